// jxl_coder_amd/csrc/kernels_lf.hip — HIP kernels (gfx950): Modular (MA-tree + rANS) stream decode — k_lf_group[_batch] (LF coefficients + HF metadata of a VarDCT
// frame, one wave per 2048x2048 LF group).
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_common.h"

namespace jxlamd {

__global__ void __launch_bounds__(64) k_lf_group(DevBuffers B, DevAux A) {
  __shared__ DevModScratch S;
  lf_group_body(B, A, S, (int)blockIdx.x, (int)threadIdx.x, 64, SyncBlock());
}

// batch variants: block -> (frame, local group) through a small map; the per-frame DevBuffers live in HBM
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(128))) k_lf_group_batch(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map) {
  __shared__ DevModScratch S;
  // Issue priority: this wave walks one long dependency chain (one instruction in flight at a time) next to data-parallel
  // waves with many ready instructions; without priority it waits for an issue slot each time it becomes ready, which
  // stretches the 240 ms it holds its LDS / register footprint.  It uses < 1/4 of the SIMD's issue slots at full speed.
  __builtin_amdgcn_s_setprio(3);
  // readfirstlane: the frame index is wave-uniform, so the DevBuffers fields come through scalar loads into SGPRs
  // (as with the by-value kernel argument of k_lf_group) instead of occupying ~60 VGPRs
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  lf_group_body(Bs[f], As[f], S, g, (int)threadIdx.x, 64, SyncBlock());
}
void launch_lf_groups(const DevBuffers &B, const DevAux &A, int n, hipStream_t s) { hipLaunchKernelGGL(k_lf_group, dim3(n), dim3(64), 0, s, B, A); }
void launch_lf_groups_batch(const DevBuffers *Bs, const DevAux *As, const int *map, int n, hipStream_t s) {
  // JXLAMD_LF_EXTRA_LDS: dynamic LDS bytes added to the kernel's static 50 KB — an occupancy knob (how many LF streams share a CU with
  // the LDS-using kernels of other decoder contexts), no functional effect
  static const unsigned extra = getenv("JXLAMD_LF_EXTRA_LDS") ? (unsigned)atoi(getenv("JXLAMD_LF_EXTRA_LDS")) : 0u;
  hipLaunchKernelGGL(k_lf_group_batch, dim3(n), dim3(64), extra, s, Bs, As, map);
}
}  // namespace jxlamd
