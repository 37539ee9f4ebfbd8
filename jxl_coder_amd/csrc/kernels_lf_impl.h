// jxl_coder_amd/csrc/kernels_lf_impl.h — the LF-group kernel bodies shared by the translation units that instantiate them (kernels_lf.hip: the lean
// builds; kernels_lf_general.hip / kernels_lf_general_b.hip: the builds with the general lock-step loops — one kernel per file so that the three
// compile side by side: each takes minutes).
#pragma once
#include <stddef.h>
#include "kernels_common.h"

namespace jxlamd {

// The workgroup's DevModScratch lives in DYNAMIC LDS: offsetof(pool) + the pool bytes of this launch (kModPoolMin .. kModPoolBytes).
template <bool kGeneral>
__device__ __forceinline__ void lf_group_kernel(const DevBuffers &B, const DevAux &A, int g, int pool_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lf_smem[];
  DevModScratch &S = *(DevModScratch *)lf_smem;
  S.ch = (DevChanOut *)(lf_smem + offsetof(DevModScratch, pool) + pool_bytes);      // every lane stores the same value; the body's first barrier orders it
  lf_group_body<true, kGeneral>(B, A, S, g, (int)threadIdx.x, 64, SyncBlock(), pool_bytes);
}
// batch variants: block -> (frame, local group) through a small map; the per-frame DevBuffers live in HBM
template <bool kGeneral>
__device__ __forceinline__ void lf_group_batch_kernel(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map, int pool_bytes) {
  // Issue priority: this wave walks one long dependency chain (one instruction in flight at a time) next to data-parallel
  // waves with many ready instructions; without priority it waits for an issue slot each time it becomes ready, which
  // stretches the time it holds its LDS / register footprint.
  __builtin_amdgcn_s_setprio(3);
  // readfirstlane: the frame index is wave-uniform, so the DevBuffers fields come through scalar loads into SGPRs
  // (as with the by-value kernel argument of k_lf_group) instead of occupying ~60 VGPRs
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  lf_group_kernel<kGeneral>(Bs[f], As[f], g, pool_bytes);
}
// dynamic LDS of a launch; `kernel`: opt in once to more than the default limit
inline size_t lf_lds_bytes(const void *kernel, bool *once, int pool_bytes) {
  if (!*once) { (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(offsetof(DevModScratch, pool) + kModPoolBytes + kLfMaxCh * sizeof(DevChanOut))); *once = true; }
  return offsetof(DevModScratch, pool) + (size_t)pool_bytes + kLfMaxCh * sizeof(DevChanOut);
}
// the general builds' launchers (their own translation units)
void launch_lf_groups_general(const DevBuffers &B, const DevAux &A, int n, int pool_bytes, hipStream_t s);
void launch_lf_groups_batch_general(const DevBuffers *Bs, const DevAux *As, const int *map, int n, int pool_bytes, hipStream_t s);
}  // namespace jxlamd
