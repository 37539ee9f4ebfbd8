// jxl_coder_amd/csrc/kernels_lf.hip — HIP kernels (gfx950): Modular (MA-tree + rANS) stream decode — k_lf_group[_batch] (LF coefficients + HF metadata of a VarDCT
// frame, one wave per 2048x2048 LF group).
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include <stddef.h>
#include "kernels_common.h"

namespace jxlamd {

// The workgroup's DevModScratch lives in DYNAMIC LDS: offsetof(pool) + the pool bytes of this launch (kModPoolMin .. kModPoolBytes).
// An LF stream holds its LDS for ~100 ms; 50.8 KB per stream (the full pool) lets three streams into a CU and leaves 8 KB for everybody
// else, 32.8 KB (12 KB pool: all that libjxl's streaming encoder needs) four with 29 KB to spare.
// Two builds of each kernel.  The lean one (default) carries the specialised lock-step loops only — weighted-predictor threshold trees
// (libjxl's LF coefficients), uniform-leaf and y / x / N / W channels (its HF metadata), the serial walker — in 123 VGPRs; a channel that
// needs a general lock-step loop ends its frame with kErrNeedGeneral and the host runs the *_general build (184 VGPRs) from then on.
template <bool kGeneral>
__device__ __forceinline__ void lf_group_kernel(const DevBuffers &B, const DevAux &A, int g, int pool_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lf_smem[];
  lf_group_body<true, kGeneral>(B, A, *(DevModScratch *)lf_smem, g, (int)threadIdx.x, 64, SyncBlock(), pool_bytes);
}
__global__ void __launch_bounds__(64) k_lf_group(DevBuffers B, DevAux A, int pool_bytes) { lf_group_kernel<false>(B, A, (int)blockIdx.x, pool_bytes); }
__global__ void __launch_bounds__(64) k_lf_group_general(DevBuffers B, DevAux A, int pool_bytes) { lf_group_kernel<true>(B, A, (int)blockIdx.x, pool_bytes); }

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
__global__ void __launch_bounds__(64) k_lf_group_batch(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map, int pool_bytes) {
  lf_group_batch_kernel<false>(Bs, As, map, pool_bytes);
}
__global__ void __launch_bounds__(64) k_lf_group_batch_general(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map, int pool_bytes) {
  lf_group_batch_kernel<true>(Bs, As, map, pool_bytes);
}
static size_t lf_lds_bytes(int pool_bytes) {
  static const bool once = [] {           // dynamic LDS beyond the default opt-in limit
    const int most = (int)(offsetof(DevModScratch, pool) + kModPoolBytes);
    (void)hipFuncSetAttribute((const void *)k_lf_group, hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute((const void *)k_lf_group_batch, hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute((const void *)k_lf_group_general, hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute((const void *)k_lf_group_batch_general, hipFuncAttributeMaxDynamicSharedMemorySize, most);
    return true;
  }();
  (void)once;
  return offsetof(DevModScratch, pool) + (size_t)pool_bytes;
}
int lf_pool_clamp(uint32_t wanted) {        // the pool the next launch gets for what the streams of the last one reported
#ifdef JXL_LF_POOL_FORCE
  wanted = JXL_LF_POOL_FORCE;               // experiment builds (tools/build_variant.sh)
#endif
  const int w = (int)((wanted + 2047u) & ~2047u);
  return w < kModPoolMin ? kModPoolMin : w > kModPoolBytes ? kModPoolBytes : w;
}
void launch_lf_groups(const DevBuffers &B, const DevAux &A, int n, int pool_bytes, bool general, hipStream_t s) {
  if (general) hipLaunchKernelGGL(k_lf_group_general, dim3(n), dim3(64), lf_lds_bytes(pool_bytes), s, B, A, pool_bytes);
  else hipLaunchKernelGGL(k_lf_group, dim3(n), dim3(64), lf_lds_bytes(pool_bytes), s, B, A, pool_bytes);
}
void launch_lf_groups_batch(const DevBuffers *Bs, const DevAux *As, const int *map, int n, int pool_bytes, bool general, hipStream_t s) {
  if (general) hipLaunchKernelGGL(k_lf_group_batch_general, dim3(n), dim3(64), lf_lds_bytes(pool_bytes), s, Bs, As, map, pool_bytes);
  else hipLaunchKernelGGL(k_lf_group_batch, dim3(n), dim3(64), lf_lds_bytes(pool_bytes), s, Bs, As, map, pool_bytes);
}
}  // namespace jxlamd
