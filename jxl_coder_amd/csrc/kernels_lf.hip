// jxl_coder_amd/csrc/kernels_lf.hip — HIP kernels (gfx950): Modular (MA-tree + rANS) stream decode — k_lf_group[_batch] (LF coefficients + HF metadata of a VarDCT
// frame, one wave per 2048x2048 LF group).
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_lf_impl.h"

namespace jxlamd {

// An LF stream holds its LDS for ~100 ms; round 6: 19.9 KB of fixed scratch + the table pool + 1 KB of descriptors — 39.3 KB with the 18.4 KB pool the bench's
// frames need (four streams per CU), 33.1 KB with the smallest pool.
// Two builds of each kernel.  The lean one (here, the default of flights) carries the specialised lock-step loops only — weighted-predictor
// threshold trees (libjxl's LF coefficients), uniform-leaf channels and y / x / N / W channels with predictors 0..5 (its HF metadata), the serial
// walker — in 125 VGPRs; a channel that needs a general lock-step loop ends its frame with kErrNeedGeneral and the host runs the *_general build
// (181 VGPRs; kernels_lf_general*.hip) from then on.
__global__ void __launch_bounds__(64) k_lf_group(DevBuffers B, DevAux A, int pool_bytes) { lf_group_kernel<false>(B, A, (int)blockIdx.x, pool_bytes); }
__global__ void __launch_bounds__(64) k_lf_group_batch(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map, int pool_bytes) {
  lf_group_batch_kernel<false>(Bs, As, map, pool_bytes);
}
int lf_pool_clamp(uint32_t wanted) {        // the pool the next launch gets for what the streams of the last one reported
#ifdef JXL_LF_POOL_FORCE
  wanted = JXL_LF_POOL_FORCE;               // experiment builds (tools/build_variant.sh)
#endif
  const int w = (int)((wanted + 511u) & ~511u);          // (round 6: steps of 512 bytes, not 2 048 — every LF workgroup of the process carries the rounding)
  return w < kModPoolMin ? kModPoolMin : w > kModPoolBytes ? kModPoolBytes : w;
}
void launch_lf_groups(const DevBuffers &B, const DevAux &A, int n, int pool_bytes, bool general, hipStream_t s) {
  static bool once = false;
  if (general) launch_lf_groups_general(B, A, n, pool_bytes, s);
  else hipLaunchKernelGGL(k_lf_group, dim3(n), dim3(64), lf_lds_bytes((const void *)k_lf_group, &once, pool_bytes), s, B, A, pool_bytes);
}
void launch_lf_groups_batch(const DevBuffers *Bs, const DevAux *As, const int *map, int n, int pool_bytes, bool general, hipStream_t s) {
  static bool once = false;
  if (general) launch_lf_groups_batch_general(Bs, As, map, n, pool_bytes, s);
  else hipLaunchKernelGGL(k_lf_group_batch, dim3(n), dim3(64), lf_lds_bytes((const void *)k_lf_group_batch, &once, pool_bytes), s, Bs, As, map, pool_bytes);
}
}  // namespace jxlamd
