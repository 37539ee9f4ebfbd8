// jxl_coder_amd/csrc/kernels_lf_multi.hip — k_lf_coef_multi (gfx950): the LF coefficients of the LfGroup streams of a flight, sixteen streams per
// wavefront, a quad of lanes per stream (dev_lf_multi.h).  Runs before k_lf_group_batch, which resumes each section at its HF metadata.
#include "kernels_common.h"
#include "dev_lf_multi.h"

namespace jxlamd {

__global__ void __launch_bounds__(64) k_lf_coef_multi(const DevBuffers *__restrict__ Bs, const int *__restrict__ map, int nstreams) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lfm_smem[];
  __builtin_amdgcn_s_setprio(3);       // one dependency chain per step next to data-parallel waves with many ready instructions (as k_lf_group_batch)
  lf_coef_multi_body(Bs, map, nstreams, lfm_smem);
}

bool launch_lf_coef_multi(const DevBuffers *Bs, const int *map, int nstreams, hipStream_t s) {
  static int state = 0;                 // 0 untried, 1 available, -1 not available (LDS per workgroup)
  if (state == 0) {
    state = hipFuncSetAttribute((const void *)k_lf_coef_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLfmLdsBytes) == hipSuccess ? 1 : -1;
    if (state < 0) (void)hipGetLastError();
  }
  if (state < 0 || nstreams <= 0) return false;
  hipLaunchKernelGGL(k_lf_coef_multi, dim3((unsigned)((nstreams + kLfmStreams - 1) / kLfmStreams)), dim3(64), kLfmLdsBytes, s, Bs, map, nstreams);
  return true;
}

}  // namespace jxlamd
