// jxl_coder_amd/csrc/kernels_lf_general.hip — k_lf_group_general: the LF-group kernel WITH the general lock-step loops (any MA property, any predictor;
// 181 VGPRs), for single decodes (the latency path) and for contexts whose frames need it (kernels_lf.hip, decoder.hip).
#include "kernels_lf_impl.h"

namespace jxlamd {
__global__ void __launch_bounds__(64) k_lf_group_general(DevBuffers B, DevAux A, int pool_bytes) { lf_group_kernel<true>(B, A, (int)blockIdx.x, pool_bytes); }
void launch_lf_groups_general(const DevBuffers &B, const DevAux &A, int n, int pool_bytes, hipStream_t s) {
  static bool once = false;
  hipLaunchKernelGGL(k_lf_group_general, dim3(n), dim3(64), lf_lds_bytes((const void *)k_lf_group_general, &once, pool_bytes), s, B, A, pool_bytes);
}
}  // namespace jxlamd
