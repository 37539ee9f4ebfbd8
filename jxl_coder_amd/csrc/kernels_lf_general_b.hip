// jxl_coder_amd/csrc/kernels_lf_general_b.hip — k_lf_group_batch_general: the flight form of k_lf_group_general (kernels_lf_general.hip).
#include "kernels_lf_impl.h"

namespace jxlamd {
__global__ void __launch_bounds__(64) k_lf_group_batch_general(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map, int pool_bytes) {
  lf_group_batch_kernel<true>(Bs, As, map, pool_bytes);
}
void launch_lf_groups_batch_general(const DevBuffers *Bs, const DevAux *As, const int *map, int n, int pool_bytes, hipStream_t s) {
  static bool once = false;
  hipLaunchKernelGGL(k_lf_group_batch_general, dim3(n), dim3(64), lf_lds_bytes((const void *)k_lf_group_batch_general, &once, pool_bytes), s, Bs, As, map, pool_bytes);
}
}  // namespace jxlamd
