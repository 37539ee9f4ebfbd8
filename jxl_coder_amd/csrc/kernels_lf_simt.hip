// jxl_coder_amd/csrc/kernels_lf_simt.hip — HIP kernels (gfx950): lane-per-stream LfGroup decode for flights (dev_lf_simt.h) and the
// data-parallel LF epilogue (dequantisation, block-context buckets, CfL / sharpness maps) as its own launch.
#include "kernels_common.h"
#include "dev_lf_simt.h"

namespace jxlamd {

// 64 LfGroup sections per wavefront, one workgroup (= one wavefront) per CU: 128 KB of lane-private LDS, no register ceiling.
__global__ void __launch_bounds__(64) k_lf_group_simt(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map, int total,
                                                      LfSimtWave *waves, DevModScratch *scratch) {
  __shared__ LfSimtLds L;
  const int lane = (int)threadIdx.x;
  __builtin_amdgcn_s_setprio(3);             // a long dependency chain next to data-parallel kernels (see k_lf_group_batch)
  L.divlut[lane] = (1u << 24) / (uint32_t)(lane + 1);
  __syncthreads();
  const int i = (int)blockIdx.x * 64 + lane;
  if (i >= total) return;
  const int f = map[2 * i], g = map[2 * i + 1];
  const DevBuffers &B = Bs[f];
  const uint32_t e = lf_group_lane(B, As[f], scratch[i], waves[blockIdx.x], L, g, lane);
  if (e) atomicOr(B.err, e);
}

__global__ void __launch_bounds__(256) k_lf_epilogue_b(const DevBuffers *__restrict__ Bs, const int *__restrict__ map) {
  const int f = map[2 * blockIdx.x], g = map[2 * blockIdx.x + 1];
  const DevBuffers &B = Bs[f];
  if (frame_failed(B)) return;
  lf_group_epilogue(B, g, (int)threadIdx.x, 256);
}

size_t lf_simt_wave_bytes() { return sizeof(LfSimtWave); }
size_t lf_simt_scratch_bytes() { return sizeof(DevModScratch); }
void launch_lf_groups_simt(const DevBuffers *Bs, const DevAux *As, const int *map, int n, void *waves, void *scratch, hipStream_t s) {
  hipLaunchKernelGGL(k_lf_group_simt, dim3((n + 63) / 64), dim3(64), 0, s, Bs, As, map, n, (LfSimtWave *)waves, (DevModScratch *)scratch);
  hipLaunchKernelGGL(k_lf_epilogue_b, dim3(n), dim3(256), 0, s, Bs, map);
}

}  // namespace jxlamd
