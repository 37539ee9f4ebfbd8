// jxl_coder_amd/csrc/kernels_filter.hip — HIP kernels (gfx950): Gaborish / EPF iterations, the last one fused with the XYB -> RGB -> RGBA8/16 writer.
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_common.h"

namespace jxlamd {

// ---- batched data-parallel stages: blockIdx.z = frame of the flight (per-frame dims come from its DevFrame)
// Which plane set holds the image before filter stage `stage` (0 gab, 1 epf0, 2 epf1, 3 epf2, 4 write), and does the
// frame run that stage at all?
__device__ __forceinline__ bool stage_runs(const DevFrame &F, int stage) {
  return stage == 0 ? F.gab != 0 : stage == 1 ? F.epf_iters >= 3 : stage == 2 ? F.epf_iters >= 1 : stage == 3 ? F.epf_iters >= 2 : true;
}
__device__ __forceinline__ bool stage_src_is_a(const DevFrame &F, int stage) {
  int n = 0;
  for (int s = 0; s < stage; s++) n += stage_runs(F, s) ? 1 : 0;
  return (n & 1) == 0;
}
// One instantiation per stage (0 = Gaborish, 1..3 = EPF iterations 0..2, 4 = XYB -> RGBA writer): the writer needs 14
// VGPRs and Gaborish 48, so they must not inherit the unrolled EPF's register footprint — these kernels share the
// SIMDs with resident entropy-decode waves, and their occupancy is what is left of the register file.
// The last filter stage of a frame (EPF iteration 1 or 2, or Gaborish when there is no EPF) is fused with the writer: its
// XYB value goes straight through the colour transform into the RGBA buffer (no plane store + reload, no writer launch).
__device__ __forceinline__ int last_filter_stage(const DevFrame &F) { return F.epf_iters >= 2 ? 3 : F.epf_iters == 1 ? 2 : F.gab ? 0 : -1; }
// Band decode: rows of context the stages AFTER `stage` still need around the band (EPF iteration 0 reads +-3 rows, 1: +-2, 2: +-1),
// i.e. how far beyond the band this stage has to produce output.  0 for the frame's last stage.
__device__ __forceinline__ int stage_halo_after(const DevFrame &F, int stage) {
  return (stage < 1 && F.epf_iters >= 3 ? 3 : 0) + (stage < 2 && F.epf_iters >= 1 ? 2 : 0) + (stage < 3 && F.epf_iters >= 2 ? 1 : 0);
}
template <int STAGE>
__global__ void __launch_bounds__(256) k_filter_b(const DevBuffers *Bs, const uint8_t *stat) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || !stage_runs(F, STAGE) || frame_failed(B)) return;
  const int last = last_filter_stage(F);
  if (STAGE == 4 && last >= 0) return;                       // the writer was fused into stage `last`
  const int halo = stage_halo_after(F, STAGE);
  const int y_begin = F.band_py0 - halo > 0 ? F.band_py0 - halo : 0, y_end = F.band_py1 + halo < F.height ? F.band_py1 + halo : F.height;
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = y_begin + (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= y_end) return;
  const bool a = stage_src_is_a(F, STAGE);
  float *src[3], *dst[3];
  for (int c = 0; c < 3; c++) { src[c] = a ? B.plane_a[c] : B.plane_b[c]; dst[c] = a ? B.plane_b[c] : B.plane_a[c]; }
  if (STAGE == 4) { xyb_write_pixel(B, stat, *(const DevStatic *)stat, src, B.out_bits, x, y); return; }
  float v[3];
  if (STAGE == 0) gab_value(F, src, x, y, v);
  else epf_value_p<(STAGE >= 1 && STAGE <= 3 ? STAGE - 1 : 0)>(B, F, src, x, y, v);
  if (STAGE == last) {
    // keep the filter's last multiply and the writer's first add apart (no FMA contraction across the fusion seam): the fused
    // path must give the very pixels of the stage-by-stage path (single decodes, tests/test_gpu_parity.py batch == single)
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    xyb_write_value(B, stat, *(const DevStatic *)stat, v[0], v[1], v[2], B.out_bits, x, y);
  }
  else for (int c = 0; c < 3; c++) dst[c][(size_t)y * (size_t)F.pw + (size_t)x] = v[c];
}

void launch_filters_batch(const DevBuffers *Bs, const uint8_t *stat, int nframes, int max_w, int max_h, int stage_mask, hipStream_t s) {
  dim3 grid((max_w + 63) / 64, (max_h + 3) / 4, nframes);
  if (stage_mask & 1) hipLaunchKernelGGL(k_filter_b<0>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 2) hipLaunchKernelGGL(k_filter_b<1>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 4) hipLaunchKernelGGL(k_filter_b<2>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 8) hipLaunchKernelGGL(k_filter_b<3>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 16) hipLaunchKernelGGL(k_filter_b<4>, grid, dim3(256), 0, s, Bs, stat);
}

}  // namespace jxlamd
