// jxl_coder_amd/csrc/kernels_compose.hip — HIP kernels (gfx950) of the composition stages (dev_compose.h): Modular planes -> f32 planes,
// patch blending, copy into a reference slot, the stand-alone writer of composed frames.  All of them are plain streaming kernels over
// pixels (HBM-bound, a few bytes per pixel); frames that need them are rare next to the flights of ordinary frames and run one by one.
#include "kernels_common.h"
#include "dev_compose.h"

namespace jxlamd {

__global__ void __launch_bounds__(256) k_mod_to_planes(DevBuffers B) {
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height || frame_failed(B)) return;
  mod_to_planes_pixel(B, F, x, y);
}
// grid (placements, ceil(largest patch / 256)): one placement per blockIdx.x, its pixels over the threads
__global__ void __launch_bounds__(256) k_patch_blend(DevBuffers B) {
  const DevFrame &F = frame_of(B);
  if (frame_failed(B)) return;
  const DevPatch P = ((const DevPatch *)(B.tables + F.patch_off))[blockIdx.x];
  const int n = P.w * P.h;          // <= 2^31: both bounded by the reference frame's size
  for (int item = (int)(blockIdx.y * 256 + threadIdx.x); item < n; item += (int)(gridDim.y * 256)) patch_blend_sample(B, F, P, item);
}
__global__ void __launch_bounds__(256) k_splines(DevBuffers B) {
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height || frame_failed(B)) return;
  spline_pixel(B, F, x, y);
}
__global__ void __launch_bounds__(256) k_save_ref(DevBuffers B, float *d0, float *d1, float *d2) {
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height) return;
  float *dst[3] = {d0, d1, d2};
  save_ref_pixel(B, F, dst, x, y);
}
__global__ void __launch_bounds__(256) k_compose_write(DevBuffers B, const uint8_t *stat) {
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height || frame_failed(B)) return;
  if (F.is_modular && !F.xyb_modular) { plain_write_pixel(B, stat, B.out_bits, x, y); return; }
  float *src[3];
  for (int c = 0; c < 3; c++) src[c] = compose_final_is_a(F) ? B.plane_a[c] : B.plane_b[c];
  if (F.not_xyb) { const size_t po = (size_t)y * (size_t)F.pw + (size_t)x; plain_write_value(B, stat, *(const DevStatic *)stat, src[0][po], src[1][po], src[2][po], B.out_bits, x, y); return; }
  xyb_write_pixel(B, stat, *(const DevStatic *)stat, src, B.out_bits, x, y);
}
// chroma-subsampled YCbCr frames: the three channels at full resolution into the second plane set (blockIdx.z = channel)
__global__ void __launch_bounds__(256) k_chroma_upsample(DevBuffers B) {
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height || frame_failed(B)) return;
  chroma_upsample_pixel(B, F, (int)blockIdx.z, x, y);
}
// noise synthesis: the random planes (one work-item per generator: 8 per group), then one work-item per pixel
__global__ void __launch_bounds__(64) k_noise_gen(DevBuffers B) {
  const DevFrame &F = frame_of(B);
  const int item = (int)(blockIdx.x * 64 + threadIdx.x);
  const NoiseGeom G = noise_geom(B, F);
  if (item >= G.xtiles * G.ytiles * 8 || frame_failed(B)) return;
  noise_gen_lane(B, F, item >> 3, item & 7);
}
__global__ void __launch_bounds__(256) k_noise_add(DevBuffers B) {
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= (F.upsampling > 1 ? F.full_w : F.width) || y >= (F.upsampling > 1 ? F.full_h : F.height) || frame_failed(B)) return;
  noise_add_pixel(B, F, x, y);
}
void launch_noise(const DevBuffers &B, int w, int h, hipStream_t s) {      // w x h: the resolution the noise is drawn at (the upsampled one for an upsampled frame)
  const int tiles = ((w + 255) / 256) * ((h + 255) / 256);
  hipLaunchKernelGGL(k_noise_gen, dim3((unsigned)(tiles * 8 + 63) / 64), dim3(64), 0, s, B);
  hipLaunchKernelGGL(k_noise_add, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, B);
}
// frames laid over a canvas (animations): one work-item per canvas pixel
__global__ void __launch_bounds__(256) k_blend_canvas(DevBuffers B, const uint8_t *stat) {
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.canvas_w || y >= F.canvas_h || frame_failed(B)) return;
  blend_canvas_pixel(B, stat, B.out_bits, x, y);
}
void launch_blend_canvas(const DevBuffers &B, const uint8_t *stat, int canvas_w, int canvas_h, hipStream_t s) {
  hipLaunchKernelGGL(k_blend_canvas, dim3((canvas_w + 63) / 64, (canvas_h + 3) / 4), dim3(256), 0, s, B, stat);
}
void launch_chroma_upsample(const DevBuffers &B, int w, int h, hipStream_t s) { hipLaunchKernelGGL(k_chroma_upsample, dim3((w + 63) / 64, (h + 3) / 4, 3), dim3(256), 0, s, B); }

__global__ void __launch_bounds__(256) k_upsample(DevBuffers B, const uint8_t *stat) {
  const DevFrame &F = frame_of(B);
  const int X = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), Y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (X >= F.full_w || Y >= F.full_h || frame_failed(B)) return;
  upsample_pixel(B, F, stat, X, Y);
}
__global__ void __launch_bounds__(256) k_upsampled_write(DevBuffers B, const uint8_t *stat) {
  const DevFrame &F = frame_of(B);
  const int X = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), Y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (X >= F.full_w || Y >= F.full_h || frame_failed(B)) return;
  upsampled_write_pixel(B, stat, B.out_bits, X, Y);
}
__global__ void __launch_bounds__(256) k_upsample_alpha(DevBuffers B, const uint8_t *stat) {
  const DevFrame &F = frame_of(B);
  const int X = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), Y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (X >= F.full_w || Y >= F.full_h || frame_failed(B)) return;
  upsample_alpha_pixel(B, F, stat, X, Y);
}
void launch_upsample_alpha(const DevBuffers &B, const uint8_t *stat, int full_w, int full_h, hipStream_t s) {
  hipLaunchKernelGGL(k_upsample_alpha, dim3((full_w + 63) / 64, (full_h + 3) / 4), dim3(256), 0, s, B, stat);
}
void launch_upsample_and_write(const DevBuffers &B, const uint8_t *stat, int full_w, int full_h, bool noise, bool write, hipStream_t s) {
  const dim3 g((full_w + 63) / 64, (full_h + 3) / 4);
  hipLaunchKernelGGL(k_upsample, g, dim3(256), 0, s, B, stat);
  if (noise) launch_noise(B, full_w, full_h, s);          // libjxl's stage order: Upsampling, Noise, colour transform (, Blending)
  if (write) hipLaunchKernelGGL(k_upsampled_write, g, dim3(256), 0, s, B, stat);      // (a blended frame: the blend kernel reads the upsampled planes and writes)
}

void launch_mod_to_planes(const DevBuffers &B, int w, int h, hipStream_t s) { hipLaunchKernelGGL(k_mod_to_planes, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, B); }
void launch_splines(const DevBuffers &B, int w, int h, hipStream_t s) {
  hipLaunchKernelGGL(k_splines, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, B);
}
void launch_patch_blend(const DevBuffers &B, int num_patches, size_t max_px, hipStream_t s) {
  if (num_patches <= 0) return;
  // placements beyond 2^22 are refused by the parser; large patches loop
  hipLaunchKernelGGL(k_patch_blend, dim3((unsigned)num_patches, (unsigned)std::min<size_t>((max_px + 255) / 256, 4096)), dim3(256), 0, s, B);
}
void launch_save_ref(const DevBuffers &B, int w, int h, float *dst, hipStream_t s) {
  const size_t n = (size_t)w * (size_t)h;
  hipLaunchKernelGGL(k_save_ref, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, B, dst, dst + n, dst + 2 * n);
}
void launch_compose_write(const DevBuffers &B, const uint8_t *stat, int w, int h, hipStream_t s) {
  hipLaunchKernelGGL(k_compose_write, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, B, stat);
}

}  // namespace jxlamd
