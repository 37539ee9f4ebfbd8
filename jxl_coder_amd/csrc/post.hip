// jxl_coder_amd/csrc/post.hip — the reference's first-party post-decode stages as HIP kernels (gfx950), on buffers that
// stay in HBM after the decode:
//   A11 ReformatColorConfig (jxlcoder/src/main/cpp/ReformatBitmap.cpp:46-263): premultiply (imagebit/RGBAlpha.cpp:67-117),
//       u16 -> f16 (RgbaU16toHF.cpp:42-144), u8 -> f16 (Rgba8ToF16.cpp:44-138), u16 -> u8 (Rgba16.cpp:32-68),
//       -> RGB565 (Rgb565.cpp:99-160), -> RGBA1010102 (Rgb1010102.cpp:177-249)
//   A10 applyColorMatrix / applyColorMatrix16Bit (colorspaces/ColorMatrix.cpp:35-219) with the Rec.2408 tone mapper
//       (colorspaces/Rec2408ToneMapper.cpp:80-100), including its stuck-pointer behaviour on zero-luma pixels.
// All of them are streaming kernels (HBM-bound): one work-item per pixel, 4-16 bytes in, 2-8 bytes out.
// Integer stages are bit-exact with the reference; the float stage keeps the reference's operation order with FMA
// contraction disabled, so that it differs from the reference only through the host-built LUTs / matrix.
#include "dev_post.h"

namespace jxlamd {

__global__ void __launch_bounds__(256) k_post_premul8(uint8_t *px, uint32_t stride, uint32_t w, uint32_t h) {
  const uint32_t x = blockIdx.y * 256 + threadIdx.x, y = blockIdx.x;      // rows in grid.x (grid.y is capped at 65 535)
  if (x >= w) return;
  uint32_t *p = (uint32_t *)(px + (size_t)y * stride) + x;
  const uint32_t v = *p, a = v >> 24;
  const uint32_t r = ((v & 0xff) * a) / 255u, g = (((v >> 8) & 0xff) * a) / 255u, b = (((v >> 16) & 0xff) * a) / 255u;
  *p = r | (g << 8) | (b << 16) | (a << 24);
}
__global__ void __launch_bounds__(256) k_post_premul16(uint8_t *px, uint32_t stride, uint32_t w, uint32_t h, uint32_t maxv) {
  const uint32_t x = blockIdx.y * 256 + threadIdx.x, y = blockIdx.x;      // rows in grid.x (grid.y is capped at 65 535)
  if (x >= w) return;
  ushort4 *p = (ushort4 *)(px + (size_t)y * stride) + x;
  ushort4 v = *p;
  const uint32_t a = v.w;
  v.x = (uint16_t)(((uint32_t)v.x * a) / maxv); v.y = (uint16_t)(((uint32_t)v.y * a) / maxv); v.z = (uint16_t)(((uint32_t)v.z * a) / maxv);
  *p = v;
}

template <int KIND>
__global__ void __launch_bounds__(256) k_post_convert(const uint8_t *src, uint32_t src_stride, uint8_t *dst, uint32_t dst_stride, uint32_t w, uint32_t h,
                                                      uint32_t depth, int attenuate) {
  const uint32_t x = blockIdx.y * 256 + threadIdx.x, y = blockIdx.x;      // rows in grid.x (grid.y is capped at 65 535)
  if (x >= w) return;
  const uint8_t *srow = src + (size_t)y * src_stride;
  uint32_t r, g, b, a;
  if (post_src16<KIND>()) { const ushort4 v = ((const ushort4 *)srow)[x]; r = v.x; g = v.y; b = v.z; a = v.w; }
  else { const uint32_t v = ((const uint32_t *)srow)[x]; r = v & 0xff; g = (v >> 8) & 0xff; b = (v >> 16) & 0xff; a = v >> 24; }
  post_convert_store<KIND>(dst + (size_t)y * dst_stride, x, r, g, b, a, depth, attenuate);
}

// the first pixel of the row whose linear luma is exactly 0 — the reference's tone-mapping loop never advances past it, so the pixels
// from there on stay un-mapped (colorspaces/Rec2408ToneMapper.cpp:80-100).  All work-items of the workgroup (one row) call it.
template <bool kU16>
__device__ __forceinline__ uint32_t post_row_first_zero(const uint8_t *row, uint32_t w, const ColorMatrixDev &P, uint32_t *first_zero) {
#pragma clang fp contract(off)
  if (threadIdx.x == 0) *first_zero = w;
  __syncthreads();
  const uint32_t cap = kU16 ? P.index_max : 255u;
  if (P.tone_map) {
    uint32_t mine = w;
    for (uint32_t x = threadIdx.x; x < w; x += 256) {
      uint32_t r, g, b;
      if (kU16) { const ushort4 v = ((const ushort4 *)row)[x]; r = v.x; g = v.y; b = v.z; }
      else { const uint32_t v = ((const uint32_t *)row)[x]; r = v & 0xff; g = (v >> 8) & 0xff; b = (v >> 16) & 0xff; }
      const float fr = P.lin_lut[r < cap ? r : cap], fg = P.lin_lut[g < cap ? g : cap], fb = P.lin_lut[b < cap ? b : cap];
      const float y = 0.2627f * fr + 0.6780f * fg + 0.0593f * fb;
      if (y == 0.0f) { mine = x; break; }      // x ascends within a work-item: the first hit is its minimum
    }
    if (mine < w) atomicMin(first_zero, mine);
  }
  __syncthreads();
  return *first_zero;
}
// One workgroup per row, in place.
template <bool kU16>
__global__ void __launch_bounds__(256) k_post_color_matrix(uint8_t *px, uint32_t stride, uint32_t w, ColorMatrixDev P) {
  __shared__ uint32_t first_zero;
  uint8_t *row = px + (size_t)blockIdx.x * stride;
  const uint32_t fz = post_row_first_zero<kU16>(row, w, P, &first_zero);
  for (uint32_t x = threadIdx.x; x < w; x += 256) {
    uint32_t r, g, b, a;
    if (kU16) { const ushort4 v = ((const ushort4 *)row)[x]; r = v.x; g = v.y; b = v.z; a = v.w; }
    else { const uint32_t v = ((const uint32_t *)row)[x]; r = v & 0xff; g = (v >> 8) & 0xff; b = (v >> 16) & 0xff; a = v >> 24; }
    post_matrix_px<kU16>(P, P.tone_map && x < fz, r, g, b);
    if (kU16) { ushort4 o; o.x = (uint16_t)r; o.y = (uint16_t)g; o.z = (uint16_t)b; o.w = (uint16_t)a; ((ushort4 *)row)[x] = o; }
    else ((uint32_t *)row)[x] = r | (g << 8) | (b << 16) | (a << 24);
  }
}
// A10 + A11 in ONE pass over the decoded buffer: colour matrix / tone map (when `matrix`), premultiply (when `premul`),
// conversion into the Bitmap's format — the values jxlamd_color_matrix followed by jxlamd_reformat produce, bit for bit, without the two
// in-place passes over the RGBA buffer in between (a 4K RGBA16 frame: 66 MB read + 66 MB written, each).  src is read-only.
template <int KIND>
__global__ void __launch_bounds__(256) k_post_fused(const uint8_t *src, uint32_t src_stride, uint8_t *dst, uint32_t dst_stride, uint32_t w, ColorMatrixDev P, int matrix,
                                                    int premul, uint32_t depth, int attenuate) {
  constexpr bool kU16 = post_src16<KIND>();
  __shared__ uint32_t first_zero;
  const uint8_t *row = src + (size_t)blockIdx.x * src_stride;
  uint8_t *drow = dst + (size_t)blockIdx.x * dst_stride;
  uint32_t fz = w;
  if (matrix) fz = post_row_first_zero<kU16>(row, w, P, &first_zero);
  const uint32_t maxv = (1u << depth) - 1u;
  for (uint32_t x = threadIdx.x; x < w; x += 256) {
    uint32_t r, g, b, a;
    if (kU16) { const ushort4 v = ((const ushort4 *)row)[x]; r = v.x; g = v.y; b = v.z; a = v.w; }
    else { const uint32_t v = ((const uint32_t *)row)[x]; r = v & 0xff; g = (v >> 8) & 0xff; b = (v >> 16) & 0xff; a = v >> 24; }
    if (matrix) post_matrix_px<kU16>(P, P.tone_map && x < fz, r, g, b);
    if (premul) {                                        // k_post_premul8 / k_post_premul16
      if (kU16) { r = (uint16_t)((r * a) / maxv); g = (uint16_t)((g * a) / maxv); b = (uint16_t)((b * a) / maxv); }
      else { r = (r * a) / 255u; g = (g * a) / 255u; b = (b * a) / 255u; }
    }
    post_convert_store<KIND>(drow, x, r, g, b, a, depth, attenuate);
  }
}

void launch_post_premultiply(void *px, uint32_t stride, uint32_t w, uint32_t h, bool is_u16, uint32_t depth, hipStream_t s) {
  dim3 grid(h, (w + 255) / 256);
  if (is_u16) hipLaunchKernelGGL(k_post_premul16, grid, dim3(256), 0, s, (uint8_t *)px, stride, w, h, (1u << depth) - 1u);
  else hipLaunchKernelGGL(k_post_premul8, grid, dim3(256), 0, s, (uint8_t *)px, stride, w, h);
}

void launch_post_convert(PostKind kind, const void *src, uint32_t ss, void *dst, uint32_t ds, uint32_t w, uint32_t h, uint32_t depth, bool att, hipStream_t s) {
  dim3 grid(h, (w + 255) / 256), block(256);
  const uint8_t *a = (const uint8_t *)src; uint8_t *b = (uint8_t *)dst; const int at = att ? 1 : 0;
  switch (kind) {
    case kPostU16ToF16: hipLaunchKernelGGL(k_post_convert<kPostU16ToF16>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostRgba8ToF16: hipLaunchKernelGGL(k_post_convert<kPostRgba8ToF16>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostRgba16To8: hipLaunchKernelGGL(k_post_convert<kPostRgba16To8>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostRgba8To565: hipLaunchKernelGGL(k_post_convert<kPostRgba8To565>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostRgba16To565: hipLaunchKernelGGL(k_post_convert<kPostRgba16To565>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostRgba8To1010102: hipLaunchKernelGGL(k_post_convert<kPostRgba8To1010102>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostRgba16To1010102: hipLaunchKernelGGL(k_post_convert<kPostRgba16To1010102>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostCopy8: hipLaunchKernelGGL(k_post_convert<kPostCopy8>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
    case kPostCopy16: hipLaunchKernelGGL(k_post_convert<kPostCopy16>, grid, block, 0, s, a, ss, b, ds, w, h, depth, at); break;
  }
}

void launch_post_fused(PostKind kind, const void *src, uint32_t ss, void *dst, uint32_t ds, uint32_t w, uint32_t h, const ColorMatrixDev *P, bool premul, uint32_t depth,
                       bool att, hipStream_t s) {
  const uint8_t *a = (const uint8_t *)src; uint8_t *b = (uint8_t *)dst;
  ColorMatrixDev Z = {}; const ColorMatrixDev &M = P ? *P : Z; const int mt = P ? 1 : 0, pm = premul ? 1 : 0, at = att ? 1 : 0;
  switch (kind) {
    case kPostU16ToF16: hipLaunchKernelGGL(k_post_fused<kPostU16ToF16>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostRgba8ToF16: hipLaunchKernelGGL(k_post_fused<kPostRgba8ToF16>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostRgba16To8: hipLaunchKernelGGL(k_post_fused<kPostRgba16To8>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostRgba8To565: hipLaunchKernelGGL(k_post_fused<kPostRgba8To565>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostRgba16To565: hipLaunchKernelGGL(k_post_fused<kPostRgba16To565>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostRgba8To1010102: hipLaunchKernelGGL(k_post_fused<kPostRgba8To1010102>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostRgba16To1010102: hipLaunchKernelGGL(k_post_fused<kPostRgba16To1010102>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostCopy8: hipLaunchKernelGGL(k_post_fused<kPostCopy8>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
    case kPostCopy16: hipLaunchKernelGGL(k_post_fused<kPostCopy16>, dim3(h), dim3(256), 0, s, a, ss, b, ds, w, M, mt, pm, depth, at); break;
  }
}

void launch_post_color_matrix(void *px, uint32_t stride, uint32_t w, uint32_t h, bool is_u16, const ColorMatrixDev &P, hipStream_t s) {
  if (is_u16) hipLaunchKernelGGL(k_post_color_matrix<true>, dim3(h), dim3(256), 0, s, (uint8_t *)px, stride, w, P);
  else hipLaunchKernelGGL(k_post_color_matrix<false>, dim3(h), dim3(256), 0, s, (uint8_t *)px, stride, w, P);
}

// ---- A8: ICC transform through a 3-D lattice
__device__ __forceinline__ void icc_lut_sample(const uint16_t *__restrict__ lut, int n, float r, float g, float b, float out[3]) {      // r, g, b in [0, 1]
  const float s = (float)(n - 1);
  const float fr = r * s, fg = g * s, fb = b * s;
  int ir = (int)fr, ig = (int)fg, ib = (int)fb;
  ir = ir > n - 2 ? n - 2 : ir; ig = ig > n - 2 ? n - 2 : ig; ib = ib > n - 2 ? n - 2 : ib;
  const float tr = fr - (float)ir, tg = fg - (float)ig, tb = fb - (float)ib;
  const size_t sg = (size_t)n * 3, sb = (size_t)n * n * 3;
  const uint16_t *p = lut + (size_t)ib * sb + (size_t)ig * sg + (size_t)ir * 3;
  for (int c = 0; c < 3; c++) {
    const float c000 = p[c], c100 = p[3 + c], c010 = p[sg + c], c110 = p[sg + 3 + c];
    const float c001 = p[sb + c], c101 = p[sb + 3 + c], c011 = p[sb + sg + c], c111 = p[sb + sg + 3 + c];
    const float a0 = c000 + (c100 - c000) * tr, a1 = c010 + (c110 - c010) * tr, a2 = c001 + (c101 - c001) * tr, a3 = c011 + (c111 - c011) * tr;
    const float b0 = a0 + (a1 - a0) * tg, b1 = a2 + (a3 - a2) * tg;
    out[c] = b0 + (b1 - b0) * tb;                              // 0 .. 65535
  }
}
template <bool kU16>
__global__ void __launch_bounds__(256) k_post_icc_lut(void *px, uint32_t stride, uint32_t w, uint32_t h, const uint16_t *__restrict__ lut, int n) {
  const uint32_t x = blockIdx.y * 256 + threadIdx.x, y = blockIdx.x;      // rows in grid.x (grid.y is capped at 65 535)
  if (x >= w || y >= h) return;
  float v[3];
  if (kU16) {
    uint16_t *p = (uint16_t *)((uint8_t *)px + (size_t)y * stride) + (size_t)x * 4;
    const float a = (float)p[3];
    if (a <= 0.0f) return;                                   // premultiplied by zero alpha: nothing to transform
    const float ia = 1.0f / a;                               // TYPE_RGBA_16_PREMUL: the colour is divided by alpha before and multiplied back after
    float r = (float)p[0] * ia, g = (float)p[1] * ia, b = (float)p[2] * ia;
    r = r > 1.0f ? 1.0f : r; g = g > 1.0f ? 1.0f : g; b = b > 1.0f ? 1.0f : b;
    icc_lut_sample(lut, n, r, g, b, v);
    const float back = a * (1.0f / 65535.0f);
    for (int c = 0; c < 3; c++) { float o = rintf(v[c] * back); p[c] = (uint16_t)(o < 0.0f ? 0.0f : o > 65535.0f ? 65535.0f : o); }
  } else {
    uint8_t *p = (uint8_t *)px + (size_t)y * stride + (size_t)x * 4;
    icc_lut_sample(lut, n, (float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, v);          // n = 256: lands on lattice points
    for (int c = 0; c < 3; c++) { float o = rintf(v[c] * (255.0f / 65535.0f)); p[c] = (uint8_t)(o < 0.0f ? 0.0f : o > 255.0f ? 255.0f : o); }
  }
}
void launch_post_icc_lut(void *px, uint32_t stride, uint32_t w, uint32_t h, bool is_u16, const uint16_t *lut, int n, hipStream_t s) {
  dim3 grid(h, (w + 255) / 256);
  if (is_u16) hipLaunchKernelGGL(k_post_icc_lut<true>, grid, dim3(256), 0, s, px, stride, w, h, lut, n);
  else hipLaunchKernelGGL(k_post_icc_lut<false>, grid, dim3(256), 0, s, px, stride, w, h, lut, n);
}

}  // namespace jxlamd
