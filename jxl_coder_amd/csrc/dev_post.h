// jxl_coder_amd/csrc/dev_post.h — per-pixel device functions of the post-decode stages A10 (colour matrix / tone map) and A11 (premultiply, conversion
// into the Bitmap's format): shared by the stand-alone kernels (post.hip) and by the decoder's writer when the stages run inside it (kernels_filter.hip).
// References: jxlcoder/src/main/cpp/colorspaces/ColorMatrix.cpp:35-219, Rec2408ToneMapper.cpp:80-100, ReformatBitmap.cpp:46-263, imagebit/*.cpp.
#pragma once
#include "post.h"

namespace jxlamd {

__device__ __forceinline__ uint16_t half_bits(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }   // RNE

// one pixel of the conversion stage: (r, g, b, a) as the source holds them -> destination format KIND at column x of drow
template <int KIND>
__device__ __forceinline__ void post_convert_store(uint8_t *drow, uint32_t x, uint32_t r, uint32_t g, uint32_t b, uint32_t a, uint32_t depth, int attenuate) {
  if (attenuate && (KIND == kPostRgba8ToF16 || KIND == kPostRgba8To565 || KIND == kPostRgba8To1010102)) { r = (r * a) / 255u; g = (g * a) / 255u; b = (b * a) / 255u; }
  if (KIND == kPostU16ToF16 || KIND == kPostRgba8ToF16) {
    const float scale = 1.0f / (float)((1u << (KIND == kPostU16ToF16 ? depth : 8u)) - 1u);
    ushort4 o;
    o.x = half_bits((float)r * scale); o.y = half_bits((float)g * scale); o.z = half_bits((float)b * scale); o.w = half_bits((float)a * scale);
    ((ushort4 *)drow)[x] = o;
  } else if (KIND == kPostRgba16To8) {
    const uint32_t d = depth - 8;
    ((uint32_t *)drow)[x] = ((r >> d) & 0xff) | (((g >> d) & 0xff) << 8) | (((b >> d) & 0xff) << 16) | (((a >> d) & 0xff) << 24);
  } else if (KIND == kPostRgba8To565) {
    ((uint16_t *)drow)[x] = (uint16_t)(((r >> 3) << 11) | ((g >> 2) << 5) | (b >> 3));
  } else if (KIND == kPostRgba16To565) {
    const uint32_t rb = depth - 8 + 3, gd = depth - 8 + 2;
    ((uint16_t *)drow)[x] = (uint16_t)((((r >> rb) << 11) & 0xffffu) | (((g >> gd) << 5) & 0xffffu) | (b >> rb));
  } else if (KIND == kPostRgba8To1010102) {
    ((uint32_t *)drow)[x] = ((a >> 6) << 30) | ((b << 2) << 20) | ((g << 2) << 10) | (r << 2);
  } else if (KIND == kPostRgba16To1010102) {
    const uint32_t d = depth - 10, ad = depth - 2;
    ((uint32_t *)drow)[x] = (((a >> ad) & 3u) << 30) | (((b >> d) & 0x3ffu) << 20) | (((g >> d) & 0x3ffu) << 10) | ((r >> d) & 0x3ffu);
  } else if (KIND == kPostCopy8) {
    ((uint32_t *)drow)[x] = r | (g << 8) | (b << 16) | (a << 24);
  } else {
    ushort4 o; o.x = (uint16_t)r; o.y = (uint16_t)g; o.z = (uint16_t)b; o.w = (uint16_t)a;
    ((ushort4 *)drow)[x] = o;
  }
}
template <int KIND> constexpr bool post_src16() { return KIND == kPostU16ToF16 || KIND == kPostRgba16To8 || KIND == kPostRgba16To565 || KIND == kPostRgba16To1010102 || KIND == kPostCopy16; }
// A10, one pixel: LUT -> (tone map unless the row is stuck) -> matrix -> LUT.  FMA contraction off: the reference's operation order.
// ... from the linearised values on (the writer's pass has looked them up already for its zero-luma test)
__device__ __forceinline__ void post_matrix_lin(const ColorMatrixDev &P, bool tone, float fr, float fg, float fb, uint32_t &r, uint32_t &g, uint32_t &b) {
#pragma clang fp contract(off)
  if (tone) {
    const float y = 0.2627f * fr + 0.6780f * fg + 0.0593f * fb;
    const float scale = (1.0f + P.weight_a * y) / (1.0f + P.weight_b * y);
    fr = fminf(fr * scale, 1.0f); fg = fminf(fg * scale, 1.0f); fb = fminf(fb * scale, 1.0f);
  }
  const float nr = fr * P.m[0] + fg * P.m[1] + fb * P.m[2];
  const float ng = fr * P.m[3] + fg * P.m[4] + fb * P.m[5];
  const float nb = fr * P.m[6] + fg * P.m[7] + fb * P.m[8];
  #define IDX(v) ({ float c_ = (v) < 0.0f ? 0.0f : (v) > 1.0f ? 1.0f : (v); if (!((v) == (v))) c_ = 0.0f; uint32_t i_ = (uint32_t)(c_ * P.index_scale) & 0xffffu; i_ < P.index_max ? i_ : P.index_max; })
  r = P.gam_lut[IDX(nr)]; g = P.gam_lut[IDX(ng)]; b = P.gam_lut[IDX(nb)];
  #undef IDX
}
template <bool kU16>
__device__ __forceinline__ void post_matrix_px(const ColorMatrixDev &P, bool tone, uint32_t &r, uint32_t &g, uint32_t &b) {
  const uint32_t cap = kU16 ? P.index_max : 255u;
  const float fr = P.lin_lut[r < cap ? r : cap], fg = P.lin_lut[g < cap ? g : cap], fb = P.lin_lut[b < cap ? b : cap];
  post_matrix_lin(P, tone, fr, fg, fb, r, g, b);
}
// A10 + A11 of ONE pixel whose RGBA codes (8- or 16-bit, as the writer would have stored them) are r, g, b, a: output position (ox, oy) of the Bitmap.
// MODE 1: the writer's pass — tone-mapped as if the row held no pixel of zero linear luma; such a pixel records its column in row_fz (the reference's tone
// mapper never advances past the first one: the rest of the row stays un-mapped, Rec2408ToneMapper.cpp:91-93).  MODE 2: the pass after it — pixels at
// or behind their row's first zero-luma pixel are written again, un-mapped.  Together: bit for bit what k_post_fused makes of the stored RGBA.
template <int MODE>
__device__ __forceinline__ void post_emit(const DevPost &Q, uint32_t r, uint32_t g, uint32_t b, uint32_t a, int ox, int oy, bool src16) {
#pragma clang fp contract(off)
  bool tone = Q.matrix && Q.P.tone_map;
  if (MODE == 2) {
    if (!tone || (uint32_t)ox < Q.row_fz[1 + oy]) return;
    tone = false;
  }
  if (Q.matrix) {
    const uint32_t cap = src16 ? Q.P.index_max : 255u;
    const float fr = Q.P.lin_lut[r < cap ? r : cap], fg = Q.P.lin_lut[g < cap ? g : cap], fb = Q.P.lin_lut[b < cap ? b : cap];
    if (MODE == 1 && tone) {
      const float y = 0.2627f * fr + 0.6780f * fg + 0.0593f * fb;
      if (y == 0.0f) { atomicMin(&Q.row_fz[1 + oy], (uint32_t)ox); Q.row_fz[0] = 1u; }      // word 0: some row of the frame has one (the second pass has work)
    }
    post_matrix_lin(Q.P, tone, fr, fg, fb, r, g, b);
  }
  if (Q.premul) {
    const uint32_t maxv = (1u << Q.depth) - 1u;
    if (src16) { r = (uint16_t)((r * a) / maxv); g = (uint16_t)((g * a) / maxv); b = (uint16_t)((b * a) / maxv); }
    else { r = (r * a) / 255u; g = (g * a) / 255u; b = (b * a) / 255u; }
  }
  uint8_t *drow = Q.dst + (size_t)oy * Q.dst_stride;
  const uint32_t x = (uint32_t)ox, depth = (uint32_t)Q.depth; const int at = Q.attenuate;
  switch (Q.kind) {
    case kPostU16ToF16: post_convert_store<kPostU16ToF16>(drow, x, r, g, b, a, depth, at); break;
    case kPostRgba8ToF16: post_convert_store<kPostRgba8ToF16>(drow, x, r, g, b, a, depth, at); break;
    case kPostRgba16To8: post_convert_store<kPostRgba16To8>(drow, x, r, g, b, a, depth, at); break;
    case kPostRgba8To565: post_convert_store<kPostRgba8To565>(drow, x, r, g, b, a, depth, at); break;
    case kPostRgba16To565: post_convert_store<kPostRgba16To565>(drow, x, r, g, b, a, depth, at); break;
    case kPostRgba8To1010102: post_convert_store<kPostRgba8To1010102>(drow, x, r, g, b, a, depth, at); break;
    case kPostRgba16To1010102: post_convert_store<kPostRgba16To1010102>(drow, x, r, g, b, a, depth, at); break;
    case kPostCopy8: post_convert_store<kPostCopy8>(drow, x, r, g, b, a, depth, at); break;
    default: post_convert_store<kPostCopy16>(drow, x, r, g, b, a, depth, at); break;
  }
}

}  // namespace jxlamd
