// jxl_coder_amd/csrc/post.h — launchers of the post-decode kernels (post.hip); buffers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jxlamd {

enum PostKind {                  // one thread per pixel, src RGBA8 / RGBA16 -> dst
  kPostU16ToF16 = 0, kPostRgba8ToF16, kPostRgba16To8, kPostRgba8To565, kPostRgba16To565, kPostRgba8To1010102, kPostRgba16To1010102,
  kPostCopy8, kPostCopy16,
};

void launch_post_premultiply(void *px, uint32_t stride, uint32_t w, uint32_t h, bool is_u16, uint32_t depth, hipStream_t s);
void launch_post_convert(PostKind kind, const void *src, uint32_t src_stride, void *dst, uint32_t dst_stride, uint32_t w, uint32_t h,
                         uint32_t depth, bool attenuate, hipStream_t s);

struct ColorMatrixDev {
  float m[9];
  int tone_map; float weight_a, weight_b;
  const float *lin_lut; const uint16_t *gam_lut;
  float index_scale; uint32_t index_max;
};
void launch_post_color_matrix(void *px, uint32_t stride, uint32_t w, uint32_t h, bool is_u16, const ColorMatrixDev &P, hipStream_t s);
// A10 + A11 INSIDE the decoder's writer: the last filter stage hands its RGBA codes straight to the colour matrix / tone map, the
// premultiply and the conversion into the Bitmap's format (kernels_filter.hip: k_filter_b<3, 1>); nothing but the Bitmap is written.  DevBuffers::post
// points at one of these (device memory) for a frame decoded that way.
struct DevPost {
  ColorMatrixDev P;
  int32_t matrix, premul, kind, depth, attenuate;     // kind: PostKind
  uint32_t dst_stride; int32_t rows;                  // rows: output rows (size of row_fz - 1)
  uint8_t *dst;                                       // the Bitmap (rows of dst_stride bytes)
  uint32_t *row_fz;                                   // [1 + output rows]: word 0 = some row has a pixel of zero linear luma; word 1 + y = the first such pixel of row y (0xFFFFFFFF: none) — see k_filter_b<3, 2>
};
// A10 (P != null) + premultiply + conversion in one pass: src (read-only) -> dst; equals launch_post_color_matrix, launch_post_premultiply, launch_post_convert in turn
void launch_post_fused(PostKind kind, const void *src, uint32_t src_stride, void *dst, uint32_t dst_stride, uint32_t w, uint32_t h, const ColorMatrixDev *P, bool premul,
                       uint32_t depth, bool attenuate, hipStream_t s);

// A8 (cpp/colorspaces/colorspace.cpp:38-86): the profile -> sRGB transform as an n^3 RGB16 lattice (host_icc_lut.cpp), applied in place with
// trilinear interpolation; alpha is copied (u8) / the colour is un-premultiplied around the transform (u16: the reference passes TYPE_RGBA_16_PREMUL)
void launch_post_icc_lut(void *px, uint32_t stride, uint32_t w, uint32_t h, bool is_u16, const uint16_t *lut, int n, hipStream_t s);

}  // namespace jxlamd
