// jxl_coder_amd/csrc/dev_compose.h — composition stages between the loop filters and the writer (ISO/IEC 18181-1 Annex K: patches;
// reference frames): Modular planes -> f32 planes, patch blending, copy into a reference slot, the writer of frames that are not XYB.
// What libjxl's render pipeline does in its "Patches" stage and when it keeps a frame for later ("save_as_reference", reference call site
// jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).  Files the reference's own encoder writes for text / screenshots at its default settings
// carry a small kReferenceOnly frame with the glyph-like patches and a main frame that adds them back (interop/JxlEncoding.cpp:145-160).
#pragma once
#include "dev_recon.h"
#include "dev_modframe.h"

namespace jxlamd {

// the image planes of a composed frame after its loop filters: the per-stage filter kernels ping-pong between the two sets
JXL_DEV bool compose_final_is_a(const DevFrame &F) {
  if (F.is_modular && !F.xyb_modular) return true;
  if (F.subsampled) return false;                       // chroma_upsample_pixel moved the image into the second set
  int n = (F.gab ? 1 : 0) + F.epf_iters;
  return (n & 1) == 0;
}

// Modular-encoded frame -> f32 planes (plane_a): integer samples / (2^bits - 1), or — XYB image — Y, X, B - Y times the LF dequantisation factors
JXL_DEV void mod_to_planes_pixel(const DevBuffers &B, const DevFrame &F, int x, int y) {
  const size_t si = (size_t)y * (size_t)F.width + (size_t)x, po = (size_t)y * (size_t)F.pw + (size_t)x;
  if (F.xyb_modular) {
    const int32_t vy = mod_plane(B, F, F.mod_out[0])[si], vx = mod_plane(B, F, F.mod_out[1])[si], vb = mod_plane(B, F, F.mod_out[2])[si];
    B.plane_a[0][po] = (float)vx * F.mod_xyb_fac[0];
    B.plane_a[1][po] = (float)vy * F.mod_xyb_fac[1];
    B.plane_a[2][po] = (float)(vb + vy) * F.mod_xyb_fac[2];
  } else {
    const float sc = 1.0f / (float)((1u << F.mod_bits) - 1);
    for (int c = 0; c < 3; c++) B.plane_a[c][po] = (float)mod_plane(B, F, F.mod_out[c])[si] * sc;
  }
}

// one sample of one patch placement: item = pixel index inside the patch rectangle
JXL_DEV void patch_blend_sample(const DevBuffers &B, const DevFrame &F, const DevPatch &P, int item) {
  const int iy = item / P.w, ix = item - iy * P.w;
  const int x = P.x + ix, y = P.y + iy;
  if ((unsigned)x >= (unsigned)F.width || (unsigned)y >= (unsigned)F.height) return;
  const bool a = compose_final_is_a(F);
  const size_t po = (size_t)y * (size_t)F.pw + (size_t)x, ro = (size_t)(P.y0 + iy) * (size_t)F.ref_w[P.ref] + (size_t)(P.x0 + ix);
  for (int c = 0; c < 3; c++) {
    float *dst = (a ? B.plane_a[c] : B.plane_b[c]) + po;
    const float r = B.ref[P.ref][c][ro];
    if (P.mode == 1) *dst = r;
    else if (P.mode == 3) *dst = *dst * r;
    else if (P.mode == 2) {
#ifdef __HIPCC__
      atomicAdd(dst, r);          // placements may overlap; without overlap this is the plain sum
#else
      *dst += r;
#endif
    }
  }
}

// Upsampling (factor 2 / 4 / 8): output pixel (X, Y) = the 5 x 5 neighbourhood of coded pixel (X / N, Y / N) weighted by the kernel of its phase
// (X % N, Y % N), clamped to the neighbourhood's range; the frame edges are mirrored.  Multiply and add stay separate (the reference's libjxl is an
// SSE2 build: no fused multiply-add), rows outer, columns inner.
JXL_DEV void upsample_pixel(const DevBuffers &B, const DevFrame &F, const uint8_t *stat, int X, int Y) {
  const DevStatic &ST = *(const DevStatic *)stat;
  const int N = F.upsampling, sh = N == 2 ? 1 : N == 4 ? 2 : 3;
  const int x = X >> sh, y = Y >> sh, ox = X & (N - 1), oy = Y & (N - 1);
  const float *k = (const float *)(stat + ST.ups_off[sh - 1]) + (size_t)(oy * N + ox) * 25;
  const bool a = compose_final_is_a(F);
  int xs[5], ys[5];
  for (int i = 0; i < 5; i++) { xs[i] = mirror(x + i - 2, F.width); ys[i] = mirror(y + i - 2, F.height); }
  for (int c = 0; c < 3; c++) {
    const float *src = a ? B.plane_a[c] : B.plane_b[c];
    float acc = 0.0f, mn = src[(size_t)ys[2] * (size_t)F.pw + (size_t)xs[2]], mx = mn;
    for (int iy = 0; iy < 5; iy++)
      for (int ix = 0; ix < 5; ix++) {
        const float v = src[(size_t)ys[iy] * (size_t)F.pw + (size_t)xs[ix]];
#ifdef __HIPCC__
        acc = __fadd_rn(__fmul_rn(k[iy * 5 + ix], v), acc);
#else
        acc = k[iy * 5 + ix] * v + acc;
#endif
        mn = v < mn ? v : mn; mx = v > mx ? v : mx;
      }
    acc = acc < mn ? mn : acc > mx ? mx : acc;
    B.up[c][(size_t)Y * (size_t)F.full_w + (size_t)X] = acc;
  }
}
// the alpha channel of a frame whose alpha is coded coarser than the image (extra-channel upsampling alpha_up = 2 / 4 / 8): same kernels, same
// clamp, on the Modular plane's samples as fractions of full scale
JXL_DEV void upsample_alpha_pixel(const DevBuffers &B, const DevFrame &F, const uint8_t *stat, int X, int Y) {
  const DevStatic &ST = *(const DevStatic *)stat;
  const int N = F.alpha_up, sh = N == 2 ? 1 : N == 4 ? 2 : 3;
  const int x = X >> sh, y = Y >> sh, ox = X & (N - 1), oy = Y & (N - 1);
  const float *k = (const float *)(stat + ST.ups_off[sh - 1]) + (size_t)(oy * N + ox) * 25;
  const int32_t *src = mod_plane(B, F, F.mod_out[3]);
  const float sc = 1.0f / (float)((1u << F.mod_alpha_bits) - 1);
  int xs[5], ys[5];
  for (int i = 0; i < 5; i++) { xs[i] = mirror(x + i - 2, F.alpha_w); ys[i] = mirror(y + i - 2, F.alpha_h); }
  float acc = 0.0f, mn = (float)src[(size_t)ys[2] * (size_t)F.alpha_w + (size_t)xs[2]] * sc, mx = mn;
  for (int iy = 0; iy < 5; iy++)
    for (int ix = 0; ix < 5; ix++) {
      const float v = (float)src[(size_t)ys[iy] * (size_t)F.alpha_w + (size_t)xs[ix]] * sc;
#ifdef __HIPCC__
      acc = __fadd_rn(__fmul_rn(k[iy * 5 + ix], v), acc);
#else
      acc = k[iy * 5 + ix] * v + acc;
#endif
      mn = v < mn ? v : mn; mx = v > mx ? v : mx;
    }
  acc = acc < mn ? mn : acc > mx ? mx : acc;
  B.up[3][(size_t)Y * (size_t)F.full_w + (size_t)X] = acc;
}
// writer of an upsampled XYB frame: full-resolution planes -> colour transform -> RGBA
JXL_DEV void upsampled_write_pixel(const DevBuffers &B, const uint8_t *stat, int out_bits, int X, int Y) {
  const DevFrame &F = frame_of(B);
  const size_t o = (size_t)Y * (size_t)F.full_w + (size_t)X;
  xyb_write_value(B, stat, *(const DevStatic *)stat, B.up[0][o], B.up[1][o], B.up[2][o], out_bits, X, Y);
}

// Chroma upsampling of a YCbCr frame whose chroma is coded at half resolution (libjxl's render stages HChromaUps, then VChromaUps, in front of the
// loop filters — which such a frame, a recompressed JPEG, does not have): out[2x] = 0.25 in[x - 1] + 0.75 in[x], out[2x + 1] = 0.25 in[x + 1] + 0.75 in[x],
// first along the rows, then along the columns of the result; the channel's edges (ceil(size / 2) samples) are mirrored; every product and sum rounded on
// its own.  One output sample of channel c, plane_a -> plane_b; channels at full resolution are copied.
JXL_DEV int mirror1(int x, int n) { return x < 0 ? -x - 1 : x >= n ? 2 * n - 1 - x : x; }
JXL_DEV void chroma_upsample_pixel(const DevBuffers &B, const DevFrame &F, int c, int X, int Y) {
  const int hs = F.hshift[c], vs = F.vshift[c];
  const int cw = hs ? (F.width + 1) / 2 : F.width, chh = vs ? (F.height + 1) / 2 : F.height;
  const float *in = B.plane_a[c];
  const size_t pw = (size_t)F.pw;
  const int x = hs ? X >> 1 : X, xn = hs ? mirror1((X & 1) ? x + 1 : x - 1, cw) : 0;
  const int y = vs ? Y >> 1 : Y, yn = vs ? mirror1((Y & 1) ? y + 1 : y - 1, chh) : 0;
  float cur, nb = 0.0f;
  if (hs) {
    cur = mul_add_rn(0.25f, in[(size_t)y * pw + (size_t)xn], in[(size_t)y * pw + (size_t)x] * 0.75f);
    if (vs) nb = mul_add_rn(0.25f, in[(size_t)yn * pw + (size_t)xn], in[(size_t)yn * pw + (size_t)x] * 0.75f);
  } else {
    cur = in[(size_t)y * pw + (size_t)x];
    if (vs) nb = in[(size_t)yn * pw + (size_t)x];
  }
  B.plane_b[c][(size_t)Y * pw + (size_t)X] = vs ? mul_add_rn(nb, 0.25f, cur * 0.75f) : cur;
}

// copy the composed frame into a reference slot (dense w x h planes)
JXL_DEV void save_ref_pixel(const DevBuffers &B, const DevFrame &F, float *const dst[3], int x, int y) {
  const bool a = compose_final_is_a(F);
  const size_t po = (size_t)y * (size_t)F.pw + (size_t)x, ro = (size_t)y * (size_t)F.width + (size_t)x;
  for (int c = 0; c < 3; c++) dst[c][ro] = (a ? B.plane_a[c] : B.plane_b[c])[po];
}

// writer of a composed frame that is not XYB (Modular-encoded, samples already in the image's own colour space): clamp, scale, round
JXL_DEV void plain_write_pixel(const DevBuffers &B, int out_bits, int x, int y) {
  const DevFrame &F = frame_of(B);
  const size_t po = (size_t)y * (size_t)F.pw + (size_t)x, si = (size_t)y * (size_t)F.width + (size_t)x;
  const float maxv = out_bits == 16 ? 65535.0f : 255.0f;
  uint32_t px[4];
  for (int c = 0; c < 4; c++) {
    float t;
    if (c < 3) t = B.plane_a[c][po];
    else if (F.mod_out[3] < 0) t = 1.0f;
    else t = (float)mod_plane(B, F, F.mod_out[3])[si] / (float)((1u << F.mod_alpha_bits) - 1);
    t = t < 0.0f ? 0.0f : t > 1.0f ? 1.0f : t;
    if (!(t == t)) t = 0.0f;
    px[c] = (uint32_t)(int)rintf(t * maxv);
  }
  x += F.crop_x0; y += F.crop_y0;
  const int W = F.canvas_w, H = F.canvas_h;
  if ((unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H) return;
  int ox = x, oy = y;
  switch (F.orientation) {
    case 2: ox = W - 1 - x; break;
    case 3: ox = W - 1 - x; oy = H - 1 - y; break;
    case 4: oy = H - 1 - y; break;
    case 5: ox = y; oy = x; break;
    case 6: ox = H - 1 - y; oy = x; break;
    case 7: ox = H - 1 - y; oy = W - 1 - x; break;
    case 8: ox = y; oy = W - 1 - x; break;
    default: break;
  }
  const size_t di = ((size_t)oy * (size_t)F.out_w + (size_t)ox) * 4;
  if (out_bits == 8) *(uint32_t *)(B.out + di) = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
  else { uint16_t *o = (uint16_t *)B.out + di; for (int c = 0; c < 4; c++) o[c] = (uint16_t)px[c]; }
}

}  // namespace jxlamd
