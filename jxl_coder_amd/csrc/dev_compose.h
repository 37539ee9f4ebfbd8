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
    for (int c = 0; c < 3; c++) { const int32_t v = mod_plane(B, F, F.mod_out[c])[si]; B.plane_a[c][po] = F.mod_exp_bits ? sample_bits_to_float(v, F.mod_bits, F.mod_exp_bits) : int_sample_to_unit(v, F.mod_bits); }
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
  const float *k = (const float *)(F.ups_custom_off[sh - 1] ? B.tables + F.ups_custom_off[sh - 1] : stat + ST.ups_off[sh - 1]) + (size_t)(oy * N + ox) * 25;
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
  const float *k = (const float *)(F.ups_custom_off[sh - 1] ? B.tables + F.ups_custom_off[sh - 1] : stat + ST.ups_off[sh - 1]) + (size_t)(oy * N + ox) * 25;
  const int32_t *src = mod_plane(B, F, F.mod_out[3]);
  const float sc = 1.0f / (float)((1u << F.mod_alpha_bits) - 1);
  int xs[5], ys[5];
  for (int i = 0; i < 5; i++) { xs[i] = mirror(x + i - 2, F.alpha_w); ys[i] = mirror(y + i - 2, F.alpha_h); }
  float acc = 0.0f, mn = F.mod_alpha_exp_bits ? alpha_sample_value(F, src[(size_t)ys[2] * (size_t)F.alpha_w + (size_t)xs[2]], true) : (float)src[(size_t)ys[2] * (size_t)F.alpha_w + (size_t)xs[2]] * sc, mx = mn;
  for (int iy = 0; iy < 5; iy++)
    for (int ix = 0; ix < 5; ix++) {
      const float v = F.mod_alpha_exp_bits ? alpha_sample_value(F, src[(size_t)ys[iy] * (size_t)F.alpha_w + (size_t)xs[ix]], true) : (float)src[(size_t)ys[iy] * (size_t)F.alpha_w + (size_t)xs[ix]] * sc;
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
  if ((F.is_modular && !F.xyb_modular) || F.not_xyb) plain_write_value(B, stat, *(const DevStatic *)stat, B.up[0][o], B.up[1][o], B.up[2][o], out_bits, X, Y);      // the image's own samples
  else xyb_write_value(B, stat, *(const DevStatic *)stat, B.up[0][o], B.up[1][o], B.up[2][o], out_bits, X, Y);
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

// ---- Noise synthesis (libjxl: PrepareNoiseInput / RandomImage, the "ConvolveNoise" and "AddNoise" render stages).
// The random planes: every 256 x 256 group runs its own Xorshift128+ — eight generators side by side, seeded through SplitMix64 with (frame counters, group
// origin) — over plane 0, 1, 2 in turn, row by row; a call yields eight 64-bit words = sixteen floats in [1, 2) (the upper 23 bits of each half as a
// mantissa); a row takes one call per whole sixteen-float batch that ends BEFORE its last sample, plus one more for the rest.  Work item = one generator.
JXL_DEV uint64_t noise_splitmix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Where the noise goes: the frame's final planes, or — an upsampled frame — the upsampled image (libjxl's stage order: Upsampling, then Noise: the random planes
// are drawn at the full resolution, in tiles of 256 x 256 seeded by their position there; Random3Planes under PrepareNoiseInput)
struct NoiseGeom { int w, h, stride, xtiles, ytiles; float *dst[3]; };
JXL_DEV NoiseGeom noise_geom(const DevBuffers &B, const DevFrame &F) {
  NoiseGeom G;
  if (F.upsampling > 1) { G.w = F.full_w; G.h = F.full_h; G.stride = F.full_w; for (int c = 0; c < 3; c++) G.dst[c] = B.up[c]; }
  else { G.w = F.width; G.h = F.height; G.stride = F.pw; const bool a = compose_final_is_a(F); for (int c = 0; c < 3; c++) G.dst[c] = a ? B.plane_a[c] : B.plane_b[c]; }
  G.xtiles = (G.w + 255) / 256; G.ytiles = (G.h + 255) / 256;
  return G;
}
JXL_DEV void noise_gen_lane(const DevBuffers &B, const DevFrame &F, int g, int lane) {
  const NoiseGeom G = noise_geom(B, F);
  const int gx = g % G.xtiles, gy = g / G.xtiles;
  const int x0 = gx * 256, y0 = gy * 256;
  const int xs = G.w - x0 < 256 ? G.w - x0 : 256, ys = G.h - y0 < 256 ? G.h - y0 : 256;
  uint64_t s0 = noise_splitmix((((uint64_t)F.noise_seed[0]) << 32) + F.noise_seed[1] + 0x9E3779B97F4A7C15ull);
  uint64_t s1 = noise_splitmix((((uint64_t)(uint32_t)x0) << 32) + (uint32_t)y0 + 0x9E3779B97F4A7C15ull);
  for (int i = 0; i < lane; i++) { s0 = noise_splitmix(s0); s1 = noise_splitmix(s1); }
  int nfull = 0;
  while ((nfull + 1) * 16 < xs) nfull++;                      // batches with x + 16 < xsize
  for (int c = 0; c < 3; c++)
    for (int y = 0; y < ys; y++) {
      float *row = B.noise[c] + (size_t)(y0 + y) * (size_t)G.stride + (size_t)x0;
      for (int f = 0; f <= nfull; f++) {
        uint64_t a = s0; const uint64_t b = s1;
        const uint64_t bits = a + b;
        s0 = b;
        a ^= a << 23;
        s1 = a ^ b ^ (a >> 18) ^ (b >> 5);
        for (int k = 0; k < 2; k++) {
          const int x = f * 16 + 2 * lane + k;
          if (x >= xs) continue;
          const uint32_t w = (uint32_t)(k ? bits >> 32 : bits);
          const uint32_t fb = (w >> 9) | 0x3F800000u;
#ifdef __HIPCC__
          row[x] = __uint_as_float(fb);
#else
          float fv; memcpy(&fv, &fb, 4); row[x] = fv;
#endif
        }
      }
    }
}
// one pixel: the three planes through the 5 x 5 high-pass (0.16 everywhere, -3.84 in the middle; the frame's edges mirrored), scaled by 0.22 and by the
// strength the 8-point curve gives for the local red / green intensities ((Y + X) / 2, (Y - X) / 2), added to X, Y, B with libjxl's correlations
JXL_DEV float noise_strength(const DevFrame &F, float x) {
  float sx = x * 6.0f; sx = sx > 0.0f ? sx : 0.0f;
  float fl = floorf(sx), fr = sx - fl;
  if (sx >= 7.0f) { fl = 6.0f; fr = 1.0f; }
  const int i = (int)fl;
  const float lo = F.noise_lut[i], hi = F.noise_lut[i + 1];
  const float v = mul_add_rn(hi - lo, fr, lo);
  return v < 0.0f ? 0.0f : v > 1.0f ? 1.0f : v;
}
JXL_DEV void noise_add_pixel(const DevBuffers &B, const DevFrame &F, int x, int y) {
  const NoiseGeom G = noise_geom(B, F);
  float rnd[3];
  int xs[5], ys[5];
  for (int i = 0; i < 5; i++) { xs[i] = mirror(x + i - 2, G.w); ys[i] = mirror(y + i - 2, G.h); }
  for (int c = 0; c < 3; c++) {
    const float *p = B.noise[c];
    float others = 0.0f;
    for (int i = 0; i < 5; i++) {
      others += p[(size_t)ys[0] * (size_t)G.stride + (size_t)xs[i]]; others += p[(size_t)ys[1] * (size_t)G.stride + (size_t)xs[i]];
      others += p[(size_t)ys[3] * (size_t)G.stride + (size_t)xs[i]]; others += p[(size_t)ys[4] * (size_t)G.stride + (size_t)xs[i]];
    }
    const float *mid = p + (size_t)ys[2] * (size_t)G.stride;
    others += mid[xs[0]]; others += mid[xs[1]]; others += mid[xs[3]]; others += mid[xs[4]];
    rnd[c] = mul_add_rn(others, 0.16f, mid[xs[2]] * -3.84f) * 0.22f;
  }
  const size_t po = (size_t)y * (size_t)G.stride + (size_t)x;
  float *px = G.dst[0] + po, *py = G.dst[1] + po, *pb = G.dst[2] + po;
  const float vx = *px, vy = *py;
  const float sg = noise_strength(F, (vy - vx) * 0.5f), sr = noise_strength(F, (vy + vx) * 0.5f);
  const float red = sr * mul_add_rn(0.0078125f, rnd[0], 0.9921875f * rnd[2]);
  const float green = sg * mul_add_rn(0.0078125f, rnd[1], 0.9921875f * rnd[2]);
  const float rg = red + green;
  *px = mul_add_rn(F.base_x, rg, red - green) + vx;
  *py = vy + rg;
  *pb = mul_add_rn(F.base_b, rg, *pb);
}

// ---- Splines (K.4; libjxl's "Splines" stage, after the patches): pixel (x, y) of the frame receives, from every spline sample whose box reaches it, colour x
// sigma / 4 x intensity x (erf((d / 2 + 1 / (2 sqrt 2)) / sigma) - erf((d / 2 - 1 / (2 sqrt 2)) / sigma))^2 at its distance d — in the order libjxl adds them (the
// row's segment list).  erf as libjxl evaluates it: 1 - 1 / (1 + a1 |x| + a2 x^2 + a3 |x|^3 + a4 x^4)^4 (its FastErff; 5e-4 of a blob's peak at most from the exact one).
JXL_DEV float spline_erf(float v) {
  const float a = fabsf(v);
  float d = a * 7.77394369e-02f + 2.05260015e-04f;
  d = d * a + 2.32120216e-01f;
  d = d * a + 2.77820801e-01f;
  d = d * a + 1.0f;
  const float d2 = d * d, inv = 1.0f / d2;
  const float r = 1.0f - inv * inv;
  return v <= 0.0f ? -r : r;
}
JXL_DEV void spline_pixel(const DevBuffers &B, const DevFrame &F, int x, int y) {
  const DevSplineSeg *segs = (const DevSplineSeg *)(B.tables + F.spline_seg_off);
  const uint32_t *rows = (const uint32_t *)(B.tables + F.spline_row_off), *idx = (const uint32_t *)(B.tables + F.spline_idx_off);
  const uint32_t i0 = rows[y], i1 = rows[y + 1];
  if (i0 == i1) return;
  const bool a = compose_final_is_a(F);
  const size_t po = (size_t)y * (size_t)F.pw + (size_t)x;
  float acc[3];
  for (int c = 0; c < 3; c++) acc[c] = (a ? B.plane_a[c] : B.plane_b[c])[po];
  bool any = false;
  for (uint32_t i = i0; i < i1; i++) {
    const DevSplineSeg sg = segs[idx[i]];
    const long long xs = llroundf(sg.cx - sg.maxdist), xe = llroundf(sg.cx + sg.maxdist);
    if ((long long)x < xs || (long long)x > xe) continue;
    const float dx = (float)x - sg.cx, dy = (float)y - sg.cy;
    const float dist = sqrtf(dx * dx + dy * dy);
    const float f = spline_erf((dist * 0.5f + 0.353553391f) * sg.inv_sigma) - spline_erf((dist * 0.5f - 0.353553391f) * sg.inv_sigma);
    const float li = sg.sigma_over_4_times_intensity * f * f;
    for (int c = 0; c < 3; c++) acc[c] += sg.color[c] * li;
    any = true;
  }
  if (any) for (int c = 0; c < 3; c++) (a ? B.plane_a[c] : B.plane_b[c])[po] = acc[c];
}

// ---- Blending (libjxl's "Blending" stage, after the colour transform): one pixel (x, y) of the CANVAS.  Background = the blend source's canvas (reference
// slot bl_src; transparent black when the slot is empty); inside the frame's rectangle the frame's colour (in the image's colour encoding, not clamped) and
// alpha are combined with it by the frame's BlendingInfo, outside the background shows.  The result is kept (canvas_save: a later frame's background)
// and / or written out with the writer's clamp, dither and orientation.  Alpha blending as libjxl's PerformAlphaBlending / PerformAlphaWeightedAdd /
// PerformMulBlending (not premultiplied: out = (fg fa + bg ba (1 - fa)) / (1 - (1 - fa)(1 - ba)); premultiplied: fg + bg (1 - fa)).
JXL_DEV void blend_canvas_pixel(const DevBuffers &B, const uint8_t *stat, int out_bits, int x, int y) {
  const DevFrame &F = frame_of(B);
  const DevStatic &ST = *(const DevStatic *)stat;
  const int W = F.canvas_w, H = F.canvas_h;
  const size_t ci = (size_t)y * (size_t)W + (size_t)x;
  const bool has_alpha = (F.has_ec || F.is_modular) && F.mod_out[3] >= 0;
  float bg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int s = F.bl_src;
  if (s >= 0 && B.ref[s][0]) {
    for (int c = 0; c < 3; c++) bg[c] = B.ref[s][c][ci];
    bg[3] = B.ref_a[s] ? B.ref_a[s][ci] : 1.0f;
  } else if (!has_alpha) bg[3] = 1.0f;
  float out[4] = {bg[0], bg[1], bg[2], bg[3]};
  const int fx = x - F.crop_x0, fy = y - F.crop_y0;
  const bool ups = F.upsampling > 1;                       // an upsampled frame is blended at its full resolution: the upsampled planes (DevBuffers::up), after the noise
  if (fx >= 0 && fy >= 0 && fx < (ups ? F.full_w : F.width) && fy < (ups ? F.full_h : F.height)) {
    const size_t po = ups ? (size_t)fy * (size_t)F.full_w + (size_t)fx : (size_t)fy * (size_t)F.pw + (size_t)fx;
    const bool a = compose_final_is_a(F);
    const float p0 = ups ? B.up[0][po] : (a ? B.plane_a[0] : B.plane_b[0])[po], p1 = ups ? B.up[1][po] : (a ? B.plane_a[1] : B.plane_b[1])[po], p2 = ups ? B.up[2][po] : (a ? B.plane_a[2] : B.plane_b[2])[po];
    float fg[3];
    if ((F.is_modular && !F.xyb_modular) || F.not_xyb) plain_to_rgb(F, p0, p1, p2, fg); else xyb_to_rgb(F, p0, p1, p2, fg);
    float fa = 1.0f;
    if (has_alpha) fa = F.alpha_up > 1 ? B.up[3][(size_t)fy * (size_t)F.full_w + (size_t)fx]      // enlarged beforehand (upsample_alpha_pixel)
                                       : alpha_sample_value(F, mod_plane(B, F, F.mod_out[3])[(size_t)fy * (size_t)F.alpha_w + (size_t)fx], true);
    // colour channels
    {
      float wa = fa;
      if (F.bl_clamp_c) wa = wa < 0.0f ? 0.0f : wa > 1.0f ? 1.0f : wa;
      switch (F.bl_mode_c) {
        case 1: for (int c = 0; c < 3; c++) out[c] = bg[c] + fg[c]; break;
        case 2:
          if (F.bl_premultiplied) { for (int c = 0; c < 3; c++) out[c] = fg[c] + bg[c] * (1.0f - wa); }
          else {
            const float na = 1.0f - (1.0f - wa) * (1.0f - bg[3]);
            const float rna = na > 0.0f ? 1.0f / na : 0.0f;
            for (int c = 0; c < 3; c++) out[c] = (fg[c] * wa + bg[c] * bg[3] * (1.0f - wa)) * rna;
          }
          break;
        case 3: for (int c = 0; c < 3; c++) out[c] = bg[c] + fg[c] * wa; break;
        case 4: for (int c = 0; c < 3; c++) { float m = fg[c]; if (F.bl_clamp_c) m = m < 0.0f ? 0.0f : m > 1.0f ? 1.0f : m; out[c] = bg[c] * m; } break;
        default: for (int c = 0; c < 3; c++) out[c] = fg[c]; break;
      }
    }
    // the alpha channel, by its own BlendingInfo
    if (has_alpha) {
      float wa = fa;
      if (F.bl_clamp_a) wa = wa < 0.0f ? 0.0f : wa > 1.0f ? 1.0f : wa;
      switch (F.bl_mode_a) {
        case 1: out[3] = bg[3] + fa; break;
        case 2: out[3] = 1.0f - (1.0f - wa) * (1.0f - bg[3]); break;
        case 3: out[3] = bg[3]; break;
        case 4: { float m = fa; if (F.bl_clamp_a) m = m < 0.0f ? 0.0f : m > 1.0f ? 1.0f : m; out[3] = bg[3] * m; } break;
        default: out[3] = fa; break;
      }
      // libjxl blends the extra channels first and the colour afterwards, and its four-channel PerformAlphaBlending writes the blended alpha
      // 1 - (1 - fa)(1 - ba) into the colour's alpha channel whatever that channel's own mode asked for (established on the reference binary:
      // tests/golden/an_split_modes_lossless)
      if (F.bl_mode_c == 2) {
        float wc = fa;
        if (F.bl_clamp_c) wc = wc < 0.0f ? 0.0f : wc > 1.0f ? 1.0f : wc;
        out[3] = 1.0f - (1.0f - wc) * (1.0f - bg[3]);
      }
    }
  }
  if (B.canvas_save[0]) { for (int c = 0; c < 3; c++) B.canvas_save[c][ci] = out[c]; if (B.canvas_save[3]) B.canvas_save[3][ci] = out[3]; }
  if (F.no_output) return;
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
  if (!has_alpha) out[3] = 1.0f;
  for (int c = 0; c < 4; c++) { const float t = out[c]; out[c] = t < 0.0f ? 0.0f : t > 1.0f ? 1.0f : t; if (!(t == t)) out[c] = 0.0f; }
  if (out_bits == 8) {
    // The 8-bit writer's dither.  The reference's libjxl (an SSE2 build) loads FOUR consecutive entries of the flat 32 x 32 table at
    // (row & 31) * 32 + (first column of the vector & 31): a vector that straddles a multiple of 32 takes its upper lanes from the NEXT row of the table.
    // Vectors start at the left edge of the row segment the writer is handed — column 0 for whole rows, the frame's left edge in the frame's columns, its
    // right edge behind them (in every row of the canvas) — so the straddle only shows next to layers whose edges are not multiples of four (established on the reference binary's output of
    // blended animations; tests/golden/an*).  Unoriented images only: with an orientation the writer is handed transposed / mirrored rows.
    int di32 = (oy & 31) * 32 + (ox & 31);
    if (F.orientation == 1) {
      const int rx0 = F.crop_x0 > 0 ? F.crop_x0 : 0, rx1 = F.crop_x0 + F.width;
      const int seg = x >= rx1 ? rx1 : x >= rx0 ? rx0 : 0;          // the canvas left of the frame, the frame's columns, the canvas right of it: three strips, every row
      const int vs = seg + ((x - seg) & ~3);
      di32 = ((y & 31) * 32 + (vs & 31) + (x - vs)) & 1023;
    }
    const float d = st_f(stat, ST.dither_off)[F.orientation > 4 ? (ox & 31) * 32 + (oy & 31) : di32];
    uint32_t px = 0;
    for (int c = 0; c < 4; c++) px |= (uint32_t)(uint8_t)(int)rintf(out[c] * 255.0f + d) << (8 * c);
    *(uint32_t *)(B.out + di) = px;
  } else {
    uint16_t *o16 = (uint16_t *)B.out + di;
    for (int c = 0; c < 4; c++) o16[c] = (uint16_t)(int)rintf(out[c] * 65535.0f);
  }
}

// copy the composed frame into a reference slot (dense w x h planes)
JXL_DEV void save_ref_pixel(const DevBuffers &B, const DevFrame &F, float *const dst[3], int x, int y) {
  const bool a = compose_final_is_a(F);
  const size_t po = (size_t)y * (size_t)F.pw + (size_t)x, ro = (size_t)y * (size_t)F.width + (size_t)x;
  for (int c = 0; c < 3; c++) dst[c][ro] = (a ? B.plane_a[c] : B.plane_b[c])[po];
}

// writer of a composed frame that is not XYB (Modular-encoded, samples already in the image's own colour space): clamp, scale, round — through the common store
// (rgba_codes: the 8-bit writer's dither, which leaves exact n / 255 samples where they are — its largest entry is 0.492 — and shows on a fractional value only: the
// alpha of a frame whose alpha channel is coded at a lower resolution, round 6; crop, orientation)
JXL_DEV void plain_write_pixel(const DevBuffers &B, const uint8_t *stat, int out_bits, int x, int y) {
  const DevFrame &F = frame_of(B);
  const size_t po = (size_t)y * (size_t)F.pw + (size_t)x;
  plain_write_value(B, stat, *(const DevStatic *)stat, B.plane_a[0][po], B.plane_a[1][po], B.plane_a[2][po], out_bits, x, y);
}

}  // namespace jxlamd
