// jxl_coder_amd/csrc/kernels_filter.hip — HIP kernels (gfx950): Gaborish / EPF iterations, the last one fused with the XYB -> RGB -> RGBA8/16 writer.
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include <stdlib.h>
#include "kernels_common.h"
#include "dev_post.h"

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
__device__ __forceinline__ int last_filter_stage(const DevFrame &F) { return F.compose ? -1 : F.epf_iters >= 2 ? 3 : F.epf_iters == 1 ? 2 : F.gab ? 0 : -1; }     // composed frames: every stage stores its planes (dev_compose.h)
// Band decode: rows of context the stages AFTER `stage` still need around the band (EPF iteration 0 reads +-3 rows, 1: +-2, 2: +-1),
// i.e. how far beyond the band this stage has to produce output.  0 for the frame's last stage.
__device__ __forceinline__ int stage_halo_after(const DevFrame &F, int stage) {
  return (stage < 1 && F.epf_iters >= 3 ? 3 : 0) + (stage < 2 && F.epf_iters >= 1 ? 2 : 0) + (stage < 3 && F.epf_iters >= 2 ? 1 : 0);
}
__device__ __forceinline__ bool frame_uses_sweep(const DevFrame &F);
// POST (EPF iteration 2 as the last stage only — frames with three iterations, BASELINE config 5): 0 = frames whose writer stores RGBA;
// 1 = frames with DevBuffers::post: the stage's pixel goes through the colour transform into RGBA codes and from there straight through A10 (colour matrix /
// tone map) and A11 (premultiply, Bitmap format) — nothing but the Bitmap is written; 2 = the pass behind it that re-writes, un-mapped, the pixels at or
// behind their row's first zero-luma pixel (dev_post.h: post_emit).  Separate instantiations: the LUT lookups and the format switch stay out of the
// ordinary writer's registers.
#ifndef JXL_FILTER_B_WAVES
#define JXL_FILTER_B_WAVES 1
#endif
template <int STAGE, int POST = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JXL_FILTER_B_WAVES))) k_filter_b(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int sweep_on) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if ((F.is_modular && !F.xyb_modular) || !stage_runs(F, STAGE) || frame_failed(B)) return;
  if ((POST != 0) != (B.post != nullptr && STAGE == 3)) return;        // the other instantiation's frame
  if (POST == 2 && B.post->row_fz[0] == 0) return;                     // no pixel of zero luma anywhere in the frame: the first pass was final
  if (STAGE == 4 && F.compose) return;                       // composed frames have their own writer stage (k_compose_write)
  if (sweep_on && frame_uses_sweep(F)) return;                // k_filter_sweep produces this frame's pixels
  const int last = last_filter_stage(F);
  if (STAGE == 4 && last >= 0) return;                       // the writer was fused into stage `last`
  const int halo = stage_halo_after(F, STAGE);
  const int y_begin = F.band_py0 - halo > 0 ? F.band_py0 - halo : 0, y_end = F.band_py1 + halo < F.height ? F.band_py1 + halo : F.height;
  const int y = y_begin + (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (y >= y_end) return;
  // POST == 2 is launched one workgroup wide and walks its rows (a frame with a zero-luma pixel somewhere has them in few rows); the others cover the width
  if (POST == 2 && F.orientation == 1 && B.post->row_fz[1 + y + F.crop_y0] == 0xFFFFFFFFu) return;
  for (int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)); x < F.width; x += (int)gridDim.x * 64) {
  if (POST == 2) {                                            // only the pixels at or behind their row's first zero-luma pixel are computed again
    int ox, oy;
    if (!out_position(F, x, y, ox, oy) || (uint32_t)ox < B.post->row_fz[1 + oy]) continue;
  }
  const bool a = stage_src_is_a(F, STAGE);
  float *src[3], *dst[3];
  for (int c = 0; c < 3; c++) { src[c] = a ? B.plane_a[c] : B.plane_b[c]; dst[c] = a ? B.plane_b[c] : B.plane_a[c]; }
  if (STAGE == 4) { xyb_write_pixel(B, stat, *(const DevStatic *)stat, src, B.out_bits, x, y); continue; }
  float v[3];
  if (STAGE == 0) gab_value(F, src, x, y, v);
  else epf_value_p<(STAGE >= 1 && STAGE <= 3 ? STAGE - 1 : 0)>(B, F, src, x, y, v);
  if (STAGE == last) {
    // keep the filter's last multiply and the writer's first add apart (no FMA contraction across the fusion seam): the fused
    // path must give the very pixels of the stage-by-stage path (single decodes, tests/test_gpu_parity.py batch == single)
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    if (POST != 0) {
      float c[3];
      xyb_to_rgb(F, v[0], v[1], v[2], c);
      for (int k = 0; k < 3; k++) { const float lin = c[k]; c[k] = lin < 0.0f ? 0.0f : lin > 1.0f ? 1.0f : lin; if (!(lin == lin)) c[k] = 0.0f; }
      uint32_t px[4]; int ox, oy;
      if (rgba_codes(B, stat, *(const DevStatic *)stat, c, B.out_bits, x, y, px, ox, oy)) post_emit<POST == 2 ? 2 : 1>(*B.post, px[0], px[1], px[2], px[3], ox, oy, B.out_bits == 16);
    } else
    xyb_write_value(B, stat, *(const DevStatic *)stat, v[0], v[1], v[2], B.out_bits, x, y);
  }
  else for (int c = 0; c < 3; c++) dst[c][(size_t)y * (size_t)F.pw + (size_t)x] = v[c];
  }
}

// ---- Column sweep: Gaborish + EPF iteration 1 (+ iteration 2) + writer in ONE pass over the reconstructed planes, all in registers.
// A wave owns a strip of 64 - 2 HX columns (lane = column, HX halo lanes on each side) and walks down a segment of rows.  Every stage
// keeps a rolling window of its input rows in registers; horizontal neighbours come from the adjacent lanes (DPP wave shifts), vertical
// ones from the window, so each plane sample is loaded once per strip and nothing goes back to HBM but the RGBA pixels.
// The 5-pixel SADs of EPF iteration 1 are built from two difference maps (per pixel: the channel-weighted absolute difference to the
// pixel above, E_v, and to the pixel on the left, E_h): SAD_up(x, y) is the plus-shaped sum of E_v around (x, y), SAD_down(x, y) =
// SAD_up(x, y + 1), SAD_left the plus-shaped sum of E_h, SAD_right(x, y) = SAD_left(x + 1, y) — 20 operations per pixel instead of
// 120.  Coordinates outside the image are evaluated at their mirror image (virtual rows / columns hold the stage output AT the
// mirrored position, which is what the per-stage kernels read there); summation order differs from the per-stage kernels in the
// last bits only.  Frames with three EPF iterations (12-tap first pass) stay on the per-stage kernels.
__device__ __forceinline__ float dpp_left(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }    // lane - 1: x - 1
__device__ __forceinline__ float dpp_right(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }   // lane + 1: x + 1
// epf_inv_sigma (dev_recon.h) for the sweep: the frame's constants in scalar registers and the eight sharpness factors selected, not loaded — the cell's quant-field and
// sharpness bytes are the only loads (one round trip every eight rows instead of three dependent ones: frame kind, the two bytes, the factor).  Same operations, same order.
__device__ __forceinline__ float sweep_inv_sigma(const uint8_t *qfm1, const uint8_t *sharp, size_t o, float epf_quant_mul, float quant_scale,
                                                 float e0, float e1, float e2, float e3, float e4, float e5, float e6, float e7) {      // (eight values, not an array: an array argument is promoted to LDS)
  const int q = qfm1[o], sh = sharp[o];
  const float sigma_quant = epf_quant_mul / (quant_scale * (float)(q + 1) * -1.1715728752538099024f);
  float es = e0;
  es = sh == 1 ? e1 : es; es = sh == 2 ? e2 : es; es = sh == 3 ? e3 : es; es = sh == 4 ? e4 : es; es = sh == 5 ? e5 : es; es = sh == 6 ? e6 : es; es = sh == 7 ? e7 : es;
  float sigma = sigma_quant * es;
  if (sigma > -1e-4f) sigma = -1e-4f;
  return 1.0f / sigma;
}
__device__ __forceinline__ bool frame_uses_sweep(const DevFrame &F) { return !F.is_modular && !F.compose && F.epf_iters <= 2; }
__device__ __forceinline__ float sgpr_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// kFast (sweep_fast_frame): the common output — sRGB transfer curve, RGBA8, identity orientation, the frame covers the canvas, no alpha plane.  The
// general writer decides all of that per pixel (seven transfer functions, eight orientations, crop, alpha, 8 / 16 bits: ~90 branches in the loop body
// and two thirds of its instructions); here the colour constants sit in scalar registers and the pixel goes straight from the filter to one 32-bit store.
__device__ __forceinline__ bool sweep_fast_frame(const DevBuffers &B, const DevFrame &F) {
  return F.transfer == 13 && F.orientation == 1 && B.out_bits == 8 && !(F.has_ec && F.mod_out[3] >= 0) && F.crop_x0 == 0 && F.crop_y0 == 0 &&
         F.canvas_w == F.width && F.canvas_h == F.height && B.post == nullptr;
}
// kPost (SURVEY.md §8f-1 for the common path, round 5): 1 = the frame's pixels leave through A10 + A11 (post_emit: colour matrix / tone map, premultiply, Bitmap
// format) instead of the RGBA store — bit for bit what k_post_fused makes of the stored RGBA; 2 = the pass behind it that rewrites, un-mapped, the pixels at or
// behind their row's first zero-luma pixel (the reference's tone mapper stops there, Rec2408ToneMapper.cpp:91-93): same sweep, segments without such a row leave at once.
template <bool kGab, int kEpf, bool kFast, int kPost>
__device__ __forceinline__ void filter_sweep(const DevBuffers &B, const DevFrame &F, const uint8_t *stat, int strip, int seg, int rows_per_wave, int lane) {
  constexpr int DG = kGab ? 1 : 0, DE = kEpf >= 1 ? 2 : 0, DF = kEpf >= 2 ? 1 : 0;     // row delay of each stage behind its input
  constexpr int HX = DG + DE + DF, SW = 64 - 2 * HX;
  // the frame's dimensions and strides in scalar registers: read through F inside the loop they are re-loaded behind every pixel store (the compiler cannot rule out
  // that the store hits the DevFrame), and a row index mirrored against a vector-register height is a masked vector loop instead of three scalar instructions
  const int w = __builtin_amdgcn_readfirstlane(F.width), h = __builtin_amdgcn_readfirstlane(F.height);
  const int plane_w = __builtin_amdgcn_readfirstlane(F.pw), out_stride = __builtin_amdgcn_readfirstlane(F.out_w);
  const int band_py1 = __builtin_amdgcn_readfirstlane(F.band_py1);
  const int x0 = __builtin_amdgcn_readfirstlane(strip * SW);     // (strip and seg come from the wave's index: uniform)
  const int y0 = __builtin_amdgcn_readfirstlane(F.band_py0 + seg * rows_per_wave);
  if (x0 >= w || y0 >= band_py1) return;
  const int y1 = y0 + rows_per_wave < band_py1 ? y0 + rows_per_wave : band_py1;
  if (kPost == 2 && F.orientation == 1) {                        // (output row = frame row: a frame that takes this path is not cropped)
    const int yr = y0 + lane;
    if (__ballot(yr < y1 && B.post->row_fz[1 + yr] != 0xFFFFFFFFu) == 0) return;
  }
  const int x = x0 - HX + lane;                                  // virtual column of this lane
  const int ex = mirror(x < w + 8 ? x : w + 7, w);               // lanes far beyond the image never produce output
  const bool lane_out = lane >= HX && lane < 64 - HX && x < w;
  const bool lane_border = (ex & 7) == 0 || (ex & 7) == 7;
  const float *src[3] = {B.plane_a[0] + ex, B.plane_a[1] + ex, B.plane_a[2] + ex};
  const DevStatic &ST = *(const DevStatic *)stat;
  const uint16_t *rcp_lut = epf_rcp_table(F, stat);              // null: the EPF normalises with the exact quotient (wave-uniform)
  float gn[3], g1[3], g2[3], cs[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float w1 = F.gab_w[c][0], w2 = F.gab_w[c][1];
    const float norm = 1.0f / (1.0f + 4 * w1 + 4 * w2);
    gn[c] = sgpr_f(norm); g1[c] = sgpr_f(w1 * norm); g2[c] = sgpr_f(w2 * norm); cs[c] = sgpr_f(F.epf_chscale[c]);      // wave-uniform: scalar registers
  }
  const float es0 = sgpr_f(F.epf_sharp[0]), es1 = sgpr_f(F.epf_sharp[1]), es2 = sgpr_f(F.epf_sharp[2]), es3 = sgpr_f(F.epf_sharp[3]);
  const float es4 = sgpr_f(F.epf_sharp[4]), es5 = sgpr_f(F.epf_sharp[5]), es6 = sgpr_f(F.epf_sharp[6]), es7 = sgpr_f(F.epf_sharp[7]);
  const float eqm = sgpr_f(F.epf_quant_mul), qsc = sgpr_f(F.quant_scale);
  const uint8_t *cell_qf = B.qfm1 + (ex >> 3), *cell_sharp = B.sharp + (ex >> 3);      // this lane's column of cells
  const int cells_w = __builtin_amdgcn_readfirstlane(F.xb);
  const float sm1 = 1.65f, sm1b = sgpr_f(sm1 * F.epf_border_sad), sm2 = sgpr_f(1.65f * F.epf_pass2), sm2b = sgpr_f(sm2 * F.epf_border_sad);
  float in[3][3] = {}, G[4][3] = {}, Ev[3] = {}, Eh[3] = {}, E1[3][3] = {};
  float sad_up_e = 0.0f, is1 = 0.0f, is2 = 0.0f;
  int cell1 = -1, cell2 = -1;
  // the fast writer's constants (wave-uniform: scalar registers) and this lane's column of the dither table
  float obc[3], ob[3], oi[9];
  const float *dith = nullptr;
  uint32_t *orow = nullptr;
  if (kFast) {
#pragma unroll
    for (int c = 0; c < 3; c++) { obc[c] = sgpr_f(F.opsin_bias_cbrt[c]); ob[c] = sgpr_f(F.opsin_bias[c]); }
#pragma unroll
    for (int i = 0; i < 9; i++) oi[i] = sgpr_f(F.opsin_inv[i]);
    dith = st_f(stat, ST.dither_off) + (x & 31);
    orow = (uint32_t *)B.out + x;
  }
  // the next row's samples are requested one iteration ahead: the ~300 instructions of a row hide the fetch whatever else is resident on the SIMD
  float nx[3];
  {
    const size_t r0 = (size_t)mirror(y0 - HX, h) * (size_t)plane_w;
#pragma unroll
    for (int c = 0; c < 3; c++) nx[c] = src[c][r0];
  }
#ifdef JXL_SWEEP_UNROLL
#pragma unroll JXL_SWEEP_UNROLL
#endif
  for (int t = y0 - HX; t < y1 + HX; t++) {
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { in[0][c] = in[1][c]; in[1][c] = in[2][c]; in[2][c] = nx[c]; }
    if (t + 1 < y1 + HX) {
      const size_t rn = (size_t)mirror(t + 1, h) * (size_t)plane_w;
#pragma unroll
      for (int c = 0; c < 3; c++) nx[c] = src[c][rn];
    }
    // Gaborish at row g = t - DG
#pragma unroll
    for (int c = 0; c < 3; c++) {
      G[0][c] = G[1][c]; G[1][c] = G[2][c]; G[2][c] = G[3][c];
      if (kGab) {
        const float U = in[0][c], C = in[1][c], D = in[2][c];
        const float side = U + D + dpp_left(C) + dpp_right(C);
        const float diag = dpp_left(U) + dpp_right(U) + dpp_left(D) + dpp_right(D);
        G[3][c] = C * gn[c] + side * g1[c] + diag * g2[c];
      } else {
        G[3][c] = in[2][c];
      }
      v[c] = G[3][c];
    }
    if (kEpf >= 1) {
      // difference maps: E_v of row g, E_h of row g - 1
      const float evn = fabsf(G[3][0] - G[2][0]) * cs[0] + fabsf(G[3][1] - G[2][1]) * cs[1] + fabsf(G[3][2] - G[2][2]) * cs[2];
      const float ehn = fabsf(G[2][0] - dpp_left(G[2][0])) * cs[0] + fabsf(G[2][1] - dpp_left(G[2][1])) * cs[1] + fabsf(G[2][2] - dpp_left(G[2][2])) * cs[2];
      Ev[0] = Ev[1]; Ev[1] = Ev[2]; Ev[2] = evn;
      Eh[0] = Eh[1]; Eh[1] = Eh[2]; Eh[2] = ehn;
      // EPF iteration 1 at row e = g - 2: window rows G[0] = e - 1, G[1] = e, G[2] = e + 1
      const int e = t - DG - 2, ey = mirror(e, h);
      if ((ey >> 3) != cell1) { cell1 = ey >> 3; is1 = sweep_inv_sigma(cell_qf, cell_sharp, (size_t)cell1 * (size_t)cells_w, eqm, qsc, es0, es1, es2, es3, es4, es5, es6, es7); }
      const float sad_dn = Ev[0] + Ev[1] + Ev[2] + dpp_left(Ev[1]) + dpp_right(Ev[1]);     // SAD_up(e + 1)
      const float sad_lf = Eh[0] + Eh[1] + Eh[2] + dpp_left(Eh[1]) + dpp_right(Eh[1]);
      const float sad_rt = dpp_right(sad_lf);
      const bool border = (ey & 7) == 0 || (ey & 7) == 7 || lane_border;
      const float isig = is1 * (border ? sm1b : sm1);
      float wu = 1.0f + sad_up_e * isig, wl = 1.0f + sad_lf * isig, wr = 1.0f + sad_rt * isig, wd = 1.0f + sad_dn * isig;
      wu = wu < 0.0f ? 0.0f : wu; wl = wl < 0.0f ? 0.0f : wl; wr = wr < 0.0f ? 0.0f : wr; wd = wd < 0.0f ? 0.0f : wd;
      const float wsum = 1.0f + wu + wl + wr + wd;
      const float inv = rcp_lut ? rcp12_lookup(rcp_lut, wsum) : 1.0f / wsum;       // (uniform; jxlamd_decoder_set_epf_reciprocal)
      const bool skip = is1 < -3.90524291751269967465540850526868f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float acc = G[1][c];
        acc += wu * G[0][c]; acc += wl * dpp_left(G[1][c]); acc += wr * dpp_right(G[1][c]); acc += wd * G[2][c];
        v[c] = skip ? G[1][c] : acc * inv;
      }
      sad_up_e = sad_dn;
    }
    if (kEpf >= 2) {
      // EPF iteration 2 (1-pixel SAD) at row f = e - 1 on the window E1[0] = f - 1, E1[1] = f, E1[2] = f + 1
#pragma unroll
      for (int c = 0; c < 3; c++) { E1[0][c] = E1[1][c]; E1[1][c] = E1[2][c]; E1[2][c] = v[c]; }
      const int f = t - DG - 3, fy = mirror(f, h);
      if ((fy >> 3) != cell2) { cell2 = fy >> 3; is2 = sweep_inv_sigma(cell_qf, cell_sharp, (size_t)cell2 * (size_t)cells_w, eqm, qsc, es0, es1, es2, es3, es4, es5, es6, es7); }
      const bool border = (fy & 7) == 0 || (fy & 7) == 7 || lane_border;
      const float isig = is2 * (border ? sm2b : sm2);
      float tl[3], tr[3];
#pragma unroll
      for (int c = 0; c < 3; c++) { tl[c] = dpp_left(E1[1][c]); tr[c] = dpp_right(E1[1][c]); }
      float su = 0.0f, sl = 0.0f, sr = 0.0f, sd = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        su += fabsf(E1[1][c] - E1[0][c]) * cs[c]; sl += fabsf(E1[1][c] - tl[c]) * cs[c];
        sr += fabsf(E1[1][c] - tr[c]) * cs[c]; sd += fabsf(E1[1][c] - E1[2][c]) * cs[c];
      }
      float wu = 1.0f + su * isig, wl = 1.0f + sl * isig, wr = 1.0f + sr * isig, wd = 1.0f + sd * isig;
      wu = wu < 0.0f ? 0.0f : wu; wl = wl < 0.0f ? 0.0f : wl; wr = wr < 0.0f ? 0.0f : wr; wd = wd < 0.0f ? 0.0f : wd;
      const float wsum = 1.0f + wu + wl + wr + wd;
      const float inv = rcp_lut ? rcp12_lookup(rcp_lut, wsum) : 1.0f / wsum;       // (uniform; jxlamd_decoder_set_epf_reciprocal)
      const bool skip = is2 < -3.90524291751269967465540850526868f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float acc = E1[1][c];
        acc += wu * E1[0][c]; acc += wl * tl[c]; acc += wr * tr[c]; acc += wd * E1[2][c];
        v[c] = skip ? E1[1][c] : acc * inv;
      }
    }
    const int o = t - HX;                                         // the row that left the last stage
    if (o >= y0 && lane_out) {
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));      // keep the filter's last multiply and the writer's first add apart (no FMA across the seam)
      if (kFast) {
        // xyb_to_rgb + tf_srgb + the clamp + rgba_codes for the fast frame, same operations in the same order
        float mix[3];
        opsin_mix(obc, ob, v[0], v[1], v[2], mix);
        const float d = dith[(o & 31) * 32];
        uint32_t px = 0xFF000000u;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float lin = tf_srgb(opsin_lin(oi, c, mix));
          const float cv = fminf(fmaxf(lin, 0.0f), 1.0f);          // the writer's clamp, a NaN becomes 0 (fmaxf returns its other operand): two instructions instead of seven
          px |= (uint32_t)(uint8_t)(int)rintf(fmaf(cv, 255.0f, d)) << (8 * c);
        }
        orow[(size_t)o * (size_t)out_stride] = px;
      } else if (kPost != 0) {
        float c[3];
        xyb_to_rgb(F, v[0], v[1], v[2], c);
        for (int k = 0; k < 3; k++) { const float lin = c[k]; c[k] = lin < 0.0f ? 0.0f : lin > 1.0f ? 1.0f : lin; if (!(lin == lin)) c[k] = 0.0f; }
        uint32_t px[4]; int ox, oy;
        if (rgba_codes(B, stat, ST, c, B.out_bits, x, o, px, ox, oy)) post_emit<kPost == 2 ? 2 : 1>(*B.post, px[0], px[1], px[2], px[3], ox, oy, B.out_bits == 16);
      } else
      xyb_write_value(B, stat, ST, v[0], v[1], v[2], B.out_bits, x, o);
    }
  }
}
// one instantiation per stage combination: the register footprint of the longest pipeline (Gaborish + two EPF iterations) must not be
// charged to the common one (Gaborish + one iteration)
template <bool kGab, int kEpf, bool kFast, int kPost = 0>
__global__ void __launch_bounds__(256) k_filter_sweep(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int rows_per_wave, int fast_on) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (!frame_uses_sweep(F) || frame_failed(B)) return;
  if ((F.gab != 0) != kGab || F.epf_iters != kEpf || (fast_on && sweep_fast_frame(B, F)) != kFast || (B.post != nullptr) != (kPost != 0)) return;           // another instantiation's frame
  if (kPost == 2 && (B.post->row_fz[0] == 0 || !(B.post->matrix && B.post->P.tone_map))) return;      // no pixel of zero luma anywhere / nothing is tone-mapped: the first pass was final
  const int lane = (int)(threadIdx.x & 63), seg = (int)(blockIdx.y * 4 + (threadIdx.x >> 6)), strip = (int)blockIdx.x;
  filter_sweep<kGab, kEpf, kFast, kPost>(B, F, stat, strip, seg, rows_per_wave, lane);
}

void launch_filters_batch(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int nframes, int max_w, int max_h, int stage_mask, hipStream_t s) {
  const bool post = (stage_mask & 64) != 0;                      // some frame of the launch hands its pixels to the post stages (DevBuffers::post)
  // Column sweep (k_filter_sweep) for every frame with at most two EPF iterations; frames with three (stage_mask & 2: the 12-tap first
  // pass) go through the per-stage kernels, which skip the frames the sweep has produced.
  const int sweep = 1;
  {
    static const int batch_rows = [] { const char *e = getenv("JXLAMD_SWEEP_ROWS"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= 4096 ? v : 64; }();     // (measurement knob)
    const int rows = nframes == 1 ? 16 : batch_rows;           // a single decode has the chip to itself: shorter segments, more waves
    const dim3 g((max_w + 55) / 56, (max_h + 4 * rows - 1) / (4 * rows), nframes);
    // bit (gab ? 3 : 0) + epf_iters: which stage combinations the frames of this launch use; bits 6..11: the same for its fast-writer frames (kSweepFastShift)
    int combos = (stage_mask >> 8) & 63, fast = (stage_mask >> (8 + kSweepFastShift)) & 63;
    static const int fast_on = !(getenv("JXLAMD_SWEEP_FAST") && atoi(getenv("JXLAMD_SWEEP_FAST")) == 0);      // A/B switch for measurements
    if (!fast_on) { combos |= fast; fast = 0; }
    if (combos & 1) hipLaunchKernelGGL((k_filter_sweep<false, 0, false>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (combos & 2) hipLaunchKernelGGL((k_filter_sweep<false, 1, false>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (combos & 4) hipLaunchKernelGGL((k_filter_sweep<false, 2, false>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (combos & 8) hipLaunchKernelGGL((k_filter_sweep<true, 0, false>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (combos & 16) hipLaunchKernelGGL((k_filter_sweep<true, 1, false>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (combos & 32) hipLaunchKernelGGL((k_filter_sweep<true, 2, false>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (fast & 1) hipLaunchKernelGGL((k_filter_sweep<false, 0, true>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (fast & 2) hipLaunchKernelGGL((k_filter_sweep<false, 1, true>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (fast & 4) hipLaunchKernelGGL((k_filter_sweep<false, 2, true>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (fast & 8) hipLaunchKernelGGL((k_filter_sweep<true, 0, true>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (fast & 16) hipLaunchKernelGGL((k_filter_sweep<true, 1, true>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    if (fast & 32) hipLaunchKernelGGL((k_filter_sweep<true, 2, true>), g, dim3(256), 0, s, Bs, stat, rows, fast_on);
    // frames whose writer hands the pixels to the post stages (DevBuffers::post): the emitting pass, then the zero-luma repair pass
    const int pst = (stage_mask >> (8 + kSweepPostShift)) & 63;
#define JXL_SWEEP_POST(bit, G, E) if (pst & (bit)) { hipLaunchKernelGGL((k_filter_sweep<G, E, false, 1>), g, dim3(256), 0, s, Bs, stat, rows, fast_on); \
                                                        hipLaunchKernelGGL((k_filter_sweep<G, E, false, 2>), g, dim3(256), 0, s, Bs, stat, rows, fast_on); }
    JXL_SWEEP_POST(1, false, 0) JXL_SWEEP_POST(2, false, 1) JXL_SWEEP_POST(4, false, 2) JXL_SWEEP_POST(8, true, 0) JXL_SWEEP_POST(16, true, 1) JXL_SWEEP_POST(32, true, 2)
#undef JXL_SWEEP_POST
    if (!(stage_mask & (2 | 32))) return;                       // 32: a composed frame — stage by stage whatever its filters
  }
  dim3 grid((max_w + 63) / 64, (max_h + 3) / 4, nframes);
  if (stage_mask & 1) hipLaunchKernelGGL(k_filter_b<0>, grid, dim3(256), 0, s, Bs, stat, sweep);
  if (stage_mask & 2) hipLaunchKernelGGL(k_filter_b<1>, grid, dim3(256), 0, s, Bs, stat, sweep);
  if (stage_mask & 4) hipLaunchKernelGGL(k_filter_b<2>, grid, dim3(256), 0, s, Bs, stat, sweep);
  if ((stage_mask & 8) && !(stage_mask & 128)) hipLaunchKernelGGL(k_filter_b<3>, grid, dim3(256), 0, s, Bs, stat, sweep);      // 128: every frame of the launch goes through the post instantiations
  if ((stage_mask & 8) && post) {
    hipLaunchKernelGGL((k_filter_b<3, 1>), grid, dim3(256), 0, s, Bs, stat, sweep);
    hipLaunchKernelGGL((k_filter_b<3, 2>), dim3(1, grid.y, grid.z), dim3(256), 0, s, Bs, stat, sweep);
  }
  if (stage_mask & 16) hipLaunchKernelGGL(k_filter_b<4>, grid, dim3(256), 0, s, Bs, stat, sweep);
}

}  // namespace jxlamd
