// jxl_coder_amd/csrc/dev_recon.h — device-side pixel reconstruction for VarDCT frames:
// adaptive LF smoothing, dequantisation + chroma-from-luma + LLF insertion + inverse variable-size DCT,
// Gaborish, edge-preserving filter, XYB -> RGB and the RGBA writer (ISO/IEC 18181-1 Annexes I, J, K, L).
// Replaces libjxl's dec_group / render-pipeline stages ("Gab", "EPF0-2", "XYB", "FromLinear", "WriteIB") that the
// reference reaches through JxlDecoderProcessInput (jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).
//
// All functions are written per work-item (tid / pixel) so that the HIP kernels are thin index wrappers and
// the test suite can drive the identical code from a CPU loop (tests/emul, test-only).
#pragma once
#include <math.h>
#include <string.h>
#include "dev_vardct.h"

namespace jxlamd {

JXL_DEV const float *st_f(const uint8_t *st, uint32_t off) { return (const float *)(st + off); }
JXL_DEV int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }
// dequant multipliers of quant table qt, channel c: the library's (static tables) unless the frame codes its own (DevFrame::qw_frame_off: RAW tables of a
// recompressed JPEG, DCT band parameters)
JXL_DEV const float *quant_mul(const DevBuffers &B, const DevFrame &F, const uint8_t *stat, const DevStatic &ST, int qt, int c) {
  const uint32_t off = F.qw_frame_off[qt][c];
  return off ? (const float *)(B.tables + off) : st_f(stat, ST.qw_off[qt][c]);
}

// ------------------------------------------------------------------ adaptive LF smoothing (one cell)
JXL_DEV void lf_smooth_cell(const DevBuffers &B, int x, int y) {
  const DevFrame &F = frame_of(B);
  const int w = F.xb, h = F.yb;
  const size_t o = (size_t)y * (size_t)w + (size_t)x;
  if (F.skip_lf_smoothing || x == 0 || y == 0 || x == w - 1 || y == h - 1 || w < 3 || h < 3) {
    for (int c = 0; c < 3; c++) B.lf_s[c][o] = B.lf[c][o];
    return;
  }
  const float w0 = 0.05226273532324128f, w1 = 0.20345139757231578f, w2 = 0.0334829185968739f;
  float sm[3], p0[3], gap = 0.5f;
  for (int c = 0; c < 3; c++) {
    const float *p = B.lf[c] + o;
    float side = p[-1] + p[1] + p[-w] + p[w];
    float corner = p[-w - 1] + p[-w + 1] + p[w - 1] + p[w + 1];
    p0[c] = p[0];
    sm[c] = w0 * p[0] + w1 * side + w2 * corner;
    float g = fabsf((sm[c] - p[0]) / F.lf_fac[c]);
    if (g > gap) gap = g;
  }
  float factor = 3.0f - 4.0f * gap;
  if (factor < 0) factor = 0;
  for (int c = 0; c < 3; c++) B.lf_s[c][o] = (sm[c] - p0[c]) * factor + p0[c];
}

// chroma from luma: X (B) coefficient + factor * Y coefficient as ONE fused multiply-add, spelled out so that every reconstruction front end —
// dense planes or sparse lists, whatever the compiler would have contracted — rounds the same way (batch == single decode, bit for bit)
JXL_DEV float cfl_add(float v0, float k, float v1) { return fmaf(k, v1, v0); }

// ------------------------------------------------------------------ varblock reconstruction
// LDS layout for one varblock: S[3][n] dequantised coefficients (storage layout), T[n] scratch.
// Phase A (tid over n): dequant + CfL.  Phase B: LLF from LF.  Phase C/D per channel: two 1-D passes.
// only_c < 0: all three channels into S[3][n]; only_c = 0..2: that channel alone into S[n] (large blocks, LDS diet)
JXL_DEV void recon_phaseA(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *S, int n, int bx, int by,
                          int tid, int nthreads, int only_c = -1) {
  const DevFrame &F = frame_of(B);
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  const int st = B.strategy[o];
  const int qt = kQuantTableOf[st];
  const int g = (by / 32) * F.xgroups + (bx / 32);
  uint32_t off = B.coef_off[o];
  // coef_off is only written by pass 0 of a PassGroup stream that reached this block: anything else (a stream that stopped early is
  // flagged, but a stale value must still not index outside the group's 65 536-coefficient pool)
  if (off + (uint32_t)n > 65536u) { if (tid == 0) *B.err |= kErrBitstream | kErrStageRecon; off = 0; }
  const float mul = F.inv_global_scale / (float)((int)B.qfm1[o] + 1);
  const size_t to = (size_t)(by / 8) * (size_t)F.tiles_x + (size_t)(bx / 8);
  const float kx = F.base_x + (float)B.xfromy[to] * F.inv_color_factor;
  const float kb = F.base_b + (float)B.bfromy[to] * F.inv_color_factor;
  const float *qw[3] = {quant_mul(B, F, stat, ST, qt, 0), quant_mul(B, F, stat, ST, qt, 1), quant_mul(B, F, stat, ST, qt, 2)};
  for (int k = tid; k < n; k += nthreads) {
    float v[3];
    for (int c = 0; c < 3; c++) {
      if (only_c >= 0 && c != 1 && c != only_c) { v[c] = 0.0f; continue; }
      int32_t *qp = &B.coef[c][(size_t)g * 65536 + off + (uint32_t)k];
      const int q = *qp;
      // Each coefficient is consumed exactly once: the last reader clears it, which leaves the frame slot's coefficient
      // planes all-zero for the next frame (no 106 MB memset per 4K frame).  With only_c the Y channel is read by all
      // three per-channel rounds; the last one (only_c == 2) clears it.
      if (q != 0 && (only_c < 0 || (c != 1 && c == only_c) || (c == 1 && only_c == 2))) *qp = 0;
      float a;
      if (q == 0) a = 0.0f;
      else if (q == 1) a = F.quant_bias[c];
      else if (q == -1) a = -F.quant_bias[c];
      else a = (float)q - F.quant_bias[3] / (float)q;
      v[c] = a * (mul * F.dm[c] * qw[c][k]);
    }
    if (only_c >= 0) { S[k] = only_c == 0 ? cfl_add(v[0], kx, v[1]) : only_c == 1 ? v[1] : cfl_add(v[2], kb, v[1]); continue; }
    S[k] = cfl_add(v[0], kx, v[1]);
    S[n + k] = v[1];
    S[2 * n + k] = cfl_add(v[2], kb, v[1]);
  }
}

// One entry of a sparse coefficient list, dequantised: the value recon_phaseA computes for (channel, position) from the dense planes
JXL_DEV float sp_dequant(const DevFrame &F, uint32_t e, float mul_dm_c, const float *qw_c) {
  const int q = sp_val(e);
  const int c = sp_chan(e);
  float a;
  if (q == 1) a = F.quant_bias[c];
  else if (q == -1) a = -F.quant_bias[c];
  else a = (float)q - F.quant_bias[3] / (float)q;
  return a * (mul_dm_c * qw_c[sp_pos(e)]);
}
// Phase A from a varblock's sparse list (DevBuffers::coef_sp, flights): clear the tile, scatter the X and B entries, then the Y entries —
// each one also adds its chroma-from-luma share to X and B at its position, which is why they come last: cfl_add(X, kx, Y) needs X in place.
// Same values as recon_phaseA (a zero coefficient dequantises to +0 and contributes nothing).  Contains barriers: every work-item calls it.
template <class Sync>
JXL_DEV void recon_phaseA_sparse(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *S, int n, int bx, int by,
                                 int tid, int nthreads, Sync sync, int only_c = -1, int st_known = -1) {
  const DevFrame &F = frame_of(B);
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  // st_known: the walkers of one strategy pass it — the strategy map → quant-table index → weight-table offsets are three dependent loads in front of every block otherwise
  const int st = st_known >= 0 ? st_known : B.strategy[o];
  const int qt = kQuantTableOf[st];
  const uint32_t *ent = B.coef_sp + B.coef_off[o];
  uint32_t cnt = B.coef_cnt[o];
  if (cnt > 3u * (uint32_t)n) { if (tid == 0) *B.err |= kErrBitstream | kErrStageRecon; cnt = 0; }      // stale record (a stream that stopped early is flagged anyway)
  const float mul = F.inv_global_scale / (float)((int)B.qfm1[o] + 1);
  const size_t to = (size_t)(by / 8) * (size_t)F.tiles_x + (size_t)(bx / 8);
  const float kx = F.base_x + (float)B.xfromy[to] * F.inv_color_factor;
  const float kb = F.base_b + (float)B.bfromy[to] * F.inv_color_factor;
  const float *qw[3] = {quant_mul(B, F, stat, ST, qt, 0), quant_mul(B, F, stat, ST, qt, 1), quant_mul(B, F, stat, ST, qt, 2)};
  const int total = only_c >= 0 ? n : 3 * n;
  for (int k = tid; k < total; k += nthreads) S[k] = 0.0f;
  sync();
  // The Y entries wait for the X and B ones to be in place (second loop), but their values do not: a work-item's first two Y entries are dequantised in the first
  // loop, next to the others, and kept in registers — the second loop then has nothing to fetch for them (it used to read every entry and its weight again: two more
  // dependent round trips per varblock in front of the transform).  Entries beyond a work-item's first two Y ones take the old way.
  int yk0 = -1, yk1 = -1, ny = 0; float yv0 = 0.0f, yv1 = 0.0f;
  for (uint32_t i = (uint32_t)tid; i < cnt; i += (uint32_t)nthreads) {
    const uint32_t e = ent[i];
    const int c = sp_chan(e);
    if (c > 2 || sp_pos(e) >= (uint32_t)n) continue;
    if (c == 1) {
      if (ny < 2) { const float v1 = sp_dequant(F, e, mul * F.dm[1], qw[1]); if (ny == 0) { yk0 = (int)sp_pos(e); yv0 = v1; } else { yk1 = (int)sp_pos(e); yv1 = v1; } }
      ny++;
      continue;
    }
    if (only_c >= 0 && c != only_c) continue;
    S[(only_c >= 0 ? 0 : c * n) + (int)sp_pos(e)] = sp_dequant(F, e, mul * F.dm[c], qw[c]);
  }
  sync();
  const auto put_y = [&](int k, float v1) {
    if (only_c < 0) { S[k] = cfl_add(S[k], kx, v1); S[n + k] = v1; S[2 * n + k] = cfl_add(S[2 * n + k], kb, v1); }
    else S[k] = only_c == 0 ? cfl_add(S[k], kx, v1) : only_c == 1 ? v1 : cfl_add(S[k], kb, v1);
  };
  if (yk0 >= 0) put_y(yk0, yv0);
  if (yk1 >= 0) put_y(yk1, yv1);
  if (ny > 2) {
    int seen = 0;
    for (uint32_t i = (uint32_t)tid; i < cnt; i += (uint32_t)nthreads) {
      const uint32_t e = ent[i];
      if (sp_chan(e) != 1 || sp_pos(e) >= (uint32_t)n) continue;
      if (seen++ < 2) continue;                        // placed above
      put_y((int)sp_pos(e), sp_dequant(F, e, mul * F.dm[1], qw[1]));
    }
  }
}

JXL_DEV void recon_phaseB(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *S, int n, int bx, int by,
                          int tid, int nthreads, int only_c = -1) {
  const DevFrame &F = frame_of(B);
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  const int st = B.strategy[o];
  const int cx = kCoveredX[st], cy = kCoveredY[st];
  const int srows = cy < cx ? cy : cx, scols = cy < cx ? cx : cy;
  const int total = (only_c >= 0 ? 1 : 3) * cx * cy;
  const float *ccx = st_f(stat, ST.cos_off[ilog2(cx)]), *ccy = st_f(stat, ST.cos_off[ilog2(cy)]);
  const float *lsx = st_f(stat, ST.llf_off) + 32 * ilog2(cx), *lsy = st_f(stat, ST.llf_off) + 32 * ilog2(cy);
  for (int i = tid; i < total; i += nthreads) {
    const int ci = i / (cx * cy), r = i - ci * cx * cy;
    const int c = only_c >= 0 ? only_c : ci;
    const int a = r / scols, b = r - a * scols;       // storage position
    const int u = cy >= cx ? a : b, v = cy >= cx ? b : a;   // horizontal / vertical frequency
    float s = 0.0f;
    for (int iy = 0; iy < cy; iy++) {
      float rs = 0.0f;
      for (int ix = 0; ix < cx; ix++) rs += B.lf_s[c][o + (size_t)iy * (size_t)F.xb + (size_t)ix] * ccx[u * cx + ix];
      s += rs * ccy[v * cy + iy];
    }
    s *= (1.0f / (float)(cx * cy)) * lsx[u] * lsy[v];
    S[ci * n + a * scols * 8 + b] = s;
  }
  (void)srows;
}

// generic separable inverse DCT: pass 1 (rows of M -> T), pass 2 (columns -> pixels)
JXL_DEV void recon_idct_pass1(const uint8_t *stat, const DevStatic &ST, const float *Sc, float *T, int R, int C, int tid, int nthreads) {
  const float *cc = st_f(stat, ST.cos_off[ilog2(C)]);
  for (int i = tid; i < R * C; i += nthreads) {
    const int v = i / C, x = i - v * C;
    float s = 0.0f;
    if (R < C) for (int u = 0; u < C; u++) s += Sc[v * C + u] * cc[u * C + x];
    else for (int u = 0; u < C; u++) s += Sc[u * R + v] * cc[u * C + x];
    T[i] = s;
  }
}
JXL_DEV void recon_idct_pass2(const uint8_t *stat, const DevStatic &ST, const float *T, float *out, int ostride, int R, int C, int tid, int nthreads) {
  const float *cr = st_f(stat, ST.cos_off[ilog2(R)]);
  for (int i = tid; i < R * C; i += nthreads) {
    const int y = i / C, x = i - y * C;
    float s = 0.0f;
    for (int v = 0; v < R; v++) s += T[v * C + x] * cr[v * R + y];
    out[(size_t)y * (size_t)ostride + (size_t)x] = s;
  }
}

// small dense helpers for the 8x8 "special" strategies (run by one lane per channel)
JXL_DEV void small_idct(const uint8_t *stat, const DevStatic &ST, const float *S, int R, int C, float *out, int ostride) {
  const float *cr = st_f(stat, ST.cos_off[ilog2(R)]), *cc = st_f(stat, ST.cos_off[ilog2(C)]);
  float tmp[32];
  for (int v = 0; v < R; v++)
    for (int x = 0; x < C; x++) {
      float s = 0.0f;
      for (int u = 0; u < C; u++) s += (R < C ? S[v * C + u] : S[u * R + v]) * cc[u * C + x];
      tmp[v * C + x] = s;
    }
  for (int y = 0; y < R; y++)
    for (int x = 0; x < C; x++) {
      float s = 0.0f;
      for (int v = 0; v < R; v++) s += tmp[v * C + x] * cr[v * R + y];
      out[(size_t)y * (size_t)ostride + (size_t)x] = s;
    }
}

JXL_DEV void recon_special(const uint8_t *stat, const DevStatic &ST, int st, const float *S, float *out, int ostride) {
  if (st == 1) {   // IDENTITY
    float dcs[4];
    float b00 = S[0], b01 = S[1], b10 = S[8], b11 = S[9];
    dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11; dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
    for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
      float block_dc = dcs[y * 2 + x], rs = 0.0f;
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (!ix && !iy) continue; rs += S[(y + iy * 2) * 8 + x + ix * 2]; }
      float v11 = block_dc - rs * (1.0f / 16);
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) {
        if (ix == 1 && iy == 1) continue;
        out[(size_t)(y * 4 + iy) * ostride + x * 4 + ix] = S[(y + iy * 2) * 8 + x + ix * 2] + v11;
      }
      out[(size_t)(4 * y + 1) * ostride + 4 * x + 1] = v11;
      out[(size_t)(y * 4) * ostride + x * 4] = S[(y + 2) * 8 + x + 2] + v11;
    }
  } else if (st == 2) {   // DCT2X2
    float a[64], b[64];
    for (int i = 0; i < 64; i++) a[i] = S[i];
    for (int sz = 2; sz <= 8; sz *= 2) {
      int n2 = sz / 2;
      for (int i = 0; i < 64; i++) b[i] = a[i];
      for (int y = 0; y < n2; y++) for (int x = 0; x < n2; x++) {
        float c00 = a[y * 8 + x], c01 = a[y * 8 + n2 + x], c10 = a[(y + n2) * 8 + x], c11 = a[(y + n2) * 8 + n2 + x];
        b[y * 2 * 8 + x * 2] = c00 + c01 + c10 + c11;
        b[y * 2 * 8 + x * 2 + 1] = c00 + c01 - c10 - c11;
        b[(y * 2 + 1) * 8 + x * 2] = c00 - c01 + c10 - c11;
        b[(y * 2 + 1) * 8 + x * 2 + 1] = c00 - c01 - c10 + c11;
      }
      for (int i = 0; i < 64; i++) a[i] = b[i];
    }
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[(size_t)y * ostride + x] = a[y * 8 + x];
  } else if (st == 3) {   // DCT4X4
    float dcs[4];
    float b00 = S[0], b01 = S[1], b10 = S[8], b11 = S[9];
    dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11; dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
    for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
      float blk[16];
      blk[0] = dcs[y * 2 + x];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (!ix && !iy) continue; blk[iy * 4 + ix] = S[(y + iy * 2) * 8 + x + ix * 2]; }
      small_idct(stat, ST, blk, 4, 4, out + (size_t)(y * 4) * ostride + x * 4, ostride);
    }
  } else if (st == 12 || st == 13) {   // DCT4X8 / DCT8X4
    float dcs[2] = {S[0] + S[8], S[0] - S[8]};
    for (int k = 0; k < 2; k++) {
      float blk[32];
      blk[0] = dcs[k];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) { if (!ix && !iy) continue; blk[iy * 8 + ix] = S[(k + iy * 2) * 8 + ix]; }
      if (st == 12) small_idct(stat, ST, blk, 4, 8, out + (size_t)(k * 4) * ostride, ostride);
      else small_idct(stat, ST, blk, 8, 4, out + k * 4, ostride);
    }
  } else {   // AFV0..3
    const int kind = st - 14, afv_x = kind & 1, afv_y = kind / 2;
    const float *basis = st_f(stat, ST.afv_off);
    float dcs[3];
    float b00 = S[0], b01 = S[1], b10 = S[8];
    dcs[0] = (b00 + b10 + b01) * 4.0f; dcs[1] = (b00 + b10 - b01); dcs[2] = b00 - b10;
    float coeff[16], blk[32];
    coeff[0] = dcs[0];
    for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (!ix && !iy) continue; coeff[iy * 4 + ix] = S[iy * 2 * 8 + ix * 2]; }
    for (int i = 0; i < 16; i++) { float s = 0.0f; for (int j = 0; j < 16; j++) s += coeff[j] * basis[j * 16 + i]; blk[i] = s; }
    for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++)
      out[(size_t)(iy + afv_y * 4) * ostride + afv_x * 4 + ix] = blk[(afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix)];
    float b2[32];
    b2[0] = dcs[1];
    for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (!ix && !iy) continue; b2[iy * 4 + ix] = S[iy * 2 * 8 + ix * 2 + 1]; }
    small_idct(stat, ST, b2, 4, 4, out + (size_t)(afv_y * 4) * ostride + (afv_x == 1 ? 0 : 4), ostride);
    b2[0] = dcs[2];
    for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) { if (!ix && !iy) continue; b2[iy * 8 + ix] = S[(1 + iy * 2) * 8 + ix]; }
    small_idct(stat, ST, b2, 4, 8, out + (size_t)((afv_y == 1 ? 0 : 4)) * ostride, ostride);
  }
}

JXL_DEV bool strategy_is_special(int st) { return st == 1 || st == 2 || st == 3 || (st >= 12 && st <= 17); }

// ------------------------------------------------------------------ loop filters (per pixel)
JXL_DEV int mirror(int x, int n) {
  while (x < 0 || x >= n) { if (x < 0) x = -x - 1; else x = 2 * n - 1 - x; }
  return x;
}

// the filtered XYB value of one pixel (the *_pixel wrappers store it into the destination planes; the fused last stage of a
// flight hands it straight to the writer)
JXL_DEV void gab_value(const DevFrame &F, float *const src[3], int x, int y, float out[3]) {
  const int w = F.width, h = F.height, pw = F.pw;
  const int ym = mirror(y - 1, h), yp = mirror(y + 1, h), xm = mirror(x - 1, w), xp = mirror(x + 1, w);
  for (int c = 0; c < 3; c++) {
    const float w1 = F.gab_w[c][0], w2 = F.gab_w[c][1];
    const float norm = 1.0f / (1.0f + 4 * w1 + 4 * w2);
    const float *p = src[c];
    #define P(yy, xx) p[(size_t)(yy) * (size_t)pw + (size_t)(xx)]
    float side = P(ym, x) + P(yp, x) + P(y, xm) + P(y, xp);
    float diag = P(ym, xm) + P(ym, xp) + P(yp, xm) + P(yp, xp);
    out[c] = P(y, x) * norm + side * (w1 * norm) + diag * (w2 * norm);
    #undef P
  }
}
JXL_DEV void gab_pixel(const DevFrame &F, float *const src[3], float *const dst[3], int x, int y) {
  float v[3];
  gab_value(F, src, x, y, v);
  for (int c = 0; c < 3; c++) dst[c][(size_t)y * (size_t)F.pw + (size_t)x] = v[c];
}

// 1 / (sum of weights).  libjxl spells it ApproximateReciprocal: exact in its AVX2 / NEON-with-refinement builds to within float rounding, but in the reference's
// SSE2-only x86_64 build the host CPU's 12-bit rcpps, biased low by up to 3e-4 relative — invisible in most pixels, up to 4 codes where the inverse opsin matrix
// amplifies it (one channel near 0 beside two near 1 on hard edges).  Default: the exact quotient.  epf_rcp_x86: the golden host's instruction as a table
// (rcp12_lut.h: a function of the operand's top 11 mantissa bits, scaled exactly by its exponent), for "what that reference build returned" bit for bit.
JXL_DEV float rcp12_lookup(const uint16_t *lut, float wsum) {
  uint32_t u; memcpy(&u, &wsum, 4);                     // wsum is in [1, 13]: a normal positive number
  uint32_t r = 0x3f000000u + ((uint32_t)lut[(u >> 12) & 2047u] << 11) - (((u >> 23) - 127u) << 23);
  float inv; memcpy(&inv, &r, 4);
  return inv;
}
// the table when the frame asks for it, else null (the column sweep resolves this once per launch)
JXL_DEV const uint16_t *epf_rcp_table(const DevFrame &F, const uint8_t *stat) { return F.epf_rcp_x86 ? (const uint16_t *)(stat + ((const DevStatic *)stat)->rcp12_off) : nullptr; }
JXL_DEV float epf_reciprocal(const DevBuffers &B, const DevFrame &F, float wsum) {
  if (!F.epf_rcp_x86) return 1.0f / wsum;
  return rcp12_lookup(epf_rcp_table(F, B.stat), wsum);
}

JXL_DEV float epf_inv_sigma(const DevBuffers &B, const DevFrame &F, int x, int y) {
  if (F.is_modular) return -1.1715728752538099024f / F.epf_sigma_modular;      // Modular-encoded XYB frame: one sigma, no quant field / sharpness map
  const size_t o = (size_t)(y >> 3) * (size_t)F.xb + (size_t)(x >> 3);
  float sigma_quant = F.epf_quant_mul / (F.quant_scale * (float)((int)B.qfm1[o] + 1) * -1.1715728752538099024f);
  float sigma = sigma_quant * F.epf_sharp[B.sharp[o]];
  if (sigma > -1e-4f) sigma = -1e-4f;
  return 1.0f / sigma;
}

// kInterior: every tap lies inside the image (x, y at least 3 pixels from each edge): plain indexing, which also lets the
// compiler share the overlapping loads of the 5-pixel SAD patterns; the border version mirrors every coordinate.
// kPass: 0 = the 12-tap first iteration, 1 = 4 taps with the 5-pixel SAD, 2 = 4 taps with the 1-pixel SAD.
template <bool kInterior, int kPass>
JXL_DEV void epf_value_t(const DevBuffers &B, const DevFrame &F, float *const src[3], int x, int y, float out[3]) {
  const int w = F.width, h = F.height, pw = F.pw;
  const size_t po = (size_t)y * (size_t)pw + (size_t)x;
  const float is = epf_inv_sigma(B, F, x, y);
  if (is < -3.90524291751269967465540850526868f) { for (int c = 0; c < 3; c++) out[c] = src[c][po]; return; }
  const float sm = 1.65f * (kPass == 0 ? F.epf_pass0 : kPass == 2 ? F.epf_pass2 : 1.0f);
  const bool border = ((y & 7) == 0 || (y & 7) == 7 || (x & 7) == 0 || (x & 7) == 7);
  const float isig = is * (border ? sm * F.epf_border_sad : sm);
  constexpr int px[5] = {0, 0, -1, 1, 0}, py[5] = {0, -1, 0, 0, 1};
  constexpr int t0x[12] = {0, -1, 0, 1, -2, -1, 1, 2, -1, 0, 1, 0}, t0y[12] = {-2, -1, -1, -1, 0, 0, 0, 0, 1, 1, 1, 2};
  constexpr int t1x[4] = {0, -1, 1, 0}, t1y[4] = {-1, 0, 0, 1};
  constexpr int ntaps = kPass == 0 ? 12 : 4;
  #define PX(c, dy, dx) (kInterior ? src[c][(int64_t)po + (int64_t)(dy) * (int64_t)pw + (int64_t)(dx)] \
                                   : src[c][(size_t)mirror(y + (dy), h) * (size_t)pw + (size_t)mirror(x + (dx), w)])
  float wsum = 1.0f, acc[3];
  for (int c = 0; c < 3; c++) acc[c] = src[c][po];
  if (kPass == 0) {
    // The 12-tap iteration, channel by channel (round 6): one channel's 25-sample neighbourhood is live at a time instead of all three (k_filter_b<1>: 178 VGPRs ->
    // what lets more of its waves sit between the resident entropy waves; BASELINE config 5 spends a fifth of its kernel time here).  Every sum keeps its order — a
    // tap's SAD over the channels 0, 1, 2, the weights over the taps, a channel's accumulator over the taps — so the values are the very same.
    float sadv[12], wgtv[12];
#ifdef __HIPCC__
    #pragma unroll
#endif
    for (int t = 0; t < 12; t++) sadv[t] = 0.0f;
    if (kInterior) {
#ifdef __HIPCC__
      #pragma unroll 1
#endif
      for (int c = 0; c < 3; c++) {
        const float cs = F.epf_chscale[c];
#ifdef __HIPCC__
        #pragma unroll
#endif
        for (int t = 0; t < 12; t++) {
          float sc = 0.0f;
#ifdef __HIPCC__
          #pragma unroll
#endif
          for (int k = 0; k < 5; k++) sc += fabsf(PX(c, py[k], px[k]) - PX(c, t0y[t] + py[k], t0x[t] + px[k]));
          sadv[t] += sc * cs;
        }
      }
    } else {
      // pixels within three of an edge (every coordinate mirrored): few, and their 75 mirrored addresses are what the kernel's register count used to be sized by —
      // rolled loops here
#ifdef __HIPCC__
      #pragma unroll 1
#endif
      for (int c = 0; c < 3; c++) {
        const float cs = F.epf_chscale[c];
#ifdef __HIPCC__
        #pragma unroll 1
#endif
        for (int t = 0; t < 12; t++) {
          float sc = 0.0f;
#ifdef __HIPCC__
          #pragma unroll 1
#endif
          for (int k = 0; k < 5; k++) sc += fabsf(PX(c, py[k], px[k]) - PX(c, t0y[t] + py[k], t0x[t] + px[k]));
          sadv[t] += sc * cs;
        }
      }
    }
#ifdef __HIPCC__
    #pragma unroll
#endif
    for (int t = 0; t < 12; t++) { float wgt = 1.0f + sadv[t] * isig; if (wgt < 0.0f) wgt = 0.0f; wgtv[t] = wgt; wsum += wgt; }
#ifdef __HIPCC__
    #pragma unroll 1
#endif
    for (int c = 0; c < 3; c++) {
      float a = acc[c];
#ifdef __HIPCC__
      #pragma unroll
#endif
      for (int t = 0; t < 12; t++) a += wgtv[t] * PX(c, t0y[t], t0x[t]);
      acc[c] = a;
    }
  } else
#ifdef __HIPCC__
  #pragma unroll
#endif
  for (int t = 0; t < ntaps; t++) {
    const int tx = kPass == 0 ? t0x[t] : t1x[t], ty = kPass == 0 ? t0y[t] : t1y[t];
    float sad = 0.0f;
    if (kPass == 2) {
      for (int c = 0; c < 3; c++) sad += fabsf(src[c][po] - PX(c, ty, tx)) * F.epf_chscale[c];
    } else {
      for (int c = 0; c < 3; c++) {
        float sc = 0.0f;
#ifdef __HIPCC__
        #pragma unroll
#endif
        for (int k = 0; k < 5; k++) sc += fabsf(PX(c, py[k], px[k]) - PX(c, ty + py[k], tx + px[k]));
        sad += sc * F.epf_chscale[c];
      }
    }
    float wgt = 1.0f + sad * isig;
    if (wgt < 0.0f) wgt = 0.0f;
    wsum += wgt;
    for (int c = 0; c < 3; c++) acc[c] += wgt * PX(c, ty, tx);
  }
  #undef PX
  const float inv = epf_reciprocal(B, F, wsum);
  for (int c = 0; c < 3; c++) out[c] = acc[c] * inv;
}
template <int kPass>
JXL_DEV void epf_value_p(const DevBuffers &B, const DevFrame &F, float *const src[3], int x, int y, float out[3]) {
  if (x >= 3 && y >= 3 && x + 3 < F.width && y + 3 < F.height) epf_value_t<true, kPass>(B, F, src, x, y, out);
  else epf_value_t<false, kPass>(B, F, src, x, y, out);
}
template <int kPass>
JXL_DEV void epf_pixel_p(const DevBuffers &B, const DevFrame &F, float *const src[3], float *const dst[3], int x, int y) {
  float v[3];
  epf_value_p<kPass>(B, F, src, x, y, v);
  for (int c = 0; c < 3; c++) dst[c][(size_t)y * (size_t)F.pw + (size_t)x] = v[c];
}
JXL_DEV void epf_pixel(const DevBuffers &B, const DevFrame &F, float *const src[3], float *const dst[3], int pass, int x, int y) {
  if (pass == 0) epf_pixel_p<0>(B, F, src, dst, x, y);
  else if (pass == 1) epf_pixel_p<1>(B, F, src, dst, x, y);
  else epf_pixel_p<2>(B, F, src, dst, x, y);
}

// ------------------------------------------------------------------ the same filters over an accessor (fused LDS-tiled kernel)
// acc(c, y, x) returns channel c at FRAME coordinates (y, x), which may lie up to 7 pixels outside the image: the accessor
// hands back the mirrored sample.  Same expression order as gab_value / epf_value_t.
template <class Acc>
JXL_DEV void gab_value_acc(const DevFrame &F, const Acc &acc, int x, int y, float out[3]) {
  for (int c = 0; c < 3; c++) {
    const float w1 = F.gab_w[c][0], w2 = F.gab_w[c][1];
    const float norm = 1.0f / (1.0f + 4 * w1 + 4 * w2);
    float side = acc(c, y - 1, x) + acc(c, y + 1, x) + acc(c, y, x - 1) + acc(c, y, x + 1);
    float diag = acc(c, y - 1, x - 1) + acc(c, y - 1, x + 1) + acc(c, y + 1, x - 1) + acc(c, y + 1, x + 1);
    out[c] = acc(c, y, x) * norm + side * (w1 * norm) + diag * (w2 * norm);
  }
}
// (x, y): the pixel whose sigma / border class applies — inside the image; the taps go through acc
template <int kPass, class Acc>
JXL_DEV void epf_value_acc(const DevBuffers &B, const DevFrame &F, const Acc &acc, int x, int y, float out[3]) {
  const float is = epf_inv_sigma(B, F, x, y);
  if (is < -3.90524291751269967465540850526868f) { for (int c = 0; c < 3; c++) out[c] = acc(c, y, x); return; }
  const float sm = 1.65f * (kPass == 0 ? F.epf_pass0 : kPass == 2 ? F.epf_pass2 : 1.0f);
  const bool border = ((y & 7) == 0 || (y & 7) == 7 || (x & 7) == 0 || (x & 7) == 7);
  const float isig = is * (border ? sm * F.epf_border_sad : sm);
  constexpr int px[5] = {0, 0, -1, 1, 0}, py[5] = {0, -1, 0, 0, 1};
  constexpr int t0x[12] = {0, -1, 0, 1, -2, -1, 1, 2, -1, 0, 1, 0}, t0y[12] = {-2, -1, -1, -1, 0, 0, 0, 0, 1, 1, 1, 2};
  constexpr int t1x[4] = {0, -1, 1, 0}, t1y[4] = {-1, 0, 0, 1};
  constexpr int ntaps = kPass == 0 ? 12 : 4;
  float wsum = 1.0f, accv[3];
  for (int c = 0; c < 3; c++) accv[c] = acc(c, y, x);
#ifdef __HIPCC__
  #pragma unroll
#endif
  for (int t = 0; t < ntaps; t++) {
    const int tx = kPass == 0 ? t0x[t] : t1x[t], ty = kPass == 0 ? t0y[t] : t1y[t];
    float sad = 0.0f;
    if (kPass == 2) {
      for (int c = 0; c < 3; c++) sad += fabsf(acc(c, y, x) - acc(c, y + ty, x + tx)) * F.epf_chscale[c];
    } else {
      for (int c = 0; c < 3; c++) {
        float sc = 0.0f;
#ifdef __HIPCC__
        #pragma unroll
#endif
        for (int k = 0; k < 5; k++) sc += fabsf(acc(c, y + py[k], x + px[k]) - acc(c, y + ty + py[k], x + tx + px[k]));
        sad += sc * F.epf_chscale[c];
      }
    }
    float wgt = 1.0f + sad * isig;
    if (wgt < 0.0f) wgt = 0.0f;
    wsum += wgt;
    for (int c = 0; c < 3; c++) accv[c] += wgt * acc(c, y + ty, x + tx);
  }
  const float inv = epf_reciprocal(B, F, wsum);
  for (int c = 0; c < 3; c++) out[c] = accv[c] * inv;
}

// ------------------------------------------------------------------ XYB -> RGB -> RGBA writer (one pixel)
// a^e for a >= 0: on the GPU two transcendental instructions (v_log_f32 / v_exp_f32, ~1 ulp each: the result moves
// by < 1e-6 relative, far below half an 8-bit or 16-bit step); the CPU harness uses libm.
JXL_DEV float pow_pos(float a, float e) {
#ifdef __HIPCC__
  return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(a));
#else
  return powf(a, e);
#endif
}
JXL_DEV float tf_srgb(float v) {
  // negative (out-of-gamut) input: libjxl 0.12's sRGB curve has no sign handling — everything at or below the threshold takes the linear segment, so a
  // negative value is 12.92 v (established on the reference binary's float output of a blended animation; it only shows where values are not clamped
  // straight away, i.e. in the background a later frame is blended over)
  if (v <= 0.0031308f) return 12.92f * v;
  return fmaf(1.055f, pow_pos(v, 1.0f / 2.4f), -0.055f);      // (spelled out: every writer instantiation must round alike)
}
JXL_DEV float tf_pq(float v, float intensity_target) {
  float a = fabsf(v) * (intensity_target * 1e-4f);
  const float m1 = 2610.0f / 16384, m2 = 2523.0f / 4096 * 128, c1 = 3424.0f / 4096, c2 = 2413.0f / 4096 * 32, c3 = 2392.0f / 4096 * 32;
  float p = pow_pos(a, m1);
  float r = pow_pos((c1 + c2 * p) / (1 + c3 * p), m2);
  return v < 0 ? -r : r;
}
JXL_DEV float tf_hlg(float v) {            // ARIB STD-B67 OETF on [0, 1] (after the inverse OOTF)
  float a = fabsf(v);
#ifdef __HIPCC__
  float r = a <= (1.0f / 12.0f) ? sqrtf(3.0f * a) : 0.17883277f * (__builtin_amdgcn_logf(12.0f * a - 0.28466892f) * 0.69314718056f) + 0.5599107295f;
#else
  float r = a <= (1.0f / 12.0f) ? sqrtf(3.0f * a) : 0.17883277f * logf(12.0f * a - 0.28466892f) + 0.5599107295f;
#endif
  return v < 0 ? -r : r;
}
JXL_DEV float tf_709(float v) {
  float a = fabsf(v);
  float r = a < 0.018f ? 4.5f * a : 1.099f * pow_pos(a, 0.45f) - 0.099f;
  return v < 0 ? -r : r;
}

// the writer's last step: three colour values in [0, 1] of frame pixel (x, y) -> canvas position, orientation, alpha, dither, RGBA8 / RGBA16
// output position of frame pixel (x, y): canvas offset, then the image's orientation; false when it falls outside the canvas
JXL_DEV bool out_position(const DevFrame &F, int x, int y, int &ox, int &oy) {
  x += F.crop_x0; y += F.crop_y0;
  const int w = F.canvas_w, h = F.canvas_h;
  if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return false;
  ox = x; oy = y;
  switch (F.orientation) {
    case 2: ox = w - 1 - x; break;
    case 3: ox = w - 1 - x; oy = h - 1 - y; break;
    case 4: oy = h - 1 - y; break;
    case 5: ox = y; oy = x; break;
    case 6: ox = h - 1 - y; oy = x; break;
    case 7: ox = h - 1 - y; oy = w - 1 - x; break;
    case 8: ox = y; oy = w - 1 - x; break;
    default: break;
  }
  return true;
}
// ... as integer codes: px[4] (8- or 16-bit) and the output position; false when the pixel falls outside the canvas
JXL_DEV bool rgba_codes(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, const float (&v)[3], int out_bits, int x, int y, uint32_t (&px)[4], int &ox, int &oy) {
  const DevFrame &F = frame_of(B);
  const int fx = x, fy = y;                            // frame position (the alpha plane below is read there)
  x += F.crop_x0; y += F.crop_y0;                      // canvas position
  const int w = F.canvas_w, h = F.canvas_h;
  if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return false;
  ox = x; oy = y;
  switch (F.orientation) {
    case 2: ox = w - 1 - x; break;
    case 3: ox = w - 1 - x; oy = h - 1 - y; break;
    case 4: oy = h - 1 - y; break;
    case 5: ox = y; oy = x; break;
    case 6: ox = h - 1 - y; oy = x; break;
    case 7: ox = h - 1 - y; oy = w - 1 - x; break;
    case 8: ox = y; oy = w - 1 - x; break;
    default: break;
  }
  float alpha = 1.0f;                                  // extra channel of type alpha (Modular-coded, integer samples)
  if ((F.has_ec || F.is_modular) && F.mod_out[3] >= 0) {
    if (F.alpha_up > 1) alpha = B.up[3][(size_t)fy * (size_t)F.full_w + (size_t)fx];       // enlarged beforehand (upsample_alpha_pixel); fx, fy are full-resolution here
    else {
      const int32_t av = (B.mod_pool + F.mod_plane_off[F.mod_out[3]])[(size_t)fy * (size_t)F.width + (size_t)fx];
      alpha = alpha_sample_value(F, av, false);
    }
    alpha = alpha < 0.0f ? 0.0f : alpha > 1.0f ? 1.0f : alpha;
  }
  if (out_bits == 8) {
    // libjxl's 8-bit writer dither (oracle/README.md): indexed by the OUTPUT position; for the transposing orientations (5..8) with row
    // and column swapped (established on the reference's output: tests/golden/vo72x40_e3_o5..8)
    const float d = st_f(stat, ST.dither_off)[F.orientation > 4 ? (ox & 31) * 32 + (oy & 31) : (oy & 31) * 32 + (ox & 31)];
    for (int c = 0; c < 3; c++) px[c] = (uint32_t)(uint8_t)(int)rintf(fmaf(v[c], 255.0f, d));
    px[3] = (uint32_t)(uint8_t)(int)rintf(fmaf(alpha, 255.0f, d));        // the dither goes on every channel; it only shows on a fractional (upsampled) alpha: |d| < 0.5
  } else {
    for (int c = 0; c < 3; c++) px[c] = (uint32_t)(uint16_t)(int)rintf(v[c] * 65535.0f);
    px[3] = (uint32_t)(uint16_t)(int)rintf(alpha * 65535.0f);
  }
  return true;
}
JXL_DEV void rgba_store(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, const float (&v)[3], int out_bits, int x, int y) {
  uint32_t px[4]; int ox, oy;
  if (!rgba_codes(B, stat, ST, v, out_bits, x, y, px, ox, oy)) return;
  const size_t di = ((size_t)oy * (size_t)frame_of(B).out_w + (size_t)ox) * 4;
  if (out_bits == 8) *(uint32_t *)(B.out + di) = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
  else { uint16_t *o16 = (uint16_t *)B.out + di; for (int c = 0; c < 4; c++) o16[c] = (uint16_t)px[c]; }
}

// XYB -> the image's colour encoding (opsin inverse, target primaries, transfer function), NOT clamped: what libjxl's "XYB" + "FromLinear" stages hand to
// the blending stage / the writer
// XYB -> linear light of the target primaries: the cube of the biased gamma-domain values, then the (scaled) inverse opsin matrix.  Every multiply-add is
// spelled out, so that the writer instantiations (general, the sweep's fast one, the ones that hand the pixel to the post stages) give the same bits whatever
// the compiler would have contracted in each of them — jxlamd_decoder_set_writer_post promises the bytes of decode + jxlamd_post_fused.
JXL_DEV void opsin_mix(const float *bias_cbrt, const float *bias, float X, float Y, float Bc, float (&mix)[3]) {
  const float gl = (Y + X) - bias_cbrt[0], gm = (Y - X) - bias_cbrt[1], gs = Bc - bias_cbrt[2];
  mix[0] = fmaf(gl * gl, gl, bias[0]); mix[1] = fmaf(gm * gm, gm, bias[1]); mix[2] = fmaf(gs * gs, gs, bias[2]);
}
JXL_DEV float opsin_lin(const float *inv, int c, const float (&mix)[3]) { return fmaf(inv[c * 3], mix[0], fmaf(inv[c * 3 + 1], mix[1], inv[c * 3 + 2] * mix[2])); }
JXL_DEV void xyb_to_rgb(const DevFrame &F, float X, float Y, float Bc, float (&v)[3]) {
  float mix[3];
  opsin_mix(F.opsin_bias_cbrt, F.opsin_bias, X, Y, Bc, mix);
  float hlg_ratio = 1.0f;
  if (F.transfer == 18 && F.hlg_exponent != 0.0f) {        // inverse OOTF: scale by luminance^(gamma - 1)
    float lum = 0.0f;
    for (int c = 0; c < 3; c++) lum += F.hlg_lum[c] * opsin_lin(F.opsin_inv, c, mix);
    // libjxl: min(pow(luminance, exponent), 1e9) — pow(0, negative exponent) is +inf there, i.e. the clamp value
    hlg_ratio = lum > 0.0f ? pow_pos(lum, F.hlg_exponent) : (lum == 0.0f && F.hlg_exponent < 0.0f) ? 1e9f : 0.0f;
    if (hlg_ratio > 1e9f) hlg_ratio = 1e9f;
  }
  for (int c = 0; c < 3; c++) {
    float lin = opsin_lin(F.opsin_inv, c, mix);
    switch (F.transfer) {
      case 18: lin = tf_hlg(lin * hlg_ratio); break;
      case 17: { float a = pow_pos(fabsf(lin), 1.0f / 2.6f); lin = lin < 0 ? -a : a; } break;
      case 13: lin = tf_srgb(lin); break;
      case 16: lin = tf_pq(lin, F.intensity_target); break;
      case 1: lin = tf_709(lin); break;
      case -1: { float a = pow_pos(fabsf(lin), F.gamma); lin = lin < 0 ? -a : a; } break;
      default: break;   // 8 = linear
    }
    v[c] = lin;
  }
}
JXL_DEV void xyb_write_value(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float X, float Y, float Bc, int out_bits, int x, int y) {
  float v[3];
  xyb_to_rgb(frame_of(B), X, Y, Bc, v);
  for (int c = 0; c < 3; c++) { const float lin = v[c]; v[c] = lin < 0.0f ? 0.0f : lin > 1.0f ? 1.0f : lin; if (!(lin == lin)) v[c] = 0.0f; }
  rgba_store(B, stat, ST, v, out_bits, x, y);
}
JXL_DEV void xyb_write_pixel(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *const src[3], int out_bits, int x, int y) {
  const size_t po = (size_t)y * (size_t)frame_of(B).pw + (size_t)x;
  xyb_write_value(B, stat, ST, src[0][po], src[1][po], src[2][po], out_bits, x, y);
}

// Writer of a VarDCT frame that is not XYB (a recompressed JPEG): the planes hold the image's own samples — R, G, B, or Cb, Y, Cr centred on zero, which
// libjxl's YCbCr stage turns into RGB with the full-range BT.601 matrix of JFIF (multiply and add rounded separately: the reference is an SSE2 build).
// No transfer function (the samples are already in the image's colour space); clamp, the 8-bit writer's dither, orientation as for every frame.
JXL_DEV float mul_add_rn(float a, float b, float c) {
#ifdef __HIPCC__
  return __fadd_rn(__fmul_rn(a, b), c);
#else
  volatile float p = a * b;      // (the CPU harness may be built with contraction enabled)
  return p + c;
#endif
}
JXL_DEV void plain_to_rgb(const DevFrame &F, float c0, float c1, float c2, float (&v)[3]) {
  v[0] = c0; v[1] = c1; v[2] = c2;
  if (F.not_xyb == 2) {
    const float yv = c1 + 128.0f / 255;
    v[0] = mul_add_rn(1.402f, c2, yv);
    v[1] = mul_add_rn(-0.299f * 1.402f / 0.587f, c2, mul_add_rn(-0.114f * 1.772f / 0.587f, c0, yv));
    v[2] = mul_add_rn(1.772f, c0, yv);
  }
}
JXL_DEV void plain_write_value(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float c0, float c1, float c2, int out_bits, int x, int y) {
  const DevFrame &F = frame_of(B);
  float v[3];
  plain_to_rgb(F, c0, c1, c2, v);
  for (int c = 0; c < 3; c++) { const float t = v[c]; v[c] = t < 0.0f ? 0.0f : t > 1.0f ? 1.0f : t; if (!(t == t)) v[c] = 0.0f; }
  rgba_store(B, stat, ST, v, out_bits, x, y);
}

}  // namespace jxlamd
