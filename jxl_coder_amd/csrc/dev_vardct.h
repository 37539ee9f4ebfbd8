// jxl_coder_amd/csrc/dev_vardct.h — device-side VarDCT section decoders:
//   * LfGroup  (2048x2048 px): LF coefficients + HF metadata (CfL maps, block strategies, quant field, EPF
//     sharpness) — two Modular streams — then varblock placement and LF dequantisation;
//   * PassGroup (256x256 px): per-block nonzero counts and AC coefficients (ISO/IEC 18181-1 Annex C.8 / I).
// These are the entropy-decode stages that libjxl runs per group under JxlDecoderProcessInput (reference call
// site jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).  One wavefront per section; lane 0 is the serial
// bitstream walker, the whole wave does the data-parallel epilogues.
#pragma once
#include "dev_modular.h"
#include "dev_modular_wave.h"
#include "dev_tables.h"

namespace jxlamd {

struct DevPost;                  // post.h: A10 + A11 inside the writer (HIP builds only)
struct DevBuffers {
  const uint8_t *codestream;
  const uint8_t *tables;        // blob; DevFrame at offset 0
  uint8_t *strategy;            // per 8x8 cell: raw AcStrategy of the covering varblock (0xFF = unset)
  uint8_t *first;               // per cell: 1 = top-left cell of its varblock
  uint8_t *qfm1;                // per cell: quant field - 1
  uint8_t *sharp;               // per cell: EPF sharpness 0..7
  uint8_t *lf_idx;              // per cell: LF-threshold bucket for the block context
  int8_t *xfromy, *bfromy;      // per 64x64 tile
  float *lf[3];                 // dequantised LF (xb*yb)
  float *lf_s[3];               // after adaptive smoothing
  uint32_t *coef_off;           // per first-cell: offset (in coefficients) inside its group's pool
  int32_t *coef[3];             // [num_groups][65536]
  float *plane_a[3], *plane_b[3];
  int32_t *lf_scratch;          // [num_lf_groups][kLfScratchInts]
  LocalTreeScratch *local;      // [max(num_lf_groups, num_groups)]: local MA trees / histograms parsed on the device
  int32_t *mod_pool;            // Modular-encoded frames: int32 channel planes (DevFrame::mod_plane_off)
  int32_t *mod_scratch;         // [num_groups][group channels x 65536]: per-group channel rectangles
  uint32_t *big_list[4];        // cell indices of the varblocks with 512..1024 / 2048..4096 / <= 256 / more than 4096 (DCT128, DCT256 families) coefficients (filled at placement)
  uint32_t *big_count;          // [4] their counts
  uint64_t *mod_end_bit;        // [1]: where the GlobalModular stream of an extra-channel frame ended (single-section frames: LfGroup 0 starts there)
  uint64_t *pass_end_bits;      // [num_passes][num_groups]: where the AC stream of a group ended (extra-channel frames: its ModularGroup stream starts there)
  uint8_t *pass_nz;             // [num_groups][3072]: per-group nonzero-count maps of the lane-per-stream PassGroup kernel
  uint32_t *err;
  uint8_t *out;                 // RGBA8 / RGBA16
  int32_t out_bits;             // 8 or 16 (used by the batched writer)
  uint32_t *lz_win;             // Modular-encoded frames whose code uses LZ77: [DevFrame::lz_win_len] + [num_groups][DevFrame::lz_win_group] decoded integers (else null)
  const uint8_t *stat;          // the decoder's static tables (DevStatic at 0): natural coefficient orders for the PassGroup kernels
  float *up[4];                 // upsampled frames: dense full_w x full_h f32 planes between the upsampling stage and the writer ([3]: an upsampled alpha channel)
  float *noise[3];              // DevFrame::noise: the three random planes (pw x ph f32 each)
  const float *lf_frame[3];     // DevFrame::use_lf_frame: the LF frame's X, Y, B planes (dense lf_frame_w x lf_frame_h)
  const DevPost *post;          // non-null: the frame's last filter stage hands its pixels to the post stages instead of storing RGBA (kernels_filter.hip: k_filter_b<3, 1>)
  float *ref_a[4];              // ... and, for a slot that holds a blended canvas (frames of an animation), its alpha plane (null: the image has none)
  float *canvas_save[4];        // where blend_canvas_pixel keeps the blended canvas (R, G, B, A planes of canvas_w x canvas_h; null: not kept)
  float *ref[4][3];             // reference slots (patch dictionaries): 3 f32 planes of DevFrame::ref_w x ref_h samples each, XYB or RGB as the image is coded
  // Sparse coefficient lists (flights, single-pass frames; null: the dense coef[] planes).  k_pass_flat appends one 32-bit entry per NONZERO
  // coefficient to its group's arena — varblocks in stream order, inside a varblock channel Y, then X, then B — and records per varblock
  // (at its first cell) where its entries start (coef_off, an index into coef_sp) and how many there are (coef_cnt).  The reconstruction
  // front ends clear their LDS tile and scatter the entries into it: no 3 x 65 536 x 4-byte planes per group to write, read and clear.
  uint32_t *coef_sp;            // entries: position in the varblock's storage layout (12 bits) | channel << 12 | quantised value << 14
  uint32_t *coef_cnt;           // per first-cell: entries of the varblock
  const uint32_t *sp_group;     // [num_groups + 1]: first entry of each group's arena (the host sizes it from the group's section bytes)
};
// entry of a sparse coefficient list; values outside 18 bits / positions beyond 4 095 do not fit (kErrNeedDense: the flight is decoded again with the dense planes)
JXL_DEV uint32_t sp_pack(uint32_t pos, int c, int32_t v) { return pos | ((uint32_t)c << 12) | ((uint32_t)v << 14); }
JXL_DEV bool sp_fits(uint32_t pos, int32_t v) { return pos < 4096u && v >= -(1 << 17) && v < (1 << 17); }
JXL_DEV uint32_t sp_pos(uint32_t e) { return e & 4095u; }
JXL_DEV int sp_chan(uint32_t e) { return (int)((e >> 12) & 3u); }
JXL_DEV int32_t sp_val(uint32_t e) { return (int32_t)e >> 14; }

constexpr int kLfScratchInts = 6 * 65536 + 2048 + 16;   // LF ints (3 planes) + CfL maps + block info + sharpness + [last] extra_precision

JXL_DEV const DevFrame &frame_of(const DevBuffers &B) { return *(const DevFrame *)B.tables; }
// coefficient order of (pass, order bucket, channel): coded permutations sit in the frame blob, natural orders in the static tables
JXL_DEV const uint32_t *order_ptr(const DevBuffers &B, const DevFrame &F, int pass, int o, int c) {
  const uint32_t off = F.order_off[pass][o][c];
  return (off & kOrderInStatic) ? (const uint32_t *)(B.stat + (off & ~kOrderInStatic)) : (const uint32_t *)(B.tables + off);
}
JXL_DEV int ceil_log2u(uint32_t x) { int r = 0; while ((1u << r) < x) r++; return r; }

// ------------------------------------------------------------------ LfGroup (serial phases run by lane 0)
struct LfGeom { int bx0, by0, bw, bh, tw, th; };
JXL_DEV LfGeom lf_geom(const DevFrame &F, int g) {
  LfGeom q;
  const int gx = g % F.xlfg, gy = g / F.xlfg;
  q.bx0 = gx * 256; q.by0 = gy * 256;
  q.bw = F.xb - q.bx0 < 256 ? F.xb - q.bx0 : 256; q.bh = F.yb - q.by0 < 256 ? F.yb - q.by0 : 256;
  q.tw = (q.bw + 7) / 8; q.th = (q.bh + 7) / 8;
  return q;
}

// phase 1: open the section, read extra_precision, begin the LF-coefficient stream
JXL_DEV void lf_phase_open(const DevBuffers &B, DevModScratch &S, int g) {
  const DevFrame &F = frame_of(B);
  const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
  const DevSection sec = secs[F.nsec == 1 ? 0 : 1 + g];
  DevBits b;
  if (F.nsec == 1) {   // single-section frame: LfGroup 0 follows LfGlobal (host-parsed) and, with extra channels, the GlobalModular stream the device decoded
    bits_init_at_bit(b, B.codestream, sec.off, F.has_ec ? *B.mod_end_bit : (uint64_t)F.single_lf_bit, F.cs_size);
  } else bits_init(b, B.codestream, sec.off, F.cs_size);
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  if (F.use_lf_frame) { scr[kLfScratchInts - 1] = 0; S.st.b = b; S.lz.win = nullptr; return; }      // no LF coefficients in the section: the LF image is an LF frame's pixels
  scr[kLfScratchInts - 1] = (int32_t)bits_read(b, 2);      // extra_precision
  S.st.b = b;
  S.lz.win = nullptr;                                       // LZ77 is confined to Modular-encoded frames (the host rejects it elsewhere)
  modular_stream_begin(B.tables, F, B.local[g], S);
}
// one stream's channels: the whole wave on the GPU (dev_modular_wave.h), lane 0 alone in the CPU harness
// kWave = false: the calling lane decodes the stream alone with the serial walker (lane-per-stream kernels)
template <bool kWave = true, bool kGeneral = true>
JXL_DEV uint32_t lf_decode_stream(DevModScratch &S, const DevChanOut *ch, int nch, int stream_id, int tid) {
#ifdef __HIPCC__
  if (!kWave) return modular_stream_decode(S, ch, nch, stream_id);
  return modular_stream_decode_wave<kGeneral>(S, ch, nch, stream_id, tid);
#else
  return tid == 0 ? modular_stream_decode(S, ch, nch, stream_id) : 0;
#endif
}
// phase 2a (all lanes): decode LF coefficients (channels Y, X, B)
template <bool kWave = true, bool kGeneral = true>
JXL_DEV uint32_t lf_phase_coeffs(const DevBuffers &B, DevModScratch &S, int g, int tid) {
  const DevFrame &F = frame_of(B);
  const LfGeom q = lf_geom(F, g);
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  DevChanOut *ch = S.ch;
  for (int c = 0; c < 3; c++) { ch[c].d = scr + (size_t)c * 65536; ch[c].w = q.bw; ch[c].h = q.bh; ch[c].hs = ch[c].vs = 0; }
  if (F.subsampled) {       // YCbCr frame with subsampled chroma: stream channels Y, Cb, Cr carry the group's rectangle >> the channel's shifts (the block grid is whole MCUs)
    for (int i = 0; i < 3; i++) { const int c = i == 0 ? 1 : i == 1 ? 0 : 2; ch[i].w = q.bw >> F.hshift[c]; ch[i].h = q.bh >> F.vshift[c]; ch[i].hs = (int16_t)F.hshift[c]; ch[i].vs = (int16_t)F.vshift[c]; }
  }
  return lf_decode_stream<kWave, kGeneral>(S, ch, 3, 1 + g, tid);
}
// phase 2b (lane 0): block count, begin the HF-metadata stream
JXL_DEV uint32_t lf_phase_meta_open(const DevBuffers &B, DevModScratch &S, int g) {
  const DevFrame &F = frame_of(B);
  const LfGeom q = lf_geom(F, g);
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  const int count = 1 + (int)bits_read(S.st.b, ceil_log2u((uint32_t)(q.bw * q.bh)));
  if (count > q.bw * q.bh) return kErrBitstream;
  scr[kLfScratchInts - 2] = count;
  modular_stream_begin(B.tables, F, B.local[g], S);
  return 0;
}
// phase 3a (all lanes): decode HF metadata
template <bool kWave = true, bool kGeneral = true>
JXL_DEV uint32_t lf_phase_meta(const DevBuffers &B, DevModScratch &S, int g, int tid) {
  const DevFrame &F = frame_of(B);
  const LfGeom q = lf_geom(F, g);
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  const int count = scr[kLfScratchInts - 2];
  int32_t *m_x = scr + 3 * 65536, *m_b = m_x + 1024, *m_blk = m_b + 1024, *m_sharp = m_blk + 2 * 65536;
  DevChanOut *ch = S.ch;
  for (int i = 0; i < 4; i++) ch[i].hs = ch[i].vs = 0;
  ch[0].d = m_x; ch[0].w = q.tw; ch[0].h = q.th;
  ch[1].d = m_b; ch[1].w = q.tw; ch[1].h = q.th;
  ch[2].d = m_blk; ch[2].w = count; ch[2].h = 2;
  ch[3].d = m_sharp; ch[3].w = q.bw; ch[3].h = q.bh;
  return lf_decode_stream<kWave, kGeneral>(S, ch, 4, 1 + 2 * F.num_lf_groups + g, tid);
}
#ifdef __HIPCC__
// lanes 0 .. n-1 hold list entries (class << 28 | cell): one atomicAdd per class reserves the slots, the lanes store
__device__ __forceinline__ void place_flush_lists(const DevBuffers &B, int pend, int n, int tid) {
  const int lane = tid & 63;
  const bool have = lane < n;
  const int cls = (int)((uint32_t)pend >> 28);
  for (int c = 0; c < 4; c++) {
    const uint64_t mask = __ballot(have && cls == c);
    if (!mask) continue;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&B.big_count[c], (uint32_t)__builtin_popcountll(mask));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (have && cls == c) B.big_list[c][base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint32_t)pend & 0x0fffffffu;
  }
}
#endif
// phase 3b (all lanes in lock-step): place the varblocks.  The raster scan over the occupancy bitmap is serial by definition
// (a block goes to the first cell still free) and every lane walks it identically; what the lanes share out is the
// per-cell bookkeeping of each block (up to 64 covered cells x 3 byte planes).  Lane 0 owns the bitmap and the lists;
// a wavefront's LDS accesses complete in program order, so the other lanes read what it wrote the iteration before.
template <class Sync>
JXL_DEV uint32_t lf_phase_place(const DevBuffers &B, DevModScratch &S, int g, uint64_t *end_bits, int tid, int nthreads, Sync sync) {
  const DevFrame &F = frame_of(B);
  const LfGeom q = lf_geom(F, g);
  const int bx0 = q.bx0, by0 = q.by0, bw = q.bw, bh = q.bh;
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  const int count = scr[kLfScratchInts - 2];
  int32_t *m_blk = scr + 3 * 65536 + 2048;
  const DevBits &b = S.st.b;
  const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
  const DevSection sec = secs[F.nsec == 1 ? 0 : 1 + g];
  if (end_bits && tid == 0) end_bits[g] = b.consumed;
  if (b.consumed > (uint64_t)sec.size * 8 + 64 && F.nsec != 1) return kErrBitstream;
  // --- varblock placement: the next block goes to the first unoccupied cell in raster order.  Occupancy is a
  // bitmap in LDS (the alias-table area is free once the streams are decoded), scanned a 32-bit word at a time, so
  // the serial cost scales with the number of varblocks, not with the 65 536 cells.
  uint32_t *occ = (uint32_t *)S.pool;                  // 256 rows x 8 words (8 KiB <= kModPoolBytes)
  for (int i = tid; i < 256 * 8; i += nthreads) occ[i] = 0;
  sync();
  int num = 0;
#ifdef __HIPCC__
  // block records 64 at a time (lane l holds record pf_base + l): the serial loop used to wait for two dependent global loads per block
  int pf_base = -64, pf_st = 0, pf_q = 0;
  int pend = 0, npend = 0;
#endif
  for (int y = 0; y < bh; y++) {
    for (int wx = 0; wx < (bw + 31) / 32; wx++) {
      for (;;) {
        uint32_t freebits = ~occ[y * 8 + wx];
        if (wx * 32 + 32 > bw) freebits &= (1u << (bw - wx * 32)) - 1u;
        if (!freebits) break;
        const int x = wx * 32 + __builtin_ctz(freebits);
        if (num >= count) return kErrBitstream;
#ifdef __HIPCC__
        if (num - pf_base >= 64) {
          pf_base = num;
          const int idx = num + (tid & 63);
          pf_st = idx < count ? m_blk[idx] : 0; pf_q = idx < count ? m_blk[count + idx] : 0;
        }
        const int pl = __builtin_amdgcn_readfirstlane(num - pf_base);
        // Runs of one-cell varblocks (round 6: a busy 4K frame places 24 000 blocks per LF group, nearly all of them 8 x 8 — 19 ms of a 118 ms stream at 0.8 us per
        // block): the next m records are one-cell blocks and the word has free cells — the k-th of them takes the k-th free cell, exactly what the serial scan
        // does one block at a time.  Cell-centric: lane i < 32 is bit i of the word, its rank among the free bits picks its record.
        {
          const int ridx = pf_base + (tid & 63);
          const bool one_cell = ridx < count && pf_st >= 0 && pf_st <= 26 && ((0x3F00Fu >> pf_st) & 1u) && pf_q >= 0 && pf_q <= 255;      // strategies 0 - 3, 12 - 17: kCoveredX = kCoveredY = 1
          const uint64_t okm = __ballot(one_cell) >> pl;
          const int run = okm == ~0ull ? 64 - pl : __builtin_ctzll(~okm);
          const int nf = __builtin_popcount(freebits);
          const int m = run < nf ? run : nf;
          if (m >= 2) {
            const int lane = tid & 63;
            const bool free_i = lane < 32 && ((freebits >> lane) & 1u);
            const int rank = __builtin_popcount(freebits & ((1u << (lane & 31)) - 1u));
            const bool take = free_i && rank < m;
            const int src = (pl + (take ? rank : 0)) & 63;
            const int rst = __shfl(pf_st, src, 64), rq = __shfl(pf_q, src, 64);
            const size_t oo = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + wx * 32 + lane);
            if (take) { B.strategy[oo] = (uint8_t)rst; B.first[oo] = 1; B.qfm1[oo] = (uint8_t)rq; }
            const uint32_t taken = (uint32_t)__ballot(take);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (tid == 0) occ[y * 8 + wx] |= taken;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            uint32_t base = 0;                             // the size-class list of the small blocks (an unordered set: the reconstruction kernels walk it)
            if (lane == 0) base = atomicAdd(&B.big_count[2], (uint32_t)m);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (take) B.big_list[2][base + (uint32_t)rank] = (uint32_t)oo & 0x0fffffffu;
            num += m;
            continue;
          }
        }
        const int st = __builtin_amdgcn_readlane(pf_st, pl), q = __builtin_amdgcn_readlane(pf_q, pl);
#else
        const int st = m_blk[num], q = m_blk[count + num];
#endif
        num++;
        if (st < 0 || st > 26 || q < 0 || q > 255) return kErrBitstream;
        const int cx = kCoveredX[st], cy = kCoveredY[st];
        if (x + cx > bw || y + cy > bh) return kErrBitstream;
        if ((x & 31) + cx > 32 || (y & 31) + cy > 32) return kErrBitstream;   // must not straddle a 256x256 group
        const uint32_t mask = (cx == 32 ? 0xFFFFFFFFu : ((1u << cx) - 1u)) << (x & 31);
        const size_t o = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + x);
        for (int iy = 0; iy < cy; iy++) {
          if (occ[(y + iy) * 8 + wx] & mask) return kErrBitstream;               // overlapping varblocks (uniform: every lane sees the same word)
        }
        // all lanes have read the words before lane 0 updates them.  On the GPU the section is ONE wave whose LDS accesses execute in
        // program order: a compiler-level fence is enough, and unlike a workgroup barrier it does not wait for the global stores of
        // the previous block (that wait was most of the serial loop's time)
#ifdef __HIPCC__
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#else
        sync();
#endif
        if (tid == 0) for (int iy = 0; iy < cy; iy++) occ[(y + iy) * 8 + wx] |= mask;
#ifdef __HIPCC__
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#else
        sync();
#endif
        for (int c = tid; c < cx * cy; c += nthreads) {
          const int iy = c / cx, ix = c - iy * cx;
          const size_t oo = o + (size_t)iy * (size_t)F.xb + (size_t)ix;
          B.strategy[oo] = (uint8_t)st; B.first[oo] = (uint8_t)(c == 0); B.qfm1[oo] = (uint8_t)q;
        }
        const int ncoef = cx * cy * 64;
#ifdef __HIPCC__
        // size-class lists (the reconstruction kernels walk them): entries collect in a lane shift register and reach the lists 64 at a
        // time with one atomic per class — an atomic with return per block was the serial loop's longest wait (1.2 us per block)
        {
          const int cls = ncoef <= 256 ? 2 : ncoef <= 1024 ? 0 : ncoef <= 4096 ? 1 : 3;
          pend = __builtin_amdgcn_update_dpp((int)(((uint32_t)cls << 28) | (uint32_t)o), pend, 0x138, 0xF, 0xF, false);   // lane l: the entry of l blocks ago
          if (++npend == 64) { place_flush_lists(B, pend, npend, tid); npend = 0; }
        }
#else
        if (tid == 0) {         // size-class lists: the reconstruction kernels walk them
          const int cls = ncoef <= 256 ? 2 : ncoef <= 1024 ? 0 : ncoef <= 4096 ? 1 : 3;
          const uint32_t slot = B.big_count[cls]++;
          B.big_list[cls][slot] = (uint32_t)o;
        }
#endif
      }
    }
  }
#ifdef __HIPCC__
  place_flush_lists(B, pend, npend, tid);
#endif
  return 0;
}

// ------------------------------------------------------------------ LfGroup, parallel epilogue (all lanes)
JXL_DEV void lf_group_epilogue(const DevBuffers &B, int g, int lane, int nlanes) {
  const DevFrame &F = frame_of(B);
  const int gx = g % F.xlfg, gy = g / F.xlfg;
  const int bx0 = gx * 256, by0 = gy * 256;
  const int bw = F.xb - bx0 < 256 ? F.xb - bx0 : 256, bh = F.yb - by0 < 256 ? F.yb - by0 : 256;
  const int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  const float mul = 1.0f / (float)(1 << scr[kLfScratchInts - 1]);
  const float fx = F.lf_fac[0] * mul, fy = F.lf_fac[1] * mul, fb = F.lf_fac[2] * mul;
  const int32_t *m_x = scr + 3 * 65536, *m_b = m_x + 1024, *m_sharp = m_b + 1024 + 2 * 65536;
  if (F.use_lf_frame) {
    // progressive_dc: one pixel of the LF frame per cell, already dequantised XYB; the block context sees LF bucket 0 (libjxl zero-fills quant_dc)
    for (int i = lane; i < bw * bh; i += nlanes) {
      const int y = i / bw, x = i - y * bw;
      const size_t o = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + x);
      const size_t so = (size_t)(by0 + y) * (size_t)F.lf_frame_w + (size_t)(bx0 + x);
      for (int c = 0; c < 3; c++) B.lf[c][o] = B.lf_frame[c][so];
      B.lf_idx[o] = 0;
      const int sh = m_sharp[i];
      B.sharp[o] = (uint8_t)(sh < 0 ? 0 : sh > 7 ? 7 : sh);
    }
  } else
  if (F.subsampled) {
    // every channel on its own grid in the top-left corner of its LF plane (stride xb), no chroma from luma on LF; the block-context bucket of a
    // full-resolution cell comes from the samples that cover it
    const int cof[3] = {1, 0, 2};                       // stream channel i holds channel cof[i]
    const float fac[3] = {fx, fy, fb};
    for (int i = 0; i < 3; i++) {
      const int c = cof[i], cw = bw >> F.hshift[c], chh = bh >> F.vshift[c];
      for (int k = lane; k < cw * chh; k += nlanes) {
        const int y = k / cw, x = k - y * cw;
        B.lf[c][(size_t)((by0 >> F.vshift[c]) + y) * (size_t)F.xb + (size_t)((bx0 >> F.hshift[c]) + x)] = (float)scr[(size_t)i * 65536 + (size_t)k] * fac[c];
      }
    }
    for (int i = lane; i < bw * bh; i += nlanes) {
      const int y = i / bw, x = i - y * bw;
      const size_t o = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + x);
      const int32_t qy = scr[(y >> F.vshift[1]) * (bw >> F.hshift[1]) + (x >> F.hshift[1])];
      const int32_t qx = scr[65536 + (y >> F.vshift[0]) * (bw >> F.hshift[0]) + (x >> F.hshift[0])];
      const int32_t qb = scr[2 * 65536 + (y >> F.vshift[2]) * (bw >> F.hshift[2]) + (x >> F.hshift[2])];
      int ix = 0, iy = 0, ib = 0;
      for (int t = 0; t < F.nb_lf_thr[0]; t++) if (qx > F.lf_thr[0][t]) ix++;
      for (int t = 0; t < F.nb_lf_thr[1]; t++) if (qy > F.lf_thr[1][t]) iy++;
      for (int t = 0; t < F.nb_lf_thr[2]; t++) if (qb > F.lf_thr[2][t]) ib++;
      int bucket = ix; bucket = bucket * (F.nb_lf_thr[2] + 1) + ib; bucket = bucket * (F.nb_lf_thr[1] + 1) + iy;
      B.lf_idx[o] = (uint8_t)bucket;
      const int sh = m_sharp[i];
      B.sharp[o] = (uint8_t)(sh < 0 ? 0 : sh > 7 ? 7 : sh);
    }
  } else
  for (int i = lane; i < bw * bh; i += nlanes) {
    int y = i / bw, x = i - y * bw;
    size_t o = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + x);
    int32_t qy = scr[i], qx = scr[65536 + i], qb = scr[2 * 65536 + i];
    float Y = (float)qy * fy;
    B.lf[1][o] = Y;
    B.lf[0][o] = (float)qx * fx + F.cfl_dc_x * Y;
    B.lf[2][o] = (float)qb * fb + F.cfl_dc_b * Y;
    int ix = 0, iy = 0, ib = 0;
    for (int t = 0; t < F.nb_lf_thr[0]; t++) if (qx > F.lf_thr[0][t]) ix++;
    for (int t = 0; t < F.nb_lf_thr[1]; t++) if (qy > F.lf_thr[1][t]) iy++;
    for (int t = 0; t < F.nb_lf_thr[2]; t++) if (qb > F.lf_thr[2][t]) ib++;
    int bucket = ix; bucket = bucket * (F.nb_lf_thr[2] + 1) + ib; bucket = bucket * (F.nb_lf_thr[1] + 1) + iy;
    B.lf_idx[o] = (uint8_t)bucket;
    int sh = m_sharp[i];
    B.sharp[o] = (uint8_t)(sh < 0 ? 0 : sh > 7 ? 7 : sh);
  }
  const int tw = (bw + 7) / 8, th = (bh + 7) / 8;
  for (int i = lane; i < tw * th; i += nlanes) {
    int y = i / tw, x = i - y * tw;
    size_t o = (size_t)(by0 / 8 + y) * (size_t)F.tiles_x + (size_t)(bx0 / 8 + x);
    int vx = m_x[i], vb = m_b[i];
    B.xfromy[o] = (int8_t)(vx < -128 ? -128 : vx > 127 ? 127 : vx);
    B.bfromy[o] = (int8_t)(vb < -128 ? -128 : vb > 127 ? 127 : vb);
  }
}

// ------------------------------------------------------------------ PassGroup (lane 0)
constexpr int kPassCtxLds = 495 * 16;     // context-map slice of one HF preset cached in LDS (num_bctx <= 16; larger maps stay in HBM/L2)
struct DevPassScratch {
  uint8_t nz[3][32 * 32];
  uint8_t ctx_map[kPassCtxLds];
  uint32_t cfg[256];
  uint16_t freq_ctx[64], nnz_ctx[64];
  uint32_t order8[2][3][64];              // coefficient orders of the two 64-coefficient order ids (DCT8 / other 8x8)
  DevBits b;
  int32_t sel, pass;
  uint32_t err;
};

// phase 1 (lane 0): open the section, read the HF preset selector
JXL_DEV void pass_phase_open(const DevBuffers &B, DevPassScratch &S, int pass, int g) {
  const DevFrame &F = frame_of(B);
  const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
  const DevSection sec = secs[F.nsec == 1 ? 0 : 2 + F.num_lf_groups + pass * F.num_groups + g];
  DevBits b;
  bits_init(b, B.codestream, sec.off, F.cs_size);
  if (F.nsec == 1) {
    uint32_t skip = F.single_pass_bit;
    while (skip >= 32) { bits_read(b, 32); skip -= 32; }
    bits_read(b, (int)skip);
  }
  S.sel = (int)bits_read(b, ceil_log2u((uint32_t)F.num_presets));
  S.err = S.sel >= F.num_presets ? (uint32_t)kErrBitstream : 0u;
  S.b = b; S.pass = pass;
}
// phase 2 (all lanes): stage this preset's context-map slice, the hybrid-uint configs and small tables in LDS
JXL_DEV void pass_phase_stage(const DevBuffers &B, DevPassScratch &S, int tid, int nthreads) {
  const DevFrame &F = frame_of(B);
  if (S.err) return;
  const DevEC &e = F.hf_ec[S.pass];
  const int n = 495 * F.num_bctx;
  const uint8_t *src = B.tables + e.ctx_map_off + (size_t)S.sel * (size_t)n;
  if (n <= kPassCtxLds) for (int i = tid; i < n; i += nthreads) S.ctx_map[i] = src[i];
  const uint32_t *cfg = (const uint32_t *)(B.tables + e.cfg_off);
  for (int i = tid; i < e.num_clusters && i < 256; i += nthreads) S.cfg[i] = cfg[i];
  for (int i = tid; i < 64; i += nthreads) { S.freq_ctx[i] = kCoeffFreqContext[i]; S.nnz_ctx[i] = kCoeffNumNonzeroContext[i]; }
  for (int i = tid; i < 2 * 3 * 64; i += nthreads) {
    const int o = i / 192, c = (i / 64) % 3, k = i & 63;
    S.order8[o][c][k] = order_ptr(B, F, S.pass, o, c)[k];
  }
  { uint8_t *nzflat = &S.nz[0][0]; for (int i = tid; i < 3 * 32 * 32; i += nthreads) nzflat[i] = 0; }
}
// rANS symbol + hybrid uint with the context map and configs read straight from LDS (ds_read) and one 64-bit alias
// load from the (L2-resident) frame tables
JXL_DEV uint32_t pass_ec_read(const DevPassScratch &S, const DevAlias *alias, int log_alpha, bool lds_ctx, const uint8_t *gctx,
                              DevBits &b, uint32_t &state, uint32_t ctx) {
  const uint32_t cluster = lds_ctx ? S.ctx_map[ctx] : gctx[ctx];
  const int lb = 12 - log_alpha;
  const uint32_t res = state & 0xfff;
  const uint32_t i = res >> lb, pos = res & ((1u << lb) - 1);
  const DevAlias e = alias[(cluster << log_alpha) + i];
  const uint32_t cfg = S.cfg[cluster & 255];
  const bool right = pos >= e.cutoff;
  const uint32_t sym = right ? e.right : i;
  const uint32_t off = right ? (uint32_t)e.off1 + pos : pos;
  const uint32_t freq = right ? e.freq1 : e.freq0;
  state = freq * (state >> 12) + off;
  if (state < (1u << 16)) state = (state << 16) | bits_read(b, 16);
  return ec_hybrid(b, cfg, sym);
}

#ifdef JXL_EMUL_TRACE
static long g_pass_syms = 0, g_pass_blocks = 0, g_pass_nz = 0;
#define PASS_COUNT(v, n) ((v) += (n))
#else
#define PASS_COUNT(v, n) ((void)0)
#endif
// phase 3 (lane 0): the serial rANS walk over the group's varblocks
JXL_DEV uint32_t pass_phase_decode(const DevBuffers &B, DevPassScratch &S, int g) {
  const DevFrame &F = frame_of(B);
  if (S.err) return S.err;
  const int pass = S.pass;
  const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
  const DevSection sec = secs[F.nsec == 1 ? 0 : 2 + F.num_lf_groups + pass * F.num_groups + g];
  DevBits b = S.b;
  const int gx = g % F.xgroups, gy = g / F.xgroups;
  const int bx0 = gx * 32, by0 = gy * 32;
  const int bw = F.xb - bx0 < 32 ? F.xb - bx0 : 32, bh = F.yb - by0 < 32 ? F.yb - by0 : 32;
  const int nslice = 495 * F.num_bctx;
  DevECView ev = ec_view(B.tables, F.hf_ec[pass]);
  // contexts below are relative to the preset's slice
  if (nslice <= kPassCtxLds) ev.ctx_map = S.ctx_map; else ev.ctx_map += (size_t)S.sel * (size_t)nslice;
  if (F.hf_ec[pass].num_clusters <= 256) ev.cfg = S.cfg;
  uint32_t state = ans_init(ev, b);
  const bool fast = !ev.use_prefix && F.hf_ec[pass].num_clusters <= 256;
  const bool lds_ctx = nslice <= kPassCtxLds;
  const uint8_t *gctx = ev.ctx_map;
  const int shift = F.pass_shift[pass];
  const bool accumulate = F.num_passes > 1;
  const uint8_t *bctx_map = B.tables + F.bctx_map_off;
  const int nlf = (F.nb_lf_thr[0] + 1) * (F.nb_lf_thr[1] + 1) * (F.nb_lf_thr[2] + 1);
  uint32_t pool = 0;
  for (int y = 0; y < bh; y++)
    for (int x = 0; x < bw; x++) {
      const size_t o = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + x);
      if (!B.first[o]) continue;
      const int st = B.strategy[o];
      const int cx = kCoveredX[st], cy = kCoveredY[st];
      const int covered = cx * cy, log2c = ceil_log2u((uint32_t)covered);
      const int size = covered * 64;
      const int ord = kStrategyOrder[st];
      uint32_t off;
      if (pass == 0) { off = pool; B.coef_off[o] = off; pool += (uint32_t)size; } else off = B.coef_off[o];
      if (off + (uint32_t)size > 65536u) return kErrBitstream;      // a group holds at most 32x32 cells of coefficients (stale / corrupt placement data)
      const uint32_t qf = (uint32_t)B.qfm1[o] + 1;
      int qf_idx = 0;
      for (int t = 0; t < F.nb_qf_thr; t++) if (qf > F.qf_thr[t]) qf_idx++;
      const int lfi = B.lf_idx[o];
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
        uint8_t *nzc = S.nz[c];
        // chroma-subsampled frames: a channel only has the blocks aligned to its sampling, and predicts its nonzero counts on its own grid
        const int hs = F.subsampled ? F.hshift[c] : 0, vs = F.subsampled ? F.vshift[c] : 0;
        if (((x >> hs) << hs) != x || ((y >> vs) << vs) != y) continue;
        const int sx = x >> hs, sy = y >> vs;
        int predicted;
        if (sx == 0) predicted = sy == 0 ? 32 : nzc[(sy - 1) * 32];
        else if (sy == 0) predicted = nzc[sx - 1];
        else predicted = (nzc[(sy - 1) * 32 + sx] + nzc[sy * 32 + sx - 1] + 1) / 2;
        int idx = c < 2 ? c ^ 1 : 2;
        idx = idx * 13 + ord;
        idx = idx * (F.nb_qf_thr + 1) + qf_idx;
        idx = idx * nlf + lfi;
        const int bctx = bctx_map[idx];
        const int nzp = predicted >= 64 ? 64 : predicted;
        const int nzctx = (nzp < 8 ? nzp : 4 + nzp / 2) * F.num_bctx + bctx;
        PASS_COUNT(g_pass_syms, 1); PASS_COUNT(g_pass_blocks, 1);
        int nzeros = (int)(fast ? pass_ec_read(S, ev.alias, ev.log_alpha, lds_ctx, gctx, b, state, (uint32_t)nzctx) : ec_read(ev, b, state, (uint32_t)nzctx));
        if (nzeros > size - covered) return kErrBitstream;
        const uint8_t nzv = (uint8_t)((nzeros + covered - 1) >> log2c);
        for (int iy = 0; iy < cy; iy++) for (int ix = 0; ix < cx; ix++) nzc[(sy + iy) * 32 + sx + ix] = nzv;
        const int histo = F.num_bctx * 37 + 458 * bctx;
        const uint32_t *order = ord < 2 ? S.order8[ord][c] : order_ptr(B, F, pass, ord, c);
        int32_t *blk = B.coef[c] + (size_t)g * 65536 + off;
        int prev = nzeros > size / 16 ? 0 : 1;
        for (int k = covered; k < size && nzeros != 0; k++) {
          const int nl = (nzeros + covered - 1) >> log2c;
          const int kk = k >> log2c;
          const int ctx = histo + (S.nnz_ctx[nl] + S.freq_ctx[kk]) * 2 + prev;
          PASS_COUNT(g_pass_syms, 1);
          const uint32_t u = fast ? pass_ec_read(S, ev.alias, ev.log_alpha, lds_ctx, gctx, b, state, (uint32_t)ctx) : ec_read(ev, b, state, (uint32_t)ctx);
          if (u) {
            const int32_t v = unpack_signed(u) * (1 << shift);
            if (accumulate) blk[order[k]] += v; else blk[order[k]] = v;
          }
          prev = u != 0;
          nzeros -= prev;
        }
        if (nzeros != 0) return kErrBitstream;
      }
    }
#ifdef JXL_EMUL_TRACE
  if (g == F.num_groups - 1) fprintf(stderr, "pass stage: %ld symbols, %ld block-channels over %d groups\n", g_pass_syms, g_pass_blocks, F.num_groups);
#endif
  if (state != 0x130000u) return kErrAnsFinal;
  if (F.nsec != 1 && b.consumed > (uint64_t)sec.size * 8 + 64) return kErrBitstream;
  if (F.has_ec) B.pass_end_bits[(size_t)pass * (size_t)F.num_groups + (size_t)g] = b.consumed;
  return 0;
}

// ------------------------------------------------------------------ bit supply of the lane-per-group PassGroup kernel (dev_pass_flat.h)
// 64 lanes refill their bit buffers at different steps, so with per-lane global loads nearly every step of the wavefront waits for
// somebody's refill (and, through the in-order memory counter, for the coefficient stores issued just before).  Every lane reads its
// stream through a 16-dword ring in LDS instead (lane-interleaved: bank = lane), topped up for ALL lanes of the wave at once whenever
// any lane runs below 6 dwords — one memory wait per ~20-60 symbols instead of one per symbol.
#ifndef JXL_SIMT_RING
#define JXL_SIMT_RING 16
#endif
constexpr int kSimtRing = JXL_SIMT_RING;                              // dwords per lane
struct SimtBits {
  uint64_t buf; int32_t n;                                 // valid bits in the low n positions
  uint32_t rd, wr;                                         // ring cursors (dwords, monotonic): slot = cursor & (kSimtRing - 1)
  const uint32_t *next, *end;                              // next dword of the stream to put into the ring; first dword past the padded stream
  uint64_t consumed;
};
#define SIMT_RING(r, i, lane) (r)[(((i) & (kSimtRing - 1)) << 6) + (lane)]
JXL_DEV void sbits_topup(SimtBits &b, uint32_t *ring, int lane) {            // all lanes of the wave together (uniform call site)
  for (int j = 0; j < kSimtRing; j++) {
    if (b.wr - b.rd < (uint32_t)kSimtRing) { SIMT_RING(ring, b.wr, lane) = b.next < b.end ? *b.next : 0u; b.next++; b.wr++; }
  }
}
JXL_DEV void sbits_init(SimtBits &b, uint32_t *ring, int lane, const uint8_t *base, uint64_t byte_off, uint64_t total_bytes) {
  if (byte_off > total_bytes) byte_off = total_bytes;
  const uint8_t *p = base + byte_off;
  b.end = (const uint32_t *)(base + ((total_bytes + 48) & ~(uint64_t)3));
  const uint64_t mis = (uint64_t)(uintptr_t)p & 3;
  b.next = (const uint32_t *)(p - mis);
  b.rd = b.wr = 0; b.consumed = 0;
  sbits_topup(b, ring, lane);
  const uint32_t w0 = SIMT_RING(ring, 0, lane), w1 = SIMT_RING(ring, 1, lane);
  b.rd = 2;
  b.buf = ((uint64_t)w0 | ((uint64_t)w1 << 32)) >> (8 * mis);
  b.n = 64 - 8 * (int32_t)mis;
}
JXL_DEV uint32_t sbits_read(SimtBits &b, uint32_t *ring, int lane, int n) {  // n <= 32
  if (n == 0) return 0;
  if (b.n <= 32) {
    uint32_t w;
    if (b.rd != b.wr) w = SIMT_RING(ring, b.rd, lane);
    else { w = b.next < b.end ? *b.next : 0u; b.next++; b.wr++; }           // ring ran dry between top-ups (a burst of long symbols): fetch directly
    b.rd++;
    b.buf |= (uint64_t)w << b.n; b.n += 32;
  }
  const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1));
  b.buf >>= n; b.n -= n; b.consumed += (uint64_t)n;
  return v;
}
#ifdef __HIPCC__
#define SIMT_ANY(cond) (__ballot(cond) != 0)
#else
#define SIMT_ANY(cond) (cond)
#endif

}  // namespace jxlamd
