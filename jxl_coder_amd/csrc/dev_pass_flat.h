// jxl_coder_amd/csrc/dev_pass_flat.h — PassGroup decode for flights, lane-per-group, as a FLAT state machine (k_pass_flat).
//
// pass_group_lane2 (dev_vardct.h) kept the reference's loop nest — cells of the group, three channels, coefficients — per lane.  In a
// wavefront of 64 groups that nest runs in lock-step: every (cell, channel) step lasts as long as the lane with the most symbols there,
// lanes whose varblocks start at other cells idle, and every block start waits for five dependent global loads (placement map, strategy,
// quant field, LF bucket, block-context map).  Here
//   * k_pass_prep (one wave per group, data-parallel, microseconds) turns the placement maps into a per-group list of 8-byte varblock
//     descriptors in stream order — position, shape, coefficient order bucket, quant / LF context indices, coefficient offset (the
//     running sum the serial walker used to carry) — and writes DevBuffers::coef_off for the reconstruction kernels;
//   * every lane of k_pass_flat owns one group and advances ITS OWN stream by exactly one symbol per iteration of a single loop, whatever
//     block or channel it is in: the wave finishes after max-over-lanes(symbols of a group) iterations instead of the sum over steps of
//     the per-step maxima; the next descriptor is prefetched a block ahead, the block-context map sits in LDS;
//   * the nonzero-count predictor only ever looks at the most recent varblock that covered a column (the block above (y - 1, x) and the
//     block on the left (y, x - 1) are the last ones to have covered columns x and x - 1 when (y, x) starts a block, because varblocks
//     tile the group in raster order without overlap), so its state is 32 bytes per channel in LDS instead of a 3 KB map in HBM;
//   * bits come through the per-lane LDS rings of pass_group_lane2 (topped up for the whole wave at once).
// Same integer arithmetic as pass_phase_decode / pass_group_lane2 (ISO/IEC 18181-1 C.8 / I.3); what libjxl does per group under
// JxlDecoderProcessInput (reference call site jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).  The CPU harness runs one lane at a time.
#pragma once
#include "dev_vardct.h"

namespace jxlamd {

constexpr int kPassBlkStride = 8 + 1024 * 8;      // bytes of DevBuffers::pass_nz per group: {count, 0} + up to 1024 descriptors
struct PassBlk { uint32_t a, off; };              // a: cell (y * 32 + x) | log2 cx << 10 | log2 cy << 13 | order bucket << 16 | qf_idx << 20 | lf bucket << 24
JXL_DEV uint32_t pass_blk_pack(int cell, int st, int qf_idx, int lfi) {
  const int cx = kCoveredX[st], cy = kCoveredY[st];
  return (uint32_t)cell | ((uint32_t)ceil_log2u((uint32_t)cx) << 10) | ((uint32_t)ceil_log2u((uint32_t)cy) << 13) | ((uint32_t)kStrategyOrder[st] << 16) |
         ((uint32_t)qf_idx << 20) | ((uint32_t)lfi << 24);
}
JXL_DEV int pass_qf_idx(const DevFrame &F, uint32_t qf) {
  int qf_idx = 0;
  for (int t = 0; t < F.nb_qf_thr; t++) if (qf > F.qf_thr[t]) qf_idx++;
  return qf_idx;
}

// Descriptor list of one group, serial form (CPU harness; k_pass_prep in kernels_pass.hip is the wave-parallel form of the same).
JXL_DEV uint32_t pass_prep_group_serial(const DevBuffers &B, int g) {
  const DevFrame &F = frame_of(B);
  const int gx = g % F.xgroups, gy = g / F.xgroups;
  const int bx0 = gx * 32, by0 = gy * 32;
  const int bw = F.xb - bx0 < 32 ? F.xb - bx0 : 32, bh = F.yb - by0 < 32 ? F.yb - by0 : 32;
  uint8_t *base = B.pass_nz + (size_t)g * kPassBlkStride;
  PassBlk *d = (PassBlk *)(base + 8);
  uint32_t n = 0, pool = 0;
  for (int y = 0; y < bh; y++)
    for (int x = 0; x < bw; x++) {
      const size_t o = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + x);
      if (!B.first[o]) continue;
      const int st = B.strategy[o];
      const uint32_t size = (uint32_t)kCoveredX[st] * kCoveredY[st] * 64u;
      if (pool + size > 65536u) return kErrBitstream;            // a group holds at most 32x32 cells of coefficients (stale / corrupt placement data)
      d[n].a = pass_blk_pack(y * 32 + x, st, pass_qf_idx(F, (uint32_t)B.qfm1[o] + 1), B.lf_idx[o]);
      d[n].off = pool;
      B.coef_off[o] = pool;
      pool += size; n++;
    }
  ((uint32_t *)base)[0] = n; ((uint32_t *)base)[1] = 0;
  return 0;
}

// entries of a group's sparse coefficient arena, from the bytes of its PassGroup section: four per byte (a nonzero coefficient of a photographic
// stream costs 4 - 8 bits with its share of the zero runs and the nonzero counts; measured 1.1 - 1.9 entries per byte on the fixtures and the bench frames) + slack for tiny
// sections, never more than the group has coefficients.  A stream that beats it ends with kErrNeedDense and its flight is decoded densely.
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint32_t sparse_group_entries(uint32_t section_bytes) {      // (host: decoder.hip sizes the arenas with it)
  const uint64_t e = 4ull * section_bytes + 1024u;
  return e > 3u * 65536u ? 3u * 65536u : (uint32_t)e;
}

// LDS of a PassGroup wave is what the data-parallel kernels of the other contexts do not get (round 6: 7.9 KB more per wave cost 8 % of the headline): the tables are sized
// for what libjxl writes — a block-context map of 39 x qf buckets x LF buckets bytes (78 on the bench's frames), 58 - 119 clusters — and anything larger is read from the
// frame tables in HBM / L2
constexpr int kFlatBctxLds = 256;                           // block-context map kept in LDS up to this size
constexpr int kFlatCfgLds = 128;                            // hybrid-uint configurations of the first clusters in LDS
struct FlatPassLds {                                       // per wavefront
  uint32_t ring[kSimtRing * 64];                           // lane-interleaved bit rings (sbits_*)
  uint32_t cfg[kFlatCfgLds];                               // hybrid-uint configs per cluster (clusters beyond: from the frame tables)
  uint32_t order8[2][3][64];                               // the two 64-coefficient orders
  const uint32_t *order_ptrs[13 * 3];                      // coefficient order of (bucket, channel) for this pass
  uint16_t freq_ctx[64], nnz_ctx[64];
  uint8_t nzcol[3 * 32 * 64];                              // [channel][column][lane]: nonzero-count context value of the last varblock that covered the column
  uint8_t bctx[kFlatBctxLds];
  const uint8_t *bctx_ptr;                                 // bctx above, or the frame tables
  // (round 6, measured and not kept: the context map of the pass's code — 495 entries per block context, 7.4 KB — in LDS as well, so that a symbol's cluster is not a
  // load from the frame tables in front of the alias entry's.  The PassGroup stage went from 146.6 to 142 ms per flight, but a wave of 23.8 instead of 15.9 KB took the
  // LDS the data-parallel kernels run in: reconstruction + filters 148 -> 170 ms, headline 17 078 / 16 852 -> 15 680 / 15 479 MP/s)
};
// frames the flat kernel takes (host twin: frame_flat_ok in decoder.hip): ANS codes with <= 256 clusters
JXL_DEV bool flat_frame_ok(const DevFrame &F) {
  for (int p = 0; p < F.num_passes; p++) if (F.hf_ec[p].use_prefix || F.hf_ec[p].num_clusters > 256) return false;
  return true;
}
JXL_DEV void flat_stage(const DevBuffers &B, FlatPassLds &L, int pass, int tid, int nthreads) {
  const DevFrame &F = frame_of(B);
  const DevEC &e = F.hf_ec[pass];
  const uint32_t *cfg = (const uint32_t *)(B.tables + e.cfg_off);
  for (int i = tid; i < e.num_clusters && i < kFlatCfgLds; i += nthreads) L.cfg[i] = cfg[i];
  for (int i = tid; i < 64; i += nthreads) { L.freq_ctx[i] = kCoeffFreqContext[i]; L.nnz_ctx[i] = kCoeffNumNonzeroContext[i]; }
  for (int i = tid; i < 2 * 3 * 64; i += nthreads) {
    const int o = i / 192, c = (i / 64) % 3, k = i & 63;
    L.order8[o][c][k] = order_ptr(B, F, pass, o, c)[k];
  }
  for (int i = tid; i < 13 * 3; i += nthreads) L.order_ptrs[i] = i < 6 ? &L.order8[i / 3][i % 3][0] : order_ptr(B, F, pass, i / 3, i % 3);
  const int nlf = (F.nb_lf_thr[0] + 1) * (F.nb_lf_thr[1] + 1) * (F.nb_lf_thr[2] + 1);
  const int nb = 39 * (F.nb_qf_thr + 1) * nlf;
  const uint8_t *src = B.tables + F.bctx_map_off;
#ifdef JXL_EMUL_TRACE
  if (tid == 0 && getenv("JXLEMUL_STATS")) fprintf(stderr, "flat_stage pass %d: %d clusters, log_alpha %d, block-context map %d bytes, %d block contexts, %d presets\n", pass, e.num_clusters, e.log_alpha, nb, F.num_bctx, F.num_presets);
#endif
  if (nb <= kFlatBctxLds) for (int i = tid; i < nb; i += nthreads) L.bctx[i] = src[i];
  if (tid == 0) L.bctx_ptr = nb <= kFlatBctxLds ? L.bctx : src;
  for (int i = tid; i < 3 * 32 * 64; i += nthreads) L.nzcol[i] = 0;

}
JXL_DEV uint32_t flat_ec_read(const uint32_t *cfg_lds, const uint32_t *cfg_all, const uint8_t *ctx_map, const DevAlias *alias, int log_alpha, SimtBits &b, uint32_t *ring, int lane,
                              uint32_t &state, uint32_t ctx) {
  const uint32_t cluster = ctx_map[ctx];
  const int lb = 12 - log_alpha;
  const uint32_t res = state & 0xfff;
  const uint32_t i = res >> lb, pos = res & ((1u << lb) - 1);
  const DevAlias e = alias[(cluster << log_alpha) + i];
  const uint32_t cfg = cluster < (uint32_t)kFlatCfgLds ? cfg_lds[cluster] : cfg_all[cluster];
  const bool right = pos >= e.cutoff;
  const uint32_t sym = right ? e.right : i;
  const uint32_t off = right ? (uint32_t)e.off1 + pos : pos;
  const uint32_t freq = right ? e.freq1 : e.freq0;
  state = freq * (state >> 12) + off;
  if (state < (1u << 16)) state = (state << 16) | sbits_read(b, ring, lane, 16);
  const uint32_t split_exp = cfg & 0xff, msb = (cfg >> 8) & 0xff, lsb = (cfg >> 16) & 0xff;
  const uint32_t split = 1u << split_exp;
  if (sym < split) return sym;
  uint32_t nbits = split_exp - (msb + lsb) + ((sym - split) >> (msb + lsb));
  if (nbits > 31) nbits = 31;
  const uint32_t low = sym & ((1u << lsb) - 1);
  const uint32_t tok = sym >> lsb;
  const uint32_t bits = sbits_read(b, ring, lane, (int)nbits);
  return (((((1u << msb) | (tok & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
}

#define FLAT_NZ(c, x) L.nzcol[((((c) << 5) + (x)) << 6) + lane]
// One pass of one group by one lane.  Every lane of the wave calls it (lanes without a group pass g < 0 and only take part in the
// wave-wide ring top-ups).
// kSparse: the nonzero coefficients go into the group's entry arena (DevBuffers::coef_sp) instead of the dense planes, and each varblock's
// (first entry, entries) pair into coef_off / coef_cnt at its first cell — single-pass frames only (the host decides: decoder.hip)
// g2 >= 0 (round 6): the lane takes a SECOND group the moment its first one ends — a frame's groups are not a multiple of 64 (a 4K frame: 135 = 64 + 64 + 7), and a wave of
// seven lanes holds a wave's LDS and registers for as long as a full one.  The host chains the tail groups onto lanes of the last full wave whose first group is short
// (decoder.hip: flat_wave_map — the bottom row of a frame whose height is not a multiple of 256), so the wave ends no later than before.
template <bool kSparse = false>
JXL_DEV uint32_t pass_group_flat(const DevBuffers &B, FlatPassLds &L, int pass, int g, int lane, int g2 = -1) {
  const DevFrame &F = frame_of(B);
  uint32_t *ring = L.ring;
  SimtBits b;
  bool done = g < 0;
  uint32_t err = 0;
  const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
  DevSection sec = secs[done ? 0 : (F.nsec == 1 ? 0 : 2 + F.num_lf_groups + pass * F.num_groups + g)];
  sbits_init(b, ring, lane, B.codestream, sec.off, F.cs_size);
  if (F.nsec == 1) {
    uint32_t skip = F.single_pass_bit;
    while (skip >= 32) { sbits_read(b, ring, lane, 32); skip -= 32; if (SIMT_ANY(b.wr - b.rd < 6u)) sbits_topup(b, ring, lane); }
    sbits_read(b, ring, lane, (int)skip);
    g2 = -1;                                               // (single-section frames do not ride in flights; their one group has no successor)
  }
  int sel = (int)sbits_read(b, ring, lane, ceil_log2u((uint32_t)F.num_presets));
  if (!done && sel >= F.num_presets) { err = kErrBitstream; done = true; }
  const int nslice = 495 * F.num_bctx;
  const uint8_t *ctx_map = B.tables + F.hf_ec[pass].ctx_map_off + (size_t)(done ? 0 : sel) * (size_t)nslice;
  const DevAlias *alias = (const DevAlias *)(B.tables + F.hf_ec[pass].alias_off);
  const uint32_t *cfg_all = (const uint32_t *)(B.tables + F.hf_ec[pass].cfg_off);
  const int la = F.hf_ec[pass].log_alpha;
  uint32_t state = sbits_read(b, ring, lane, 32);
  const int shift = F.pass_shift[pass];
  const bool accumulate = F.num_passes > 1;
  const int nlf = (F.nb_lf_thr[0] + 1) * (F.nb_lf_thr[1] + 1) * (F.nb_lf_thr[2] + 1), nqf = F.nb_qf_thr + 1;
  const uint8_t *bctx_map = L.bctx_ptr;
  const int num_bctx = F.num_bctx, frame_xb = F.xb;        // (read through F inside the loop they are re-loaded behind every coefficient store)
  const uint8_t *blkbase = B.pass_nz + (size_t)(done ? 0 : g) * kPassBlkStride;
  uint32_t nblk = done ? 0u : *(const uint32_t *)blkbase;
  const PassBlk *desc = (const PassBlk *)(blkbase + 8);
  PassBlk dn = {0u, 0u};
  if (nblk) dn = desc[0];
  uint32_t bi = 0;
  int ci = 3;                                              // channel step of the current varblock (3: fetch the next one)
  int nzeros = 0, k = 0, size = 64, covered = 1, log2c = 0, prev = 0, histo = 0, bctx = 0;
  int x = 0, y = 0, cx = 1, ord = 0, qf_idx = 0, lfi = 0, c = 1;
  uint32_t off = 0, o_cur = 0;
  const uint32_t *order = &L.order8[0][0][0];
  int32_t *blk = B.coef[0];
  // sparse emission: this lane's arena, its fill, where the current varblock's entries began and the cell that owns them
  uint32_t sp_base = (kSparse && !done) ? B.sp_group[g] : 0u, sp_cap = (kSparse && !done) ? B.sp_group[g + 1] - sp_base : 0u;
  uint32_t *ent = kSparse ? B.coef_sp + sp_base : nullptr;
  uint32_t ne = 0, blk_start = 0;
  size_t blk_cell = 0;
  bool sp_bad = false;
  size_t cell0 = kSparse ? (size_t)((done ? 0 : g / F.xgroups) * 32) * (size_t)F.xb + (size_t)((done ? 0 : g % F.xgroups) * 32) : 0;
  while (SIMT_ANY(!done)) {
    uint32_t ctx = 0, o_next = 0;
    const bool run = nzeros > 0;                           // inside the coefficients of a (varblock, channel)
    if (!done) {
      if (run) {
        const int nl = (nzeros + covered - 1) >> log2c, kk = k >> log2c;
        ctx = (uint32_t)(histo + (L.nnz_ctx[nl] + L.freq_ctx[kk]) * 2 + prev);
        o_next = order[k + 1 < size ? k + 1 : k];          // the order entry one symbol ahead of its use
      } else {
        if (ci == 3) {
          if (kSparse && bi > 0) { B.coef_off[blk_cell] = sp_base + blk_start; B.coef_cnt[blk_cell] = ne - blk_start; }      // the varblock just finished
          if (bi == nblk) {
            done = true;
            if (g2 >= 0) {
              // ---- this lane's first group is complete: close it, open the second one (the same steps as above, for this lane alone)
              if (kSparse && (sp_bad || ne > sp_cap)) err = kErrNeedDense;
              else if (state != 0x130000u) err = kErrAnsFinal;
              else if (b.consumed > (uint64_t)sec.size * 8 + 64) err = kErrBitstream;
              else {
                if (F.has_ec) B.pass_end_bits[(size_t)pass * (size_t)F.num_groups + (size_t)g] = b.consumed;
                g = g2; g2 = -1;
                sec = secs[2 + F.num_lf_groups + pass * F.num_groups + g];
                sbits_init(b, ring, lane, B.codestream, sec.off, F.cs_size);
                sel = (int)sbits_read(b, ring, lane, ceil_log2u((uint32_t)F.num_presets));
                if (sel >= F.num_presets) err = kErrBitstream;
                else {
                  ctx_map = B.tables + F.hf_ec[pass].ctx_map_off + (size_t)sel * (size_t)nslice;
                  state = sbits_read(b, ring, lane, 32);
                  blkbase = B.pass_nz + (size_t)g * kPassBlkStride;
                  nblk = *(const uint32_t *)blkbase;
                  desc = (const PassBlk *)(blkbase + 8);
                  dn.a = 0u; dn.off = 0u;
                  if (nblk) dn = desc[0];
                  bi = 0; nzeros = 0; k = 0; prev = 0;
                  for (int i = 0; i < 3 * 32; i++) L.nzcol[(i << 6) + lane] = 0;      // the nonzero-count columns of the new group start from zero
                  if (kSparse) { sp_base = B.sp_group[g]; sp_cap = B.sp_group[g + 1] - sp_base; ent = B.coef_sp + sp_base; ne = 0; blk_start = 0; sp_bad = false;
                                 cell0 = (size_t)((g / F.xgroups) * 32) * (size_t)F.xb + (size_t)((g % F.xgroups) * 32); }
                  done = nblk == 0;                          // (an empty group: nothing to read)
                }
              }
            }
          }
          if (!done) {
            const PassBlk d = dn;
            bi++;
            dn = desc[bi < nblk ? bi : bi - 1];            // the following descriptor, a whole varblock ahead of its use
            const int cell = (int)(d.a & 1023u), lcx = (int)((d.a >> 10) & 7u), lcy = (int)((d.a >> 13) & 7u);
            x = cell & 31; y = cell >> 5; cx = 1 << lcx;
            log2c = lcx + lcy; covered = 1 << log2c; size = covered * 64;
            ord = (int)((d.a >> 16) & 15u); qf_idx = (int)((d.a >> 20) & 15u); lfi = (int)(d.a >> 24);
            off = d.off;
            if (kSparse) { blk_start = ne; blk_cell = cell0 + (size_t)y * (size_t)frame_xb + (size_t)x; }
            ci = 0;
          }
        }
        if (!done) {
          c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
          const int above = FLAT_NZ(c, x), left = FLAT_NZ(c, x > 0 ? x - 1 : 0);
          int predicted;
          if (x == 0) predicted = y == 0 ? 32 : above;
          else if (y == 0) predicted = left;
          else predicted = (above + left + 1) / 2;
          int idx = c < 2 ? c ^ 1 : 2;
          idx = idx * 13 + ord;
          idx = idx * nqf + qf_idx;
          idx = idx * nlf + lfi;
          bctx = bctx_map[idx];
          const int nzp = predicted >= 64 ? 64 : predicted;
          ctx = (uint32_t)((nzp < 8 ? nzp : 4 + nzp / 2) * num_bctx + bctx);
        }
      }
    }
    if (SIMT_ANY(!done && b.wr - b.rd < 6u)) sbits_topup(b, ring, lane);
    if (!done) {
      const uint32_t u = flat_ec_read(L.cfg, cfg_all, ctx_map, alias, la, b, ring, lane, state, ctx);
      if (run) {
        if (u) {
          const int32_t v = unpack_signed(u) * (1 << shift);
          if (kSparse) {
            if (ne < sp_cap) ent[ne] = sp_pack(o_cur, c, v);
            ne++;
            if (!sp_fits(o_cur, v)) sp_bad = true;
          } else
          if (accumulate) blk[o_cur] += v; else blk[o_cur] = v;
        }
        o_cur = o_next;
        prev = u != 0;
        nzeros -= prev;
        k++;
        if (nzeros == 0) ci++;
        else if (k >= size) { err = kErrBitstream; done = true; }
      } else {
        const int nz = (int)u;
        if (nz > size - covered || nz < 0) { err = kErrBitstream; done = true; }
        else {
          const uint8_t nzv = (uint8_t)((nz + covered - 1) >> log2c);
          for (int ix = 0; ix < cx; ix++) FLAT_NZ(c, x + ix) = nzv;
          histo = num_bctx * 37 + 458 * bctx;
          order = L.order_ptrs[ord * 3 + c];
          if (!kSparse) blk = B.coef[c] + (size_t)g * 65536 + off;
          prev = nz > size / 16 ? 0 : 1;
          k = covered; nzeros = nz;
          if (nz == 0) ci++;
          else o_cur = order[covered];
        }
      }
    }
  }
  if (kSparse && g >= 0 && !err && (sp_bad || ne > sp_cap)) err = kErrNeedDense;
  if (g >= 0 && !err) {
    if (state != 0x130000u) err = kErrAnsFinal;
    else if (F.nsec != 1 && b.consumed > (uint64_t)sec.size * 8 + 64) err = kErrBitstream;
    else if (F.has_ec) B.pass_end_bits[(size_t)pass * (size_t)F.num_groups + (size_t)g] = b.consumed;
  }
  return err;
}
#undef FLAT_NZ

}  // namespace jxlamd
