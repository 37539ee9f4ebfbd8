// jxl_coder_amd/csrc/dev_lf_simt.h — lane-per-stream (SIMT) decode of the LfGroup sections of a flight of frames.
//
// The wave-per-stream kernel (dev_modular_wave.h) is the latency path: one wavefront, 40 KB of LDS and 128 VGPRs per serial
// stream.  A flight of 128 4K frames has 512 such streams and they hold their CU resources for the whole LF stage, which is
// what starves the data-parallel kernels of the other decoder contexts.  Here every LANE decodes one LfGroup section (its
// LF-coefficient stream, its HF-metadata stream and the varblock placement), 64 sections per wavefront, so the LF stage of a
// 256-frame batch is 16 wavefronts.  What makes that work:
//   * lanes of a wavefront get sections of the SAME geometry (the host sorts the (frame, LF group) map), so they run in
//     lock-step over (channel, y, x) and every access to the per-stream state is one coalesced 256-byte row of a
//     lane-interleaved (SoA) buffer: LfSimtWave, 857 KB per wavefront, L2-resident;
//   * LDS is per-lane private through the dword-interleaved layout priv[i][lane]: bank = lane for any per-lane index, so
//     table lookups with different indices per lane never conflict.  It holds the context LUT of the weighted-predictor
//     channels (leaf cluster as a function of the max-error property), then the pruned MA tree of the metadata channels,
//     then the placement's occupancy window;
//   * same integer arithmetic as dev_modular.h (bit-exact); streams the fast paths do not cover (other tree shapes, prefix
//     codes, 32-bit samples) fall back to the serial walker on the lane's own DevModScratch in HBM.
// Per-lane plain C: the CPU harness (tests/emul, test-only) runs it one lane at a time.
#pragma once
#include "dev_vardct.h"

namespace jxlamd {

constexpr int kSimtPriv = 512;                       // dwords of private LDS per lane
struct LfSimtLds {                                   // 128 KB + 256 B per wavefront
  uint32_t priv[kSimtPriv][64];
  uint32_t divlut[64];                               // (1 << 24) / (i + 1): 64 entries = 64 banks, conflict-free for any indices
};
struct LfSimtWave {                                  // HBM, one per wavefront: lane-interleaved rows of the current channel
  int32_t vals[3][kModMaxW][64];                     // rows y, y - 1, y - 2 (ring by y % 3)
  int32_t te[2][kModMaxW + 2][64];                   // weighted predictor: true error, rows y / y - 1 (ring by y & 1)
  uint32_t pe[4][2][kModMaxW + 2][64];               // weighted predictor: sub-predictor errors
};

JXL_DEV int32_t sabs32(int32_t v) { return v < 0 ? -v : v; }

// ------------------------------------------------------------------ weighted-predictor channels (LF coefficients)
// Sweep the (channel, stream)-pruned tree over the whole range of property 15: every decision must test property 15, every leaf
// must be (weighted predictor, multiplier 1, offset 0), every threshold must lie in [-512, 510].  Fills the lane's LUT
// lut[p + 512] = cluster of the leaf for max-error p (p clamped to [-512, 511]).  Returns false when the fast path does not apply.
JXL_DEV bool simt_wp_lut_build(const DevTreeNode *tree, int count, int chan, int stream, const uint8_t *ctx_map, LfSimtLds &L, int lane) {
  int64_t p = -(int64_t)0x80000000ll;
  int guard = 0;
  while (p <= 0x7fffffffll) {
    if (++guard > 4096) return false;
    int64_t next = 0x80000000ll;                     // first p' > p whose walk may differ
    int idx = 0, depth = 0;
    for (;;) {
      if (idx < 0 || idx >= count || ++depth > 4096) return false;
      const DevTreeNode nd = tree[idx];
      if (nd.prop < 0) {
        if (nd.lchild != 6 || nd.rchild != 1 || nd.offset != 0) return false;
        const uint32_t clu = ctx_map[nd.splitval];
        const int64_t lo = p < -512 ? -512 : p, hi = next - 1 > 511 ? 511 : next - 1;
        for (int64_t v = lo; v <= hi; v++) {
          const int i = (int)v + 512;
          uint32_t &wd = L.priv[i >> 2][lane];
          wd = (wd & ~(0xffu << ((i & 3) * 8))) | (clu << ((i & 3) * 8));
        }
        break;
      }
      if (nd.prop == 0 || nd.prop == 1) { idx = (nd.prop == 0 ? chan : stream) > nd.splitval ? nd.lchild : nd.rchild; continue; }
      if (nd.prop != 15) return false;
      if (nd.splitval < -512 || nd.splitval > 510) return false;
      if (p > nd.splitval) idx = nd.lchild;
      else { if ((int64_t)nd.splitval + 1 < next) next = (int64_t)nd.splitval + 1; idx = nd.rchild; }
    }
    p = next;
  }
  return true;
}

// One weighted-predictor channel of one lane's stream.  Rows live in W (lane-interleaved), the decoded plane goes to c.d.
// kLdsAlias: not used (alias tables are read through L2 from the stream's own tables).
JXL_DEV void simt_wp_channel(const DevECView &ev, DevBits &b, uint32_t &state, const DevWP &wp, LfSimtWave &W, LfSimtLds &L, const DevChanOut c, int lane) {
  const int w = c.w, h = c.h;
  const int la = ev.log_alpha, lb = 12 - la;
  const uint32_t posmask = (1u << lb) - 1;
  for (int i = 0; i < w + 2; i++) { W.te[0][i][lane] = 0; W.te[1][i][lane] = 0; for (int k = 0; k < 4; k++) { W.pe[k][0][i][lane] = 0; W.pe[k][1][i][lane] = 0; } }
  for (int y = 0; y < h; y++) {
    const int rc = y % 3, rn = (y + 2) % 3, rnn = (y + 1) % 3;
    const int cur = y & 1, prv = cur ^ 1;
    int32_t *out = c.d + (size_t)y * (size_t)w;
    int32_t vW = 0;
    int32_t vN = y > 0 ? W.vals[rn][0][lane] : 0, vNW = vN, vNE = (y > 0 && w > 1) ? W.vals[rn][1][lane] : vN;
    uint32_t peN[4], peNW[4], peNE[4];
    int32_t teN, teNW, teNE, teW = 0;
    for (int k = 0; k < 4; k++) { peN[k] = W.pe[k][prv][0][lane]; peNW[k] = peN[k]; peNE[k] = w > 1 ? W.pe[k][prv][1][lane] : peN[k]; }
    teN = W.te[prv][0][lane]; teNW = teN; teNE = w > 1 ? W.te[prv][1][lane] : teN;
    for (int x = 0; x < w; x++) {
      // loads for the next sample (independent of this one's value): the row above at x + 2
      const bool more = x + 2 < w;
      const int32_t nxtNE = (y > 0 && more) ? W.vals[rn][x + 2][lane] : 0;
      const int32_t nxt_te = more ? W.te[prv][x + 2][lane] : 0;
      uint32_t nxt_pe[4];
      for (int k = 0; k < 4; k++) nxt_pe[k] = more ? W.pe[k][prv][x + 2][lane] : 0u;
      const int32_t W_ = x > 0 ? vW : (y > 0 ? vN : 0);
      const int32_t N_ = y > 0 ? vN : W_;
      const int32_t NW_ = (x > 0 && y > 0) ? vNW : W_;
      const int32_t NE_ = (x + 1 < w && y > 0) ? vNE : N_;
      const int32_t NN_ = y > 1 ? W.vals[rnn][x][lane] : N_;
      uint32_t wgt[4];
      for (int k = 0; k < 4; k++) {
        const uint32_t e = peN[k] + (x < w - 1 ? peNE[k] : peN[k]) + (x > 0 ? peNW[k] : peN[k]);
        int shift = floor_log2_u32(e + 1) - 5;
        if (shift < 0) shift = 0;
        wgt[k] = 4 + (((uint32_t)wp.w[k] * L.divlut[e >> shift]) >> shift);
      }
      const int32_t N8 = N_ * 8, W8 = W_ * 8, NE8 = NE_ * 8, NW8 = NW_ * 8, NN8 = NN_ * 8;
      const int32_t tW = x == 0 ? 0 : teW, tN = teN, tNW = x > 0 ? teNW : teN, tNE = x < w - 1 ? teNE : teN;
      const int32_t sumWN = tN + tW;
      int32_t p = tW;
      if (sabs32(tN) > sabs32(p)) p = tN;
      if (sabs32(tNW) > sabs32(p)) p = tNW;
      if (sabs32(tNE) > sabs32(p)) p = tNE;
      int32_t wpred[4];
      wpred[0] = W8 + NE8 - N8;
      wpred[1] = N8 - (((sumWN + tNE) * wp.p1) >> 5);
      wpred[2] = W8 - (((sumWN + tNW) * wp.p2) >> 5);
      wpred[3] = N8 - ((tNW * wp.p3a + tN * wp.p3b + tNE * wp.p3c + (NN8 - N8) * wp.p3d + (NW8 - W8) * wp.p3e) >> 5);
      uint32_t wsum = wgt[0] + wgt[1] + wgt[2] + wgt[3];
      const int lw = floor_log2_u32(wsum);
      wsum = 0;
      for (int k = 0; k < 4; k++) { wgt[k] >>= lw - 4; wsum += wgt[k]; }
      int32_t sum = (int32_t)(wsum >> 1) - 1;
      for (int k = 0; k < 4; k++) sum += wpred[k] * (int32_t)wgt[k];
      int32_t raw = (int32_t)(((int64_t)sum * (int64_t)L.divlut[wsum - 1]) >> 24);
      if (!(((tN ^ tW) | (tN ^ tNW)) > 0)) {
        int32_t mx = W8 > NE8 ? W8 : NE8; if (N8 > mx) mx = N8;
        int32_t mn = W8 < NE8 ? W8 : NE8; if (N8 < mn) mn = N8;
        if (raw > mx) raw = mx;
        if (raw < mn) raw = mn;
      }
      // context -> cluster through the lane's LUT, then one rANS step
      const int pc = (p < -512 ? -512 : p > 511 ? 511 : p) + 512;
      const uint32_t clu = (L.priv[pc >> 2][lane] >> ((pc & 3) * 8)) & 0xff;
      const uint32_t res = state & 0xfff;
      const uint32_t ai = res >> lb, apos = res & posmask;
      const DevAlias a = ev.alias[(clu << la) + ai];
      const uint32_t cfg = ev.cfg[clu];
      const bool right = apos >= a.cutoff;
      const uint32_t sym = right ? a.right : ai;
      const uint32_t off = right ? (uint32_t)a.off1 + apos : apos;
      const uint32_t freq = right ? a.freq1 : a.freq0;
      state = freq * (state >> 12) + off;
      if (state < (1u << 16)) state = (state << 16) | bits_read(b, 16);
      const uint32_t u = ec_hybrid(b, cfg, sym);
      const int32_t val = unpack_signed(u) + ((raw + 3) >> 3);
      W.vals[rc][x][lane] = val;
      out[x] = val;
      const int32_t v8 = val * 8;
      const int32_t terr = raw - v8;
      W.te[cur][x][lane] = terr;
      for (int k = 0; k < 4; k++) {
        const uint32_t err = (uint32_t)((sabs32(wpred[k] - v8) + 3) >> 3);
        W.pe[k][cur][x][lane] = err;
        peNW[k] = peN[k]; peN[k] = peNE[k] + err;        // the error also counts for position x + 1 of the row above
        peNE[k] = more ? nxt_pe[k] : peN[k];
      }
      teNW = teN; teN = teNE; teW = terr; teNE = more ? nxt_te : teN;
      vW = val; vNW = vN; vN = vNE; vNE = nxtNE;
    }
  }
}

// ------------------------------------------------------------------ channels without the weighted predictor (HF metadata)
// Pruned tree of one channel in the lane's LDS: 3 dwords per node — interior: prop, splitval, lchild | rchild << 16;
// leaf: 0x80000000 | predictor << 8 | cluster, offset, multiplier.  The DFS stack sits at the end of the private region.
constexpr int kSimtTreeNodes = 150, kSimtTreeStack = kSimtPriv - 3 * kSimtTreeNodes;     // 62 stack entries
JXL_DEV int simt_tree_build(const DevTreeNode *tree, int count, int chan, int stream, const uint8_t *ctx_map, LfSimtLds &L, int lane, int *max_prop) {
  int n = 0, sp = 0, guard = 0;
  *max_prop = 0;
  #define SIMT_STACK(i) L.priv[3 * kSimtTreeNodes + (i)][lane]
  // stack entry: source node index (15 bits) | is-right-child << 15 | pruned index of the parent << 16 | has-parent << 31; the parent's child
  // field is patched when the child is emitted
  SIMT_STACK(sp++) = 0u;                               // the root: no parent field to patch
  while (sp > 0) {
    if (++guard > 8 * count + 64) return -1;
    const uint32_t ent = SIMT_STACK(--sp);
    int idx = (int)(ent & 0x7fff); const int parent = (int)((ent >> 16) & 0x7fff); const bool is_right = (ent >> 15) & 1, has_parent = (ent >> 31) != 0;
    // follow static decisions (properties 0 and 1) straight away
    DevTreeNode nd;
    for (;;) {
      if (idx < 0 || idx >= count) return -1;
      nd = tree[idx];
      if (nd.prop == 0 || nd.prop == 1) { idx = (nd.prop == 0 ? chan : stream) > nd.splitval ? nd.lchild : nd.rchild; if (++guard > 8 * count + 64) return -1; continue; }
      break;
    }
    if (n >= kSimtTreeNodes) return -1;
    const int me = n++;
    if (has_parent) { uint32_t &pw = L.priv[3 * parent + 2][lane]; pw = is_right ? (pw & 0xffffu) | ((uint32_t)me << 16) : (pw & 0xffff0000u) | (uint32_t)me; }
    if (nd.prop < 0) {
      if (nd.lchild == 6) return -1;                                   // weighted predictor: not this path
      L.priv[3 * me][lane] = 0x80000000u | ((uint32_t)nd.lchild << 8) | ctx_map[nd.splitval];
      L.priv[3 * me + 1][lane] = (uint32_t)nd.offset; L.priv[3 * me + 2][lane] = (uint32_t)nd.rchild;
      continue;
    }
    if (nd.prop >= 15 || nd.lchild > 0x7fff || nd.rchild > 0x7fff) return -1;
    if (nd.prop > *max_prop) *max_prop = nd.prop;
    L.priv[3 * me][lane] = (uint32_t)nd.prop; L.priv[3 * me + 1][lane] = (uint32_t)nd.splitval; L.priv[3 * me + 2][lane] = 0;
    if (sp + 2 > kSimtTreeStack) return -1;
    SIMT_STACK(sp++) = (uint32_t)nd.rchild | (1u << 15) | ((uint32_t)me << 16) | (1u << 31);
    SIMT_STACK(sp++) = (uint32_t)nd.lchild | ((uint32_t)me << 16) | (1u << 31);
  }
  #undef SIMT_STACK
  return n;
}

// One channel without weighted predictor.  Narrow channels (<= kModMaxW) keep their rows in W; the wide block-info channel
// (count x 2) reads its neighbourhood back from the plane it writes.
JXL_DEV void simt_plain_channel(const DevECView &ev, DevBits &b, uint32_t &state, LfSimtWave &W, LfSimtLds &L, const DevChanOut c, int lane) {
  const int w = c.w, h = c.h;
  const bool wide = w > kModMaxW;
  const int la = ev.log_alpha, lb = 12 - la;
  const uint32_t posmask = (1u << lb) - 1;
  for (int y = 0; y < h; y++) {
    const int rc = y % 3, rn = (y + 2) % 3, rnn = (y + 1) % 3;
    int32_t *out = c.d + (size_t)y * (size_t)w;
    #define ROW_N(xx) (wide ? out[(xx) - w] : W.vals[rn][(xx)][lane])
    #define ROW_NN(xx) (wide ? out[(xx) - 2 * w] : W.vals[rnn][(xx)][lane])
    int32_t vW = 0, vWW = 0;
    int32_t vN = y > 0 ? ROW_N(0) : 0, vNW = vN, vNE = (y > 0 && w > 1) ? ROW_N(1) : vN, vNEE = (y > 0 && w > 2) ? ROW_N(2) : vNE;
    int32_t prev_prop9 = 0;
    for (int x = 0; x < w; x++) {
      const int32_t nextNEE = (y > 0 && x + 3 < w) ? ROW_N(x + 3) : 0;
      const int64_t W_ = x > 0 ? vW : (y > 0 ? vN : 0);
      const int64_t N_ = y > 0 ? vN : W_;
      const int64_t NW_ = (x > 0 && y > 0) ? vNW : W_;
      const int64_t NE_ = (x + 1 < w && y > 0) ? vNE : N_;
      const int64_t NN_ = y > 1 ? ROW_NN(x) : N_;
      const int64_t NEE_ = (x + 2 < w && y > 0) ? vNEE : NE_;
      const int64_t WW_ = x > 1 ? vWW : W_;
      const int32_t p9 = (int32_t)(W_ + N_ - NW_);
      // walk the lane's pruned tree (LDS)
      int node = 0;
      uint32_t w0 = L.priv[0][lane];
      while (!(w0 & 0x80000000u)) {
        const int32_t split = (int32_t)L.priv[3 * node + 1][lane];
        const uint32_t kids = L.priv[3 * node + 2][lane];
        int32_t v;
        switch (w0) {
          case 2: v = y; break;
          case 3: v = x; break;
          case 4: v = (int32_t)iabs64(N_); break;
          case 5: v = (int32_t)iabs64(W_); break;
          case 6: v = (int32_t)N_; break;
          case 7: v = (int32_t)W_; break;
          case 8: v = (int32_t)(W_ - prev_prop9); break;
          case 9: v = p9; break;
          case 10: v = (int32_t)(W_ - NW_); break;
          case 11: v = (int32_t)(NW_ - N_); break;
          case 12: v = (int32_t)(N_ - NE_); break;
          case 13: v = (int32_t)(N_ - NN_); break;
          default: v = (int32_t)(W_ - WW_); break;      // 14
        }
        node = v > split ? (int)(kids & 0xffff) : (int)(kids >> 16);
        w0 = L.priv[3 * node][lane];
      }
      prev_prop9 = p9;
      const int32_t l_off = (int32_t)L.priv[3 * node + 1][lane];
      const uint32_t l_mul = L.priv[3 * node + 2][lane];
      const int64_t guess = predict_plain((int)((w0 >> 8) & 0xff), W_, N_, NW_, NE_, NN_, WW_, NEE_, 0);
      const uint32_t clu = w0 & 0xff;
      const uint32_t res = state & 0xfff;
      const uint32_t ai = res >> lb, apos = res & posmask;
      const DevAlias a = ev.alias[(clu << la) + ai];
      const uint32_t cfg = ev.cfg[clu];
      const bool right = apos >= a.cutoff;
      const uint32_t sym = right ? a.right : ai;
      const uint32_t off = right ? (uint32_t)a.off1 + apos : apos;
      const uint32_t freq = right ? a.freq1 : a.freq0;
      state = freq * (state >> 12) + off;
      if (state < (1u << 16)) state = (state << 16) | bits_read(b, 16);
      const uint32_t u = ec_hybrid(b, cfg, sym);
      const int64_t val = (int64_t)unpack_signed(u) * (int64_t)l_mul + l_off + guess;
      out[x] = (int32_t)val;
      if (!wide) W.vals[rc][x][lane] = (int32_t)val;
      vWW = vW; vW = (int32_t)val;
      vNW = vN; vN = vNE; vNE = vNEE; vNEE = nextNEE;
    }
    #undef ROW_N
    #undef ROW_NN
  }
}

// All channels of the stream that S.st describes (header and tree already parsed by modular_stream_begin).  Returns error bits.
JXL_DEV uint32_t simt_stream_decode(DevModScratch &S, LfSimtWave &W, LfSimtLds &L, const DevChanOut *chans, int nch, int stream_id, int lane) {
  DevModStream &st = S.st;
  if (st.err) return st.err;
  const DevECView ev = st.ev;                          // tables through L2: the stream's own (local) or the frame's (global)
  bool fast = !ev.use_prefix && st.m16 != 0;
  for (int ci = 0; ci < nch && fast; ci++) if (chans[ci].w > kModMaxW && chans[ci].h > 2) fast = false;
  if (!fast) {                                         // serial walker on the lane's own scratch (tables unstaged)
    S.alias = st.ev.alias; S.alias_lds = 0; S.ctx_map = st.ev.ctx_map; S.ctx_lds = 0; S.tree = st.tree; S.tree_ncache = 0;
    for (int i = 0; i < 64; i++) S.divlut[i] = (1u << 24) / (uint32_t)(i + 1);
    for (int i = 0; i < st.num_clusters && i < kLocMaxClusters; i++) S.cfg[i] = st.ev.cfg[i];
    return modular_stream_decode(S, chans, nch, stream_id);
  }
  DevBits b = st.b;
  uint32_t state = ans_init(ev, b);
  uint32_t err = 0;
  for (int ci = 0; ci < nch && !err; ci++) {
    const DevChanOut c = chans[ci];
    if (c.w == 0 || c.h == 0) continue;
    int max_prop = 0;
    const int n = simt_tree_build(st.tree, st.count, ci, stream_id, ev.ctx_map, L, lane, &max_prop);
    if (n > 0) { simt_plain_channel(ev, b, state, W, L, c, lane); continue; }
    if (c.w <= kModMaxW && simt_wp_lut_build(st.tree, st.count, ci, stream_id, ev.ctx_map, L, lane)) { simt_wp_channel(ev, b, state, st.wp, W, L, c, lane); continue; }
    err = kErrUnsupportedTransform | (1u << 21);       // a tree neither fast path covers (the host routes such flights to the wave kernel)
  }
  if (!err && state != 0x130000u) err |= kErrAnsFinal;
  st.b = b;
  return err;
}

// ------------------------------------------------------------------ varblock placement, one lane
// Same raster rule as lf_phase_place; the occupancy bitmap is a 64-row window in the lane's private LDS (a varblock spans at most
// 32 cell rows), rows are cleared as the scan leaves them.
JXL_DEV uint32_t simt_place(const DevBuffers &B, DevModScratch &S, LfSimtLds &L, int g, uint64_t *end_bits, int lane) {
  const DevFrame &F = frame_of(B);
  const LfGeom q = lf_geom(F, g);
  const int bx0 = q.bx0, by0 = q.by0, bw = q.bw, bh = q.bh;
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  const int count = scr[kLfScratchInts - 2];
  const int32_t *m_blk = scr + 3 * 65536 + 2048;
  const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
  const DevSection sec = secs[F.nsec == 1 ? 0 : 1 + g];
  if (end_bits) end_bits[g] = S.st.b.consumed;
  if (S.st.b.consumed > (uint64_t)sec.size * 8 + 64 && F.nsec != 1) return kErrBitstream;
  #define OCC(yy, ww) L.priv[(((yy) & 63) << 3) + (ww)][lane]
  for (int i = 0; i < kSimtPriv; i++) L.priv[i][lane] = 0;
  int num = 0;
  for (int y = 0; y < bh; y++) {
    for (int wx = 0; wx < (bw + 31) / 32; wx++) {
      for (;;) {
        uint32_t freebits = ~OCC(y, wx);
        if (wx * 32 + 32 > bw) freebits &= (1u << (bw - wx * 32)) - 1u;
        if (!freebits) break;
        const int x = wx * 32 + __builtin_ctz(freebits);
        if (num >= count) return kErrBitstream;
        const int st = m_blk[num], qf = m_blk[count + num];
        num++;
        if (st < 0 || st > 26 || qf < 0 || qf > 255) return kErrBitstream;
        const int cx = kCoveredX[st], cy = kCoveredY[st];
        if (x + cx > bw || y + cy > bh) return kErrBitstream;
        if ((x & 31) + cx > 32 || (y & 31) + cy > 32) return kErrBitstream;   // must not straddle a 256x256 group
        const uint32_t mask = (cx == 32 ? 0xFFFFFFFFu : ((1u << cx) - 1u)) << (x & 31);
        const size_t o = (size_t)(by0 + y) * (size_t)F.xb + (size_t)(bx0 + x);
        for (int iy = 0; iy < cy; iy++) if (OCC(y + iy, wx) & mask) return kErrBitstream;
        for (int iy = 0; iy < cy; iy++) OCC(y + iy, wx) |= mask;
        for (int iy = 0; iy < cy; iy++)
          for (int ix = 0; ix < cx; ix++) {
            const size_t oo = o + (size_t)iy * (size_t)F.xb + (size_t)ix;
            B.strategy[oo] = (uint8_t)st; B.first[oo] = (uint8_t)((iy | ix) == 0); B.qfm1[oo] = (uint8_t)qf;
          }
        const int ncoef = cx * cy * 64;
        if (ncoef <= 4096) {
          const int cls = ncoef <= 256 ? 2 : ncoef <= 1024 ? 0 : 1;
#ifdef __HIPCC__
          const uint32_t slot = atomicAdd(&B.big_count[cls], 1u);
#else
          const uint32_t slot = B.big_count[cls]++;
#endif
          B.big_list[cls][slot] = (uint32_t)o;
        }
      }
    }
    for (int wx = 0; wx < 8; wx++) OCC(y, wx) = 0;     // row y leaves the window: its slot serves row y + 64
  }
  #undef OCC
  return 0;
}

// ------------------------------------------------------------------ one LfGroup section, one lane
JXL_DEV uint32_t lf_group_lane(const DevBuffers &B, const DevAux &A, DevModScratch &S, LfSimtWave &W, LfSimtLds &L, int g, int lane) {
  const DevFrame &F = frame_of(B);
  lf_phase_open(B, S, g);
  const LfGeom q = lf_geom(F, g);
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  DevChanOut *ch = S.ch;
  for (int c = 0; c < 3; c++) { ch[c].d = scr + (size_t)c * 65536; ch[c].w = q.bw; ch[c].h = q.bh; }
  uint32_t e = simt_stream_decode(S, W, L, ch, 3, 1 + g, lane);
  if (!e) e = lf_phase_meta_open(B, S, g);
  if (e) return e | kErrStageLf;
  const int count = scr[kLfScratchInts - 2];
  int32_t *m_x = scr + 3 * 65536, *m_b = m_x + 1024, *m_blk = m_b + 1024, *m_sharp = m_blk + 2 * 65536;
  ch[0].d = m_x; ch[0].w = q.tw; ch[0].h = q.th;
  ch[1].d = m_b; ch[1].w = q.tw; ch[1].h = q.th;
  ch[2].d = m_blk; ch[2].w = count; ch[2].h = 2;
  ch[3].d = m_sharp; ch[3].w = q.bw; ch[3].h = q.bh;
  e = simt_stream_decode(S, W, L, ch, 4, 1 + 2 * F.num_lf_groups + g, lane);
  if (!e) e = simt_place(B, S, L, g, A.lf_end_bits, lane);
  return e ? (e | kErrStageLf | (1u << 20)) : 0;
}

}  // namespace jxlamd
