// jxl_coder_amd/csrc/dev_lf_multi.h — the LF coefficients of SIXTEEN LfGroup streams per wavefront (gfx950 only; flights).
//
// libjxl codes the LF image of a VarDCT frame as three Modular channels per 2048 x 2048 LF group with the weighted (self-correcting)
// predictor and an MA tree that only tests the predictor's max-error property: 73 % of the Modular samples of a 4K frame, one serial
// rANS + predictor chain per stream.  The lock-step kernel (dev_modular_wave.h, wave_decode_channel_wpfixed) gives such a stream a whole
// wavefront: four lanes evaluate the four sub-predictors, the other sixty repeat wave-uniform work — ~95 wave-instructions per sample, a
// wave and 33 - 52 KB of LDS held for ~100 ms per stream (what the flights of several contexts queue for).
//
// Here a QUAD of lanes owns a stream — lane k of the quad is sub-predictor k, exactly as in the lock-step loop — and the sixteen quads of
// a wavefront advance sixteen independent streams by one sample per iteration of one loop:
//   * everything that was wave-uniform is now quad-uniform and lives in VGPRs (rANS state, 64-bit bit window, position, neighbourhood);
//     the rarely-taken branches of the lock-step loop (state refill, hybrid-uint extra bits, bit-window refill) become predicated code;
//   * per stream the LDS holds only what the predictor reads from the previous row: two sample rows (int16: modular_16bit_buffers), one
//     row of true errors, one row of the four sub-predictor errors — updated IN PLACE behind the read position (7 KB) — and a 1 KB table
//     max-error -> cluster (the threshold tree of the channel, flattened over [-512, 511]);
//   * alias tables stay where the header parser left them (HBM / L2, one 8-byte load per symbol: k_pass_flat's arrangement).
// ~12 wave-instructions per sample instead of ~95, sixteen times fewer resident waves, no table pool.
//
// A stream this kernel does not cover (prefix codes, LZ77, a tree that is not a threshold tree on property 15 with weighted-predictor
// leaves, thresholds outside [-512, 510], samples beyond int16) is left to the lock-step kernel, which decodes it from the start; the
// others hand it the bit position where the HF metadata begins (lf_scratch tail words, kLfmStatus / kLfmEndBit).  Same integer arithmetic
// as the serial walker (dev_modular.h: wp_predict / wp_update), bit-exact; the GPU parity tests compare flights with single decodes.
#pragma once
#include "dev_vardct.h"

#ifdef __HIPCC__
namespace jxlamd {

constexpr int kLfmStreams = 16;                        // per wavefront
constexpr int kLfmTab = 1024;                         // max-error values -512 .. 511

struct LfmHdr {                                       // what the header phase (quad lane 0) leaves for the quad
  DevBits b;
  DevWP wp;
  const DevTreeNode *tree; int32_t count;
  const uint8_t *ctx_map; const uint32_t *cfg; const DevAlias *alias;
  int32_t log_alpha, num_clusters, num_ctx;
  int32_t ok;                                         // 1: this kernel decodes the stream
  int32_t bw, bh;
  int32_t *scr;
  uint32_t *err;
  int32_t stream_id;
  int32_t nthr;                                       // thresholds of the current channel
};
struct LfmStream {                                    // per stream, LDS
  union {
    struct {
      uint8_t ctab[kLfmTab];                          // cluster of max-error p (index p + 512)
      int16_t rowN[256], rowNN[256];                  // samples of rows y - 1 and y - 2 (row y overwrites y - 1 behind the read position, y - 1 moves to NN)
      int32_t te[256];                                // true errors (prediction - 8 * sample) of row y - 1, row y in place
      uint32_t pe[4][256];                            // sub-predictor errors likewise
      int32_t thr[64]; uint8_t clu[68];               // the channel's thresholds (descending) and the clusters of the intervals between them
      int32_t stack[64];
    } d;
    LocalTmp tmp;                                     // header phase: working arrays of the histogram / alias-table builder
  };
  LfmHdr h;
};
constexpr size_t kLfmLdsBytes = 256 + (size_t)kLfmStreams * sizeof(LfmStream);

__device__ __forceinline__ int lfm_quad_sum(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ int lfm_med3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

// header of one stream (quad lane 0): section open, extra_precision, GroupHeader, (local | global) tree + code
__device__ __forceinline__ void lfm_header(const DevBuffers &B, LfmStream &S, int g) {
  const DevFrame &F = frame_of(B);
  LfmHdr &H = S.h;
  const LfGeom q = lf_geom(F, g);
  H.bw = q.bw; H.bh = q.bh; H.ok = 0; H.err = B.err; H.stream_id = 1 + g;
  int32_t *scr = B.lf_scratch + (size_t)g * kLfScratchInts;
  H.scr = scr;
  scr[kLfmStatus] = 0;
  if (F.nsec == 1 || !F.modular_16bit || *B.err) return;
  const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
  DevBits b;
  bits_init(b, B.codestream, secs[1 + g].off, F.cs_size);
  const int32_t extra_precision = (int32_t)bits_read(b, 2);
  int ntr, use_global;
  modular_read_header(b, H.wp, ntr, use_global);
  if (ntr != 0) return;
  if (use_global) {
    if (F.tree_count <= 0 || F.tree_ec.lz77 || F.tree_ec.use_prefix) return;
    H.tree = (const DevTreeNode *)(B.tables + F.tree_off); H.count = F.tree_count;
    const DevECView v = ec_view(B.tables, F.tree_ec);
    H.ctx_map = v.ctx_map; H.cfg = v.cfg; H.alias = v.alias; H.log_alpha = v.log_alpha;
    H.num_clusters = F.tree_ec.num_clusters; H.num_ctx = F.tree_ec.num_ctx;
  } else {
    LocalTreeScratch &L = B.local[g];
    if (d_read_local_tree(b, L, &S.tmp)) return;                  // (an error: the lock-step kernel meets it again and reports it)
    if (L.leaf_code.use_prefix) return;
    H.tree = L.nodes; H.count = L.count;
    H.ctx_map = L.leaf_code.ctx_map; H.cfg = L.leaf_code.cfg; H.alias = L.leaf_code.alias; H.log_alpha = L.leaf_code.log_alpha;
    H.num_clusters = L.leaf_code.num_clusters; H.num_ctx = L.leaf_code.num_ctx;
  }
  if (H.num_clusters > 64 || H.bw < 1 || H.bh < 1) return;
  H.b = b;
  scr[kLfScratchInts - 1] = extra_precision;
  H.ok = 1;
}

// Threshold tree of channel `chan` (quad lane 0): in-order walk of the pruned tree, greater-than side first.  Accepts only decision nodes
// on property 15 whose thresholds come out strictly descending (a search tree over one variable) inside [-512, 510] and leaves with the
// weighted predictor, multiplier 1, offset 0.  Result: S.d.thr[0 .. n), S.d.clu[0 .. n] — cluster of p is clu[#{i : thr[i] >= p}].
__device__ __forceinline__ bool lfm_tree(LfmStream &S, int chan) {
  const LfmHdr &H = S.h;
  int sp = 0, nthr = 0, nleaf = 0, guard = 0;
  int32_t *stack = S.d.stack;
  // iterative in-order: the stack holds nodes whose greater-than side is being visited (their threshold follows)
  int cur = 0;
  for (;;) {
    if (++guard > 4 * H.count + 64) return false;
    const DevTreeNode nd = H.tree[cur];
    if (nd.prop >= 0) {
      if (nd.prop == 0 || nd.prop == 1) { const int v = nd.prop == 0 ? chan : H.stream_id; cur = v > nd.splitval ? nd.lchild : nd.rchild; continue; }
      if (nd.prop != 15 || sp >= 63) return false;
      stack[sp++] = cur;
      cur = nd.lchild;                                 // p > splitval
      continue;
    }
    // leaf
    if (nd.lchild != 6 || nd.rchild != 1 || nd.offset != 0 || nleaf > 64) return false;
    if (nd.splitval < 0 || nd.splitval >= H.num_ctx) return false;
    S.d.clu[nleaf++] = H.ctx_map[nd.splitval];
    if (sp == 0) break;
    const DevTreeNode up = H.tree[stack[--sp]];
    if (up.splitval < -512 || up.splitval > 510 || nthr >= 64) return false;
    if (nthr > 0 && up.splitval >= S.d.thr[nthr - 1]) return false;
    S.d.thr[nthr++] = up.splitval;
    cur = up.rchild;                                   // p <= splitval
  }
  if (nleaf != nthr + 1) return false;
  S.h.nthr = nthr;
  return true;
}

// All sixty-four lanes.  map: {frame, LF group} per stream, nstreams entries.
__device__ __forceinline__ void lf_coef_multi_body(const DevBuffers *__restrict__ Bs, const int *__restrict__ map, int nstreams, uint8_t *smem) {
  uint32_t *divlut = (uint32_t *)smem;
  LfmStream *SL = (LfmStream *)(smem + 256);
  const int lane = (int)threadIdx.x, qd = lane >> 2, k = lane & 3;
  const int sidx = (int)blockIdx.x * kLfmStreams + qd;
  const bool live = sidx < nstreams;
  LfmStream &S = SL[qd];
  divlut[lane] = (1u << 24) / (uint32_t)(lane + 1);
  if (k == 0) S.h.ok = 0;
  __syncthreads();
  if (live && k == 0) lfm_header(Bs[map[2 * sidx]], S, map[2 * sidx + 1]);
  __syncthreads();
  bool ok = live && S.h.ok != 0;
  // quad-uniform stream state
  uint64_t buf = 0; int nbit = 0; uint32_t ahead = 0; const uint32_t *nextp = nullptr, *endp = nullptr; uint64_t consumed = 0;
  uint32_t state = 0;
  int la = 5;
  const DevAlias *alias = nullptr;
  int wp_p1 = 0, wp_p2 = 0, wp_p3a = 0, wp_p3b = 0, wp_p3c = 0, wp_p3d = 0, wp_p3e = 0, wp_wk = 0;      // (scalars: an indexed struct member would live in scratch)
  int bw = 1, bh = 1;
  int32_t *scr = nullptr;
  if (ok) {
    const LfmHdr &H = S.h;
    buf = H.b.buf; nbit = H.b.n; ahead = H.b.ahead; nextp = H.b.next; endp = H.b.end; consumed = H.b.consumed;
    la = H.log_alpha; alias = H.alias; bw = H.bw; bh = H.bh; scr = H.scr;
    wp_p1 = H.wp.p1; wp_p2 = H.wp.p2; wp_p3a = H.wp.p3a; wp_p3b = H.wp.p3b; wp_p3c = H.wp.p3c; wp_p3d = H.wp.p3d; wp_p3e = H.wp.p3e;
    wp_wk = k == 0 ? H.wp.w[0] : k == 1 ? H.wp.w[1] : k == 2 ? H.wp.w[2] : H.wp.w[3];
  }
  // a bit window of at least 33 valid bits (the caller consumes at most 32 before the next call)
  auto refill = [&]() {
    if (nbit <= 32) {
      buf |= (uint64_t)ahead << nbit;
      nbit += 32;
      ahead = nextp < endp ? *nextp : 0u;
      nextp++;
    }
  };
  auto take = [&](int n) -> uint32_t {       // n <= 32 (0 allowed), quad-uniform
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1ull));
    buf >>= n; nbit -= n; consumed += (uint64_t)n;
    return v;
  };
  if (ok) { refill(); state = take(32); }
  // per-lane constants of sub-predictor k (see wave_decode_channel_wpfixed: same split of the four predictions)
  const int PT = k == 1 ? wp_p1 : k == 2 ? wp_p2 : 0;                       // x tW
  const int PW = k == 3 ? -wp_p3e : 0;                                      // x W8
  const int cA = k == 2 ? wp_p2 : k == 3 ? wp_p3a : 0;                      // x tNW
  const int cB = k == 1 ? wp_p1 : k == 2 ? wp_p2 : k == 3 ? wp_p3b : 0;     // x tN
  const int cC = k == 1 ? wp_p1 : k == 3 ? wp_p3c : 0;                      // x tNE
  const int cD = k == 3 ? wp_p3d : 0;                                       // x (NN8 - N8)
  const int cE = k == 3 ? wp_p3e : 0;                                       // x NW8
  const int cWmask = (k & 1) ? 0 : -1;                                      // sub-predictors 0 and 2 start from W
  const uint32_t wk = (uint32_t)wp_wk;
  const int lb = 12 - la;
  uint32_t errbits = 0;
  for (int chan = 0; chan < 3; chan++) {
    // ---- the channel's threshold tree -> cluster table; clear the previous-row state (row 0 reads zeros)
    __syncthreads();
    if (ok && k == 0) { if (!lfm_tree(S, chan)) S.h.ok = 0; }
    __syncthreads();
    if (ok && S.h.ok == 0) {
      // Not a stream for this kernel after all (only possible before anything was decoded when chan == 0; later channels: the stream is
      // re-decoded from the start by the lock-step kernel, the samples written so far are overwritten)
      ok = false;
    }
    if (ok) {
      const int nthr = S.h.nthr;
      for (int i = k; i < kLfmTab; i += 4) {
        const int p = i - 512;
        int j = 0;
        while (j < nthr && S.d.thr[j] >= p) j++;
        S.d.ctab[i] = S.d.clu[j];
      }
      for (int i = k; i < 256; i += 4) { S.d.rowN[i] = 0; S.d.rowNN[i] = 0; S.d.te[i] = 0; S.d.pe[0][i] = 0; S.d.pe[1][i] = 0; S.d.pe[2][i] = 0; S.d.pe[3][i] = 0; }
    }
    __syncthreads();
    int32_t *out = ok ? scr + (size_t)chan * 65536 : nullptr;
    // ---- the sample loop: every live quad advances its stream by one sample per iteration
    int x = 0, y = 0;
    bool run = ok;
    int32_t W8 = 0, tW = 0;
    uint32_t e1 = 0, e2 = 0;
    int32_t vNW = 0, vN = 0, vNE = 0, tNW = 0, tN = 0, tNE = 0;
    uint32_t pNW = 0, pN = 0, pNE = 0;
    while (__ballot(run)) {
      if (run) {
        const bool row0 = y == 0;
        if (x == 0) {                                      // row start: W is replaced by N (0 in row 0), its error by 0
          vN = S.d.rowN[0]; vNE = bw > 1 ? S.d.rowN[1] : vN; vNW = vN;
          tN = S.d.te[0]; tNE = bw > 1 ? S.d.te[1] : tN; tNW = tN;
          pN = S.d.pe[k][0]; pNE = bw > 1 ? S.d.pe[k][1] : pN; pNW = pN;
          W8 = vN * 8; tW = 0; e1 = 0; e2 = 0;
        }
        const bool has_r = x + 1 < bw;
        // what the previous rows contribute (independent of the sample before this one)
        const int32_t NN = (int32_t)S.d.rowNN[x];
        const int xn = x + 2 < bw ? x + 2 : bw - 1;
        const int32_t nvNE = (int32_t)S.d.rowN[xn];       // prefetch for the next step (x + 2)
        const int32_t ntNE = S.d.te[xn];
        const uint32_t npNE = S.d.pe[k][xn];
        const int32_t aNE = has_r ? vNE : vN, atNE = has_r ? tNE : tN;
        const uint32_t apNE = has_r ? pNE : pN;
        const int32_t N8 = vN * 8, NE8 = aNE * 8, NW8 = vNW * 8, NN8 = (y > 1 ? NN : vN) * 8;
        int32_t qv = tN;
        if (abs(tNW) > abs(qv)) qv = tNW;
        if (abs(atNE) > abs(qv)) qv = atNE;
        const int32_t mxN = row0 ? (int32_t)0x80000000 : (N8 > NE8 ? N8 : NE8), mnN = row0 ? 0x7fffffff : (N8 < NE8 ? N8 : NE8);
        const uint32_t esum = pN + apNE + pNW;                                  // EN + ENE + ENW of the previous row
        const int32_t Ak = k == 0 ? NE8 - N8 : k == 2 ? 0 : N8;
        const int32_t Bk = __mul24(cA, tNW) + __mul24(cB, tN) + __mul24(cC, atNE) + __mul24(cD, NN8 - N8) + __mul24(cE, NW8);
        // ---- context: property 15 = the error of largest magnitude among W, N, NW, NE (W wins ties)
        const int32_t p = abs(qv) > abs(tW) ? qv : tW;
        const uint32_t clu = S.d.ctab[lfm_med3(p, -512, 511) + 512];
        const uint32_t ai = (state >> lb) & ((1u << la) - 1u);
        const DevAlias ent = alias[(clu << la) + ai];                          // L2: cutoff, right symbol, offset, both frequencies
        const uint32_t cfg = S.h.cfg[clu];
        // ---- weighted predictor, sub-predictor k in this lane
        uint32_t e = esum + e1 + e2;
        if (!has_r) e += e1;                               // no NE: the N error sum (which carries e1) counts twice
        int sh = 26 - __builtin_clz(e + 1);
        sh = sh < 0 ? 0 : sh;
        const uint32_t wd = wk * divlut[e >> sh];            // (divlut[0] = 2^24 does not fit a 24-bit multiply)
        const int cwm = row0 ? -1 : cWmask;
        int32_t inner = Bk + __mul24(tW, PT);
        if (!row0) inner += __mul24(W8, PW);
        const int32_t wpk = Ak + (W8 & cwm) - (inner >> 5);
        const int32_t mx = mxN > W8 ? mxN : W8, mn = mnN < W8 ? mnN : W8;
        const bool no_clamp = ((tN ^ tW) | (tN ^ tNW)) > 0;
        uint32_t wgt = 4 + (wd >> sh);
        const uint32_t wsum = (uint32_t)lfm_quad_sum((int)wgt);
        wgt >>= (27 - __builtin_clz(wsum));
        const uint32_t wsum2 = (uint32_t)lfm_quad_sum((int)wgt);
        const uint32_t dv = divlut[wsum2 - 1];
        const int32_t sum = lfm_quad_sum(__mul24(wpk, (int32_t)wgt)) + (int32_t)(wsum2 >> 1) - 1;
        int32_t raw = __mulhi(sum, (int32_t)(dv << 8));                        // (sum * dv) >> 24
        const int32_t cl = lfm_med3(raw, mn, mx);
        raw = no_clamp ? raw : cl;
        // ---- rANS symbol + hybrid uint (predicated: sixteen streams never agree on the rare paths)
        const uint32_t apos = state & ((1u << lb) - 1u);
        const bool right = apos >= ent.cutoff;
        uint32_t u = right ? ent.right : ai;
        const uint32_t off = right ? (uint32_t)ent.off1 + apos : apos;
        const uint32_t freq = right ? ent.freq1 : ent.freq0;
        state = freq * (state >> 12) + off;
        refill();
        {
          const bool need = state < (1u << 16);
          const uint32_t bits16 = take(need ? 16 : 0);
          state = need ? ((state << 16) | bits16) : state;
        }
        {
          const uint32_t split_exp = cfg & 0xff, msb = (cfg >> 8) & 0xff, lsb = (cfg >> 16) & 0xff;
          const bool big = u >= (1u << split_exp);
          uint32_t nbits = split_exp - (msb + lsb) + ((u - (1u << split_exp)) >> (msb + lsb));
          if (nbits > 31) nbits = 31;                      // corrupt stream; the final-state check flags it
          nbits = big ? nbits : 0;
          refill();
          const uint32_t bits = take((int)nbits);
          const uint32_t low = u & ((1u << lsb) - 1u), tok = u >> lsb;
          const uint32_t full = (((((1u << msb) | (tok & ((1u << msb) - 1u))) << nbits) | bits) << lsb) | low;
          u = big ? full : u;
        }
        const int32_t val = unpack_signed(u) + ((raw + 3) >> 3);
        // ---- this sample's errors; previous-row state moves on in place
        const int32_t v8 = val * 8;
        const int32_t d = wpk - v8;
        const uint32_t err = (uint32_t)((abs(d) + 3) >> 3);
        tW = raw - v8; W8 = v8; e2 = e1; e1 = err;
        S.d.pe[k][x] = err;
        if (k == 0) { S.d.rowNN[x] = (int16_t)vN; S.d.rowN[x] = (int16_t)val; S.d.te[x] = tW; out[(size_t)y * (size_t)bw + (size_t)x] = val; }
        if ((uint32_t)(val + 32768) > 65535u) errbits |= kErrNeedGeneral;       // beyond int16 although the image metadata promised it: not for this kernel
        vNW = vN; vN = aNE; vNE = nvNE; tNW = tN; tN = atNE; tNE = ntNE; pNW = pN; pN = apNE; pNE = npNE;
        x++;
        if (x == bw) { x = 0; y++; run = y < bh; }
      }
    }
  }
  // ---- stream end: final state, where the HF metadata begins
  if (ok) {
    if (state != 0x130000u) errbits |= kErrAnsFinal;
    if (k == 0) {
      if (errbits) atomicOr(S.h.err, errbits | kErrStageLf);
      scr[kLfmEndBit] = (int32_t)(uint32_t)consumed; scr[kLfmEndBit + 1] = (int32_t)(uint32_t)(consumed >> 32);
      scr[kLfmStatus] = 1;
    }
  }
}

}  // namespace jxlamd
#endif
