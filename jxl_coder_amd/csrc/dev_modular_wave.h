// jxl_coder_amd/csrc/dev_modular_wave.h — wave-cooperative Modular channel decoder (gfx950 only).
//
// The bit stream is serial, but the cost per sample is dominated by CONTEXT MODELLING, not by the rANS step:
// an MA-tree walk is depth x (node load + property load) of dependent LDS round trips.  Here the whole wavefront
// decodes one stream in lock-step:
//   * every lane keeps the (wave-uniform) bit reader, rANS state and a sliding window of the neighbourhood in
//     registers; only lane 0 stores;
//   * lane i evaluates decision node i of the MA tree (pruned for this channel/stream): one compare per lane,
//     one 64-bit ballot for all decisions; lane j then tests leaf j's (must-be-1, must-be-0) masks and a second
//     ballot names the leaf — 2 ballots instead of a pointer-chasing walk;
//   * the weighted predictor's error state slides through registers (5 LDS loads + 5 stores per sample instead
//     of 21 + 9), divisions are table lookups.
// Falls back to the serial walker (dev_modular.h) when the pruned tree has more than 64 decision nodes or leaves.
// Bit-exact with the serial path (same integer arithmetic); the CPU harness exercises the serial path, the
// -m gpu parity tests exercise this one.
#pragma once
#include "dev_modular.h"
#include <type_traits>

#ifdef __HIPCC__
namespace jxlamd {

// lane 0: flatten the tree reachable for (chan, stream) into ballot form
__device__ __forceinline__ void wave_tree_build(const DevTreeNode *tree, int count, int chan, int stream, DevWaveTree &W) {
  W.ni = 0; W.nl = 0; W.ok = 1; W.uses_wp = 0;
  int32_t *stack_node = W.stack_node; uint64_t *stack_n1 = W.stack_n1, *stack_n0 = W.stack_n0;
  int sp = 0;
  stack_node[0] = 0; stack_n1[0] = 0; stack_n0[0] = 0; sp = 1;
  int guard = 0;
  while (sp > 0) {
    if (++guard > 4 * count + 16) { W.ok = 0; return; }
    --sp;
    const int idx = stack_node[sp]; const uint64_t n1 = stack_n1[sp], n0 = stack_n0[sp];
    const DevTreeNode nd = tree[idx];
    if (nd.prop < 0) {
      if (W.nl >= 64) { W.ok = 0; return; }
      const int j = W.nl++;
      W.leaf_need1[j] = n1; W.leaf_need0[j] = n0;
      W.leaf_ctx[j] = nd.splitval; W.leaf_pred[j] = nd.lchild; W.leaf_off[j] = nd.offset; W.leaf_mul[j] = nd.rchild;
      if (nd.lchild == 6) W.uses_wp = 1;
      continue;
    }
    if (nd.prop == 0 || nd.prop == 1) {
      const int v = nd.prop == 0 ? chan : stream;
      stack_node[sp] = v > nd.splitval ? nd.lchild : nd.rchild; stack_n1[sp] = n1; stack_n0[sp] = n0; sp++;
      continue;
    }
    if (nd.prop > 15) { W.ok = 0; return; }
    if (nd.prop == 15) W.uses_wp = 1;
    if (W.ni >= 64 || sp + 2 > 64) { W.ok = 0; return; }
    const int i = W.ni++;
    W.int_prop[i] = nd.prop; W.int_split[i] = nd.splitval;
    stack_node[sp] = nd.lchild; stack_n1[sp] = n1 | (1ull << i); stack_n0[sp] = n0; sp++;   // decision true  -> left
    stack_node[sp] = nd.rchild; stack_n1[sp] = n1; stack_n0[sp] = n0 | (1ull << i); sp++;   // decision false -> right
  }
}

// rANS symbol + hybrid uint with every table addressed directly in LDS (ds_read instead of flat loads through
// generic pointers); used when the stream's code fits the LDS staging area, which is the case for libjxl's streams.
// `cluster` = ctx_map[ctx], resolved once per leaf at channel start (it used to be a dependent LDS round trip per sample)
template <bool kLds>
__device__ __forceinline__ uint32_t wave_ec_read(const DevECView &v, DevModScratch &S, DevBits &b, uint32_t &state, uint32_t ctx, uint32_t cluster) {
  if (!kLds) return ec_read(v, b, state, ctx);
  const int lb = 12 - v.log_alpha;
  const uint32_t res = state & 0xfff;
  const uint32_t i = res >> lb, pos = res & ((1u << lb) - 1);
  const DevAlias e = ((const DevAlias *)S.pool)[(cluster << v.log_alpha) + i];   // the alias tables open the pool
  const uint32_t cfg = S.cfg[cluster];
  const bool right = pos >= e.cutoff;
  const uint32_t sym = right ? e.right : i;
  const uint32_t off = right ? (uint32_t)e.off1 + pos : pos;
  const uint32_t freq = right ? e.freq1 : e.freq0;
  state = freq * (state >> 12) + off;
  if (state < (1u << 16)) state = (state << 16) | bits_read(b, 16);
  return ec_hybrid(b, cfg, sym);
}

template <class T> __device__ __forceinline__ T tabs(T v) { return v < 0 ? -v : v; }
template <class T>
__device__ __forceinline__ T predict_plain_t(int predictor, T W, T N, T NW, T NE, T NN, T WW, T NEE, T wp) {
  switch (predictor) {
    case 0: return 0;
    case 1: return W;
    case 2: return N;
    case 3: return (W + N) / 2;
    case 4: { T p = W + N - NW; return tabs<T>(p - W) < tabs<T>(p - N) ? W : N; }
    case 5: { T m = N < W ? N : W, M = N < W ? W : N, g = N + W - NW; return g < m ? m : g > M ? M : g; }
    case 6: return wp;
    case 7: return NE;
    case 8: return NW;
    case 9: return WW;
    case 10: return (W + NW) / 2;
    case 11: return (N + NW) / 2;
    case 12: return (N + NE) / 2;
    case 13: return (6 * N - 2 * NN + 7 * W + WW + NEE + 3 * NE + 8) / 16;
  }
  return 0;
}

// One channel, all 64 lanes in lock-step.  kM16: every sample fits int16 (ImageMetadata.modular_16bit_buffers), so the
// neighbourhood, the properties, the predictors and the weighted predictor run in 32-bit arithmetic (exactly the
// same results as the 64-bit reference arithmetic, half the vector instructions).  kWP: this channel's pruned tree
// uses the weighted predictor (property 15 or predictor 6).
template <bool kLds, bool kM16, bool kWP>
__device__ __forceinline__ void wave_decode_channel(const DevECView &ev, DevBits &b, uint32_t &state, const DevWP &wp, DevModScratch &S,
                                                    DevWaveTree &WT, const DevChanOut c, int lane) {
  typedef typename std::conditional<kM16, int32_t, int64_t>::type T;
  const int w = c.w, h = c.h;
  const bool wide = w > kModMaxW;
  const int ni = WT.ni, nl = WT.nl;
  const int my_prop = lane < ni ? WT.int_prop[lane] : 0;
  const int my_split = lane < ni ? WT.int_split[lane] : 0x7fffffff;
  const uint64_t my_need1 = lane < nl ? WT.leaf_need1[lane] : ~0ull;
  const uint64_t my_need0 = lane < nl ? WT.leaf_need0[lane] : ~0ull;
  // lane j keeps leaf j's record in registers; the selected leaf's record is fetched with v_readlane (no LDS trip)
  const int my_lctx = lane < nl ? WT.leaf_ctx[lane] : 0, my_lpred = lane < nl ? WT.leaf_pred[lane] : 0;
  const int my_loff = lane < nl ? WT.leaf_off[lane] : 0, my_lmul = lane < nl ? WT.leaf_mul[lane] : 1;
  const int my_lclu = (kLds && lane < nl) ? (int)((const uint8_t *)S.pool)[S.ctx_off + my_lctx] : 0;
  // the WP's reciprocal table (1<<24)/(i+1), i < 64: lane i holds entry i; lookups are v_readlane with a uniform index
  // kM16: every MA property is a signed sum of per-sample inputs (W, N, NW, NE, NN, WW, the previous sample's property 9,
  // x, y, the WP's max error), two of them under |.|.  Lane i keeps the coefficient row of ITS decision node and splits
  // the sum into the part that is known before the previous sample's value arrives (everything but W and the WP error)
  // and two 24-bit multiply-adds on the critical path — the 13-deep select chain over all properties used to sit there.
  int cW = 0, cN = 0, cNW = 0, cNE = 0, cNN = 0, cWW = 0, cP9 = 0, cX = 0, cY = 0, cE = 0;
  bool cAbs = false;
  switch (my_prop) {
    case 2: cY = 1; break;
    case 3: cX = 1; break;
    case 4: cN = 1; cAbs = true; break;
    case 5: cW = 1; cAbs = true; break;
    case 6: cN = 1; break;
    case 7: cW = 1; break;
    case 8: cW = 1; cP9 = -1; break;
    case 9: cW = 1; cN = 1; cNW = -1; break;
    case 10: cW = 1; cNW = -1; break;
    case 11: cNW = 1; cN = -1; break;
    case 12: cN = 1; cNE = -1; break;
    case 13: cN = 1; cNN = -1; break;
    case 14: cW = 1; cWW = -1; break;
    case 15: cE = 1; break;
    default: break;
  }
  const int my_div = (int)S.divlut[lane];
  #define WAVE_DIV(idx) ((uint32_t)__builtin_amdgcn_readlane(my_div, __builtin_amdgcn_readfirstlane((int)(idx))))
  if (kWP) {
    for (int i = lane; i < 2 * (w + 2); i += 64) { S.wp_err[i] = 0; for (int k = 0; k < 4; k++) S.wp_pred_err[k][i] = 0; }
  }
  __syncthreads();
  for (int y = 0; y < h; y++) {
    int32_t *out = c.d + (size_t)y * (size_t)w;
    int32_t *row = wide ? out : S.rows[y % 3];
    const int32_t *rN = wide ? out - w : S.rows[(y + 2) % 3];
    const int32_t *rNN = wide ? out - 2 * w : S.rows[(y + 1) % 3];
    int32_t vW = 0, vWW = 0;
    int32_t vN = y > 0 ? rN[0] : 0, vNW = vN, vNE = (y > 0 && w > 1) ? rN[1] : vN, vNEE = (y > 0 && w > 2) ? rN[2] : vNE;
    const int cur_row = (y & 1) ? 0 : (w + 2), prev_row = (y & 1) ? (w + 2) : 0;
    uint32_t peNW[4] = {0, 0, 0, 0}, peN[4] = {0, 0, 0, 0}, peNE[4] = {0, 0, 0, 0};
    int32_t teNW = 0, teN = 0, teNE = 0, teW = 0;
    if (kWP) {
      for (int k = 0; k < 4; k++) { peN[k] = S.wp_pred_err[k][prev_row]; peNW[k] = peN[k]; peNE[k] = w > 1 ? S.wp_pred_err[k][prev_row + 1] : peN[k]; }
      teN = S.wp_err[prev_row]; teNW = teN; teNE = w > 1 ? S.wp_err[prev_row + 1] : teN;
    }
    int32_t prev_prop9 = 0;
    for (int x = 0; x < w; x++) {
      const T W_ = x > 0 ? vW : (y > 0 ? vN : 0);
      const T N_ = y > 0 ? vN : W_;
      const T NW_ = (x > 0 && y > 0) ? vNW : W_;
      const T NE_ = (x + 1 < w && y > 0) ? vNE : N_;
      const T NN_ = y > 1 ? rNN[x] : N_;
      const T NEE_ = (x + 2 < w && y > 0) ? vNEE : NE_;
      const T WW_ = x > 1 ? vWW : W_;
      const int32_t nextNEE = (y > 0 && x + 3 < w) ? rN[x + 3] : 0;     // independent of this sample: issued early
      const int32_t p9 = (int32_t)(W_ + N_ - NW_);
      const int32_t p9_prev = prev_prop9;
      int32_t pv[16];
      #define M24(a, b) __mul24((a), (int)(b))
      int32_t early = 0;
      if (kM16) {
        // the raw window registers, not the edge-substituted W_/N_/...: at the image edges the substitutes ARE late values
        early = (M24(cN, N_) + M24(cNW, NW_)) + (M24(cNE, NE_) + M24(cNN, NN_)) + (M24(cWW, WW_) + M24(cP9, p9_prev)) + (M24(cX, x) + M24(cY, y));
      } else {
        pv[2] = y; pv[3] = x;
        pv[4] = (int32_t)tabs<T>(N_); pv[5] = (int32_t)tabs<T>(W_);
        pv[6] = (int32_t)N_; pv[7] = (int32_t)W_;
        pv[8] = (int32_t)(W_ - prev_prop9); pv[9] = p9;
        pv[10] = (int32_t)(W_ - NW_); pv[11] = (int32_t)(NW_ - N_); pv[12] = (int32_t)(N_ - NE_);
        pv[13] = (int32_t)(N_ - NN_); pv[14] = (int32_t)(W_ - WW_);
      }
      pv[15] = 0;
      prev_prop9 = p9;
      T wp_pred = 0, wpred[4] = {0, 0, 0, 0}, wp_raw = 0;
      if (kWP) {
        uint32_t wgt[4];
        for (int k = 0; k < 4; k++) {
          const uint32_t e = peN[k] + (x < w - 1 ? peNE[k] : peN[k]) + (x > 0 ? peNW[k] : peN[k]);
          int shift = floor_log2_u32(e + 1) - 5;
          if (shift < 0) shift = 0;
          wgt[k] = 4 + (((uint32_t)wp.w[k] * WAVE_DIV(e >> shift)) >> shift);
        }
        const T N8 = N_ * 8, W8 = W_ * 8, NE8 = NE_ * 8, NW8 = NW_ * 8, NN8 = NN_ * 8;
        const T tW = x == 0 ? 0 : teW, tN = teN, tNW = x > 0 ? teNW : teN, tNE = x < w - 1 ? teNE : teN;
        const T sumWN = tN + tW;
        T p = tW;
        if (tabs<T>(tN) > tabs<T>(p)) p = tN;
        if (tabs<T>(tNW) > tabs<T>(p)) p = tNW;
        if (tabs<T>(tNE) > tabs<T>(p)) p = tNE;
        pv[15] = (int32_t)p;
        wpred[0] = W8 + NE8 - N8;
        wpred[1] = N8 - (((sumWN + tNE) * wp.p1) >> 5);
        wpred[2] = W8 - (((sumWN + tNW) * wp.p2) >> 5);
        wpred[3] = N8 - ((tNW * wp.p3a + tN * wp.p3b + tNE * wp.p3c + (NN8 - N8) * wp.p3d + (NW8 - W8) * wp.p3e) >> 5);
        uint32_t wsum = wgt[0] + wgt[1] + wgt[2] + wgt[3];
        const int lw = floor_log2_u32(wsum);
        wsum = 0;
        for (int k = 0; k < 4; k++) { wgt[k] >>= lw - 4; wsum += wgt[k]; }
        T sum = (T)(wsum >> 1) - 1;
        for (int k = 0; k < 4; k++) sum += wpred[k] * (T)wgt[k];
        wp_raw = (T)(((int64_t)sum * (int64_t)WAVE_DIV(wsum - 1)) >> 24);
        if (!((((tN ^ tW) | (tN ^ tNW))) > 0)) {
          T mx = W8 > NE8 ? W8 : NE8; if (N8 > mx) mx = N8;
          T mn = W8 < NE8 ? W8 : NE8; if (N8 < mn) mn = N8;
          if (wp_raw > mx) wp_raw = mx;
          if (wp_raw < mn) wp_raw = mn;
        }
        wp_pred = (wp_raw + 3) >> 3;
      }
      // MA tree by ballot: lane i decides node i, lane j tests leaf j, the chosen leaf's record comes by readlane
      int32_t myv;
      if (kM16) {
        myv = early + M24(cW, W_) + M24(cE, pv[15]);
        if (cAbs) myv = myv < 0 ? -myv : myv;
      } else {
        myv = pv[2];
        #pragma unroll
        for (int k = 3; k < 16; k++) myv = my_prop == k ? pv[k] : myv;
      }
      #undef M24
      const uint64_t dec = __ballot(lane < ni && myv > my_split);
      const uint64_t lm = __ballot(lane < nl && (dec & my_need1) == my_need1 && (~dec & my_need0) == my_need0);
      const int leaf = lm ? __builtin_ctzll(lm) : 0;
      const int l_ctx = __builtin_amdgcn_readlane(my_lctx, leaf), l_pred = __builtin_amdgcn_readlane(my_lpred, leaf);
      const int l_off = __builtin_amdgcn_readlane(my_loff, leaf), l_mul = __builtin_amdgcn_readlane(my_lmul, leaf);
      const T guess = predict_plain_t<T>(l_pred, W_, N_, NW_, NE_, NN_, WW_, NEE_, wp_pred);
      const int l_clu = __builtin_amdgcn_readlane(my_lclu, leaf);
      const uint32_t u = wave_ec_read<kLds>(ev, S, b, state, (uint32_t)l_ctx, (uint32_t)l_clu);
      const T res = (T)unpack_signed(u);
      const T val = (l_mul == 1 ? res : res * (T)l_mul) + (T)l_off + guess;      // l_mul is wave-uniform: the multiply is branched around
      if (lane == 0) { row[x] = (int32_t)val; if (!wide) out[x] = (int32_t)val; }
      vWW = vW; vW = (int32_t)val;
      vNW = vN; vN = vNE; vNE = vNEE; vNEE = nextNEE;
      if (kWP) {
        const T v8 = val * 8;
        const int32_t terr = (int32_t)(wp_raw - v8);
        uint32_t err[4];
        for (int k = 0; k < 4; k++) err[k] = (uint32_t)((tabs<T>(wpred[k] - v8) + 3) >> 3);
        if (lane == 0) { S.wp_err[cur_row + x] = terr; for (int k = 0; k < 4; k++) S.wp_pred_err[k][cur_row + x] = err[k]; }
        for (int k = 0; k < 4; k++) { peNW[k] = peN[k]; peN[k] = peNE[k] + err[k]; }   // carry to (x+1) of the previous row
        teNW = teN; teN = teNE; teW = terr;
        if (x + 2 < w) { for (int k = 0; k < 4; k++) peNE[k] = S.wp_pred_err[k][prev_row + x + 2]; teNE = S.wp_err[prev_row + x + 2]; }
        else { for (int k = 0; k < 4; k++) peNE[k] = peN[k]; teNE = teN; }
      }
    }
    __syncthreads();     // row[] written by lane 0 is read by every lane in the next row
  }
  #undef WAVE_DIV
}

// All 64 lanes call this with identical arguments.  Returns error bits (uniform).
template <bool kLds>
__device__ __forceinline__ uint32_t modular_decode_channels_wave(const DevECView &ev, DevBits &b, uint32_t &state, const DevTreeNode *gtree,
                                                        int tree_count, const DevWP &wp, DevModScratch &S, DevWaveTree &WT,
                                                        const DevChanOut *chans, int nch, int stream_id, int lane, bool m16) {
  for (int ci = 0; ci < nch; ci++) {
    const DevChanOut c = chans[ci];
    if (c.w == 0 || c.h == 0) continue;
    __syncthreads();
    if (lane == 0) wave_tree_build(gtree, tree_count, ci, stream_id, WT);
    __syncthreads();
    if (!WT.ok) return kErrWaveFallback;                    // caller re-runs the stream with the serial walker
    const bool uses_wp = WT.uses_wp != 0;
    if (c.w > kModMaxW && uses_wp) return kErrUnsupportedTransform;
    if (m16) { if (uses_wp) wave_decode_channel<kLds, true, true>(ev, b, state, wp, S, WT, c, lane); else wave_decode_channel<kLds, true, false>(ev, b, state, wp, S, WT, c, lane); }
    else { if (uses_wp) wave_decode_channel<kLds, false, true>(ev, b, state, wp, S, WT, c, lane); else wave_decode_channel<kLds, false, false>(ev, b, state, wp, S, WT, c, lane); }
  }
  return 0;
}

// Stream-level wrapper: every lane calls it; falls back to the serial walker when the tree is too large.
__device__ __forceinline__ uint32_t modular_stream_decode_wave(DevModScratch &S, const DevChanOut *chans, int nch, int stream_id, int lane) {
  DevModStream &st = S.st;
  if (st.err) return st.err;
  DevECView ev = st.ev;
  ev.ctx_map = S.ctx_map;
  if (st.num_clusters <= kLocMaxClusters) ev.cfg = S.cfg;
  if (!ev.use_prefix) ev.alias = S.alias;
  DevBits b = st.b;
  uint32_t state = ans_init(ev, b);
  const bool lds = S.ctx_lds && S.alias_lds && !ev.use_prefix;
  uint32_t err = lds ? modular_decode_channels_wave<true>(ev, b, state, st.tree, st.count, st.wp, S, S.wt, chans, nch, stream_id, lane, st.m16 != 0)
                     : modular_decode_channels_wave<false>(ev, b, state, st.tree, st.count, st.wp, S, S.wt, chans, nch, stream_id, lane, st.m16 != 0);
  __syncthreads();
  if (err == kErrWaveFallback) {
    if (lane == 0) S.fallback_err = modular_stream_decode(S, chans, nch, stream_id);
    __syncthreads();
    return S.fallback_err;
  }
  if (!err && state != 0x130000u) err |= kErrAnsFinal;
  if (lane == 0) st.b = b;
  __syncthreads();
  return err;
}

}  // namespace jxlamd
#endif
