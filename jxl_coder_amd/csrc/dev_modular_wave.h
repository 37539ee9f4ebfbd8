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
#ifndef JXL_MOD_CHAIN
#define JXL_MOD_CHAIN 1      // the chain form of the lean loop (0: an A/B build without it)
#endif
namespace jxlamd {

// lane 0: flatten the tree reachable for (chan, stream) into ballot form
__device__ __forceinline__ void wave_tree_build(const DevTreeNode *tree, int count, int chan, int stream, DevWaveTree &W, int32_t *stack_node, uint64_t *stack_n1, uint64_t *stack_n0) {
  W.ni = 0; W.nl = 0; W.ok = 1; W.uses_wp = 0;
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

// rANS symbol + hybrid uint with the alias tables in HBM (a code of more clusters than the LDS pool holds: libjxl's one-shot encoder writes 128 for an RGBA
// photograph): alias entry and configuration as two global loads issued together; `cluster` comes from the tree's exit word
__device__ __forceinline__ uint32_t wave_ec_read_global(const DevECView &v, DevModScratch &S, DevBits &b, uint32_t &state, uint32_t cluster, bool cfg_lds) {
  if (v.use_prefix) { const uint32_t token = ec_token(v, b, state, cluster); return ec_hybrid(b, v.cfg[cluster], token); }
  const int lb = 12 - v.log_alpha;
  const uint32_t res = state & 0xfff;
  const uint32_t i = res >> lb, pos = res & ((1u << lb) - 1);
  const uint64_t raw8 = gld((const uint64_t *)&v.alias[(cluster << v.log_alpha) + i]);
  const uint32_t raw_x = (uint32_t)raw8, raw_y = (uint32_t)(raw8 >> 32);
  const uint32_t cfg = cfg_lds ? S.cfg[cluster] : gld(&S.st.ev.cfg[cluster]);
  const uint32_t cutoff = raw_x & 255u, rsym = (raw_x >> 8) & 255u, off1 = raw_x >> 16, freq0 = raw_y & 0xffffu, freq1 = raw_y >> 16;      // DevAlias, field by field
  const bool right = pos >= cutoff;
  const uint32_t sym = right ? rsym : i;
  const uint32_t off = right ? off1 + pos : pos;
  const uint32_t freq = right ? freq1 : freq0;
  state = freq * (state >> 12) + off;
  if (state < (1u << 16)) state = (state << 16) | bits_read(b, 16);
  return ec_hybrid(b, cfg, sym);
}

template <class T> __device__ __forceinline__ T tabs(T v) { return v < 0 ? -v : v; }
// sum over the four lanes of every quad, in each of them (two DPP quad-permute adds)
__device__ __forceinline__ int quad_sum_i32(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
  return v;
}
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
// kBig: the channel's tree is in block form (big_tree_build) in the LDS area `big`: block 0 sits in registers like a small tree, an exit that names
// another block costs one more round (that block's nodes and masks from LDS, the property values by ds_bpermute from a lane-indexed vector).
template <bool kLds, bool kM16, bool kWP, bool kBig = false>
__device__ __forceinline__ void wave_decode_channel(const DevECView &ev, DevBits &b, uint32_t &state, const DevWP &wp, DevModScratch &S,
                                                    DevWaveTree &WT, const DevChanOut c, int lane, int y_end = 0x7fffffff, const uint32_t *big = nullptr) {
  typedef typename std::conditional<kM16, int32_t, int64_t>::type T;
  const int w = c.w, h = c.h < y_end ? c.h : y_end;      // rows [0, y_end): the specialised loops below take over from there
  const bool wide = w > kModMaxW;
  const DevBigHdr *BH = (const DevBigHdr *)big;
  const uint32_t root_word = kBig ? BH->root_word : 0u;
  const int ni = kBig ? (int)((root_word >> 25) & 63) : WT.ni, nl = kBig ? ni + 1 : WT.nl;
  const uint64_t *big_nodes = kBig ? (const uint64_t *)(big + BH->off_nodes) : nullptr;      // (prop, split) pairs
  const uint4 *big_need = kBig ? (const uint4 *)(big + BH->off_need) : nullptr;
  const uint32_t *big_exit = kBig ? big + BH->off_exit : nullptr;
  const int32_t *big_mulo = kBig ? (const int32_t *)big + BH->off_mulo : nullptr;
  const bool big_cfg_lds = S.st.num_clusters <= kLocMaxClusters;      // (then ev.cfg is S.cfg)
  const bool big_refs = kBig && BH->max_prop > 15;
  const int32_t *my_ref = (big_refs && lane < kModMaxRefs) ? S.refp[lane] : nullptr;
#ifdef JXL_MOD_DEBUG_SWITCHES      // tools/build_variant.sh <name> kernels_mod.hip -DJXL_MOD_DEBUG_SWITCHES: the removal switches of DESIGN 7d (six tests per sample the product does not pay for)
  const uint32_t dbg = (kBig && S.walk_stat) ? S.walk_stat[2] : 0u;      // MEASUREMENT ONLY (JXLAMD_DEBUG_MOD, word 4 of the flag block): 1 = every symbol from cluster 0's tables, 2 = block 0's exits taken as leaves — wrong pixels, same loop
#else
  constexpr uint32_t dbg = 0u;
#endif
  int my_prop, my_split; uint64_t my_need1, my_need0;
  uint32_t my_exit = 0;
  if (kBig) {
    const uint64_t nd = lane < ni ? big_nodes[lane] : 0x7fffffff00000000ull;
    my_prop = (int)(uint32_t)nd; my_split = (int)(nd >> 32);
    const uint4 m = lane < nl ? big_need[lane] : make_uint4(~0u, ~0u, ~0u, ~0u);
    my_need1 = m.x | ((uint64_t)m.y << 32); my_need0 = m.z | ((uint64_t)m.w << 32);
    my_exit = lane < nl ? big_exit[lane] : 0u;
  } else {
    my_prop = lane < ni ? WT.int_prop[lane] : 0;
    my_split = lane < ni ? WT.int_split[lane] : 0x7fffffff;
    my_need1 = lane < nl ? WT.leaf_need1[lane] : ~0ull;
    my_need0 = lane < nl ? WT.leaf_need0[lane] : ~0ull;
  }
  // lane j keeps leaf j's record in registers; the selected leaf's record is fetched with v_readlane (no LDS trip)
  const int my_lctx = (!kBig && lane < nl) ? WT.leaf_ctx[lane] : 0, my_lpred = (!kBig && lane < nl) ? WT.leaf_pred[lane] : 0;
  const int my_loff = (!kBig && lane < nl) ? WT.leaf_off[lane] : 0, my_lmul = (!kBig && lane < nl) ? WT.leaf_mul[lane] : 1;
  const int my_lclu = (!kBig && kLds && lane < nl) ? (int)((const uint8_t *)S.pool)[S.ctx_off + my_lctx] : 0;
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
#ifndef JXL_NO_LANE_PROPS      // (-DJXL_NO_LANE_PROPS: the lane-0 form) lane k computes property k of the sample from ITS coefficient row and stores it — one LDS store of 16 lanes instead of four 16-byte stores of one
  int dW = 0, dN = 0, dNW = 0, dNE = 0, dNN = 0, dWW = 0, dP9 = 0, dX = 0, dY = 0, dE = 0;
  bool dAbs = false;
  switch (lane) {
    case 2: dY = 1; break;
    case 3: dX = 1; break;
    case 4: dN = 1; dAbs = true; break;
    case 5: dW = 1; dAbs = true; break;
    case 6: dN = 1; break;
    case 7: dW = 1; break;
    case 8: dW = 1; dP9 = -1; break;
    case 9: dW = 1; dN = 1; dNW = -1; break;
    case 10: dW = 1; dNW = -1; break;
    case 11: dNW = 1; dN = -1; break;
    case 12: dN = 1; dNE = -1; break;
    case 13: dN = 1; dNN = -1; break;
    case 14: dW = 1; dWW = -1; break;
    case 15: dE = 1; break;
    default: break;
  }
#endif
  const int my_div = (int)S.divlut[lane];
  #define WAVE_DIV(idx) ((uint32_t)__builtin_amdgcn_readlane(my_div, __builtin_amdgcn_readfirstlane((int)(idx))))
  // kPack (32-bit arithmetic): the weighted predictor with lane l on sub-predictor l & 3, as in wave_decode_channel_wpfixed — one error-weight chain, one
  // prediction and one error update per sample instead of four of each on wave-uniform values; the sums over the four are DPP quad adds (every quad holds the same
  // four, so the sums are wave-uniform again).  The prediction of sub-predictor k as base_k - ((sum_j a_kj * in_j) >> 5): the same ring arithmetic term by term.
  constexpr bool kPack = kWP && kM16;
  constexpr int kPeStride = 2 * (kWpMaxW + 2);
  static_assert(offsetof(DevModScratch, wp_err) == offsetof(DevModScratch, wp_pred_err) + 4 * kPeStride * sizeof(uint32_t), "wp_err is row 4 behind the four error rows");
  const int kk = lane & 3;
  const int32_t a_tW = kk == 1 ? wp.p1 : kk == 2 ? wp.p2 : 0, a_tN = kk == 1 ? wp.p1 : kk == 2 ? wp.p2 : kk == 3 ? wp.p3b : 0;
  const int32_t a_tNW = kk == 2 ? wp.p2 : kk == 3 ? wp.p3a : 0, a_tNE = kk == 1 ? wp.p1 : kk == 3 ? wp.p3c : 0;
  const int32_t a_d1 = kk == 3 ? wp.p3d : 0, a_d2 = kk == 3 ? wp.p3e : 0;
  const bool baseW = (kk & 1) == 0, base0 = kk == 0;
  const uint32_t *my_wdiv = S.wdiv[kk];
  uint32_t *my_pe = S.wp_pred_err[kk];
  uint32_t *pe_flat = &S.wp_pred_err[0][0];
  if (kWP) {
    for (int i = lane; i < 2 * (w + 2); i += 64) { S.wp_err[i] = 0; for (int k = 0; k < 4; k++) S.wp_pred_err[k][i] = 0; }
    if (kPack) for (int i = lane; i < 256; i += 64) S.wdiv[i >> 6][i & 63] = (uint32_t)wp.w[i >> 6] * S.divlut[i & 63];
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
    uint32_t qNW = 0, qN = 0, qNE = 0;      // kPack: the error sums of sub-predictor kk
    if (kWP) {
      if (kPack) { qN = my_pe[prev_row]; qNW = qN; qNE = w > 1 ? my_pe[prev_row + 1] : qN; }
      else for (int k = 0; k < 4; k++) { peN[k] = S.wp_pred_err[k][prev_row]; peNW[k] = peN[k]; peNE[k] = w > 1 ? S.wp_pred_err[k][prev_row + 1] : peN[k]; }
      teN = S.wp_err[prev_row]; teNW = teN; teNE = w > 1 ? S.wp_err[prev_row + 1] : teN;
    }
    int32_t prev_prop9 = 0;
    // One sample.  kE = false: an INTERIOR sample (y >= 2, 2 <= x, x + 3 < w) — every edge substitution below is known not to apply, and with it go ~45 of the
    // ~510 instructions a sample costs (profiles/r05_pmc_block_form_loop_by_removal_rgba4k.json: the loop is bound by its instruction count)
    #define E(cond) (!kE || (cond))
    auto step = [&](auto edge_c, const int x) __attribute__((always_inline)) {
      constexpr bool kE = decltype(edge_c)::value;
      const T W_ = E(x > 0) ? vW : (y > 0 ? vN : 0);
      const T N_ = E(y > 0) ? vN : W_;
      const T NW_ = E(x > 0 && y > 0) ? vNW : W_;
      const T NE_ = E(x + 1 < w && y > 0) ? vNE : N_;
      const T NN_ = E(y > 1) ? rNN[x] : N_;
      const T NEE_ = E(x + 2 < w && y > 0) ? vNEE : NE_;
      const T WW_ = E(x > 1) ? vWW : W_;
      const int32_t nextNEE = E(y > 0 && x + 3 < w) ? rN[x + 3] : 0;     // independent of this sample: issued early
      const int32_t p9 = (int32_t)(W_ + N_ - NW_);
      const int32_t p9_prev = prev_prop9;
      int32_t pv[16];
      #define M24(a, b) __mul24((a), (int)(b))
      int32_t early = 0;
      if (kM16 && !kBig) {
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
      int32_t my_pred = 0;
      if (kWP && !(dbg & 8u)) {
        uint32_t wgt[4] = {0, 0, 0, 0};
        uint32_t my_wd = 0; int my_sh = 0;
        if (kPack) {
          const uint32_t e = qN + (E(x < w - 1) ? qNE : qN) + (E(x > 0) ? qNW : qN);
          my_sh = 26 - __builtin_clz(e + 1);
          my_sh = my_sh < 0 ? 0 : my_sh;
          my_wd = my_wdiv[e >> my_sh];                     // LDS, lane-indexed: (weight * reciprocal) of this sub-predictor's error sum
        } else for (int k = 0; k < 4; k++) {
          const uint32_t e = peN[k] + (E(x < w - 1) ? peNE[k] : peN[k]) + (E(x > 0) ? peNW[k] : peN[k]);
          int shift = floor_log2_u32(e + 1) - 5;
          if (shift < 0) shift = 0;
          wgt[k] = 4 + (((uint32_t)wp.w[k] * WAVE_DIV(e >> shift)) >> shift);
        }
        const T N8 = N_ * 8, W8 = W_ * 8, NE8 = NE_ * 8, NW8 = NW_ * 8, NN8 = NN_ * 8;
        const T tW = E(x > 0) ? teW : 0, tN = teN, tNW = E(x > 0) ? teNW : teN, tNE = E(x < w - 1) ? teNE : teN;
        const T sumWN = tN + tW;
        T p = tW;
        if (tabs<T>(tN) > tabs<T>(p)) p = tN;
        if (tabs<T>(tNW) > tabs<T>(p)) p = tNW;
        if (tabs<T>(tNE) > tabs<T>(p)) p = tNE;
        pv[15] = (int32_t)p;
        if (kPack) {
          const int32_t inner = a_tW * (int32_t)tW + a_tN * (int32_t)tN + a_tNW * (int32_t)tNW + a_tNE * (int32_t)tNE + a_d1 * (int32_t)(NN8 - N8) + a_d2 * (int32_t)(NW8 - W8);
          my_pred = (int32_t)(baseW ? W8 : N8) + (base0 ? (int32_t)(NE8 - N8) : 0) - (inner >> 5);
          uint32_t g = 4 + (my_wd >> my_sh);
          const uint32_t wsum = (uint32_t)quad_sum_i32((int)g);
          g >>= (27 - __builtin_clz(wsum));
          const uint32_t wsum2 = (uint32_t)quad_sum_i32((int)g);
          const int32_t sum = quad_sum_i32(my_pred * (int32_t)g) + (int32_t)(wsum2 >> 1) - 1;
          wp_raw = (T)__mulhi(sum, (int32_t)(WAVE_DIV(wsum2 - 1) << 8));      // (sum * dv) >> 24; wsum2 - 1 in [12, 30]: dv < 2^21
        } else {
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
        }
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
      if (kBig) {
        // the 14 property values (wave-uniform) go to LDS once per sample — lane 0, four 16-byte stores — and every lane of every block reads the one its
        // node tests: a single wave issues one instruction every four cycles whatever its kind, so what counts here is the instruction count (a lane-indexed
        // vector for ds_bpermute took 14 selects whose lane masks the compiler kept spilling: 56 instructions per sample)
#ifndef JXL_NO_LANE_PROPS
        if (kM16) {
          int32_t pvl = (M24(dN, N_) + M24(dNW, NW_)) + (M24(dNE, NE_) + M24(dNN, NN_)) + (M24(dWW, WW_) + M24(dP9, p9_prev)) + (M24(dX, x) + M24(dY, y)) + (M24(dW, W_) + M24(dE, pv[15]));
          if (dAbs) pvl = pvl < 0 ? -pvl : pvl;
          if (lane < 16) S.props[lane] = pvl;
        } else
#endif
        if (lane == 0) {
          #pragma unroll
          for (int k = 0; k < 16; k++) S.props[k] = k < 2 ? 0 : pv[k];
        }
        if (big_refs) {                      // (uniform) lane r: the four properties of previous channel r at this position — |v|, v, |v - g|, v - g with g the clamped gradient of its W, N, NW
          if (lane < kModMaxRefs) {
            int32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
            if (my_ref) {
              const int32_t *rp = my_ref + (size_t)y * (size_t)w;
              const int64_t v = rp[x], vl = x ? rp[x - 1] : 0, vt = y ? rp[x - w] : vl, vtl = (x && y) ? rp[x - w - 1] : vl;
              const int64_t lo = vl < vt ? vl : vt, hi = vl < vt ? vt : vl, grad = vl + vt - vtl;
              const int64_t vp = vtl > hi ? lo : vtl < lo ? hi : grad;
              q0 = (int32_t)(v < 0 ? -v : v); q1 = (int32_t)v; q2 = (int32_t)(v - vp < 0 ? vp - v : v - vp); q3 = (int32_t)(v - vp);
            }
            S.props[16 + 4 * lane] = q0; S.props[17 + 4 * lane] = q1; S.props[18 + 4 * lane] = q2; S.props[19 + 4 * lane] = q3;
          }
        }
        myv = S.props[my_prop & 63];
      } else if (kM16) {
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
      int l_ctx, l_pred, l_off, l_mul, l_clu;
      if (kBig) {
        uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)my_exit, leaf);
        int eidx = (int)(root_word & 8191u) + leaf;
        if ((dbg & 2u) && (e >> 31)) e = 5u << 26;
        if (dbg & 32u) e = 5u << 26;
        while (e >> 31) {                    // (uniform) the exit names another block
          const int bni = (int)((e >> 25) & 63), noff = (int)((e >> 13) & 4095), eoff = (int)(e & 8191);
          const int jl = lane < bni ? lane : bni;      // lanes beyond the block read its last exit / the entry behind its last node (inside the area; masked below): no branches around the loads
          const uint64_t nd = big_nodes[noff + jl];
          const uint4 m = big_need[eoff + jl];
          const uint32_t bex = big_exit[eoff + jl];
          const int bv = S.props[(uint32_t)nd & 63u];
          const uint64_t bdec = __ballot(lane < bni && bv > (int)(nd >> 32));
          const uint64_t b1 = m.x | ((uint64_t)m.y << 32), b0 = m.z | ((uint64_t)m.w << 32);
          const uint64_t blm = __ballot(lane <= bni && (bdec & b1) == b1 && (~bdec & b0) == b0);
          const int bleaf = blm ? __builtin_ctzll(blm) : 0;
          e = (uint32_t)__builtin_amdgcn_readlane((int)bex, bleaf);
          eidx = eoff + bleaf;
          if (!blm) e = 0;                   // (cannot happen for a built tree; keeps the loop finite)
        }
        l_ctx = (int)(e & 0x3ffffu); l_clu = (int)((e >> 18) & 255u); l_pred = (int)((e >> 26) & 15u);
        if (dbg & 1u) l_clu = 0;
        l_mul = 1; l_off = 0;
        if (e & (1u << 30)) { l_mul = big_mulo[2 * eidx]; l_off = big_mulo[2 * eidx + 1]; }
      } else {
        l_ctx = __builtin_amdgcn_readlane(my_lctx, leaf); l_pred = __builtin_amdgcn_readlane(my_lpred, leaf);
        l_off = __builtin_amdgcn_readlane(my_loff, leaf); l_mul = __builtin_amdgcn_readlane(my_lmul, leaf);
        l_clu = __builtin_amdgcn_readlane(my_lclu, leaf);
      }
      const T guess = predict_plain_t<T>(l_pred, W_, N_, NW_, NE_, NN_, WW_, NEE_, wp_pred);
      uint32_t u;
      if (kBig && (dbg & 16u)) u = (uint32_t)(x & 3);
      else if (kBig && !kLds) u = wave_ec_read_global(ev, S, b, state, (uint32_t)l_clu, big_cfg_lds);      // the exit word carries the cluster: no context-map trip
      else u = wave_ec_read<kLds>(ev, S, b, state, (uint32_t)l_ctx, (uint32_t)l_clu);
      const T res = (T)unpack_signed(u);
      const T val = (l_mul == 1 ? res : res * (T)l_mul) + (T)l_off + guess;      // l_mul is wave-uniform: the multiply is branched around
      if (lane == 0) { if (!wide) row[x] = (int32_t)val; else gst(&out[x], (int32_t)val); }      // (a wide channel's row IS the plane; the others go to HBM row by row, below)
      vWW = vW; vW = (int32_t)val;
      vNW = vN; vN = vNE; vNE = vNEE; vNEE = nextNEE;
      if (kWP && !(dbg & 8u)) {
        const T v8 = val * 8;
        const int32_t terr = (int32_t)(wp_raw - v8);
        if (kPack) {
          const uint32_t my_err = (uint32_t)((tabs<int32_t>(my_pred - (int32_t)v8) + 3) >> 3);
          if (lane < 5) pe_flat[lane * kPeStride + cur_row + x] = lane < 4 ? my_err : (uint32_t)terr;      // lanes 0..3: their sub-predictor's error, lane 4: the true error (S.wp_err)
          qNW = qN; qN = qNE + my_err;                   // carry to (x+1) of the previous row
          teNW = teN; teN = teNE; teW = terr;
          if (E(x + 2 < w)) { qNE = my_pe[prev_row + x + 2]; teNE = S.wp_err[prev_row + x + 2]; }
          else { qNE = qN; teNE = teN; }
        } else {
        uint32_t err[4];
        for (int k = 0; k < 4; k++) err[k] = (uint32_t)((tabs<T>(wpred[k] - v8) + 3) >> 3);
        if (lane == 0) { S.wp_err[cur_row + x] = terr; for (int k = 0; k < 4; k++) S.wp_pred_err[k][cur_row + x] = err[k]; }
        for (int k = 0; k < 4; k++) { peNW[k] = peN[k]; peN[k] = peNE[k] + err[k]; }   // carry to (x+1) of the previous row
        teNW = teN; teN = teNE; teW = terr;
        if (E(x + 2 < w)) { for (int k = 0; k < 4; k++) peNE[k] = S.wp_pred_err[k][prev_row + x + 2]; teNE = S.wp_err[prev_row + x + 2]; }
        else { for (int k = 0; k < 4; k++) peNE[k] = peN[k]; teNE = teN; }
        }
      }
    };
    #undef E
    for (int x = 0; x < w;) {
      if (y > 1 && x >= 2 && x + 3 < w) { for (const int xe = w - 3; x < xe; x++) step(std::false_type(), x); }
      else { step(std::true_type(), x); x++; }
    }
    __syncthreads();     // row[] written by lane 0 is read by every lane in the next row
    if (!wide) for (int i = lane; i < w; i += 64) gst(&out[i], row[i]);      // one coalesced store per row instead of a one-lane store per sample
  }
  #undef WAVE_DIV
}

// ------------------------------------------------------------------------------------------------------------------
// Specialised loop for libjxl's LF-coefficient streams (effort >= 7 "kWPFixedDC": every decision node of the channel's tree tests
// property 15 — the weighted predictor's max error — and every leaf uses the weighted predictor, multiplier 1, offset 0).
// That is 73 % of the Modular samples of a 4K VarDCT frame and the longest dependency chain of the whole decode.
//   * context: the decision nodes are thresholds on ONE value, so the leaf is a function of how many thresholds lie below it:
//     one v_cmp + s_bcnt1 + v_readlane (lane c keeps the cluster / hybrid-uint config of "c thresholds below") instead of two
//     ballots and five readlanes;
//   * weighted predictor: lane l evaluates sub-predictor k = l & 3 (error weight, prediction, product), the sums over k are two
//     DPP quad-permute adds each — the four-fold scalar repetition of the generic loop is gone;
//   * everything that only depends on the PREVIOUS row (neighbour values and errors, error-sum bases, the parts of the four
//     predictions that do not involve W, clamp bounds, the max-error candidate) is computed for 32 samples at a time by 32 lanes
//     in parallel and handed to the serial loop through LDS records (3 ds_reads per sample);
//   * the reciprocal table is pre-multiplied by the header weights (S.wdiv), the row is copied to HBM once per row.
// Same integer arithmetic as wave_decode_channel<true, true, true> (bit-exact).  Row 0 runs through the same step with constant records:
// there every neighbour is the late value W and every previous-row error is 0, so sub-predictor k predicts W8 - ((tW * {0, p1, p2, 0}[k]) >> 5),
// the clamp interval is [W8, W8] and the max-error property is tW — no generic loop (and none of its ~180 registers) in this path.
struct alignas(16) DevWpFixedLds {       // overlays DevModScratch::wt, props, refp (the tree sits in registers by then; the other two serve other loops)
  int32_t U[32][4];                      // per sample of the chunk: q, |q|, tN, tN ^ tNW
  int32_t V[32][2];                      // ... max(N8, NE8), min(N8, NE8)
  int32_t K[32][4][4];                   // per sample and sub-predictor k: error-sum base, A_k, B_k
};
static_assert(sizeof(DevWpFixedLds) <= offsetof(DevModScratch, fallback_err) - offsetof(DevModScratch, wt) && offsetof(DevModScratch, wt) % 16 == 0, "the chunk records live in the LDS of the tree, the property vector and the reference pointers");

// lane l takes `keep` from lane l - 1 (wave_shr:1) / l - 4 within its row of 16 (row_shr:4); the lanes without a source take `fresh`
__device__ __forceinline__ int shift_in_wave1(int fresh, int keep) { return __builtin_amdgcn_update_dpp(fresh, keep, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ int shift_in_row4(int fresh, int keep) { return __builtin_amdgcn_update_dpp(fresh, keep, 0x114, 0xF, 0xF, false); }
__device__ __forceinline__ int med3_i32(int v, int lo, int hi) { int r; asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi)); return r; }

// Bit reader of the lock-step loops: every lane holds the same reader, so its conditions are wave-uniform — branching on a ballot
// gives scalar branches (no exec save / restore around the rarely taken refill).
__device__ __forceinline__ void ubits_refill(DevBits &b) {
  if (__ballot(b.n <= 32)) {
    b.buf |= (uint64_t)b.ahead << b.n;
    b.n += 32;
    b.ahead = __ballot(b.next < b.end) ? *b.next : 0u;
    b.next++;
  }
}
__device__ __forceinline__ uint32_t ubits_read(DevBits &b, int n) {       // n <= 32, uniform; n == 0 reads nothing
  ubits_refill(b);
  const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1));
  b.buf >>= n; b.n -= n; b.consumed += (uint64_t)n;
  return v;
}

// Alias tables of a whole stream in 1.25 KB per cluster instead of 2 KB: entry i -> 32 bits (cutoff | right symbol << 8 | offsets1 << 16),
// the two frequencies of an entry are D[i] and D[right], so one table of D per cluster (u16 x 128 symbols) serves both — at the price
// of a dependent LDS read on the rANS chain, which has slack next to the weighted predictor's.  libjxl's LF streams carry up to one
// cluster per context (34); 23 fit the pool.  Returns false (pool untouched) when a symbol >= 128 can occur.
// Only the clusters the channel's leaves use are packed (`used`: bit per cluster id; compact index = number of used clusters below it):
// libjxl's non-streaming encoder writes ONE global tree for all Modular streams of a frame (39 clusters and log_alpha 7 in the
// reference's 4K demo photographs; the LF channels use 28 of them).  log_alpha 5..8: 4 << log_alpha bytes of entries per cluster.
// Returns false (pool untouched) when the tables do not fit or a symbol >= 128 can occur.
// Frequency tables sized by the alphabet (round 5): the largest symbol the used clusters can produce + 1, rounded up to a power of two >= 16 — libjxl's LF
// residual tokens stay below 64, so a cluster costs 512 + 128 bytes at log_alpha 7 instead of 768: 28 clusters 17.5 KB instead of 21 KB, which with the channel
// descriptors out of the fixed part lets a fourth LF stream into a CU.  0: a symbol >= 128 can occur (not packed).
// the clusters a channel's leaves use (a context map holds up to 256): libjxl's one-shot encoder writes ONE code of ~128 clusters for every Modular stream of a
// frame, of which an LF channel touches ~30
struct CluSet {
  uint64_t w[4];
  __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = w[3] = 0; }
  __device__ __forceinline__ void add(int clu) { const uint64_t bit = 1ull << (clu & 63); const int k = (clu >> 6) & 3; w[0] |= k == 0 ? bit : 0; w[1] |= k == 1 ? bit : 0; w[2] |= k == 2 ? bit : 0; w[3] |= k == 3 ? bit : 0; }
  __device__ __forceinline__ int count() const { return __builtin_popcountll(w[0]) + __builtin_popcountll(w[1]) + __builtin_popcountll(w[2]) + __builtin_popcountll(w[3]); }
  __device__ __forceinline__ int rank(int clu) const {      // members below clu = its index in the packed pool
    const int k = (clu >> 6) & 3;
    int r = __builtin_popcountll((k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : w[3]) & ((1ull << (clu & 63)) - 1ull));
    if (k > 0) r += __builtin_popcountll(w[0]);
    if (k > 1) r += __builtin_popcountll(w[1]);
    if (k > 2) r += __builtin_popcountll(w[2]);
    return r;
  }
};
__device__ __forceinline__ int wave_alias_dsz(const DevAlias *galias, const CluSet &used, int log_alpha, int lane) {
  const int table = 1 << log_alpha;
  int mx = 0;
  for (int k = 0; k < 4; k++)
    for (uint64_t m = used.w[k]; m; m &= m - 1) {
      const int s = 64 * k + __builtin_ctzll(m);
      for (int i = lane; i < table; i += 64) {
        const DevAlias e = galias[(s << log_alpha) + i];
        if (e.freq0 != 0 && i > mx) mx = i;
        if (e.freq1 != 0 && (int)e.right > mx) mx = (int)e.right;
      }
    }
  for (int d = 32; d; d >>= 1) { const int o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
  int dsz = 16;
  while (dsz <= mx) dsz <<= 1;
  return dsz > 128 ? 0 : dsz;
}
// Round 6: THREE bytes per alias entry instead of four — a cluster is [u16 x table: offsets1 | (cutoff & 15) << 12][u8 x table: right symbol | (cutoff >> 4) << log2(dsz)].
// offsets1 < 4096 (a slot of the 4096), the right symbol < dsz, and the cutoff < bucket size = 4096 >> log_alpha takes 12 - log_alpha bits, of which the byte has
// 8 - log2(dsz) to spare: log_alpha 8 / 7 / 6 / 5 with alphabets of at most 128 / 128 / 64 / 32 symbols (an alphabet never exceeds its table) always fit.  At log_alpha 8
// (every LF stream of libjxl's streaming encoder, i.e. the bench's frames) a cluster is 768 + 128 bytes instead of 1 024 + 128: 21 clusters 18.4 KB instead of 23.6 KB —
// LDS that every LF workgroup of the process holds for ~100 ms.
__device__ __forceinline__ int wave_dsz_log2(int dsz) { return 31 - __builtin_clz((unsigned)dsz); }
__device__ __forceinline__ int wave_packed_bytes(const CluSet &used, int log_alpha, int dsz) { return used.count() * ((3 << log_alpha) + 2 * dsz); }
__device__ __forceinline__ bool wave_pack_alias(const DevAlias *galias, const CluSet &used, int log_alpha, int dsz, DevModScratch &S, int lane) {
  if (log_alpha < 5 || log_alpha > 8 || dsz <= 0 || wave_packed_bytes(used, log_alpha, dsz) > S.pool_bytes) return false;
  const int table = 1 << log_alpha, nc = used.count(), rbits = wave_dsz_log2(dsz);
  if (12 - log_alpha > 4 + 8 - rbits) return false;       // (cannot happen: see above)
  __syncthreads();
  uint8_t *base = (uint8_t *)S.pool;
  uint16_t *D = (uint16_t *)(base + (size_t)nc * (size_t)(3 << log_alpha));
  int cid = 0;
  for (int k = 0; k < 4; k++)
    for (uint64_t m = used.w[k]; m; m &= m - 1, cid++) {
      const int s = 64 * k + __builtin_ctzll(m);
      uint16_t *e16 = (uint16_t *)(base + (size_t)cid * (size_t)(3 << log_alpha));
      uint8_t *e8 = (uint8_t *)(e16 + table);
      for (int i = lane; i < table; i += 64) {
        const DevAlias e = galias[(s << log_alpha) + i];
        e16[i] = (uint16_t)((uint32_t)e.off1 | (((uint32_t)e.cutoff & 15u) << 12));
        e8[i] = (uint8_t)((uint32_t)e.right | (((uint32_t)e.cutoff >> 4) << rbits));
        if (i < dsz) D[cid * dsz + i] = e.freq0;
      }
      for (int i = table + lane; i < dsz; i += 64) D[cid * dsz + i] = 0;       // symbols beyond the table never occur
    }
  __syncthreads();
  return true;
}

// Bit reader over an LDS ring (wave_decode_channel_wpfixed): the next 512 bytes of the stream sit in S.ring, refilled 256 bytes at a
// time by all lanes ~450 samples before the reader gets there, so the serial loop never waits on a global load — next to the
// memory-bound kernels of other decoder contexts such a load takes microseconds, and one was due every ~7 samples.
struct RingBits {
  uint64_t buf; int32_t n; uint32_t ahead; int32_t widx; uint64_t consumed;
  const uint32_t *base, *end;         // word 0 of the ring's index space; first word past the padded stream
};
__device__ __forceinline__ void ring_fill_half(uint32_t *ring, const RingBits &r, int first_word, int lane) {
  const uint32_t *p = r.base + first_word + lane;
  ring[(first_word + lane) & 127] = p < r.end ? *p : 0u;
}
__device__ __forceinline__ void ring_open(uint32_t *ring, RingBits &r, const DevBits &b, int lane) {
  r.buf = b.buf; r.n = b.n; r.ahead = b.ahead; r.widx = 0; r.consumed = b.consumed; r.base = b.next; r.end = b.end;
  ring_fill_half(ring, r, 0, lane); ring_fill_half(ring, r, 64, lane);
  __syncthreads();
}
__device__ __forceinline__ void ring_close(const RingBits &r, DevBits &b) {
  b.buf = r.buf; b.n = r.n; b.ahead = r.ahead; b.next = r.base + r.widx; b.consumed = r.consumed;
}
__device__ __forceinline__ uint32_t ring_read(uint32_t *ring, RingBits &r, int n, int lane) {     // n <= 32, uniform
  if (__ballot(r.n <= 32)) {
    r.buf |= (uint64_t)r.ahead << r.n;
    r.n += 32;
    r.ahead = ring[r.widx & 127];
    r.widx++;
    if ((__builtin_amdgcn_readfirstlane(r.widx) & 63) == 0) ring_fill_half(ring, r, r.widx + 64, lane);      // the half just left behind
  }
  const uint32_t v = (uint32_t)(r.buf & ((1ull << n) - 1));
  r.buf >>= n; r.n -= n; r.consumed += (uint64_t)n;
  return v;
}

__device__ __forceinline__ void wave_decode_channel_wpfixed(const DevECView &ev, DevBits &b, uint32_t &state, const DevWP &wp, DevModScratch &S,
                                                            const DevChanOut c, int lane, int my_split, int my_off, int my_doff, int my_cfg, int y_begin, int rbits) {
  const int w = c.w, h = c.h;
  DevWpFixedLds &R = *(DevWpFixedLds *)&S.wt;
  const int k = lane & 3;
  const int PT = k == 1 ? wp.p1 : k == 2 ? wp.p2 : 0;
  const int PW = k == 3 ? -wp.p3e : 0;
  const int cWmask = (k & 1) ? 0 : -1;         // sub-predictors 0 and 2 start from W
  const bool use_pw = wp.p3e != 0;             // uniform
  const uint32_t *wdiv = S.wdiv[k];
  const int la = ev.log_alpha, lb = 12 - la;
  const uint8_t *pool8 = (const uint8_t *)S.pool;          // packed alias entries + per-symbol frequencies (wave_pack_alias)
  const uint32_t e8_base = 2u << la, rmask = (1u << rbits) - 1u;      // a cluster's byte array follows its 16-bit array
  for (int i = lane; i < 256; i += 64) S.wdiv[i >> 6][i & 63] = (uint32_t)wp.w[i >> 6] * S.divlut[i & 63];
  RingBits rb;
  ring_open(S.ring, rb, b, lane);
  for (int y = y_begin; y < h; y++) {
    int32_t *row = S.rows[y % 3];
    const int32_t *rN = S.rows[(y + 2) % 3];
    const int32_t *rNN = S.rows[(y + 1) % 3];
    const int cur_row = (y & 1) ? 0 : (w + 2), prev_row = (y & 1) ? (w + 2) : 0;
    const bool row0 = y == 0;
    const int cwm = row0 ? -1 : cWmask;       // row 0: every sub-predictor starts from W (N, NE, NW, NN are W there)
    const bool upw = use_pw && !row0;         // ... and the (NW - W) term vanishes
    int32_t W8 = row0 ? 0 : rN[0] * 8, tW = 0;           // x = 0: W is replaced by N (0 in row 0), its error by 0
    uint32_t e1 = 0, e2 = 0;                  // this lane's sub-predictor error of the previous / second previous sample
    for (int x0 = 0; x0 < w; x0 += 32) {
      const int n = w - x0 < 32 ? w - x0 : 32;
      if (row0) {
        if (lane < n) {
          int4 u0; int2 u1; u0.x = 0; u0.y = 0; u0.z = 0; u0.w = 0; u1.x = (int)0x80000000; u1.y = 0x7fffffff;
          *(int4 *)&R.U[lane][0] = u0; *(int2 *)&R.V[lane][0] = u1;
          int4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
#pragma unroll
          for (int kk = 0; kk < 4; kk++) *(int4 *)&R.K[lane][kk][0] = z;
        }
      } else if (lane < n) {                  // ---- parallel part: everything sample x0 + lane takes from the previous rows
        const int x = x0 + lane, pos = prev_row + x;
        const bool has_l = x > 0, has_r = x + 1 < w;
        const int32_t N = rN[x], NE = has_r ? rN[x + 1] : N, NW = has_l ? rN[x - 1] : N, NN = y > 1 ? rNN[x] : N;
        const int32_t tN = S.wp_err[pos], tNE = has_r ? S.wp_err[pos + 1] : tN, tNW = has_l ? S.wp_err[pos - 1] : tN;
        const int32_t N8 = N * 8, NE8 = NE * 8, NW8 = NW * 8, NN8 = NN * 8;
        int32_t q = tN;
        if (tabs<int32_t>(tNW) > tabs<int32_t>(q)) q = tNW;
        if (tabs<int32_t>(tNE) > tabs<int32_t>(q)) q = tNE;
        int4 u0; int2 u1;
        u0.x = q; u0.y = tabs<int32_t>(q); u0.z = tN; u0.w = tN ^ tNW; u1.x = N8 > NE8 ? N8 : NE8; u1.y = N8 < NE8 ? N8 : NE8;
        *(int4 *)&R.U[lane][0] = u0; *(int2 *)&R.V[lane][0] = u1;
        #pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const uint32_t EN = S.wp_pred_err[kk][pos], ENE = has_r ? S.wp_pred_err[kk][pos + 1] : EN, ENW = has_l ? S.wp_pred_err[kk][pos - 1] : EN;
          int4 r;
          r.x = (int32_t)(EN + ENE + ENW);
          r.y = kk == 0 ? NE8 - N8 : kk == 2 ? 0 : N8;
          r.z = kk == 0 ? 0 : kk == 1 ? (tN + tNE) * wp.p1 : kk == 2 ? (tN + tNW) * wp.p2
                            : tNW * wp.p3a + tN * wp.p3b + tNE * wp.p3c + (NN8 - N8) * wp.p3d + NW8 * wp.p3e;
          r.w = 0;
          *(int4 *)&R.K[lane][kk][0] = r;
        }
      }
      __syncthreads();
      // ---- serial part.  The results leave through lane shift registers (one DPP move each per sample, no masked stores): after the
      // chunk lane l holds value and true error of sample n - 1 - l; lanes 0..15 hold the sub-predictor errors of the last four samples.
      int32_t keep_val = 0, keep_te = 0, keep_e = 0;
      int4 u0 = *(const int4 *)&R.U[0][0]; int2 u1 = *(const int2 *)&R.V[0][0];
      int4 kr = *(const int4 *)&R.K[0][k][0];
      auto step = [&](const int i, auto last_tag) {
        constexpr bool kLastOfRow = decltype(last_tag)::value;
        // context: property 15 = the error of largest magnitude among W, N, NW, NE (first one wins ties); the leaf is a function of
        // how many thresholds lie below it
        const int32_t aW = tabs<int32_t>(tW);
        const int32_t p = u0.y > aW ? u0.x : tW;
        const int cnt = __builtin_popcountll(__ballot(p > my_split));
        const uint32_t aoff = (uint32_t)__builtin_amdgcn_readlane(my_off, cnt);       // byte offsets of the cluster's packed alias entries
        const uint32_t doff = (uint32_t)__builtin_amdgcn_readlane(my_doff, cnt);      // ... and of its frequency table
        const uint32_t cfg = (uint32_t)__builtin_amdgcn_readlane(my_cfg, cnt);
        const uint32_t ai = __builtin_amdgcn_ubfe(state, lb, la);
        const uint32_t e16 = *(const uint16_t *)(pool8 + aoff + (ai << 1));           // LDS: offsets1 | (cutoff & 15) << 12
        const uint32_t e8 = *(const uint8_t *)(pool8 + aoff + e8_base + ai);          // LDS: right symbol | (cutoff >> 4) << rbits
        // weighted predictor, sub-predictor k per lane: error weight
        uint32_t e = (uint32_t)kr.x + e1 + e2;
        if (kLastOfRow) e += e1;              // no NE: the N error sum (which carries e1) counts twice
        int sh = 26 - __builtin_clz(e + 1);
        sh = sh < 0 ? 0 : sh;
        const uint32_t wd = wdiv[e >> sh];                                            // LDS
        // records of the next sample: issued behind the two reads this sample waits for (LDS returns in order), used next step
        const int inext = i + 1 < 32 ? i + 1 : 31;
        const int4 u0n = *(const int4 *)&R.U[inext][0]; const int2 u1n = *(const int2 *)&R.V[inext][0];
        const int4 krn = *(const int4 *)&R.K[inext][k][0];
        __builtin_amdgcn_sched_barrier(0);
        // ... prediction of sub-predictor k and the clamp bounds while the tables come in
        int32_t inner = kr.z + __mul24(tW, PT);
        if (upw) inner += __mul24(W8, PW);
        const int32_t wpk = kr.y + (W8 & cwm) - (inner >> 5);
        const int32_t mx = u1.x > W8 ? u1.x : W8, mn = u1.y < W8 ? u1.y : W8;
        const bool no_clamp = ((u0.z ^ tW) | u0.w) > 0;
        uint32_t wgt = 4 + (wd >> sh);
        const uint32_t wsum = (uint32_t)quad_sum_i32((int)wgt);
        wgt >>= (27 - __builtin_clz(wsum));
        const uint32_t wsum2 = (uint32_t)quad_sum_i32((int)wgt);
        const uint32_t dv = S.divlut[wsum2 - 1];                                      // LDS; wsum2 in [13, 31]: dv < 2^21
        const int32_t sum = quad_sum_i32(__mul24(wpk, (int32_t)wgt)) + (int32_t)(wsum2 >> 1) - 1;
        // rANS symbol + hybrid uint
        const uint32_t apos = state & ((1u << lb) - 1);
        const bool right = apos >= ((e16 >> 12) | ((e8 >> rbits) << 4));
        uint32_t u = right ? e8 & rmask : ai;
        const uint32_t off = right ? (e16 & 4095u) + apos : apos;
        const uint32_t freq = *(const uint16_t *)(pool8 + doff + (u << 1));           // LDS, dependent: the symbol's frequency
        state = __umul24(freq, state >> 12) + off;                                    // freq <= 4096, state >> 12 < 2^20
        if (__ballot(state < (1u << 16))) state = (state << 16) | ring_read(S.ring, rb, 16, lane);
        const uint32_t split_exp = cfg & 0xff;
        if (__ballot(u >= (1u << split_exp))) {                                       // hybrid uint: tokens below the split are the value
          const uint32_t msb = (cfg >> 8) & 0xff, lsb = (cfg >> 16) & 0xff;
          uint32_t nbits = split_exp - (msb + lsb) + ((u - (1u << split_exp)) >> (msb + lsb));
          if (nbits > 31) nbits = 31;         // corrupt stream; the final-state check flags it
          const uint32_t low = u & ((1u << lsb) - 1);
          const uint32_t tok = u >> lsb;
          const uint32_t bits = ring_read(S.ring, rb, (int)nbits, lane);
          u = (((((1u << msb) | (tok & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
        }
        // (sum * dv) >> 24 as the high half of sum * (dv << 8): one multiply instead of a 64-bit product
        int32_t raw = __mulhi(sum, (int32_t)(dv << 8));
        const int32_t cl = med3_i32(raw, mn, mx);
        raw = no_clamp ? raw : cl;
        const int32_t val = unpack_signed(u) + ((raw + 3) >> 3);
        // this sample's errors feed the next one
        const int32_t v8 = val * 8;
        const int32_t d = wpk - v8;
        const int32_t err = (tabs<int32_t>(d) + 3) >> 3;
        tW = raw - v8; W8 = v8; e2 = e1; e1 = (uint32_t)err;
        keep_val = shift_in_wave1(val, keep_val); keep_te = shift_in_wave1(tW, keep_te);
        keep_e = shift_in_row4(err, keep_e);
        u0 = u0n; u1 = u1n; kr = krn;
      };
      const bool row_ends = x0 + n == w;          // this chunk carries the row's last sample (no NE neighbour): its step is a separate instantiation
      const int nn = row_ends ? n - 1 : n;
      int i = 0;
      #pragma unroll 1
      for (; i + 4 <= nn; i += 4) {
        step(i, std::false_type()); step(i + 1, std::false_type()); step(i + 2, std::false_type()); step(i + 3, std::false_type());
        // lanes 4j + k, j = 0..3, hold the error of sub-predictor k for sample i + 3 - j
        if (lane < 16) S.wp_pred_err[k][cur_row + x0 + i + 3 - (lane >> 2)] = (uint32_t)keep_e;
      }
      const int rem = n - i;                      // 0..4 samples left, the row's last one among them when row_ends
      #pragma unroll 1
      for (; i < nn; i++) step(i, std::false_type());
      if (row_ends) step(i, std::true_type());
      if (lane < 4 * rem) S.wp_pred_err[k][cur_row + x0 + n - 1 - (lane >> 2)] = (uint32_t)keep_e;
      if (lane < n) { row[x0 + n - 1 - lane] = keep_val; S.wp_err[cur_row + x0 + n - 1 - lane] = keep_te; }
      __syncthreads();
    }
    int32_t *out = c.d + (size_t)y * (size_t)w;
    for (int x = lane; x < w; x += 64) out[x] = row[x];      // one coalesced copy per row
  }
  ring_close(rb, b);
}

// A channel whose reachable leaves all carry the same cluster, the zero predictor, multiplier 1 and offset 0: no tree walk, no
// neighbourhood — the rANS chain alone (libjxl's EPF-sharpness channel, 86 % of the samples of an HF-metadata stream: its contexts differ
// only by N > 3 / W > 3 and share one cluster when the encoder left the map flat).  When that cluster's histogram holds a single
// symbol (frequency 4096) below the hybrid-uint split, the state never changes and no bit is read: the channel is a constant.
template <bool kLds>
__device__ __forceinline__ void wave_decode_channel_uniform(const DevECView &ev, DevBits &b, uint32_t &state, DevModScratch &S, const DevChanOut c, int lane, uint32_t cluster) {
  const int la = ev.log_alpha, lb = 12 - la;
  const DevAlias *tab = (kLds ? (const DevAlias *)S.pool : ev.alias) + ((size_t)cluster << la);
  const uint32_t cfg = kLds ? S.cfg[cluster] : ev.cfg[cluster];
  const uint32_t split_exp = cfg & 0xff, msb = (cfg >> 8) & 0xff, lsb = (cfg >> 16) & 0xff;
  const size_t n = (size_t)c.w * (size_t)c.h;
  {
    const DevAlias e0 = tab[0];
    const bool r0 = 0 >= e0.cutoff;
    const uint32_t sym0 = r0 ? e0.right : 0u, f0 = r0 ? e0.freq1 : e0.freq0;
    if (f0 == 4096u && sym0 < (1u << split_exp)) {
      const int32_t v = unpack_signed(sym0);
      for (size_t i = (size_t)lane; i < n; i += 64) c.d[i] = v;
      return;
    }
  }
  int32_t keep = 0;
  for (size_t i0 = 0; i0 < n; i0 += 64) {
    const int m = n - i0 < 64 ? (int)(n - i0) : 64;
    #pragma unroll 1
    for (int i = 0; i < m; i++) {
      const uint32_t res = state & 0xfff, bi = res >> lb, pos = res & ((1u << lb) - 1);
      const DevAlias e = tab[bi];
      const bool right = pos >= e.cutoff;
      uint32_t u = right ? e.right : bi;
      state = (right ? e.freq1 : e.freq0) * (state >> 12) + (right ? (uint32_t)e.off1 + pos : pos);
      if (__ballot(state < (1u << 16))) state = (state << 16) | ubits_read(b, 16);
      if (__ballot(u >= (1u << split_exp))) {
        uint32_t nbits = split_exp - (msb + lsb) + ((u - (1u << split_exp)) >> (msb + lsb));
        if (nbits > 31) nbits = 31;           // corrupt stream; the final-state check flags it
        const uint32_t low = u & ((1u << lsb) - 1), tok = u >> lsb;
        const uint32_t bits = ubits_read(b, (int)nbits);
        u = (((((1u << msb) | (tok & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
      }
      keep = shift_in_wave1(unpack_signed(u), keep);
    }
    if (lane < m) c.d[i0 + (size_t)(m - 1 - lane)] = keep;     // after m steps lane l holds sample i0 + m - 1 - l: one coalesced store
  }
}

// "Lean" lock-step loop for the channels of libjxl's HF-metadata streams whose tree only looks at y, x, N, W (properties 2..7) and whose
// leaves predict with zero, W or N: the general loop's seven-sample neighbourhood window, its sixteen-property sum and the predictor
// switch shrink to two multiply-adds, the two tree ballots and a select.  Results leave through a lane shift register (one coalesced
// store per 64 samples); a row buffer in LDS is only kept when N is needed, which also makes channels wider than the LDS rows (the
// block-info channel: thousands of samples x 2 rows, W and y only) eligible.
template <bool kLds>
__device__ __forceinline__ void wave_decode_channel_lean(const DevECView &ev, DevBits &b, uint32_t &state, DevModScratch &S, DevWaveTree &WT,
                                                         const DevChanOut c, int lane, bool needs_n, int y_end) {
  const int w = c.w, h = y_end;            // rows [0, y_end): the rest of an eligible channel goes through wave_decode_channel_chain
  const int ni = WT.ni, nl = WT.nl;
  const int my_prop = lane < ni ? WT.int_prop[lane] : 0;
  const int my_split = lane < ni ? WT.int_split[lane] : 0x7fffffff;
  const uint64_t my_need1 = lane < nl ? WT.leaf_need1[lane] : ~0ull, my_need0 = lane < nl ? WT.leaf_need0[lane] : ~0ull;
  const int my_lctx = lane < nl ? WT.leaf_ctx[lane] : 0, my_lpred = lane < nl ? WT.leaf_pred[lane] : 0;
  const int my_loff = lane < nl ? WT.leaf_off[lane] : 0, my_lmul = lane < nl ? WT.leaf_mul[lane] : 1;
  const int my_lclu = (kLds && lane < nl) ? (int)((const uint8_t *)S.pool)[S.ctx_off + my_lctx] : 0;
  const int cY = my_prop == 2, cX = my_prop == 3, cN = (my_prop == 4 || my_prop == 6), cW = (my_prop == 5 || my_prop == 7);
  const bool cAbs = my_prop == 4 || my_prop == 5;
  __syncthreads();
  for (int y = 0; y < h; y++) {
    int32_t *out = c.d + (size_t)y * (size_t)w;
    int32_t *row = S.rows[y & 1];
    const int32_t *rN = S.rows[(y + 1) & 1];
    int32_t vW = 0, keep = 0, vNW = 0;
    int32_t vN = y > 0 ? (needs_n ? rN[0] : out[-(ptrdiff_t)w]) : 0;      // the sample above x = 0 stands in for W there (needed even when no property reads N)
    for (int x0 = 0; x0 < w; x0 += 64) {
      const int m = w - x0 < 64 ? w - x0 : 64;
      #pragma unroll 1
      for (int i = 0; i < m; i++) {
        const int x = x0 + i;
        const int32_t W_ = x > 0 ? vW : vN;                       // x == 0: the sample above (0 in the first row)
        const int32_t N_ = y > 0 ? vN : W_;
        const int32_t NW_ = (x > 0 && y > 0) ? vNW : W_;
        const int32_t nextN = (needs_n && y > 0 && x + 1 < w) ? rN[x + 1] : 0;
        int32_t myv = __mul24(cN, N_) + __mul24(cW, W_) + __mul24(cX, x) + __mul24(cY, y);
        if (cAbs) myv = myv < 0 ? -myv : myv;
        const uint64_t dec = __ballot(lane < ni && myv > my_split);
        const uint64_t lm = __ballot(lane < nl && (dec & my_need1) == my_need1 && (~dec & my_need0) == my_need0);
        const int leaf = lm ? __builtin_ctzll(lm) : 0;
        const int l_pred = __builtin_amdgcn_readlane(my_lpred, leaf), l_off = __builtin_amdgcn_readlane(my_loff, leaf);
        const int l_mul = __builtin_amdgcn_readlane(my_lmul, leaf);
        const int l_ctx = __builtin_amdgcn_readlane(my_lctx, leaf), l_clu = __builtin_amdgcn_readlane(my_lclu, leaf);
        const uint32_t u = wave_ec_read<kLds>(ev, S, b, state, (uint32_t)l_ctx, (uint32_t)l_clu);
        const int32_t res = unpack_signed(u);
        // predictors 0 .. 5: zero, W, N, their average, the select and the clamped gradient (libjxl's chroma-from-luma maps: one leaf, gradient)
        const int32_t guess = l_pred == 0 ? 0 : l_pred == 1 ? W_ : l_pred == 2 ? N_ : predict_plain_t<int32_t>(l_pred, W_, N_, NW_, 0, 0, 0, 0, 0);
        const int32_t val = (l_mul == 1 ? res : res * l_mul) + l_off + guess;
        keep = shift_in_wave1(val, keep);
        vNW = vN; vW = val; vN = nextN;
      }
      if (lane < m) { const int32_t v = keep; out[x0 + m - 1 - lane] = v; if (needs_n) row[x0 + m - 1 - lane] = v; }
    }
    __syncthreads();        // the row written above is the next row's N
  }
}

// "Chain" form of the lean loop (round 6) for the two long channels of libjxl's HF-metadata streams — block info (count x 2: `y > 0`, then thresholds on W) and the
// EPF sharpness when its contexts do not share a cluster (`N > 3`, then `W > 3`): 9 000 .. 55 000 + 65 536 samples per 2048 x 2048 LF group at ~850 clocks each in the
// lean loop.  Every decision node is a threshold, so with the decisions on y, x, N fixed (a "class": known for a whole row segment before its first sample is decoded,
// evaluated by 64 lanes for 64 samples at once) the leaf is a function of HOW MANY of the tree's W thresholds lie below W — the weighted-predictor loop's counting form:
// one v_cmp + s_bcnt1 + two v_readlane (lane e = class * (nW + 1) + count keeps the leaf's alias-table offset and its hybrid-uint configuration | predictor << 24)
// instead of two ballots over 64-bit need masks and five readlanes; the bit reader in its wave-uniform form.  Eligible: properties y, x, N, |N| on at most six nodes,
// W nodes of ONE kind (W or |W|), (1 << classes) * (nW + 1) <= 64 entries, leaves with predictor zero / W / N, multiplier 1, offset 0, tables in LDS.  Row 0 of a channel
// whose tree or predictors read N (there N stands for W: not known ahead) is left to the lean loop.  Bit-exact with it by construction: same samples, same order.
struct ChainPlan { int nW, nC, wabs; uint64_t wmask, cmask; bool ok; };
__device__ __forceinline__ ChainPlan wave_chain_plan(const DevWaveTree &WT, int lane) {
  ChainPlan P;
  const int ni = WT.ni, nl = WT.nl;
  const int pr = lane < ni ? WT.int_prop[lane] : 0;
  const uint64_t w7 = __ballot(lane < ni && pr == 7), w5 = __ballot(lane < ni && pr == 5);
  P.cmask = __ballot(lane < ni && (pr == 2 || pr == 3 || pr == 4 || pr == 6));
  P.wabs = w5 != 0; P.wmask = w5 | w7;
  P.nW = __builtin_popcountll(P.wmask); P.nC = __builtin_popcountll(P.cmask);
  const bool leaves_ok = __ballot(lane < nl && (WT.leaf_pred[lane] < 0 || WT.leaf_pred[lane] > 2 || WT.leaf_mul[lane] != 1 || WT.leaf_off[lane] != 0)) == 0;
  P.ok = leaves_ok && !(w5 && w7) && P.nW + P.nC == ni && P.nC <= 6 && ((P.nW + 1) << P.nC) <= 64;
  return P;
}
__device__ __forceinline__ void wave_decode_channel_chain(const DevECView &ev, DevBits &b, uint32_t &state, DevModScratch &S, DevWaveTree &WT,
                                                          const DevChanOut c, int lane, bool needs_n, int y_begin, const ChainPlan P) {
  const int w = c.w, h = c.h;
  const int ni = WT.ni, nl = WT.nl, nW = P.nW, nC = P.nC, stride = nW + 1;
  const int la = ev.log_alpha, lb = 12 - la;
  const int my_prop = lane < ni ? WT.int_prop[lane] : 0;
  const int my_split = lane < ni ? WT.int_split[lane] : 0x7fffffff;
  const bool isW = (P.wmask >> lane) & 1, isC = (P.cmask >> lane) & 1;
  const int my_wsplit = isW ? my_split : 0x7fffffff;                    // the counting compare: lanes that are no W node never count
  const int my_cbit = __builtin_popcountll(P.cmask & ((1ull << lane) - 1ull));      // a class node's bit in the class number
  const uint64_t my_need1 = lane < nl ? WT.leaf_need1[lane] : ~0ull, my_need0 = lane < nl ? WT.leaf_need0[lane] : ~0ull;
  const int my_lctx = lane < nl ? WT.leaf_ctx[lane] : 0, my_lpred = lane < nl ? WT.leaf_pred[lane] : 0;
  const int my_lclu = lane < nl ? (int)((const uint8_t *)S.pool)[S.ctx_off + my_lctx] : 0;
  // the W thresholds in ascending order (S.wdiv: the weighted-predictor loop's table, idle in a channel without that predictor)
  int32_t *sorted = (int32_t *)&S.wdiv[0][0];
  {
    int rank = 0;
    for (uint64_t m = P.wmask; m; m &= m - 1) { const int j = __builtin_ctzll(m); const int tj = __builtin_amdgcn_readlane(my_split, j); rank += (tj < my_split || (tj == my_split && j < lane)) ? 1 : 0; }
    __syncthreads();
    if (isW) sorted[rank] = my_split;
    __syncthreads();
  }
  // entry e = class * stride + count: a property value with exactly `count` thresholds below it stands for W, the class bits for the other decisions
  int my_aoff = 0, my_cp = 0;
  for (int k = 0, e = 0; k < (1 << nC); k++)
  for (int cc = 0; cc < stride; cc++, e++) {
    const int pc = cc == 0 ? (nW ? sorted[0] : 0) : (int)((uint32_t)sorted[cc - 1] + 1u);
    const uint64_t dec = __ballot(lane < ni && (isW ? pc > my_split : ((k >> my_cbit) & 1) != 0));
    const uint64_t lm = __ballot(lane < nl && (dec & my_need1) == my_need1 && (~dec & my_need0) == my_need0);
    const int leaf = lm ? __builtin_ctzll(lm) : 0;
    const int clu = __builtin_amdgcn_readlane(my_lclu, leaf), pred = __builtin_amdgcn_readlane(my_lpred, leaf);
    if (lane == e) { my_aoff = (clu << la) * (int)sizeof(DevAlias); my_cp = (int)(S.cfg[clu] & 0xffffffu) | (pred << 24); }
  }
  const uint8_t *pool8 = (const uint8_t *)S.pool;
  const bool any_n_pred = __ballot(lane < nl && my_lpred == 2) != 0;
  __syncthreads();
  for (int y = y_begin; y < h; y++) {
    int32_t *out = c.d + (size_t)y * (size_t)w;
    int32_t *row = S.rows[y & 1];
    const int32_t *rN = S.rows[(y + 1) & 1];
    int32_t W_ = y > 0 ? (needs_n ? rN[0] : out[-(ptrdiff_t)w]) : 0;      // x == 0: the sample above stands in for W (0 in the first row)
    int32_t keep = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
      const int m = w - x0 < 64 ? w - x0 : 64;
      // ---- parallel part: the class of sample x0 + lane
      const int x = x0 + lane;
      const int32_t Nx = (needs_n && y > 0) ? rN[x < w ? x : w - 1] : 0;
      int cls = 0;
      for (uint64_t cm = P.cmask; cm; cm &= cm - 1) {
        const int j = __builtin_ctzll(cm);
        const int pj = __builtin_amdgcn_readlane(my_prop, j), sj = __builtin_amdgcn_readlane(my_split, j), bj = __builtin_amdgcn_readlane(my_cbit, j);
        const int32_t v = pj == 2 ? y : pj == 3 ? x : pj == 4 ? (Nx < 0 ? -Nx : Nx) : Nx;
        cls |= (v > sj ? 1 : 0) << bj;
      }
      const int my_base = cls * stride;
      // ---- serial part
      #pragma unroll 1
      for (int i = 0; i < m; i++) {
        const int base_i = __builtin_amdgcn_readlane(my_base, i);
        const int32_t pv = P.wabs ? (W_ < 0 ? -W_ : W_) : W_;
        const int en = base_i + __builtin_popcountll(__ballot(pv > my_wsplit));
        const uint32_t aoff = (uint32_t)__builtin_amdgcn_readlane(my_aoff, en), cp = (uint32_t)__builtin_amdgcn_readlane(my_cp, en);
        const uint32_t res = state & 0xfff, bi = res >> lb, pos = res & ((1u << lb) - 1);
        const DevAlias e = *(const DevAlias *)(pool8 + aoff + bi * (uint32_t)sizeof(DevAlias));
        const bool right = pos >= e.cutoff;
        uint32_t u = right ? e.right : bi;
        state = (right ? e.freq1 : e.freq0) * (state >> 12) + (right ? (uint32_t)e.off1 + pos : pos);
        if (__ballot(state < (1u << 16))) state = (state << 16) | ubits_read(b, 16);
        const uint32_t split_exp = cp & 0xff;
        if (__ballot(u >= (1u << split_exp))) {
          const uint32_t msb = (cp >> 8) & 0xff, lsb = (cp >> 16) & 0xff;
          uint32_t nbits = split_exp - (msb + lsb) + ((u - (1u << split_exp)) >> (msb + lsb));
          if (nbits > 31) nbits = 31;           // corrupt stream; the final-state check flags it
          const uint32_t low = u & ((1u << lsb) - 1), tok = u >> lsb;
          const uint32_t bits = ubits_read(b, (int)nbits);
          u = (((((1u << msb) | (tok & ((1u << msb) - 1))) << nbits) | bits) << lsb) | low;
        }
        const uint32_t pred = cp >> 24;
        int32_t guess = pred == 1 ? W_ : 0;
        if (any_n_pred) { const int32_t N_i = y > 0 ? __builtin_amdgcn_readlane(Nx, i) : W_; guess = pred == 2 ? N_i : guess; }
        const int32_t val = unpack_signed(u) + guess;
        keep = shift_in_wave1(val, keep);
        W_ = val;
      }
      if (lane < m) { const int32_t v = keep; out[x0 + m - 1 - lane] = v; if (needs_n) row[x0 + m - 1 - lane] = v; }
    }
    __syncthreads();        // the row written above is the next row's N
  }
}

// Global-tree frames (libjxl's non-streaming encoder: one MA tree and one code for ALL Modular streams of the frame, 39 clusters in the
// reference's demo photographs) do not fit the LDS table pool whole, but a channel only uses the clusters of its own leaves: stage those
// 8-byte alias tables compactly, with the context-map entries of the leaves and the hybrid-uint configs renumbered to match, so that the
// lock-step loops run in their LDS (kLds) form.  Returns false when even the compact set does not fit (pool untouched).
__device__ __forceinline__ bool wave_restage_compact(const DevECView &g, int num_ctx, DevModScratch &S, const DevWaveTree &WT, int lane, uint64_t *used_out) {
  const int nl = WT.nl, la = g.log_alpha;
  const int my_ctx = lane < nl ? WT.leaf_ctx[lane] : 0;
  const int my_clu = lane < nl ? (int)g.ctx_map[my_ctx] : 0;
  uint64_t used = 0;
  for (int j = 0; j < nl; j++) used |= 1ull << (__builtin_amdgcn_readlane(my_clu, j) & 63);
  const int nc = __builtin_popcountll(used);
  const int alias_bytes = nc * (8 << la), ctx_bytes = (num_ctx + 7) & ~7;
  // (no mod_pool_want here: these are the short HF-metadata channels — a few milliseconds of a stream that holds its LDS for ~100 ms; when their tables
  // do not fit the pool the loops read them through L2 instead of making every LF wave of the process 10 KB larger for good)
  if (g.use_prefix || alias_bytes + ctx_bytes > S.pool_bytes) return false;
  __syncthreads();
  DevAlias *dst = (DevAlias *)S.pool;
  int cid = 0;
  for (uint64_t m = used; m; m &= m - 1, cid++) {
    const int s = __builtin_ctzll(m);
    for (int i = lane; i < (1 << la); i += 64) dst[(cid << la) + i] = g.alias[(s << la) + i];
    if (lane == 0) S.cfg[cid] = g.cfg[s];
  }
  if (lane < nl) ((uint8_t *)S.pool)[alias_bytes + my_ctx] = (uint8_t)__builtin_popcountll(used & ((1ull << (my_clu & 63)) - 1ull));
  if (lane == 0) S.ctx_off = alias_bytes;
  __syncthreads();
  *used_out = used;
  return true;
}

// All 64 lanes call this with identical arguments.  Returns error bits (uniform).
// kGeneral = false: the kernel carries no general lock-step loop (wave_decode_channel: ~180 VGPRs for the lifetime of the wave); a channel
// that needs one ends the stream with kErrNeedGeneral and the host decodes the frame again with the general kernel.
// kBig: trees beyond one ballot go through the block form (big_tree_build) when it fits what the pool has left behind the stream's tables.
template <bool kLds, bool kGeneral, bool kBig = false>
__device__ __forceinline__ uint32_t modular_decode_channels_wave(const DevECView &ev, DevBits &b, uint32_t &state, const DevTreeNode *gtree,
                                                        int tree_count, const DevWP &wp, DevModScratch &S, DevWaveTree &WT,
                                                        const DevChanOut *chans, int nch, int stream_id, int lane, bool m16) {
  DevECView evg = S.st.ev;                 // the stream's tables where the parser left them (HBM / L2)
  if (S.st.num_clusters <= kLocMaxClusters) evg.cfg = S.cfg;
  bool pool_packed = false;                // S.pool re-used for the packed alias tables of the weighted-predictor loop
  for (int ci = 0; ci < nch; ci++) {
    const DevChanOut c = chans[ci];
    if (c.w == 0 || c.h == 0) continue;
    __syncthreads();
    if (lane == 0) wave_tree_build(gtree, tree_count, ci, stream_id, WT, mod_stack_node(S), mod_stack_n1(S), mod_stack_n0(S));
    __syncthreads();
    if (!WT.ok) {
      if (!kBig || !kGeneral) return kErrWaveFallback;      // caller re-runs the stream with the serial walker
      const int at = (S.pool_used + 15) & ~15;              // the head of the tree cached there serves only the serial walker (which stages the stream again)
      uint32_t *big = (uint32_t *)((uint8_t *)S.pool + at);
      if (lane == 0) {
        const BigCount cnt = big_tree_count(gtree, tree_count, ci, stream_id, mod_stack_node(S));
        (void)big_tree_build(gtree, tree_count, ci, stream_id, S.st.ev.ctx_map, cnt, mod_stack_node(S), mod_stack_n1(S), mod_stack_n0(S), big, S.pool_bytes - at);
      }
      __syncthreads();
      const DevBigHdr *BH = (const DevBigHdr *)big;
      if (S.pool_bytes - at < (int)sizeof(DevBigHdr) + 64 || !BH->ok) return kErrWaveFallback;
      const bool bwp = BH->uses_wp != 0;
      if (c.w > kModMaxW && bwp) return kErrWaveFallback;
      if (BH->max_prop > 15) {
        // properties 16 ..: four per earlier channel of the stream with this channel's size and shifts, nearest first (libjxl: PrecomputeReferences; the serial
        // statement is in modular_decode_channels); lane r of the loop evaluates the four of reference r for every sample
        if (c.w > kModMaxW) return kErrWaveFallback;
        const int want = (BH->max_prop - 16) / 4 + 1;
        if (lane == 0) {
          int nref = 0;
          for (int j = ci - 1; j >= 0 && nref < want; j--)
            if (chans[j].w == c.w && chans[j].h == c.h && chans[j].hs == c.hs && chans[j].vs == c.vs) S.refp[nref++] = chans[j].d;
          for (int k = nref; k < kModMaxRefs; k++) S.refp[k] = nullptr;
        }
        __threadfence();                                    // the reference planes were written by this wave (lane 0) moments ago
        __syncthreads();
      }
      if (lane == 0 && S.walk_stat) atomicAdd(S.walk_stat + 1, 1u);
      if (kLds && !pool_packed) {
        if (m16) { if (bwp) wave_decode_channel<true, true, true, true>(ev, b, state, wp, S, WT, c, lane, 0x7fffffff, big); else wave_decode_channel<true, true, false, true>(ev, b, state, wp, S, WT, c, lane, 0x7fffffff, big); }
        else { if (bwp) wave_decode_channel<true, false, true, true>(ev, b, state, wp, S, WT, c, lane, 0x7fffffff, big); else wave_decode_channel<true, false, false, true>(ev, b, state, wp, S, WT, c, lane, 0x7fffffff, big); }
      } else {
        if (m16) { if (bwp) wave_decode_channel<false, true, true, true>(evg, b, state, wp, S, WT, c, lane, 0x7fffffff, big); else wave_decode_channel<false, true, false, true>(evg, b, state, wp, S, WT, c, lane, 0x7fffffff, big); }
        else { if (bwp) wave_decode_channel<false, false, true, true>(evg, b, state, wp, S, WT, c, lane, 0x7fffffff, big); else wave_decode_channel<false, false, false, true>(evg, b, state, wp, S, WT, c, lane, 0x7fffffff, big); }
      }
      continue;
    }
    const bool uses_wp = WT.uses_wp != 0;
    if (c.w > kModMaxW && uses_wp) return kErrWaveFallback;       // the serial walker keeps the predictor's error rows of such channels in HBM
    // the threshold-tree / weighted-predictor specialisation (see wave_decode_channel_wpfixed)
    if (m16 && uses_wp && !ev.use_prefix && c.w >= 4 && c.h >= 2 && WT.ni >= 1 && WT.ni <= 63 && WT.nl <= 64 &&
        ev.log_alpha >= 5 && ev.log_alpha <= 8 &&      // (any number of clusters: the channel's own are packed into the pool — round 5: the 128-cluster codes of libjxl's one-shot files took the general loop here, 1.4 us per LF sample instead of 0.4)
        __ballot(lane < WT.ni && WT.int_prop[lane] != 15) == 0 &&
        __ballot(lane < WT.nl && (WT.leaf_pred[lane] != 6 || WT.leaf_mul[lane] != 1 || WT.leaf_off[lane] != 0)) == 0) {
      const int ni = WT.ni, nl = WT.nl;
      const int my_split = lane < ni ? WT.int_split[lane] : 0x7fffffff;
      const uint64_t my_need1 = lane < nl ? WT.leaf_need1[lane] : ~0ull, my_need0 = lane < nl ? WT.leaf_need0[lane] : ~0ull;
      const int my_lclu = lane < nl ? (int)evg.ctx_map[WT.leaf_ctx[lane]] : 0;
      CluSet used; used.clear();                      // clusters of this channel's leaves
      for (int j = 0; j < nl; j++) used.add(__builtin_amdgcn_readlane(my_lclu, j));
      const int la_p = ev.log_alpha;
      const int dsz = wave_alias_dsz(evg.alias, used, la_p, lane);
      const int packed = dsz ? wave_packed_bytes(used, la_p, dsz) : kModPoolBytes + 1;
      if (dsz) mod_pool_want(S, packed, lane);
      // the lean build has no general loop to fall back to: a pool sized by the previous flight that is too small for this channel is a retry
      if (!kGeneral && packed > S.pool_bytes && packed <= kModPoolBytes) return kErrNeedPool;
      if (packed <= S.pool_bytes) {
      // rank the thresholds; lane c then holds a value with exactly c thresholds below it
      int rank = 0;
      for (int j = 0; j < ni; j++) { const int tj = __builtin_amdgcn_readlane(my_split, j); rank += (tj < my_split || (tj == my_split && j < lane)) ? 1 : 0; }
      __syncthreads();
      if (lane < ni) S.rows[0][rank] = my_split;     // rows[0] as 64-entry scratch (row 0 below rewrites it)
      __syncthreads();
      const int rep = lane == 0 ? S.rows[0][0] : (lane <= ni ? S.rows[0][lane - 1] + 1 : 0);
      int my_off = 0, my_doff = 0, my_cfg = 0;
      for (int cc = 0; cc <= ni; cc++) {
        const int pc = __builtin_amdgcn_readlane(rep, cc);
        const uint64_t dec = __ballot(lane < ni && pc > my_split);
        const uint64_t lm = __ballot(lane < nl && (dec & my_need1) == my_need1 && (~dec & my_need0) == my_need0);
        const int leaf = lm ? __builtin_ctzll(lm) : 0;
        const int clu = __builtin_amdgcn_readlane(my_lclu, leaf);
        const int cid = used.rank(clu);      // compact index of the cluster in the packed pool
        if (lane == cc) { my_off = cid * (3 << la_p); my_doff = used.count() * (3 << la_p) + cid * 2 * dsz; my_cfg = (int)S.st.ev.cfg[clu]; }      // the stream's own config table (S.cfg may hold a compact renumbering)
      }
      __syncthreads();
      // once the pool holds packed tables the other loops of this stream read their tables through L2 (or restage them compactly)
      if (wave_pack_alias(evg.alias, used, la_p, dsz, S, lane)) {   // per channel: the set of clusters may differ
        pool_packed = true;                            // the pool no longer holds the stream's 8-byte tables / context map
        wave_decode_channel_wpfixed(ev, b, state, wp, S, c, lane, my_split, my_off, my_doff, my_cfg, /*y_begin=*/0, wave_dsz_log2(dsz));
        continue;
      }
      // symbols >= 128 in an LF stream (not produced by libjxl): the pool is untouched, the general loops below take the channel
      }
    }
    // the LDS (kLds) form of the remaining loops needs the stream's tables in the pool: there from modular_stream_stage, or staged
    // compactly for this channel now
    bool lds_now = kLds && !pool_packed;
    uint64_t used = 0;
    bool compact = false;
    if (!lds_now && !ev.use_prefix && S.st.num_clusters <= kLocMaxClusters && WT.nl >= 1 && WT.nl <= 64 &&
        wave_restage_compact(S.st.ev, S.st.num_ctx, S, WT, lane, &used)) {
      lds_now = true; compact = true; pool_packed = true;
      evg.cfg = S.st.ev.cfg;                        // S.cfg now holds the compact renumbering
    }
    // uniform-leaf channel (see wave_decode_channel_uniform)
    if (!ev.use_prefix && WT.nl >= 1 && WT.nl <= 64) {
      const int nl = WT.nl;
      const int my_clu = lane < nl ? (int)S.st.ev.ctx_map[WT.leaf_ctx[lane]] : -1;
      const int clu0 = __builtin_amdgcn_readfirstlane(my_clu);
      if (__ballot(lane < nl && (my_clu != clu0 || WT.leaf_pred[lane] != 0 || WT.leaf_mul[lane] != 1 || WT.leaf_off[lane] != 0)) == 0) {
        const int cl = compact ? __builtin_popcountll(used & ((1ull << (clu0 & 63)) - 1ull)) : clu0;
        if (lds_now) wave_decode_channel_uniform<true>(ev, b, state, S, c, lane, (uint32_t)cl);
        else wave_decode_channel_uniform<false>(evg, b, state, S, c, lane, (uint32_t)clu0);
        continue;
      }
    }
    // lean loop (see wave_decode_channel_lean): y / x / N / W properties only, zero / W / N predictors
    if (m16 && !ev.use_prefix && !uses_wp && WT.ni <= 64 && WT.nl >= 1 && WT.nl <= 64) {
      const bool props_ok = __ballot(lane < WT.ni && (WT.int_prop[lane] < 2 || WT.int_prop[lane] > 7)) == 0;
      const bool preds_ok = __ballot(lane < WT.nl && (WT.leaf_pred[lane] < 0 || WT.leaf_pred[lane] > 5)) == 0;
      const bool needs_n = __ballot((lane < WT.ni && (WT.int_prop[lane] == 4 || WT.int_prop[lane] == 6)) || (lane < WT.nl && WT.leaf_pred[lane] >= 2)) != 0;
      if (props_ok && preds_ok && (!needs_n || c.w <= kModMaxW)) {
        // rows [0, y_split) in the lean loop, the rest in its chain form where the tree allows it (see wave_decode_channel_chain)
        ChainPlan P; P.nW = P.nC = P.wabs = 0; P.wmask = P.cmask = 0; P.ok = false;
        if (lds_now && JXL_MOD_CHAIN) P = wave_chain_plan(WT, lane);
        const int y_split = P.ok ? (needs_n ? 1 : 0) : c.h;
        if (y_split > 0) {
          if (lds_now) wave_decode_channel_lean<true>(ev, b, state, S, WT, c, lane, needs_n, y_split);
          else wave_decode_channel_lean<false>(evg, b, state, S, WT, c, lane, needs_n, y_split);
        }
        if (y_split < c.h) wave_decode_channel_chain(ev, b, state, S, WT, c, lane, needs_n, y_split, P);
        continue;
      }
    }
    if (!kGeneral) return kErrNeedGeneral;
    if (lds_now) {
      if (m16) { if (uses_wp) wave_decode_channel<true, true, true>(ev, b, state, wp, S, WT, c, lane); else wave_decode_channel<true, true, false>(ev, b, state, wp, S, WT, c, lane); }
      else { if (uses_wp) wave_decode_channel<true, false, true>(ev, b, state, wp, S, WT, c, lane); else wave_decode_channel<true, false, false>(ev, b, state, wp, S, WT, c, lane); }
    } else {        // the tables stay in HBM / L2
      if (m16) { if (uses_wp) wave_decode_channel<false, true, true>(evg, b, state, wp, S, WT, c, lane); else wave_decode_channel<false, true, false>(evg, b, state, wp, S, WT, c, lane); }
      else { if (uses_wp) wave_decode_channel<false, false, true>(evg, b, state, wp, S, WT, c, lane); else wave_decode_channel<false, false, false>(evg, b, state, wp, S, WT, c, lane); }
    }
  }
  return 0;
}

// Stream-level wrapper: every lane calls it; falls back to the serial walker when the tree is too large.
template <bool kGeneral = true, bool kBig = false>
__device__ __forceinline__ uint32_t modular_stream_decode_wave(DevModScratch &S, const DevChanOut *chans, int nch, int stream_id, int lane) {
  DevModStream &st = S.st;
  if (st.err) return st.err;
  if (st.ev.lz77 || st.wide32) {          // LZ77 copies, float32 samples: the serial walker (lane 0) only
    if (lane == 0) S.fallback_err = modular_stream_decode(S, chans, nch, stream_id);
    __syncthreads();
    return S.fallback_err;
  }
  DevECView ev = st.ev;
  ev.ctx_map = S.ctx_map;
  if (st.num_clusters <= kLocMaxClusters) ev.cfg = S.cfg;
  if (!ev.use_prefix) ev.alias = S.alias;
  DevBits b = st.b;
  uint32_t state = ans_init(ev, b);
  const bool lds = S.ctx_lds && S.alias_lds && !ev.use_prefix;
  uint32_t err = lds ? modular_decode_channels_wave<true, kGeneral, kBig>(ev, b, state, st.tree, st.count, st.wp, S, S.wt, chans, nch, stream_id, lane, st.m16 != 0)
                     : modular_decode_channels_wave<false, kGeneral, kBig>(ev, b, state, st.tree, st.count, st.wp, S, S.wt, chans, nch, stream_id, lane, st.m16 != 0);
  __syncthreads();
  if (err == kErrWaveFallback) {
    if (lane == 0 && S.walk_stat) atomicAdd(S.walk_stat, 1u);
    // An earlier channel of the stream may have re-used the table pool (packed alias tables of the weighted-predictor loop, compact restaging):
    // the serial walker starts the stream over and needs the tables as modular_stream_stage laid them out
    modular_stream_stage(S, lane, 64);
    __syncthreads();
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
