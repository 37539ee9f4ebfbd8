// jxl_coder_amd/csrc/dev_modular.h — device-side Modular sub-bitstream decoder (ISO/IEC 18181-1 Annex H):
// MA-tree context modelling, the 14 predictors and the weighted (self-correcting) predictor.
// Used for the LF image and HF metadata of VarDCT frames (28 % of the bits of a 4K q90 frame), i.e. what
// libjxl's ModularFrameDecoder does under JxlDecoderProcessInput (reference call site
// jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).  Integer-exact by construction.
//
// Execution model: one wavefront per stream, lane 0 walks the stream; the previous two rows, the WP error
// state, the property vector and the head of the MA tree live in LDS so that the serial lane never waits on
// HBM for its neighbourhood.
#pragma once
#include <stddef.h>
#include "dev_echeader.h"

namespace jxlamd {

constexpr int kLfMaxCh = 40;          // channel descriptors of an LF-group workgroup: 3 LF channels, 4 HF-metadata channels, <= 32 ModularLfGroup channels (host_parse.cpp)
constexpr int kModGroupDesc = 80;     // channel descriptors of a ModularGroup workgroup: <= 64 channel rectangles (host_parse.cpp: kModMaxGroupCh) + the palettes of its own transforms (kModMaxLocalTr)
constexpr int kModMaxW = 256;         // widest channel a device stream may carry (LF group = 256 LF samples; 256-px lossless groups)
constexpr int kWpMaxW = 256;
constexpr int kTreeLds = 128;         // MA-tree nodes cached in LDS

struct DevWP { int32_t p1, p2, p3a, p3b, p3c, p3d, p3e, w[4]; };
constexpr int kModMaxRefs = 12;                    // previous channels an MA tree may look at (properties 16 ..: four per channel; libjxl's encoder offers up to 11)
struct DevChanOut { int32_t *d; int32_t w, h; int16_t hs, vs; };      // hs / vs: the channel's shifts (-1: a meta channel) — what decides, with the size, which earlier channels the "previous channel" MA properties read

#ifndef JXL_MOD_POOL_BYTES
#define JXL_MOD_POOL_BYTES 30720
#endif
constexpr int kModPoolBytes = JXL_MOD_POOL_BYTES;  // LARGEST LDS table pool of one stream: 8-byte alias tables of up to 15 clusters + context map + tree head, or the packed tables of up to 23 clusters (dev_modular_wave.h)
// The pool is the tail of DevModScratch and the LF kernels allocate it as dynamic LDS: what a stream's tables need decides how many LF
// streams a CU holds (a stream keeps its LDS for ~100 ms, and LDS-time is what the flights of several decoder contexts run out of first).
// libjxl's streaming encoder (log_alpha 8): 8 - 21 clusters x 896 B packed for the LF coefficients of the 256 bench frames (round 6: three-byte entries), <= 7 x 2 KB for the HF
// metadata; its one-shot encoder (global tree, 39 clusters, log_alpha 7): 28 x 512 B.  Streams report what they would have liked (DevModScratch::pool_want), the host sizes the next launch.
constexpr int kModPoolMin = 12288;                 // header parser's working arrays (LocalTmp) and the placement bitmap (8 KB) live there too

struct DevTr { int32_t id, begin_c, rct_type, num_c, nb_colours, nb_deltas, d_pred; };
constexpr int kModMaxLocalTr = 8;      // transforms of one group stream (libjxl: per-channel palettes + one multi-channel palette + an RCT)
struct DevTrList { DevTr t[kModMaxLocalTr]; int32_t n; };

struct DevModStream {                 // what lane 0 hands to the other lanes / to the next phase of a stream
  DevBits b;
  DevWP wp;
  const DevTreeNode *tree; int32_t count;
  DevECView ev;
  int32_t num_ctx, num_clusters;
  int32_t m16;
  int32_t wide32;                     // float32 samples (the planes hold bit patterns of any magnitude): serial loop only, and a neighbourhood sum that leaves 32 bits is refused —
                                      // libjxl's specialised loops (gradient-only / weighted-only trees) evaluate such sums without the wrap of its generic loop, which is the one restated here
  uint32_t err;
};

struct alignas(16) DevWaveTree {      // LDS: the pruned tree of one channel in ballot form (16-byte aligned: the WP-only loop overlays b128 records)
  int32_t int_prop[64], int_split[64];
  uint64_t leaf_need1[64], leaf_need0[64];
  int32_t leaf_ctx[64], leaf_pred[64], leaf_off[64], leaf_mul[64];
  int32_t ni, nl, ok, uses_wp;
};      // (the builders' DFS stacks — 1.25 KB — no longer live here: mod_stack_node / _n1 / _n0 below lay them over the weighted predictor's error rows)

struct DevModScratch {                // per-wave working memory (LDS on the GPU)
  int32_t rows[3][kModMaxW + 8];      // cur / prev / prevprev rows
  uint32_t wp_pred_err[4][2 * (kWpMaxW + 2)];      // (between channels: the tree builders' DFS stacks, mod_stack_*)
  int32_t wp_err[2 * (kWpMaxW + 2)];
  uint32_t divlut[64];                // (1<<24)/(i+1): the WP's division-free reciprocal table
  uint32_t wdiv[4][64];               // divlut pre-multiplied by the WP header weights (wave_decode_channel_wpfixed)
  uint32_t ring[128];                 // the next 512 bytes of the stream, refilled half by half far ahead of the reader (wave_decode_channel_wpfixed)
  uint32_t cfg[kLocMaxClusters];
  // Table pool, carved per stream by modular_stream_stage: [alias tables | context map | head of the MA tree].  A part
  // that does not fit stays in HBM (the pointers below then address the HBM copy).  libjxl's LF streams need
  // 9 clusters x 256 alias entries = 18 KiB; 20 KiB keeps the workgroup under 40 KiB LDS => 4 streams per CU.
  const DevAlias *alias;              // LDS when alias_lds
  const uint8_t *ctx_map;             // LDS when ctx_lds
  const DevTreeNode *tree;            // first tree_ncache nodes of the stream's tree
  int32_t alias_lds, ctx_lds, tree_ncache, ctx_off;   // ctx_off: byte offset of the context map inside the pool
  int32_t pool_used;                  // bytes of the pool the stream's [alias tables | context map] occupy: what follows is the tree head — or, in the big-tree wave loop, the channel's tree in block form
  DevModStream st;
  DevChanOut *ch;                     // channel descriptors of the current stream: an array of the KERNEL's choosing (LDS) — the LF kernels keep 40 behind their table pool,
                                      // the Modular-frame kernels kModMaxCh (round 5: 3 KB of every LF workgroup's fixed LDS were descriptors only the latter use)
  int32_t grp_dec;                    // channels the current group stream carries after its own transforms' meta-apply (palette channels in front)
  int32_t grp_src[kModMaxGroupCh], grp_n;         // a group stream's channels: which stream channel of the frame each one is a rectangle of (LDS: keeps the kernel free of scratch)
  DevTrList trs;                      // transforms of the current stream header
  DevChanOut pal_saved[kModMaxLocalTr][3];   // group-level palettes: the colour channels 1.. each one folded away (their buffers receive the colours again)
  // wt, props, refp in this order: the weighted-predictor loop's chunk records (dev_modular_wave.h: DevWpFixedLds, 2 816 bytes) overlay the three — the tree sits in
  // registers by then, props serve the serial walker and refp the block-form loops only
  DevWaveTree wt;
  int32_t props[16 + 4 * kModMaxRefs];
  const int32_t *refp[kModMaxRefs];    // wave loop, block-form trees: the previous channels whose samples feed properties 16 .. (same size and shifts, nearest first)
  uint32_t fallback_err;
  uint32_t *wide_wp;                  // HBM: the weighted predictor's error rows for channels wider than the LDS rows (kWideWpInts; null: such channels are rejected)
  DevLz lz;                           // LZ77 state of the current stream (serial walker; window in HBM, set by the stream's caller)
  int32_t pool_bytes;                 // bytes of `pool` actually backed by LDS in this launch (kModPoolMin .. kModPoolBytes)
  uint32_t *pool_want;                // where to report (max) the pool bytes this stream's per-channel table sets need; may be null
  uint32_t *walk_stat;                // two counters (words 2 and 3 of the frame's flag block): streams that went to the serial walker, channels decoded with their tree in block form; may be null
  uint64_t pool[kModPoolBytes / 8];   // LAST member: the kernels allocate only pool_bytes of it
};
// DFS stacks of the tree builders (wave_tree_build, tree_facts, big_tree_count / big_tree_build): they run at the start of a channel, when the weighted predictor's error
// rows of the previous channel are dead and this channel's are not yet cleared — LDS, not scratch, and no bytes of their own
JXL_DEV int32_t *mod_stack_node(DevModScratch &S) { return (int32_t *)&S.wp_pred_err[0][0]; }
JXL_DEV uint64_t *mod_stack_n1(DevModScratch &S) { return (uint64_t *)&S.wp_pred_err[1][0]; }
JXL_DEV uint64_t *mod_stack_n0(DevModScratch &S) { return (uint64_t *)&S.wp_pred_err[2][0]; }
static_assert(sizeof(((DevModScratch *)0)->wp_pred_err[0]) >= 64 * 8 && (offsetof(DevModScratch, wp_pred_err) % 8) == 0 && (sizeof(((DevModScratch *)0)->wp_pred_err[0]) % 8) == 0, "the DFS stacks fit one error row each");
JXL_DEV void mod_pool_want(DevModScratch &S, int bytes, int tid) {
  if (tid != 0 || !S.pool_want) return;
#ifdef __HIPCC__
  atomicMax(S.pool_want, (uint32_t)bytes);
#else
  if ((uint32_t)bytes > *S.pool_want) *S.pool_want = (uint32_t)bytes;
#endif
}



JXL_DEV int32_t floor_log2_u32(uint32_t x) { return 31 - __builtin_clz(x); }
JXL_DEV int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

struct WPState {
  int64_t prediction[4];
  int64_t pred;
};

JXL_DEV uint32_t wp_error_weight(const uint32_t *divlut, uint32_t x, uint32_t maxweight) {
  int shift = floor_log2_u32(x + 1) - 5;
  if (shift < 0) shift = 0;
  return 4 + ((maxweight * divlut[x >> shift]) >> shift);
}

// where the weighted predictor keeps its two rows of errors: LDS, or — channels wider than kWpMaxW (Modular group sizes 512 / 1024) — HBM
constexpr int kWideMaxW = 1024;
constexpr int kWideWpInts = 5 * 2 * (kWideMaxW + 2);
struct WpRows { uint32_t *pe[4]; int32_t *te; };
JXL_DEV WpRows wp_rows(DevModScratch &S, bool wide, int w) {
  WpRows r;
  if (!wide) { for (int k = 0; k < 4; k++) r.pe[k] = S.wp_pred_err[k]; r.te = S.wp_err; }
  else { for (int k = 0; k < 4; k++) r.pe[k] = S.wide_wp + (size_t)k * 2 * (size_t)(w + 2); r.te = (int32_t *)(S.wide_wp + (size_t)4 * 2 * (size_t)(w + 2)); }
  return r;
}

JXL_DEV int64_t wp_predict(DevModScratch &S, const WpRows &R, WPState &st, const DevWP &h, int x, int y, int xs, int64_t N, int64_t W,
                           int64_t NE, int64_t NW, int64_t NN, int32_t &max_err) {
  int cur_row = (y & 1) ? 0 : (xs + 2);
  int prev_row = (y & 1) ? (xs + 2) : 0;
  int pos_N = prev_row + x;
  int pos_NE = x < xs - 1 ? pos_N + 1 : pos_N;
  int pos_NW = x > 0 ? pos_N - 1 : pos_N;
  uint32_t w[4];
  for (int i = 0; i < 4; i++) {
    uint32_t e = R.pe[i][pos_N] + R.pe[i][pos_NE] + R.pe[i][pos_NW];
    w[i] = wp_error_weight(S.divlut, e, (uint32_t)h.w[i]);
  }
  N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
  int64_t teW = x == 0 ? 0 : R.te[cur_row + x - 1];
  int64_t teN = R.te[pos_N], teNW = R.te[pos_NW], teNE = R.te[pos_NE];
  int64_t sumWN = teN + teW;
  int64_t p = teW;
  if (iabs64(teN) > iabs64(p)) p = teN;
  if (iabs64(teNW) > iabs64(p)) p = teNW;
  if (iabs64(teNE) > iabs64(p)) p = teNE;
  max_err = (int32_t)p;
  st.prediction[0] = W + NE - N;
  st.prediction[1] = N - (((sumWN + teNE) * h.p1) >> 5);
  st.prediction[2] = W - (((sumWN + teNW) * h.p2) >> 5);
  st.prediction[3] = N - ((teNW * h.p3a + teN * h.p3b + teNE * h.p3c + (NN - N) * h.p3d + (NW - W) * h.p3e) >> 5);
  uint32_t wsum = w[0] + w[1] + w[2] + w[3];
  int lw = floor_log2_u32(wsum);
  wsum = 0;
  for (int i = 0; i < 4; i++) { w[i] >>= lw - 4; wsum += w[i]; }
  int64_t sum = (int64_t)(wsum >> 1) - 1;
  for (int i = 0; i < 4; i++) sum += st.prediction[i] * (int64_t)w[i];
  st.pred = (sum * (int64_t)S.divlut[wsum - 1]) >> 24;
  if (((teN ^ teW) | (teN ^ teNW)) > 0) return (st.pred + 3) >> 3;
  int64_t mx = W > NE ? W : NE; if (N > mx) mx = N;
  int64_t mn = W < NE ? W : NE; if (N < mn) mn = N;
  if (st.pred > mx) st.pred = mx;
  if (st.pred < mn) st.pred = mn;
  return (st.pred + 3) >> 3;
}

JXL_DEV void wp_update(const WpRows &R, const WPState &st, int64_t val, int x, int y, int xs) {
  int cur_row = (y & 1) ? 0 : (xs + 2);
  int prev_row = (y & 1) ? (xs + 2) : 0;
  val *= 8;
  R.te[cur_row + x] = (int32_t)(st.pred - val);
  for (int i = 0; i < 4; i++) {
    uint32_t err = (uint32_t)((iabs64(st.prediction[i] - val) + 3) >> 3);
    R.pe[i][cur_row + x] = err;
    R.pe[i][prev_row + x + 1] += err;
  }
}

JXL_DEV int64_t clamped_gradient(int64_t n, int64_t w, int64_t l) {
  int64_t m = n < w ? n : w, M = n < w ? w : n;
  int64_t g = n + w - l;
  return g < m ? m : g > M ? M : g;
}

JXL_DEV int64_t predict_plain(int predictor, int64_t W, int64_t N, int64_t NW, int64_t NE, int64_t NN, int64_t WW,
                              int64_t NEE, int64_t wp) {
  switch (predictor) {
    case 0: return 0;
    case 1: return W;
    case 2: return N;
    case 3: return (W + N) / 2;
    case 4: { int64_t p = W + N - NW; return iabs64(p - W) < iabs64(p - N) ? W : N; }
    case 5: return clamped_gradient(N, W, NW);
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

// Does the part of the MA tree reachable for this (channel, stream) use the weighted predictor (property 15 or
// predictor 6), and which is the largest property it tests?  Properties 0 and 1 are static per channel, so the
// unreachable branches are pruned exactly as libjxl's tree filtering does.
struct TreeFacts { int uses_wp; int max_prop; };
JXL_DEV TreeFacts tree_facts(const DevTreeNode *tree, int count, int chan, int stream, int32_t *stack) {
  TreeFacts f; f.uses_wp = 0; f.max_prop = 0;
  int sp = 0;
  stack[sp++] = 0;
  int guard = 0;
  while (sp > 0 && guard++ < 4 * count + 8) {
    const DevTreeNode nd = tree[stack[--sp]];
    if (nd.prop < 0) { if (nd.lchild == 6) f.uses_wp = 1; continue; }
    if (nd.prop == 0 || nd.prop == 1) {
      int v = nd.prop == 0 ? chan : stream;
      stack[sp++] = v > nd.splitval ? nd.lchild : nd.rchild;
      continue;
    }
    if (nd.prop == 15) f.uses_wp = 1;
    if (nd.prop > f.max_prop) f.max_prop = nd.prop;
    if (sp + 2 > 64) { f.uses_wp = 1; f.max_prop = 99; break; }   // pathological depth: take the safe answer
    stack[sp++] = nd.lchild; stack[sp++] = nd.rchild;
  }
  return f;
}

// ---- MA trees beyond one ballot (more than 64 decision nodes or leaves after pruning): the tree in BLOCKS for the wave loop (dev_modular_wave.h).
// A block is a connected piece of the tree with at most 63 decision nodes — so at most 64 exits, each a leaf or the root of another block — cut breadth
// first, so that a walk crosses ~depth / 6 blocks.  Inside a block the wave evaluates as it does a small tree: lane i decides node i, lane j tests exit j's
// (must-be-1, must-be-0) masks over the block's decisions.  libjxl's one-shot encoder writes ONE tree for all Modular streams of a frame (a default lossy
// RGBA photograph: 1015 nodes, 459 leaves for the alpha channel's group streams); the serial walker took 3.6 - 6 us per sample there.
// Layout (32-bit words from the area's start): DevBigHdr | nodes (prop, split) | exit masks (need1, need0: 4 words) | exit words | (mul, off) of the exits
// when some leaf has a multiplier / offset | builder's block queue.  Exit word: bit 31 set = block (ni << 25 | first node << 13 | first exit), else leaf
// (bit 30: has multiplier / offset, predictor << 26, cluster << 18, context).
constexpr int kBigBlkNodes = 63;
struct DevBigHdr {
  int32_t ok, nblocks, nnodes, nexits, nonunit, uses_wp;
  int32_t off_nodes, off_need, off_exit, off_mulo, off_blk, capB, capE;
  uint32_t root_word;
  int32_t max_prop;              // largest property the tree tests (> 15: properties of previous channels)
  int32_t pad[1];
};
static_assert(sizeof(DevBigHdr) == 64, "DevBigHdr: 16 words");
struct BigCount { int32_t ni, nl, nonunit, uses_wp, ok, max_prop; };
// how large is the tree reachable for (chan, stream), and which is the largest property it tests?  ok = 0: a property beyond the previous channels the loops keep, or deeper than the stack
JXL_DEV BigCount big_tree_count(const DevTreeNode *tree, int count, int chan, int stream, int32_t *stack) {
  BigCount r; r.ni = r.nl = r.nonunit = r.uses_wp = 0; r.ok = 1; r.max_prop = 0;
  int sp = 0, guard = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    if (++guard > 4 * count + 16) { r.ok = 0; return r; }
    const DevTreeNode nd = tree[stack[--sp]];
    if (nd.prop < 0) { r.nl++; if (nd.lchild == 6) r.uses_wp = 1; if (nd.rchild != 1 || nd.offset != 0) r.nonunit = 1; continue; }
    if (nd.prop == 0 || nd.prop == 1) { const int v = nd.prop == 0 ? chan : stream; stack[sp++] = v > nd.splitval ? nd.lchild : nd.rchild; continue; }
    if (nd.prop >= 16 + 4 * kModMaxRefs || sp + 2 > 64) { r.ok = 0; return r; }      // (properties 16 ..: four per previous channel, up to kModMaxRefs of them)
    if (nd.prop == 15) r.uses_wp = 1;
    if (nd.prop > r.max_prop) r.max_prop = nd.prop;
    r.ni++;
    stack[sp++] = nd.lchild; stack[sp++] = nd.rchild;
  }
  return r;
}
// one work-item builds the blocks into `big` (big_bytes of LDS); queue = three arrays of 64 entries (the small-tree builder's DFS stacks)
JXL_DEV bool big_tree_build(const DevTreeNode *tree, int count, int chan, int stream, const uint8_t *ctx_map, const BigCount &cnt,
                            int32_t *q_node, uint64_t *q_n1, uint64_t *q_n0, uint32_t *big, int big_bytes) {
  DevBigHdr &H = *(DevBigHdr *)big;
  H.ok = 0;
  if (!cnt.ok || cnt.ni > 4095 || big_bytes < (int)sizeof(DevBigHdr) + 64) return false;
  const int total = big_bytes / 4, E = 5 + (cnt.nonunit ? 2 : 0);
  const int off_nodes = 16, off_need = (off_nodes + 2 * cnt.ni + 3) & ~3;
  int capB = (total - off_need - (cnt.nl - 1) * E) / (E + 2);
  if (capB < 1) return false;
  if (capB > cnt.ni + 1) capB = cnt.ni + 1;
  int capE = cnt.nl + capB - 1;
  if (capE > 8191) return false;
  H.nonunit = cnt.nonunit; H.uses_wp = cnt.uses_wp; H.max_prop = cnt.max_prop;
  H.off_nodes = off_nodes; H.off_need = off_need; H.off_exit = off_need + 4 * capE; H.off_mulo = H.off_exit + capE;
  H.off_blk = H.off_mulo + (cnt.nonunit ? 2 * capE : 0); H.capB = capB; H.capE = capE;
  int32_t *nodes = (int32_t *)big + off_nodes;
  uint32_t *need = big + off_need, *exw = big + H.off_exit;
  int32_t *mulo = (int32_t *)big + H.off_mulo, *blk = (int32_t *)big + H.off_blk;     // blk[2 * b] = root node, blk[2 * b + 1] = the exit (index) of its parent block that names it
  int nblocks = 1, nnodes = 0, nexits = 0;
  blk[0] = 0; blk[1] = -1;
  for (int bk = 0; bk < nblocks; bk++) {
    const int node_off = nnodes, exit_off = nexits;
    int ni = 0, nl = 0, head = 0, pend = 1;
    q_node[0] = blk[2 * bk]; q_n1[0] = 0; q_n0[0] = 0;
    while (pend > 0) {
      int idx = q_node[head]; const uint64_t n1 = q_n1[head], n0 = q_n0[head];
      head = (head + 1) & 63; pend--;
      DevTreeNode nd = tree[idx];
      for (int guard = 0; nd.prop == 0 || nd.prop == 1; guard++) {      // static decisions: the channel and the stream are known
        if (guard > count) return false;
        const int v = nd.prop == 0 ? chan : stream;
        idx = v > nd.splitval ? nd.lchild : nd.rchild; nd = tree[idx];
      }
      if (nd.prop >= 0 && ni < kBigBlkNodes) {
        const int i = ni++;
        if (nnodes >= cnt.ni) return false;
        nodes[2 * nnodes] = nd.prop; nodes[2 * nnodes + 1] = nd.splitval; nnodes++;
        int t = (head + pend) & 63;
        q_node[t] = nd.lchild; q_n1[t] = n1 | (1ull << i); q_n0[t] = n0; pend++;      // decision true  -> left
        t = (head + pend) & 63;
        q_node[t] = nd.rchild; q_n1[t] = n1; q_n0[t] = n0 | (1ull << i); pend++;      // decision false -> right
        continue;
      }
      if (nexits >= capE) return false;
      const int ex = nexits++; nl++;
      need[4 * ex] = (uint32_t)n1; need[4 * ex + 1] = (uint32_t)(n1 >> 32); need[4 * ex + 2] = (uint32_t)n0; need[4 * ex + 3] = (uint32_t)(n0 >> 32);
      if (nd.prop < 0) {
        const uint32_t ctx = (uint32_t)nd.splitval, pred = (uint32_t)nd.lchild;
        if (ctx >= (1u << 18) || pred > 13) return false;
        const bool nonunit = nd.rchild != 1 || nd.offset != 0;
        exw[ex] = (nonunit ? 1u << 30 : 0u) | (pred << 26) | ((uint32_t)ctx_map[ctx] << 18) | ctx;
        if (cnt.nonunit) { mulo[2 * ex] = nd.rchild; mulo[2 * ex + 1] = nd.offset; }
      } else {                                       // the block is full: this node roots another one (its word is written when that block is built)
        if (nblocks >= capB) return false;
        blk[2 * nblocks] = idx; blk[2 * nblocks + 1] = ex; nblocks++;
        exw[ex] = 0x80000000u;
      }
    }
    (void)nl;
    if (exit_off > 8191 || node_off > 4095) return false;
    const uint32_t word = 0x80000000u | ((uint32_t)ni << 25) | ((uint32_t)node_off << 13) | (uint32_t)exit_off;
    if (bk == 0) H.root_word = word; else exw[blk[2 * bk + 1]] = word;
  }
  H.nblocks = nblocks; H.nnodes = nnodes; H.nexits = nexits; H.ok = 1;
  return true;
}
// reference evaluation of a tree in block form (one work-item; the CPU harness checks the builder with it against the plain walk)
JXL_DEV uint32_t big_tree_eval(const uint32_t *big, const int32_t *props, int *exit_index) {
  const DevBigHdr &H = *(const DevBigHdr *)big;
  uint32_t e = H.root_word; int eidx = 0;
  while (e >> 31) {
    const int ni = (int)((e >> 25) & 63), noff = (int)((e >> 13) & 4095), eoff = (int)(e & 8191);
    uint64_t dec = 0;
    for (int i = 0; i < ni; i++) { const int32_t *nd = (const int32_t *)big + H.off_nodes + 2 * (noff + i); if (props[nd[0]] > nd[1]) dec |= 1ull << i; }      // (props: 16 + 4 per previous channel)
    int leaf = -1;
    for (int j = 0; j <= ni && leaf < 0; j++) {
      const uint32_t *m = big + H.off_need + 4 * (eoff + j);
      const uint64_t n1 = m[0] | ((uint64_t)m[1] << 32), n0 = m[2] | ((uint64_t)m[3] << 32);
      if ((dec & n1) == n1 && (~dec & n0) == n0) leaf = j;
    }
    if (leaf < 0) return 0xFFFFFFFFu;
    eidx = eoff + leaf; e = big[H.off_exit + eidx];
    if (e == 0x80000000u) return 0xFFFFFFFFu;      // (a block word always carries ni >= 0 and offsets; the bare flag means "never built")
  }
  *exit_index = eidx;
  return e;
}

// Decode the channels of one modular stream (lane 0 only).  `chans[i].d` are dense w*h int32 planes in HBM.
// Channels up to kModMaxW wide keep their three live rows in LDS; wider ones (the count x 2 block-info channel
// of HF metadata) read their neighbourhood back from the plane itself.  Returns 0 or error bits.
JXL_DEV uint32_t modular_decode_channels(const DevECView &ev, DevBits &b, uint32_t &state, const DevTreeNode *gtree,
                                         int tree_count, const DevWP &wp, DevModScratch &S,
                                         const DevChanOut *chans, int nch, int stream_id) {
  int32_t *props = S.props;
  props[1] = stream_id;
  for (int ci = 0; ci < nch; ci++) {
    const DevChanOut &c = chans[ci];
    const int w = c.w, h = c.h;
    if (w == 0 || h == 0) continue;
    const TreeFacts tf = tree_facts(gtree, tree_count, ci, stream_id, mod_stack_node(S));
#ifdef JXL_EMUL_TRACE
    {
      unsigned props_used = 0, preds_used = 0; int leaves = 0, nonunit = 0, sp = 0, st[256]; st[sp++] = 0;
      bool reach[256] = {false}; int nreach = 0;
      while (sp > 0) { const DevTreeNode nd = gtree[st[--sp]];
        if (nd.prop < 0) { preds_used |= 1u << nd.lchild; leaves++; if (nd.rchild != 1 || nd.offset != 0) nonunit++; const int cl = ev.ctx_map[nd.splitval]; if (!reach[cl]) { reach[cl] = true; nreach++; } continue; }
        if (nd.prop == 0 || nd.prop == 1) { int v = nd.prop == 0 ? ci : stream_id; st[sp++] = v > nd.splitval ? nd.lchild : nd.rchild; continue; }
        props_used |= 1u << nd.prop; if (sp + 2 < 256) { st[sp++] = nd.lchild; st[sp++] = nd.rchild; } }
      fprintf(stderr, "stream %d chan %d %dx%d: tree %d nodes uses_wp=%d max_prop=%d props 0x%x predictors 0x%x leaves %d (mul/offset != 1/0: %d) clusters reachable %d\n", stream_id, ci, w, h, tree_count, tf.uses_wp, tf.max_prop, props_used, preds_used, leaves, nonunit, nreach);
    }
#endif
    // properties 16 ..: four per earlier channel of the stream with this channel's size and shifts, nearest first (libjxl: PrecomputeReferences); properties
    // beyond the channels there are read as zero.  This is their serial statement; on the GPU the block-form wave loop evaluates them too (dev_modular_wave.h: kBig — the one-ballot loops pass such trees on to it)
    int nref = 0, nref_props = 0;
    const int32_t *refp[kModMaxRefs];
    if (tf.max_prop > 15) {
      nref_props = (tf.max_prop - 16) / 4 + 1;
      if (nref_props > kModMaxRefs) return kErrUnsupportedTransform;
      for (int j = ci - 1; j >= 0 && nref < nref_props; j--)
        if (chans[j].w == w && chans[j].h == h && chans[j].hs == c.hs && chans[j].vs == c.vs) refp[nref++] = chans[j].d;
      for (int k = 16 + 4 * nref; k < 16 + 4 * nref_props; k++) props[k] = 0;
    }
    const bool wide = w > kModMaxW;
    if (wide && tf.uses_wp && (!S.wide_wp || w > kWideMaxW)) return kErrUnsupportedTransform;
    props[0] = ci;
    const bool wide32 = S.st.wide32 != 0;
#ifdef JXL_EMUL_TRACE
    long trace_clu[256]; for (int i = 0; i < 256; i++) trace_clu[i] = 0;
#endif
    bool wide_overflow = false;
    WPState wst;
    const WpRows WR = wp_rows(S, wide && tf.uses_wp, w);
    if (tf.uses_wp) {
      for (int i = 0; i < 2 * (w + 2); i++) { WR.te[i] = 0; for (int k = 0; k < 4; k++) WR.pe[k][i] = 0; }
    }
    for (int y = 0; y < h; y++) {
      int32_t *out = c.d + (size_t)y * (size_t)w;
      int32_t *row = wide ? out : S.rows[y % 3];
      const int32_t *rN = wide ? out - w : S.rows[(y + 2) % 3];
      const int32_t *rNN = wide ? out - 2 * w : S.rows[(y + 1) % 3];
      int64_t prev_prop9 = 0;
      props[2] = y;
      for (int x = 0; x < w; x++) {
        int64_t W = x > 0 ? row[x - 1] : (y > 0 ? rN[x] : 0);
        int64_t N = y > 0 ? rN[x] : W;
        int64_t NW = (x > 0 && y > 0) ? rN[x - 1] : W;
        int64_t NE = (x + 1 < w && y > 0) ? rN[x + 1] : N;
        int64_t NN = y > 1 ? rNN[x] : N;
        int64_t NEE = (x + 2 < w && y > 0) ? rN[x + 2] : NE;
        int64_t WW = x > 1 ? row[x - 2] : W;
        props[3] = x;
        props[4] = (int32_t)iabs64(N);
        props[5] = (int32_t)iabs64(W);
        props[6] = (int32_t)N;
        props[7] = (int32_t)W;
        props[8] = (int32_t)(W - prev_prop9);
        props[9] = (int32_t)(W + N - NW);
        if (wide32 && (W + N - NW) != (int64_t)props[9]) wide_overflow = true;
        prev_prop9 = props[9];
        props[10] = (int32_t)(W - NW);
        props[11] = (int32_t)(NW - N);
        props[12] = (int32_t)(N - NE);
        props[13] = (int32_t)(N - NN);
        props[14] = (int32_t)(W - WW);
        int64_t wp_pred = 0;
        if (tf.uses_wp) { int32_t me; wp_pred = wp_predict(S, WR, wst, wp, x, y, w, N, W, NE, NW, NN, me); props[15] = me; }
        else props[15] = 0;
        for (int r = 0; r < nref; r++) {
          const int32_t *rp = refp[r] + (size_t)y * (size_t)w;
          const int64_t v = rp[x], vl = x ? rp[x - 1] : 0, vt = y ? rp[x - w] : vl, vtl = (x && y) ? rp[x - w - 1] : vl;
          const int64_t lo = vl < vt ? vl : vt, hi = vl < vt ? vt : vl, grad = vl + vt - vtl;
          const int64_t vp = vtl > hi ? lo : vtl < lo ? hi : grad;      // the clamped gradient of the reference channel at this position
          props[16 + 4 * r] = (int32_t)iabs64(v); props[17 + 4 * r] = (int32_t)v;
          props[18 + 4 * r] = (int32_t)iabs64(v - vp); props[19 + 4 * r] = (int32_t)(v - vp);
        }
        const DevTreeNode *nd = S.tree_ncache > 0 ? &S.tree[0] : &gtree[0];
        while (nd->prop >= 0) {
          int idx = props[nd->prop] > nd->splitval ? nd->lchild : nd->rchild;
          nd = idx < S.tree_ncache ? &S.tree[idx] : &gtree[idx];
        }
        int64_t guess = predict_plain(nd->lchild, W, N, NW, NE, NN, WW, NEE, wp_pred);
#ifdef JXL_EMUL_TRACE
        trace_clu[ev.ctx_map[nd->splitval]]++;
#endif
        uint32_t u = ev.lz77 ? ec_read_lz(ev, b, state, (uint32_t)nd->splitval, S.lz) : ec_read(ev, b, state, (uint32_t)nd->splitval);
        int64_t val = (int64_t)unpack_signed(u) * (int64_t)(uint32_t)nd->rchild + nd->offset + guess;
        row[x] = (int32_t)val;
        if (!wide) out[x] = (int32_t)val;
        if (tf.uses_wp) wp_update(WR, wst, val, x, y, w);
      }
    }
#ifdef JXL_EMUL_TRACE
    {                                     // samples per cluster of this channel: how many clusters carry 99 % / 99.9 % of them (what an LDS cache of hot clusters would have to hold)
      long tot = 0; int used = 0; long v[256]; for (int i = 0; i < 256; i++) { v[i] = trace_clu[i]; tot += v[i]; used += v[i] != 0; }
      for (int i = 0; i < 256; i++) for (int j = i + 1; j < 256; j++) if (v[j] > v[i]) { long t = v[i]; v[i] = v[j]; v[j] = t; }
      long acc = 0; int n99 = 0, n999 = 0; for (int i = 0; i < 256 && tot; i++) { acc += v[i]; if (!n99 && acc * 100 >= tot * 99) n99 = i + 1; if (!n999 && acc * 1000 >= tot * 999) { n999 = i + 1; break; } }
      fprintf(stderr, "  stream %d chan %d: %ld samples over %d clusters (log_alpha %d); 99 %% in %d, 99.9 %% in %d; top: %ld %ld %ld %ld\n", stream_id, ci, tot, used, ev.log_alpha, n99, n999, v[0], v[1], v[2], v[3]);
    }
#endif
    if (wide_overflow) return kErrUnsupportedTransform;
  }
  return 0;
}

// GroupHeader of a modular stream (H.2): use_global_tree, WP header, transforms.
JXL_DEV void modular_read_header(DevBits &b, DevWP &wp, int &nb_transforms, int &use_global) {
  use_global = (int)bits_read(b, 1);
  wp.p1 = 16; wp.p2 = 10; wp.p3a = 7; wp.p3b = 7; wp.p3c = 7; wp.p3d = 0; wp.p3e = 0;
  wp.w[0] = 13; wp.w[1] = 12; wp.w[2] = 12; wp.w[3] = 12;
  if (!bits_read(b, 1)) {
    wp.p1 = (int)bits_read(b, 5); wp.p2 = (int)bits_read(b, 5);
    wp.p3a = (int)bits_read(b, 5); wp.p3b = (int)bits_read(b, 5); wp.p3c = (int)bits_read(b, 5);
    wp.p3d = (int)bits_read(b, 5); wp.p3e = (int)bits_read(b, 5);
    for (int i = 0; i < 4; i++) wp.w[i] = (int)bits_read(b, 4);
  }
  nb_transforms = (int)bits_u32(b, -1, 0, -1, 1, 4, 2, 8, 18);
}

// One modular stream in three phases so that the whole wave can stage the stream's tables in LDS:
//   begin (lane 0): GroupHeader, (global | local) MA tree + leaf code;  stage (all lanes): tree head, context map,
//   hybrid-uint configs and alias tables -> LDS;  decode (lane 0): channels + final-state check.
// Transform list of a stream header (H.6): RCT and palette are parsed, squeeze parameters skipped (the caller decides what it accepts).
JXL_DEV uint32_t modular_read_transforms(DevBits &b, int ntr, DevTrList *out) {
  if (ntr > 0 && (!out || ntr > kModMaxLocalTr)) return kErrUnsupportedTransform;
  if (out) out->n = ntr;
  for (int i = 0; i < ntr; i++) {
    DevTr &t = out->t[i];
    t.id = (int)bits_read(b, 2);
    t.begin_c = t.rct_type = t.num_c = t.nb_colours = t.nb_deltas = t.d_pred = 0;
    if (t.id == 0) {
      t.begin_c = (int)bits_u32(b, 3, 0, 6, 8, 10, 72, 13, 1096);
      t.rct_type = (int)bits_u32(b, -1, 6, 2, 0, 4, 2, 6, 10);
      if (t.rct_type >= 42) return kErrBitstream;
    } else if (t.id == 1) {
      t.begin_c = (int)bits_u32(b, 3, 0, 6, 8, 10, 72, 13, 1096);
      t.num_c = (int)bits_u32(b, -1, 1, -1, 3, -1, 4, 13, 1);
      t.nb_colours = (int)bits_u32(b, 8, 0, 10, 256, 12, 1280, 16, 5376);
      t.nb_deltas = (int)bits_u32(b, -1, 0, 8, 1, 10, 257, 16, 1281);
      t.d_pred = (int)bits_read(b, 4);
    } else if (t.id == 2) {               // squeeze: the steps are resolved on the host (GlobalModular); here they are only skipped
      const int num_sq = (int)bits_u32(b, -1, 0, 4, 1, 6, 9, 8, 41);
      for (int q = 0; q < num_sq; q++) {
        (void)bits_read(b, 2);
        (void)bits_u32(b, 3, 0, 6, 8, 10, 72, 13, 1096);
        (void)bits_u32(b, -1, 1, -1, 2, -1, 3, 4, 4);
      }
    }
    else return kErrBitstream;
  }
  return 0;
}

JXL_DEV void modular_stream_begin(const uint8_t *tables, const DevFrame &F, LocalTreeScratch &L, DevModScratch &S, DevTrList *trs = nullptr) {
  DevModStream &st = S.st;
  int ntr, use_global;
  modular_read_header(st.b, st.wp, ntr, use_global);
  st.err = 0; st.m16 = F.modular_16bit;
  st.wide32 = (F.is_modular ? (F.mod_exp_bits && F.mod_bits == 32) : (F.mod_alpha_exp_bits && F.mod_alpha_bits == 32)) ? 1 : 0;
  if (trs) trs->n = 0;
  { uint32_t e = modular_read_transforms(st.b, ntr, trs); if (e) { st.err = e; return; } }
  if (use_global) {
    if (F.tree_count <= 0) { st.err = kErrBitstream; return; }
    st.tree = (const DevTreeNode *)(tables + F.tree_off); st.count = F.tree_count; st.ev = ec_view(tables, F.tree_ec);
    st.num_ctx = F.tree_ec.num_ctx + (F.tree_ec.lz77 ? 1 : 0); st.num_clusters = F.tree_ec.num_clusters;      // the map carries one more entry (distances) with LZ77
  } else {
    static_assert(sizeof(LocalTmp) <= (size_t)kModPoolMin, "the header parser's working arrays fit the table pool");
    uint32_t e = d_read_local_tree(st.b, L, (LocalTmp *)S.pool);      // the pool is free until modular_stream_stage fills it
    if (e) { st.err = e; return; }
    st.tree = L.nodes; st.count = L.count; st.ev = local_view(L.leaf_code);
    st.num_ctx = L.leaf_code.num_ctx; st.num_clusters = L.leaf_code.num_clusters;
  }
#ifdef JXL_EMUL_TRACE
  if (getenv("JXLEMUL_TREE")) {
    fprintf(stderr, "stream tree: %d nodes (global %d), %d ctx, %d clusters, log_alpha %d, prefix %d; wp p1 %d p2 %d p3 %d %d %d %d %d w %d %d %d %d\n", st.count, use_global, st.num_ctx,
            st.num_clusters, st.ev.log_alpha, st.ev.use_prefix, st.wp.p1, st.wp.p2, st.wp.p3a, st.wp.p3b, st.wp.p3c, st.wp.p3d, st.wp.p3e, st.wp.w[0], st.wp.w[1], st.wp.w[2], st.wp.w[3]);
    for (int i = 0; i < st.count && i < 300; i++) {
      const DevTreeNode &t = st.tree[i];
      if (t.prop < 0) fprintf(stderr, "  %3d leaf ctx %d (cluster %d) pred %d mul %d off %d\n", i, t.splitval, st.ev.ctx_map[t.splitval], t.lchild, t.rchild, t.offset);
      else fprintf(stderr, "  %3d prop %d > %d ? %d : %d\n", i, t.prop, t.splitval, t.lchild, t.rchild);
    }
    for (int i = 0; i < st.num_clusters; i++) fprintf(stderr, "  cfg[%d] split_exp %d msb %d lsb %d\n", i, st.ev.cfg[i] & 255, (st.ev.cfg[i] >> 8) & 255, (st.ev.cfg[i] >> 16) & 255);
  }
#endif
}

JXL_DEV void modular_stream_stage(DevModScratch &S, int tid, int nthreads) {
  DevModStream &st = S.st;
  if (st.err) return;
#ifdef JXL_EMUL_TRACE
  if (tid == 0 && getenv("JXLEMUL_STATS")) fprintf(stderr, "stream: tree %d nodes, %d ctx, %d clusters, log_alpha %d, prefix %d\n", st.count, st.num_ctx, st.num_clusters, st.ev.log_alpha, st.ev.use_prefix);
#endif
  // carve the pool (every work-item computes the same layout; the stores of the descriptors are benign duplicates)
  uint8_t *pool = (uint8_t *)S.pool;
  int used = 0;
  const int alias_bytes = st.ev.use_prefix ? 0 : (int)((st.num_clusters << st.ev.log_alpha) * sizeof(DevAlias));
  const bool alias_lds = !st.ev.use_prefix && st.num_clusters <= kLocMaxClusters && alias_bytes <= S.pool_bytes;
  DevAlias *l_alias = (DevAlias *)pool;
  if (alias_lds) used = alias_bytes;
  const int ctx_bytes = (st.num_ctx + 7) & ~7;
  const bool ctx_lds = used + ctx_bytes <= S.pool_bytes;
  uint8_t *l_ctx = pool + used;
  if (ctx_lds) used += ctx_bytes;
  DevTreeNode *l_tree = (DevTreeNode *)(pool + used);
  int ncache = (S.pool_bytes - used) / (int)sizeof(DevTreeNode);
  if (ncache > st.count) ncache = st.count;
  S.alias = alias_lds ? l_alias : st.ev.alias; S.alias_lds = alias_lds;
  S.ctx_map = ctx_lds ? l_ctx : st.ev.ctx_map; S.ctx_lds = ctx_lds; S.ctx_off = (int32_t)(l_ctx - pool);
  S.tree = l_tree; S.tree_ncache = ncache; S.pool_used = used;
  for (int i = tid; i < ncache; i += nthreads) l_tree[i] = st.tree[i];
  for (int i = tid; i < 64; i += nthreads) S.divlut[i] = (1u << 24) / (uint32_t)(i + 1);
  if (ctx_lds) for (int i = tid; i < st.num_ctx; i += nthreads) l_ctx[i] = st.ev.ctx_map[i];
  for (int i = tid; i < st.num_clusters && i < kLocMaxClusters; i += nthreads) S.cfg[i] = st.ev.cfg[i];
  if (alias_lds) {
    const int n = st.num_clusters << st.ev.log_alpha;
    for (int i = tid; i < n; i += nthreads) l_alias[i] = st.ev.alias[i];
  }
}

JXL_DEV uint32_t modular_stream_decode(DevModScratch &S, const DevChanOut *chans, int nch, int stream_id) {
  DevModStream &st = S.st;
  if (st.err) return st.err;
  DevECView ev = st.ev;
  ev.ctx_map = S.ctx_map;
  if (st.num_clusters <= kLocMaxClusters) ev.cfg = S.cfg;
  if (!ev.use_prefix) ev.alias = S.alias;
  DevBits b = st.b;
  uint32_t state = ans_init(ev, b);
  if (ev.lz77) {
    if (!S.lz.win) return kErrLz77;                      // no window behind this stream (LZ77 outside Modular-encoded frames)
    S.lz.ncopy = 0; S.lz.pos = 0; S.lz.ndec = 0; S.lz.err = 0; S.lz.dist_mult = 0;
    for (int i = 0; i < nch; i++) if ((uint32_t)chans[i].w > S.lz.dist_mult) S.lz.dist_mult = (uint32_t)chans[i].w;
  }
  uint32_t err = modular_decode_channels(ev, b, state, st.tree, st.count, st.wp, S, chans, nch, stream_id);
  if (ev.lz77 && S.lz.err) err |= kErrBitstream;
  if (!err && state != 0x130000u) err |= kErrAnsFinal;
  st.b = b;
  return err;
}

}  // namespace jxlamd
