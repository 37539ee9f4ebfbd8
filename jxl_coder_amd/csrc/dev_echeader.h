// jxl_coder_amd/csrc/dev_echeader.h — device-side parser for entropy-code HEADERS and MA trees that live INSIDE
// group sections (ISO/IEC 18181-1 Annex C.2 / D.3 / H.4.2): libjxl's streaming encoder (used for frames of
// 2048x2048 and up, i.e. the 4K benchmark frame) gives every LfGroup stream its own MA tree and histograms
// instead of a global one, and the second stream's header sits behind the first stream's data — so its bit
// position is only known on the device.  Lane 0 of the section's wavefront parses: LZ77 flag (must be off),
// context map (simple / entropy-coded + MTF), prefix (Brotli-style) or ANS histograms, alias tables, and the
// tree itself, into per-section scratch memory in HBM.
#pragma once
#include "dev_entropy.h"
#include "dev_tables.h"

namespace jxlamd {

constexpr int kLocMaxClusters = 64;       // clusters whose configurations (and alias tables) the streams stage in LDS; codes with more read them from HBM
constexpr int kLeafMaxClusters = 256;     // a context map's entries are bytes: no code has more
constexpr int kLocMaxCtx = 4096;
constexpr int kLocMaxNodes = 2 * kLocMaxCtx;
constexpr int kLocPool = 1 << 16;

// Working arrays of the header parser.  They live in HBM next to the tables they produce, NOT in per-lane private
// (scratch) memory: kernels of several decoder contexts run side by side and the runtime's scratch backing proved
// unreliable under that concurrency on this stack (tables came out corrupted), so the entropy kernels use no scratch.
struct LocalTmp {
  uint8_t logc[258]; uint16_t same[258]; int16_t cnt[258];
  uint16_t cut[256], under[512], over[512]; uint8_t right[256]; uint16_t off[256];
  uint32_t offs[17], cloffs[17]; int32_t sym[4]; uint8_t cll[18]; uint16_t clsorted[18];
  uint8_t mtf[256]; uint16_t counts[kLeafMaxClusters]; uint16_t D[256];
  DevPrefix clp;
};

template <int NC>
struct LocalECT {
  static constexpr int kMaxClusters = NC;
  LocalTmp tmp;
  uint8_t ctx_map[kLocMaxCtx + 8];
  uint32_t cfg[NC];
  DevAlias alias[NC * 256];
  DevPrefix prefix[NC];
  uint16_t pool[kLocPool];
  uint32_t pool_used;
  int32_t num_ctx, num_clusters, use_prefix, log_alpha;
};
using LocalEC = LocalECT<kLeafMaxClusters>;      // a leaf code: up to 256 clusters (libjxl's encoder goes past 64 with squeezed channels at effort 7)
using LocalECSmall = LocalECT<8>;                // the tree's own code (6 contexts) and the nested code of a context map (1)

struct LocalTreeScratch {       // per LF group, reused by its two streams
  LocalECSmall tree_code;       // 6 contexts
  LocalEC leaf_code;
  LocalECSmall nested;          // for entropy-coded context maps
  DevTreeNode nodes[kLocMaxNodes];
  uint8_t lens[1 << 15];        // prefix-code lengths scratch
  int32_t count;
};

template <class EC>
JXL_DEV DevECView local_view(const EC &e) {
  DevECView v;
  v.ctx_map = e.ctx_map; v.cfg = e.cfg; v.alias = e.alias; v.prefix = e.prefix; v.pool = e.pool;
  v.use_prefix = e.use_prefix; v.log_alpha = e.log_alpha;
  v.lz77 = 0; v.lz_min_symbol = 0; v.lz_min_length = 0; v.dist_ctx = 0; v.lz_len_cfg = 0;     // codes parsed on the device reject LZ77 (d_ec_read_header_t)
  return v;
}

JXL_DEV int dceil_log2(uint32_t x) { int r = 0; while ((1u << r) < x) r++; return r; }

JXL_DEV uint32_t d_read_huc(DevBits &b, int log_alpha, uint32_t &err) {
  uint32_t split = bits_read(b, dceil_log2((uint32_t)log_alpha + 1)), msb = 0, lsb = 0;
  if ((int)split != log_alpha) {
    msb = bits_read(b, dceil_log2(split + 1));
    if (msb > split) { err |= kErrBitstream; msb = split; }
    lsb = bits_read(b, dceil_log2(split - msb + 1));
    if (lsb + msb > split) { err |= kErrBitstream; lsb = 0; }
  }
  return split | (msb << 8) | (lsb << 16);
}

JXL_DEV uint32_t d_varlen_u8(DevBits &b) {
  if (!bits_read(b, 1)) return 0;
  int n = (int)bits_read(b, 3);
  if (n == 0) return 1;
  return bits_read(b, n) + (1u << n);
}

// ANS histogram (12-bit) into D[table]; returns error bits
JXL_DEV uint32_t d_read_histogram(DevBits &b, uint16_t *D, int table, LocalTmp &T) {
  for (int i = 0; i < table; i++) D[i] = 0;
  if (bits_read(b, 1)) {
    int ns = (int)bits_read(b, 1) + 1;
    uint32_t s0 = d_varlen_u8(b), s1 = 0;
    if (ns == 2) s1 = d_varlen_u8(b);
    if ((int)s0 >= table || (int)s1 >= table) return kErrBitstream;
    if (ns == 1) D[s0] = 4096;
    else { if (s0 == s1) return kErrBitstream; D[s0] = (uint16_t)bits_read(b, 12); D[s1] = (uint16_t)(4096 - D[s0]); }
    return 0;
  }
  if (bits_read(b, 1)) {
    int n = (int)d_varlen_u8(b) + 1;
    if (n > table) return kErrBitstream;
    for (int i = 0; i < n; i++) D[i] = (uint16_t)(4096 / n + (i < 4096 % n ? 1 : 0));
    return 0;
  }
  int len = 0;
  while (len < 3 && bits_read(b, 1)) len++;
  int shift = (int)((bits_read(b, len) | (1u << len)) - 1);
  if (shift > 13) return kErrBitstream;
  int n = (int)d_varlen_u8(b) + 3;
  if (n > table) return kErrBitstream;
  uint8_t *logc = T.logc; uint16_t *same = T.same; int16_t *cnt = T.cnt;
  for (int i = 0; i < n; i++) { same[i] = 0; cnt[i] = 0; logc[i] = 0; }
  int omit_log = -1, omit_pos = -1;
  for (int i = 0; i < n; i++) {
    uint32_t idx = bits_peek(b, 7);
    int l, v;
    uint32_t lo = idx & 15;
    const uint32_t len_tab = 0x43334343u, len_tab2 = 0x43334303u;   // nibble tables: lengths for lo = 8..15 / 0..7
    const uint64_t val_tab = 0x2986174A5986370Aull;                 // values (hex digits) for lo = 0..15
    if (lo != 1) { l = (int)(((lo < 8 ? len_tab2 : len_tab) >> (4 * (lo & 7))) & 15); v = (int)((val_tab >> (4 * lo)) & 15); }
    else if (idx & 16) { l = 5; v = 0; }
    else if (idx & 32) { l = 6; v = 11; }
    else if (idx & 64) { l = 7; v = 13; }
    else { l = 7; v = 12; }
    bits_skip(b, l);
    logc[i] = (uint8_t)v;
    if (v == 13) {
      int rle = (int)d_varlen_u8(b);
      same[i] = (uint16_t)(rle + 5);
      i += rle + 3;
      continue;
    }
    if (v > omit_log) { omit_log = v; omit_pos = i; }
  }
  if (omit_pos < 0) return kErrBitstream;
  if (omit_pos + 1 < n && logc[omit_pos + 1] == 13) return kErrBitstream;
  int prev = 0, numsame = 0, total = 0;
  for (int i = 0; i < n; i++) {
    if (same[i]) { numsame = same[i] - 1; prev = i > 0 ? cnt[i - 1] : 0; }
    if (numsame > 0) { cnt[i] = (int16_t)prev; numsame--; }
    else {
      int code = logc[i];
      if (i == omit_pos || code == 0) continue;
      if (code == 1) cnt[i] = 1;
      else {
        int lc = code - 1;
        int bc = shift - ((12 - lc) >> 1);
        if (bc > lc) bc = lc;
        if (bc < 0) bc = 0;
        cnt[i] = (int16_t)((1 << lc) + (int)(bits_read(b, bc) << (lc - bc)));
      }
    }
    total += cnt[i];
  }
  int rest = 4096 - total;
  if (rest <= 0) return kErrBitstream;
  cnt[omit_pos] = (int16_t)rest;
  for (int i = 0; i < n; i++) D[i] = (uint16_t)cnt[i];
  return 0;
}

JXL_DEV void d_build_alias(const uint16_t *D, int log_alpha, DevAlias *a, LocalTmp &T) {
  const int table = 1 << log_alpha, bucket = 4096 >> log_alpha;
  // (every entry is put together in registers and leaves as ONE 8-byte store: the tables lie in HBM, and a field stored there and read back — the right symbol's
  // frequency used to be looked up through a[i].right — costs the serial lane a store-to-load round trip per entry, 0.25 ms per cluster as measured in round 6)
  int n = table;
  while (n > 0 && D[n - 1] == 0) n--;
  // a symbol of frequency 4096 is the histogram's only one (the frequencies sum to 4096), hence its last
  if (n > 0 && D[n - 1] == 4096) {
    const int s = n - 1;
    for (int i = 0; i < table; i++) { DevAlias e; e.cutoff = 0; e.right = (uint8_t)s; e.off1 = (uint16_t)(bucket * i); e.freq0 = 4096; e.freq1 = 4096; a[i] = e; }
    return;
  }
  uint16_t *cut = T.cut, *under = T.under, *over = T.over; uint8_t *right = T.right; uint16_t *off = T.off;
  // libjxl's InitAliasTable pairs the top of a stack of overfull buckets with the top of a stack of underfull ones until one of them is empty.  The same pairs
  // in the same order with the stacks' tops in registers (the working arrays are LDS or HBM: every read is a round trip of the serial lane): the overfull bucket
  // stays `cur` while it remains overfull (the reference pushes it back and pops it again); the underfull stack is, from the top, the bucket that has just dropped
  // below full (`pend`: at most one, taken next), the empty buckets table - 1 .. n (implicit: nothing to read) and the underfull buckets below n (`under`).
  int nr = 0, no = 0;
  for (int i = 0; i < table; i++) { right[i] = 0; off[i] = 0; }
  for (int i = 0; i < n; i++) { const uint16_t c = D[i]; cut[i] = c; if (c > bucket) over[no++] = (uint16_t)i; else if (c < bucket) under[nr++] = (uint16_t)i; }
  for (int i = n; i < table; i++) cut[i] = 0;
  int ne = table - n, cur = -1, co = 0, pend = -1, pcut = 0;
  for (;;) {
    if (cur < 0) { if (no == 0) break; cur = over[--no]; co = cut[cur]; }
    int u, cu;
    if (pend >= 0) { u = pend; cu = pcut; pend = -1; }
    else if (ne > 0) { u = n + --ne; cu = 0; }
    else if (nr > 0) { u = under[--nr]; cu = cut[u]; }
    else break;
    co -= bucket - cu;
    right[u] = (uint8_t)cur; off[u] = (uint16_t)co;
    if (co < bucket) { cut[cur] = (uint16_t)co; pend = cur; pcut = co; cur = -1; }
    else if (co == bucket) { cut[cur] = (uint16_t)co; cur = -1; }
  }
  if (cur >= 0) cut[cur] = (uint16_t)co;
  for (int i = 0; i < table; i++) {
    DevAlias e;
    const uint16_t ci = cut[i];
    if (ci == bucket) { e.right = (uint8_t)i; e.off1 = 0; e.cutoff = 0; }
    else { e.right = right[i]; e.off1 = (uint16_t)(off[i] - ci); e.cutoff = (uint8_t)ci; }
    e.freq0 = D[i];
    e.freq1 = D[e.right];
    a[i] = e;
  }
}

// canonical code from lengths -> DevPrefix (+ symbols appended to the pool)
template <class EC>
JXL_DEV uint32_t d_build_canonical(EC &ec, DevPrefix &p, const uint8_t *lens, int n) {
  for (int l = 0; l < 16; l++) p.cnt[l] = 0;
  int nz = 0, last = -1;
  for (int i = 0; i < n; i++) if (lens[i]) { p.cnt[lens[i]]++; nz++; last = i; }
  if (ec.pool_used + (uint32_t)nz > (uint32_t)kLocPool) return kErrUnsupportedTransform;
  p.sorted_off = ec.pool_used;
  uint32_t *offs = ec.tmp.offs; offs[1] = 0;
  for (int l = 1; l < 16; l++) offs[l + 1] = offs[l] + p.cnt[l];
  for (int i = 0; i < n; i++) if (lens[i]) ec.pool[p.sorted_off + offs[lens[i]]++] = (uint16_t)i;
  ec.pool_used += (uint32_t)nz;
  p.single = nz == 1 ? last : nz == 0 ? 0 : -1;
  return 0;
}

JXL_DEV int d_prefix_decode_raw(const DevPrefix &p, const uint16_t *pool, DevBits &b) {
  if (p.single >= 0) return p.single;
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)bits_read(b, 1);
    int count = p.cnt[len];
    if (code - first < count) return pool[p.sorted_off + (uint32_t)(index + code - first)];
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return 0;
}

template <class EC>
JXL_DEV uint32_t d_read_prefix_code(DevBits &b, EC &ec, DevPrefix &p, uint8_t *lens, int alphabet) {
  for (int i = 0; i < alphabet; i++) lens[i] = 0;
  if (alphabet == 1) { uint32_t e = d_build_canonical(ec, p, lens, 1); p.single = 0; return e; }
  int hskip = (int)bits_read(b, 2);
  if (hskip == 1) {
    int max_bits = 0;
    for (int t = alphabet - 1; t; t >>= 1) max_bits++;
    int ns = (int)bits_read(b, 2) + 1;
    int32_t *sym = ec.tmp.sym; sym[0] = sym[1] = sym[2] = sym[3] = 0;
    for (int i = 0; i < ns; i++) { sym[i] = (int)bits_read(b, max_bits); if (sym[i] >= alphabet) return kErrBitstream; }
    for (int i = 0; i < ns; i++) for (int j = i + 1; j < ns; j++) if (sym[i] == sym[j]) return kErrBitstream;
    if (ns == 1) { uint32_t e = d_build_canonical(ec, p, lens, alphabet); p.single = sym[0]; return e; }
    if (ns == 2) { lens[sym[0]] = 1; lens[sym[1]] = 1; }
    else if (ns == 3) { lens[sym[0]] = 1; lens[sym[1]] = 2; lens[sym[2]] = 2; }
    else if (bits_read(b, 1)) { lens[sym[0]] = 1; lens[sym[1]] = 2; lens[sym[2]] = 3; lens[sym[3]] = 3; }
    else { lens[sym[0]] = lens[sym[1]] = lens[sym[2]] = lens[sym[3]] = 2; }
    return d_build_canonical(ec, p, lens, alphabet);
  }
  const uint8_t *order = kClOrder, *cl_len = kClLen, *cl_val = kClVal;   // constant memory (dev_tables.h)
  uint8_t *cll = ec.tmp.cll;
  for (int i = 0; i < 18; i++) cll[i] = 0;
  int space = 32, num_codes = 0;
  for (int i = hskip; i < 18 && space > 0; i++) {
    uint32_t pk = bits_peek(b, 4);
    bits_skip(b, cl_len[pk]);
    int v = cl_val[pk];
    cll[order[i]] = (uint8_t)v;
    if (v) { space -= 32 >> v; num_codes++; }
  }
  if (!(num_codes == 1 || space == 0)) return kErrBitstream;
  // code-length code: tiny canonical decoder kept local
  DevPrefix &clp = ec.tmp.clp; uint16_t *clsorted = ec.tmp.clsorted;
  {
    for (int l = 0; l < 16; l++) clp.cnt[l] = 0;
    int nz = 0, last = -1;
    for (int i = 0; i < 18; i++) if (cll[i]) { clp.cnt[cll[i]]++; nz++; last = i; }
    uint32_t *offs = ec.tmp.cloffs; offs[1] = 0;
    for (int l = 1; l < 16; l++) offs[l + 1] = offs[l] + clp.cnt[l];
    for (int i = 0; i < 18; i++) if (cll[i]) clsorted[offs[cll[i]]++] = (uint16_t)i;
    clp.sorted_off = 0; clp.single = nz == 1 ? last : nz == 0 ? 0 : -1;
  }
  int symbol = 0, prev_len = 8, repeat = 0, repeat_len = 0, sp = 32768;
  while (symbol < alphabet && sp > 0) {
    int cl = d_prefix_decode_raw(clp, clsorted, b);
    if (cl < 16) {
      repeat = 0;
      lens[symbol++] = (uint8_t)cl;
      if (cl) { prev_len = cl; sp -= 32768 >> cl; }
    } else {
      int extra = cl - 14;
      int new_len = cl == 16 ? prev_len : 0;
      if (repeat_len != new_len) { repeat = 0; repeat_len = new_len; }
      int old = repeat;
      if (repeat > 0) { repeat -= 2; repeat <<= extra; }
      repeat += (int)bits_read(b, extra) + 3;
      int delta = repeat - old;
      if (symbol + delta > alphabet) return kErrBitstream;
      for (int i = 0; i < delta; i++) lens[symbol++] = (uint8_t)repeat_len;
      if (repeat_len) sp -= delta << (15 - repeat_len);
    }
  }
  if (sp != 0) return kErrBitstream;
  return d_build_canonical(ec, p, lens, alphabet);
}

// The nested code of an entropy-coded context map has a single context, so it never carries a context map itself:
// two template instances instead of recursion keep everything inlined (no call stack, no scratch).
template <bool kNested, class EC>
JXL_DEV uint32_t d_ec_read_header_t(DevBits &b, int num_ctx, EC &ec, LocalECSmall *nested, uint8_t *lens, LocalTmp *fast_tmp = nullptr);

JXL_DEV uint32_t d_read_ctx_map(DevBits &b, uint8_t *map, int n, int &num_clusters, LocalECSmall *nested, uint8_t *lens) {
  uint32_t err = 0;
  if (bits_read(b, 1)) {
    int nb = (int)bits_read(b, 2);
    for (int i = 0; i < n; i++) map[i] = (uint8_t)bits_read(b, nb);
  } else {
    int use_mtf = (int)bits_read(b, 1);
    if (!nested) return kErrBitstream;
    err |= d_ec_read_header_t<true>(b, 1, *nested, nullptr, lens);
    if (err) return err;
    DevECView v = local_view(*nested);
    uint32_t state = ans_init(v, b);
    for (int i = 0; i < n; i++) {
      uint32_t val = ec_read(v, b, state, 0);
      if (val > 255) return kErrBitstream;
      map[i] = (uint8_t)val;
    }
    if (state != 0x130000u) return kErrAnsFinal;
    if (use_mtf) {
      uint8_t *mtf = nested->tmp.mtf;
      for (int i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
      for (int i = 0; i < n; i++) {
        uint8_t idx = map[i], val = mtf[idx];
        map[i] = val;
        for (; idx; idx--) mtf[idx] = mtf[idx - 1];
        mtf[0] = val;
      }
    }
  }
  int mx = 0;
  for (int i = 0; i < n; i++) if (map[i] > mx) mx = map[i];
  num_clusters = mx + 1;
  return err;
}

template <bool kNested, class EC>
JXL_DEV uint32_t d_ec_read_header_t(DevBits &b, int num_ctx, EC &ec, LocalECSmall *nested, uint8_t *lens, LocalTmp *fast_tmp) {
  uint32_t err = 0;
  if (kNested && num_ctx != 1) return kErrBitstream;
  if (num_ctx > kLocMaxCtx) return kErrUnsupportedTransform;
  ec.num_ctx = num_ctx; ec.pool_used = 0;
  if (bits_read(b, 1)) return kErrLz77;          // LZ77 inside a group-level code: not on the device
  ec.num_clusters = 1;
  for (int i = 0; i < num_ctx; i++) ec.ctx_map[i] = 0;
  if (!kNested && num_ctx > 1) { err |= d_read_ctx_map(b, ec.ctx_map, num_ctx, ec.num_clusters, nested, lens); if (err) return err; }
  if (ec.num_clusters > num_ctx || ec.num_clusters > EC::kMaxClusters) return kErrBitstream;      // more clusters than contexts: libjxl refuses that
  ec.use_prefix = (int)bits_read(b, 1);
  ec.log_alpha = ec.use_prefix ? 15 : 5 + (int)bits_read(b, 2);
  for (int i = 0; i < ec.num_clusters; i++) ec.cfg[i] = d_read_huc(b, ec.log_alpha, err);
  if (ec.use_prefix) {
    uint16_t *counts = ec.tmp.counts;
    for (int i = 0; i < ec.num_clusters; i++) {
      if (!bits_read(b, 1)) counts[i] = 1;
      else { int nb = (int)bits_read(b, 4); uint32_t c = 1 + (1u << nb) + bits_read(b, nb); if (c > (1u << 15)) return kErrBitstream; counts[i] = (uint16_t)c; }
    }
    for (int i = 0; i < ec.num_clusters; i++) { err |= d_read_prefix_code(b, ec, ec.prefix[i], lens, counts[i]); if (err) return err; }
  } else {
    // histogram + alias-table working arrays: the caller's LDS scratch when it has one (the serial lane's dependent accesses then
    // cost an LDS round trip instead of an L2 one — the pairing loop of an alias table is ~230 such steps per cluster), else HBM
    const int table = 1 << ec.log_alpha;
    LocalTmp &T = fast_tmp ? *fast_tmp : ec.tmp;
    uint16_t *D = T.D;
    for (int i = 0; i < ec.num_clusters; i++) {
      err |= d_read_histogram(b, D, table, T);
      if (err) return err;
      d_build_alias(D, ec.log_alpha, ec.alias + (size_t)i * (size_t)table, T);
    }
  }
  return err;
}

// MA tree (H.4.2) + its leaf code
JXL_DEV uint32_t d_read_local_tree(DevBits &b, LocalTreeScratch &L, LocalTmp *fast_tmp = nullptr) {
  uint32_t err = d_ec_read_header_t<false>(b, 6, L.tree_code, &L.nested, L.lens, fast_tmp);
  if (err) return err;
  DevECView v = local_view(L.tree_code);
  uint32_t state = ans_init(v, b);
  int to_decode = 1, leaf = 0, count = 0;
  while (to_decode > 0) {
    to_decode--;
    if (count >= kLocMaxNodes) return kErrUnsupportedTransform;
    DevTreeNode &nd = L.nodes[count];
    int prop = (int)ec_read(v, b, state, 1) - 1;
    if (prop < 0) {
      nd.prop = -1;
      nd.lchild = (int32_t)ec_read(v, b, state, 2);                     // predictor
      nd.offset = unpack_signed(ec_read(v, b, state, 3));
      uint32_t mul_log = ec_read(v, b, state, 4), mul_bits = ec_read(v, b, state, 5);
      if (nd.lchild > 13 || mul_log > 30) return kErrBitstream;
      nd.rchild = (int32_t)((mul_bits + 1u) << mul_log);                // multiplier
      nd.splitval = leaf++;                                             // context
      count++;
      continue;
    }
    nd.prop = prop;
    nd.splitval = unpack_signed(ec_read(v, b, state, 0));
    nd.lchild = count + to_decode + 1;
    nd.rchild = count + to_decode + 2;
    nd.offset = 0;
    count++;
    to_decode += 2;
    if (b.consumed > (1ull << 34)) return kErrBitstream;
  }
  if (state != 0x130000u) return kErrAnsFinal;
  L.count = count;
  return d_ec_read_header_t<false>(b, leaf, L.leaf_code, &L.nested, L.lens, fast_tmp);
}

}  // namespace jxlamd
