/* host_bits.cpp — PRODUCT host code: bit reader, field codes, ANS / prefix histogram + context-map reader (ISO/IEC 18181-1 Annex C/D)
 * used by host_parse.cpp to read headers, the embedded ICC stream, the TOC and the LfGlobal / HfGlobal sections into the frame-tables
 * blob.  It decodes no pixel data.  Nothing here includes or links anything under oracle/.  Note on independence: this file and the
 * oracle's oracle/jxo_entropy.c were written together from the same clauses and differ in little but the prefix, so the C oracle is NOT an
 * independent check of header / histogram / context-map parsing — the reference binary's golden outputs (tests/golden) are. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "host_bits.h"

void hx_br_init(hx_br *br, const uint8_t *p, size_t len) { br->p = p; br->len = len; br->pos = 0; br->err = 0; }

uint32_t hx_bits(hx_br *br, int n) {
  if (n == 0) return 0;
  size_t byte = br->pos >> 3;
  int sh = (int)(br->pos & 7);
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) {
    size_t b = byte + (size_t)i;
    if (b < br->len) v |= (uint64_t)br->p[b] << (8 * i);
  }
  if (br->pos + (size_t)n > br->len * 8) br->err = 1;
  br->pos += (size_t)n;
  v >>= sh;
  return (uint32_t)(n == 32 ? v : (v & ((1ull << n) - 1)));
}

void hx_align(hx_br *br) { br->pos = (br->pos + 7) & ~(size_t)7; }

uint32_t hx_u32(hx_br *br, int b0, uint32_t o0, int b1, uint32_t o1, int b2, uint32_t o2, int b3, uint32_t o3) {
  uint32_t sel = hx_bits(br, 2);
  int b = sel == 0 ? b0 : sel == 1 ? b1 : sel == 2 ? b2 : b3;
  uint32_t o = sel == 0 ? o0 : sel == 1 ? o1 : sel == 2 ? o2 : o3;
  if (b < 0) return o;
  return hx_bits(br, b) + o;
}

uint64_t hx_u64(hx_br *br) {
  uint32_t sel = hx_bits(br, 2);
  if (sel == 0) return 0;
  if (sel == 1) return 1 + hx_bits(br, 4);
  if (sel == 2) return 17 + hx_bits(br, 8);
  uint64_t v = hx_bits(br, 12);
  int shift = 12;
  while (hx_bits(br, 1)) {
    if (shift == 60) { v |= (uint64_t)hx_bits(br, 4) << shift; break; }
    v |= (uint64_t)hx_bits(br, 8) << shift;
    shift += 8;
  }
  return v;
}

float hx_f16(hx_br *br) {
  uint32_t h = hx_bits(br, 16);
  uint32_t sign = h >> 15, exp = (h >> 10) & 31, mant = h & 1023;
  float v;
  if (exp == 0) v = (float)mant * (1.0f / 16777216.0f);           /* subnormal: mant * 2^-24 */
  else {
    union { uint32_t u; float f; } c;
    c.u = ((exp + 112) << 23) | (mant << 13);
    v = c.f;
  }
  return sign ? -v : v;
}

uint32_t hx_enum(hx_br *br) { return hx_u32(br, -1, 0, -1, 1, 4, 2, 6, 18); }

/* ------------------------------------------------------------------------------------------------ */
static int ceil_log2(uint32_t x) { int r = 0; while ((1u << r) < x) r++; return r; }   /* x>=1 */

static void read_huc(hx_br *br, hx_huc *c, int log_alpha) {
  c->split_exp = (uint8_t)hx_bits(br, ceil_log2((uint32_t)log_alpha + 1));
  c->msb = c->lsb = 0;
  if (c->split_exp != log_alpha) {
    c->msb = (uint8_t)hx_bits(br, ceil_log2((uint32_t)c->split_exp + 1));
    if (c->msb > c->split_exp) { br->err = 1; c->msb = c->split_exp; }
    c->lsb = (uint8_t)hx_bits(br, ceil_log2((uint32_t)(c->split_exp - c->msb) + 1));
    if (c->lsb + c->msb > c->split_exp) { br->err = 1; c->lsb = 0; }
  }
}

static uint32_t read_varlen_u8(hx_br *br) {
  if (!hx_bits(br, 1)) return 0;
  int n = (int)hx_bits(br, 3);
  if (n == 0) return 1;
  return hx_bits(br, n) + (1u << n);
}

/* ANS histogram, 12-bit precision */
static int read_ans_histogram(hx_br *br, uint16_t *D, int table_size) {
  memset(D, 0, sizeof(uint16_t) * (size_t)table_size);
  if (hx_bits(br, 1)) {                         /* simple */
    int ns = (int)hx_bits(br, 1) + 1;
    uint32_t s0 = read_varlen_u8(br), s1 = 0;
    if (ns == 2) s1 = read_varlen_u8(br);
    if ((int)s0 >= table_size || (int)s1 >= table_size) return -1;
    if (ns == 1) D[s0] = 4096;
    else {
      if (s0 == s1) return -1;
      D[s0] = (uint16_t)hx_bits(br, 12);
      D[s1] = (uint16_t)(4096 - D[s0]);
    }
    return 0;
  }
  if (hx_bits(br, 1)) {                         /* flat */
    int n = (int)read_varlen_u8(br) + 1;
    if (n > table_size) return -1;
    for (int i = 0; i < n; i++) D[i] = (uint16_t)(4096 / n + (i < 4096 % n ? 1 : 0));
    return 0;
  }
  int len = 0;
  while (len < 3 && hx_bits(br, 1)) len++;
  int shift = (int)((hx_bits(br, len) | (1u << len)) - 1);
  if (shift > 13) return -1;
  int n = (int)read_varlen_u8(br) + 3;
  if (n > table_size) return -1;
  /* prefix code over log-counts: (length, value) by 7 peeked bits; see 18181-1 C.2.? */
  int logc[258], same[258];
  memset(same, 0, sizeof(same));
  int omit_log = -1, omit_pos = -1;
  for (int i = 0; i < n; i++) {
    /* peek 7 bits */
    hx_br save = *br;
    uint32_t idx = hx_bits(br, 7);
    *br = save;
    int l, v;
    uint32_t lo = idx & 15;
    static const uint8_t len16[16] = {3, 0, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4};
    static const uint8_t val16[16] = {10, 0, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2};
    if (lo != 1) { l = len16[lo]; v = val16[lo]; }
    else if (idx & 16) { l = 5; v = 0; }
    else if (idx & 32) { l = 6; v = 11; }
    else if (idx & 64) { l = 7; v = 13; }
    else { l = 7; v = 12; }
    (void)hx_bits(br, l);
    logc[i] = v;
    if (v == 13) {
      int rle = (int)read_varlen_u8(br);
      same[i] = rle + 5;
      i += rle + 3;
      continue;
    }
    if (v > omit_log) { omit_log = v; omit_pos = i; }
  }
  if (omit_pos < 0) return -1;
  if (omit_pos + 1 < n && logc[omit_pos + 1] == 13) return -1;
  int prev = 0, numsame = 0, total = 0;
  int cnt[258];
  memset(cnt, 0, sizeof(cnt));
  for (int i = 0; i < n; i++) {
    if (same[i]) { numsame = same[i] - 1; prev = i > 0 ? cnt[i - 1] : 0; }
    if (numsame > 0) { cnt[i] = prev; numsame--; }
    else {
      int code = logc[i];
      if (i == omit_pos || code == 0) continue;
      if (code == 1) cnt[i] = 1;
      else {
        int lc = code - 1;
        int bc = shift - ((12 - lc) >> 1);
        if (bc > lc) bc = lc;
        if (bc < 0) bc = 0;
        cnt[i] = (1 << lc) + (int)(hx_bits(br, bc) << (lc - bc));
      }
    }
    total += cnt[i];
  }
  cnt[omit_pos] = 4096 - total;
  if (cnt[omit_pos] <= 0) return -1;
  for (int i = 0; i < n; i++) D[i] = (uint16_t)cnt[i];
  return 0;
}

static void build_alias(hx_cluster *c, int log_alpha) {
  int table = 1 << log_alpha;
  int bucket = 4096 >> log_alpha;
  c->a_sym = (uint8_t *)calloc((size_t)table, 1);
  c->a_cutoff = (uint16_t *)calloc((size_t)table, 2);
  c->a_off = (uint32_t *)calloc((size_t)table, 4);
  for (int s = 0; s < table; s++)
    if (c->D[s] == 4096) {
      for (int i = 0; i < table; i++) { c->a_sym[i] = (uint8_t)s; c->a_cutoff[i] = 0; c->a_off[i] = (uint32_t)(bucket * i); }
      return;
    }
  int n = table;
  while (n > 0 && c->D[n - 1] == 0) n--;
  uint32_t *cut = (uint32_t *)calloc((size_t)table, 4);
  int *under = (int *)malloc(sizeof(int) * (size_t)table * 2), *over = (int *)malloc(sizeof(int) * (size_t)table * 2);
  int nu = 0, no = 0;
  for (int i = 0; i < n; i++) {
    cut[i] = c->D[i];
    if ((int)cut[i] > bucket) over[no++] = i; else if ((int)cut[i] < bucket) under[nu++] = i;
  }
  for (int i = n; i < table; i++) { cut[i] = 0; under[nu++] = i; }
  while (no > 0) {
    int o = over[--no];
    int u = under[--nu];
    uint32_t by = (uint32_t)bucket - cut[u];
    cut[o] -= by;
    c->a_sym[u] = (uint8_t)o;
    c->a_off[u] = cut[o];
    if ((int)cut[o] < bucket) under[nu++] = o; else if ((int)cut[o] > bucket) over[no++] = o;
  }
  for (int i = 0; i < table; i++) {
    if ((int)cut[i] == bucket) { c->a_sym[i] = (uint8_t)i; c->a_off[i] = 0; c->a_cutoff[i] = 0; }
    else { c->a_off[i] -= cut[i]; c->a_cutoff[i] = (uint16_t)cut[i]; }
  }
  free(cut); free(under); free(over);
}

/* ---- prefix codes (Brotli-style, RFC 7932 §3.4/3.5) */
static int build_canonical(hx_cluster *c, const uint8_t *lens, int n) {
  memset(c->cnt, 0, sizeof(c->cnt));
  int nz = 0, last = -1;
  for (int i = 0; i < n; i++) if (lens[i]) { c->cnt[lens[i]]++; nz++; last = i; }
  c->sorted = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(nz > 0 ? nz : 1));
  int offs[17]; offs[1] = 0;
  for (int l = 1; l < 16; l++) offs[l + 1] = offs[l] + c->cnt[l];
  for (int i = 0; i < n; i++) if (lens[i]) c->sorted[offs[lens[i]]++] = (uint16_t)i;
  c->single = -1;
  if (nz == 1) c->single = last;
  if (nz == 0) c->single = 0;
  return 0;
}

static int prefix_decode(const hx_cluster *c, hx_br *br) {
  if (c->single >= 0) return c->single;
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)hx_bits(br, 1);
    int count = c->cnt[len];
    if (code - first < count) return c->sorted[index + (code - first)];
    index += count; first += count; first <<= 1; code <<= 1;
  }
  br->err = 1;
  return 0;
}

static int read_prefix_code(hx_br *br, hx_cluster *c, int alphabet) {
  uint8_t *lens = (uint8_t *)calloc((size_t)alphabet + 1, 1);
  int rc = 0;
  if (alphabet == 1) { lens[0] = 0; build_canonical(c, lens, 1); c->single = 0; free(lens); return 0; }
  int hskip = (int)hx_bits(br, 2);
  if (hskip == 1) {                              /* simple code */
    int max_bits = 0;
    for (int t = alphabet - 1; t; t >>= 1) max_bits++;
    int ns = (int)hx_bits(br, 2) + 1;
    int sym[4];
    for (int i = 0; i < ns; i++) { sym[i] = (int)hx_bits(br, max_bits); if (sym[i] >= alphabet) rc = -1; }
    for (int i = 0; i < ns && !rc; i++) for (int j = i + 1; j < ns; j++) if (sym[i] == sym[j]) rc = -1;
    if (!rc) {
      if (ns == 1) { build_canonical(c, lens, alphabet); c->single = sym[0]; free(lens); return 0; }
      if (ns == 2) { lens[sym[0]] = 1; lens[sym[1]] = 1; }
      else if (ns == 3) { lens[sym[0]] = 1; lens[sym[1]] = 2; lens[sym[2]] = 2; }
      else {
        if (hx_bits(br, 1)) { lens[sym[0]] = 1; lens[sym[1]] = 2; lens[sym[2]] = 3; lens[sym[3]] = 3; }
        else { lens[sym[0]] = lens[sym[1]] = lens[sym[2]] = lens[sym[3]] = 2; }
      }
      build_canonical(c, lens, alphabet);
    }
    free(lens);
    return rc;
  }
  /* complex code: code-length code */
  static const uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t cl_len[16] = {2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4};
  static const uint8_t cl_val[16] = {0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2, 0, 4, 3, 5};
  uint8_t cll[18];
  memset(cll, 0, sizeof(cll));
  int space = 32, num_codes = 0;
  for (int i = hskip; i < 18 && space > 0; i++) {
    hx_br save = *br;
    uint32_t p = hx_bits(br, 4);
    *br = save;
    (void)hx_bits(br, cl_len[p]);
    int v = cl_val[p];
    cll[order[i]] = (uint8_t)v;
    if (v) { space -= 32 >> v; num_codes++; }
  }
  if (!(num_codes == 1 || space == 0)) { free(lens); return -1; }
  hx_cluster clc;
  memset(&clc, 0, sizeof(clc));
  build_canonical(&clc, cll, 18);
  int symbol = 0, prev_len = 8, repeat = 0, repeat_len = 0;
  int sp = 32768;
  while (symbol < alphabet && sp > 0) {
    int cl = prefix_decode(&clc, br);
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
      repeat += (int)hx_bits(br, extra) + 3;
      int delta = repeat - old;
      if (symbol + delta > alphabet) { rc = -1; break; }
      for (int i = 0; i < delta; i++) lens[symbol++] = (uint8_t)repeat_len;
      if (repeat_len) sp -= delta << (15 - repeat_len);
    }
    if (br->err) { rc = -1; break; }
  }
  if (!rc && sp != 0) rc = -1;
  free(clc.sorted);
  if (!rc) build_canonical(c, lens, alphabet);
  free(lens);
  return rc;
}

/* ---- context map */
int hx__read_ctx_map(hx_br *br, uint8_t *map, int n, int *num_clusters) {
  if (hx_bits(br, 1)) {                         /* simple */
    int b = (int)hx_bits(br, 2);
    for (int i = 0; i < n; i++) map[i] = (uint8_t)hx_bits(br, b);
  } else {
    int use_mtf = (int)hx_bits(br, 1);
    hx_ec ec;
    if (hx_ec_read_header(&ec, br, 1)) return -1;
    hx_ec_begin(&ec, br, 0);
    for (int i = 0; i < n; i++) {
      uint32_t v = hx_ec_read(&ec, br, 0);
      if (v > 255) { hx_ec_free(&ec); return -1; }
      map[i] = (uint8_t)v;
    }
    int ok = hx_ec_final_ok(&ec);
    hx_ec_free(&ec);
    if (!ok) return -1;
    if (use_mtf) {
      uint8_t mtf[256];
      for (int i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
      for (int i = 0; i < n; i++) {
        uint8_t idx = map[i], v = mtf[idx];
        map[i] = v;
        for (; idx; idx--) mtf[idx] = mtf[idx - 1];
        mtf[0] = v;
      }
    }
  }
  int mx = 0;
  for (int i = 0; i < n; i++) if (map[i] > mx) mx = map[i];
  *num_clusters = mx + 1;
  /* all cluster ids below max must be used */
  uint8_t seen[256];
  memset(seen, 0, sizeof(seen));
  for (int i = 0; i < n; i++) seen[map[i]] = 1;
  for (int i = 0; i <= mx; i++) if (!seen[i]) return -1;
  return br->err ? -1 : 0;
}

int hx_ec_read_header(hx_ec *ec, hx_br *br, int num_ctx) {
  memset(ec, 0, sizeof(*ec));
  ec->num_ctx = num_ctx;
  ec->lz77 = (int)hx_bits(br, 1);
  int n = num_ctx;
  if (ec->lz77) {
    ec->lz_min_symbol = (int)hx_u32(br, -1, 224, -1, 512, -1, 4096, 15, 8);
    ec->lz_min_length = (int)hx_u32(br, -1, 3, -1, 4, 2, 5, 8, 9);
    read_huc(br, &ec->lz_len_cfg, 8);
    n++;
  }
  ec->ctx_map = (uint8_t *)calloc((size_t)n, 1);
  ec->num_clusters = 1;
  if (n > 1 && hx__read_ctx_map(br, ec->ctx_map, n, &ec->num_clusters)) return -1;
  ec->use_prefix = (int)hx_bits(br, 1);
  ec->log_alpha = ec->use_prefix ? 15 : 5 + (int)hx_bits(br, 2);
  ec->cfg = (hx_huc *)calloc((size_t)ec->num_clusters, sizeof(hx_huc));
  ec->cl = (hx_cluster *)calloc((size_t)ec->num_clusters, sizeof(hx_cluster));
  for (int i = 0; i < ec->num_clusters; i++) read_huc(br, &ec->cfg[i], ec->log_alpha);
  if (ec->use_prefix) {
    int *counts = (int *)malloc(sizeof(int) * (size_t)ec->num_clusters);
    for (int i = 0; i < ec->num_clusters; i++) {
      if (!hx_bits(br, 1)) counts[i] = 1;
      else { int nb = (int)hx_bits(br, 4); counts[i] = 1 + (1 << nb) + (int)hx_bits(br, nb); }
      if (counts[i] > (1 << 15)) { free(counts); return -1; }
    }
    for (int i = 0; i < ec->num_clusters; i++) {
      ec->cl[i].nsym = counts[i];
      if (read_prefix_code(br, &ec->cl[i], counts[i])) { free(counts); return -1; }
    }
    free(counts);
  } else {
    int table = 1 << ec->log_alpha;
    for (int i = 0; i < ec->num_clusters; i++) {
      ec->cl[i].D = (uint16_t *)calloc((size_t)table, 2);
      if (read_ans_histogram(br, ec->cl[i].D, table)) return -1;
      build_alias(&ec->cl[i], ec->log_alpha);
    }
  }
  if (ec->lz77) ec->window = (uint32_t *)calloc(1u << 20, 4);
  return br->err ? -1 : 0;
}

void hx_ec_begin(hx_ec *ec, hx_br *br, uint32_t dist_mult) {
  ec->dist_mult = dist_mult;
  ec->num_to_copy = ec->copy_pos = ec->num_decoded = 0;
  ec->state = ec->use_prefix ? 0x130000u : hx_bits(br, 32);
}

int hx_ec_final_ok(const hx_ec *ec) { return ec->state == 0x130000u; }

static inline uint32_t read_token(hx_ec *ec, hx_br *br, int cluster) {
  hx_cluster *c = &ec->cl[cluster];
  if (ec->use_prefix) return (uint32_t)prefix_decode(c, br);
  int log_bucket = 12 - ec->log_alpha;
  uint32_t res = ec->state & 0xfff;
  uint32_t i = res >> log_bucket, pos = res & ((1u << log_bucket) - 1);
  uint32_t sym, off;
  if (pos >= c->a_cutoff[i]) { sym = c->a_sym[i]; off = c->a_off[i] + pos; }
  else { sym = i; off = pos; }
  ec->state = c->D[sym] * (ec->state >> 12) + off;
  if (ec->state < (1u << 16)) ec->state = (ec->state << 16) | hx_bits(br, 16);
  return sym;
}

static inline uint32_t read_hybrid(hx_br *br, const hx_huc *c, uint32_t token) {
  uint32_t split = 1u << c->split_exp;
  if (token < split) return token;
  uint32_t nbits = c->split_exp - (c->msb + c->lsb) + ((token - split) >> (c->msb + c->lsb));
  if (nbits > 31) { br->err = 1; return 0; }
  uint32_t low = token & ((1u << c->lsb) - 1);
  token >>= c->lsb;
  uint32_t bits = hx_bits(br, (int)nbits);
  return (((((1u << c->msb) | (token & ((1u << c->msb) - 1))) << nbits) | bits) << c->lsb) | low;
}

static const int8_t kSpecialDist[120][2] = {
    {0, 1},  {1, 0},  {1, 1},  {-1, 1}, {0, 2},  {2, 0},  {1, 2},  {-1, 2}, {2, 1},  {-2, 1}, {2, 2},  {-2, 2},
    {0, 3},  {3, 0},  {1, 3},  {-1, 3}, {3, 1},  {-3, 1}, {2, 3},  {-2, 3}, {3, 2},  {-3, 2}, {0, 4},  {4, 0},
    {1, 4},  {-1, 4}, {4, 1},  {-4, 1}, {3, 3},  {-3, 3}, {2, 4},  {-2, 4}, {4, 2},  {-4, 2}, {0, 5},  {3, 4},
    {-3, 4}, {4, 3},  {-4, 3}, {5, 0},  {1, 5},  {-1, 5}, {5, 1},  {-5, 1}, {2, 5},  {-2, 5}, {5, 2},  {-5, 2},
    {4, 4},  {-4, 4}, {3, 5},  {-3, 5}, {5, 3},  {-5, 3}, {0, 6},  {6, 0},  {1, 6},  {-1, 6}, {6, 1},  {-6, 1},
    {2, 6},  {-2, 6}, {6, 2},  {-6, 2}, {4, 5},  {-4, 5}, {5, 4},  {-5, 4}, {3, 6},  {-3, 6}, {6, 3},  {-6, 3},
    {0, 7},  {7, 0},  {1, 7},  {-1, 7}, {5, 5},  {-5, 5}, {7, 1},  {-7, 1}, {4, 6},  {-4, 6}, {6, 4},  {-6, 4},
    {2, 7},  {-2, 7}, {7, 2},  {-7, 2}, {3, 7},  {-3, 7}, {7, 3},  {-7, 3}, {5, 6},  {-5, 6}, {6, 5},  {-6, 5},
    {8, 0},  {4, 7},  {-4, 7}, {7, 4},  {-7, 4}, {8, 1},  {8, 2},  {6, 6},  {-6, 6}, {8, 3},  {5, 7},  {-5, 7},
    {7, 5},  {-7, 5}, {8, 4},  {6, 7},  {-6, 7}, {7, 6},  {-7, 6}, {8, 5},  {7, 7},  {-7, 7}, {8, 6},  {8, 7}};

uint32_t hx_ec_read(hx_ec *ec, hx_br *br, int ctx) {
  const uint32_t mask = (1u << 20) - 1;
  if (ec->num_to_copy > 0) {
    uint32_t r = ec->window[(ec->copy_pos++) & mask];
    ec->num_to_copy--;
    ec->window[(ec->num_decoded++) & mask] = r;
    return r;
  }
  int cluster = ec->ctx_map[ctx];
  uint32_t token = read_token(ec, br, cluster);
  if (ec->lz77 && token >= (uint32_t)ec->lz_min_symbol) {
    ec->num_to_copy = read_hybrid(br, &ec->lz_len_cfg, token - (uint32_t)ec->lz_min_symbol) + (uint32_t)ec->lz_min_length;
    int dc = ec->ctx_map[ec->num_ctx];
    uint32_t dtok = read_token(ec, br, dc);
    uint32_t distance = read_hybrid(br, &ec->cfg[dc], dtok);
    uint32_t nspecial = ec->dist_mult ? 120 : 0;
    if (distance < nspecial) {
      int d = (int)ec->dist_mult * kSpecialDist[distance][1] + kSpecialDist[distance][0];
      distance = d < 1 ? 1u : (uint32_t)d;
    } else distance = distance + 1 - nspecial;
    if (distance > ec->num_decoded) distance = ec->num_decoded;
    if (distance > (1u << 20)) distance = 1u << 20;
    ec->copy_pos = ec->num_decoded - distance;
    if (distance == 0) {
      uint32_t n = ec->num_to_copy < (1u << 20) ? ec->num_to_copy : (1u << 20);
      memset(ec->window, 0, n * 4);
    }
    if (ec->num_to_copy < (uint32_t)ec->lz_min_length || br->err) { br->err = 1; ec->num_to_copy = 0; return 0; }
    return hx_ec_read(ec, br, ctx);
  }
  uint32_t r = read_hybrid(br, &ec->cfg[cluster], token);
  if (ec->lz77) ec->window[(ec->num_decoded++) & mask] = r;
  return r;
}

void hx_ec_free(hx_ec *ec) {
  if (ec->cl)
    for (int i = 0; i < ec->num_clusters; i++) {
      free(ec->cl[i].D); free(ec->cl[i].a_sym); free(ec->cl[i].a_cutoff); free(ec->cl[i].a_off); free(ec->cl[i].sorted);
    }
  free(ec->cl); free(ec->cfg); free(ec->ctx_map); free(ec->window);
  memset(ec, 0, sizeof(*ec));
}

int hx_read_permutation(hx_ec *ec, hx_br *br, uint32_t *out, uint32_t size, uint32_t skip) {
  uint32_t *lehmer = (uint32_t *)calloc(size ? size : 1, 4);
  #define PCTX(v) ({ uint32_t _v = (v); int _b = 0; while (_v) { _b++; _v >>= 1; } _b > 7 ? 7 : _b; })
  uint32_t end = hx_ec_read(ec, br, PCTX(size));
  if (end > size - skip) { free(lehmer); return -1; }
  uint32_t last = 0;
  for (uint32_t i = skip; i < end + skip; i++) {
    lehmer[i] = hx_ec_read(ec, br, PCTX(last));
    last = lehmer[i];
    if (lehmer[i] >= size - i) { free(lehmer); return -1; }
  }
  #undef PCTX
  /* decode Lehmer code: out[i] = lehmer[i]-th unused element */
  uint32_t *tmp = (uint32_t *)malloc(4 * (size_t)(size ? size : 1));
  for (uint32_t i = 0; i < size; i++) tmp[i] = i;
  uint32_t remaining = size;
  for (uint32_t i = 0; i < size; i++) {
    uint32_t idx = lehmer[i];
    out[i] = tmp[idx];
    memmove(tmp + idx, tmp + idx + 1, 4 * (size_t)(remaining - idx - 1));
    remaining--;
  }
  free(tmp); free(lehmer);
  return br->err ? -1 : 0;
}

#include <stdarg.h>
static thread_local char g_hx_err[512];
void hx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_hx_err, sizeof(g_hx_err), fmt, ap); va_end(ap); }
const char *hx_last_error(void) { return g_hx_err; }
