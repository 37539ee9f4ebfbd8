/* oracle/jxo_modular.c — Modular sub-bitstream decoder: MA tree, predictors, weighted (self-correcting)
 * predictor, RCT / palette inverse transforms (ISO/IEC 18181-1 Annex H).  CPU restatement, checker only
 * (see jxo.h).  Integer-exact: the lossless path must be bit-exact against the reference's libjxl. */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jxo_int.h"

static char g_err[512];
void jxo_set_error(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char *jxo_last_error(void) { return g_err; }

void jxo_modimg_init(jxo_modimg *m) { memset(m, 0, sizeof(*m)); }
int jxo_modimg_add(jxo_modimg *m, int w, int h, int hshift, int vshift) {
  if (m->nch == m->cap) { m->cap = m->cap ? m->cap * 2 : 8; m->ch = (jxo_chan *)realloc(m->ch, sizeof(jxo_chan) * (size_t)m->cap); }
  jxo_chan *c = &m->ch[m->nch];
  c->w = w; c->h = h; c->hshift = hshift; c->vshift = vshift;
  c->d = (int32_t *)calloc((size_t)w * (size_t)h + 1, 4);
  return m->nch++;
}
static void modimg_insert(jxo_modimg *m, int at, int w, int h, int hshift, int vshift) {
  jxo_modimg_add(m, w, h, hshift, vshift);
  jxo_chan last = m->ch[m->nch - 1];
  memmove(&m->ch[at + 1], &m->ch[at], sizeof(jxo_chan) * (size_t)(m->nch - 1 - at));
  m->ch[at] = last;
}
static void modimg_erase(jxo_modimg *m, int at, int n) {
  for (int i = 0; i < n; i++) free(m->ch[at + i].d);
  memmove(&m->ch[at], &m->ch[at + n], sizeof(jxo_chan) * (size_t)(m->nch - at - n));
  m->nch -= n;
}
void jxo_modimg_free(jxo_modimg *m) {
  for (int i = 0; i < m->nch; i++) free(m->ch[i].d);
  free(m->ch);
  memset(m, 0, sizeof(*m));
}

/* ------------------------------------------------------------------ MA tree */
int jxo_tree_read(jxo_tree *t, jxo_br *br) {
  memset(t, 0, sizeof(*t));
  jxo_ec ec;
  if (jxo_ec_read_header(&ec, br, 6)) JXO_FAIL("tree: bad entropy header");
  jxo_ec_begin(&ec, br, 0);
  int cap = 64, to_decode = 1, leaf = 0;
  t->n = (jxo_tnode *)malloc(sizeof(jxo_tnode) * (size_t)cap);
  while (to_decode > 0) {
    to_decode--;
    if (t->count == cap) { cap *= 2; t->n = (jxo_tnode *)realloc(t->n, sizeof(jxo_tnode) * (size_t)cap); }
    if (t->count > (1 << 22) || br->err) { jxo_ec_free(&ec); JXO_FAIL("tree: too large / truncated"); }
    jxo_tnode *nd = &t->n[t->count];
    int prop = (int)jxo_ec_read(&ec, br, 1) - 1;
    if (prop < 0) {
      nd->prop = -1;
      nd->predictor = (int)jxo_ec_read(&ec, br, 2);
      nd->offset = jxo_unpack_signed(jxo_ec_read(&ec, br, 3));
      uint32_t mul_log = jxo_ec_read(&ec, br, 4);
      uint32_t mul_bits = jxo_ec_read(&ec, br, 5);
      if (nd->predictor > 13 || mul_log > 30) { jxo_ec_free(&ec); JXO_FAIL("tree: bad leaf"); }
      nd->multiplier = (mul_bits + 1u) << mul_log;
      nd->ctx = leaf++;
      nd->lchild = nd->rchild = -1;
      t->count++;
      continue;
    }
    nd->prop = prop;
    nd->splitval = jxo_unpack_signed(jxo_ec_read(&ec, br, 0));
    nd->lchild = t->count + to_decode + 1;
    nd->rchild = t->count + to_decode + 2;
    nd->predictor = 0; nd->offset = 0; nd->multiplier = 1; nd->ctx = -1;
    t->count++;
    to_decode += 2;
  }
  int ok = jxo_ec_final_ok(&ec);
  jxo_ec_free(&ec);
  if (!ok) JXO_FAIL("tree: ANS final state mismatch");
  t->num_leaves = leaf;
  if (jxo_ec_read_header(&t->code, br, leaf)) JXO_FAIL("tree: bad symbol-code header");
  t->valid = 1;
  return 0;
}
void jxo_tree_free(jxo_tree *t) {
  if (t->valid) jxo_ec_free(&t->code);
  free(t->n);
  memset(t, 0, sizeof(*t));
}

/* ------------------------------------------------------------------ weighted predictor */
typedef struct {
  int64_t prediction[4];
  int64_t pred;
  uint32_t *pred_errors[4];
  int32_t *error;
  jxo_wp_header h;
  uint32_t divlookup[64];
  int xsize;
} wp_state;

static void wp_init(wp_state *s, const jxo_wp_header *h, int xsize) {
  memset(s, 0, sizeof(*s));
  s->h = *h; s->xsize = xsize;
  for (int i = 0; i < 4; i++) s->pred_errors[i] = (uint32_t *)calloc((size_t)(xsize + 2) * 2, 4);
  s->error = (int32_t *)calloc((size_t)(xsize + 2) * 2, 4);
  for (int i = 0; i < 64; i++) s->divlookup[i] = (1u << 24) / (uint32_t)(i + 1);
}
static void wp_free(wp_state *s) { for (int i = 0; i < 4; i++) free(s->pred_errors[i]); free(s->error); }
static inline int floor_log2_u64(uint64_t x) { int r = 0; while (x >>= 1) r++; return r; }
static inline uint32_t wp_error_weight(const wp_state *s, uint64_t x, uint32_t maxweight) {
  int shift = floor_log2_u64(x + 1) - 5;
  if (shift < 0) shift = 0;
  return 4 + ((maxweight * s->divlookup[x >> shift]) >> shift);
}
static int64_t wp_predict(wp_state *s, int x, int y, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN, int32_t *max_err) {
  int xs = s->xsize;
  size_t cur_row = (y & 1) ? 0 : (size_t)(xs + 2);
  size_t prev_row = (y & 1) ? (size_t)(xs + 2) : 0;
  size_t pos_N = prev_row + (size_t)x;
  size_t pos_NE = x < xs - 1 ? pos_N + 1 : pos_N;
  size_t pos_NW = x > 0 ? pos_N - 1 : pos_N;
  uint32_t w[4];
  for (int i = 0; i < 4; i++) {
    uint32_t e = s->pred_errors[i][pos_N] + s->pred_errors[i][pos_NE] + s->pred_errors[i][pos_NW];
    w[i] = wp_error_weight(s, e, (uint32_t)s->h.w[i]);
  }
  N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
  int64_t teW = x == 0 ? 0 : s->error[cur_row + (size_t)x - 1];
  int64_t teN = s->error[pos_N], teNW = s->error[pos_NW], teNE = s->error[pos_NE];
  int64_t sumWN = teN + teW;
  {
    int64_t p = teW;
    if (llabs(teN) > llabs(p)) p = teN;
    if (llabs(teNW) > llabs(p)) p = teNW;
    if (llabs(teNE) > llabs(p)) p = teNE;
    *max_err = (int32_t)p;
  }
  s->prediction[0] = W + NE - N;
  s->prediction[1] = N - (((sumWN + teNE) * s->h.p1) >> 5);
  s->prediction[2] = W - (((sumWN + teNW) * s->h.p2) >> 5);
  s->prediction[3] = N - ((teNW * s->h.p3a + teN * s->h.p3b + teNE * s->h.p3c + (NN - N) * s->h.p3d + (NW - W) * s->h.p3e) >> 5);
  uint32_t wsum = w[0] + w[1] + w[2] + w[3];
  int lw = floor_log2_u64(wsum);
  wsum = 0;
  for (int i = 0; i < 4; i++) { w[i] >>= lw - 4; wsum += w[i]; }
  int64_t sum = (int64_t)(wsum >> 1) - 1;
  for (int i = 0; i < 4; i++) sum += s->prediction[i] * (int64_t)w[i];
  s->pred = (sum * (int64_t)s->divlookup[wsum - 1]) >> 24;
  if (((teN ^ teW) | (teN ^ teNW)) > 0) return (s->pred + 3) >> 3;
  int64_t mx = W > NE ? W : NE; if (N > mx) mx = N;
  int64_t mn = W < NE ? W : NE; if (N < mn) mn = N;
  if (s->pred > mx) s->pred = mx;
  if (s->pred < mn) s->pred = mn;
  return (s->pred + 3) >> 3;
}
static void wp_update(wp_state *s, int64_t val, int x, int y) {
  int xs = s->xsize;
  size_t cur_row = (y & 1) ? 0 : (size_t)(xs + 2);
  size_t prev_row = (y & 1) ? (size_t)(xs + 2) : 0;
  val *= 8;
  s->error[cur_row + (size_t)x] = (int32_t)(s->pred - val);
  for (int i = 0; i < 4; i++) {
    int64_t err = (llabs(s->prediction[i] - val) + 3) >> 3;
    s->pred_errors[i][cur_row + (size_t)x] = (uint32_t)err;
    s->pred_errors[i][prev_row + (size_t)x + 1] += (uint32_t)err;
  }
}

/* ------------------------------------------------------------------ predictors */
static inline int64_t clamped_gradient(int64_t n, int64_t w, int64_t l) {
  int64_t m = n < w ? n : w, M = n < w ? w : n;
  int64_t g = n + w - l;
  return g < m ? m : g > M ? M : g;
}

static int64_t predict_plain(int predictor, int64_t W, int64_t N, int64_t NW, int64_t NE, int64_t NN, int64_t WW, int64_t NEE, int64_t wp) {
  switch (predictor) {
    case 0: return 0;
    case 1: return W;
    case 2: return N;
    case 3: return (W + N) / 2;
    case 4: { int64_t p = W + N - NW; int64_t pa = llabs(p - W), pb = llabs(p - N); return pa < pb ? W : N; }
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

static int decode_channel(jxo_br *br, jxo_ec *ec, const jxo_tree *tree, const jxo_wp_header *wph, jxo_modimg *img,
                          int chan, int stream_id) {
  jxo_chan *c = &img->ch[chan];
  int w = c->w, h = c->h;
  /* which properties does the tree use / does it need the weighted predictor? */
  int max_prop = 15, uses_wp = 0;
  for (int i = 0; i < tree->count; i++) {
    if (tree->n[i].prop > max_prop) max_prop = tree->n[i].prop;
    if (tree->n[i].prop == 15) uses_wp = 1;
    if (tree->n[i].prop < 0 && tree->n[i].predictor == 6) uses_wp = 1;
  }
  { extern int jxo_debug; if (jxo_debug > 1 && chan == 0) fprintf(stderr, "dbg tree: stream %d nodes %d max_prop %d uses_wp %d w %d h %d\n", stream_id, tree->count, max_prop, uses_wp, w, h); }
  int nprops = max_prop + 1;
  int32_t *props = (int32_t *)calloc((size_t)nprops + 4, 4);
  /* reference channels for props >= 16 */
  int nref = (nprops - 16 + 3) / 4;
  int refs[64]; int nrefs = 0;
  for (int j = chan - 1; j >= 0 && nrefs < nref && nrefs < 64; j--) {
    jxo_chan *r = &img->ch[j];
    if (r->w != w || r->h != h || r->hshift != c->hshift || r->vshift != c->vshift) continue;
    refs[nrefs++] = j;
  }
  wp_state wp;
  if (uses_wp) wp_init(&wp, wph, w);
  props[0] = chan; props[1] = stream_id;
  for (int y = 0; y < h; y++) {
    int32_t *row = c->d + (size_t)y * (size_t)w;
    const int32_t *rN = y > 0 ? row - w : NULL, *rNN = y > 1 ? row - 2 * w : NULL;
    int64_t prev_prop9 = 0;
    props[2] = y;
    for (int x = 0; x < w; x++) {
      int64_t W = x > 0 ? row[x - 1] : (rN ? rN[x] : 0);
      int64_t N = rN ? rN[x] : W;
      int64_t NW = (x > 0 && rN) ? rN[x - 1] : W;
      int64_t NE = (x + 1 < w && rN) ? rN[x + 1] : N;
      int64_t NN = rNN ? rNN[x] : N;
      int64_t NEE = (x + 2 < w && rN) ? rN[x + 2] : NE;
      int64_t WW = x > 1 ? row[x - 2] : W;
      props[3] = x;
      props[4] = (int32_t)llabs(N);
      props[5] = (int32_t)llabs(W);
      props[6] = (int32_t)N;
      props[7] = (int32_t)W;
      props[8] = (int32_t)(W - prev_prop9);
      props[9] = (int32_t)(W + N - NW);
      prev_prop9 = props[9];
      props[10] = (int32_t)(W - NW);
      props[11] = (int32_t)(NW - N);
      props[12] = (int32_t)(N - NE);
      props[13] = (int32_t)(N - NN);
      props[14] = (int32_t)(W - WW);
      int64_t wp_pred = 0;
      if (uses_wp) { int32_t me; wp_pred = wp_predict(&wp, x, y, N, W, NE, NW, NN, &me); props[15] = me; }
      else props[15] = 0;
      for (int r = 0; r < nrefs; r++) {
        const jxo_chan *rc = &img->ch[refs[r]];
        const int32_t *rp = rc->d + (size_t)y * (size_t)w;
        int64_t v = rp[x];
        int64_t vl = x ? rp[x - 1] : 0;
        int64_t vt = y ? rp[x - w] : vl;
        int64_t vtl = (x && y) ? rp[x - w - 1] : vl;
        int64_t vp = clamped_gradient(vl, vt, vtl);
        int o = 16 + 4 * r;
        if (o < nprops) props[o] = (int32_t)llabs(v);
        if (o + 1 < nprops) props[o + 1] = (int32_t)v;
        if (o + 2 < nprops) props[o + 2] = (int32_t)llabs(v - vp);
        if (o + 3 < nprops) props[o + 3] = (int32_t)(v - vp);
      }
      const jxo_tnode *nd = &tree->n[0];
      while (nd->prop >= 0) nd = &tree->n[props[nd->prop] > nd->splitval ? nd->lchild : nd->rchild];
      int64_t guess = predict_plain(nd->predictor, W, N, NW, NE, NN, WW, NEE, wp_pred);
      uint32_t u = jxo_ec_read(ec, br, nd->ctx);
      int64_t val = (int64_t)jxo_unpack_signed(u) * (int64_t)nd->multiplier + nd->offset + guess;
      row[x] = (int32_t)val;
      if (uses_wp) wp_update(&wp, val, x, y);
    }
    if (br->err) break;
  }
  if (uses_wp) wp_free(&wp);
  free(props);
  return br->err ? -1 : 0;
}

/* ------------------------------------------------------------------ transforms */
static int meta_apply(jxo_modimg *img, const jxo_transform *t) {
  if (t->id == JXO_TR_RCT) {
    if (t->begin_c + 3 > img->nch) JXO_FAIL("rct: channel range");
    return 0;
  }
  if (t->id == JXO_TR_PALETTE) {
    int b = t->begin_c, e = t->begin_c + t->num_c - 1;
    if (e >= img->nch) JXO_FAIL("palette: channel range begin_c=%d num_c=%d nch=%d nbcol=%d nbdelta=%d pred=%d", t->begin_c, t->num_c, img->nch, t->nb_colours, t->nb_deltas, t->d_pred);
    if (b >= img->nb_meta) img->nb_meta += 1; else img->nb_meta += 2 - t->num_c;
    int hs = img->ch[b].hshift, vs = img->ch[b].vshift;
    (void)hs; (void)vs;
    modimg_erase(img, b + 1, t->num_c - 1);
    modimg_insert(img, 0, t->nb_colours + t->nb_deltas, t->num_c, -1, -1);
    return 0;
  }
  /* Squeeze (ISO/IEC 18181-1 H.6.2; what libjxl's MetaSqueeze does under JxlDecoderProcessInput, reference call site
     jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75): every step halves channels [begin_c, begin_c + num_c) along one axis and inserts
     their residual channels right behind the range (in_place) or at the end of the list. */
  for (int q = 0; q < t->nsq; q++) {
    const jxo_squeeze_step *p = &t->sq[q];
    int b = p->begin_c, e = p->begin_c + p->num_c - 1;
    if (b < img->nb_meta || e >= img->nch) JXO_FAIL("squeeze: channel range");
    int offset = p->in_place ? e + 1 : img->nch;
    for (int c = b; c <= e; c++) {
      jxo_chan *a = &img->ch[c];
      int w = a->w, h = a->h, rw = w, rh = h;
      if (p->horizontal) { a->w = (w + 1) / 2; a->hshift++; rw = w - a->w; } else { a->h = (h + 1) / 2; a->vshift++; rh = h - a->h; }
      free(a->d); a->d = (int32_t *)calloc((size_t)a->w * (size_t)a->h + 1, sizeof(int32_t));
      int hs = a->hshift, vs = a->vshift;
      modimg_insert(img, offset + (c - b), rw, rh, hs, vs);
    }
  }
  return 0;
}

/* default squeeze sequence (num_sq == 0): chroma-like channels first, then alternate until both sides are <= 8 */
static void default_squeeze(const jxo_modimg *img, jxo_transform *t) {
  int first = img->nb_meta, nbc = img->nch - img->nb_meta;
  int w = img->ch[first].w, h = img->ch[first].h;
  t->nsq = 0;
  if (nbc > 2 && img->ch[first + 1].w == w && img->ch[first + 1].h == h) {
    jxo_squeeze_step p = {1, 0, first + 1, 2};
    t->sq[t->nsq++] = p; p.horizontal = 0; t->sq[t->nsq++] = p;
  }
  jxo_squeeze_step p = {0, 1, first, nbc};
  if (!(w > h) && h > 8) { p.horizontal = 0; t->sq[t->nsq++] = p; h = (h + 1) / 2; }
  while ((w > 8 || h > 8) && t->nsq + 2 <= 48) {
    if (w > 8) { p.horizontal = 1; t->sq[t->nsq++] = p; w = (w + 1) / 2; }
    if (h > 8) { p.horizontal = 0; t->sq[t->nsq++] = p; h = (h + 1) / 2; }
  }
}

static int64_t smooth_tendency(int64_t B, int64_t a, int64_t n) {
  int64_t diff = 0;
  if (B >= a && a >= n) {
    diff = (4 * B - 3 * n - a + 6) / 12;
    if (diff - (diff & 1) > 2 * (B - a)) diff = 2 * (B - a) + 1;
    if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
  } else if (B <= a && a <= n) {
    diff = (4 * B - 3 * n - a - 6) / 12;
    if (diff + (diff & 1) < 2 * (B - a)) diff = 2 * (B - a) - 1;
    if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
  }
  return diff;
}

static int inv_squeeze(jxo_modimg *img, const jxo_transform *t) {
  for (int q = t->nsq - 1; q >= 0; q--) {
    const jxo_squeeze_step *p = &t->sq[q];
    int b = p->begin_c, e = p->begin_c + p->num_c - 1;
    int offset = p->in_place ? e + 1 : img->nch - p->num_c;
    if (e >= img->nch || offset <= e || offset + p->num_c > img->nch) JXO_FAIL("squeeze: bookkeeping");
    for (int c = b; c <= e; c++) {
      jxo_chan *a = &img->ch[c]; const jxo_chan *r = &img->ch[offset + c - b];
      int ow = p->horizontal ? a->w + r->w : a->w, oh = p->horizontal ? a->h : a->h + r->h;
      if (p->horizontal ? (r->h != a->h || (r->w != a->w && r->w != a->w - 1)) : (r->w != a->w || (r->h != a->h && r->h != a->h - 1))) JXO_FAIL("squeeze: channel sizes");
      int32_t *out = (int32_t *)calloc((size_t)ow * (size_t)oh + 1, sizeof(int32_t));
      int lines = p->horizontal ? a->h : a->w, na = p->horizontal ? a->w : a->h, nr = p->horizontal ? r->w : r->h;
      for (int i = 0; i < lines; i++) {
        size_t as = p->horizontal ? 1 : (size_t)a->w, rs = p->horizontal ? 1 : (size_t)r->w, os = p->horizontal ? 1 : (size_t)ow;
        const int32_t *av = a->d + (p->horizontal ? (size_t)i * (size_t)a->w : (size_t)i), *rv = r->d + (p->horizontal ? (size_t)i * (size_t)r->w : (size_t)i);
        int32_t *o = out + (p->horizontal ? (size_t)i * (size_t)ow : (size_t)i);
        int64_t left = 0;
        for (int k = 0; k < nr; k++) {
          int64_t A0 = av[(size_t)k * as], nx = k + 1 < na ? av[(size_t)(k + 1) * as] : A0;
          if (k == 0) left = A0;
          int64_t diff = (int64_t)rv[(size_t)k * rs] + smooth_tendency(left, A0, nx);
          int64_t first = A0 + diff / 2, second = first - diff;
          o[(size_t)(2 * k) * os] = (int32_t)first; o[(size_t)(2 * k + 1) * os] = (int32_t)second;
          left = second;
        }
        if (na > nr) o[(size_t)(2 * nr) * os] = av[(size_t)nr * as];
      }
      free(a->d); a->d = out; a->w = ow; a->h = oh;
      if (p->horizontal) a->hshift--; else a->vshift--;
    }
    modimg_erase(img, offset, p->num_c);
  }
  return 0;
}

#include "jxo_delta_palette.h"
static int32_t palette_value(const jxo_chan *pal, int index, int c, int bit_depth) {
  int psize = pal->w;
  if (index < 0) {           /* implicit delta entries: 143 of them, +/- the rows of the table */
    if (c >= 3) return 0;
    int k = (int)((uint32_t)(-(index + 1)) % 143u);
    int32_t v = (k & 1) ? kJxoDeltaPalette[(k + 1) >> 1][c] : -kJxoDeltaPalette[(k + 1) >> 1][c];
    return bit_depth > 8 ? v * (1 << (bit_depth - 8)) : v;
  }
  if (psize <= index && index < psize + 64) {
    if (c >= 3) return 0;
    index -= psize;
    index >>= c * 2;
    return (int32_t)(((int64_t)(index % 4) * ((1 << bit_depth) - 1)) / 4) + (1 << (bit_depth - 3 > 0 ? bit_depth - 3 : 0));
  } else if (psize + 64 <= index) {
    if (c >= 3) return 0;
    index -= psize + 64;
    if (c == 1) index /= 5; else if (c == 2) index /= 25;
    return (int32_t)(((int64_t)(index % 5) * ((1 << bit_depth) - 1)) / 4);
  }
  return pal->d[(size_t)c * (size_t)pal->w + (size_t)index];
}

static int inv_palette(jxo_modimg *img, const jxo_transform *t) {
  int nb = img->ch[0].h;
  int c0 = t->begin_c + 1;
  if (c0 >= img->nch) JXO_FAIL("palette: index channel");
  int w = img->ch[c0].w, h = img->ch[c0].h;
  for (int i = 1; i < nb; i++) modimg_insert(img, c0 + 1, w, h, img->ch[c0].hshift, img->ch[c0].vshift);
  const jxo_chan *pal = &img->ch[0];
  int bit_depth = img->bitdepth < 24 ? img->bitdepth : 24;
  if (t->d_pred == 6 && t->nb_deltas) JXO_FAIL("unsupported: weighted-predictor delta palette");
  /* index plane copy (channel c0 is overwritten as colour 0) */
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)w * (size_t)h + 4);
  memcpy(idx, img->ch[c0].d, sizeof(int32_t) * (size_t)w * (size_t)h);
  for (int c = 0; c < nb; c++) {
    int32_t *out = img->ch[c0 + c].d;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int index = idx[(size_t)y * (size_t)w + (size_t)x];
        if (t->nb_deltas == 0 && t->d_pred == 0 && nb == 1) index = index < 0 ? 0 : index > pal->w - 1 ? pal->w - 1 : index;
        int64_t v = palette_value(pal, index, c, bit_depth);
        if (index < t->nb_deltas) {
          int32_t *row = out + (size_t)y * (size_t)w;
          const int32_t *rN = y > 0 ? row - w : NULL, *rNN = y > 1 ? row - 2 * w : NULL;
          int64_t W = x > 0 ? row[x - 1] : (rN ? rN[x] : 0);
          int64_t N = rN ? rN[x] : W;
          int64_t NW = (x > 0 && rN) ? rN[x - 1] : W;
          int64_t NE = (x + 1 < w && rN) ? rN[x + 1] : N;
          int64_t NN = rNN ? rNN[x] : N;
          int64_t NEE = (x + 2 < w && rN) ? rN[x + 2] : NE;
          int64_t WW = x > 1 ? row[x - 2] : W;
          v += predict_plain(t->d_pred, W, N, NW, NE, NN, WW, NEE, 0);
        }
        out[(size_t)y * (size_t)w + (size_t)x] = (int32_t)v;
      }
  }
  free(idx);
  img->nb_meta--;
  modimg_erase(img, 0, 1);
  return 0;
}

static int inv_rct(jxo_modimg *img, const jxo_transform *t) {
  int m = t->begin_c;
  if (m + 3 > img->nch) JXO_FAIL("rct: channel range");
  int perm = t->rct_type / 7, type = t->rct_type % 7;
  jxo_chan *a = &img->ch[m], *b = &img->ch[m + 1], *c = &img->ch[m + 2];
  if (a->w != b->w || a->w != c->w || a->h != b->h || a->h != c->h) JXO_FAIL("rct: channel dims differ");
  size_t n = (size_t)a->w * (size_t)a->h;
  for (size_t i = 0; i < n; i++) {
    int64_t F = a->d[i], S = b->d[i], T = c->d[i];
    if (type == 6) {
      int64_t tmp = F - (T >> 1);
      int64_t G = T + tmp;
      int64_t B = tmp - (S >> 1);
      int64_t R = B + S;
      F = R; S = G; T = B;
    } else {
      if (type & 1) T += F;
      if ((type >> 1) == 1) S += F;
      else if ((type >> 1) == 2) S += (F + T) >> 1;
    }
    a->d[i] = (int32_t)F; b->d[i] = (int32_t)S; c->d[i] = (int32_t)T;
  }
  /* permutation: decoded (a,b,c) go to channels m+perm%3, m+(perm+1+perm/3)%3, m+(perm+2-perm/3)%3 */
  int32_t *src[3] = {a->d, b->d, c->d};
  int dst[3] = {perm % 3, (perm + 1 + perm / 3) % 3, (perm + 2 - perm / 3) % 3};
  int32_t *out[3];
  for (int i = 0; i < 3; i++) out[dst[i]] = src[i];
  img->ch[m].d = out[0]; img->ch[m + 1].d = out[1]; img->ch[m + 2].d = out[2];
  return 0;
}

int jxo_modular_undo_transforms(jxo_modimg *img) {
  for (int i = img->ntr - 1; i >= 0; i--) {
    const jxo_transform *t = &img->tr[i];
    int rc = t->id == JXO_TR_RCT ? inv_rct(img, t) : t->id == JXO_TR_PALETTE ? inv_palette(img, t) : t->id == JXO_TR_SQUEEZE ? inv_squeeze(img, t) : -1;
    if (rc) return rc;
  }
  img->ntr = 0;
  return 0;
}

/* ------------------------------------------------------------------ stream */
int jxo_modular_decode(jxo_br *br, jxo_modimg *img, int stream_id, int max_chan_size, jxo_tree *global_tree,
                       int undo_transforms, int *first_undecoded) {
  if (first_undecoded) *first_undecoded = img->nch;
  if (img->nch == 0) return 0;
  int use_global = jxo_bool(br);
  jxo_wp_header wph = {16, 10, 7, 7, 7, 0, 0, {13, 12, 12, 12}};
  if (!jxo_bool(br)) {
    wph.p1 = (int)jxo_bits(br, 5); wph.p2 = (int)jxo_bits(br, 5);
    wph.p3a = (int)jxo_bits(br, 5); wph.p3b = (int)jxo_bits(br, 5); wph.p3c = (int)jxo_bits(br, 5);
    wph.p3d = (int)jxo_bits(br, 5); wph.p3e = (int)jxo_bits(br, 5);
    for (int i = 0; i < 4; i++) wph.w[i] = (int)jxo_bits(br, 4);
  }
  int ntr = (int)jxo_u32(br, -1, 0, -1, 1, 4, 2, 8, 18);
  if (ntr > 64) JXO_FAIL("too many transforms");
  img->ntr = 0;
  for (int i = 0; i < ntr; i++) {
    jxo_transform *t = &img->tr[img->ntr];
    memset(t, 0, sizeof(*t));
    t->id = (int)jxo_bits(br, 2);
    if (t->id == JXO_TR_RCT) {
      t->begin_c = (int)jxo_u32(br, 3, 0, 6, 8, 10, 72, 13, 1096);
      t->rct_type = (int)jxo_u32(br, -1, 6, 2, 0, 4, 2, 6, 10);
      if (t->rct_type >= 42) JXO_FAIL("bad rct type");
    } else if (t->id == JXO_TR_PALETTE) {
      t->begin_c = (int)jxo_u32(br, 3, 0, 6, 8, 10, 72, 13, 1096);
      t->num_c = (int)jxo_u32(br, -1, 1, -1, 3, -1, 4, 13, 1);
      t->nb_colours = (int)jxo_u32(br, 8, 0, 10, 256, 12, 1280, 16, 5376);
      t->nb_deltas = (int)jxo_u32(br, -1, 0, 8, 1, 10, 257, 16, 1281);
      t->d_pred = (int)jxo_bits(br, 4);
      if (t->d_pred > 13) JXO_FAIL("bad palette predictor");
    } else if (t->id == JXO_TR_SQUEEZE) {
      int num_sq = (int)jxo_u32(br, -1, 0, 4, 1, 6, 9, 8, 41);
      if (num_sq > 48) JXO_FAIL("unsupported: more than 48 squeeze steps");
      t->nsq = num_sq;
      for (int q = 0; q < num_sq; q++) {
        t->sq[q].horizontal = (unsigned char)jxo_bool(br); t->sq[q].in_place = (unsigned char)jxo_bool(br);
        t->sq[q].begin_c = (int)jxo_u32(br, 3, 0, 6, 8, 10, 72, 13, 1096);
        t->sq[q].num_c = (int)jxo_u32(br, -1, 1, -1, 2, -1, 3, 4, 4);
      }
      if (num_sq == 0) { if (img->nch - img->nb_meta < 1) JXO_FAIL("squeeze without channels"); default_squeeze(img, t); }
    } else JXO_FAIL("bad transform id");
    if (br->err) JXO_FAIL("truncated modular header");
    { extern int jxo_debug; if (jxo_debug > 1) fprintf(stderr, "dbg transforms: stream %d id %d begin_c %d rct %d num_c %d nbcol %d nbdeltas %d pred %d\n", stream_id, t->id, t->begin_c, t->rct_type, t->num_c, t->nb_colours, t->nb_deltas, t->d_pred); }
    if (meta_apply(img, t)) return -1;
    img->ntr++;
  }
  jxo_tree local;
  memset(&local, 0, sizeof(local));
  jxo_tree *tree = global_tree;
  if (!use_global) {
    if (jxo_tree_read(&local, br)) return -1;
    tree = &local;
  } else if (!global_tree || !global_tree->valid) JXO_FAIL("modular stream wants a global tree but none exists");
  uint32_t dist_mult = 0;
  for (int i = 0; i < img->nch; i++) {
    jxo_chan *c = &img->ch[i];
    if (i >= img->nb_meta && max_chan_size && (c->w > max_chan_size || c->h > max_chan_size)) break;
    if ((uint32_t)c->w > dist_mult) dist_mult = (uint32_t)c->w;
  }
  jxo_ec *ec = &tree->code;
  jxo_ec_begin(ec, br, dist_mult);
  int rc = 0, i;
  for (i = 0; i < img->nch; i++) {
    jxo_chan *c = &img->ch[i];
    if (!c->w || !c->h) continue;
    if (i >= img->nb_meta && max_chan_size && (c->w > max_chan_size || c->h > max_chan_size)) break;
    if (decode_channel(br, ec, tree, &wph, img, i, stream_id)) { jxo_set_error("modular: truncated channel %d (stream %d)", i, stream_id); rc = -1; break; }
  }
  if (first_undecoded) *first_undecoded = i;
  if (!rc && !jxo_ec_final_ok(ec)) { jxo_set_error("modular: ANS final state mismatch (stream %d)", stream_id); rc = -1; }
  if (!use_global) jxo_tree_free(&local);
  if (rc) return rc;
  if (undo_transforms) return jxo_modular_undo_transforms(img);
  return 0;
}
