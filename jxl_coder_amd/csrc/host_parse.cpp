// jxl_coder_amd/csrc/host_parse.cpp — see host_parse.h.
#include "host_parse.h"
#include <math.h>
#include <algorithm>
#include <vector>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include "host_bits.h"

namespace {
#define PI 3.14159265358979323846
typedef struct { int prop; int32_t splitval; int lchild, rchild; int predictor; int64_t offset; uint32_t multiplier; int ctx; } hx_tnode;
typedef struct { hx_tnode *n; int count; int num_leaves; hx_ec code; int valid; } hx_tree;
typedef struct {
  uint32_t xsize, ysize, bits_per_sample, exp_bits, num_color_channels, num_extra_channels, alpha_bits, alpha_premultiplied;
  uint32_t orientation, have_animation, xyb_encoded;
  float intensity_target;
  uint32_t want_icc, color_space, white_point, primaries, transfer_function, rendering_intent, have_gamma;
  float gamma;
} hx_info;
typedef struct { int nbands; float b[3][8]; } hx_dctparams;
#include "host_format.inc"
#include "host_icc.inc"
#include "dither_lut.h"
#include "rcp12_lut.h"
#include "upsampling_weights.h"
#include "host_icc_synth.inc"
#include "host_modular_small.inc"
}  // namespace

namespace jxlamd {

namespace {

struct Blob {
  std::vector<uint8_t> &v;
  explicit Blob(std::vector<uint8_t> &vv) : v(vv) {}
  uint32_t append(const void *p, size_t n) {
    size_t off = (v.size() + 15) & ~(size_t)15;
    v.resize(off + n);
    if (n) memcpy(v.data() + off, p, n);
    return (uint32_t)off;
  }
};

static int ceil_log2u(uint32_t x) { int r = 0; while ((1u << r) < x) r++; return r; }

// pack a parsed entropy code into the device layout
static int pack_ec(const hx_ec &ec, Blob &b, DevEC *out, std::string *err, bool allow_lz77 = false) {
  if (ec.lz77 && !allow_lz77) { *err = "unsupported: LZ77 in a group-level entropy code"; return -1; }
  memset(out, 0, sizeof(*out));
  out->num_ctx = ec.num_ctx; out->num_clusters = ec.num_clusters; out->use_prefix = ec.use_prefix; out->log_alpha = ec.log_alpha;
  out->lz77 = ec.lz77; out->lz_min_symbol = ec.lz_min_symbol; out->lz_min_length = ec.lz_min_length;
  out->lz_len_cfg = ec.lz_len_cfg.split_exp | (ec.lz_len_cfg.msb << 8) | (ec.lz_len_cfg.lsb << 16);
  out->ctx_map_off = b.append(ec.ctx_map, (size_t)ec.num_ctx + (ec.lz77 ? 1u : 0u));      // + the distance context
  std::vector<uint32_t> cfg((size_t)ec.num_clusters);
  for (int i = 0; i < ec.num_clusters; i++) cfg[(size_t)i] = ec.cfg[i].split_exp | (ec.cfg[i].msb << 8) | (ec.cfg[i].lsb << 16);
  out->cfg_off = b.append(cfg.data(), cfg.size() * 4);
  if (!ec.use_prefix) {
    const int table = 1 << ec.log_alpha;
    std::vector<DevAlias> al((size_t)ec.num_clusters * (size_t)table);
    for (int c = 0; c < ec.num_clusters; c++)
      for (int i = 0; i < table; i++) {
        DevAlias &a = al[(size_t)c * (size_t)table + (size_t)i];
        const hx_cluster &cl = ec.cl[c];
        a.cutoff = (uint8_t)cl.a_cutoff[i]; a.right = cl.a_sym[i]; a.off1 = (uint16_t)cl.a_off[i];
        a.freq0 = cl.D[i]; a.freq1 = cl.D[cl.a_sym[i]];
        // a single-symbol distribution keeps freq 4096 for every bucket (state must not change)
        if (cl.D[cl.a_sym[i]] == 4096) { a.freq0 = 4096; a.freq1 = 4096; }
      }
    out->alias_off = b.append(al.data(), al.size() * sizeof(DevAlias));
    out->prefix_off = out->pool_off = 0;
  } else {
    std::vector<DevPrefix> pf((size_t)ec.num_clusters);
    std::vector<uint16_t> pool;
    for (int c = 0; c < ec.num_clusters; c++) {
      const hx_cluster &cl = ec.cl[c];
      DevPrefix &p = pf[(size_t)c];
      memcpy(p.cnt, cl.cnt, sizeof(p.cnt));
      p.single = cl.single;
      p.sorted_off = (uint32_t)pool.size();
      int nz = 0; for (int l = 1; l < 16; l++) nz += cl.cnt[l];
      for (int i = 0; i < nz; i++) pool.push_back(cl.sorted[i]);
    }
    out->prefix_off = b.append(pf.data(), pf.size() * sizeof(DevPrefix));
    out->pool_off = b.append(pool.data(), pool.size() * 2);
    out->alias_off = 0;
  }
  return 0;
}

struct Priv {                 // state between phase 1 and phase 2
  img_meta m;
  frame_hdr f;
  DevFrame F;
  std::vector<DevSection> secs;
  int ref_w[4] = {0, 0, 0, 0}, ref_h[4] = {0, 0, 0, 0};     // reference slots as they are when this frame is decoded
  bool ref_patch_ok[4] = {false, false, false, false};       // ... and whether a slot holds a frame as it was before the colour transform (a patch source) rather than a blended canvas / nothing this decode keeps
  bool blend = false, save_canvas = false, has_src = false; int alpha_ec = -1, lf_w = 0, lf_h = 0;      // composition over a canvas (plan_parse: the frame walk)
  int64_t tree_bit = -1;        // where the global MA tree starts inside LfGlobal (-1: none): a RAW dequant matrix of HfGlobal may be coded with it
};

// the upper triangle of the symmetric 5n x 5n matrix of factor N = 2n = 2 << t (15 / 55 / 210 weights: the defaults of upsampling_weights.h or an image's own)
// expanded to one 5 x 5 kernel per output phase (the phases of the right / lower half are the mirror images of the left / upper half)
static std::vector<float> expand_upsampling_kernels(const float *w, int t) {
  const int N = 2 << t, n = N / 2;
  std::vector<float> sym((size_t)(5 * n) * (size_t)(5 * n)), k((size_t)N * N * 25);
  for (int i = 0; i < 5 * n; i++) for (int j = 0; j < 5 * n; j++) { const int y = std::min(i, j), x = std::max(i, j); sym[(size_t)j * (size_t)(5 * n) + (size_t)i] = w[5 * n * y - y * (y - 1) / 2 + x - y]; }
  for (int oy = 0; oy < N; oy++) for (int ox = 0; ox < N; ox++) for (int iy = 0; iy < 5; iy++) for (int ix = 0; ix < 5; ix++) {
    const int py = oy < n ? oy : N - 1 - oy, px = ox < n ? ox : N - 1 - ox, ty = oy < n ? iy : 4 - iy, tx = ox < n ? ix : 4 - ix;
    k[(size_t)((oy * N + ox) * 25 + iy * 5 + ix)] = sym[(size_t)(py * 5 + ty) * (size_t)(5 * n) + (size_t)(px * 5 + tx)];      // kernel[py][px][ty][tx] = sym[5 py + ty][5 px + tx]
  }
  return k;
}

static void fill_info(const img_meta &m, ImageInfo *i) {
  const hx_info &p = m.pub;
  i->xsize = p.xsize; i->ysize = p.ysize; i->bits_per_sample = p.bits_per_sample; i->exp_bits = p.exp_bits;
  i->num_color_channels = p.num_color_channels; i->num_extra_channels = p.num_extra_channels; i->alpha_bits = p.alpha_bits;
  i->alpha_premultiplied = p.alpha_premultiplied; i->orientation = 1; i->have_animation = p.have_animation;
  i->xyb_encoded = p.xyb_encoded; i->uses_original_profile = !p.xyb_encoded; i->intensity_target = p.intensity_target;
  i->want_icc = p.want_icc; i->color_space = p.color_space; i->white_point = p.white_point; i->primaries = p.primaries;
  i->transfer_function = p.transfer_function; i->rendering_intent = p.rendering_intent; i->have_gamma = p.have_gamma; i->gamma = p.gamma;
  memcpy(i->wp_xy, m.wp_xy, sizeof(i->wp_xy)); memcpy(i->prim_xy, m.prim_xy, sizeof(i->prim_xy));
}

// Patch dictionary (K.3.1), at the head of LfGlobal when FrameHeader.flags has kPatches: entropy-coded with 10 contexts.  Every placement
// becomes one DevPatch; the device blends them after the loop filters (dev_compose.h).
// Splines (K.4), in LfGlobal behind the patch dictionary when FrameHeader.flags has kSplines: quantised control points (double deltas from a starting point),
// 32 DCT coefficients per colour channel and for the thickness (sigma), one quantisation adjustment for all splines; 6 contexts.  What libjxl's
// Splines::Decode / QuantizedSpline::Dequantize / InitializeDrawCache do before a pixel is touched: the curve through the control points (centripetal
// Catmull-Rom, 16 steps per span), resampled at unit arc length; colour and sigma of every sample from the continuous inverse DCTs; one segment per sample.
struct QSpline { std::vector<std::pair<int64_t, int64_t>> cp; int32_t color[3][32]; int32_t sigma[32]; float sx, sy; };
static int parse_splines(FramePlan *plan, hx_br *sb, uint64_t num_pixels, std::vector<QSpline> *out, int32_t *quant_adjust) {
  hx_ec ec;
  if (hx_ec_read_header(&ec, sb, 6)) { plan->error = "splines: bad entropy header"; return -1; }
  hx_ec_begin(&ec, sb, 0);
  const char *err = nullptr;
  const uint64_t max_cp = std::min<uint64_t>(1u << 20, num_pixels / 2);
  const uint64_t num = (uint64_t)hx_ec_read(&ec, sb, 2) + 1;
  if (num > max_cp) err = "splines: too many splines";      // (libjxl: num_splines > max_control_points fails — ADVICE r5: one more was accepted here)
  std::vector<QSpline> q;
  if (!err) q.resize((size_t)num);
  int64_t lx = 0, ly = 0;
  for (size_t i = 0; i < q.size() && !err; i++) {
    int64_t x = hx_ec_read(&ec, sb, 1), y = hx_ec_read(&ec, sb, 1);
    if (i) { x = hx_unpack_signed((uint32_t)x) + lx; y = hx_unpack_signed((uint32_t)y) + ly; }
    if (x >= (1 << 23) || x <= -(1 << 23) || y >= (1 << 23) || y <= -(1 << 23)) { err = "splines: starting point out of range"; break; }
    q[i].sx = (float)x; q[i].sy = (float)y; lx = x; ly = y;
  }
  if (!err) *quant_adjust = hx_unpack_signed(hx_ec_read(&ec, sb, 0));
  uint64_t total_cp = 0;
  for (size_t i = 0; i < q.size() && !err; i++) {
    const uint64_t n = hx_ec_read(&ec, sb, 3);
    total_cp += n;
    if (total_cp > max_cp) { err = "splines: too many control points"; break; }
    q[i].cp.resize((size_t)n);
    for (auto &p : q[i].cp) {
      p.first = hx_unpack_signed(hx_ec_read(&ec, sb, 4)); p.second = hx_unpack_signed(hx_ec_read(&ec, sb, 4));
      if (std::llabs(p.first) >= (1 << 30) || std::llabs(p.second) >= (1 << 30)) { err = "splines: control point delta out of range"; break; }
    }
    for (int c = 0; c < 3 && !err; c++) for (int k = 0; k < 32; k++) { q[i].color[c][k] = hx_unpack_signed(hx_ec_read(&ec, sb, 5)); if (q[i].color[c][k] == INT32_MIN) err = "splines: DCT coefficient out of range"; }
    for (int k = 0; k < 32 && !err; k++) { q[i].sigma[k] = hx_unpack_signed(hx_ec_read(&ec, sb, 5)); if (q[i].sigma[k] == INT32_MIN) err = "splines: DCT coefficient out of range"; }      // (libjxl refuses INT_MIN: its magnitude does not exist)
    if (sb->err) { err = "truncated splines"; break; }
  }
  const int ok = hx_ec_final_ok(&ec);
  hx_ec_free(&ec);
  if (err) { plan->error = err; return -1; }
  if (!ok || sb->err) { plan->error = "splines: ANS final state"; return -1; }
  out->swap(q);
  return 0;
}
struct SPoint { float x, y; };
static float spline_idct(const float *dct, float t) {       // libjxl's ContinuousIDCT: a 32-point DCT-III evaluated at a real position, sqrt(2)-scaled (dct[0] carries 1 / sqrt(2))
  float r = 0.0f;
  for (int i = 0; i < 32; i++) r += dct[i] * cosf((3.14159265358979323846f / 32.0f) * (float)i * (t + 0.5f));
  return 1.41421356237f * r;
}
static int build_splines(FramePlan *plan, Blob &blob, DevFrame &F, const std::vector<QSpline> &qs, int32_t quant_adjust, float y_to_x, float y_to_b, int width, int height) {
  static const float kChannelWeight[4] = {0.0042f, 0.075f, 0.07f, 0.3333f};
  const float inv_quant = quant_adjust >= 0 ? 1.0f / (1.0f + 0.125f * (float)quant_adjust) : 1.0f - 0.125f * (float)quant_adjust;
  std::vector<DevSplineSeg> segs;
  std::vector<std::pair<int32_t, uint32_t>> by_y;
  // libjxl's complexity limits (QuantizedSpline::Dequantize): a spline's control polygon may not be longer (Manhattan) than area_limit, and the sum over the
  // splines of [estimated width x that length] may not exceed it either — a few hundred bytes of a crafted file could otherwise ask for 2^25 row entries x the image width
  // of work in k_splines.  The estimate is libjxl's: per sigma coefficient ceil(inv_quant |q|) clamped to [1, weight_limit], squared, times log2 of the largest colour sum.
  const uint64_t image_size = (uint64_t)width * (uint64_t)height;
  const uint64_t area_limit = std::min<uint64_t>(1024ull * image_size + (1ull << 32), 1ull << 42);
  uint64_t total_estimated_area = 0;
  for (const QSpline &q : qs) {
    // control points: the starting point, then running sums of running sums of the coded deltas
    std::vector<SPoint> cp;
    cp.push_back({q.sx, q.sy});
    int64_t cx = (int64_t)llroundf(q.sx), cy = (int64_t)llroundf(q.sy), dx = 0, dy = 0;
    uint64_t manhattan = 0;
    for (const auto &p : q.cp) {
      dx += p.first; dy += p.second; cx += dx; cy += dy;
      manhattan += (uint64_t)std::llabs(dx) + (uint64_t)std::llabs(dy);
      if (manhattan > area_limit) { plan->error = "splines: control polygon beyond the area limit"; return -1; }
      if (std::llabs(cx) >= (1ll << 30) || std::llabs(cy) >= (1ll << 30) || std::llabs(dx) >= (1ll << 30) || std::llabs(dy) >= (1ll << 30)) { plan->error = "splines: control point out of range"; return -1; }
      cp.push_back({(float)cx, (float)cy});
    }
    float cdct[3][32], sdct[32];
    for (int c = 0; c < 3; c++) for (int i = 0; i < 32; i++) cdct[c][i] = (float)q.color[c][i] * (i == 0 ? 0.70710678118f : 1.0f) * kChannelWeight[c] * inv_quant;
    for (int i = 0; i < 32; i++) { cdct[0][i] += y_to_x * cdct[1][i]; cdct[2][i] += y_to_b * cdct[1][i]; }
    for (int i = 0; i < 32; i++) sdct[i] = (float)q.sigma[i] * (i == 0 ? 0.70710678118f : 1.0f) * kChannelWeight[3] * inv_quant;
    {
      uint64_t col[3] = {0, 0, 0};
      for (int c = 0; c < 3; c++) for (int i = 0; i < 32; i++) col[c] += (uint64_t)ceilf(inv_quant * fabsf((float)q.color[c][i]));
      col[0] += (uint64_t)ceilf(fabsf(y_to_x)) * col[1]; col[2] += (uint64_t)ceilf(fabsf(y_to_b)) * col[1];
      const uint64_t max_col = std::max(col[1], std::max(col[0], col[2]));
      uint64_t logcolor = 1; while (logcolor < 64 && (1ull << logcolor) < 1ull + max_col) logcolor++;           // max(1, ceil(log2(1 + max colour sum)))
      const float weight_limit = ceilf(sqrtf(((float)area_limit / (float)logcolor) / (float)std::max<uint64_t>(1, manhattan)));
      uint64_t width_estimate = 0;
      for (int i = 0; i < 32; i++) {
        const float wf = ceilf(inv_quant * fabsf((float)q.sigma[i]));
        const uint64_t wgt = (uint64_t)std::min(weight_limit, std::max(1.0f, wf));
        width_estimate += wgt * wgt * logcolor;
      }
      total_estimated_area += width_estimate * manhattan;
      if (total_estimated_area > area_limit) { plan->error = "splines: estimated area beyond the limit"; return -1; }
    }
    // centripetal Catmull-Rom through the control points, 16 intermediate points per span
    std::vector<SPoint> pts;
    if (cp.size() == 1) pts.push_back(cp[0]);
    else {
      std::vector<SPoint> e;
      e.push_back({cp[0].x + (cp[0].x - cp[1].x), cp[0].y + (cp[0].y - cp[1].y)});
      e.insert(e.end(), cp.begin(), cp.end());
      const size_t n = cp.size();
      e.push_back({cp[n - 1].x + (cp[n - 1].x - cp[n - 2].x), cp[n - 1].y + (cp[n - 1].y - cp[n - 2].y)});
      for (size_t st = 0; st + 3 < e.size(); st++) {
        const SPoint *p = &e[st];
        pts.push_back(p[1]);
        float d[3], t[4]; t[0] = 0.0f;
        for (int k = 0; k < 3; k++) { d[k] = sqrtf(hypotf(p[k + 1].x - p[k].x, p[k + 1].y - p[k].y)); t[k + 1] = t[k] + d[k]; }
        for (int i = 1; i < 16; i++) {
          const float tt = d[0] + ((float)i / 16.0f) * d[1];
          SPoint a[3], b[2];
          for (int k = 0; k < 3; k++) { const float w = (tt - t[k]) / d[k]; a[k] = {p[k].x + w * (p[k + 1].x - p[k].x), p[k].y + w * (p[k + 1].y - p[k].y)}; }
          for (int k = 0; k < 2; k++) { const float w = (tt - t[k]) / (d[k] + d[k + 1]); b[k] = {a[k].x + w * (a[k + 1].x - a[k].x), a[k].y + w * (a[k + 1].y - a[k].y)}; }
          const float w = (tt - t[1]) / d[1];
          pts.push_back({b[0].x + w * (b[1].x - b[0].x), b[0].y + w * (b[1].y - b[0].y)});
        }
      }
      pts.push_back(e[e.size() - 2]);
    }
    // samples at unit arc length along that polyline; the last one carries what is left of the last step as its weight
    std::vector<std::pair<SPoint, float>> draw;
    {
      SPoint current = pts[0];
      draw.push_back({current, 1.0f});
      size_t next_id = 0;
      bool done = false;
      while (!done && next_id < pts.size()) {
        SPoint previous = current;
        float from_prev = 0.0f;
        for (;;) {
          if (next_id >= pts.size()) { draw.push_back({previous, from_prev}); done = true; break; }
          const SPoint nx = pts[next_id];
          const float to_next = sqrtf((nx.x - previous.x) * (nx.x - previous.x) + (nx.y - previous.y) * (nx.y - previous.y));
          if (from_prev + to_next >= 1.0f) {
            const float w = (1.0f - from_prev) / to_next;
            current = {previous.x + w * (nx.x - previous.x), previous.y + w * (nx.y - previous.y)};
            draw.push_back({current, 1.0f});
            break;
          }
          from_prev += to_next; previous = nx; next_id++;
        }
        if (draw.size() > (1u << 22)) { plan->error = "unsupported: spline longer than 2^22 pixels"; return -1; }
      }
    }
    const float arc = (float)((double)draw.size() - 2.0) + draw.back().second;
    if (!(arc > 0.0f)) continue;                               // no effect
    const float inv_arc = 1.0f / arc;
    for (size_t k = 0; k < draw.size(); k++) {
      const float prog = std::min(1.0f, (float)k * inv_arc);
      float color[3];
      for (int c = 0; c < 3; c++) color[c] = spline_idct(cdct[c], 31.0f * prog);
      const float sigma = spline_idct(sdct, 31.0f * prog);
      const float mult = draw[k].second;
      if (!std::isfinite(sigma) || sigma == 0.0f || !std::isfinite(1.0f / sigma) || !std::isfinite(color[0]) || !std::isfinite(color[1]) || !std::isfinite(color[2])) continue;
      float max_color = 0.01f;
      for (int c = 0; c < 3; c++) max_color = std::max(max_color, fabsf(color[c] * mult));
      // beyond this distance the blob is below 1e-5 of an intensity step
      const float maxdist = sqrtf(-2.0f * sigma * sigma * (logf(0.1f) * 5.0f - logf(max_color)));
      DevSplineSeg sg;
      sg.cx = draw[k].first.x; sg.cy = draw[k].first.y;
      for (int c = 0; c < 3; c++) sg.color[c] = color[c];
      sg.inv_sigma = 1.0f / sigma; sg.sigma_over_4_times_intensity = 0.25f * sigma * mult; sg.maxdist = maxdist;
      if (!std::isfinite(maxdist)) continue;
      const long long y0 = llroundf(sg.cy - maxdist), y1 = llroundf(sg.cy + maxdist) + 1;
      for (long long y = std::max<long long>(y0, 0); y < std::min<long long>(y1, height); y++) by_y.push_back({(int32_t)y, (uint32_t)segs.size()});
      segs.push_back(sg);
      if (by_y.size() > (1u << 25) || segs.size() > (1u << 22)) { plan->error = "unsupported: splines cover more than 2^25 row segments"; return -1; }
    }
  }
  std::stable_sort(by_y.begin(), by_y.end(), [](const std::pair<int32_t, uint32_t> &a, const std::pair<int32_t, uint32_t> &b) { return a.first < b.first; });
  std::vector<uint32_t> rows((size_t)height + 1, 0), idx(by_y.size());
  for (size_t i = 0; i < by_y.size(); i++) { rows[(size_t)by_y[i].first + 1]++; idx[i] = by_y[i].second; }
  for (int y = 0; y < height; y++) rows[(size_t)y + 1] += rows[(size_t)y];
  F.num_spline_segs = (int32_t)segs.size();
  F.spline_seg_off = blob.append(segs.data(), segs.size() * sizeof(DevSplineSeg));
  F.spline_row_off = blob.append(rows.data(), rows.size() * 4);
  F.spline_idx_off = blob.append(idx.data(), idx.size() * 4);
  (void)width;
  return 0;
}

static int parse_patches(FramePlan *plan, Priv *pv, hx_br *sb, Blob &blob) {
  DevFrame &F = pv->F; const frame_hdr &f = pv->f; const img_meta &m = pv->m;
  hx_ec ec;
  if (hx_ec_read_header(&ec, sb, 10)) { plan->error = "patch dictionary: bad entropy header"; return -1; }
  hx_ec_begin(&ec, sb, 0);
  std::vector<DevPatch> out;
  const char *err = nullptr;
  const uint32_t num_ref = hx_ec_read(&ec, sb, 0);
  const uint64_t frame_px = (uint64_t)f.coded_width * (uint64_t)f.coded_height;
  if ((uint64_t)num_ref > frame_px + 1) err = "patch dictionary: too many patches";
  uint64_t total = 0;
  for (uint32_t id = 0; id < num_ref && !err; id++) {
    DevPatch P; memset(&P, 0, sizeof(P));
    P.ref = (int32_t)hx_ec_read(&ec, sb, 1);
    P.x0 = (int32_t)hx_ec_read(&ec, sb, 3); P.y0 = (int32_t)hx_ec_read(&ec, sb, 3);
    const uint32_t pw = hx_ec_read(&ec, sb, 2), ph = hx_ec_read(&ec, sb, 2);
    if (P.ref < 0 || P.ref > 3 || pv->ref_w[P.ref] <= 0) { err = "patch dictionary: reference to an empty slot"; break; }
    if (!pv->ref_patch_ok[P.ref]) { err = "unsupported: patch taken from a frame saved after the colour transform"; break; }
    if (pw >= (1u << 24) || ph >= (1u << 24) || P.x0 < 0 || P.y0 < 0 || (uint64_t)P.x0 + pw + 1 > (uint64_t)pv->ref_w[P.ref] || (uint64_t)P.y0 + ph + 1 > (uint64_t)pv->ref_h[P.ref]) { err = "patch dictionary: patch outside its reference frame"; break; }
    P.w = (int32_t)pw + 1; P.h = (int32_t)ph + 1;
    const uint32_t count = hx_ec_read(&ec, sb, 7);
    if ((uint64_t)count + 1 > frame_px) { err = "patch dictionary: too many placements"; break; }
    for (uint32_t i = 0; i <= count && !err; i++) {
      if (i == 0) { P.x = (int32_t)hx_ec_read(&ec, sb, 4); P.y = (int32_t)hx_ec_read(&ec, sb, 4); }
      else { P.x += hx_unpack_signed(hx_ec_read(&ec, sb, 6)); P.y += hx_unpack_signed(hx_ec_read(&ec, sb, 6)); }
      if (P.x < 0 || P.y < 0 || (int64_t)P.x + P.w > f.coded_width || (int64_t)P.y + P.h > f.coded_height) { err = "patch dictionary: placement outside the frame"; break; }
      for (int j = 0; j <= m.num_extra; j++) {
        const uint32_t mode = hx_ec_read(&ec, sb, 5);
        if (mode >= 8) { err = "patch dictionary: bad blend mode"; break; }
        if (mode >= 4 && m.num_extra > 1) (void)hx_ec_read(&ec, sb, 8);      // alpha channel of the alpha-weighted modes
        if (mode >= 3) (void)hx_ec_read(&ec, sb, 9);                          // clamp flag (kMul and the alpha modes)
        if (j == 0) {
          if (mode >= 4) { err = "unsupported: alpha-weighted patch blending"; break; }
          P.mode = (int32_t)mode;
        } else if (mode != 0) { err = "unsupported: patches on extra channels"; break; }
      }
      if (err) break;
      total += (uint64_t)P.w * (uint64_t)P.h;
      if (out.size() >= (1u << 22) || total > ((uint64_t)1 << 33)) { err = "unsupported: patch dictionary beyond 2^22 placements"; break; }
      if (P.mode != 0) out.push_back(P);
    }
    if (sb->err) { err = "truncated patch dictionary"; break; }
  }
  const int ok = hx_ec_final_ok(&ec);
  hx_ec_free(&ec);
  if (err) { plan->error = err; return -1; }
  if (!ok || sb->err) { plan->error = "patch dictionary: ANS final state"; return -1; }
  // libjxl applies the placements in file order; the device blends them concurrently (k_patch_blend), which only kAdd survives (atomic adds commute).
  // A kReplace / kMul placement that overlaps any other one would depend on the order: not produced by libjxl's encoder — refused, not raced
  {
    std::vector<size_t> ordered;
    for (size_t i = 0; i < out.size(); i++) if (out[i].mode != 2) ordered.push_back(i);
    if (ordered.size() > 4096) { plan->error = "unsupported: patch dictionary with more than 4096 replace / multiply placements"; return -1; }
    // the check below is ordered x all rectangle tests on the parse thread: bounded at 4096 x 65536 (a fraction of a second) — an untrusted file with millions of
    // placements beside one replacing placement is refused here instead of costing tens of seconds
    if (!ordered.empty() && out.size() > 65536) { plan->error = "unsupported: patch dictionary with replace / multiply placements among more than 65536 placements"; return -1; }
    for (size_t a : ordered)
      for (size_t b = 0; b < out.size(); b++) {
        if (b == a) continue;
        const DevPatch &A = out[a], &Bp = out[b];
        if (A.x < Bp.x + Bp.w && Bp.x < A.x + A.w && A.y < Bp.y + Bp.h && Bp.y < A.y + A.h) { plan->error = "unsupported: overlapping patch placements with replace / multiply blending"; return -1; }
      }
  }
  F.num_patches = (int32_t)out.size();
  F.patch_off = blob.append(out.data(), out.size() * sizeof(DevPatch));
  int mw = 1, mh = 1;
  for (const DevPatch &P : out) { mw = std::max(mw, (int)P.w); mh = std::max(mh, (int)P.h); }
  plan->patch_max_px = (size_t)mw * (size_t)mh;
  return 0;
}

static int parse_hf_global(FramePlan *plan, Priv *pv, hx_br *br) {
  DevFrame &F = pv->F;
  const frame_hdr &f = pv->f;
  Blob blob(plan->tables);
  if (!hx_bool(br)) {
    // DequantMatrices (ISO/IEC 18181-1 I.2.4), one encoding per quant table: 0 library, 1 - 5 the special 8 x 8 tables from their own parameters, 6 DCT band parameters,
    // 7 RAW (a Modular image of three channels — libjxl's 8x8 table of a recompressed JPEG).  The multipliers go into the frame blob (DevFrame::qw_frame_off).
    for (int t = 0; t < 17; t++) {
      const int mode = (int)hx_bits(br, 3);
      const int rows = kQTRows[t] * 8, cols = kQTCols[t] * 8, n = rows * cols;
      if (mode == 0) continue;
      std::vector<float> mul[3];
      for (int c = 0; c < 3; c++) mul[c].assign((size_t)n, 0.0f);
      if (mode == 6) {
        const int nb = (int)hx_bits(br, 4) + 1;
        double b[3][17];
        for (int c = 0; c < 3; c++) {
          for (int i = 0; i < nb; i++) b[c][i] = hx_f16(br);
          if (b[c][0] < 1e-8) { plan->error = "bad DCT quant parameters"; return -1; }
          b[c][0] *= 64.0;
        }
        for (int c = 0; c < 3; c++) {
          double bands[17];
          bands[0] = b[c][0];
          for (int i = 1; i < nb; i++) { bands[i] = bands[i - 1] * band_mult(b[c][i]); if (bands[i] < 1e-8) { plan->error = "bad DCT quant parameters"; return -1; } }
          for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) {
            const double dx = (double)x / (cols - 1), dy = (double)y / (rows - 1);
            mul[c][(size_t)(y * cols + x)] = 1.0f / (float)interp_bands(sqrt(dx * dx + dy * dy), sqrt(2.0) + 1e-6, bands, nb);
          }
        }
      } else if (mode == 7) {
        const float den = hx_f16(br);
        if (den < 1e-8f) { plan->error = "bad RAW quant table denominator"; return -1; }
        std::vector<int32_t> q;
        if (modsmall::decode(br, cols, rows, 3, 1 + 3 * f.num_lf_groups + t, plan->cs + pv->secs[0].off,
                             plan->single_section ? plan->cs_size - pv->secs[0].off : pv->secs[0].size, pv->tree_bit, &q, &plan->error)) return -1;
        for (int c = 0; c < 3; c++) for (int i = 0; i < n; i++) {
          const int32_t v = q[(size_t)c * (size_t)n + (size_t)i];
          if (v <= 0) { plan->error = "RAW quant table entry <= 0"; return -1; }
          const float wgt = 1.0f / (den * (float)v);           // libjxl keeps this "weight" and multiplies by its reciprocal (~ den * q)
          if (wgt >= 1e8f || wgt < 1e-8f) { plan->error = "RAW quant table entry out of range"; return -1; }
          mul[c][(size_t)i] = 1.0f / wgt;
        }
      } else {                                          // modes 1 - 5: the special 8 x 8 tables from their own parameters (host_format.inc: read_special_quant_weights)
        float wbuf[3][64]; float *w[3] = {wbuf[0], wbuf[1], wbuf[2]};
        if (read_special_quant_weights(br, t, mode, w)) { plan->error = "invalid: dequant matrix parameters (encoding mode " + std::to_string(mode) + " for table " + std::to_string(t) + ")"; return -1; }
        for (int c = 0; c < 3; c++) for (int i = 0; i < 64; i++) mul[c][(size_t)i] = 1.0f / w[c][i];
      }
      if (br->err) { plan->error = "truncated HfGlobal"; return -1; }
      for (int c = 0; c < 3; c++) F.qw_frame_off[t][c] = blob.append(mul[c].data(), (size_t)n * 4);
    }
  }
  F.num_presets = 1 + (int)hx_bits(br, ceil_log2u((uint32_t)f.num_groups));
  if (f.num_passes > kMaxPasses) { plan->error = "invalid: more than 11 passes"; return -1; }
  if (getenv("JXLAMD_PARSE_TRACE")) fprintf(stderr, "HfGlobal: %d presets, %d block contexts, %d groups\n", F.num_presets, F.num_bctx, f.num_groups);
  for (int p = 0; p < f.num_passes; p++) {
    uint32_t used = hx_u32(br, -1, 0x5F, -1, 0x13, -1, 0, 13, 0);
    hx_ec oc; int have_oc = 0;
    if (used) { if (hx_ec_read_header(&oc, br, 8)) { plan->error = "bad coefficient-order code"; return -1; } hx_ec_begin(&oc, br, 0); have_oc = 1; }
    const std::vector<uint8_t> &stat = static_tables();
    const DevStatic &ST = *(const DevStatic *)stat.data();
    for (int o = 0; o < 13; o++) {
      // natural orders live in the static tables (1.5 MB for the 13 buckets: not rebuilt and uploaded with every frame)
      if (!(used & (1u << o))) { for (int c = 0; c < 3; c++) F.order_off[p][o][c] = kOrderInStatic | ST.nat_order_off[o]; continue; }
      int st = kOrderStrategy[o];
      uint32_t size = (uint32_t)kCoveredX[st] * kCoveredY[st] * 64;
      const uint32_t *nat = (const uint32_t *)(stat.data() + ST.nat_order_off[o]);
      std::vector<uint32_t> ord(size), perm(size);
      for (int c = 0; c < 3; c++) {
        if (hx_read_permutation(&oc, br, perm.data(), size, size / 64)) { hx_ec_free(&oc); plan->error = "bad coefficient order"; return -1; }
        for (uint32_t i = 0; i < size; i++) ord[i] = nat[perm[i]];
        F.order_off[p][o][c] = blob.append(ord.data(), (size_t)size * 4);
      }
    }
    if (have_oc) { int ok = hx_ec_final_ok(&oc); hx_ec_free(&oc); if (!ok) { plan->error = "coefficient orders: ANS final state"; return -1; } }
    hx_ec hf;
    if (hx_ec_read_header(&hf, br, 495 * F.num_presets * F.num_bctx)) { plan->error = "bad HF histograms"; return -1; }
    int rc = pack_ec(hf, blob, &F.hf_ec[p], &plan->error);
    hx_ec_free(&hf);
    if (rc) return -1;
  }
  if (br->err) { plan->error = "truncated HfGlobal"; return -1; }
  plan->hf_parsed = true;
  return 0;
}

static void finish_blob(FramePlan *plan, Priv *pv) { memcpy(plan->tables.data(), &pv->F, sizeof(DevFrame)); }


// GlobalModular of a Modular-encoded frame: parse the stream header (transforms) on the host, derive the channel list
// the device decodes and the inverse-transform program with resolved plane indices.  Sample data stays on the device.
static int parse_modular_global(FramePlan *plan, Priv *pv, hx_br *sb, bool vardct = false) {
  DevFrame &F = pv->F; const frame_hdr &f = pv->f; const img_meta &m = pv->m;
  F.is_modular = vardct ? 0 : 1; F.has_ec = vardct ? 1 : 0;      // vardct: the Modular image holds only the extra channels
  F.mod_global_bit = (uint32_t)sb->pos;
  F.mod_group_dim = f.group_dim;
  F.mod_bits = (int)m.pub.bits_per_sample;
  // float samples: the integer planes hold the floats' bit patterns (dev_entropy.h: sample_bits_to_float); an XYB Modular frame's samples are quantised XYB whatever the image declares
  F.mod_exp_bits = (vardct || m.pub.xyb_encoded) ? 0 : (int)m.pub.exp_bits;
  if (F.mod_exp_bits ? (F.mod_bits > 32 || F.mod_exp_bits > 8 || F.mod_bits - F.mod_exp_bits - 1 < 1 || F.mod_bits - F.mod_exp_bits - 1 > 23) : (F.mod_bits > 24 && !vardct && !m.pub.xyb_encoded)) { plan->error = "unsupported: integer samples of more than 24 bits / float format in Modular frames"; return -1; }
  // (round 6: integer samples of 17 .. 24 bits — what libjxl's encoder writes at most, from 24-bit PNM / PNG-16 + padding sources: the planes are int32 like libjxl's, the
  // decode loops of images without modular_16bit_buffers run on 64-bit neighbourhoods, and RCT / palette / squeeze sums of 24-bit samples stay inside 32 bits; 25 .. 31 bits
  // would need the inverse transforms' intermediate sums audited against libjxl's 64-bit-then-truncate arithmetic)
  const int ncol = vardct ? 0 : (m.pub.num_color_channels == 1 && !m.pub.xyb_encoded) ? 1 : 3;
  struct Ch { int w, h, hs, vs, plane; };
  std::vector<Ch> L;
  for (int i = 0; i < ncol + m.num_extra; i++) {
    if (i >= ncol && m.ec[i - ncol].dim_shift) { plan->error = "unsupported: extra channel dim_shift"; return -1; }
    // colour channels: the coded size of the frame; an extra channel: ceil(full size / its own upsampling factor) (>= the frame's)
    if (i < ncol) L.push_back({f.coded_width, f.coded_height, 0, 0, -1});
    else {
      const int u = f.ec_upsampling[i - ncol];
      int sh = 0; while ((f.upsampling << sh) < u) sh++;            // coarser than the colour channels by this shift: its group rectangles shrink with it
      // (an LF frame is the image at an eighth: its extra channels shrink with it)
      const int fw = f.frame_type == 1 ? f.coded_width : f.width, fh = f.frame_type == 1 ? f.coded_height : f.height;
      L.push_back({(fw + u - 1) / u, (fh + u - 1) / u, sh, sh, -1});
    }
  }
  const std::vector<Ch> L0 = L;                 // what the inverse transforms must arrive at again
  int nb_meta = 0;
  // GroupHeader (H.2): use_global_tree, WP header (skipped here, the device parses it again), transforms
  const bool use_global_tree = hx_bool(sb);        // a local tree + code follow the transforms: the device parses them (as it does for LF groups)
  if (use_global_tree && F.tree_count <= 0) { plan->error = "modular frame: missing global MA tree"; return -1; }
  if (!hx_bool(sb)) { for (int i = 0; i < 7; i++) (void)hx_bits(sb, 5); for (int i = 0; i < 4; i++) (void)hx_bits(sb, 4); }
  const int ntr = (int)hx_u32(sb, -1, 0, -1, 1, 4, 2, 8, 18);
  if (ntr > 32) { plan->error = "unsupported: more than 32 global transforms"; return -1; }      // (every channel may bring a palette of its own — libjxl compacts channels of few distinct values, e.g. float16 — next to RCT and squeeze; the operation list and the plane table bound the total)
  struct Sq { int horizontal, in_place, begin_c, num_c; };
  struct Tr { int id, begin_c, rct_type, num_c, nb_colours, nb_deltas, d_pred; std::vector<Sq> sq; };
  std::vector<Tr> trs;
  for (int i = 0; i < ntr; i++) {
    Tr t{}; t.id = (int)hx_bits(sb, 2);
    if (t.id == 0) {
      t.begin_c = (int)hx_u32(sb, 3, 0, 6, 8, 10, 72, 13, 1096);
      t.rct_type = (int)hx_u32(sb, -1, 6, 2, 0, 4, 2, 6, 10);
      if (t.rct_type >= 42 || t.begin_c + 3 > (int)L.size()) { plan->error = "bad RCT transform"; return -1; }
    } else if (t.id == 1) {
      t.begin_c = (int)hx_u32(sb, 3, 0, 6, 8, 10, 72, 13, 1096);
      t.num_c = (int)hx_u32(sb, -1, 1, -1, 3, -1, 4, 13, 1);
      t.nb_colours = (int)hx_u32(sb, 8, 0, 10, 256, 12, 1280, 16, 5376);
      t.nb_deltas = (int)hx_u32(sb, -1, 0, 8, 1, 10, 257, 16, 1281);
      t.d_pred = (int)hx_bits(sb, 4);
      if (t.nb_deltas > 0 && t.d_pred == 6) { plan->error = "unsupported: weighted-predictor delta palette"; return -1; }
      if (t.d_pred > 13) { plan->error = "bad palette predictor"; return -1; }
      if (t.num_c < 1 || t.begin_c + t.num_c > (int)L.size() || t.nb_colours < 0 || t.begin_c < nb_meta) { plan->error = "unsupported: palette layout"; return -1; }
      for (int c = 1; c < t.num_c; c++)
        if (L[(size_t)t.begin_c + c].w != L[(size_t)t.begin_c].w || L[(size_t)t.begin_c + c].h != L[(size_t)t.begin_c].h || L[(size_t)t.begin_c + c].hs != L[(size_t)t.begin_c].hs ||
            L[(size_t)t.begin_c + c].vs != L[(size_t)t.begin_c].vs) { plan->error = "palette over channels of different size"; return -1; }
      // meta-apply: the num_c channels collapse into one index channel at begin_c, the palette ((nb_colours + nb_deltas) x num_c: the delta entries first) becomes meta channel 0
      L.erase(L.begin() + t.begin_c + 1, L.begin() + t.begin_c + t.num_c);
      L.insert(L.begin(), {t.nb_colours + t.nb_deltas, t.num_c, -1, -1, -1});
      nb_meta++;
    } else if (t.id == 2) {
      // Squeeze (H.6.2): explicit steps, or — num_sq == 0 — the default sequence derived from the channel list
      const int num_sq = (int)hx_u32(sb, -1, 0, 4, 1, 6, 9, 8, 41);
      if (num_sq > 64) { plan->error = "unsupported: more than 64 squeeze steps"; return -1; }
      for (int q = 0; q < num_sq; q++) {
        Sq p{};
        p.horizontal = hx_bool(sb) ? 1 : 0; p.in_place = hx_bool(sb) ? 1 : 0;
        p.begin_c = (int)hx_u32(sb, 3, 0, 6, 8, 10, 72, 13, 1096);
        p.num_c = (int)hx_u32(sb, -1, 1, -1, 2, -1, 3, 4, 4);
        t.sq.push_back(p);
      }
      if (num_sq == 0) {
        const int first = nb_meta, nbc = (int)L.size() - nb_meta;
        if (nbc < 1) { plan->error = "squeeze without channels"; return -1; }
        int w = L[(size_t)first].w, h = L[(size_t)first].h;
        if (nbc > 2 && L[(size_t)first + 1].w == w && L[(size_t)first + 1].h == h) {      // channels 1 and 2 as chroma: squeezed first (4:2:0-like previews)
          t.sq.push_back({1, 0, first + 1, 2});
          t.sq.push_back({0, 0, first + 1, 2});
        }
        Sq p{0, 1, first, nbc};
        if (!(w > h) && h > 8) { p.horizontal = 0; t.sq.push_back(p); h = (h + 1) / 2; }     // tall (or square) images start with a vertical step
        while (w > 8 || h > 8) {
          if (w > 8) { p.horizontal = 1; t.sq.push_back(p); w = (w + 1) / 2; }
          if (h > 8) { p.horizontal = 0; t.sq.push_back(p); h = (h + 1) / 2; }
        }
      }
      // meta-apply: each step halves channels [begin_c, begin_c + num_c) along one axis and inserts their residual channels
      for (const Sq &p : t.sq) {
        const int b = p.begin_c, e = p.begin_c + p.num_c - 1;
        if (b < nb_meta || e >= (int)L.size()) { plan->error = "squeeze: channel range"; return -1; }      // (squeezing meta channels: not produced by libjxl)
        const int offset = p.in_place ? e + 1 : (int)L.size();
        for (int c = b; c <= e; c++) {
          Ch &a = L[(size_t)c];
          if (a.hs < 0 || a.hs > 30 || a.vs > 30) { plan->error = "squeeze: bad channel"; return -1; }
          Ch r = a;
          if (p.horizontal) { const int w = a.w; a.w = (w + 1) / 2; a.hs++; r.w = w - a.w; r.hs = a.hs; }
          else { const int h = a.h; a.h = (h + 1) / 2; a.vs++; r.h = h - a.h; r.vs = a.vs; }
          L.insert(L.begin() + offset + (c - b), r);
        }
      }
    } else { plan->error = "bad transform id"; return -1; }
    trs.push_back(t);
  }
  if (sb->err) { plan->error = "truncated GlobalModular header"; return -1; }
  if ((int)L.size() > kModMaxCh) { plan->error = "unsupported: more than 128 modular channels"; return -1; }
  F.mod_nch = (int)L.size(); F.mod_nb_meta = nb_meta;
  // plane offsets (in samples) accumulate in 64 bits: the inverse squeeze steps add one output plane each (~3x the image), and nothing above
  // bounds a frame's channels by 2^32 samples in total.  The device indexes the pool with 32-bit offsets: beyond that the frame is refused
  uint64_t off = 0;
  constexpr uint64_t kPoolLimit = 0xFFFFFFFFull - 4096;
  int first_group = F.mod_nch;
  for (int i = 0; i < F.mod_nch; i++) {
    const Ch &c = L[(size_t)i];
    F.mod_w[i] = c.w; F.mod_h[i] = c.h; F.mod_hs[i] = (uint8_t)(c.hs < 0 ? 0 : c.hs); F.mod_vs[i] = (uint8_t)(c.vs < 0 ? 0 : c.vs);
    if (first_group == F.mod_nch && i >= nb_meta && (c.w > f.group_dim || c.h > f.group_dim)) first_group = i;
  }
  // channels after the first one that exceeds a group travel in the ModularLfGroup (both shifts >= 3) or ModularGroup streams; the device
  // decodes the latter (single-pass frames: shifts 0..2)
  F.mod_lf_nch = 0;
  for (int i = first_group; i < F.mod_nch; i++) if (std::min(L[(size_t)i].hs, L[(size_t)i].vs) >= 3) F.mod_lf_nch++;
  if (F.mod_lf_nch > 32) { plan->error = "unsupported: more than 32 ModularLfGroup channels"; return -1; }      // (S.grp_src holds kModMaxGroupCh; 16 for RGBA with squeeze at 8200 x 8200)
  // planes are indexed by stream channel position (the device decodes "channel i" into plane i); the inverse squeeze steps append theirs
  int nplanes = F.mod_nch;
  for (int i = 0; i < F.mod_nch; i++) {
    L[(size_t)i].plane = i; F.mod_plane_off[i] = (uint32_t)off; off += (uint64_t)F.mod_w[i] * (uint64_t)F.mod_h[i] + 64;
    if (off > kPoolLimit) { plan->error = "unsupported: Modular image beyond 2^32 samples"; return -1; }
  }
  F.mod_first_group_ch = first_group;
  F.lz_win_len = 0; F.lz_win_group = 0;
  if (F.tree_ec.lz77 && !vardct) {            // a stream never holds more integers than the image has samples; the window is 2^20 at most
    // the GlobalModular stream holds the channels before first_group, a group stream at most group_dim^2 samples of each later channel
    uint64_t total = 0;
    for (int i = 0; i < first_group; i++) total += (uint64_t)F.mod_w[i] * (uint64_t)F.mod_h[i];
    F.lz_win_len = (uint32_t)std::min<uint64_t>(total + 64, 1u << 20);
    F.lz_win_group = (uint32_t)std::min<uint64_t>((uint64_t)(F.mod_nch - first_group) * (uint64_t)f.group_dim * (uint64_t)f.group_dim + 64, 1u << 20);
  }
  if (F.mod_nch - first_group > kModMaxGroupCh) { plan->error = "unsupported: more than 64 group channels"; return -1; }      // dev_modular.h: kModMaxGroupCh
  // inverse program (last transform first)
  F.mod_nops = 0;
  for (int i = ntr - 1; i >= 0; i--) {
    const Tr &t = trs[(size_t)i];
    if (t.id == 2) {
      for (int q = (int)t.sq.size() - 1; q >= 0; q--) {
        const Sq &p = t.sq[(size_t)q];
        const int b = p.begin_c, e = p.begin_c + p.num_c - 1;
        const int offset = p.in_place ? e + 1 : (int)L.size() - p.num_c;          // where this step's residual channels sit now
        if (e >= (int)L.size() || offset + p.num_c > (int)L.size() || offset <= e) { plan->error = "squeeze: bookkeeping"; return -1; }
        for (int c = b; c <= e; c++) {
          if (F.mod_nops >= kModMaxOps || nplanes >= kModMaxPlanes) { plan->error = "unsupported: too many squeeze steps"; return -1; }
          Ch &a = L[(size_t)c]; const Ch &r = L[(size_t)(offset + c - b)];
          const int o = F.mod_nops++;
          F.mod_op_kind[o] = p.horizontal ? 2 : 3;
          F.mod_op_a[o] = a.plane; F.mod_op_b[o] = r.plane; F.mod_op_d[o] = nplanes;
          F.mod_op_x[o] = a.w; F.mod_op_y[o] = a.h;
          if (p.horizontal) {
            if (r.h != a.h || (r.w != a.w && r.w != a.w - 1)) { plan->error = "squeeze: channel sizes"; return -1; }
            F.mod_op_e[o] = r.w; F.mod_op_c[o] = a.h;
            a.w += r.w; a.hs--;
          } else {
            if (r.w != a.w || (r.h != a.h && r.h != a.h - 1)) { plan->error = "squeeze: channel sizes"; return -1; }
            F.mod_op_e[o] = r.h; F.mod_op_c[o] = a.w;
            a.h += r.h; a.vs--;
          }
          a.plane = nplanes;
          F.mod_plane_off[nplanes++] = (uint32_t)off; off += (uint64_t)a.w * (uint64_t)a.h + 64;
          if (off > kPoolLimit) { plan->error = "unsupported: Modular image beyond 2^32 samples"; return -1; }
        }
        L.erase(L.begin() + offset, L.begin() + offset + p.num_c);
      }
      continue;
    }
    if (F.mod_nops >= kModMaxOps) { plan->error = "unsupported: too many transforms"; return -1; }
    int o = F.mod_nops++;
    if (t.id == 0) {
      F.mod_op_kind[o] = 0;
      F.mod_op_a[o] = L[(size_t)t.begin_c].plane; F.mod_op_b[o] = L[(size_t)t.begin_c + 1].plane; F.mod_op_c[o] = L[(size_t)t.begin_c + 2].plane;
      F.mod_op_x[o] = t.rct_type;
      const Ch &a = L[(size_t)t.begin_c], &b2 = L[(size_t)t.begin_c + 1], &c2 = L[(size_t)t.begin_c + 2];
      if (a.w != b2.w || a.w != c2.w || a.h != b2.h || a.h != c2.h) { plan->error = "RCT over channels of different size"; return -1; }
      F.mod_op_y[o] = a.w * a.h;
    } else {
      // inverse palette: the index channel (now at begin_c + 1, behind the palette meta channel) becomes num_c colour channels
      const Ch ix = L[(size_t)t.begin_c + 1];
      const bool deltas = t.nb_deltas > 0 || t.d_pred != 0;      // entries added to a prediction of the pixel (explicit ones, or implicit ones = indices below zero): raster order per channel
      F.mod_op_kind[o] = deltas ? 4 : 1;
      F.mod_op_a[o] = ix.plane; F.mod_op_b[o] = L[0].plane;
      F.mod_op_x[o] = t.nb_colours + t.nb_deltas; F.mod_op_e[o] = t.num_c;
      F.mod_op_f[o] = t.nb_deltas; F.mod_op_g[o] = t.d_pred; F.mod_op_h[o] = ix.w;
      F.mod_op_y[o] = deltas ? (F.mod_bits | (ix.h << 8)) : F.mod_bits;
      F.mod_op_c[o] = deltas ? t.num_c : ix.w * ix.h;
      if (deltas && ix.h >= (1 << 23)) { plan->error = "unsupported: palette channel height"; return -1; }
      const int fresh = deltas ? t.num_c : t.num_c - 1;       // without deltas colour 0 replaces the index in place
      if (nplanes + fresh > kModMaxPlanes) { plan->error = "unsupported: too many planes"; return -1; }
      F.mod_op_d[o] = nplanes;
      std::vector<Ch> outs;
      for (int c = 0; c < t.num_c; c++) {
        Ch oc = ix;
        if (deltas || c > 0) {
          oc.plane = nplanes; F.mod_plane_off[nplanes++] = (uint32_t)off; off += (uint64_t)ix.w * (uint64_t)ix.h + 64;
          if (off > kPoolLimit) { plan->error = "unsupported: Modular image beyond 2^32 samples"; return -1; }
        }
        outs.push_back(oc);
      }
      L.erase(L.begin() + t.begin_c + 1);
      L.insert(L.begin() + t.begin_c + 1, outs.begin(), outs.end());
      L.erase(L.begin());
    }
  }
  plan->mod_pool_ints = (size_t)off;
  if ((int)L.size() != ncol + m.num_extra) { plan->error = "modular channel bookkeeping"; return -1; }
  for (size_t i = 0; i < L.size(); i++) if (L[i].w != L0[i].w || L[i].h != L0[i].h) { plan->error = "modular channel bookkeeping (sizes)"; return -1; }
  for (int c = 0; c < 3; c++) F.mod_out[c] = vardct ? -1 : L[(size_t)(ncol == 1 ? 0 : c)].plane;
  F.mod_out[3] = -1; F.mod_alpha_bits = 8;
  F.alpha_up = 1; F.alpha_w = f.width; F.alpha_h = f.height;
  for (int i = 0; i < m.num_extra; i++) if (m.ec[i].type == 0) {
    F.mod_out[3] = L[(size_t)(ncol + i)].plane; F.mod_alpha_bits = m.ec[i].bits; F.mod_alpha_exp_bits = m.ec[i].float_sample ? m.ec[i].exp_bits : 0;
    if (F.mod_alpha_exp_bits ? (F.mod_alpha_bits > 32 || F.mod_alpha_exp_bits > 8 || F.mod_alpha_bits - F.mod_alpha_exp_bits - 1 < 1 || F.mod_alpha_bits - F.mod_alpha_exp_bits - 1 > 23) : F.mod_alpha_bits > 16) { plan->error = "unsupported: alpha sample format"; return -1; }
    F.alpha_up = f.ec_upsampling[i]; F.alpha_w = L[(size_t)(ncol + i)].w; F.alpha_h = L[(size_t)(ncol + i)].h;
    break;
  }
  return 0;
}

}  // namespace

int parse_basic_info(const uint8_t *data, size_t size, ImageInfo *info, std::string *error, std::vector<uint8_t> *icc) {
  uint8_t *cs; size_t csn; int owned;
  if (extract_codestream(data, size, &cs, &csn, &owned)) { if (error) *error = hx_last_error(); return -1; }
  hx_br br; hx_br_init(&br, cs, csn);
  img_meta m;
  int rc = read_image_header(&br, &m);
  if (rc) { if (error) *error = hx_last_error(); }
  else {
    fill_info(m, info);
    info->icc_size = 0;
    // only a non-XYB image reports its embedded profile (see plan_parse): the reference's libjxl outputs XYB + ICC images as sRGB
    if (m.pub.want_icc && !m.pub.xyb_encoded) {
      std::vector<uint8_t> tmp;
      if (read_icc_stream(&br, &tmp) == 0) { info->icc_size = (uint32_t)tmp.size(); if (icc) icc->swap(tmp); }
      else { rc = -1; if (error) *error = hx_last_error(); }
    } else if (m.pub.want_icc) {
      info->want_icc = 0; info->color_space = 0; info->white_point = 1; info->primaries = 1; info->transfer_function = 13; info->have_gamma = 0; info->rendering_intent = 1;
    } else if (!m.pub.have_gamma && m.pub.transfer_function == 8) {
      // an enum encoding with a transfer function the reference does not handle itself (linear light): it asks libjxl for the data profile — which
      // libjxl synthesises — and hands it to Little CMS (interop/JxlDecoding.cpp:126-141, JniDecoding.cpp:103-114).  Same here (host_icc_synth.inc)
      std::vector<uint8_t> tmp;
      if (icc_synth::synthesize(m.pub, m.wp_xy, m.prim_xy, &tmp)) { info->icc_size = (uint32_t)tmp.size(); if (icc) icc->swap(tmp); }
    }
  }
  if (owned) free(cs);
  return rc;
}

// One frame of the walk: its header, where its TOC starts and where the next frame header begins
struct FrameRec { frame_hdr f; size_t toc_bit = 0; size_t end_byte = 0; bool needed = false; bool blend = false; bool canvas_needed = false; int src_frame = -1; bool lf_needed = false; int visible_index = 0, nonvisible_index = 0; };

// TOC of the frame whose header ended at toc_bit: section table (logical order) and the byte where the frame's sections end
static int read_toc(const uint8_t *cs, size_t csn, const frame_hdr &f, size_t toc_bit, std::vector<DevSection> *secs, size_t *end_byte, std::string *error) {
  hx_br br; hx_br_init(&br, cs, csn); br.pos = toc_bit;
  const int nsec = (f.num_groups == 1 && f.num_passes == 1) ? 1 : 1 + f.num_lf_groups + 1 + f.num_groups * f.num_passes;
  std::vector<uint32_t> perm;
  if (hx_bool(&br)) {
    hx_ec tc;
    if (hx_ec_read_header(&tc, &br, 8)) { *error = "bad TOC permutation code"; return -1; }
    hx_ec_begin(&tc, &br, 0);
    perm.resize((size_t)nsec);
    int e = hx_read_permutation(&tc, &br, perm.data(), (uint32_t)nsec, 0);
    int ok = hx_ec_final_ok(&tc);
    hx_ec_free(&tc);
    if (e || !ok) { *error = "bad TOC permutation"; return -1; }
  }
  hx_align(&br);
  std::vector<uint32_t> sz((size_t)nsec);
  for (int i = 0; i < nsec; i++) sz[(size_t)i] = hx_u32(&br, 10, 0, 14, 1024, 22, 17408, 30, 4211712);
  hx_align(&br);
  size_t base = br.pos / 8, acc = 0;
  std::vector<size_t> phys((size_t)nsec);
  for (int i = 0; i < nsec; i++) { phys[(size_t)i] = base + acc; acc += sz[(size_t)i]; }
  if (base + acc > csn || br.err) { *error = "truncated file (TOC exceeds input)"; return -1; }
  *end_byte = base + acc;
  if (secs) {
    secs->resize((size_t)nsec);
    for (int i = 0; i < nsec; i++) {
      size_t src = perm.empty() ? (size_t)i : perm[(size_t)i];
      if (phys[src] > 0xFFFFFFFFull) { *error = "unsupported: codestream beyond 4 GiB"; return -1; }
      (*secs)[(size_t)i].off = (uint32_t)phys[src]; (*secs)[(size_t)i].size = sz[src];
    }
  }
  return 0;
}

static int build_frame(FramePlan *plan, Priv *pv, const FrameRec &rec, bool is_shown, uint32_t raw_w, uint32_t raw_h);

// The preview frame (image metadata have_preview): a frame of the PreviewHeader's size in front of the image's frames.  DecodeJpegXlOneShot never subscribes
// to it (interop/JxlDecoding.cpp:60-75), libjxl then walks over it: its header and TOC are read, its sections skipped.
static int skip_preview_frame(hx_br *br, const img_meta &m, const uint8_t *cs, size_t csn, std::string *error) {
  if (m.have_animation) { *error = "unsupported: preview frame of an animation"; return -1; }
  hx_align(br);
  frame_hdr pf;
  if (read_frame_header(br, &m, m.preview_w, m.preview_h, &pf)) { *error = hx_last_error(); return -1; }
  if (pf.frame_type != 0) { *error = "invalid: the preview is not a regular frame"; return -1; }
  size_t end_byte = 0;
  if (read_toc(cs, csn, pf, br->pos, nullptr, &end_byte, error)) return -1;
  hx_br_init(br, cs, csn);
  br->pos = end_byte * 8;
  return 0;
}

int parse_anim_info(const uint8_t *data, size_t size, std::vector<AnimFrame> *frames, AnimHeader *hdr, std::string *error) {
  uint8_t *cs0; size_t csn; int owned_flag;
  if (extract_codestream(data, size, &cs0, &csn, &owned_flag)) { *error = hx_last_error(); return -1; }
  std::vector<uint8_t> owned;
  if (owned_flag) { owned.assign(cs0, cs0 + csn); free(cs0); }
  const uint8_t *cs = owned_flag ? owned.data() : cs0;
  hx_br br; hx_br_init(&br, cs, csn);
  img_meta m;
  if (read_image_header(&br, &m)) { *error = hx_last_error(); return -1; }
  if (m.pub.want_icc && read_icc_stream(&br, nullptr)) { *error = hx_last_error(); return -1; }
  if (m.have_preview && skip_preview_frame(&br, m, cs, csn, error)) return -1;
  const uint32_t raw_w = m.orientation > 4 ? m.pub.ysize : m.pub.xsize, raw_h = m.orientation > 4 ? m.pub.xsize : m.pub.ysize;
  hdr->have_animation = (uint32_t)m.have_animation; hdr->tps_numerator = m.have_animation ? m.tps_num : 0; hdr->tps_denominator = m.have_animation ? m.tps_den : 0;
  hdr->num_loops = m.have_animation ? m.num_loops : 0; hdr->have_timecodes = m.have_animation ? (uint32_t)m.have_timecodes : 0;
  frames->clear();
  int shown = 0;
  for (size_t n = 0;; n++) {
    hx_align(&br);
    frame_hdr f;
    if (read_frame_header(&br, &m, raw_w, raw_h, &f)) { *error = hx_last_error(); return -1; }
    size_t end_byte = 0;
    if (read_toc(cs, csn, f, br.pos, nullptr, &end_byte, error)) return -1;
    if (f.frame_type == 0 || f.frame_type == 3) {
      AnimFrame a;
      a.ticks = (uint32_t)f.duration; a.ms = 0; a.is_last = f.is_last;
      if (m.have_animation && m.tps_num) a.ms = (int)roundf(1000.0f * (float)f.duration * (float)m.tps_den / (float)m.tps_num);
      a.coalesced = (f.is_last || f.duration > 0) ? shown++ : -1;
      frames->push_back(a);
    }
    if (f.is_last) break;
    if (n > (1 << 20)) { *error = "too many frames"; return -1; }      // (a walk over headers and TOCs: the reference has no limit; this one only bounds a corrupt file's loop)
    hx_br_init(&br, cs, csn);
    br.pos = end_byte * 8;
  }
  return 0;
}

int plan_parse(const uint8_t *data, size_t size, FramePlan *plan, int target_frame) {
  plan->error.clear();
  plan->tables.clear();
  plan->refs.clear();
  uint8_t *cs; size_t csn; int owned;
  if (extract_codestream(data, size, &cs, &csn, &owned)) { plan->error = hx_last_error(); return -1; }
  if (owned) { plan->cs_owned.assign(cs, cs + csn); free(cs); plan->cs = plan->cs_owned.data(); }
  else plan->cs = cs;                                   /* the bare codestream, or the one codestream box of a container, in place */
  plan->cs_size = csn;
  std::shared_ptr<Priv> pvs = std::make_shared<Priv>();
  plan->priv = pvs;
  Priv *pv = pvs.get();
  hx_br br; hx_br_init(&br, plan->cs, csn);
  img_meta &m = pv->m;
  if (read_image_header(&br, &m)) { plan->error = hx_last_error(); return -1; }
  fill_info(m, &plan->info);
  if (m.pub.want_icc) {
    // The reference never hands libjxl a CMS (interop/JxlDecoding.cpp:46-75), so an XYB image with an embedded profile comes out in sRGB with
    // an sRGB enum profile (probed on the reference's own assets: jxl_icc_12.bit.jxl, wide_gamut.jxl, ...): the stream is skipped here.
    // A non-XYB (Modular) image keeps its samples in the profile's space; the bytes are available through jxlamd_get_icc.
    std::vector<uint8_t> icc;
    if (read_icc_stream(&br, m.pub.xyb_encoded ? nullptr : &icc)) { plan->error = hx_last_error(); return -1; }
    if (m.pub.xyb_encoded) { m.pub.want_icc = 0; m.pub.color_space = 0; m.pub.white_point = 1; m.pub.primaries = 1; m.pub.transfer_function = 13; m.pub.have_gamma = 0; m.pub.rendering_intent = 1; }
    fill_info(m, &plan->info);
    plan->info.icc_size = (uint32_t)icc.size();
  } else if (!m.pub.have_gamma && m.pub.transfer_function == 8) {
    std::vector<uint8_t> tmp;                                  // linear-light enum encoding: the synthesised data profile (parse_basic_info)
    if (icc_synth::synthesize(m.pub, m.wp_xy, m.prim_xy, &tmp)) plan->info.icc_size = (uint32_t)tmp.size();
  }
  if (m.have_preview && skip_preview_frame(&br, m, plan->cs, csn, &plan->error)) return -1;
  const uint32_t raw_w = m.orientation > 4 ? m.pub.ysize : m.pub.xsize, raw_h = m.orientation > 4 ? m.pub.xsize : m.pub.ysize;
  // ---- the frame walk: every frame's header and TOC (its sections are skipped by their sizes), up to the frame marked is_last
  std::vector<FrameRec> recs;
  for (;;) {
    hx_align(&br);
    FrameRec r;
    if (read_frame_header(&br, &m, raw_w, raw_h, &r.f)) { plan->error = hx_last_error(); return -1; }
    const frame_hdr &f = r.f;
    if (getenv("JXLAMD_PARSE_TRACE"))
      fprintf(stderr, "frame %zu: type %d encoding %d is_last %d crop %d (%d,%d %dx%d) blend %d (all replace: %d) src %d duration %d save_ref %d before_ct %d flags %llu upsampling %d gab %d epf %d\n", recs.size(), f.frame_type,
              f.encoding, f.is_last, f.have_crop, f.x0, f.y0, f.width, f.height, f.blend_mode, !f.blend_not_replace, f.blend_source, f.duration, f.save_as_ref, f.save_before_ct, (unsigned long long)f.flags, f.upsampling, f.gab, f.epf_iters);
    r.toc_bit = br.pos;
    if (read_toc(plan->cs, csn, f, r.toc_bit, nullptr, &r.end_byte, &plan->error)) return -1;
    recs.push_back(r);
    if (f.is_last) break;
    if (recs.size() > (1u << 20)) { plan->error = "too many frames"; return -1; }
    hx_br_init(&br, plan->cs, csn);
    br.pos = r.end_byte * 8;
  }
  {   // libjxl's frame counters (they seed the noise generator), advanced BEFORE a frame is decoded: shown frames up to and including this one / frames
      // since the last shown one (established on the reference binary: the first frame of a still image is seeded with (1, 0))
    int vis = 0, nonvis = 0;
    for (auto &r : recs) {
      const bool shown = (r.f.frame_type == 0 || r.f.frame_type == 3) && (r.f.is_last || r.f.duration > 0);
      if (shown) { vis++; nonvis = 0; } else nonvis++;
      r.visible_index = vis; r.nonvisible_index = nonvis;
    }
  }
  // ---- which frame is shown, and which earlier frames does it need?  The reference keeps what the LAST coalesced frame shows (interop/JxlDecoding.cpp:
  // 164-166); its animated decoder asks for coalesced frame i (JxlAnimatedDecoder.cpp:28-144).  A coalesced frame is a regular frame of non-zero duration
  // (or the last one) laid over the canvas its BlendingInfo names: the content of a reference slot, i.e. an earlier frame's blended canvas.  Needed are
  // therefore, recursively: the occupants of the slots a needed frame's patches may draw on, and the occupant of the slot a needed frame is blended over.
  size_t last = recs.size() - 1;
  if (target_frame >= 0) {
    int seen = 0; bool found = false;
    for (size_t i = 0; i < recs.size(); i++) {
      const frame_hdr &f = recs[i].f;
      if ((f.frame_type == 0 || f.frame_type == 3) && (f.is_last || f.duration > 0)) { if (seen == target_frame) { last = i; found = true; break; } seen++; }
    }
    if (!found) { plan->error = "frame index beyond the frames of the file"; return -1; }
  }
  {
    const frame_hdr &f = recs[last].f;
    if (f.frame_type != 0 && f.frame_type != 3) { plan->error = "unsupported: non-regular last frame"; return -1; }
  }
  const int alpha_ec = [&] { for (int i = 0; i < m.num_extra; i++) if (m.ec[i].type == 0) return i; return -1; }();
  // does a regular frame look at its blend source?  (colour and the alpha channel count: the other extra channels are not part of the output)
  const auto full_frame = [&](const frame_hdr &f) { return !f.have_crop || (f.x0 == 0 && f.y0 == 0 && f.width == (int)raw_w && f.height == (int)raw_h); };
  const auto uses_canvas = [&](const frame_hdr &f) {
    if (f.frame_type != 0 && f.frame_type != 3) return false;
    return !full_frame(f) || f.bl_mode[0] != 0 || (alpha_ec >= 0 && f.bl_mode[1 + alpha_ec] != 0);
  };
  recs[last].needed = true;
  std::vector<int> occupant_at((recs.size()) * 4, -1);      // [frame][slot]: which frame sits in the slot when this frame is decoded
  {
    int occ[4] = {-1, -1, -1, -1};
    for (size_t i = 0; i < recs.size(); i++) {
      for (int k = 0; k < 4; k++) occupant_at[i * 4 + (size_t)k] = occ[k];
      const frame_hdr &f = recs[i].f;
      // FrameHeader::CanBeReferenced: a frame of non-zero duration only stays around when it names a slot other than 0
      if (!f.is_last && f.frame_type != 1 && (f.duration == 0 || f.save_as_ref != 0)) occ[f.save_as_ref & 3] = (int)i;
    }
  }
  for (size_t i = last + 1; i-- > 0;) {
    if (!recs[i].needed) continue;
    const frame_hdr &f = recs[i].f;
    // patches may name any slot that holds a frame as it was before the colour transform (libjxl's encoder keeps its patch frames that way: kReferenceOnly,
    // save_before_color_transform); a slot with a frame of the animation itself, saved after it, is not a patch source — the dictionary parser refuses a patch that names one
    if (f.flags & 2) for (int k = 0; k < 4; k++) {
      const int occ = occupant_at[i * 4 + (size_t)k];
      if (occ >= 0 && (recs[(size_t)occ].f.save_before_ct || recs[(size_t)occ].f.frame_type == 2 || !m.pub.xyb_encoded)) recs[(size_t)occ].needed = true;
    }
    if (f.flags & 32) {
      // kUseDcFrame (progressive_dc): the frame's LF image is the latest LF frame of the next level before it (libjxl's dc_frames[lf_level]) instead of LF
      // coefficients; an LF frame may itself be a VarDCT frame with an LF frame of its own (progressive_dc = 2: level 1 VarDCT over level 2 Modular)
      const int want = (f.frame_type == 1 ? f.lf_level : 0) + 1;
      if (want > 4) { plan->error = "LF frame level out of range"; return -1; }
      int lf = -1;
      for (size_t j = i; j-- > 0;) if (recs[j].f.frame_type == 1 && recs[j].f.lf_level == want) { lf = (int)j; break; }
      if (lf < 0) { plan->error = "frame refers to an LF frame the file does not have"; return -1; }
      recs[(size_t)lf].needed = true; recs[(size_t)lf].lf_needed = true;
    }
    if (uses_canvas(f)) {
      const int src = f.bl_source[0] & 3;
      if (alpha_ec >= 0 && (f.bl_source[1 + alpha_ec] & 3) != src && (f.bl_mode[1 + alpha_ec] != 0 || !full_frame(f))) { plan->error = "unsupported: colour and alpha blended over different reference slots"; return -1; }
      const int occ = occupant_at[i * 4 + (size_t)src];
      recs[i].src_frame = occ;
      // over an empty slot a replacing frame shows the cleared canvas: the writer's cropped-frame path; everything else is laid over a canvas
      if (occ >= 0 || f.bl_mode[0] != 0 || (alpha_ec >= 0 && f.bl_mode[1 + alpha_ec] != 0)) recs[i].blend = true;
      if (occ >= 0) {
        FrameRec &o = recs[(size_t)occ];
        if (o.f.frame_type != 0 && o.f.frame_type != 3) { plan->error = "unsupported: blending over a reference-only frame"; return -1; }
        // (libjxl keeps such a frame BEFORE blending: one that is itself laid over a canvas would hand on its blended canvas here — refused)
        if (o.f.save_before_ct && (m.pub.xyb_encoded || !full_frame(o.f) || uses_canvas(o.f))) { plan->error = "unsupported: blending over a frame saved before the colour transform"; return -1; }
        o.needed = true; o.canvas_needed = true; o.blend = true;
      }
    }
  }
  // ---- the needed earlier frames first, in file order (each its own FramePlan over the same codestream bytes), then the shown frame
  int slot_w[4] = {0, 0, 0, 0}, slot_h[4] = {0, 0, 0, 0}, lf_w[6] = {0, 0, 0, 0, 0, 0}, lf_h[6] = {0, 0, 0, 0, 0, 0};      // lf_w[level]: the latest LF frame of that level
  const auto blend_checks = [&](const FrameRec &r) -> bool {
    const frame_hdr &f = r.f;
    if (!r.blend) return true;
    for (int ch : {0, alpha_ec >= 0 ? 1 + alpha_ec : 0}) {
      if ((f.bl_mode[ch] == 2 || f.bl_mode[ch] == 3) && m.num_extra > 0 && f.bl_alpha[ch] != alpha_ec) { plan->error = "unsupported: blending weighted by an extra channel that is not the alpha channel"; return false; }
    }
    if (m.orientation != 1 && false) return false;
    return true;
  };
  for (size_t i = 0; i < last; i++) {
    if (!recs[i].needed) continue;
    const frame_hdr &f = recs[i].f;
    std::shared_ptr<FramePlan> sub = std::make_shared<FramePlan>();
    sub->info = plan->info; sub->cs = plan->cs; sub->cs_size = plan->cs_size;
    std::shared_ptr<Priv> spv = std::make_shared<Priv>();
    sub->priv = spv;
    spv->m = m; spv->f = f;
    memcpy(spv->ref_w, slot_w, sizeof(slot_w)); memcpy(spv->ref_h, slot_h, sizeof(slot_h)); 
    for (int k = 0; k < 4; k++) { const int occ = occupant_at[i * 4 + (size_t)k]; spv->ref_patch_ok[k] = occ >= 0 && recs[(size_t)occ].needed && !recs[(size_t)occ].canvas_needed; }
    { const int lv = (f.frame_type == 1 ? f.lf_level : 0) + 1; spv->lf_w = lf_w[lv]; spv->lf_h = lf_h[lv]; }
    if (f.frame_type == 1 && (f.lf_level < 1 || f.lf_level > 4)) { plan->error = "LF frame level out of range"; return -1; }
    if (!recs[i].canvas_needed && f.frame_type != 1) {
      // a frame kept for a patch dictionary: stored as it is, before the colour transform
      if (f.frame_type != 2 && (recs[i].blend || !full_frame(f))) { plan->error = "unsupported: blended / cropped regular frame used as a patch source"; return -1; }
      if (!f.save_before_ct && m.pub.xyb_encoded) { plan->error = "unsupported: reference frame saved after the colour transform"; return -1; }
    }
    if (!blend_checks(recs[i])) return -1;
    spv->blend = recs[i].blend; spv->save_canvas = recs[i].canvas_needed; spv->has_src = recs[i].src_frame >= 0; spv->alpha_ec = alpha_ec;
    if (build_frame(sub.get(), spv.get(), recs[i], /*is_shown=*/false, raw_w, raw_h)) { plan->error = sub->error; return -1; }
    if (plan->refs.size() >= 2048) { plan->error = "unsupported: more than 2048 frames to decode for one shown frame"; return -1; }      // (a blend chain through every frame of a long animation)
    sub->save_slot = f.frame_type == 1 ? 3 + f.lf_level : (f.save_as_ref & 3);          // slots 4..7: the LF frames of level 1..4 (the LF image of the next frame of the level below that asks for one)
    sub->save_canvas = recs[i].canvas_needed;
    plan->refs.push_back(sub);
    if (f.frame_type == 1) { lf_w[f.lf_level] = f.coded_width; lf_h[f.lf_level] = f.coded_height; continue; }
    if (recs[i].canvas_needed) { slot_w[f.save_as_ref & 3] = (int)raw_w; slot_h[f.save_as_ref & 3] = (int)raw_h; }
    else { slot_w[f.save_as_ref & 3] = f.width; slot_h[f.save_as_ref & 3] = f.height; }
  }
  if (!blend_checks(recs[last])) return -1;
  pv->lf_w = lf_w[1]; pv->lf_h = lf_h[1];
  pv->blend = recs[last].blend; pv->save_canvas = false; pv->has_src = recs[last].src_frame >= 0; pv->alpha_ec = alpha_ec;
  pv->f = recs[last].f;
  memcpy(pv->ref_w, slot_w, sizeof(slot_w)); memcpy(pv->ref_h, slot_h, sizeof(slot_h));
  for (int k = 0; k < 4; k++) { const int occ = occupant_at[last * 4 + (size_t)k]; pv->ref_patch_ok[k] = occ >= 0 && recs[(size_t)occ].needed && !recs[(size_t)occ].canvas_needed; }
  return build_frame(plan, pv, recs[last], /*is_shown=*/true, raw_w, raw_h);
}

// Everything a decoded frame needs from the host: checks of what the device path covers, TOC, LfGlobal (patch dictionary, quantiser, block
// context map, colour correlation, global MA tree, GlobalModular header), HfGlobal, the DevFrame parameter block.
static int build_frame(FramePlan *plan, Priv *pv, const FrameRec &rec, bool is_shown, uint32_t raw_w, uint32_t raw_h) {
  img_meta &m = pv->m;
  frame_hdr &f = pv->f;
  const size_t csn = plan->cs_size;
  std::vector<DevSection> &secs = pv->secs;
  // a frame may exceed the canvas (cropped frames, reference frames of a patch dictionary), but one that is orders of magnitude larger than the
  // image it belongs to only sizes allocations: refused
  if ((int64_t)f.coded_width * (int64_t)f.coded_height > 64 * (int64_t)raw_w * (int64_t)raw_h + ((int64_t)1 << 24)) { plan->error = "unsupported: frame far larger than the image"; return -1; }
  if (f.encoding == 0 && !m.pub.xyb_encoded) {
    // a VarDCT frame of an image that is not XYB: a recompressed JPEG (YCbCr, chroma possibly subsampled, RAW dequant matrices).  Decoded like any
    // VarDCT frame up to the planes, which then hold the image's own samples (dev_compose.h: chroma upsampling, YCbCr -> RGB in the writer)
    if (m.num_extra) { plan->error = "unsupported: extra channels on a VarDCT frame that is not XYB"; return -1; }
    if (m.pub.num_color_channels != 3 && !f.do_ycbcr) { plan->error = "unsupported: grey VarDCT frame without XYB / YCbCr"; return -1; }
    if (f.upsampling != 1 || (f.flags & 2) || !is_shown) { plan->error = "unsupported: upsampling / patches / reference use of a VarDCT frame that is not XYB"; return -1; }
    if (f.subsampled && (f.gab || f.epf_iters)) { plan->error = "unsupported: loop filters on a chroma-subsampled frame"; return -1; }
    if (f.subsampled && !(f.flags & 128)) { plan->error = "unsupported: adaptive LF smoothing of a chroma-subsampled frame"; return -1; }
    if (f.subsampled && f.num_passes != 1) { plan->error = "unsupported: multi-pass chroma-subsampled frame"; return -1; }
  }
  if (f.encoding == 1 && !m.pub.xyb_encoded && (f.gab || f.epf_iters)) { f.gab = 0; f.epf_iters = 0; }   // loop filters only apply to XYB frames
  for (int i = 0; i < m.num_extra; i++) {
    // an extra channel is coded at 1 / ec_upsampling of the full size, never finer than the colour channels
    if (f.ec_upsampling[i] < f.upsampling) { plan->error = "extra channel upsampling below the frame's"; return -1; }
  }
  if (f.upsampling != 1) {
    // an upsampled frame (what the reference's encoder writes from distance 10 up, i.e. its quality <= 12: interop/JxlEncoding.cpp:38-46) is coded at
    // ceil(size / upsampling) and enlarged after the patches (dev_compose.h)
    // (round 6: also frames that are not XYB — `cjxl --resampling=2 -d 0` through the reference's encoder writes a Modular frame of the image's own samples at half size —:
    // the integer planes become [0, 1] floats (k_mod_to_planes), are enlarged like any other, and leave through the plain writer)
    if (!m.pub.xyb_encoded && f.encoding == 0) { plan->error = "unsupported: upsampling of a VarDCT frame that is not XYB"; return -1; }
    if (!is_shown && !pv->blend) { plan->error = "unsupported: upsampled reference frame"; return -1; }      // (a frame kept as a canvas goes through the blend kernel at its full resolution; one kept as a patch source does not)
  }
  if (is_shown && !pv->blend && f.have_crop && (f.x0 || f.y0 || f.width != (int)raw_w || f.height != (int)raw_h)) {
    // a frame that does not cover the canvas shows the blend source's canvas around it: the cleared canvas (plan_parse has checked that no
    // earlier frame was saved into that slot)
    if (f.blend_not_replace) { plan->error = "unsupported: cropped frame blended with a reference frame"; return -1; }
    if (f.width < 1 || f.height < 1) { plan->error = "empty frame"; return -1; }
    plan->cropped = true;
  }
  if (!is_shown && (f.width < 1 || f.height < 1)) { plan->error = "empty frame"; return -1; }
  if (f.do_ycbcr && f.encoding == 1) { plan->error = "unsupported: YCbCr Modular frame"; return -1; }
  if ((f.flags & 16) && (f.frame_type == 1 || !m.pub.xyb_encoded)) { plan->error = "unsupported: splines on an LF frame / an image that is not XYB"; return -1; }
  if (f.flags & 1) {
    if (f.encoding != 0 || !m.pub.xyb_encoded) { plan->error = "unsupported: noise on a frame that is not a VarDCT XYB frame"; return -1; }
    if (!is_shown) { plan->error = "unsupported: noise on a reference frame"; return -1; }
  }
  if (f.flags & 32) {
    if (f.encoding != 0) { plan->error = "unsupported: Modular frame with an LF frame"; return -1; }
    if (f.subsampled) { plan->error = "unsupported: chroma-subsampled frame with an LF frame"; return -1; }
    if (pv->lf_w != (f.coded_width + 7) / 8 || pv->lf_h != (f.coded_height + 7) / 8) { plan->error = "LF frame does not match the frame it serves"; return -1; }
  }
  if (f.frame_type == 1 && !m.pub.xyb_encoded) { plan->error = "unsupported: LF frame of an image that is not XYB"; return -1; }      // Modular, or VarDCT over an LF frame of its own (progressive_dc = 2)      // (its extra channels, coded at an eighth too, are decoded and not used: the main frame carries its own)
  if (f.group_dim != 256 && f.encoding != 1) { plan->error = "unsupported: group size"; return -1; }      // VarDCT frames: 256 (libjxl never writes another); Modular frames: 128 .. 1024
  const int nsec = (f.num_groups == 1 && f.num_passes == 1) ? 1 : 1 + f.num_lf_groups + 1 + f.num_groups * f.num_passes;
  { size_t end_byte = 0; if (read_toc(plan->cs, csn, f, rec.toc_bit, &secs, &end_byte, &plan->error)) return -1; }
  // ---- DevFrame
  DevFrame &F = pv->F;
  memset(&F, 0, sizeof(F));
  F.width = f.coded_width; F.height = f.coded_height;          // (an upsampled frame: the coded size; its pixels are full_w x full_h)
  F.upsampling = f.upsampling; F.full_w = f.width; F.full_h = f.height;
  F.xb = (F.width + 7) / 8; F.yb = (F.height + 7) / 8;
  if (f.encoding == 0 && f.subsampled) {          // the block grid is padded to whole MCUs: ceil(size / (8 << max shift)) << max shift
    int mh = 0, mv = 0;
    for (int c = 0; c < 3; c++) { mh = std::max(mh, f.hshift[c]); mv = std::max(mv, f.vshift[c]); }
    F.xb = ((F.width + (8 << mh) - 1) / (8 << mh)) << mh; F.yb = ((F.height + (8 << mv) - 1) / (8 << mv)) << mv;
    F.subsampled = 1;
    for (int c = 0; c < 3; c++) { F.hshift[c] = f.hshift[c]; F.vshift[c] = f.vshift[c]; }
  }
  F.pw = F.xb * 8; F.ph = F.yb * 8;
  F.not_xyb = (f.encoding == 0 && !m.pub.xyb_encoded) ? (f.do_ycbcr ? 2 : 1) : 0;
  F.tiles_x = (F.xb + 7) / 8; F.tiles_y = (F.yb + 7) / 8;
  F.xgroups = f.xgroups; F.ygroups = f.ygroups; F.num_groups = f.num_groups;
  F.xlfg = f.xlfg; F.ylfg = f.ylfg; F.num_lf_groups = f.num_lf_groups;
  F.num_passes = f.num_passes;
  for (int i = 0; i < 12; i++) F.pass_shift[i] = i < f.num_passes - 1 ? f.pass_shift[i] : 0;
  {
    // which Modular channels travel in which pass (libjxl: Passes::GetDownsamplingBracket): pass p takes the shifts [min, max]; a pass that completes a
    // downsampling level (last_pass) lowers min to that level's shift, the last pass goes down to 0, and the next pass starts just below
    int max_shift = 2, min_shift = 3;
    for (int p = 0; p < kMaxPasses; p++) { F.pass_min_shift[p] = 3; F.pass_max_shift[p] = 2; }
    for (int p = 0; p < f.num_passes && p < kMaxPasses; p++) {
      for (int j = 0; j < f.num_ds; j++) if (p == f.ds_last[j]) min_shift = f.ds[j] == 8 ? 3 : f.ds[j] == 4 ? 2 : f.ds[j] == 2 ? 1 : 0;
      if (p == f.num_passes - 1) min_shift = 0;
      F.pass_min_shift[p] = min_shift; F.pass_max_shift[p] = max_shift;
      max_shift = min_shift - 1;
    }
  }
  F.nsec = nsec;
  F.cs_size = (uint32_t)csn;
  plan->tables.assign(((sizeof(DevFrame) + 15) / 16) * 16, 0);
  Blob blob(plan->tables);
  F.sec_off = blob.append(secs.data(), secs.size() * sizeof(DevSection));
  if (m.custom_upsampling) {      // the image's own upsampling weights (metadata), expanded like the defaults of the static tables
    const float *cw[3] = {m.up_w2, m.up_w4, m.up_w8};
    for (int t = 0; t < 3; t++) if (m.custom_upsampling & (1 << t)) { const std::vector<float> k = expand_upsampling_kernels(cw[t], t); F.ups_custom_off[t] = blob.append(k.data(), k.size() * 4); }
  }
  // ---- LfGlobal (section 0)
  hx_br sb; hx_br_init(&sb, plan->cs + secs[0].off, nsec == 1 ? csn - secs[0].off : secs[0].size);
  if ((f.flags & 2) && parse_patches(plan, pv, &sb, blob)) return -1;
  std::vector<QSpline> qsplines; int32_t spline_quant_adjust = 0;
  if ((f.flags & 16) && parse_splines(plan, &sb, (uint64_t)f.coded_width * (uint64_t)f.coded_height, &qsplines, &spline_quant_adjust)) return -1;
  if (f.flags & 1) for (int i = 0; i < 8; i++) F.noise_lut[i] = (float)hx_bits(&sb, 10) * (1.0f / 1024);      // NoiseParameters: eight points of the strength curve
  float lf_dequant[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  if (!hx_bool(&sb)) for (int c = 0; c < 3; c++) lf_dequant[c] = hx_f16(&sb) * (1.0f / 128);
  uint32_t global_scale = 1, quant_lf = 1;
  uint32_t color_factor = 84; float base_x = 0.0f, base_b = 1.0f; int ytox_dc = 0, ytob_dc = 0;
  if (f.encoding == 0) {
  global_scale = hx_u32(&sb, 11, 1, 11, 2049, 12, 4097, 16, 8193);
  quant_lf = hx_u32(&sb, -1, 16, 5, 1, 8, 1, 16, 1);
  std::vector<uint8_t> bctx;
  if (hx_bool(&sb)) { bctx.assign(kDefaultBlockCtxMap, kDefaultBlockCtxMap + 39); F.num_bctx = 15; }
  else {
    int nlf = 1;
    for (int c = 0; c < 3; c++) {
      F.nb_lf_thr[c] = (int)hx_bits(&sb, 4);
      for (int i = 0; i < F.nb_lf_thr[c]; i++) F.lf_thr[c][i] = hx_unpack_signed(hx_u32(&sb, 4, 0, 8, 16, 16, 272, 32, 65808));
      nlf *= F.nb_lf_thr[c] + 1;
    }
    F.nb_qf_thr = (int)hx_bits(&sb, 4);
    for (int i = 0; i < F.nb_qf_thr; i++) F.qf_thr[i] = 1 + hx_u32(&sb, 2, 0, 3, 4, 5, 12, 8, 44);
    size_t n = (size_t)39 * (size_t)(F.nb_qf_thr + 1) * (size_t)nlf;
    if (n > 39 * 64 || nlf > 64) { plan->error = "block ctx map too large"; return -1; }
    bctx.assign(n, 0);
    if (hx__read_ctx_map(&sb, bctx.data(), (int)n, &F.num_bctx)) { plan->error = "bad block ctx map"; return -1; }
  }
  F.bctx_map_off = blob.append(bctx.data(), bctx.size());
  if (!hx_bool(&sb)) {
    color_factor = hx_u32(&sb, -1, 84, -1, 256, 8, 2, 16, 258);
    base_x = hx_f16(&sb); base_b = hx_f16(&sb);
    ytox_dc = (int)hx_bits(&sb, 8) - 128; ytob_dc = (int)hx_bits(&sb, 8) - 128;
  }
  }
  if (!qsplines.empty() && build_splines(plan, blob, F, qsplines, spline_quant_adjust, base_x, base_b, f.coded_width, f.coded_height)) return -1;
  // GlobalModular: MA tree flag (+ tree); no channels on this path (no extra channels)
  hx_tree tree; memset(&tree, 0, sizeof(tree));
  const bool have_tree = hx_bool(&sb);
  pv->tree_bit = have_tree ? (int64_t)sb.pos : -1;
  if (have_tree && hx_tree_read(&tree, &sb)) { hx_tree_free(&tree); plan->error = std::string("global MA tree: ") + hx_last_error(); return -1; }
  if (sb.err) { hx_tree_free(&tree); plan->error = "truncated LfGlobal"; return -1; }
  if (have_tree) {
    std::vector<DevTreeNode> nodes((size_t)tree.count);
    for (int i = 0; i < tree.count; i++) {
      DevTreeNode &d = nodes[(size_t)i]; const hx_tnode &s = tree.n[i];
      memset(&d, 0, sizeof(d));
      if (s.prop < 0) { d.prop = -1; d.splitval = s.ctx; d.lchild = s.predictor; d.rchild = (int32_t)s.multiplier; d.offset = (int32_t)s.offset; }
      else { d.prop = s.prop; d.splitval = s.splitval; d.lchild = s.lchild; d.rchild = s.rchild; }
    }
    F.tree_count = tree.count;
    F.tree_off = blob.append(nodes.data(), nodes.size() * sizeof(DevTreeNode));
    int rc = pack_ec(tree.code, blob, &F.tree_ec, &plan->error, /*allow_lz77=*/f.encoding == 1);      // Modular-encoded frames: the serial walker copies
    hx_tree_free(&tree);
    if (rc) return -1;
  } else F.tree_count = 0;     // streaming-encoded frames: every LfGroup stream carries its own tree (parsed on the device)
  plan->lf_global_end_bit = (uint32_t)sb.pos;
  F.single_lf_bit = (uint32_t)sb.pos;
  if (f.encoding == 1 && parse_modular_global(plan, pv, &sb)) return -1;
  if (f.encoding == 0 && m.num_extra) {
    if (parse_modular_global(plan, pv, &sb, /*vardct=*/true)) return -1;
    plan->has_ec = true;
  }
  // quantiser-derived constants
  float inv_quant_dc = 65536.0f / ((float)global_scale * (float)quant_lf);
  for (int c = 0; c < 3; c++) F.lf_fac[c] = lf_dequant[c] * inv_quant_dc;
  F.cfl_dc_x = base_x + (float)ytox_dc / (float)color_factor;
  F.cfl_dc_b = base_b + (float)ytob_dc / (float)color_factor;
  F.inv_global_scale = 65536.0f / (float)global_scale;
  F.quant_scale = (float)global_scale / 65536.0f;
  F.dm[0] = powf(1.0f / 1.25f, (float)f.x_qm - 2.0f); F.dm[1] = 1.0f; F.dm[2] = powf(1.0f / 1.25f, (float)f.b_qm - 2.0f);
  F.base_x = base_x; F.base_b = base_b; F.inv_color_factor = 1.0f / (float)color_factor;
  memcpy(F.quant_bias, m.quant_bias, sizeof(F.quant_bias));
  F.skip_lf_smoothing = (f.flags & (128 | 32)) ? 1 : 0;      // (a frame that takes its LF image from an LF frame is not smoothed either)
  F.use_lf_frame = (f.flags & 32) ? 1 : 0; F.lf_frame_w = pv->lf_w; F.lf_frame_h = pv->lf_h;
  F.lf_frame_slot = 4 + (f.frame_type == 1 ? f.lf_level : 0);
  F.modular_16bit = m.modular_16;
  F.gab = f.gab; memcpy(F.gab_w, f.gab_w, sizeof(F.gab_w));
  F.epf_iters = f.epf_iters; memcpy(F.epf_sharp, f.epf_sharp, sizeof(F.epf_sharp)); memcpy(F.epf_chscale, f.epf_chscale, sizeof(F.epf_chscale));
  F.epf_quant_mul = f.epf_quant_mul; F.epf_pass0 = f.epf_pass0; F.epf_pass2 = f.epf_pass2; F.epf_border_sad = f.epf_border_sad;
  F.epf_sigma_modular = f.epf_sigma_modular;
  F.xyb_modular = (f.encoding == 1 && m.pub.xyb_encoded) ? 1 : 0;
  for (int c = 0; c < 3; c++) F.mod_xyb_fac[c] = lf_dequant[c];
  // colour: opsin inverse scaled to display-relative linear, then to the data profile's primaries
  {
    double T[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    static const double srgb[8] = {0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204, 0.3127, 0.3290};
    double dst[8]; memcpy(dst, srgb, sizeof(dst));      // r, g, b, white
    bool custom = false;
    if (m.pub.primaries == 9) { const double t[6] = {0.708, 0.292, 0.170, 0.797, 0.131, 0.046}; memcpy(dst, t, sizeof(t)); }
    else if (m.pub.primaries == 11) { const double t[6] = {0.680, 0.320, 0.265, 0.690, 0.150, 0.060}; memcpy(dst, t, sizeof(t)); }
    else if (m.pub.primaries == 2) { for (int k = 0; k < 6; k++) dst[k] = (double)m.prim_xy[k]; custom = true; }
    else if (m.pub.primaries != 1) { plan->error = "unsupported: primaries"; return -1; }
    if (m.pub.white_point == 2) { dst[6] = (double)m.wp_xy[0]; dst[7] = (double)m.wp_xy[1]; custom = true; }
    else if (m.pub.white_point == 10) { dst[6] = dst[7] = 1.0 / 3; custom = true; }
    else if (m.pub.white_point == 11) { dst[6] = 0.314; dst[7] = 0.351; custom = true; }
    else if (m.pub.white_point != 1) { plan->error = "unsupported: white point"; return -1; }
    for (int k = 0; k < 4; k++) if (!(dst[2 * k + 1] > 1e-6)) { plan->error = "bad chromaticities"; return -1; }
    // A grey target (ColourEncoding.colour_space = kGrey; the reference's own encoder writes it for mono bitmaps, interop/JxlEncoding.cpp:66-68,103-106): libjxl keeps
    // the sRGB inverse matrix whatever the white point says and multiplies it from the left by three identical rows of the sRGB luminances, so the three
    // channels of the pipeline carry the same value and the 4-channel output the reference asks for (interop/JxlDecoding.cpp:63) has R = G = B on every sample.
    const bool grey_target = m.grey_target != 0;
    if ((m.pub.primaries != 1 || custom) && !grey_target) {
      // linear sRGB -> the target's linear RGB.  Same white point on both sides: the two RGB -> XYZ matrices.  Another white point (custom, E, DCI): both
      // sides go to XYZ-D50 through the Bradford adaptation of their white point, as libjxl's output stage does (PrimariesToXYZD50 on the image's encoding and
      // on sRGB); for a D65 target the two adaptations cancel, and the enum cases keep the arithmetic they had
      double A[9], Bm[9], Bi[9];
      primaries_to_xyz(srgb, A); primaries_to_xyz(dst, Bm);
      if (dst[6] != srgb[6] || dst[7] != srgb[7]) {
        const auto bradford_to_d50 = [](double wx, double wy, double out[9]) {
          static const double kB[9] = {0.8951, 0.2664, -0.1614, -0.7502, 1.7135, 0.0367, 0.0389, -0.0685, 1.0296};
          static const double kBi[9] = {0.9869929, -0.1470543, 0.1599627, 0.4323053, 0.5183603, 0.0492912, -0.0085287, 0.0400428, 0.9684867};
          const double w[3] = {wx / wy, 1.0, (1.0 - wx - wy) / wy}, w50[3] = {0.96422, 1.0, 0.82521};
          double lms[3], lms50[3];
          for (int i = 0; i < 3; i++) { lms[i] = kB[i * 3] * w[0] + kB[i * 3 + 1] * w[1] + kB[i * 3 + 2] * w[2]; lms50[i] = kB[i * 3] * w50[0] + kB[i * 3 + 1] * w50[1] + kB[i * 3 + 2] * w50[2]; }
          double t[9];
          for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t[r * 3 + c] = (lms50[r] / lms[r]) * kB[r * 3 + c];
          for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { out[r * 3 + c] = 0; for (int k = 0; k < 3; k++) out[r * 3 + c] += kBi[r * 3 + k] * t[k * 3 + c]; }
        };
        double As[9], Ad[9], t[9];
        bradford_to_d50(srgb[6], srgb[7], As); bradford_to_d50(dst[6], dst[7], Ad);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { t[r * 3 + c] = 0; for (int k = 0; k < 3; k++) t[r * 3 + c] += As[r * 3 + k] * A[k * 3 + c]; }
        memcpy(A, t, sizeof(A));
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { t[r * 3 + c] = 0; for (int k = 0; k < 3; k++) t[r * 3 + c] += Ad[r * 3 + k] * Bm[k * 3 + c]; }
        memcpy(Bm, t, sizeof(Bm));
      }
      inv3(Bm, Bi);
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { T[r * 3 + c] = 0; for (int k = 0; k < 3; k++) T[r * 3 + c] += Bi[r * 3 + k] * A[k * 3 + c]; }
      for (int k = 0; k < 9; k++) if (!std::isfinite(T[k]) || fabs(T[k]) > 1e6) { plan->error = "bad chromaticities"; return -1; }      // collinear primaries, a white point on the axis
    }
    float itscale = 255.0f / m.pub.intensity_target;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
      double s = 0; for (int k = 0; k < 3; k++) s += T[r * 3 + k] * (double)m.opsin_inv[k * 3 + c];
      F.opsin_inv[r * 3 + c] = (float)(s * itscale);
    }
    if (grey_target) {
      static const float kLuma[3] = {0.2126f, 0.7152f, 0.0722f};
      for (int c = 0; c < 3; c++) {
        double s = 0; for (int k = 0; k < 3; k++) s += (double)(kLuma[k] * m.opsin_inv[k * 3 + c]);       // float products summed in a double, stored as a float
        const float g = (float)s;
        for (int r = 0; r < 3; r++) F.opsin_inv[r * 3 + c] = g * itscale;
      }
    }
    for (int c = 0; c < 3; c++) { F.opsin_bias[c] = m.opsin_bias[c]; F.opsin_bias_cbrt[c] = cbrtf(m.opsin_bias[c]); }
    F.transfer = m.pub.have_gamma ? -1 : (int)m.pub.transfer_function;
    if (!m.pub.have_gamma && F.transfer != 13 && F.transfer != 8 && F.transfer != 16 && F.transfer != 1 && F.transfer != 17 && F.transfer != 18) { plan->error = "unsupported: transfer function"; return -1; }
    if (F.transfer == 18) {
      // HLG encodes scene light: the display-referred linear values go through the inverse OOTF first — every channel is scaled by
      // Y^(gamma - 1), gamma = (1 / 1.2) * 1.111^(-log2(intensity_target / 1000)), Y from the target primaries' luminance weights
      double M[9]; primaries_to_xyz(dst, M);
      for (int c = 0; c < 3; c++) F.hlg_lum[c] = (float)M[3 + c];
      const float gamma = (1.0f / 1.2f) * powf(1.111f, -log2f(m.pub.intensity_target / 1000.0f));
      F.hlg_exponent = gamma - 1.0f;
      if (F.hlg_exponent > -0.01f && F.hlg_exponent < 0.01f) F.hlg_exponent = 0.0f;
    }
    F.gamma = m.pub.gamma; F.intensity_target = m.pub.intensity_target;
  }
  F.orientation = m.orientation; F.out_w = (int)m.pub.xsize; F.out_h = (int)m.pub.ysize;
  F.canvas_w = (int)raw_w; F.canvas_h = (int)raw_h; F.crop_x0 = f.have_crop ? f.x0 : 0; F.crop_y0 = f.have_crop ? f.y0 : 0;
  if (!is_shown && !pv->blend) { F.canvas_w = f.width; F.canvas_h = f.height; F.crop_x0 = F.crop_y0 = 0; F.orientation = 1; F.out_w = f.width; F.out_h = f.height; }
  F.no_output = is_shown ? 0 : 1;
  if (pv->blend) {
    // the frame is laid over a canvas of the image's size (dev_compose.h: blend_canvas_pixel): BlendingInfo of the colour channels and of the alpha channel
    const int a = pv->alpha_ec;
    F.blend = 1; F.bl_src = pv->has_src ? (f.bl_source[0] & 3) : -1;
    F.bl_mode_c = f.bl_mode[0]; F.bl_clamp_c = f.bl_clamp[0];
    F.bl_mode_a = a >= 0 ? f.bl_mode[1 + a] : 0; F.bl_clamp_a = a >= 0 ? f.bl_clamp[1 + a] : 0;
    if (a < 0) { if (F.bl_mode_c == 2) F.bl_mode_c = 0; else if (F.bl_mode_c == 3) F.bl_mode_c = 1; }      // without an alpha channel "blend" replaces and the weighted sum is a plain one
    F.bl_premultiplied = (a >= 0 && m.ec[a].alpha_assoc) ? 1 : 0;
    plan->blend = true;
  }
  // composition: a reference frame keeps its image in the f32 planes (copied into its slot), a frame with patches blends them there; the
  // writer then runs as a stage of its own.  A Modular-encoded frame of an XYB image (libjxl's patch frames) always takes this route
  F.noise = (f.flags & 1) ? 1 : 0; F.noise_seed[0] = (uint32_t)rec.visible_index; F.noise_seed[1] = (uint32_t)rec.nonvisible_index;
  F.compose = (!is_shown || pv->blend || F.noise || F.num_patches > 0 || F.num_spline_segs > 0 || (f.encoding == 1 && m.pub.xyb_encoded) || f.upsampling != 1 || F.alpha_up > 1 || F.not_xyb) ? 1 : 0;
  plan->compose = F.compose != 0;
  memcpy(F.ref_w, pv->ref_w, sizeof(F.ref_w)); memcpy(F.ref_h, pv->ref_h, sizeof(F.ref_h));
  F.band_gr0 = 0; F.band_gr1 = F.ygroups; F.band_cy0 = 0; F.band_cy1 = F.yb; F.band_py0 = 0; F.band_py1 = F.height;
  F.band_scy0 = 0; F.band_scy1 = F.yb; F.band_g0 = 0; F.band_lfg0 = 0;
  plan->single_section = nsec == 1;
  plan->xb = F.xb; plan->yb = F.yb; plan->num_groups = F.num_groups; plan->num_lf_groups = F.num_lf_groups;
  plan->num_passes = F.num_passes; plan->width = F.width; plan->height = F.height;
  plan->modular = f.encoding == 1;
  if (f.encoding == 1) plan->single_section = false;    // nothing to re-parse on the host: no HfGlobal
  if (!plan->single_section && f.encoding == 0) {
    hx_br hb; hx_br_init(&hb, plan->cs + secs[(size_t)(1 + f.num_lf_groups)].off, secs[(size_t)(1 + f.num_lf_groups)].size);
    if (parse_hf_global(plan, pv, &hb)) return -1;
  }
  finish_blob(plan, pv);
  return 0;
}

int plan_parse_hf_single(FramePlan *plan, uint64_t lf_end_bit) {
  Priv *pv = (Priv *)plan->priv.get();
  hx_br hb; hx_br_init(&hb, plan->cs + pv->secs[0].off, plan->cs_size - pv->secs[0].off);
  hb.pos = (size_t)lf_end_bit;
  if (parse_hf_global(plan, pv, &hb)) return -1;
  pv->F.single_pass_bit = (uint32_t)hb.pos;
  finish_blob(plan, pv);
  return 0;
}

static void build_static_tables(std::vector<uint8_t> &tab);
const std::vector<uint8_t> &static_tables() {
  static std::vector<uint8_t> tab;
  static std::once_flag once;
  std::call_once(once, [] { build_static_tables(tab); });      // decodes may start concurrently on many threads
  return tab;
}
static void build_static_tables(std::vector<uint8_t> &tab) {
  std::vector<uint8_t> v(((sizeof(DevStatic) + 15) / 16) * 16, 0);
  Blob blob(v);
  DevStatic ST; memset(&ST, 0, sizeof(ST));
  init_quant_tables();
  for (int t = 0; t < 17; t++) {
    size_t n = (size_t)kQTRows[t] * 8 * (size_t)kQTCols[t] * 8;
    for (int c = 0; c < 3; c++) {
      std::vector<float> inv(n);
      for (size_t i = 0; i < n; i++) inv[i] = 1.0f / qt_weights[t][c][i];
      ST.qw_off[t][c] = blob.append(inv.data(), n * 4);
    }
  }
  for (int l = 0; l < 9; l++) {
    int n = 1 << l;
    std::vector<float> c((size_t)n * (size_t)n);
    for (int k = 0; k < n; k++) for (int i = 0; i < n; i++) c[(size_t)k * n + i] = (float)((k ? sqrt(2.0) : 1.0) * cos((2 * i + 1) * k * PI / (2.0 * n)));
    ST.cos_off[l] = blob.append(c.data(), c.size() * 4);
  }
  { float a[256]; for (int j = 0; j < 16; j++) for (int i = 0; i < 16; i++) a[j * 16 + i] = (float)kAFVBasis[j][i]; ST.afv_off = blob.append(a, sizeof(a)); }
  ST.dither_off = blob.append(kDither32, sizeof(kDither32));
  ST.rcp12_off = blob.append(kRcp12, sizeof(kRcp12));
  {
    // default upsampling kernels: the symmetric 5n x 5n matrix of each factor N = 2n, expanded to one 5 x 5 kernel per output phase (the phases
    // of the right / lower half are the mirror images of the left / upper half)
    const float *w[3] = {kUpsampling2, kUpsampling4, kUpsampling8};
    for (int t = 0; t < 3; t++) {
      const std::vector<float> k = expand_upsampling_kernels(w[t], t);
      ST.ups_off[t] = blob.append(k.data(), k.size() * 4);
    }
  }
  { float l[6 * 32]; memset(l, 0, sizeof(l));
    for (int i = 0; i < 6; i++) { int N = 1 << i; for (int k = 0; k < N; k++) { double t = k * PI / (16.0 * N); l[i * 32 + k] = (float)(1.0 / (cos(t) * cos(2 * t) * cos(4 * t))); } }
    ST.llf_off = blob.append(l, sizeof(l)); }
  for (int o = 0; o < 13; o++) {
    const int st = kOrderStrategy[o];
    std::vector<uint32_t> nat((size_t)kCoveredX[st] * kCoveredY[st] * 64);
    natural_order(st, nat.data());
    ST.nat_order_off[o] = blob.append(nat.data(), nat.size() * 4);
  }
  memcpy(v.data(), &ST, sizeof(ST));
  tab.swap(v);
}

}  // namespace jxlamd
