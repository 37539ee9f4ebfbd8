// jxl_coder_amd/csrc/decoder.hip — decoder context: device buffers, H2D of the codestream + frame tables,
// kernel sequencing on one HIP stream, error collection; and the C-ABI of include/jxl_amd.h.
// Host counterpart of the reference's DecodeJpegXlOneShot driver loop (interop/JxlDecoding.cpp:36-176).
#include <atomic>
#include <limits>
#include <chrono>
#include <thread>
#include "decoder_ctx.h"

namespace jxlamd {
std::string &tls_error() { static thread_local std::string e; return e; }
}
#define g_tls_error (tls_error())

static void fill_public_info(const ImageInfo &i, uint32_t flags, jxlamd_info *o) {
  memset(o, 0, sizeof(*o));
  o->xsize = i.xsize; o->ysize = i.ysize; o->bits_per_sample = i.bits_per_sample; o->exponent_bits_per_sample = i.exp_bits;
  o->num_color_channels = i.num_color_channels; o->num_extra_channels = i.num_extra_channels; o->alpha_bits = i.alpha_bits;
  o->alpha_premultiplied = i.alpha_premultiplied; o->orientation = 1; o->have_animation = i.have_animation;
  o->uses_original_profile = i.uses_original_profile;
  o->intensity_target = i.intensity_target <= 0.f ? 255.f : i.intensity_target;
  o->have_encoded_profile = !i.want_icc;
  o->color_space = i.color_space; o->white_point = i.white_point;
  o->primaries = (i.color_space == 0 || i.color_space == 3) ? i.primaries : 0;      // grey / XYB encodings carry no primaries: libjxl leaves the field 0
  o->transfer_function = i.have_gamma ? 65535u : i.transfer_function; o->rendering_intent = i.rendering_intent;
  o->gamma = i.have_gamma ? (double)i.gamma : 0.0;
  o->out_bits = (i.bits_per_sample > 8 && (flags & JXLAMD_ALLOW_16BIT)) ? 16 : 8;
  // JxlDecoding.cpp:126-133: `cs == RGB && tf == HLG || tf == PQ || ...` (&& binds tighter than ||)
  uint32_t tf = o->transfer_function;
  o->prefer_encoding = o->have_encoded_profile &&
                       ((o->color_space == 0 && tf == 18) || tf == 16 || tf == 17 || tf == 1 || tf == 13 || tf == 65535u);
  o->has_alpha_in_origin = i.num_extra_channels > 0 && i.alpha_bits > 0;
  o->icc_size = i.icc_size;
  if (o->have_encoded_profile) {          // chromaticities of the enum values, as libjxl's ColorEncoding reports them
    static const double kWp[4][2] = {{0.3127, 0.3290}, {0, 0}, {1.0 / 3, 1.0 / 3}, {0.314, 0.351}};       // D65, custom, E (10), DCI (11)
    const double *wp = i.white_point == 1 ? kWp[0] : i.white_point == 10 ? kWp[2] : i.white_point == 11 ? kWp[3] : nullptr;
    if (wp) { o->white_point_xy[0] = wp[0]; o->white_point_xy[1] = wp[1]; }
    else { o->white_point_xy[0] = i.wp_xy[0]; o->white_point_xy[1] = i.wp_xy[1]; }
    if (i.color_space == 0 || i.color_space == 3) {      // RGB / unknown carry primaries
      static const double kPr[3][6] = {{0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204},      // sRGB (1)
                                       {0.708, 0.292, 0.170, 0.797, 0.131, 0.046},                                           // BT.2100 (9)
                                       {0.680, 0.320, 0.265, 0.690, 0.150, 0.060}};                                          // P3 (11)
      double pr[6];
      const double *src = i.primaries == 1 ? kPr[0] : i.primaries == 9 ? kPr[1] : i.primaries == 11 ? kPr[2] : nullptr;
      for (int k = 0; k < 6; k++) pr[k] = src ? src[k] : (double)i.prim_xy[k];
      o->primaries_red_xy[0] = pr[0]; o->primaries_red_xy[1] = pr[1]; o->primaries_green_xy[0] = pr[2]; o->primaries_green_xy[1] = pr[3];
      o->primaries_blue_xy[0] = pr[4]; o->primaries_blue_xy[1] = pr[5];
    }
  }
}

static int size_guard(const jxlamd_info &o, uint32_t flags, std::string *err) {
  uint64_t cur = (uint64_t)o.xsize * o.ysize * 4 * (o.out_bits == 16 ? 2 : 1);
  if (!(flags & JXLAMD_NO_SIZE_GUARD) && cur >= (uint64_t)std::numeric_limits<int32_t>::max()) {
    *err = "Invalid image size exceed allowance, current size w: " + std::to_string(o.xsize) + ", h: " + std::to_string(o.ysize);
    return JXLAMD_ERR_SIZE;
  }
  return 0;
}

static int ensure_color_plan(jxlamd_decoder *d, int is_u16, uint32_t depth, uint32_t primaries, uint32_t tf, const double *xy8, float intensity_target, bool *runs);
static PostKind reformat_kind(uint32_t resolved_config, int src_is_u16);
static int err_class(const std::string &e) { return e.rfind("unsupported", 0) == 0 ? JXLAMD_ERR_UNSUPPORTED : JXLAMD_ERR_INVALID; }
// internal return code: the lean LF kernel met a channel that needs a general lock-step loop (kErrNeedGeneral) — decode again with the general build
// what one decoder context learnt about the frames of this process serves the others (contexts of a service see the same kind of content)
static std::atomic<int> g_lf_pool_floor{0};
// measurement only (jxlamd_debug_set_ablate; profiles/r06_ablations_stage_floors.json): stages the FLIGHT path leaves out — 1 the LF stage (clear, extra-channel globals, LfGroup
// streams, LF smoothing), 2 the PassGroup stage, 4 reconstruction + filters + writer.  The frames then decode to whatever the slots held before: only the clock is of interest.
static std::atomic<int> g_ablate{0};
static constexpr int kRetryGeneral = 0x7e7e;
// ... or a channel whose packed tables did not fit the LDS table pool this launch was sized with (kErrNeedPool): decode again with the largest
static constexpr int kRetryPool = 0x7e7f;
// ... or the shared HF pools moved while this flight's LF stage ran (decode_batch_once): start the flight over
static constexpr int kRetryMoved = 0x7e80;
// ... or a frame's sparse coefficient lists did not hold its coefficients (kErrNeedDense): decode the flight again with the dense planes
static constexpr int kRetryDense = 0x7e81;
// ... or a frame holds varblocks of the DCT128 / DCT256 families and the flight did not launch their kernel: decode it again with it (the context keeps launching it)
static constexpr int kRetryHuge = 0x7e82;
int dev_err_class(uint32_t derr) { return (derr & 0xFFFFu & ~(kErrBitstream | kErrAnsFinal)) ? JXLAMD_ERR_UNSUPPORTED : JXLAMD_ERR_INVALID; }

// host twin of mod_group_scratch_ints (dev_modframe.h)
static size_t mod_scratch_ints_host(const DevFrame &F) {
  const size_t gd = (size_t)(F.mod_group_dim > 0 ? F.mod_group_dim : 256);
  return (size_t)(F.mod_nch - F.mod_first_group_ch + 1) * gd * gd + (size_t)kWideWpInts;
}
// ... + the ModularLfGroup rectangles (mod_lfgroup_body): 256 x 256 samples per channel and LF group
static size_t mod_scratch_total_ints(const DevFrame &F, int num_groups, int num_lf_groups) {
  return ((size_t)num_groups + 1) * mod_scratch_ints_host(F) + (size_t)num_lf_groups * (size_t)F.mod_lf_nch * 65536;
}

__global__ void k_or_flags(const uint32_t *src, uint32_t *dst) { if (threadIdx.x == 0 && *src) atomicOr(dst, *src); }
__global__ void __launch_bounds__(256) k_fill_opaque_alpha(void *out, size_t npx, int bits) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npx) return;
  if (bits == 16) ((uint16_t *)out)[i * 4 + 3] = 65535; else ((uint8_t *)out)[i * 4 + 3] = 255;
}

BandGeom band_geometry(const DevFrame &F, int gr0, int gr1) {
  BandGeom q;
  q.gr0 = gr0; q.gr1 = gr1;
  q.cy0 = gr0 * 32; q.cy1 = std::min(gr1 * 32, (int)F.yb);
  q.py0 = gr0 * 256; q.py1 = std::min(gr1 * 256, (int)F.height);
  q.lr0 = gr0 / 8; q.lr1 = std::min((gr1 + 7) / 8, (int)F.ylfg);
  q.scy0 = std::max(0, q.lr0 * 256 - 1); q.scy1 = std::min((int)F.yb, q.lr1 * 256 + 1);
  q.st0 = q.scy0 / 8; q.st1 = (q.scy1 + 7) / 8;
  q.g0 = gr0 * F.xgroups; q.ng = (gr1 - gr0) * F.xgroups;
  q.lfg0 = q.lr0 * F.xlfg; q.nlfg = (q.lr1 - q.lr0) * F.xlfg;
  q.prow0 = std::max(0, q.py0 - 8); q.prow1 = std::min((int)F.ph, q.cy1 * 8 + 8);
  q.halo = (F.gab ? 1 : 0) + (F.epf_iters >= 3 ? 3 : 0) + (F.epf_iters >= 1 ? 2 : 0) + (F.epf_iters >= 2 ? 1 : 0);
  q.whole = gr0 == 0 && gr1 == F.ygroups;
  if (F.is_modular) {               // Modular-encoded frames (group sizes 128 .. 1024) are never banded: every row, whatever the group grid
    q.cy0 = 0; q.cy1 = F.yb; q.py0 = 0; q.py1 = F.height; q.scy0 = 0; q.scy1 = F.yb; q.st0 = 0; q.st1 = (F.yb + 7) / 8; q.prow0 = 0; q.prow1 = F.ph; q.whole = true;
  }
  return q;
}

// host parse + buffers + H2D + clears for one frame (everything before the first kernel).  band_rows = {first, last+1} group row
// for a band decode (band.hip): the same buffers, sized for the band's rows and addressed through biased pointers.
int jxlamd_decoder::prepare(FrameSlot &S, const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info,
                            bool parsed, bool own_planes, const int *band_rows) {
  FramePlan &plan = S.plan;
  static const bool trace_prep = getenv("JXLAMD_TRACE_BANDS") && atoi(getenv("JXLAMD_TRACE_BANDS"));
  const auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tp0 = now_ms();
  if (!parsed) { plan = FramePlan(); (void)plan_parse(jxl, size, &plan); }
  const double tp1 = now_ms();
  if (!plan.error.empty() || plan.tables.empty()) { set_error(plan.error); return err_class(plan.error); }
  fill_public_info(plan.info, flags, &S.pi);
  if (info) *info = S.pi;
  DevFrame *Fh = (DevFrame *)plan.tables.data();       // host copy of the frame parameters: the band fields are patched in before the upload
  if (band_rows) {
    if (plan.compose || !plan.refs.empty()) { set_error("unsupported: band decode of a frame with patches / reference frames"); return JXLAMD_ERR_UNSUPPORTED; }
    if (plan.modular || plan.has_ec || plan.single_section) { set_error("unsupported: band decode of a Modular / extra-channel / single-group frame"); return JXLAMD_ERR_UNSUPPORTED; }
    if (Fh->orientation != 1) { set_error("unsupported: band decode of a frame with a non-identity orientation"); return JXLAMD_ERR_UNSUPPORTED; }
    if (plan.cropped) { set_error("unsupported: band decode of a frame that does not cover the image"); return JXLAMD_ERR_UNSUPPORTED; }
    if (band_rows[0] < 0 || band_rows[1] <= band_rows[0] || band_rows[1] > Fh->ygroups) { set_error("band rows outside the frame"); return JXLAMD_ERR_BUFFER; }
  }
  const BandGeom q = band_geometry(*Fh, band_rows ? band_rows[0] : 0, band_rows ? band_rows[1] : Fh->ygroups);
  S.band = q;
  Fh->epf_rcp_x86 = epf_rcp_mode;
  Fh->band_gr0 = q.gr0; Fh->band_gr1 = q.gr1; Fh->band_cy0 = q.cy0; Fh->band_cy1 = q.cy1; Fh->band_py0 = q.py0; Fh->band_py1 = q.py1;
  Fh->band_scy0 = q.scy0; Fh->band_scy1 = q.scy1; Fh->band_g0 = q.g0; Fh->band_lfg0 = q.lfg0;
  const size_t bpp = S.pi.out_bits == 16 ? 8 : 4;
  if (q.whole) { std::string e; int rc = size_guard(S.pi, flags, &e); if (rc) { set_error(e); return rc; } }     // a band is below the Bitmap layer (BASELINE config 4)
  S.out_bytes = q.whole ? (size_t)S.pi.xsize * S.pi.ysize * bpp : (size_t)S.pi.xsize * (size_t)(q.py1 - q.py0) * bpp;
  if (Fh->no_output) S.out_bytes = 0;                   // a reference frame: nothing is written out
  const bool post_wanted = wpost_enabled && q.whole && !Fh->no_output;      // the buffer then receives the Bitmap format (checked against its size below)
  if (!post_wanted && out_cap < S.out_bytes) { set_error("output buffer too small"); return JXLAMD_ERR_BUFFER; }
  const size_t ncell = (size_t)plan.xb * (size_t)(q.scy1 - q.scy0);
  const size_t ntile = (size_t)((plan.xb + 7) / 8) * (size_t)(q.st1 - q.st0);
  const size_t npx = (size_t)plan.xb * 8 * (size_t)(q.prow1 - q.prow0);
  if (!stat_uploaded) {
    const std::vector<uint8_t> &st = static_tables();
    HIPCHECK(stat.ensure(st.size()));
    HIPCHECK(hipMemcpy(stat.p, st.data(), st.size(), hipMemcpyHostToDevice));   // one-time, synchronous (shared host source)
    stat_uploaded = true;
  }
  // The kernels' bit reader fetches aligned words and relies on >= 64 zero bytes after the codestream, so the stream always goes
  // through the slot's own padded buffer: H2D from page-locked staging, or — JXLAMD_IN_DEVICE — a device-to-device copy of the
  // caller's resident bytes (no requirement on the caller's buffer: neither alignment nor padding).
  const bool cs_resident = (flags & JXLAMD_IN_DEVICE) && jxl_dev && plan.cs_owned.empty();
  const bool in_flight = !own_planes;                 // frames of a batched flight: decode_batch uploads tables and streams of all of them at once
  S.up_cs_dev = cs_resident ? (const uint8_t *)jxl_dev + (plan.cs - jxl) : nullptr;
  const uint8_t *d_cs = nullptr;
  if (!in_flight) {
    HIPCHECK(S.cs.ensure(plan.cs_size + 64));
    if (cs_resident) HIPCHECK(hipMemcpyAsync(S.cs.p, S.up_cs_dev, plan.cs_size, hipMemcpyDeviceToDevice, stream));
    else if (band_rows && Fh->nsec > 1) {
      // A band only reads the sections of its own LF groups and PassGroups: stage and upload those byte ranges (at their codestream offsets:
      // the kernels keep addressing sections by absolute offset) instead of the whole file — for the 32768 x 32768 frame of config 4 a band of
      // an eighth of the rows moves ~20 MB instead of 157 MB through page-locked memory and PCIe, and eight bands did that side by side.
      const DevSection *secs = (const DevSection *)(plan.tables.data() + Fh->sec_off);
      std::vector<std::pair<size_t, size_t>> need;                      // [begin, end) byte ranges, + 64 bytes the bit reader may look ahead
      const auto add = [&](int si) { if (si >= 0 && si < Fh->nsec && secs[si].size) need.push_back({secs[si].off, std::min<size_t>(plan.cs_size, (size_t)secs[si].off + secs[si].size + 64)}); };
      for (int g = q.lfg0; g < q.lfg0 + q.nlfg; g++) add(1 + g);
      for (int p = 0; p < Fh->num_passes; p++) for (int g = q.g0; g < q.g0 + q.ng; g++) add(2 + Fh->num_lf_groups + p * Fh->num_groups + g);
      std::sort(need.begin(), need.end());
      std::vector<std::pair<size_t, size_t>> runs;
      for (const auto &r : need) { if (!runs.empty() && r.first <= runs.back().second + 4096) runs.back().second = std::max(runs.back().second, r.second); else runs.push_back(r); }
      size_t total = 0;
      for (const auto &r : runs) total += r.second - r.first;
      HIPCHECK(S.h_cs.ensure(total + 64));
      size_t at = 0;
      for (const auto &r : runs) {
        memcpy((uint8_t *)S.h_cs.p + at, plan.cs + r.first, r.second - r.first);
        HIPCHECK(hipMemcpyAsync((uint8_t *)S.cs.p + r.first, (uint8_t *)S.h_cs.p + at, r.second - r.first, hipMemcpyHostToDevice, stream));
        at += r.second - r.first;
      }
    } else {
      HIPCHECK(S.h_cs.ensure(plan.cs_size));
      memcpy(S.h_cs.p, plan.cs, plan.cs_size);
      HIPCHECK(hipMemcpyAsync(S.cs.p, S.h_cs.p, plan.cs_size, hipMemcpyHostToDevice, stream));
    }
    HIPCHECK(hipMemsetAsync((uint8_t *)S.cs.p + plan.cs_size, 0, 64, stream));
    d_cs = (const uint8_t *)S.cs.p;
    // single-section frames: room for the phase-2 (HfGlobal) tables — orders of all 13 order ids x 3 channels x passes + histograms
    HIPCHECK(S.tables.ensure(plan.tables.size() + (plan.single_section ? (size_t)(12u << 20) : 0u)));
    HIPCHECK(S.h_tables.ensure(plan.tables.size()));
    memcpy(S.h_tables.p, plan.tables.data(), plan.tables.size());
    HIPCHECK(hipMemcpyAsync(S.tables.p, S.h_tables.p, plan.tables.size(), hipMemcpyHostToDevice, stream));
  }
  const double tp2 = now_ms();
  if (trace_prep && band_rows) fprintf(stderr, "[prepare band %d] parse %.1f ms, codestream + tables staging and upload calls %.1f ms (tables %zu B, cs_owned %zu B)\n", band_rows[0], tp1 - tp0, tp2 - tp1, plan.tables.size(), plan.cs_owned.size());
  if (!plan.modular) {
    for (int i = 0; i < 5; i++) HIPCHECK(S.cells8[i].ensure(ncell));
    for (int i = 0; i < 2; i++) HIPCHECK(S.tiles[i].ensure(ntile));
    for (int i = 0; i < 6; i++) HIPCHECK(S.lf[i].ensure(ncell * 4));
    HIPCHECK(S.coef_off.ensure(ncell * 4));
    HIPCHECK(S.coef_cnt.ensure(ncell * 4));
    if (own_planes) for (int c = 0; c < 3; c++) HIPCHECK(S.coef[c].ensure((size_t)q.ng * 65536 * 4));   // flights: the decoder's coefficient pool
    if (own_planes) for (int i = 0; i < 6; i++) HIPCHECK(S.planes[i].ensure(npx * 4));   // flights borrow sets of the decoder's plane pool instead
    HIPCHECK(S.lf_scratch.ensure((size_t)q.nlfg * kLfScratchInts * 4));
    HIPCHECK(S.local.ensure((size_t)q.nlfg * sizeof(LocalTreeScratch)));
    HIPCHECK(S.pass_nz.ensure((size_t)q.ng * kPassBlkStride));
    HIPCHECK(S.big_list[0].ensure((ncell / 8 + 16) * 4));
    HIPCHECK(S.big_list[1].ensure((ncell / 32 + 16) * 4));
    HIPCHECK(S.big_list[2].ensure((ncell + 16) * 4));
    HIPCHECK(S.big_list[3].ensure((ncell / 256 + 16) * 4));
    if (plan.has_ec) {                                   // extra channels: a Modular image next to the VarDCT one
      HIPCHECK(S.mod_pool.ensure(plan.mod_pool_ints * 4 + 256));
      HIPCHECK(S.mod_scratch.ensure(mod_scratch_total_ints(*Fh, plan.num_groups, plan.num_lf_groups) * 4 + 256));      // + 1: the GlobalModular stream's slot
      HIPCHECK(S.local.ensure((size_t)std::max(plan.num_groups, plan.num_lf_groups) * sizeof(LocalTreeScratch)));
      HIPCHECK(S.pass_end.ensure((size_t)plan.num_groups * (size_t)plan.num_passes * 8));
    }
  } else {
    if (Fh->lz_win_len) HIPCHECK(S.lz_win.ensure(((size_t)Fh->lz_win_len + (size_t)plan.num_groups * (size_t)Fh->lz_win_group) * 4));
    HIPCHECK(S.mod_pool.ensure(plan.mod_pool_ints * 4 + 256));
    HIPCHECK(S.mod_scratch.ensure(mod_scratch_total_ints(*Fh, plan.num_groups, plan.num_lf_groups) * 4 + 256));      // + 1: the GlobalModular stream's slot
    HIPCHECK(S.local.ensure((size_t)(plan.num_groups > 1 ? plan.num_groups : 1) * sizeof(LocalTreeScratch)));
    if (plan.compose) for (int i = 0; i < (Fh->xyb_modular ? 6 : 3); i++) HIPCHECK(S.planes[i].ensure(npx * 4));      // composed: the image moves into the f32 planes (k_mod_to_planes)
  }
  HIPCHECK(S.misc.ensure(4096 + (size_t)plan.num_lf_groups * 72));
  S.host_out = nullptr; S.d_out = out_ptr;
  // A10 + A11 with the decode (jxlamd_decoder_set_writer_post): the caller's buffer receives the Bitmap format.  Decided per frame as the
  // reference's JNI layer does (cpp/JniDecoding.cpp:131-137: colour matrix when the enum encoding is 'preferred', RGB, and the API level is below 34)
  S.post_active = false; S.post_fused = false;
  if (post_wanted) {
    const bool is16 = S.pi.out_bits == 16;
    jxlamd_reformat_info ri;
    int rc = jxlamd_reformat_query(S.pi.xsize, S.pi.ysize, is16, wpost_cfg, (int)S.pi.has_alpha_in_origin, wpost_api, &ri);
    if (rc) { set_error(g_tls_error); return rc; }
    if (out_cap < ri.bytes) { set_error("output buffer too small"); return JXLAMD_ERR_BUFFER; }
    const uint32_t tf = S.pi.transfer_function;
    const bool matrix = S.pi.prefer_encoding && (tf == 16 || tf == 18 || tf == 17 || tf == 1 || tf == 65535u || tf == 13) && S.pi.color_space == 0 && wpost_api < 34;
    S.post_depth = is16 ? 16u : 8u;                       // bitDepth as DecodeJpegXlOneShot reports it (JxlDecoding.cpp:92-101)
    S.post_runs = false;
    if (matrix) {
      const double xy8[8] = {S.pi.primaries_red_xy[0], S.pi.primaries_red_xy[1], S.pi.primaries_green_xy[0], S.pi.primaries_green_xy[1], S.pi.primaries_blue_xy[0], S.pi.primaries_blue_xy[1],
                             S.pi.white_point_xy[0], S.pi.white_point_xy[1]};
      rc = ensure_color_plan(this, is16, S.post_depth, S.pi.primaries, tf, xy8, S.pi.intensity_target, &S.post_runs);
      if (rc) return rc;
    }
    S.post_active = true; S.post_gen = post_lut_gen;
    S.post_kind = (int)reformat_kind(ri.resolved_config, is16);
    S.post_premul = !S.pi.alpha_premultiplied && S.pi.has_alpha_in_origin; S.post_att = !S.pi.alpha_premultiplied;
    S.post_stride = ri.stride; S.post_bytes = ri.bytes;
    // inside the writer where the frame's last stage is a per-stage kernel (three EPF iterations: BASELINE config 5); elsewhere one pass behind it
    // (round 5: the column sweep's writer has the post instantiation too — every VarDCT frame that is written straight from its last filter stage)
    S.post_fused = !plan.modular && !plan.compose && !plan.cropped && plan.refs.empty() && Fh->orientation >= 1;
    S.post_final = out_ptr;
    if (!(flags & JXLAMD_OUT_DEVICE)) { HIPCHECK(S.out.ensure(ri.bytes)); S.post_final = S.out.p; S.host_out = out_ptr; }
    const uint32_t line = ri.format == JXLAMD_FMT_RGB_565 ? S.pi.xsize * 2 : ri.format == JXLAMD_FMT_RGBA_F16 ? S.pi.xsize * 8 : S.pi.xsize * 4;
    if (ri.stride != line) HIPCHECK(hipMemsetAsync(S.post_final, 0, ri.bytes, stream));
    if (S.post_fused) {
      HIPCHECK(S.post_fz.ensure((size_t)(1 + S.pi.ysize) * 4));
      if (!in_flight) {
        HIPCHECK(hipMemsetAsync(S.post_fz.p, 0xFF, (size_t)(1 + S.pi.ysize) * 4, stream));
        HIPCHECK(hipMemsetAsync(S.post_fz.p, 0, 4, stream));
      }                                                   // (frames of a flight: k_clear_b initialises it, and the DevPost block travels inside the frame's tables — no extra dispatch per frame: every one of them waits its turn among the kernels of the other contexts)
      DevPost Q; memset(&Q, 0, sizeof(Q));
      if (S.post_runs) Q.P = post_dev;
      Q.matrix = S.post_runs ? 1 : 0; Q.premul = S.post_premul ? 1 : 0; Q.kind = S.post_kind; Q.depth = (int32_t)S.post_depth; Q.attenuate = S.post_att ? 1 : 0;
      Q.dst_stride = ri.stride; Q.dst = (uint8_t *)S.post_final; Q.row_fz = (uint32_t *)S.post_fz.p;
      Q.rows = (int32_t)S.pi.ysize;
      if (in_flight) {
        plan.tables.resize((plan.tables.size() + 15) & ~(size_t)15);
        S.post_off = plan.tables.size();
        plan.tables.insert(plan.tables.end(), (const uint8_t *)&Q, (const uint8_t *)&Q + sizeof(Q));
        Fh = (DevFrame *)plan.tables.data();              // (the vector may have moved)
      } else {
        HIPCHECK(S.post_dev.ensure(sizeof(DevPost)));
        HIPCHECK(S.h_post.ensure(sizeof(DevPost)));
        memcpy(S.h_post.p, &Q, sizeof(Q));
        HIPCHECK(hipMemcpyAsync(S.post_dev.p, S.h_post.p, sizeof(DevPost), hipMemcpyHostToDevice, stream));
      }
      S.d_out = S.post_final;                             // (the RGBA writer is not used: B.out only has to be a valid address)
    } else {
      HIPCHECK(S.post_tmp.ensure(S.out_bytes));
      S.d_out = S.post_tmp.p;
    }
  } else
  if (!(flags & JXLAMD_OUT_DEVICE)) { HIPCHECK(S.out.ensure(S.out_bytes)); S.d_out = S.out.p; S.host_out = out_ptr; }
  if (plan.cropped) {
    // the frame does not cover the image: what it leaves out shows the cleared canvas — transparent black, or opaque black when the image
    // has no alpha channel
    HIPCHECK(hipMemsetAsync(S.d_out, 0, S.out_bytes, stream));
    if (!S.pi.has_alpha_in_origin) {
      const size_t npx = S.out_bytes / bpp;
      hipLaunchKernelGGL(k_fill_opaque_alpha, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, S.d_out, npx, (int)S.pi.out_bits);
    }
  }
  DevBuffers &B = S.B;
  memset(&B, 0, sizeof(B));
  // pointer biases: element [frame coordinate] lands at [frame coordinate - band origin] of the allocation
  const ptrdiff_t cb = (ptrdiff_t)q.scy0 * plan.xb, tb = (ptrdiff_t)q.st0 * ((plan.xb + 7) / 8), pb = (ptrdiff_t)q.prow0 * plan.xb * 8;
  B.codestream = d_cs; B.tables = in_flight ? nullptr : (const uint8_t *)S.tables.p;      // in_flight: decode_batch points both into the flight's buffers
  B.strategy = (uint8_t *)S.cells8[0].p - cb; B.first = (uint8_t *)S.cells8[1].p - cb; B.qfm1 = (uint8_t *)S.cells8[2].p - cb;
  B.sharp = (uint8_t *)S.cells8[3].p - cb; B.lf_idx = (uint8_t *)S.cells8[4].p - cb;
  B.xfromy = (int8_t *)S.tiles[0].p - tb; B.bfromy = (int8_t *)S.tiles[1].p - tb;
  for (int c = 0; c < 3; c++) { B.lf[c] = (float *)S.lf[c].p - cb; B.lf_s[c] = (float *)S.lf[3 + c].p - cb;
                                B.coef[c] = (int32_t *)S.coef[c].p - (ptrdiff_t)q.g0 * 65536;
                                B.plane_a[c] = (float *)S.planes[c].p - pb; B.plane_b[c] = (float *)S.planes[3 + c].p - pb; }
  B.coef_cnt = (uint32_t *)S.coef_cnt.p - cb;      // (coef_sp / sp_group: set by decode_batch for the frames of a sparse flight)
  B.coef_off = (uint32_t *)S.coef_off.p - cb; B.lf_scratch = (int32_t *)S.lf_scratch.p - (ptrdiff_t)q.lfg0 * kLfScratchInts;
  B.local = (LocalTreeScratch *)S.local.p - q.lfg0;
  B.mod_pool = (int32_t *)S.mod_pool.p; B.mod_scratch = (int32_t *)S.mod_scratch.p; B.pass_nz = (uint8_t *)S.pass_nz.p - (ptrdiff_t)q.g0 * kPassBlkStride;
  B.pass_end_bits = (uint64_t *)S.pass_end.p; B.mod_end_bit = (uint64_t *)((uint8_t *)S.misc.p + 256);
  B.err = (uint32_t *)S.misc.p; B.out = (uint8_t *)S.d_out - (ptrdiff_t)q.py0 * (ptrdiff_t)S.pi.xsize * (ptrdiff_t)bpp; B.out_bits = (int32_t)S.pi.out_bits; B.stat = (const uint8_t *)stat.p;
  B.lz_win = (plan.modular && Fh->lz_win_len) ? (uint32_t *)S.lz_win.p : nullptr;
  if (Fh->upsampling > 1 || Fh->alpha_up > 1) {
    const size_t n = (size_t)Fh->full_w * (size_t)Fh->full_h;
    HIPCHECK(S.up_planes.ensure(4 * n * 4));
    for (int c = 0; c < 4; c++) B.up[c] = (float *)S.up_planes.p + (size_t)c * n;
  }
  for (int k = 0; k < 4; k++) {
    const bool have = ref_store[k].p && Fh->ref_w[k] == ref_w[k] && Fh->ref_h[k] == ref_h[k] && ref_w[k] > 0;
    const size_t n = (size_t)ref_w[k] * (size_t)ref_h[k];
    for (int c = 0; c < 3; c++) B.ref[k][c] = have ? (float *)ref_store[k].p + (size_t)c * n : nullptr;
    B.ref_a[k] = (have && ref_alpha[k]) ? (float *)ref_store[k].p + 3 * n : nullptr;      // a blended canvas kept with its alpha plane
  }
  for (int c = 0; c < 3; c++) B.noise[c] = nullptr;
  if (Fh->noise) {
    const size_t nn = Fh->upsampling > 1 ? (size_t)Fh->full_w * (size_t)Fh->full_h : npx;      // drawn at the resolution it is added at
    HIPCHECK(S.noise_planes.ensure(3 * nn * 4));
    for (int c = 0; c < 3; c++) B.noise[c] = (float *)S.noise_planes.p + (size_t)c * nn;
  }
  for (int c = 0; c < 3; c++) B.lf_frame[c] = nullptr;
  if (Fh->use_lf_frame) {
    const int k = Fh->lf_frame_slot;
    if (k < 4 || k > 7 || !ref_store[k].p || ref_w[k] != Fh->lf_frame_w || ref_h[k] != Fh->lf_frame_h) { set_error("LF frame missing"); return JXLAMD_ERR_INVALID; }
    for (int c = 0; c < 3; c++) B.lf_frame[c] = (const float *)ref_store[k].p + (size_t)c * (size_t)ref_w[k] * (size_t)ref_h[k];
  }
  for (int c = 0; c < 4; c++) B.canvas_save[c] = nullptr;
  B.post = (S.post_active && S.post_fused && !in_flight) ? (const DevPost *)S.post_dev.p : nullptr;      // in a flight: inside the flight's tables (decode_batch)
  if (Fh->blend && Fh->bl_src >= 0 && !B.ref[Fh->bl_src][0]) { set_error("blending: the source canvas is missing"); return JXLAMD_ERR_INVALID; }
  if (Fh->num_patches > 0) {
    const DevPatch *P = (const DevPatch *)(plan.tables.data() + Fh->patch_off);
    for (int i = 0; i < Fh->num_patches; i++) if (!B.ref[P[i].ref][0]) { set_error("patch dictionary: reference frame missing"); return JXLAMD_ERR_INVALID; }
  }
  B.big_list[0] = (uint32_t *)S.big_list[0].p; B.big_list[1] = (uint32_t *)S.big_list[1].p; B.big_list[2] = (uint32_t *)S.big_list[2].p; B.big_list[3] = (uint32_t *)S.big_list[3].p; B.big_count = (uint32_t *)((uint8_t *)S.misc.p + 64);
  S.A.lf_end_bits = (uint64_t *)((uint8_t *)S.misc.p + 4096);
  S.A.lf_times = (uint64_t *)((uint8_t *)S.misc.p + 4096 + (size_t)plan.num_lf_groups * 8);
  // frames of a batched flight (in_flight): one k_clear_b launch clears these for all of them
  if (!in_flight) HIPCHECK(hipMemsetAsync(S.misc.p, 0, 4096 + (size_t)plan.num_lf_groups * 72, stream));
  { static const uint32_t dbg_mod = getenv("JXLAMD_DEBUG_MOD") ? (uint32_t)atoi(getenv("JXLAMD_DEBUG_MOD")) : 0u;      // measurement switches of the block-form tree loop (single decodes only)
    if (dbg_mod && !in_flight) { static uint32_t word; word = dbg_mod; HIPCHECK(hipMemcpyAsync((uint8_t *)S.misc.p + 16, &word, 4, hipMemcpyHostToDevice, stream)); } }
  if (!plan.modular) {
    if (!in_flight) HIPCHECK(hipMemsetAsync(S.cells8[1].p, 0, ncell, stream));
    // The reconstruction kernels clear every coefficient they consume, so a slot whose previous decode completed is
    // already all-zero; only fresh / regrown / failed slots are cleared here.
    if (in_flight) return JXLAMD_OK;                    // flights use the decoder's coefficient pool (decode_batch)
    const size_t coef_bytes = (size_t)q.ng * 65536 * 4;
    const bool clean = S.coef_clean && S.coef_clean_bytes >= coef_bytes && S.coef_clean_ptr[0] == S.coef[0].p && S.coef_clean_ptr[1] == S.coef[1].p &&
                       S.coef_clean_ptr[2] == S.coef[2].p;
    if (!clean) for (int c = 0; c < 3; c++) HIPCHECK(hipMemsetAsync(S.coef[c].p, 0, S.coef[c].cap, stream));
    S.coef_clean = false;                              // until collect() has seen this decode succeed
    for (int c = 0; c < 3; c++) S.coef_clean_ptr[c] = S.coef[c].p;
    S.coef_clean_bytes = std::min(std::min(S.coef[0].cap, S.coef[1].cap), S.coef[2].cap);
  }
  return JXLAMD_OK;
}

// single-section frames: HfGlobal follows LfGroup 0 in the same section; its bit position is only known after the LF kernel
int jxlamd_decoder::finish_single_section(FrameSlot &S) {
  uint64_t end_bit = 0; uint32_t derr = 0;
  HIPCHECK(h_flags.ensure(256));
  HIPCHECK(hipMemcpyAsync(h_flags.p, S.A.lf_end_bits, 8, hipMemcpyDeviceToHost, stream));
  HIPCHECK(hipMemcpyAsync((uint8_t *)h_flags.p + 8, S.B.err, 4, hipMemcpyDeviceToHost, stream));
  HIPCHECK(hipStreamSynchronize(stream));
  memcpy(&end_bit, h_flags.p, 8); memcpy(&derr, (uint8_t *)h_flags.p + 8, 4);
  if ((derr & kErrNeedPool) && lf_pool_bytes < kModPoolBytes) { lf_pool_floor = kModPoolBytes; g_lf_pool_floor.store(kModPoolBytes); lf_pool_bytes = kModPoolBytes; return kRetryPool; }      // (the other flags of an attempt that stopped for the pool say nothing: see decode_batch_once)
  if ((derr & kErrNeedGeneral) && !lf_general) return kRetryGeneral;
  if (derr) { set_error("corrupt or unsupported stream (device flags " + std::to_string(derr) + ", LfGroup)"); return dev_err_class(derr); }
  if (plan_parse_hf_single(&S.plan, end_bit)) { set_error(S.plan.error); return err_class(S.plan.error); }
  ((DevFrame *)S.plan.tables.data())->epf_rcp_x86 = epf_rcp_mode;      // (the blob was built again from the parser's own copy of the frame parameters)
  HIPCHECK(S.tables.ensure(S.plan.tables.size()));
  S.B.tables = (const uint8_t *)S.tables.p;              // ensure() may have moved the buffer
  HIPCHECK(S.h_tables.ensure(S.plan.tables.size()));
  memcpy(S.h_tables.p, S.plan.tables.data(), S.plan.tables.size());
  HIPCHECK(hipMemcpyAsync(S.tables.p, S.h_tables.p, S.plan.tables.size(), hipMemcpyHostToDevice, stream));
  return JXLAMD_OK;
}

// everything after the entropy stages: reconstruction, loop filters, RGBA writer, D2H of the output if asked
int jxlamd_decoder::launch_rest(FrameSlot &S, int parts, bool upload_B) {
  const FramePlan &plan = S.plan;
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  if ((parts & 1) || upload_B) {   // the buffer table goes up once per decode (page-locked staging)
    HIPCHECK(S.dB.ensure(sizeof(DevBuffers)));
    HIPCHECK(S.h_B.ensure(sizeof(DevBuffers)));
    memcpy(S.h_B.p, &S.B, sizeof(DevBuffers));
    HIPCHECK(hipMemcpyAsync(S.dB.p, S.h_B.p, sizeof(DevBuffers), hipMemcpyHostToDevice, stream));
  }
  int stage_mask = 0;
  if (F->gab) stage_mask |= 1;
  if (F->epf_iters >= 3) stage_mask |= 2;
  if (F->epf_iters >= 1) stage_mask |= 4;
  if (F->epf_iters >= 2) stage_mask |= 8;
  if (F->epf_iters <= 2) stage_mask |= sweep_stage_bit(*F, (int)S.pi.out_bits, S.post_active && S.post_fused);     // column-sweep instantiation
  if (!F->gab && !F->epf_iters) stage_mask |= 1 << 4;
  if (F->compose) stage_mask = (stage_mask & 15) | 32;           // stage by stage into the planes; patches, reference copy and writer follow (launch_compose_tail)
  if (S.post_active && S.post_fused) stage_mask |= 64 | 128;     // the last filter stage emits the Bitmap format (k_filter_b<3, 1> / <3, 2>); 128: no other frame in this launch
  if (parts & 1) HIPCHECK(huge_scratch.ensure((size_t)kHugeSlots * 2 * 65536 * 4));      // (a single decode always launches the DCT128 / DCT256 kernel: alone it costs microseconds)
  launch_rest_batch((const DevBuffers *)S.dB.p, (const uint8_t *)stat.p, 1, plan.xb * plan.yb, plan.width, plan.height, stage_mask, /*expect_large=*/true, parts, stream, false, (float *)huge_scratch.p);
  return JXLAMD_OK;
}

int mod_group_pool_bytes(const FramePlan &plan);
// Extra channels of a VarDCT frame (alpha).  The GlobalModular part (meta channels, channels that fit one group) is decoded
// BEFORE the LF stage (launch_mod_global at the call sites: in a single-section frame LfGroup 0 starts where it ends); this is the
// rest: the ModularGroup stream that follows each group's AC stream, then the inverse global transforms.  The writer reads the planes.
int jxlamd_decoder::launch_extra_channels(FrameSlot &S) {
  const FramePlan &plan = S.plan;
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  if (F->mod_first_group_ch < F->mod_nch) launch_mod_groups(S.B, plan.num_groups, mod_group_pool_bytes(plan), stream);
  for (int o = 0; o < F->mod_nops; o++) launch_mod_op(S.B, o, (size_t)(F->mod_op_kind[o] == 0 ? F->mod_op_y[o] : F->mod_op_c[o]), stream);
  return JXLAMD_OK;
}

// Modular-encoded (lossless) frame: GlobalModular stream, per-group streams, inverse global transforms, writer
int jxlamd_decoder::launch_modular(FrameSlot &S) {
  const FramePlan &plan = S.plan;
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  launch_mod_global(S.B, mod_group_pool_bytes(plan), stream);
  if (F->mod_lf_nch > 0) launch_mod_lfgroups(S.B, plan.num_lf_groups, stream);
  if (F->mod_first_group_ch < F->mod_nch) launch_mod_groups(S.B, plan.num_groups, mod_group_pool_bytes(plan), stream);
  for (int o = 0; o < F->mod_nops; o++) launch_mod_op(S.B, o, (size_t)(F->mod_op_kind[o] == 0 ? F->mod_op_y[o] : F->mod_op_c[o]), stream);
  if (!F->compose) { launch_mod_write(S.B, plan.width, plan.height, (int)S.pi.out_bits, stream); return JXLAMD_OK; }
  launch_mod_to_planes(S.B, plan.width, plan.height, stream);
  if (F->xyb_modular && (F->gab || F->epf_iters)) { int rc = launch_rest(S, 2, /*upload_B=*/true); if (rc) return rc; }      // loop filters of an XYB frame, stage by stage
  return launch_compose_tail(S);
}

// composed frames (dev_compose.h): patches onto the filtered planes, copy into the frame's reference slot, stand-alone writer
int jxlamd_decoder::launch_compose_tail(FrameSlot &S) {
  const FramePlan &plan = S.plan;
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  if (F->subsampled) launch_chroma_upsample(S.B, plan.width, plan.height, stream);      // recompressed JPEG: chroma to full resolution (no loop filters in between)
  launch_patch_blend(S.B, F->num_patches, plan.patch_max_px, stream);
  if (F->num_spline_segs > 0) launch_splines(S.B, plan.width, plan.height, stream);
  if (F->noise && F->upsampling == 1) launch_noise(S.B, plan.width, plan.height, stream);      // after the patches, before the colour transform (libjxl's stage order); an upsampled frame: after the upsampling
  if (F->blend) {
    if (F->alpha_up > 1 && F->mod_out[3] >= 0) launch_upsample_alpha(S.B, (const uint8_t *)stat.p, F->full_w, F->full_h, stream);      // an upsampled frame is blended at its full resolution
    if (F->upsampling > 1) launch_upsample_and_write(S.B, (const uint8_t *)stat.p, F->full_w, F->full_h, F->noise != 0, /*write=*/false, stream);
    // a frame of an animation over its canvas (dev_compose.h: blend_canvas_pixel): the background is a reference slot's canvas, the result goes out and / or
    // becomes the new canvas of the frame's slot — in place when it is the slot it was read from (every pixel reads before it writes)
    DevBuffers Bb = S.B;
    if (plan.save_slot >= 0 && plan.save_canvas) {
      const int k = plan.save_slot;
      const size_t n = (size_t)F->canvas_w * (size_t)F->canvas_h;
      const bool has_alpha = (F->has_ec || F->is_modular) && F->mod_out[3] >= 0;
      const bool in_place = F->bl_src == k && ref_w[k] == F->canvas_w && ref_h[k] == F->canvas_h && ref_store[k].p;
      if (!in_place) {
        if (F->bl_src == k) { set_error("blending: canvas geometry changed"); return JXLAMD_ERR_INVALID; }
        HIPCHECK(ref_store[k].ensure(4 * n * 4));
      }
      ref_w[k] = F->canvas_w; ref_h[k] = F->canvas_h; ref_alpha[k] = has_alpha;
      for (int c = 0; c < 4; c++) Bb.canvas_save[c] = (c < 3 || has_alpha) ? (float *)ref_store[k].p + (size_t)c * n : nullptr;
      if (in_place) for (int c = 0; c < 3; c++) Bb.ref[k][c] = Bb.canvas_save[c];
    }
    launch_blend_canvas(Bb, (const uint8_t *)stat.p, F->canvas_w, F->canvas_h, stream);
    return JXLAMD_OK;
  }
  if (plan.save_slot >= 0) {
    const int k = plan.save_slot;
    const size_t n = (size_t)plan.width * (size_t)plan.height;
    HIPCHECK(ref_store[k].ensure(3 * n * 4));
    ref_w[k] = plan.width; ref_h[k] = plan.height; ref_alpha[k] = false;
    launch_save_ref(S.B, plan.width, plan.height, (float *)ref_store[k].p, stream);
  }
  if (F->no_output) return JXLAMD_OK;
  if (F->alpha_up > 1 && F->mod_out[3] >= 0) launch_upsample_alpha(S.B, (const uint8_t *)stat.p, F->full_w, F->full_h, stream);
  if (F->upsampling > 1) launch_upsample_and_write(S.B, (const uint8_t *)stat.p, F->full_w, F->full_h, F->noise != 0, /*write=*/true, stream);
  else launch_compose_write(S.B, (const uint8_t *)stat.p, plan.width, plan.height, stream);
  return JXLAMD_OK;
}

// A10 + A11 behind the writer for a frame whose last filter stage could not take them (column-sweep frames, Modular and composed frames): one pass over
// the RGBA the writer stored (k_post_fused), into the caller's buffer
void jxlamd_decoder::launch_post_pass(FrameSlot &S) {
  if (!S.post_active || S.post_fused) return;
  const bool is16 = S.pi.out_bits == 16;
  launch_post_fused((PostKind)S.post_kind, S.post_tmp.p, S.pi.xsize * (is16 ? 8u : 4u), S.post_final, S.post_stride, S.pi.xsize, S.pi.ysize, S.post_runs ? &post_dev : nullptr,
                    S.post_premul, S.post_depth, S.post_att, stream);
}

int jxlamd_decoder::collect(FrameSlot &S, uint32_t flags) {
  uint32_t derr = 0;
  uint32_t head[20] = {0};                              // flags word and, at byte 64, the size-class block counters
  HIPCHECK(h_flags.ensure(256));                        // page-locked: a copy into pageable memory is synchronous inside the runtime (see band.hip)
  launch_post_pass(S);
  HIPCHECK(hipMemcpyAsync(h_flags.p, S.B.err, sizeof(head), hipMemcpyDeviceToHost, stream));
  if (S.host_out) HIPCHECK(hipMemcpyAsync(S.host_out, S.post_active ? S.post_final : S.d_out, S.post_active ? S.post_bytes : S.out_bytes, hipMemcpyDeviceToHost, stream));
  HIPCHECK(hipStreamSynchronize(stream));
  HIPCHECK(hipGetLastError());
  (void)flags;
  memcpy(head, h_flags.p, sizeof(head));
  derr = head[0];
  serial_streams += head[2]; block_tree_channels += head[3];
  const int pool_of_this_attempt = lf_pool_bytes;
  if (!S.plan.modular) lf_pool_bytes = std::max(std::max(lf_pool_floor, g_lf_pool_floor.load()), lf_pool_clamp(head[1]));
  if (!S.plan.modular && head[17] > 0) large_blocks_seen = true;      // big_count[1]: varblocks with 2048 / 4096 coefficients
  if ((derr & kErrNeedPool) && pool_of_this_attempt < kModPoolBytes) { lf_pool_floor = kModPoolBytes; g_lf_pool_floor.store(kModPoolBytes); lf_pool_bytes = kModPoolBytes; return kRetryPool; }      // (the other flags of an attempt that stopped for the pool say nothing: see decode_batch_once)
  if ((derr & kErrNeedGeneral) && !lf_general) return kRetryGeneral;
  if (derr) { set_error("corrupt or unsupported stream (device flags " + std::to_string(derr) + ")"); return dev_err_class(derr); }
  S.coef_clean = !S.plan.modular;
  return JXLAMD_OK;
}

int jxlamd_decoder::decode(const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info, int frame) {
  target_frame = frame;
  int rc0 = decode_once(jxl, size, jxl_dev, flags, out_ptr, out_cap, info);
  if (rc0 == kRetryPool) { pool_retries++; rc0 = decode_once(jxl, size, jxl_dev, flags, out_ptr, out_cap, info); }       // with the largest table pool
  if (rc0 == kRetryPool) { set_error("LF table pool: the stream asked for a larger pool twice"); return JXLAMD_ERR_DEVICE; }
  if (rc0 != kRetryGeneral) return rc0;
  lf_general = true; general_retries++;
  return decode_once(jxl, size, jxl_dev, flags, out_ptr, out_cap, info);
}

int jxlamd_decoder::decode_once(const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info) {
  HIPCHECK(hipSetDevice(device));
  FrameSlot &S = slot(0);
  S.plan = FramePlan();
  (void)plan_parse(jxl, size, &S.plan, target_frame);
  if (!S.plan.error.empty() || S.plan.tables.empty()) { set_error(S.plan.error); return err_class(S.plan.error); }
  ref_cursor = 0;
  int rc = decode_refs(S, flags);
  if (rc) return rc;
  rc = prepare(S, jxl, size, jxl_dev, flags, out_ptr, out_cap, info, /*parsed=*/true);
  if (rc) return rc;
  return run_frame(S, flags, /*single_latency=*/true);
}

// The frames a patch dictionary draws on (FramePlan::refs), in file order: each one is decoded like a frame of its own — same kernels — and
// its image copied into its reference slot instead of being written out.
int jxlamd_decoder::decode_refs(FrameSlot &main, uint32_t flags, bool deferred) {
  // deferred (frames of a flight whose references are Modular frames — a screenshot's patch sprites): no flag read-back and no synchronisation per reference frame (655
  // one-wave launches each behind a host round trip per mixed run in round 5); the frame's kernels are queued, its flags OR-ed into ref_err_dev, the batch checks that word once
  for (const auto &r : main.plan.refs) if (!r || !r->modular) deferred = false;
  for (size_t k = 0; k < main.plan.refs.size(); k++) {
    const size_t si = deferred ? ref_cursor++ : ref_cursor + k;      // (synchronous reference decodes reuse the slots behind the deferred ones of this batch, whose staging may still be read)
    while (ref_slots.size() <= si) ref_slots.push_back(new FrameSlot());
    FrameSlot &RS = *ref_slots[si];
    RS.plan = *main.plan.refs[k];
    int rc = prepare(RS, nullptr, 0, nullptr, (flags & ~(uint32_t)(JXLAMD_IN_DEVICE | JXLAMD_OUT_DEVICE)) | JXLAMD_NO_SIZE_GUARD, nullptr, ~(size_t)0, nullptr, /*parsed=*/true);
    if (rc) return rc;
    if (deferred) {
      rc = launch_modular(RS); if (rc) return rc;
      hipLaunchKernelGGL(k_or_flags, dim3(1), dim3(64), 0, stream, (const uint32_t *)RS.B.err, (uint32_t *)ref_err_dev.p);
      refs_deferred = true;
      continue;
    }
    rc = run_frame(RS, flags, false);
    if (rc) return rc;
  }
  return JXLAMD_OK;
}

// one prepared frame through every stage on this context's stream, then the flag readback (and the output copy if the caller's buffer is on the host)
int jxlamd_decoder::run_frame(FrameSlot &S, uint32_t flags, bool single_latency) {
  int rc;
  HIPCHECK(hipEventRecord(ev[0], stream));
  if (S.plan.modular) {
    rc = launch_modular(S); if (rc) return rc;
    for (int i = 1; i <= 4; i++) HIPCHECK(hipEventRecord(ev[i], stream));
    rc = collect(S, flags);
    for (int i = 0; i < 4; i++) timing[i] = 0;
    (void)hipEventElapsedTime(&timing[4], ev[0], ev[4]);
    return rc;
  }
  if (S.plan.has_ec) launch_mod_global(S.B, mod_group_pool_bytes(S.plan), stream);
  // a single decode is the latency path and has the chip to itself: the general build (181 VGPRs) runs a lone stream ~5 % faster than the lean
  // one (125), whose smaller footprint only pays next to the data-parallel kernels of other flights
  launch_lf_groups(S.B, S.A, S.plan.num_lf_groups, lf_pool_bytes, /*general=*/single_latency || lf_general, stream);
  if (S.plan.single_section) { rc = finish_single_section(S); if (rc) return rc; }
  HIPCHECK(hipEventRecord(ev[1], stream));
  launch_lf_smooth(S.B, S.plan.xb, S.plan.yb, stream);
  launch_pass_groups(S.B, S.plan.num_groups, stream);
  if (S.plan.has_ec) launch_extra_channels(S);
  HIPCHECK(hipEventRecord(ev[2], stream));
  rc = launch_rest(S, 1); if (rc) return rc;
  HIPCHECK(hipEventRecord(ev[3], stream));
  rc = launch_rest(S, 2); if (rc) return rc;
  if (S.plan.compose) { rc = launch_compose_tail(S); if (rc) return rc; }      // (plan flag: finish_single_section re-packs the tables blob, a DevFrame pointer taken earlier would dangle)
  HIPCHECK(hipEventRecord(ev[4], stream));
  rc = collect(S, flags);
  for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&timing[i], ev[i], ev[i + 1]);
  (void)hipEventElapsedTime(&timing[4], ev[0], ev[4]);
  return rc;
}

// LDS table pool (bytes) the ModularGroup streams of a frame should run with — the kernels clamp it to kModPoolMin .. kModPoolBytes; any size decodes the
// same pixels, what fits stays out of HBM.  Streams with their own trees (no global tree in the frame) get the full pool: their tables are only known on the
// device.  With the frame's global tree and code: [alias tables, when the LDS form takes them | context map | the tree: its head for a small one, the block
// form (dev_modular.h: big_tree_build) of the largest pruned tree any sampled (channel, stream) reaches for one beyond a ballot].
int mod_group_pool_bytes(const FramePlan &plan) {
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  if (plan.tables.empty() || F->tree_count <= 0) return kModPoolBytes;
  const DevEC &ec = F->tree_ec;
  const int alias_bytes = ec.use_prefix ? 0 : (int)((size_t)(ec.num_clusters << ec.log_alpha) * sizeof(DevAlias));
  const bool alias_lds = !ec.use_prefix && ec.num_clusters <= kLocMaxClusters && alias_bytes <= kModPoolBytes;
  int used = (alias_lds ? alias_bytes : 0) + ((ec.num_ctx + (ec.lz77 ? 1 : 0) + 7) & ~7);
  if (used > kModPoolBytes) return kModPoolBytes;
  const DevTreeNode *tree = (const DevTreeNode *)(plan.tables.data() + F->tree_off);
  int ni_max = 0, nl_max = 0;
  const int nch = std::min(std::max(F->mod_nch - F->mod_first_group_ch, 1), (int)kModMaxGroupCh), ng = std::max(plan.num_groups, 1);
  std::vector<int> stack;
  for (int pass = 0; pass < std::max(F->num_passes, 1); pass++)
    for (int gs = 0; gs < 3; gs++) {
      const int g = gs == 0 ? 0 : gs == 1 ? ng / 2 : ng - 1;
      const int stream = 1 + 3 * F->num_lf_groups + 17 + pass * ng + g;
      for (int c = 0; c < nch; c++) {
        int ni = 0, nl = 0; size_t guard = 0;
        stack.assign(1, 0);
        while (!stack.empty() && guard++ < (size_t)4 * (size_t)F->tree_count + 16) {
          const int idx = stack.back(); stack.pop_back();
          if (idx < 0 || idx >= F->tree_count) break;
          const DevTreeNode &nd = tree[idx];
          if (nd.prop < 0) { nl++; continue; }
          if (nd.prop == 0 || nd.prop == 1) { stack.push_back((nd.prop == 0 ? c : stream) > nd.splitval ? nd.lchild : nd.rchild); continue; }
          ni++; stack.push_back(nd.lchild); stack.push_back(nd.rchild);
        }
        ni_max = std::max(ni_max, ni); nl_max = std::max(nl_max, nl);
      }
    }
  int tree_bytes;
  if (ni_max <= 64 && nl_max <= 64) return kModPoolBytes;      // one ballot: the specialised loops re-pack the tables of the clusters a channel uses into whatever the pool has (measured: a lossless e3 4K frame 81 -> 85 ms with a pool cut to the stream's own tables)
  tree_bytes = (int)sizeof(DevBigHdr) + 8 * ni_max + 28 * (nl_max + ni_max / 6 + 16) + 512;                  // nodes + (masks, word, queue entry) per exit and per block, with room
  return ((used + 15) & ~15) + tree_bytes + 64;
}

// host-side twin of flat_frame_ok (dev_pass_flat.h)
bool frame_flat_ok(const FramePlan &plan) {
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  if (plan.modular || plan.tables.empty()) return false;
  for (int p = 0; p < F->num_passes; p++) if (F->hf_ec[p].use_prefix || F->hf_ec[p].num_clusters > 256) return false;
  return true;
}

// A frame whose groups are not a multiple of 64 ends with a wave of few lanes (4K: 135 = 64 + 64 + 7) that holds a wave's LDS and registers as long as a full one.  When the
// tail is short (<= 16 groups) and each tail group, added to the group of the lane it would follow (the last lanes of the last full wave: with a frame height that is not a
// multiple of 256 both are groups of the short bottom row), stays within the frame's largest group — judged by the sections' bytes, which is what a stream's length follows —
// the tail rides on those lanes and the wave is not launched.
static bool flat_tail_chain(const FramePlan &plan) {
  static const bool on = !(getenv("JXLAMD_PASS_CHAIN") && atoi(getenv("JXLAMD_PASS_CHAIN")) == 0);      // A/B switch for measurements
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  const int G = plan.num_groups, tail = G % 64;
  if (!on || plan.tables.empty() || F->num_passes != 1 || F->nsec == 1 || G < 64 || tail == 0 || tail > 16) return false;
  const DevSection *secs = (const DevSection *)(plan.tables.data() + F->sec_off);
  const DevSection *pg = secs + 2 + F->num_lf_groups;      // PassGroup sections of the one pass
  uint32_t largest = 0;
  for (int g = 0; g < G; g++) largest = std::max(largest, pg[g].size);
  const int full = G - tail;
  for (int j = 0; j < tail; j++) if ((uint64_t)pg[full - tail + j].size + (uint64_t)pg[full + j].size > (uint64_t)largest + largest / 16) return false;
  return true;
}
// chain[k] (round 6): frame k's tail groups (num_groups % 64 of them) ride as SECOND groups of the last lanes of its last full wave instead of in a wave of their own
// (flat_tail_chain decides; dev_pass_flat.h: pass_group_flat's g2)
std::vector<int> flat_wave_map(const std::vector<int> &ngroups, const std::vector<int> &chain) {
  std::vector<int> per_xcd[8];
  for (size_t k = 0; k < ngroups.size(); k++) {
    const int tail = (k < chain.size() && chain[k]) ? ngroups[k] % 64 : 0, full = ngroups[k] - tail;
    for (int g = 0; g < full; g += 64) {
      std::vector<int> &v = per_xcd[k & 7];
      const int cnt = std::min(64, full - g);
      v.push_back((int)k); v.push_back(g); v.push_back(cnt | ((g + cnt == full ? tail : 0) << 8));
    }
  }
  size_t rows = 0;
  for (const auto &v : per_xcd) rows = std::max(rows, v.size() / 3);
  std::vector<int> out;
  out.reserve(rows * 24);
  for (size_t r = 0; r < rows; r++)
    for (int x = 0; x < 8; x++) {
      const std::vector<int> &v = per_xcd[x];
      if (3 * r < v.size()) { out.push_back(v[3 * r]); out.push_back(v[3 * r + 1]); out.push_back(v[3 * r + 2]); }
      else { out.push_back(0); out.push_back(0); out.push_back(0); }       // padding wavefront (exits at once): keeps workgroup index % 8 == frame % 8
    }
  return out;
}

// n independent frames: the entropy stages of ALL frames go into ONE launch each (grid = sum of LF groups / groups
// over the batch), so that their serial streams run side by side on the chip; the cheap data-parallel stages follow
// per frame on the same stream.  Frames that need the single-section round trip are decoded one by one.
int jxlamd_decoder::decode_batch(int n, const uint8_t *const *jxl, const size_t *sizes, const void *const *jxl_dev, uint32_t flags,
                                 void *const *outs, const size_t *caps, jxlamd_info *infos) {
  int rc0 = JXLAMD_OK, moved = 0;
  bool pool_retried = false, general_retried = false;
  dense_flight = false;
  for (;;) {
    rc0 = decode_batch_once(n, jxl, sizes, jxl_dev, flags, outs, caps, infos);
    if (rc0 == kRetryMoved && moved++ < 16) continue;
    if (rc0 == kRetryPool && !pool_retried) { pool_retried = true; pool_retries++; continue; }                 // with the largest table pool
    if (rc0 == kRetryGeneral && !general_retried) { general_retried = true; lf_general = true; general_retries++; continue; }   // some frame needs a general lock-step loop: this context runs the general LF build from now on
    if (rc0 == kRetryDense && !dense_flight) { dense_flight = true; sparse_misses++; continue; }               // with the dense coefficient planes
    if (rc0 == kRetryHuge && !huge_blocks_seen) { huge_blocks_seen = true; continue; }                         // with k_recon_huge_b in the launch list
    break;
  }
  dense_flight = false;
  if (rc0 == kRetryPool || rc0 == kRetryMoved || rc0 == kRetryGeneral || rc0 == kRetryDense || rc0 == kRetryHuge) { set_error("LF table pool / shared HF pools / coefficient lists: the flight was restarted too often"); return JXLAMD_ERR_DEVICE; }
  return rc0;
}

int jxlamd_decoder::decode_batch_once(int n, const uint8_t *const *jxl, const size_t *sizes, const void *const *jxl_dev, uint32_t flags,
                                      void *const *outs, const size_t *caps, jxlamd_info *infos) {
  HIPCHECK(hipSetDevice(device));
  std::vector<int> batched, mod_batched;
  // JXLAMD_TRACE_FLIGHT=1: wall-clock split of every flight on stderr (host parse / per-frame prepare + uploads / launches / wait)
  static const bool trace = getenv("JXLAMD_TRACE_FLIGHT") && atoi(getenv("JXLAMD_TRACE_FLIGHT"));
  const auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  // host parse of all frames in parallel (pure CPU work, independent per frame)
  for (int i = 0; i < n; i++) { slot((size_t)i).plan.reset(); }
  {
    const int nthr = n < 16 ? n : 16;
    std::vector<std::thread> th;
    std::atomic<int> next{0};
    for (int t = 0; t < nthr; t++) th.emplace_back([&] { for (;;) { int i = next.fetch_add(1); if (i >= n) return; (void)plan_parse(jxl[i], sizes[i], &slots[(size_t)i]->plan); } });
    for (auto &t : th) t.join();
  }
  const double t_parsed = now();
  if (trace) HIPCHECK(hipEventRecord(ev[5], stream));
  HIPCHECK(ref_err_dev.ensure(256)); HIPCHECK(h_ref_err.ensure(256));
  HIPCHECK(hipMemsetAsync(ref_err_dev.p, 0, 4, stream));
  ref_cursor = 0; refs_deferred = false;
  for (int i = 0; i < n; i++) {
    FrameSlot &S = slot((size_t)i);
    // Frames that draw on other frames (patch dictionaries, LF frames, animation layers over a canvas) keep state in the context's reference slots and are
    // decoded one by one, like a single decode.  A composed frame that stands alone — coded at a lower resolution and upsampled (the reference's quality
    // <= 12), with noise, with splines — rides in the flight: its entropy stages in the flight's launches, its
    // filters stage by stage into the plane set it borrows, its composition stages (launch_compose_tail) right behind its sub-batch's filters.
    // A frame whose references are plain patch sources (a screenshot: the patch dictionary's sprite frame, then the image) rides too: its reference frames are
    // decoded first — one by one — and their images move from the context's slots into the frame's own (FrameSlot::own_ref), so that the next such frame of
    // the flight does not overwrite what this one's patch stage reads much later.
    bool rides = false, rides_with_refs = false;
    if (S.plan.compose && S.plan.error.empty() && !S.plan.tables.empty() && !S.plan.modular && !S.plan.single_section && S.plan.save_slot < 0) {
      const DevFrame *Fc = (const DevFrame *)S.plan.tables.data();
      static const bool compose_in_flights = !(getenv("JXLAMD_COMPOSE_IN_FLIGHTS") && atoi(getenv("JXLAMD_COMPOSE_IN_FLIGHTS")) == 0);      // A/B switch for measurements (2: only frames without references)
      static const bool refs_in_flights = !(getenv("JXLAMD_COMPOSE_IN_FLIGHTS") && atoi(getenv("JXLAMD_COMPOSE_IN_FLIGHTS")) == 2);
      rides = compose_in_flights && !Fc->blend && !Fc->use_lf_frame && !Fc->no_output && !Fc->subsampled;      // (subsampled chroma: the flights' list-driven reconstruction kernels do not place such blocks)
      if (S.plan.refs.empty()) rides = rides && Fc->num_patches == 0;
      else {
        rides = rides && refs_in_flights;
        for (const auto &r : S.plan.refs) if (!r || r->save_canvas || r->save_slot < 0 || r->save_slot > 3) rides = false;
        rides_with_refs = rides;
      }
    }
    const bool composed = (S.plan.compose || !S.plan.refs.empty()) && !rides;
    if ((composed || rides_with_refs) && S.plan.error.empty()) {
      static const bool refs_async = !(getenv("JXLAMD_REFS_ASYNC") && atoi(getenv("JXLAMD_REFS_ASYNC")) == 0);      // A/B switch for measurements
      int rc = decode_refs(S, flags, /*deferred=*/rides_with_refs && refs_async); if (rc) return rc;
    }
    int rc = prepare(S, jxl[i], sizes[i], jxl_dev ? jxl_dev[i] : nullptr, jxl_dev && jxl_dev[i] ? (flags | JXLAMD_IN_DEVICE) : (flags & ~JXLAMD_IN_DEVICE),
                     outs[i], caps[i], infos ? &infos[i] : nullptr, /*parsed=*/true,
                     /*own_planes=*/S.plan.modular || S.plan.single_section || composed);
    if (rc) return rc;
    if (rides_with_refs) for (int k = 0; k < 4; k++) if (S.B.ref[k][0]) {      // (prepare pointed the frame at the context's slots: they are this frame's from here on)
      S.own_ref[k].swap(ref_store[k]);
      ref_w[k] = ref_h[k] = 0; ref_alpha[k] = false;      // the slot now holds an older image (or nothing): not a reference any more
    }
    if (composed) { rc = run_frame(S, flags, false); if (rc) return rc; continue; }
    if (S.plan.modular) { mod_batched.push_back(i); continue; }
    if (S.plan.single_section) {
      if (S.plan.has_ec) launch_mod_global(S.B, mod_group_pool_bytes(S.plan), stream);
      launch_lf_groups(S.B, S.A, S.plan.num_lf_groups, lf_pool_bytes, lf_general, stream);
      if (S.plan.single_section) { rc = finish_single_section(S); if (rc) return rc; }
      launch_lf_smooth(S.B, S.plan.xb, S.plan.yb, stream);
      launch_pass_groups(S.B, S.plan.num_groups, stream);
      if (S.plan.has_ec) launch_extra_channels(S);
      launch_rest(S);
      rc = collect(S, flags); if (rc) return rc;
    } else batched.push_back(i);
  }
  if (refs_deferred) {      // the reference frames queued above: one read-back for all of them
    HIPCHECK(hipMemcpyAsync(h_ref_err.p, ref_err_dev.p, 4, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    uint32_t rerr = 0; memcpy(&rerr, h_ref_err.p, 4);
    if (rerr) { set_error("corrupt or unsupported stream (device flags " + std::to_string(rerr) + ", reference frame of a flight)"); return dev_err_class(rerr); }
  }
  // ---- Modular-encoded (lossless) frames of the batch: their streams are as serial as the LF streams, so they too go into one
  // launch per stage over all of them (GlobalModular streams, then every 256x256 group stream, inverse transforms, writer)
  if (!mod_batched.empty()) {
    std::vector<DevBuffers> hb; std::vector<int> gmap;
    int max_ops = 0, mw = 0, mh = 0, mod_pool = kModPoolMin;
    for (size_t k = 0; k < mod_batched.size(); k++) {
      const FrameSlot &S = slot((size_t)mod_batched[k]);
      mod_pool = std::max(mod_pool, mod_group_pool_bytes(S.plan));
      const DevFrame *F = (const DevFrame *)S.plan.tables.data();
      hb.push_back(S.B);
      if (F->mod_first_group_ch < F->mod_nch) for (int g = 0; g < S.plan.num_groups; g++) { gmap.push_back((int)k); gmap.push_back(g); }
      max_ops = std::max(max_ops, (int)F->mod_nops); mw = std::max(mw, S.plan.width); mh = std::max(mh, S.plan.height);
    }
    const size_t o_g = (hb.size() * sizeof(DevBuffers) + 255) & ~(size_t)255, total = o_g + gmap.size() * 4 + 4;
    HIPCHECK(mod_tab.ensure(total));
    HIPCHECK(h_mod_tab.ensure(total));
    memcpy(h_mod_tab.p, hb.data(), hb.size() * sizeof(DevBuffers));
    if (!gmap.empty()) memcpy((uint8_t *)h_mod_tab.p + o_g, gmap.data(), gmap.size() * 4);
    HIPCHECK(hipMemcpyAsync(mod_tab.p, h_mod_tab.p, total, hipMemcpyHostToDevice, stream));
    launch_modular_batch((const DevBuffers *)mod_tab.p, (const int *)((uint8_t *)mod_tab.p + o_g), (int)hb.size(), (int)gmap.size() / 2, max_ops, mw, mh, mod_pool, stream);
    int first_mod_rc = JXLAMD_OK;
    for (int i : mod_batched) { int rc = collect(slot((size_t)i), flags); if (rc && !first_mod_rc) first_mod_rc = rc; }
    if (first_mod_rc) return first_mod_rc;
  }
  if (batched.empty()) return JXLAMD_OK;
  const double t_prepared = now();
  // ---- the flight.  Two phases with different buffer lifetimes:
  //   LF phase : ONE launch decodes the LfGroup streams of all frames of the flight.  Its outputs are small (per-cell planes,
  //              LF image: ~4 MB per 4K frame), so a flight can be long (hundreds of frames) — which is what the
  //              latency-bound entropy kernel needs.  (A lane-per-stream variant of this kernel — 64 streams per
  //              wavefront, state in HBM, no LDS — was measured in round 1: 1.5 s per launch alone, 3-3.8 s next to
  //              other flights' kernels, i.e. not yet enough frames in flight to win)
  //   HF phase : PassGroup decode + reconstruction + filters + writer in sub-flights of hf_sets frames that share a pool
  //              of coefficient sets (106 MB per 4K frame) and, inside, sub-batches of plane_sets frames that share the
  //              f32 pixel planes (200 MB per frame).  Everything is stream-ordered, so a set is reused only after its
  //              previous user has been reconstructed (and the reconstruction leaves the coefficient planes all-zero).
  static const int plane_sets_env = getenv("JXLAMD_PLANE_SETS") ? std::max(1, atoi(getenv("JXLAMD_PLANE_SETS"))) : 0;
  static const size_t plane_budget = (size_t)(getenv("JXLAMD_PLANE_BUDGET_MB") ? std::max(64, atoi(getenv("JXLAMD_PLANE_BUDGET_MB"))) : 3200) << 20;
  static const int hf_sets_env = getenv("JXLAMD_HF_SETS") ? atoi(getenv("JXLAMD_HF_SETS")) : 128;
  const int nb = (int)batched.size();
  size_t max_npx = 0, max_coef = 0;
  int max_cells = 0, max_w = 0, max_h = 0, stage_mask = 0;
  bool all_post = true, compose_filters = false;
  for (int i : batched) {
    const FrameSlot &S = slot((size_t)i);
    const DevFrame *F = (const DevFrame *)S.plan.tables.data();
    if (!F->gab && !F->epf_iters && !F->compose) stage_mask |= 1 << 4;       // no filter stage to fuse the writer into: stand-alone writer launch
    max_npx = std::max(max_npx, (size_t)S.plan.xb * S.plan.yb * 64);
    max_coef = std::max(max_coef, (size_t)S.plan.num_groups * 65536);
    max_cells = std::max(max_cells, S.plan.xb * S.plan.yb); max_w = std::max(max_w, S.plan.width); max_h = std::max(max_h, S.plan.height);
    if (F->gab) stage_mask |= 1;
    if (F->epf_iters >= 3) stage_mask |= 2;
    if (F->epf_iters >= 1) stage_mask |= 4;
    if (F->epf_iters >= 2) stage_mask |= 8;
    if (F->compose) { stage_mask |= 32; if (F->gab || F->epf_iters) compose_filters = true; }      // stage by stage into the planes (the per-stage kernels skip the sweep's frames and vice versa)
    else if (F->epf_iters <= 2) stage_mask |= sweep_stage_bit(*F, (int)S.pi.out_bits, S.post_active && S.post_fused);     // column-sweep instantiation
    if (S.post_active && S.post_fused) stage_mask |= 64;      // the last filter stage emits the Bitmap format (k_filter_b<3, 1> / <3, 2>)
    else all_post = false;
  }
  if (all_post && (stage_mask & 64)) stage_mask |= 128;       // every frame of the flight: the plain instantiation of the last stage is not launched
  // Pixel planes of a sub-batch: the column sweep reads the reconstruction's planes and writes pixels — only frames with three EPF iterations (per-stage
  // kernels, ping-pong) need the second plane set.  As many frames per sub-batch as the budget holds (JXLAMD_PLANE_BUDGET_MB, 3.2 GB: 32 4K frames of
  // three planes).  Measured (round 5, quick bench, one box): sub-batches of 16 / 32 / 64 frames 13 020 - 13 050 / 12 940 / 12 590 - 13 150 MP/s — the
  // sub-batch size is not what the rate depends on, although every launch of a busy stream takes 1.5 - 4 ms whatever it computes (k_recon_large_b:
  // 0.07 ms alone, 4 ms in the mix).
  const int planes_per_set = ((stage_mask & 2) || compose_filters) ? 6 : 3;      // (a composed frame's filter stages ping-pong too)
  const int plane_sets = plane_sets_env ? plane_sets_env : (int)std::max<size_t>(1, std::min<size_t>(128, plane_budget / ((size_t)planes_per_set * std::max<size_t>(max_npx, 1) * 4)));
  const int hf_sets = std::max(plane_sets, hf_sets_env / plane_sets * plane_sets);
  // HF-phase memory (HfPools): sized now — the frames' DevBuffers carry its addresses — but only held from the PassGroup stage on, so that
  // contexts sharing it overlap one's LF stage with the other's HF phase
  const int used_sets = std::min(hf_sets, nb);
  // Sparse coefficient lists (DevBuffers::coef_sp) when every frame of the flight is a single-pass frame the flat PassGroup kernel takes: an arena per
  // group sized from the bytes of its section (sparse_group_entries), ~16 MB per 4K q90 frame instead of the 106 MB of dense planes — written once,
  // read once, nothing to clear.  The dense pool is then not even allocated.
  bool sparse = sparse_enabled && !dense_flight && sparse_misses < 3;
  int flight_groups = 0;
  for (int i : batched) {
    const FrameSlot &S = slot((size_t)i);
    const DevFrame *F = (const DevFrame *)S.plan.tables.data();
    sparse = sparse && frame_flat_ok(S.plan) && F->num_passes == 1 && !F->subsampled;
    flight_groups += S.plan.num_groups;
  }
  sparse = sparse && flight_groups >= flat_min_groups;
  last_flight_sparse = sparse;
  static const int sparse_cap_override = getenv("JXLAMD_SPARSE_CAP") ? atoi(getenv("JXLAMD_SPARSE_CAP")) : 0;      // test hook: entries per group (a tiny value forces the dense retry)
  std::vector<uint32_t> sp_groups;                   // per frame: [num_groups + 1] first entry of each group's arena
  std::vector<size_t> sp_frame_off, sp_tab_off;      // per frame: its arena inside the pool (entries); its offsets inside sp_groups
  size_t sp_total = 0;
  if (sparse) for (int i : batched) {
    const FrameSlot &S = slot((size_t)i);
    const DevFrame *F = (const DevFrame *)S.plan.tables.data();
    const DevSection *secs = (const DevSection *)(S.plan.tables.data() + F->sec_off);
    sp_frame_off.push_back(sp_total); sp_tab_off.push_back(sp_groups.size());
    uint32_t at = 0;
    for (int g = 0; g < S.plan.num_groups; g++) { sp_groups.push_back(at); at += sparse_cap_override > 0 ? (uint32_t)sparse_cap_override : sparse_group_entries(secs[2 + F->num_lf_groups + g].size); }
    sp_groups.push_back(at);
    sp_total += ((size_t)at + 63) & ~(size_t)63;
  }
  const size_t coef_need = sparse ? 0 : (size_t)used_sets * 3 * max_coef * 4;
  uint64_t pool_gen;
  {
    std::lock_guard<std::mutex> lk(pools->mu);
    const void *p0 = pools->plane_pool.p, *c0 = pools->coef_pool.p, *s0 = pools->sp_pool.p;
    HIPCHECK(pools->plane_pool.ensure((size_t)std::min(plane_sets, nb) * planes_per_set * max_npx * 4));
    HIPCHECK(pools->coef_pool.ensure(coef_need));
    HIPCHECK(pools->sp_pool.ensure(sp_total * 4));
    if (pools->plane_pool.p != p0 || pools->coef_pool.p != c0 || pools->sp_pool.p != s0) pools->generation++;
    if (pools->coef_pool.p != c0) pools->coef_pool_clean = false;
    pool_gen = pools->generation;
  }
  DevMem &plane_pool = pools->plane_pool, &coef_pool = pools->coef_pool;
  // tables and compressed bytes of all frames: one page-locked staging buffer and ONE upload each (a frame used to cost three
  // stream operations here and one more in collect(); next to seven other busy contexts those ~500 tiny operations per flight
  // took up to 200 ms of the stream's time)
  std::vector<GatherDesc> gdesc;
  uint32_t gather_max = 0;
  {
    const auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t tab_total = 0, cs_total = 0;
    for (int i : batched) { const FramePlan &P = slot((size_t)i).plan; tab_total += al(P.tables.size()); cs_total += al(P.cs_size + 64); }
    HIPCHECK(flight_tables.ensure(tab_total)); HIPCHECK(h_flight_tables.ensure(tab_total)); HIPCHECK(flight_cs.ensure(cs_total));
    size_t to = 0, co = 0;
    bool host_cs = false;
    for (int i : batched) host_cs = host_cs || !slot((size_t)i).up_cs_dev;
    if (host_cs) HIPCHECK(h_flight_cs.ensure(cs_total));
    for (int i : batched) {
      FrameSlot &S = slot((size_t)i);
      const FramePlan &P = S.plan;
      memcpy((uint8_t *)h_flight_tables.p + to, P.tables.data(), P.tables.size());
      S.B.tables = (const uint8_t *)flight_tables.p + to;
      if (S.post_active && S.post_fused) S.B.post = (const DevPost *)(S.B.tables + S.post_off);
      S.B.codestream = (const uint8_t *)flight_cs.p + co;
      if (S.up_cs_dev) { gdesc.push_back({S.up_cs_dev, (uint8_t *)flight_cs.p + co, (uint32_t)P.cs_size, 64u}); gather_max = std::max(gather_max, (uint32_t)P.cs_size); }
      else {
        memcpy((uint8_t *)h_flight_cs.p + co, P.cs, P.cs_size); memset((uint8_t *)h_flight_cs.p + co + P.cs_size, 0, 64);
        HIPCHECK(hipMemcpyAsync((uint8_t *)flight_cs.p + co, (uint8_t *)h_flight_cs.p + co, P.cs_size + 64, hipMemcpyHostToDevice, stream));
      }
      to += al(P.tables.size()); co += al(P.cs_size + 64);
    }
    HIPCHECK(hipMemcpyAsync(flight_tables.p, h_flight_tables.p, tab_total, hipMemcpyHostToDevice, stream));
  }
  std::vector<DevBuffers> hb; std::vector<DevAux> ha; std::vector<int> lf_map, pg_map, ec_map, w_map, sf_groups;
  std::vector<size_t> pg_off, ec_off, w_off;                 // per sub-flight: first entry of its PassGroup map / extra-channel group map / wavefront map
  bool all_flat = true;
  for (int i : batched) all_flat = all_flat && frame_flat_ok(slot((size_t)i).plan);
  std::vector<int> sf_chain;
  const auto close_subflight = [&] { const std::vector<int> m = flat_wave_map(sf_groups, sf_chain); w_map.insert(w_map.end(), m.begin(), m.end()); sf_groups.clear(); sf_chain.clear(); };
  std::vector<int> ec_ops;                                   // per sub-flight: most inverse transforms any of its frames has
  bool any_ec = false;
  int ec_pool = kModPoolMin;                                 // LDS table pool of the extra channels' group streams: the largest any frame of the flight asks for
  for (int k = 0; k < nb; k++) {
    FrameSlot &S = slot((size_t)batched[(size_t)k]);
    float *set = (float *)plane_pool.p + (size_t)(k % plane_sets) * planes_per_set * max_npx;
    int32_t *cset = (int32_t *)coef_pool.p + (size_t)(k % used_sets) * 3 * max_coef;
    for (int c = 0; c < 3; c++) { S.B.plane_a[c] = set + (size_t)c * max_npx; S.B.plane_b[c] = planes_per_set == 6 ? set + (size_t)(3 + c) * max_npx : nullptr; S.B.coef[c] = sparse ? nullptr : cset + (size_t)c * max_coef; }
    S.B.coef_sp = sparse ? (uint32_t *)pools->sp_pool.p + sp_frame_off[(size_t)k] : nullptr;      // (sp_group: below, once the flight's table block is laid out)
    hb.push_back(S.B); ha.push_back(S.A);
    if (k % hf_sets == 0) { if (k) close_subflight(); pg_off.push_back(pg_map.size() / 2); ec_off.push_back(ec_map.size() / 2); ec_ops.push_back(0); w_off.push_back(w_map.size() / 3); }
    sf_groups.push_back(S.plan.num_groups); sf_chain.push_back(flat_tail_chain(S.plan) ? 1 : 0);
    if (sf_chain.back() && (sparse || (all_flat && flight_groups >= flat_min_groups))) chained_tail_frames++;
    for (int g = 0; g < S.plan.num_groups; g++) { pg_map.push_back(k - k / hf_sets * hf_sets); pg_map.push_back(g); }   // frame index inside its sub-flight
    if (S.plan.has_ec) {
      const DevFrame *F = (const DevFrame *)S.plan.tables.data();
      any_ec = true;
      ec_pool = std::max(ec_pool, mod_group_pool_bytes(S.plan));
      ec_ops.back() = std::max(ec_ops.back(), (int)F->mod_nops);
      if (F->mod_first_group_ch < F->mod_nch) for (int g = 0; g < S.plan.num_groups; g++) { ec_map.push_back(k - k / hf_sets * hf_sets); ec_map.push_back(g); }
    }
  }
  close_subflight();
  pg_off.push_back(pg_map.size() / 2); ec_off.push_back(ec_map.size() / 2); w_off.push_back(w_map.size() / 3);
  // LF map: group-major — the long streams (full 256x256-cell LF groups, 240 ms) are dispatched first and the short edge
  // groups (15 ms) fill the slots they leave, instead of long and short workgroups alternating
  int max_lfg = 0;
  for (int i : batched) max_lfg = std::max(max_lfg, slot((size_t)i).plan.num_lf_groups);
  for (int g = 0; g < max_lfg; g++)
    for (int k = 0; k < nb; k++) if (g < slot((size_t)batched[(size_t)k]).plan.num_lf_groups) { lf_map.push_back(k); lf_map.push_back(g); }
  const size_t o_b = 0, o_a = (hb.size() * sizeof(DevBuffers) + 255) & ~(size_t)255, o_lf = (o_a + ha.size() * sizeof(DevAux) + 255) & ~(size_t)255,
               o_pg = (o_lf + lf_map.size() * 4 + 255) & ~(size_t)255, o_ec = (o_pg + pg_map.size() * 4 + 255) & ~(size_t)255,
               o_w = (o_ec + ec_map.size() * 4 + 255) & ~(size_t)255, o_gd = (o_w + w_map.size() * 4 + 255) & ~(size_t)255,
               o_sp = (o_gd + gdesc.size() * sizeof(GatherDesc) + 255) & ~(size_t)255,
               o_fl = (o_sp + sp_groups.size() * 4 + 255) & ~(size_t)255, total = o_fl + (size_t)nb * kFlagWords * 4 + 4;
  HIPCHECK(batch_tab.ensure(total));
  HIPCHECK(h_batch.ensure(total));
  uint8_t *bt = (uint8_t *)batch_tab.p, *hbt = (uint8_t *)h_batch.p;
  if (sparse) {
    for (int k = 0; k < nb; k++) { hb[(size_t)k].sp_group = (const uint32_t *)(bt + o_sp) + sp_tab_off[(size_t)k]; slot((size_t)batched[(size_t)k]).B.sp_group = hb[(size_t)k].sp_group; }
    memcpy(hbt + o_sp, sp_groups.data(), sp_groups.size() * 4);
  }
  memcpy(hbt + o_b, hb.data(), hb.size() * sizeof(DevBuffers));
  memcpy(hbt + o_a, ha.data(), ha.size() * sizeof(DevAux));
  memcpy(hbt + o_lf, lf_map.data(), lf_map.size() * 4);
  memcpy(hbt + o_pg, pg_map.data(), pg_map.size() * 4);
  if (!ec_map.empty()) memcpy(hbt + o_ec, ec_map.data(), ec_map.size() * 4);
  memcpy(hbt + o_w, w_map.data(), w_map.size() * 4);
  if (!gdesc.empty()) memcpy(hbt + o_gd, gdesc.data(), gdesc.size() * sizeof(GatherDesc));
  HIPCHECK(hipMemcpyAsync(bt, hbt, o_fl, hipMemcpyHostToDevice, stream));
  if (!gdesc.empty()) launch_gather_streams((const GatherDesc *)(bt + o_gd), (int)gdesc.size(), gather_max, stream);
  const DevBuffers *dB = (const DevBuffers *)(bt + o_b);
  const DevAux *dA = (const DevAux *)(bt + o_a);
  if (huge_blocks_seen) HIPCHECK(huge_scratch.ensure((size_t)kHugeSlots * 2 * 65536 * 4));
  HIPCHECK(hipEventRecord(ev[0], stream));
  const int ablate = g_ablate.load();
  if (!(ablate & 1)) {
  launch_clear_batch(dB, nb, max_cells, stream);
  if (any_ec) launch_ec_global_batch(dB, nb, ec_pool, stream);          // GlobalModular parts of the extra channels (skips frames without)
  static const int lf_pool_min_env = getenv("JXLAMD_LF_POOL_MIN") ? atoi(getenv("JXLAMD_LF_POOL_MIN")) : 0;      // measurement switch: the LDS of an LF workgroup as a variable (streams per CU)
  launch_lf_groups_batch(dB, dA, (const int *)(bt + o_lf), (int)lf_map.size() / 2, std::min(kModPoolBytes, std::max(std::max(lf_pool_bytes, g_lf_pool_floor.load()), lf_pool_min_env)), lf_general, stream);
  }
  HIPCHECK(hipEventRecord(ev[1], stream));
  if (!(ablate & 1)) launch_lf_smooth_batch(dB, nb, max_cells, stream);
  // ---- HF phase: this context's turn on the pools (uncontended unless shared).  The wait is host-side and overlaps the LF stage launched above
  std::unique_lock<std::mutex> pool_lock(pools->mu);
  if (pools->generation != pool_gen) return kRetryMoved;      // a sharing context re-allocated the pools meanwhile (first flights only): the tables above hold stale addresses
  if (!sparse) {
    if (!pools->coef_pool_clean) HIPCHECK(hipMemsetAsync(coef_pool.p, 0, coef_pool.cap, stream));      // fresh, or left dirty by a failed flight: clear once
    pools->coef_pool_clean = false;                          // until every frame of this flight has been collected without error
  }
  for (int sf = 0, k0 = 0; k0 < nb; sf++, k0 += hf_sets) {
    const int cnt = std::min(hf_sets, nb - k0);
    const int *map = (const int *)(bt + o_pg) + 2 * pg_off[(size_t)sf];
    const int n_pg = (int)(pg_off[(size_t)sf + 1] - pg_off[(size_t)sf]);
    // >= flat_min_groups groups: one LANE per group (k_pass_prep + k_pass_flat, 64 streams per wavefront); below that the
    // one-wave-per-group kernel has the shorter critical path
    if (ablate & 2) {
    } else if (sparse || (all_flat && n_pg >= flat_min_groups)) {
      launch_pass_prep(dB + k0, map, n_pg, stream);
      launch_pass_flat(dB + k0, (const int *)(bt + o_w) + 3 * w_off[(size_t)sf], (int)(w_off[(size_t)sf + 1] - w_off[(size_t)sf]), sparse, stream);
    } else launch_pass_groups_batch(dB + k0, map, n_pg, stream);
    if (any_ec) launch_ec_groups_batch(dB + k0, (const int *)(bt + o_ec) + 2 * ec_off[(size_t)sf], cnt, (int)(ec_off[(size_t)sf + 1] - ec_off[(size_t)sf]),
                                       ec_ops[(size_t)sf], ec_pool, stream);
    if (sf == 0) HIPCHECK(hipEventRecord(ev[2], stream));
    for (int j0 = 0; j0 < cnt && !(ablate & 4); j0 += plane_sets) {
      launch_rest_batch(dB + k0 + j0, (const uint8_t *)stat.p, std::min(plane_sets, cnt - j0), max_cells, max_w, max_h, stage_mask, large_hint, 3, stream, sparse,
                        huge_blocks_seen ? (float *)huge_scratch.p : nullptr);
      if (stage_mask & 32) for (int j = j0; j < std::min(j0 + plane_sets, cnt); j++) {      // composed frames of the sub-batch: splines / noise / upsampling + their writer, before the plane set moves on
        FrameSlot &S = slot((size_t)batched[(size_t)(k0 + j)]);
        if (S.plan.compose) { const int rc = launch_compose_tail(S); if (rc) return rc; }
      }
    }
  }
  HIPCHECK(hipEventRecord(ev[4], stream));
  const double t_launched = now();
  int first_rc = JXLAMD_OK;
  large_blocks_seen = false;
  uint32_t pool_want = 0;
  bool need_pool = false, need_dense = false, need_huge = false;
  // flags / counters of all frames in one device-to-host copy and one synchronisation
  launch_gather_flags(dB, nb, (uint32_t *)(bt + o_fl), stream);
  HIPCHECK(h_flags.ensure((size_t)nb * kFlagWords * 4));
  HIPCHECK(hipMemcpyAsync(h_flags.p, bt + o_fl, (size_t)nb * kFlagWords * 4, hipMemcpyDeviceToHost, stream));
  for (int i : batched) if (slot((size_t)i).post_active && slot((size_t)i).post_runs && slot((size_t)i).post_gen != post_lut_gen) {
    // the context holds ONE set of tone-map LUTs: frames of a flight that need different ones (other primaries / transfer function / bit depth) cannot share it
    set_error("unsupported: frames of one batch with different colour-matrix plans under jxlamd_decoder_set_writer_post"); return JXLAMD_ERR_UNSUPPORTED;
  }
  for (int i : batched) launch_post_pass(slot((size_t)i));      // frames whose writer could not take A10 + A11 itself: one pass over their RGBA each
  for (int i : batched) { FrameSlot &S = slot((size_t)i); if (S.host_out) HIPCHECK(hipMemcpyAsync(S.host_out, S.post_active ? S.post_final : S.d_out, S.post_active ? S.post_bytes : S.out_bytes, hipMemcpyDeviceToHost, stream)); }
  HIPCHECK(hipStreamSynchronize(stream));
  HIPCHECK(hipGetLastError());
  for (int k = 0; k < nb; k++) {
    FrameSlot &S = slot((size_t)batched[(size_t)k]);
    const uint32_t *head = (const uint32_t *)h_flags.p + (size_t)k * kFlagWords;
    if (head[17] > 0) large_blocks_seen = true;            // big_count[1]: varblocks with 2048 / 4096 coefficients
    serial_streams += head[2]; block_tree_channels += head[3];
    if (head[19] > 0 && (!huge_blocks_seen || sparse)) need_huge = true;      // big_count[3]: DCT128 / DCT256 families, and their kernel was not in this flight's launch list
    pool_want = std::max(pool_want, head[1]);
    if (sparse && (head[0] & kErrNeedDense) && !(head[0] & 0xFFFFu & ~kErrNeedDense)) { need_dense = true; continue; }      // (judged again in the dense flight)
    // A stream that stopped for a larger pool leaves its frame's later stages to decode whatever the slot held before (a fresh slot: zeros, nothing flagged; a used one: another
    // frame's metadata, flagged as corrupt by the PassGroup kernels): with kErrNeedPool set and the pool not yet the largest, the other flags of this attempt say nothing —
    // the flight runs again and is judged then (kErrNeedPool cannot be raised with the largest pool, so this repeats at most once)
    if ((head[0] & kErrNeedPool) && lf_pool_bytes < kModPoolBytes) { need_pool = true; continue; }
    else if ((head[0] & kErrNeedGeneral) && !lf_general) return kRetryGeneral;      // (coef_pool_clean stays false: the second attempt clears the pool)
    if (head[0]) { set_error("corrupt or unsupported stream (device flags " + std::to_string(head[0]) + ")"); if (!first_rc) first_rc = dev_err_class(head[0]); }
    (void)S;        // S.coef_clean describes the slot's OWN coefficient planes (single decodes); a flight uses the decoder's pool and leaves it as it is
  }
  // one miss is enough evidence that this context's frames vary: it keeps the largest pool from here on (a repeated flight costs more than a
  // fourth LF stream per CU gains; measured on 256 distinct frames: wanted pools 12 .. 25 KB, 8 % of the flights repeated with a creeping floor)
  if (need_dense && !need_pool) return kRetryDense;
  if (need_huge && !need_pool) { if (sparse) return kRetryDense; return kRetryHuge; }
  // A miss: the flight runs again with the largest pool, and what its streams then report (every one of them, this time) becomes the floor of this
  // process's later launches — the content's own maximum (12 - 26 KB on the bench's distinct frames), not the largest pool for good: an LF wave keeps
  // its LDS for ~100 ms and three of them at 53 KB leave a CU's other 3 KB to nobody (round 5: their launches wait 150 - 230 ms for a slot).
  if (need_pool) { lf_pool_bytes = kModPoolBytes; pool_missed = true; return kRetryPool; }
  if (pool_missed) {
    pool_missed = false;
    const int want = lf_pool_clamp(pool_want + 2048);      // + headroom: the next flights' frames differ a little (256 distinct bench frames: 12 - 25 KB), every further miss repeats a flight
    lf_pool_floor = std::max(lf_pool_floor, want);
    for (int cur = g_lf_pool_floor.load(); cur < want && !g_lf_pool_floor.compare_exchange_weak(cur, want);) {}
  }
  if (!sparse) pools->coef_pool_clean = first_rc == JXLAMD_OK;
  lf_pool_bytes = std::max(std::max(lf_pool_floor, g_lf_pool_floor.load()), lf_pool_clamp(pool_want));
  { static const bool forget = getenv("JXLAMD_LF_POOL_FORGET") != nullptr;      // tests: every flight starts from the smallest pool again, so that flights on USED slots miss it and are repeated
    if (forget) { lf_pool_floor = 0; g_lf_pool_floor.store(0); lf_pool_bytes = kModPoolMin; } }
  large_hint = large_blocks_seen;                      // the next flight of this context most likely looks like this one
  (void)hipEventElapsedTime(&timing[0], ev[0], ev[1]); (void)hipEventElapsedTime(&timing[1], ev[1], ev[2]);   // LF; first sub-flight's PassGroup
  (void)hipEventElapsedTime(&timing[2], ev[2], ev[4]); timing[3] = 0; (void)hipEventElapsedTime(&timing[4], ev[0], ev[4]);
  if (trace) {
    float to_ev0 = 0; (void)hipEventElapsedTime(&to_ev0, ev[5], ev[0]);       // ev[5]: recorded before the first upload of the flight
    fprintf(stderr, "[flight %p] begin %.1f end %.1f | n=%d parse %.1f prepare %.1f launch %.1f wait+collect %.1f ms | GPU: uploads %.1f LF %.1f pass0 %.1f rest %.1f\n",
            (void *)this, t_begin, now(), nb, t_parsed - t_begin, t_prepared - t_parsed, t_launched - t_prepared, now() - t_launched, to_ev0, timing[0], timing[1], timing[2]);
    // the LF streams' own stamps (100 MHz device wall clock): when each started after the launch's first one (workgroups waiting for a
    // slot) and how long the launch really ran — against the stage time the events bracket
    std::vector<double> st, dur; uint64_t first = ~0ull, last = 0;
    for (int k = 0; k < nb; k++) {
      const FrameSlot &S = slot((size_t)batched[(size_t)k]);
      const int n = S.plan.num_lf_groups;
      std::vector<uint64_t> tt((size_t)n * 8);
      if (hipMemcpy(tt.data(), (uint8_t *)S.misc.p + 4096 + (size_t)n * 8, (size_t)n * 64, hipMemcpyDeviceToHost) != hipSuccess) break;
      for (int g = 0; g < n; g++) if (tt[(size_t)g * 8]) { first = std::min(first, tt[(size_t)g * 8]); last = std::max(last, tt[(size_t)g * 8 + 6]); st.push_back((double)tt[(size_t)g * 8]); dur.push_back((double)(tt[(size_t)g * 8 + 6] - tt[(size_t)g * 8]) / 1e5); }
    }
    if (!st.empty()) {
      for (double &v : st) v = (v - (double)first) / 1e5;
      std::sort(st.begin(), st.end()); std::sort(dur.begin(), dur.end());
      fprintf(stderr, "[flight %p]   LF streams %zu: start offsets ms p50 %.1f p75 %.1f p90 %.1f max %.1f | durations ms p50 %.1f p90 %.1f max %.1f | first start -> last end %.1f ms (stage %.1f)\n",
              (void *)this, st.size(), st[st.size() / 2], st[st.size() * 3 / 4], st[st.size() * 9 / 10], st.back(), dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur.back(),
              (double)(last - first) / 1e5, timing[0]);
    }
  }
  return first_rc;
}

extern "C" {

jxlamd_decoder *jxlamd_decoder_create(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    g_tls_error = "no HIP device " + std::to_string(device) + " (the MI355X path has no CPU fallback)";
    return nullptr;
  }
  jxlamd_decoder *d = new jxlamd_decoder();
  d->device = device;
  if (hipSetDevice(device) != hipSuccess) { g_tls_error = "cannot open HIP device"; delete d; return nullptr; }
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
    g_tls_error = "cannot open HIP device"; delete d; return nullptr;
  }
  for (auto &e : d->ev) (void)hipEventCreate(&e);
  return d;
}

void jxlamd_decoder_destroy(jxlamd_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  (void)hipStreamSynchronize(d->stream);
  for (FrameSlot *fs : d->slots) delete fs;
  for (FrameSlot *fs : d->ref_slots) delete fs;          // DevMem / PinnedMem members release themselves (slots and the decoder's own pools)
  for (auto &e : d->ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(d->stream);
  delete d;
}

int jxlamd_decoder_share_pools(jxlamd_decoder *owner, jxlamd_decoder *peer) {
  if (!owner || !peer || owner == peer || owner->device != peer->device) { g_tls_error = "share_pools: two decoders of one device"; return JXLAMD_ERR_BUFFER; }
  peer->pools = owner->pools;           // the peer's own pools (if any were allocated) are released with its last reference
  return JXLAMD_OK;
}

const char *jxlamd_last_error(const jxlamd_decoder *d) { return d ? d->error.c_str() : g_tls_error.c_str(); }

int jxlamd_basic_info(const uint8_t *jxl, size_t size, jxlamd_info *info) {
  return jxlamd_guarded(nullptr, [&]() -> int {
    ImageInfo ii; std::string err;
    if (parse_basic_info(jxl, size, &ii, &err)) { g_tls_error = err; return JXLAMD_ERR_INVALID; }
    fill_public_info(ii, JXLAMD_ALLOW_16BIT, info);
    return JXLAMD_OK;
  });
}

int jxlamd_get_icc(const uint8_t *jxl, size_t size, uint8_t *icc, size_t capacity, size_t *icc_size) {
  return jxlamd_guarded(nullptr, [&]() -> int {
    ImageInfo ii; std::string err; std::vector<uint8_t> bytes;
    if (parse_basic_info(jxl, size, &ii, &err, &bytes)) { g_tls_error = err; return JXLAMD_ERR_INVALID; }
    if (icc_size) *icc_size = bytes.size();
    if (bytes.size() > capacity || (!icc && !bytes.empty())) { g_tls_error = "output buffer too small"; return JXLAMD_ERR_BUFFER; }
    if (!bytes.empty()) memcpy(icc, bytes.data(), bytes.size());
    return JXLAMD_OK;
  });
}

int jxlamd_output_size(const uint8_t *jxl, size_t size, uint32_t flags, size_t *bytes) {
  return jxlamd_guarded(nullptr, [&]() -> int {
    ImageInfo ii; std::string err;
    if (parse_basic_info(jxl, size, &ii, &err)) { g_tls_error = err; return JXLAMD_ERR_INVALID; }
    jxlamd_info o; fill_public_info(ii, flags, &o);
    int rc = size_guard(o, flags, &err);
    if (rc) { g_tls_error = err; return rc; }
    *bytes = (size_t)o.xsize * o.ysize * 4 * (o.out_bits == 16 ? 2 : 1);
    return JXLAMD_OK;
  });
}

int jxlamd_decode(jxlamd_decoder *d, const uint8_t *jxl, size_t size, uint32_t flags, void *out, size_t cap, jxlamd_info *info) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  return jxlamd_guarded(d, [&]() -> int { return d->decode(jxl, size, nullptr, flags & ~JXLAMD_IN_DEVICE, out, cap, info); });
}

// A10 + A11 with the decode: from now on this context's decodes deliver the Bitmap format of jxlamd_reformat_query(w, h, 16-bit?, cfg,
// has alpha, api_level) — colour matrix / tone map when the reference's JNI layer would apply it (cpp/JniDecoding.cpp:131-137), premultiply, conversion —
// INSIDE the writer for frames whose last filter stage is a per-stage kernel (three EPF iterations: BASELINE config 5), one pass behind it otherwise.
int jxlamd_decoder_set_epf_reciprocal(jxlamd_decoder *d, int mode) {
  if (!d) return JXLAMD_ERR_INVALID;
  if (mode != 0 && mode != 1) { d->set_error("EPF reciprocal mode must be 0 (exact) or 1 (reference x86 build)"); return JXLAMD_ERR_INVALID; }
  d->epf_rcp_mode = mode;
  return JXLAMD_OK;
}
int jxlamd_decoder_set_writer_post(jxlamd_decoder *d, int enabled, int cfg, int api_level) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  if (enabled && (cfg < 1 || cfg > 6)) { d->set_error("Invalid Color Config: " + std::to_string(cfg) + " was passed"); return JXLAMD_ERR_BUFFER; }
  d->wpost_enabled = enabled != 0; d->wpost_cfg = cfg; d->wpost_api = api_level;
  return JXLAMD_OK;
}

// Coalesced frame `frame` of an animation (what the reference's JxlAnimatedDecoder::getFrame returns, interop/JxlAnimatedDecoder.cpp:28-144): the frames it
// is blended over are decoded into their reference slots first, all on the device.
int jxlamd_decode_frame(jxlamd_decoder *d, const uint8_t *jxl, size_t size, int frame, uint32_t flags, void *out, size_t cap, jxlamd_info *info) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  if (frame < 0) { d->set_error("Frame position must be positive"); return JXLAMD_ERR_INVALID; }
  return jxlamd_guarded(d, [&]() -> int { return d->decode(jxl, size, nullptr, flags & ~JXLAMD_IN_DEVICE, out, cap, info, frame); });
}
// The frame list of an animation as the reference's JxlAnimatedDecoder constructor collects it (interop/JxlAnimatedDecoder.hpp:68-185): *num_frames
// entries (the first min(cap, *num_frames) durations, in milliseconds, are stored) and the loop count (-1: the file is not an animation).  Host only.
int jxlamd_anim_info(const uint8_t *jxl, size_t size, int32_t *durations_ms, int cap, int32_t *num_frames, int32_t *loops) {
  return jxlamd_guarded(nullptr, [&]() -> int {
    std::vector<AnimFrame> d; AnimHeader h; std::string err;
    if (parse_anim_info(jxl, size, &d, &h, &err)) { g_tls_error = err; return err_class(err); }
    if (num_frames) *num_frames = (int32_t)d.size();
    if (loops) *loops = h.have_animation ? (int32_t)h.num_loops : -1;
    for (int i = 0; i < cap && i < (int)d.size() && durations_ms; i++) durations_ms[i] = d[(size_t)i].ms;
    return JXLAMD_OK;
  });
}
// The same walk with everything libjxl's frame events carry (JxlFrameHeader::duration in ticks, is_last; JxlAnimationHeader): what the libjxl-named
// compat library needs to answer JxlDecoderGetFrameHeader / JxlDecoderGetBasicInfo of an animation (csrc/libjxl_abi.cpp).
int jxlamd_anim_frames(const uint8_t *jxl, size_t size, jxlamd_anim_frame *frames, int cap, int32_t *num_frames, jxlamd_anim_header *header) {
  return jxlamd_guarded(nullptr, [&]() -> int {
    std::vector<AnimFrame> d; AnimHeader h; std::string err;
    if (parse_anim_info(jxl, size, &d, &h, &err)) { g_tls_error = err; return err_class(err); }
    if (num_frames) *num_frames = (int32_t)d.size();
    if (header) { header->have_animation = h.have_animation; header->tps_numerator = h.tps_numerator; header->tps_denominator = h.tps_denominator; header->num_loops = h.num_loops; header->have_timecodes = h.have_timecodes; }
    for (int i = 0; i < cap && i < (int)d.size() && frames; i++) { frames[i].duration_ticks = d[(size_t)i].ticks; frames[i].duration_ms = d[(size_t)i].ms; frames[i].is_last = d[(size_t)i].is_last; frames[i].coalesced_index = d[(size_t)i].coalesced; }
    return JXLAMD_OK;
  });
}

int jxlamd_decode_resident(jxlamd_decoder *d, const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out,
                           size_t cap, jxlamd_info *info) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  return jxlamd_guarded(d, [&]() -> int { return d->decode(jxl, size, jxl_dev, flags | JXLAMD_IN_DEVICE, out, cap, info); });
}

int jxlamd_decode_batch(jxlamd_decoder *d, int n, const uint8_t *const *jxl, const size_t *sizes, uint32_t flags, void *const *outs,
                        const size_t *caps, jxlamd_info *infos) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  return jxlamd_guarded(d, [&]() -> int { return d->decode_batch(n, jxl, sizes, nullptr, flags & ~JXLAMD_IN_DEVICE, outs, caps, infos); });
}

int jxlamd_decode_batch_resident(jxlamd_decoder *d, int n, const uint8_t *const *jxl, const size_t *sizes, const void *const *jxl_dev,
                                 uint32_t flags, void *const *outs, const size_t *caps, jxlamd_info *infos) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  return jxlamd_guarded(d, [&]() -> int { return d->decode_batch(n, jxl, sizes, jxl_dev, flags, outs, caps, infos); });
}

// ---- post-decode stages (A10, A11)
static uint32_t aligned64(uint32_t line) { return line + (64u - line % 64u) % 64u; }

int jxlamd_reformat_query(uint32_t w, uint32_t h, int src_is_u16, int cfg, int has_alpha, int api_level, jxlamd_reformat_info *o) {
  if (!o || cfg < JXLAMD_CFG_DEFAULT || cfg > JXLAMD_CFG_HARDWARE) { g_tls_error = "Invalid Color Config"; return JXLAMD_ERR_BUFFER; }
  if (cfg == JXLAMD_CFG_DEFAULT) {                       // ReformatBitmap.cpp:52-63 (depth > 8 <=> the decode produced u16)
    if (src_is_u16 && api_level >= 26) cfg = (api_level >= 33 && !has_alpha) ? JXLAMD_CFG_RGBA_1010102 : JXLAMD_CFG_RGBA_F16;
    else cfg = JXLAMD_CFG_RGBA_8888;
  }
  o->resolved_config = (uint32_t)cfg;
  switch (cfg) {
    case JXLAMD_CFG_RGBA_8888: o->stride = w * 4; o->format = JXLAMD_FMT_RGBA_8888; o->use_floats = 0; break;
    case JXLAMD_CFG_RGBA_F16: o->stride = src_is_u16 ? w * 8 : aligned64(w * 8); o->format = JXLAMD_FMT_RGBA_F16; o->use_floats = 1; break;
    case JXLAMD_CFG_RGB_565: o->stride = aligned64(w * 2); o->format = JXLAMD_FMT_RGB_565; o->use_floats = 0; break;
    case JXLAMD_CFG_RGBA_1010102: o->stride = aligned64(w * 4); o->format = JXLAMD_FMT_RGBA_1010102; o->use_floats = 0; break;
    default: o->stride = src_is_u16 ? w * 8 : w * 4; o->format = src_is_u16 ? JXLAMD_FMT_RGBA_F16 : JXLAMD_FMT_RGBA_8888; o->use_floats = src_is_u16 ? 1 : 0; break;
  }
  o->bytes = (uint64_t)o->stride * h;
  return JXLAMD_OK;
}

static int jxlamd_reformat_impl(jxlamd_decoder *d, void *src, uint32_t w, uint32_t h, int src_is_u16, uint32_t depth, int cfg, int alpha_premultiplied,
                    int has_alpha, int api_level, void *dst, size_t dst_cap, jxlamd_reformat_info *out) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  jxlamd_reformat_info info;
  int rc = jxlamd_reformat_query(w, h, src_is_u16, cfg, has_alpha, api_level, &info);
  if (rc) { d->error = g_tls_error; return rc; }
  if (out) *out = info;
  if (!src || !dst || dst_cap < info.bytes) { d->set_error("output buffer too small"); return JXLAMD_ERR_BUFFER; }
  if (src_is_u16 ? (depth < 10 || depth > 16) : depth != 8) { d->set_error("bit depth does not match the source format"); return JXLAMD_ERR_BUFFER; }
  if (hipSetDevice(d->device) != hipSuccess) { d->set_error("cannot select device"); return JXLAMD_ERR_DEVICE; }
  const hipStream_t s = d->stream;
  const uint32_t ss = w * (src_is_u16 ? 8u : 4u);
  if (!alpha_premultiplied && has_alpha) launch_post_premultiply(src, ss, w, h, src_is_u16 != 0, depth, s);
  const bool att = !alpha_premultiplied;
  const uint32_t line = info.format == JXLAMD_FMT_RGB_565 ? w * 2 : info.format == JXLAMD_FMT_RGBA_F16 ? w * 8 : w * 4;
  if (info.stride != line && hipMemsetAsync(dst, 0, info.bytes, s) != hipSuccess) { d->set_error("HIP: memset failed"); return JXLAMD_ERR_DEVICE; }
  PostKind k;
  switch (info.resolved_config) {
    case JXLAMD_CFG_RGBA_8888: k = src_is_u16 ? kPostRgba16To8 : kPostCopy8; break;
    case JXLAMD_CFG_RGBA_F16: k = src_is_u16 ? kPostU16ToF16 : kPostRgba8ToF16; break;
    case JXLAMD_CFG_RGB_565: k = src_is_u16 ? kPostRgba16To565 : kPostRgba8To565; break;
    case JXLAMD_CFG_RGBA_1010102: k = src_is_u16 ? kPostRgba16To1010102 : kPostRgba8To1010102; break;
    default: k = src_is_u16 ? kPostU16ToF16 : kPostCopy8; break;     // HARDWARE: ReformatBitmap.cpp:231-245
  }
  launch_post_convert(k, src, ss, dst, info.stride, w, h, depth, att, s);
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { d->set_error("HIP: post stage failed"); return JXLAMD_ERR_DEVICE; }
  return JXLAMD_OK;
}

int jxlamd_reformat(jxlamd_decoder *d, void *src, uint32_t w, uint32_t h, int src_is_u16, uint32_t depth, int cfg, int alpha_premultiplied,
                    int has_alpha, int api_level, void *dst, size_t dst_cap, jxlamd_reformat_info *out) {
  return jxlamd_guarded(d, [&]() -> int { return jxlamd_reformat_impl(d, src, w, h, src_is_u16, depth, cfg, alpha_premultiplied, has_alpha, api_level, dst, dst_cap, out); });
}

// The colour-matrix parameters of (format, depth, primaries, transfer, intensity target): matrix + the two LUTs on the device, cached per
// decoder context (the LUTs — 2 x 65 536 powf for u16 — only depend on these).  *runs = 0: the reference's stage is the identity here.
static int ensure_color_plan(jxlamd_decoder *d, int is_u16, uint32_t depth, uint32_t primaries, uint32_t tf, const double *xy8, float intensity_target, bool *runs) {
  static const double zeros[8] = {0.64, 0.33, 0.30, 0.60, 0.15, 0.06, 0.3127, 0.3290};
  double key[13] = {(double)(is_u16 != 0), (double)depth, (double)primaries, (double)tf, (double)intensity_target};
  for (int i = 0; i < 8; i++) key[5 + i] = (primaries == 1 || primaries == 9 || primaries == 11 || !xy8) ? 0.0 : xy8[i];
  const hipStream_t s = d->stream;
  if (d->post_key_valid && memcmp(key, d->post_key, sizeof(key)) == 0) { *runs = d->post_plan_runs; return JXLAMD_OK; }
  d->post_key_valid = false;
  ColorMatrixPlan P;
  if (!plan_color_matrix(is_u16 != 0, depth, primaries, tf, xy8 ? xy8 : zeros, intensity_target, &P)) {
    memcpy(d->post_key, key, sizeof(key)); d->post_key_valid = true; d->post_plan_runs = false;
    *runs = false;
    return JXLAMD_OK;
  }
  if (d->post_lin_lut.ensure(P.lin_lut.size() * 4) != hipSuccess || d->post_gam_lut.ensure(P.gam_lut.size() * 2) != hipSuccess ||
      hipMemcpyAsync(d->post_lin_lut.p, P.lin_lut.data(), P.lin_lut.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d->post_gam_lut.p, P.gam_lut.data(), P.gam_lut.size() * 2, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {          // the LUT vectors are pageable and die with this call
    d->set_error("HIP: LUT upload failed"); return JXLAMD_ERR_DEVICE;
  }
  ColorMatrixDev D;
  memcpy(D.m, P.m, sizeof(D.m));
  D.tone_map = P.tone_map; D.weight_a = P.weight_a; D.weight_b = P.weight_b;
  D.lin_lut = (const float *)d->post_lin_lut.p; D.gam_lut = (const uint16_t *)d->post_gam_lut.p;
  D.index_scale = P.index_scale; D.index_max = P.index_max;
  d->post_dev = D; memcpy(d->post_key, key, sizeof(key)); d->post_key_valid = true; d->post_plan_runs = true; d->post_lut_gen++;
  *runs = true;
  return JXLAMD_OK;
}

static int jxlamd_color_matrix_impl(jxlamd_decoder *d, void *px, uint32_t w, uint32_t h, int is_u16, uint32_t depth, uint32_t primaries, uint32_t tf,
                        const double *xy8, float intensity_target) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  if (!px || (is_u16 ? (depth < 9 || depth > 16) : depth != 8)) { d->set_error("bad pixel buffer / bit depth"); return JXLAMD_ERR_BUFFER; }
  if (hipSetDevice(d->device) != hipSuccess) { d->set_error("cannot select device"); return JXLAMD_ERR_DEVICE; }
  bool runs = false;
  int rc = ensure_color_plan(d, is_u16, depth, primaries, tf, xy8, intensity_target, &runs);
  if (rc || !runs) return rc;
  launch_post_color_matrix(px, w * (is_u16 ? 8u : 4u), w, h, is_u16 != 0, d->post_dev, d->stream);
  if (hipStreamSynchronize(d->stream) != hipSuccess || hipGetLastError() != hipSuccess) { d->set_error("HIP: colour matrix stage failed"); return JXLAMD_ERR_DEVICE; }
  return JXLAMD_OK;
}

int jxlamd_color_matrix(jxlamd_decoder *d, void *px, uint32_t w, uint32_t h, int is_u16, uint32_t depth, uint32_t primaries, uint32_t tf,
                        const double *xy8, float intensity_target) {
  return jxlamd_guarded(d, [&]() -> int { return jxlamd_color_matrix_impl(d, px, w, h, is_u16, depth, primaries, tf, xy8, intensity_target); });
}

static PostKind reformat_kind(uint32_t resolved_config, int src_is_u16) {
  switch (resolved_config) {
    case JXLAMD_CFG_RGBA_8888: return src_is_u16 ? kPostRgba16To8 : kPostCopy8;
    case JXLAMD_CFG_RGBA_F16: return src_is_u16 ? kPostU16ToF16 : kPostRgba8ToF16;
    case JXLAMD_CFG_RGB_565: return src_is_u16 ? kPostRgba16To565 : kPostRgba8To565;
    case JXLAMD_CFG_RGBA_1010102: return src_is_u16 ? kPostRgba16To1010102 : kPostRgba8To1010102;
    default: return src_is_u16 ? kPostU16ToF16 : kPostCopy8;     // HARDWARE: ReformatBitmap.cpp:231-245
  }
}

static int jxlamd_post_fused_impl(jxlamd_decoder *d, const void *src, uint32_t w, uint32_t h, int src_is_u16, uint32_t depth, int apply_color_matrix, uint32_t primaries,
                      uint32_t tf, const double *xy8, float intensity_target, int cfg, int alpha_premultiplied, int has_alpha, int api_level, void *dst,
                      size_t dst_cap, jxlamd_reformat_info *out) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  jxlamd_reformat_info info;
  int rc = jxlamd_reformat_query(w, h, src_is_u16, cfg, has_alpha, api_level, &info);
  if (rc) { d->error = g_tls_error; return rc; }
  if (out) *out = info;
  if (!src || !dst || dst_cap < info.bytes) { d->set_error("output buffer too small"); return JXLAMD_ERR_BUFFER; }
  if (src_is_u16 ? (depth < 10 || depth > 16) : depth != 8) { d->set_error("bit depth does not match the source format"); return JXLAMD_ERR_BUFFER; }
  if (hipSetDevice(d->device) != hipSuccess) { d->set_error("cannot select device"); return JXLAMD_ERR_DEVICE; }
  bool runs = false;
  if (apply_color_matrix) { rc = ensure_color_plan(d, src_is_u16, depth, primaries, tf, xy8, intensity_target, &runs); if (rc) return rc; }
  const hipStream_t s = d->stream;
  const uint32_t line = info.format == JXLAMD_FMT_RGB_565 ? w * 2 : info.format == JXLAMD_FMT_RGBA_F16 ? w * 8 : w * 4;
  if (info.stride != line && hipMemsetAsync(dst, 0, info.bytes, s) != hipSuccess) { d->set_error("HIP: memset failed"); return JXLAMD_ERR_DEVICE; }
  launch_post_fused(reformat_kind(info.resolved_config, src_is_u16), src, w * (src_is_u16 ? 8u : 4u), dst, info.stride, w, h, runs ? &d->post_dev : nullptr,
                    !alpha_premultiplied && has_alpha, depth, !alpha_premultiplied, s);
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { d->set_error("HIP: post stage failed"); return JXLAMD_ERR_DEVICE; }
  return JXLAMD_OK;
}

int jxlamd_post_fused(jxlamd_decoder *d, const void *src, uint32_t w, uint32_t h, int src_is_u16, uint32_t depth, int apply_color_matrix, uint32_t primaries,
                      uint32_t tf, const double *xy8, float intensity_target, int cfg, int alpha_premultiplied, int has_alpha, int api_level, void *dst,
                      size_t dst_cap, jxlamd_reformat_info *out) {
  return jxlamd_guarded(d, [&]() -> int { return jxlamd_post_fused_impl(d, src, w, h, src_is_u16, depth, apply_color_matrix, primaries, tf, xy8, intensity_target, cfg, alpha_premultiplied, has_alpha, api_level, dst, dst_cap, out); });
}

static int jxlamd_icc_transform_impl(jxlamd_decoder *d, void *px, uint32_t w, uint32_t h, int is_u16, const uint8_t *icc, size_t icc_size) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  if (!px || !icc || !icc_size) { d->set_error("bad pixel buffer / profile"); return JXLAMD_ERR_BUFFER; }
  constexpr int kN = 256;               // every 8-bit level is a lattice point (100 MB of HBM per cached profile)
  if (hipSetDevice(d->device) != hipSuccess) { d->set_error("cannot select device"); return JXLAMD_ERR_DEVICE; }
  const hipStream_t s = d->stream;
  if (d->icc_lut_key.size() != icc_size || memcmp(d->icc_lut_key.data(), icc, icc_size) != 0 || !d->icc_lut.p || d->icc_lut_u16 != (is_u16 != 0)) {
    std::vector<uint16_t> lut; std::string err;
    if (!build_icc_lut(icc, icc_size, kN, &lut, &err, /*eight_bit=*/!is_u16)) {
      if (err.rfind("unsupported", 0) == 0) { d->set_error(err); return JXLAMD_ERR_UNSUPPORTED; }
      return JXLAMD_OK;              // colorspace.cpp:47-51, :70-74: "better proceed with invalid photo than crash"
    }
    if (d->icc_lut.ensure(lut.size() * 2) != hipSuccess || hipMemcpy(d->icc_lut.p, lut.data(), lut.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
      d->set_error("HIP: ICC lattice upload failed"); return JXLAMD_ERR_DEVICE;
    }
    d->icc_lut_key.assign(icc, icc + icc_size); d->icc_lut_u16 = is_u16 != 0;
  }
  launch_post_icc_lut(px, w * (is_u16 ? 8u : 4u), w, h, is_u16 != 0, (const uint16_t *)d->icc_lut.p, kN, s);
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { d->set_error("HIP: ICC stage failed"); return JXLAMD_ERR_DEVICE; }
  return JXLAMD_OK;
}

int jxlamd_icc_transform(jxlamd_decoder *d, void *px, uint32_t w, uint32_t h, int is_u16, const uint8_t *icc, size_t icc_size) {
  return jxlamd_guarded(d, [&]() -> int { return jxlamd_icc_transform_impl(d, px, w, h, is_u16, icc, icc_size); });
}

int jxlamd_debug_lf_phases(jxlamd_decoder *d, int num_lf_groups, uint64_t *out) {
  if (!d || d->slots.empty() || !d->slots[0]->misc.p) return JXLAMD_ERR_DEVICE;
  if (num_lf_groups < 0 || num_lf_groups > d->slots[0]->plan.num_lf_groups) return JXLAMD_ERR_BUFFER;
  return hipMemcpy(out, (uint8_t *)d->slots[0]->misc.p + 4096 + (size_t)num_lf_groups * 8, (size_t)num_lf_groups * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : JXLAMD_ERR_DEVICE;
}

int jxlamd_debug_lf_general(const jxlamd_decoder *dec) { return dec && dec->lf_general ? 1 : 0; }
int jxlamd_debug_pass_chain(const jxlamd_decoder *dec, uint32_t *out) { if (!dec || !out) return JXLAMD_ERR_BUFFER; *out = dec->chained_tail_frames; return JXLAMD_OK; }
int jxlamd_debug_set_ablate(int mask) { g_ablate.store(mask & 7); return JXLAMD_OK; }
int jxlamd_debug_lf_retries(const jxlamd_decoder *dec, uint32_t out[3]) {
  if (!dec || !out) return JXLAMD_ERR_BUFFER;
  out[0] = dec->pool_retries; out[1] = dec->general_retries; out[2] = (uint32_t)dec->lf_pool_bytes;
  return JXLAMD_OK;
}
int jxlamd_debug_sparse(const jxlamd_decoder *dec, uint32_t out[2]) {      // {did the last flight hand its coefficients over as sparse lists, flights decoded again densely}
  if (!dec || !out) return JXLAMD_ERR_BUFFER;
  out[0] = dec->last_flight_sparse ? 1u : 0u; out[1] = dec->sparse_misses;
  return JXLAMD_OK;
}
int jxlamd_debug_modular(const jxlamd_decoder *dec, uint64_t out[2]) {      // {Modular streams decoded by the serial walker, channels decoded with their MA tree in block form} since the context was created
  if (!dec || !out) return JXLAMD_ERR_BUFFER;
  out[0] = dec->serial_streams; out[1] = dec->block_tree_channels;
  return JXLAMD_OK;
}
int jxlamd_debug_lf_phases_frame(jxlamd_decoder *d, int frame, int num_lf_groups, uint64_t *out) {     // frame = slot index inside the last flight
  if (!d || frame < 0 || (size_t)frame >= d->slots.size() || !d->slots[(size_t)frame]->misc.p) return JXLAMD_ERR_DEVICE;
  FrameSlot &S = *d->slots[(size_t)frame];
  if (num_lf_groups < 0 || num_lf_groups > S.plan.num_lf_groups) return JXLAMD_ERR_BUFFER;
  return hipMemcpy(out, (uint8_t *)S.misc.p + 4096 + (size_t)S.plan.num_lf_groups * 8, (size_t)num_lf_groups * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : JXLAMD_ERR_DEVICE;
}

int jxlamd_last_timing(const jxlamd_decoder *d, float ms[5]) {
  if (!d) return JXLAMD_ERR_DEVICE;
  memcpy(ms, d->timing, sizeof(d->timing));
  return JXLAMD_OK;
}

}  // extern "C"
