// tests/emul/emul.cpp — TEST-ONLY harness: runs the product's device code (jxl_coder_amd/csrc/dev_*.h, the
// exact functions the HIP kernels wrap) on the CPU, one "workgroup" at a time as a single serial thread
// (nthreads = 1, barriers are no-ops).  It lets the CPU test suite check the bitstream logic and the table
// packing of the product's host parser without a GPU.  It is never linked into libjxlamd.so and is not a
// fallback: the product fails loudly without a HIP device.
#include <algorithm>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../jxl_coder_amd/csrc/dev_bodies.h"
#include "../../jxl_coder_amd/csrc/dev_modframe.h"
#include "../../jxl_coder_amd/csrc/dev_compose.h"
#include "../../jxl_coder_amd/csrc/dev_pass_flat.h"
#include "../../jxl_coder_amd/csrc/host_parse.h"

using namespace jxlamd;
struct NoSync { void operator()() const {} };
static std::string g_err;
// the product's classification (decoder.hip: dev_err_class): any flag besides "bitstream" / "ANS final state" names something the device path does not cover
static std::string flag_message(uint32_t err, const char *where) {
  const bool unsupported = (err & 0xFFFFu & ~(uint32_t)(kErrBitstream | kErrAnsFinal)) != 0;
  return std::string(unsupported ? "unsupported: " : "") + "device flags " + std::to_string(err) + " (" + where + ")";
}

extern "C" const char *emul_last_error() { return g_err.c_str(); }



// the loop filters, one stage after the other (the per-stage kernels k_filter_b)
static void filter_stages(const DevBuffers &B, const DevFrame &F) {
  bool a = true;
  auto planes = [&](bool isa, float *p[3]) { for (int c = 0; c < 3; c++) p[c] = isa ? B.plane_a[c] : B.plane_b[c]; };
  float *src[3], *dst[3];
  if (F.gab) { planes(a, src); planes(!a, dst); for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) gab_pixel(F, src, dst, x, y); a = !a; }
  for (int pass = 0; pass < 3; pass++) {
    bool run = pass == 0 ? F.epf_iters >= 3 : pass == 1 ? F.epf_iters >= 1 : F.epf_iters >= 2;
    if (!run) continue;
    planes(a, src); planes(!a, dst);
    for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) epf_pixel(B, F, src, dst, pass, x, y);
    a = !a;
  }
}

struct RefStore { std::vector<float> p[8][4]; int w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };      // [slot][R, G, B, alpha (blended canvases only)]

// composition tail (jxlamd_decoder::launch_compose_tail): patches, copy into the reference slot, stand-alone writer
static void compose_tail(FramePlan &plan, const DevBuffers &B, const DevFrame &F, int out_bits, RefStore &refs, const uint8_t *stat) {
  if (F.subsampled) for (int c = 0; c < 3; c++) for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) chroma_upsample_pixel(B, F, c, x, y);      // k_chroma_upsample
  const DevPatch *P = (const DevPatch *)(B.tables + F.patch_off);
  for (int i = 0; i < F.num_patches; i++) for (int k = 0; k < P[i].w * P[i].h; k++) patch_blend_sample(B, F, P[i], k);
  if (F.num_spline_segs > 0) for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) spline_pixel(B, F, x, y);      // k_splines
  const auto noise = [&]() {                                                              // k_noise_gen, k_noise_add
    const NoiseGeom G = noise_geom(B, F);
    for (int g = 0; g < G.xtiles * G.ytiles; g++) for (int lane = 0; lane < 8; lane++) noise_gen_lane(B, F, g, lane);
    for (int y = 0; y < G.h; y++) for (int x = 0; x < G.w; x++) noise_add_pixel(B, F, x, y);
  };
  if (F.noise && F.upsampling == 1 && !getenv("JXLEMUL_NO_NOISE")) noise();
  if (F.blend) {                                           // jxlamd_decoder::launch_compose_tail: the frame over its canvas (k_blend_canvas)
    if (F.alpha_up > 1 && F.mod_out[3] >= 0) for (int Y = 0; Y < F.full_h; Y++) for (int X = 0; X < F.full_w; X++) upsample_alpha_pixel(B, F, stat, X, Y);
    if (F.upsampling > 1) {
      for (int Y = 0; Y < F.full_h; Y++) for (int X = 0; X < F.full_w; X++) upsample_pixel(B, F, stat, X, Y);
      if (F.noise && !getenv("JXLEMUL_NO_NOISE")) noise();
    }
    DevBuffers Bb = B;
    std::vector<float> keep[4];
    const bool has_alpha = (F.has_ec || F.is_modular) && F.mod_out[3] >= 0;
    if (plan.save_slot >= 0 && plan.save_canvas) for (int c = 0; c < 4; c++) { if (c == 3 && !has_alpha) continue; keep[c].assign((size_t)F.canvas_w * F.canvas_h, 0.f); Bb.canvas_save[c] = keep[c].data(); }
    for (int y = 0; y < F.canvas_h; y++) for (int x = 0; x < F.canvas_w; x++) blend_canvas_pixel(Bb, stat, out_bits, x, y);
    if (getenv("JXLEMUL_TRACE_PX") && plan.save_slot >= 0 && plan.save_canvas) { int px, py; sscanf(getenv("JXLEMUL_TRACE_PX"), "%d,%d", &px, &py); size_t i = (size_t)py * F.canvas_w + px; fprintf(stderr, "canvas px (%d,%d): %g %g %g a %g | planes %g %g %g\n", px, py, keep[0][i], keep[1][i], keep[2][i], keep[3].empty() ? -1.f : keep[3][i], (compose_final_is_a(F) ? B.plane_a[0] : B.plane_b[0])[(size_t)(py - F.crop_y0) * F.pw + (px - F.crop_x0)], (compose_final_is_a(F) ? B.plane_a[1] : B.plane_b[1])[(size_t)(py - F.crop_y0) * F.pw + (px - F.crop_x0)], (compose_final_is_a(F) ? B.plane_a[2] : B.plane_b[2])[(size_t)(py - F.crop_y0) * F.pw + (px - F.crop_x0)]); }
    if (getenv("JXLEMUL_TRACE_CANVAS") && plan.save_slot >= 0 && plan.save_canvas) for (int c = 0; c < 3; c++) { float mn = 1e9f, mx = -1e9f; for (float v : keep[c]) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; } fprintf(stderr, "canvas slot %d ch %d min %g max %g (blend mode %d src %d)\n", plan.save_slot, c, mn, mx, F.bl_mode_c, F.bl_src); }
    if (plan.save_slot >= 0 && plan.save_canvas) { for (int c = 0; c < 4; c++) refs.p[plan.save_slot][c].swap(keep[c]); refs.w[plan.save_slot] = F.canvas_w; refs.h[plan.save_slot] = F.canvas_h; }
    return;
  }
  if (plan.save_slot >= 0) {
    float *dst[3];
    for (int c = 0; c < 3; c++) { refs.p[plan.save_slot][c].assign((size_t)F.width * F.height, 0.f); dst[c] = refs.p[plan.save_slot][c].data(); }
    for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) save_ref_pixel(B, F, dst, x, y);
  }
  if (F.no_output) return;
  if (F.alpha_up > 1 && F.mod_out[3] >= 0) for (int Y = 0; Y < F.full_h; Y++) for (int X = 0; X < F.full_w; X++) upsample_alpha_pixel(B, F, stat, X, Y);
  if (F.upsampling > 1) {                                  // k_upsample, then the writer at full resolution
    for (int Y = 0; Y < F.full_h; Y++) for (int X = 0; X < F.full_w; X++) upsample_pixel(B, F, stat, X, Y);
    if (F.noise && !getenv("JXLEMUL_NO_NOISE")) noise();                                  // libjxl's stage order: Upsampling, Noise, colour transform
    for (int Y = 0; Y < F.full_h; Y++) for (int X = 0; X < F.full_w; X++) upsampled_write_pixel(B, stat, out_bits, X, Y);
    return;
  }
  float *src[3];
  for (int c = 0; c < 3; c++) src[c] = compose_final_is_a(F) ? B.plane_a[c] : B.plane_b[c];
  for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) {
    if (F.is_modular && !F.xyb_modular) plain_write_pixel(B, stat, out_bits, x, y);
    else if (F.not_xyb) { const size_t po = (size_t)y * (size_t)F.pw + (size_t)x; plain_write_value(B, stat, *(const DevStatic *)stat, src[0][po], src[1][po], src[2][po], out_bits, x, y); }
    else xyb_write_pixel(B, stat, *(const DevStatic *)stat, src, out_bits, x, y);
  }
}

static int run_frame(FramePlan &plan, int out_bits, uint8_t *out, RefStore &refs);

static int g_target_frame = -1;
static int g_epf_rcp = 0;
extern "C" void emul_set_epf_reciprocal(int mode) { g_epf_rcp = mode; }      // what jxlamd_decoder_set_epf_reciprocal patches into the frame parameters
extern "C" void emul_set_target_frame(int i) { g_target_frame = i; }
extern "C" int emul_decode(const uint8_t *jxl, size_t size, int allow16, uint8_t *out, size_t out_cap, uint32_t *w, uint32_t *h, uint32_t *bits) {
  FramePlan plan;
  if (plan_parse(jxl, size, &plan, g_target_frame)) { g_err = plan.error; return -1; }
  const int out_bits = (plan.info.bits_per_sample > 8 && allow16) ? 16 : 8;
  *w = plan.info.xsize; *h = plan.info.ysize; *bits = (uint32_t)out_bits;
  const size_t out_bytes = (size_t)plan.info.xsize * plan.info.ysize * 4 * (out_bits / 8);
  if (plan.cropped && out_bytes <= out_cap) {             // what jxlamd_decoder::prepare does: the cleared canvas (opaque when the image has no alpha)
    memset(out, 0, out_bytes);
    if (!(plan.info.num_extra_channels > 0 && plan.info.alpha_bits > 0))
      for (size_t i = 0; i < out_bytes / (4 * (size_t)(out_bits / 8)); i++) { if (out_bits == 16) ((uint16_t *)out)[i * 4 + 3] = 65535; else out[i * 4 + 3] = 255; }
  }
  if (out_cap < out_bytes) { g_err = "buffer"; return -5; }
  RefStore refs;
  for (auto &r : plan.refs) { int rc = run_frame(*r, out_bits, nullptr, refs); if (rc) return rc; }      // the frames the patch dictionary draws on, each into its slot
  return run_frame(plan, out_bits, out, refs);
}

static int run_frame(FramePlan &plan, int out_bits, uint8_t *out, RefStore &refs) {
  const size_t ncell = (size_t)plan.xb * plan.yb, ntile = (size_t)((plan.xb + 7) / 8) * ((plan.yb + 7) / 8), npx = ncell * 64;
  std::vector<uint8_t> cs(plan.cs, plan.cs + plan.cs_size); cs.resize(cs.size() + 64, 0);
  std::vector<uint8_t> c8[5]; for (auto &v : c8) v.assign(ncell, 0);
  std::vector<int8_t> tl[2]; for (auto &v : tl) v.assign(ntile, 0);
  std::vector<float> lf[6]; for (auto &v : lf) v.assign(ncell, 0.f);
  std::vector<uint32_t> coef_off(ncell, 0);
  std::vector<int32_t> coef[3]; for (auto &v : coef) v.assign((size_t)plan.num_groups * 65536, 0);
  std::vector<float> pl[6]; for (auto &v : pl) v.assign(npx, 0.f);
  std::vector<int32_t> scr((size_t)plan.num_lf_groups * kLfScratchInts, 0);
  std::vector<uint64_t> endbits((size_t)plan.num_lf_groups, 0);
  std::vector<uint32_t> bl0(ncell / 8 + 16), bl1(ncell / 32 + 16), bl2(ncell + 16), bl3(ncell / 256 + 16); uint32_t bcount[4] = {0, 0, 0, 0};
  uint32_t errw[32] = {0}; uint32_t &err = errw[0];       // the frame's flag block (word 1: LF table pool the streams asked for)
  std::vector<uint8_t> tables = plan.tables; tables.reserve(tables.size() + (8u << 20));
  ((DevFrame *)tables.data())->epf_rcp_x86 = g_epf_rcp;
  DevBuffers B; memset(&B, 0, sizeof(B));
  B.codestream = cs.data(); B.tables = tables.data(); B.stat = static_tables().data();
  memset(c8[0].data(), 0xFF, ncell);
  B.strategy = c8[0].data(); B.first = c8[1].data(); B.qfm1 = c8[2].data(); B.sharp = c8[3].data(); B.lf_idx = c8[4].data();
  B.xfromy = tl[0].data(); B.bfromy = tl[1].data();
  for (int c = 0; c < 3; c++) { B.lf[c] = lf[c].data(); B.lf_s[c] = lf[3 + c].data(); B.coef[c] = coef[c].data(); B.plane_a[c] = pl[c].data(); B.plane_b[c] = pl[3 + c].data(); }
  B.coef_off = coef_off.data(); B.lf_scratch = scr.data(); B.err = errw; B.out = out;
  const DevFrame &F0 = *(const DevFrame *)plan.tables.data();
  std::vector<float> upv[4];
  if (F0.upsampling > 1) for (int c = 0; c < 3; c++) { upv[c].assign((size_t)F0.full_w * F0.full_h, 0.f); B.up[c] = upv[c].data(); }
  if (F0.alpha_up > 1) { upv[3].assign((size_t)F0.full_w * F0.full_h, 0.f); B.up[3] = upv[3].data(); }
  for (int k = 0; k < 4; k++) { for (int c = 0; c < 3; c++) B.ref[k][c] = refs.p[k][c].empty() ? nullptr : refs.p[k][c].data(); B.ref_a[k] = refs.p[k][3].empty() ? nullptr : refs.p[k][3].data(); }
  std::vector<float> noisev[3];
  if (F0.noise) for (int c = 0; c < 3; c++) { noisev[c].assign(F0.upsampling > 1 ? (size_t)F0.full_w * F0.full_h : npx, 0.f); B.noise[c] = noisev[c].data(); }
  { const int lk = ((const DevFrame *)plan.tables.data())->lf_frame_slot; for (int c = 0; c < 3; c++) B.lf_frame[c] = (lk < 4 || lk > 7 || refs.p[lk][c].empty()) ? nullptr : refs.p[lk][c].data(); }      // slots 4..7: the LF frames of a progressive_dc file
  std::vector<LocalTreeScratch> loc((size_t)((plan.modular || plan.has_ec) ? std::max(plan.num_groups > 1 ? plan.num_groups : 1, plan.num_lf_groups) : plan.num_lf_groups)); B.local = loc.data();
  std::vector<int32_t> mpool(plan.mod_pool_ints + 64, 0), mscr((plan.modular || plan.has_ec) ? ((size_t)plan.num_groups + 1) * mod_group_scratch_ints(*(const DevFrame *)plan.tables.data()) + (size_t)plan.num_lf_groups * (size_t)((const DevFrame *)plan.tables.data())->mod_lf_nch * 65536 + 64 : 1, 0);
  std::vector<uint64_t> pend((size_t)plan.num_groups * (size_t)plan.num_passes + 1, 0); B.pass_end_bits = pend.data();
  B.mod_pool = mpool.data(); B.mod_scratch = mscr.data();
  const uint32_t lzl = ((const DevFrame *)plan.tables.data())->lz_win_len, lzg = ((const DevFrame *)plan.tables.data())->lz_win_group;
  std::vector<uint32_t> lzw(plan.modular && lzl ? (size_t)lzl + (size_t)plan.num_groups * lzg : 1, 0); B.lz_win = plan.modular && lzl ? lzw.data() : nullptr;
  B.big_list[0] = bl0.data(); B.big_list[1] = bl1.data(); B.big_list[2] = bl2.data(); B.big_list[3] = bl3.data(); B.big_count = bcount;
  DevAux A; A.lf_end_bits = endbits.data(); A.lf_times = nullptr;
  const std::vector<uint8_t> &stat = static_tables();
  if (plan.modular) {
    const DevFrame &F = *(const DevFrame *)tables.data();
    DevModScratch *MS = new DevModScratch(); std::vector<DevChanOut> chbuf(kModMaxCh); MS->ch = chbuf.data();
    mod_global_body(B, *MS, 0, 1, NoSync());
    for (int g = 0; F.mod_lf_nch > 0 && g < plan.num_lf_groups && !err; g++) {       // ModularLfGroup streams (k_mod_lfgroup)
      const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
      DevBits b; bits_init(b, B.codestream, secs[1 + g].off, F.cs_size);
      MS->st.b = b; MS->wide_wp = nullptr;
      uint32_t e = mod_lfgroup_body(B, *MS, g, 0, 1, NoSync()); if (e) err |= e;
    }
    if (F.mod_first_group_ch < F.mod_nch) for (int g = 0; g < plan.num_groups && !err; g++) mod_group_body(B, *MS, g, 0, 1, NoSync());
    delete MS;
    if (err) { g_err = flag_message(err, "Modular"); return -2; }
    for (int o = 0; o < F.mod_nops; o++) { size_t n = (size_t)(F.mod_op_kind[o] == 0 ? F.mod_op_y[o] : F.mod_op_c[o]); for (size_t i = 0; i < n; i++) mod_op_element(B, F, o, i); }
    if (!F.compose) {
      for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) mod_write_pixel(B, out_bits, x, y);
      return 0;
    }
    for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) mod_to_planes_pixel(B, F, x, y);
    if (F.xyb_modular) filter_stages(B, F);
    compose_tail(plan, B, F, out_bits, refs, stat.data());
    return 0;
  }
  DevModScratch *MS = new DevModScratch(); std::vector<DevChanOut> chbuf(kModMaxCh); MS->ch = chbuf.data();
  uint64_t mod_end = 0; B.mod_end_bit = &mod_end;
  if (plan.has_ec) mod_global_body(B, *MS, 0, 1, NoSync());       // GlobalModular part of the extra channels: before LfGroup 0
  for (int g = 0; g < plan.num_lf_groups; g++) lf_group_body(B, A, *MS, g, 0, 1, NoSync());
  delete MS;
  if (err) { g_err = flag_message(err, "LfGroup"); return -2; }
  if (plan.single_section) {
    if (plan_parse_hf_single(&plan, endbits[0])) { g_err = plan.error; return -1; }
    tables = plan.tables; B.tables = tables.data();
    ((DevFrame *)tables.data())->epf_rcp_x86 = g_epf_rcp;
  }
  for (int y = 0; y < plan.yb; y++) for (int x = 0; x < plan.xb; x++) lf_smooth_cell(B, x, y);
  DevPassScratch *PS = new DevPassScratch();
  std::vector<uint8_t> pnz((size_t)plan.num_groups * kPassBlkStride, 0); B.pass_nz = pnz.data();
  if (getenv("JXLEMUL_STATS")) {
    int hist[32] = {0};
    for (size_t o = 0; o < ncell; o++) if (B.first[o]) hist[B.strategy[o] & 31]++;
    for (int i = 0; i < 27; i++) if (hist[i]) fprintf(stderr, "strategy %d (%dx%d cells): %d blocks\n", i, kCoveredX[i], kCoveredY[i], hist[i]);
    fprintf(stderr, "lists: %u medium, %u large, %u small\n", bcount[0], bcount[1], bcount[2]);
  }
  // JXLEMUL_SPARSE (with JXLEMUL_FLAT_PASS): the flights' sparse coefficient lists — arenas sized like decoder.hip's sparse_group_entries, or by
  // JXLEMUL_SPARSE_CAP entries per group (a tiny value exercises the overflow flag)
  const bool sparse = getenv("JXLEMUL_FLAT_PASS") && getenv("JXLEMUL_SPARSE") && plan.num_passes == 1 && !((const DevFrame *)tables.data())->subsampled;
  std::vector<uint32_t> sp_ent, sp_grp, sp_cnt;
  if (sparse) {
    const DevFrame &F0 = *(const DevFrame *)tables.data();
    const DevSection *secs = (const DevSection *)(tables.data() + F0.sec_off);
    sp_grp.assign((size_t)plan.num_groups + 1, 0);
    for (int g = 0; g < plan.num_groups; g++) {
      const uint32_t bytes = F0.nsec == 1 ? secs[0].size : secs[2 + F0.num_lf_groups + g].size;
      sp_grp[(size_t)g + 1] = sp_grp[(size_t)g] + (getenv("JXLEMUL_SPARSE_CAP") ? (uint32_t)atoi(getenv("JXLEMUL_SPARSE_CAP")) : sparse_group_entries(bytes));
    }
    sp_ent.assign((size_t)sp_grp.back() + 1, 0); sp_cnt.assign(ncell, 0);
    B.coef_sp = sp_ent.data(); B.coef_cnt = sp_cnt.data(); B.sp_group = sp_grp.data();
  }
  if (getenv("JXLEMUL_FLAT_PASS")) {       // k_pass_prep + k_pass_flat: group descriptor lists, then the flat lane-per-group state machine, one lane at a time
    const DevFrame &F0 = *(const DevFrame *)tables.data();
    if (!flat_frame_ok(F0)) { g_err = "frame not eligible for the flat PassGroup path"; return -3; }
    for (int g = 0; g < plan.num_groups; g++) { uint32_t e = pass_prep_group_serial(B, g); if (e) err |= e; }
    FlatPassLds *L2 = new FlatPassLds();
    for (int pass = 0; pass < F0.num_passes && !err; pass++)
      for (int g = 0; g < plan.num_groups; g++) {
        flat_stage(B, *L2, pass, 0, 1);                 // per lane here: the nonzero-count columns start from zero for every group
        // JXLEMUL_FLAT_CHAIN: groups in pairs, the second one as the lane's chained group (what k_pass_flat does with a frame's tail groups)
        const int g2 = (getenv("JXLEMUL_FLAT_CHAIN") && g + 1 < plan.num_groups && F0.nsec != 1) ? g + 1 : -1;
        uint32_t e = sparse ? pass_group_flat<true>(B, *L2, pass, g, (g * 5 + 1) % 64, g2) : pass_group_flat<false>(B, *L2, pass, g, (g * 5 + 1) % 64, g2); if (e) err |= e;
        if (g2 >= 0) g++;
      }
    delete L2;
    if (sparse && getenv("JXLEMUL_STATS")) {
      const DevSection *secs = (const DevSection *)(tables.data() + F0.sec_off);
      double worst = 0; uint64_t tot_e = 0, tot_b = 0;
      for (int g = 0; g < plan.num_groups; g++) {
        uint32_t used = 0;
        const int gx = g % F0.xgroups, gy = g / F0.xgroups;
        for (int y = 0; y < 32 && gy * 32 + y < F0.yb; y++) for (int x = 0; x < 32 && gx * 32 + x < F0.xb; x++) { const size_t o = (size_t)(gy * 32 + y) * F0.xb + gx * 32 + x; if (B.first[o]) used += sp_cnt[o]; }
        const uint32_t bytes = F0.nsec == 1 ? secs[0].size : secs[2 + F0.num_lf_groups + g].size;
        tot_e += used; tot_b += bytes;
        if (bytes && (double)used / bytes > worst) worst = (double)used / bytes;
      }
      fprintf(stderr, "sparse: %llu entries for %llu section bytes (%.3f per byte), worst group %.3f per byte\n", (unsigned long long)tot_e, (unsigned long long)tot_b, tot_b ? (double)tot_e / tot_b : 0.0, worst);
    }
  } else
  for (int g = 0; g < plan.num_groups; g++) pass_group_body(B, *PS, g, 0, 1, NoSync());
  delete PS;
  if (err) { g_err = flag_message(err, "PassGroup"); return -2; }
  if (plan.has_ec) {                          // extra channels (alpha): same order as jxlamd_decoder::launch_extra_channels
    const DevFrame &F = *(const DevFrame *)tables.data();
    DevModScratch *MS2 = new DevModScratch(); std::vector<DevChanOut> chbuf2(kModMaxCh); MS2->ch = chbuf2.data();
    if (F.mod_first_group_ch < F.mod_nch) for (int g = 0; g < plan.num_groups && !err; g++) mod_group_body(B, *MS2, g, 0, 1, NoSync());
    delete MS2;
    if (err) { g_err = flag_message(err, "extra channels"); return -2; }
    for (int o = 0; o < F.mod_nops; o++) { size_t n = (size_t)(F.mod_op_kind[o] == 0 ? F.mod_op_y[o] : F.mod_op_c[o]); for (size_t i = 0; i < n; i++) mod_op_element(B, F, o, i); }
  }
  std::vector<float> S(3 * 65536), T(65536);       // (on the device the DCT128 / DCT256 families keep these tiles in HBM: k_recon_huge_b)
  for (int y = 0; y < plan.yb; y++) for (int x = 0; x < plan.xb; x++) {
    if (sparse) {
      recon_block_body<true, false, true>(B, stat.data(), S.data(), T.data(), x, y, 0, 1024, 0, 1, NoSync());
      recon_block_body<false, true, true>(B, stat.data(), S.data(), T.data(), x, y, 1025, 65536, 0, 1, NoSync());
    } else if (getenv("JXLEMUL_FLAT_PASS")) {       // also exercise the one-channel-at-a-time path the large-block kernel uses
      recon_block_body<true>(B, stat.data(), S.data(), T.data(), x, y, 0, 1024, 0, 1, NoSync());
      recon_block_body<false, true>(B, stat.data(), S.data(), T.data(), x, y, 1025, 65536, 0, 1, NoSync());
    } else recon_block_body<true>(B, stat.data(), S.data(), T.data(), x, y, 0, 65536, 0, 1, NoSync());
  }
  if (err) { g_err = flag_message(err, "recon"); return -2; }
  const DevFrame &F = *(const DevFrame *)tables.data();
  filter_stages(B, F);
  if (F.compose) { compose_tail(plan, B, F, out_bits, refs, stat.data()); return 0; }
  float *src[3];
  for (int c = 0; c < 3; c++) src[c] = compose_final_is_a(F) ? B.plane_a[c] : B.plane_b[c];
  for (int y = 0; y < F.height; y++) for (int x = 0; x < F.width; x++) xyb_write_pixel(B, stat.data(), *(const DevStatic *)stat.data(), src, out_bits, x, y);
  return 0;
}

// ---- the block form of large MA trees (dev_modular.h: big_tree_build) against the plain walk: random trees (seeded), random property vectors.
// Returns mismatches (0 = good), -1 when the tree could not be built in big_bytes, -2 when the count pass disagrees with the tree.  stats[0..2] = blocks, nodes, exits.
extern "C" int emul_bigtree_selftest(uint32_t seed, int internal_nodes, int big_bytes, int trials, int with_nonunit, int32_t *stats) {
  using namespace jxlamd;
  uint64_t rs = seed * 0x9E3779B97F4A7C15ull + 12345;
  auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); };
  // a random full binary tree in libjxl's order (a node's children come later): grow by turning random leaves into decision nodes
  std::vector<DevTreeNode> tree(1);
  std::vector<int> leaves{0};
  tree[0].prop = -1;
  for (int i = 0; i < internal_nodes; i++) {
    const size_t pick = (rnd() % 4 == 0) ? leaves.size() - 1 : rnd() % leaves.size();      // sometimes the newest leaf: deep chains
    const int at = leaves[pick];
    leaves.erase(leaves.begin() + (long)pick);
    const int l = (int)tree.size(), r = l + 1;
    tree.resize(tree.size() + 2);
    const uint32_t k = rnd() % 20;
    tree[(size_t)at].prop = k < 2 ? (int)k : 2 + (int)(rnd() % 14);       // properties 0 / 1 (static) now and then
    tree[(size_t)at].splitval = (int)(rnd() % 9) - 4;
    tree[(size_t)at].lchild = l; tree[(size_t)at].rchild = r; tree[(size_t)at].offset = 0;
    tree[(size_t)l].prop = tree[(size_t)r].prop = -1;
    leaves.push_back(l); leaves.push_back(r);
  }
  int nctx = 0;
  for (auto &nd : tree) if (nd.prop < 0) { nd.splitval = nctx++; nd.lchild = (int)(rnd() % 14); nd.rchild = 1; nd.offset = 0; if (with_nonunit && rnd() % 7 == 0) { nd.rchild = 1 + (int)(rnd() % 5); nd.offset = (int)(rnd() % 11) - 5; } }
  std::vector<uint8_t> ctx_map((size_t)nctx);
  for (auto &c : ctx_map) c = (uint8_t)(rnd() % 200);
  const int chan = (int)(rnd() % 5) - 2, stream = (int)(rnd() % 5) - 2;
  int32_t stack[64]; int32_t qn[64]; uint64_t q1[64], q0[64];
  const BigCount cnt = big_tree_count(tree.data(), (int)tree.size(), chan, stream, stack);
  if (!cnt.ok) return -3;           // deeper than the stack: the decoder falls back, nothing to compare
  // plain count of the pruned tree
  { int ni = 0, nl = 0; std::vector<int> st{0}; while (!st.empty()) { const DevTreeNode nd = tree[(size_t)st.back()]; st.pop_back();
      if (nd.prop < 0) { nl++; continue; } if (nd.prop < 2) { const int v = nd.prop == 0 ? chan : stream; st.push_back(v > nd.splitval ? nd.lchild : nd.rchild); continue; }
      ni++; st.push_back(nd.lchild); st.push_back(nd.rchild); }
    if (ni != cnt.ni || nl != cnt.nl) return -2; }
  std::vector<uint32_t> big((size_t)big_bytes / 4 + 16);
  if (!big_tree_build(tree.data(), (int)tree.size(), chan, stream, ctx_map.data(), cnt, qn, q1, q0, big.data(), big_bytes)) return -1;
  const DevBigHdr &H = *(const DevBigHdr *)big.data();
  if (stats) { stats[0] = H.nblocks; stats[1] = H.nnodes; stats[2] = H.nexits; }
  if (H.nnodes != cnt.ni || H.nexits != cnt.nl + H.nblocks - 1) return -2;
  int bad = 0;
  for (int t = 0; t < trials; t++) {
    int32_t props[16];
    props[0] = chan; props[1] = stream;
    for (int k = 2; k < 16; k++) props[k] = (int)(rnd() % 13) - 6;
    const DevTreeNode *nd = &tree[0];
    while (nd->prop >= 0) nd = &tree[(size_t)(props[nd->prop] > nd->splitval ? nd->lchild : nd->rchild)];
    int eidx = -1;
    const uint32_t e = big_tree_eval(big.data(), props, &eidx);
    const uint32_t want = ((nd->rchild != 1 || nd->offset != 0) ? 1u << 30 : 0u) | ((uint32_t)nd->lchild << 26) | ((uint32_t)ctx_map[(size_t)nd->splitval] << 18) | (uint32_t)nd->splitval;
    if (e != want) { bad++; continue; }
    if (e & (1u << 30)) { const int32_t *mulo = (const int32_t *)big.data() + H.off_mulo; if (mulo[2 * eidx] != nd->rchild || mulo[2 * eidx + 1] != nd->offset) bad++; }
  }
  return bad;
}
