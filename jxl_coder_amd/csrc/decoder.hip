// jxl_coder_amd/csrc/decoder.hip — decoder context: device buffers, H2D of the codestream + frame tables,
// kernel sequencing on one HIP stream, error collection; and the C-ABI of include/jxl_amd.h.
// Host counterpart of the reference's DecodeJpegXlOneShot driver loop (interop/JxlDecoding.cpp:36-176).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <limits>
#include <string>
#include <vector>
#include "../../include/jxl_amd.h"
#include "host_parse.h"
#include "kernels.h"

using namespace jxlamd;

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string("HIP: ") + hipGetErrorString(e_) + " at " #x); return JXLAMD_ERR_DEVICE; } } while (0)

static thread_local std::string g_tls_error;

struct DevMem {
  void *p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = n + n / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct jxlamd_decoder {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  std::string error;
  DevMem cs, tables, stat, cells8[5], tiles[2], lf[6], coef_off, coef[3], planes[6], lf_scratch, local, misc, out;
  bool stat_uploaded = false;
  float timing[5] = {0, 0, 0, 0, 0};
  void set_error(const std::string &e) { error = e; g_tls_error = e; }

  int decode(const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info);
};

static void fill_public_info(const ImageInfo &i, uint32_t flags, jxlamd_info *o) {
  memset(o, 0, sizeof(*o));
  o->xsize = i.xsize; o->ysize = i.ysize; o->bits_per_sample = i.bits_per_sample; o->exponent_bits_per_sample = i.exp_bits;
  o->num_color_channels = i.num_color_channels; o->num_extra_channels = i.num_extra_channels; o->alpha_bits = i.alpha_bits;
  o->alpha_premultiplied = i.alpha_premultiplied; o->orientation = 1; o->have_animation = i.have_animation;
  o->uses_original_profile = i.uses_original_profile;
  o->intensity_target = i.intensity_target <= 0.f ? 255.f : i.intensity_target;
  o->have_encoded_profile = !i.want_icc;
  o->color_space = i.color_space; o->white_point = i.white_point; o->primaries = i.primaries;
  o->transfer_function = i.have_gamma ? 65535u : i.transfer_function; o->rendering_intent = i.rendering_intent;
  o->gamma = i.have_gamma ? (double)i.gamma : 0.0;
  o->out_bits = (i.bits_per_sample > 8 && (flags & JXLAMD_ALLOW_16BIT)) ? 16 : 8;
  // JxlDecoding.cpp:126-133: `cs == RGB && tf == HLG || tf == PQ || ...` (&& binds tighter than ||)
  uint32_t tf = o->transfer_function;
  o->prefer_encoding = o->have_encoded_profile &&
                       ((o->color_space == 0 && tf == 18) || tf == 16 || tf == 17 || tf == 1 || tf == 13 || tf == 65535u);
  o->has_alpha_in_origin = i.num_extra_channels > 0 && i.alpha_bits > 0;
}

static int size_guard(const jxlamd_info &o, uint32_t flags, std::string *err) {
  uint64_t cur = (uint64_t)o.xsize * o.ysize * 4 * (o.out_bits == 16 ? 2 : 1);
  if (!(flags & JXLAMD_NO_SIZE_GUARD) && cur >= (uint64_t)std::numeric_limits<int32_t>::max()) {
    *err = "Invalid image size exceed allowance, current size w: " + std::to_string(o.xsize) + ", h: " + std::to_string(o.ysize);
    return JXLAMD_ERR_SIZE;
  }
  return 0;
}

int jxlamd_decoder::decode(const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info) {
  HIPCHECK(hipSetDevice(device));
  FramePlan plan;
  if (plan_parse(jxl, size, &plan)) {
    set_error(plan.error);
    return plan.error.rfind("unsupported", 0) == 0 ? JXLAMD_ERR_UNSUPPORTED : JXLAMD_ERR_INVALID;
  }
  jxlamd_info pi;
  fill_public_info(plan.info, flags, &pi);
  if (info) *info = pi;
  { std::string e; int rc = size_guard(pi, flags, &e); if (rc) { set_error(e); return rc; } }
  const size_t out_bytes = (size_t)pi.xsize * pi.ysize * 4 * (pi.out_bits == 16 ? 2 : 1);
  if (out_cap < out_bytes) { set_error("output buffer too small"); return JXLAMD_ERR_BUFFER; }
  const size_t ncell = (size_t)plan.xb * plan.yb;
  const size_t ntile = (size_t)((plan.xb + 7) / 8) * ((plan.yb + 7) / 8);
  const size_t npx = ncell * 64;
  // ---- buffers
  if (!stat_uploaded) {
    const std::vector<uint8_t> &st = static_tables();
    HIPCHECK(stat.ensure(st.size()));
    HIPCHECK(hipMemcpyAsync(stat.p, st.data(), st.size(), hipMemcpyHostToDevice, stream));
    stat_uploaded = true;
  }
  const bool cs_alias = (flags & JXLAMD_IN_DEVICE) && jxl_dev && plan.cs_owned.empty();
  const uint8_t *d_cs;
  if (cs_alias) d_cs = (const uint8_t *)jxl_dev + (plan.cs - jxl);
  else {
    HIPCHECK(cs.ensure(plan.cs_size + 64));
    HIPCHECK(hipMemcpyAsync(cs.p, plan.cs, plan.cs_size, hipMemcpyHostToDevice, stream));
    HIPCHECK(hipMemsetAsync((uint8_t *)cs.p + plan.cs_size, 0, 64, stream));
    d_cs = (const uint8_t *)cs.p;
  }
  HIPCHECK(tables.ensure(plan.tables.size() + (8u << 20)));     // room for the phase-2 (HfGlobal) tables
  HIPCHECK(hipMemcpyAsync(tables.p, plan.tables.data(), plan.tables.size(), hipMemcpyHostToDevice, stream));
  for (int i = 0; i < 5; i++) HIPCHECK(cells8[i].ensure(ncell));
  for (int i = 0; i < 2; i++) HIPCHECK(tiles[i].ensure(ntile));
  for (int i = 0; i < 6; i++) HIPCHECK(lf[i].ensure(ncell * 4));
  HIPCHECK(coef_off.ensure(ncell * 4));
  for (int c = 0; c < 3; c++) HIPCHECK(coef[c].ensure((size_t)plan.num_groups * 65536 * 4));
  for (int i = 0; i < 6; i++) HIPCHECK(planes[i].ensure(npx * 4));
  HIPCHECK(lf_scratch.ensure((size_t)plan.num_lf_groups * kLfScratchInts * 4));
  HIPCHECK(local.ensure((size_t)plan.num_lf_groups * sizeof(LocalTreeScratch)));
  HIPCHECK(misc.ensure(4096 + (size_t)plan.num_lf_groups * 72));
  void *d_out = out_ptr;
  if (!(flags & JXLAMD_OUT_DEVICE)) { HIPCHECK(out.ensure(out_bytes)); d_out = out.p; }
  DevBuffers B;
  memset(&B, 0, sizeof(B));
  B.codestream = d_cs; B.tables = (const uint8_t *)tables.p;
  B.strategy = (uint8_t *)cells8[0].p; B.first = (uint8_t *)cells8[1].p; B.qfm1 = (uint8_t *)cells8[2].p;
  B.sharp = (uint8_t *)cells8[3].p; B.lf_idx = (uint8_t *)cells8[4].p;
  B.xfromy = (int8_t *)tiles[0].p; B.bfromy = (int8_t *)tiles[1].p;
  for (int c = 0; c < 3; c++) { B.lf[c] = (float *)lf[c].p; B.lf_s[c] = (float *)lf[3 + c].p; B.coef[c] = (int32_t *)coef[c].p;
                                B.plane_a[c] = (float *)planes[c].p; B.plane_b[c] = (float *)planes[3 + c].p; }
  B.coef_off = (uint32_t *)coef_off.p; B.lf_scratch = (int32_t *)lf_scratch.p; B.local = (LocalTreeScratch *)local.p;
  B.err = (uint32_t *)misc.p; B.out = (uint8_t *)d_out;
  DevAux A; A.lf_end_bits = (uint64_t *)((uint8_t *)misc.p + 4096); A.lf_times = (uint64_t *)((uint8_t *)misc.p + 4096 + (size_t)plan.num_lf_groups * 8);
  HIPCHECK(hipMemsetAsync(misc.p, 0, 4096 + (size_t)plan.num_lf_groups * 72, stream));
  HIPCHECK(hipMemsetAsync(B.strategy, 0xFF, ncell, stream));
  HIPCHECK(hipMemsetAsync(B.first, 0, ncell, stream));
  for (int c = 0; c < 3; c++) HIPCHECK(hipMemsetAsync(B.coef[c], 0, (size_t)plan.num_groups * 65536 * 4, stream));
  // ---- kernels
  HIPCHECK(hipEventRecord(ev[0], stream));
  launch_lf_groups(B, A, plan.num_lf_groups, stream);
  if (plan.single_section) {
    // HfGlobal follows LfGroup 0 in the same section: its bit position is only known now.
    uint64_t end_bit = 0; uint32_t derr = 0;
    HIPCHECK(hipMemcpyAsync(&end_bit, A.lf_end_bits, 8, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipMemcpyAsync(&derr, B.err, 4, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    if (derr) { set_error("corrupt or unsupported stream (device flags " + std::to_string(derr) + ", LfGroup)"); return (derr & ~(kErrBitstream | kErrAnsFinal)) ? JXLAMD_ERR_UNSUPPORTED : JXLAMD_ERR_INVALID; }
    if (plan_parse_hf_single(&plan, end_bit)) { set_error(plan.error); return plan.error.rfind("unsupported", 0) == 0 ? JXLAMD_ERR_UNSUPPORTED : JXLAMD_ERR_INVALID; }
    HIPCHECK(tables.ensure(plan.tables.size()));
    HIPCHECK(hipMemcpyAsync(tables.p, plan.tables.data(), plan.tables.size(), hipMemcpyHostToDevice, stream));
  }
  HIPCHECK(hipEventRecord(ev[1], stream));
  launch_lf_smooth(B, plan.xb, plan.yb, stream);
  launch_pass_groups(B, plan.num_groups, stream);
  HIPCHECK(hipEventRecord(ev[2], stream));
  launch_recon(B, (const uint8_t *)stat.p, plan.xb, plan.yb, stream);
  HIPCHECK(hipEventRecord(ev[3], stream));
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  bool src_a = launch_filters(B, plan.width, plan.height, F->gab, F->epf_iters, true, stream);
  launch_write(B, (const uint8_t *)stat.p, plan.width, plan.height, (int)pi.out_bits, src_a, stream);
  HIPCHECK(hipEventRecord(ev[4], stream));
  uint32_t derr = 0;
  HIPCHECK(hipMemcpyAsync(&derr, B.err, 4, hipMemcpyDeviceToHost, stream));
  if (!(flags & JXLAMD_OUT_DEVICE)) HIPCHECK(hipMemcpyAsync(out_ptr, d_out, out_bytes, hipMemcpyDeviceToHost, stream));
  HIPCHECK(hipStreamSynchronize(stream));
  HIPCHECK(hipGetLastError());
  for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&timing[i], ev[i], ev[i + 1]);
  (void)hipEventElapsedTime(&timing[4], ev[0], ev[4]);
  if (derr) {
    set_error("corrupt or unsupported stream (device flags " + std::to_string(derr) + ")");
    return (derr & ~(kErrBitstream | kErrAnsFinal)) ? JXLAMD_ERR_UNSUPPORTED : JXLAMD_ERR_INVALID;
  }
  return JXLAMD_OK;
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
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
    g_tls_error = "cannot open HIP device"; delete d; return nullptr;
  }
  for (auto &e : d->ev) (void)hipEventCreate(&e);
  return d;
}

void jxlamd_decoder_destroy(jxlamd_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  (void)hipStreamSynchronize(d->stream);
  DevMem *all[] = {&d->cs, &d->tables, &d->stat, &d->coef_off, &d->lf_scratch, &d->local, &d->misc, &d->out};
  for (auto *m : all) m->release();
  for (auto &m : d->cells8) m.release();
  for (auto &m : d->tiles) m.release();
  for (auto &m : d->lf) m.release();
  for (auto &m : d->coef) m.release();
  for (auto &m : d->planes) m.release();
  for (auto &e : d->ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(d->stream);
  delete d;
}

const char *jxlamd_last_error(const jxlamd_decoder *d) { return d ? d->error.c_str() : g_tls_error.c_str(); }

int jxlamd_basic_info(const uint8_t *jxl, size_t size, jxlamd_info *info) {
  ImageInfo ii; std::string err;
  if (parse_basic_info(jxl, size, &ii, &err)) { g_tls_error = err; return JXLAMD_ERR_INVALID; }
  fill_public_info(ii, JXLAMD_ALLOW_16BIT, info);
  return JXLAMD_OK;
}

int jxlamd_output_size(const uint8_t *jxl, size_t size, uint32_t flags, size_t *bytes) {
  ImageInfo ii; std::string err;
  if (parse_basic_info(jxl, size, &ii, &err)) { g_tls_error = err; return JXLAMD_ERR_INVALID; }
  jxlamd_info o; fill_public_info(ii, flags, &o);
  int rc = size_guard(o, flags, &err);
  if (rc) { g_tls_error = err; return rc; }
  *bytes = (size_t)o.xsize * o.ysize * 4 * (o.out_bits == 16 ? 2 : 1);
  return JXLAMD_OK;
}

int jxlamd_decode(jxlamd_decoder *d, const uint8_t *jxl, size_t size, uint32_t flags, void *out, size_t cap, jxlamd_info *info) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  return d->decode(jxl, size, nullptr, flags & ~JXLAMD_IN_DEVICE, out, cap, info);
}

int jxlamd_decode_resident(jxlamd_decoder *d, const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out,
                           size_t cap, jxlamd_info *info) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  return d->decode(jxl, size, jxl_dev, flags | JXLAMD_IN_DEVICE, out, cap, info);
}

int jxlamd_decode_batch(jxlamd_decoder *d, int n, const uint8_t *const *jxl, const size_t *sizes, uint32_t flags, void *const *outs,
                        const size_t *caps, jxlamd_info *infos) {
  if (!d) { g_tls_error = "null decoder"; return JXLAMD_ERR_DEVICE; }
  for (int i = 0; i < n; i++) {
    int rc = d->decode(jxl[i], sizes[i], nullptr, flags & ~JXLAMD_IN_DEVICE, outs[i], caps[i], infos ? &infos[i] : nullptr);
    if (rc) return rc;
  }
  return JXLAMD_OK;
}

int jxlamd_debug_lf_phases(jxlamd_decoder *d, int num_lf_groups, uint64_t *out) {
  if (!d || !d->misc.p) return JXLAMD_ERR_DEVICE;
  return hipMemcpy(out, (uint8_t *)d->misc.p + 4096 + (size_t)num_lf_groups * 8, (size_t)num_lf_groups * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : JXLAMD_ERR_DEVICE;
}

int jxlamd_last_timing(const jxlamd_decoder *d, float ms[5]) {
  if (!d) return JXLAMD_ERR_DEVICE;
  memcpy(ms, d->timing, sizeof(d->timing));
  return JXLAMD_OK;
}

}  // extern "C"
