// jxl_coder_amd/csrc/libjxl_abi.cpp — the libjxl C-API subset of include/jxl_amd_libjxl.h over the C-ABI of include/jxl_amd.h
// (secondary drop-in boundary).  Host-only glue: an event state machine in libjxl's order around jxlamd_basic_info /
// jxlamd_get_icc / jxlamd_decode.  Reference call sites: jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:46-171 (decode), :181-224 (size).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>
#include "../../include/jxl_amd.h"
#include "../../include/jxl_amd_libjxl.h"

// built twice: -DJXLC_ONLY_DECODER -> compat/libjxl.so, -DJXLC_ONLY_THREADS -> compat/libjxl_threads.so (the reference links both names)
#ifndef JXLC_ONLY_THREADS
struct JxlAmdCompatDecoder {
  const uint8_t *in = nullptr; size_t in_size = 0; bool closed = false;
  int events = 0;
  // progress: 0 nothing, 1 headers parsed, 2 BASIC_INFO delivered, 3 COLOR_ENCODING delivered, 4 pixels decoded (FULL_IMAGE delivered), 5 done
  int stage = 0;
  bool failed = false;
  jxlamd_info info;
  std::vector<uint8_t> icc; bool icc_tried = false;
  void *out = nullptr; size_t out_size = 0; int out_type = 0;
  jxlamd_decoder *dev = nullptr;
  void reset_state() { stage = 0; failed = false; icc.clear(); icc_tried = false; out = nullptr; out_size = 0; out_type = 0; memset(&info, 0, sizeof(info)); }
};
typedef JxlAmdCompatDecoder D;
#endif

extern "C" {
#ifndef JXLC_ONLY_THREADS

uint32_t JxlDecoderVersion(void) { return 10 * 1000 + 1; }        // the API level of the reference's vendored headers (jxl/version.h:16-18: 0.10.1)

int JxlSignatureCheck(const uint8_t *buf, size_t len) {
  static const uint8_t kBox[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
  if (len == 0) return JXLC_SIG_NOT_ENOUGH_BYTES;
  if (buf[0] == 0xFF) { if (len < 2) return JXLC_SIG_NOT_ENOUGH_BYTES; return buf[1] == 0x0A ? JXLC_SIG_CODESTREAM : JXLC_SIG_INVALID; }
  const size_t n = std::min<size_t>(len, 12);
  if (memcmp(buf, kBox, n) != 0) return JXLC_SIG_INVALID;
  return len < 12 ? JXLC_SIG_NOT_ENOUGH_BYTES : JXLC_SIG_CONTAINER;
}

D *JxlDecoderCreate(const void *) { D *d = new (std::nothrow) D(); if (d) d->reset_state(); return d; }
void JxlDecoderReset(D *d) { if (!d) return; d->in = nullptr; d->in_size = 0; d->closed = false; d->events = 0; d->reset_state(); }
void JxlDecoderDestroy(D *d) { if (!d) return; if (d->dev) jxlamd_decoder_destroy(d->dev); delete d; }

int JxlDecoderSubscribeEvents(D *d, int events_wanted) {
  if (!d || d->stage != 0 || events_wanted < 0) return JXLC_DEC_ERROR;       // libjxl: only before the first ProcessInput
  d->events = events_wanted;
  return JXLC_DEC_SUCCESS;
}
int JxlDecoderSetParallelRunner(D *d, JxlcParallelRunner, void *) { return (d && d->stage == 0) ? JXLC_DEC_SUCCESS : JXLC_DEC_ERROR; }
int JxlDecoderSetInput(D *d, const uint8_t *data, size_t size) {
  if (!d || d->in) return JXLC_DEC_ERROR;                                    // libjxl: previous input must be released first
  d->in = data; d->in_size = size;
  return JXLC_DEC_SUCCESS;
}
size_t JxlDecoderReleaseInput(D *d) { if (!d) return 0; d->in = nullptr; d->in_size = 0; return 0; }
void JxlDecoderCloseInput(D *d) { if (d) d->closed = true; }

static int fail(D *d) { d->failed = true; return JXLC_DEC_ERROR; }

int JxlDecoderProcessInput(D *d) {
  if (!d || d->failed) return JXLC_DEC_ERROR;
  if (!d->in) return d->closed ? JXLC_DEC_ERROR : JXLC_DEC_NEED_MORE_INPUT;
  if (d->stage == 0) {
    const int sig = JxlSignatureCheck(d->in, d->in_size);
    if (sig == JXLC_SIG_INVALID) return fail(d);
    if (jxlamd_basic_info(d->in, d->in_size, &d->info) != JXLAMD_OK) {
      if (!d->closed) return JXLC_DEC_NEED_MORE_INPUT;                       // the one-shot parser cannot tell "truncated" from "corrupt": ask while input may still come
      return fail(d);
    }
    d->stage = 1;
  }
  if (d->stage == 1) { d->stage = 2; if (d->events & JXLC_DEC_BASIC_INFO) return JXLC_DEC_BASIC_INFO; }
  if (d->stage == 2) { d->stage = 3; if (d->events & JXLC_DEC_COLOR_ENCODING) return JXLC_DEC_COLOR_ENCODING; }
  if (d->stage == 3) {
    if (!(d->events & JXLC_DEC_FULL_IMAGE)) { d->stage = 5; return JXLC_DEC_SUCCESS; }
    if (!d->out) return JXLC_DEC_NEED_IMAGE_OUT_BUFFER;
    if (!d->dev) {
      const char *dv = getenv("JXLAMD_DEVICE");
      d->dev = jxlamd_decoder_create(dv ? atoi(dv) : 0);
      if (!d->dev) return fail(d);                                           // no GPU: never a CPU route
    }
    // the INT32_MAX allowance is the CALLER's check in the reference (JxlDecoding.cpp:103-109), libjxl itself decodes larger images
    const uint32_t flags = (d->out_type == JXLC_TYPE_UINT16 ? JXLAMD_ALLOW_16BIT : 0u) | JXLAMD_NO_SIZE_GUARD;
    jxlamd_info got;
    if (jxlamd_decode(d->dev, d->in, d->in_size, flags, d->out, d->out_size, &got) != JXLAMD_OK) return fail(d);
    d->stage = 4;
    return JXLC_DEC_FULL_IMAGE;
  }
  if (d->stage == 4) { d->stage = 5; return JXLC_DEC_SUCCESS; }
  return JXLC_DEC_SUCCESS;
}

int JxlDecoderGetBasicInfo(const D *d, JxlcBasicInfo *o) {
  if (!d || d->stage < 1) return JXLC_DEC_NEED_MORE_INPUT;
  if (!o) return JXLC_DEC_SUCCESS;
  memset(o, 0, sizeof(*o));
  const jxlamd_info &i = d->info;
  o->have_container = JxlSignatureCheck(d->in, d->in_size) == JXLC_SIG_CONTAINER;
  o->xsize = i.xsize; o->ysize = i.ysize; o->bits_per_sample = i.bits_per_sample; o->exponent_bits_per_sample = i.exponent_bits_per_sample;
  // jxlamd_info carries the reference's *derived* intensity (255 when the stream says <= 0, JxlDecoding.cpp:91); the derivation is idempotent
  o->intensity_target = i.intensity_target; o->min_nits = 0.f; o->relative_to_max_display = 0; o->linear_below = 0.f;
  o->uses_original_profile = (int)i.uses_original_profile; o->have_preview = 0; o->have_animation = (int)i.have_animation;
  o->orientation = (int)i.orientation;                                      // 1: the pixels come out re-oriented, as libjxl's default (keep_orientation off)
  o->num_color_channels = i.num_color_channels; o->num_extra_channels = i.num_extra_channels; o->alpha_bits = i.alpha_bits;
  o->alpha_exponent_bits = 0; o->alpha_premultiplied = (int)i.alpha_premultiplied;
  o->intrinsic_xsize = i.xsize; o->intrinsic_ysize = i.ysize;
  return JXLC_DEC_SUCCESS;
}

int JxlDecoderGetColorAsEncodedProfile(const D *d, int, JxlcColorEncoding *c) {
  if (!d || d->stage < 1) return JXLC_DEC_NEED_MORE_INPUT;
  const jxlamd_info &i = d->info;
  if (!i.have_encoded_profile) return JXLC_DEC_ERROR;                        // ICC-only image: libjxl has no enum description either
  if (!c) return JXLC_DEC_SUCCESS;
  memset(c, 0, sizeof(*c));
  c->color_space = (int)i.color_space; c->white_point = (int)i.white_point; c->primaries = (int)i.primaries;
  c->transfer_function = i.transfer_function == 65535u ? 65535 : (int)i.transfer_function; c->gamma = i.gamma;
  c->rendering_intent = (int)i.rendering_intent;
  for (int k = 0; k < 2; k++) { c->white_point_xy[k] = i.white_point_xy[k]; c->primaries_red_xy[k] = i.primaries_red_xy[k];
                                c->primaries_green_xy[k] = i.primaries_green_xy[k]; c->primaries_blue_xy[k] = i.primaries_blue_xy[k]; }
  return JXLC_DEC_SUCCESS;
}

static bool load_icc(D *d) {
  if (d->icc_tried) return !d->icc.empty();
  d->icc_tried = true;
  if (!d->info.icc_size) return false;
  d->icc.resize(d->info.icc_size);
  size_t got = 0;
  if (jxlamd_get_icc(d->in, d->in_size, d->icc.data(), d->icc.size(), &got) != JXLAMD_OK) { d->icc.clear(); return false; }
  d->icc.resize(got);
  return !d->icc.empty();
}
// The embedded profile of an ICC-coded image, or — enum-coded image with the LINEAR transfer function, the one case in which the reference reads
// the bytes of an enum encoding (it does not 'prefer' it: JxlDecoding.cpp:126-144) — the profile libjxl synthesises, byte for byte
// (host_icc_synth.inc; jxlamd_info::icc_size carries its size).  Other enum encodings: size 0, as before; the reference never asks for them.
int JxlDecoderGetICCProfileSize(const D *d, int, size_t *size) {
  if (!d || d->stage < 1) return JXLC_DEC_NEED_MORE_INPUT;
  if (size) *size = d->info.icc_size;
  return JXLC_DEC_SUCCESS;
}
int JxlDecoderGetColorAsICCProfile(const D *d, int, uint8_t *icc_profile, size_t size) {
  if (!d || d->stage < 1) return JXLC_DEC_NEED_MORE_INPUT;
  D *m = const_cast<D *>(d);
  if (!load_icc(m) || size < m->icc.size()) return JXLC_DEC_ERROR;
  memcpy(icc_profile, m->icc.data(), m->icc.size());
  return JXLC_DEC_SUCCESS;
}

static bool format_ok(const D *d, const JxlcPixelFormat *f, size_t *bytes) {
  if (!f || f->num_channels != 4 || f->align > 1) return false;
  if (f->endianness != 0 && f->endianness != 1) return false;               // native or little endian (the host is little endian)
  size_t bps;
  if (f->data_type == JXLC_TYPE_UINT8) bps = 1;
  else if (f->data_type == JXLC_TYPE_UINT16 && d->info.bits_per_sample > 8) bps = 2;    // libjxl would also widen 8-bit images; the reference never asks
  else return false;
  *bytes = (size_t)d->info.xsize * (size_t)d->info.ysize * 4 * bps;
  return true;
}
int JxlDecoderImageOutBufferSize(const D *d, const JxlcPixelFormat *format, size_t *size) {
  if (!d || d->stage < 1) return JXLC_DEC_NEED_MORE_INPUT;
  size_t b = 0;
  if (!format_ok(d, format, &b)) return JXLC_DEC_ERROR;
  if (size) *size = b;
  return JXLC_DEC_SUCCESS;
}
int JxlDecoderSetImageOutBuffer(D *d, const JxlcPixelFormat *format, void *buffer, size_t size) {
  if (!d || d->stage < 1) return JXLC_DEC_ERROR;
  size_t b = 0;
  if (!format_ok(d, format, &b) || !buffer || size < b) return JXLC_DEC_ERROR;
  d->out = buffer; d->out_size = size; d->out_type = format->data_type;
  return JXLC_DEC_SUCCESS;
}

#endif   // !JXLC_ONLY_THREADS
#ifndef JXLC_ONLY_DECODER
// ---- libjxl_threads: the runner object exists for the caller's sake only
struct CompatRunner { size_t threads = 1; };
int JxlResizableParallelRunner(void *, void *jpegxl_opaque, JxlcParallelRunInit init, JxlcParallelRunFunction func, uint32_t start_range, uint32_t end_range) {
  if (init) { const int r = init(jpegxl_opaque, 1); if (r) return r; }      // a correct sequential runner, should anybody call it
  for (uint32_t i = start_range; i < end_range; i++) func(jpegxl_opaque, i, 0);
  return 0;
}
void *JxlResizableParallelRunnerCreate(const void *) { return new (std::nothrow) CompatRunner(); }
void JxlResizableParallelRunnerSetThreads(void *r, size_t n) { if (r) ((CompatRunner *)r)->threads = n ? n : 1; }
uint32_t JxlResizableParallelRunnerSuggestThreads(uint64_t xsize, uint64_t ysize) {
  return (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, xsize * ysize / (256 * 256)));      // one thread per 256x256 group, at most 64
}
void JxlResizableParallelRunnerDestroy(void *r) { delete (CompatRunner *)r; }
#endif   // !JXLC_ONLY_DECODER

}  // extern "C"
