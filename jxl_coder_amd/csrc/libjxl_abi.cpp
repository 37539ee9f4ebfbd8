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
  // frames (animations): the list of regular frames, which of them the current mode emits (coalescing on: frames of non-zero duration and the last one),
  // the cursor, frames still to skip (JxlDecoderSkipFrames), and whether the current frame's pixels were waived (JxlDecoderSkipCurrentFrame)
  bool coalescing = true;
  std::vector<jxlamd_anim_frame> frames; jxlamd_anim_header anim;
  std::vector<int> emit;          // indices into `frames` of the frames this walk emits
  size_t cur = 0, skip = 0; int fstage = 0; bool waived = false;      // fstage: 0 before the FRAME event, 1 FRAME delivered (pixels pending), 2 done
  void reset_state() { stage = 0; failed = false; icc.clear(); icc_tried = false; out = nullptr; out_size = 0; out_type = 0; memset(&info, 0, sizeof(info));
                       frames.clear(); emit.clear(); cur = 0; skip = 0; fstage = 0; waived = false; memset(&anim, 0, sizeof(anim)); }
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
void JxlDecoderReset(D *d) { if (!d) return; d->in = nullptr; d->in_size = 0; d->closed = false; d->events = 0; d->coalescing = true; d->reset_state(); }
// back to the start of the file: the input has to be set again, the subscribed events and settings stay (jxl/decode.h: JxlDecoderRewind)
void JxlDecoderRewind(D *d) { if (!d) return; const int ev = d->events; const bool co = d->coalescing; d->in = nullptr; d->in_size = 0; d->closed = false; d->reset_state(); d->events = ev; d->coalescing = co; }
void JxlDecoderSkipFrames(D *d, size_t amount) { if (d) d->skip += amount; }
int JxlDecoderSetCoalescing(D *d, int coalescing) { if (!d || d->stage > 3) return JXLC_DEC_ERROR; d->coalescing = coalescing != 0; return JXLC_DEC_SUCCESS; }
int JxlDecoderSkipCurrentFrame(D *d) {
  if (!d || d->stage != 3 || d->fstage != 1) return JXLC_DEC_ERROR;        // only between a frame's FRAME event and its FULL_IMAGE
  d->waived = true;
  return JXLC_DEC_SUCCESS;
}
void JxlDecoderDestroy(D *d) { if (!d) return; if (d->dev) jxlamd_decoder_destroy(d->dev); delete d; }

int JxlDecoderSubscribeEvents(D *d, int events_wanted) {
  if (!d || d->stage != 0 || events_wanted < 0) return JXLC_DEC_ERROR;       // libjxl: only before the first ProcessInput (or after a rewind)
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
    if (!(d->events & (JXLC_DEC_FULL_IMAGE | JXLC_DEC_FRAME))) { d->stage = 5; return JXLC_DEC_SUCCESS; }
    if (d->frames.empty()) {
      // the frame walk: every regular frame of the file; with coalescing on libjxl emits the frames of non-zero duration and the last one
      int32_t n = 0;
      if (jxlamd_anim_frames(d->in, d->in_size, nullptr, 0, &n, &d->anim) != JXLAMD_OK || n < 1) return fail(d);
      d->frames.resize((size_t)n);
      if (jxlamd_anim_frames(d->in, d->in_size, d->frames.data(), n, &n, &d->anim) != JXLAMD_OK) return fail(d);
      for (int i = 0; i < n; i++) if (!d->coalescing || d->frames[(size_t)i].coalesced_index >= 0) d->emit.push_back(i);
      d->cur = std::min(d->skip, d->emit.size()); d->skip = 0; d->fstage = 0;
    }
    for (;;) {
      // JxlDecoderSkipFrames called after a frame's events (libjxl: skips from the current position): applied between frames
      if (d->fstage == 0 && d->skip) { d->cur = std::min(d->cur + d->skip, d->emit.size()); d->skip = 0; }
      if (d->cur >= d->emit.size()) { d->stage = 5; return JXLC_DEC_SUCCESS; }
      if (d->fstage == 0) { d->fstage = 1; d->waived = false; if (d->events & JXLC_DEC_FRAME) return JXLC_DEC_FRAME; }
      if (d->fstage == 1) {
        if (!(d->events & JXLC_DEC_FULL_IMAGE) || d->waived) { d->fstage = 0; d->cur++; d->out = nullptr; continue; }
        if (!d->out) return JXLC_DEC_NEED_IMAGE_OUT_BUFFER;
        // a frame's own layer (coalescing off) is not something this library renders: the reference only walks such frames (JxlDecoderSkipCurrentFrame)
        if (!d->coalescing) return fail(d);
        if (!d->dev) {
          const char *dv = getenv("JXLAMD_DEVICE");
          d->dev = jxlamd_decoder_create(dv ? atoi(dv) : 0);
          if (!d->dev) return fail(d);                                       // no GPU: never a CPU route
        }
        // the INT32_MAX allowance is the CALLER's check in the reference (JxlDecoding.cpp:103-109), libjxl itself decodes larger images
        const uint32_t flags = (d->out_type == JXLC_TYPE_UINT16 ? JXLAMD_ALLOW_16BIT : 0u) | JXLAMD_NO_SIZE_GUARD;
        jxlamd_info got;
        const int ci = d->frames[(size_t)d->emit[d->cur]].coalesced_index;
        if (jxlamd_decode_frame(d->dev, d->in, d->in_size, ci, flags, d->out, d->out_size, &got) != JXLAMD_OK) return fail(d);
        d->fstage = 0; d->cur++; d->out = nullptr;                           // libjxl: the output buffer is per frame
        return JXLC_DEC_FULL_IMAGE;
      }
    }
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
  if (i.have_animation) {
    jxlamd_anim_header h; int32_t n = 0;
    if (jxlamd_anim_frames(d->in, d->in_size, nullptr, 0, &n, &h) == JXLAMD_OK) {
      o->animation.tps_numerator = h.tps_numerator; o->animation.tps_denominator = h.tps_denominator; o->animation.num_loops = h.num_loops; o->animation.have_timecodes = (int)h.have_timecodes;
    }
  }
  return JXLC_DEC_SUCCESS;
}
// the frame the last JXL_DEC_FRAME event announced.  With coalescing on every frame covers the image and replaces it (what libjxl reports then); the
// layer geometry of a frame walked with coalescing off is not exposed (the reference reads the duration only: JxlAnimatedDecoder.hpp:139-151)
int JxlDecoderGetFrameHeader(const D *d, JxlcFrameHeader *h) {
  if (!d || d->stage != 3 || d->fstage != 1 || d->cur >= d->emit.size()) return JXLC_DEC_ERROR;
  if (!h) return JXLC_DEC_SUCCESS;
  memset(h, 0, sizeof(*h));
  const jxlamd_anim_frame &f = d->frames[(size_t)d->emit[d->cur]];
  uint32_t ticks = f.duration_ticks;
  h->duration = ticks; h->timecode = 0; h->name_length = 0; h->is_last = d->cur + 1 == d->emit.size();
  h->layer_info.have_crop = 0; h->layer_info.xsize = d->info.xsize; h->layer_info.ysize = d->info.ysize;
  h->layer_info.blend_info.blendmode = 0; h->layer_info.save_as_reference = 0;
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
  // an enum encoding whose profile this library does not synthesise reports size 0 above: "copying" those zero bytes succeeds — the reference's animated
  // decoder asks for the bytes of every file (JxlAnimatedDecoder.hpp:158-174) and only uses them when it does not 'prefer' the enum encoding
  if (m->info.icc_size == 0) return JXLC_DEC_SUCCESS;
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
