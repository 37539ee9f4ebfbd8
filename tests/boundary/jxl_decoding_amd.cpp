// tests/boundary/jxl_decoding_amd.cpp — the reference-side binding of INTEGRATION.md §2, COMPILED: DecodeJpegXlOneShot and
// DecodeBasicInfo with the reference's exact signatures (declared by the reference's own header, included from where it lies:
// jxlcoder/src/main/cpp/interop/JxlDecoding.h:54-65), implemented over the C-ABI of include/jxl_amd.h.  This is the file a
// maintainer of the reference would drop in place of interop/JxlDecoding.cpp.  Built by tests/test_boundary.py in the build
// container only (the reference tree is not on the GPU box); nothing of the reference is copied — the header is only #included.
#include "interop/JxlDecoding.h"
#include <string.h>
#include <memory>
#include "jxl_amd.h"

namespace {
struct DecoderDeleter { void operator()(jxlamd_decoder *d) const { jxlamd_decoder_destroy(d); } };
// one decoder context per thread, destroyed with the thread (the reference creates a libjxl decoder + runner per call, JxlDecoding.cpp:46-48)
jxlamd_decoder *thread_decoder() {
  static thread_local std::unique_ptr<jxlamd_decoder, DecoderDeleter> dec;
  if (!dec) dec.reset(jxlamd_decoder_create(0));
  return dec.get();
}
void fill_color_encoding(const jxlamd_info &i, JxlColorEncoding *c) {
  memset(c, 0, sizeof(*c));
  c->color_space = (JxlColorSpace)i.color_space; c->white_point = (JxlWhitePoint)i.white_point; c->primaries = (JxlPrimaries)i.primaries;
  c->transfer_function = (JxlTransferFunction)(i.transfer_function == 65535u ? JXL_TRANSFER_FUNCTION_GAMMA : i.transfer_function);
  c->rendering_intent = (JxlRenderingIntent)i.rendering_intent; c->gamma = i.gamma;
  for (int k = 0; k < 2; k++) { c->white_point_xy[k] = i.white_point_xy[k]; c->primaries_red_xy[k] = i.primaries_red_xy[k];
                                c->primaries_green_xy[k] = i.primaries_green_xy[k]; c->primaries_blue_xy[k] = i.primaries_blue_xy[k]; }
}
}  // namespace

bool DecodeJpegXlOneShot(const uint8_t *jxl, size_t size, std::vector<uint8_t> *pixels, size_t *xsize, size_t *ysize, std::vector<uint8_t> *iccProfile,
                         bool *useFloats, uint32_t *bitDepth, bool *alphaPremultiplied, bool allowedFloats, JxlOrientation *jxlOrientation,
                         bool *preferEncoding, JxlColorEncoding *colorEncoding, bool *hasAlphaInOrigin, float *intensityTarget) {
  const uint32_t flags = allowedFloats ? JXLAMD_ALLOW_16BIT : 0u;
  jxlamd_info info;
  if (jxlamd_basic_info(jxl, size, &info) != JXLAMD_OK) return false;
  size_t bytes = 0;
  int rc = jxlamd_output_size(jxl, size, flags, &bytes);
  if (rc == JXLAMD_ERR_SIZE) throw InvalidImageSizeException(info.xsize, info.ysize);          // JxlDecoding.cpp:103-109
  if (rc != JXLAMD_OK) return false;
  jxlamd_decoder *dec = thread_decoder();
  if (!dec) return false;
  pixels->resize(bytes);
  rc = jxlamd_decode(dec, jxl, size, flags, pixels->data(), pixels->size(), &info);
  if (rc == JXLAMD_ERR_SIZE) throw InvalidImageSizeException(info.xsize, info.ysize);
  if (rc != JXLAMD_OK) return false;                                                           // -> InvalidJXLException (JniDecoding.cpp:78)
  *xsize = info.xsize; *ysize = info.ysize;
  *useFloats = info.out_bits == 16; *bitDepth = info.out_bits;                                 // JxlDecoding.cpp:92-101
  *alphaPremultiplied = info.alpha_premultiplied != 0;
  *jxlOrientation = (JxlOrientation)info.orientation;                                          // already applied: identity
  *preferEncoding = info.prefer_encoding != 0;
  if (info.have_encoded_profile) fill_color_encoding(info, colorEncoding);
  *hasAlphaInOrigin = info.has_alpha_in_origin != 0;
  *intensityTarget = info.intensity_target;
  iccProfile->clear();                                                                         // JxlDecoding.cpp:135-144
  if (!*preferEncoding && info.icc_size) {
    iccProfile->resize(info.icc_size);
    size_t got = 0;
    if (jxlamd_get_icc(jxl, size, iccProfile->data(), iccProfile->size(), &got) != JXLAMD_OK) iccProfile->clear(); else iccProfile->resize(got);
  }
  return true;
}

bool DecodeBasicInfo(const uint8_t *jxl, size_t size, size_t *xsize, size_t *ysize) {
  jxlamd_info info;
  if (jxlamd_basic_info(jxl, size, &info) != JXLAMD_OK) return false;
  *xsize = info.xsize; *ysize = info.ysize;
  return true;
}

#include "boundary_entry.inc"
