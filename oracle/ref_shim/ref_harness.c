/* oracle/_ref harness — TEST INFRASTRUCTURE ONLY (never linked by the product).
 *
 * Drives the reference's own prebuilt libjxl (jxlcoder/src/main/cpp/lib/x86_64/libjxl.so,
 * libjxl 0.12.0, loaded under the bionic shim next to this file) with exactly the call
 * sequence of the reference's driver:
 *   decode: jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:46-171  (DecodeJpegXlOneShot)
 *   encode: jxlcoder/src/main/cpp/interop/JxlEncoding.cpp:54-192  (EncodeJxlOneshot)
 * Compiled as C against the public headers where they lie (-I <reference>/jxlcoder/src/main/cpp).
 * The libraries are dlopen()ed RTLD_LOCAL (libjxl_threads exports libc++abi symbols).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jxl/decode.h"
#include "jxl/encode.h"
#include "jxl/resizable_parallel_runner.h"
#include "jxl/thread_parallel_runner.h"

typedef struct {
  uint32_t xsize, ysize, bits_per_sample, exponent_bits;
  uint32_t num_color_channels, num_extra_channels, alpha_bits, alpha_premultiplied;
  uint32_t orientation, have_animation, uses_original_profile, out_bits; /* 8,16 or 32(float) */
  float intensity_target;
  uint32_t prefer_encoding;       /* JxlDecoding.cpp:126-133 quirk reproduced */
  uint32_t have_encoded_profile;
  uint32_t color_space, white_point, primaries, transfer_function, rendering_intent;
  double gamma;
  uint32_t icc_size;
  uint32_t version;
  double xy[8];                   /* white_point_xy, primaries_red/green/blue_xy as JxlDecoderGetColorAsEncodedProfile fills them */
} RefInfo;

static void *h_jxl, *h_thr;
#define SYM(h, name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
  if (!p_##name) { fprintf(stderr, "ref_harness: missing %s\n", #name); return -100; }

static int load_libs(void) {
  if (h_jxl && h_thr) return 0;
  /* All libraries sit next to this harness (oracle/_ref/). Load by absolute path in dependency
   * order: glibc matches later DT_NEEDED entries ("libc.so", "libbrotlidec.so", ...) against the
   * sonames of objects that are already loaded, so no LD_LIBRARY_PATH is needed. */
  Dl_info di;
  char dir[4096];
  if (!dladdr((void *)&load_libs, &di) || !di.dli_fname) return -1;
  snprintf(dir, sizeof(dir), "%s", di.dli_fname);
  char *slash = strrchr(dir, '/');
  if (slash) *slash = 0; else strcpy(dir, ".");
  static const char *order[] = {"libc.so", "libm.so", "libdl.so", "liblog.so", "libbrotlicommon.so",
                                "libbrotlidec.so", "libbrotlienc.so", "libjxl_cms.so", "libjxl.so",
                                "libjxl_threads.so"};
  for (unsigned i = 0; i < sizeof(order) / sizeof(order[0]); i++) {
    char path[4300];
    snprintf(path, sizeof(path), "%s/%s", dir, order[i]);
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "ref_harness: %s\n", dlerror()); return -1; }
    if (!strcmp(order[i], "libjxl.so")) h_jxl = h;
    if (!strcmp(order[i], "libjxl_threads.so")) h_thr = h;
  }
  return 0;
}

int ref_version(void) {
  if (load_libs()) return -1;
  SYM(h_jxl, JxlDecoderVersion);
  return (int)p_JxlDecoderVersion();
}

void ref_free(void *p) { free(p); }

/* mode: 0 = what DecodeJpegXlOneShot does (u8, or u16 when bits>8 && allow16);
 *       1 = float32 RGBA in the data profile (debug aid);
 * threads: 0 = JxlResizableParallelRunnerSuggestThreads (the reference's choice), else fixed. */
int ref_decode(const uint8_t *jxl, size_t size, int threads, int allow16, int mode,
               uint8_t **out, size_t *out_size, RefInfo *ri, uint8_t *icc_out, size_t icc_cap) {
  if (load_libs()) return -1;
  SYM(h_jxl, JxlDecoderCreate); SYM(h_jxl, JxlDecoderDestroy); SYM(h_jxl, JxlDecoderSubscribeEvents);
  SYM(h_jxl, JxlDecoderSetParallelRunner); SYM(h_jxl, JxlDecoderSetInput); SYM(h_jxl, JxlDecoderCloseInput);
  SYM(h_jxl, JxlDecoderProcessInput); SYM(h_jxl, JxlDecoderGetBasicInfo); SYM(h_jxl, JxlDecoderGetICCProfileSize);
  SYM(h_jxl, JxlDecoderGetColorAsEncodedProfile); SYM(h_jxl, JxlDecoderGetColorAsICCProfile);
  SYM(h_jxl, JxlDecoderImageOutBufferSize); SYM(h_jxl, JxlDecoderSetImageOutBuffer); SYM(h_jxl, JxlDecoderVersion);
  SYM(h_thr, JxlResizableParallelRunner); SYM(h_thr, JxlResizableParallelRunnerCreate);
  SYM(h_thr, JxlResizableParallelRunnerDestroy); SYM(h_thr, JxlResizableParallelRunnerSetThreads);
  SYM(h_thr, JxlResizableParallelRunnerSuggestThreads);

  int rc = -2;
  *out = NULL; *out_size = 0;
  memset(ri, 0, sizeof(*ri));
  ri->version = p_JxlDecoderVersion();
  void *runner = p_JxlResizableParallelRunnerCreate(NULL);
  JxlDecoder *dec = p_JxlDecoderCreate(NULL);
  if (JXL_DEC_SUCCESS != p_JxlDecoderSubscribeEvents(dec, JXL_DEC_BASIC_INFO | JXL_DEC_COLOR_ENCODING | JXL_DEC_FULL_IMAGE)) goto done;
  if (JXL_DEC_SUCCESS != p_JxlDecoderSetParallelRunner(dec, p_JxlResizableParallelRunner, runner)) goto done;
  JxlBasicInfo info;
  JxlPixelFormat format = {4, JXL_TYPE_UINT8, JXL_NATIVE_ENDIAN, 0};
  size_t bps = 1;
  p_JxlDecoderSetInput(dec, jxl, size);
  p_JxlDecoderCloseInput(dec);
  ri->intensity_target = 255;
  for (;;) {
    JxlDecoderStatus st = p_JxlDecoderProcessInput(dec);
    if (st == JXL_DEC_ERROR) { rc = -3; goto done; }
    else if (st == JXL_DEC_NEED_MORE_INPUT) { rc = -4; goto done; }
    else if (st == JXL_DEC_BASIC_INFO) {
      if (JXL_DEC_SUCCESS != p_JxlDecoderGetBasicInfo(dec, &info)) { rc = -5; goto done; }
      ri->xsize = info.xsize; ri->ysize = info.ysize; ri->bits_per_sample = info.bits_per_sample;
      ri->exponent_bits = info.exponent_bits_per_sample;
      ri->num_color_channels = info.num_color_channels; ri->num_extra_channels = info.num_extra_channels;
      ri->alpha_bits = info.alpha_bits; ri->alpha_premultiplied = info.alpha_premultiplied;
      ri->orientation = info.orientation; ri->have_animation = info.have_animation;
      ri->uses_original_profile = info.uses_original_profile;
      ri->intensity_target = info.intensity_target <= 0.f ? 255.f : info.intensity_target;
      if (mode == 1) { format.data_type = JXL_TYPE_FLOAT; bps = 4; ri->out_bits = 32; }
      else if (info.bits_per_sample > 8 && allow16) { format.data_type = JXL_TYPE_UINT16; bps = 2; ri->out_bits = 16; }
      else { ri->out_bits = 8; }
      uint32_t nthr = threads > 0 ? (uint32_t)threads
                                  : p_JxlResizableParallelRunnerSuggestThreads(info.xsize, info.ysize);
      p_JxlResizableParallelRunnerSetThreads(runner, nthr);
    } else if (st == JXL_DEC_COLOR_ENCODING) {
      size_t icc_size = 0;
      if (JXL_DEC_SUCCESS != p_JxlDecoderGetICCProfileSize(dec, JXL_COLOR_PROFILE_TARGET_DATA, &icc_size)) { rc = -6; goto done; }
      ri->icc_size = (uint32_t)icc_size;
      JxlColorEncoding clr;
      if (JXL_DEC_SUCCESS == p_JxlDecoderGetColorAsEncodedProfile(dec, JXL_COLOR_PROFILE_TARGET_DATA, &clr)) {
        ri->have_encoded_profile = 1;
        ri->color_space = clr.color_space; ri->white_point = clr.white_point; ri->primaries = clr.primaries;
        ri->transfer_function = clr.transfer_function; ri->rendering_intent = clr.rendering_intent; ri->gamma = clr.gamma;
        ri->xy[0] = clr.white_point_xy[0]; ri->xy[1] = clr.white_point_xy[1]; ri->xy[2] = clr.primaries_red_xy[0]; ri->xy[3] = clr.primaries_red_xy[1];
        ri->xy[4] = clr.primaries_green_xy[0]; ri->xy[5] = clr.primaries_green_xy[1]; ri->xy[6] = clr.primaries_blue_xy[0]; ri->xy[7] = clr.primaries_blue_xy[1];
        if ((clr.color_space == JXL_COLOR_SPACE_RGB && clr.transfer_function == JXL_TRANSFER_FUNCTION_HLG) ||
            clr.transfer_function == JXL_TRANSFER_FUNCTION_PQ || clr.transfer_function == JXL_TRANSFER_FUNCTION_DCI ||
            clr.transfer_function == JXL_TRANSFER_FUNCTION_709 || clr.transfer_function == JXL_TRANSFER_FUNCTION_SRGB ||
            clr.transfer_function == JXL_TRANSFER_FUNCTION_GAMMA)
          ri->prefer_encoding = 1;
      }
      if (icc_out && icc_size && icc_size <= icc_cap)
        if (JXL_DEC_SUCCESS != p_JxlDecoderGetColorAsICCProfile(dec, JXL_COLOR_PROFILE_TARGET_DATA, icc_out, icc_size)) { rc = -7; goto done; }
    } else if (st == JXL_DEC_NEED_IMAGE_OUT_BUFFER) {
      size_t need = 0;
      if (JXL_DEC_SUCCESS != p_JxlDecoderImageOutBufferSize(dec, &format, &need)) { rc = -8; goto done; }
      size_t stride = (size_t)ri->xsize * 4 * bps;
      if (need != stride * ri->ysize) { rc = -9; goto done; }
      if (!*out) { *out = (uint8_t *)malloc(need); *out_size = need; }
      if (JXL_DEC_SUCCESS != p_JxlDecoderSetImageOutBuffer(dec, &format, *out, need)) { rc = -10; goto done; }
    } else if (st == JXL_DEC_FULL_IMAGE) {
      /* last frame wins (JxlDecoding.cpp:164-166) */
    } else if (st == JXL_DEC_SUCCESS) { rc = 0; goto done; }
    else { rc = -11; goto done; }
  }
done:
  p_JxlDecoderDestroy(dec);
  p_JxlResizableParallelRunnerDestroy(runner);
  if (rc != 0 && *out) { free(*out); *out = NULL; *out_size = 0; }
  return rc;
}

typedef struct {
  uint32_t xsize, ysize, num_channels /*1,3,4*/, bits /*8,16*/;
  int32_t lossless;
  float distance;
  int32_t effort, decoding_speed;
  int32_t gaborish, epf;           /* -1 = encoder default */
  int32_t primaries, transfer;     /* 0 = sRGB defaults; else JxlPrimaries / JxlTransferFunction ints */
  float intensity_target;          /* 0 = default */
  int32_t modular;                 /* -1 default, 0 VarDCT, 1 modular */
  int32_t threads;                 /* 0 = default */
  int32_t extra[8][2];             /* further (JxlEncoderFrameSettingId, value) pairs; id<0 = unused */
} RefEncParams;

/* ICC bytes for the NEXT ref_encode call: the profile is set with JxlEncoderSetICCProfile instead of an enum colour encoding, as the
 * reference's encoder does when the Bitmap carries a profile (interop/JxlEncoding.cpp:125-129).  size 0 clears it. */
static uint8_t *g_icc; static size_t g_icc_size;
void ref_set_icc(const uint8_t *icc, size_t size) {
  free(g_icc); g_icc = NULL; g_icc_size = 0;
  if (size) { g_icc = (uint8_t *)malloc(size); memcpy(g_icc, icc, size); g_icc_size = size; }
}

/* JxlBasicInfo.orientation (1..8) for the NEXT ref_encode call (test fixtures with a non-identity orientation); 0 / 1 = identity. */
static double g_custom_xy[8]; static int g_custom_xy_set;     /* white point xy, red, green, blue xy: custom colour encoding of the next ref_encode (0 entries: enum) */
void ref_set_custom_xy(const double *xy8) { g_custom_xy_set = xy8 != NULL; if (xy8) memcpy(g_custom_xy, xy8, sizeof(g_custom_xy)); }
static int g_orientation;
void ref_set_orientation(int o) { g_orientation = o; }

/* float samples for the next ref_encode: 1 = float32 (32 bits, 8 exponent bits; float32 buffer), 2 = float16 (16 / 5; float16 buffer) */
static int g_float = 0;
void ref_set_float(int mode) { g_float = mode; }
static int g_int_bits = 0;      /* > 0 with ref_set_float(1): the float32 pixel buffer is an INTEGER image of that many bits (17 .. 31; what cjxl makes of 24- / 32-bit PNM sources) */
void ref_set_int_bits(int bits) { g_int_bits = bits; }
/* the alpha of the next encodes is declared premultiplied (the pixels are taken as they are) */
static int g_premultiplied = 0;
void ref_set_premultiplied(int on) { g_premultiplied = on; }
/* one more extra channel for the next ref_encode: an 8-bit plane of the image's size and its JxlExtraChannelType (NULL: none) */
static const uint8_t *g_extra_plane = NULL; static int g_extra_type = 0;
void ref_set_extra_channel(const uint8_t *plane, int type) { g_extra_plane = plane; g_extra_type = type; }
int ref_encode(const void *pixels, size_t pixels_size, const RefEncParams *p, uint8_t **out, size_t *out_size) {
  if (load_libs()) return -1;
  SYM(h_jxl, JxlEncoderSetICCProfile);
  SYM(h_jxl, JxlEncoderCreate); SYM(h_jxl, JxlEncoderDestroy); SYM(h_jxl, JxlEncoderSetParallelRunner);
  SYM(h_jxl, JxlEncoderInitBasicInfo); SYM(h_jxl, JxlEncoderSetBasicInfo); SYM(h_jxl, JxlEncoderInitExtraChannelInfo);
  SYM(h_jxl, JxlEncoderSetExtraChannelInfo); SYM(h_jxl, JxlEncoderSetColorEncoding); SYM(h_jxl, JxlEncoderSetExtraChannelBuffer);
  SYM(h_jxl, JxlEncoderFrameSettingsCreate); SYM(h_jxl, JxlEncoderSetFrameDistance);
  SYM(h_jxl, JxlEncoderFrameSettingsSetOption); SYM(h_jxl, JxlEncoderSetFrameLossless);
  SYM(h_jxl, JxlEncoderAddImageFrame); SYM(h_jxl, JxlEncoderCloseInput); SYM(h_jxl, JxlEncoderProcessOutput);
  SYM(h_jxl, JxlColorEncodingSetToSRGB);
  SYM(h_thr, JxlThreadParallelRunner); SYM(h_thr, JxlThreadParallelRunnerCreate);
  SYM(h_thr, JxlThreadParallelRunnerDestroy); SYM(h_thr, JxlThreadParallelRunnerDefaultNumWorkerThreads);

  int rc = -2;
  *out = NULL; *out_size = 0;
  JxlEncoder *enc = p_JxlEncoderCreate(NULL);
  size_t nthr = p->threads > 0 ? (size_t)p->threads : p_JxlThreadParallelRunnerDefaultNumWorkerThreads();
  void *runner = p_JxlThreadParallelRunnerCreate(NULL, nthr);
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetParallelRunner(enc, p_JxlThreadParallelRunner, runner)) goto done;
  JxlPixelFormat pf = {p->num_channels, g_float == 2 ? JXL_TYPE_FLOAT16 : g_float ? JXL_TYPE_FLOAT : p->bits == 16 ? JXL_TYPE_UINT16 : JXL_TYPE_UINT8, JXL_NATIVE_ENDIAN, 0};
  JxlBasicInfo bi;
  p_JxlEncoderInitBasicInfo(&bi);
  bi.xsize = p->xsize; bi.ysize = p->ysize; bi.bits_per_sample = p->bits;
  if (g_float) { bi.bits_per_sample = g_float == 2 ? 16 : 32; bi.exponent_bits_per_sample = g_float == 2 ? 5 : 8; }      /* float samples: the pixel buffer holds float32 */
  if (g_float == 1 && g_int_bits > 0) { bi.bits_per_sample = (uint32_t)g_int_bits; bi.exponent_bits_per_sample = 0; }
  bi.uses_original_profile = p->lossless ? JXL_TRUE : JXL_FALSE;
  const int has_alpha = p->num_channels == 2 || p->num_channels == 4;       /* 2: grey + alpha */
  bi.num_color_channels = p->num_channels <= 2 ? 1 : 3;
  bi.alpha_premultiplied = g_premultiplied ? JXL_TRUE : JXL_FALSE;
  if (p->intensity_target > 0) bi.intensity_target = p->intensity_target;
  if (g_orientation >= 1 && g_orientation <= 8) bi.orientation = (JxlOrientation)g_orientation;
  if (has_alpha) { bi.num_extra_channels = 1; bi.alpha_bits = g_float ? bi.bits_per_sample : p->bits; if (g_float) bi.alpha_exponent_bits = bi.exponent_bits_per_sample; }
  if (g_extra_plane) bi.num_extra_channels += 1;      /* one more extra channel (depth, spot colour, ...): ref_set_extra_channel */
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetBasicInfo(enc, &bi)) { rc = -3; goto done; }
  if (has_alpha) {
    JxlExtraChannelInfo ci;
    p_JxlEncoderInitExtraChannelInfo(JXL_CHANNEL_ALPHA, &ci);
    ci.bits_per_sample = g_float ? bi.bits_per_sample : p->bits; if (g_float) ci.exponent_bits_per_sample = bi.exponent_bits_per_sample;
    ci.alpha_premultiplied = g_premultiplied ? JXL_TRUE : JXL_FALSE;
    if (JXL_ENC_SUCCESS != p_JxlEncoderSetExtraChannelInfo(enc, 0, &ci)) { rc = -4; goto done; }
  }
  if (g_extra_plane) {
    JxlExtraChannelInfo ci;
    p_JxlEncoderInitExtraChannelInfo((JxlExtraChannelType)g_extra_type, &ci);
    ci.bits_per_sample = 8;
    if (JXL_ENC_SUCCESS != p_JxlEncoderSetExtraChannelInfo(enc, has_alpha ? 1 : 0, &ci)) { rc = -4; goto done; }
  }
  JxlColorEncoding ce;
  p_JxlColorEncodingSetToSRGB(&ce, p->num_channels <= 2);
  if (p->primaries) ce.primaries = (JxlPrimaries)p->primaries;
  if (p->transfer) ce.transfer_function = (JxlTransferFunction)p->transfer;
  if (g_custom_xy_set) {
    ce.white_point = JXL_WHITE_POINT_CUSTOM; ce.primaries = JXL_PRIMARIES_CUSTOM;
    ce.white_point_xy[0] = g_custom_xy[0]; ce.white_point_xy[1] = g_custom_xy[1];
    ce.primaries_red_xy[0] = g_custom_xy[2]; ce.primaries_red_xy[1] = g_custom_xy[3]; ce.primaries_green_xy[0] = g_custom_xy[4]; ce.primaries_green_xy[1] = g_custom_xy[5];
    ce.primaries_blue_xy[0] = g_custom_xy[6]; ce.primaries_blue_xy[1] = g_custom_xy[7];
  }
  if (g_icc_size) { if (JXL_ENC_SUCCESS != p_JxlEncoderSetICCProfile(enc, g_icc, g_icc_size)) { rc = -5; goto done; } }
  else if (JXL_ENC_SUCCESS != p_JxlEncoderSetColorEncoding(enc, &ce)) { rc = -5; goto done; }
  JxlEncoderFrameSettings *fs = p_JxlEncoderFrameSettingsCreate(enc, NULL);
  if (!p->lossless && JXL_ENC_SUCCESS != p_JxlEncoderSetFrameDistance(fs, p->distance)) { rc = -6; goto done; }
  if (JXL_ENC_SUCCESS != p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_EFFORT, p->effort)) { rc = -7; goto done; }
  if (p->lossless && JXL_ENC_SUCCESS != p_JxlEncoderSetFrameLossless(fs, JXL_TRUE)) { rc = -8; goto done; }
  if (JXL_ENC_SUCCESS != p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_DECODING_SPEED, p->decoding_speed)) { rc = -9; goto done; }
  if (p->gaborish >= 0) p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_GABORISH, p->gaborish);
  if (p->epf >= 0) p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_EPF, p->epf);
  if (p->modular >= 0) p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_MODULAR, p->modular);
  for (int i = 0; i < 8; i++)
    if (p->extra[i][0] >= 0)
      if (JXL_ENC_SUCCESS != p_JxlEncoderFrameSettingsSetOption(fs, (JxlEncoderFrameSettingId)p->extra[i][0], p->extra[i][1])) { rc = -20 - i; goto done; }
  if (JXL_ENC_SUCCESS != p_JxlEncoderAddImageFrame(fs, &pf, pixels, pixels_size)) { rc = -10; goto done; }
  if (g_extra_plane) {
    JxlPixelFormat pf1 = {1, JXL_TYPE_UINT8, JXL_NATIVE_ENDIAN, 0};
    if (JXL_ENC_SUCCESS != p_JxlEncoderSetExtraChannelBuffer(fs, &pf1, g_extra_plane, (size_t)p->xsize * p->ysize, has_alpha ? 1 : 0)) { rc = -12; goto done; }
  }
  p_JxlEncoderCloseInput(enc);
  size_t cap = 1 << 16;
  uint8_t *buf = (uint8_t *)malloc(cap);
  uint8_t *next = buf;
  size_t avail = cap;
  JxlEncoderStatus st = JXL_ENC_NEED_MORE_OUTPUT;
  while (st == JXL_ENC_NEED_MORE_OUTPUT) {
    st = p_JxlEncoderProcessOutput(enc, &next, &avail);
    if (st == JXL_ENC_NEED_MORE_OUTPUT) {
      size_t off = (size_t)(next - buf);
      cap *= 2;
      buf = (uint8_t *)realloc(buf, cap);
      next = buf + off; avail = cap - off;
    }
  }
  if (st != JXL_ENC_SUCCESS) { free(buf); rc = -11; goto done; }
  *out = buf; *out_size = (size_t)(next - buf);
  rc = 0;
done:
  p_JxlEncoderDestroy(enc);
  p_JxlThreadParallelRunnerDestroy(runner);
  return rc;
}

/* JPEG -> JPEG XL as the reference's JxlConstruction does (interop/JxlConstruction.hpp:46-90): StoreJPEGMetadata, lossless frame settings,
 * effort, decoding speed 3, JxlEncoderAddJPEGFrame.  The result is a VarDCT frame in YCbCr with the JPEG's quantisation tables. */
int ref_encode_jpeg(const uint8_t *jpeg, size_t jpeg_size, int effort, uint8_t **out, size_t *out_size) {
  if (load_libs()) return -1;
  SYM(h_jxl, JxlEncoderCreate); SYM(h_jxl, JxlEncoderDestroy); SYM(h_jxl, JxlEncoderSetParallelRunner);
  SYM(h_jxl, JxlEncoderStoreJPEGMetadata); SYM(h_jxl, JxlEncoderFrameSettingsCreate); SYM(h_jxl, JxlEncoderSetFrameLossless);
  SYM(h_jxl, JxlEncoderFrameSettingsSetOption); SYM(h_jxl, JxlEncoderAddJPEGFrame); SYM(h_jxl, JxlEncoderCloseInput); SYM(h_jxl, JxlEncoderProcessOutput);
  SYM(h_thr, JxlThreadParallelRunner); SYM(h_thr, JxlThreadParallelRunnerCreate);
  SYM(h_thr, JxlThreadParallelRunnerDestroy); SYM(h_thr, JxlThreadParallelRunnerDefaultNumWorkerThreads);
  int rc = -2;
  *out = NULL; *out_size = 0;
  JxlEncoder *enc = p_JxlEncoderCreate(NULL);
  void *runner = p_JxlThreadParallelRunnerCreate(NULL, p_JxlThreadParallelRunnerDefaultNumWorkerThreads());
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetParallelRunner(enc, p_JxlThreadParallelRunner, runner)) goto done;
  if (JXL_ENC_SUCCESS != p_JxlEncoderStoreJPEGMetadata(enc, JXL_TRUE)) { rc = -3; goto done; }
  JxlEncoderFrameSettings *fs = p_JxlEncoderFrameSettingsCreate(enc, NULL);
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetFrameLossless(fs, JXL_TRUE)) { rc = -4; goto done; }
  if (JXL_ENC_SUCCESS != p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_EFFORT, effort)) { rc = -5; goto done; }
  if (JXL_ENC_SUCCESS != p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_DECODING_SPEED, 3)) { rc = -6; goto done; }
  if (JXL_ENC_SUCCESS != p_JxlEncoderAddJPEGFrame(fs, jpeg, jpeg_size)) { rc = -7; goto done; }
  p_JxlEncoderCloseInput(enc);
  size_t cap = 1 << 16;
  uint8_t *buf = (uint8_t *)malloc(cap), *next = buf;
  size_t avail = cap;
  JxlEncoderStatus st = JXL_ENC_NEED_MORE_OUTPUT;
  while (st == JXL_ENC_NEED_MORE_OUTPUT) {
    st = p_JxlEncoderProcessOutput(enc, &next, &avail);
    if (st == JXL_ENC_NEED_MORE_OUTPUT) { size_t off = (size_t)(next - buf); cap *= 2; buf = (uint8_t *)realloc(buf, cap); next = buf + off; avail = cap - off; }
  }
  if (st != JXL_ENC_SUCCESS) { free(buf); rc = -11; goto done; }
  *out = buf; *out_size = (size_t)(next - buf);
  rc = 0;
done:
  p_JxlEncoderDestroy(enc);
  p_JxlThreadParallelRunnerDestroy(runner);
  return rc;
}

/* ---- animations: an encoder entry for building fixtures with cropped / blended layers, and the reference's frame-indexed decode.
 * Encode: libjxl's public API as the reference's animated encoder drives it (interop/JxlAnimatedEncoder: JxlEncoderSetFrameHeader with duration),
 *         plus layer_info (crop, blend mode, source, save_as_reference) so that the fixtures cover what files from other encoders (cjxl from GIF / APNG) hold.
 * Decode: jxlcoder/src/main/cpp/interop/JxlAnimatedDecoder.cpp:28-144 (getFrame: rewind, JxlDecoderSkipFrames(position), coalescing on, first full image)
 *         and JxlAnimatedDecoder.hpp:68-185 (the constructor's frame walk with coalescing off: frame count and durations). */
typedef struct {
  const uint8_t *rgba; uint32_t w, h; int32_t x0, y0;
  int32_t blend_mode, source, save_as_reference; uint32_t duration;
  int32_t alpha_blend_mode;      /* < 0: the alpha channel blends as libjxl's encoder defaults it (like the colour); else its own BlendingInfo mode (JxlEncoderSetExtraChannelBlendInfo) */
} RefAnimFrame;

int ref_encode_anim(const RefAnimFrame *frames, int nframes, uint32_t W, uint32_t H, int lossless, float distance, int effort,
                    uint32_t tps_num, uint32_t tps_den, uint32_t loops, uint8_t **out, size_t *out_size) {
  if (load_libs()) return -1;
  SYM(h_jxl, JxlEncoderCreate); SYM(h_jxl, JxlEncoderDestroy); SYM(h_jxl, JxlEncoderSetParallelRunner); SYM(h_jxl, JxlEncoderInitBasicInfo);
  SYM(h_jxl, JxlEncoderSetBasicInfo); SYM(h_jxl, JxlEncoderInitExtraChannelInfo); SYM(h_jxl, JxlEncoderSetExtraChannelInfo);
  SYM(h_jxl, JxlColorEncodingSetToSRGB); SYM(h_jxl, JxlEncoderSetColorEncoding); SYM(h_jxl, JxlEncoderFrameSettingsCreate);
  SYM(h_jxl, JxlEncoderSetFrameDistance); SYM(h_jxl, JxlEncoderFrameSettingsSetOption); SYM(h_jxl, JxlEncoderSetFrameLossless);
  SYM(h_jxl, JxlEncoderAddImageFrame); SYM(h_jxl, JxlEncoderCloseInput); SYM(h_jxl, JxlEncoderProcessOutput);
  SYM(h_jxl, JxlEncoderInitFrameHeader); SYM(h_jxl, JxlEncoderSetFrameHeader); SYM(h_jxl, JxlEncoderSetExtraChannelBlendInfo);
  SYM(h_thr, JxlThreadParallelRunner); SYM(h_thr, JxlThreadParallelRunnerCreate); SYM(h_thr, JxlThreadParallelRunnerDestroy);
  SYM(h_thr, JxlThreadParallelRunnerDefaultNumWorkerThreads);
  int rc = -2;
  *out = NULL; *out_size = 0;
  JxlEncoder *enc = p_JxlEncoderCreate(NULL);
  void *runner = p_JxlThreadParallelRunnerCreate(NULL, p_JxlThreadParallelRunnerDefaultNumWorkerThreads());
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetParallelRunner(enc, p_JxlThreadParallelRunner, runner)) goto done;
  JxlBasicInfo bi;
  p_JxlEncoderInitBasicInfo(&bi);
  bi.xsize = W; bi.ysize = H; bi.bits_per_sample = 8; bi.num_color_channels = 3; bi.num_extra_channels = 1; bi.alpha_bits = 8;
  bi.alpha_premultiplied = g_premultiplied ? JXL_TRUE : JXL_FALSE;
  bi.uses_original_profile = lossless ? JXL_TRUE : JXL_FALSE;
  bi.have_animation = tps_num ? JXL_TRUE : JXL_FALSE;          /* tps_num == 0: a layered STILL (every frame a layer of one image; durations are not written) */
  bi.animation.tps_numerator = tps_num; bi.animation.tps_denominator = tps_den; bi.animation.num_loops = loops;
  bi.animation.have_timecodes = JXL_FALSE;
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetBasicInfo(enc, &bi)) { rc = -3; goto done; }
  JxlExtraChannelInfo ci;
  p_JxlEncoderInitExtraChannelInfo(JXL_CHANNEL_ALPHA, &ci);
  ci.bits_per_sample = 8; ci.alpha_premultiplied = g_premultiplied ? JXL_TRUE : JXL_FALSE;
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetExtraChannelInfo(enc, 0, &ci)) { rc = -4; goto done; }
  JxlColorEncoding ce;
  p_JxlColorEncodingSetToSRGB(&ce, JXL_FALSE);
  if (JXL_ENC_SUCCESS != p_JxlEncoderSetColorEncoding(enc, &ce)) { rc = -5; goto done; }
  JxlPixelFormat pf = {4, JXL_TYPE_UINT8, JXL_NATIVE_ENDIAN, 0};
  for (int i = 0; i < nframes; i++) {
    const RefAnimFrame *fr = &frames[i];
    JxlEncoderFrameSettings *fs = p_JxlEncoderFrameSettingsCreate(enc, NULL);
    if (!lossless && JXL_ENC_SUCCESS != p_JxlEncoderSetFrameDistance(fs, distance)) { rc = -6; goto done; }
    if (JXL_ENC_SUCCESS != p_JxlEncoderFrameSettingsSetOption(fs, JXL_ENC_FRAME_SETTING_EFFORT, effort)) { rc = -7; goto done; }
    if (lossless && JXL_ENC_SUCCESS != p_JxlEncoderSetFrameLossless(fs, JXL_TRUE)) { rc = -8; goto done; }
    JxlFrameHeader fh;
    p_JxlEncoderInitFrameHeader(&fh);
    fh.duration = tps_num ? fr->duration : 0;
    fh.layer_info.have_crop = (fr->x0 || fr->y0 || fr->w != W || fr->h != H) ? JXL_TRUE : JXL_FALSE;
    fh.layer_info.crop_x0 = fr->x0; fh.layer_info.crop_y0 = fr->y0; fh.layer_info.xsize = fr->w; fh.layer_info.ysize = fr->h;
    fh.layer_info.blend_info.blendmode = (JxlBlendMode)fr->blend_mode; fh.layer_info.blend_info.source = (uint32_t)fr->source;
    fh.layer_info.blend_info.alpha = 0; fh.layer_info.blend_info.clamp = JXL_FALSE;
    fh.layer_info.save_as_reference = (uint32_t)fr->save_as_reference;
    if (JXL_ENC_SUCCESS != p_JxlEncoderSetFrameHeader(fs, &fh)) { rc = -9; goto done; }
    if (fr->alpha_blend_mode >= 0) {
      JxlBlendInfo abi = fh.layer_info.blend_info;
      abi.blendmode = (JxlBlendMode)fr->alpha_blend_mode;
      if (JXL_ENC_SUCCESS != p_JxlEncoderSetExtraChannelBlendInfo(fs, 0, &abi)) { rc = -12; goto done; }
    }
    if (JXL_ENC_SUCCESS != p_JxlEncoderAddImageFrame(fs, &pf, fr->rgba, (size_t)fr->w * fr->h * 4)) { rc = -10; goto done; }
  }
  p_JxlEncoderCloseInput(enc);
  {
    size_t cap = 1 << 16, used = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    JxlEncoderStatus st;
    do {
      uint8_t *next = buf + used; size_t avail = cap - used;
      st = p_JxlEncoderProcessOutput(enc, &next, &avail);
      used = (size_t)(next - buf);
      if (st == JXL_ENC_NEED_MORE_OUTPUT) { cap *= 2; buf = (uint8_t *)realloc(buf, cap); }
    } while (st == JXL_ENC_NEED_MORE_OUTPUT);
    if (st != JXL_ENC_SUCCESS) { free(buf); rc = -11; goto done; }
    *out = buf; *out_size = used; rc = 0;
  }
done:
  p_JxlEncoderDestroy(enc);
  p_JxlThreadParallelRunnerDestroy(runner);
  return rc;
}

/* frame count and durations (ms) as the reference's JxlAnimatedDecoder constructor collects them (coalescing OFF); returns the count or < 0 */
int ref_anim_info(const uint8_t *jxl, size_t size, int32_t *durations_ms, int cap, int32_t *loops) {
  if (load_libs()) return -1;
  SYM(h_jxl, JxlDecoderCreate); SYM(h_jxl, JxlDecoderDestroy); SYM(h_jxl, JxlDecoderSubscribeEvents); SYM(h_jxl, JxlDecoderSetInput);
  SYM(h_jxl, JxlDecoderCloseInput); SYM(h_jxl, JxlDecoderProcessInput); SYM(h_jxl, JxlDecoderGetBasicInfo); SYM(h_jxl, JxlDecoderSetCoalescing);
  SYM(h_jxl, JxlDecoderGetFrameHeader); SYM(h_jxl, JxlDecoderSkipCurrentFrame);
  JxlDecoder *dec = p_JxlDecoderCreate(NULL);
  int n = -2;
  if (JXL_DEC_SUCCESS != p_JxlDecoderSubscribeEvents(dec, JXL_DEC_BASIC_INFO | JXL_DEC_FULL_IMAGE | JXL_DEC_FRAME)) goto done;
  if (JXL_DEC_SUCCESS != p_JxlDecoderSetCoalescing(dec, JXL_FALSE)) goto done;
  p_JxlDecoderSetInput(dec, jxl, size);
  p_JxlDecoderCloseInput(dec);
  JxlBasicInfo info; memset(&info, 0, sizeof(info));
  n = 0;
  for (;;) {
    JxlDecoderStatus st = p_JxlDecoderProcessInput(dec);
    if (st == JXL_DEC_ERROR || st == JXL_DEC_NEED_MORE_INPUT) { n = -3; break; }
    else if (st == JXL_DEC_BASIC_INFO) { if (JXL_DEC_SUCCESS != p_JxlDecoderGetBasicInfo(dec, &info)) { n = -4; break; } *loops = info.have_animation ? (int32_t)info.animation.num_loops : -1; }
    else if (st == JXL_DEC_FRAME) {
      JxlFrameHeader fh;
      if (JXL_DEC_SUCCESS != p_JxlDecoderGetFrameHeader(dec, &fh)) { n = -5; break; }
      int ms = 0;
      if (info.animation.tps_numerator) ms = (int)__builtin_roundf(1000.0f * (float)fh.duration * (float)info.animation.tps_denominator / (float)info.animation.tps_numerator);
      if (n < cap) durations_ms[n] = ms;
      n++;
    }
    else if (st == JXL_DEC_NEED_IMAGE_OUT_BUFFER) { if (JXL_DEC_SUCCESS != p_JxlDecoderSkipCurrentFrame(dec)) { n = -6; break; } }
    else if (st == JXL_DEC_FULL_IMAGE) continue;       /* (the reference stops at the first one and rewinds: a quirk that truncates its list to one entry when a frame is not skipped) */
    else if (st == JXL_DEC_SUCCESS) break;
  }
done:
  p_JxlDecoderDestroy(dec);
  return n;
}

/* coalesced frame `index` as RGBA8: getFrame's call sequence (JxlAnimatedDecoder.cpp:28-144) */
int ref_decode_frame(const uint8_t *jxl, size_t size, int index, uint8_t **out, size_t *out_size, uint32_t *w, uint32_t *h) {
  if (load_libs()) return -1;
  SYM(h_jxl, JxlDecoderCreate); SYM(h_jxl, JxlDecoderDestroy); SYM(h_jxl, JxlDecoderSubscribeEvents); SYM(h_jxl, JxlDecoderSetInput);
  SYM(h_jxl, JxlDecoderCloseInput); SYM(h_jxl, JxlDecoderProcessInput); SYM(h_jxl, JxlDecoderGetBasicInfo); SYM(h_jxl, JxlDecoderSetCoalescing);
  SYM(h_jxl, JxlDecoderSkipFrames); SYM(h_jxl, JxlDecoderImageOutBufferSize); SYM(h_jxl, JxlDecoderSetImageOutBuffer);
  SYM(h_jxl, JxlDecoderSetParallelRunner);
  SYM(h_thr, JxlResizableParallelRunner); SYM(h_thr, JxlResizableParallelRunnerCreate); SYM(h_thr, JxlResizableParallelRunnerDestroy);
  int rc = -2;
  *out = NULL; *out_size = 0;
  void *runner = p_JxlResizableParallelRunnerCreate(NULL);
  JxlDecoder *dec = p_JxlDecoderCreate(NULL);
  if (JXL_DEC_SUCCESS != p_JxlDecoderSubscribeEvents(dec, JXL_DEC_BASIC_INFO | JXL_DEC_FULL_IMAGE | JXL_DEC_FRAME)) goto done;
  if (JXL_DEC_SUCCESS != p_JxlDecoderSetParallelRunner(dec, p_JxlResizableParallelRunner, runner)) goto done;
  p_JxlDecoderSetInput(dec, jxl, size);
  p_JxlDecoderCloseInput(dec);
  p_JxlDecoderSkipFrames(dec, (size_t)index);
  if (JXL_DEC_SUCCESS != p_JxlDecoderSetCoalescing(dec, JXL_TRUE)) goto done;
  JxlPixelFormat format = {4, JXL_TYPE_UINT8, JXL_NATIVE_ENDIAN, 0};
  JxlBasicInfo info; memset(&info, 0, sizeof(info));
  int got = 0;
  for (;;) {
    JxlDecoderStatus st = p_JxlDecoderProcessInput(dec);
    if (st == JXL_DEC_ERROR || st == JXL_DEC_NEED_MORE_INPUT) { rc = -3; goto done; }
    else if (st == JXL_DEC_BASIC_INFO) { if (JXL_DEC_SUCCESS != p_JxlDecoderGetBasicInfo(dec, &info)) { rc = -4; goto done; } *w = info.xsize; *h = info.ysize; }
    else if (st == JXL_DEC_FRAME) continue;
    else if (st == JXL_DEC_NEED_IMAGE_OUT_BUFFER) {
      size_t need = 0;
      if (JXL_DEC_SUCCESS != p_JxlDecoderImageOutBufferSize(dec, &format, &need)) { rc = -5; goto done; }
      free(*out);
      *out = (uint8_t *)malloc(need); *out_size = need;
      if (JXL_DEC_SUCCESS != p_JxlDecoderSetImageOutBuffer(dec, &format, *out, need)) { rc = -6; goto done; }
      got = 1;
    }
    else if (st == JXL_DEC_FULL_IMAGE || st == JXL_DEC_SUCCESS) { rc = got ? 0 : -7; goto done; }
  }
done:
  p_JxlDecoderDestroy(dec);
  p_JxlResizableParallelRunnerDestroy(runner);
  if (rc) { free(*out); *out = NULL; *out_size = 0; }
  return rc;
}
