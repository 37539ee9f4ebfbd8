/* include/jxl_amd_libjxl.h — the SECONDARY drop-in boundary: the subset of libjxl's C API that the reference's decode
 * path calls, exported by jxl_coder_amd/compat/libjxl.so (+ the runner symbols of libjxl_threads.so) on top of libjxlamd.so, so that the
 * reference's own interop/JxlDecoding.cpp — compiled unchanged — decodes through the HIP kernels.
 *
 * Types and values restate the reference's vendored headers (jxlcoder/src/main/cpp/jxl/, libjxl 0.10 API); a CPU test compiles this
 * header next to them and compares every size, offset and enum value (tests/test_libjxl_abi.py).
 *   JxlDecoderStatus values            jxl/decode.h:122-327
 *   JxlPixelFormat / JxlDataType       jxl/types.h:40-104
 *   JxlBasicInfo                       jxl/codestream_header.h:95-261
 *   JxlColorEncoding                   jxl/color_encoding.h:114-153
 *   JxlParallelRunner                  jxl/parallel_runner.h:83-129, jxl/resizable_parallel_runner.h:44-72
 * Entry points (the 12 + 5 the decode path uses, reference call sites interop/JxlDecoding.cpp:46-171 and :181-224):
 *   JxlDecoderCreate :98  JxlDecoderDestroy :114  JxlDecoderReset :107  JxlDecoderSubscribeEvents :474  JxlDecoderSetParallelRunner :441
 *   JxlDecoderSetInput :615  JxlDecoderCloseInput :658  JxlDecoderReleaseInput :638  JxlDecoderProcessInput :599  JxlDecoderGetBasicInfo :671
 *   JxlDecoderGetICCProfileSize :796  JxlDecoderGetColorAsEncodedProfile :770  JxlDecoderGetColorAsICCProfile :814
 *   JxlDecoderImageOutBufferSize :999  JxlDecoderSetImageOutBuffer :1018  JxlDecoderVersion :38  JxlSignatureCheck :76
 *   JxlResizableParallelRunner / Create / Destroy / SetThreads / SuggestThreads.
 *
 * Behaviour (one-shot use, as the reference drives it): events come in libjxl's order — JXL_DEC_BASIC_INFO, JXL_DEC_COLOR_ENCODING,
 * JXL_DEC_NEED_IMAGE_OUT_BUFFER, JXL_DEC_FULL_IMAGE, JXL_DEC_SUCCESS — each only if subscribed; header-level calls are host-only, the pixels
 * are decoded by jxlamd_decode on the device named by JXLAMD_DEVICE (default 0) when the output buffer has been set.  Output formats:
 * 4 channels, JXL_TYPE_UINT8, or JXL_TYPE_UINT16 for images with more than 8 bits per sample (what the reference asks for,
 * JxlDecoding.cpp:63,96); anything else, a truncated input, or a feature the GPU path rejects gives JXL_DEC_ERROR (the reference
 * turns every failure into `return false`).  The parallel runner is accepted and never called: the device is the runner. */
#ifndef JXL_AMD_LIBJXL_H_
#define JXL_AMD_LIBJXL_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct JxlAmdCompatDecoder JxlAmdCompatDecoder;      /* what a JxlDecoder* of this library points to */

enum {                                   /* JxlDecoderStatus */
  JXLC_DEC_SUCCESS = 0, JXLC_DEC_ERROR = 1, JXLC_DEC_NEED_MORE_INPUT = 2, JXLC_DEC_NEED_IMAGE_OUT_BUFFER = 5,
  JXLC_DEC_BASIC_INFO = 0x40, JXLC_DEC_COLOR_ENCODING = 0x100, JXLC_DEC_FRAME = 0x400, JXLC_DEC_FULL_IMAGE = 0x1000
};
enum { JXLC_TYPE_FLOAT = 0, JXLC_TYPE_UINT8 = 2, JXLC_TYPE_UINT16 = 3, JXLC_TYPE_FLOAT16 = 5 };      /* JxlDataType */
enum { JXLC_SIG_NOT_ENOUGH_BYTES = 0, JXLC_SIG_INVALID = 1, JXLC_SIG_CODESTREAM = 2, JXLC_SIG_CONTAINER = 3 };   /* JxlSignature */

typedef struct { uint32_t num_channels; int data_type; int endianness; size_t align; } JxlcPixelFormat;
typedef struct { uint32_t xsize, ysize; } JxlcPreviewHeader;
typedef struct { uint32_t tps_numerator, tps_denominator, num_loops; int have_timecodes; } JxlcAnimationHeader;
/* jxl/codestream_header.h:325-425 (JxlBlendInfo, JxlLayerInfo, JxlFrameHeader) */
typedef struct { int blendmode; uint32_t source, alpha; int clamp; } JxlcBlendInfo;
typedef struct { int have_crop; int32_t crop_x0, crop_y0; uint32_t xsize, ysize; JxlcBlendInfo blend_info; uint32_t save_as_reference; } JxlcLayerInfo;
typedef struct { uint32_t duration, timecode, name_length; int is_last; JxlcLayerInfo layer_info; } JxlcFrameHeader;
typedef struct {
  int have_container;
  uint32_t xsize, ysize, bits_per_sample, exponent_bits_per_sample;
  float intensity_target, min_nits;
  int relative_to_max_display;
  float linear_below;
  int uses_original_profile, have_preview, have_animation;
  int orientation;
  uint32_t num_color_channels, num_extra_channels, alpha_bits, alpha_exponent_bits;
  int alpha_premultiplied;
  JxlcPreviewHeader preview;
  JxlcAnimationHeader animation;
  uint32_t intrinsic_xsize, intrinsic_ysize;
  uint8_t padding[100];
} JxlcBasicInfo;
typedef struct {
  int color_space; int white_point; double white_point_xy[2];
  int primaries; double primaries_red_xy[2], primaries_green_xy[2], primaries_blue_xy[2];
  int transfer_function; double gamma; int rendering_intent;
} JxlcColorEncoding;

typedef int (*JxlcParallelRunInit)(void *jpegxl_opaque, size_t num_threads);
typedef void (*JxlcParallelRunFunction)(void *jpegxl_opaque, uint32_t value, size_t thread_id);
typedef int (*JxlcParallelRunner)(void *runner_opaque, void *jpegxl_opaque, JxlcParallelRunInit init, JxlcParallelRunFunction func,
                                  uint32_t start_range, uint32_t end_range);

#ifndef JXL_AMD_LIBJXL_NO_PROTOTYPES     /* (the layout test includes the reference's jxl/decode.h, which declares the same names) */
uint32_t JxlDecoderVersion(void);
int JxlSignatureCheck(const uint8_t *buf, size_t len);
JxlAmdCompatDecoder *JxlDecoderCreate(const void *memory_manager);
void JxlDecoderReset(JxlAmdCompatDecoder *dec);
void JxlDecoderDestroy(JxlAmdCompatDecoder *dec);
int JxlDecoderSubscribeEvents(JxlAmdCompatDecoder *dec, int events_wanted);
int JxlDecoderSetParallelRunner(JxlAmdCompatDecoder *dec, JxlcParallelRunner runner, void *runner_opaque);
int JxlDecoderSetInput(JxlAmdCompatDecoder *dec, const uint8_t *data, size_t size);
size_t JxlDecoderReleaseInput(JxlAmdCompatDecoder *dec);
void JxlDecoderCloseInput(JxlAmdCompatDecoder *dec);
int JxlDecoderProcessInput(JxlAmdCompatDecoder *dec);
int JxlDecoderGetBasicInfo(const JxlAmdCompatDecoder *dec, JxlcBasicInfo *info);
int JxlDecoderGetColorAsEncodedProfile(const JxlAmdCompatDecoder *dec, int target, JxlcColorEncoding *color_encoding);
int JxlDecoderGetICCProfileSize(const JxlAmdCompatDecoder *dec, int target, size_t *size);
int JxlDecoderGetColorAsICCProfile(const JxlAmdCompatDecoder *dec, int target, uint8_t *icc_profile, size_t size);
int JxlDecoderImageOutBufferSize(const JxlAmdCompatDecoder *dec, const JxlcPixelFormat *format, size_t *size);
int JxlDecoderSetImageOutBuffer(JxlAmdCompatDecoder *dec, const JxlcPixelFormat *format, void *buffer, size_t size);
/* the calls of the reference's animated decoder (interop/JxlAnimatedDecoder.hpp:68-185, .cpp:28-222; jxl/decode.h): frame events, frame-indexed decode */
void JxlDecoderRewind(JxlAmdCompatDecoder *dec);
void JxlDecoderSkipFrames(JxlAmdCompatDecoder *dec, size_t amount);
int JxlDecoderSkipCurrentFrame(JxlAmdCompatDecoder *dec);
int JxlDecoderSetCoalescing(JxlAmdCompatDecoder *dec, int coalescing);
int JxlDecoderGetFrameHeader(const JxlAmdCompatDecoder *dec, JxlcFrameHeader *header);
/* libjxl_threads */
int JxlResizableParallelRunner(void *runner_opaque, void *jpegxl_opaque, JxlcParallelRunInit init, JxlcParallelRunFunction func,
                               uint32_t start_range, uint32_t end_range);
void *JxlResizableParallelRunnerCreate(const void *memory_manager);
void JxlResizableParallelRunnerSetThreads(void *runner_opaque, size_t num_threads);
uint32_t JxlResizableParallelRunnerSuggestThreads(uint64_t xsize, uint64_t ysize);
void JxlResizableParallelRunnerDestroy(void *runner_opaque);
#endif

#ifdef __cplusplus
}
#endif
#endif
