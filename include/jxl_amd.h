/* include/jxl_amd.h — C-ABI of the MI355X-native JPEG XL decode path (libjxlamd.so).
 *
 * Drop-in boundary: these entry points are what the reference's decode driver would bind instead of libjxl:
 *   jxlamd_decode        <->  DecodeJpegXlOneShot  (jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:36-176,
 *                                                   declaration interop/JxlDecoding.h:54-63)
 *   jxlamd_basic_info    <->  DecodeBasicInfo      (interop/JxlDecoding.cpp:178-225, JxlDecoding.h:65)
 * i.e. the libjxl calls JxlDecoderCreate/SetInput/ProcessInput/GetBasicInfo/GetColorAsEncodedProfile/
 * ImageOutBufferSize/SetImageOutBuffer (jxlcoder/src/main/cpp/jxl/decode.h:441-1018) collapsed into one call.
 * Plain pointers and sizes only; no C++/torch types.  See INTEGRATION.md for the reference-side stub.
 */
#ifndef JXL_AMD_H_
#define JXL_AMD_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct jxlamd_decoder jxlamd_decoder;

/* Mirrors the JxlBasicInfo / JxlColorEncoding fields the reference driver reads
 * (interop/JxlDecoding.cpp:81-144; jxl/codestream_header.h:95-261; jxl/color_encoding.h:153). */
typedef struct {
  uint32_t xsize, ysize;            /* oriented output size */
  uint32_t bits_per_sample;         /* of the ORIGINAL image */
  uint32_t exponent_bits_per_sample;
  uint32_t num_color_channels, num_extra_channels, alpha_bits, alpha_premultiplied;
  uint32_t orientation;             /* as libjxl reports after re-orienting: 1 (JXL_ORIENT_IDENTITY) */
  uint32_t have_animation, uses_original_profile;
  float intensity_target;           /* 255 when the stream says <= 0 (JxlDecoding.cpp:91) */
  /* colour encoding of the data profile (JxlColorEncoding enums) */
  uint32_t have_encoded_profile;
  uint32_t color_space, white_point, primaries, transfer_function, rendering_intent;
  double gamma;
  /* what DecodeJpegXlOneShot derives */
  uint32_t out_bits;                /* 8 or 16: bitDepth out-param */
  uint32_t prefer_encoding;         /* JxlDecoding.cpp:126-133 (operator-precedence quirk reproduced) */
  uint32_t has_alpha_in_origin;     /* num_extra_channels > 0 && alpha_bits > 0 (JxlDecoding.cpp:111) */
  /* JxlColorEncoding's chromaticities as libjxl fills them for every encoded profile (enum values resolved to their xy;
   * JniDecoding.cpp:177-186 reads them for custom primaries): white_point_xy, primaries_red/green/blue_xy; zero for grey primaries */
  double white_point_xy[2], primaries_red_xy[2], primaries_green_xy[2], primaries_blue_xy[2];
  uint32_t icc_size;                /* bytes jxlamd_get_icc returns for this file (0: none obtainable) */
  uint32_t reserved;
} jxlamd_info;

enum {
  JXLAMD_OK = 0,
  JXLAMD_ERR_INVALID = -1,       /* corrupt / truncated stream: the reference returns false -> InvalidJXLException */
  JXLAMD_ERR_UNSUPPORTED = -2,   /* valid JPEG XL feature this build does not decode on the GPU (message says which) */
  JXLAMD_ERR_SIZE = -3,          /* w*h*4*bytes >= INT32_MAX: the reference throws InvalidImageSizeException (JxlDecoding.cpp:103-109) */
  JXLAMD_ERR_DEVICE = -4,        /* HIP failure / no GPU: never falls back to a CPU path */
  JXLAMD_ERR_BUFFER = -5         /* output buffer too small */
};

/* flags for jxlamd_decode */
#define JXLAMD_ALLOW_16BIT 1u     /* "allowedFloats": bits_per_sample > 8 -> RGBA u16 (JxlDecoding.cpp:92-101) */
#define JXLAMD_OUT_DEVICE 2u      /* `out` is a device pointer (HBM-resident output, no D2H) */
#define JXLAMD_NO_SIZE_GUARD 4u   /* skip the INT32_MAX guard (BASELINE config 4: 32768^2 below the Bitmap layer) */
#define JXLAMD_IN_DEVICE 8u       /* `jxl_dev` passed to jxlamd_decode_resident holds the same bytes in HBM (any alignment, no padding needed:
                                     the decoder copies them device-to-device into its own padded stream buffer) */
#define JXLAMD_BAND_SHARED_GPU 16u /* jxlamd_band_begin: other bands of the frame are decoded on this GPU at the same time (several decoder contexts): the band's
                                     PassGroup stage takes the lane-per-group kernel from 1 024 groups on (throughput) instead of 4 096 (latency of a lone band) */

jxlamd_decoder *jxlamd_decoder_create(int device);        /* NULL if the HIP device cannot be opened */
void jxlamd_decoder_destroy(jxlamd_decoder *dec);
const char *jxlamd_last_error(const jxlamd_decoder *dec);  /* dec may be NULL: last error of this thread's stateless calls */

/* DecodeBasicInfo: header-only parse, host only. */
int jxlamd_basic_info(const uint8_t *jxl, size_t size, jxlamd_info *info);

/* The ICC profile DecodeJpegXlOneShot hands back when !prefer_encoding (interop/JxlDecoding.cpp:135-141): the codestream's embedded ICC
 * profile, decoded on the host (ISO/IEC 18181-1 Annex E.4).  info.icc_size = its size (0: none).  Host only, no decoder needed.
 * Profiles libjxl would SYNTHESISE for an enum colour encoding that the reference does not 'prefer' (linear / unknown transfer) are not
 * generated: icc_size stays 0 for those. */
int jxlamd_get_icc(const uint8_t *jxl, size_t size, uint8_t *icc, size_t capacity, size_t *icc_size);

/* Bytes needed for the RGBA output of this file under `flags`. */
int jxlamd_output_size(const uint8_t *jxl, size_t size, uint32_t flags, size_t *bytes);

/* DecodeJpegXlOneShot: RGBA interleaved, top-down, no row padding; u8, or u16 when JXLAMD_ALLOW_16BIT and the
 * image has more than 8 bits.  `out` is host memory unless JXLAMD_OUT_DEVICE. */
int jxlamd_decode(jxlamd_decoder *dec, const uint8_t *jxl, size_t size, uint32_t flags, void *out, size_t out_capacity,
                  jxlamd_info *info);

/* The Edge-Preserving Filter's normalisation, 1 / (sum of weights).  libjxl spells it ApproximateReciprocal; in the reference's prebuilt x86_64 library
 * (cpp/lib/x86_64/libjxl.so, an SSE2-only build reached through interop/JxlDecoding.cpp:75) that is the CPU's 12-bit rcpps, which leaves a filtered sample up to
 * 3e-4 (relative) off the exact quotient — at most one 8-bit code on photographs, up to 4 codes on a channel near 0 beside two near 1 (saturated hard edges),
 * where the inverse opsin matrix amplifies it.  mode 0 (default): the exact quotient (what libjxl's other builds approximate).  mode 1: that build's
 * instruction as a 2 048-entry table of the golden host's results (csrc/rcp12_lut.h) — the pixels the reference returned there, max |difference| 1 on every
 * fixture.  Applies to every later decode of this context.  Environment default for new decoders: JXLAMD_EPF_RCP=x86. */
int jxlamd_decoder_set_epf_reciprocal(jxlamd_decoder *dec, int mode);

/* A10 + A11 with the decode ("fuse the colour matrix / tone map and the reformat into the writer").  Once set, every decode of this
 * context (single, batch, resident) delivers the Bitmap format of jxlamd_reformat_query(xsize, ysize, out_bits == 16, cfg, has_alpha_in_origin, api_level)
 * into `out` — rows of that stride — instead of RGBA8 / RGBA16: the colour matrix / Rec.2408 tone map where the reference's JNI layer applies it
 * (cpp/JniDecoding.cpp:131-228: 'preferred' RGB enum encoding, API level < 34), premultiply and conversion (cpp/ReformatBitmap.cpp:46-263).  Frames whose
 * last filter stage is a per-stage kernel (three EPF iterations: BASELINE config 5) emit it from that stage — the RGBA image is never stored —, the others
 * take one pass over their RGBA behind the writer; the pixels equal jxlamd_decode + jxlamd_post_fused bit for bit.  enabled = 0 restores RGBA output. */
int jxlamd_decoder_set_writer_post(jxlamd_decoder *dec, int enabled, int cfg, int api_level);

/* Animations (the reference's JxlAnimatedDecoder, interop/JxlAnimatedDecoder.hpp:68-185 and .cpp:28-144).
 * jxlamd_anim_info  = what its constructor collects: *num_frames regular frames, their durations in ms (round(1000 * ticks * tps_den / tps_num); the first
 *                     min(capacity, *num_frames) are stored), *loops = animation.num_loops (-1: not an animation).  Host only.
 * jxlamd_decode_frame = getFrame(frame): coalesced frame `frame` (frames of non-zero duration and the last one count), i.e. the frame laid over the canvas its
 *                     BlendingInfo names — kReplace / kAdd / kBlend / kMulAdd / kMul on colour and alpha, cropped layers, reference slots — decoded and
 *                     blended on the GPU.  jxlamd_decode is jxlamd_decode_frame of the last frame (interop/JxlDecoding.cpp:164-166). */
int jxlamd_anim_info(const uint8_t *jxl, size_t size, int32_t *durations_ms, int capacity, int32_t *num_frames, int32_t *loops);
/* the same walk with what libjxl's frame events carry (JxlFrameHeader, JxlAnimationHeader: jxl/codestream_header.h:77-90, 391-425) */
typedef struct { uint32_t duration_ticks; int32_t duration_ms; int32_t is_last; int32_t coalesced_index; /* -1: a zero-duration layer, shown as part of the next frame */ } jxlamd_anim_frame;
typedef struct { uint32_t have_animation, tps_numerator, tps_denominator, num_loops, have_timecodes; } jxlamd_anim_header;
int jxlamd_anim_frames(const uint8_t *jxl, size_t size, jxlamd_anim_frame *frames, int capacity, int32_t *num_frames, jxlamd_anim_header *header);
int jxlamd_decode_frame(jxlamd_decoder *dec, const uint8_t *jxl, size_t size, int frame, uint32_t flags, void *out, size_t out_capacity,
                        jxlamd_info *info);

/* Same, with the compressed bytes ALSO resident in device memory at `jxl_dev` (skips the H2D of the codestream;
 * the host copy is still needed for header/TOC parsing). */
int jxlamd_decode_resident(jxlamd_decoder *dec, const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags,
                           void *out, size_t out_capacity, jxlamd_info *info);

/* Batch extension (BASELINE configs 3/5): n independent files, outputs[i] sized by jxlamd_output_size.
 * Produces exactly what n jxlamd_decode calls would; the entropy stages of all frames share one launch each so that
 * their serial streams run side by side.  Returns the first failing status. */
int jxlamd_decode_batch(jxlamd_decoder *dec, int n, const uint8_t *const *jxl, const size_t *sizes, uint32_t flags,
                        void *const *outs, const size_t *out_capacities, jxlamd_info *infos);

/* Flights of several contexts on ONE set of HF-phase buffers: after this call `peer` runs the PassGroup / reconstruction / filter stages of its
 * batches in `owner`'s coefficient and pixel-plane pools (106 MB + 12 MB per 4K frame in flight) instead of its own, taking turns with the other
 * contexts that share them.  A flight needs those pools for about two thirds of its time; during its LF stage (a few MB per frame) they serve a
 * sharing context, so 2 contexts per pool set keep the pools busy without doubling the memory.  Both decoders must be on the same device; call
 * before the first batch of `peer`.  Single decodes are not affected.  (Extension; the reference decodes one file per call.) */
int jxlamd_decoder_share_pools(jxlamd_decoder *owner, jxlamd_decoder *peer);

/* Batch with the compressed bytes of frame i also resident in HBM at jxl_dev[i] (may be NULL per frame). */
int jxlamd_decode_batch_resident(jxlamd_decoder *dec, int n, const uint8_t *const *jxl, const size_t *sizes,
                                 const void *const *jxl_dev, uint32_t flags, void *const *outs, const size_t *out_capacities,
                                 jxlamd_info *infos);

/* ------------------------------------------------------------------------------------------------------------------
 * Band-sharded decode of ONE frame (BASELINE config 4; SURVEY.md §8b "decode_sharded", §8e).  Not in the reference: libjxl decodes a
 * frame in one process; the boundary this sits under is DecodeJpegXlOneShot's size guard (interop/JxlDecoding.cpp:103-109).
 * A band = group rows [group_row0, group_row1) (256-pixel rows of groups; use multiples of 8 to keep 2048x2048 LF groups whole —
 * other splits work but decode the shared LF groups on both sides).  `out` receives ONLY the band's pixel rows
 * [group_row0*256, min(group_row1*256, ysize)), tight RGBA.  Protocol per band context (every call returns with its work complete):
 *   jxlamd_band_begin        parse, upload, LF-group entropy stage of the band
 *   jxlamd_band_export/import(JXLAMD_HALO_LF, ...)      one cell row of LF / quant-field / sharpness per border
 *   jxlamd_band_reconstruct  LF smoothing, PassGroup entropy decode, dequant + inverse DCT
 *   jxlamd_band_export/import(JXLAMD_HALO_PIXELS, ...)  H rows of pre-filter XYB per border (H = 1 Gaborish + 3/2/1 per EPF iteration)
 *   jxlamd_band_finish       Gaborish / EPF / XYB->RGBA writer for the band's rows
 * export side 0 = this band's top edge (goes to the band above), 1 = its bottom edge; import side 0 = rows above this band (the
 * upper neighbour's side-1 export), 1 = rows below.  Bands at the image edge skip that side.  Halo buffers are device memory
 * (jxlamd_band_halo_bytes bytes) that the caller moves between GPUs (ncclSend/ncclRecv) or hands to the neighbour context directly.
 * The pixels of a band equal the same rows of jxlamd_decode of the whole frame bit for bit.
 * Supported: single-frame VarDCT without extra channels, orientation 1.  flags: JXLAMD_ALLOW_16BIT, JXLAMD_OUT_DEVICE. */
enum { JXLAMD_HALO_LF = 0, JXLAMD_HALO_PIXELS = 1 };
int jxlamd_band_begin(jxlamd_decoder *dec, const uint8_t *jxl, size_t size, uint32_t flags, int group_row0, int group_row1, void *out,
                      size_t out_capacity, jxlamd_info *info);
int jxlamd_band_halo_bytes(jxlamd_decoder *dec, int kind, size_t *bytes);
int jxlamd_band_export(jxlamd_decoder *dec, int kind, int side, void *halo_dev, size_t capacity);
int jxlamd_band_import(jxlamd_decoder *dec, int kind, int side, const void *halo_dev, size_t size);
int jxlamd_band_reconstruct(jxlamd_decoder *dec);
int jxlamd_band_finish(jxlamd_decoder *dec);
/* The cut of a frame of `ygroups` group rows (ceil(ysize / 256)) into `nbands` bands: rows[2 b] .. rows[2 b + 1] = the group rows of band b (borders on whole
 * 2048-pixel LF groups when there are enough of them). */
int jxlamd_band_rows(int ygroups, int nbands, int *rows);
/* BASELINE config 4 for the bands ONE process holds, without a host-language driver (SURVEY.md §8b "decode_sharded"; the reference has no counterpart: libjxl
 * decodes a frame in one process under interop/JxlDecoding.cpp:75): the frame as `nbands` bands of group rows (jxlamd_band_rows), band b on decoder context
 * decs[b] (all on one device, one context per band), its tight RGBA rows into the device buffer outs[b] of caps[b] bytes.  The bands go through every phase of
 * the protocol above side by side and hand their halo rows to their neighbours through device buffers.  Pixels = the same rows of a whole-frame decode. */
int jxlamd_decode_sharded_local(jxlamd_decoder *const *decs, int nbands, const uint8_t *jxl, size_t size, uint32_t flags, void *const *outs, const size_t *caps,
                                jxlamd_info *info);

/* Timing of the last decode in milliseconds (HIP events on the decoder's stream):
 * [0]=LF groups, [1]=pass groups, [2]=reconstruction, [3]=filters+write, [4]=total device time. */
int jxlamd_last_timing(const jxlamd_decoder *dec, float ms[5]);

/* Profiling aid: per-LF-group phase timestamps (100 MHz device wall clock), 8 x uint64 per LF group:
 * [0] start, [1] LF stream staged, [2] LF coefficients decoded, [3] metadata stream staged, [4] metadata decoded,
 * [5] varblocks placed, [6] epilogue done. */
int jxlamd_debug_lf_phases(jxlamd_decoder *dec, int num_lf_groups, uint64_t *out);
int jxlamd_debug_lf_phases_frame(jxlamd_decoder *dec, int frame, int num_lf_groups, uint64_t *out);   /* same, frame `frame` of the last flight */
/* 1 once a frame of this context needed the LF kernel build with the general lock-step loops (decoded again with it, and every later
 * decode of the context uses it): the lean build covers libjxl's LF-coefficient and HF-metadata streams. */
int jxlamd_debug_lf_general(const jxlamd_decoder *dec);
/* Measurement only: flights of every decoder of the process leave stages out — 1 the LF stage, 2 the PassGroup stage, 4 reconstruction + filters + writer (the pixels are
 * then whatever the slots held; 0 restores the decoder).  Stage floors of the flight pipeline: tools/gpu/ab.sh with JXLAMD_BENCH_ABLATE, profiles/r06_ablations_stage_floors.json. */
int jxlamd_debug_set_ablate(int mask);
/* out[0]: decodes / flights of this context that ran a second time because the LF table pool was too small (kErrNeedPool), out[1]: ... because
 * a stream needed the general build, out[2]: the pool (bytes) the next LF launch will get. */
int jxlamd_debug_lf_retries(const jxlamd_decoder *dec, uint32_t out[3]);
/* Flights of single-pass VarDCT frames hand their coefficients from the PassGroup stage to the reconstruction as per-varblock sparse lists (4 bytes per
 * nonzero coefficient) instead of dense 3 x 65 536 x int32 planes per group.  out[0]: the context's last flight did; out[1]: flights of this context that
 * were decoded again with the dense planes because a stream did not fit its lists. */
int jxlamd_debug_sparse(const jxlamd_decoder *dec, uint32_t out[2]);
/* *out: frames of this context's flights whose tail PassGroups (num_groups % 64, e.g. 7 of a 4K frame's 135) were decoded as second groups of the last full wave's lanes
 * instead of in a wave of their own (k_pass_flat, round 6; JXLAMD_PASS_CHAIN=0 switches it off for A/B measurements). */
int jxlamd_debug_pass_chain(const jxlamd_decoder *dec, uint32_t *out);
/* Modular streams (lossless frames, alpha, LF coefficients) are decoded by one wavefront in lock-step; an MA tree with more than 64 decision nodes or
 * leaves after pruning runs from its block form (at most 63 nodes per block), and what fits neither goes to a one-lane serial walker.  out[0]: streams
 * this context sent to the serial walker, out[1]: channels decoded from the block form — since the context was created. */
int jxlamd_debug_modular(const jxlamd_decoder *dec, uint64_t out[2]);

/* ------------------------------------------------------------------------------------------------------------------
 * Post-decode stages of the reference's JNI layer, on buffers that stay in HBM.
 * They run after jxlamd_decode*(…, JXLAMD_OUT_DEVICE) in the reference's order (cpp/JniDecoding.cpp:45-331):
 *   [A8 ICC, A9 rescale: not here]  ->  A10 jxlamd_color_matrix  ->  A11 jxlamd_reformat.
 * ------------------------------------------------------------------------------------------------------------------ */

/* PreferredColorConfig, the reference's values (cpp/Support.h:37-44, kt/PreferredColorConfig.kt) */
enum { JXLAMD_CFG_DEFAULT = 1, JXLAMD_CFG_RGBA_8888 = 2, JXLAMD_CFG_RGBA_F16 = 3, JXLAMD_CFG_RGB_565 = 4, JXLAMD_CFG_RGBA_1010102 = 5,
       JXLAMD_CFG_HARDWARE = 6 };
/* pixel layout of a reformatted buffer */
enum { JXLAMD_FMT_RGBA_8888 = 1, JXLAMD_FMT_RGBA_F16 = 2, JXLAMD_FMT_RGB_565 = 3, JXLAMD_FMT_RGBA_1010102 = 4 };

typedef struct jxlamd_reformat_info {
  uint32_t stride;            /* bytes per row of the destination (64-byte aligned rows for F16-from-u8 / 565 / 1010102, ReformatBitmap.cpp:105-107) */
  uint32_t format;            /* JXLAMD_FMT_* */
  uint32_t use_floats;        /* the reference's *useFloats after the stage */
  uint32_t resolved_config;   /* DEFAULT resolved as ReformatBitmap.cpp:52-63 does (needs the Android API level) */
  uint64_t bytes;             /* stride * height */
} jxlamd_reformat_info;

/* Destination geometry of jxlamd_reformat for a decoded w x h RGBA8 (src_is_u16 = 0) or RGBA16 (1) image.
 * Replaces the sizing logic of ReformatColorConfig (cpp/ReformatBitmap.cpp:46-263). */
int jxlamd_reformat_query(uint32_t w, uint32_t h, int src_is_u16, int color_config, int has_alpha_in_origin, int api_level,
                          jxlamd_reformat_info *out);

/* ReformatColorConfig (cpp/ReformatBitmap.cpp:46-263) on device buffers: premultiplies src IN PLACE when
 * !alpha_premultiplied && has_alpha_in_origin (imagebit/RGBAlpha.cpp:67-117), then converts into dst
 * (imagebit/RgbaU16toHF.cpp, Rgba8ToF16.cpp, Rgba16.cpp, Rgb565.cpp, Rgb1010102.cpp — including the reference's second
 * attenuation of u8 sources when !alpha_premultiplied).  src rows are tight (w*4 or w*8 bytes, what jxlamd_decode writes).
 * depth = the bitDepth DecodeJpegXlOneShot reports (8 or 16). */
int jxlamd_reformat(jxlamd_decoder *dec, void *src_dev, uint32_t w, uint32_t h, int src_is_u16, uint32_t depth, int color_config,
                    int alpha_premultiplied, int has_alpha_in_origin, int api_level, void *dst_dev, size_t dst_capacity,
                    jxlamd_reformat_info *out);

/* A10 + A11 in ONE pass: what jxlamd_color_matrix (when apply_color_matrix; same gate and parameters as the
 * applyColorMatrix block of cpp/JniDecoding.cpp:131-228, cpp/colorspaces/ColorMatrix.cpp:35-219) followed by jxlamd_reformat
 * (cpp/ReformatBitmap.cpp:46-263) produce, bit for bit, without the two in-place passes over the RGBA buffer in between.  src_dev is
 * read-only here (the two-call form premultiplies it in place). */
int jxlamd_post_fused(jxlamd_decoder *dec, const void *src_dev, uint32_t w, uint32_t h, int src_is_u16, uint32_t depth, int apply_color_matrix,
                      uint32_t primaries, uint32_t transfer_function, const double *xy8, float intensity_target, int color_config,
                      int alpha_premultiplied, int has_alpha_in_origin, int api_level, void *dst_dev, size_t dst_capacity, jxlamd_reformat_info *out);

/* applyColorMatrix / applyColorMatrix16Bit (cpp/colorspaces/ColorMatrix.cpp:35-219) with the set-up of the call site
 * (cpp/JniDecoding.cpp:138-228): linearise with the image's transfer function, Rec.2408 tone map for PQ / HLG
 * (colorspaces/Rec2408ToneMapper.cpp:80-100), source primaries -> Rec.709, sRGB OETF; in place, alpha untouched.
 * primaries / transfer_function: jxlamd_info values; xy8 = {rx,ry,gx,gy,bx,by,wx,wy} for custom primaries (may be NULL otherwise).
 * Returns JXLAMD_OK without touching the pixels when the reference would skip the stage for this transfer function
 * (the caller checks preferEncoding / colour space / API level < 34 as JniDecoding.cpp:131-137 does). */
int jxlamd_color_matrix(jxlamd_decoder *dec, void *pixels_dev, uint32_t w, uint32_t h, int is_u16, uint32_t depth, uint32_t primaries,
                        uint32_t transfer_function, const double *xy8, float intensity_target);

/* convertUseDefinedColorSpace (cpp/colorspaces/colorspace.cpp:38-86; cpp/JniDecoding.cpp:103-114): runs right after jxlamd_decode when the ICC
 * vector is non-empty, i.e. info.icc_size > 0 (jxlamd_get_icc).  Embedded profile -> sRGB with the reference's Little CMS parameters
 * (perceptual, black-point compensation, no-white-on-white-fixup, alpha copied; RGBA8, or RGBA16 treated as premultiplied), in place on the
 * device: the host samples the transform from the system's liblcms2 on a 256^3 lattice per profile (every 8-bit level a lattice point;
 * cached per decoder context), the kernel looks up (RGBA8) / interpolates (RGBA16).  Like the reference, a profile Little CMS cannot open leaves the pixels untouched (JXLAMD_OK). */
int jxlamd_icc_transform(jxlamd_decoder *dec, void *pixels_dev, uint32_t w, uint32_t h, int is_u16, const uint8_t *icc, size_t icc_size);

/* RescaleImage of `decodeSampled` (cpp/SizeScaler.cpp:38-144 -> weaver/src/scale.rs): runs between jxlamd_decode and jxlamd_color_matrix
 * (cpp/JniDecoding.cpp:116-136).  new_w / new_h: requested size, -1 = keep the aspect ratio, -2 = same rounded up to even
 * (resolve_dimensions, scale.rs:94-130); scale_mode: 1 Fit, 2 Fill, 3 Resize (cpp/SizeScaler.h:36-40; Fit / Fill scale uniformly and centre-crop,
 * scale.rs:202-234); sampler: XSampler 1..10 (cpp/XScaler.h).  premultiply_alpha = the reference's doesOriginHasAlpha.
 * The geometry is integer-exact; the filter arithmetic restates the published filter definitions (pic-scale is not vendored). */
typedef struct jxlamd_rescale_info { uint32_t scaled_w, scaled_h, crop_x, crop_y, out_w, out_h; } jxlamd_rescale_info;
int jxlamd_rescale_query(uint32_t w, uint32_t h, int new_w, int new_h, int scale_mode, jxlamd_rescale_info *out);
int jxlamd_rescale(jxlamd_decoder *dec, const void *src_dev, uint32_t w, uint32_t h, int src_is_u16, uint32_t depth, int new_w, int new_h, int scale_mode,
                   int sampler, int premultiply_alpha, void *dst_dev, size_t dst_capacity, jxlamd_rescale_info *out);

#ifdef __cplusplus
}
#endif
#endif
