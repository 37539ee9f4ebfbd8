// jxl_coder_amd/csrc/host_parse.h — host side of the decode: parses container, codestream headers, TOC and the
// global sections (LfGlobal, HfGlobal) and packs everything the kernels need into one "frame tables" blob
// (dev_types.h).  No pixel work happens on the host.
#pragma once
#include <stdint.h>
#include <memory>
#include <string>
#include <vector>
#include "dev_types.h"

namespace jxlamd {

struct ImageInfo {            // what DecodeJpegXlOneShot / DecodeBasicInfo report (JxlBasicInfo + colour encoding subset)
  uint32_t xsize = 0, ysize = 0;          // oriented
  uint32_t bits_per_sample = 8, exp_bits = 0;
  uint32_t num_color_channels = 3, num_extra_channels = 0, alpha_bits = 0, alpha_premultiplied = 0;
  uint32_t orientation = 1;               // reported AFTER re-orientation, like libjxl (always 1)
  uint32_t have_animation = 0, xyb_encoded = 1, uses_original_profile = 0;
  float intensity_target = 255.f;
  uint32_t want_icc = 0;
  uint32_t icc_size = 0;                   // bytes of the embedded ICC profile a non-XYB image reports (jxlamd_get_icc)
  uint32_t color_space = 0, white_point = 1, primaries = 1, transfer_function = 13, rendering_intent = 1;
  uint32_t have_gamma = 0; float gamma = 0;
  float wp_xy[2] = {0, 0}, prim_xy[6] = {0, 0, 0, 0, 0, 0};     // custom white point / primaries of the header (white_point == 2 / primaries == 2)
};

struct FramePlan {
  ImageInfo info;
  const uint8_t *cs = nullptr; size_t cs_size = 0;      // codestream bytes (may alias the input)
  std::vector<uint8_t> cs_owned;                         // assembled from jxlp boxes when needed
  std::vector<uint8_t> tables;                           // blob: DevFrame at 0
  bool single_section = false;
  bool modular = false;                                  // Modular-encoded (lossless) frame
  bool cropped = false;                                  // the decoded frame does not cover the image (have_crop): the output is cleared first
  bool has_ec = false;                                   // VarDCT frame with Modular-coded extra channels (alpha)
  size_t mod_pool_ints = 0;                              // int32 samples in the channel-plane pool
  // Composition (dev_compose.h).  refs: the earlier frames of the file this frame's patch dictionary draws on, in file order — each a complete
  // plan of its own over the same codestream bytes, decoded first and copied into reference slot save_slot.
  std::vector<std::shared_ptr<FramePlan>> refs;
  int save_slot = -1;                                    // a plan inside `refs`: where its image goes
  bool save_canvas = false;                              // ... as the blended canvas after the colour transform (4 planes of the IMAGE's size: a blend source), not as the frame itself
  bool blend = false;                                    // the frame is laid over a canvas (dev_compose.h: blend_canvas_pixel) instead of being written straight out
  bool compose = false;                                  // the frame keeps its image in the f32 planes after the filters (reference frame / patches / XYB Modular)
  size_t patch_max_px = 0;                               // largest patch rectangle (launch geometry of the blend kernel)
  bool hf_parsed = false;
  uint32_t lf_global_end_bit = 0;                        // single-section: where LfGroup 0 begins
  // geometry copies for the launcher
  int xb = 0, yb = 0, num_groups = 0, num_lf_groups = 0, num_passes = 1, width = 0, height = 0;
  std::string error;
  // internal parse state kept between phase 1 and 2 (single-section frames)
  std::shared_ptr<void> priv;
  // ready for the next parse; keeps the capacity of the byte vectors (a flight re-parses hundreds of frames per second per context:
  // fresh megabyte-sized allocations go through mmap / page faults, which serialise the parsing threads of all contexts)
  void reset() {
    std::vector<uint8_t> t; t.swap(tables); t.clear();
    std::vector<uint8_t> c; c.swap(cs_owned); c.clear();
    *this = FramePlan();
    tables.swap(t); cs_owned.swap(c);
  }
};

// Phase 1: everything up to and including LfGlobal; for multi-section frames also HfGlobal (phase 2 implicit).
// Returns 0 on success; on failure plan->error says why (unsupported feature or corrupt stream).
// target_frame: which coalesced frame of an animation to decode (what libjxl emits with coalescing on: every frame of non-zero duration, and the last
// one), -1 = the last — the reference's decode() keeps that one (interop/JxlDecoding.cpp:164-166), its JxlAnimatedDecoder::getFrame(i) frame i.
int plan_parse(const uint8_t *data, size_t size, FramePlan *plan, int target_frame = -1);
// The frame list of an animation as the reference's JxlAnimatedDecoder constructor collects it (interop/JxlAnimatedDecoder.hpp:68-185): one entry per
// regular frame, durations in milliseconds (round(1000 * ticks * tps_denominator / tps_numerator)), the loop count (-1: not an animation).  Returns 0.
struct AnimFrame { uint32_t ticks; int32_t ms; int32_t is_last; int32_t coalesced; };      // coalesced: index among the frames libjxl emits with coalescing on (-1: a zero-duration layer, merged into the next one)
struct AnimHeader { uint32_t have_animation, tps_numerator, tps_denominator, num_loops, have_timecodes; };
int parse_anim_info(const uint8_t *data, size_t size, std::vector<AnimFrame> *frames, AnimHeader *hdr, std::string *error);
// Phase 2 for single-section frames: HfGlobal starts at `lf_end_bit` (reported by the LF kernel).
int plan_parse_hf_single(FramePlan *plan, uint64_t lf_end_bit);
// Header-only parse (DecodeBasicInfo).
int parse_basic_info(const uint8_t *data, size_t size, ImageInfo *info, std::string *error, std::vector<uint8_t> *icc = nullptr);

// Per-process constant tables (inverse quant weights, cosine bases, AFV basis, dither LUT); header DevStatic at 0.
const std::vector<uint8_t> &static_tables();

}  // namespace jxlamd
