// jxl_coder_amd/csrc/host_post.h — host-side parameter preparation for the post-decode stages (no pixel work):
// the 3x3 gamut matrix, the linearisation / sRGB-gamma LUTs and the tone-mapper weights of the reference's
// applyColorMatrix (jxlcoder/src/main/cpp/colorspaces/ColorMatrix.cpp:35-219, call site JniDecoding.cpp:138-228).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace jxlamd {

struct ColorMatrixPlan {
  float m[9];                    // Rec.709 <- source primaries, row-major
  int tone_map;                  // PQ / HLG only (JniDecoding.cpp:138-166)
  float weight_a, weight_b;      // Rec2408ToneMapper.h:36-45 with display 250 nits, white 203 nits
  std::vector<float> lin_lut;    // 256 entries (u8) or 2^depth (u16)
  std::vector<uint16_t> gam_lut; // 2049 entries holding u8 values (u8) or 2^depth entries (u16)
  float index_scale;             // 2048 (u8) or 2^depth - 1 (u16)
  uint32_t index_max;
};

// false when the reference does not run the stage for this transfer function (linear, unknown)
bool plan_color_matrix(bool is_u16, uint32_t depth, uint32_t primaries, uint32_t transfer_function, const double xy[8],
                       float intensity_target, ColorMatrixPlan *plan);

// A8: the embedded profile -> sRGB transform of convertUseDefinedColorSpace (cpp/colorspaces/colorspace.cpp:38-86) sampled from Little CMS
// on an n^3 RGB16 lattice (host_icc_lut.cpp)
bool build_icc_lut(const uint8_t *icc, size_t icc_size, int n, std::vector<uint16_t> *lut, std::string *err, bool eight_bit = false);

}  // namespace jxlamd
