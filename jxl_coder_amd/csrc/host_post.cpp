// jxl_coder_amd/csrc/host_post.cpp — see host_post.h.  Curves written from their defining standards (IEC 61966-2-1,
// BT.709, BT.2100 PQ / HLG with the BT.2100-2 OOTF, SMPTE 428-1, pure gamma 2.2) with the constants and the
// "SDR white = 203 nits = 1.0" scaling the reference uses (colorspaces/Trc.cpp:31-329), evaluated in float like there.
#include "host_post.h"
#include <cmath>
#include <cfloat>
#include <algorithm>

namespace jxlamd {
namespace {

enum Curve { kSrgb, k709, kGamma22, k428, kPq, kHlg };

float clamp01(float v) { return v < 0.0f ? 0.0f : v > 1.0f ? 1.0f : v; }

float decode_curve(float e, Curve c) {      // encoded value -> linear light
  switch (c) {
    case kSrgb: {
      const float knee = 0.0030412825601275209f, a = 0.0550107189475866f;
      if (e < 0.0f) return 0.0f;
      if (e < 12.92f * knee) return e / 12.92f;
      if (e < 1.0f) return powf((e + a) / (1.0f + a), 2.4f);
      return 1.0f;
    }
    case k709: {
      const float beta = 0.018053968510807f, alpha = 1.09929682680944f;
      if (e < 0.0f) return 0.0f;
      if (e < 4.5f * beta) return e / 4.5f;
      if (e < 1.0f) return powf((e + (alpha - 1.0f)) / alpha, 1.0f / 0.45f);
      return 1.0f;
    }
    case kGamma22: return powf(clamp01(e), 2.2f);
    case k428: return powf(std::max(e, 0.0f), 2.6f) / 0.91655527974030934f;
    case kPq: {
      if (!(e > 0.0f)) return 0.0f;
      const float m2 = 78.84375f, m1 = 0.1593017578125f, c1 = 0.8359375f, c2 = 18.8515625f, c3 = 18.6875f;
      const float p = powf(e, 1.0f / m2);
      const float num = std::max(p - c1, 0.0f), den = std::max(c2 - c3 * p, FLT_MIN);
      return powf(num / den, 1.0f / m1) * 10000.0f / 203.0f;
    }
    case kHlg: {
      if (e < 0.0f) return 0.0f;
      const float a = 0.17883277f, b = 0.28466892f, cc = 0.55991073f;
      const float scene = e <= 0.5f ? powf((e * e) * (1.0f / 3.0f), 1.2f) : powf((expf((e - cc) / a) + b) / 12.0f, 1.2f);
      return scene * 1000.0f / 203.0f;
    }
  }
  return clamp01(e);
}

float encode_srgb(float l) {
  const float knee = 0.0030412825601275209f, a = 0.0550107189475866f;
  if (l < 0.0f) return 0.0f;
  if (l < knee) return l * 12.92f;
  if (l < 1.0f) return (1.0f + a) * powf(l, 1.0f / 2.4f) - a;
  return 1.0f;
}

struct M3 { double v[3][3]; };

M3 inverse(const M3 &a) {
  M3 r;
  const double (*m)[3] = a.v;
  const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                     m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  const double id = 1.0 / det;
  r.v[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * id; r.v[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id; r.v[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
  r.v[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id; r.v[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id; r.v[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  r.v[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id; r.v[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id; r.v[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
  return r;
}

// RGB -> XYZ for chromaticities (rx, ry, gx, gy, bx, by) and white (wx, wy); entries rounded to float as the reference
// keeps them (Eigen::Matrix3f, colorspaces/ColorSpaceProfile.h:131-143)
M3 rgb_to_xyz(const double p[6], const double w[2]) {
  M3 c;
  for (int i = 0; i < 3; i++) {
    const float x = (float)p[2 * i], y = (float)p[2 * i + 1];
    c.v[0][i] = (double)(x / y); c.v[1][i] = 1.0; c.v[2][i] = (double)((1.0f - x - y) / y);
  }
  const float wx = (float)w[0], wy = (float)w[1];
  const double W[3] = {(double)(wx / wy), 1.0, (double)((1.0f - wx - wy) / wy)};
  const M3 ci = inverse(c);
  M3 r;
  for (int j = 0; j < 3; j++) {
    const float s = (float)(ci.v[j][0] * W[0] + ci.v[j][1] * W[1] + ci.v[j][2] * W[2]);
    for (int i = 0; i < 3; i++) r.v[i][j] = (double)(float)((float)c.v[i][j] * s);
  }
  return r;
}

}  // namespace

bool plan_color_matrix(bool is_u16, uint32_t depth, uint32_t primaries, uint32_t tf, const double xy[8], float intensity_target, ColorMatrixPlan *P) {
  Curve curve; bool tone = true;
  switch (tf) {                                   // JxlTransferFunction values; selection as in JniDecoding.cpp:140-165
    case 18: curve = kHlg; break;
    case 17: curve = k428; tone = false; break;
    case 16: curve = kPq; break;
    case 65535: curve = kGamma22; tone = false; break;      // the reference replaces any gamma by 2.2 here ("Make real gamma")
    case 1: curve = k709; tone = false; break;
    case 13: curve = kSrgb; tone = false; break;
    default: return false;
  }
  static const double kSrgbP[6] = {0.640, 0.330, 0.300, 0.600, 0.150, 0.060}, kP3[6] = {0.680, 0.320, 0.265, 0.690, 0.150, 0.060},
                      k2020[6] = {0.708, 0.292, 0.170, 0.797, 0.131, 0.046}, kD65[2] = {0.3127, 0.3290};
  const double *prim = primaries == 9 ? k2020 : primaries == 11 ? kP3 : primaries == 1 ? kSrgbP : xy;
  const double *white = (primaries == 9 || primaries == 11 || primaries == 1) ? kD65 : xy + 6;
  const M3 src = rgb_to_xyz(prim, white), dst = rgb_to_xyz(kSrgbP, kD65), di = inverse(dst);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) P->m[i * 3 + j] = (float)(di.v[i][0] * src.v[0][j] + di.v[i][1] * src.v[1][j] + di.v[i][2] * src.v[2][j]);
  P->tone_map = tone;
  const float ld = intensity_target / 203.0f;
  P->weight_a = (250.0f / 203.0f) / (ld * ld);
  P->weight_b = 1.0f / (250.0f / 203.0f);
  if (is_u16) {
    const uint32_t n = 1u << depth;
    const float cut = (float)n - 1.0f, inv = 1.0f / cut;
    P->lin_lut.resize(n); P->gam_lut.resize(n);
    for (uint32_t j = 0; j < n; j++) {
      P->lin_lut[j] = decode_curve((float)j * inv, curve);
      P->gam_lut[j] = (uint16_t)std::min(std::max(roundf(encode_srgb((float)j * inv) * cut), 0.0f), cut);
    }
    P->index_scale = cut; P->index_max = n - 1;
  } else {
    P->lin_lut.resize(256); P->gam_lut.resize(2049);
    for (uint32_t j = 0; j < 256; j++) P->lin_lut[j] = decode_curve((float)j * (1.0f / 255.0f), curve);
    for (uint32_t j = 0; j < 2049; j++) P->gam_lut[j] = (uint16_t)std::min(std::max(roundf(encode_srgb((float)j * (1.0f / 2048.0f)) * 255.0f), 0.0f), 255.0f);
    P->index_scale = 2048.0f; P->index_max = 2048;
  }
  return true;
}

}  // namespace jxlamd
