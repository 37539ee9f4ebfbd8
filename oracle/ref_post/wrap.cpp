// oracle/ref_post/wrap.cpp — TEST INFRASTRUCTURE ONLY (never linked into or loaded by the product).
// C entry points around the reference's own first-party post-decode stages, compiled from the sources where they lie
// under /root/reference (recipe: Makefile next to this file; output: oracle/_ref/libref_post.so):
//   A10  applyColorMatrix / applyColorMatrix16Bit          jxlcoder/src/main/cpp/colorspaces/ColorMatrix.cpp:35-219
//   A11  AssociateAlphaRgba8/16, RgbaU16ToF, Rgba8ToF16, Rgba16ToRgba8, Rgba8To565, Rgba16To565,
//        Rgba8ToRGBA1010102, Rgba16ToRGBA1010102             jxlcoder/src/main/cpp/imagebit/*.cpp
// The colour-matrix set-up below follows the reference's call site (JniDecoding.cpp:138-228): source primaries ->
// XYZ -> Rec.709, transfer function selection and the "tone map only for PQ / HLG" rule.
#include <cstdint>
#include <cfloat>
#include <cstring>
#include "imagebit/RGBAlpha.h"
#include "imagebit/RgbaU16toHF.h"
#include "imagebit/Rgba8ToF16.h"
#include "imagebit/Rgba16.h"
#include "imagebit/Rgb565.h"
#include "imagebit/Rgb1010102.h"
#include "colorspaces/ColorMatrix.h"
#include "colorspaces/ColorSpaceProfile.h"
#include "colorspaces/ITUR.h"
#include "colorspaces/Trc.h"

extern "C" {

void refpost_associate8(uint8_t *p, uint32_t stride, uint32_t w, uint32_t h) { coder::AssociateAlphaRgba8(p, stride, p, stride, w, h); }
void refpost_associate16(uint16_t *p, uint32_t stride, uint32_t w, uint32_t h, uint32_t depth) { coder::AssociateAlphaRgba16(p, stride, p, stride, w, h, depth); }
void refpost_u16_to_f16(const uint16_t *s, uint32_t ss, uint16_t *d, uint32_t ds, uint32_t w, uint32_t h, uint32_t depth) { coder::RgbaU16ToF(s, ss, d, ds, w, h, depth); }
void refpost_rgba8_to_f16(const uint8_t *s, uint32_t ss, uint16_t *d, uint32_t ds, uint32_t w, uint32_t h, int attenuate) { coder::Rgba8ToF16(s, ss, d, ds, w, h, attenuate != 0); }
void refpost_rgba16_to_8(const uint16_t *s, uint32_t ss, uint8_t *d, uint32_t ds, uint32_t w, uint32_t h, uint32_t depth) { coder::Rgba16ToRgba8(s, ss, d, ds, w, h, depth); }
void refpost_rgba8_to_565(const uint8_t *s, uint32_t ss, uint16_t *d, uint32_t ds, uint32_t w, uint32_t h, int attenuate) { coder::Rgba8To565(s, ss, d, ds, w, h, attenuate != 0); }
void refpost_rgba16_to_565(const uint16_t *s, uint32_t ss, uint16_t *d, uint32_t ds, uint32_t w, uint32_t h, uint32_t depth) { coder::Rgba16To565(s, ss, d, ds, w, h, depth); }
void refpost_rgba8_to_1010102(const uint8_t *s, uint32_t ss, uint8_t *d, uint32_t ds, uint32_t w, uint32_t h, int attenuate) { coder::Rgba8ToRGBA1010102(s, ss, d, ds, w, h, attenuate != 0); }
void refpost_rgba16_to_1010102(const uint16_t *s, uint32_t ss, uint8_t *d, uint32_t ds, uint32_t w, uint32_t h, uint32_t depth) { coder::Rgba16ToRGBA1010102(s, ss, d, ds, w, h, depth); }

// primaries: libjxl JxlPrimaries (1 sRGB, 2 custom, 9 Rec.2100, 11 P3); tf: JxlTransferFunction (1 709, 8 linear, 13 sRGB,
// 16 PQ, 17 DCI, 18 HLG, 65535 gamma).  xy = {rx, ry, gx, gy, bx, by, wx, wy} for custom primaries.
// Returns 0 when the reference would not run the stage for this transfer function.
int refpost_color_matrix(void *pixels, uint32_t stride, uint32_t w, uint32_t h, int is16, int depth, int primaries, int tf, const double *xy,
                         float intensity_target, float *matrix_out) {
  TransferFunction fn = TransferFunction::Srgb;
  bool tone = true;
  switch (tf) {
    case 18: fn = TransferFunction::Hlg; break;
    case 17: fn = TransferFunction::Smpte428; tone = false; break;
    case 16: fn = TransferFunction::Pq; break;
    case 65535: fn = TransferFunction::Gamma2p2; tone = false; break;
    case 1: fn = TransferFunction::Itur709; tone = false; break;
    case 13: fn = TransferFunction::Srgb; tone = false; break;
    default: return 0;
  }
  Eigen::Matrix<float, 3, 2> prim;
  Eigen::Vector2f white;
  Eigen::Matrix3f src;
  if (primaries == 9) { prim << getRec2020Primaries(); white << getIlluminantD65(); }
  else if (primaries == 11) { prim << getDisplayP3Primaries(); white << getIlluminantD65(); }
  else if (primaries == 1) { prim << getSRGBPrimaries(); white << getIlluminantD65(); }
  else {
    prim << (float)xy[0], (float)xy[1], (float)xy[2], (float)xy[3], (float)xy[4], (float)xy[5];
    white << (float)xy[6], (float)xy[7];
  }
  src = GamutRgbToXYZ(prim, white);
  Eigen::Matrix3f dst = GamutRgbToXYZ(getRec709Primaries(), getIlluminantD65());
  Eigen::Matrix3f conv = dst.inverse() * src;
  ITURColorCoefficients coeffs = colorPrimariesComputeYCoeffs(prim, white);
  const float m[9] = {conv(0, 0), conv(0, 1), conv(0, 2), conv(1, 0), conv(1, 1), conv(1, 2), conv(2, 0), conv(2, 1), conv(2, 2)};
  if (matrix_out) memcpy(matrix_out, m, sizeof(m));
  if (is16) applyColorMatrix16Bit((uint16_t *)pixels, stride, w, h, (uint8_t)depth, m, fn, TransferFunction::Srgb, tone, coeffs, intensity_target);
  else applyColorMatrix((uint8_t *)pixels, stride, w, h, m, fn, TransferFunction::Srgb, tone, coeffs, intensity_target);
  return 1;
}

float refpost_to_linear(float v, int fn) { return toLinear(v, (TransferFunction)fn); }
float refpost_to_gamma(float v, int fn) { return toGamma(v, (TransferFunction)fn); }

}  // extern "C"
