// jxl_coder_amd/csrc/host_icc_lut.cpp — convertUseDefinedColorSpace (cpp/colorspaces/colorspace.cpp:38-86) for HBM-resident pixels.
// The reference runs Little CMS over every pixel on the CPU: embedded profile -> sRGB, perceptual intent, black-point compensation |
// no-white-on-white-fixup | copy-alpha, TYPE_RGBA_8 or TYPE_RGBA_16_PREMUL.  Here the host asks the system's Little CMS
// (liblcms2.so.2, loaded at run time — the reference vendors the same library under cpp/icc) for the transform ONCE per profile,
// sampled on a 3-D lattice with the reference's intent and flags, and the device applies it to the whole image with trilinear
// interpolation (post.hip k_post_icc_lut).  Only the few libary entry points below are declared; their prototypes and the format /
// flag constants are Little CMS's published API (lcms2.h).
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>
#include "host_post.h"

namespace jxlamd {
namespace {
typedef void *(*fn_open_mem)(const void *, uint32_t);
typedef void *(*fn_srgb)();
typedef void *(*fn_create_xform)(void *, uint32_t, void *, uint32_t, uint32_t, uint32_t);
typedef void (*fn_do_xform)(void *, const void *, void *, uint32_t);
typedef void (*fn_del_xform)(void *);
typedef int (*fn_close)(void *);
struct Lcms { fn_open_mem open_mem; fn_srgb srgb; fn_create_xform create; fn_do_xform run; fn_del_xform del; fn_close close; bool ok = false; };
const Lcms &lcms() {
  static Lcms L; static std::once_flag once;
  std::call_once(once, [] {
    void *h = dlopen("liblcms2.so.2", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    L.open_mem = (fn_open_mem)dlsym(h, "cmsOpenProfileFromMem"); L.srgb = (fn_srgb)dlsym(h, "cmsCreate_sRGBProfile");
    L.create = (fn_create_xform)dlsym(h, "cmsCreateTransform"); L.run = (fn_do_xform)dlsym(h, "cmsDoTransform");
    L.del = (fn_del_xform)dlsym(h, "cmsDeleteTransform"); L.close = (fn_close)dlsym(h, "cmsCloseProfile");
    L.ok = L.open_mem && L.srgb && L.create && L.run && L.del && L.close;
  });
  return L;
}
// lcms2.h: TYPE_RGB_16 = COLORSPACE_SH(PT_RGB = 4) | CHANNELS_SH(3) | BYTES_SH(2); INTENT_PERCEPTUAL = 0;
// cmsFLAGS_BLACKPOINTCOMPENSATION 0x2000, cmsFLAGS_NOWHITEONWHITEFIXUP 0x0004 (COPY_ALPHA is the kernel's job: alpha never enters the lattice)
constexpr uint32_t kTypeRgb16 = (4u << 16) | (3u << 3) | 2u, kTypeRgb8 = (4u << 16) | (3u << 3) | 1u, kIntentPerceptual = 0, kFlags = 0x2000u | 0x0004u;
}  // namespace

// lattice of n^3 RGB16 triples (r fastest); returns false with *err set when lcms or the profile is unusable
// eight_bit: the lattice of an RGBA8 image.  Little CMS optimises an 8-bit transform differently from a 16-bit one (a matrix-shaper profile runs
// through exact per-level tables, the 16-bit form through a resampled CLUT that is coarse near black: linear code 2 comes out as 12 instead
// of 22), and the reference transforms 8-bit images with TYPE_RGBA_8 (colorspace.cpp:59-66): the 256^3 lattice is therefore sampled through the
// 8-BIT transform — every possible input is a lattice point, the stage is then exactly the reference's per-pixel result.
bool build_icc_lut(const uint8_t *icc, size_t icc_size, int n, std::vector<uint16_t> *lut, std::string *err, bool eight_bit) {
  const Lcms &L = lcms();
  if (!L.ok) { *err = "unsupported: liblcms2.so.2 is not available for the ICC colour-space stage"; return false; }
  void *src = L.open_mem(icc, (uint32_t)icc_size);
  if (!src) { *err = "ColorProfile Allocation Failed"; return false; }                  // the reference logs this and returns the pixels untouched
  void *dst = L.srgb();
  if (eight_bit && n != 256) { *err = "8-bit lattice needs 256 points per axis"; L.close(dst); L.close(src); return false; }
  void *xf = eight_bit ? L.create(src, kTypeRgb8, dst, kTypeRgb8, kIntentPerceptual, kFlags) : L.create(src, kTypeRgb16, dst, kTypeRgb16, kIntentPerceptual, kFlags);
  bool ok = xf != nullptr;
  if (ok && eight_bit) {
    const size_t cnt = (size_t)256 * 256 * 256;
    std::vector<uint8_t> in(cnt * 3), outb(cnt * 3);
    size_t o = 0;
    for (int b = 0; b < 256; b++) for (int g = 0; g < 256; g++) for (int r = 0; r < 256; r++) { in[o++] = (uint8_t)r; in[o++] = (uint8_t)g; in[o++] = (uint8_t)b; }
    L.run(xf, in.data(), outb.data(), (uint32_t)cnt);
    lut->resize(cnt * 3);
    for (size_t i = 0; i < cnt * 3; i++) (*lut)[i] = (uint16_t)(outb[i] * 257u);      // the kernel's 8-bit path scales by 255 / 65535 and rounds: exact
    L.del(xf);
  } else if (ok) {
    std::vector<uint16_t> in((size_t)n * n * n * 3);
    size_t o = 0;
    // uniform lattice: with n = 256 every 8-bit level is a lattice point (no interpolation for RGBA8: gamut / white clipping puts kinks into
    // the transform that a coarser lattice smears by several code values), 16-bit input interpolates between them
    std::vector<uint16_t> axis((size_t)n);
    for (int i = 0; i < n; i++) axis[(size_t)i] = (uint16_t)lrint((double)i / (n - 1) * 65535.0);
    for (int b = 0; b < n; b++) for (int g = 0; g < n; g++) for (int r = 0; r < n; r++) { in[o++] = axis[(size_t)r]; in[o++] = axis[(size_t)g]; in[o++] = axis[(size_t)b]; }
    lut->resize(in.size());
    L.run(xf, in.data(), lut->data(), (uint32_t)((size_t)n * n * n));
    L.del(xf);
  } else *err = "ColorProfile Creation has hailed";
  L.close(dst); L.close(src);
  return ok;
}

}  // namespace jxlamd
