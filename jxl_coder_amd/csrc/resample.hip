// jxl_coder_amd/csrc/resample.hip — the `decodeSampled` resampler on HBM-resident RGBA buffers:
// what RescaleImage (cpp/SizeScaler.cpp:38-144) gets from the Rust crate `weaver` (weaver/src/scale.rs:94-130 resolve_dimensions,
// :202-234 Fit / Fill / Resize geometry, weaver/src/scaling_function.rs:65-78 filter map) -> pic-scale 0.7.6.
// pic-scale is an un-vendored dependency (weaver/Cargo.toml:6-9): its filter arithmetic is restated from the published definitions of
// the ten filters, as a two-pass separable convolution in f32 with alpha pre-multiplication when the origin has alpha
// (ScalingOptions.premultiply_alpha, scale.rs:184).  Parity for this stage is "own stated tolerance" (oracle/resample_oracle.py, +-1 LSB);
// the geometry (output size, crop window) is integer-exact.
#include "decoder_ctx.h"
#include <math.h>

namespace {

enum { kBilinear = 1, kNearest = 2, kCubic = 3, kMitchell = 4, kLanczos = 5, kCatmullRom = 6, kHermite = 7, kBSpline = 8, kHann = 9, kBicubic = 10 };   // XSampler, cpp/XScaler.h

__host__ __device__ inline float bc_spline(float x, float B, float C) {
  x = fabsf(x);
  if (x < 1.0f) return ((12 - 9 * B - 6 * C) * x * x * x + (-18 + 12 * B + 6 * C) * x * x + (6 - 2 * B)) * (1.0f / 6);
  if (x < 2.0f) return ((-B - 6 * C) * x * x * x + (6 * B + 30 * C) * x * x + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) * (1.0f / 6);
  return 0.0f;
}
__host__ __device__ inline float keys_cubic(float x, float a) {
  x = fabsf(x);
  if (x < 1.0f) return (a + 2) * x * x * x - (a + 3) * x * x + 1;
  if (x < 2.0f) return a * x * x * x - 5 * a * x * x + 8 * a * x - 4 * a;
  return 0.0f;
}
__host__ __device__ inline float sinc_pi(float x) { if (x == 0.0f) return 1.0f; const float t = 3.14159265358979f * x; return sinf(t) / t; }
__host__ __device__ inline float filter_radius(int f) {
  switch (f) { case kBilinear: case kHermite: return 1.0f; case kNearest: return 0.5f; case kLanczos: case kHann: return 3.0f; default: return 2.0f; }
}
__host__ __device__ inline float filter_weight(int f, float x) {
  switch (f) {
    case kBilinear: { x = fabsf(x); return x < 1.0f ? 1.0f - x : 0.0f; }
    case kNearest: return (x >= -0.5f && x < 0.5f) ? 1.0f : 0.0f;
    case kCubic: return keys_cubic(x, -0.5f);
    case kMitchell: return bc_spline(x, 1.0f / 3, 1.0f / 3);
    case kLanczos: case kHann: return fabsf(x) < 3.0f ? sinc_pi(x) * sinc_pi(x * (1.0f / 3)) : 0.0f;     // SizeScaler.cpp:79-82 maps hann to Lanczos too
    case kCatmullRom: return bc_spline(x, 0.0f, 0.5f);
    case kHermite: return bc_spline(x, 0.0f, 0.0f);
    case kBSpline: return bc_spline(x, 1.0f, 0.0f);
    default: return keys_cubic(x, -0.75f);                                                                 // bicubic
  }
}

// One 1-D pass.  kFirst: source is u8 / u16 RGBA (premultiplied here when asked), destination f32 RGBA [in_h][out_w];
// second pass: source f32 [in_h][w], destination u8 / u16 RGBA [out_h][w] (un-premultiplied, rounded, clamped).
template <bool kFirst, bool kU16>
__global__ void __launch_bounds__(256) k_resample(const void *__restrict__ src, void *__restrict__ dst, int in_len, int out_full, int other, int crop0, int out_len,
                                                  int src_stride_px, int filter, int premul, float maxv) {
  // consecutive work-items walk along x in both passes (coalesced rows): x is the resampled axis in pass 1, the other one in pass 2
  const int tx = (int)(blockIdx.y * 256 + threadIdx.x), ty = (int)blockIdx.x;      // rows in grid.x (grid.y is capped at 65 535)
  const int o = kFirst ? tx : ty;                                 // index along the resampled axis (inside the crop window)
  const int line = kFirst ? ty : tx;                              // index along the other axis
  if (o >= out_len || line >= other) return;
  const float scale = (float)in_len / (float)out_full;            // source samples per output sample
  const float fscale = scale > 1.0f ? scale : 1.0f;               // minification widens the kernel
  const float center = ((float)(o + crop0) + 0.5f) * scale;
  const float radius = filter_radius(filter) * fscale;
  int j0 = (int)floorf(center - radius), j1 = (int)ceilf(center + radius);
  if (filter == kNearest) { j0 = (int)floorf(center); j1 = j0 + 1; }
  float acc[4] = {0, 0, 0, 0}, wsum = 0.0f;
  for (int j = j0; j < j1; j++) {
    const float wgt = filter == kNearest ? 1.0f : filter_weight(filter, ((float)j + 0.5f - center) / fscale);
    if (wgt == 0.0f) continue;
    const int jj = j < 0 ? 0 : j >= in_len ? in_len - 1 : j;       // clamp-to-edge
    float px[4];
    if (kFirst) {
      const size_t idx = ((size_t)line * (size_t)src_stride_px + (size_t)jj) * 4;
      if (kU16) { const uint16_t *s = (const uint16_t *)src + idx; for (int c = 0; c < 4; c++) px[c] = (float)s[c]; }
      else { const uint8_t *s = (const uint8_t *)src + idx; for (int c = 0; c < 4; c++) px[c] = (float)s[c]; }
      if (premul) { const float a = px[3] / maxv; px[0] *= a; px[1] *= a; px[2] *= a; }
    } else {
      const float *s = (const float *)src + ((size_t)jj * (size_t)src_stride_px + (size_t)line) * 4;
      for (int c = 0; c < 4; c++) px[c] = s[c];
    }
    for (int c = 0; c < 4; c++) acc[c] += wgt * px[c];
    wsum += wgt;
  }
  const float inv = wsum != 0.0f ? 1.0f / wsum : 0.0f;
  for (int c = 0; c < 4; c++) acc[c] *= inv;
  if (kFirst) {
    float *d = (float *)dst + ((size_t)line * (size_t)out_len + (size_t)o) * 4;
    for (int c = 0; c < 4; c++) d[c] = acc[c];
  } else {
    if (premul) { const float a = acc[3] > 0.0f ? maxv / acc[3] : 0.0f; acc[0] *= a; acc[1] *= a; acc[2] *= a; }
    const size_t idx = ((size_t)o * (size_t)other + (size_t)line) * 4;
    for (int c = 0; c < 4; c++) {
      float v = rintf(acc[c]); v = v < 0.0f ? 0.0f : v > maxv ? maxv : v;
      if (kU16) ((uint16_t *)dst)[idx + c] = (uint16_t)v; else ((uint8_t *)dst)[idx + c] = (uint8_t)v;
    }
  }
}

long rust_round(double v) { return (long)(v < 0 ? -floor(-v + 0.5) : floor(v + 0.5)); }     // f64::round: half away from zero

}  // namespace

extern "C" {

// resolve_dimensions + the Fit / Fill / Resize window (weaver/src/scale.rs:94-130, :202-234), integer-exact
int jxlamd_rescale_query(uint32_t w, uint32_t h, int new_w, int new_h, int scale_mode, jxlamd_rescale_info *o) {
  if (!o || !w || !h || scale_mode < 1 || scale_mode > 3) { tls_error() = "Invalid Scale Mode was passed"; return JXLAMD_ERR_BUFFER; }
  size_t nw, nh;
  if (new_w > 0 && new_h == -1) { nw = (size_t)new_w; nh = (size_t)std::max(1l, rust_round((double)h * ((double)new_w / (double)w))); }
  else if (new_w > 0 && new_h == -2) { nw = (size_t)new_w; nh = ((size_t)std::max(1l, rust_round((double)h * ((double)new_w / (double)w))) + 1) & ~(size_t)1; }
  else if (new_w == -1 && new_h > 0) { nh = (size_t)new_h; nw = (size_t)std::max(1l, rust_round((double)w * ((double)new_h / (double)h))); }
  else if (new_w == -2 && new_h > 0) { nh = (size_t)new_h; nw = ((size_t)std::max(1l, rust_round((double)w * ((double)new_h / (double)h))) + 1) & ~(size_t)1; }
  else { nw = (size_t)std::max(1, new_w); nh = (size_t)std::max(1, new_h); }
  size_t sw, sh, cx, cy, cw, ch;
  if (scale_mode == 3) { sw = nw; sh = nh; cx = cy = 0; cw = nw; ch = nh; }
  else {
    const double xf = (double)nw / (double)w, yf = (double)nh / (double)h;
    const double sc = scale_mode == 2 ? std::max(xf, yf) : std::min(xf, yf);
    sw = (size_t)std::max(1l, rust_round((double)w * sc)); sh = (size_t)std::max(1l, rust_round((double)h * sc));
    cx = (size_t)std::max(0l, ((long)sw - (long)nw) / 2); cy = (size_t)std::max(0l, ((long)sh - (long)nh) / 2);
    cw = std::min(nw, sw); ch = std::min(nh, sh);
  }
  o->scaled_w = (uint32_t)sw; o->scaled_h = (uint32_t)sh; o->crop_x = (uint32_t)cx; o->crop_y = (uint32_t)cy; o->out_w = (uint32_t)cw; o->out_h = (uint32_t)ch;
  return JXLAMD_OK;
}

int jxlamd_rescale(jxlamd_decoder *d, const void *src_dev, uint32_t w, uint32_t h, int is_u16, uint32_t depth, int new_w, int new_h, int scale_mode, int sampler,
                   int premultiply_alpha, void *dst_dev, size_t dst_cap, jxlamd_rescale_info *out) {
  if (!d) { tls_error() = "null decoder"; return JXLAMD_ERR_DEVICE; }
  jxlamd_rescale_info q;
  int rc = jxlamd_rescale_query(w, h, new_w, new_h, scale_mode, &q);
  if (rc) { d->error = tls_error(); return rc; }
  if (out) *out = q;
  if (sampler < 1 || sampler > 10) { d->set_error("Invalid Sampler: " + std::to_string(sampler) + " was passed"); return JXLAMD_ERR_BUFFER; }
  const size_t bpp = is_u16 ? 8 : 4;
  if (!src_dev || !dst_dev || dst_cap < (size_t)q.out_w * q.out_h * bpp) { d->set_error("output buffer too small"); return JXLAMD_ERR_BUFFER; }
  if (is_u16 ? (depth < 9 || depth > 16) : depth != 8) { d->set_error("bit depth does not match the source format"); return JXLAMD_ERR_BUFFER; }
  if (hipSetDevice(d->device) != hipSuccess) { d->set_error("cannot select device"); return JXLAMD_ERR_DEVICE; }
  // pass 1 (horizontal) -> f32 [h][out_w]; pass 2 (vertical) -> destination
  const size_t tmp_bytes = (size_t)h * q.out_w * 16;
  if (d->resample_tmp.ensure(tmp_bytes) != hipSuccess) { d->set_error("HIP: out of memory for the resampler's intermediate"); return JXLAMD_ERR_DEVICE; }
  const float maxv = (float)((1u << depth) - 1);
  const hipStream_t s = d->stream;
  dim3 g1(h, (q.out_w + 255) / 256), g2(q.out_h, (q.out_w + 255) / 256);
  if (is_u16) {
    hipLaunchKernelGGL((k_resample<true, true>), g1, dim3(256), 0, s, src_dev, d->resample_tmp.p, (int)w, (int)q.scaled_w, (int)h, (int)q.crop_x, (int)q.out_w, (int)w, sampler, premultiply_alpha, maxv);
    hipLaunchKernelGGL((k_resample<false, true>), g2, dim3(256), 0, s, d->resample_tmp.p, dst_dev, (int)h, (int)q.scaled_h, (int)q.out_w, (int)q.crop_y, (int)q.out_h, (int)q.out_w, sampler, premultiply_alpha, maxv);
  } else {
    hipLaunchKernelGGL((k_resample<true, false>), g1, dim3(256), 0, s, src_dev, d->resample_tmp.p, (int)w, (int)q.scaled_w, (int)h, (int)q.crop_x, (int)q.out_w, (int)w, sampler, premultiply_alpha, maxv);
    hipLaunchKernelGGL((k_resample<false, false>), g2, dim3(256), 0, s, d->resample_tmp.p, dst_dev, (int)h, (int)q.scaled_h, (int)q.out_w, (int)q.crop_y, (int)q.out_h, (int)q.out_w, sampler, premultiply_alpha, maxv);
  }
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { d->set_error("HIP: resample stage failed"); return JXLAMD_ERR_DEVICE; }
  return JXLAMD_OK;
}

}  // extern "C"
