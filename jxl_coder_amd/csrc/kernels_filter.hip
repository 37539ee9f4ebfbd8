// jxl_coder_amd/csrc/kernels_filter.hip — HIP kernels (gfx950): Gaborish / EPF iterations, the last one fused with the XYB -> RGB -> RGBA8/16 writer.
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_common.h"

namespace jxlamd {

// ---- batched data-parallel stages: blockIdx.z = frame of the flight (per-frame dims come from its DevFrame)
// Which plane set holds the image before filter stage `stage` (0 gab, 1 epf0, 2 epf1, 3 epf2, 4 write), and does the
// frame run that stage at all?
__device__ __forceinline__ bool stage_runs(const DevFrame &F, int stage) {
  return stage == 0 ? F.gab != 0 : stage == 1 ? F.epf_iters >= 3 : stage == 2 ? F.epf_iters >= 1 : stage == 3 ? F.epf_iters >= 2 : true;
}
__device__ __forceinline__ bool stage_src_is_a(const DevFrame &F, int stage) {
  int n = 0;
  for (int s = 0; s < stage; s++) n += stage_runs(F, s) ? 1 : 0;
  return (n & 1) == 0;
}
// One instantiation per stage (0 = Gaborish, 1..3 = EPF iterations 0..2, 4 = XYB -> RGBA writer): the writer needs 14
// VGPRs and Gaborish 48, so they must not inherit the unrolled EPF's register footprint — these kernels share the
// SIMDs with resident entropy-decode waves, and their occupancy is what is left of the register file.
// The last filter stage of a frame (EPF iteration 1 or 2, or Gaborish when there is no EPF) is fused with the writer: its
// XYB value goes straight through the colour transform into the RGBA buffer (no plane store + reload, no writer launch).
__device__ __forceinline__ int last_filter_stage(const DevFrame &F) { return F.epf_iters >= 2 ? 3 : F.epf_iters == 1 ? 2 : F.gab ? 0 : -1; }
// Band decode: rows of context the stages AFTER `stage` still need around the band (EPF iteration 0 reads +-3 rows, 1: +-2, 2: +-1),
// i.e. how far beyond the band this stage has to produce output.  0 for the frame's last stage.
__device__ __forceinline__ int stage_halo_after(const DevFrame &F, int stage) {
  return (stage < 1 && F.epf_iters >= 3 ? 3 : 0) + (stage < 2 && F.epf_iters >= 1 ? 2 : 0) + (stage < 3 && F.epf_iters >= 2 ? 1 : 0);
}
template <int STAGE>
__global__ void __launch_bounds__(256) k_filter_b(const DevBuffers *Bs, const uint8_t *stat) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || !stage_runs(F, STAGE) || frame_failed(B)) return;
  const int last = last_filter_stage(F);
  if (STAGE == 4 && last >= 0) return;                       // the writer was fused into stage `last`
  const int halo = stage_halo_after(F, STAGE);
  const int y_begin = F.band_py0 - halo > 0 ? F.band_py0 - halo : 0, y_end = F.band_py1 + halo < F.height ? F.band_py1 + halo : F.height;
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = y_begin + (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= y_end) return;
  const bool a = stage_src_is_a(F, STAGE);
  float *src[3], *dst[3];
  for (int c = 0; c < 3; c++) { src[c] = a ? B.plane_a[c] : B.plane_b[c]; dst[c] = a ? B.plane_b[c] : B.plane_a[c]; }
  if (STAGE == 4) { xyb_write_pixel(B, stat, *(const DevStatic *)stat, src, B.out_bits, x, y); return; }
  float v[3];
  if (STAGE == 0) gab_value(F, src, x, y, v);
  else epf_value_p<(STAGE >= 1 && STAGE <= 3 ? STAGE - 1 : 0)>(B, F, src, x, y, v);
  if (STAGE == last) {
    // keep the filter's last multiply and the writer's first add apart (no FMA contraction across the fusion seam): the fused
    // path must give the very pixels of the stage-by-stage path (single decodes, tests/test_gpu_parity.py batch == single)
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    xyb_write_value(B, stat, *(const DevStatic *)stat, v[0], v[1], v[2], B.out_bits, x, y);
  }
  else for (int c = 0; c < 3; c++) dst[c][(size_t)y * (size_t)F.pw + (size_t)x] = v[c];
}

// ---- Fused Gaborish + EPF iterations + writer over LDS tiles (one launch per frame set instead of one per stage).
// A workgroup owns a 64 x 16 output tile.  It loads the tile plus the H-pixel halo the enabled stages need (H = 1 + 3 + 2 + 1 at
// most) from the reconstructed planes ONCE — coordinates outside the image are mirrored at load time — and runs the stages between two
// LDS buffers, each over a margin that shrinks by the stage's radius; the last stage feeds the XYB -> RGBA writer directly.  Values
// the stage-by-stage path would read at mirrored coordinates are computed there too (position p outside the image holds the stage
// output AT mirror(p)), so the result is what the per-stage kernels produce.  HBM traffic per 4K frame: 3 planes read once (+ halo
// overlap) and the RGBA written, instead of three plane sets per stage.
struct LdsTile {
  const float *base; int stride, plane, ox, oy;             // element (c, y, x) at base[c * plane + (y - oy) * stride + (x - ox)]
  __device__ __forceinline__ float operator()(int c, int y, int x) const { return base[c * plane + (y - oy) * stride + (x - ox)]; }
};
constexpr int kTileW = 64, kTileH = 16;
__global__ void __launch_bounds__(256) k_filter_fused(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat) {
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || frame_failed(B)) return;
  const int tx0 = (int)blockIdx.x * kTileW, ty0 = F.band_py0 + (int)blockIdx.y * kTileH;
  if (tx0 >= F.width || ty0 >= F.band_py1) return;
  const int tid = (int)threadIdx.x;
  const bool run[4] = {F.gab != 0, F.epf_iters >= 3, F.epf_iters >= 1, F.epf_iters >= 2};
  const int rad[4] = {1, 3, 2, 1};
  int H = 0, last = -1;
  for (int s = 0; s < 4; s++) if (run[s]) { H += rad[s]; last = s; }
  const int RW = kTileW + 2 * H, RH = kTileH + 2 * H, plane = RW * RH;
  float *bufA = tile_lds, *bufB = tile_lds + 3 * plane;
  // load: tile + halo, mirrored at the image edges
  for (int i = tid; i < plane; i += 256) {
    const int ly = i / RW, lx = i - ly * RW;
    const size_t so = (size_t)mirror(ty0 - H + ly, F.height) * (size_t)F.pw + (size_t)mirror(tx0 - H + lx, F.width);
    for (int c = 0; c < 3; c++) bufA[c * plane + i] = B.plane_a[c][so];
  }
  __syncthreads();
  const DevStatic &ST = *(const DevStatic *)stat;
  if (last < 0) {                                           // no loop filter: writer only
    for (int i = tid; i < kTileW * kTileH; i += 256) {
      const int x = tx0 + (i & (kTileW - 1)), y = ty0 + i / kTileW;
      if (x < F.width && y < F.band_py1) xyb_write_value(B, stat, ST, bufA[i], bufA[plane + i], bufA[2 * plane + i], B.out_bits, x, y);
    }
    return;
  }
  int m = H;                                                // margin of valid data around the tile in the current source buffer
  for (int s = 0; s < 4; s++) {
    if (!run[s]) continue;
    const int mo = m - rad[s];                              // margin this stage produces
    const LdsTile src{bufA, RW, plane, tx0 - H, ty0 - H};
    const int ow = kTileW + 2 * mo, oh = kTileH + 2 * mo;
    for (int i = tid; i < ow * oh; i += 256) {
      const int py = i / ow, px = i - py * ow;
      const int gx = tx0 - mo + px, gy = ty0 - mo + py;     // frame coordinates of the output position (may be outside the image)
      // outside the image only the `mo` pixels the later stages' taps can reach are ever read (their mirror images lie inside the loaded region)
      if (gx < -mo || gx > F.width - 1 + mo || gy < -mo || gy > F.height - 1 + mo) continue;
      const int ex = mirror(gx, F.width), ey = mirror(gy, F.height);   // where the stage-by-stage path evaluates it
      float v[3];
      if (s == 0) gab_value_acc(F, src, ex, ey, v);
      else if (s == 1) epf_value_acc<0>(B, F, src, ex, ey, v);
      else if (s == 2) epf_value_acc<1>(B, F, src, ex, ey, v);
      else epf_value_acc<2>(B, F, src, ex, ey, v);
      if (s == last) {
        if (gx < F.width && gy < F.band_py1) {
          asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));      // keep the filter's last multiply and the writer's first add apart (no FMA across the seam)
          xyb_write_value(B, stat, ST, v[0], v[1], v[2], B.out_bits, gx, gy);
        }
      } else {
        const int o = (gy - (ty0 - H)) * RW + (gx - (tx0 - H));
        for (int c = 0; c < 3; c++) bufB[c * plane + o] = v[c];
      }
    }
    __syncthreads();
    float *t = bufA; bufA = bufB; bufB = t;
    m = mo;
  }
}

void launch_filters_fused(const DevBuffers *Bs, const uint8_t *stat, int nframes, int max_w, int max_h, int stage_mask, hipStream_t s) {
  const int H = (stage_mask & 1 ? 1 : 0) + (stage_mask & 2 ? 3 : 0) + (stage_mask & 4 ? 2 : 0) + (stage_mask & 8 ? 1 : 0);
  const size_t lds = (size_t)2 * 3 * (kTileW + 2 * H) * (kTileH + 2 * H) * sizeof(float);
  hipLaunchKernelGGL(k_filter_fused, dim3((max_w + kTileW - 1) / kTileW, (max_h + kTileH - 1) / kTileH, nframes), dim3(256), lds, s, Bs, stat);
}

void launch_filters_batch(const DevBuffers *Bs, const uint8_t *stat, int nframes, int max_w, int max_h, int stage_mask, hipStream_t s) {
  // k_filter_fused halves the filters' HBM traffic but runs 1.8x longer than the per-stage kernels (LDS-operation bound, 4 waves per SIMD;
  // profiles/r02_*): off by default
  static const int fused = getenv("JXLAMD_FUSED_FILTERS") ? atoi(getenv("JXLAMD_FUSED_FILTERS")) : 0;
  if (fused) { launch_filters_fused(Bs, stat, nframes, max_w, max_h, stage_mask, s); return; }
  dim3 grid((max_w + 63) / 64, (max_h + 3) / 4, nframes);
  if (stage_mask & 1) hipLaunchKernelGGL(k_filter_b<0>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 2) hipLaunchKernelGGL(k_filter_b<1>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 4) hipLaunchKernelGGL(k_filter_b<2>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 8) hipLaunchKernelGGL(k_filter_b<3>, grid, dim3(256), 0, s, Bs, stat);
  if (stage_mask & 16) hipLaunchKernelGGL(k_filter_b<4>, grid, dim3(256), 0, s, Bs, stat);
}

}  // namespace jxlamd
