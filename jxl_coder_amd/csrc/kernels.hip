// jxl_coder_amd/csrc/kernels.hip — hand-written HIP kernels (gfx950) of the JPEG XL decode pipeline.
//   k_lf_group   : Modular (MA-tree + rANS) decode of LF coefficients and HF metadata, one wave per LF group
//   k_pass_group : rANS decode of AC coefficients, one wave per 256x256 group
//   k_lf_smooth  : adaptive LF smoothing, one thread per 8x8 cell
//   k_recon_*    : dequant + chroma-from-luma + LLF + inverse var-size DCT, LDS-staged per varblock
//   k_gab/k_epf  : loop filters, one thread per pixel;  k_write : XYB -> RGB -> RGBA8/16
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels.h"

namespace jxlamd {

struct SyncBlock { __device__ void operator()() const { __syncthreads(); } };

__global__ void __launch_bounds__(64) k_lf_group(DevBuffers B, DevAux A) {
  __shared__ DevModScratch S;
  lf_group_body(B, A, S, (int)blockIdx.x, (int)threadIdx.x, 64, SyncBlock());
}

__global__ void __launch_bounds__(64) k_pass_group(DevBuffers B) {
  __shared__ DevPassScratch S;
  pass_group_body(B, S, (int)blockIdx.x, (int)threadIdx.x, 64, SyncBlock());
}

// batch variants: block -> (frame, local group) through a small map; the per-frame DevBuffers live in HBM
__global__ void __launch_bounds__(64) k_lf_group_batch(const DevBuffers *Bs, const DevAux *As, const int *map) {
  __shared__ DevModScratch S;
  const int f = map[2 * blockIdx.x], g = map[2 * blockIdx.x + 1];
  lf_group_body(Bs[f], As[f], S, g, (int)threadIdx.x, 64, SyncBlock());
}
__global__ void __launch_bounds__(64) k_pass_group_batch(const DevBuffers *Bs, const int *map) {
  __shared__ DevPassScratch S;
  const int f = map[2 * blockIdx.x], g = map[2 * blockIdx.x + 1];
  pass_group_body(Bs[f], S, g, (int)threadIdx.x, 64, SyncBlock());
}

__global__ void __launch_bounds__(256) k_lf_smooth(DevBuffers B, int xb, int yb) {
  int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= xb * yb) return;
  lf_smooth_cell(B, i % xb, i / xb);
}

// Varblock reconstruction in three size classes so that the LDS budget (and with it the occupancy) fits the block:
//   small  (<= 256 coefficients):  one wave per 8x8 cell, 4 KiB LDS
//   medium (512, 1024):            256 threads, 16 KiB LDS, scans kScan consecutive cells per workgroup
//   large  (2048, 4096):           256 threads, 64 KiB LDS, scans kScan consecutive cells per workgroup
constexpr int kScan = 16;
__global__ void __launch_bounds__(64) k_recon_small(DevBuffers B, const uint8_t *stat, int xb) {
  __shared__ float S[3 * 256];
  __shared__ float T[256];
  int cell = (int)blockIdx.x;
  recon_block_body(B, stat, S, T, cell % xb, cell / xb, 0, 256, (int)threadIdx.x, 64, SyncBlock());
}
template <int NMIN, int NMAX>
__global__ void __launch_bounds__(256) k_recon_scan(DevBuffers B, const uint8_t *stat, int xb, int ncell) {
  __shared__ float S[3 * NMAX];
  __shared__ float T[NMAX];
  for (int i = 0; i < kScan; i++) {
    const int cell = (int)blockIdx.x * kScan + i;
    if (cell >= ncell) return;
    recon_block_body(B, stat, S, T, cell % xb, cell / xb, NMIN, NMAX, (int)threadIdx.x, 256, SyncBlock());
    __syncthreads();
  }
}

struct Planes { float *p[3]; };

__global__ void __launch_bounds__(256) k_gab(DevBuffers B, Planes src, Planes dst, int w, int h) {
  int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= w || y >= h) return;
  gab_pixel(frame_of(B), src.p, dst.p, x, y);
}
__global__ void __launch_bounds__(256) k_epf(DevBuffers B, Planes src, Planes dst, int pass, int w, int h) {
  int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= w || y >= h) return;
  epf_pixel(B, frame_of(B), src.p, dst.p, pass, x, y);
}
__global__ void __launch_bounds__(256) k_write(DevBuffers B, const uint8_t *stat, Planes src, int out_bits, int w, int h) {
  int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= w || y >= h) return;
  xyb_write_pixel(B, stat, *(const DevStatic *)stat, src.p, out_bits, x, y);
}

// ---- Modular-encoded frames
__global__ void __launch_bounds__(64) k_mod_global(DevBuffers B) {
  __shared__ DevModScratch S;
  mod_global_body(B, S, (int)threadIdx.x, 64, SyncBlock());
}
__global__ void __launch_bounds__(64) k_mod_group(DevBuffers B) {
  __shared__ DevModScratch S;
  mod_group_body(B, S, (int)blockIdx.x, (int)threadIdx.x, 64, SyncBlock());
}
__global__ void __launch_bounds__(256) k_mod_op(DevBuffers B, int op, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) mod_op_element(B, frame_of(B), op, i);
}
__global__ void __launch_bounds__(256) k_mod_write(DevBuffers B, int out_bits, int w, int h) {
  int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= w || y >= h) return;
  mod_write_pixel(B, out_bits, x, y);
}
void launch_mod_global(const DevBuffers &B, hipStream_t s) { hipLaunchKernelGGL(k_mod_global, dim3(1), dim3(64), 0, s, B); }
void launch_mod_groups(const DevBuffers &B, int n, hipStream_t s) { hipLaunchKernelGGL(k_mod_group, dim3(n), dim3(64), 0, s, B); }
void launch_mod_op(const DevBuffers &B, int op, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_mod_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, op, n); }
void launch_mod_write(const DevBuffers &B, int w, int h, int out_bits, hipStream_t s) {
  hipLaunchKernelGGL(k_mod_write, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, B, out_bits, w, h);
}

void launch_lf_groups(const DevBuffers &B, const DevAux &A, int n, hipStream_t s) { hipLaunchKernelGGL(k_lf_group, dim3(n), dim3(64), 0, s, B, A); }
void launch_pass_groups(const DevBuffers &B, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group, dim3(n), dim3(64), 0, s, B); }
void launch_lf_groups_batch(const DevBuffers *Bs, const DevAux *As, const int *map, int n, hipStream_t s) { hipLaunchKernelGGL(k_lf_group_batch, dim3(n), dim3(64), 0, s, Bs, As, map); }
void launch_pass_groups_batch(const DevBuffers *Bs, const int *map, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group_batch, dim3(n), dim3(64), 0, s, Bs, map); }
void launch_lf_smooth(const DevBuffers &B, int xb, int yb, hipStream_t s) {
  hipLaunchKernelGGL(k_lf_smooth, dim3((xb * yb + 255) / 256), dim3(256), 0, s, B, xb, yb);
}
void launch_recon(const DevBuffers &B, const uint8_t *stat, int xb, int yb, hipStream_t s) {
  hipLaunchKernelGGL(k_recon_small, dim3(xb * yb), dim3(64), 0, s, B, stat, xb);
  hipLaunchKernelGGL((k_recon_scan<257, 1024>), dim3((xb * yb + kScan - 1) / kScan), dim3(256), 0, s, B, stat, xb, xb * yb);
  hipLaunchKernelGGL((k_recon_scan<1025, 4096>), dim3((xb * yb + kScan - 1) / kScan), dim3(256), 0, s, B, stat, xb, xb * yb);
}
static Planes planes_of(const DevBuffers &B, bool a) {
  Planes p;
  for (int c = 0; c < 3; c++) p.p[c] = a ? B.plane_a[c] : B.plane_b[c];
  return p;
}
bool launch_filters(const DevBuffers &B, int w, int h, int gab, int epf_iters, bool src_is_a, hipStream_t s) {
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
  if (gab) { hipLaunchKernelGGL(k_gab, grid, block, 0, s, B, planes_of(B, src_is_a), planes_of(B, !src_is_a), w, h); src_is_a = !src_is_a; }
  if (epf_iters >= 3) { hipLaunchKernelGGL(k_epf, grid, block, 0, s, B, planes_of(B, src_is_a), planes_of(B, !src_is_a), 0, w, h); src_is_a = !src_is_a; }
  if (epf_iters >= 1) { hipLaunchKernelGGL(k_epf, grid, block, 0, s, B, planes_of(B, src_is_a), planes_of(B, !src_is_a), 1, w, h); src_is_a = !src_is_a; }
  if (epf_iters >= 2) { hipLaunchKernelGGL(k_epf, grid, block, 0, s, B, planes_of(B, src_is_a), planes_of(B, !src_is_a), 2, w, h); src_is_a = !src_is_a; }
  return src_is_a;
}
void launch_write(const DevBuffers &B, const uint8_t *stat, int w, int h, int out_bits, bool src_is_a, hipStream_t s) {
  dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
  hipLaunchKernelGGL(k_write, grid, block, 0, s, B, stat, planes_of(B, src_is_a), out_bits, w, h);
}

}  // namespace jxlamd
