// jxl_coder_amd/csrc/kernels.hip — hand-written HIP kernels (gfx950) of the JPEG XL decode pipeline.
//   k_lf_group   : Modular (MA-tree + rANS) decode of LF coefficients and HF metadata, one wave per LF group
//   k_pass_group : rANS decode of AC coefficients, one wave per 256x256 group
//   k_lf_smooth  : adaptive LF smoothing, one thread per 8x8 cell
//   k_recon_*    : dequant + chroma-from-luma + LLF + inverse var-size DCT, LDS-staged per varblock
//   k_filter_b<stage> : Gaborish / EPF iterations (one thread per pixel), the last one fused with the XYB -> RGB -> RGBA8/16 writer
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include <stdlib.h>
#include <algorithm>
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
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(128))) k_lf_group_batch(const DevBuffers *__restrict__ Bs, const DevAux *__restrict__ As, const int *__restrict__ map) {
  __shared__ DevModScratch S;
  // Issue priority: this wave walks one long dependency chain (one instruction in flight at a time) next to data-parallel
  // waves with many ready instructions; without priority it waits for an issue slot each time it becomes ready, which
  // stretches the 240 ms it holds its LDS / register footprint.  It uses < 1/4 of the SIMD's issue slots at full speed.
  __builtin_amdgcn_s_setprio(3);
  // readfirstlane: the frame index is wave-uniform, so the DevBuffers fields come through scalar loads into SGPRs
  // (as with the by-value kernel argument of k_lf_group) instead of occupying ~60 VGPRs
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  lf_group_body(Bs[f], As[f], S, g, (int)threadIdx.x, 64, SyncBlock());
}
__global__ void __launch_bounds__(64) k_pass_group_batch(const DevBuffers *Bs, const int *map) {
  __shared__ DevPassScratch S;
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  pass_group_body(Bs[f], S, g, (int)threadIdx.x, 64, SyncBlock());
}

// lane-per-stream PassGroup kernel (batch mode): lane l of block b decodes group map[b*64+l]
__global__ void __launch_bounds__(64) k_pass_group_simt(const DevBuffers *Bs, const int *map, int total) {
  __shared__ uint16_t freq_ctx[64], nnz_ctx[64];
  __builtin_amdgcn_s_setprio(2);           // latency-bound like the LF waves (see k_lf_group_batch), but 64 streams per wave
  freq_ctx[threadIdx.x] = kCoeffFreqContext[threadIdx.x]; nnz_ctx[threadIdx.x] = kCoeffNumNonzeroContext[threadIdx.x];
  __syncthreads();
  const int i = (int)(blockIdx.x * 64 + threadIdx.x);
  if (i >= total) return;
  const int f = map[2 * i], g = map[2 * i + 1];
  const DevBuffers &B = Bs[f];
  uint32_t e = pass_group_lane(B, freq_ctx, nnz_ctx, B.pass_nz + (size_t)g * 3072, g);
  if (e) atomicOr(B.err, e | kErrStagePass);
}

__global__ void __launch_bounds__(256) k_lf_smooth(DevBuffers B, int xb, int yb) {
  int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= xb * yb) return;
  lf_smooth_cell(B, i % xb, i / xb);
}

// Varblock reconstruction in three size classes so that the LDS budget (and with it the occupancy) fits the block:
//   small  (<= 256 coefficients):  one wave per 8x8 cell, 4 KiB LDS
//   medium (512, 1024):            256 threads, 28 KiB LDS, walks the list of such blocks recorded at placement
//   large  (2048, 4096):           256 threads, 32 KiB LDS (one channel at a time), walks its list
// medium / large varblocks: the placement step recorded their cells; a fixed-size grid walks the list.
//
// DCT32x32 (the dominant transform of smooth 4K content) takes a register-blocked path: the 32-point cosine table sits in
// LDS for the lifetime of the workgroup, the three channels go through each 1-D pass together, and every work-item owns
// a 4 (frequencies / rows) x 3 (channels) tile of outputs for one column x, so that one b128 LDS broadcast feeds 4 FMAs:
//   pass 1: T[c][v][x] = sum_u S[c][u][v] * cc[u][x]      pass 2: out[c][y][x] = sum_v T[c][v][x] * cc[v][y]
// 12 FMAs per 4 LDS reads instead of 1 FMA per (LDS + global) read of the generic path.
constexpr int kStrategyDct32 = 5;
template <int NMAX>
struct ReconLds {                       // medium: S[3][1024] T[3][1024] CC (28 KiB); large: one channel at a time, S[4096] T[4096] (32 KiB)
  float S[NMAX > 1024 ? NMAX : 3 * NMAX];
  float T[NMAX > 1024 ? NMAX : 3 * 1024];
  float CC[NMAX > 1024 ? 4 : 1024];
};

__device__ __forceinline__ void recon_dct32_passes(const DevBuffers &B, const float *S, float *T, const float *CC, int bx, int by, int tid) {
  const DevFrame &F = frame_of(B);
  const int x = tid & 31, q0 = (tid >> 5) * 4;
  float acc[3][4];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[c][j] = 0.0f;
#pragma unroll 4
  for (int u = 0; u < 32; u++) {
    const float ccv = CC[u * 32 + x];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float4 s4 = *(const float4 *)&S[c * 1024 + u * 32 + q0];
      acc[c][0] += s4.x * ccv; acc[c][1] += s4.y * ccv; acc[c][2] += s4.z * ccv; acc[c][3] += s4.w * ccv;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int j = 0; j < 4; j++) { T[c * 1024 + (q0 + j) * 32 + x] = acc[c][j]; acc[c][j] = 0.0f; }
  __syncthreads();
#pragma unroll 4
  for (int v = 0; v < 32; v++) {
    const float4 c4 = *(const float4 *)&CC[v * 32 + q0];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float tv = T[c * 1024 + v * 32 + x];
      acc[c][0] += tv * c4.x; acc[c][1] += tv * c4.y; acc[c][2] += tv * c4.z; acc[c][3] += tv * c4.w;
    }
  }
  const size_t po = (size_t)(by * 8 + q0) * (size_t)F.pw + (size_t)(bx * 8 + x);
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int j = 0; j < 4; j++) B.plane_a[c][po + (size_t)j * (size_t)F.pw] = acc[c][j];
}

template <int NMIN, int NMAX>
__device__ __forceinline__ void recon_list_walk(const DevBuffers &B, const uint8_t *stat, int cls, int xb) {
  __shared__ __attribute__((aligned(16))) ReconLds<NMAX> L;
  const int tid = (int)threadIdx.x;
  const DevStatic &ST = *(const DevStatic *)stat;
  const uint32_t count = B.big_count[cls];
  if (blockIdx.x >= count) return;
  if (NMAX == 1024) {
    const float *cc = st_f(stat, ST.cos_off[5]);
    for (int i = tid; i < 1024; i += 256) L.CC[i] = cc[i];
  }
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const int cell = (int)B.big_list[cls][i];
    const int bx = cell % xb, by = cell / xb;
    if (NMAX == 1024 && B.strategy[cell] == kStrategyDct32) {
      __syncthreads();                               // previous item's pass 2 has finished reading T; CC is in place
      recon_phaseA(B, stat, ST, L.S, 1024, bx, by, tid, 256);
      __syncthreads();
      recon_phaseB(B, stat, ST, L.S, 1024, bx, by, tid, 256);
      __syncthreads();
      recon_dct32_passes(B, L.S, L.T, L.CC, bx, by, tid);
    } else {
      __syncthreads();
      recon_block_body<false, (NMAX > 1024)>(B, stat, L.S, L.T, bx, by, NMIN, NMAX, tid, 256, SyncBlock());
    }
  }
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

// ---- Modular-encoded frames of a flight: the same bodies, (frame, group) through a map / blockIdx.z = frame
__global__ void __launch_bounds__(64) k_mod_global_b(const DevBuffers *Bs) {
  __shared__ DevModScratch S;
  const DevFrame &F = frame_of(Bs[blockIdx.x]);
  if (!F.is_modular && !F.has_ec) return;          // VarDCT frame without extra channels: no Modular image
  mod_global_body(Bs[blockIdx.x], S, (int)threadIdx.x, 64, SyncBlock());
}
__global__ void __launch_bounds__(64) k_mod_group_b(const DevBuffers *Bs, const int *map) {
  __shared__ DevModScratch S;
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  const DevBuffers &B = Bs[f];
  const DevFrame &F = frame_of(B);
  if (F.mod_first_group_ch >= F.mod_nch) return;
  mod_group_body(B, S, g, (int)threadIdx.x, 64, SyncBlock());
}
__global__ void __launch_bounds__(256) k_mod_op_b(const DevBuffers *Bs, int op) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if ((!F.is_modular && !F.has_ec) || op >= F.mod_nops) return;
  const size_t n = (size_t)(F.mod_op_kind[op] == 0 ? F.mod_op_y[op] : F.mod_op_c[op]);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) mod_op_element(B, F, op, i);
}
__global__ void __launch_bounds__(256) k_mod_write_b(const DevBuffers *Bs) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height) return;
  mod_write_pixel(B, B.out_bits, x, y);
}
// extra channels (alpha) of the VarDCT frames of a flight: GlobalModular parts before the LF stage, the per-group streams and the
// inverse transforms after the PassGroup stage of each sub-flight
void launch_ec_global_batch(const DevBuffers *Bs, int nframes, hipStream_t s) { hipLaunchKernelGGL(k_mod_global_b, dim3(nframes), dim3(64), 0, s, Bs); }
void launch_ec_groups_batch(const DevBuffers *Bs, const int *group_map, int nframes, int ngroups, int max_ops, hipStream_t s) {
  if (ngroups > 0) hipLaunchKernelGGL(k_mod_group_b, dim3(ngroups), dim3(64), 0, s, Bs, group_map);
  for (int o = 0; o < max_ops; o++) hipLaunchKernelGGL(k_mod_op_b, dim3(1024, 1, nframes), dim3(256), 0, s, Bs, o);
}
void launch_modular_batch(const DevBuffers *Bs, const int *group_map, int nframes, int ngroups, int max_ops, int max_w, int max_h, hipStream_t s) {
  hipLaunchKernelGGL(k_mod_global_b, dim3(nframes), dim3(64), 0, s, Bs);
  if (ngroups > 0) hipLaunchKernelGGL(k_mod_group_b, dim3(ngroups), dim3(64), 0, s, Bs, group_map);
  for (int o = 0; o < max_ops; o++) hipLaunchKernelGGL(k_mod_op_b, dim3(1024, 1, nframes), dim3(256), 0, s, Bs, o);
  hipLaunchKernelGGL(k_mod_write_b, dim3((max_w + 63) / 64, (max_h + 3) / 4, nframes), dim3(256), 0, s, Bs);
}

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
__global__ void __launch_bounds__(256) k_lf_smooth_b(const DevBuffers *Bs) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular) return;
  int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= F.xb * F.yb) return;
  lf_smooth_cell(B, i % F.xb, i / F.xb);
}
__global__ void __launch_bounds__(64) k_recon_small_b(const DevBuffers *Bs, const uint8_t *stat) {
  __shared__ float S[3 * 256];
  __shared__ float T[256];
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular) return;
  const uint32_t count = B.big_count[2];
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const int cell = (int)B.big_list[2][i];
    __syncthreads();
    recon_block_body<true>(B, stat, S, T, cell % F.xb, cell / F.xb, 0, 256, (int)threadIdx.x, 64, SyncBlock());
  }
}
template <int NMIN, int NMAX>
__global__ void __launch_bounds__(256) k_recon_list_b(const DevBuffers *Bs, const uint8_t *stat, int cls) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular) return;
  recon_list_walk<NMIN, NMAX>(B, stat, cls, F.xb);
}
// One instantiation per stage (0 = Gaborish, 1..3 = EPF iterations 0..2, 4 = XYB -> RGBA writer): the writer needs 14
// VGPRs and Gaborish 48, so they must not inherit the unrolled EPF's register footprint — these kernels share the
// SIMDs with resident entropy-decode waves, and their occupancy is what is left of the register file.
// The last filter stage of a frame (EPF iteration 1 or 2, or Gaborish when there is no EPF) is fused with the writer: its
// XYB value goes straight through the colour transform into the RGBA buffer (no plane store + reload, no writer launch).
__device__ __forceinline__ int last_filter_stage(const DevFrame &F) { return F.epf_iters >= 2 ? 3 : F.epf_iters == 1 ? 2 : F.gab ? 0 : -1; }
template <int STAGE>
__global__ void __launch_bounds__(256) k_filter_b(const DevBuffers *Bs, const uint8_t *stat) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || !stage_runs(F, STAGE)) return;
  const int last = last_filter_stage(F);
  if (STAGE == 4 && last >= 0) return;                       // the writer was fused into stage `last`
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height) return;
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

void launch_rest_batch(const DevBuffers *Bs, const uint8_t *stat, int nframes, int max_cells, int max_w, int max_h, int stage_mask, bool expect_large,
                       int parts, hipStream_t s) {
  if (parts & 1) {
    // a single decode has the chip to itself: more, shorter workgroups for the list walkers
    const int gm = nframes == 1 ? 2048 : 256, gl = nframes == 1 ? 512 : 64;
    hipLaunchKernelGGL(k_recon_small_b, dim3(std::min(max_cells, nframes == 1 ? 8192 : 1024), 1, nframes), dim3(64), 0, s, Bs, stat);
    hipLaunchKernelGGL((k_recon_list_b<257, 1024>), dim3(gm, 1, nframes), dim3(256), 0, s, Bs, stat, 0);
    // 2048/4096-coefficient blocks: any grid is correct (the workgroups stride over the list); when the previous flight had
    // none, one workgroup per frame keeps the (then empty) launch from queueing 64 x 32 KB of LDS requests behind resident LF waves
    hipLaunchKernelGGL((k_recon_list_b<1025, 4096>), dim3(expect_large ? gl : 1, 1, nframes), dim3(256), 0, s, Bs, stat, 1);
  }
  if (parts & 2) {
    dim3 grid((max_w + 63) / 64, (max_h + 3) / 4, nframes);
    if (stage_mask & 1) hipLaunchKernelGGL(k_filter_b<0>, grid, dim3(256), 0, s, Bs, stat);
    if (stage_mask & 2) hipLaunchKernelGGL(k_filter_b<1>, grid, dim3(256), 0, s, Bs, stat);
    if (stage_mask & 4) hipLaunchKernelGGL(k_filter_b<2>, grid, dim3(256), 0, s, Bs, stat);
    if (stage_mask & 8) hipLaunchKernelGGL(k_filter_b<3>, grid, dim3(256), 0, s, Bs, stat);
    if (stage_mask & 16) hipLaunchKernelGGL(k_filter_b<4>, grid, dim3(256), 0, s, Bs, stat);
  }
}
// one launch clears what hipMemsetAsync cleared per frame: the placement map and the flags / counters / LF bookkeeping block
__global__ void __launch_bounds__(256) k_clear_b(const DevBuffers *Bs) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular) return;
  if (blockIdx.x == 0) {
    uint32_t *m = B.err;                                   // misc block: 4096 bytes of flags + 72 bytes per LF group
    const int words = (4096 + F.num_lf_groups * 72) / 4;
    for (int i = (int)threadIdx.x; i < words; i += 256) m[i] = 0;
    return;
  }
  const int ncell = F.xb * F.yb;
  const int i = (int)((blockIdx.x - 1) * 256 + threadIdx.x) * 16;
  if (i + 16 <= ncell) { uint4 z = {0, 0, 0, 0}; *(uint4 *)(B.first + i) = z; }       // cell arrays are 256-byte aligned allocations
  else for (int k = i; k < ncell; k++) B.first[k] = 0;
}
void launch_clear_batch(const DevBuffers *Bs, int nframes, int max_cells, hipStream_t s) {
  hipLaunchKernelGGL(k_clear_b, dim3((max_cells + 4095) / 4096 + 1, 1, nframes), dim3(256), 0, s, Bs);
}
void launch_lf_smooth_batch(const DevBuffers *Bs, int nframes, int max_cells, hipStream_t s) {
  hipLaunchKernelGGL(k_lf_smooth_b, dim3((max_cells + 255) / 256, 1, nframes), dim3(256), 0, s, Bs);
}

void launch_lf_groups(const DevBuffers &B, const DevAux &A, int n, hipStream_t s) { hipLaunchKernelGGL(k_lf_group, dim3(n), dim3(64), 0, s, B, A); }
void launch_pass_groups(const DevBuffers &B, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group, dim3(n), dim3(64), 0, s, B); }
void launch_lf_groups_batch(const DevBuffers *Bs, const DevAux *As, const int *map, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_lf_group_batch, dim3(n), dim3(64), 0, s, Bs, As, map);
}
void launch_pass_groups_batch(const DevBuffers *Bs, const int *map, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group_batch, dim3(n), dim3(64), 0, s, Bs, map); }
void launch_pass_groups_simt(const DevBuffers *Bs, const int *map, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group_simt, dim3((n + 63) / 64), dim3(64), 0, s, Bs, map, n); }
void launch_lf_smooth(const DevBuffers &B, int xb, int yb, hipStream_t s) {
  hipLaunchKernelGGL(k_lf_smooth, dim3((xb * yb + 255) / 256), dim3(256), 0, s, B, xb, yb);
}

}  // namespace jxlamd
