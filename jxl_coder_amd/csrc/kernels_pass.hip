// jxl_coder_amd/csrc/kernels_pass.hip — HIP kernels (gfx950): rANS decode of AC coefficients — k_pass_group[_batch] (one wave per 256x256 group) and
// k_pass_group_simt (one LANE per group, flights).
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_common.h"

namespace jxlamd {

__global__ void __launch_bounds__(64) k_pass_group(DevBuffers B) {
  __shared__ DevPassScratch S;
  if (frame_failed(B)) return;
  pass_group_body(B, S, (int)blockIdx.x, (int)threadIdx.x, 64, SyncBlock());
}

__global__ void __launch_bounds__(64) k_pass_group_batch(const DevBuffers *Bs, const int *map) {
  __shared__ DevPassScratch S;
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  if (frame_failed(Bs[f])) return;
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
  if (frame_failed(B)) return;
  uint32_t e = pass_group_lane(B, freq_ctx, nnz_ctx, B.pass_nz + (size_t)g * 3072, g);
  if (e) atomicOr(B.err, e | kErrStagePass);
}

// lane-per-group with the stall-free bit supply (dev_vardct.h pass_group_lane2): one wavefront = up to 64 groups of ONE frame.
// wmap: {frame, first group, number of groups <= 64} per workgroup
__global__ void __launch_bounds__(64) k_pass_group_simt2(const DevBuffers *__restrict__ Bs, const int *__restrict__ wmap, int ctx_in_lds) {
  // dynamic LDS: the context-map slices (the struct's last member, 31 KB) are only allocated when they are kept in LDS
  extern __shared__ __attribute__((aligned(16))) uint8_t simt2_smem[];
  SimtPassLds &L = *(SimtPassLds *)simt2_smem;
  const int lane = (int)threadIdx.x;
  const int f = wmap[3 * blockIdx.x], g0 = wmap[3 * blockIdx.x + 1], n = wmap[3 * blockIdx.x + 2];
  const DevBuffers &B = Bs[f];
  if (frame_failed(B)) return;
  const DevFrame &F = frame_of(B);
  __builtin_amdgcn_s_setprio(2);
  uint32_t e = 0;
  for (int pass = 0; pass < F.num_passes; pass++) {
    __syncthreads();
    simt2_stage(B, L, pass, lane, 64, ctx_in_lds != 0);
    __syncthreads();
    if (lane < n && !e) e = pass_group_lane2(B, L, pass, B.pass_nz + (size_t)(g0 + lane) * 3072, g0 + lane, lane);
  }
  if (e) atomicOr(B.err, e | kErrStagePass);
}
void launch_pass_groups_simt2(const DevBuffers *Bs, const int *wmap, int nwg, hipStream_t s) {
  static const int ctx_in_lds = getenv("JXLAMD_SIMT2_CTX_LDS") ? atoi(getenv("JXLAMD_SIMT2_CTX_LDS")) : 0;
  const size_t lds = offsetof(SimtPassLds, ctx_map) + (ctx_in_lds ? sizeof(((SimtPassLds *)nullptr)->ctx_map) : 16);
  hipLaunchKernelGGL(k_pass_group_simt2, dim3(nwg), dim3(64), lds, s, Bs, wmap, ctx_in_lds);
}

// One workgroup per frame (per 256 groups of a larger frame): the pass's HF code sits in LDS (DevFrame::hf_lds), every lane decodes one
// 256x256 group.  wmap: {frame, first group, number of groups} per workgroup.
constexpr int kHfLdsBytes = 150 * 1024;
__global__ void __launch_bounds__(256) k_pass_frame(const DevBuffers *__restrict__ Bs, const int *__restrict__ wmap) {
  __shared__ __attribute__((aligned(16))) uint8_t img[kHfLdsBytes];
  __shared__ uint16_t freq_ctx[64], nnz_ctx[64];
  const int tid = (int)threadIdx.x;
  const int f = wmap[3 * blockIdx.x], g0 = wmap[3 * blockIdx.x + 1], n = wmap[3 * blockIdx.x + 2];
  const DevBuffers &B = Bs[f];
  if (frame_failed(B)) return;
  const DevFrame &F = frame_of(B);
  if (tid < 64) { freq_ctx[tid] = kCoeffFreqContext[tid]; nnz_ctx[tid] = kCoeffNumNonzeroContext[tid]; }
  __builtin_amdgcn_s_setprio(2);
  uint32_t e = 0;
  for (int pass = 0; pass < F.num_passes; pass++) {
    const uint32_t bytes = F.hf_lds[pass].bytes;
    const uint4 *src = (const uint4 *)(B.tables + F.hf_lds[pass].off);
    __syncthreads();                                   // the previous pass's lanes are done with the image
    for (uint32_t i = (uint32_t)tid; i < (bytes + 15) / 16; i += 256) ((uint4 *)img)[i] = src[i];
    __syncthreads();
    if (tid < n && !e) e = pass_group_lane_lds(B, img, pass, freq_ctx, nnz_ctx, B.pass_nz + (size_t)(g0 + tid) * 3072, g0 + tid);
  }
  if (e) atomicOr(B.err, e | kErrStagePass);
}
void launch_pass_frames(const DevBuffers *Bs, const int *wmap, int nwg, hipStream_t s) { hipLaunchKernelGGL(k_pass_frame, dim3(nwg), dim3(256), 0, s, Bs, wmap); }
int pass_frame_lds_capacity() { return kHfLdsBytes; }

void launch_pass_groups(const DevBuffers &B, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group, dim3(n), dim3(64), 0, s, B); }
void launch_pass_groups_batch(const DevBuffers *Bs, const int *map, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group_batch, dim3(n), dim3(64), 0, s, Bs, map); }
void launch_pass_groups_simt(const DevBuffers *Bs, const int *map, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group_simt, dim3((n + 63) / 64), dim3(64), 0, s, Bs, map, n); }

}  // namespace jxlamd
