// jxl_coder_amd/csrc/kernels_pass.hip — HIP kernels (gfx950): rANS decode of AC coefficients — k_pass_group[_batch] (one wave per 256x256 group: single
// decodes, small bands) and k_pass_prep + k_pass_flat (one LANE per group, flat state machine: flights and large bands).
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_common.h"

namespace jxlamd {

__global__ void __launch_bounds__(64) k_pass_group(DevBuffers B) {
  __shared__ DevPassScratch S;
  if (frame_failed(B)) return;
  pass_group_body(B, S, (int)blockIdx.x, (int)threadIdx.x, 64, SyncBlock());
}

__global__ void __launch_bounds__(64) k_pass_group_batch(const DevBuffers *__restrict__ Bs, const int *__restrict__ map) {
  __shared__ DevPassScratch S;
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  if (frame_failed(Bs[f])) return;
  pass_group_body(Bs[f], S, g, (int)threadIdx.x, 64, SyncBlock());
}

// ---- flights: lane-per-group, flat state machine (dev_pass_flat.h)
// k_pass_prep: one wave per group turns the placement maps into the group's descriptor list (stream order) and writes coef_off.
// Lane l owns the 16 cells of half a row (y = l / 2): block counts and coefficient totals are prefix-summed across the wave.
__global__ void __launch_bounds__(64) k_pass_prep(const DevBuffers *__restrict__ Bs, const int *__restrict__ map) {
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  const DevBuffers &B = Bs[f];
  if (frame_failed(B)) return;
  const DevFrame &F = frame_of(B);
  const int lane = (int)threadIdx.x;
  const int gx = g % F.xgroups, gy = g / F.xgroups;
  const int bx0 = gx * 32, by0 = gy * 32;
  const int bw = F.xb - bx0 < 32 ? F.xb - bx0 : 32, bh = F.yb - by0 < 32 ? F.yb - by0 : 32;
  const int y = lane >> 1, x0 = (lane & 1) * 16;
  uint32_t fmask = 0, cnt = 0, sz = 0;
  const size_t row = (size_t)(by0 + y) * (size_t)F.xb + (size_t)bx0;
  if (y < bh)
    for (int j = 0; j < 16; j++) {
      const int x = x0 + j;
      if (x < bw && B.first[row + x]) { fmask |= 1u << j; cnt++; const int st = B.strategy[row + x]; sz += (uint32_t)kCoveredX[st] * kCoveredY[st] * 64u; }
    }
  uint32_t pc = cnt, ps = sz;                                  // inclusive prefix sums over the lanes
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t tc = (uint32_t)__shfl_up((int)pc, d, 64), ts = (uint32_t)__shfl_up((int)ps, d, 64);
    if (lane >= d) { pc += tc; ps += ts; }
  }
  const uint32_t total_n = (uint32_t)__shfl((int)pc, 63, 64), total_sz = (uint32_t)__shfl((int)ps, 63, 64);
  uint8_t *base = B.pass_nz + (size_t)g * kPassBlkStride;
  if (total_sz > 65536u) {                                     // a group holds at most 32x32 cells of coefficients (stale / corrupt placement data)
    if (lane == 0) { atomicOr(B.err, kErrBitstream | kErrStagePass); ((uint32_t *)base)[0] = 0; }
    return;
  }
  PassBlk *d = (PassBlk *)(base + 8);
  uint32_t n = pc - cnt, pool = ps - sz;
  for (uint32_t m = fmask; m; m &= m - 1) {
    const int x = x0 + __builtin_ctz(m);
    const size_t o = row + (size_t)x;
    const int st = B.strategy[o];
    PassBlk e;
    e.a = pass_blk_pack(y * 32 + x, st, pass_qf_idx(F, (uint32_t)B.qfm1[o] + 1), B.lf_idx[o]); e.off = pool;
    d[n] = e;
    B.coef_off[o] = pool;
    pool += (uint32_t)kCoveredX[st] * kCoveredY[st] * 64u; n++;
  }
  if (lane == 0) { ((uint32_t *)base)[0] = total_n; ((uint32_t *)base)[1] = 0; }
}
// wmap: {frame, first group, number of groups <= 64 | chained tail groups << 8} per wavefront (entries with 0 groups pad the XCD interleave, see decoder.hip)
template <bool kSparse>
__global__ void __launch_bounds__(64) k_pass_flat(const DevBuffers *__restrict__ Bs, const int *__restrict__ wmap) {
  __shared__ __attribute__((aligned(16))) FlatPassLds L;
  const int lane = (int)threadIdx.x;
  const int f = wmap[3 * blockIdx.x], g0 = wmap[3 * blockIdx.x + 1], nf = wmap[3 * blockIdx.x + 2];
  const int n = nf & 255, nch = (nf >> 8) & 255;      // nch: the wave's last nch lanes take a second group each — the frame's tail groups g0 + n .. (decoder.hip: flat_wave_map)
  if (n <= 0) return;
  const DevBuffers &B = Bs[f];
  if (frame_failed(B)) return;
  const DevFrame &F = frame_of(B);
  __builtin_amdgcn_s_setprio(2);          // latency-bound like the LF waves (see k_lf_group_batch), but 64 streams per wave
  uint32_t e = 0;
  for (int pass = 0; pass < F.num_passes; pass++) {
    __syncthreads();
    flat_stage(B, L, pass, lane, 64);
    __syncthreads();
    const uint32_t ep = pass_group_flat<kSparse>(B, L, pass, (lane < n && !e) ? g0 + lane : -1, lane, (lane >= n - nch && lane < n && !e) ? g0 + n + (lane - (n - nch)) : -1);
    e |= ep;
  }
  if (e) atomicOr(B.err, e | kErrStagePass);
}
void launch_pass_prep(const DevBuffers *__restrict__ Bs, const int *__restrict__ map, int ngroups, hipStream_t s) { hipLaunchKernelGGL(k_pass_prep, dim3(ngroups), dim3(64), 0, s, Bs, map); }
void launch_pass_flat(const DevBuffers *__restrict__ Bs, const int *wmap, int nwg, bool sparse, hipStream_t s) {
  if (sparse) hipLaunchKernelGGL(k_pass_flat<true>, dim3(nwg), dim3(64), 0, s, Bs, wmap);
  else hipLaunchKernelGGL(k_pass_flat<false>, dim3(nwg), dim3(64), 0, s, Bs, wmap);
}

void launch_pass_groups(const DevBuffers &B, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group, dim3(n), dim3(64), 0, s, B); }
void launch_pass_groups_batch(const DevBuffers *__restrict__ Bs, const int *__restrict__ map, int n, hipStream_t s) { hipLaunchKernelGGL(k_pass_group_batch, dim3(n), dim3(64), 0, s, Bs, map); }

}  // namespace jxlamd
