// jxl_coder_amd/csrc/kernels_mod.hip — HIP kernels (gfx950): Modular-encoded (lossless) frames and the Modular-coded extra channels
// (alpha) of VarDCT frames: k_mod_global / k_mod_group (MA-tree + rANS stream decode, one wave per stream), k_mod_op (inverse
// RCT / palette), k_mod_write (int -> RGBA), and their flight forms.
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_common.h"

namespace jxlamd {

// ---- Modular-encoded frames
// GlobalModular workgroups in DYNAMIC LDS too (round 6): offsetof(pool) + the table pool the frame's tree and code need + kModMaxCh channel descriptors.  With the
// static 54.9 KB of rounds 1 - 5 a reference frame of a screenshot — one wave, a few milliseconds of work — waited for a CU with 55 KB of free LDS next to the LF
// and PassGroup waves of fifteen other flights; a patch-sprite frame needs 12 KB of pool: 35 KB.
__device__ __forceinline__ DevModScratch &mod_global_smem(int pool_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t mod_smem_g[];
  DevModScratch &S = *(DevModScratch *)mod_smem_g;
  S.ch = (DevChanOut *)(mod_smem_g + offsetof(DevModScratch, pool) + pool_bytes);      // (every lane stores the same value)
  __syncthreads();
  return S;
}
__global__ void __launch_bounds__(64) k_mod_global(DevBuffers B, int pool_bytes) {
  DevModScratch &S = mod_global_smem(pool_bytes);
  mod_global_body(B, S, (int)threadIdx.x, 64, SyncBlock(), pool_bytes);
}
// The group streams' workgroup lives in DYNAMIC LDS like the LF kernels': offsetof(pool) + the table pool of this launch + kModGroupDesc channel descriptors.
// A frame's group streams are as many as its 256 x 256 groups and each keeps its LDS for the ~100 ms it runs: with the full 30 KB pool (54 KB per
// workgroup) a CU holds two of them, with the 16 KB that libjxl's one-shot streams need (context map + the channel's tree in block form; their 128
// clusters' alias tables never fit) four (mod_group_pool_bytes, decoder.hip).
__device__ __forceinline__ DevModScratch &mod_group_smem(int pool_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t mod_smem[];
  DevModScratch &S = *(DevModScratch *)mod_smem;
  S.ch = (DevChanOut *)(mod_smem + offsetof(DevModScratch, pool) + pool_bytes);      // (every lane stores the same value)
  __syncthreads();
  return S;
}
__global__ void __launch_bounds__(64) k_mod_group(DevBuffers B, int pool_bytes) {
  DevModScratch &S = mod_group_smem(pool_bytes);
  mod_group_body(B, S, (int)blockIdx.x, (int)threadIdx.x, 64, SyncBlock(), pool_bytes);
}
// ModularLfGroup streams of a Modular-encoded frame (section 1 + g holds nothing else there)
__device__ __forceinline__ void mod_lfgroup_kernel(const DevBuffers &B, DevModScratch &S, int g) {
  const DevFrame &F = frame_of(B);
  if (threadIdx.x == 0) {
    S.pool_bytes = kModPoolBytes; S.pool_want = nullptr; S.walk_stat = B.err + 2; S.wide_wp = nullptr;
    const DevSection *secs = (const DevSection *)(B.tables + F.sec_off);
    DevBits b;
    bits_init(b, B.codestream, secs[1 + g].off, F.cs_size);
    S.st.b = b;
  }
  __syncthreads();
  const uint32_t e = mod_lfgroup_body<true, true>(B, S, g, (int)threadIdx.x, 64, SyncBlock());
  if (threadIdx.x == 0 && e) *B.err |= e | kErrStageLf;
}
__global__ void __launch_bounds__(64) k_mod_lfgroup(DevBuffers B) {
  __shared__ DevModScratch S;
  __shared__ DevChanOut chbuf[kModMaxCh];
  S.ch = chbuf;                                       // (every lane stores the same value)
  __syncthreads();
  mod_lfgroup_kernel(B, S, (int)blockIdx.x);
}
// Palette with delta entries (libjxl's lossy palette; mod_op_element kind 4 is its serial statement): a delta entry is added to a prediction from the pixel's
// already reconstructed neighbours (W, WW, N, NW, NE, NEE, NN), so the rows depend on each other — but a row only needs the one above it three pixels ahead.
// One wave per colour channel, lane r on row 64 b + r, three pixels behind lane r - 1: each lane keeps its last seven outputs in registers and takes the
// neighbours of the rows above from lanes r - 1 / r - 2 by wave shifts; only a band's first rows read the previous band's last rows back from memory.
// 1.2 s -> milliseconds for a 1080p screenshot; the same integers as the serial form (64-bit sums, predict_plain).
__device__ __forceinline__ void delta_palette_wave(const DevBuffers &B, const DevFrame &F, int op, int c, int lane) {
  const int32_t *idx = mod_plane(B, F, F.mod_op_a[op]);
  const int32_t *pal = mod_plane(B, F, F.mod_op_b[op]);
  int32_t *out = mod_plane(B, F, F.mod_op_d[op] + c);
  const int psize = F.mod_op_x[op], bd = F.mod_op_y[op] & 0xff, bit_depth = bd < 24 ? bd : 24;
  const int nb_deltas = F.mod_op_f[op], d_pred = F.mod_op_g[op], w = F.mod_op_h[op], h = (int)((uint32_t)F.mod_op_y[op] >> 8);
  for (int y0 = 0; y0 < h; y0 += 64) {
    const int y = y0 + lane;
    const bool act = y < h;
    const int rows = h - y0 < 64 ? h - y0 : 64;
    int32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0;      // this lane's outputs at x - 1 .. x - 7 (zeros outside its row)
    const int32_t *irow = idx + (size_t)(act ? y : 0) * (size_t)w;
    int32_t *orow = out + (size_t)(act ? y : 0) * (size_t)w;
    const int32_t *mN = y > 0 ? orow - w : nullptr, *mNN = y > 1 ? orow - 2 * w : nullptr;      // the rows above in memory (used by lane 0 / lanes 0 and 1 only)
    const int steps = w + 3 * (rows - 1);
    int xn = -3 * lane;
    int32_t inext = (act && xn >= 0 && xn < w) ? irow[xn] : 0;
    for (int t = 0; t < steps; t++) {
      const int x = t - 3 * lane;
      const int32_t index = inext;
      { const int x1 = x + 1; inext = (act && x1 >= 0 && x1 < w) ? irow[x1] : 0; }      // the next step's index, requested a step ahead
      // lane r - 1 stands at x + 3: its h0 .. h3 are the row above at x + 2, x + 1, x, x - 1; lane r - 2 at x + 6: its h5 is two rows up at x
      const int32_t sNEE = __shfl_up(h0, 1, 64), sNE = __shfl_up(h1, 1, 64), sN = __shfl_up(h2, 1, 64), sNW = __shfl_up(h3, 1, 64), sNN = __shfl_up(h5, 2, 64);
      int32_t v = 0;
      if (act && x >= 0 && x < w) {
        int64_t val = palette_value(pal, psize, index, c, bit_depth);
        if (index < nb_deltas) {
          const bool hN = y > 0, hNN = y > 1;
          const int64_t rNx = hN ? (lane >= 1 ? sN : mN[x]) : 0;
          const int64_t W = x > 0 ? (int64_t)h0 : (hN ? rNx : 0);
          const int64_t N = hN ? rNx : W;
          const int64_t NW = (x > 0 && hN) ? (int64_t)(lane >= 1 ? sNW : mN[x - 1]) : W;
          const int64_t NE = (x + 1 < w && hN) ? (int64_t)(lane >= 1 ? sNE : mN[x + 1]) : N;
          const int64_t NN = hNN ? (int64_t)(lane >= 2 ? sNN : mNN[x]) : N;
          const int64_t NEE = (x + 2 < w && hN) ? (int64_t)(lane >= 1 ? sNEE : mN[x + 2]) : NE;
          const int64_t WW = x > 1 ? (int64_t)h1 : W;
          val += predict_plain(d_pred, W, N, NW, NE, NN, WW, NEE, 0);
        }
        v = (int32_t)val;
        orow[x] = v;
      }
      h6 = h5; h5 = h4; h4 = h3; h3 = h2; h2 = h1; h1 = h0; h0 = v;      // every step, in or out of the row: the shift offsets above count steps
    }
    __threadfence();                                      // the next band's first rows read this band's last rows from memory
  }
}
__global__ void __launch_bounds__(256) k_mod_op(DevBuffers B, int op, size_t n) {
  const DevFrame &F = frame_of(B);
  if (F.mod_op_kind[op] == 4) {         // one wave per colour channel; the grid may hold fewer workgroups than the palette has channels
    if (threadIdx.x < 64) for (int c = (int)blockIdx.x; c < F.mod_op_c[op]; c += (int)gridDim.x) delta_palette_wave(B, F, op, c, (int)threadIdx.x);
    return;
  }
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) mod_op_element(B, F, op, i);
}
__global__ void __launch_bounds__(256) k_mod_write(DevBuffers B, int out_bits, int w, int h) {
  int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= w || y >= h) return;
  mod_write_pixel(B, out_bits, x, y);
}
static size_t mod_global_lds(const void *kernel, bool *once, int pool_bytes) {
  if (!*once) { (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(offsetof(DevModScratch, pool) + kModPoolBytes + kModMaxCh * sizeof(DevChanOut))); *once = true; }
  return offsetof(DevModScratch, pool) + (size_t)pool_bytes + kModMaxCh * sizeof(DevChanOut);
}
static int mod_pool_clamp(int pool_bytes);
void launch_mod_global(const DevBuffers &B, int pool_bytes, hipStream_t s) {
  static bool once = false;
  pool_bytes = mod_pool_clamp(pool_bytes);
  hipLaunchKernelGGL(k_mod_global, dim3(1), dim3(64), mod_global_lds((const void *)k_mod_global, &once, pool_bytes), s, B, pool_bytes);
}
static size_t mod_group_lds(const void *kernel, bool *once, int pool_bytes) {
  if (!*once) { (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(offsetof(DevModScratch, pool) + kModPoolBytes + kModGroupDesc * sizeof(DevChanOut))); *once = true; }
  return offsetof(DevModScratch, pool) + (size_t)pool_bytes + kModGroupDesc * sizeof(DevChanOut);
}
static int mod_pool_clamp(int pool_bytes) { return pool_bytes < kModPoolMin ? kModPoolMin : pool_bytes > kModPoolBytes ? kModPoolBytes : (pool_bytes + 255) & ~255; }
void launch_mod_groups(const DevBuffers &B, int n, int pool_bytes, hipStream_t s) {
  static bool once = false;
  pool_bytes = mod_pool_clamp(pool_bytes);
  hipLaunchKernelGGL(k_mod_group, dim3(n), dim3(64), mod_group_lds((const void *)k_mod_group, &once, pool_bytes), s, B, pool_bytes);
}
void launch_mod_lfgroups(const DevBuffers &B, int n, hipStream_t s) { hipLaunchKernelGGL(k_mod_lfgroup, dim3(n), dim3(64), 0, s, B); }
void launch_mod_op(const DevBuffers &B, int op, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_mod_op, dim3((unsigned)std::max<size_t>((n + 255) / 256, 4)), dim3(256), 0, s, B, op, n); }      // (>= 4 workgroups: a delta palette takes one per colour channel)
void launch_mod_write(const DevBuffers &B, int w, int h, int out_bits, hipStream_t s) {
  hipLaunchKernelGGL(k_mod_write, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, B, out_bits, w, h);
}

// ---- Modular-encoded frames of a flight: the same bodies, (frame, group) through a map / blockIdx.z = frame
__global__ void __launch_bounds__(64) k_mod_global_b(const DevBuffers *__restrict__ Bs, int pool_bytes) {
  DevModScratch &S = mod_global_smem(pool_bytes);
  const DevFrame &F = frame_of(Bs[blockIdx.x]);
  if (!F.is_modular && !F.has_ec) return;          // VarDCT frame without extra channels: no Modular image
  mod_global_body(Bs[blockIdx.x], S, (int)threadIdx.x, 64, SyncBlock(), pool_bytes);
}
__global__ void __launch_bounds__(64) k_mod_group_b(const DevBuffers *__restrict__ Bs, const int *__restrict__ map, int pool_bytes) {
  DevModScratch &S = mod_group_smem(pool_bytes);
  const int f = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x]), g = __builtin_amdgcn_readfirstlane(map[2 * blockIdx.x + 1]);
  const DevBuffers &B = Bs[f];
  const DevFrame &F = frame_of(B);
  if (F.mod_first_group_ch >= F.mod_nch) return;
  mod_group_body(B, S, g, (int)threadIdx.x, 64, SyncBlock(), pool_bytes);
}
__global__ void __launch_bounds__(64) k_mod_lfgroup_b(const DevBuffers *__restrict__ Bs) {      // grid (max LF groups, frames)
  __shared__ DevModScratch S;
  __shared__ DevChanOut chbuf[kModMaxCh];
  S.ch = chbuf;                                       // (every lane stores the same value)
  __syncthreads();
  const DevBuffers &B = Bs[blockIdx.y];
  const DevFrame &F = frame_of(B);
  if (!F.is_modular || F.mod_lf_nch <= 0 || (int)blockIdx.x >= F.num_lf_groups || frame_failed(B)) return;
  mod_lfgroup_kernel(B, S, (int)blockIdx.x);
}
__global__ void __launch_bounds__(256) k_mod_op_b(const DevBuffers *__restrict__ Bs, int op) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if ((!F.is_modular && !F.has_ec) || op >= F.mod_nops) return;
  if (F.mod_op_kind[op] == 4) { if (threadIdx.x < 64) for (int c = (int)blockIdx.x; c < F.mod_op_c[op]; c += (int)gridDim.x) delta_palette_wave(B, F, op, c, (int)threadIdx.x); return; }      // one wave per colour channel
  const size_t n = (size_t)(F.mod_op_kind[op] == 0 ? F.mod_op_y[op] : F.mod_op_c[op]);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) mod_op_element(B, F, op, i);
}
__global__ void __launch_bounds__(256) k_mod_write_b(const DevBuffers *__restrict__ Bs) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= F.width || y >= F.height) return;
  mod_write_pixel(B, B.out_bits, x, y);
}
// extra channels (alpha) of the VarDCT frames of a flight: GlobalModular parts before the LF stage, the per-group streams and the
// inverse transforms after the PassGroup stage of each sub-flight
static void launch_mod_global_b(const DevBuffers *__restrict__ Bs, int nframes, int pool_bytes, hipStream_t s) {
  static bool once = false;
  pool_bytes = mod_pool_clamp(pool_bytes);
  hipLaunchKernelGGL(k_mod_global_b, dim3(nframes), dim3(64), mod_global_lds((const void *)k_mod_global_b, &once, pool_bytes), s, Bs, pool_bytes);
}
void launch_ec_global_batch(const DevBuffers *__restrict__ Bs, int nframes, int pool_bytes, hipStream_t s) { launch_mod_global_b(Bs, nframes, pool_bytes, s); }
static void launch_mod_group_b(const DevBuffers *__restrict__ Bs, const int *group_map, int ngroups, int pool_bytes, hipStream_t s) {
  static bool once = false;
  pool_bytes = mod_pool_clamp(pool_bytes);
  hipLaunchKernelGGL(k_mod_group_b, dim3(ngroups), dim3(64), mod_group_lds((const void *)k_mod_group_b, &once, pool_bytes), s, Bs, group_map, pool_bytes);
}
void launch_ec_groups_batch(const DevBuffers *__restrict__ Bs, const int *group_map, int nframes, int ngroups, int max_ops, int pool_bytes, hipStream_t s) {
  if (ngroups > 0) launch_mod_group_b(Bs, group_map, ngroups, pool_bytes, s);
  for (int o = 0; o < max_ops; o++) hipLaunchKernelGGL(k_mod_op_b, dim3(1024, 1, nframes), dim3(256), 0, s, Bs, o);
}
void launch_modular_batch(const DevBuffers *__restrict__ Bs, const int *group_map, int nframes, int ngroups, int max_ops, int max_w, int max_h, int pool_bytes, hipStream_t s) {
  launch_mod_global_b(Bs, nframes, pool_bytes, s);
  const int max_lfg = ((max_w + 1023) / 1024) * ((max_h + 1023) / 1024);       // LF groups are 8 x group_dim pixels wide (>= 1024)
  if (max_w > 1024 || max_h > 1024) hipLaunchKernelGGL(k_mod_lfgroup_b, dim3(max_lfg, nframes), dim3(64), 0, s, Bs);   // only images beyond one LF group can carry such streams
  if (ngroups > 0) launch_mod_group_b(Bs, group_map, ngroups, pool_bytes, s);
  for (int o = 0; o < max_ops; o++) hipLaunchKernelGGL(k_mod_op_b, dim3(1024, 1, nframes), dim3(256), 0, s, Bs, o);
  hipLaunchKernelGGL(k_mod_write_b, dim3((max_w + 63) / 64, (max_h + 3) / 4, nframes), dim3(256), 0, s, Bs);
}


}  // namespace jxlamd
