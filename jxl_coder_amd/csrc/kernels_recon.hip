// jxl_coder_amd/csrc/kernels_recon.hip — HIP kernels (gfx950): adaptive LF smoothing, dequant + chroma-from-luma + LLF + inverse var-size DCT (LDS-staged per varblock),
// per-flight clears.
// Bodies live in dev_*.h (shared with the CPU test harness); this file only maps blockIdx/threadIdx.
#include "kernels_common.h"
#include "post.h"

namespace jxlamd {

__global__ void __launch_bounds__(256) k_lf_smooth(DevBuffers B) {
  const DevFrame &F = frame_of(B);
  int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= F.xb * (F.band_cy1 - F.band_cy0) || frame_failed(B)) return;
  lf_smooth_cell(B, i % F.xb, F.band_cy0 + i / F.xb);
}

// Varblock reconstruction, one kernel per transform family so that registers and LDS (and with them the occupancy next to resident
// entropy waves) fit the block; the placement step recorded the varblocks' cells in three size-class lists, fixed-size grids walk them.
constexpr int kStrategyDct32 = 5;
// DCT32x32 on the matrix cores (north_star: "MFMA only for the >= 32x32 matrix-form blocks").  Both 1-D passes are 32x32x32 products
//   pass 1: T_c = S_c^T x CC          pass 2: out_c = CC^T x T_c
// issued as 16 v_mfma_f32_32x32x2_f32 each (f32 in, f32 accumulate: exact f32, the only MFMA precision that keeps the +-1 LSB parity);
// waves 0..2 of the workgroup take one channel each.  Operand maps (cdna_hip_programming.md): lane l feeds A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; accumulator register r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// On gfx950 the f32 MFMA peak equals the f32 VALU peak (157 TFLOP/s both), so this is a change of execution unit — it takes the
// multiply-adds off the VALU that the co-resident entropy waves compete for — not a change of arithmetic throughput.
typedef __attribute__((ext_vector_type(16))) float f32x16;
// ccr[st] = cc[2 st + (lane >> 5)][lane & 31]: the cosine operand of step st of BOTH passes (pass 1 multiplies by cc[u][x], pass 2 by cc[v][y], same lane map) —
// sixteen registers loaded once per workgroup instead of a 4 KB table in LDS (round 5: LDS next to resident LF streams is what these workgroups wait for)
__device__ __forceinline__ void recon_dct32_mfma(const DevBuffers &B, const float *S, float *T, const float (&ccr)[16], int bx, int by, int tid) {
  const DevFrame &F = frame_of(B);
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kh = lane >> 5;
  const float *Sc = S + wave * 1024;
  float *Tc = T + wave * 1024;
  if (wave < 3) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < 16; st++) {
      const int u = 2 * st + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Sc[u * 32 + j], ccr[st], acc, 0, 0, 0);      // A[v][u] = S[u][v], B[u][x] = cc[u][x]
    }
#pragma unroll
    for (int r = 0; r < 16; r++) Tc[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + j] = acc[r];
  }
  __syncthreads();
  if (wave < 3) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int st = 0; st < 16; st++) {
      const int v = 2 * st + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ccr[st], Tc[v * 32 + j], acc, 0, 0, 0);      // A[y][v] = cc[v][y], B[v][x] = T[v][x]
    }
    float *out = B.plane_a[wave] + (size_t)(by * 8) * (size_t)F.pw + (size_t)(bx * 8 + j);
#pragma unroll
    for (int r = 0; r < 16; r++) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * kh) * (size_t)F.pw] = acc[r];
  }
}

// Front end of a DCT32x32 varblock (dequantisation + chroma from luma into S, lowest frequencies from the LF image), written for
// memory-level parallelism: recon_phaseA / recon_phaseB (dev_recon.h, the generic path) interleave dependent global loads with
// conditional stores, one coefficient at a time per work-item — four round trips for the coefficients and sixteen more for the 48
// work-items of the LLF corner, with a barrier in between.  Here every work-item first issues ALL its loads (12 coefficients and
// their weights; wave 3 also the 16 LF samples of its LLF output), then computes; the LLF corner is skipped by the dequantiser
// and written by wave 3, so the two need no barrier between them.  Same float operations in the same order as the generic path.
// kSparse: the coefficients come from the varblock's sparse list (DevBuffers::coef_sp; recon_phaseA_sparse, the same values): the tile is cleared and
// the few nonzero entries scattered into it — no 12 KB of mostly zero coefficients to fetch and to clear per block
template <bool kSparse>
__device__ __forceinline__ void recon_dct32_front(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *S, float *LL, int bx, int by, int tid) {
  const DevFrame &F = frame_of(B);
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  const int qt = kQuantTableOf[kStrategyDct32];
  const int li = tid - 192;
  const bool llf = li >= 0 && li < 48;
  float lf1 = 0.0f;
  if (kSparse) {
    if (llf) lf1 = B.lf_s[li >> 4][o + (size_t)((li >> 2) & 3) * (size_t)F.xb + (size_t)(li & 3)];      // issued before the barriers of the scatter
    recon_phaseA_sparse(B, stat, ST, S, 1024, bx, by, tid, 256, SyncBlock(), -1, kStrategyDct32);
    __syncthreads();                                   // the corner's positions may have received entries: the LLF values below replace them
  } else {
  const int g = (by / 32) * F.xgroups + (bx / 32);
  uint32_t off = B.coef_off[o];
  if (off + 1024u > 65536u) { if (tid == 0) *B.err |= kErrBitstream | kErrStageRecon; off = 0; }   // see recon_phaseA
  // block base addresses are wave-uniform: scalar base + one 32-bit lane offset per access (no 64-bit address registers per plane)
  const auto uptr = [](const void *p) { const uint64_t v = (uint64_t)p; return (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32))) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); };
  int32_t *qp[3]; const float *qw[3];
#pragma unroll
  for (int c = 0; c < 3; c++) { qp[c] = (int32_t *)uptr(&B.coef[c][(size_t)g * 65536 + off]) + tid; qw[c] = (const float *)uptr(quant_mul(B, F, stat, ST, qt, c)) + tid; }
  int q[3][4]; float w[3][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int c = 0; c < 3; c++) { q[c][j] = qp[c][256 * j]; w[c][j] = qw[c][256 * j]; }
  // LLF corner: work-item 192 + i, i < 48 (all in wave 3), produces coefficient (a, b) = (i / 4 % 4, i % 4) of channel i / 16.  Each of the
  // 48 loads ONE of the 3 x 16 LF samples (same index arithmetic: sample (iy, ix) = (i / 4 % 4, i % 4) of channel i / 16) and the wave
  // shares them through LDS — sixteen loads per work-item cost sixteen registers in every lane of the workgroup
  if (llf) lf1 = B.lf_s[li >> 4][o + (size_t)((li >> 2) & 3) * (size_t)F.xb + (size_t)(li & 3)];
  const auto sg = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };    // per-block values are wave-uniform: scalar registers
  const float mul = sg(F.inv_global_scale / (float)((int)B.qfm1[o] + 1));
  const size_t to = (size_t)(by / 8) * (size_t)F.tiles_x + (size_t)(bx / 8);
  const float kx = sg(F.base_x + (float)B.xfromy[to] * F.inv_color_factor);
  const float kb = sg(F.base_b + (float)B.bfromy[to] * F.inv_color_factor);
  const float qb[4] = {sg(F.quant_bias[0]), sg(F.quant_bias[1]), sg(F.quant_bias[2]), sg(F.quant_bias[3])};
  const float md[3] = {sg(mul * F.dm[0]), sg(mul * F.dm[1]), sg(mul * F.dm[2])};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = tid + 256 * j;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int qq = q[c][j];
      if (qq != 0) qp[c][256 * j] = 0;                 // consumed exactly once: the reader clears it (see recon_phaseA)
      float a;
      if (qq == 0) a = 0.0f;
      else if (qq == 1) a = qb[c];
      else if (qq == -1) a = -qb[c];
      else a = (float)qq - qb[3] / (float)qq;
      v[c] = a * (md[c] * w[c][j]);
    }
    if ((k >> 5) < 4 && (k & 31) < 4) continue;        // the LLF corner belongs to wave 3
    S[k] = cfl_add(v[0], kx, v[1]);
    S[1024 + k] = v[1];
    S[2048 + k] = cfl_add(v[2], kb, v[1]);
  }
  }
  if (llf) LL[32 + li] = lf1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");            // wave 3 only: its LDS accesses execute in order, the fences keep the compiler from reordering
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (llf) {
    const int r = li & 15, a = r >> 2, b = r & 3;      // cx == cy: horizontal frequency u = a, vertical v = b (recon_phaseB)
    const float *lf = LL + 32 + (li >> 4) * 16;
    float s = 0.0f;
#pragma unroll
    for (int iy = 0; iy < 4; iy++) {
      float rs = 0.0f;
#pragma unroll
      for (int ix = 0; ix < 4; ix++) rs += lf[iy * 4 + ix] * LL[a * 4 + ix];
      s += rs * LL[b * 4 + iy];
    }
    s *= (1.0f / 16.0f) * LL[16 + a] * LL[16 + b];
    S[(li >> 4) * 1024 + a * 32 + b] = s;
  }
}

// DCT64x64 on the matrix cores, one channel at a time (S and T hold 4096 coefficients each): wave w owns the 32x32 output tile
// (w >> 1, w & 1) of both 64x64x64 products, 32 v_mfma_f32_32x32x2_f32 each; the 64-point cosine table is read from the static tables
// (L2-resident).  Same operand maps as recon_dct32_mfma.
constexpr int kStrategyDct64 = 18;
template <bool kSparse>
__device__ __forceinline__ void recon_dct64_mfma(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *S, float *T, int bx, int by, int tid) {
  const DevFrame &F = frame_of(B);
  const float *cc = st_f(stat, ST.cos_off[6]);
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int t0 = (wave >> 1) * 32, t1 = (wave & 1) * 32;
  for (int c = 0; c < 3; c++) {
    if (kSparse) recon_phaseA_sparse(B, stat, ST, S, 4096, bx, by, tid, 256, SyncBlock(), c, kStrategyDct64);
    else recon_phaseA(B, stat, ST, S, 4096, bx, by, tid, 256, c);
    __syncthreads();
    recon_phaseB(B, stat, ST, S, 4096, bx, by, tid, 256, c);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll 8
    for (int st = 0; st < 32; st++) {
      const int u = 2 * st + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(S[u * 64 + t0 + j], cc[u * 64 + t1 + j], acc, 0, 0, 0);      // T[v][x] = sum_u S[u][v] cc[u][x]
    }
    __syncthreads();                                 // T may be S itself (k_recon_large_b): every wave has finished reading S
#pragma unroll
    for (int r = 0; r < 16; r++) T[(t0 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 64 + t1 + j] = acc[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll 8
    for (int st = 0; st < 32; st++) {
      const int v = 2 * st + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cc[v * 64 + t0 + j], T[v * 64 + t1 + j], acc, 0, 0, 0);      // out[y][x] = sum_v cc[v][y] T[v][x]
    }
    float *out = B.plane_a[c] + (size_t)(by * 8 + t0) * (size_t)F.pw + (size_t)(bx * 8 + t1 + j);
#pragma unroll
    for (int r = 0; r < 16; r++) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * kh) * (size_t)F.pw] = acc[r];
    __syncthreads();                                 // T is rewritten by the next channel's first pass
  }
}

__global__ void __launch_bounds__(256) k_lf_smooth_b(const DevBuffers *__restrict__ Bs) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || frame_failed(B)) return;
  int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= F.xb * (F.band_cy1 - F.band_cy0)) return;
  lf_smooth_cell(B, i % F.xb, F.band_cy0 + i / F.xb);
}
// DCT8x8 — nearly all of the small varblocks: one lane per coefficient / pixel, the three channels side by side, the 8-point cosine
// rows a lane needs (its x in the first pass, its y in the second) in registers.  Same operations in the same order as
// recon_phaseA / recon_phaseB / recon_idct_pass1 / recon_idct_pass2 for this strategy (one tenth of their instructions: no per-element
// index arithmetic, no table loads inside the sums).
template <bool kSparse>
__device__ __forceinline__ void recon_dct8_block(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *S, float *T, int bx, int by, int lane,
                                                 const float (&cx8)[8], const float (&cy8)[8]) {
  const DevFrame &F = frame_of(B);
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  const int qt = kQuantTableOf[0];
  if (kSparse) {
    float l3[3] = {0.0f, 0.0f, 0.0f};
    if (lane < 3) l3[0] = B.lf_s[lane][o];             // the LLF "corner" of a 1x1 block is the LF sample itself (all scales are 1)
    recon_phaseA_sparse(B, stat, ST, S, 64, bx, by, lane, 64, SyncBlock(), -1, 0);
    __syncthreads();
    if (lane < 3) S[lane * 64] = l3[0];
  } else {
  const int g = (by / 32) * F.xgroups + (bx / 32);
  uint32_t off = B.coef_off[o];
  if (off + 64u > 65536u) { if (lane == 0) *B.err |= kErrBitstream | kErrStageRecon; off = 0; }   // see recon_phaseA
  const float mul = F.inv_global_scale / (float)((int)B.qfm1[o] + 1);
  const size_t to = (size_t)(by / 8) * (size_t)F.tiles_x + (size_t)(bx / 8);
  const float kx = F.base_x + (float)B.xfromy[to] * F.inv_color_factor;
  const float kb = F.base_b + (float)B.bfromy[to] * F.inv_color_factor;
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    int32_t *qp = &B.coef[c][(size_t)g * 65536 + off + (uint32_t)lane];
    const int q = *qp;
    if (q != 0) *qp = 0;                               // consumed exactly once: the reader clears it
    float a;
    if (q == 0) a = 0.0f;
    else if (q == 1) a = F.quant_bias[c];
    else if (q == -1) a = -F.quant_bias[c];
    else a = (float)q - F.quant_bias[3] / (float)q;
    v[c] = a * (mul * F.dm[c] * quant_mul(B, F, stat, ST, qt, c)[lane]);
  }
  float s0 = cfl_add(v[0], kx, v[1]), s1 = v[1], s2 = cfl_add(v[2], kb, v[1]);
  if (lane == 0) { s0 = B.lf_s[0][o]; s1 = B.lf_s[1][o]; s2 = B.lf_s[2][o]; }      // the LLF "corner" of a 1x1 block is the LF sample itself (all scales are 1)
  S[lane] = s0; S[64 + lane] = s1; S[128 + lane] = s2;
  }
  __syncthreads();
  const int hi = lane >> 3, lo = lane & 7;
  float t[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int u = 0; u < 8; u++)
#pragma unroll
    for (int c = 0; c < 3; c++) t[c] += S[c * 64 + u * 8 + hi] * cx8[u];             // T[v][x], lane = v * 8 + x
#pragma unroll
  for (int c = 0; c < 3; c++) T[c * 64 + lane] = t[c];
  __syncthreads();
  float r[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int vv = 0; vv < 8; vv++)
#pragma unroll
    for (int c = 0; c < 3; c++) r[c] += T[c * 64 + vv * 8 + lo] * cy8[vv];            // out[y][x], lane = y * 8 + x
  const size_t po = (size_t)(by * 8 + hi) * (size_t)F.pw + (size_t)(bx * 8 + lo);
#pragma unroll
  for (int c = 0; c < 3; c++) B.plane_a[c][po] = r[c];
}

template <bool kSparse>
__global__ void __launch_bounds__(64) k_recon_small_b(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int skip_dct8) {
  __shared__ float S[3 * 256];
  __shared__ float T[256];
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular) return;
  if (frame_failed(B)) {
    if (kSparse) return;                               // nothing to tidy: a sparse arena is rewritten from its start by its next user
    // A PassGroup stream that stopped early leaves coefficients nobody will consume: the set must be all-zero again before its next
    // user (the frame hf_sets later in this flight, or the next decode), so the failed frame's launch clears it instead.
    const size_t g0 = (size_t)F.band_gr0 * (size_t)F.xgroups * 65536, g1 = (size_t)F.band_gr1 * (size_t)F.xgroups * 65536;
    for (int c = 0; c < 3; c++)
      for (size_t i = g0 + (size_t)blockIdx.x * 64 + threadIdx.x; i < g1; i += (size_t)gridDim.x * 64) B.coef[c][i] = 0;
    return;
  }
  const uint32_t count = B.big_count[2];
  // (the list, the strategy map, the row of cells and the band limits in registers: read through B / F inside the loop they are re-loaded behind every barrier)
  const uint32_t *list2 = B.big_list[2];
  const uint8_t *strategy = B.strategy;
  const int xb = F.xb, cy0 = F.band_cy0, cy1 = F.band_cy1;
  const bool skip_here = skip_dct8 && !F.subsampled;
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const int cell = (int)list2[i];
    const int by = cell / xb;
    if (by < cy0 || by >= cy1) continue;
    if (skip_here) { const int st = strategy[cell]; if (st == 0 || st == 4 || (st >= 6 && st <= 9)) continue; }      // k_recon_lists_a / _b have reconstructed it
    __syncthreads();
    recon_block_body<true, false, kSparse>(B, stat, S, T, cell % xb, by, 0, 256, (int)threadIdx.x, 64, SyncBlock());
  }
}
// A walker's position in its size-class list, two entries ahead: the cell of the block after next is requested while this block is worked on, and the strategy byte of the
// next block (whose cell arrived a block ago) with it — the list entry and the strategy byte used to be two dependent loads in front of every block.
struct CellWalk {
  const uint32_t *list; const uint8_t *strategy;
  uint32_t count, stride, i;
  int c0, c1, st0;
  __device__ __forceinline__ CellWalk(const uint32_t *l, const uint8_t *s, uint32_t n, uint32_t first, uint32_t step) : list(l), strategy(s), count(n), stride(step), i(first) {
    c0 = i < count ? (int)list[i] : -1;
    c1 = i + stride < count ? (int)list[i + stride] : -1;
    st0 = c0 >= 0 ? (int)strategy[c0] : -1;
  }
  __device__ __forceinline__ bool next(int &cell, int &st) {       // false: the list is done
    if (c0 < 0) return false;
    cell = c0; st = st0;
    i += stride;
    c0 = c1;
    c1 = i + stride < count ? (int)list[i + stride] : -1;
    st0 = c0 >= 0 ? (int)strategy[c0] : -1;
    return true;
  }
};
// list walkers of the one-wave-per-block families: workgroup `wg` of `nwg` takes every nwg-th entry of the class's size list and
// reconstructs the blocks of its own strategy.  smem: the workgroup's LDS (k_recon_lists_*: one launch for several families)
template <bool kSparse>
__device__ __forceinline__ void recon_dct8_walk(const DevBuffers &B, const uint8_t *stat, float *smem, uint32_t wg, uint32_t nwg) {
  float *S = smem, *T = smem + 3 * 64;
  const DevFrame &F = frame_of(B);
  const uint32_t count = B.big_count[2];
  if (wg >= count || F.subsampled) return;           // chroma-subsampled frames (recompressed JPEGs): the generic small-block kernel places each channel on its own grid
  const int lane = (int)threadIdx.x;
  const DevStatic &ST = *(const DevStatic *)stat;
  float cx8[8], cy8[8];                               // 8-point cosine table, row k: this lane's column x = lane & 7 / its row y = lane >> 3
  { const float *cc = st_f(stat, ST.cos_off[3]);
#pragma unroll
    for (int k = 0; k < 8; k++) { cx8[k] = cc[k * 8 + (lane & 7)]; cy8[k] = cc[k * 8 + (lane >> 3)]; } }
  // (the list, the strategy map, the row of cells and the band limits in registers: read through B / F inside the loop they are re-loaded behind every barrier)
  const uint32_t *list2 = B.big_list[2];
  const uint8_t *strategy = B.strategy;
  const int xb = F.xb, cy0 = F.band_cy0, cy1 = F.band_cy1;
  CellWalk walk(list2, strategy, count, wg, nwg);
  for (int cell, st; walk.next(cell, st);) {
    const int by = cell / xb;
    if (by < cy0 || by >= cy1 || st != 0) continue;
    __syncthreads();
    recon_dct8_block<kSparse>(B, stat, ST, S, T, cell % xb, by, lane, cx8, cy8);
  }
}
// DCT16x16, DCT16x8, DCT8x16 — with DCT8x8 the bulk of the varblocks of photographic content (the reference's 4K demo photograph: 9 700
// DCT16x16, 2 800 16x8 / 8x16, 1 800 DCT8x8, 4 400 DCT32x32).  One wave per block; lane = column x of a group of R C / 64 consecutive rows,
// so that the 1-D passes are register-blocked (one cosine value / one T value feeds R C / 64 multiply-adds) with both cosine tables in
// LDS; all coefficient loads of the block are issued up front.  Same operations in the same order as the generic path
// (recon_phaseA / recon_phaseB / recon_idct_pass1 / recon_idct_pass2, incl. the transposed storage of blocks with R >= C).
template <int R, int C, bool kSparse>
__device__ __forceinline__ void recon_dct_rc_block(const DevBuffers &B, const uint8_t *stat, const DevStatic &ST, float *S, float *T, const float *ccC, const float *crR,
                                                   int st, int bx, int by, int lane) {
  constexpr int N = R * C, NJ = N / 64;
  const DevFrame &F = frame_of(B);
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  if (kSparse) recon_phaseA_sparse(B, stat, ST, S, N, bx, by, lane, 64, SyncBlock(), -1, st);
  else {
  const int qt = kQuantTableOf[st];
  const int g = (by / 32) * F.xgroups + (bx / 32);
  uint32_t off = B.coef_off[o];
  if (off + (uint32_t)N > 65536u) { if (lane == 0) *B.err |= kErrBitstream | kErrStageRecon; off = 0; }   // see recon_phaseA
  int q[3][NJ]; float w[3][NJ];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int j = 0; j < NJ; j++) { q[c][j] = B.coef[c][(size_t)g * 65536 + off + (uint32_t)(lane + 64 * j)]; w[c][j] = quant_mul(B, F, stat, ST, qt, c)[lane + 64 * j]; }
  const float mul = F.inv_global_scale / (float)((int)B.qfm1[o] + 1);
  const size_t to = (size_t)(by / 8) * (size_t)F.tiles_x + (size_t)(bx / 8);
  const float kx = F.base_x + (float)B.xfromy[to] * F.inv_color_factor;
  const float kb = F.base_b + (float)B.bfromy[to] * F.inv_color_factor;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int k = lane + 64 * j;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int qq = q[c][j];
      if (qq != 0) B.coef[c][(size_t)g * 65536 + off + (uint32_t)k] = 0;     // consumed exactly once: the reader clears it
      float a;
      if (qq == 0) a = 0.0f;
      else if (qq == 1) a = F.quant_bias[c];
      else if (qq == -1) a = -F.quant_bias[c];
      else a = (float)qq - F.quant_bias[3] / (float)qq;
      v[c] = a * (mul * F.dm[c] * w[c][j]);
    }
    S[k] = cfl_add(v[0], kx, v[1]);
    S[N + k] = v[1];
    S[2 * N + k] = cfl_add(v[2], kb, v[1]);
  }
  }
  __syncthreads();
  recon_phaseB(B, stat, ST, S, N, bx, by, lane, 64);          // the LLF corner (C / 8 x R / 8 values per channel) overwrites its positions
  __syncthreads();
  const int x = lane % C, r0 = (lane / C) * NJ;                // this lane: column x, rows r0 .. r0 + NJ - 1 of both passes
  float acc[3][NJ];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[c][j] = 0.0f;
#pragma unroll 4
  for (int u = 0; u < C; u++) {
    const float ccv = ccC[u * C + x];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[c][j] += (R < C ? S[c * N + (r0 + j) * C + u] : S[c * N + u * R + r0 + j]) * ccv;      // T[v][x], v = r0 + j
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int j = 0; j < NJ; j++) { T[c * N + (r0 + j) * C + x] = acc[c][j]; acc[c][j] = 0.0f; }
  __syncthreads();
#pragma unroll 4
  for (int v = 0; v < R; v++) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float tv = T[c * N + v * C + x];
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[c][j] += tv * crR[v * R + r0 + j];              // out[y][x], y = r0 + j
    }
  }
  const size_t po = (size_t)(by * 8 + r0) * (size_t)F.pw + (size_t)(bx * 8 + x);
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int j = 0; j < NJ; j++) B.plane_a[c][po + (size_t)j * (size_t)F.pw] = acc[c][j];
}
template <int R, int C, int STRAT, int LIST, bool kSparse>      // LIST: the size-class list the strategy's blocks are on (2: <= 256 coefficients, 0: 512 / 1024)
__device__ __forceinline__ void recon_dct_rc_walk(const DevBuffers &B, const uint8_t *stat, float *smem, uint32_t wg, uint32_t nwg) {
  float *S = smem, *T = smem + 3 * R * C, *ccC = smem + 6 * R * C, *crR = smem + 6 * R * C + C * C;
  const DevFrame &F = frame_of(B);
  const uint32_t count = B.big_count[LIST];
  if (wg >= count) return;
  const int lane = (int)threadIdx.x;
  const DevStatic &ST = *(const DevStatic *)stat;
  for (int i = lane; i < C * C; i += 64) ccC[i] = st_f(stat, ST.cos_off[C == 8 ? 3 : C == 16 ? 4 : 5])[i];
  for (int i = lane; i < R * R; i += 64) crR[i] = st_f(stat, ST.cos_off[R == 8 ? 3 : R == 16 ? 4 : 5])[i];
  const uint32_t *list = B.big_list[LIST];         // (in registers: see recon_dct8_walk)
  const uint8_t *strategy = B.strategy;
  const int xb = F.xb, cy0 = F.band_cy0, cy1 = F.band_cy1;
  CellWalk walk(list, strategy, count, wg, nwg);
  for (int cell, st; walk.next(cell, st);) {
    const int by = cell / xb;
    if (by < cy0 || by >= cy1 || st != STRAT) continue;
    __syncthreads();
    recon_dct_rc_block<R, C, kSparse>(B, stat, ST, S, T, ccC, crR, STRAT, cell % xb, by, lane);
  }
}
// The one-wave-per-block families in TWO launches instead of nine (blockIdx.y = family): every launch of a flight's stream is a
// serialisation point — the previous kernel drains, the next one waits for slots among the kernels of 15 other contexts — and these
// kernels carry a few per cent of the area of smooth content.  _a: DCT8x8, 16x16, 16x8, 8x16 (8 KB of LDS, <= 90 VGPRs: the bulk of
// photographic content); _b: the 32-wide / 32-tall rectangles (17 KB).
template <bool kSparse>
__global__ void __launch_bounds__(64) k_recon_lists_a(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat) {
  __shared__ __attribute__((aligned(16))) float smem[6 * 256 + 2 * 256];
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || frame_failed(B)) return;
  switch (blockIdx.y) {
    case 0: recon_dct8_walk<kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;
    case 1: recon_dct_rc_walk<16, 16, 4, 2, kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;     // AcStrategy 4: DCT16x16
    case 2: recon_dct_rc_walk<16, 8, 6, 2, kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;      // 6: 16 rows x 8 columns
    default: recon_dct_rc_walk<8, 16, 7, 2, kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;     // 7: 8 x 16
  }
}
template <bool kSparse>
__global__ void __launch_bounds__(64) k_recon_lists_b(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat) {
  __shared__ __attribute__((aligned(16))) float smem[6 * 512 + 256 + 1024];
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || frame_failed(B)) return;
  switch (blockIdx.y) {
    case 0: recon_dct_rc_walk<32, 8, 8, 2, kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;      // 8: 32 x 8
    case 1: recon_dct_rc_walk<8, 32, 9, 2, kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;      // 9: 8 x 32
    case 2: recon_dct_rc_walk<32, 16, 10, 0, kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;    // 10: 32 x 16
    default: recon_dct_rc_walk<16, 32, 11, 0, kSparse>(B, stat, smem, blockIdx.x, gridDim.x); break;   // 11: 16 x 32
  }
}
// DCT32x32 blocks only (98 % of the area of smooth 4K content): half the LDS of the general medium kernel (the second pass runs in
// place: wave c reads all of channel c before it writes) and its own, smaller register footprint — what the data-parallel kernels can
// use next to resident entropy waves is what decides their speed in a flight mix.
struct ReconDct32Lds { float S[3 * 1024]; float LL[96]; };
template <bool kSparse>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_recon_dct32_b(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat) {
  __shared__ __attribute__((aligned(16))) ReconDct32Lds L;
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || frame_failed(B)) return;
  const int tid = (int)threadIdx.x, xb = F.xb;
  const DevStatic &ST = *(const DevStatic *)stat;
  const uint32_t count = B.big_count[0];
  if (blockIdx.x >= count) return;
  float ccr[16];
  {
    const float *cc = st_f(stat, ST.cos_off[5]);
    const int lane = tid & 63;
#pragma unroll
    for (int st = 0; st < 16; st++) ccr[st] = cc[(2 * st + (lane >> 5)) * 32 + (lane & 31)];
    if (tid < 16) L.LL[tid] = st_f(stat, ST.cos_off[2])[tid];
    else if (tid < 20) L.LL[tid] = (st_f(stat, ST.llf_off) + 64)[tid - 16];
  }
  // (the list, the strategy map and the band limits in registers: read through B / F inside the loop they are re-loaded behind every barrier, each a round trip in
  // front of the block's own dependent loads)
  const uint32_t *list0 = B.big_list[0];
  const uint8_t *strategy = B.strategy;
  const int cy0 = F.band_cy0, cy1 = F.band_cy1;
  CellWalk walk(list0, strategy, count, blockIdx.x, gridDim.x);
  for (int cell, st; walk.next(cell, st);) {
    const int bx = cell % xb, by = cell / xb;
    if (by < cy0 || by >= cy1 || st != kStrategyDct32) continue;
    __syncthreads();                                 // the previous block's second pass has finished reading S
    recon_dct32_front<kSparse>(B, stat, ST, L.S, L.LL, bx, by, tid);
    __syncthreads();
    recon_dct32_mfma(B, L.S, L.S, ccr, bx, by, tid);
  }
}
// The 512 / 1024-coefficient blocks that are NOT DCT32x32 (DCT16x32, 32x16, 8x32, ... — a few per cent of the blocks), one channel at a
// time: 8 KB of LDS instead of the general medium kernel's 33 KB, so that this short launch is not kept waiting for LDS by resident LF waves.
template <bool kSparse>
__global__ void __launch_bounds__(256) k_recon_medium_pc_b(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int skip_rc) {
  __shared__ __attribute__((aligned(16))) float S[1024];
  __shared__ __attribute__((aligned(16))) float T[1024];
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || frame_failed(B)) return;
  const int tid = (int)threadIdx.x, xb = F.xb;
  const uint32_t count = B.big_count[0];
  const uint32_t *list0 = B.big_list[0];           // (in registers: see recon_dct8_walk)
  const uint8_t *strategy = B.strategy;
  const int cy0 = F.band_cy0, cy1 = F.band_cy1;
  CellWalk walk(list0, strategy, count, blockIdx.x, gridDim.x);
  for (int cell, st; walk.next(cell, st);) {
    const int bx = cell % xb, by = cell / xb;
    if (by < cy0 || by >= cy1 || st == kStrategyDct32 || (skip_rc && (st == 10 || st == 11))) continue;     // k_recon_dct32_b / k_recon_lists_b
    __syncthreads();
    recon_block_body<false, true, kSparse>(B, stat, S, T, bx, by, 257, 1024, tid, 256, SyncBlock());
  }
}
// The 2048 / 4096-coefficient blocks with half the LDS of k_recon_list_b<1025, 4096>: DCT64x64 on the matrix cores with the second pass in
// place (16 KB), the 64x32 / 32x64 blocks one channel at a time in S[2048] + T[2048].
template <bool kSparse>
__global__ void __launch_bounds__(256) k_recon_large_b(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat) {
  __shared__ __attribute__((aligned(16))) float S[4096];
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular || frame_failed(B)) return;
  const int tid = (int)threadIdx.x, xb = F.xb;
  const DevStatic &ST = *(const DevStatic *)stat;
  const uint32_t count = B.big_count[1];
  const uint32_t *list1 = B.big_list[1];           // (in registers: see recon_dct8_walk)
  const uint8_t *strategy = B.strategy;
  const int cy0 = F.band_cy0, cy1 = F.band_cy1;
  CellWalk walk(list1, strategy, count, blockIdx.x, gridDim.x);
  for (int cell, st; walk.next(cell, st);) {
    const int bx = cell % xb, by = cell / xb;
    if (by < cy0 || by >= cy1) continue;
    __syncthreads();
    if (st == kStrategyDct64) recon_dct64_mfma<kSparse>(B, stat, ST, S, S, bx, by, tid);
    else recon_block_body<false, true, kSparse>(B, stat, S, S + 2048, bx, by, 1025, 2048, tid, 256, SyncBlock());
  }
}
// The DCT128 / DCT256 families (AcStrategy 21 .. 26: 8 192 .. 65 536 coefficients; libjxl's encoder never selects them, its decoder takes them): one
// workgroup per varblock, one channel at a time through the generic front end and the two separable passes — matrix products on the matrix cores (recon_huge_block) —, with the S and T tiles in HBM — a
// workgroup's own 512 KB slice of the context's scratch (L2-resident while it is worked on).  kHugeSlots workgroups walk the lists of ALL frames of the launch.
// OUT (M x N) = A (M x K) * B (K x N) on the matrix cores, operands straight from HBM / L2 (the S and T tiles of one block and the cosine tables: 64 KB - 256 KB each,
// L2-resident while the block is worked on): one wave per 32 x 32 output tile, v_mfma_f32_32x32x2_f32 over K two at a time — operand maps as in recon_dct32_mfma.
template <class FA, class FB, class FO>
__device__ __forceinline__ void huge_gemm_mfma(int M, int N, int K, FA a, FB b, FO out, int tid) {
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int tn_count = N >> 5, tiles = (M >> 5) * tn_count;
  for (int t = wave; t < tiles; t += 4) {
    const int tm = t / tn_count, tn = t - tm * tn_count;
    const int row = tm * 32 + j, col = tn * 32 + j;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a(row, k + kh), b(k + kh, col), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) out(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, col, acc[r]);
  }
}
// one varblock of the DCT128 / DCT256 families, one channel at a time: the generic front end (dequantisation, chroma from luma, LLF) into S, then both separable
// passes as matrix products (recon_idct_pass1 / pass2 restated: T = A x CC with A[v][u] = S[v][u] or, for R >= C, S's transposed storage; pixels = CR^T x T)
__device__ __forceinline__ void recon_huge_block(const DevBuffers &B, const uint8_t *stat, float *S, float *T, int bx, int by, int tid) {
  const DevFrame &F = frame_of(B);
  const DevStatic &ST = *(const DevStatic *)stat;
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  if (!B.first[o]) return;
  const int st = B.strategy[o];
  const int cx = kCoveredX[st], cy = kCoveredY[st];
  const int n = cx * cy * 64;
  if (n <= 4096 || F.subsampled) return;
  const int R = cy * 8, C = cx * 8;
  const float *cc = st_f(stat, ST.cos_off[ilog2(C)]), *cr = st_f(stat, ST.cos_off[ilog2(R)]);
  const size_t po = (size_t)by * 8 * (size_t)F.pw + (size_t)bx * 8;
  for (int c = 0; c < 3; c++) {
    recon_phaseA(B, stat, ST, S, n, bx, by, tid, 256, c);
    __syncthreads();
    recon_phaseB(B, stat, ST, S, n, bx, by, tid, 256, c);
    __syncthreads();
    if (R < C) huge_gemm_mfma(R, C, C, [&](int v, int u) { return S[v * C + u]; }, [&](int u, int x) { return cc[u * C + x]; }, [&](int v, int x, float val) { T[v * C + x] = val; }, tid);
    else huge_gemm_mfma(R, C, C, [&](int v, int u) { return S[u * R + v]; }, [&](int u, int x) { return cc[u * C + x]; }, [&](int v, int x, float val) { T[v * C + x] = val; }, tid);
    __syncthreads();
    float *dst = B.plane_a[c] + po;
    const size_t pw = (size_t)F.pw;
    huge_gemm_mfma(R, C, R, [&](int y, int v) { return cr[v * R + y]; }, [&](int v, int x) { return T[v * C + x]; }, [&](int y, int x, float val) { dst[(size_t)y * pw + (size_t)x] = val; }, tid);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_recon_huge_b(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int nframes, float *scratch) {
  float *S = scratch + (size_t)blockIdx.x * 2 * 65536, *T = S + 65536;
  const int tid = (int)threadIdx.x;
  for (int f = 0; f < nframes; f++) {
    const DevBuffers &B = Bs[f];
    const DevFrame &F = frame_of(B);
    if (F.is_modular || frame_failed(B)) continue;
    const uint32_t count = B.big_count[3];
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
      const int cell = (int)B.big_list[3][i];
      const int bx = cell % F.xb, by = cell / F.xb;
      if (by < F.band_cy0 || by >= F.band_cy1) continue;
      __syncthreads();
      recon_huge_block(B, stat, S, T, bx, by, tid);
    }
  }
}
void launch_recon_huge(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int nframes, float *scratch, hipStream_t s) {
  hipLaunchKernelGGL(k_recon_huge_b, dim3(kHugeSlots), dim3(256), 0, s, Bs, stat, nframes, scratch);
}
template <bool kSparse>
static void launch_recon_batch_t(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int nframes, int max_cells, bool expect_large, hipStream_t s) {
  // a single decode has the chip to itself: more, shorter workgroups for the list walkers; in a flight (16 frames per launch) 256 workgroups
  // per frame fill the chip, and the launches of the families a frame does not use cost 4 096 empty workgroups instead of 16 384
  const int gm = nframes == 1 ? 2048 : 256, gl = nframes == 1 ? 512 : 64;
  const dim3 gs(std::min(max_cells, nframes == 1 ? 8192 : 256), 1, nframes);
  const dim3 gs4(gs.x, 4, nframes);
  hipLaunchKernelGGL(k_recon_dct32_b<kSparse>, dim3(gm, 1, nframes), dim3(256), 0, s, Bs, stat);
  hipLaunchKernelGGL(k_recon_lists_a<kSparse>, gs4, dim3(64), 0, s, Bs, stat);                     // DCT8x8, 16x16, 16x8, 8x16
  hipLaunchKernelGGL(k_recon_lists_b<kSparse>, gs4, dim3(64), 0, s, Bs, stat);                     // 32x8, 8x32, 32x16, 16x32
  hipLaunchKernelGGL(k_recon_small_b<kSparse>, gs, dim3(64), 0, s, Bs, stat, 1);                   // the other <= 256-coefficient transforms (AFV, DCT4x8, ...)
  hipLaunchKernelGGL(k_recon_medium_pc_b<kSparse>, dim3(nframes > 1 ? 64 : gm, 1, nframes), dim3(256), 0, s, Bs, stat, 1);
  // the 2048 / 4096-coefficient list: one workgroup per frame when the previous flight had none
  hipLaunchKernelGGL(k_recon_large_b<kSparse>, dim3(expect_large ? gl : 1, 1, nframes), dim3(256), 0, s, Bs, stat);
}
// sparse: the frames' coefficients are per-varblock sparse lists (DevBuffers::coef_sp, written by k_pass_flat<true>) — every frame of the launch
void launch_recon_batch(const DevBuffers *__restrict__ Bs, const uint8_t *__restrict__ stat, int nframes, int max_cells, bool expect_large, bool sparse, hipStream_t s) {
  if (sparse) launch_recon_batch_t<true>(Bs, stat, nframes, max_cells, expect_large, s);
  else launch_recon_batch_t<false>(Bs, stat, nframes, max_cells, expect_large, s);
}
// one launch clears what hipMemsetAsync cleared per frame: the placement map and the flags / counters / LF bookkeeping block
__global__ void __launch_bounds__(256) k_clear_b(const DevBuffers *__restrict__ Bs) {
  const DevBuffers &B = Bs[blockIdx.z];
  const DevFrame &F = frame_of(B);
  if (F.is_modular) return;
  if (blockIdx.x == 0) {
    uint32_t *m = B.err;                                   // misc block: 4096 bytes of flags + 72 bytes per LF group
    const int words = (4096 + F.num_lf_groups * 72) / 4;
    for (int i = (int)threadIdx.x; i < words; i += 256) m[i] = 0;
    if (B.post) {                                          // the writer hands this frame's pixels to the post stages: "no zero-luma pixel seen" (dev_post.h: post_emit)
      uint32_t *fz = B.post->row_fz;
      for (int i = (int)threadIdx.x; i <= B.post->rows; i += 256) fz[i] = i ? 0xFFFFFFFFu : 0u;
    }
    return;
  }
  const int ncell = F.xb * (F.band_scy1 - F.band_scy0);        // the rows backed by storage (whole frame unless this is a band decode)
  uint8_t *first = B.first + (size_t)F.band_scy0 * (size_t)F.xb;
  const int i = (int)((blockIdx.x - 1) * 256 + threadIdx.x) * 16;
  if (i + 16 <= ncell) { uint4 z = {0, 0, 0, 0}; *(uint4 *)(first + i) = z; }       // cell arrays are 256-byte aligned allocations
  else for (int k = i; k < ncell; k++) first[k] = 0;
}
__global__ void __launch_bounds__(256) k_gather_streams(const GatherDesc *descs) {
  const GatherDesc g = descs[blockIdx.y];
  const uint32_t words = (g.size + g.pad + 3) / 4;            // dst is 256-byte aligned; bytes beyond `size` are written as zeros (the bit reader's padding)
  const bool aligned = ((uintptr_t)g.src & 3) == 0;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < words; w += gridDim.x * 256) {
    const uint32_t base = w * 4;
    uint32_t v = 0;
    if (aligned && base + 4 <= g.size) v = ((const uint32_t *)g.src)[w];
    else for (uint32_t b = 0; b < 4; b++) if (base + b < g.size) v |= (uint32_t)g.src[base + b] << (8 * b);
    ((uint32_t *)g.dst)[w] = v;
  }
}
void launch_gather_streams(const GatherDesc *descs, int n, uint32_t max_bytes, hipStream_t s) {
  const uint32_t words = (max_bytes + 64 + 3) / 4;
  hipLaunchKernelGGL(k_gather_streams, dim3(std::min<uint32_t>((words + 1023) / 1024, 256u), n), dim3(256), 0, s, descs);
}
__global__ void __launch_bounds__(256) k_gather_flags(const DevBuffers *__restrict__ Bs, int n, uint32_t *out) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i < n * kFlagWords) out[i] = Bs[i / kFlagWords].err[i % kFlagWords];
}
void launch_gather_flags(const DevBuffers *__restrict__ Bs, int n, uint32_t *out, hipStream_t s) {
  hipLaunchKernelGGL(k_gather_flags, dim3((n * kFlagWords + 255) / 256), dim3(256), 0, s, Bs, n, out);
}
void launch_clear_batch(const DevBuffers *__restrict__ Bs, int nframes, int max_cells, hipStream_t s) {
  hipLaunchKernelGGL(k_clear_b, dim3((max_cells + 4095) / 4096 + 1, 1, nframes), dim3(256), 0, s, Bs);
}
void launch_lf_smooth(const DevBuffers &B, int xb, int yb, hipStream_t s) {
  hipLaunchKernelGGL(k_lf_smooth, dim3((xb * yb + 255) / 256), dim3(256), 0, s, B);
}
void launch_lf_smooth_batch(const DevBuffers *__restrict__ Bs, int nframes, int max_cells, hipStream_t s) {
  hipLaunchKernelGGL(k_lf_smooth_b, dim3((max_cells + 255) / 256, 1, nframes), dim3(256), 0, s, Bs);
}

}  // namespace jxlamd
