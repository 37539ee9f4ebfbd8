// jxl_coder_amd/csrc/dev_bodies.h — per-workgroup bodies of the decode kernels, written against
// (tid, nthreads, sync) so that kernels.hip instantiates them with __syncthreads() and the CPU test harness
// (tests/emul, test-only) runs each workgroup as one serial "thread" (nthreads = 1, sync = no-op).
#pragma once
#include "dev_recon.h"
#include "dev_modframe.h"

namespace jxlamd {

struct DevAux {
  uint64_t *lf_end_bits;     // [num_lf_groups]: bits consumed by each LfGroup section (single-section frames need it)
  uint64_t *lf_times;        // [num_lf_groups][8]: phase timestamps (100 MHz wall clock) for profiling
};

// ---- LfGroup: one workgroup (one wave) per 2048x2048 LF group
#ifdef __HIPCC__
#define JXL_STAMP(i) do { if (tid == 0 && A.lf_times) A.lf_times[g * 8 + (i)] = wall_clock64(); } while (0)
#else
#define JXL_STAMP(i) do { } while (0)
#endif
template <bool kWave = true, bool kGeneral = true, class Sync>
JXL_DEV void lf_group_body(const DevBuffers &B, const DevAux &A, DevModScratch &S, int g, int tid, int nthreads, Sync sync, int pool_bytes = kModPoolBytes) {
  JXL_STAMP(0);
  if (tid == 0) { S.pool_bytes = pool_bytes; S.pool_want = B.err + 1; S.walk_stat = B.err + 2; S.wide_wp = nullptr; }      // word 1 of the frame's flag block: LDS table pool the streams would have liked
#ifdef __HIPCC__
  const uint64_t cyc0 = __builtin_readcyclecounter();      // shader clock (s_memtime): with the 100 MHz wall stamps it gives the effective clock
#endif
  if (tid == 0) lf_phase_open(B, S, g);
  sync();
  uint32_t e = 0;
  if (!frame_of(B).use_lf_frame) {                          // (uniform) progressive_dc frames have no LF-coefficient stream
    modular_stream_stage(S, tid, nthreads);
    sync();
    JXL_STAMP(1);
    e = lf_phase_coeffs<kWave, kGeneral>(B, S, g, tid);     // whole wave on the GPU (kWave), or this lane alone
  }
  JXL_STAMP(2);
  {
    const DevFrame &Fm = frame_of(B);
    if (Fm.has_ec && Fm.mod_lf_nch > 0) {                   // ModularLfGroup stream of the extra channels: between the LF coefficients and the HF metadata
      sync();
      if (tid == 0) S.st.err = e;
      sync();
      if (!S.st.err) { const uint32_t e2 = mod_lfgroup_body<kGeneral>(B, S, g, tid, nthreads, sync); if (e2) e = e2; }
      if (tid == 0) S.st.err = 0;
      sync();
    }
  }
  if (tid == 0) { if (!e) e = lf_phase_meta_open(B, S, g); if (e) { S.st.err = e; *B.err |= e | kErrStageLf; } }
  sync();
  if (S.st.err) return;                                       // uniform: read from LDS after the barrier
  modular_stream_stage(S, tid, nthreads);
  sync();
  JXL_STAMP(3);
  e = lf_phase_meta<kWave, kGeneral>(B, S, g, tid);
  JXL_STAMP(4);
  if (tid == 0 && e) { S.st.err = e; *B.err |= e | kErrStageLf | (1u << 20); }
  sync();
  if (S.st.err) return;
  e = lf_phase_place(B, S, g, A.lf_end_bits, tid, nthreads, sync);      // uniform result
  if (tid == 0 && e) { S.st.err = e; *B.err |= e | kErrStageLf | (1u << 20); }
  sync();
  if (S.st.err) return;
  JXL_STAMP(5);
  lf_group_epilogue(B, g, tid, nthreads);
  JXL_STAMP(6);
#ifdef __HIPCC__
  if (tid == 0 && A.lf_times) A.lf_times[g * 8 + 7] = __builtin_readcyclecounter() - cyc0;
#endif
}

// ---- PassGroup: one workgroup (one wave) per 256x256 group; passes are sequential inside
template <class Sync>
JXL_DEV void pass_group_body(const DevBuffers &B, DevPassScratch &S, int g, int tid, int nthreads, Sync sync) {
  const DevFrame &F = frame_of(B);
  for (int p = 0; p < F.num_passes; p++) {
    if (tid == 0) pass_phase_open(B, S, p, g);
    sync();
    pass_phase_stage(B, S, tid, nthreads);
    sync();
    if (tid == 0) { uint32_t e = pass_phase_decode(B, S, g); if (e) { S.err = e; *B.err |= e | kErrStagePass; } }
    sync();
    if (S.err) return;
  }
}

// ---- a DCT8x8 varblock of a chroma-subsampled YCbCr frame (recompressed JPEG): channel c has a block at cell (bx, by) only when the cell is aligned to its
// sampling, and that block lives at (bx >> hshift, by >> vshift) of the channel's own grid — LF sample, pixels.  Other strategies do not occur there.
template <class Sync>
JXL_DEV void recon_block_subsampled(const DevBuffers &B, const uint8_t *stat, float *S, float *T, int bx, int by, int st, int tid, int nthreads, Sync sync) {
  const DevFrame &F = frame_of(B);
  const DevStatic &ST = *(const DevStatic *)stat;
  if (st != 0) { if (tid == 0) *B.err |= kErrUnsupportedBlock | kErrStageRecon; return; }
  recon_phaseA(B, stat, ST, S, 64, bx, by, tid, nthreads);      // a channel without a block here reads zeros: its coefficients were never written
  sync();
  bool present[3]; int cbx[3], cby[3];
  for (int c = 0; c < 3; c++) {
    cbx[c] = bx >> F.hshift[c]; cby[c] = by >> F.vshift[c];
    present[c] = (cbx[c] << F.hshift[c]) == bx && (cby[c] << F.vshift[c]) == by;
  }
  if (tid == 0) for (int c = 0; c < 3; c++) if (present[c]) S[c * 64] = B.lf_s[c][(size_t)cby[c] * (size_t)F.xb + (size_t)cbx[c]];      // LLF of a 1 x 1 block: the LF sample (all scales are 1)
  sync();
  for (int c = 0; c < 3; c++) {
    if (!present[c]) continue;
    recon_idct_pass1(stat, ST, S + c * 64, T, 8, 8, tid, nthreads);
    sync();
    recon_idct_pass2(stat, ST, T, B.plane_a[c] + (size_t)cby[c] * 8 * (size_t)F.pw + (size_t)cbx[c] * 8, F.pw, 8, 8, tid, nthreads);
    sync();
  }
}

// ---- varblock reconstruction; LDS: S[3*n] + T[n]
// kSpecial: the 8x8 special transforms can occur (small-block launch only); kPerChannel: S holds ONE channel (LDS: S[n] + T[n])
// kSparse: the coefficients come from the varblock's sparse list (DevBuffers::coef_sp) instead of the dense planes
template <bool kSpecial, bool kPerChannel = false, bool kSparse = false, class Sync>
JXL_DEV void recon_block_body(const DevBuffers &B, const uint8_t *stat, float *S, float *T, int bx, int by, int nmin, int nmax,
                              int tid, int nthreads, Sync sync) {
  const DevFrame &F = frame_of(B);
  const DevStatic &ST = *(const DevStatic *)stat;
  const size_t o = (size_t)by * (size_t)F.xb + (size_t)bx;
  if (!B.first[o]) return;                       // uniform across the workgroup
  const int st = B.strategy[o];
  const int cx = kCoveredX[st], cy = kCoveredY[st];
  const int n = cx * cy * 64;
  // (the DCT128 / DCT256 families — up to 65 536 coefficients — have their own launch, with S and T in HBM: k_recon_huge_b)
  if (n < nmin || n > nmax) return;                 // another size class' launch handles it
  if (F.subsampled) { recon_block_subsampled(B, stat, S, T, bx, by, st, tid, nthreads, sync); return; }
  float *dst[3] = {B.plane_a[0], B.plane_a[1], B.plane_a[2]};
  const size_t po = (size_t)by * 8 * (size_t)F.pw + (size_t)bx * 8;
  if (kPerChannel) {
    const int R = cy * 8, C = cx * 8;
    for (int c = 0; c < 3; c++) {
      if (kSparse) recon_phaseA_sparse(B, stat, ST, S, n, bx, by, tid, nthreads, sync, c);
      else recon_phaseA(B, stat, ST, S, n, bx, by, tid, nthreads, c);
      sync();
      recon_phaseB(B, stat, ST, S, n, bx, by, tid, nthreads, c);
      sync();
      recon_idct_pass1(stat, ST, S, T, R, C, tid, nthreads);
      sync();
      recon_idct_pass2(stat, ST, T, dst[c] + po, F.pw, R, C, tid, nthreads);
      sync();
    }
    return;
  }
  if (kSparse) recon_phaseA_sparse(B, stat, ST, S, n, bx, by, tid, nthreads, sync);
  else recon_phaseA(B, stat, ST, S, n, bx, by, tid, nthreads);
  sync();
  recon_phaseB(B, stat, ST, S, n, bx, by, tid, nthreads);
  sync();
  if (kSpecial && strategy_is_special(st)) {
    for (int c = tid; c < 3; c += nthreads) recon_special(stat, ST, st, S + c * n, dst[c] + po, F.pw);
    return;
  }
  const int R = cy * 8, C = cx * 8;
  for (int c = 0; c < 3; c++) {
    recon_idct_pass1(stat, ST, S + c * n, T, R, C, tid, nthreads);
    sync();
    recon_idct_pass2(stat, ST, T, dst[c] + po, F.pw, R, C, tid, nthreads);
    sync();
  }
}

}  // namespace jxlamd
