// jxl_coder_amd/csrc/decoder_ctx.h — the decoder context shared by decoder.hip (whole-frame / flight decode, C-ABI) and band.hip
// (band-sharded decode of one frame, BASELINE config 4): device buffers, page-locked staging, per-frame slot.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include "../../include/jxl_amd.h"
#include "host_parse.h"
#include "kernels.h"
#include "post.h"
#include "host_post.h"

namespace jxlamd {
std::string &tls_error();
}
using namespace jxlamd;

// No C++ exception may cross the C-ABI (a crafted file that makes a std::vector throw must not std::terminate the host process): every
// extern "C" entry point that parses input or allocates runs its body through this.  Mapping as the reference's JNI layer does
// (JniDecoding.cpp:81-93): bad_alloc -> "Not enough memory to decode this image", anything else -> an invalid-stream error.
template <class Fn> static inline int jxlamd_guarded(jxlamd_decoder *d, Fn &&fn);
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string("HIP: ") + hipGetErrorString(e_) + " at " #x); return JXLAMD_ERR_DEVICE; } } while (0)



// default of jxlamd_decoder_set_epf_reciprocal for decoders of this process: JXLAMD_EPF_RCP=x86 (callers without a handle on the decoder — the libjxl-named compat library)
static inline int epf_rcp_mode_default() { const char *e = getenv("JXLAMD_EPF_RCP"); return e && (!strcmp(e, "x86") || !strcmp(e, "1")) ? 1 : 0; }

struct DevMem {
  void *p = nullptr; size_t cap = 0;
  DevMem() = default;
  DevMem(const DevMem &) = delete; DevMem &operator=(const DevMem &) = delete;
  ~DevMem() { release(); }            // owners cannot forget a member (the flight buffers leaked on destroy)
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    // growth headroom: a slot sees frames of different sizes flight after flight, and every re-allocation is a hipFree (device-wide
    // synchronisation that stalls the flights of all contexts); the big pools (GBs) cannot afford more than 1/8
    size_t want = n < ((size_t)64 << 20) ? n + n / 2 + 65536 : n + n / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  void swap(DevMem &o) { void *tp = p; p = o.p; o.p = tp; size_t tc = cap; cap = o.cap; o.cap = tc; }
};

struct PinnedMem {             // page-locked host staging: true async DMA, no shared pageable-copy staging in the runtime
  void *p = nullptr; size_t cap = 0;
  PinnedMem() = default;
  PinnedMem(const PinnedMem &) = delete; PinnedMem &operator=(const PinnedMem &) = delete;
  ~PinnedMem() { release(); }
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    size_t want = n + n / 4 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Rows of a frame one decode covers and the storage that backs them.  Kernels address cells / groups / pixels with FRAME
// coordinates; the buffer pointers in DevBuffers are biased by the *_0 origins below, so a band of a 32768^2 frame only allocates
// its own rows (+ halo).  Whole-frame decode: every origin 0.
struct BandGeom {
  int gr0 = 0, gr1 = 0;          // group rows
  int cy0 = 0, cy1 = 0;          // cell rows reconstructed here
  int py0 = 0, py1 = 0;          // pixel rows written here
  int lr0 = 0, lr1 = 0;          // LF-group rows decoded here (superset of the band when it is not LF-group aligned)
  int scy0 = 0, scy1 = 0;        // cell rows backed by storage: the LF-group rows + one halo row each side
  int st0 = 0, st1 = 0;          // 64x64 tile rows backed by storage
  int g0 = 0, ng = 0;            // first group / number of groups
  int lfg0 = 0, nlfg = 0;        // first LF group / number of LF groups
  int prow0 = 0, prow1 = 0;      // plane rows backed by storage (band + 8 rows of halo each side, clipped to the padded plane)
  int halo = 0;                  // H: pre-filter rows the loop filters read beyond the band (Gaborish 1 + EPF 3 / 2 / 1 per iteration)
  bool whole = true;
};
BandGeom band_geometry(const DevFrame &F, int gr0, int gr1);
int dev_err_class(uint32_t derr);
bool frame_flat_ok(const FramePlan &plan);
// {frame, first group, groups} entries of k_pass_flat for frames [0, n) with ngroups[i] groups each, ordered so that the wavefronts of one frame
// land on one XCD (workgroup b runs on XCD b % 8): its tables are then read through ONE 4 MB L2 instead of all eight
std::vector<int> flat_wave_map(const std::vector<int> &ngroups, const std::vector<int> &chain = std::vector<int>());

struct FrameSlot {             // HBM work buffers of one in-flight frame
  bool coef_clean = false; size_t coef_clean_bytes = 0; const void *coef_clean_ptr[3] = {nullptr, nullptr, nullptr};   // own coefficient planes known all-zero?
  PinnedMem h_tables, h_cs, h_B;
  DevMem dB;                     // device copy of B: single decodes run the flight kernels over a one-frame array (same pixels on every path)
  DevMem cs, tables, cells8[5], tiles[2], lf[6], coef_off, coef_cnt, coef[3], planes[6], lf_scratch, local, misc, out, mod_pool, mod_scratch, pass_nz, pass_end, big_list[4], lz_win, up_planes;
  DevMem own_ref[4];             // a frame of a flight that draws patches from reference frames: their images, taken over from the context's slots (decode_batch_once)
  FramePlan plan;
  DevBuffers B;
  DevAux A;
  jxlamd_info pi;
  size_t out_bytes = 0;
  void *d_out = nullptr; void *host_out = nullptr;
  const uint8_t *up_cs_dev = nullptr;   // frames of a flight: the caller's resident compressed bytes (null: host bytes), copied by the flight's gather launch
  // A10 + A11 behind / inside the writer (jxlamd_decoder_set_writer_post): `out` receives the Bitmap format.  fused: the frame's last filter stage emits it
  // (DevBuffers::post); otherwise the writer fills post_tmp (RGBA8 / RGBA16) and one k_post_fused pass over it follows the frame's kernels
  bool post_active = false, post_fused = false, post_runs = false, post_premul = false, post_att = false;
  int post_kind = 0; uint32_t post_depth = 8, post_stride = 0; size_t post_bytes = 0; uint64_t post_gen = 0; size_t post_off = 0;      // post_off: where the frame's DevPost sits inside its tables (flights); post_gen: which upload of the context's tone-map LUTs the frame was prepared against
  void *post_final = nullptr;
  DevMem post_dev, post_fz, post_tmp; PinnedMem h_post;
  DevMem noise_planes;           // frames with noise synthesis: three random planes
  BandGeom band;                 // rows this decode covers (whole frame unless jxlamd_band_begin set it up) and the storage behind them
  int band_stage = 0;            // band decode protocol: 0 idle, 1 LF stage done, 2 reconstructed, (finish returns to 0)
};

// HF-phase memory of a flight: hf_sets coefficient sets (3 x groups x 65536 i32, 106 MB per 4K frame) shared by its sub-flights and plane_sets x 6
// f32 planes shared by the frames of a sub-batch.  A flight only needs it from its PassGroup stage on; while its LF stage runs (a third of
// the flight's time, a few MB per frame) the memory can serve another context's HF phase: contexts that share one HfPools take turns (mu).
struct HfPools {
  DevMem plane_pool, coef_pool;
  DevMem sp_pool;                   // sparse flights: the frames' coefficient-entry arenas (4 bytes per nonzero coefficient; ~16 bytes per byte of PassGroup section) instead of coef_pool
  bool coef_pool_clean = false;     // the reconstruction kernels clear every coefficient they consume: after a flight without errors the pool is all-zero
  uint64_t generation = 0;          // bumped whenever a buffer is (re)allocated: a flight that baked the old addresses into its tables starts over
  std::mutex mu;
};

struct jxlamd_decoder {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  std::string error;
  DevMem stat, batch_tab, mod_tab, post_lin_lut, post_gam_lut, resample_tmp, icc_lut;
  ColorMatrixDev post_dev; double post_key[13] = {0}; bool post_key_valid = false, post_plan_runs = false;   // cached colour-matrix parameters
  bool huge_blocks_seen = false;         // DCT128 / DCT256-family varblocks met by this context: its flights launch their kernel from then on (k_recon_huge_b)
  DevMem huge_scratch;                   // ... and its S / T tiles (kHugeSlots workgroups x 2 x 65 536 floats)
  bool large_blocks_seen = false, large_hint = true;   // 64x64-class varblocks in the previous flight? (sizes the launch of their kernel)
  std::shared_ptr<HfPools> pools = std::make_shared<HfPools>();     // HF-phase memory of this context's flights (own, or shared: jxlamd_decoder_share_pools)
  PinnedMem h_batch, h_mod_tab, h_flight_tables, h_flight_cs, h_flags;
  DevMem flight_tables, flight_cs;       // tables / padded compressed bytes of all frames of a flight: one upload (or one gather launch) per flight
  std::vector<FrameSlot *> slots;
  std::vector<FrameSlot *> ref_slots;     // reference frames of the file being decoded (patch dictionaries), one slot each
  DevMem ref_store[8]; int ref_w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ref_h[8] = {0, 0, 0, 0, 0, 0, 0, 0}; bool ref_alpha[8] = {false, false, false, false, false, false, false, false};      // [4..7]: the LF frames (level 1..4) of a progressive_dc file
  int target_frame = -1;
  int epf_rcp_mode = epf_rcp_mode_default();      // jxlamd_decoder_set_epf_reciprocal: 0 exact quotient, 1 the reference x86 build's rcpps (rcp12_lut.h)
  bool wpost_enabled = false; int wpost_cfg = 0, wpost_api = 34; uint64_t post_lut_gen = 0;      // jxlamd_decoder_set_writer_post     // the four reference slots: 3 dense f32 planes each
  bool stat_uploaded = false;
  std::vector<uint8_t> icc_lut_key;        // the profile whose lattice icc_lut holds
  bool icc_lut_u16 = false;                // ... sampled through Little CMS's 16-bit transform (RGBA16 images) or its 8-bit one (RGBA8: host_icc_lut.cpp)
  struct { const DevBuffers *dB = nullptr; const DevAux *dA = nullptr; const int *lf_map = nullptr, *pg_map = nullptr, *wmap = nullptr; int nwg = 0; uint32_t flags = 0; } bandtab;   // band decode: device tables of the one-frame 'flight'
  // Groups in a (sub-)flight from which the lane-per-group kernel (k_pass_prep + k_pass_flat: ~100 ms of latency whatever the size, 17 M VALU
  // per 4K frame) takes over from the wave-per-group one (20 ms alone, 285 M VALU per frame: throughput-bound as soon as several contexts
  // run it).  Measured with 16 contexts (tools/gpu/run_flatmin.sh): flights of 8 / 16 / 24 4K frames 3 750 -> 4 800, 4 660 -> 8 180,
  // 4 750 -> 10 360 MP/s with the flat kernel.  JXLAMD_FLAT_MIN_GROUPS: test hook (1 forces it everywhere).
  int flat_min_groups = getenv("JXLAMD_FLAT_MIN_GROUPS") ? atoi(getenv("JXLAMD_FLAT_MIN_GROUPS")) : 1024;
  // ... and in a band of one frame (config 4: eight concurrent bands of 2 048 groups reconstruct in 33 - 36 ms each with the wave-per-group kernel)
  int band_flat_min_groups = getenv("JXLAMD_FLAT_MIN_GROUPS") ? atoi(getenv("JXLAMD_FLAT_MIN_GROUPS")) : 4096;
  float timing[5] = {0, 0, 0, 0, 0};
  uint32_t pool_retries = 0, general_retries = 0;   // decodes / flights run a second time (kErrNeedPool / kErrNeedGeneral)
  // Flights hand their coefficients over as sparse per-varblock lists (DevBuffers::coef_sp) unless JXLAMD_SPARSE=0; a flight whose streams beat the
  // arenas' sizing or carry values / positions an entry cannot hold (kErrNeedDense) is decoded again with the dense planes, and a context that met
  // three such flights stays dense
  bool sparse_enabled = !(getenv("JXLAMD_SPARSE") && atoi(getenv("JXLAMD_SPARSE")) == 0);
  uint32_t sparse_misses = 0; bool dense_flight = false, last_flight_sparse = false;
  uint32_t chained_tail_frames = 0;       // frames of flat flights whose tail groups rode as second groups (flat_tail_chain)
  uint64_t serial_streams = 0, block_tree_channels = 0;      // Modular streams that went to the serial walker / channels decoded with their MA tree in block form (device counters, summed over decodes)
  bool pool_missed = false;              // the flight in progress is the repeat of one that missed the pool
  int lf_pool_floor = 0;                 // the pool never shrinks below what a stream of this context once missed (kErrNeedPool)
  bool lf_general = false;               // the LF kernel build with the general lock-step loops (set for good the first time a frame of this context needs one)
  int lf_pool_bytes = kModPoolBytes;      // LDS table pool of the next LF launch: what the streams of the previous decode of this context asked for (first decode: the largest)
  void set_error(const std::string &e) { error = e; tls_error() = e; }
  FrameSlot &slot(size_t i) { while (slots.size() <= i) slots.push_back(new FrameSlot()); return *slots[i]; }

  int prepare(FrameSlot &S, const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info,
              bool parsed = false, bool own_planes = true, const int *band_rows = nullptr);
  // band-sharded decode of one frame (band.hip)
  int band_begin(const uint8_t *jxl, size_t size, uint32_t flags, int gr0, int gr1, void *out_ptr, size_t out_cap, jxlamd_info *info);
  int band_halo_bytes(int kind, size_t *bytes);
  int band_export(int kind, int side, void *dev_buf, size_t cap);
  int band_import(int kind, int side, const void *dev_buf, size_t size);
  int band_reconstruct();
  int band_finish();
  int finish_single_section(FrameSlot &S);
  int launch_rest(FrameSlot &S, int parts = 3, bool upload_B = false);     // parts: 1 = reconstruction, 2 = filters + writer
  int launch_compose_tail(FrameSlot &S);
  int decode_refs(FrameSlot &main, uint32_t flags, bool deferred = false);
  // reference frames of a flight's frames decoded WITHOUT a host synchronisation each (round 6): every one in a slot of its own (the page-locked staging of a slot is reused
  // only after the batch's one synchronisation), their device flags OR-ed into ref_err_dev, which decode_batch_once reads once behind the last of them
  DevMem ref_err_dev; PinnedMem h_ref_err; size_t ref_cursor = 0; bool refs_deferred = false;
  int run_frame(FrameSlot &S, uint32_t flags, bool single_latency);
  int launch_modular(FrameSlot &S);
  int launch_extra_channels(FrameSlot &S);
  int collect(FrameSlot &S, uint32_t flags);
  int decode(const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info, int frame = -1);
  void launch_post_pass(FrameSlot &S);
  int decode_once(const uint8_t *jxl, size_t size, const void *jxl_dev, uint32_t flags, void *out_ptr, size_t out_cap, jxlamd_info *info);
  int decode_batch_once(int n, const uint8_t *const *jxl, const size_t *sizes, const void *const *jxl_dev, uint32_t flags, void *const *outs,
                        const size_t *caps, jxlamd_info *infos);
  int decode_batch(int n, const uint8_t *const *jxl, const size_t *sizes, const void *const *jxl_dev, uint32_t flags, void *const *outs,
                   const size_t *caps, jxlamd_info *infos);
};

template <class Fn> static inline int jxlamd_guarded(jxlamd_decoder *d, Fn &&fn) {
  try { return fn(); }
  catch (const std::bad_alloc &) { const std::string m = "Not enough memory to decode this image"; if (d) d->set_error(m); else jxlamd::tls_error() = m; return JXLAMD_ERR_DEVICE; }
  catch (const std::exception &e) { const std::string m = std::string("Error: ") + e.what(); if (d) d->set_error(m); else jxlamd::tls_error() = m; return JXLAMD_ERR_INVALID; }
  catch (...) { const std::string m = "Error: unknown exception"; if (d) d->set_error(m); else jxlamd::tls_error() = m; return JXLAMD_ERR_INVALID; }
}
