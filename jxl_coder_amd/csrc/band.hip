// jxl_coder_amd/csrc/band.hip — band-sharded decode of ONE frame (BASELINE config 4: a 32768x32768 VarDCT image whose 256x256
// groups are sharded over the GPUs of a node, RCCL used only for the loop-filter halo at the band borders).
//
// A band = the group rows [gr0, gr1) of the frame.  PassGroups are independently entropy-decodable and reconstructable; only the
// image-space stages look across group borders:
//   * adaptive LF smoothing reads the LF sample above / below        -> "LF halo":    one cell row (3 x xb f32 LF + the quant-field
//     and EPF-sharpness bytes the EPF sigma of the first halo rows needs), exchanged after the LF stage;
//   * Gaborish (+-1 row) and EPF (+-3 / +-2 / +-1 rows per iteration) -> "pixel halo": H = gab + 3 + 2 + 1 (as configured) rows of the
//     PRE-filter XYB planes, exchanged after reconstruction.  Each later filter stage then recomputes a shrinking margin
//     around the band (k_filter_b: stage_halo_after), so the band's own rows come out bit-identical to a whole-frame decode.
// The decoder context runs the protocol  begin -> [LF halo export / import] -> reconstruct -> [pixel halo export / import] -> finish;
// halo buffers are plain device memory so that the caller moves them with ncclSend/ncclRecv (jxl_coder_amd/shard.py) or, for bands
// on the same GPU, hands one context's export straight to its neighbour's import.
// The reference has no counterpart (libjxl decodes a frame in one process); the boundary it sits under is the size guard of
// DecodeJpegXlOneShot (jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:103-109): a band is smaller than a Bitmap.
#include <chrono>
#include <functional>
#include <mutex>
#include <thread>
#include "decoder_ctx.h"

namespace {
constexpr int kHaloLf = 0, kHaloPixels = 1;
size_t lf_halo_bytes(const FramePlan &plan) { return (((size_t)plan.xb * (3 * 4 + 2)) + 15) & ~(size_t)15; }
size_t px_halo_bytes(const FramePlan &plan, const BandGeom &q) { return (size_t)q.halo * (size_t)plan.xb * 8 * 3 * 4; }
}  // namespace

int jxlamd_decoder::band_begin(const uint8_t *jxl, size_t size, uint32_t flags, int gr0, int gr1, void *out_ptr, size_t out_cap, jxlamd_info *info) {
  HIPCHECK(hipSetDevice(device));
  FrameSlot &S = slot(0);
  S.band_stage = 0;
  const int rows[2] = {gr0, gr1};
  static const bool trace = getenv("JXLAMD_TRACE_BANDS") && atoi(getenv("JXLAMD_TRACE_BANDS"));      // host wall-clock split of band_begin on stderr
  const auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  // The bands of one frame parse the SAME file (header, 16 642 TOC entries and the global tables of a 32768 x 32768 frame: 14 - 17 ms each): the
  // first band of a process to see a file parses it, the others copy the plan.  One entry, keyed by the caller's buffer, its size, a sample of all its
  // bytes and a FULL hash of the byte ranges a plan is made from — everything up to the end of LfGlobal (container boxes, headers, TOC) and the
  // HfGlobal section; the group sections are not part of a plan (the kernels read them from the buffer of the decode at hand).  Only codestreams
  // that alias the caller's buffer are cached: a plan that owns its codestream (assembled from jxlp boxes) points into its own copy.
  {
    static std::mutex mu; static const uint8_t *k_data = nullptr; static size_t k_size = 0; static uint64_t k_sig = 0, k_head_sig = 0; static size_t k_head_end = 0, k_hf0 = 0, k_hf1 = 0;
    static std::shared_ptr<FramePlan> cached;
    const auto fnv = [&](uint64_t h, size_t a, size_t b) { for (size_t i = a; i < b && i < size; i++) h = (h ^ jxl[i]) * 1099511628211ull; return h; };
    uint64_t sig = 1469598103934665603ull;
    for (size_t i = 0; i < size; i += (size <= 8192 ? 1 : (i < 4096 || i + 8192 >= size ? 1 : size / 4096))) sig = (sig ^ jxl[i]) * 1099511628211ull;
    std::lock_guard<std::mutex> lk(mu);
    if (cached && k_data == jxl && k_size == size && k_sig == sig && fnv(fnv(1469598103934665603ull, 0, k_head_end), k_hf0, k_hf1) == k_head_sig) S.plan = *cached;
    else {
      S.plan = FramePlan(); (void)plan_parse(jxl, size, &S.plan);
      cached.reset();
      if (S.plan.error.empty() && !S.plan.tables.empty() && S.plan.cs_owned.empty() && S.plan.cs >= jxl && S.plan.cs + S.plan.cs_size <= jxl + size) {
        const DevFrame *F = (const DevFrame *)S.plan.tables.data();
        const DevSection *secs = (const DevSection *)(S.plan.tables.data() + F->sec_off);
        const size_t cs_off = (size_t)(S.plan.cs - jxl);
        k_head_end = cs_off + (size_t)secs[0].off + secs[0].size;
        const int hf = F->nsec > 1 ? 1 + F->num_lf_groups : 0;
        k_hf0 = cs_off + (size_t)secs[hf].off; k_hf1 = k_hf0 + secs[hf].size;
        k_head_sig = fnv(fnv(1469598103934665603ull, 0, k_head_end), k_hf0, k_hf1);
        cached = std::make_shared<FramePlan>(S.plan); k_data = jxl; k_size = size; k_sig = sig;
      }
    }
  }
  int rc = prepare(S, jxl, size, nullptr, flags & ~JXLAMD_IN_DEVICE, out_ptr, out_cap, info, /*parsed=*/true, true, rows);
  if (rc) return rc;
  const double t_prepared = now();
  const BandGeom &q = S.band;
  // the band runs the flight kernels over a one-frame array: DevBuffers, DevAux and the (frame 0, group) maps of its LF groups / groups
  std::vector<int> maps;
  for (int g = q.lfg0; g < q.lfg0 + q.nlfg; g++) { maps.push_back(0); maps.push_back(g); }
  for (int g = q.g0; g < q.g0 + q.ng; g++) { maps.push_back(0); maps.push_back(g); }
  const size_t n_pairs = maps.size();
  int nwg = 0;
  for (int g = q.g0; g < q.g0 + q.ng; g += 64) { maps.push_back(0); maps.push_back(g); maps.push_back(std::min(64, q.g0 + q.ng - g)); nwg++; }     // k_pass_flat wavefronts
  const size_t o_a = (sizeof(DevBuffers) + 255) & ~(size_t)255, o_m = (o_a + sizeof(DevAux) + 255) & ~(size_t)255, total = o_m + maps.size() * 4;
  HIPCHECK(batch_tab.ensure(total));
  HIPCHECK(h_batch.ensure(total));
  uint8_t *hb = (uint8_t *)h_batch.p, *db = (uint8_t *)batch_tab.p;
  memcpy(hb, &S.B, sizeof(DevBuffers)); memcpy(hb + o_a, &S.A, sizeof(DevAux)); memcpy(hb + o_m, maps.data(), maps.size() * 4);
  HIPCHECK(hipMemcpyAsync(db, hb, total, hipMemcpyHostToDevice, stream));
  bandtab.dB = (const DevBuffers *)db; bandtab.dA = (const DevAux *)(db + o_a);
  bandtab.lf_map = (const int *)(db + o_m); bandtab.pg_map = bandtab.lf_map + 2 * q.nlfg; bandtab.wmap = bandtab.lf_map + n_pairs; bandtab.nwg = nwg; bandtab.flags = flags;
  HIPCHECK(hipEventRecord(ev[0], stream));
  launch_lf_groups_batch(bandtab.dB, bandtab.dA, bandtab.lf_map, q.nlfg, lf_pool_bytes, /*general=*/true, stream);
  HIPCHECK(hipEventRecord(ev[1], stream));
  // (into page-locked memory: a device-to-host copy into pageable memory is synchronous inside the runtime and kept the other bands' host
  // threads — eight decoder contexts of one process — out of their own enqueue calls for as long as this band's kernels ran)
  HIPCHECK(h_flags.ensure(256));
  HIPCHECK(hipMemcpyAsync(h_flags.p, S.B.err, 4, hipMemcpyDeviceToHost, stream));
  const double t_launched = now();
  HIPCHECK(hipStreamSynchronize(stream));
  HIPCHECK(hipGetLastError());
  const uint32_t derr = *(const uint32_t *)h_flags.p;
  if (trace) {
    float lf_ms = 0; (void)hipEventElapsedTime(&lf_ms, ev[0], ev[1]);
    fprintf(stderr, "[band %d-%d] parse + buffers + uploads queued %.1f ms, launch %.1f ms, wait %.1f ms (LF kernel %.1f ms on the device)\n", gr0, gr1, t_prepared - t_begin, t_launched - t_prepared,
            now() - t_launched, lf_ms);
  }
  if (derr) { set_error("corrupt or unsupported stream (device flags " + std::to_string(derr) + ", LfGroup of a band)"); return dev_err_class(derr); }
  S.band_stage = 1;
  return JXLAMD_OK;
}

int jxlamd_decoder::band_halo_bytes(int kind, size_t *bytes) {
  FrameSlot &S = slot(0);
  if (S.band_stage < 1 || !bytes || (kind != kHaloLf && kind != kHaloPixels)) { set_error("band: no band decode in progress"); return JXLAMD_ERR_BUFFER; }
  *bytes = kind == kHaloLf ? lf_halo_bytes(S.plan) : px_halo_bytes(S.plan, S.band);
  return JXLAMD_OK;
}

// side 0 = this band's TOP edge (sent to the band above), 1 = its BOTTOM edge (sent to the band below)
int jxlamd_decoder::band_export(int kind, int side, void *dev_buf, size_t cap) {
  HIPCHECK(hipSetDevice(device));
  FrameSlot &S = slot(0);
  const BandGeom &q = S.band; const FramePlan &plan = S.plan;
  const size_t xb = (size_t)plan.xb, pw = xb * 8;
  uint8_t *dst = (uint8_t *)dev_buf;
  if (kind == kHaloLf) {
    if (S.band_stage < 1) { set_error("band: LF halo exported before the LF stage"); return JXLAMD_ERR_BUFFER; }
    if (!dev_buf || cap < lf_halo_bytes(plan)) { set_error("band: halo buffer too small"); return JXLAMD_ERR_BUFFER; }
    const size_t r = (size_t)(side == 0 ? q.cy0 : q.cy1 - 1);
    for (int c = 0; c < 3; c++) HIPCHECK(hipMemcpyAsync(dst + (size_t)c * xb * 4, S.B.lf[c] + r * xb, xb * 4, hipMemcpyDeviceToDevice, stream));
    HIPCHECK(hipMemcpyAsync(dst + 12 * xb, S.B.qfm1 + r * xb, xb, hipMemcpyDeviceToDevice, stream));
    HIPCHECK(hipMemcpyAsync(dst + 13 * xb, S.B.sharp + r * xb, xb, hipMemcpyDeviceToDevice, stream));
  } else if (kind == kHaloPixels) {
    if (S.band_stage < 2) { set_error("band: pixel halo exported before reconstruction"); return JXLAMD_ERR_BUFFER; }
    if (q.halo == 0) return JXLAMD_OK;
    if (!dev_buf || cap < px_halo_bytes(plan, q)) { set_error("band: halo buffer too small"); return JXLAMD_ERR_BUFFER; }
    const size_t r = (size_t)(side == 0 ? q.py0 : q.cy1 * 8 - q.halo), n = (size_t)q.halo * pw;
    for (int c = 0; c < 3; c++) HIPCHECK(hipMemcpyAsync(dst + (size_t)c * n * 4, S.B.plane_a[c] + r * pw, n * 4, hipMemcpyDeviceToDevice, stream));
  } else { set_error("band: unknown halo kind"); return JXLAMD_ERR_BUFFER; }
  HIPCHECK(hipStreamSynchronize(stream));               // the buffer is complete when the call returns (the caller's send may use another stream)
  return JXLAMD_OK;
}

// side 0 = the rows ABOVE this band (the upper neighbour's bottom-edge export), 1 = the rows BELOW it
int jxlamd_decoder::band_import(int kind, int side, const void *dev_buf, size_t size) {
  HIPCHECK(hipSetDevice(device));
  FrameSlot &S = slot(0);
  const BandGeom &q = S.band; const FramePlan &plan = S.plan;
  const size_t xb = (size_t)plan.xb, pw = xb * 8;
  const uint8_t *src = (const uint8_t *)dev_buf;
  if ((side == 0 && q.gr0 == 0) || (side == 1 && q.cy1 >= plan.yb)) { set_error("band: no neighbour on that side (image edge)"); return JXLAMD_ERR_BUFFER; }
  if (kind == kHaloLf) {
    if (S.band_stage != 1) { set_error("band: LF halo must be imported between begin and reconstruct"); return JXLAMD_ERR_BUFFER; }
    if (!dev_buf || size < lf_halo_bytes(plan)) { set_error("band: halo message too small"); return JXLAMD_ERR_BUFFER; }
    const size_t r = (size_t)(side == 0 ? q.cy0 - 1 : q.cy1);
    for (int c = 0; c < 3; c++) HIPCHECK(hipMemcpyAsync(S.B.lf[c] + r * xb, src + (size_t)c * xb * 4, xb * 4, hipMemcpyDeviceToDevice, stream));
    HIPCHECK(hipMemcpyAsync(S.B.qfm1 + r * xb, src + 12 * xb, xb, hipMemcpyDeviceToDevice, stream));
    HIPCHECK(hipMemcpyAsync(S.B.sharp + r * xb, src + 13 * xb, xb, hipMemcpyDeviceToDevice, stream));
  } else if (kind == kHaloPixels) {
    if (S.band_stage != 2) { set_error("band: pixel halo must be imported between reconstruct and finish"); return JXLAMD_ERR_BUFFER; }
    if (q.halo == 0) return JXLAMD_OK;
    if (!dev_buf || size < px_halo_bytes(plan, q)) { set_error("band: halo message too small"); return JXLAMD_ERR_BUFFER; }
    const size_t r = (size_t)(side == 0 ? q.py0 - q.halo : q.cy1 * 8), n = (size_t)q.halo * pw;
    for (int c = 0; c < 3; c++) HIPCHECK(hipMemcpyAsync(S.B.plane_a[c] + r * pw, src + (size_t)c * n * 4, n * 4, hipMemcpyDeviceToDevice, stream));
  } else { set_error("band: unknown halo kind"); return JXLAMD_ERR_BUFFER; }
  HIPCHECK(hipStreamSynchronize(stream));               // the caller may reuse / free the message buffer
  return JXLAMD_OK;
}

// LF smoothing, PassGroup entropy decode, dequant + inverse DCT of the band's groups
int jxlamd_decoder::band_reconstruct() {
  HIPCHECK(hipSetDevice(device));
  FrameSlot &S = slot(0);
  if (S.band_stage != 1) { set_error("band: reconstruct needs a band whose LF stage is done"); return JXLAMD_ERR_BUFFER; }
  const BandGeom &q = S.band; const FramePlan &plan = S.plan;
  launch_lf_smooth(S.B, plan.xb, q.cy1 - q.cy0, stream);
  // a band of thousands of groups fills the chip with one LANE per group; below that the wave-per-group kernel has the shorter critical path
  if (q.ng >= ((bandtab.flags & JXLAMD_BAND_SHARED_GPU) ? flat_min_groups : band_flat_min_groups) && frame_flat_ok(plan)) { launch_pass_prep(bandtab.dB, bandtab.pg_map, q.ng, stream); launch_pass_flat(bandtab.dB, bandtab.wmap, bandtab.nwg, /*sparse=*/false, stream); }
  else launch_pass_groups_batch(bandtab.dB, bandtab.pg_map, q.ng, stream);
  HIPCHECK(hipEventRecord(ev[2], stream));
  HIPCHECK(huge_scratch.ensure((size_t)kHugeSlots * 2 * 65536 * 4));
  launch_rest_batch(bandtab.dB, (const uint8_t *)stat.p, 1, plan.xb * (q.scy1 - q.scy0), plan.width, q.py1 - q.py0, 0, /*expect_large=*/true, 1, stream, false, (float *)huge_scratch.p);
  HIPCHECK(hipEventRecord(ev[3], stream));
  HIPCHECK(h_flags.ensure(256));
  HIPCHECK(hipMemcpyAsync(h_flags.p, S.B.err, 4, hipMemcpyDeviceToHost, stream));
  HIPCHECK(hipStreamSynchronize(stream));
  HIPCHECK(hipGetLastError());
  const uint32_t derr = *(const uint32_t *)h_flags.p;
  if (derr) { S.band_stage = 0; set_error("corrupt or unsupported stream (device flags " + std::to_string(derr) + ", band)"); return dev_err_class(derr); }
  S.band_stage = 2;
  return JXLAMD_OK;
}

// Gaborish / EPF over the band (+ the margins the later stages need) and the RGBA writer for the band's rows
int jxlamd_decoder::band_finish() {
  HIPCHECK(hipSetDevice(device));
  FrameSlot &S = slot(0);
  if (S.band_stage != 2) { set_error("band: finish needs a reconstructed band"); return JXLAMD_ERR_BUFFER; }
  const BandGeom &q = S.band; const FramePlan &plan = S.plan;
  const DevFrame *F = (const DevFrame *)plan.tables.data();
  int stage_mask = 0;
  if (F->gab) stage_mask |= 1;
  if (F->epf_iters >= 3) stage_mask |= 2;
  if (F->epf_iters >= 1) stage_mask |= 4;
  if (F->epf_iters >= 2) stage_mask |= 8;
  if (F->epf_iters <= 2) stage_mask |= sweep_stage_bit(*F, (int)S.pi.out_bits, false);     // column-sweep instantiation
  if (!F->gab && !F->epf_iters) stage_mask |= 1 << 4;
  launch_rest_batch(bandtab.dB, (const uint8_t *)stat.p, 1, plan.xb * (q.scy1 - q.scy0), plan.width, (q.py1 - q.py0) + 2 * q.halo, stage_mask, true, 2, stream);
  HIPCHECK(hipEventRecord(ev[4], stream));
  S.band_stage = 0;
  int rc = collect(S, bandtab.flags);
  for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&timing[i], ev[i], ev[i + 1]);      // LF | halo wait + PassGroup | reconstruction | halo wait + filters
  (void)hipEventElapsedTime(&timing[4], ev[0], ev[4]);
  return rc;
}

extern "C" {
int jxlamd_band_begin(jxlamd_decoder *d, const uint8_t *jxl, size_t size, uint32_t flags, int group_row0, int group_row1, void *out, size_t cap,
                      jxlamd_info *info) {
  if (!d) { tls_error() = "null decoder"; return JXLAMD_ERR_DEVICE; }
  return jxlamd_guarded(d, [&]() -> int { return d->band_begin(jxl, size, flags, group_row0, group_row1, out, cap, info); });
}
int jxlamd_band_halo_bytes(jxlamd_decoder *d, int kind, size_t *bytes) { return d ? d->band_halo_bytes(kind, bytes) : JXLAMD_ERR_DEVICE; }
int jxlamd_band_export(jxlamd_decoder *d, int kind, int side, void *dev_buf, size_t cap) { return d ? d->band_export(kind, side, dev_buf, cap) : JXLAMD_ERR_DEVICE; }
int jxlamd_band_import(jxlamd_decoder *d, int kind, int side, const void *dev_buf, size_t size) { return d ? d->band_import(kind, side, dev_buf, size) : JXLAMD_ERR_DEVICE; }
int jxlamd_band_reconstruct(jxlamd_decoder *d) { return d ? d->band_reconstruct() : JXLAMD_ERR_DEVICE; }
int jxlamd_band_finish(jxlamd_decoder *d) { return d ? d->band_finish() : JXLAMD_ERR_DEVICE; }

// The cut of a frame of `ygroups` group rows into `nbands` bands (jxl_coder_amd/shard.py: band_rows): borders on multiples of 8 group rows — whole 2048-pixel
// LF groups, no LF stream decoded twice — whenever there are at least nbands LF-group rows.  rows[2 b], rows[2 b + 1] = first / one past the last group row of band b.
int jxlamd_band_rows(int ygroups, int nbands, int *rows) {
  if (!rows || nbands < 1 || nbands > ygroups) { tls_error() = "band rows: cannot cut the frame that way"; return JXLAMD_ERR_BUFFER; }
  const int lf_rows = (ygroups + 7) / 8;
  const int unit = lf_rows >= nbands ? 8 : 1, units = lf_rows >= nbands ? lf_rows : ygroups;
  for (int b = 0; b <= nbands; b++) {
    int cut = std::min(ygroups, (int)((int64_t)units * b / nbands) * unit);
    if (b == nbands) cut = ygroups;
    if (b < nbands) rows[2 * b] = cut;
    if (b > 0) rows[2 * b - 1] = cut;
  }
  return JXLAMD_OK;
}

// BASELINE config 4 for the bands ONE process holds, without Python (SURVEY.md §8b "decode_sharded"; what jxl_coder_amd/shard.py: decode_sharded does for a
// rank's bands): band b of `nbands` on decoder context decs[b] (same device; one context per band), its pixel rows into outs[b].  Every phase of the protocol
// runs the bands side by side — one host thread each, their own HIP streams — and the halo rows of neighbouring bands are handed from one context to the
// next through device buffers; a multi-process decode uses the same calls per band and moves the rank-border halos itself (ncclSend / ncclRecv).
int jxlamd_decode_sharded_local(jxlamd_decoder *const *decs, int nbands, const uint8_t *jxl, size_t size, uint32_t flags, void *const *outs, const size_t *caps,
                                jxlamd_info *info) {
  if (!decs || nbands < 1 || !jxl || !outs || !caps) { tls_error() = "decode_sharded_local: bad arguments"; return JXLAMD_ERR_BUFFER; }
  for (int b = 0; b < nbands; b++) if (!decs[b]) { tls_error() = "null decoder"; return JXLAMD_ERR_DEVICE; }
  // the halo buffers below are allocated by THIS thread and handed between the contexts' streams: one device for all bands, and this thread on it (on a
  // non-default GPU they landed on device 0 and every export / import became a cross-device copy)
  for (int b = 1; b < nbands; b++) if (decs[b]->device != decs[0]->device) { tls_error() = "decode_sharded_local: the bands' decoder contexts must be on one device"; return JXLAMD_ERR_DEVICE; }
  if (hipSetDevice(decs[0]->device) != hipSuccess) { tls_error() = "decode_sharded_local: cannot select the decoders' device"; return JXLAMD_ERR_DEVICE; }
  jxlamd_info bi;
  int rc = jxlamd_basic_info(jxl, size, &bi);
  if (rc) return rc;
  const int ygroups = (int)((bi.ysize + 255) / 256);      // (oriented height: a frame with orientation > 4 is refused by jxlamd_band_begin before its rows are looked at)
  std::vector<int> rows((size_t)nbands * 2);
  if ((rc = jxlamd_band_rows(ygroups, nbands, rows.data()))) return rc;
  const uint32_t fl = (flags & ~(uint32_t)JXLAMD_BAND_SHARED_GPU) | (nbands > 1 ? JXLAMD_BAND_SHARED_GPU : 0u);
  std::vector<int> rcs((size_t)nbands, JXLAMD_OK);
  std::vector<std::string> errs((size_t)nbands);
  const auto phase = [&](const std::function<int(int)> &fn) -> int {
    if (nbands == 1) { rcs[0] = fn(0); if (rcs[0]) errs[0] = decs[0]->error; }
    else {
      std::vector<std::thread> th;
      for (int b = 0; b < nbands; b++) th.emplace_back([&, b] { rcs[(size_t)b] = fn(b); if (rcs[(size_t)b]) errs[(size_t)b] = decs[b]->error; });
      for (auto &t : th) t.join();
    }
    for (int b = 0; b < nbands; b++) if (rcs[(size_t)b]) { tls_error() = errs[(size_t)b]; return rcs[(size_t)b]; }
    return JXLAMD_OK;
  };
  std::vector<jxlamd_info> infos((size_t)nbands);
  if ((rc = phase([&](int b) { return jxlamd_band_begin(decs[b], jxl, size, fl, rows[2 * (size_t)b], rows[2 * (size_t)b + 1], outs[b], caps[b], &infos[(size_t)b]); }))) return rc;
  if (info) *info = infos[0];
  DevMem halo[2];
  const auto exchange = [&](int kind) -> int {
    for (int b = 0; b + 1 < nbands; b++) {
      size_t n = 0;
      int e = jxlamd_band_halo_bytes(decs[b], kind, &n);
      if (e) return e;
      if (n == 0) continue;
      if (halo[0].ensure(n) != hipSuccess || halo[1].ensure(n) != hipSuccess) { tls_error() = "decode_sharded_local: no memory for the halo rows"; return JXLAMD_ERR_DEVICE; }
      if ((e = jxlamd_band_export(decs[b], kind, 1, halo[0].p, n)) || (e = jxlamd_band_export(decs[b + 1], kind, 0, halo[1].p, n))) { tls_error() = decs[b]->error.empty() ? decs[b + 1]->error : decs[b]->error; return e; }
      if ((e = jxlamd_band_import(decs[b + 1], kind, 0, halo[0].p, n)) || (e = jxlamd_band_import(decs[b], kind, 1, halo[1].p, n))) { tls_error() = decs[b]->error.empty() ? decs[b + 1]->error : decs[b]->error; return e; }
      // the imports are copies on the two contexts' streams: done before the buffers carry the next border
      if (hipStreamSynchronize(decs[b]->stream) != hipSuccess || hipStreamSynchronize(decs[b + 1]->stream) != hipSuccess) { tls_error() = "decode_sharded_local: halo copy failed"; return JXLAMD_ERR_DEVICE; }
    }
    return JXLAMD_OK;
  };
  if ((rc = exchange(kHaloLf))) return rc;
  if ((rc = phase([&](int b) { return jxlamd_band_reconstruct(decs[b]); }))) return rc;
  if ((rc = exchange(kHaloPixels))) return rc;
  return phase([&](int b) { return jxlamd_band_finish(decs[b]); });
}
}  // extern "C"
