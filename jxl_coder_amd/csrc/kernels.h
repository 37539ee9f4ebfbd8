// jxl_coder_amd/csrc/kernels.h — launch interface of the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_bodies.h"
#include "dev_modframe.h"
#include "dev_pass_flat.h"
#include "dev_compose.h"
namespace jxlamd {
// pool_bytes: LDS table pool per stream of this launch (kModPoolMin .. kModPoolBytes, dev_modular.h); lf_pool_clamp turns what the streams of
// a decode reported (word 1 of a frame's flag block) into the value for the next one
// general: the build with the general lock-step loops (a frame the lean build ended with kErrNeedGeneral)
void launch_lf_groups(const DevBuffers &B, const DevAux &A, int num_lf_groups, int pool_bytes, bool general, hipStream_t s);
void launch_lf_groups_batch(const DevBuffers *Bs, const DevAux *As, const int *map, int nblocks, int pool_bytes, bool general, hipStream_t s);
int lf_pool_clamp(uint32_t wanted);
// flights / large bands: k_pass_prep (group descriptor lists; map = {frame, group} pairs) then k_pass_flat (lane per group; wmap = {frame,
// first group, groups <= 64} per wavefront, entries with 0 groups allowed); frames must pass flat_frame_ok (dev_pass_flat.h)
void launch_pass_prep(const DevBuffers *Bs, const int *map, int ngroups, hipStream_t s);
void launch_pass_flat(const DevBuffers *Bs, const int *wmap, int nwg, bool sparse, hipStream_t s);      // sparse: coefficients into DevBuffers::coef_sp (single-pass frames)
void launch_pass_groups_batch(const DevBuffers *Bs, const int *map, int nblocks, hipStream_t s);
void launch_recon_batch(const DevBuffers *Bs, const uint8_t *stat, int nframes, int max_cells, bool expect_large, bool sparse, hipStream_t s);
// varblocks of more than 4 096 coefficients (big_list[3]); scratch: kHugeSlots x 2 x 65 536 floats.  Dense coefficient planes only (a sparse entry's
// 12-bit position does not reach them: such a frame ends its PassGroup stage with kErrNeedDense)
constexpr int kHugeSlots = 32;
void launch_recon_huge(const DevBuffers *Bs, const uint8_t *stat, int nframes, float *scratch, hipStream_t s);
void launch_filters_batch(const DevBuffers *Bs, const uint8_t *stat, int nframes, int max_w, int max_h, int stage_mask, hipStream_t s);
// stage_mask bit of a frame's column-sweep instantiation: 8 + (gab ? 3 : 0) + epf_iters, + kSweepFastShift when the frame takes the sweep's fast writer
// (host twin of sweep_fast_frame, kernels_filter.hip: sRGB curve, RGBA8, identity orientation, frame = canvas, no alpha plane, no post stages in the writer)
// ... + kSweepPostShift when the frame's writer emits the Bitmap format through the post stages (jxlamd_decoder_set_writer_post: k_filter_sweep<.., .., false, 1 / 2>)
constexpr int kSweepFastShift = 6, kSweepPostShift = 12;
inline int sweep_stage_bit(const DevFrame &F, int out_bits, bool writer_post) {
  const bool fast = F.transfer == 13 && F.orientation == 1 && out_bits == 8 && !(F.has_ec && F.mod_out[3] >= 0) && F.crop_x0 == 0 && F.crop_y0 == 0 &&
                    F.canvas_w == F.width && F.canvas_h == F.height && !writer_post;
  return 1 << (8 + (writer_post ? kSweepPostShift : fast ? kSweepFastShift : 0) + (F.gab ? 3 : 0) + F.epf_iters);
}
// parts: 1 = reconstruction kernels, 2 = filters + writer, 3 = both
inline void launch_rest_batch(const DevBuffers *Bs, const uint8_t *stat, int nframes, int max_cells, int max_w, int max_h, int stage_mask, bool expect_large,
                              int parts, hipStream_t s, bool sparse = false, float *huge_scratch = nullptr) {
  if (parts & 1) launch_recon_batch(Bs, stat, nframes, max_cells, expect_large, sparse, s);
  if ((parts & 1) && huge_scratch && !sparse) launch_recon_huge(Bs, stat, nframes, huge_scratch, s);
  if (parts & 2) launch_filters_batch(Bs, stat, nframes, max_w, max_h, stage_mask, s);
}
void launch_clear_batch(const DevBuffers *Bs, int nframes, int max_cells, hipStream_t s);
// flight plumbing: the compressed bytes of all (device-resident) frames of a flight into the flight's padded stream buffer in ONE launch,
// and the flag / counter words of all frames into one array for ONE device-to-host copy (instead of ~5 stream operations per frame)
struct GatherDesc { const uint8_t *src; uint8_t *dst; uint32_t size, pad; };
void launch_gather_streams(const GatherDesc *descs, int n, uint32_t max_bytes, hipStream_t s);
constexpr int kFlagWords = 20;
void launch_gather_flags(const DevBuffers *Bs, int n, uint32_t *out, hipStream_t s);
void launch_lf_smooth_batch(const DevBuffers *Bs, int nframes, int max_cells, hipStream_t s);
void launch_lf_smooth(const DevBuffers &B, int xb, int yb, hipStream_t s);
void launch_pass_groups(const DevBuffers &B, int num_groups, hipStream_t s);
// filters ping-pong between plane_a and plane_b; `src_is_a` tells where the current image is; returns the new flag
// Modular-encoded (lossless) frames
void launch_mod_global(const DevBuffers &B, int pool_bytes, hipStream_t s);
void launch_mod_groups(const DevBuffers &B, int num_groups, int pool_bytes, hipStream_t s);      // pool_bytes: LDS table pool of every group stream's workgroup (mod_group_pool_bytes, decoder.hip)
void launch_mod_lfgroups(const DevBuffers &B, int num_lf_groups, hipStream_t s);      // ModularLfGroup streams of a Modular-encoded frame
void launch_mod_op(const DevBuffers &B, int op, size_t n, hipStream_t s);
void launch_mod_write(const DevBuffers &B, int width, int height, int out_bits, hipStream_t s);
// composition stages (dev_compose.h): frames with a patch dictionary and the reference frames it draws on
void launch_mod_to_planes(const DevBuffers &B, int w, int h, hipStream_t s);
void launch_patch_blend(const DevBuffers &B, int num_patches, size_t max_px, hipStream_t s);
void launch_splines(const DevBuffers &B, int w, int h, hipStream_t s);      // after the patches (dev_compose.h: spline_pixel)
void launch_noise(const DevBuffers &B, int w, int h, hipStream_t s);
void launch_chroma_upsample(const DevBuffers &B, int w, int h, hipStream_t s);
void launch_blend_canvas(const DevBuffers &B, const uint8_t *stat, int canvas_w, int canvas_h, hipStream_t s);
void launch_save_ref(const DevBuffers &B, int w, int h, float *dst, hipStream_t s);
void launch_compose_write(const DevBuffers &B, const uint8_t *stat, int w, int h, hipStream_t s);
void launch_upsample_alpha(const DevBuffers &B, const uint8_t *stat, int full_w, int full_h, hipStream_t s);      // alpha coded coarser than the image -> DevBuffers::up[3]
void launch_upsample_and_write(const DevBuffers &B, const uint8_t *stat, int full_w, int full_h, bool noise, bool write, hipStream_t s);      // upsampled frames: enlarge, then write at full resolution
void launch_ec_global_batch(const DevBuffers *Bs, int nframes, int pool_bytes, hipStream_t s);
void launch_ec_groups_batch(const DevBuffers *Bs, const int *group_map, int nframes, int ngroups, int max_ops, int pool_bytes, hipStream_t s);
void launch_modular_batch(const DevBuffers *Bs, const int *group_map, int nframes, int ngroups, int max_ops, int max_w, int max_h, int pool_bytes, hipStream_t s);
}  // namespace jxlamd
