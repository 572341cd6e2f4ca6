// sgs_kernels.h -- host-side launcher declarations (internal to libsgs_hip.so).
#pragma once
#include "sgs_device.h"
#include <stddef.h>

namespace sgs {

// ---- preprocess.hip
void launch_preprocess_fwd(hipStream_t st, int P, int D, int M, const float* means3D,
			   const float* scales, float mod, const float* rotations,
			   const float* opacities, const float* shs, const float* cov3D_precomp,
			   const float* colors_precomp, const float* view, const float* proj,
			   const float* campos, int W, int H, float tanx, float tany, float fx,
			   float fy, int gx, int gy, int prefiltered, int num_channels, int* radii,
			   float2* means2D, float* depths, float* cov3Ds, float* rgb,
			   uint8_t* clamped, float4* conic_opacity, uint32_t* tiles_touched,
			   int* trap_flag, uint32_t* ds_cnt0 = nullptr, uint32_t* ds_gcnt0 = nullptr);
void launch_mark_visible(hipStream_t st, int P, const float* means3D, const float* view,
			 uint8_t* present);

// ---- depth_sort.hip: the depth presort of the Gaussians (4 kernels, no look-back, no fills of its own)
constexpr int DS_TILE = 4096;   // keys per workgroup
constexpr int DS_GRP = 32;      // tiles per group row of the count matrices
struct DepthSortLayout {        // byte offsets inside the sort's scratch
	size_t counts, counts_bytes;   // 4 passes x (tiles + groups) rows of 256 counters; must be ZERO before pass 0's
	                               // matrices are filled (preprocess.hip does that, with atomics)
	size_t keys[2], vals[2], total;
	size_t chain;                  // (round 6, inside the zeroed counts region) ghist[3][256] | ticket[4]: the chained passes' global digit histograms and tile tickets
	size_t sg;                     // (round 6, inside the zeroed counts region) 4 passes x sgroups rows: a third level of count rows (32 groups each) for more than 64 groups
	int tiles, groups, sgroups;
};
struct DepthSortSpanOut {   // optional by-product of the last pass: what binning_rows.hip's span_counts_kernel writes
	const int* radii;
	const float2* means2D;
	int gx, gy, major_x;
	uint64_t* counts64;
	uint4* rrec;
	unsigned long long* total;   // += sum(counts64); zero before the sort
	// stage A's one-segment tables for binning_rows.hip (segstart[2] | chunk0[2] | grp0[2]): six words the last pass's first
	// workgroup writes, so that the span partitions start without a table kernel of their own (nullptr: not wanted)
	uint32_t* stage_a_tab;
	uint32_t stage_a_chunks, stage_a_groups;
	// (round 6, deferred-count forwards) the count record written by the LAST workgroup of the sort's last pass instead of a kernel of its own
	// (capi.hip count_check_kernel): cc_done = a zeroed word (nullptr: not wanted); capacities, the trap flag, the device and the pinned host record
	uint32_t* cc_done;
	uint32_t cc_L_cap, cc_R_cap;
	const int* cc_trap;
	uint32_t* cc_rec;
	volatile uint32_t* cc_host;
};
void depth_sort_layout(int P, DepthSortLayout* lay);
hipError_t launch_depth_sort(hipStream_t st, int P, const DepthSortLayout& lay, char* scratch,
			     const uint32_t* depth_bits, uint32_t* perm, const DepthSortSpanOut* span = nullptr);
// the same from a bare array of keys (clears and fills pass 0's count matrices itself)
hipError_t launch_depth_sort_standalone(hipStream_t st, int P, const DepthSortLayout& lay, char* scratch,
					const uint32_t* keys, uint32_t* perm);

// ---- binning.hip
size_t scan_temp_bytes(int P);
hipError_t launch_inclusive_scan(hipStream_t st, void* temp, size_t temp_bytes,
				 const uint32_t* in, uint32_t* out, int P);
void launch_duplicate_with_keys(hipStream_t st, int P, const float2* means2D, const float* depths,
				const uint32_t* offsets, const int* radii, int gx, int gy,
				uint64_t* keys, uint32_t* vals, uint32_t L, const uint32_t* perm);
void launch_gather_counts(hipStream_t st, int P, const uint32_t* perm, const uint32_t* tiles_touched,
			  uint32_t* counts_sorted);
void launch_emit_tile_keys(hipStream_t st, int P, uint32_t L, const float2* means2D,
			   const uint32_t* offsets, const int* radii, const uint32_t* perm, int gx,
			   int gy, uint32_t* keys32, uint32_t* vals);
size_t sort32_temp_bytes(size_t L, int end_bit);
hipError_t launch_sort32_pairs(hipStream_t st, void* temp, size_t temp_bytes, uint32_t* keys_in,
			       uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t L,
			       int end_bit);
void launch_tile_ranges32(hipStream_t st, size_t L, const uint32_t* tiles, uint2* ranges, int ntiles);
// ---- binning_rows.hip (binning mode 0)
void row_binning_scratch(int P, uint32_t R, int gx, int gy, size_t* tab_words, size_t* cmat_words,
			 size_t* gtot_words, size_t* len_words);
// R may be an upper bound of the major-instance count (grids and scratch are sized from it, the kernels read the
// actual counts from device tables); abort: optional device word, != 0 -> every kernel exits
hipError_t launch_row_binning(hipStream_t st, int P, uint32_t R, int gx, int gy, const uint4* rrec, uint2* items, uint32_t* tabs, uint32_t* cmat,
			      uint32_t* gtot, uint32_t* lens, uint2* ranges, uint32_t* point_list, const uint32_t* abort = nullptr,
			      uint32_t* arena_counter = nullptr, uint32_t arena_first_free = 0,
			      const uint32_t* stage_a_tab = nullptr);   // stage_a_tab: DepthSortSpanOut::stage_a_tab, already written (else a table kernel runs)
// chunks / scan groups of stage A (the P ranked Gaussians as one segment): what DepthSortSpanOut::stage_a_chunks / _groups must hold
void row_binning_stage_a_counts(int P, uint32_t* chunks, uint32_t* groups);   // optional: also reset the split blend's counter
void launch_reconstruct_keys_ranges(hipStream_t st, int ntiles, const uint2* ranges, const uint32_t* point_list,
				    const float* depths, uint64_t* keys_sorted);
size_t sort_temp_bytes(size_t L, int begin_bit, int end_bit);
hipError_t launch_sort_pairs(hipStream_t st, void* temp, size_t temp_bytes, uint64_t* keys_in,
			     uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t L,
			     int begin_bit, int end_bit);
void launch_tile_ranges(hipStream_t st, size_t L, const uint64_t* keys, uint2* ranges, int ntiles);

// ---- blend_fwd.hip
struct BlendFwdArgs {
	const uint2* ranges;
	const uint32_t* point_list;
	int W, H, C;
	int gx, gy;
	const float2* means2D;
	const float* features;       // (P,C) row-major
	const float4* conic_opacity;
	const float* depths;         // per Gaussian view z (RGB-D variant)
	const float* bg;             // (C)
	float* final_T;              // (H*W)
	uint32_t* n_contrib;         // (H*W)
	float* out;                  // (C,H,W), rows `pitch` floats apart (pitch >= W; pitch == W: contiguous)
	float* out_depth;            // (H*W) or null
	int pitch;                   // output row pitch in pixels
	const uint32_t* abort;       // optional device word: != 0 -> every blend kernel exits (deferred-count forward, capi.hip)
	uint32_t* usage_host;        // optional pinned {work-list slots requested, overflow flag}: written by the sweep plan kernel
	bool counter_reset_done;     // the work-list counter was already reset (launch_row_binning): no arena_reset_kernel
	bool norm_plane;             // SGS_OPT_NORM_PLANE: `out` is an (H, pitch) plane that receives sum_c out[c]^2 (atomics), no feature map
	// optional, per (device, stream): [0] = number of tiles the order below is for (0 = none yet), [1 ..] = the tiles sorted
	// by the work-list length they had in the stream's PREVIOUS frame, longest first.  The weights pre-pass takes its
	// tiles in this order when [0] matches (a scheduling hint: results do not depend on it); the sweep plan kernel of
	// every frame rewrites it from the frame's own counts.
	uint32_t* tile_order = nullptr;
	int bands = 0;               // SGS_OPT_OUT_BANDS: > 1 = `out` is written band-major for that many image bands (sgs_band_of, sgs_device.h)
};
// gate: optional device word; when non-null the 128-channel-aligned kernels exit unless
// *gate != 0 (used as the arena-overflow fallback of the split path).
hipError_t launch_blend_forward(hipStream_t st, const BlendFwdArgs& a, int variant,
				const uint32_t* gate = nullptr, int c_skip = 0);

// ---- blend_fwd_split.hip (weights pre-pass + streaming accumulate)
struct SplitArena {   // byte offsets inside the arena chunk
	size_t counter, nbatches, table, act_id, act_idx, order, wgt, total;
	uint32_t capacity;   // work-list slots (1 KB of weights + 4 B id each)
};
// slot_bytes: 1536 = three bf16 terms per weight (the forward's default hand-over), 1024 = fp32 rows / two bf16 terms (the
// exact and two-term variants, and the backward's work list: its stream-ordered scratch is a third smaller for it)
size_t split_arena_bytes(uint32_t capacity, size_t L, int ntiles, SplitArena* lay, int slot_bytes = 1536);
// `mark` is called (with `mark_user`) between the weights pre-pass and the accumulate kernel, on `st`
// (stage timing).
// *usage_reported: the plan kernel wrote a.usage_host (no copy of the counter needed)
hipError_t launch_blend_forward_split(hipStream_t st, const BlendFwdArgs& a, char* arena,
				      const SplitArena& lay, void (*mark)(void*), void* mark_user,
				      int split_mode, bool* usage_reported = nullptr);

// ---- blend_weights2.hip: the weights pre-pass with a lane owning two pixels (mode 3 = fp32 rows, 4 = three bf16 terms)
hipError_t launch_blend_weights2(hipStream_t st, int mode, const uint2* ranges, const uint32_t* point_list,
				 const float2* means2D, const float4* conic_opacity, float* final_T, uint32_t* n_contrib,
				 uint32_t* act_id, uint32_t* act_idx, float* wgt, uint32_t* table, uint32_t* nact, uint32_t* counter,
				 uint32_t capacity, int W, int H, int gx, int ntiles, float* clear_ptr, size_t clear_floats,
				 const uint32_t* tile_order = nullptr);

// ---- blend_sweep2.hip: the accumulate sweep in fp32-class arithmetic (arith: 0 = exact fp32 MFMA, 1 = six bf16 products,
// 2 = the same on the x16 MFMA), LDS-polled DMA arrival, stores spread over the next tile; takes fp32 weight rows
// SGS_OPT_NORM_PLANE with nothing to blend: plane[i] = sum_c bg[c]^2
hipError_t launch_norm_plane_background(hipStream_t st, float* plane, size_t n, const float* bg, int C);
hipError_t launch_accum_sweep2(hipStream_t st, int arith, int dbg, const BlendFwdArgs& a, const uint32_t* table,
			       const uint32_t* nbatches, const uint32_t* act_id, const char* wgt, const uint32_t* counter,
			       int nc, int seg, int nseg, int pxcd, int items, unsigned long long* trace,
			       const uint32_t* order, int dealt);

// round 4: the same sweep as one 8-wave workgroup per (segment, 128 channels) for both row parities ("ping-pong", sweep
// nibble 6): items / pxcd count (segment, chunk) pairs, not (segment, chunk, parity) triples
hipError_t launch_accum_sweep3(hipStream_t st, int dbg, const BlendFwdArgs& a, const uint32_t* table,
			       const uint32_t* nbatches, const uint32_t* act_id, const char* wgt, const uint32_t* counter,
			       int nc, int seg, int nseg, int pxcd, int items, unsigned long long* trace,
			       const uint32_t* order, int dealt, int tune, int form = 0, int stp = 0);   // (a.bands is honoured by the x16 forms, tune == 1)   // form: 0 lock step, 1 fp32 hand-over (nibble 5), 2 free-running halves (nibble 4)

// debug: 4 x uint64 per sweep workgroup (begin, end on the 100 MHz steady counter, HW_ID | XCC_ID << 32, batches | tiles << 32)
void set_sweep_trace(void* device_words);
unsigned long long* get_sweep_trace();

// ---- blend_fused.hip: the C % 128 == 0 forward blend as one kernel (weights never leave the CU)
bool blend_forward_fused_eligible(const BlendFwdArgs& a);
hipError_t launch_blend_forward_fused(hipStream_t st, const BlendFwdArgs& a, bool exact, int seg_tiles);
// ---- blend_fused_pc.hip: one kernel, one producer wave (weights, once per strip) + C / 64 consumer waves per workgroup
bool blend_forward_fused_pc_eligible(const BlendFwdArgs& a);
hipError_t launch_blend_forward_fused_pc(hipStream_t st, const BlendFwdArgs& a, bool exact, int seg_tiles, int dbg = 0);

hipError_t launch_blend_weights_rows(hipStream_t st, const uint2* ranges, const uint32_t* point_list,
				     const float2* means2D, const float4* conic_opacity, float* final_T,
				     uint32_t* n_contrib, char* arena, const SplitArena& lay, int W, int H, int gx,
				     int gy, float* clear_ptr = nullptr, size_t clear_floats = 0,   // clear: optional buffer this kernel zero-fills (a multiple of 4 floats, 16-B aligned)
				     const uint32_t* tile_order = nullptr);

// ---- blend_bwd.hip
struct BlendBwdArgs {
	const uint2* ranges;
	const uint32_t* point_list;
	int W, H, C;
	int gx, gy;
	const float* bg;
	const float2* means2D;
	const float4* conic_opacity;
	const float* colors;         // (P,C)
	const float* final_T;
	const uint32_t* n_contrib;
	const float* dL_dpix;        // (C,H,W)
	float* dL_dmean2D;           // (P,3)
	float* dL_dconic;            // (P,4)
	float* dL_dopacity;          // (P)
	float* dL_dcolors;           // (P,C)
	const uint32_t* tile_order = nullptr;   // optional: [0] = tile count, [1..] = tiles longest-first (the stream's forward wrote it)
};
// `gate` (optional): device flag pair; the kernel runs only if gate[1] != 0 (fallback of the MFMA path).
hipError_t launch_blend_backward(hipStream_t st, const BlendBwdArgs& a, const uint32_t* gate = nullptr);
// ---- blend_bwd_mfma.hip: the backward blend for C >= 128, C % 32 == 0 as two matrix products over the
// forward's work list + a scalar recurrence (see the file header).  `arena` is scratch of
// split_arena_bytes(capacity, ...) bytes.
bool blend_backward_mfma_eligible(const BlendBwdArgs& a);
// the x16-MFMA kernels against the loaded code object (blend_sweep2.hip / blend_bwd_mfma.hip): 1 = they own their compute unit
bool x16_kernel_owns_cu(const void* fn, const char* name);
int sweep3_x16_ownership();
int bwd_fused_x16_ownership();
// fp32_products: the two channel products on v_mfma_f32_32x32x2_f32 (exact fp32 products) instead of split bf16
// clear_dcolor: a.dL_dcolors (P x C floats) is zero-filled by the first kernel instead of by the caller
// two_kernels: rounds 2-4's form (bwd_dcolor + bwd_dot, each streaming the gradient) instead of the fused kernel
hipError_t launch_blend_backward_mfma(hipStream_t st, const BlendBwdArgs& a, char* arena, const SplitArena& lay,
				      bool fp32_products, size_t clear_dcolor_floats = 0, bool two_kernels = false);
void launch_preprocess_bwd(hipStream_t st, int P, int D, int M, const float* means3D,
			   const int* radii, const float* shs, const uint8_t* clamped,
			   const float* scales, const float* rotations, float mod,
			   const float* cov3Ds, const float* view, const float* proj, float fx,
			   float fy, float tanx, float tany, const float* campos,
			   const float* dL_dmean2D, const float* dL_dconic, float* dL_dmeans,
			   const float* dL_dcolor, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
			   float* dL_drot);

// ---- composite.hip: the "over" composite of depth-ordered shard partials in one pass
constexpr int SGS_MAX_SHARDS = 16;
hipError_t launch_composite_over(hipStream_t st, int S, const float* const* A, const float* const* T, const float* bg,
				 float* out, float* t_out, int C, size_t npix);

// ---- knn.hip
size_t knn_scratch_bytes(int P);
hipError_t launch_knn(hipStream_t st, int P, const float* points, float* out, void* scratch,
		      size_t scratch_bytes);

// ---- misc
void launch_debug_expf(hipStream_t st, int n, const float* in, float* out);

// ---- fusion_map.hip (dataset/fusion_utils.py:30-78, fusion.py:127-147)
hipError_t launch_fusion_mapping(hipStream_t st, int N, const float* coords, const float* wvt,
				 const double intr[4], int W, int H, int cut, double vis_thres, int depth_mode,
				 const float* depth, double* zbuf, long long* mapping, double* weight);
hipError_t launch_fusion_accumulate(hipStream_t st, int N, int C, const float* feat_hwc, int W,
				    const long long* mapping, float* feat_sum, float* times);

} // namespace sgs
