/*
 * sgs_raster.h -- C-ABI of libsgs_hip.so, the MI355X (gfx950) N-channel
 * Gaussian-splat rasteriser + Morton kNN.
 *
 * This is the drop-in boundary for the hot path of sharinka0715/semantic-gaussians:
 * every entry point replaces one native entry point of the reference's three CUDA
 * extensions (CR = submodules/channel-rasterization, RR = submodules/rgbd-rasterization,
 * SK = submodules/simple-knn).  Plain pointers and sizes only -- no torch types.
 *
 * Conventions (same as the reference):
 *   - all float data is fp32, all pointers are DEVICE pointers on the GPU that
 *     `stream` belongs to, unless marked [host];
 *   - a NULL pointer for an optional input means "not provided"
 *     (reference: empty tensor -> nullptr, CR/cuda_rasterizer/forward.cu:205,241);
 *   - viewmatrix / projmatrix are the 16 floats of the reference's transposed
 *     matrices (scene/camera.py:87-93; CR/cuda_rasterizer/auxiliary.h:58-77);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  The
 *     reference launches on the legacy default stream; here every kernel, scan and
 *     sort is enqueued on `stream`;
 *   - functions return >= 0 on success and a negative SGS_E* code on failure;
 *     sgs_last_error() returns a thread-local message for the last failure.
 */
#ifndef SGS_RASTER_H_INCLUDED
#define SGS_RASTER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGS_ABI_VERSION 1

#define SGS_EINVAL (-1)   /* bad argument (reference: AT_ERROR / std::runtime_error) */
#define SGS_EHIP (-2)     /* HIP runtime / kernel error */
#define SGS_EALLOC (-3)   /* allocator callback returned NULL */
#define SGS_ETRAP (-4)    /* prefiltered=1 but a point was culled (reference: __trap(),
                             CR/cuda_rasterizer/auxiliary.h:156-160) */
#define SGS_ERETRY (-5)   /* sgs_forward_result: the deferred-count frame did not fit its capacity guess */
#define SGS_ENOTREADY (-6) /* sgs_forward_result(wait = 0): the counts have not arrived yet */

/* Resizable scratch buffer callback.  Replaces std::function<char*(size_t)>
 * (CR/cuda_rasterizer/rasterizer.h:31-33, built by resizeFunctional,
 * CR/rasterize_points.cu:28-36): must return a device pointer to at least `bytes`
 * bytes (128-B aligned) that stays valid until the caller releases it.  */
typedef void *(*sgs_alloc_fn)(void *user, size_t bytes);

int sgs_abi_version(void);
const char *sgs_last_error(void);

/* ---- forward ---------------------------------------------------------------
 * Replaces CudaRasterizer::Rasterizer::forward
 *   (CR/cuda_rasterizer/rasterizer_impl.cu:198-341, declared rasterizer.h:30-53;
 *    RR/cuda_rasterizer/rasterizer_impl.cu:198-339 when out_depth != NULL).
 * Pipeline (default, binning mode 0): preprocess -> depth presort of the P Gaussians (own LSD radix sort, four launches; its last
 * pass also writes the per-rank span counts, adds up their totals and -- deferred counts -- the count record) -> (8-byte D2H: row instances | num_rendered)
 * -> two span partitions that write every tile's depth-ordered list and `ranges` with no sort of the tile instances
 * -> blend (weights pre-pass + accumulate sweep for num_channels >= 128).  Binning mode 1 keeps the
 * reference's order of operations (inclusive scan -> 4-byte D2H -> duplicateWithKeys -> stable 64-bit radix sort
 * on bits [0, 32+msb(tiles)) -> tile ranges); lists, ranges and (reconstructed) keys are bit-identical in all modes.
 *   P,D,M          #Gaussians, active SH degree, SH coeffs per Gaussian (0 if no shs)
 *   background     (C) floats
 *   means3D (P,3) shs (P,M,3)|NULL colors_precomp (P,C)|NULL opacities (P)
 *   scales (P,3)|NULL rotations (P,4)|NULL cov3D_precomp (P,6)|NULL
 *   out_color      (C,H,W) written in full
 *   out_depth      (1,H,W) or NULL: the RGB-D median depth of RR/forward.cu:308,368-372,391
 *   radii          (P) int32, or NULL (internal)
 *   debug          !=0: launch errors are checked after every stage and the stream is synchronised and checked once at
 *                  the end of the call (the reference's CHECK_CUDA synchronises after EVERY stage,
 *                  CR/cuda_rasterizer/auxiliary.h:166-173; SGS_DEBUG_SYNC_EVERY_STAGE=1 in the environment does
 *                  the same -- the reference's own render_chn passes debug=True on every production frame)
 * Returns num_rendered (Sum of tiles touched) >= 0, or a negative error code.
 * Blocks the host once (the 8-byte read of the instance counts), like the reference (rasterizer_impl.cu:283) --
 * unless SGS_OPT_DEFER_COUNT is set on the stream (below). */
int sgs_rasterize_forward(
	sgs_alloc_fn geometry_buffer, void *geometry_user,
	sgs_alloc_fn binning_buffer, void *binning_user,
	sgs_alloc_fn image_buffer, void *image_user,
	int P, int D, int M,
	const float *background,
	int width, int height,
	const float *means3D,
	const float *shs,
	const float *colors_precomp,
	const float *opacities,
	const float *scales,
	float scale_modifier,
	const float *rotations,
	const float *cov3D_precomp,
	const float *viewmatrix,
	const float *projmatrix,
	const float *cam_pos,
	float tan_fovx, float tan_fovy,
	int prefiltered,
	int num_channels,
	float *out_color,
	float *out_depth,
	int *radii,
	int debug,
	void *stream);

/* ---- backward --------------------------------------------------------------
 * Replaces CudaRasterizer::Rasterizer::backward
 *   (CR/cuda_rasterizer/rasterizer_impl.cu:345-441, rasterizer.h:55-84), with the
 * colour-channel count a RUNTIME argument: the reference instantiates its backward on
 * the compile-time NUM_CHANNELS=3 (CR/cuda_rasterizer/config.h:15, backward.cu:599,636);
 * num_channels==3 reproduces it, other values are its runtime-C generalisation.
 *   R                         num_rendered returned by the forward
 *   geom/binning/image_buffer the three buffers the forward filled (opaque layout)
 *   dL_dpix                   (C,H,W)
 * Outputs must be zero-initialised by the caller (reference: torch::zeros,
 * CR/rasterize_points.cu:157-165):
 *   dL_dmean2D (P,3) dL_dconic (P,4) dL_dopacity (P) dL_dcolor (P,C) dL_dmean3D (P,3)
 *   dL_dcov3D (P,6) dL_dsh (P,M,3) dL_dscale (P,3) dL_drot (P,4) */
int sgs_rasterize_backward(
	int P, int D, int M, int R,
	const float *background,
	int width, int height,
	const float *means3D,
	const float *shs,
	const float *colors_precomp,
	const float *scales,
	float scale_modifier,
	const float *rotations,
	const float *cov3D_precomp,
	const float *viewmatrix,
	const float *projmatrix,
	const float *campos,
	float tan_fovx, float tan_fovy,
	const int *radii,
	char *geom_buffer,
	char *binning_buffer,
	char *image_buffer,
	const float *dL_dpix,
	int num_channels,
	float *dL_dmean2D,
	float *dL_dconic,
	float *dL_dopacity,
	float *dL_dcolor,
	float *dL_dmean3D,
	float *dL_dcov3D,
	float *dL_dsh,
	float *dL_dscale,
	float *dL_drot,
	int debug,
	void *stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible
 * (CR/cuda_rasterizer/rasterizer_impl.cu:141-153).  present: (P) bytes, 1 = view z > 0.2. */
int sgs_mark_visible(int P, const float *means3D, const float *viewmatrix,
		     const float *projmatrix, uint8_t *present, void *stream);

/* ---- Gaussian-sharded scenes (BASELINE config 5; no counterpart in the reference) ----------------------
 * Depth-ordered shard partials (A_s: the (C, rows, W) feature map of shard s rendered with a ZERO background, T_s: its
 * (rows, W) final transmittance -- what sgs_rasterize_forward + the image buffer's accum_alpha give) combine with the
 * associative, non-commutative "over" operator; this is the whole front-to-back chain in one pass over memory:
 *     out[c][px] = sum_s (prod_{s' < s} T_s'[px]) A_s[c][px] + (prod_s T_s[px]) background[c]
 *     T_out[px]  = prod_s T_s[px]                                                     (optional, may be NULL)
 * partial_A / partial_T are HOST arrays of num_shards (<= 16) device pointers, front-most shard first; every plane is
 * contiguous (rows * width floats) and 16-byte aligned; background may be NULL (no background term). */
int sgs_composite_over(int num_shards, const float *const *partial_A, const float *const *partial_T,
		       const float *background, float *out, float *T_out, int num_channels, int rows, int width,
		       void *stream);

/* Replaces SimpleKNN::knn (SK/simple_knn.cu:185-221, simple_knn.h:18): mean squared
 * distance to the three nearest other points.  points (P,3), meanDists (P).
 * Scratch comes from `scratch(scratch_user, bytes)` (one call). */
int sgs_knn_mean_dist2(int P, const float *points, float *meanDists,
		       sgs_alloc_fn scratch, void *scratch_user, void *stream);

/* ---- introspection of the opaque buffers (debug / parity tests) --------------
 * The reference's buffers are opaque to callers as well (only its own backward reads
 * them, CR/cuda_rasterizer/rasterizer_impl.cu:376-378); these accessors exist so the
 * bit-exactness of tile ids / sort keys can be tested without freezing a layout.
 * Each writes byte offsets (relative to the buffer start) of the named arrays. */
typedef struct {
	size_t depths;         /* float  [P]   view-space z */
	size_t clamped;        /* uint8  [3P]  SH clamp flags */
	size_t radii;          /* int32  [P]   internal radii (when caller passed NULL) */
	size_t means2D;        /* float2 [P]   pixel centre */
	size_t cov3D;          /* float  [6P]  */
	size_t conic_opacity;  /* float4 [P]   */
	size_t rgb;            /* float  [3P]  SH->RGB result */
	size_t tiles_touched;  /* uint32 [P]   */
	size_t point_offsets;  /* uint32 [P]   inclusive scan of tiles_touched */
	size_t total;          /* bytes required */
} sgs_geometry_layout;

typedef struct {
	size_t keys_unsorted;  /* uint64 [L] (tile<<32 | depth bits), emission order */
	size_t vals_unsorted;  /* uint32 [L] */
	size_t keys_sorted;    /* uint64 [L] */
	size_t point_list;     /* uint32 [L] sorted Gaussian ids */
	size_t total;
} sgs_binning_layout;

typedef struct {
	size_t accum_alpha;    /* float  [H*W] final T */
	size_t n_contrib;      /* uint32 [H*W] */
	size_t ranges;         /* uint2  [tiles] */
	size_t total;
} sgs_image_layout;

int sgs_geometry_layout_of(int P, sgs_geometry_layout *out);
int sgs_binning_layout_of(int num_rendered, sgs_binning_layout *out);
int sgs_image_layout_of(int width, int height, sgs_image_layout *out);

/* Number of key bits the forward sorts on: 32 + getHigherMsb(tiles)
 * (CR/cuda_rasterizer/rasterizer_impl.cu:35-50,302). */
int sgs_sort_bits(int width, int height);

/* The default binning never builds the reference's 64-bit sort keys (tile << 32 | depth bits);
 * this call materialises them, in sorted order, into the binning buffer's keys_sorted area from
 * `ranges`, `point_list` and the depths (parity tests).  Call it only on buffers of a forward
 * in binning mode 0 or 2: a mode-1 forward has written the real sorted keys already. */
int sgs_debug_sorted_keys(int P, int num_rendered, int width, int height,
			  const char *geom_buffer, char *binning_buffer,
			  const char *image_buffer, void *stream);

/* Device exp() used by the blend kernels, exposed for the numerics contract test
 * (DESIGN.md "exp contract"): out[i] = sgs_expf(in[i]). */
int sgs_debug_expf(int n, const float *in, float *out, void *stream);

/* Tuning / measurement hooks (not part of the reference's interface).
 *
 * State model.  The library keeps NO process-global mutable state on the data path: the adaptive work-list
 * capacities, their pinned feedback words and the counters below live in a context keyed by (device, stream),
 * created on first use and freed by sgs_stream_release().  The sgs_set_* functions set process-wide DEFAULTS of
 * the tuning options; sgs_stream_set_option() overrides one option for one stream, so two callers in one
 * process (different streams or devices) can run different arithmetic / binning / backward modes concurrently. */
#define SGS_OPT_BLEND_VARIANT 0
#define SGS_OPT_BINNING_MODE 1
#define SGS_OPT_BACKWARD_MODE 2
#define SGS_OPT_STAGE_TIMING 3
/* Row pitch, in pixels, of the out_color planes the NEXT forward on this stream writes (ONE call: the forward consumes
 * the override, so a pitch can never leak into a later call that passes a contiguous buffer): out_color is then
 * (C, H, pitch) floats and pixel (c, y, x) lives at out_color[(c * H + y) * pitch + x].  0 (default) = width
 * (contiguous (C,H,W), the reference's layout).  A pitch that is a multiple of 32 pixels makes every 16-pixel tile
 * pair a whole number of 128-byte lines whatever the image width is (BASELINE config 4: width 1297); the padding
 * columns hold unspecified values.  Must be >= width; ignored by the RGB-D variant's depth plane. */
#define SGS_OPT_OUT_PITCH 4
/* 1: deferred-count forwards on this stream (inference).  The reference's forward -- and this one by default --
 * blocks the host once per frame on a device-to-host copy of num_rendered (rasterizer_impl.cu:283), because the
 * binning buffer is sized from it.  With this option the buffers are sized from the stream's capacity guesses
 * (1.25 x what its previous frame needed; the first frame of a stream still blocks), the true counts stay on the
 * device, and a frame that does not fit aborts itself on the device.  sgs_rasterize_forward then returns at once
 * with the CAPACITY the binning buffer was laid out for; the caller MUST call sgs_forward_result() before it uses
 * the outputs AND before the next forward on the same stream (the context keeps one pending record per stream):
 * 0 = valid (and the true num_rendered), SGS_ERETRY = render that frame again (the guess has grown).
 * One host thread can so keep several streams full: 1M Gaussians x 512 channels, 4 views in flight:
 * see DESIGN.md 7.  Binning mode 0 only; under debug the frame is still synchronised once at the end of the call
 * (ignored altogether with SGS_DEBUG_SYNC_EVERY_STAGE=1); not for frames that will be differentiated (the
 * backward locates the lists from num_rendered).  2: as 1 with a capacity no frame fits (tests). */
#define SGS_OPT_DEFER_COUNT 5
/* 1: sgs_rasterize_backward clears dL_dcolor itself (the caller may pass uninitialised memory).  The reference's
 * binding hands over torch::zeros for every gradient (CR/rasterize_points.cu:156-164) and so must a caller by
 * default; for the N-channel gradient that fill is P * C * 4 bytes (2 GB at 1M x 512) of pure HBM writes in front
 * of the backward -- with this option it is folded into the backward's first kernel (the work-list pre-pass, which
 * is instruction bound and leaves the memory system idle): -0.2 ms at 1M x 512 x 968x1296. */
#define SGS_OPT_BWD_CLEARS_DCOLOR 6
/* 1: the NEXT forward on this stream (ONE call, like SGS_OPT_OUT_PITCH) renders the per-pixel squared L2 norm of the
 * feature map instead of the feature map: out_color is then ONE (H, pitch) plane of floats that receives
 * sum_c out[c]^2 (the library clears it; summation order over the channel groups is not fixed, so the last bits may
 * differ between runs).  This is the only quantity the reference's consumer needs from the full map beyond the
 * similarities themselves (eval_segmentation.py:155: rendering / (rendering.norm(dim=0) + 1e-8)); rendering it this
 * way saves the C * H * W * 4 bytes of stores (2.57 GB at 512 x 968 x 1296) and the caller's read of them.
 * Needs num_channels % 128 == 0, no depth plane and the default blend (variants 0 / 15); SGS_EINVAL otherwise.
 * A frame whose work list overflows is rendered by the same gated fallback as always; its epilogue adds the squares. */
#define SGS_OPT_NORM_PLANE 7
/* n > 1: the NEXT forward on this stream (ONE call, like SGS_OPT_OUT_PITCH) writes its feature map BAND-major for an image-partitioned
 * exchange between n ranks (Gaussian-sharded rendering, sgs_hip.dist.render_gaussian_sharded): band b = image rows [lo_b, hi_b),
 * lo_b = min(H, 16 * (ceil(H / 16) * b / n)), stored as a contiguous (num_channels, hi_b - lo_b, pitch) block, the blocks one behind the
 * other in out_color (band b starts num_channels * pitch * lo_b floats in; the buffer's size is unchanged).  Every band is then one
 * contiguous message -- no staging copy per peer.  Needs num_channels % 128 == 0, no depth plane, no SGS_OPT_NORM_PLANE and the default /
 * ping-pong sweep (variants 0, 6, word nibbles 4 / 6); SGS_EINVAL otherwise. */
#define SGS_OPT_OUT_BANDS 8
#define SGS_OPT_COUNT 9
/* value < 0 removes the override (the stream follows the process default again).  Returns the previous override,
 * or 0x7fffffff if there was none. */
int sgs_stream_set_option(void *stream, int option, int value);
#define SGS_STAT_ARENA_SLOTS 0     /* current work-list capacity of the split forward (slots of 1 KB) */
#define SGS_STAT_FWD_OVERFLOWS 1   /* split forwards whose work list overflowed (frame rendered by the gated fallback) */
#define SGS_STAT_BWD_OVERFLOWS 2   /* work-list backwards that overflowed (gradients by the per-chunk fallback) */
#define SGS_STAT_FORWARDS 3        /* forwards issued on this stream */
#define SGS_STAT_DEFERRED_FORWARDS 4   /* of those, deferred-count ones */
#define SGS_STAT_DEFERRED_RETRIES 5    /* deferred-count frames that did not fit (sgs_forward_result: SGS_ERETRY) */
#define SGS_STAT_BWD_POOL_FALLBACKS 6  /* work-list backwards that got no scratch from the stream-ordered pool and ran on the
                                         per-chunk kernel instead (the pool keeps up to SGS_BWD_POOL_RELEASE_MB, default
                                         4096, resident outside the caller's allocator) */
#define SGS_STAT_TILE_ORDER_ALLOC_FAILURES 7   /* forwards that could not get the stream's tile-order feedback buffer (4 B per tile): they
                                                  run without the longest-first schedule of the weights pre-pass -- same results */
#define SGS_STAT_COUNT 8
int sgs_stream_get_stat(void *stream, int stat, uint64_t *out);
/* The counts of the last forward on `stream` (see SGS_OPT_DEFER_COUNT): the deferred form of the reference's blocking
 * `cudaMemcpy(&num_rendered, geomState.point_offsets + P - 1, sizeof(int), cudaMemcpyDeviceToHost)`
 * (CR/cuda_rasterizer/rasterizer_impl.cu:283).  wait != 0: blocks until they have arrived
 * (that is after the frame's scan, long before its blend); wait == 0: SGS_ENOTREADY if they have not.  After an
 * ordinary forward: 0 and its num_rendered, immediately.  SGS_EINVAL if no forward was ever issued on (current
 * device, stream) -- contexts are keyed by the CURRENT device: call it under the device the forward ran on. */
int sgs_forward_result(void *stream, int wait, int *num_rendered);
/* Frees the context of (current device, stream); returns 1 if there was one.  Drains `stream` first (kernels already
 * enqueued may still write the context's pinned feedback words); calls in flight on other host threads keep the
 * context alive until they return. */
int sgs_stream_release(void *stream);

/* ---- compute-unit partitions (round 6; no counterpart in the reference, which runs one frame at a time on the null stream,
 * CR/cuda_rasterizer/rasterizer_impl.cu:198-341) ----------------------------------------------------------------------------
 * The accumulate sweep's workgroup (8 waves x 256 registers) can only start on a compute unit that holds nothing else; with several
 * views in flight the ~17 short, latency-bound front-end kernels of the other views keep a few waves on ALL compute units and hold
 * every sweep workgroup off for their duration.  These calls give the front end its own small share of the chip:
 *   sgs_device_cu_count            compute units of the current device
 *   sgs_stream_create_cu_range     a stream whose kernels only run on compute units [cu_first, cu_first + cu_count) of the current
 *                                  device (hipExtStreamCreateWithCUMask; the driver deals consecutive mask bits round-robin over the
 *                                  XCDs, so a range is spread evenly over them).  Destroy it with sgs_stream_destroy.
 *   sgs_stream_set_front           forwards issued on `stream` enqueue preprocess -> depth sort -> span partitions (and the count
 *                                  record / read-back) on `front_stream` and only the blend on `stream`; the two are ordered by events
 *                                  inside the call (front end behind the stream's earlier work, blend behind the front end), so the
 *                                  caller keeps addressing ONE stream.  front_stream NULL (or == stream): one stream again.
 *                                  The library keeps the pointer: detach (sgs_stream_set_front(stream, NULL)) or release / destroy `stream`
 *                                  BEFORE front_stream is destroyed.
 * Typical use (sgs_hip.raster.PartitionedStreams, bench.py --front-cus): per view slot a blend stream on CUs [f, n) and a front stream on [0, f).
 * Measured on MI355X: not a win for this pipeline -- the sweep slows in proportion to the compute units it loses. */
int sgs_device_cu_count(void);
int sgs_stream_create_cu_range(int cu_first, int cu_count, void **stream_out);
int sgs_stream_destroy(void *stream);
int sgs_stream_set_front(void *stream, void *front_stream);
/* The kernels that issue the double-rate MFMA (v_mfma_f32_32x32x16_bf16) must own their compute unit.  The build
 * checks their code objects (csrc/check_code_object.py, a step of `make`); this asks the runtime about the LOADED kernels on the
 * current device (hipFuncGetAttributes: 256 registers, 512 threads, > 80 KB LDS, no scratch), once per process:
 * bit 0 = the forward's ping-pong sweeps, bit 1 = the fused backward.  A kernel whose bit is clear is never launched: the x8 sweep /
 * the fp32-product backward (same interface, bit-identical / fp32-exact results) run in its place, with one line on stderr. */
int sgs_x16_cu_ownership(void);

/* Selects the forward blend kernels.  The product library (sgs_build_flags() == 0) knows:
 *   0  (default) num_channels >= 128: weights pre-pass + ping-pong row sweep for the 128-channel-aligned part in "f32-equivalent"
 *      arithmetic -- features and weights split EXACTLY into three bf16 terms each, the six products with i + j <= 4 on
 *      v_mfma_f32_32x32x16_bf16 (the sweep's workgroups own their compute units; bits [19:16] = 0 in the word form: on
 *      v_mfma_f32_32x32x8_bf16), fp32 accumulate: against the exact (float64) composite as accurate as the reference's fp32
 *      multiply-add chain (CR/cuda_rasterizer/forward.cu:355-356), every integer output bit-exact; the px1 kernel for the
 *      remainder channels, below 128 channels and for RGB-D;
 *   15 as 0 with the fp32-input MFMA sweep: the feature map is BIT-IDENTICAL to the contract's fp32 fma chain (SGS_BLEND_EXACT=1);
 *   14 round 2's arithmetic: two bf16 terms per operand, three products (<= 3 * 2^-16 of sum |f| w): the fastest, NOT fp32-class;
 *   6  the single-kernel px4 form for the 128-aligned part (the gated fallback of 0 when its work list overflows);
 *   >= 16, the word form:  bits [3:0] sweep (4 / 6 = 0's ping-pong sweep: 4 its free-running form, 6 its lock-step form; 11 = 15's,
 *      0 / 8 = 14's) | [7:4] segment length / 8 tiles (0 = adaptive)
 *      | [13:12] workgroup order (0 / 3 = segments sorted by work and dealt to the XCDs, 1 = row-major, 2 = dealt unsorted)
 *      | [19:16] with sweeps 4 / 6: 1 = on the x16 MFMA, 0 = on the x8 MFMA (sweep 6 only: round 4's default, bit-identical).
 *      | [21:20] with the word's sweep 4 on x16 only: where a finished tile pair's stores are issued (0 = rounds 3-5: blocks 0, 1 of both tiles at
 *      once, blocks 2, 3 in four chunks behind the next tile's steps; 1 = pixel block 1 in the next tile's first matrix phase) -- the same
 *      sums in the same order, bit-identical maps.
 *      What 0 selects is the word 0x110004 (free-running halves on x16, store placement 1; 0x10004 = round 5's default, 0x10006 = the same in
 *      lock step: all bit-identical).
 * Everything else -- sweep nibble 4 with bits [19:16] = 0, nibbles 5 / 7 / 9 / 10 / 13 / 14, ablation bits [11:8], pre-pass switches
 * [15:14], single-kernel forms 1-5 -- is a development form (`make EXPERIMENTS=1`; DESIGN.md 5.x has the measurements), 12 / 15 / [19:16] the
 * double-rate-MFMA reproducers (`make X16=1`), 32-35 round 2's fused kernels (`make FUSED=1`): SGS_EINVAL here.
 * Returns the previous value. */
int sgs_set_blend_variant(int variant);
/* Device time (ms, hipEvents on `stream`) of each stage of the forward.
 * stages: 0 preprocess 1 scan+readback 2 duplicate 3 sort 4 ranges 5 blend weights pre-pass
 *         6 blend accumulate (the whole blend on the single-kernel paths)
 * mode 0 off; 1 = resolve at the end of each call (adds a host sync per forward);
 * 2 = deferred: events are parked and sgs_get_stage_ms() returns the MEAN over all
 * forwards since the last query (no extra synchronisation inside the timed region).
 * sgs_get_stage_ms returns the number of forwards averaged (0 in mode 1). */
int sgs_set_stage_timing(int mode);
/* Binning algorithm (1 and 2 run on rocPRIM's scan / radix sort and are `make EXPERIMENTS=1` builds only since round 6 -- SGS_EINVAL in the
 * product library, which carries no library kernel): 0 (default) = Gaussians presorted by depth, per-tile lists built from row
 * instances without sorting the tile instances (binning_rows.hip; 3 is accepted as an alias); 1 = the reference's order of
 * operations (emit in index order, sort on all 32+msb(tiles) key bits); 2 = depth presort,
 * instances emitted in that order, stable radix sort on the 32-bit tile id.  Lists (point_list),
 * ranges and the (reconstructed) sorted keys are bit-identical in all modes; point_offsets and
 * the UNSORTED key/value arrays exist only in mode 1.  Returns the previous mode. */
int sgs_set_binning_mode(int mode);
/* Backward blend: 0 (default) = for num_channels >= 32 with num_channels % 32 == 0 the channel work runs as two matrix
 * products over the forward's work list, both in ONE kernel that reads dL/dpixel once (round 5, blend_bwd_mfma.hip
 * bwd_fused_kernel; scratch comes from a stream-ordered pool on `stream`), the per-chunk kernel otherwise.
 * Arithmetic of the two products in mode 0: every operand (features, weights, gradient) is split into TWO bf16 terms,
 * x = hi + lo + O(2^-16 x), the three products lo*hi + hi*lo + hi*hi run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
 * <= 3 * 2^-16 of sum |a||b| per product -- NARROWER than the forward's default (three exact terms, six products).
 * 3 = the same kernel with exact fp32 products (v_mfma_f32_32x32x2_f32), ~1.3 ms slower at 1M x 512 x 968x1296;
 * 1 = always the per-chunk VALU kernel (blend_bwd.hip); 2 = as 0 with a deliberately undersized work-list arena
 * (exercises the overflow fallback; tests only); 4 / 5 = rounds 2-4's form of 0 / 3 (one kernel per product, each
 * streaming the gradient; `make EXPERIMENTS=1` builds only, SGS_EINVAL otherwise).  All within 1e-4 of the largest gradient entry of the float64 oracle
 * (tests, also at the headline configuration's full size).  Returns the previous mode.
 * Mode 0 with more tiles than the device has compute units launches the kernel as PERSISTENT workgroups (round 6): one per compute unit, tiles by
 * ticket, the next tile's first loads requested in the current tile's tail; same arithmetic, same results up to the order of the colour gradient's
 * atomics.  SGS_BWD_PERSIST=0 in the environment (read once, at load) keeps one workgroup per tile. */
int sgs_set_backward_mode(int mode);
/* Which optional parts this libsgs_hip.so was built with: bit 0 `make FUSED=1` (blend variants 32-35), bit 1 `make X16=1` (the
 * double-rate-MFMA reproducers), bit 2 `make EXPERIMENTS=1` (development forms of the blend kernels: ablations, superseded sweeps and
 * pre-passes, the two-kernel backward).  The product build returns 0 and answers those variants with SGS_EINVAL. */
int sgs_build_flags(void);
int sgs_get_stage_ms(float *ms7);
/* The depth presort of the forward (csrc/depth_sort.hip) on a bare array of 32-bit keys: perm[r] = index of the
 * r-th smallest key, equal keys in index order (tests).  With keys / perm / scratch NULL: returns the scratch
 * bytes needed for P keys. */
long long sgs_debug_depth_sort(int P, const unsigned *keys, unsigned *perm, void *scratch, void *stream);
/* Debug (tools/sweep_trace.py): device buffer of 4 x uint64 per workgroup of the accumulate sweep, filled by the
 * next forwards with (begin, end) on the 100 MHz steady counter, HW_ID | XCC_ID << 32, batches | tiles << 32;
 * NULL switches the trace off.  Process-wide, not for production use. */
void sgs_debug_set_sweep_trace(void *device_words);

/* ---- 2-D -> 3-D fusion step (the callers either side of the depth render) ----
 *
 * sgs_fusion_compute_mapping replaces PointCloudToImageMapper.compute_mapping
 * (dataset/fusion_utils.py:30-78), which the reference evaluates in NumPy on the host once per view
 * (fusion.py:127-133) after copying the Gaussian centres and the rendered depth off the device.
 *   coords                (N,3) float32, device: Gaussian centres
 *   world_view_transform  16 float32, device: the view's matrix exactly as the reference holds it
 *                         (the TRANSPOSED world-to-camera matrix; it is applied transposed, :45)
 *   intrinsics4           HOST doubles fx, fy, cx, cy AFTER the mapper's constructor adjustment (:22-28)
 *   depth_mode            0 = no depth (front test z > 0, :71-73); 1 = `depth` is the (H,W) float32 map
 *                         (occlusion test |d - z| <= vis_thres * d, :63-69); 2 = "surface": the z-buffer of
 *                         the points themselves (:57-62), built in `zbuf` ((H,W) float64 scratch)
 *   mapping               (N,3) int64 out: (y, x, 1) for visible points, (0,0,0) otherwise
 *   weight                (N) float64 out: exp(-distance of the pixel from the image centre / 10) (:77)
 * Arithmetic is float64 in the reference's operation order; the integer outputs are the reference's.
 *
 * sgs_fusion_accumulate is the per-view accumulation of fuse_one_scene (fusion.py:139-147): for every
 * visible point  feat_sum[i,:] += features[y,x,:]  and  times[i] += 1.  `features_hwc` is the 2-D feature map
 * channel-LAST, (H,W,C) float32 (the reference indexes a (C,H,W) map on the host); fp32 adds, one per
 * point and channel, so the sums are the reference's bit for bit. */
int sgs_fusion_compute_mapping(int N, const float *coords, const float *world_view_transform,
                               const double *intrinsics4, int image_w, int image_h, int cut_bound,
                               double vis_thres, int depth_mode, const float *depth, double *zbuf,
                               long long *mapping, double *weight, void *stream);
int sgs_fusion_accumulate(int N, int C, const float *features_hwc, int image_w, int image_h,
                          const long long *mapping, float *feat_sum, float *times, void *stream);

#ifdef __cplusplus
}
#endif
#endif
