// blend_fwd.hip -- N-channel front-to-back alpha-composite (forward) for gfx950.
//
// Behaviour restated from CR/cuda_rasterizer/forward.cu:262-375 (runtime C) and
// RR/cuda_rasterizer/forward.cu:261-393 (C=3 + median depth), see SURVEY.md A.4.
//
// MI355X design (DESIGN.md "blend forward").  Two kernels:
//
//  blend_fwd_px1_kernel  (any C; also the RGB-D path)
//    one 16x16 tile x one chunk of CC channels per 256-lane workgroup (4 wave64);
//    lane = pixel, a wave owns a 16x4 strip, the CC accumulators live in VGPRs (the
//    reference keeps a per-thread `float C[768]` in scratch memory).  "This Gaussian
//    touches none of my 64 pixels" is one ballot -> the wave skips the feature fetch and
//    the CC FMAs.  The Gaussian's feature row is wave-uniform, so it is fetched with SCALAR
//    loads (s_load_dwordx16 through the scalar cache) and consumed as the SGPR-pair operand
//    of v_pk_fma_f32 -- no LDS traffic and no VGPRs for features.
//
//  blend_fwd_px4_kernel  (C a multiple of 4*CW; the C=512 headline path)
//    one tile x 4*CW channels per workgroup.  The per-(pixel,Gaussian) weights alpha*T are
//    computed ONCE per workgroup (wave w evaluates strip w, one pixel per lane) and parked in
//    LDS; then each wave accumulates a different CW-channel slice for ALL 256 pixels, four
//    pixels per lane: every scalar-loaded feature pair feeds four v_pk_fma_f32
//    (measured 143 TFLOP/s form on MI355X, tools/ubench_fma.hip), the weight evaluation is
//    shared by 4*CW channels instead of CC, and each feature byte is fetched exactly once per
//    tile.
//
// Common: tile-list entries (id, xy, conic, opacity) are staged in LDS by one coalesced
// gather per batch; id*C is computed in 64 bit (the reference overflows int at 5M x 768);
// blockIdx -> (tile, chunk) is XCD-aware (block b runs on XCD b % 8): each XCD owns a
// contiguous band of tiles so neighbouring tiles re-read shared feature rows from that
// XCD's L2.
#include "sgs_kernels.h"

namespace sgs {

typedef float f2 __attribute__((ext_vector_type(2)));

struct StagedEntry {   // 32 B per list entry in LDS
	float a2, b2, c2, o;   // -0.5*conic.x, -conic.y, -0.5*conic.z, opacity
	float x, y;            // pixel centre
	uint32_t id;
	float depth;
};

__device__ __forceinline__ StagedEntry stage_entry(uint32_t id, const float2* __restrict__ means2D,
						   const float4* __restrict__ conic_opacity,
						   const float* __restrict__ depths, bool want_depth)
{
	const float2 xy = means2D[id];
	const float4 co = conic_opacity[id];
	StagedEntry e;
	e.a2 = -0.5f * co.x;   // exact scalings: the contract's power uses a2,b2,c2
	e.b2 = -co.y;
	e.c2 = -0.5f * co.z;
	e.o = co.w;
	e.x = xy.x;
	e.y = xy.y;
	e.id = id;
	e.depth = want_depth ? depths[id] : 0.f;
	return e;
}

// Exact per-(pixel, Gaussian) evaluation shared by all kernels.
//   power = fma(b2*dx, dy, fma(c2*dy, dy, (a2*dx)*dx));  skip if power > 0
//   alpha = min(0.99, o*exp(power));                     skip if alpha < 1/255
//   test_T = T*(1-alpha);                                stop pixel if test_T < 1e-4
// Returns `take`; updates `done`.
__device__ __forceinline__ bool eval_pixel(const StagedEntry& e, float pxf, float pyf, float T,
					   bool& done, float& alpha, float& test_T)
{
	const float dx = e.x - pxf, dy = e.y - pyf;
	const float power =
		__builtin_fmaf(e.b2 * dx, dy, __builtin_fmaf(e.c2 * dy, dy, (e.a2 * dx) * dx));
	alpha = fmin_(0.99f, e.o * expf_contract(power));
	test_T = T * (1.0f - alpha);
	const bool cand = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
	const bool stop = cand && (test_T < 0.0001f);
	done = done || stop;
	return cand && !stop;
}

// -------------------------------------------------------------------------------------
// px1: lane = pixel, CC channels per workgroup.
//   FULL : every chunk has exactly CC channels (no per-channel bounds checks)
//   XCD  : XCD-aware block map (grid is padded to 8*per_xcd)
template <int CC, bool DEPTH, bool FULL>
__global__ __launch_bounds__(256) void blend_fwd_px1_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float2* __restrict__ means2D, const float* __restrict__ features,
	const float4* __restrict__ conic_opacity, const float* __restrict__ depths,
	const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
	float* __restrict__ out, float* __restrict__ out_depth, int W, int H, int C, int gx,
	int c_begin, int nchunks, int write_aux, int per_xcd, int total, int pitch, const uint32_t* __restrict__ abort)
{
	if (abort && *abort != 0u) return;   // deferred-count forward whose capacity guess was too small (capi.hip)
	const int b = blockIdx.x;
	const int v = (b & 7) * per_xcd + (b >> 3);
	if (v >= total) return;
	const int tile = v / nchunks;
	const int chunk = v - tile * nchunks;
	const int c0 = c_begin + chunk * CC;
	const int cn = FULL ? CC : ((C - c0) < CC ? (C - c0) : CC);
	const int tx = tile % gx, ty = tile / gx;

	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
	const int px = tx * SGS_TILE + (lane & 15);
	const int py = ty * SGS_TILE + wave * 4 + (lane >> 4);
	const bool inside = px < W && py < H;
	const float pxf = (float)px, pyf = (float)py;
	const size_t HW = (size_t)H * pitch;   // channel plane stride of the (possibly row-padded) output

	const uint2 range = ranges[tile];
	const int n_total = (int)(range.y - range.x);

	__shared__ StagedEntry s_e[256];
	__shared__ int s_alive[4];

	float acc[CC];
#pragma unroll
	for (int c = 0; c < CC; c++) acc[c] = 0.f;
	float T = 1.0f;
	float D = 15.0f;   // RR/forward.cu:308
	uint32_t last = 0;
	bool done = !inside;

	for (int base = 0; base < n_total; base += 256) {
		const bool wave_alive = __ballot(!done) != 0ull;
		if (lane == 0) s_alive[wave] = wave_alive ? 1 : 0;
		__syncthreads();
		const int alive = s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3];
		if (!alive) break;   // every pixel of the tile is done
		const int n = (n_total - base) < 256 ? (n_total - base) : 256;
		if ((int)threadIdx.x < n)
			s_e[threadIdx.x] = stage_entry(point_list[range.x + base + threadIdx.x], means2D,
						       conic_opacity, depths, DEPTH);
		__syncthreads();
		if (wave_alive) {
			for (int j = 0; j < n; j++) {
				const StagedEntry e = s_e[j];
				float alpha, test_T;
				const bool take = eval_pixel(e, pxf, pyf, T, done, alpha, test_T);
				if (__ballot(take) != 0ull) {
					const float w = take ? alpha * T : 0.0f;
					const uint32_t id = __builtin_amdgcn_readfirstlane(e.id);
					const float* __restrict__ f = features + (size_t)id * C + c0;
#pragma unroll
					for (int c = 0; c < CC; c++)
						if (FULL || c < cn) acc[c] = __builtin_fmaf(f[c], w, acc[c]);
					if (DEPTH) {
						if (take && T > 0.5f && test_T < 0.5f) D = e.depth;
					}
					if (take) {
						T = test_T;
						last = (uint32_t)(base + j + 1);
					}
				}
				if (__ballot(!done) == 0ull) break;
			}
		}
	}

	if (inside) {
		const size_t pix = (size_t)py * W + px;
		if (write_aux && chunk == 0) {
			final_T[pix] = T;
			n_contrib[pix] = last;
			if (DEPTH) out_depth[pix] = D;
		}
#pragma unroll
		for (int c = 0; c < CC; c++)
			if (FULL || c < cn) out[(size_t)(c0 + c) * HW + (size_t)py * pitch + px] = __builtin_fmaf(T, bg[c0 + c], acc[c]);
	}
}

// -------------------------------------------------------------------------------------
// px4: weights shared through LDS, 4 pixels per lane, CW channels per wave.
//
// LDS per workgroup: BATCH staged entries (32 B each) + BATCH x 256 weights (4 B each),
// weight layout [entry][lane][strip] so that lane l reads its four pixels
// (strip 0..3, position l) with one ds_read_b128.
template <int CW, int BATCH>
__global__ __launch_bounds__(256) void blend_fwd_px4_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float2* __restrict__ means2D, const float* __restrict__ features,
	const float4* __restrict__ conic_opacity, const float* __restrict__ bg,
	float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out, int W,
	int H, int C, int gx, int nchunks, int per_xcd, int total, const uint32_t* __restrict__ gate, int pitch,
	const uint32_t* __restrict__ abort, int norm, int bands)
{
	if (gate && gate[1] != 1u) return;   // fallback instance: runs only if the split path's work list overflowed (1; 2 = aborted frame)
	if (abort && *abort != 0u) return;
	__shared__ StagedEntry s_e[BATCH];
	__shared__ float4 s_w[BATCH * 64];          // [entry][lane] -> (strip0..3)
	__shared__ unsigned s_active[BATCH];        // per entry: any pixel of the tile takes it
	__shared__ int s_alive[4];
	// The gated fallback instance is launched with a SMALL grid whose workgroups stride over the items: when the gate
	// is closed (every frame but an overflowing one) only that many workgroups have to start and exit.
	auto one_item = [&](const int b) __attribute__((always_inline)) {
	const int v = (b & 7) * per_xcd + (b >> 3);
	if (v >= total) return;
	const int tile = v / nchunks;
	const int chunk = v - tile * nchunks;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
	const int c0 = (chunk * 4 + wave) * CW;   // this wave's channel slice
	const int tx = tile % gx, ty = tile / gx;
	const size_t HW = (size_t)H * pitch;   // channel plane stride of the (possibly row-padded) output

	// weight-phase pixel (strip = wave)
	const int px = tx * SGS_TILE + (lane & 15);
	const int py = ty * SGS_TILE + wave * 4 + (lane >> 4);
	const bool inside = px < W && py < H;
	const float pxf = (float)px, pyf = (float)py;

	const uint2 range = ranges[tile];
	const int n_total = (int)(range.y - range.x);

	f2 acc[4][CW / 2];   // packed pairs: v_pk_fma_f32 with an SGPR-pair feature operand
#pragma unroll
	for (int p = 0; p < 4; p++)
#pragma unroll
		for (int c = 0; c < CW / 2; c++) acc[p][c] = (f2){0.f, 0.f};
	float T = 1.0f;
	uint32_t last = 0;
	bool done = !inside;

	for (int base = 0; base < n_total; base += BATCH) {
		const bool wave_alive = __ballot(!done) != 0ull;
		if (lane == 0) s_alive[wave] = wave_alive ? 1 : 0;
		__syncthreads();   // also fences the previous batch's reads of s_w / s_e / s_active
		const int alive = s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3];
		if (!alive) break;
		if (threadIdx.x < BATCH) s_active[threadIdx.x] = 0u;
		const int n = (n_total - base) < BATCH ? (n_total - base) : BATCH;
		if ((int)threadIdx.x < n)
			s_e[threadIdx.x] = stage_entry(point_list[range.x + base + threadIdx.x], means2D,
						       conic_opacity, nullptr, false);
		__syncthreads();
		// ---- weight phase: wave w evaluates strip w for the whole batch
		float* s_wf = reinterpret_cast<float*>(s_w);
		for (int j = 0; j < n; j++) {
			float w = 0.0f;
			if (wave_alive) {
				const StagedEntry e = s_e[j];
				float alpha, test_T;
				const bool take = eval_pixel(e, pxf, pyf, T, done, alpha, test_T);
				if (take) {
					w = alpha * T;
					T = test_T;
					last = (uint32_t)(base + j + 1);
				}
				const bool any_take = __ballot(take) != 0ull;
				if (lane == 0 && any_take) s_active[j] = 1u;   // benign same-value race
			}
			s_wf[(j * 64 + lane) * 4 + wave] = w;
		}
		__syncthreads();
		// ---- accumulate phase: 4 pixels per lane, CW channels per wave
		for (int j = 0; j < n; j++) {
			if (s_active[j] == 0u) continue;
			const float4 w4 = s_w[j * 64 + lane];
			const uint32_t id = __builtin_amdgcn_readfirstlane(s_e[j].id);
			const f2* __restrict__ f = reinterpret_cast<const f2*>(features + (size_t)id * C + c0);
			const f2 w0 = {w4.x, w4.x}, w1 = {w4.y, w4.y}, w2 = {w4.z, w4.z}, w3 = {w4.w, w4.w};
#pragma unroll
			for (int c = 0; c < CW / 2; c++) {
				const f2 fv = f[c];
				acc[0][c] = __builtin_elementwise_fma(fv, w0, acc[0][c]);
				acc[1][c] = __builtin_elementwise_fma(fv, w1, acc[1][c]);
				acc[2][c] = __builtin_elementwise_fma(fv, w2, acc[2][c]);
				acc[3][c] = __builtin_elementwise_fma(fv, w3, acc[3][c]);
			}
		}
	}

	// ---- epilogue.  T of the accumulate-phase pixels comes from their owner lanes via LDS.
	__syncthreads();
	float* s_T = reinterpret_cast<float*>(s_w);   // [strip][pos]
	s_T[wave * 64 + lane] = T;
	if (inside && chunk == 0) {
		const size_t pix = (size_t)py * W + px;
		final_T[pix] = T;
		n_contrib[pix] = last;
	}
	__syncthreads();
	// SGS_OPT_OUT_BANDS: the tile row's band is a (C, rows, pitch) image of its own
	float* outb = out;
	size_t HWb = HW;
	int row0 = 0;
	if (bands > 1) {
		int lo_tile, rows;
		sgs_band_of(ty, (H + SGS_TILE - 1) / SGS_TILE, bands, H, lo_tile, rows);
		row0 = SGS_TILE * lo_tile;
		outb = out + (size_t)C * (size_t)pitch * (size_t)row0;
		HWb = (size_t)rows * pitch;
	}
#pragma unroll
	for (int p = 0; p < 4; p++) {
		const int qx = tx * SGS_TILE + (lane & 15);
		const int qy = ty * SGS_TILE + p * 4 + (lane >> 4);
		if (qx < W && qy < H) {
			const float Tp = s_T[p * 64 + lane];
			const size_t pix = (size_t)(qy - row0) * pitch + qx;
			if (norm) {   // SGS_OPT_NORM_PLANE: `out` is one (H, pitch) plane that receives sum_c out[c]^2 (blend_sweep2.hip)
				float ss = 0.f;
#pragma unroll
				for (int c = 0; c < CW; c++) {
					const float v = __builtin_fmaf(Tp, bg[c0 + c], acc[p][c >> 1][c & 1]);
					ss = __builtin_fmaf(v, v, ss);
				}
				atomicAdd(out + pix, ss);
				continue;
			}
#pragma unroll
			for (int c = 0; c < CW; c++)
				outb[(size_t)(c0 + c) * HWb + pix] = __builtin_fmaf(Tp, bg[c0 + c], acc[p][c >> 1][c & 1]);
		}
	}
	};
	for (int b = blockIdx.x; b < 8 * per_xcd; b += gridDim.x) {
		one_item(b);
		__syncthreads();   // the next item reuses the shared arrays
	}
}

// -------------------------------------------------------------------------------------
template <int CC, bool DEPTH, bool FULL>
static void launch_px1(hipStream_t st, const BlendFwdArgs& a, int c_begin, int nchunks,
		       int write_aux)
{
	const int total = a.gx * a.gy * nchunks;
	const int per_xcd = (total + 7) / 8;
	hipLaunchKernelGGL((blend_fwd_px1_kernel<CC, DEPTH, FULL>), dim3(per_xcd * 8), dim3(256), 0, st,
			   a.ranges, a.point_list, a.means2D, a.features, a.conic_opacity, a.depths,
			   a.bg, a.final_T, a.n_contrib, a.out, a.out_depth, a.W, a.H, a.C, a.gx, c_begin,
			   nchunks, write_aux, per_xcd, total, a.pitch, a.abort);
}

// workgroups of the GATED instance (it exits at once unless the split path's work list overflowed): every one of them
// has to be scheduled -- behind other views' sweeps with several views in flight -- before the stream moves on
#ifndef SGS_GATED_GRID
#define SGS_GATED_GRID 2048
#endif
template <int CW, int BATCH>
static void launch_px4(hipStream_t st, const BlendFwdArgs& a, int nchunks, const uint32_t* gate)
{
	const int total = a.gx * a.gy * nchunks;
	const int per_xcd = (total + 7) / 8;
	const int grid = gate && per_xcd * 8 > SGS_GATED_GRID ? SGS_GATED_GRID : per_xcd * 8;   // (gated fallback: see the kernel)
	hipLaunchKernelGGL((blend_fwd_px4_kernel<CW, BATCH>), dim3(grid), dim3(256), 0, st,
			   a.ranges, a.point_list, a.means2D, a.features, a.conic_opacity, a.bg,
			   a.final_T, a.n_contrib, a.out, a.W, a.H, a.C, a.gx, nchunks, per_xcd, total, gate, a.pitch, a.abort, a.norm_plane ? 1 : 0, a.bands);
}

// variant: 0 = default (px4 CW=32 for the 128-channel-aligned part, px1 for the rest)
//          1 = px1 CC=64   2 = px1 CC=128   3 = px1 CC=32   4 = px4 CW=32 BATCH=64
//          5 = px4 CW=16
hipError_t launch_blend_forward(hipStream_t st, const BlendFwdArgs& a, int variant,
				const uint32_t* gate, int c_skip)
{
	const int ntiles = a.gx * a.gy;
	if (ntiles == 0 || a.C == 0) return hipSuccess;
	const bool depth = a.out_depth != nullptr;
	int c_done = 0;
	if (c_skip > 0 && !gate) {
		c_done = c_skip;   // channels [0, c_skip) were rendered by the split path
	} else if (!depth) {
#ifndef SGS_WITH_EXPERIMENTS   // the product library: the 128-channel px4 form (variant 0 / 6; the gated fallback of the split path); 1-5 are make EXPERIMENTS=1
		if (variant != 0) return hipErrorInvalidValue;
		{
			const int nch = a.C / 128;
			if (nch > 0) {
				launch_px4<32, 32>(st, a, nch, gate);
				c_done = nch * 128;
			}
		}
#else
		if (variant == 0 || variant == 4 || variant == 5) {
			const int width = (variant == 5) ? 64 : 128;
			const int nch = a.C / width;
			if (nch > 0) {
				if (variant == 0) launch_px4<32, 32>(st, a, nch, gate);
				else if (variant == 4) launch_px4<32, 64>(st, a, nch, nullptr);
				else launch_px4<16, 32>(st, a, nch, nullptr);
				c_done = nch * width;
			}
		} else {
			const int width = (variant == 2) ? 128 : (variant == 3 ? 32 : 64);
			const int nch = a.C / width;
			if (nch > 0) {
				if (width == 128) launch_px1<128, false, true>(st, a, 0, nch, 1);
				else if (width == 64) launch_px1<64, false, true>(st, a, 0, nch, 1);
				else launch_px1<32, false, true>(st, a, 0, nch, 1);
				c_done = nch * width;
			}
		}
#endif
	}
	if (gate) return hipGetLastError();   // gated call renders only the 128-aligned part
	const int rem = a.C - c_done;
	if (rem > 0) {
		const int write_aux = (c_done == 0) ? 1 : 0;
		if (rem <= 4) {
			if (depth) launch_px1<4, true, false>(st, a, c_done, 1, write_aux);
			else launch_px1<4, false, false>(st, a, c_done, 1, write_aux);
		} else {
			const int nch = (rem + 31) / 32;
			if (depth) launch_px1<32, true, false>(st, a, c_done, nch, write_aux);
			else launch_px1<32, false, false>(st, a, c_done, nch, write_aux);
		}
	}
	return hipGetLastError();
}

__global__ void debug_expf_kernel(int n, const float* __restrict__ in, float* __restrict__ out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = expf_contract(in[i]);
}

void launch_debug_expf(hipStream_t st, int n, const float* in, float* out)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(debug_expf_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, in, out);
}

} // namespace sgs
