// blend_bwd.hip -- backward of the alpha-composite (runtime channel count) and of the
// per-Gaussian preprocess, for gfx950.
//
// Behaviour restated from CR/cuda_rasterizer/backward.cu:394-552 (renderCUDA<C> backward),
// :141-271 (computeCov2DCUDA), :341-391 (preprocessCUDA backward), :275-336 (computeCov3D
// backward), :20-136 (computeColorFromSH backward); see SURVEY.md A.5.
//
// The reference instantiates its blend backward on the compile-time NUM_CHANNELS = 3
// (config.h:15).  Here the channel count is a runtime argument; the colour state is
// processed in chunks of CC channels, one (tile, chunk) per workgroup.  Every geometry
// gradient is linear in dL/dalpha and dL/dalpha is a plain sum over channels, so each chunk
// adds its partial contribution through the same atomics with no cross-chunk ordering.
//
// MI355X notes:
//  * lane = pixel, wave = 16x4 strip (as in the forward); the back-to-front walk starts at
//    the strip's largest n_contrib, so whole batches behind it are skipped;
//  * gradients are reduced over the 64 lanes of the wave with DPP adds before ONE atomic
//    per (wave, Gaussian, component) -- the reference issues one atomic per pixel;
//  * `last_color` of the reference is folded into the recurrence
//    rec <- alpha*c + (1-alpha)*rec evaluated at the end of the step (same values, no
//    second C-sized register array).
#include "sgs_kernels.h"

namespace sgs {

struct StagedEntryB {
	float a2, b2, c2, o;
	float x, y;
	uint32_t id;
	float ca;   // conic.x
	float cb, cc;
	float pad0, pad1;
};

// one (tile, channel chunk) of the walk: the whole workgroup
template <int CC>
__device__ __forceinline__ void blend_bwd_block(
	const int blk, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float* __restrict__ bg, const float2* __restrict__ means2D,
	const float4* __restrict__ conic_opacity, const float* __restrict__ colors,
	const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
	const float* __restrict__ dL_dpixels, float* __restrict__ dL_dmean2D,
	float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
	int W, int H, int C, int gx, int nchunks)
{
	const int tile = blk / nchunks;
	const int chunk = blk - tile * nchunks;
	const int c0 = chunk * CC;
	const int cn = (C - c0) < CC ? (C - c0) : CC;
	const int tx = tile % gx, ty = tile / gx;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int px = tx * SGS_TILE + (lane & 15);
	const int py = ty * SGS_TILE + wave * 4 + (lane >> 4);
	const bool inside = px < W && py < H;
	const float pxf = (float)px, pyf = (float)py;
	const size_t HW = (size_t)H * W;
	const size_t pix = (size_t)py * W + px;
	const uint2 range = ranges[tile];

	__shared__ StagedEntryB s_e[256];
	__shared__ int s_max[4];
	// CC <= 4 (RGB / RGB-D): the wave-reduced sums of an entry (CC colour + 6 geometry components) of the four waves
	// meet here and leave as one atomic per (entry, component), issued 64 lanes wide, after every 64 entries -- 9 x 4
	// lane-0 device-scope atomics per entry were what bound the kernel.  CC = 32: the colour sums leave as one
	// coalesced row of atomics per wave (lane c carries channel c) instead of 32 single-lane ones.
	constexpr bool COMBINE = CC <= 4;
	constexpr int NV = CC + 6;
	__shared__ float s_acc[COMBINE ? 4 * 64 * NV : 1];

	const float T_final = inside ? final_Ts[pix] : 0.f;
	float T = T_final;
	const int last_contributor = inside ? (int)n_contrib[pix] : 0;

	float rec[CC], g[CC];
	float bg_dot = 0.f;
#pragma unroll
	for (int c = 0; c < CC; c++) {
		rec[c] = 0.f;
		g[c] = (inside && c < cn) ? dL_dpixels[(size_t)(c0 + c) * HW + pix] : 0.f;
	}
#pragma unroll
	for (int c = 0; c < CC; c++)
		if (c < cn) bg_dot += bg[c0 + c] * g[c];

	const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

	// strip-level and tile-level starting points of the back-to-front walk
	int wave_max = last_contributor;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const int o = __shfl_xor(wave_max, off);
		wave_max = o > wave_max ? o : wave_max;
	}
	if (lane == 0) s_max[wave] = wave_max;
	__syncthreads();
	int tile_max = s_max[0];
	tile_max = s_max[1] > tile_max ? s_max[1] : tile_max;
	tile_max = s_max[2] > tile_max ? s_max[2] : tile_max;
	tile_max = s_max[3] > tile_max ? s_max[3] : tile_max;

	// entries [0, tile_max) are walked from tile_max-1 down to 0 in batches of 256
	for (int hi = tile_max; hi > 0; hi -= 256) {
		const int n = hi < 256 ? hi : 256;   // this batch covers indices [hi-n, hi)
		__syncthreads();
		if ((int)threadIdx.x < n) {
			// slot k holds entry index hi-1-k (descending)
			const uint32_t id = point_list[range.x + (uint32_t)(hi - 1 - (int)threadIdx.x)];
			const float2 xy = means2D[id];
			const float4 co = conic_opacity[id];
			StagedEntryB e;
			e.a2 = -0.5f * co.x;
			e.b2 = -co.y;
			e.c2 = -0.5f * co.z;
			e.o = co.w;
			e.x = xy.x;
			e.y = xy.y;
			e.id = id;
			e.ca = co.x;
			e.cb = co.y;
			e.cc = co.z;
			e.pad0 = e.pad1 = 0.f;
			s_e[threadIdx.x] = e;
		}
		__syncthreads();
		for (int kb = 0; kb < n; kb += 64) {
		const int ke = kb + 64 < n ? kb + 64 : n;
		if (COMBINE) {
			for (int q = threadIdx.x; q < 4 * 64 * NV; q += 256) s_acc[q] = 0.f;
			__syncthreads();
		}
		for (int k = kb; k < ke; k++) {
			const int idx = hi - 1 - k;          // 0-based list index of this entry
			if (idx >= wave_max) continue;       // nobody in this strip got that far
			const StagedEntryB e = s_e[k];
			const float dx = e.x - pxf, dy = e.y - pyf;
			const float power =
				__builtin_fmaf(e.b2 * dx, dy, __builtin_fmaf(e.c2 * dy, dy, (e.a2 * dx) * dx));
			const float G = expf_contract(power);
			const float alpha = fmin_(0.99f, e.o * G);
			const bool valid = inside && (idx < last_contributor) && !(power > 0.0f) &&
					   !(alpha < 1.0f / 255.0f);
			if (__ballot(valid) == 0ull) continue;
			const float oma = 1.f - alpha;
			if (valid) T = T / oma;
			const float wgt = valid ? alpha * T : 0.f;   // dchannel_dcolor
			const uint32_t id = __builtin_amdgcn_readfirstlane(e.id);
			const float* __restrict__ col = colors + (size_t)id * C + c0;
			float* __restrict__ dcol = dL_dcolors + (size_t)id * C + c0;
			float S = 0.f;
			float* acc_row = COMBINE ? &s_acc[(wave * 64 + (k - kb)) * NV] : nullptr;
			const int comp = wave_sum8_component(lane);
			const bool head = (lane & 7) == 0;   // the lanes wave_sum8 leaves its eight totals in
			float pc[8];   // this lane's wgt * g[c] of the current eight channels, 0 beyond cn
#pragma unroll
			for (int j = 0; j < 8; j++) pc[j] = 0.f;
#pragma unroll
			for (int c = 0; c < CC; c++) {
				if (c < cn) {
					const float cv = col[c];
					// contribution to dL/dalpha uses the colour accumulated BEHIND this entry
					S += (cv - rec[c]) * g[c];
					pc[c & 7] = wgt * g[c];
					// fold this entry into the running "behind" colour for the next one
					if (valid) rec[c] = alpha * cv + oma * rec[c];
				}
				if (!COMBINE && (c & 7) == 7) {   // 32-channel chunk: eight channel sums per transposed reduction
					if (c - 7 < cn) {
						const float u8 = wave_sum8(pc[0], pc[1], pc[2], pc[3], pc[4], pc[5], pc[6], pc[7]);
						if (head && c - 7 + comp < cn) atomicAdd(&dcol[c - 7 + comp], u8);
					}
#pragma unroll
					for (int j = 0; j < 8; j++) pc[j] = 0.f;
				}
			}
			float dL_dalpha = S * T;
			dL_dalpha += (-T_final / oma) * bg_dot;
			if (!valid) dL_dalpha = 0.f;
			const float Gv = valid ? G : 0.f;   // keeps inf/NaN of rejected lanes out of the sums
			float u;
			if constexpr (COMBINE) {
				// (round 6, as blend_bwd_mfma.hip's geometry walk) the wave sums are the six MOMENTS of r = G dL/dalpha over the pixels -- r dx, r dy,
				// r dx^2, r dx dy, r dy^2, r -- and meet the entry's opacity and conic once per entry where the four waves' sums are added up
				const float r = Gv * dL_dalpha;
				const float rx = r * dx, ry = r * dy;
				u = wave_sum8(rx, ry, rx * dx, rx * dy, ry * dy, r, pc[0], pc[1]);   // (+ the first two colour sums of an RGB chunk)
			} else {
				const float dL_dG = e.o * dL_dalpha;
				const float gdx = Gv * dx, gdy = Gv * dy;
				const float dG_ddelx = -gdx * e.ca - gdy * e.cb;
				const float dG_ddely = -gdy * e.cc - gdx * e.cb;
				// the six geometry sums in one transposed reduction
				u = wave_sum6(dL_dG * dG_ddelx * ddelx_dx, dL_dG * dG_ddely * ddely_dy, -0.5f * gdx * dx * dL_dG,
					      -0.5f * gdx * dy * dL_dG, -0.5f * gdy * dy * dL_dG, Gv * dL_dalpha);
			}
			if (COMBINE) {
				if (head) acc_row[comp] = u;   // slots 0-5 geometry, 6 / 7 = colour channels 0 / 1
#pragma unroll
				for (int c = 2; c < CC; c++) {
					if (c < cn) {
						const float gc = wave_sum(pc[c]);
						if (lane == 0) acc_row[6 + c] = gc;
					}
				}
			} else if (head && comp < 6) {
				float* dst = comp < 2 ? dL_dmean2D + 3 * (size_t)id + comp
					   : comp < 5 ? dL_dconic + 4 * (size_t)id + (comp == 4 ? 3 : comp - 2)
						      : dL_dopacity + id;
				atomicAdd(dst, u);
			}
		}
		if (COMBINE) {   // one atomic per (entry, component) of this group of 64 entries
			__syncthreads();
			for (int q = threadIdx.x; q < (ke - kb) * NV; q += 256) {
				const int e = q / NV, c = q - NV * e;
				auto tot = [&](int k) __attribute__((always_inline)) {
					return (s_acc[(0 * 64 + e) * NV + k] + s_acc[(1 * 64 + e) * NV + k]) +
					       (s_acc[(2 * 64 + e) * NV + k] + s_acc[(3 * 64 + e) * NV + k]);
				};
				const StagedEntryB& E = s_e[kb + e];
				float v;   // slots 0 .. 5: the moments Rx, Ry, Rxx, Rxy, Ryy, R0
				if (c == 0) v = -ddelx_dx * E.o * (E.ca * tot(0) + E.cb * tot(1));
				else if (c == 1) v = -ddely_dy * E.o * (E.cc * tot(1) + E.cb * tot(0));
				else if (c < 5) v = -0.5f * E.o * tot(c);
				else v = tot(c);
				if (v != 0.f) {
					const size_t id = s_e[kb + e].id;
					float* dst = c < 2 ? dL_dmean2D + 3 * id + c
						   : c < 5 ? dL_dconic + 4 * id + (c == 4 ? 3 : c - 2)
						   : c == 5 ? dL_dopacity + id
							    : dL_dcolors + id * C + c0 + (c - 6);
					if (c < 6 || c - 6 < cn) atomicAdd(dst, v);
				}
			}
			__syncthreads();
		}
		}
	}
}

// gate == nullptr: one workgroup per (tile, chunk).  gate != nullptr (the fallback behind the work-list path: it runs only if that path's arena
// overflowed): a strided grid -- 79 056 workgroups that each read the gate word and leave were 21 us of every cfg3 backward; 2 048 are 2.
template <int CC>
__global__ __launch_bounds__(256) void blend_bwd_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float* __restrict__ bg, const float2* __restrict__ means2D,
	const float4* __restrict__ conic_opacity, const float* __restrict__ colors,
	const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
	const float* __restrict__ dL_dpixels, float* __restrict__ dL_dmean2D,
	float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
	int W, int H, int C, int gx, int nchunks, int nblocks, const uint32_t* __restrict__ gate)
{
	if (gate) {
		if (gate[1] == 0u) return;   // the work-list path did the job
		for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
			blend_bwd_block<CC>(blk, ranges, point_list, bg, means2D, conic_opacity, colors, final_Ts, n_contrib, dL_dpixels, dL_dmean2D,
					    dL_dconic, dL_dopacity, dL_dcolors, W, H, C, gx, nchunks);
			__syncthreads();   // (the block's LDS is the next block's)
		}
	} else {
		blend_bwd_block<CC>(blockIdx.x, ranges, point_list, bg, means2D, conic_opacity, colors, final_Ts, n_contrib, dL_dpixels, dL_dmean2D,
				    dL_dconic, dL_dopacity, dL_dcolors, W, H, C, gx, nchunks);
	}
}

hipError_t launch_blend_backward(hipStream_t st, const BlendBwdArgs& a, const uint32_t* gate)
{
	const int ntiles = a.gx * a.gy;
	if (ntiles == 0 || a.C == 0) return hipSuccess;
	if (a.C <= 4) {
		hipLaunchKernelGGL((blend_bwd_kernel<4>), dim3(gate && ntiles > 2048 ? 2048 : ntiles), dim3(256), 0, st, a.ranges,
				   a.point_list, a.bg, a.means2D, a.conic_opacity, a.colors, a.final_T,
				   a.n_contrib, a.dL_dpix, a.dL_dmean2D, a.dL_dconic, a.dL_dopacity,
				   a.dL_dcolors, a.W, a.H, a.C, a.gx, 1, ntiles, gate);
	} else {
		const int nch = (a.C + 31) / 32;
		const int nblocks = ntiles * nch;
		hipLaunchKernelGGL((blend_bwd_kernel<32>), dim3(gate && nblocks > 2048 ? 2048 : nblocks), dim3(256), 0, st, a.ranges,
				   a.point_list, a.bg, a.means2D, a.conic_opacity, a.colors, a.final_T,
				   a.n_contrib, a.dL_dpix, a.dL_dmean2D, a.dL_dconic, a.dL_dopacity,
				   a.dL_dcolors, a.W, a.H, a.C, a.gx, nch, nblocks, gate);
	}
	return hipGetLastError();
}

} // namespace sgs
