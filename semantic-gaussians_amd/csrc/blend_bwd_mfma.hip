// blend_bwd_mfma.hip -- backward of the N-channel alpha-composite as matrix products (C >= 32, C % 32 == 0).
//
// Behaviour: CR/cuda_rasterizer/backward.cu:394-552 (renderCUDA backward), restated for a runtime
// channel count in blend_bwd.hip.  That kernel walks every tile's list once per 32-channel chunk and
// does all of its channel work on the VALU (78 ms at 1M x 512 x 968x1296).  Per (list entry k, pixel p)
// the channel dimension only enters through two contractions:
//
//   D[k][p]      = sum_c F[k][c] * g[c][p]             (g = dL/dpixel)       -> dL/dalpha
//   dL/dF[k][c]  = sum_p w[k][p] * g[c][p]             (w = alpha * T, the forward's blend weight)
//
// and the reference's running "colour behind the entry" only ever appears dotted with g, so it
// collapses to a scalar recurrence:  R_k = sum_c rec_k[c] g[c] = alpha_{k+1} D_{k+1} + (1 - alpha_{k+1}) R_{k+1},
//   dL/dalpha_k = T_k (D_k - R_k) - T_final / (1 - alpha_k) * (bg . g).
// So the backward blend is:
//   1. blend_weights_kernel<3>   (blend_fwd_split.hip) the forward's work list again: per tile the
//                                contributing entries, their fp32 weight rows w[k][256 px'], ids and
//                                list positions; the list's closing pseudo entry (id SGS_BG_ID) stands
//                                for the background: its D row is  bg . g.
//   2. bwd_dcolor_kernel         dL/dF = W G^T per (tile, 128 channels): fp32 MFMA, one coalesced
//                                atomic row per (entry, 32 channels).
//   3. bwd_dot_kernel            D = F G per tile, fp32 MFMA; D rows overwrite the weight rows.
//   4. bwd_geom_kernel           lane = pixel, back-to-front over the work list: recomputes G / alpha,
//                                walks T back from T_final (as the reference does), the scalar
//                                recurrence above, and the wave-reduced atomics of blend_bwd.hip for
//                                mean2D / conic / opacity.
// fp32 MFMA (v_mfma_f32_32x32x2_f32): products and sums are fp32; only the summation order differs from
// blend_bwd.hip (the test bar is 1e-4 of the largest gradient entry, same as for the atomics' order).
// If the work list overflows its arena a device flag makes 2-4 exit and blend_bwd.hip's kernel
// (launched behind them, gated on the same flag) do the work.
#include "sgs_kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace sgs {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t BG_ID = 0xFFFFFFFFu;     // blend_fwd_split.hip SGS_BG_ID
constexpr uint32_t NO_ID = 0xFFFFFFFEu;
constexpr int CHUNK = 128;                  // work-list slots per arena chunk (blend_fwd_split.hip ACH)
constexpr int LDP __attribute__((unused)) = 36;   // LDS row pitch (floats) of the 32-wide operand slabs (the two-kernel form)

// row of the 32x32 MFMA result held in accumulator register r of a lane in half h
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Four consecutive pixels of a gradient row, branch free: `row` points at a (clamped, in-bounds) row, x is
// the first pixel.  VEC (W % 4 == 0, x % 4 == 0): one 16-byte load, the piece is entirely inside or outside
// the image.  Otherwise four clamped scalar loads.  The values of out-of-image pixels are zeroed by
// mask_px4 -- LATER, when the piece is written to LDS: a select right behind the load would make the wave
// wait for it before the matrix work the load is meant to overlap.
template <bool VEC>
__device__ __forceinline__ float4 load_px4(const float* __restrict__ row, int x, int W)
{
	float4 v;
	if (VEC) {
		v = *reinterpret_cast<const float4*>(row + (x < W ? x : W - 4));
	} else {
		v.x = row[x < W ? x : W - 1];
		v.y = row[x + 1 < W ? x + 1 : W - 1];
		v.z = row[x + 2 < W ? x + 2 : W - 1];
		v.w = row[x + 3 < W ? x + 3 : W - 1];
	}
	return v;
}
__device__ __forceinline__ float4 mask_px4(float4 v, int x, int W, bool ok)
{
	v.x = (ok && x < W) ? v.x : 0.f;
	v.y = (ok && x + 1 < W) ? v.y : 0.f;
	v.z = (ok && x + 2 < W) ? v.z : 0.f;
	v.w = (ok && x + 3 < W) ? v.w : 0.f;
	return v;
}
__device__ __forceinline__ float4 mask4(float4 v, bool ok)
{
	v.x = ok ? v.x : 0.f;
	v.y = ok ? v.y : 0.f;
	v.z = ok ? v.z : 0.f;
	v.w = ok ? v.w : 0.f;
	return v;
}

#ifdef SGS_WITH_EXPERIMENTS   // rounds 1-4's two-kernel form (one kernel per product, each streaming the gradient): make EXPERIMENTS=1, backward modes 4 / 5
// ---- 3. D[slot][px'] = sum_c F[id(slot)][c] * g[c][px']
// 512 threads: wave w owns the 32 px' [32 w, 32 w + 32) for up to 128 entries (4 x 16 accumulators), so
// two workgroups fit a CU (the any-width instantiation spills; widths that are not a multiple of 4 are
// rare).  All global loads are unconditional (clamped addresses + selects): with
// branches around them the compiler can no longer count outstanding loads and drains them one by one.
template <bool VEC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void bwd_dot_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const float* __restrict__ features, const float* __restrict__ bg,
	const float* __restrict__ dL_dpix, float* __restrict__ Drows,
	const uint32_t* __restrict__ counter, int W, int H, int C, int gx, int per_xcd, int ntiles)
{
	if (counter[1] != 0u) return;
	const int b = blockIdx.x;
	const int tile = (b & 7) * per_xcd + (b >> 3);
	if (tile >= ntiles) return;
	const int t = threadIdx.x;
	const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int l31 = lane & 31, h = lane >> 5;
	const int tx = tile % gx, ty = tile / gx;
	const uint32_t HW = (uint32_t)H * (uint32_t)W;
	const uint32_t chunk_base = (ranges[tile].x >> 7) + (uint32_t)tile;
	const int total = (int)nact[tile];

	__shared__ float sF[CHUNK * LDP];
	__shared__ float sG[32 * 256];
	__shared__ uint32_t s_id[CHUNK];

	// this thread's share of a 32-channel slab of the tile's gradient: 4 x (channel wave + 8 i, row, 4 px)
	const int g_rem = t & 63, g_row = g_rem >> 2, g_x4 = (g_rem & 3) * 4;
	const int g_y = ty * SGS_TILE + g_row, g_x = tx * SGS_TILE + g_x4;
	const int g_pxp = (g_row & 1) * 128 + (g_row >> 1) * 16 + g_x4;
	const bool g_ok = g_y < H;
	const uint32_t g_off = (uint32_t)wave * HW + (uint32_t)(g_ok ? g_y : H - 1) * (uint32_t)W;   // row start within a slab

	for (int ci = 0; ci * CHUNK < total; ci++) {
		const int cnt = (total - ci * CHUNK) < CHUNK ? (total - ci * CHUNK) : CHUNK;
		const int mb = (cnt + 31) >> 5;
		const uint32_t cstart = sgs_chunk_start(table, chunk_base, (uint32_t)tile, (uint32_t)ci);
		__syncthreads();
		if (t < CHUNK) s_id[t] = t < cnt ? act_id[cstart + t] : NO_ID;
		f32x16 acc[4];
#pragma unroll
		for (int m = 0; m < 4; m++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
		__syncthreads();   // s_id visible
		// this thread's two feature-row pieces of the chunk (fixed for all slabs)
		const float* frow[2];
		bool fvalid[2];
#pragma unroll
		for (int i = 0; i < 2; i++) {
			const int q = t + 512 * i, e = q >> 3, f = q & 7;
			const uint32_t id = s_id[e];
			fvalid[i] = id != NO_ID;
			frow[i] = ((id == BG_ID || id == NO_ID) ? bg : features + (size_t)id * C) + 4 * f;
		}
		// the next 32-channel slab is fetched into registers while the current one is multiplied
		float4 pg[4], pf[2];
		auto fetch = [&](int c0) __attribute__((always_inline)) {
			const float* gb = dL_dpix + (size_t)c0 * HW;   // (uniform)
#pragma unroll
			for (int i = 0; i < 4; i++) pg[i] = load_px4<VEC>(gb + g_off + (uint32_t)(8 * i) * HW, g_x, W);
#pragma unroll
			for (int i = 0; i < 2; i++) pf[i] = *reinterpret_cast<const float4*>(frow[i] + c0);
		};
		fetch(0);
		for (int c0 = 0; c0 < C; c0 += 32) {
			__syncthreads();   // the previous slab's readers are done
#pragma unroll
			for (int i = 0; i < 4; i++)
				*reinterpret_cast<float4*>(&sG[(wave + 8 * i) * 256 + g_pxp]) = mask_px4(pg[i], g_x, W, g_ok);
#pragma unroll
			for (int i = 0; i < 2; i++) {
				const int q = t + 512 * i, e = q >> 3, f = q & 7;
				*reinterpret_cast<float4*>(&sF[e * LDP + 4 * f]) = mask4(pf[i], fvalid[i]);
			}
			__syncthreads();
			if (c0 + 32 < C) fetch(c0 + 32);
			// MFMA k index = lane half h  <->  channel c0 + 16 h + s
#pragma unroll
			for (int s4 = 0; s4 < 4; s4++) {
				float bv[4];
#pragma unroll
				for (int u = 0; u < 4; u++) bv[u] = sG[(16 * h + 4 * s4 + u) * 256 + 32 * wave + l31];
#pragma unroll
				for (int m = 0; m < 4; m++)
					if (m < mb) {
						const float4 a = *reinterpret_cast<const float4*>(&sF[(32 * m + l31) * LDP + 16 * h + 4 * s4]);
						acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv[0], acc[m], 0, 0, 0);
						acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv[1], acc[m], 0, 0, 0);
						acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv[2], acc[m], 0, 0, 0);
						acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv[3], acc[m], 0, 0, 0);
					}
			}
		}
#pragma unroll
		for (int m = 0; m < 4; m++)
			if (m < mb)
#pragma unroll
				for (int r = 0; r < 16; r++) {
					const int e = 32 * m + mfma_row(r, h);
					if (e < cnt) Drows[(size_t)(cstart + e) * 256 + 32 * wave + l31] = acc[m][r];
				}
	}
}

// ---- 2. dL/dF[id(slot)][c] += sum_px' w[slot][px'] * g[c][px']
// 512 threads: wave w owns channels [32 (w & 3), +32) of the 128-channel group and entries
// [64 (w >> 2), +64) of the chunk (2 x 16 accumulators).
template <bool VEC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void bwd_dcolor_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const float* __restrict__ Wrows, const float* __restrict__ dL_dpix,
	float* __restrict__ dL_dcolors, const uint32_t* __restrict__ counter, int W, int H, int C, int gx,
	int nch, int per_xcd, int items)
{
	if (counter[1] != 0u) return;
	const int b = blockIdx.x;
	const int item = (b & 7) * per_xcd + (b >> 3);
	if (item >= items) return;
	const int tile = item / nch, cc = item - tile * nch;
	const int t = threadIdx.x;
	const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int nb = wave & 3, mh = wave >> 2;
	const int l31 = lane & 31, h = lane >> 5;
	const int tx = tile % gx, ty = tile / gx;
	const uint32_t HW = (uint32_t)H * (uint32_t)W;
	const uint32_t chunk_base = (ranges[tile].x >> 7) + (uint32_t)tile;
	const int total = (int)nact[tile];
	const int cbase = cc * 128;

	__shared__ float sW[CHUNK * LDP];
	__shared__ float sG[128 * LDP];
	__shared__ uint32_t s_id[CHUNK];

	// this thread's two pieces of a slab: (row q >> 3, 4 floats at 4 (q & 7)) of BOTH operand slabs --
	// entry q >> 3 of the weights, channel cbase + (q >> 3) of the gradient (image row 2 (f >> 2), px 4 (f & 3))
	const float* gch[2];
	bool cvalid[2];
	int gy0[2], gxx[2];
#pragma unroll
	for (int i = 0; i < 2; i++) {
		const int q = t + 512 * i, e = q >> 3, f = q & 7;
		cvalid[i] = cbase + e < C;
		gch[i] = dL_dpix + (size_t)(cvalid[i] ? cbase + e : cbase) * HW;
		gy0[i] = ty * SGS_TILE + 2 * (f >> 2);
		gxx[i] = tx * SGS_TILE + (f & 3) * 4;
	}

	for (int ci = 0; ci * CHUNK < total; ci++) {
		const int cnt = (total - ci * CHUNK) < CHUNK ? (total - ci * CHUNK) : CHUNK;
		const int cnt16 = (cnt + 15) & ~15;   // rows up to here are initialised (zero padding of the work list)
		const int mb = ((cnt + 31) >> 5) - 2 * mh;   // M blocks of this wave's entry half that hold entries
		const bool wave_on = cbase + 32 * nb < C && mb > 0;   // (C % 32 == 0)
		const uint32_t cstart = sgs_chunk_start(table, chunk_base, (uint32_t)tile, (uint32_t)ci);
		__syncthreads();
		if (t < CHUNK) s_id[t] = t < cnt ? act_id[cstart + t] : NO_ID;
		f32x16 acc[2];
#pragma unroll
		for (int m = 0; m < 2; m++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
		const float* wrow[2];
		bool wvalid[2];
#pragma unroll
		for (int i = 0; i < 2; i++) {
			const int q = t + 512 * i, e = q >> 3, f = q & 7;
			wvalid[i] = e < cnt16;
			wrow[i] = Wrows + (size_t)(cstart + (wvalid[i] ? e : 0)) * 256 + 4 * f;
		}
		// slabs of 32 px' = two image rows of one parity; slab j + 1 is fetched into registers while slab j
		// is multiplied
		float4 pw[2], pg[2];
		bool pok[2];
		auto fetch = [&](int j) __attribute__((always_inline)) {
			const int yo = 4 * (j & 3) + (j >> 2);   // y = 16 ty + 4 (j & 3) + 2 r2 + parity
#pragma unroll
			for (int i = 0; i < 2; i++) {
				pw[i] = *reinterpret_cast<const float4*>(wrow[i] + 32 * j);
				const int y = gy0[i] + yo;
				pok[i] = cvalid[i] && y < H;
				pg[i] = load_px4<VEC>(gch[i] + (uint32_t)(y < H ? y : H - 1) * (uint32_t)W, gxx[i], W);
			}
		};
		fetch(0);
		for (int j = 0; j < 8; j++) {
			__syncthreads();   // the previous slab's readers are done (and s_id is visible)
#pragma unroll
			for (int i = 0; i < 2; i++) {
				const int q = t + 512 * i, e = q >> 3, f = q & 7;
				*reinterpret_cast<float4*>(&sW[e * LDP + 4 * f]) = mask4(pw[i], wvalid[i]);
				*reinterpret_cast<float4*>(&sG[e * LDP + 4 * f]) = mask_px4(pg[i], gxx[i], W, pok[i]);
			}
			__syncthreads();
			if (j + 1 < 8) fetch(j + 1);
			if (wave_on) {
				// MFMA k index = lane half h  <->  px' 32 j + 16 h + s
#pragma unroll
				for (int s4 = 0; s4 < 4; s4++) {
					const float4 b4 = *reinterpret_cast<const float4*>(&sG[(32 * nb + l31) * LDP + 16 * h + 4 * s4]);
#pragma unroll
					for (int m = 0; m < 2; m++)
						if (m < mb) {
							const float4 a = *reinterpret_cast<const float4*>(&sW[(64 * mh + 32 * m + l31) * LDP + 16 * h + 4 * s4]);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b4.x, acc[m], 0, 0, 0);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b4.y, acc[m], 0, 0, 0);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b4.z, acc[m], 0, 0, 0);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b4.w, acc[m], 0, 0, 0);
						}
				}
			}
		}
		if (wave_on) {
#pragma unroll
			for (int m = 0; m < 2; m++)
				if (m < mb)
#pragma unroll
					for (int r = 0; r < 16; r++) {
						const int e = 64 * mh + 32 * m + mfma_row(r, h);
						const uint32_t id = s_id[e];
						if (id < NO_ID) atomicAdd(&dL_dcolors[(size_t)id * C + cbase + 32 * nb + l31], acc[m][r]);
					}
		}
	}
}

#endif   // SGS_WITH_EXPERIMENTS

// ================= split-bf16 forms of the two products (the default) =================
// x = hi + lo + O(2^-16 x) with hi, lo bf16 (round to nearest even), and  F.G ~ Fl.Gh + Fh.Gl + Fh.Gh  on
// v_mfma_f32_32x32x8_bf16_1k with fp32 accumulation: every product is exact in fp32, what is dropped is
// <= 3 * 2^-16 of sum |F||G| -- the forward's default arithmetic (5.2) applied to the backward's two products.
// 3 x 32 cycles per 8 k instead of 4 x 64: the matrix work drops 2.7x and both kernels become staging bound.
// Operand slabs in LDS: one row per M / N index, 36 dwords: [16 dwords: 32 k as bf16 hi][16 dwords: lo][4 pad];
// a lane of half h reads its 16 k (positions 16 h .. 16 h + 15) as ds_read_b128 pairs (conflict free at this pitch).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int LDQ = 36;   // row pitch (dwords) of the split slabs

__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo)
{
	const bf16x2 hv = {(__bf16)a, (__bf16)b};
	hi = __builtin_bit_cast(uint32_t, hv);
	const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xFFFF0000u);
	const bf16x2 lv = {(__bf16)(a - ha), (__bf16)(b - hb)};
	lo = __builtin_bit_cast(uint32_t, lv);
}
__device__ __forceinline__ void split_px4(float4 v, bool ok, uint2& hi, uint2& lo)
{
	split_pair(v.x, v.y, hi.x, lo.x);
	split_pair(v.z, v.w, hi.y, lo.y);
	hi.x = ok ? hi.x : 0u;
	hi.y = ok ? hi.y : 0u;
	lo.x = ok ? lo.x : 0u;
	lo.y = ok ? lo.y : 0u;
}
struct Op2 {   // two k-steps (2 x 4 bf16) of one operand row
	s16x4 k0, k1;
};
__device__ __forceinline__ Op2 lds_op2(const uint32_t* p)
{
	const uint4 v = *reinterpret_cast<const uint4*>(p);
	Op2 r;
	r.k0 = __builtin_bit_cast(s16x4, uint2{v.x, v.y});
	r.k1 = __builtin_bit_cast(s16x4, uint2{v.z, v.w});
	return r;
}
#define SGS_MFMA_BF16(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(A_, B_, C_, 0, 0, 0)
// the double-rate form (fused kernel only): an Op2 -- the lane's 8 consecutive k positions -- IS one operand of
// v_mfma_f32_32x32x16_bf16 (lane half h supplies k = 8 h .. 8 h + 7 of the instruction's 16)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 op16(const Op2& o)
{
	return __builtin_bit_cast(bf16x8, uint4{__builtin_bit_cast(uint2, o.k0).x, __builtin_bit_cast(uint2, o.k0).y,
						 __builtin_bit_cast(uint2, o.k1).x, __builtin_bit_cast(uint2, o.k1).y});
}
#define SGS_MFMA_X16(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(op16(A_), op16(B_), C_, 0, 0, 0)

#ifdef SGS_WITH_EXPERIMENTS
// ---- 3'. D = F G.  The gradient slab has to be transposed on its way into LDS (k = channel is the slow
// dimension of dL_dpix): a thread owns ONE pixel and 16 channels of the slab (16 dword loads, lanes along px':
// four 64-byte row pieces per wave instruction) and writes its 16 k as two 16-byte pieces per half.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void bwd_dot_split_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const float* __restrict__ features, const float* __restrict__ bg,
	const float* __restrict__ dL_dpix, float* __restrict__ Drows,
	const uint32_t* __restrict__ counter, int W, int H, int C, int gx, int per_xcd, int ntiles)
{
	if (counter[1] != 0u) return;
	const int b = blockIdx.x;
	const int tile = (b & 7) * per_xcd + (b >> 3);
	if (tile >= ntiles) return;
	const int t = threadIdx.x;
	const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int l31 = lane & 31, h = lane >> 5;
	const int tx = tile % gx, ty = tile / gx;
	const uint32_t HW = (uint32_t)H * (uint32_t)W;
	const uint32_t chunk_base = (ranges[tile].x >> 7) + (uint32_t)tile;
	const int total = (int)nact[tile];

	__shared__ __attribute__((aligned(16))) uint32_t sF[CHUNK * LDQ];
	__shared__ __attribute__((aligned(16))) uint32_t sG[256 * LDQ];
	__shared__ uint32_t s_id[CHUNK];

	// this thread's pixel px' = t & 255 and channel half (t >> 8) of every slab
	const int gp = t & 255, gh = t >> 8;
	const int g_y = ty * SGS_TILE + 2 * ((gp & 127) >> 4) + (gp >> 7), g_x = tx * SGS_TILE + (gp & 15);
	const bool g_ok = g_y < H && g_x < W;
	const uint32_t g_off = (uint32_t)(16 * gh) * HW + (uint32_t)(g_y < H ? g_y : H - 1) * (uint32_t)W +
			       (uint32_t)(g_x < W ? g_x : W - 1);
	const uint32_t g_offb = 4u * g_off;   // (eligibility: 128 planes * 4 B < 2^32)

	for (int ci = 0; ci * CHUNK < total; ci++) {
		const int cnt = (total - ci * CHUNK) < CHUNK ? (total - ci * CHUNK) : CHUNK;
		const int mb = (cnt + 31) >> 5;
		const uint32_t cstart = sgs_chunk_start(table, chunk_base, (uint32_t)tile, (uint32_t)ci);
		__syncthreads();
		if (t < CHUNK) s_id[t] = t < cnt ? act_id[cstart + t] : NO_ID;
		f32x16 acc[4];
#pragma unroll
		for (int m = 0; m < 4; m++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
		__syncthreads();   // s_id visible
		const float* frow[2];
		bool fvalid[2];
#pragma unroll
		for (int i = 0; i < 2; i++) {
			const int q = t + 512 * i, e = q >> 3, f = q & 7;
			const uint32_t id = s_id[e];
			fvalid[i] = id != NO_ID;
			frow[i] = ((id == BG_ID || id == NO_ID) ? bg : features + (size_t)id * C) + 4 * f;
		}
		float pg[16];
		float4 pf[2];
		auto fetch = [&](int c0) __attribute__((always_inline)) {
			// buffer loads: SGPR descriptor over this slab + scalar plane offset + ONE 32-bit lane offset (as plain
			// pointer arithmetic the compiler tabulates 16 64-bit lane addresses and spills them)
			const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
				const_cast<float*>(dL_dpix + (size_t)c0 * HW), 0, 0xFFFFFFFF, 0x00020000);
#pragma unroll
			for (int j = 0; j < 16; j++)
				pg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, g_offb, (uint32_t)j * HW * 4u, 0));
#pragma unroll
			for (int i = 0; i < 2; i++) pf[i] = *reinterpret_cast<const float4*>(frow[i] + c0);
		};
		fetch(0);
		for (int c0 = 0; c0 < C; c0 += 32) {
			__syncthreads();   // the previous slab's readers are done
			{
				uint32_t hi[8], lo[8];
#pragma unroll
				for (int j = 0; j < 8; j++) {
					split_pair(pg[2 * j], pg[2 * j + 1], hi[j], lo[j]);
					hi[j] = g_ok ? hi[j] : 0u;
					lo[j] = g_ok ? lo[j] : 0u;
				}
				uint32_t* row = &sG[gp * LDQ + 8 * gh];
				*reinterpret_cast<uint4*>(row) = uint4{hi[0], hi[1], hi[2], hi[3]};
				*reinterpret_cast<uint4*>(row + 4) = uint4{hi[4], hi[5], hi[6], hi[7]};
				*reinterpret_cast<uint4*>(row + 16) = uint4{lo[0], lo[1], lo[2], lo[3]};
				*reinterpret_cast<uint4*>(row + 20) = uint4{lo[4], lo[5], lo[6], lo[7]};
			}
#pragma unroll
			for (int i = 0; i < 2; i++) {
				const int q = t + 512 * i, e = q >> 3, f = q & 7;
				uint2 hi, lo;
				split_px4(pf[i], fvalid[i], hi, lo);
				*reinterpret_cast<uint2*>(&sF[e * LDQ + 2 * f]) = hi;
				*reinterpret_cast<uint2*>(&sF[e * LDQ + 16 + 2 * f]) = lo;
			}
			__syncthreads();
			if (c0 + 32 < C) fetch(c0 + 32);
			// k = position 16 h + 4 s + j of the slab (= channel c0 + that) for both operands
#pragma unroll
			for (int sp = 0; sp < 2; sp++) {
				const uint32_t* gb_ = &sG[(32 * wave + l31) * LDQ + 8 * h + 4 * sp];
				const Op2 bh = lds_op2(gb_), bl = lds_op2(gb_ + 16);
#pragma unroll
				for (int m = 0; m < 4; m++)
					if (m < mb) {
						const uint32_t* fa = &sF[(32 * m + l31) * LDQ + 8 * h + 4 * sp];
						const Op2 ah = lds_op2(fa), al = lds_op2(fa + 16);
						acc[m] = SGS_MFMA_BF16(al.k0, bh.k0, acc[m]);
						acc[m] = SGS_MFMA_BF16(ah.k0, bl.k0, acc[m]);
						acc[m] = SGS_MFMA_BF16(ah.k0, bh.k0, acc[m]);
						acc[m] = SGS_MFMA_BF16(al.k1, bh.k1, acc[m]);
						acc[m] = SGS_MFMA_BF16(ah.k1, bl.k1, acc[m]);
						acc[m] = SGS_MFMA_BF16(ah.k1, bh.k1, acc[m]);
					}
			}
		}
#pragma unroll
		for (int m = 0; m < 4; m++)
			if (m < mb)
#pragma unroll
				for (int r = 0; r < 16; r++) {
					const int e = 32 * m + mfma_row(r, h);
					if (e < cnt) Drows[(size_t)(cstart + e) * 256 + 32 * wave + l31] = acc[m][r];
				}
	}
}

// ---- 2'. dL/dF = W G^T: k = px' is the fast dimension of both operands, so staging is convert + pack.
template <bool VEC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void bwd_dcolor_split_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const float* __restrict__ Wrows, const float* __restrict__ dL_dpix,
	float* __restrict__ dL_dcolors, const uint32_t* __restrict__ counter, int W, int H, int C, int gx,
	int nch, int per_xcd, int items)
{
	if (counter[1] != 0u) return;
	const int b = blockIdx.x;
	const int item = (b & 7) * per_xcd + (b >> 3);
	if (item >= items) return;
	const int tile = item / nch, cc = item - tile * nch;
	const int t = threadIdx.x;
	const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int nb = wave & 3, mh = wave >> 2;
	const int l31 = lane & 31, h = lane >> 5;
	const int tx = tile % gx, ty = tile / gx;
	const uint32_t HW = (uint32_t)H * (uint32_t)W;
	const uint32_t chunk_base = (ranges[tile].x >> 7) + (uint32_t)tile;
	const int total = (int)nact[tile];
	const int cbase = cc * 128;

	__shared__ __attribute__((aligned(16))) uint32_t sW[CHUNK * LDQ];
	__shared__ __attribute__((aligned(16))) uint32_t sG[128 * LDQ];
	__shared__ uint32_t s_id[CHUNK];

	const float* gch[2];
	bool cvalid[2];
	int gy0[2], gxx[2];
#pragma unroll
	for (int i = 0; i < 2; i++) {
		const int q = t + 512 * i, e = q >> 3, f = q & 7;
		cvalid[i] = cbase + e < C;
		gch[i] = dL_dpix + (size_t)(cvalid[i] ? cbase + e : cbase) * HW;
		gy0[i] = ty * SGS_TILE + 2 * (f >> 2);
		gxx[i] = tx * SGS_TILE + (f & 3) * 4;
	}

	for (int ci = 0; ci * CHUNK < total; ci++) {
		const int cnt = (total - ci * CHUNK) < CHUNK ? (total - ci * CHUNK) : CHUNK;
		const int cnt16 = (cnt + 15) & ~15;
		const int mb = ((cnt + 31) >> 5) - 2 * mh;
		const bool wave_on = cbase + 32 * nb < C && mb > 0;
		const uint32_t cstart = sgs_chunk_start(table, chunk_base, (uint32_t)tile, (uint32_t)ci);
		__syncthreads();
		if (t < CHUNK) s_id[t] = t < cnt ? act_id[cstart + t] : NO_ID;
		f32x16 acc[2];
#pragma unroll
		for (int m = 0; m < 2; m++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
		const float* wrow[2];
		bool wvalid[2];
#pragma unroll
		for (int i = 0; i < 2; i++) {
			const int q = t + 512 * i, e = q >> 3, f = q & 7;
			wvalid[i] = e < cnt16;
			wrow[i] = Wrows + (size_t)(cstart + (wvalid[i] ? e : 0)) * 256 + 4 * f;
		}
		float4 pw[2], pg[2];
		bool pok[2];
		auto fetch = [&](int j) __attribute__((always_inline)) {
			const int yo = 4 * (j & 3) + (j >> 2);
#pragma unroll
			for (int i = 0; i < 2; i++) {
				pw[i] = *reinterpret_cast<const float4*>(wrow[i] + 32 * j);
				const int y = gy0[i] + yo;
				pok[i] = cvalid[i] && y < H;
				pg[i] = load_px4<VEC>(gch[i] + (uint32_t)(y < H ? y : H - 1) * (uint32_t)W, gxx[i], W);
			}
		};
		fetch(0);
		for (int j = 0; j < 8; j++) {
			__syncthreads();
#pragma unroll
			for (int i = 0; i < 2; i++) {
				const int q = t + 512 * i, e = q >> 3, f = q & 7;
				uint2 hi, lo;
				split_px4(pw[i], wvalid[i], hi, lo);
				*reinterpret_cast<uint2*>(&sW[e * LDQ + 2 * f]) = hi;
				*reinterpret_cast<uint2*>(&sW[e * LDQ + 16 + 2 * f]) = lo;
				split_px4(mask_px4(pg[i], gxx[i], W, pok[i]), true, hi, lo);
				*reinterpret_cast<uint2*>(&sG[e * LDQ + 2 * f]) = hi;
				*reinterpret_cast<uint2*>(&sG[e * LDQ + 16 + 2 * f]) = lo;
			}
			__syncthreads();
			if (j + 1 < 8) fetch(j + 1);
			if (wave_on) {
				// k = position 16 h + 4 s + jj of the slab (= px' 32 j + that) for both operands
#pragma unroll
				for (int sp = 0; sp < 2; sp++) {
					const uint32_t* gb_ = &sG[(32 * nb + l31) * LDQ + 8 * h + 4 * sp];
					const Op2 bh = lds_op2(gb_), bl = lds_op2(gb_ + 16);
#pragma unroll
					for (int m = 0; m < 2; m++)
						if (m < mb) {
							const uint32_t* wa = &sW[(64 * mh + 32 * m + l31) * LDQ + 8 * h + 4 * sp];
							const Op2 ah = lds_op2(wa), al = lds_op2(wa + 16);
							acc[m] = SGS_MFMA_BF16(al.k0, bh.k0, acc[m]);
							acc[m] = SGS_MFMA_BF16(ah.k0, bl.k0, acc[m]);
							acc[m] = SGS_MFMA_BF16(ah.k0, bh.k0, acc[m]);
							acc[m] = SGS_MFMA_BF16(al.k1, bh.k1, acc[m]);
							acc[m] = SGS_MFMA_BF16(ah.k1, bl.k1, acc[m]);
							acc[m] = SGS_MFMA_BF16(ah.k1, bh.k1, acc[m]);
						}
				}
			}
		}
		if (wave_on) {
#pragma unroll
			for (int m = 0; m < 2; m++)
				if (m < mb)
#pragma unroll
					for (int r = 0; r < 16; r++) {
						const int e = 64 * mh + 32 * m + mfma_row(r, h);
						const uint32_t id = s_id[e];
						if (id < NO_ID) atomicAdd(&dL_dcolors[(size_t)id * C + cbase + 32 * nb + l31], acc[m][r]);
					}
		}
	}
}
#endif   // SGS_WITH_EXPERIMENTS

// ================= 2 + 3 fused (round 5, the default): both products of a tile from ONE read of its gradient =================
// The two products contract the gradient slab g[32 c][256 px'] over different dimensions (D = F g over c, dL/dF = W g^T over
// px'), which is why rounds 2-4 ran them as two kernels that each streamed the 2.57 GB gradient.  One workgroup of eight waves
// per tile, all channels, the work list in chunks of up to 128 entries; per 32-channel slab:
//   * a lane owns ONE pixel (px' = 32 wave + lane % 32) and the sixteen channels 16 h .. 16 h + 15 of its lane half: sixteen
//     dword loads a slab ahead.  Split into bf16 (hi, lo) pairs those registers ARE the B operand of D's products for the
//     wave's own N block of 32 px' (accumulators: 4 x 16 registers for 128 entries), no LDS involved.  A = the chunk's
//     feature rows, staged per slab by all threads (double buffered).
//   * the same registers are written, 16 bits at a time, into a [c][256 px'] tile in LDS -- the transpose, K = px' contiguous
//     -- which is the B operand of W g^T one slab LATER (double buffered: the one barrier per slab publishes it).  For that
//     product wave w is (entry block w & 3, K half w >> 2): its A operand, the weights of 32 entries x 128 px', lives in 64
//     registers for the whole chunk, 48 products per slab go into ONE accumulator, the upper half's partial sum crosses to the
//     lower half's wave through LDS (plain stores: ds_add_f32 runs at ~170 cycles per wave instruction on this chip,
//     tools/ubench_ldsadd.hip -- a reduction of eight partial tiles through LDS float atomics made this kernel 13 ms) and
//     leaves as one coalesced 128-B atomic row per (entry, 32 channels), issued before the next loads so that the in-order
//     memory counter never waits for an atomic's round trip.
//   * the two waves of a SIMD run their phases in opposite order (lower half: split / transpose, then D, then W g^T of the
//     previous slab; upper half: W g^T first), so one wave's VALU / LDS phase lies beside the other's matrix phase.
// FP32 (backward mode 3): the same data flow with the operands left in fp32 and v_mfma_f32_32x32x2_f32 (k = lane half, the
// sixteen positions 16 h + s of a 32-wide K block one by one) -- identical register and LDS footprints.
constexpr int GROW = 260;                   // dwords per channel row of the transposed slab: [128 dwords hi | 128 lo | 4 pad] (or 256 floats + 4)
constexpr int GTILE = 32 * GROW + 32;       // rows 16 .. 31 start 32 dwords later: the two lane halves of a transposing store hit different banks
__device__ __forceinline__ int g_row(int c) { return c * GROW + (c >> 4) * 32; }

// PERSIST (round 6): one workgroup per compute unit that takes its tiles from a per-XCD ticket and requests the NEXT tile's first loads
// -- the count words, the entry ids, slab 0 of the gradient, then (once the registers of the split weights are dead) the weight rows and the first feature
// pieces -- during the current tile's tail (phase clocks on x16, profiles/r06_backward_phases.txt: that round trip and the wait for the slowest wave's are
// 10 % of an iteration with one workgroup per tile, and with one workgroup per CU nothing hides it).  The ticket behind the current tile's is requested at
// the tile's top and read in its tail: the dynamic balance of separately dispatched workgroups without their start-up.  Kernel: -3.9 %.
template <bool FP32, int DBG = 0, bool PERSIST = false>   // DBG (development, wrong results): 2 no global atomics, 4 no products, 8 no transposing stores, 16 phase stamps
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_fused_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const float* Wrows, const float* __restrict__ features, const float* __restrict__ bg,
	const float* __restrict__ dL_dpix, float* Drows, float* __restrict__ dL_dcolors,
	const uint32_t* __restrict__ counter, uint32_t capacity, int W, int H, int C, int gx, int per_xcd, int ntiles,
	unsigned long long* __restrict__ trace, uint32_t* tickets)
{
	// (split form: double-rate MFMAs) 256 registers per wave, pinned: with two waves per SIMD the workgroup owns its compute unit's register
	// files outright -- no foreign wave can be resident beside its matrix instructions; all eight waves are resident from
	// dispatch, and the last products of a tile lie before a barrier that every wave passes before it can leave
	if constexpr (!FP32) asm volatile("" : : : "v255");
	const int b = blockIdx.x;
	const int t = threadIdx.x;
	// (XCD bands, not the forward's longest-first order: measured, that order is 1.5 % slower here too -- neighbouring tiles share
	// feature rows in an XCD's L2 -- profiles/r05_backward_fused.txt)
	int tile = (b & 7) * per_xcd + (b >> 3);
	// PERSIST: the tiles of this XCD's band by ticket (tickets[x]: zeroed by the arena's reset kernel)
	__shared__ int s_tk;
	const int band_lo = (b & 7) * per_xcd;
	const int band_n = (ntiles - band_lo) < per_xcd ? (ntiles - band_lo) : per_xcd;
	if constexpr (PERSIST) {
		if (band_n <= 0) return;
		if (t == 0) s_tk = (int)__hip_atomic_fetch_add(&tickets[b & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
		const int tk = __builtin_amdgcn_readfirstlane(s_tk);
		__syncthreads();   // (s_tk is written again below)
		if (tk >= band_n) return;
		tile = band_lo + tk;
	}
	// (the overflow word is NOT tested here: a dependent load in front of everything costs every workgroup a memory round trip;
	// it travels with the first chunk's loads below.  A first chunk beyond the arena's capacity means overflow without asking.)
	if (tile >= ntiles || ((uint32_t)tile + 1u) * 128u > capacity) return;
	const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int l31 = lane & 31, h = lane >> 5;
	// DBG & 16 (tools/bwd_phases.py): shader-clock stamps at the phase boundaries of an iteration, summed per wave.  s_memtime is an
	// SMEM access (reading it drains lgkmcnt), so the stamps cost LDS overlap: read the shares, not the total
	constexpr bool PH = (DBG & 16) != 0;
	uint32_t ph[20] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
	uint32_t ph_prev = PH ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
	uint32_t ph_iters = 0u, ph_total = 0u;   // (phase stamps) slab iterations and entries of this workgroup's tiles
#define SGS_PH(K_)                                                          \
	if (PH) {                                                           \
		const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); \
		ph[K_] += now_ - ph_prev;                                   \
		ph_prev = now_;                                             \
	}
	const int mblk = wave & 3, kh = wave >> 2;   // W g^T: this wave's block of 32 entries and its half of the 256 px' (= row parity)
	const uint32_t HW = (uint32_t)H * (uint32_t)W;
	const int nsl = C >> 5;

	__shared__ __attribute__((aligned(16))) uint32_t sF[2][CHUNK * LDQ];   // the chunk's feature rows of a slab (A of D), double buffered
	__shared__ __attribute__((aligned(16))) uint32_t sG[2][GTILE];         // a slab transposed: [c][256 px'], double buffered
	__shared__ __attribute__((aligned(16))) float sX[2][4][2][8 * 64];     // partial dL/dF tiles crossing to the other K half: [entry block][destination half][register][lane]
	__shared__ uint32_t s_id[CHUNK];

	// this lane's pixel px' = 32 wave + l31 and its sixteen channels 16 h .. 16 h + 15 of every slab
	const int gp = 32 * wave + l31;
	// (of the CURRENT tile; PERSIST moves them on to the next tile once the current tile's last slab has been taken)
	bool g_ok;
	uint32_t g_okm, g_offb;
	auto set_tile = [&](int tl) __attribute__((always_inline)) {
		// (from an opaque copy of the thread index: the lane terms are a dozen instructions a tile -- hoisted out of the tile loop they were held
		// in registers the slab loop does not have, and spilled)
		int tq = t;
		if constexpr (PERSIST) asm volatile("" : "+v"(tq));
		const int gq = ((tq >> 6) << 5) | (tq & 31), hq = (tq >> 5) & 1;
		const int tx = tl % gx, ty = tl / gx;
		const int g_y = ty * SGS_TILE + 2 * ((gq & 127) >> 4) + (gq >> 7), g_x = tx * SGS_TILE + (gq & 15);
		g_ok = g_y < H && g_x < W;
		g_okm = g_ok ? 0xFFFFFFFFu : 0u;
		g_offb = 4u * ((uint32_t)(16 * hq) * HW + (uint32_t)(g_y < H ? g_y : H - 1) * (uint32_t)W +
			       (uint32_t)(g_x < W ? g_x : W - 1));   // (eligibility: 128 planes * 4 B < 2^32)
	};
	set_tile(tile);
	const int wr16 = 2 * g_row(16 * h) + gp;             // transposing store, 16-bit units: element (c = 16 h, px' = gp), hi term
	const int wr32 = g_row(16 * h) + gp;                 // (FP32) in floats
	const int rd = g_row(l31) + 64 * kh + 8 * h;         // W g^T's B operand, dwords: row c = l31, px' 128 kh + 16 h .. of block 0 (+ 16 j)
	const int rdf = g_row(l31) + 128 * kh + 16 * h;      // (FP32) floats (+ 32 j)

	float pg[16];
	auto fetch_g = [&](int c0) __attribute__((always_inline)) {
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
			const_cast<float*>(dL_dpix + (size_t)c0 * HW), 0, 0xFFFFFFFF, 0x00020000);
#pragma unroll
		for (int j = 0; j < 16; j++)
			pg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, g_offb, (uint32_t)j * HW * 4u, 0));
	};

	// Every tile has at least one entry (the closing background entry), and its first chunk sits at slot 128 tile: everything the
	// first chunk needs -- the gradient's first slab, the entry ids, the weight rows -- is requested in ONE round trip, beside the
	// entry count itself (masks are applied when it has arrived).  Written the obvious way (count, then ids, then pointers, then
	// rows) a workgroup spent four dependent round trips, 13 us, before its first product: 16 % of the kernel.
	// The loads a chunk starts with (ids, weight rows; fetch_g(0) goes with them) are issued at the chunk's top, or -- PERSIST, a tile's first chunk -- during the
	// previous tile's tail (`pre`).
	uint32_t my_id = 0u, f_id[2] = {0u, 0u};
	float4 wraw[4][4];
	auto load_ids = [&](uint32_t cs) __attribute__((always_inline)) {
		my_id = act_id[cs + (uint32_t)(t & (CHUNK - 1))];   // (slots beyond the count: this chunk's own, unused memory)
#pragma unroll
		for (int i = 0; i < 2; i++) f_id[i] = act_id[cs + (uint32_t)((t + 512 * i) >> 3)];
	};
	// the chunk's weights for W g^T: entries 32 mblk + l31, px' 128 kh + 32 j + 16 h .. + 15 (j = 0 .. 3), resident all chunk
	// (n: the chunk's entry count where it is known when the rows are requested -- every chunk but a workgroup's first; rows beyond it, a quarter of a
	// chunk's 128 KB on average, are then not requested at all.  They are masked after arrival either way.)
	auto load_w = [&](uint32_t cs, int n) __attribute__((always_inline)) {
		const float* wr = Wrows + (size_t)(cs + (uint32_t)(32 * mblk + l31)) * 256 + 128 * kh + 16 * h;
		if (32 * mblk + l31 < n) {
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int i = 0; i < 4; i++) wraw[j][i] = *reinterpret_cast<const float4*>(wr + 32 * j + 4 * i);
		} else {
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int i = 0; i < 4; i++) wraw[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
		}
	};
	// the feature rows of the chunk's entries (A of D): thread t stages pieces (t + 512 i) & 7 of entries (t + 512 i) >> 3
	const float* frow[2];
	bool fvalid[2];
	float4 pf[2];
	auto set_rows = [&](int n) __attribute__((always_inline)) {
#pragma unroll
		for (int i = 0; i < 2; i++) {
			const int q = t + 512 * i, e = q >> 3, f = q & 7;
			fvalid[i] = e < n;
			const uint32_t id = fvalid[i] ? f_id[i] : NO_ID;
			frow[i] = ((id == BG_ID || id == NO_ID) ? bg : features + (size_t)id * C) + 4 * f;
		}
	};
	auto fetch_f = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
		for (int i = 0; i < 2; i++) pf[i] = *reinterpret_cast<const float4*>(frow[i] + c0);
	};
	bool pre = false;
	uint32_t total_pre_v = 0u, cb_pre_v = 0u;   // (the next tile's count words: vector loads -- scalar ones would put a memory round trip in front of the tail's first LDS wait)
	for (;;) {   // (PERSIST: this workgroup's tiles; otherwise once)
	int total = 0x7fffffff;
	uint32_t chunk_base = 0u;
	int next_tile = -1;
	for (int ci = 0; ci * CHUNK < total; ci++) {
		const uint32_t cstart = ci == 0 ? (uint32_t)tile * 128u : table[chunk_base + (uint32_t)ci];
		SGS_PH(9)
		// (PERSIST) the ticket behind this tile's, requested a whole tile before it is needed: the balance of separately dispatched workgroups without their start-up
		uint32_t tk_nx = 0u;
		if constexpr (PERSIST) {
			if (ci == 0 && t == 0) tk_nx = __hip_atomic_fetch_add(&tickets[b & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		const bool was_pre = PERSIST && pre && ci == 0;
		if (!was_pre) {
			fetch_g(0);
			load_ids(cstart);
			load_w(cstart, ci == 0 ? CHUNK : total - ci * CHUNK);
		}
		if (ci == 0) {
			if (PERSIST && pre) {
				total = (int)__builtin_amdgcn_readfirstlane(total_pre_v);
				chunk_base = __builtin_amdgcn_readfirstlane(cb_pre_v);
				pre = false;
			} else {
				const uint32_t overflow = counter[1];
				total = (int)nact[tile];
				chunk_base = (ranges[tile].x >> 7) + (uint32_t)tile;
				if (overflow != 0u) return;   // (uniform; nothing has been written yet) the per-chunk kernel behind this one does the work
			}
		}
		const int cnt = (total - ci * CHUNK) < CHUNK ? (total - ci * CHUNK) : CHUNK;
		const int mb = (cnt + 31) >> 5;
		const bool e_on = mblk < mb;   // this wave's entry block holds entries
		if (PH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		SGS_PH(10)   // first round trip (count, ids, weights, gradient slab 0)
		__syncthreads();   // the previous chunk is done with s_id, sF, sG, sX
		if (t < CHUNK) s_id[t] = t < cnt ? my_id : NO_ID;
		if (!was_pre) {   // (PERSIST: a prefetched tile's first pieces were requested in the previous tile's tail)
			set_rows(cnt);
			fetch_f(0);
		}
		SGS_PH(11)   // barrier, ids, row pointers
		uint32_t wh[4][8], wl[4][8];   // (split) bf16 pairs of positions 2 i, 2 i + 1
		float wv[4][16];               // (FP32)
		{
			const bool ok = 32 * mblk + l31 < cnt;
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int i = 0; i < 4; i++) {
					const float4 v = mask4(wraw[j][i], ok);
					if (FP32) {
						wv[j][4 * i] = v.x;
						wv[j][4 * i + 1] = v.y;
						wv[j][4 * i + 2] = v.z;
						wv[j][4 * i + 3] = v.w;
					} else {
						split_pair(v.x, v.y, wh[j][2 * i], wl[j][2 * i]);
						split_pair(v.z, v.w, wh[j][2 * i + 1], wl[j][2 * i + 1]);
					}
				}
		}
		f32x16 acc[4];
#pragma unroll
		for (int m = 0; m < 4; m++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
		float held[8];   // this wave's own half of its partial dL/dF tile (lower K half: registers 0 .. 7, upper: 8 .. 15), a slab long
#pragma unroll
		for (int r = 0; r < 8; r++) held[r] = 0.f;
		uint32_t gh[8], gl[8];
		float gv[16];
		auto stage_f = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
			for (int i = 0; i < 2; i++) {
				const int q = t + 512 * i, e = q >> 3, f = q & 7;
				if (FP32) {
					*reinterpret_cast<float4*>(&sF[buf][e * LDQ + 4 * f]) = mask4(pf[i], fvalid[i]);
				} else {
					uint2 hi, lo;
					split_px4(pf[i], fvalid[i], hi, lo);
					*reinterpret_cast<uint2*>(&sF[buf][e * LDQ + 2 * f]) = hi;
					*reinterpret_cast<uint2*>(&sF[buf][e * LDQ + 16 + 2 * f]) = lo;
				}
			}
		};
		// A: the prefetched sub-slab becomes D's B operand (registers) and goes, transposed, into the slab's LDS tile
		auto take_slab = [&](int buf) __attribute__((always_inline)) {
			if (FP32) {
				float* gt = reinterpret_cast<float*>(&sG[buf][0]) + wr32;
#pragma unroll
				for (int j = 0; j < 16; j++) {
					gv[j] = g_ok ? pg[j] : 0.f;
					if (!(DBG & 8)) gt[j * GROW] = gv[j];
				}
			} else {
				uint16_t* gt = reinterpret_cast<uint16_t*>(&sG[buf][0]) + wr16;
#pragma unroll
				for (int j = 0; j < 8; j++) {
					split_pair(pg[2 * j], pg[2 * j + 1], gh[j], gl[j]);
					gh[j] &= g_okm;
					gl[j] &= g_okm;
					if (DBG & 8) continue;
					gt[2 * (2 * j) * GROW] = (uint16_t)gh[j];
					gt[2 * (2 * j + 1) * GROW] = (uint16_t)(gh[j] >> 16);
					gt[2 * (2 * j) * GROW + 256] = (uint16_t)gl[j];
					gt[2 * (2 * j + 1) * GROW + 256] = (uint16_t)(gl[j] >> 16);
				}
			}
			asm volatile("" ::: "memory");   // (the tile is read through differently typed pointers)
		};
		// D[e][px'] += sum_c F[e][c] g[c][px'] for the wave's 32 px'
		auto prod_d = [&](int buf) __attribute__((always_inline)) {
			if (DBG & 4) {
#pragma unroll
				for (int j = 0; j < 8; j++) acc[0][j] += FP32 ? gv[j] : __uint_as_float(gh[j] ^ gl[j]);
			} else if (FP32) {
				const float* sFf = reinterpret_cast<const float*>(&sF[buf][0]);
#pragma unroll
				for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
					for (int m = 0; m < 4; m++)
						if (m < mb) {
							const float4 a = *reinterpret_cast<const float4*>(&sFf[(32 * m + l31) * LDQ + 16 * h + 4 * s4]);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, gv[4 * s4], acc[m], 0, 0, 0);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, gv[4 * s4 + 1], acc[m], 0, 0, 0);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, gv[4 * s4 + 2], acc[m], 0, 0, 0);
							acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, gv[4 * s4 + 3], acc[m], 0, 0, 0);
						}
			} else {
				// the feature operand of the NEXT (k pair, block) is read from LDS before the current six products are issued: read
				// where it is used, an LDS round trip stood in front of every group of six and a lone wave's matrix phase ran at 60 %
				const uint32_t* fa0 = &sF[buf][l31 * LDQ + 8 * h];
				Op2 nah = lds_op2(fa0), nal = lds_op2(fa0 + 16);
#pragma unroll
				for (int sp = 0; sp < 2; sp++) {
					Op2 bh, bl;
					bh.k0 = __builtin_bit_cast(s16x4, uint2{gh[4 * sp], gh[4 * sp + 1]});
					bh.k1 = __builtin_bit_cast(s16x4, uint2{gh[4 * sp + 2], gh[4 * sp + 3]});
					bl.k0 = __builtin_bit_cast(s16x4, uint2{gl[4 * sp], gl[4 * sp + 1]});
					bl.k1 = __builtin_bit_cast(s16x4, uint2{gl[4 * sp + 2], gl[4 * sp + 3]});
#pragma unroll
					for (int m = 0; m < 4; m++)
						if (m < mb) {
							const Op2 ah = nah, al = nal;
							const bool wrap = m + 1 >= mb;   // (uniform) the next operand is block 0 of the second k pair
							if (!(wrap && sp == 1)) {
								const uint32_t* fn = fa0 + (wrap ? 4 : 32 * (m + 1) * LDQ + 4 * sp);
								nah = lds_op2(fn);
								nal = lds_op2(fn + 16);
							}
							acc[m] = SGS_MFMA_X16(al, bh, acc[m]);
							acc[m] = SGS_MFMA_X16(ah, bl, acc[m]);
							acc[m] = SGS_MFMA_X16(ah, bh, acc[m]);
						}
				}
			}
		};
		// E: this wave's partial dL/dF[32 mblk ..][slab's 32 c] over its 128 px' from the slab tile `buf`.  The two K halves of an
		// entry block split the FINISHING of the tile: the lower half's wave keeps accumulator registers 0 .. 7 and parks 8 .. 15 in
		// sX for its partner, the upper half's wave the other way round (with all sixteen finished by the lower half, its waves spent
		// a quarter of every iteration on the atomics while the upper half waited at the barrier)
		auto prod_e = [&](int buf) __attribute__((always_inline)) {
			if (!e_on) return;
			f32x16 a2;
#pragma unroll
			for (int r = 0; r < 16; r++) a2[r] = 0.f;
			if (DBG & 4) {
				a2[0] = __uint_as_float(sG[buf][rd]);
			} else if (FP32) {
				const float* gt = reinterpret_cast<const float*>(&sG[buf][0]) + rdf;
#pragma unroll
				for (int j = 0; j < 4; j++)
#pragma unroll
					for (int s4 = 0; s4 < 4; s4++) {
						const float4 b4 = *reinterpret_cast<const float4*>(gt + 32 * j + 4 * s4);
						a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][4 * s4], b4.x, a2, 0, 0, 0);
						a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][4 * s4 + 1], b4.y, a2, 0, 0, 0);
						a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][4 * s4 + 2], b4.z, a2, 0, 0, 0);
						a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][4 * s4 + 3], b4.w, a2, 0, 0, 0);
					}
			} else {
				const uint32_t* gt = &sG[buf][rd];
				Op2 nbh = lds_op2(gt), nbl = lds_op2(gt + 128);   // (read one (block, k pair) ahead of the products, as in D)
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const int j = q >> 1, sp = q & 1;
					const Op2 bh = nbh, bl = nbl;
					if (q + 1 < 8) {
						nbh = lds_op2(gt + 16 * ((q + 1) >> 1) + 4 * ((q + 1) & 1));
						nbl = lds_op2(gt + 128 + 16 * ((q + 1) >> 1) + 4 * ((q + 1) & 1));
					}
					Op2 ah, al;
					ah.k0 = __builtin_bit_cast(s16x4, uint2{wh[j][4 * sp], wh[j][4 * sp + 1]});
					ah.k1 = __builtin_bit_cast(s16x4, uint2{wh[j][4 * sp + 2], wh[j][4 * sp + 3]});
					al.k0 = __builtin_bit_cast(s16x4, uint2{wl[j][4 * sp], wl[j][4 * sp + 1]});
					al.k1 = __builtin_bit_cast(s16x4, uint2{wl[j][4 * sp + 2], wl[j][4 * sp + 3]});
					a2 = SGS_MFMA_X16(al, bh, a2);
					a2 = SGS_MFMA_X16(ah, bl, a2);
					a2 = SGS_MFMA_X16(ah, bh, a2);
				}
			}
			float* xo = &sX[buf][mblk][kh ^ 1][lane];
			if (kh) {
#pragma unroll
				for (int r = 0; r < 8; r++) {
					xo[r * 64] = a2[r];
					held[r] = a2[8 + r];
				}
			} else {
#pragma unroll
				for (int r = 0; r < 8; r++) {
					xo[r * 64] = a2[8 + r];
					held[r] = a2[r];
				}
			}
		};
		// B: `held` + the partner's registers in sX[buf] (a barrier old) = this wave's eight rows of dL/dF of the slab at channel cb:
		// one coalesced 128-B atomic row per (entry, 32 channels)
		auto finish_e = [&](int buf, int cb) __attribute__((always_inline)) {
			if (!e_on) return;
			// (the row index is laundered: left loop invariant, the ids and row addresses are tabulated per chunk -- registers the
			// kernel does not have; they spill and come back from scratch in every slab)
			int eb = 32 * mblk + 16 * kh + 4 * h;   // accumulator register 8 kh + r of half h = entry row 16 kh + (r & 3) + 8 (r >> 2) + 4 h
			asm volatile("" : "+v"(eb));
			const float* xi = &sX[buf][mblk][kh][lane];
			float v[8];
			uint32_t id[8];
#pragma unroll
			for (int r = 0; r < 8; r++) {   // sixteen LDS reads in flight, one wait
				v[r] = xi[r * 64];
				id[r] = s_id[eb + (r & 3) + 8 * (r >> 2)];
			}
			float* const col = dL_dcolors + cb + l31;
#pragma unroll
			for (int r = 0; r < 8; r++) {
				const float sum = held[r] + v[r];
				if (id[r] < NO_ID && (!(DBG & 2) || sum == 12345.678f)) atomicAdd(col + (size_t)id[r] * C, sum);
			}
		};

		SGS_PH(12)   // weights split
		stage_f(0);
		SGS_PH(13)   // second round trip (feature pieces) + staging
		if constexpr (PERSIST) {
			if (ci == 0 && t == 0) s_tk = (int)tk_nx;   // every thread reads it in the tail of the tile's last chunk
		}
		__syncthreads();
		SGS_PH(14)
		// iteration s: slab s is taken and multiplied into D, W g^T of slab s - 1 runs from the tile the last barrier published,
		// slab s - 2 is finished.  The atomics go out BEFORE the next slab's gradient loads: the memory counter retires in order,
		// so the wait for those loads at the top of the next iteration also covers atomics that are an iteration old by then --
		// issued behind the loads they would be the youngest entries and every slab would wait for their round trip.  The
		// feature pieces are requested FIRST for the same reason: they are staged at the end of the iteration.
		// The two waves of a SIMD (w and w + 4) run W g^T at opposite ends of the iteration, so one's VALU / LDS phases lie beside
		// the other's matrix phases.  (One instance of every phase in program order, each behind a uniform branch: with the two
		// halves' sequences written as two arms the compiler allocates them separately and copies ~100 registers where they join.
		// Also measured and dropped: the upper half's W g^T and slab take-over as ONE basic block, the split's VALU instructions and
		// the transposing stores placed between the products with sched_group_barrier -- same time, profiles/r05_backward_fused.txt.)
		for (int s = 0; s < nsl; s++) {
			const int cur = s & 1;
			const bool more = s + 1 < nsl;
			SGS_PH(9)
			if (more) fetch_f(32 * (s + 1));
			SGS_PH(0)
			if (kh && s >= 2) finish_e(cur, 32 * (s - 2));
			if (kh && s >= 1) prod_e(cur ^ 1);
			SGS_PH(1)
			take_slab(cur);
			SGS_PH(2)
			if (!kh && s >= 2) finish_e(cur, 32 * (s - 2));
			SGS_PH(3)
			if (more) fetch_g(32 * (s + 1));
			SGS_PH(4)
			prod_d(cur);
			SGS_PH(5)
			if (!kh && s >= 1) prod_e(cur ^ 1);
			SGS_PH(6)
			if (more) stage_f(cur ^ 1);
			SGS_PH(7)
			__syncthreads();
			SGS_PH(8)
		}
		SGS_PH(9)
		// PERSIST, the tile's last chunk: the next tile's first round trip goes out HERE, behind the last slab's barrier -- the gradient's slab 0 (this
		// tile's last slab has been taken, its geometry is done with), the ids, the count words; the weight rows follow once this chunk's are dead
		uint32_t cs_next = 0u;
		if constexpr (PERSIST) {
			if ((ci + 1) * CHUNK >= total) {
				const int tk2 = __builtin_amdgcn_readfirstlane(s_tk);
				const int nt = band_lo + tk2;
				if (tk2 < band_n && ((uint32_t)nt + 1u) * 128u <= capacity) {
					next_tile = nt;
					cs_next = (uint32_t)nt * 128u;
					set_tile(nt);
					{   // (count words and ids FIRST: loads return in order, and the weight rows / feature pieces below wait for these, not for the gradient)
						const uint32_t* np = nact + nt;
						const uint2* rp = ranges + nt;
						asm volatile("" : "+v"(np), "+v"(rp));
						total_pre_v = *np;
						cb_pre_v = (rp->x >> 7) + (uint32_t)nt;
					}
					load_ids(cs_next);
					fetch_g(0);
				}
			}
		}
		if (nsl >= 2) finish_e(nsl & 1, 32 * (nsl - 2));
		SGS_PH(15)
		prod_e((nsl & 1) ^ 1);
		SGS_PH(16)
		if constexpr (PERSIST) {
			if (next_tile >= 0) {
				const int n_next = (int)__builtin_amdgcn_readfirstlane(total_pre_v);
				load_w(cs_next, n_next);
				set_rows(n_next < CHUNK ? n_next : CHUNK);   // (the ids have been under way for a slab's W g^T)
				fetch_f(0);
				pre = true;
			} else {
				// (no prefetch: the next chunk's top loads everything.  Written out so that the chunk's old values END here -- left to the
				// merge, the dead weight rows stayed live through the whole slab loop and the kernel spilled)
#pragma unroll
				for (int j = 0; j < 4; j++)
#pragma unroll
					for (int i = 0; i < 4; i++) wraw[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
				for (int j = 0; j < 16; j++) pg[j] = 0.f;
				my_id = f_id[0] = f_id[1] = 0u;
				pf[0] = pf[1] = make_float4(0.f, 0.f, 0.f, 0.f);
			}
		}
		__syncthreads();
		SGS_PH(17)
		finish_e((nsl & 1) ^ 1, 32 * (nsl - 1));
		SGS_PH(18)
#pragma unroll
		for (int m = 0; m < 4; m++)
			if (m < mb)
#pragma unroll
				for (int r = 0; r < 16; r++) {
					const int e = 32 * m + mfma_row(r, h);
					if (e < cnt) Drows[(size_t)(cstart + e) * 256 + 32 * wave + l31] = acc[m][r];
				}
	}
	if (PH) {
		ph_iters += (uint32_t)(((total + CHUNK - 1) / CHUNK) * nsl);
		ph_total += (uint32_t)total;
	}
	if (!PERSIST || next_tile < 0) break;
	tile = next_tile;
	}
	if (PH && trace) {
		if (PH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		SGS_PH(19)   // D rows written
		if (lane == 0) {
			unsigned long long* o = trace + ((size_t)b * 8 + wave) * 24;
#pragma unroll
			for (int k = 0; k < 20; k++) o[k] = ph[k];
			o[20] = (unsigned long long)ph_iters;   // iterations with a slab
			o[21] = (unsigned long long)wave | ((unsigned long long)ph_total << 8);
			o[22] = wall_clock64();
		}
	}
#undef SGS_PH
}
#undef SGS_MFMA_BF16
#undef SGS_MFMA_X16

struct StagedEntryG {
	float a2, b2, c2, o;
	float x, y;
	uint32_t id;
	float ca;
	float cb, cc;
	uint32_t slot, idx1;
};

// ---- 4. geometry gradients from D (lane = pixel, back to front)
__global__ __launch_bounds__(256) void bwd_geom_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const uint32_t* __restrict__ act_idx, const float* __restrict__ Drows,
	const float2* __restrict__ means2D, const float4* __restrict__ conic_opacity,
	const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
	float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
	const uint32_t* __restrict__ counter, int W, int H, int gx, int per_xcd, int ntiles)
{
	if (counter[1] != 0u) return;
	const int b = blockIdx.x;
	const int tile = (b & 7) * per_xcd + (b >> 3);
	if (tile >= ntiles) return;
	const int tx = tile % gx, ty = tile / gx;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ry = wave * 4 + (lane >> 4), rx = lane & 15;
	const int px = tx * SGS_TILE + rx, py = ty * SGS_TILE + ry;
	const bool inside = px < W && py < H;
	const int pxp = (ry & 1) * 128 + (ry >> 1) * 16 + rx;
	const float pxf = (float)px, pyf = (float)py;
	const size_t pix = (size_t)py * W + px;
	const uint32_t chunk_base = (ranges[tile].x >> 7) + (uint32_t)tile;
	const int total = (int)nact[tile];
	const int real = total - 1;   // the last entry is the background pseudo entry
	if (real <= 0) return;

	__shared__ StagedEntryG s_e[64];
	// per staged entry and wave: the six wave-reduced sums (mean2D x/y, conic xx/xy/yy, opacity).  They are added up
	// over the four waves and sent as ONE atomic per (entry, component), issued 64 lanes wide, after the block --
	// a lane-0 atomic per wave, entry and component (24 narrow device-scope atomics per entry) was what bound this
	// kernel.
	__shared__ float s_acc[4][64][6];

	const float T_final = inside ? final_Ts[pix] : 0.f;
	float T = T_final;
	const int last_contributor = inside ? (int)n_contrib[pix] : 0;
	int wave_max = last_contributor;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const int o = __shfl_xor(wave_max, off);
		wave_max = o > wave_max ? o : wave_max;
	}
	const uint32_t bg_slot = sgs_chunk_start(table, chunk_base, (uint32_t)tile, (uint32_t)real / CHUNK) + (uint32_t)real % CHUNK;
	const float bg_dot = Drows[(size_t)bg_slot * 256 + pxp];
	const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
	float R = 0.f;

	for (int hi = real; hi > 0; hi -= 64) {
		const int n = hi < 64 ? hi : 64;
		__syncthreads();
		for (int q = threadIdx.x; q < 4 * 64 * 6; q += 256) (&s_acc[0][0][0])[q] = 0.f;
		if ((int)threadIdx.x < n) {
			const uint32_t g = (uint32_t)(hi - 1 - (int)threadIdx.x);
			const uint32_t slot = sgs_chunk_start(table, chunk_base, (uint32_t)tile, g / CHUNK) + g % CHUNK;
			const uint32_t id = act_id[slot];
			const float2 xy = means2D[id];
			const float4 co = conic_opacity[id];
			StagedEntryG e;
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
			e.slot = slot;
			e.idx1 = act_idx[slot];
			s_e[threadIdx.x] = e;
		}
		__syncthreads();
		// Which staged entries can touch THIS wave's 4 x 16 pixels at all: lane l bounds entry l's power over the wave's rectangle (the
		// maximum of the quadratic over its edges when the centre lies outside -- blend_weights2.hip's tile-level rejection, on a quarter
		// tile) against the alpha test's threshold log(1 / (255 o)).  Conservative by 0.01 on every side, so a cleared bit is an entry whose
		// step would have left at its ballot after ~35 instructions with nothing changed; now the step is skipped by two scalar instructions
		// and its D row is not loaded.  One ballot per 64 entries and wave.  Measured (call AF): few of a tile's kept entries miss a whole quarter
		// tile -- 2.6 % fewer VALU instructions, 7 % fewer row loads, 3.5 % of the kernel's time (307 -> 296 us).
		unsigned long long hit;
		{
			bool keep = lane < n;
			if (keep) {
				const float ea = s_e[lane].a2, eb = s_e[lane].b2, ec = s_e[lane].c2;
				if (ea < 0.f && ec < 0.f && 4.f * ea * ec - eb * eb > 0.f) {
					const float thr = __logf(1.0f / (255.0f * s_e[lane].o)) - 0.01f;
					const float ex0 = s_e[lane].x, ey0 = s_e[lane].y;
					const float dxl = ex0 - (float)(tx * SGS_TILE + SGS_TILE - 1) - 0.01f;
					const float dxh = ex0 - (float)(tx * SGS_TILE) + 0.01f;
					const float dyl = ey0 - (float)(ty * SGS_TILE + wave * 4 + 3) - 0.01f;
					const float dyh = ey0 - (float)(ty * SGS_TILE + wave * 4) + 0.01f;
					if (!(dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f)) {
						float qmax = -__builtin_inff();
#pragma unroll
						for (int k = 0; k < 2; k++) {
							const float ex = k ? dxh : dxl;
							const float sy = fmin_(fmax_(-eb * ex / (2.f * ec), dyl), dyh);
							qmax = fmax_(qmax, ea * ex * ex + eb * ex * sy + ec * sy * sy);
							const float ey = k ? dyh : dyl;
							const float sx = fmin_(fmax_(-eb * ey / (2.f * ea), dxl), dxh);
							qmax = fmax_(qmax, ea * sx * sx + eb * sx * ey + ec * ey * ey);
						}
						keep = !(qmax < thr - 0.01f);
					}
				}
			}
			hit = __ballot(keep);
		}
		// One entry of the walk.  Dk (this pixel's D of the entry) arrives prefetched: the load is the only global
		// access of a step and its address does not depend on the recurrence, so a group of 8 is requested while the
		// previous group is processed (as written before -- one dependent load per step -- the kernel was bound by
		// that latency: 0.73 ms for 455 k entries).
		auto step = [&](const StagedEntryG& e, float Dk, int kslot) __attribute__((always_inline)) {
			if (!((hit >> kslot) & 1ull)) return;
			const int idx = (int)e.idx1 - 1;
			if (idx >= wave_max) return;
			const float dx = e.x - pxf, dy = e.y - pyf;
			const float power =
				__builtin_fmaf(e.b2 * dx, dy, __builtin_fmaf(e.c2 * dy, dy, (e.a2 * dx) * dx));
			const float G = expf_contract(power);
			const float alpha = fmin_(0.99f, e.o * G);
			const bool valid = inside && (idx < last_contributor) && !(power > 0.0f) &&
					   !(alpha < 1.0f / 255.0f);
			if (__ballot(valid) == 0ull) return;
			const float oma = 1.f - alpha;
			// one reciprocal for the two divisions by 1 - alpha (alpha <= 0.99: v_rcp_f32 + one Newton step is within an ulp of the
			// IEEE quotient, and this kernel's sums are order-dependent atomics anyway; two IEEE divisions were 18 of the step's
			// ~107 VALU instructions, and the kernel is bound by VALU issue: 195 M wave instructions x 4 cycles = its 0.35 ms.
			// A packed two-pixels-per-lane form was no faster: it loses this form's per-wave skips)
			float rinv = __builtin_amdgcn_rcpf(oma);
			rinv = __builtin_fmaf(__builtin_fmaf(-oma, rinv, 1.f), rinv, rinv);
			if (valid) T = T * rinv;
			float dL_dalpha = (Dk - R) * T;
			dL_dalpha += (-T_final * rinv) * bg_dot;
			if (valid) R = alpha * Dk + oma * R;
			// Round 6: the wave sums are the six MOMENTS of r = G dL/dalpha over the pixels -- sum r, r dx, r dy, r dx^2, r dx dy, r dy^2 -- and the
			// entry's constants (opacity, conic) meet them once per entry behind the block: every one of the six gradient sums is a combination of these
			//     dL/dmean.x = -W/2 o (ca Rx + cb Ry),  dL/dmean.y = -H/2 o (cc Ry + cb Rx),  dL/dconic = -o/2 (Rxx, Rxy, Ryy),  dL/do = R0
			// (7 VALU per step in front of the reduction instead of 22).
			const float r = valid ? G * dL_dalpha : 0.f;
			const float rx = r * dx, ry = r * dy;
			const float u = wave_sum6(rx, ry, rx * dx, rx * dy, ry * dy, r);
			const int comp = wave_sum8_component(lane);
			if ((lane & 7) == 0 && comp < 6) s_acc[wave][kslot][comp] = u;
		};
		constexpr int PF = 8;
		float dcur[PF], dnext[PF];
#pragma unroll
		for (int u = 0; u < PF; u++) dcur[u] = (u < n && ((hit >> u) & 1ull)) ? Drows[(size_t)s_e[u].slot * 256 + pxp] : 0.f;
		for (int k0 = 0; k0 < n; k0 += PF) {
#pragma unroll
			for (int u = 0; u < PF; u++) {
				const int kk = k0 + PF + u;
				dnext[u] = (kk < n && ((hit >> kk) & 1ull)) ? Drows[(size_t)s_e[kk].slot * 256 + pxp] : 0.f;
			}
#pragma unroll
			for (int u = 0; u < PF; u++)
				if (k0 + u < n) step(s_e[k0 + u], dcur[u], k0 + u);
#pragma unroll
			for (int u = 0; u < PF; u++) dcur[u] = dnext[u];
		}
		__syncthreads();
		for (int q = threadIdx.x; q < n * 6; q += 256) {
			const int e = q / 6, c = q - 6 * e;
			auto mom = [&](int k) __attribute__((always_inline)) { return (s_acc[0][e][k] + s_acc[1][e][k]) + (s_acc[2][e][k] + s_acc[3][e][k]); };
			const StagedEntryG& E = s_e[e];
			float v;
			if (c == 0) v = -ddelx_dx * E.o * (E.ca * mom(0) + E.cb * mom(1));
			else if (c == 1) v = -ddely_dy * E.o * (E.cc * mom(1) + E.cb * mom(0));
			else if (c < 5) v = -0.5f * E.o * mom(c);
			else v = mom(5);
			if (v != 0.f) {
				const size_t id = s_e[e].id;
				float* dst = c < 2 ? dL_dmean2D + 3 * id + c : (c < 5 ? dL_dconic + 4 * id + (c == 4 ? 3 : c - 2) : dL_dopacity + id);
				atomicAdd(dst, v);
			}
		}
	}
}

} // namespace

#ifdef SGS_WITH_EXPERIMENTS
static const int g_bwd_dbg = getenv("SGS_BWD_DBG") ? atoi(getenv("SGS_BWD_DBG")) : 0;   // ablations / phase stamps of the fused kernel (tools/bwd_phases.py)
#endif
// the fused kernel as persistent workgroups that prefetch the next tile's first round trip (SGS_BWD_PERSIST=0: one workgroup per tile)
static const bool g_bwd_persist = getenv("SGS_BWD_PERSIST") ? atoi(getenv("SGS_BWD_PERSIST")) != 0 : true;
static int bwd_persist_groups()   // one 8-wave workgroup per compute unit, a multiple of the 8 XCDs
{
	static const int n = [] {
		int dev = 0;
		hipDeviceProp_t pr;
		if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
		const int cu = pr.multiProcessorCount;
		return cu >= 8 ? (cu / 8) * 8 : 8;
	}();
	return n;
}

int bwd_fused_x16_ownership()   // (blend_sweep2.hip: x16_kernel_owns_cu)
{
	static const int own = (x16_kernel_owns_cu((const void*)&bwd_fused_kernel<false>, "bwd_fused_kernel<false>") &&
				x16_kernel_owns_cu((const void*)&bwd_fused_kernel<false, 0, true>, "bwd_fused_kernel<false,0,true>"))
				       ? 1
				       : 0;
	return own;
}

bool blend_backward_mfma_eligible(const BlendBwdArgs& a)
{
	return a.C >= 32 && (a.C & 31) == 0 && (((uintptr_t)a.colors | (uintptr_t)a.bg) & 15u) == 0 &&
	       ((uintptr_t)a.dL_dpix & 15u) == 0 && (size_t)a.H * a.W * 128 * 4 < ((size_t)1 << 32);   // 32-bit slab offsets
}

hipError_t launch_blend_backward_mfma(hipStream_t st, const BlendBwdArgs& a, char* arena, const SplitArena& lay,
				      bool fp32_products, size_t clear_dcolor_floats, bool two_kernels)
{
	const int ntiles = a.gx * a.gy;
	hipError_t e = launch_blend_weights_rows(st, a.ranges, a.point_list, a.means2D, a.conic_opacity,
						 const_cast<float*>(a.final_T), const_cast<uint32_t*>(a.n_contrib),
						 arena, lay, a.W, a.H, a.gx, a.gy, clear_dcolor_floats ? a.dL_dcolors : nullptr,
						 clear_dcolor_floats, a.tile_order);
	// (The pre-pass takes its tiles longest-first by the forward's own work-list lengths, like the forward's.  The
	// kernels below do NOT: measured, that order costs them their locality -- neighbouring tiles share feature rows and
	// colour-gradient rows in an XCD's L2 -- dcolor 1.10 -> 1.30 ms, dot 0.86 -> 1.07, profiles/r04_backward_tile_order.txt)
	if (e != hipSuccess) return e;
	const uint32_t* counter = (const uint32_t*)(arena + lay.counter);
	const uint32_t* nact = (const uint32_t*)(arena + lay.nbatches);
	const uint32_t* table = (const uint32_t*)(arena + lay.table);
	const uint32_t* act_id = (const uint32_t*)(arena + lay.act_id);
	const uint32_t* act_idx = (const uint32_t*)(arena + lay.act_idx);
	float* rows = (float*)(arena + lay.wgt);
	const int nch = (a.C + 127) / 128;
	const int items = ntiles * nch;
	const int ixcd = (items + 7) / 8, txcd = (ntiles + 7) / 8;
	const bool vec = (a.W & 3) == 0;   // 16-byte loads of the gradient rows (the two-kernel form)
	(void)vec; (void)ixcd; (void)items;
	if (!two_kernels) {   // round 5: one kernel, one read of the gradient for both products
		const dim3 grid(txcd * 8), block(512);
		uint32_t* tickets = (uint32_t*)(arena + lay.counter) + 2;
#define SGS_FUSED_ARGS a.ranges, table, nact, act_id, rows, a.colors, a.bg, a.dL_dpix, rows, a.dL_dcolors, counter, lay.capacity, a.W, a.H, a.C, a.gx, txcd, ntiles, \
		       get_sweep_trace(), tickets
#ifdef SGS_WITH_EXPERIMENTS
		if (g_bwd_dbg != 0 && !fp32_products) {
			switch (g_bwd_dbg) {
#define SGS_DBG_CASE(D_) case D_: hipLaunchKernelGGL((bwd_fused_kernel<false, D_>), grid, block, 0, st, SGS_FUSED_ARGS); break;
			SGS_DBG_CASE(2) SGS_DBG_CASE(4) SGS_DBG_CASE(8) SGS_DBG_CASE(15) SGS_DBG_CASE(16)
#undef SGS_DBG_CASE
			case 48: hipLaunchKernelGGL((bwd_fused_kernel<false, 16, true>), dim3(bwd_persist_groups()), block, 0, st, SGS_FUSED_ARGS); break;   // stamps, persistent form
			default: break;
			}
		} else
#endif
		if (fp32_products || !bwd_fused_x16_ownership()) {   // (fp32 products: backward mode 3; also what runs when the x16 form would not own its CU)
			hipLaunchKernelGGL((bwd_fused_kernel<true>), grid, block, 0, st, SGS_FUSED_ARGS);
		} else if (g_bwd_persist && (int)grid.x > bwd_persist_groups()) {
			hipLaunchKernelGGL((bwd_fused_kernel<false, 0, true>), dim3(bwd_persist_groups()), block, 0, st, SGS_FUSED_ARGS);
		} else {
			hipLaunchKernelGGL((bwd_fused_kernel<false>), grid, block, 0, st, SGS_FUSED_ARGS);
		}
#undef SGS_FUSED_ARGS
	} else {
#ifdef SGS_WITH_EXPERIMENTS
#define SGS_LAUNCH_BWD(DCOL_, DOT_)                                                                              \
	hipLaunchKernelGGL(DCOL_, dim3(ixcd * 8), dim3(512), 0, st, a.ranges, table, nact, act_id,                 \
			   rows, a.dL_dpix, a.dL_dcolors, counter, a.W, a.H, a.C, a.gx, nch, ixcd, items);        \
	hipLaunchKernelGGL(DOT_, dim3(txcd * 8), dim3(512), 0, st, a.ranges, table, nact, act_id, a.colors, a.bg,  \
			   a.dL_dpix, rows, counter, a.W, a.H, a.C, a.gx, txcd, ntiles)
	if (fp32_products) {
		if (vec) { SGS_LAUNCH_BWD(bwd_dcolor_kernel<true>, bwd_dot_kernel<true>); }
		else { SGS_LAUNCH_BWD(bwd_dcolor_kernel<false>, bwd_dot_kernel<false>); }
	} else {
		if (vec) { SGS_LAUNCH_BWD(bwd_dcolor_split_kernel<true>, bwd_dot_split_kernel); }
		else { SGS_LAUNCH_BWD(bwd_dcolor_split_kernel<false>, bwd_dot_split_kernel); }
	}
#undef SGS_LAUNCH_BWD
#else
		return hipErrorInvalidValue;   // (the two-kernel form is not in the product library: make EXPERIMENTS=1)
#endif
	}
	hipLaunchKernelGGL(bwd_geom_kernel, dim3(txcd * 8), dim3(256), 0, st, a.ranges, table, nact, act_id, act_idx,
			   rows, a.means2D, a.conic_opacity, a.final_T, a.n_contrib, a.dL_dmean2D, a.dL_dconic,
			   a.dL_dopacity, counter, a.W, a.H, a.gx, txcd, ntiles);
	e = hipGetLastError();
	if (e != hipSuccess) return e;
	return launch_blend_backward(st, a, counter);   // runs only if the work list overflowed
}

} // namespace sgs
