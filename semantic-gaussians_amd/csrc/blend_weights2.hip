// blend_weights2.hip -- round 3: the weights pre-pass of the C >= 128 forward blend (and of the work-list backward) with
// a lane owning TWO horizontally adjacent pixels.  Same job, same work-list formats and the same bits as
// blend_weights_kernel (blend_fwd_split.hip): walk the tile's sorted list exactly like the reference's renderCUDA
// (CR/cuda_rasterizer/forward.cu:300-364: power, alpha, the three skips, T update, n_contrib, final_T) and emit, per list
// entry that contributes to at least one pixel, the 256 blend weights w = alpha * T and the Gaussian id.
//
// Why: round 2's kernel (lane = pixel, four waves per tile) was bound by instruction ISSUE, not by arithmetic -- 108 M
// VALU and 86 M SALU instructions per cfg3 frame, SQ_WAIT_ANY 53 % (profiles/r02m_blend_pmc.txt).  Everything that is
// per-wave rather than per-pixel -- the staged entry's LDS reads, the wave-uniform candidate test and its branches, loop
// control, the activity bookkeeping -- is paid once per 64 pixels there and once per 128 here, and the per-pixel chain
// runs as packed fp32 instructions (v_pk_mul / v_pk_add / v_pk_fma_f32: two pixels per issue slot; the arithmetic
// contract is unchanged -- packed operations are the same IEEE operations).  The weights of a batch stay in registers
// until the batch's activity mask is known, so the round trip through a 16 KB LDS tile is gone as well and twelve
// workgroups of two waves fit a CU.
//
// Workgroup = one tile = two waves; wave p owns row parity p (image rows y = 2 rp + p, rp = lane >> 3), lane owns pixels
// x = 2 (lane & 7), + 1: its two weights of an entry are ADJACENT in the row-parity-major pixel order of the work list
// (px' = p * 128 + rp * 16 + x), one 8-byte store.
#include "sgs_kernels.h"

namespace sgs {

namespace {

constexpr int WB = 16;    // list entries per batch
#ifndef SGS_W2SB_GROUP
#define SGS_W2SB_GROUP 16   // kept entries per group of the super-batch kernel (A/B builds: 8 -> 80 VGPRs, 6 waves per SIMD)
#endif
constexpr int ACH = 128;  // work-list slots per chunk
constexpr uint32_t SGS_BG_ID = 0xFFFFFFFFu;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct StagedEntry2 {   // 40 B per kept list entry in LDS
	float a2, b2, c2, o;
	float x, y;
	uint32_t id;
	float thr;      // prefilter: no pixel with power < thr can pass the alpha test
	uint32_t idx1;  // 1-based position in the tile's list (n_contrib bookkeeping)
	uint32_t pad;
};

__device__ __forceinline__ void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// expf_contract (sgs_device.h) on a pair, instruction for instruction
__device__ __forceinline__ f32x2 expf_contract2(f32x2 x)
{
	const float LOG2E = 1.44269504088896341f, LN2_HI = 0.693145751953125f, LN2_LO = 1.42860682030941723e-6f, MAGIC = 12582912.0f;
	x = f32x2{fmax_(x.x, -87.0f), fmax_(x.y, -87.0f)};
	const f32x2 t = x * LOG2E;
	f32x2 nf = t + MAGIC;
	asm volatile("" : "+v"(nf));   // keep the compiler from re-associating (t + M) - M
	const f32x2 n = nf - MAGIC;
	f32x2 r = __builtin_elementwise_fma(n, f32x2{-LN2_HI, -LN2_HI}, x);
	r = __builtin_elementwise_fma(n, f32x2{-LN2_LO, -LN2_LO}, r);
	f32x2 p = {0.008182921446859837f, 0.008182921446859837f};
	p = __builtin_elementwise_fma(p, r, f32x2{0.04184672236442566f, 0.04184672236442566f});
	p = __builtin_elementwise_fma(p, r, f32x2{0.16668450832366943f, 0.16668450832366943f});
	p = __builtin_elementwise_fma(p, r, f32x2{0.4999966621398926f, 0.4999966621398926f});
	p = __builtin_elementwise_fma(p, r, f32x2{1.0f, 1.0f});
	p = __builtin_elementwise_fma(p, r, f32x2{1.0f, 1.0f});
	return f32x2{__uint_as_float(__float_as_uint(p.x) + (__float_as_uint(nf.x) << 23)),
		     __uint_as_float(__float_as_uint(p.y) + (__float_as_uint(nf.y) << 23))};
}

// entries 8 gi .. 8 gi + 7 of a tile are complete in the LDS tile pend[8][256]: split this lane's two pixel columns into
// three exact bf16 terms each (w = t1 + t2 + t3: the fp32 difference of a value and its own rounding is exact) and store
// them k-major.  A real call, on purpose: inlined at the sixteen emission sites it costs the kernel two waves per SIMD.
// Round 4: the lane does NOT split its own two columns (px' = 2 lane, 2 lane + 1 of the wave's 128: 16 bytes at a 32-byte stride,
// a 128-byte line completed by two instructions) but columns lane and 64 + lane of its wave's half: every store instruction
// writes 1 KB of whole lines, which is what the streaming (nt) hint needs (with the old mapping the hint cost 0.47 instead of
// 0.26 ms, profiles/r04_worklist_nt_stores.txt).  The columns were written by other lanes of the SAME wave a moment ago: the
// LDS unit executes a wave's instructions in order, no barrier is involved.
__device__ __noinline__ void flush_group3(const float* pend, uint4* dst, int pxq)
{
#pragma unroll 1
	for (int i = 0; i < 2; i++) {
		uint32_t t1[4], t2[4], t3[4];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const float x0 = pend[(2 * k) * 256 + pxq + 64 * i], x1 = pend[(2 * k + 1) * 256 + pxq + 64 * i];
			const f32x2 v0 = {x0, x1};
			t1[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2));
			const float r0 = x0 - __uint_as_float(t1[k] << 16), r1 = x1 - __uint_as_float(t1[k] & 0xffff0000u);
			const f32x2 v1 = {r0, r1};
			t2[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2));
			const f32x2 v2 = {r0 - __uint_as_float(t2[k] << 16), r1 - __uint_as_float(t2[k] & 0xffff0000u)};
			t3[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2));
		}
		store_nt(&dst[pxq + 64 * i], make_uint4(t1[0], t1[1], t1[2], t1[3]));
		store_nt(&dst[256 + pxq + 64 * i], make_uint4(t2[0], t2[1], t2[2], t2[3]));
		store_nt(&dst[512 + pxq + 64 * i], make_uint4(t3[0], t3[1], t3[2], t3[3]));
	}
}

} // namespace

// MODE 3: fp32 rows [slot][256 px'] (exact sweep, backward); MODE 4: three bf16 terms, per group of 8 slots
// [term][256 px'][8 x bf16] (the default six-product sweep, blend_sweep2.hip)
template <int MODE>
__global__ __launch_bounds__(128, 5) void blend_weights2_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float2* __restrict__ means2D, const float4* __restrict__ conic_opacity,
	float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
	uint32_t* __restrict__ act_id, uint32_t* __restrict__ act_idx, float* __restrict__ wgt,
	uint32_t* __restrict__ table, uint32_t* __restrict__ nact, uint32_t* __restrict__ counter,
	uint32_t capacity, int W, int H, int gx, int per_xcd, int ntiles,
	float4* __restrict__ clear_ptr, unsigned long long clear_n4, const uint32_t* __restrict__ tile_order)
{
	static_assert(MODE == 3 || MODE == 4, "weights format");
	const int b = blockIdx.x;
	// (backward, SGS_OPT_BWD_CLEARS_DCOLOR) this workgroup's slice of the gradient buffer, 32 KB per batch of the tile's
	// walk and the rest at the end: issued in one burst at the start, every resident workgroup was in its store phase at
	// the same time and the clear's 0.37 ms simply added to the kernel (0.22 -> 0.59 ms at cfg3)
	unsigned long long ci = 0, ci1 = 0;
	if (clear_ptr) {
		const unsigned long long per = (clear_n4 + gridDim.x - 1) / gridDim.x;
		const unsigned long long i0 = (unsigned long long)b * per;
		ci1 = i0 + per < clear_n4 ? i0 + per : clear_n4;
		ci = i0 + threadIdx.x;
	}
	auto clear_some = [&](int k) {
		for (int q = 0; q < k && ci < ci1; q++, ci += 128) clear_ptr[ci] = make_float4(0.f, 0.f, 0.f, 0.f);
	};
	auto clear_rest = [&]() {
		for (; ci < ci1; ci += 128) clear_ptr[ci] = make_float4(0.f, 0.f, 0.f, 0.f);
	};
	// (round 4) tiles longest-first by the work they had in the stream's previous frame when that order exists
	// (BlendFwdArgs::tile_order, blend_fwd_split.hip), XCD bands otherwise
	const int tile = (tile_order && tile_order[0] == (uint32_t)ntiles) ? (b < ntiles ? (int)tile_order[1 + b] : ntiles)
										  : (b & 7) * per_xcd + (b >> 3);
	if (tile >= ntiles || counter[1] == 2u) {   // padding workgroup / aborted frame (the lists do not exist)
		clear_rest();
		return;
	}
	const int tx = tile % gx, ty = tile / gx;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = row parity
	const int rp = lane >> 3, xp = (lane & 7) * 2;
	const int px = tx * SGS_TILE + xp, py = ty * SGS_TILE + 2 * rp + wave;
	const bool in0 = px < W && py < H, in1 = px + 1 < W && py < H;
	const int pxp = wave * 128 + rp * 16 + xp;   // this lane's first pixel in row-parity-major order (even)
	const f32x2 pxf = {(float)px, (float)(px + 1)};
	const float pyf = (float)py;
	const uint2 range = ranges[tile];
	const int n_total = (int)(range.y - range.x);
	const uint32_t chunk_base = (range.x >> 7) + (uint32_t)tile;

	__shared__ StagedEntry2 s_e[WB];
	__shared__ uint32_t s_amask[2];    // per batch parity (double-buffered: no barrier between its last read and the next clear)
	__shared__ int s_nkeep;
	__shared__ int s_alive[2];
	__shared__ uint32_t s_ovf;
	__shared__ uint32_t s_chunk[64];
	__shared__ float s_pend[MODE == 4 ? 8 * 256 : 2];   // MODE 4: the entry group being filled, [k][px']

	auto chunk_start = [&](uint32_t ci) -> uint32_t { return ci < 64 ? s_chunk[ci] : table[chunk_base + ci]; };
	auto flush_group = [&](uint32_t gi) {   // MODE 4: the group in s_pend is complete
		const uint32_t g0 = gi * 8u;
		const uint32_t slot = chunk_start(g0 / ACH) + (g0 % ACH);
		flush_group3(s_pend, reinterpret_cast<uint4*>(reinterpret_cast<char*>(wgt) + (size_t)(slot >> 3) * 12288), wave * 128 + lane);
	};
	// one work-list entry's two weights of this lane into slot position g of the tile (tile-uniform g)
	auto emit = [&](uint32_t g, f32x2 w) {
		if (MODE == 4) {
			*reinterpret_cast<f32x2*>(&s_pend[(g & 7u) * 256 + pxp]) = w;
			if ((g & 7u) == 7u) flush_group(g >> 3);
		} else {
			const uint32_t slot = chunk_start(g / ACH) + (g % ACH);
			store_nt(reinterpret_cast<f32x2*>(wgt + (size_t)slot * 256 + pxp), w);
		}
	};

	f32x2 T = {1.0f, 1.0f};
	uint32_t last0 = 0, last1 = 0;
	bool done0 = !in0, done1 = !in1;
	uint32_t total = 0;     // active entries emitted so far (tile-uniform)
	uint32_t nchunks = 0;   // chunks reserved so far (tile-uniform)
	if (threadIdx.x == 0) {
		s_ovf = 0u;
		s_amask[0] = s_amask[1] = 0u;
	}

	// staging runs one batch ahead in registers (lanes < WB of wave 0), as in blend_weights_kernel
	uint32_t pf_id = 0u, pf_id_next = 0u;
	float2 pf_xy = make_float2(0.f, 0.f);
	float4 pf_co = make_float4(0.f, 0.f, 0.f, 0.f);
	if ((int)threadIdx.x < WB) {
		if ((int)threadIdx.x < n_total) {
			pf_id = point_list[range.x + threadIdx.x];
			pf_xy = means2D[pf_id];
			pf_co = conic_opacity[pf_id];
		}
		if (WB + (int)threadIdx.x < n_total) pf_id_next = point_list[range.x + WB + threadIdx.x];
	}

	int batch = 0;
	for (int base = 0; base < n_total; base += WB, batch++) {
		const bool wave_alive = __ballot(!(done0 && done1)) != 0ull;
		if (lane == 0) s_alive[wave] = wave_alive ? 1 : 0;
		lds_barrier2();   // also: the previous batch's s_e / s_amask reads are over
		if (!(s_alive[0] | s_alive[1])) break;
		const int n = (n_total - base) < WB ? (n_total - base) : WB;
		if ((int)threadIdx.x < WB) {   // (16 lanes of wave 0)
			StagedEntry2 e;
			bool keep = false;
			const uint32_t id = pf_id;
			const float2 xy = pf_xy;
			const float4 co = pf_co;
			if (base + WB + (int)threadIdx.x < n_total) {
				pf_id = pf_id_next;
				pf_xy = means2D[pf_id];
				pf_co = conic_opacity[pf_id];
			}
			if (base + 2 * WB + (int)threadIdx.x < n_total)
				pf_id_next = point_list[range.x + base + 2 * WB + threadIdx.x];
			if ((int)threadIdx.x < n) {
				e.a2 = -0.5f * co.x;
				e.b2 = -co.y;
				e.c2 = -0.5f * co.z;
				e.o = co.w;
				e.x = xy.x;
				e.y = xy.y;
				e.id = id;
				e.idx1 = (uint32_t)(base + (int)threadIdx.x + 1);
				e.pad = 0u;
				// prefilter threshold and tile-level rejection: exactly blend_weights_kernel's (both are conservative
				// skips of work whose result is provably "no pixel takes the entry")
				e.thr = __logf(1.0f / (255.0f * co.w)) - 0.01f;
				keep = true;
				if (e.a2 < 0.f && e.c2 < 0.f && 4.f * e.a2 * e.c2 - e.b2 * e.b2 > 0.f) {
					const float dxl = xy.x - (float)(tx * SGS_TILE + SGS_TILE - 1) - 0.01f;
					const float dxh = xy.x - (float)(tx * SGS_TILE) + 0.01f;
					const float dyl = xy.y - (float)(ty * SGS_TILE + SGS_TILE - 1) - 0.01f;
					const float dyh = xy.y - (float)(ty * SGS_TILE) + 0.01f;
					if (!(dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f)) {
						float qmax = -__builtin_inff();
#pragma unroll
						for (int k = 0; k < 2; k++) {
							const float ex = k ? dxh : dxl;
							const float sy = fmin_(fmax_(-e.b2 * ex / (2.f * e.c2), dyl), dyh);
							qmax = fmax_(qmax, e.a2 * ex * ex + e.b2 * ex * sy + e.c2 * sy * sy);
							const float ey = k ? dyh : dyl;
							const float sx = fmin_(fmax_(-e.b2 * ey / (2.f * e.a2), dxl), dxh);
							qmax = fmax_(qmax, e.a2 * sx * sx + e.b2 * sx * ey + e.c2 * ey * ey);
						}
						keep = !(qmax < e.thr - 0.01f);
					}
				}
			}
			const uint32_t km = (uint32_t)__ballot(keep);
			const int nk = __popc(km);
			if (keep) s_e[__popc(km & ((1u << threadIdx.x) - 1u))] = e;
			if (threadIdx.x == 0) {
				s_nkeep = nk;
				s_amask[(batch + 1) & 1] = 0u;   // the NEXT batch's mask word (nobody reads or writes it before the next barrier pair)
			}
		}
		lds_barrier2();
		const int nkeep = s_nkeep;
		clear_some(16);
		// ---- weight phase: this wave's 128 pixels for the whole batch; the weights stay in registers
		f32x2 w[WB];
		uint32_t act = 0u;
#pragma unroll
		for (int j = 0; j < WB; j++) {
			w[j] = f32x2{0.f, 0.f};
			if (j < nkeep && wave_alive) {   // (uniform)
				const StagedEntry2 e = s_e[j];
				const f32x2 dx = f32x2{e.x, e.x} - pxf;
				const float dy = e.y - pyf;
				const float cdy = e.c2 * dy;
				const f32x2 adx = dx * e.a2;
				const f32x2 t2 = adx * dx;
				const f32x2 t4 = __builtin_elementwise_fma(f32x2{cdy, cdy}, f32x2{dy, dy}, t2);
				const f32x2 bdx = dx * e.b2;
				const f32x2 power = __builtin_elementwise_fma(bdx, f32x2{dy, dy}, t4);
				const bool pre0 = !(power.x > 0.0f) && !(power.x < e.thr), pre1 = !(power.y > 0.0f) && !(power.y < e.thr);
				const bool c0 = !done0 && pre0, c1 = !done1 && pre1;
				if (__ballot(c0 || c1) != 0ull) {
					const f32x2 ex = expf_contract2(power);
					const f32x2 oe = ex * e.o;
					const f32x2 alpha = {fmin_(0.99f, oe.x), fmin_(0.99f, oe.y)};
					const f32x2 test_T = T * (f32x2{1.0f, 1.0f} - alpha);
					const bool cand0 = c0 && !(alpha.x < 1.0f / 255.0f), cand1 = c1 && !(alpha.y < 1.0f / 255.0f);
					const bool stop0 = cand0 && (test_T.x < 0.0001f), stop1 = cand1 && (test_T.y < 0.0001f);
					const bool take0 = cand0 && !stop0, take1 = cand1 && !stop1;
					done0 = done0 || stop0;
					done1 = done1 || stop1;
					const f32x2 aT = alpha * T;
					w[j] = f32x2{take0 ? aT.x : 0.f, take1 ? aT.y : 0.f};
					T = f32x2{take0 ? test_T.x : T.x, take1 ? test_T.y : T.y};
					last0 = take0 ? e.idx1 : last0;
					last1 = take1 ? e.idx1 : last1;
					if (__ballot(take0 || take1) != 0ull) act |= 1u << j;
				}
			}
		}
		if (lane == 0 && act != 0u) atomicOr(&s_amask[batch & 1], act);
		lds_barrier2();
		// ---- emission: every thread derives the same slots from the combined activity mask
		{
			const uint32_t amask = s_amask[batch & 1];
			const uint32_t cnt = (uint32_t)__popc(amask);
			if (nchunks * ACH < total + cnt) {   // (tile-uniform) the batch crosses into a new 128-slot chunk
				if (threadIdx.x == 0) {
					uint32_t nc = nchunks;
					while (nc * ACH < total + cnt && s_ovf == 0u) {
						const uint32_t start = nc == 0 ? (uint32_t)tile * ACH : atomicAdd(&counter[0], (uint32_t)ACH);
						if (start + ACH > capacity) {   // arena overflow: flag it, emit nothing more
							atomicExch(&counter[1], 1u);
							s_ovf = 1u;
							break;
						}
						if (nc != 0) table[chunk_base + nc] = start;   // chunk 0 is implicit (sgs_chunk_start)
						if (nc < 64) s_chunk[nc] = start;
						nc++;
					}
				}
				__syncthreads();
				nchunks = (total + cnt + ACH - 1) / ACH;
			}
			if (s_ovf == 0u) {
#pragma unroll
				for (int j = 0; j < WB; j++) {
					if ((amask >> j) & 1u) {   // (uniform)
						const uint32_t g = total + (uint32_t)__popc(amask & ((1u << j) - 1u));
						emit(g, w[j]);
						if (threadIdx.x == 0) {
							const uint32_t slot = chunk_start(g / ACH) + (g % ACH);
							act_id[slot] = s_e[j].id;
							if (act_idx) act_idx[slot] = s_e[j].idx1;   // (backward: position in the tile's list)
						}
					}
				}
			}
			total += cnt;
		}
	}
	clear_rest();
	// ---- the closing T * bg pseudo entry (every tile gets one, also an empty tile), zero padding to a batch of 16
	__syncthreads();
	if (threadIdx.x == 0 && nchunks * ACH < total + 1u && s_ovf == 0u) {
		const uint32_t start = nchunks == 0 ? (uint32_t)tile * ACH : atomicAdd(&counter[0], (uint32_t)ACH);
		if (start + ACH > capacity) {
			atomicExch(&counter[1], 1u);
			s_ovf = 1u;
		} else {
			if (nchunks != 0) table[chunk_base + nchunks] = start;
			if (nchunks < 64) s_chunk[nchunks] = start;
		}
	}
	__syncthreads();
	if (s_ovf == 0u) {
		const uint32_t g = total;
		emit(g, f32x2{in0 ? T.x : 0.0f, in1 ? T.y : 0.0f});
		if (threadIdx.x == 0) act_id[chunk_start(g / ACH) + (g % ACH)] = SGS_BG_ID;
		const uint32_t pad_end = (g + 1u + 15u) & ~15u;
		for (uint32_t q = g + 1u; q < pad_end; q++) emit(q, f32x2{0.f, 0.f});
	}
	total += 1u;
	if (threadIdx.x == 0) nact[tile] = total;
	const size_t pix = (size_t)py * W + px;
	if (in0) {
		final_T[pix] = T.x;
		n_contrib[pix] = last0;
	}
	if (in1) {
		final_T[pix + 1] = T.y;
		n_contrib[pix + 1] = last1;
	}
}

// Round 4: the kernel above with blend_weights_sb_kernel's walk (blend_fwd_split.hip): one list entry per thread per
// 128-entry super-batch (gathers prefetched a super-batch ahead, the tile-level rejection 128 wide), the kept entries compacted
// in list order, the weight phase over groups of 16 KEPT entries -- two barriers per 128 list entries plus one per group.
// Same arithmetic, same entry order, same work list (bit-identical frames).  Three-term format only (the forward's default).
template <int GB>   // GB = kept entries per group (their weights stay in 2 GB registers of the lane)
__global__ __launch_bounds__(128, GB == 8 ? 6 : 5) void blend_weights2_sb_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float2* __restrict__ means2D, const float4* __restrict__ conic_opacity,
	float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
	uint32_t* __restrict__ act_id, float* __restrict__ wgt,
	uint32_t* __restrict__ table, uint32_t* __restrict__ nact, uint32_t* __restrict__ counter,
	uint32_t capacity, int W, int H, int gx, int per_xcd, int ntiles, const uint32_t* __restrict__ tile_order)
{
	constexpr int SB = 128;   // list entries staged at a time (one per thread)
	const int b = blockIdx.x;
	const int tile = (tile_order && tile_order[0] == (uint32_t)ntiles) ? (b < ntiles ? (int)tile_order[1 + b] : ntiles)
										  : (b & 7) * per_xcd + (b >> 3);
	if (tile >= ntiles || counter[1] == 2u) return;
	const int tx = tile % gx, ty = tile / gx;
	const int t = threadIdx.x, lane = t & 63;
	const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // = row parity
	const int rp = lane >> 3, xp = (lane & 7) * 2;
	const int px = tx * SGS_TILE + xp, py = ty * SGS_TILE + 2 * rp + wave;
	const bool in0 = px < W && py < H, in1 = px + 1 < W && py < H;
	const int pxp = wave * 128 + rp * 16 + xp;
	const f32x2 pxf = {(float)px, (float)(px + 1)};
	const float pyf = (float)py;
	const uint2 range = ranges[tile];
	const int n_total = (int)(range.y - range.x);
	const uint32_t chunk_base = (range.x >> 7) + (uint32_t)tile;

	__shared__ StagedEntry2 s_e[SB];
	__shared__ uint32_t s_amask[SB / GB];
	__shared__ int s_cnt[2], s_alive[2];
	__shared__ uint32_t s_ovf;
	__shared__ uint32_t s_chunk[64];
	__shared__ float s_pend[8 * 256];

	auto chunk_start = [&](uint32_t ci) -> uint32_t { return ci < 64 ? s_chunk[ci] : table[chunk_base + ci]; };
	auto emit = [&](uint32_t g, f32x2 w) {
		*reinterpret_cast<f32x2*>(&s_pend[(g & 7u) * 256 + pxp]) = w;
		if ((g & 7u) == 7u) {
			const uint32_t g0 = g & ~7u;
			const uint32_t slot = chunk_start(g0 / ACH) + (g0 % ACH);
			flush_group3(s_pend, reinterpret_cast<uint4*>(reinterpret_cast<char*>(wgt) + (size_t)(slot >> 3) * 12288), wave * 128 + lane);
		}
	};
	auto reserve = [&](uint32_t have, uint32_t upto) {   // thread 0: the chunks for entries [0, upto) exist afterwards (or s_ovf is set)
		uint32_t nc = have;
		while (nc * ACH < upto && s_ovf == 0u) {
			const uint32_t start = nc == 0 ? (uint32_t)tile * ACH : atomicAdd(&counter[0], (uint32_t)ACH);
			if (start + ACH > capacity) {
				atomicExch(&counter[1], 1u);
				s_ovf = 1u;
				break;
			}
			if (nc != 0) table[chunk_base + nc] = start;
			if (nc < 64) s_chunk[nc] = start;
			nc++;
		}
	};

	f32x2 T = {1.0f, 1.0f};
	uint32_t last0 = 0, last1 = 0;
	bool done0 = !in0, done1 = !in1;
	uint32_t total = 0, nchunks = 0;
	if (t == 0) s_ovf = 0u;

	uint32_t pf_id = 0u, pf_id_next = 0u;
	float2 pf_xy = make_float2(0.f, 0.f);
	float4 pf_co = make_float4(0.f, 0.f, 0.f, 0.f);
	if (t < n_total) {
		pf_id = point_list[range.x + t];
		pf_xy = means2D[pf_id];
		pf_co = conic_opacity[pf_id];
	}
	if (SB + t < n_total) pf_id_next = point_list[range.x + SB + t];
	const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));

	for (int base = 0; base < n_total; base += SB) {
		const uint32_t id = pf_id;
		const float2 xy = pf_xy;
		const float4 co = pf_co;
		if (base + SB + t < n_total) {
			pf_id = pf_id_next;
			pf_xy = means2D[pf_id];
			pf_co = conic_opacity[pf_id];
		}
		if (base + 2 * SB + t < n_total) pf_id_next = point_list[range.x + base + 2 * SB + t];
		StagedEntry2 e;
		bool keep = false;
		if (base + t < n_total) {
			e.a2 = -0.5f * co.x;
			e.b2 = -co.y;
			e.c2 = -0.5f * co.z;
			e.o = co.w;
			e.x = xy.x;
			e.y = xy.y;
			e.id = id;
			e.idx1 = (uint32_t)(base + t + 1);
			e.pad = 0u;
			e.thr = __logf(1.0f / (255.0f * co.w)) - 0.01f;   // (prefilter and tile-level rejection: exactly blend_weights_kernel's)
			keep = true;
			if (e.a2 < 0.f && e.c2 < 0.f && 4.f * e.a2 * e.c2 - e.b2 * e.b2 > 0.f) {
				const float dxl = xy.x - (float)(tx * SGS_TILE + SGS_TILE - 1) - 0.01f;
				const float dxh = xy.x - (float)(tx * SGS_TILE) + 0.01f;
				const float dyl = xy.y - (float)(ty * SGS_TILE + SGS_TILE - 1) - 0.01f;
				const float dyh = xy.y - (float)(ty * SGS_TILE) + 0.01f;
				if (!(dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f)) {
					float qmax = -__builtin_inff();
#pragma unroll
					for (int k = 0; k < 2; k++) {
						const float ex = k ? dxh : dxl;
						const float sy = fmin_(fmax_(-e.b2 * ex / (2.f * e.c2), dyl), dyh);
						qmax = fmax_(qmax, e.a2 * ex * ex + e.b2 * ex * sy + e.c2 * sy * sy);
						const float ey = k ? dyh : dyl;
						const float sx = fmin_(fmax_(-e.b2 * ey / (2.f * e.a2), dxl), dxh);
						qmax = fmax_(qmax, e.a2 * sx * sx + e.b2 * sx * ey + e.c2 * ey * ey);
					}
					keep = !(qmax < e.thr - 0.01f);
				}
			}
		}
		const unsigned long long km = __ballot(keep);
		const bool wave_alive = __ballot(!(done0 && done1)) != 0ull;
		if (lane == 0) {
			s_cnt[wave] = __popcll(km);
			s_alive[wave] = wave_alive ? 1 : 0;
		}
		lds_barrier2();   // B1 (also: the previous super-batch's copy-out has finished reading s_e / its mask words)
		if (!(s_alive[0] | s_alive[1])) break;
		const int c0 = s_cnt[0], nkeep = c0 + s_cnt[1];
		if (keep) s_e[(wave ? c0 : 0) + __popcll(km & below)] = e;
		if (t < SB / GB) s_amask[t] = 0u;
		lds_barrier2();   // B2
		for (int g0 = 0, gi = 0; g0 < nkeep; g0 += GB, gi++) {
			const int ng = (nkeep - g0) < GB ? (nkeep - g0) : GB;
			// ---- weight phase (the kernel above's): this wave's 128 pixels against the group's entries, weights in registers
			f32x2 w[GB];
			uint32_t act = 0u;
			const bool alive = __ballot(!(done0 && done1)) != 0ull;
#pragma unroll
			for (int j = 0; j < GB; j++) {
				w[j] = f32x2{0.f, 0.f};
				if (j < ng && alive) {   // (uniform)
					const StagedEntry2 se = s_e[g0 + j];
					const f32x2 dx = f32x2{se.x, se.x} - pxf;
					const float dy = se.y - pyf;
					const float cdy = se.c2 * dy;
					const f32x2 adx = dx * se.a2;
					const f32x2 t2 = adx * dx;
					const f32x2 t4 = __builtin_elementwise_fma(f32x2{cdy, cdy}, f32x2{dy, dy}, t2);
					const f32x2 bdx = dx * se.b2;
					const f32x2 power = __builtin_elementwise_fma(bdx, f32x2{dy, dy}, t4);
					const bool pre0 = !(power.x > 0.0f) && !(power.x < se.thr), pre1 = !(power.y > 0.0f) && !(power.y < se.thr);
					const bool q0 = !done0 && pre0, q1 = !done1 && pre1;
					if (__ballot(q0 || q1) != 0ull) {
						const f32x2 ex = expf_contract2(power);
						const f32x2 oe = ex * se.o;
						const f32x2 alpha = {fmin_(0.99f, oe.x), fmin_(0.99f, oe.y)};
						const f32x2 test_T = T * (f32x2{1.0f, 1.0f} - alpha);
						const bool cand0 = q0 && !(alpha.x < 1.0f / 255.0f), cand1 = q1 && !(alpha.y < 1.0f / 255.0f);
						const bool stop0 = cand0 && (test_T.x < 0.0001f), stop1 = cand1 && (test_T.y < 0.0001f);
						const bool take0 = cand0 && !stop0, take1 = cand1 && !stop1;
						done0 = done0 || stop0;
						done1 = done1 || stop1;
						const f32x2 aT = alpha * T;
						w[j] = f32x2{take0 ? aT.x : 0.f, take1 ? aT.y : 0.f};
						T = f32x2{take0 ? test_T.x : T.x, take1 ? test_T.y : T.y};
						last0 = take0 ? se.idx1 : last0;
						last1 = take1 ? se.idx1 : last1;
						if (__ballot(take0 || take1) != 0ull) act |= 1u << j;
					}
				}
			}
			if (lane == 0 && act != 0u) atomicOr(&s_amask[gi], act);
			lds_barrier2();   // B3: the group's mask is complete
			const uint32_t amask = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_amask[gi]);
			const uint32_t cnt = (uint32_t)__popc(amask);
			if (nchunks * ACH < total + cnt) {   // (tile-uniform) the group crosses into a new 128-slot chunk
				if (t == 0) reserve(nchunks, total + cnt);
				__syncthreads();
				nchunks = (total + cnt + ACH - 1) / ACH;
			}
			if (s_ovf == 0u) {
#pragma unroll
				for (int j = 0; j < GB; j++) {
					if ((amask >> j) & 1u) {   // (uniform)
						const uint32_t g = total + (uint32_t)__popc(amask & ((1u << j) - 1u));
						emit(g, w[j]);
						if (t == 0) act_id[chunk_start(g / ACH) + (g % ACH)] = s_e[g0 + j].id;
					}
				}
			}
			total += cnt;
		}
	}
	// ---- the closing T * bg pseudo entry (every tile gets one, also an empty tile), zero padding to a batch of 16
	__syncthreads();
	if (t == 0 && nchunks * ACH < total + 1u) reserve(nchunks, total + 1u);
	__syncthreads();
	if (s_ovf == 0u) {
		const uint32_t g = total;
		emit(g, f32x2{in0 ? T.x : 0.0f, in1 ? T.y : 0.0f});
		if (t == 0) act_id[chunk_start(g / ACH) + (g % ACH)] = SGS_BG_ID;
		const uint32_t pad_end = (g + 1u + 15u) & ~15u;
		for (uint32_t q = g + 1u; q < pad_end; q++) emit(q, f32x2{0.f, 0.f});
	}
	total += 1u;
	if (t == 0) nact[tile] = total;
	const size_t pix = (size_t)py * W + px;
	if (in0) {
		final_T[pix] = T.x;
		n_contrib[pix] = last0;
	}
	if (in1) {
		final_T[pix + 1] = T.y;
		n_contrib[pix + 1] = last1;
	}
}

hipError_t launch_blend_weights2(hipStream_t st, int mode, const uint2* ranges, const uint32_t* point_list,
				 const float2* means2D, const float4* conic_opacity, float* final_T, uint32_t* n_contrib,
				 uint32_t* act_id, uint32_t* act_idx, float* wgt, uint32_t* table, uint32_t* nact, uint32_t* counter,
				 uint32_t capacity, int W, int H, int gx, int ntiles, float* clear_ptr, size_t clear_floats,
				 const uint32_t* tile_order)
{
	const dim3 grid(((ntiles + 7) / 8) * 8);
	if (mode == 5)   // three bf16 terms, 128-entry super-batches (the forward's default pre-pass)
		hipLaunchKernelGGL(blend_weights2_sb_kernel<SGS_W2SB_GROUP>, grid, dim3(128), 0, st, ranges, point_list, means2D, conic_opacity, final_T,
				   n_contrib, act_id, wgt, table, nact, counter, capacity, W, H, gx, (ntiles + 7) / 8, ntiles, tile_order);
#ifdef SGS_WITH_EXPERIMENTS   // round 3's 16-entry-batch kernel in the three-term format (variant bits 0xC000)
	else if (mode == 4)
		hipLaunchKernelGGL(blend_weights2_kernel<4>, grid, dim3(128), 0, st, ranges, point_list, means2D, conic_opacity, final_T,
				   n_contrib, act_id, act_idx, wgt, table, nact, counter, capacity, W, H, gx, (ntiles + 7) / 8, ntiles,
				   (float4*)clear_ptr, (unsigned long long)(clear_floats / 4), tile_order);
#else
	else if (mode == 4) return hipErrorInvalidValue;
#endif
	else
		hipLaunchKernelGGL(blend_weights2_kernel<3>, grid, dim3(128), 0, st, ranges, point_list, means2D, conic_opacity, final_T,
				   n_contrib, act_id, act_idx, wgt, table, nact, counter, capacity, W, H, gx, (ntiles + 7) / 8, ntiles,
				   (float4*)clear_ptr, (unsigned long long)(clear_floats / 4), tile_order);
	return hipGetLastError();
}

} // namespace sgs
