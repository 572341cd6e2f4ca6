// blend_fwd_split.hip -- the C>=128 headline path: the forward blend as TWO kernels.
//
//   1. blend_weights_kernel   one workgroup per 16x16 tile, lane = pixel.  Walks the tile's
//      sorted list exactly like the reference's renderCUDA (CR/cuda_rasterizer/forward.cu:
//      300-364: power, alpha, the three skips, T update, n_contrib, final_T) but instead of
//      touching any feature it emits, per list entry that contributes to at least one pixel,
//      the 256 blend weights w = alpha*T (0 for pixels that skip it) and the Gaussian id, into
//      a compact "work list" in HBM.  This is the only place exp() and the sequential
//      transmittance chain are evaluated: once per tile instead of once per channel chunk.
//
//   2. blend_accum_sweep_kernel   a pure streaming weighted sum  out[ch][px] = sum_k F[k][ch] * W[k][px]
//      over the work list, i.e. a matrix product per tile, as tile-row sweeps with full-line stores:
//        <.., false> (default)          split-bf16 MFMA products, fp32 accumulate;
//        <.., true>  (SGS_BLEND_EXACT)  fp32-input MFMA, bit-identical to the contract.
//      (Measured and removed on the way, see DESIGN.md 5 and the git history: scalar-fed and LDS-fed VALU
//      forms, per-tile fp32-MFMA kernels with 2- and 3-stage rings, a per-tile and a tile-pair
//      split-bf16 kernel.)
//
// The fp32 path is bit-identical to the single-kernel paths (same contract arithmetic, same
// accumulation order; adding w = 0 is exact); the default's tolerance is derived in DESIGN.md 5.2.
//
// Work-list storage ("arena") is carved from the binning buffer.  A tile's entries are kept
// contiguous in chunks of 128 slots, bump-allocated with one atomicAdd per chunk (most tiles
// need one); chunk c of tile t starts at table[(range.x >> 7) + t + c] -- that index is
// collision free without any scan -- and nact[t] is the tile's entry count.
// If the arena overflows, a device flag makes the accumulate kernel exit immediately and the
// single-kernel path (launched right after it) take over, so no host round trip is needed;
// the host grows the arena for the next frame.
#include "sgs_kernels.h"
#include <type_traits>

namespace sgs {

namespace {

struct StagedEntryW {   // 40 B per list entry in LDS
	float a2, b2, c2, o;
	float x, y;
	uint32_t id;
	float thr;      // prefilter: no pixel with power < thr can pass the alpha test
	uint32_t idx1;  // 1-based position in the tile's list (n_contrib bookkeeping)
	uint32_t pad;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL
// store (s_waitcnt vmcnt(0)): in the weights kernel that parks all four waves on the HBM round trip of the
// weight rows written a moment earlier, three times per batch.  The barriers between the phases of a batch
// only hand LDS data from wave to wave.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int WB = 16;    // list entries per batch
constexpr int ACH = 128;  // work-list slots per chunk
constexpr uint32_t SGS_BG_ID = 0xFFFFFFFFu;   // work-list id of the closing T * bg pseudo entry

} // namespace

// Work-list weight formats.  Common to both: pixels in row-parity-major order
// px' = (y & 1) * 128 + (y >> 1) * 16 + x (the rows of one parity are a contiguous half: a sweep
// workgroup reads only its own); every tile's list ends with a pseudo entry whose "weights" are the
// pixels' final transmittance and whose id is SGS_BG_ID (the accumulate kernel feeds the background
// vector as its feature row, so  + T * bg  falls out of the matrix product); the tile's last 16-entry
// batch is padded with zero weights.
// MODE 2 (default): weights split into bf16 hi + bf16 lo (w = hi + lo + O(2^-16 w)), k-major for the bf16
//                   MFMA's B operand: per group of 8 consecutive entries [256 px'][8 x hi] (4 KB) then
//                   [256 px'][8 x lo] (4 KB).
// MODE 3 (exact):   fp32 rows, [entry][256 px'] (1 KB per entry).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void blend_weights_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float2* __restrict__ means2D, const float4* __restrict__ conic_opacity,
	float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
	uint32_t* __restrict__ act_id, uint32_t* __restrict__ act_idx, float* __restrict__ wgt,
	uint32_t* __restrict__ table, uint32_t* __restrict__ nact, uint32_t* __restrict__ counter,
	uint32_t capacity, int W, int H, int gx, int per_xcd, int ntiles, unsigned long long* __restrict__ trace,
	float4* __restrict__ clear_ptr, unsigned long long clear_n4, const uint32_t* __restrict__ tile_order)
{
	const int b = blockIdx.x;
	const unsigned long long t_begin = trace ? wall_clock64() : 0ull;   // (debug timeline, tools/sweep_trace.py)
	if (clear_ptr) {   // (backward, SGS_OPT_BWD_CLEARS_DCOLOR) this workgroup's slice of the gradient buffer: the stores
		// drain under the instruction-bound work below
		const unsigned long long per = (clear_n4 + gridDim.x - 1) / gridDim.x;
		const unsigned long long i0 = (unsigned long long)b * per;
		const unsigned long long i1 = i0 + per < clear_n4 ? i0 + per : clear_n4;
		for (unsigned long long i = i0 + threadIdx.x; i < i1; i += 256) clear_ptr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	}
	static_assert(MODE == 2 || MODE == 3 || MODE == 4, "weights format: 2 = two bf16 terms, 3 = fp32 rows, 4 = three bf16 terms");
	constexpr bool BF = MODE == 2 || MODE == 4;   // weights as bf16 terms, k-major groups of 8 (else fp32 rows of 256)
	constexpr int GROUP_BYTES = MODE == 4 ? 12288 : 8192;   // 8 entries x 256 px x (2 | 3) terms x 2 B
	constexpr bool SWEEP = true;     // parity-major pixel order, closing T * bg pseudo entry, zero padding to 16
	// Which tile: by default XCD b % 8 owns a contiguous band of tiles.  A tile's work (its number of ACTIVE list entries)
	// cannot be predicted from its list (round 2: r = 0.006 with the list length), so in that order the kernel ends with
	// a long tail (1 206 of 1 536 workgroup slots busy on average at cfg3).  Round 4: when this stream has rendered a frame
	// of the same tile grid before, the tiles are taken longest-first by the work-list length they had in THAT frame
	// (tile_order, written by the previous frame's sweep_plan_kernel): consecutive views of a scene have nearly the same
	// per-tile work, and a stale order is only a worse schedule, never a different result.
	int tile;
	if (tile_order && tile_order[0] == (uint32_t)ntiles) {
		if (b >= ntiles) return;
		tile = (int)tile_order[1 + b];
	} else {
		tile = (b & 7) * per_xcd + (b >> 3);
		if (tile >= ntiles) return;
	}
	if (counter[1] == 2u) return;   // aborted frame (arena_reset_kernel): the lists do not exist
	const int tx = tile % gx, ty = tile / gx;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int px = tx * SGS_TILE + (lane & 15);
	const int py = ty * SGS_TILE + wave * 4 + (lane >> 4);
	const bool inside = px < W && py < H;
	// this pixel's index in row-parity-major order (the sweep kernels' layout)
	const int pxp_own = ((wave * 4 + (lane >> 4)) & 1) * 128 + ((wave * 4 + (lane >> 4)) >> 1) * 16 + (lane & 15);
	const float pxf = (float)px, pyf = (float)py;
	const uint2 range = ranges[tile];
	const int n_total = (int)(range.y - range.x);
	const uint32_t chunk_base = (range.x >> 7) + (uint32_t)tile;

	__shared__ StagedEntryW s_e[WB];
	__shared__ float s_wt[WB * 256];       // [entry][strip*64 + lane]
	__shared__ uint32_t s_amask;           // entries of the batch taken by at least one pixel
	__shared__ int s_nkeep;                // entries of the batch that survive the tile-level rejection
	__shared__ int s_alive[4];
	__shared__ uint32_t s_ovf;
	__shared__ uint32_t s_chunk[64];   // first slot of each chunk (re-read from `table` beyond 64)
	__shared__ float s_pend[BF ? 8 * 256 : 1];   // BF: the entry group being filled, [k][px]

	// BF: entries 8*gi .. 8*gi+7 of this tile are complete in s_pend -> split and store them.
	// Thread p only ever touches column p of s_pend, so no barrier is involved.
	auto flush_group = [&](uint32_t gi) {
		const uint32_t g0 = gi * 8u, ci = g0 / ACH;
		const uint32_t cstart = ci < 64 ? s_chunk[ci] : table[chunk_base + ci];   // (ci >= 64 > 0: a table entry)
		const uint32_t slot = cstart + (g0 % ACH);
		if constexpr (MODE == 4) {
			// three terms, w = t1 + t2 + t3 EXACTLY (8 significant bits each; the fp32 difference of a value and its own
			// rounding is exact): what the sweep's six products need to be fp32-equivalent (blend_sweep2.hip)
			uint32_t t1[4], t2[4], t3[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const float x0 = s_pend[(2 * k) * 256 + threadIdx.x], x1 = s_pend[(2 * k + 1) * 256 + threadIdx.x];
				typedef float f32x2_ __attribute__((ext_vector_type(2)));
				typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
				const f32x2_ v0 = {x0, x1};
				t1[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2_));
				const float r0 = x0 - __uint_as_float(t1[k] << 16), r1 = x1 - __uint_as_float(t1[k] & 0xffff0000u);
				const f32x2_ v1 = {r0, r1};
				t2[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2_));
				const f32x2_ v2 = {r0 - __uint_as_float(t2[k] << 16), r1 - __uint_as_float(t2[k] & 0xffff0000u)};
				t3[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2_));
			}
			uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(wgt) + (size_t)(slot >> 3) * GROUP_BYTES);
			store_nt(&dst[pxp_own], make_uint4(t1[0], t1[1], t1[2], t1[3]));
			store_nt(&dst[256 + pxp_own], make_uint4(t2[0], t2[1], t2[2], t2[3]));
			store_nt(&dst[512 + pxp_own], make_uint4(t3[0], t3[1], t3[2], t3[3]));
			return;
		}
		bf16x8 hi, lo;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const float v = s_pend[k * 256 + threadIdx.x];
			hi[k] = (__bf16)v;
			lo[k] = (__bf16)(v - (float)hi[k]);
		}
		bf16x8* dst = reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(wgt) + (size_t)(slot >> 3) * 8192);
		store_nt(&dst[pxp_own], hi);
		store_nt(&dst[256 + pxp_own], lo);
	};

	float T = 1.0f;
	uint32_t last = 0;
	bool done = !inside;
	uint32_t total = 0;     // active entries emitted so far (tile-uniform)
	uint32_t nchunks = 0;   // chunks reserved so far (tile-uniform)
	if (threadIdx.x == 0) s_ovf = 0u;

	// Staging runs one batch ahead in registers (lanes < WB): at the top of batch b the Gaussian
	// data of batch b is complete, the gathers for batch b + 1 are issued with the ids fetched during
	// batch b - 1, and the ids of batch b + 2 are requested -- the dependent point_list -> means2D /
	// conic_opacity round trips overlap the weight phase instead of preceding it.
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

	for (int base = 0; base < n_total; base += WB) {
		const bool wave_alive = __ballot(!done) != 0ull;
		if (lane == 0) s_alive[wave] = wave_alive ? 1 : 0;
		lds_barrier();   // also: previous batch's copy-out has finished reading LDS
		const int alive = s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3];
		if (!alive) break;
		const int n = (n_total - base) < WB ? (n_total - base) : WB;
		if ((int)threadIdx.x < WB) {   // (the first 32 lanes of wave 0)
			StagedEntryW e;
			bool keep = false;
			// this batch's data (prefetched), then the next batch's gathers and the ids after that
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
				// prefilter threshold: alpha = o * exp(power) >= 1/255 needs power >= ln(1 / (255 o)).
				// 1 % below it (the contract exp is good to 5 ulp, __logf to ~1e-6) a pixel
				// provably fails the alpha test; anything else goes through the exact path.
				e.thr = __logf(1.0f / (255.0f * co.w)) - 0.01f;
				// tile-level rejection: the exact maximum of the (concave) quadratic form over the
				// tile's pixel box; below the threshold no pixel of the tile can take the entry.
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
							const float ex = k ? dxh : dxl;   // edge dx = ex
							const float sy = fmin_(fmax_(-e.b2 * ex / (2.f * e.c2), dyl), dyh);
							qmax = fmax_(qmax, e.a2 * ex * ex + e.b2 * ex * sy + e.c2 * sy * sy);
							const float ey = k ? dyh : dyl;   // edge dy = ey
							const float sx = fmin_(fmax_(-e.b2 * ey / (2.f * e.a2), dxl), dxh);
							qmax = fmax_(qmax, e.a2 * sx * sx + e.b2 * sx * ey + e.c2 * ey * ey);
						}
						keep = !(qmax < e.thr - 0.01f);
					}
				}
			}
			// compact the kept entries (order preserved), pad to a multiple of 4 with null entries
			const uint32_t km = (uint32_t)__ballot(keep);
			const int nk = __popc(km);
			if (keep) s_e[__popc(km & ((1u << threadIdx.x) - 1u))] = e;
			if ((int)threadIdx.x < ((nk + 3) & ~3) - nk) {
				StagedEntryW z;
				z.a2 = z.b2 = z.c2 = z.o = z.x = z.y = 0.f;
				z.id = 0u;
				z.idx1 = 0u;
				z.thr = __builtin_inff();
				s_e[nk + threadIdx.x] = z;
			}
			if (threadIdx.x == 0) s_nkeep = nk;
		}
		if (threadIdx.x == 0) s_amask = 0u;
		lds_barrier();
		const int nkeep = s_nkeep;   // entries of this batch some pixel of the tile might take
		// ---- weight phase: wave w evaluates strip w for the whole batch.  The quadratic form of
		// four entries is evaluated together (independent, ILP); an entry then runs the exp /
		// alpha / transmittance chain only if some pixel of the strip can pass the alpha test
		// (wave-uniform branch) -- for ~60 % of the (strip, entry) pairs none can.
		if (wave_alive) {
			uint32_t act = 0u;   // entries of this batch taken by some pixel of this strip
			const int n4 = (nkeep + 3) & ~3;
			for (int j0 = 0; j0 < n4; j0 += 4) {
				float power[4], opac[4];
				uint32_t idx1[4];
				bool pre[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const StagedEntryW e = s_e[j0 + u];
					const float dx = e.x - pxf, dy = e.y - pyf;
					power[u] = __builtin_fmaf(e.b2 * dx, dy,
								  __builtin_fmaf(e.c2 * dy, dy, (e.a2 * dx) * dx));
					opac[u] = e.o;
					idx1[u] = e.idx1;
					pre[u] = !(power[u] > 0.0f) && !(power[u] < e.thr);
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int j = j0 + u;
					const bool cand0 = !done && pre[u];
					float w = 0.0f;
					if (__ballot(cand0) != 0ull) {
						const float alpha = fmin_(0.99f, opac[u] * expf_contract(power[u]));
						const float test_T = T * (1.0f - alpha);
						const bool cand = cand0 && !(alpha < 1.0f / 255.0f);
						const bool stop = cand && (test_T < 0.0001f);
						const bool take = cand && !stop;
						done = done || stop;
						if (take) {
							w = alpha * T;
							T = test_T;
							last = idx1[u];
						}
						if (__ballot(take) != 0ull) act |= 1u << j;
					}
					s_wt[j * 256 + wave * 64 + lane] = w;
				}
			}
			if (lane == 0 && act != 0u) atomicOr(&s_amask, act);
		} else {
			for (int j = 0; j < nkeep; j++) s_wt[j * 256 + wave * 64 + lane] = 0.0f;
		}
		lds_barrier();
		// ---- compaction into the tile's contiguous chunks.  Every thread derives the same counts from
		// the activity mask; only when the batch crosses into a new 128-slot chunk (about once per tile)
		// does thread 0 reserve it and a barrier publish the chunk start.
		{
			const uint32_t amask = s_amask;
			const uint32_t cnt = (uint32_t)__popc(amask);
			if (nchunks * ACH < total + cnt) {   // (tile-uniform)
				if (threadIdx.x == 0) {
					uint32_t nc = nchunks;
					while (nc * ACH < total + cnt && s_ovf == 0u) {
						// a tile's FIRST chunk is pre-assigned (slot tile * 128; the bump counter starts behind
						// those): every tile needs one, and most need no other -- no returning device-scope
						// atomic on the tile's critical path
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
				uint32_t m = amask;
				for (uint32_t r = 0; r < cnt; r++) {
					const int e = __builtin_ctz(m);
					m &= m - 1;
					const uint32_t g = total + r, ci = g / ACH;
					const uint32_t cstart = ci < 64 ? s_chunk[ci] : table[chunk_base + ci];   // (ci >= 64 > 0: a table entry)
					const uint32_t slot = cstart + (g % ACH);
					if (BF) {
						s_pend[(g & 7u) * 256 + threadIdx.x] = s_wt[e * 256 + threadIdx.x];
						if ((g & 7u) == 7u) flush_group(g >> 3);
					} else {
						store_nt(&wgt[(size_t)slot * 256 + (SWEEP ? pxp_own : (int)threadIdx.x)], s_wt[e * 256 + threadIdx.x]);
					}
					if (threadIdx.x == 0) {
						act_id[slot] = s_e[e].id;
						if (act_idx) act_idx[slot] = s_e[e].idx1;   // (backward: position in the tile's list)
					}
				}
			}
			total += cnt;
		}
	}
	if (SWEEP) {   // the closing T * bg pseudo entry (every tile gets one, also an empty tile)
		__syncthreads();
		if (threadIdx.x == 0 && nchunks * ACH < total + 1u && s_ovf == 0u) {
			const uint32_t start = nchunks == 0 ? (uint32_t)tile * ACH : atomicAdd(&counter[0], (uint32_t)ACH);
			if (start + ACH > capacity) {
				atomicExch(&counter[1], 1u);
				s_ovf = 1u;
			} else {
				if (nchunks != 0) table[chunk_base + nchunks] = start;
				if (nchunks < 64) s_chunk[nchunks] = start;
				nchunks++;
			}
		}
		__syncthreads();
		if (s_ovf == 0u) {
			const uint32_t g = total, ci = g / ACH;
			const uint32_t cstart = ci < 64 ? s_chunk[ci] : table[chunk_base + ci];   // (ci >= 64 > 0: a table entry)
			const float wT = inside ? T : 0.0f;
			if (BF) {
				s_pend[(g & 7u) * 256 + threadIdx.x] = wT;
				if ((g & 7u) == 7u) flush_group(g >> 3);
			} else {
				store_nt(&wgt[(size_t)(cstart + (g % ACH)) * 256 + pxp_own], wT);
			}
			if (threadIdx.x == 0) act_id[cstart + (g % ACH)] = SGS_BG_ID;
		}
		total += 1u;
	}
	if (SWEEP && s_ovf == 0u) {   // zero-pad the last batch to 16 entries (all inside the tile's last chunk)
		const uint32_t pad_end = (total + 15u) & ~15u;
		for (uint32_t g = total; g < pad_end; g++) {
			if (BF) {
				s_pend[(g & 7u) * 256 + threadIdx.x] = 0.0f;
				if ((g & 7u) == 7u) flush_group(g >> 3);
			} else {
				const uint32_t ci = g / ACH;
				const uint32_t cstart = ci < 64 ? s_chunk[ci] : table[chunk_base + ci];   // (ci >= 64 > 0: a table entry)
				store_nt(&wgt[(size_t)(cstart + (g % ACH)) * 256 + pxp_own], 0.0f);
			}
		}
	}
	if (threadIdx.x == 0) nact[tile] = total;
	if (inside) {
		const size_t pix = (size_t)py * W + px;
		final_T[pix] = T;
		n_contrib[pix] = last;
	}
	if (trace && threadIdx.x == 0) {
		trace[4 * (size_t)b] = t_begin;
		trace[4 * (size_t)b + 1] = wall_clock64();
		trace[4 * (size_t)b + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
					   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
		trace[4 * (size_t)b + 3] = (unsigned long long)total | ((unsigned long long)(ranges[tile].y - ranges[tile].x) << 32);
	}
}

// Round 4: the same pre-pass with the list walked 256 entries at a time ("super-batch").  Why: at cfg3 a tile's list has
// ~3 300 entries of which ~100 end up active; the kernel above pays three barriers, an LDS staging round trip and ~30
// scalar instructions of loop control per 16 LIST entries, nearly all of them for batches in which the tile-level test
// rejects everything (PMC, round 3: 85.7 M SALU beside 111 M VALU, 55 % of the wave-cycles waiting).  Here every thread
// stages ONE list entry (gathers prefetched a super-batch ahead, the tile-level rejection evaluated 256 wide), the kept
// entries are compacted in list order (ballot + a four-word wave prefix), and only THEY go through the weight phase, 16 at a
// time as before: two barriers per 256 list entries plus one per 16 kept entries.  A group's 16 weights of a pixel never leave
// the thread's registers (the thread that evaluated pixel p is the one that copies column p out; the kernel above parks them
// in 16 KB of LDS), so between a group's weight phase and its copy-out only the activity mask crosses waves -- one barrier per
// group, a mask word per group.
// Arithmetic, order of the entries, work-list contents: identical to the kernel above (bit-identical frames).
template <int MODE>
__global__ __launch_bounds__(256) void blend_weights_sb_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float2* __restrict__ means2D, const float4* __restrict__ conic_opacity,
	float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
	uint32_t* __restrict__ act_id, uint32_t* __restrict__ act_idx, float* __restrict__ wgt,
	uint32_t* __restrict__ table, uint32_t* __restrict__ nact, uint32_t* __restrict__ counter,
	uint32_t capacity, int W, int H, int gx, int per_xcd, int ntiles, const uint32_t* __restrict__ tile_order)
{
	static_assert(MODE == 3 || MODE == 4, "weights format: 3 = fp32 rows, 4 = three bf16 terms");
	constexpr bool BF = MODE == 4;
	constexpr int GROUP_BYTES = 12288;
	constexpr int SB = 256;   // list entries staged at a time (one per thread)
	const int b = blockIdx.x;
	int tile;
	if (tile_order && tile_order[0] == (uint32_t)ntiles) {   // longest-first by the previous frame's work (see above)
		if (b >= ntiles) return;
		tile = (int)tile_order[1 + b];
	} else {
		tile = (b & 7) * per_xcd + (b >> 3);
		if (tile >= ntiles) return;
	}
	if (counter[1] == 2u) return;   // aborted frame
	const int tx = tile % gx, ty = tile / gx;
	const int t = threadIdx.x, lane = t & 63;
	const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int px = tx * SGS_TILE + (lane & 15);
	const int py = ty * SGS_TILE + wave * 4 + (lane >> 4);
	const bool inside = px < W && py < H;
	const int pxp_own = ((wave * 4 + (lane >> 4)) & 1) * 128 + ((wave * 4 + (lane >> 4)) >> 1) * 16 + (lane & 15);
	const float pxf = (float)px, pyf = (float)py;
	const uint2 range = ranges[tile];
	const int n_total = (int)(range.y - range.x);
	const uint32_t chunk_base = (range.x >> 7) + (uint32_t)tile;

	__shared__ StagedEntryW s_e[SB];          // the super-batch's kept entries, list order
	__shared__ uint32_t s_amask[SB / WB];     // per group: entries taken by at least one pixel
	__shared__ int s_cnt[4], s_alive[4];
	__shared__ uint32_t s_ovf;
	__shared__ uint32_t s_chunk[64];
	__shared__ float s_pend[BF ? 8 * 256 : 1];

	auto chunk_start_of = [&](uint32_t ci) -> uint32_t { return ci < 64 ? s_chunk[ci] : table[chunk_base + ci]; };
	auto flush_group = [&](uint32_t gi) {   // entries 8 gi .. 8 gi + 7 of the tile are complete in s_pend: split and store
		const uint32_t g0 = gi * 8u;
		const uint32_t slot = chunk_start_of(g0 / ACH) + (g0 % ACH);
		uint32_t t1[4], t2[4], t3[4];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const float x0 = s_pend[(2 * k) * 256 + t], x1 = s_pend[(2 * k + 1) * 256 + t];
			typedef float f32x2_ __attribute__((ext_vector_type(2)));
			typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
			const f32x2_ v0 = {x0, x1};
			t1[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2_));
			const float r0 = x0 - __uint_as_float(t1[k] << 16), r1 = x1 - __uint_as_float(t1[k] & 0xffff0000u);
			const f32x2_ v1 = {r0, r1};
			t2[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2_));
			const f32x2_ v2 = {r0 - __uint_as_float(t2[k] << 16), r1 - __uint_as_float(t2[k] & 0xffff0000u)};
			t3[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2_));
		}
		uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(wgt) + (size_t)(slot >> 3) * GROUP_BYTES);
		store_nt(&dst[pxp_own], make_uint4(t1[0], t1[1], t1[2], t1[3]));
		store_nt(&dst[256 + pxp_own], make_uint4(t2[0], t2[1], t2[2], t2[3]));
		store_nt(&dst[512 + pxp_own], make_uint4(t3[0], t3[1], t3[2], t3[3]));
	};
	// one work-list entry (tile-uniform position g) from this thread's weight w
	auto emit = [&](uint32_t g, float w) {
		if (BF) {
			s_pend[(g & 7u) * 256 + t] = w;
			if ((g & 7u) == 7u) flush_group(g >> 3);
		} else {
			store_nt(&wgt[(size_t)(chunk_start_of(g / ACH) + (g % ACH)) * 256 + pxp_own], w);
		}
	};
	// thread 0: make sure the chunks for entries [0, upto) exist; returns false after an arena overflow
	auto reserve = [&](uint32_t& nchunks, uint32_t upto) {
		uint32_t nc = nchunks;
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

	float T = 1.0f;
	uint32_t last = 0;
	bool done = !inside;
	uint32_t total = 0, nchunks = 0;   // (tile-uniform) active entries emitted, chunks reserved
	if (t == 0) s_ovf = 0u;

	// gathers run a super-batch ahead, the ids two
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
		StagedEntryW e;
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
			e.thr = __logf(1.0f / (255.0f * co.w)) - 0.01f;   // (the prefilter and the tile-level rejection: see the kernel above)
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
		const bool wave_alive = __ballot(!done) != 0ull;
		if (lane == 0) {
			s_cnt[wave] = __popcll(km);
			s_alive[wave] = wave_alive ? 1 : 0;
		}
		lds_barrier();   // B1 (also: the previous super-batch's copy-out has finished reading s_e / its mask words)
		if (!(s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3])) break;
		const int c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
		const int nkeep = c0 + c1 + c2 + c3;
		const int off = wave == 0 ? 0 : (wave == 1 ? c0 : (wave == 2 ? c0 + c1 : c0 + c1 + c2));
		if (keep) s_e[off + __popcll(km & below)] = e;
		if (t < ((nkeep + 3) & ~3) - nkeep) {   // pad to a multiple of 4 with null entries
			StagedEntryW z;
			z.a2 = z.b2 = z.c2 = z.o = z.x = z.y = 0.f;
			z.id = 0u;
			z.idx1 = 0u;
			z.pad = 0u;
			z.thr = __builtin_inff();
			s_e[nkeep + t] = z;
		}
		if (t < SB / WB) s_amask[t] = 0u;
		lds_barrier();   // B2
		for (int g0 = 0, gi = 0; g0 < nkeep; g0 += WB, gi++) {
			const int ng = (nkeep - g0) < WB ? (nkeep - g0) : WB;
			// ---- weight phase (as above): this thread's pixel against the group's entries; the 16 weights stay in registers
			f32x16 wv;
#pragma unroll
			for (int j = 0; j < WB; j++) wv[j] = 0.0f;
			if (__ballot(!done) != 0ull) {
				uint32_t act = 0u;
				const int n4 = (ng + 3) & ~3;
#pragma unroll
				for (int j0 = 0; j0 < WB; j0 += 4) {
					if (j0 < n4) {   // (uniform)
						float power[4], opac[4];
						uint32_t idx1[4];
						bool pre[4];
#pragma unroll
						for (int u = 0; u < 4; u++) {
							const StagedEntryW se = s_e[g0 + j0 + u];
							const float dx = se.x - pxf, dy = se.y - pyf;
							power[u] = __builtin_fmaf(se.b2 * dx, dy, __builtin_fmaf(se.c2 * dy, dy, (se.a2 * dx) * dx));
							opac[u] = se.o;
							idx1[u] = se.idx1;
							pre[u] = !(power[u] > 0.0f) && !(power[u] < se.thr);
						}
#pragma unroll
						for (int u = 0; u < 4; u++) {
							const bool cand0 = !done && pre[u];
							if (__ballot(cand0) != 0ull) {
								const float alpha = fmin_(0.99f, opac[u] * expf_contract(power[u]));
								const float test_T = T * (1.0f - alpha);
								const bool cand = cand0 && !(alpha < 1.0f / 255.0f);
								const bool stop = cand && (test_T < 0.0001f);
								const bool take = cand && !stop;
								done = done || stop;
								if (take) {
									wv[j0 + u] = alpha * T;
									T = test_T;
									last = idx1[u];
								}
								if (__ballot(take) != 0ull) act |= 1u << (j0 + u);
							}
						}
					}
				}
				if (lane == 0 && act != 0u) atomicOr(&s_amask[gi], act);
			}
			lds_barrier();   // B3: the group's mask is complete
			const uint32_t amask = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_amask[gi]);
			const uint32_t cnt = (uint32_t)__popc(amask);
			if (nchunks * ACH < total + cnt) {   // (tile-uniform) the group crosses into a new chunk: about once per tile
				if (t == 0) reserve(nchunks, total + cnt);
				__syncthreads();
				nchunks = (total + cnt + ACH - 1) / ACH;
			}
			if (s_ovf == 0u) {
				uint32_t m = amask;
				for (uint32_t r = 0; r < cnt; r++) {
					const int en = __builtin_ctz(m);   // (uniform: the register array is indexed through the scalar unit)
					m &= m - 1;
					const uint32_t g = total + r;
					emit(g, wv[en]);
					if (t == 0) {
						const uint32_t slot = chunk_start_of(g / ACH) + (g % ACH);
						act_id[slot] = s_e[g0 + en].id;
						if (act_idx) act_idx[slot] = s_e[g0 + en].idx1;
					}
				}
			}
			total += cnt;
		}
	}
	// the closing T * bg pseudo entry (every tile gets one, also an empty tile)
	__syncthreads();
	if (t == 0 && nchunks * ACH < total + 1u) reserve(nchunks, total + 1u);
	__syncthreads();
	nchunks = (total + 1u + ACH - 1) / ACH;
	if (s_ovf == 0u) {
		emit(total, inside ? T : 0.0f);
		if (t == 0) act_id[chunk_start_of(total / ACH) + (total % ACH)] = SGS_BG_ID;
	}
	total += 1u;
	if (s_ovf == 0u) {   // zero-pad the last batch to 16 entries (all inside the tile's last chunk)
		const uint32_t pad_end = (total + 15u) & ~15u;
		for (uint32_t g = total; g < pad_end; g++) emit(g, 0.0f);
	}
	if (t == 0) nact[tile] = total;
	if (inside) {
		const size_t pix = (size_t)py * W + px;
		final_T[pix] = T;
		n_contrib[pix] = last;
	}
}

constexpr int AB = 16;   // work-list entries per batch (divides ACH)


// -------------------------------------------------------------------------------------
// Row-sweep accumulate (blend_accum_sweep_kernel).
//
// With the matrix work on the MFMA pipe the output store decides the kernel: the memory system writes
// the 64-B pieces a 16-px-wide tile owns at 3.5 TB/s but complete aligned 128-B lines at 4.8-5.2 TB/s,
// and L2 does not merge halves written by different waves (tools/ubench_store.hip,
// profiles/r01_ubench_store.txt).  A line of channel c, row y covers 32 pixels = two horizontally
// adjacent tiles; with a pitch of W*4 bytes and W % 32 == 16 the lines start at x = 0 (mod 32) on even
// rows and x = 16 (mod 32) on odd rows (`stagger`; W % 32 == 0: every row alike):
//     even rows: line = tiles (2k, 2k+1)      odd rows: line = tiles (2k-1, 2k)
// So a workgroup SWEEPS: it walks `seg` consecutive tiles of a tile row for 128 channels and ONE row
// parity, all their batches as one pipeline, accumulates the current tile in one set of accumulators
// (S[1]), keeps a finished LEFT half in a second set (S[0]), and when the right-hand tile is finished
// v_permlane16_swap_b32 merges the two 16-lane half rows so that every store instruction writes two
// complete 128-B lines.  Only the first / last tile of a segment writes half lines.
//
// The sweep also amortises the per-workgroup prologue (the segment's batches flattened into one LDS
// table) and never drains the DMA pipeline between tiles: the next tiles' batches are in flight while a
// finished tile is stored.  The closing T * bg term is a work-list entry (blend_weights_kernel), so
// there is no epilogue arithmetic.
//
// Workgroup = 4 waves = 4 channel groups of 32; wave tile 32 channels x the 128 pixels of the parity
// (4 MFMA blocks of 32 px = two rows each); batch = 16 entries; ring of NST stages of 8 KB fp32
// features + 8 KB weights (this parity's half) + the ids of the batch LA bundles on.
// ---- sweep plan: the order in which the accumulate sweep's workgroups take the segments.
// A sweep workgroup's duration is ~ a * (batches of its segment) + b (tools/sweep_trace.py: 1.46 us per batch + 36 us
// at cfg3) and the hardware hands workgroup b to XCD b % 8 in blockIdx order, 64 at a time per XCD.  In row-major
// order the last workgroups of every XCD are full-size ones, and an XCD whose band of rows is lighter idles while the
// others finish: 433 of 512 slots busy on average.  This kernel (one workgroup, a few us between the weights pre-pass
// and the sweep) sorts the segments by their batch count, heaviest first; the sweep deals them to the XCDs in
// serpentine order (rank k -> XCD k % 16 < 8 ? k % 8 : 7 - k % 8), all 2 * C / 128 workgroups of a segment on one XCD
// (they share its weights and feature rows through that XCD's L2).  Longest-processing-time-first: every XCD gets
// the same work to within one small segment and its last workgroups are its shortest.
constexpr int PLAN_MAX = 4096;   // segments (beyond: row-major order)
__global__ __launch_bounds__(1024) void sweep_plan_kernel(const uint32_t* __restrict__ nact, uint32_t* __restrict__ order,
							   const uint32_t* __restrict__ counter, int gx, int gy, int seg, int nseg,
							   volatile uint32_t* usage_host, uint32_t* __restrict__ tile_order)
{
	if (usage_host && threadIdx.x == 0) {   // the work-list usage feedback, straight into the stream's pinned words
		usage_host[0] = counter[0];         // (a device-to-host copy at the end of the frame would be one more launch)
		usage_host[1] = counter[1];
		__threadfence_system();
	}
	if (counter[1] != 0u) return;
	__shared__ uint32_t key[PLAN_MAX];
	if (tile_order) {   // the NEXT frame's tile order for the weights pre-pass: a counting sort by work-list length, longest first
		uint32_t* hist = key;   // (1024 bins; the segment keys below reuse the array afterwards)
		const int ntiles = gx * gy, tid = (int)threadIdx.x;
		hist[tid] = 0u;
		__syncthreads();
		for (int t = tid; t < ntiles; t += 1024) atomicAdd(&hist[1023u - (nact[t] < 1023u ? nact[t] : 1023u)], 1u);
		__syncthreads();
		const uint32_t mine = hist[tid];
		for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan over the bins
			const uint32_t add = tid >= off ? hist[tid - off] : 0u;
			__syncthreads();
			hist[tid] += add;
			__syncthreads();
		}
		const uint32_t first = hist[tid] - mine;
		__syncthreads();
		hist[tid] = first;   // from here on: the next free position of each bin
		__syncthreads();
		for (int t = tid; t < ntiles; t += 1024)
			tile_order[1u + atomicAdd(&hist[1023u - (nact[t] < 1023u ? nact[t] : 1023u)], 1u)] = (uint32_t)t;
		if (tid == 0) tile_order[0] = (uint32_t)ntiles;
		__syncthreads();
	}
	const int n = gy * nseg;
	int np2 = 1;
	while (np2 < n) np2 <<= 1;
	for (int s = threadIdx.x; s < np2; s += 1024) {
		uint32_t k = 0;   // padding sorts last
		if (s < n) {
			const int ty = s / nseg, tx0 = (s % nseg) * seg;
			const int nt = (gx - tx0) < seg ? (gx - tx0) : seg;
			uint32_t J = 0;
			for (int i = 0; i < nt; i++) J += (nact[ty * gx + tx0 + i] + 15u) >> 4;
			J = J < 0xFFFFEu ? J : 0xFFFFEu;
			k = ((J + 1u) << 12) | (uint32_t)(PLAN_MAX - 1 - s);   // ties: lower segment first
		}
		key[s] = k;
	}
	__syncthreads();
	if (np2 <= 1024) {   // few segments: rank by counting (keys are distinct), one pass over LDS
		const int s = threadIdx.x;
		if (s < n) {
			const uint32_t mine = key[s];
			int rank = 0;
			for (int j = 0; j < n; j++) rank += key[j] > mine ? 1 : 0;
			order[rank] = (uint32_t)(PLAN_MAX - 1) - (mine & 4095u);
		}
		return;
	}
	for (int k2 = 2; k2 <= np2; k2 <<= 1)   // bitonic sort, descending
		for (int j = k2 >> 1; j > 0; j >>= 1) {
			for (int i = threadIdx.x; i < np2; i += 1024) {
				const int ixj = i ^ j;
				if (ixj > i) {
					const uint32_t a = key[i], b = key[ixj];
					const bool desc = (i & k2) == 0;
					if (desc ? a < b : a > b) {
						key[i] = b;
						key[ixj] = a;
					}
				}
			}
			__syncthreads();
		}
	for (int s = threadIdx.x; s < n; s += 1024) order[s] = (uint32_t)(PLAN_MAX - 1) - (key[s] & 4095u);
}

#ifndef SGS_SEGMAX   // (overridable for A/B builds: LDS per workgroup decides what can share a CU with the sweep)
#define SGS_SEGMAX 96
#define SGS_SW_JMAX 1024
#define SGS_NST 4
#endif
constexpr int SEGMAX = SGS_SEGMAX;   // tiles per sweep (upper bound, the launcher picks the length)
constexpr int SW_JMAX = SGS_SW_JMAX; // batch-table window (batches of a segment kept in LDS)
constexpr int NST = SGS_NST;         // ring stages (bundles of NST - 1 batches in flight)
constexpr int LA = NST - 1;
constexpr int STAGE_BYTES = 8192 + 8192 + 1024;   // features | this parity's weights | ids of the batch LA bundles on
constexpr int SW_NDMA = 5;   // LDS-DMA instructions per wave per bundle: 2 feature + 2 weight + 1 id

// The accumulator sets are touched only through these free functions with compile-time set
// indices (closures nested more than one level deep keep the array in scratch memory).
typedef f32x16 SweepSets[2][4];   // [left, right][block]
typedef int v4i __attribute__((ext_vector_type(4)));

// Block pb of a wave's row-parity group g holds rows rp = 2 pb (lanes 0-15 / 32-47) and rp + 1
// (lanes 16-31 / 48-63), y = 2 rp + g; lanes >= 32 are channel + 4.
// left (S[0]) | right (S[1]) half rows -> complete lines.
// base = &out[c0 + 4*half][ty*16 + g][xl0 + (lane & 31)];  GUARD = some lane or row is outside
// the image (edge tiles only): per-store predication instead of straight-line stores.
template <bool GUARD>
__device__ __forceinline__ void sweep_store_paired(const SweepSets& S, float* base, size_t HW, int W,
						   bool xok, int y0, int H)
{
#pragma unroll
	for (int pb = 0; pb < 4; pb++) {
		const int y = y0 + 4 * pb;
		const bool ok0 = xok && y < H, ok1 = xok && y + 2 < H;
		float* bp = base + (size_t)(4 * pb) * W;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(S[0][pb][r]),
									 __float_as_uint(S[1][pb][r]), false, false);
			float* dst = bp + (size_t)((r & 3) + 8 * (r >> 2)) * HW;
			if (!GUARD || ok0) *dst = __uint_as_float(sw[0]);
			if (!GUARD || ok1) dst[2 * (size_t)W] = __uint_as_float(sw[1]);
		}
	}
}

// The interior case (every lane and row inside the image) without per-store address arithmetic:
// `global_store_dword voffset, data, s[base]` with the channel plane as a wave-uniform SGPR base
// walked by scalar adds (planes (r & 3) + 8 (r >> 2): +1, +1, +1, +5) and one 32-bit byte offset
// per lane and pixel block.  Inline asm keeps the compiler from tabulating the 16 bases (it runs
// out of SGPRs and falls back to 64-bit VGPR addresses: ~2 VALU per store, 256 per tile and wave).
// `nt`: the image is written once and not read back by this pipeline; streaming stores were 6-10 %
// faster in tools/ubench_store.hip (modes 16 vs 18).
// ubase = &out[c0][0][0] (uniform); loff = byte offset of [4*half][ty*16 + g][xl0 + (lane & 31)]
__device__ __forceinline__ void sweep_store_paired_fast(const SweepSets& S, const float* ubase, uint32_t loff,
							size_t HW, int W)
{
	const uint64_t plane = (uint64_t)HW * 4u;
#pragma unroll
	for (int pb = 0; pb < 4; pb++) {
		const uint32_t o0 = loff + (uint32_t)(4 * pb * W) * 4u, o1 = o0 + (uint32_t)(2 * W) * 4u;
		uint64_t sb = (uint64_t)ubase;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(S[0][pb][r]),
									 __float_as_uint(S[1][pb][r]), false, false);
			asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(o0), "v"(sw[0]), "s"(sb) : "memory");
			asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(o1), "v"(sw[1]), "s"(sb) : "memory");
			sb += ((r & 3) == 3 ? 5u : 1u) * plane;
		}
	}
}

// EXPERIMENT (variant 0x408), not the default: the same pair as 16-byte stores.  Measured at cfg3: sweep 1.09 ms against
// 0.93 ms for the dword form -- eight channel planes per store instruction and 512 extra VALU per pair cost more than
// the shorter store queue gains.  After the permlane16 swap lane l31 of register r holds pixel xp + l31 of channel
// plane(r); a 4 x 4 transpose inside every quad of lanes (two rounds of v_mov_dpp quad_perm + select) turns four
// consecutive registers (four channels) x four lanes (four pixels) into: lane qi of the quad holds four CONSECUTIVE
// pixels of channel plane(4 q + qi).  One global_store_dwordx4 then writes eight complete 128-B lines (8 lanes x 16 B
// per channel and row), and a tile pair is 32 store instructions instead of 128.
// loff4 = byte offset of [4*half + (lane & 3)][ty*16 + g][xl0 + 4 * ((lane & 31) >> 2)]
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sweep_store_paired_wide(const SweepSets& S, const float* ubase, uint32_t loff4,
							 size_t HW, int W, int lane)
{
	const uint64_t plane8 = (uint64_t)HW * 32u;   // 8 channel planes
	// lane masks for the selects.  They are bit merges (v_bfi_b32) on purpose: written as ?: the compiler turns them
	// into exec-masked branches, and a DPP move executed under a partial exec mask cannot read the disabled lanes.
	const uint32_t m1 = (lane & 1) ? 0xFFFFFFFFu : 0u, m2 = (lane & 2) ? 0xFFFFFFFFu : 0u;
	auto sel = [](uint32_t m, uint32_t a, uint32_t b) __attribute__((always_inline)) { return (a & m) | (b & ~m); };
#pragma unroll
	for (int pb = 0; pb < 4; pb++) {
		const uint32_t o0 = loff4 + (uint32_t)(4 * pb * W) * 4u, o1 = o0 + (uint32_t)(2 * W) * 4u;
		uint64_t sb = (uint64_t)ubase;
#pragma unroll
		for (int q = 0; q < 4; q++) {
			uint32_t a[4], b[4];   // row 2 pb (a) and row 2 pb + 1 (b) of registers 4 q .. 4 q + 3
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(S[0][pb][4 * q + j]),
										 __float_as_uint(S[1][pb][4 * q + j]), false, false);
				a[j] = sw[0];
				b[j] = sw[1];
			}
#pragma unroll
			for (int w = 0; w < 2; w++) {
				uint32_t* v = w ? b : a;
				// 2 x 2 blocks inside lane pairs, then the off-diagonal blocks across lane pairs
				const uint32_t x0 = sel(m1, quad_xor1(v[1]), v[0]), x1 = sel(m1, v[1], quad_xor1(v[0]));
				const uint32_t x2 = sel(m1, quad_xor1(v[3]), v[2]), x3 = sel(m1, v[3], quad_xor1(v[2]));
				const uint32_t y0 = sel(m2, quad_xor2(x2), x0), y2 = sel(m2, x2, quad_xor2(x0));
				const uint32_t y1 = sel(m2, quad_xor2(x3), x1), y3 = sel(m2, x3, quad_xor2(x1));
				const u32x4 d = {y0, y1, y2, y3};
				const uint32_t off = w ? o1 : o0;
				// (s_nop: a store of more than 64 bits is still reading its data registers when the next VALU may
				// overwrite them -- the hazard the compiler pads for its own stores but cannot see inside an asm)
				asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(sb) : "memory");
			}
			sb += plane8;
		}
	}
}

// the half rows in S[1] on their own (segment ends): 64-B pieces;
// base = &out[c0 + 4*half][ty*16 + g + 2*((lane>>4)&1)][x0 + (lane & 15)]
template <bool GUARD>
__device__ __forceinline__ void sweep_store_single(const SweepSets& S, float* base, size_t HW, int W,
						   bool xok, int y0, int H)
{
#pragma unroll
	for (int pb = 0; pb < 4; pb++) {
		const bool ok = xok && y0 + 4 * pb < H;
		float* bp = base + (size_t)(4 * pb) * W;
#pragma unroll
		for (int r = 0; r < 16; r++)
			if (!GUARD || ok) bp[(size_t)((r & 3) + 8 * (r >> 2)) * HW] = S[1][pb][r];
	}
}

template <int X>
__device__ __forceinline__ void sweep_zero(SweepSets& S)
{
#pragma unroll
	for (int pb = 0; pb < 4; pb++)
#pragma unroll
		for (int r = 0; r < 16; r++) S[X][pb][r] = 0.f;
}

// One batch of 16 entries out of ring stage `st` (LDS byte address) into S[1], for the wave's
// 32 channels (cg) and its row parity g (4 pixel blocks).  All LDS reads of the ring are inline
// asm: the compiler's waitcnt pass would put s_waitcnt vmcnt(0) in front of every ds_read that
// may alias an LDS-DMA destination and drain the bundles in flight (cdna_hip_programming.md
// 5.7: early-clobber outputs, nothing consumes an output before the explicit lgkmcnt(0), that
// wait takes the values as "+v").
template <bool X16>
__device__ __forceinline__ void sweep_compute(SweepSets& S, uint32_t st, uint32_t n, int cg, int half, int l31)
{
	const uint32_t fa = st + (uint32_t)((8 * half) * 128 + cg * 32 + l31) * 4u;   // + kk * 512
	const uint32_t wa = st + 8192u + (uint32_t)half * 4096u + (uint32_t)l31 * 16u;  // + pb * 512 (+ 2048 lo)
	float f[8];
	v4i bh, bl;
	asm volatile(
		"ds_read_b32 %0, %10\n\t"
		"ds_read_b32 %1, %10 offset:512\n\t"
		"ds_read_b32 %2, %10 offset:1024\n\t"
		"ds_read_b32 %3, %10 offset:1536\n\t"
		"ds_read_b32 %4, %10 offset:2048\n\t"
		"ds_read_b32 %5, %10 offset:2560\n\t"
		"ds_read_b32 %6, %10 offset:3072\n\t"
		"ds_read_b32 %7, %10 offset:3584\n\t"
		"ds_read_b128 %8, %11\n\t"
		"ds_read_b128 %9, %11 offset:2048\n\t"
		"s_waitcnt lgkmcnt(0)"
		: "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]),
		  "=&v"(f[7]), "=&v"(bh), "=&v"(bl)
		: "v"(fa), "v"(wa)
		: "memory");
	__builtin_amdgcn_sched_barrier(0);
	// (rows past n hold a clamped duplicate of the last entry; their weights are zero)
	(void)n;
	bf16x8 ah, al;
#pragma unroll
	for (int kk = 0; kk < 8; kk++) {
		ah[kk] = (__bf16)f[kk];
		al[kk] = (__bf16)(f[kk] - (float)ah[kk]);
	}
	const s16x8 ahs = __builtin_bit_cast(s16x8, ah), als = __builtin_bit_cast(s16x8, al);
	const s16x4 ah0 = {ahs[0], ahs[1], ahs[2], ahs[3]}, ah1 = {ahs[4], ahs[5], ahs[6], ahs[7]};
	const s16x4 al0 = {als[0], als[1], als[2], als[3]}, al1 = {als[4], als[5], als[6], als[7]};
#pragma unroll
	for (int pb = 0; pb < 4; pb++) {   // the next block's operands fly while this block multiplies
		v4i nh, nl;
		if (pb < 3) {
			const uint32_t wn = wa + (uint32_t)(pb + 1) * 512u;
			asm volatile(
				"ds_read_b128 %0, %2\n\t"
				"ds_read_b128 %1, %2 offset:2048"
				: "=&v"(nh), "=&v"(nl)
				: "v"(wn)
				: "memory");
		}
		// Each of the three products is issued as two v_mfma_f32_32x32x8_bf16_1k over the two halves of the
		// lane's 8 k-values (lane half h owns k = 8 h + j: the pairing of A and B stays consistent), NOT as one
		// gfx950 v_mfma_f32_32x32x16_bf16.  With the double-rate instruction in this kernel, waves of OTHER
		// kernels resident on the same CU (a second view's preprocess / binning / weights on another stream)
		// sporadically received a wrong 256-byte beat of a global_load_dwordx4 return: ~1 forward in 1000
		// with two views in flight came out with a handful of wrong radii (24 000-view stress runs: 16-26
		// corrupted forwards with the x16 form, 0 with this form, 0 with the fp32-MFMA sweep, 0 with the MFMAs
		// replaced by VALU work; stores and LDS prefetch pattern made no difference).  The
		// halved matrix rate costs ~0.09 ms of the sweep at the headline size.
		const s16x8 h = __builtin_bit_cast(s16x8, bh), l = __builtin_bit_cast(s16x8, bl);
		const s16x4 h0 = {h[0], h[1], h[2], h[3]}, h1 = {h[4], h[5], h[6], h[7]};
		const s16x4 l0 = {l[0], l[1], l[2], l[3]}, l1 = {l[4], l[5], l[6], l[7]};
		if (X16) {   // reproducer only (tools/repro_x16_neighbour_corruption.py): the gfx950 double-rate form
			const bf16x8 hb = __builtin_bit_cast(bf16x8, bh), lb = __builtin_bit_cast(bf16x8, bl);
			S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, hb, S[1][pb], 0, 0, 0);
			S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, lb, S[1][pb], 0, 0, 0);
			S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, hb, S[1][pb], 0, 0, 0);
		} else {
		S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(al0, h0, S[1][pb], 0, 0, 0);
		S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(al1, h1, S[1][pb], 0, 0, 0);
		S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah0, l0, S[1][pb], 0, 0, 0);
		S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah1, l1, S[1][pb], 0, 0, 0);
		S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah0, h0, S[1][pb], 0, 0, 0);
		S[1][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah1, h1, S[1][pb], 0, 0, 0);
		}
		if (pb < 3) {
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nh), "+v"(nl) : : "memory");
			__builtin_amdgcn_sched_barrier(0);
			bh = nh;
			bl = nl;
		}
	}
}

// The same batch with fp32-input MFMAs (SGS_BLEND_EXACT): weights are fp32 rows [entry][128 px], one
// v_mfma_f32_32x32x2_f32 per pair of entries and pixel block -- an exact k-ordered fma chain, bit-identical
// to the contract (the closing T * bg entry is the contract's final fma(T, bg, acc)).
__device__ __forceinline__ void sweep_compute_exact(SweepSets& S, uint32_t st, int cg, int half, int l31)
{
	const uint32_t fa = st + (uint32_t)(half * 128 + cg * 32 + l31) * 4u;      // + pair * 1024
	const uint32_t wa = st + 8192u + (uint32_t)(half * 128 + l31) * 4u;       // + pair * 1024 + pb * 128
	float a, b0, b1, b2, b3;
	asm volatile(
		"ds_read_b32 %0, %5\n\t"
		"ds_read_b32 %1, %6\n\t"
		"ds_read_b32 %2, %6 offset:128\n\t"
		"ds_read_b32 %3, %6 offset:256\n\t"
		"ds_read_b32 %4, %6 offset:384\n\t"
		"s_waitcnt lgkmcnt(0)"
		: "=&v"(a), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
		: "v"(fa), "v"(wa)
		: "memory");
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int p = 0; p < 8; p++) {
		float an = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
		if (p < 7) {
			const uint32_t fn = fa + (uint32_t)(p + 1) * 1024u, wn = wa + (uint32_t)(p + 1) * 1024u;
			asm volatile(
				"ds_read_b32 %0, %5\n\t"
				"ds_read_b32 %1, %6\n\t"
				"ds_read_b32 %2, %6 offset:128\n\t"
				"ds_read_b32 %3, %6 offset:256\n\t"
				"ds_read_b32 %4, %6 offset:384"
				: "=&v"(an), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
				: "v"(fn), "v"(wn)
				: "memory");
		}
		S[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, S[1][0], 0, 0, 0);
		S[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, S[1][1], 0, 0, 0);
		S[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, S[1][2], 0, 0, 0);
		S[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, S[1][3], 0, 0, 0);
		if (p < 7) {
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an), "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3) : : "memory");
			__builtin_amdgcn_sched_barrier(0);
			a = an; b0 = n0; b1 = n1; b2 = n2; b3 = n3;
		}
	}
}

// One workgroup = (tile-row segment, 128 channels, row parity g): 4 waves = 4 channel groups of 32.
// The two parities are separate workgroups (two per CU) so that one's store phase overlaps the
// other's multiply phase; each fetches the features (the second copy comes from L2) and its own
// half of the weights.
// DBG (development ablations, 0 in production): 1 = no stores, 2 = no matrix work.
template <int DBG, bool EXACT>
__global__ __launch_bounds__(256, 2) void blend_accum_sweep_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const char* __restrict__ wgt, const float* __restrict__ features,
	const float* __restrict__ bg, float* __restrict__ out, const uint32_t* __restrict__ counter,
	int W, int H, int C, int gx, int nchunks_c, int seg, int nseg, int per_xcd, int total_items, int PW,
	unsigned long long* __restrict__ trace, const uint32_t* __restrict__ order, int dealt)
{
	if (counter[1] != 0u) return;   // arena overflowed: the single-kernel path renders this frame
	const int b = blockIdx.x;
	int chunk, g, rest;   // 128-channel chunk, row parity, segment (= ty * nseg + sg) of this workgroup
	if (dealt) {   // segments dealt to the XCDs in serpentine order of their rank (sweep_plan_kernel); per_xcd = ranks per XCD
		const int x = b & 7, pos = b >> 3, sib = 2 * nchunks_c;
		const int m = pos / sib, w = pos - m * sib;
		const int k = 16 * (m >> 1) + ((m & 1) ? 15 - x : x);
		if (k >= total_items) return;   // (total_items = segments)
		chunk = w % nchunks_c;
		g = w / nchunks_c;
		rest = order ? (int)order[k] : k;
	} else {   // row-major: XCD x takes the band [x * per_xcd, (x + 1) * per_xcd) of (segment, parity, chunk) items
		const int v = (b & 7) * per_xcd + (b >> 3);
		if (v >= total_items) return;
		chunk = v % nchunks_c;
		g = (v / nchunks_c) & 1;
		rest = v / (2 * nchunks_c);
	}
	// (debug, tools/sweep_trace.py) per-workgroup timeline: begin / end on the 100 MHz steady counter + where it ran
	const unsigned long long t_begin = trace ? wall_clock64() : 0ull;
	// PW = output row pitch in pixels (>= W; the image width itself unless the caller asked for padded rows)
	const int stagger = (PW & 31) == 16 ? 1 : 0;   // odd rows start 64 B into a line (else every row is aligned alike)
	const int sg = rest % nseg, ty = rest / nseg;
	const int tx0 = sg * seg;   // even (seg is even)
	const int nt = (gx - tx0) < seg ? (gx - tx0) : seg;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int cg = wave;
	const int half = lane >> 5, l31 = lane & 31;
	const int cbase = chunk * 128;
	const int c0 = cbase + cg * 32;
	const size_t HW = (size_t)H * PW;   // channel plane stride

	__shared__ float4 s_ring[NST * STAGE_BYTES / 16];
	// the segment's batches as one flat stream: .x = first arena slot of the batch,
	// .y = entries (1..16) | tile index in the segment << 8 | last batch of its tile << 16
	__shared__ uint2 s_bt[SW_JMAX];
	__shared__ uint32_t s_tot[SEGMAX], s_cb[SEGMAX], s_pref[SEGMAX + 1];

	// ---- prologue: the batch table (ordinary LDS / global accesses: nothing is in flight yet)
	if ((int)threadIdx.x < nt) {
		const int tile = ty * gx + tx0 + threadIdx.x;
		s_tot[threadIdx.x] = nact[tile];   // >= 1: every tile ends with the T * bg pseudo entry
		s_cb[threadIdx.x] = (ranges[tile].x >> 7) + (uint32_t)tile;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t acc = 0;
		for (int t = 0; t < nt; t++) {
			s_pref[t] = acc;
			acc += (s_tot[t] + AB - 1) / AB;
		}
		s_pref[nt] = acc;
	}
	__syncthreads();
	const uint32_t J = s_pref[nt];   // batches in this segment
	// (re)build the table window [wbase, wbase + SW_JMAX); entries past J repeat the last batch
	auto fill_table = [&](uint32_t wbase) __attribute__((always_inline)) {
		uint32_t maxnb = 0;
		for (int t = 0; t < nt; t++) maxnb = max(maxnb, s_pref[t + 1] - s_pref[t]);
		for (int tb = 0; tb < nt; tb += 8)   // 8 tiles x 32 batches per pass
			for (uint32_t qb = 0; qb < maxnb; qb += 32) {
				const int t = tb + (int)(threadIdx.x >> 5);
				const uint32_t q = qb + (threadIdx.x & 31);
				if (t < nt) {
					const uint32_t p0 = s_pref[t], nb = s_pref[t + 1] - p0;
					if (q < nb && p0 + q >= wbase && p0 + q < wbase + SW_JMAX) {
						const uint32_t tot = s_tot[t], first = q * AB;
						const uint32_t slot = sgs_chunk_start(table, s_cb[t], (uint32_t)(ty * gx + tx0 + t), first >> 7) + (first & 127u);
						const uint32_t n = (tot - first) < (uint32_t)AB ? (tot - first) : (uint32_t)AB;
						s_bt[p0 + q - wbase] = make_uint2(slot, n | ((uint32_t)t << 8) | (q + 1 == nb ? 1u << 16 : 0u));
					}
				}
			}
		// past the end: 2 LA dummy batches (re-reads of a valid batch, never consumed)
		if (threadIdx.x < 2 * LA && J + threadIdx.x >= wbase && J + threadIdx.x - wbase < SW_JMAX)
			s_bt[J + threadIdx.x - wbase] = make_uint2((uint32_t)(ty * gx + tx0 + nt - 1) * 128u, 1u | ((uint32_t)(nt - 1) << 8));
	};
	fill_table(0);
	__syncthreads();

	const uint32_t ring = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_ring;
	const uint32_t bt_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_bt;
	const uint32_t sub = (uint32_t)(4 * wave + half);   // this lane fetches the feature rows of entries sub and sub + 2
	// bundle = features + this parity's weights of the batch at `slot` into stage st, and the ids of
	// the batch (slot2, n2) into its id area.  Every wave issues exactly SW_NDMA DMA instructions.
	auto issue = [&](uint32_t slot, uint32_t id0, uint32_t id1, uint32_t slot2, uint32_t n2, uint32_t st) __attribute__((always_inline)) {
		const float* row0 = id0 == SGS_BG_ID ? bg : features + (size_t)id0 * C;
		const float* row1 = id1 == SGS_BG_ID ? bg : features + (size_t)id1 * C;
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row0 + cbase + l31 * 4),
						 (__attribute__((address_space(3))) void*)(size_t)(st + (uint32_t)(4 * wave) * 512u),
						 16, 0, 0);
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row1 + cbase + l31 * 4),
						 (__attribute__((address_space(3))) void*)(size_t)(st + (uint32_t)(4 * wave + 2) * 512u),
						 16, 0, 0);
		// weights.  bf16: piece `wave` = (k-group wave >> 1, hi/lo wave & 1), this parity's 2 KB half;
		// exact: fp32 rows of 1 KB per entry, this parity's 512 B of entries 4*wave .. 4*wave+3
		const char* wsrc = EXACT ? wgt + (size_t)(slot + 4 * wave + (lane >> 5)) * 1024 + (size_t)g * 512 +
						   (size_t)(lane & 31) * 16
					 : wgt + (size_t)((slot >> 3) + (wave >> 1)) * 8192 + (size_t)(wave & 1) * 4096 +
						   (size_t)g * 2048 + (size_t)lane * 16;
#pragma unroll
		for (int j = 0; j < 2; j++)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + j * (EXACT ? 2048 : 1024)),
							 (__attribute__((address_space(3))) void*)(size_t)(st + 8192u + (uint32_t)(wave * 2 + j) * 1024u),
							 16, 0, 0);
		const uint32_t li = (uint32_t)(lane & 15) < n2 ? (uint32_t)(lane & 15) : n2 - 1u;
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(act_id + slot2 + li),
						 (__attribute__((address_space(3))) void*)(size_t)(st + 16384u + (uint32_t)wave * 256u),
						 4, 0, 0);
	};

	SweepSets S;   // this wave's left / right half rows
	sweep_zero<0>(S);
	sweep_zero<1>(S);

	// ---- the sweep.  Step j computes batch j, issues bundle j + LA, whose id area carries the ids
	// of batch j + 2 LA.
	uint32_t wbase = 0;
#pragma unroll
	for (int k = 0; k < LA; k++) {   // prologue bundles 0 .. LA-1 (ids by ordinary loads)
		const uint2 e = s_bt[k], e2 = s_bt[k + LA];
		const uint32_t n = e.y & 255u;
		const uint32_t id0 = act_id[e.x + (sub < n ? sub : n - 1u)];
		const uint32_t id1 = act_id[e.x + (sub + 2u < n ? sub + 2u : n - 1u)];
		issue(e.x, id0, id1, e2.x, e2.y & 255u, ring + (uint32_t)k * STAGE_BYTES);
	}
	uint32_t st0 = ring, stI = ring + LA * STAGE_BYTES;   // stages of batch j and of bundle j + LA
	bool has_pending = false;   // S[0] holds left half rows waiting for their right-hand tile
	int skip_wait = 0;   // steps whose bundle is already known to have landed (see the tile-complete branch)
	for (uint32_t j = 0; j < J; j++) {
		if (j + 2 * LA >= wbase + SW_JMAX) {   // (uniform, long segments only) slide the table window
			__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
			__syncthreads();
			wbase = j;
			fill_table(wbase);
			__syncthreads();
		}
		// this step's three table entries: batch j, j + LA, j + 2 LA
		uint32_t e0y, eIx, eDx, eDy;
		{
			const uint32_t a = bt_a + (j - wbase) * 8u;
			uint32_t r0, r1;
			uint64_t rd;
			asm volatile(
				"ds_read_b32 %0, %3 offset:4\n\t"
				"ds_read_b32 %1, %3 offset:%4\n\t"
				"ds_read_b64 %2, %3 offset:%5\n\t"
				"s_waitcnt lgkmcnt(0)"
				: "=&v"(r0), "=&v"(r1), "=&v"(rd)
				: "v"(a), "n"(LA * 8), "n"(2 * LA * 8)
				: "memory");
			e0y = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0);
			eIx = (uint32_t)__builtin_amdgcn_readfirstlane((int)r1);
			eDx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rd);
			eDy = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rd >> 32));
		}
		// bundle j landed; the LA - 1 younger bundles -- and after an epilogue part of its stores --
		// may still be in flight (vmcnt is 6 bits: <= 63)
		// vmcnt counts loads and stores in one counter but only orders them within their kind: a counted wait is safe
		// only up to the number of YOUNGER LOADS, so stores behind a bundle always have to retire before the wait for it
		// returns.  The tile-complete branch therefore waits for the next LA - 1 bundles BEFORE it issues its stores;
		// the steps that consume those bundles skip the wait and the stores get that long to drain.
		if (skip_wait > 0) skip_wait--;
		else __builtin_amdgcn_s_waitcnt(((LA - 1) * SW_NDMA) | (7 << 4) | (15 << 8));
		__builtin_amdgcn_s_barrier();
		{
			uint32_t id0, id1;   // this lane's feature-row ids for bundle j + LA, from the id area of bundle j
			const uint32_t ia = st0 + 16384u + (uint32_t)wave * 256u + sub * 4u;
			asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
				     : "=&v"(id0), "=&v"(id1) : "v"(ia) : "memory");
			__builtin_amdgcn_sched_barrier(0);
			issue(eIx, id0, id1, eDx, eDy & 255u, stI);   // into the stage batch j - 1 was computed from
		}
		const uint32_t n = e0y & 255u;
		const int tx = tx0 + (int)((e0y >> 8) & 255u);
		const bool is_left = ((tx + g * stagger) & 1) == 0;   // even rows: even tiles are left halves; odd rows (staggered): odd tiles
		if (!(DBG & 2)) {   // always into S[1]
			if (EXACT) sweep_compute_exact(S, st0, cg, half, l31);
			else sweep_compute<(DBG & 8) != 0>(S, st0, n, cg, half, l31);
		}
		if ((e0y >> 16) != 0u && !((DBG & 1) && S[1][0][0] != 123.f)) {   // tile complete
			const int hi = (l31 >> 4) & 1;
			const bool will_store = !is_left || tx == tx0 + nt - 1;   // (uniform)
			if (will_store) {   // bundles j + 1 .. j + LA - 1 first (issued one and two steps ago), then the stores
				__builtin_amdgcn_s_waitcnt(SW_NDMA | (7 << 4) | (15 << 8));
				skip_wait = LA - 1;
			}
			float* cbp = out + (size_t)(c0 + 4 * half) * HW + (size_t)(ty * SGS_TILE + g) * PW;
			const int xs = tx * SGS_TILE + (l31 & 15), xp = (tx - 1) * SGS_TILE + l31;
			const int y0 = ty * SGS_TILE + g;
			if (!is_left && has_pending) {   // S[0] | S[1] are whole lines
				// uniform: the whole 32 x 8 block lies inside the (row-padded) plane -- padding columns may be written
				const bool inside = (tx + 1) * SGS_TILE <= W && y0 + 14 < H;
				if (inside && (DBG & 4)) {   // (A/B: 16-byte stores after a quad transpose -- measured 17 % SLOWER, see the helper)
					sweep_store_paired_wide(S, out + (size_t)c0 * HW,
								((uint32_t)(4 * half + (lane & 3)) * (uint32_t)HW +
								 (uint32_t)(y0 * PW + (tx - 1) * SGS_TILE + 4 * (l31 >> 2))) * 4u, HW, PW, lane);
				} else if (inside) {
					sweep_store_paired_fast(S, out + (size_t)c0 * HW,
								((uint32_t)(4 * half) * (uint32_t)HW + (uint32_t)(y0 * PW + xp)) * 4u, HW, PW);
				} else {
					sweep_store_paired<true>(S, cbp + xp, HW, PW, xp < W, y0, H);
				}
			} else if (!is_left || tx == tx0 + nt - 1) {   // a half with no partner in this segment
				sweep_store_single<true>(S, cbp + (size_t)(2 * hi) * PW + xs, HW, PW, xs < W, y0 + 2 * hi, H);
			}
			if (is_left) {   // becomes the pending left half
#pragma unroll
				for (int pb = 0; pb < 4; pb++) S[0][pb] = S[1][pb];
			}
			has_pending = is_left;
			sweep_zero<1>(S);
		}
		st0 = st0 + STAGE_BYTES == ring + NST * STAGE_BYTES ? ring : st0 + STAGE_BYTES;
		stI = stI + STAGE_BYTES == ring + NST * STAGE_BYTES ? ring : stI + STAGE_BYTES;
	}
	__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // drain the dummy tail bundles before LDS is released
	if (trace && threadIdx.x == 0) {
		trace[4 * (size_t)b] = t_begin;
		trace[4 * (size_t)b + 1] = wall_clock64();
		trace[4 * (size_t)b + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |          // HW_ID
					   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // XCC_ID
		trace[4 * (size_t)b + 3] = (unsigned long long)J | ((unsigned long long)nt << 32);
	}
}

// counter[0] = work-list slots handed out, counter[1] = 0 ok / 1 the work list overflowed (the gated single-kernel
// fallback renders the frame) / 2 the frame was aborted before the blend (deferred-count forward whose capacity
// guess was too small: every blend kernel exits).  Replaces a memset so that the abort word is folded in.
__global__ void arena_reset_kernel(uint32_t* __restrict__ counter, const uint32_t* __restrict__ abort, uint32_t first_free)
{
	counter[0] = first_free;   // slots [0, ntiles * 128) are the tiles' pre-assigned first chunks
	counter[1] = (abort && *abort != 0u) ? 2u : 0u;
#pragma unroll
	for (int x = 0; x < 8; x++) counter[2 + x] = 0u;   // the per-XCD tile tickets of the backward's persistent kernel (blend_bwd_mfma.hip)
}

// The exact-format work list alone (fp32 weight rows + ids + list positions): the backward's first step.
hipError_t launch_blend_weights_rows(hipStream_t st, const uint2* ranges, const uint32_t* point_list,
				     const float2* means2D, const float4* conic_opacity, float* final_T,
				     uint32_t* n_contrib, char* arena, const SplitArena& lay, int W, int H, int gx,
				     int gy, float* clear_ptr, size_t clear_floats, const uint32_t* tile_order)
{
	const int ntiles = gx * gy;
	uint32_t* counter = (uint32_t*)(arena + lay.counter);
	hipLaunchKernelGGL(arena_reset_kernel, dim3(1), dim3(1), 0, st, counter, (const uint32_t*)nullptr,
			   (uint32_t)ntiles * (uint32_t)ACH);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	return launch_blend_weights2(st, 3, ranges, point_list, means2D, conic_opacity, final_T, n_contrib,
				     (uint32_t*)(arena + lay.act_id), (uint32_t*)(arena + lay.act_idx), (float*)(arena + lay.wgt),
				     (uint32_t*)(arena + lay.table), (uint32_t*)(arena + lay.nbatches), counter, lay.capacity, W, H, gx,
				     ntiles, clear_ptr, clear_floats, tile_order);
}

size_t split_arena_bytes(uint32_t capacity, size_t L, int ntiles, SplitArena* lay, int slot_bytes)
{
	size_t off = 0;
	auto take = [&](size_t bytes) { off = (off + 127) & ~(size_t)127; const size_t o = off; off += bytes; return o; };
	SplitArena a;
	a.capacity = capacity;
	a.counter = take(64);   // slots handed out, overflow word, 8 tile tickets (the backward)
	a.nbatches = take((size_t)ntiles * 4);                       // nact[tile]
	a.table = take(((L >> 7) + (size_t)ntiles + 1) * 4);           // chunk starts
	a.act_id = take((size_t)capacity * 4);
	a.act_idx = take((size_t)capacity * 4);
	a.order = take((size_t)PLAN_MAX * 4);
	a.wgt = take((size_t)capacity * (size_t)slot_bytes);   // 1 KB per slot (fp32 rows / two bf16 terms), 1.5 KB with three bf16 terms
	a.total = (off + 127) & ~(size_t)127;
	if (lay) *lay = a;
	return a.total;
}

static unsigned long long* g_sweep_trace = nullptr;   // debug only (sgs_debug_set_sweep_trace)
void set_sweep_trace(void* device_words) { g_sweep_trace = (unsigned long long*)device_words; }
unsigned long long* get_sweep_trace() { return g_sweep_trace; }

hipError_t launch_blend_forward_split(hipStream_t st, const BlendFwdArgs& a, char* arena,
				      const SplitArena& lay, void (*mark)(void*), void* mark_user,
				      int split_mode, bool* usage_reported)
{
	if (usage_reported) *usage_reported = false;
	const int ntiles = a.gx * a.gy;
	uint32_t* counter = (uint32_t*)(arena + lay.counter);
	uint32_t* nbatches = (uint32_t*)(arena + lay.nbatches);
	uint32_t* table = (uint32_t*)(arena + lay.table);
	uint32_t* act_id = (uint32_t*)(arena + lay.act_id);
	float* wgt = (float*)(arena + lay.wgt);
	if (!a.counter_reset_done)
		hipLaunchKernelGGL(arena_reset_kernel, dim3(1), dim3(1), 0, st, counter, a.abort, (uint32_t)ntiles * (uint32_t)ACH);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
#define SGS_LAUNCH_W(M_, ST_, T0_, NT_)                                                             \
	hipLaunchKernelGGL(blend_weights_kernel<M_>, dim3((((NT_) + 7) / 8) * 8), dim3(256), 0, ST_,    \
			   a.ranges, a.point_list, a.means2D, a.conic_opacity, a.final_T, a.n_contrib, \
			   act_id, (uint32_t*)nullptr, wgt, table, nbatches, counter, lay.capacity, a.W, a.H, a.gx, \
			   ((NT_) + 7) / 8, NT_, g_sweep_trace ? g_sweep_trace + 4 * 4096 : nullptr, (float4*)nullptr, 0ull, \
			   (const uint32_t*)a.tile_order)
	{
		// ---- row-sweep path (default)
		// segment length: long sweeps amortise the prologue and leave few half-line stores at segment
		// ends (cfg3: 48 -> 2 segments per tile row is 3 % faster than 16), but there must be enough
		// workgroups to fill 256 CUs x 2 a few times over.
		const int nc = a.C / 128;
		// accumulate kernel (low nibble of the variant): 6 = round 4's ping-pong sweep (one 8-wave workgroup for both row
		// parities, six bf16 products: the default); 8 = round 2's split-bf16 (three products) sweep, 9 = the same sweep on
		// fp32-input MFMA; blend_sweep2.hip: 10 = f32-equivalent (six bf16 products), 11 = exact fp32 MFMA, 14 = six products
		// with pre-split weights (round 3's default), 12 / 15 = six products on the double-rate MFMA (make X16=1).
		// (the norm-plane epilogue of N1 lives in round 3's kernel: same arithmetic, same hand-over format)
		const int arith_nib = (((split_mode & 15) == 6 || (split_mode & 15) == 5 || (split_mode & 15) == 4) && a.norm_plane) ? 14 : (split_mode & 15);
		const bool sweep3c = arith_nib == 5;   // ping-pong sweep, fp32 weights handed over and split by the sweep (one step ahead)
		const bool sweep3f = arith_nib == 4;   // ping-pong sweep without the barriers: the halves run free, flags per ring stage (experiment)
		const bool sweep3 = arith_nib == 6 || sweep3c || sweep3f;
		// a ping-pong workgroup covers both parities and there is one per CU: half as many, twice as large work items
		const int wg_per_item = sweep3 ? 1 : 2, wg_target = sweep3 ? 768 : 1536;
		int seg = ((split_mode >> 4) & 15) ? ((split_mode >> 4) & 15) * 8 : 48;
		if (seg > SEGMAX) seg = SEGMAX;
		if (((split_mode >> 4) & 15) == 0)
			while (seg > 8 && (long long)a.gy * ((a.gx + seg - 1) / seg) * nc * wg_per_item < wg_target) seg /= 2;
		const int nseg = (a.gx + seg - 1) / seg;
		seg = ((a.gx + nseg - 1) / nseg + 1) & ~1;   // balanced, even (segments start on even tiles)
		const bool sweep2 = (arith_nib >= 10 && arith_nib <= 15) || arith_nib == 7 || sweep3;   // 7 = six products, fp32 hand-over, split once per workgroup (S2_X6C)
		const bool presplit3 = arith_nib >= 14 || (sweep3 && !sweep3c);   // weights handed over as three bf16 terms
		const bool exact = arith_nib == 9;
		// weights pre-pass.  blend_weights2.hip (lane = two pixels: a third fewer instructions) is used for the fp32-row format
		// (0.25 -> 0.22 ms at cfg3; the backward's pre-pass is the same kernel).  For the three-term format it is no faster
		// (0.31-0.32 vs 0.30-0.31 ms, order-balanced A/B: 96 VGPRs, the flush through LDS), so the default path keeps round
		// 2's kernel there.  Bit 14 of the word flips the choice for A/B runs; a sweep trace needs round 2's kernel (it
		// carries the trace hooks).
		const bool flip = (split_mode & 0x4000) != 0;
		const bool w_old = (presplit3 ? !flip : flip) || g_sweep_trace != nullptr;
		// round 4: the list walked a super-batch at a time for the three-term format -- blend_weights2_sb_kernel (two pixels per lane,
		// 128 entries; the default) or blend_weights_sb_kernel (lane = pixel, 256 entries; bit 14); bit 15 of the word restores the
		// 16-entry-batch kernels for A/B runs (0x8000: lane = pixel, 0xC000: two pixels per lane; a sweep trace needs the former:
		// it carries the hooks).  fp32 rows (exact sweep, backward) stay with blend_weights2_kernel<3>.
		const bool w_sb = presplit3 && (split_mode & 0x8000) == 0 && g_sweep_trace == nullptr;
#ifndef SGS_WITH_EXPERIMENTS
		// the product library: three-term format -> blend_weights2_sb_kernel, fp32 rows (exact sweep) -> blend_weights2_kernel<3>, two
		// terms (variant 14) -> blend_weights_kernel<2>; the superseded pre-passes and the sweep trace are make EXPERIMENTS=1
		(void)w_old;
		if (flip || (split_mode & 0x8000) || g_sweep_trace != nullptr) return hipErrorInvalidValue;
		if (presplit3) {
			const hipError_t ew = launch_blend_weights2(st, 5, a.ranges, a.point_list, a.means2D, a.conic_opacity, a.final_T, a.n_contrib,
								    act_id, nullptr, wgt, table, nbatches, counter, lay.capacity, a.W, a.H, a.gx, ntiles,
								    nullptr, 0, a.tile_order);
			if (ew != hipSuccess) return ew;
		} else if (exact || sweep2) {
			const hipError_t ew = launch_blend_weights2(st, 3, a.ranges, a.point_list, a.means2D, a.conic_opacity, a.final_T, a.n_contrib,
								    act_id, nullptr, wgt, table, nbatches, counter, lay.capacity, a.W, a.H, a.gx, ntiles,
								    nullptr, 0, a.tile_order);
			if (ew != hipSuccess) return ew;
		} else SGS_LAUNCH_W(2, st, 0, ntiles);
		(void)w_sb;
#else
		if (w_sb && !flip) {   // the default: the two-pixels-per-lane kernel with the super-batch walk (blend_weights2.hip, mode 5: 0.207 ms at cfg3;
			// bit 14 of the word selects the lane-per-pixel super-batch kernel below, 0.217-0.22)
			const hipError_t ew = launch_blend_weights2(st, 5, a.ranges, a.point_list, a.means2D, a.conic_opacity, a.final_T, a.n_contrib,
								    act_id, nullptr, wgt, table, nbatches, counter, lay.capacity, a.W, a.H, a.gx, ntiles,
								    nullptr, 0, a.tile_order);
			if (ew != hipSuccess) return ew;
		} else if (w_sb) {
			const int pxw = (ntiles + 7) / 8;
			hipLaunchKernelGGL(blend_weights_sb_kernel<4>, dim3(pxw * 8), dim3(256), 0, st, a.ranges, a.point_list, a.means2D,
					   a.conic_opacity, a.final_T, a.n_contrib, act_id, (uint32_t*)nullptr, wgt, table, nbatches, counter,
					   lay.capacity, a.W, a.H, a.gx, pxw, ntiles, (const uint32_t*)a.tile_order);
		} else if ((presplit3 || exact || sweep2) && !w_old) {
			const hipError_t ew = launch_blend_weights2(st, presplit3 ? 4 : 3, a.ranges, a.point_list, a.means2D, a.conic_opacity,
								    a.final_T, a.n_contrib, act_id, nullptr, wgt, table, nbatches, counter,
								    lay.capacity, a.W, a.H, a.gx, ntiles, nullptr, 0, a.tile_order);
			if (ew != hipSuccess) return ew;
		} else if (presplit3) SGS_LAUNCH_W(4, st, 0, ntiles);
		else if (exact || sweep2) SGS_LAUNCH_W(3, st, 0, ntiles);
		else SGS_LAUNCH_W(2, st, 0, ntiles);
#endif
		if (mark) mark(mark_user);
		// workgroup order (bits [13:12] of the variant): 0 / 3 = segments sorted by work and dealt to the XCDs
		// (sweep_plan_kernel), 1 = row-major bands per XCD (the previous order), 2 = dealt, unsorted
		const int nsegs = a.gy * nseg;
		int plan = (split_mode >> 12) & 3;
		if (plan == 0) plan = 3;
		if (nsegs > PLAN_MAX && plan == 3) plan = 2;
		uint32_t* order = (uint32_t*)(arena + lay.order);
		if (plan == 3)
			hipLaunchKernelGGL(sweep_plan_kernel, dim3(1), dim3(1024), 0, st, nbatches, order, counter, a.gx, a.gy, seg, nseg,
					   a.usage_host, a.tile_order);
		if (usage_reported) *usage_reported = plan == 3 && a.usage_host != nullptr;
		const int dealt = plan >= 2;
		const int items = dealt ? nsegs : nsegs * nc * wg_per_item;   // x 2 row parities (one workgroup for both: ping-pong)
		const int pxcd = dealt ? 2 * ((nsegs + 15) / 16) * nc * wg_per_item : (items + 7) / 8;   // workgroups per XCD
		const uint32_t* order_arg = plan == 3 ? order : nullptr;
#define SGS_LAUNCH_SWEEP(D_, E_)                                                                     \
	hipLaunchKernelGGL((blend_accum_sweep_kernel<D_, E_>), dim3(pxcd * 8), dim3(256), 0, st, a.ranges, table, \
			   nbatches, act_id, (const char*)wgt, a.features, a.bg, a.out, counter, a.W,   \
			   a.H, a.C, a.gx, nc, seg, nseg, pxcd, items, a.pitch, g_sweep_trace, order_arg, dealt)
		if (sweep3) {
			const hipError_t e3 = launch_accum_sweep3(st, (split_mode >> 8) & 15, a, table, nbatches, act_id, (const char*)wgt, counter,
								  nc, seg, nseg, pxcd, items, g_sweep_trace, order_arg, dealt, (split_mode >> 16) & 15, sweep3c ? 1 : (sweep3f ? 2 : 0), (split_mode >> 20) & 3);
			if (e3 != hipSuccess) return e3;
		} else if (sweep2) {
			const hipError_t e2 = launch_accum_sweep2(st, arith_nib == 7 ? 6 : arith_nib == 11 ? 0 : (arith_nib == 12 ? 2 : (arith_nib == 13 ? 3 : (arith_nib == 14 ? 4 : (arith_nib == 15 ? 5 : 1)))), a.norm_plane ? 32 : ((split_mode >> 8) & 15), a, table,
								  nbatches, act_id, (const char*)wgt, counter, nc, seg, nseg, pxcd, items,
								  g_sweep_trace, order_arg, dealt);
			if (e2 != hipSuccess) return e2;
		}
#ifndef SGS_WITH_EXPERIMENTS
		else if (exact || ((split_mode >> 8) & 15) != 0) return hipErrorInvalidValue;   // (round 2's sweep on fp32 MFMA, its ablations: make EXPERIMENTS=1)
		else SGS_LAUNCH_SWEEP(0, false);
#else
		else if (exact) SGS_LAUNCH_SWEEP(0, true);
		else
			switch ((split_mode >> 8) & 15) {
			case 1: SGS_LAUNCH_SWEEP(1, false); break;
			case 2: SGS_LAUNCH_SWEEP(2, false); break;
			case 3: SGS_LAUNCH_SWEEP(3, false); break;
			case 4: SGS_LAUNCH_SWEEP(4, false); break;
#ifdef SGS_WITH_X16
			case 8: SGS_LAUNCH_SWEEP(8, false); break;   // v_mfma_f32_32x32x16_bf16 products (reproducer only; make X16=1)
#else
			case 8: return hipErrorInvalidValue;         // (the x16 build of this sweep is not in the product library)
#endif
			default: SGS_LAUNCH_SWEEP(0, false); break;
			}
#endif
#undef SGS_LAUNCH_SWEEP
	}
#undef SGS_LAUNCH_W
	return hipGetLastError();
}

} // namespace sgs
