// binning_rows.hip -- binning mode 0: per-tile lists without sorting the tile instances.
//
// The reference turns (Gaussian, tile) instances into per-tile depth-ordered lists with a
// 45-bit radix sort of L = 16.5 M pairs (cfg3).  With the Gaussians already in depth order
// (depth_sort.hip: the depth presort) a tile's list is just "the Gaussians whose rect covers the tile,
// in rank order", and a rect is a span of columns times a span of rows.  A stable partition of
// items that each cover a SPAN of bins needs no sort at all: for 64 consecutive items (one wave)
// and one bin, the ballot of "covers the bin" is both the chunk's count for that bin and, through
// popcount(ballot & lanes_below), every covering item's rank inside the chunk.  Two such
// partitions give the lists:
//
//   stage A  items = the P ranked Gaussians, bins = the longer tile-grid axis ("major", columns for
//            cfg3: 81): output = per column the Gaussians covering it, in depth order, each carrying
//            its row span ("major instances", R = 3.9 M for cfg3, 4.3x fewer than L);
//   stage B  items = the major instances of one column (64 per wave, never straddling columns),
//            bins = the other axis (rows, 61): output = point_list, every Gaussian id written once
//            to its final position; the per-tile counts give `ranges`.
//
// Each stage is: chunk histogram (a +1/-1 difference array per span and a wave prefix), two-level
// prefix of the chunk counts along the item order (groups of 32 chunks), an exclusive scan over the
// bins / tiles, and the ballot sweep.  No keys exist, nothing L-sized is read, and the only
// L-sized write is point_list itself.  Results are bit-identical to the reference's sorted order
// (tests: point_list, ranges, reconstructed 64-bit keys).
#include "sgs_kernels.h"

#include <cstring>

namespace sgs {

namespace {
constexpr int RCH = 64;    // items per chunk (one wave)
constexpr int RGRP = 32;   // chunks per scan group
constexpr int SCAT_CPW = 1;      // chunks per wave of the scatter kernels (4 was measured: 54 -> 76 us -- the kernel wants MORE waves in flight, not fewer launches)
// scatter kernels: SCAT_NW waves per workgroup (template; 16 unless a long bin axis makes the per-wave LDS tables too big).
// The waves are independent: a big workgroup only amortises the launch (PMC: the kernel holds 1.4 waves per SIMD in
// flight with 4-wave workgroups = workgroups launched per us x wave lifetime).
constexpr int SCAT_CAP = 512;    // stage B scatter: ids a wave stages in LDS per chunk (beyond: written directly)

// wave-uniform: the segment whose chunk range contains `id`: largest s with first[s] <= id.  The
// table (nseg + 1 entries, first[nseg] = total > id) is read 64 entries per load and resolved with a
// ballot: one memory round trip per 64 segments instead of a dependent binary search.
__device__ __forceinline__ int seg_of(const uint32_t* __restrict__ first, int nseg, uint32_t id)
{
	const int lane = threadIdx.x & 63;
	int s = 0;
	for (int b = 0; b <= nseg; b += 64) {
		const int i = b + lane;
		const bool le = i <= nseg && first[i] <= id;
		const unsigned long long m = __ballot(le);   // a prefix of the lanes (the table is non-decreasing)
		if (m == 0ull) break;
		s = b + __popcll(m) - 1;
		if (m != ~0ull) break;
	}
	return s;
}

// an item of a stage: the Gaussian, the span it covers on this stage's axis, and (stage A) the
// other axis' span as payload for stage B
struct Item {
	uint32_t g, lo, hi, payload;
	bool valid;
};

// stage A item: rank r -> Gaussian perm[r], rect from means2D / radii
__device__ __forceinline__ Item item_from_rank(uint32_t r, uint32_t n, const uint32_t* __restrict__ perm,
					       const int* __restrict__ radii, const float2* __restrict__ means2D,
					       int gx, int gy, bool major_x)
{
	Item it{0u, 0u, 0u, 0u, false};
	if (r < n) {
		const uint32_t g = perm[r];
		const int rad = radii[g];
		if (rad > 0) {
			const float2 p = means2D[g];
			uint32_t x0, y0, x1, y1;
			get_rect(p.x, p.y, rad, gx, gy, x0, y0, x1, y1);
			it.g = g;
			it.lo = major_x ? x0 : y0;
			it.hi = major_x ? x1 : y1;
			it.payload = major_x ? (y0 | (y1 << 16)) : (x0 | (x1 << 16));
			it.valid = it.hi > it.lo;
		}
	}
	return it;
}
} // namespace

// counts64[r] = major-axis bins covered << 32 | tiles covered, for the Gaussian of depth rank r; the
// rank's record rrec[r] = (Gaussian, major span lo | hi << 16, minor span lo | hi << 16, 0) is kept so
// that stage A reads its items coalesced instead of gathering perm -> radii / means2D twice more
__global__ __launch_bounds__(256) void span_counts_kernel(int P, const uint32_t* __restrict__ perm,
							   const int* __restrict__ radii,
							   const float2* __restrict__ means2D, int gx, int gy, int major_x,
							   uint64_t* __restrict__ counts64, uint4* __restrict__ rrec)
{
	const int r = blockIdx.x * 256 + threadIdx.x;
	if (r >= P) return;
	const Item it = item_from_rank((uint32_t)r, (uint32_t)P, perm, radii, means2D, gx, gy, major_x != 0);
	rrec[r] = it.valid ? make_uint4(it.g, it.lo | (it.hi << 16), it.payload, 0u) : make_uint4(0u, 0u, 0u, 0u);
	uint64_t c = 0;
	if (it.valid) {
		const uint32_t nmaj = it.hi - it.lo, nmin = (it.payload >> 16) - (it.payload & 0xffffu);
		c = ((uint64_t)nmaj << 32) | (uint64_t)(nmaj * nmin);
	}
	counts64[r] = c;
}

// chunk0 / grp0 = first chunk / scan group of each segment (segment s = items [segstart[s], segstart[s+1])).
// Stage A passes segstart == nullptr: one segment of `total` items.  One workgroup.
__global__ __launch_bounds__(64) void seg_tables_kernel(int nseg, uint32_t total, uint32_t* __restrict__ segstart,
							 bool single, uint32_t* __restrict__ chunk0,
							 uint32_t* __restrict__ grp0, const uint32_t* __restrict__ abort)
{
	if (threadIdx.x != 0) return;
	if (abort && *abort != 0u) return;
	if (single) {
		segstart[0] = 0u;
		segstart[1] = total;
	}
	uint32_t c = 0, g = 0;
	for (int s = 0; s < nseg; s++) {
		chunk0[s] = c;
		grp0[s] = g;
		const uint32_t n = segstart[s + 1] - segstart[s];
		const uint32_t nc = (n + RCH - 1) / RCH;
		c += nc;
		g += (nc + RGRP - 1) / RGRP;
	}
	chunk0[nseg] = c;
	grp0[nseg] = g;
}

// per chunk and bin: how many of the chunk's items cover the bin
template <bool FROM_RANKS>
__global__ __launch_bounds__(256) void span_hist_kernel(
	int nb, int nseg, const uint32_t* __restrict__ segstart, const uint32_t* __restrict__ chunk0,
	const uint2* __restrict__ items, const uint4* __restrict__ rrec, uint32_t* __restrict__ cmat,
	const uint32_t* __restrict__ grp0, uint4* __restrict__ desc, const uint32_t* __restrict__ abort)
{
	if (abort && *abort != 0u) return;
	extern __shared__ int s_diff[];   // [4 waves][nb + 1]
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t c = blockIdx.x * 4u + (uint32_t)wave;
	int* diff = s_diff + wave * (nb + 1);
	for (int b = lane; b <= nb; b += 64) diff[b] = 0;
	if (c >= chunk0[nseg]) return;
	const int s = seg_of(chunk0, nseg, c);
	const uint32_t idx = segstart[s] + (c - chunk0[s]) * RCH + (uint32_t)lane, end = segstart[s + 1];
	// the chunk's descriptor for the scatter kernel (segment, scan group, first item, end of the segment): the scatter's
	// waves are bound by their chain of dependent loads, and this is three links of it
	if (lane == 0) desc[c] = make_uint4((uint32_t)s, grp0[s] + (c - chunk0[s]) / RGRP, idx, end);
	uint32_t lo = 0, hi = 0;
	if (FROM_RANKS) {
		if (idx < end) {
			const uint32_t sp = rrec[idx].y;
			lo = sp & 0xffffu;
			hi = sp >> 16;
		}
	} else if (idx < end) {
		const uint32_t sp = items[idx].y;
		lo = sp & 0xffffu;
		hi = sp >> 16;
	}
	if (hi > lo) {   // difference array: +1 where the span starts, -1 one past its end
		atomicAdd(&diff[lo], 1);
		atomicAdd(&diff[hi], -1);
	}
	int carry = 0;   // inclusive prefix over the bins = coverage count per bin
	for (int b0 = 0; b0 < nb; b0 += 64) {
		const int b = b0 + lane;
		int v = b < nb ? diff[b] : 0;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int u = __shfl_up(v, o);
			if (lane >= o) v += u;
		}
		v += carry;
		if (b < nb) cmat[(size_t)c * nb + b] = (uint32_t)v;
		carry = __shfl(v, 63);
	}
}

// The two kernels above in one, for bin axes that fit LDS (nb <= HG_NB_MAX): a workgroup per scan GROUP (32 chunks, 8
// waves x 4 chunks).  The chunks' coverage rows stay in LDS, the in-group exclusive prefix over the chunks is taken
// there, and cmat receives the prefix directly -- one launch and one round trip of the count matrix less per stage.
constexpr int HG_NB_MAX = 256;
template <bool FROM_RANKS>
__global__ __launch_bounds__(512) void span_hist_group_kernel(
	int nb, int nseg, const uint32_t* __restrict__ segstart, const uint32_t* __restrict__ chunk0,
	const uint2* __restrict__ items, const uint4* __restrict__ rrec, uint32_t* __restrict__ cmat,
	const uint32_t* __restrict__ grp0, uint32_t* __restrict__ gtot, uint4* __restrict__ desc,
	const uint32_t* __restrict__ abort)
{
	if (abort && *abort != 0u) return;
	extern __shared__ int s_cov[];   // [RGRP chunks][nb + 1]: difference arrays, then coverage counts
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t G = blockIdx.x;
	if (G >= grp0[nseg]) return;
	const int s = seg_of(grp0, nseg, G);
	const uint32_t cbeg = chunk0[s] + (G - grp0[s]) * RGRP;
	const uint32_t cend = cbeg + RGRP < chunk0[s + 1] ? cbeg + RGRP : chunk0[s + 1];
	const uint32_t seg_first = segstart[s], seg_end = segstart[s + 1], c_seg0 = chunk0[s];
	const int pitch = nb + 1;
	for (int q = threadIdx.x; q < RGRP * pitch; q += 512) s_cov[q] = 0;
	__syncthreads();
	for (uint32_t c = cbeg + (uint32_t)wave; c < cend; c += 8) {   // a wave: chunks wave, wave + 8, ...
		int* diff = s_cov + (c - cbeg) * pitch;
		const uint32_t idx = seg_first + (c - c_seg0) * RCH + (uint32_t)lane;
		if (lane == 0) desc[c] = make_uint4((uint32_t)s, G, idx, seg_end);
		uint32_t lo = 0, hi = 0;
		if (idx < seg_end) {
			const uint32_t sp = FROM_RANKS ? rrec[idx].y : items[idx].y;
			lo = sp & 0xffffu;
			hi = sp >> 16;
		}
		if (hi > lo) {
			atomicAdd(&diff[lo], 1);
			atomicAdd(&diff[hi], -1);
		}
		__builtin_amdgcn_wave_barrier();
		int carry = 0;   // inclusive prefix over the bins = coverage count per bin (in place)
		for (int b0 = 0; b0 < nb; b0 += 64) {
			const int b = b0 + lane;
			int v = b < nb ? diff[b] : 0;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const int u = __shfl_up(v, o);
				if (lane >= o) v += u;
			}
			v += carry;
			if (b < nb) diff[b] = v;
			carry = __shfl(v, 63);
		}
	}
	__syncthreads();
	// exclusive prefix over the group's chunks, per bin
	for (int b = threadIdx.x; b < nb; b += 512) {
		uint32_t run = 0;
		for (uint32_t c = cbeg; c < cend; c++) {
			const uint32_t v = (uint32_t)s_cov[(c - cbeg) * pitch + b];
			cmat[(size_t)c * nb + b] = run;
			run += v;
		}
		gtot[(size_t)G * nb + b] = run;
	}
}

// in-group exclusive prefix over the chunks of a scan group (in place), group total to gtot
__global__ __launch_bounds__(256) void span_scan_groups_kernel(int nb, int nseg, const uint32_t* __restrict__ chunk0,
								const uint32_t* __restrict__ grp0,
								uint32_t* __restrict__ cmat, uint32_t* __restrict__ gtot,
								const uint32_t* __restrict__ abort)
{
	if (abort && *abort != 0u) return;
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	const uint32_t G = t / (uint32_t)nb;
	const int b = (int)(t - G * (uint32_t)nb);
	if (G >= grp0[nseg]) return;
	int s = 0;   // (per-thread, not wave-uniform: plain binary search)
	{
		int lo = 0, hi = nseg;
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (grp0[mid] <= G) lo = mid;
			else hi = mid;
		}
		s = lo;
	}
	const uint32_t cbeg = chunk0[s] + (G - grp0[s]) * RGRP;
	const uint32_t cend = cbeg + RGRP < chunk0[s + 1] ? cbeg + RGRP : chunk0[s + 1];
	uint32_t run = 0;
	for (uint32_t c = cbeg; c < cend; c++) {
		const uint32_t v = cmat[(size_t)c * nb + b];
		cmat[(size_t)c * nb + b] = run;
		run += v;
	}
	gtot[(size_t)G * nb + b] = run;
}

// per (segment, bin): exclusive prefix over the segment's groups (in place); its total is the length of
// the output list `seg * seg_stride + bin * bin_stride`.  One wave per list: the groups are split into
// 64 contiguous runs, run sums are wave-scanned (stage A has one segment with hundreds of groups).
__global__ __launch_bounds__(256) void span_scan_lists_kernel(int nb, int nseg, const uint32_t* __restrict__ grp0,
							       uint32_t* __restrict__ gtot, int seg_stride, int bin_stride,
							       uint32_t* __restrict__ listlen, const uint32_t* __restrict__ abort)
{
	if (abort && *abort != 0u) return;
	const int lane = threadIdx.x & 63;
	const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (t >= nb * nseg) return;
	const int s = t / nb, b = t - s * nb;
	const uint32_t g0 = grp0[s], n = grp0[s + 1] - g0;
	const uint32_t per = (n + 63) / 64;
	const uint32_t beg = (uint32_t)lane * per < n ? (uint32_t)lane * per : n;
	const uint32_t end = beg + per < n ? beg + per : n;
	uint32_t sum = 0;
	for (uint32_t i = beg; i < end; i++) sum += gtot[(size_t)(g0 + i) * nb + b];
	uint32_t incl = sum;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t u = (uint32_t)__shfl_up((int)incl, o);
		if (lane >= o) incl += u;
	}
	uint32_t run = incl - sum;
	for (uint32_t i = beg; i < end; i++) {
		const uint32_t v = gtot[(size_t)(g0 + i) * nb + b];
		gtot[(size_t)(g0 + i) * nb + b] = run;
		run += v;
	}
	if (lane == 63) listlen[s * seg_stride + b * bin_stride] = incl;
}

// exclusive scan of n list lengths in list order.  RANGES: ranges[t] = [start, start + len), (0, 0) for an
// empty tile as the reference's memset + identifyTileRanges leaves it (CR/cuda_rasterizer/
// rasterizer_impl.cu:116-138,313); else starts[t] = start, starts[n] = total.  One workgroup.
// RANGES also writes the list starts once more, segment-major (tstart[s * nb + b]): the stage B scatter reads a
// segment's nb starts per chunk, which in `ranges` are gx * 8 bytes apart (one cache line each).
template <bool RANGES>
__global__ __launch_bounds__(1024) void list_scan_kernel(int n, const uint32_t* __restrict__ len,
							  uint2* __restrict__ ranges, uint32_t* __restrict__ starts,
							  const uint32_t* __restrict__ abort, int gx, int major_x, int nb,
							  uint32_t* __restrict__ arena_counter, uint32_t first_free)
{
	// (RANGES, the last single-workgroup kernel in front of the blend) also resets the split blend's work-list
	// counter -- what blend_fwd_split.hip's arena_reset_kernel does -- so that launch disappears from the frame
	if (arena_counter && threadIdx.x == 0) {
		arena_counter[0] = first_free;
		arena_counter[1] = (abort && *abort != 0u) ? 2u : 0u;
	}
	if (abort && *abort != 0u) return;
	__shared__ uint32_t s_part[1024];
	const int per = (n + 1023) / 1024;
	const int t0 = threadIdx.x * per < n ? threadIdx.x * per : n, t1 = (t0 + per < n) ? t0 + per : n;
	uint32_t sum = 0;
	for (int t = t0; t < t1; t++) sum += len[t];
	s_part[threadIdx.x] = sum;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan of the partials
		const uint32_t v = (int)threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
		__syncthreads();
		s_part[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t run = s_part[threadIdx.x] - sum;
	for (int t = t0; t < t1; t++) {
		const uint32_t m = len[t];
		if (RANGES) {
			ranges[t] = m ? make_uint2(run, run + m) : make_uint2(0u, 0u);
			const int x = t % gx, y = t / gx;
			starts[(major_x ? x : y) * nb + (major_x ? y : x)] = run;
		} else {
			starts[t] = run;
		}
		run += m;
	}
	if (!RANGES && threadIdx.x == 1023) starts[n] = s_part[1023];
}

// Stage A's two single-workgroup steps in one launch (round 4): the exclusive scan of the nb bin lengths (they are stage B's
// segment starts; list_scan_kernel<false>) and stage B's chunk / group tables (seg_tables_kernel).  nb <= 2048 (the row
// builder's own limit).  (The per-bin prefix over the scan groups stays a kernel of its own: 81 bins x 488 groups want more
// than one workgroup -- folded in here it took 78 us instead of 9.)
__global__ __launch_bounds__(1024) void stage_a_lists_kernel(int nb, const uint32_t* __restrict__ binlen, uint32_t* __restrict__ segB,
							      uint32_t* __restrict__ chunk0B, uint32_t* __restrict__ grp0B,
							      const uint32_t* __restrict__ abort)
{
	if (abort && *abort != 0u) return;
	__shared__ uint32_t s_len[2048 + 1];
	for (int b = threadIdx.x; b < nb; b += 1024) s_len[b] = binlen[b];
	__syncthreads();
	// exclusive scan of the nb lengths: thread t owns bins 2 t, 2 t + 1
	__shared__ uint32_t s_part[1024];
	const int b0 = 2 * (int)threadIdx.x;
	const uint32_t v0 = b0 < nb ? s_len[b0] : 0u, v1 = b0 + 1 < nb ? s_len[b0 + 1] : 0u;
	s_part[threadIdx.x] = v0 + v1;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {
		const uint32_t a = (int)threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
		__syncthreads();
		s_part[threadIdx.x] += a;
		__syncthreads();
	}
	const uint32_t start = s_part[threadIdx.x] - (v0 + v1);
	if (b0 < nb) segB[b0] = start;
	if (b0 + 1 < nb) segB[b0 + 1] = start + v0;
	if (threadIdx.x == 1023) segB[nb] = s_part[1023];
	// stage B's tables: the same scan over (chunks << 32 | scan groups) of each segment
	__shared__ unsigned long long s_cg[1024];
	const uint32_t nc0 = (v0 + RCH - 1) / RCH, nc1 = (v1 + RCH - 1) / RCH;
	const unsigned long long cg0 = ((unsigned long long)nc0 << 32) | (unsigned long long)((nc0 + RGRP - 1) / RGRP);
	const unsigned long long cg1 = ((unsigned long long)nc1 << 32) | (unsigned long long)((nc1 + RGRP - 1) / RGRP);
	s_cg[threadIdx.x] = cg0 + cg1;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {
		const unsigned long long a = (int)threadIdx.x >= off ? s_cg[threadIdx.x - off] : 0ull;
		__syncthreads();
		s_cg[threadIdx.x] += a;
		__syncthreads();
	}
	const unsigned long long cgs = s_cg[threadIdx.x] - (cg0 + cg1);
	if (b0 < nb) {
		chunk0B[b0] = (uint32_t)(cgs >> 32);
		grp0B[b0] = (uint32_t)cgs;
	}
	if (b0 + 1 < nb) {
		chunk0B[b0 + 1] = (uint32_t)((cgs + cg0) >> 32);
		grp0B[b0 + 1] = (uint32_t)(cgs + cg0);
	}
	if (threadIdx.x == 1023) {   // (bins beyond nb contribute nothing: the last partial is the total)
		chunk0B[nb] = (uint32_t)(s_cg[1023] >> 32);
		grp0B[nb] = (uint32_t)s_cg[1023];
	}
}

// (Round 6, measured and removed -- git history has the code: the two one-workgroup steps -- stage A's stage_a_lists_kernel, stage B's
// list_scan_kernel<true> -- riding in the LAST workgroup of the span_scan_lists_kernel in front of them (a device-scope ticket per stage).  With a
// __threadfence() pair around the ticket the 1 236-workgroup stage-B kernel took 129 us instead of 5 + 9 (an XCD's L2 is not coherent with the other
// seven: an agent-scope fence is a whole-L2 write-back + invalidate per workgroup); with the hand-over on device-scope word atomics instead of fences it
// still cost 8 us MORE than the two kernels it replaced -- the tail workgroup's 256 threads scan 4 941 lengths with L2-bypassing loads where the dedicated
// kernel has 1 024 threads and a warm L2, and two launches of back-to-back kernels on one stream cost almost nothing.  profiles/r06_front_end_ab.txt.
// A third form -- the depth sort's recipe, which DID pay there: roles dealt by a ticket, the last ticket a dedicated 1 024-thread scanner workgroup that polls
// the lengths as FLAG | length words while the producers are still working -- was correct (141 tests) and 15 us slower per frame than the four small
// kernels (stage A 23.7 us instead of 9.0 + 5.3, stage B 15.1 instead of 4.9 + 9.0, profiles/r06_span_chain.txt): the producers' ticket and the scanner's
// polls are L2-bypassing round trips in front of work that takes a few microseconds.  Removed as well.)

// the ballot sweep: every covering item is written to its final position in the list of
// (segment, bin).  OUT_ITEMS: stage A (the item with its payload span); else the Gaussian id.
template <bool FROM_RANKS, int SCAT_NW>
__global__ __launch_bounds__(64 * SCAT_NW) void span_scatter_kernel(
	int nb, int nseg, const uint32_t* __restrict__ segstart, const uint32_t* __restrict__ chunk0,
	const uint32_t* __restrict__ grp0, const uint2* __restrict__ items, const uint4* __restrict__ rrec,
	const uint32_t* __restrict__ cmat, const uint32_t* __restrict__ gtot, const uint32_t* __restrict__ starts,
	const uint2* __restrict__ ranges, int seg_stride, int bin_stride, uint2* __restrict__ out_items,
	uint32_t* __restrict__ point_list, const uint4* __restrict__ desc, const uint32_t* __restrict__ abort)
{
	if (abort && *abort != 0u) return;
	extern __shared__ uint32_t s_base[];   // [4 waves][nb]: first position of this chunk in each list
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// a wave takes SCAT_CPW consecutive chunks (1: see the constant)
	const uint32_t nchunks_all = chunk0[nseg];
	auto one_chunk = [&](const uint32_t c) __attribute__((always_inline)) {
	const uint4 dsc = desc[c];   // written by span_hist_kernel: (segment, scan group, first item, segment end)
	const int s = (int)__builtin_amdgcn_readfirstlane((int)dsc.x);
	const uint32_t G = (uint32_t)__builtin_amdgcn_readfirstlane((int)dsc.y);
	uint32_t* base = s_base + wave * nb;
	for (int b = lane; b < nb; b += 64) {
		const int list = s * seg_stride + b * bin_stride;
		base[b] = (FROM_RANKS ? starts[list] : starts[(size_t)s * nb + b]) + gtot[(size_t)G * nb + b] + cmat[(size_t)c * nb + b];
	}
	const uint32_t idx = dsc.z + (uint32_t)lane, end = dsc.w;
	uint32_t g = 0, lo = 0, hi = 0, payload = 0;
	if (FROM_RANKS) {
		if (idx < end) {
			const uint4 v = rrec[idx];
			g = v.x;
			lo = v.y & 0xffffu;
			hi = v.y >> 16;
			payload = v.z;
		}
	} else if (idx < end) {
		const uint2 v = items[idx];
		g = v.x;
		lo = v.y & 0xffffu;
		hi = v.y >> 16;
	}
	// the wave's bin range
	uint32_t blo = hi > lo ? lo : 0xffffffffu, bhi = hi;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) {
		const uint32_t a = (uint32_t)__shfl_xor((int)blo, o), b2 = (uint32_t)__shfl_xor((int)bhi, o);
		blo = a < blo ? a : blo;
		bhi = b2 > bhi ? b2 : bhi;
	}
	blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)blo);
	bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)bhi);
	if (blo >= bhi) return;   // no item of this chunk covers a bin (next chunk)
	// Which lanes cover bin b, as a 64-bit mask per bin in LDS: every lane ORs its bit into the bins of its own span
	// (L / R = 4 bins per item on average) -- instead of one ballot per bin of the wave's whole range, which for 64
	// depth-consecutive items is most of the axis (61 ballots + popcounts per wave; the kernel was issue bound on
	// them).  The rank of an item inside a bin's list is still popcount(mask & lanes_below): same positions, same bits.
	unsigned long long* cover = reinterpret_cast<unsigned long long*>(s_base + SCAT_NW * nb) + (size_t)wave * nb;   // (16-byte aligned)
	for (uint32_t b = blo + (uint32_t)lane; b < bhi; b += 64) cover[b] = 0ull;
	__builtin_amdgcn_wave_barrier();
	const unsigned long long mine = 1ull << lane;
	for (uint32_t b = lo; b < hi; b++) atomicOr(&cover[b], mine);
	__builtin_amdgcn_wave_barrier();
	const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	if (!FROM_RANKS) {
		// Stage B writes 4-byte ids, ~4 per item, each into a different list: written straight from the lane loop
		// that is 64 separate memory requests per instruction and the kernel is bound by the L2's request rate
		// (16.5 M requests at cfg3).  So the chunk's output is first laid out in LDS grouped by bin (a bin's
		// entries are consecutive in its list), then written with consecutive lanes on consecutive entries: one
		// request per run instead of one per id.
		uint32_t* off = s_base + 3 * SCAT_NW * nb + (size_t)wave * nb;                      // [waves][nb] after the masks
		uint32_t* st_g = s_base + 4 * SCAT_NW * nb + (size_t)wave * SCAT_CAP;               // [waves][SCAT_CAP]
		uint16_t* st_b = reinterpret_cast<uint16_t*>(s_base + 4 * SCAT_NW * nb + SCAT_NW * SCAT_CAP) + (size_t)wave * SCAT_CAP;
		uint32_t total = 0;
		for (uint32_t b0 = blo; b0 < bhi; b0 += 64) {   // per bin: entries of this chunk, exclusive prefix
			const uint32_t b = b0 + (uint32_t)lane;
			const uint32_t c = b < bhi ? (uint32_t)__popcll(cover[b]) : 0u;
			uint32_t incl = c;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const uint32_t u = (uint32_t)__shfl_up((int)incl, o);
				if (lane >= o) incl += u;
			}
			if (b < bhi) {
				off[b] = total + incl - c;
				base[b] -= total + incl - c;   // list position of staged entry i of bin b = i + base[b]
			}
			total += (uint32_t)__shfl((int)incl, 63);
		}
		__builtin_amdgcn_wave_barrier();
		if (total <= (uint32_t)SCAT_CAP) {
			for (uint32_t b = lo; b < hi; b++) {
				const uint32_t i = off[b] + (uint32_t)__popcll(cover[b] & below);
				st_g[i] = g;
				st_b[i] = (uint16_t)b;
			}
			__builtin_amdgcn_wave_barrier();
			for (uint32_t i = (uint32_t)lane; i < total; i += 64) point_list[i + base[st_b[i]]] = st_g[i];
		} else {   // (a chunk of very large items: more entries than the staging area holds)
			for (uint32_t b = lo; b < hi; b++) point_list[off[b] + base[b] + (uint32_t)__popcll(cover[b] & below)] = g;
		}
		return;
	}
	for (uint32_t b = lo; b < hi; b++) {
		const uint32_t pos = base[b] + (uint32_t)__popcll(cover[b] & below);
		out_items[pos] = make_uint2(g, payload);
	}
	};
	const uint32_t c_first = (blockIdx.x * (uint32_t)SCAT_NW + (uint32_t)wave) * (uint32_t)SCAT_CPW;
	for (int it = 0; it < SCAT_CPW; it++) {
		if (c_first + (uint32_t)it >= nchunks_all) break;
		one_chunk(c_first + (uint32_t)it);
		__builtin_amdgcn_wave_barrier();   // the next chunk reuses this wave's LDS areas
	}
}

void row_binning_stage_a_counts(int P, uint32_t* chunks, uint32_t* groups)
{
	const uint32_t chA = ((uint32_t)P + RCH - 1) / RCH;
	*chunks = chA;
	*groups = (chA + RGRP - 1) / RGRP;
}

// sizes of the builder's scratch in 32-bit words.  R = major instances (stage A output)
void row_binning_scratch(int P, uint32_t R, int gx, int gy, size_t* tab_words, size_t* cmat_words,
			 size_t* gtot_words, size_t* len_words)
{
	const int nbA = gx >= gy ? gx : gy, nbB = gx >= gy ? gy : gx;
	const size_t chA = ((size_t)P + RCH - 1) / RCH, grA = (chA + RGRP - 1) / RGRP;
	const size_t chB = (size_t)R / RCH + (size_t)nbA, grB = chB / RGRP + (size_t)nbA;
	const size_t cmA = chA * nbA, cmB = chB * nbB, gtA = grA * nbA, gtB = grB * nbB;
	*tab_words = 3 * ((size_t)nbA + 2) + 8;          // segstart | chunk0 | grp0 of stage B (+ stage A's two-entry tables)
	*cmat_words = (cmA > cmB ? cmA : cmB) + 4 * (chA > chB ? chA : chB) + 4;   // the stages run one after the other; + chunk descriptors
	*gtot_words = gtA > gtB ? gtA : gtB;
	*len_words = 2 * (size_t)gx * gy + nbA + 2;       // tile lengths (stage B) | bin lengths (stage A) | segment-major list starts
}

hipError_t launch_row_binning(hipStream_t st, int P, uint32_t R, int gx, int gy, const uint4* rrec, uint2* items, uint32_t* tabs, uint32_t* cmat,
			      uint32_t* gtot, uint32_t* lens, uint2* ranges, uint32_t* point_list, const uint32_t* abort,
			      uint32_t* arena_counter, uint32_t arena_first_free, const uint32_t* stage_a_tab)
{
	const int ntiles = gx * gy;
	if (R == 0) return hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)ntiles, st);
	const int major_x = gx >= gy;
	const int nbA = major_x ? gx : gy, nbB = major_x ? gy : gx;
	// tables: stage A (one segment) uses 3 x 2 words at the end; stage B: segstart = stage A's bin starts
	uint32_t* segB = tabs;                    // nbA + 1 (+1 spare)
	uint32_t* chunk0B = tabs + (nbA + 2);
	uint32_t* grp0B = tabs + 2 * (nbA + 2);
	// stage A's tables: written by the depth sort's last pass when the caller asked for it (stage_a_tab), else by a table kernel here
	uint32_t* segA = stage_a_tab ? const_cast<uint32_t*>(stage_a_tab) : tabs + 3 * (nbA + 2);
	uint32_t* chunk0A = segA + 2;
	uint32_t* grp0A = segA + 4;
	uint32_t* binlen = lens + ntiles;         // stage A list lengths
	uint32_t* tstart = lens + ntiles + nbA + 2;   // stage B list starts, segment-major
	const uint32_t chA = ((uint32_t)P + RCH - 1) / RCH, grA = (chA + RGRP - 1) / RGRP;
	const size_t ldsA = (size_t)4 * (nbA + 1) * 4, ldsB = (size_t)4 * (nbB + 1) * 4;
	const uint32_t chB_ub = R / RCH + (uint32_t)nbA;
	// chunk descriptors (16 B each, 16-byte aligned) behind the larger of the two count matrices
	const size_t cmA_w = (size_t)chA * nbA, cmB_w = (size_t)chB_ub * nbB;
	uint4* desc = reinterpret_cast<uint4*>(cmat + (((cmA_w > cmB_w ? cmA_w : cmB_w) + 3) & ~(size_t)3));
	// the ballot-free scatter: [4 waves][nb] list bases (uint32) + [4 waves][nb] cover masks (uint64, 8-byte aligned)
	// per wave: bases + masks (+ prefixes + staging in stage B); as many waves per workgroup as fit 64 KB
	const size_t ldsWA = (size_t)nbA * 12, ldsWB = (size_t)nbB * 16 + (size_t)SCAT_CAP * 6;
	const int nwA = 16 * ldsWA <= 65536 ? 16 : (4 * ldsWA <= 65536 ? 4 : 1);
	const int nwB = 4 * ldsWB <= 65536 ? 4 : 1;   // (16 measured slower for stage B: 58 vs 53 us; stage A: 20 vs 22)
	const size_t ldsSA = nwA * ldsWA, ldsSB = nwB * ldsWB;

	// ---- stage A: ranked Gaussians -> major instances grouped by major bin
	if (!stage_a_tab) hipLaunchKernelGGL(seg_tables_kernel, dim3(1), dim3(64), 0, st, 1, (uint32_t)P, segA, true, chunk0A, grp0A, abort);
	if (nbA <= HG_NB_MAX) {
		hipLaunchKernelGGL(span_hist_group_kernel<true>, dim3(grA), dim3(512), (size_t)RGRP * (nbA + 1) * 4, st, nbA, 1,
				   segA, chunk0A, (const uint2*)nullptr, rrec, cmat, grp0A, gtot, desc, abort);
	} else {
		hipLaunchKernelGGL(span_hist_kernel<true>, dim3((chA + 3) / 4), dim3(256), ldsA, st, nbA, 1, segA, chunk0A,
				   (const uint2*)nullptr, rrec, cmat, grp0A, desc, abort);
		hipLaunchKernelGGL(span_scan_groups_kernel, dim3((unsigned)(((size_t)grA * nbA + 255) / 256)), dim3(256), 0, st,
				   nbA, 1, chunk0A, grp0A, cmat, gtot, abort);
	}
	hipLaunchKernelGGL(span_scan_lists_kernel, dim3((nbA + 3) / 4), dim3(256), 0, st, nbA, 1, grp0A, gtot, 0, 1,
			   binlen, abort);
	// (the bin lengths' scan = stage B's segment starts, and stage B's chunk / group tables: one launch)
	hipLaunchKernelGGL(stage_a_lists_kernel, dim3(1), dim3(1024), 0, st, nbA, binlen, segB, chunk0B, grp0B, abort);
#define SGS_SCATTER_A(NW_)                                                                                          \
	hipLaunchKernelGGL((span_scatter_kernel<true, NW_>), dim3((chA + NW_ * SCAT_CPW - 1) / (NW_ * SCAT_CPW)),  \
			   dim3(64 * NW_), ldsSA, st, nbA, 1, segA, chunk0A, grp0A, (const uint2*)nullptr, rrec,    \
			   cmat, gtot, segB, (const uint2*)nullptr, 0, 1, items, (uint32_t*)nullptr, desc, abort)
	if (nwA == 16) SGS_SCATTER_A(16);
	else if (nwA == 4) SGS_SCATTER_A(4);
	else SGS_SCATTER_A(1);
#undef SGS_SCATTER_A

	// ---- stage B: the major instances of each major bin -> per-tile lists
	const uint32_t chB = R / RCH + (uint32_t)nbA, grB = chB / RGRP + (uint32_t)nbA;   // upper bounds
	const int seg_stride = major_x ? 1 : gx, bin_stride = major_x ? gx : 1;          // tile = y * gx + x
	if (nbB <= HG_NB_MAX) {
		hipLaunchKernelGGL(span_hist_group_kernel<false>, dim3(grB), dim3(512), (size_t)RGRP * (nbB + 1) * 4, st, nbB, nbA,
				   segB, chunk0B, items, rrec, cmat, grp0B, gtot, desc, abort);
	} else {
		hipLaunchKernelGGL(span_hist_kernel<false>, dim3((chB + 3) / 4), dim3(256), ldsB, st, nbB, nbA, segB, chunk0B, items,
				   rrec, cmat, grp0B, desc, abort);
		hipLaunchKernelGGL(span_scan_groups_kernel, dim3((unsigned)(((size_t)grB * nbB + 255) / 256)), dim3(256), 0, st,
				   nbB, nbA, chunk0B, grp0B, cmat, gtot, abort);
	}
	hipLaunchKernelGGL(span_scan_lists_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, nbB, nbA, grp0B, gtot,
			   seg_stride, bin_stride, lens, abort);
	hipLaunchKernelGGL(list_scan_kernel<true>, dim3(1), dim3(1024), 0, st, ntiles, lens, ranges, tstart, abort, gx, major_x, nbB,
			   arena_counter, arena_first_free);
#define SGS_SCATTER_B(NW_)                                                                                          \
	hipLaunchKernelGGL((span_scatter_kernel<false, NW_>), dim3((chB + NW_ * SCAT_CPW - 1) / (NW_ * SCAT_CPW)), \
			   dim3(64 * NW_), ldsSB, st, nbB, nbA, segB, chunk0B, grp0B, items, rrec, cmat, gtot,      \
			   tstart, ranges, seg_stride, bin_stride, (uint2*)nullptr, point_list, desc, abort)
	if (nwB == 4) SGS_SCATTER_B(4);
	else SGS_SCATTER_B(1);
#undef SGS_SCATTER_B
	return hipGetLastError();
}

// keys_sorted[i] = tile << 32 | depth bits of point_list[i], from `ranges` (parity tests only)
__global__ __launch_bounds__(256) void reconstruct_keys_ranges_kernel(int ntiles, const uint2* __restrict__ ranges,
								       const uint32_t* __restrict__ point_list,
								       const float* __restrict__ depths,
								       uint64_t* __restrict__ keys_sorted)
{
	const int t = blockIdx.x;
	if (t >= ntiles) return;
	const uint2 r = ranges[t];
	for (uint32_t i = r.x + threadIdx.x; i < r.y; i += 256)
		keys_sorted[i] = ((uint64_t)(uint32_t)t << 32) | (uint64_t)__float_as_uint(depths[point_list[i]]);
}

void launch_reconstruct_keys_ranges(hipStream_t st, int ntiles, const uint2* ranges, const uint32_t* point_list,
				    const float* depths, uint64_t* keys_sorted)
{
	if (ntiles <= 0) return;
	hipLaunchKernelGGL(reconstruct_keys_ranges_kernel, dim3(ntiles), dim3(256), 0, st, ntiles, ranges, point_list,
			   depths, keys_sorted);
}

} // namespace sgs
