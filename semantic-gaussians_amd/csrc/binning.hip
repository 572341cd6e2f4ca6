// binning.hip -- tile binning: offsets scan, key emission, 64-bit radix sort, tile ranges.
//
// All integer work; outputs must equal the reference's (and the oracle's) bit for bit:
//   offsets = inclusive_sum(tiles_touched)                   (rasterizer_impl.cu:279)
//   key     = (tile_id << 32) | float_bits(depth), val = idx, emitted row-major over the
//             Gaussian's tile rect starting at offsets[idx-1]  (rasterizer_impl.cu:70-111)
//   stable ascending sort on key bits [0, 32+msb(tiles))      (rasterizer_impl.cu:302-311)
//   ranges[tile] = [first, one-past-last) index               (rasterizer_impl.cu:116-138)
// (all citations: CR/cuda_rasterizer/ of the reference).
//
// MI355X notes.
//
// (1) Depth-presorted emission (default, binning mode 0).  The reference radix-sorts all
// L = sum(tiles touched) instances on 32+13 key bits: six 8-bit passes that each move 24 B per
// instance (2.4 GB at L = 16.5M -- as expensive on MI355X as the whole C=512 blend).  The
// required result is a total order: ascending (tile, depth bits, Gaussian index).  Here the
// P Gaussians (not the L instances) are first sorted by depth bits (a stable sort keeps
// ascending index inside equal depths); instances are then EMITTED in that order, so the
// big sort only has to be stable on the 13 tile bits: two passes instead of six, and the
// outcome is bit-identical.
//
// (2) The reference emits keys with one thread per Gaussian looping over its
// rect: lanes of a wave then write 64 unrelated 12-byte records per iteration and the
// wave runs as long as its largest Gaussian.  Here emission is instance-parallel: one lane
// per OUTPUT slot, which finds its Gaussian by a binary search over the (L2-resident)
// offsets array, so every store is a fully coalesced 512-B/256-B wave store and the work
// is perfectly balanced.  The plain scan and radix sort are rocPRIM library calls on the
// caller's stream (the reference uses CUB for the same two steps).
//
// Round 6: the two rocPRIM calls (and with them ~450 library kernel instantiations, 3.2 MB of the library's 4.2 MB of device code)
// are only reached from binning modes 1 / 2 -- the reference-order witness and round 1's default -- which no product path selects
// (mode 0 = span partitions, binning_rows.hip, on depth_sort.hip's own radix sort).  They are built by `make EXPERIMENTS=1`; the
// product library answers those modes with SGS_EINVAL and carries no rocprim:: symbol.
#include "sgs_kernels.h"
#ifdef SGS_WITH_EXPERIMENTS
#include <cstring>   // rocPRIM's texture iterator calls host memset
#include <rocprim/rocprim.hpp>
#endif

namespace sgs {

#ifndef SGS_WITH_EXPERIMENTS
size_t scan_temp_bytes(int) { return 0; }
hipError_t launch_inclusive_scan(hipStream_t, void*, size_t, const uint32_t*, uint32_t*, int) { return hipErrorNotSupported; }
size_t sort_temp_bytes(size_t, int, int) { return 0; }
hipError_t launch_sort_pairs(hipStream_t, void*, size_t, uint64_t*, uint64_t*, uint32_t*, uint32_t*, size_t, int, int) { return hipErrorNotSupported; }
size_t sort32_temp_bytes(size_t, int) { return 0; }
hipError_t launch_sort32_pairs(hipStream_t, void*, size_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, size_t, int) { return hipErrorNotSupported; }
#else
size_t scan_temp_bytes(int P)
{
	size_t bytes = 0;
	(void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
				      (size_t)P, rocprim::plus<uint32_t>(), (hipStream_t)0);
	return bytes;
}

hipError_t launch_inclusive_scan(hipStream_t st, void* temp, size_t temp_bytes,
				 const uint32_t* in, uint32_t* out, int P)
{
	return rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)P,
				       rocprim::plus<uint32_t>(), st);
}
#endif

// One lane per emitted (tile, Gaussian) instance.  `perm` (optional) maps emission rank ->
// Gaussian index: null = the reference's emission order (ascending index), else the
// depth-sorted order of mode 0 (offsets are then the scan over the permuted tile counts).
__global__ __launch_bounds__(256) void duplicate_with_keys_kernel(
	int P, uint32_t L, const float2* __restrict__ means2D, const float* __restrict__ depths,
	const uint32_t* __restrict__ offsets, const int* __restrict__ radii, int gx, int gy,
	uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, const uint32_t* __restrict__ perm)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= L) return;
	// smallest g with offsets[g] > i  (upper bound); Gaussians with 0 tiles are skipped
	// automatically because their offset equals their predecessor's.
	int lo = 0, hi = P - 1;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (offsets[mid] > i) hi = mid;
		else lo = mid + 1;
	}
	const uint32_t base = (lo == 0) ? 0u : offsets[lo - 1];
	const int g = perm ? (int)perm[lo] : lo;
	const uint32_t k = i - base;
	const float2 p = means2D[g];
	uint32_t x0, y0, x1, y1;
	get_rect(p.x, p.y, radii[g], gx, gy, x0, y0, x1, y1);
	const uint32_t w = x1 - x0;
	const uint32_t ty = y0 + k / w, tx = x0 + k % w;
	uint64_t key = (uint64_t)(ty * (uint32_t)gx + tx);
	key <<= 32;
	key |= (uint64_t)__float_as_uint(depths[g]);
	keys[i] = key;
	vals[i] = (uint32_t)g;
}

void launch_duplicate_with_keys(hipStream_t st, int P, const float2* means2D, const float* depths,
				const uint32_t* offsets, const int* radii, int gx, int gy,
				uint64_t* keys, uint32_t* vals, uint32_t L, const uint32_t* perm)
{
	if (L == 0) return;
	hipLaunchKernelGGL(duplicate_with_keys_kernel, dim3((L + 255u) / 256u), dim3(256), 0, st, P,
			   L, means2D, depths, offsets, radii, gx, gy, keys, vals, perm);
}

__global__ __launch_bounds__(256) void gather_counts_kernel(int P, const uint32_t* __restrict__ perm,
							     const uint32_t* __restrict__ tiles_touched,
							     uint32_t* __restrict__ counts_sorted)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < P) counts_sorted[i] = tiles_touched[perm[i]];
}

void launch_gather_counts(hipStream_t st, int P, const uint32_t* perm, const uint32_t* tiles_touched,
			  uint32_t* counts_sorted)
{
	hipLaunchKernelGGL(gather_counts_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, perm,
			   tiles_touched, counts_sorted);
}

#ifdef SGS_WITH_EXPERIMENTS
size_t sort_temp_bytes(size_t L, int begin_bit, int end_bit)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr,
					(uint32_t*)nullptr, (uint32_t*)nullptr, L, (unsigned)begin_bit,
					(unsigned)end_bit, (hipStream_t)0);
	return bytes;
}

// stable sort on key bits [begin_bit, end_bit)
hipError_t launch_sort_pairs(hipStream_t st, void* temp, size_t temp_bytes, uint64_t* keys_in,
			     uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t L,
			     int begin_bit, int end_bit)
{
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, L,
					 (unsigned)begin_bit, (unsigned)end_bit, st);
}
#endif

__global__ __launch_bounds__(256) void tile_ranges_kernel(uint32_t L,
							   const uint64_t* __restrict__ keys,
							   uint2* __restrict__ ranges)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= L) return;
	const uint32_t cur = (uint32_t)(keys[i] >> 32);
	if (i == 0) ranges[cur].x = 0;
	else {
		const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
		if (cur != prev) {
			ranges[prev].y = i;
			ranges[cur].x = i;
		}
	}
	if (i == L - 1) ranges[cur].y = L;
}

void launch_tile_ranges(hipStream_t st, size_t L, const uint64_t* keys, uint2* ranges, int ntiles)
{
	(void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)ntiles, st);
	if (L == 0) return;
	hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st,
			   (uint32_t)L, keys, ranges);
}

// ---------------------------------------------------------------------------------------
// Mode 0 emission: 32-bit tile keys, workgroup-staged search.
//
// After the depth presort every Gaussian with tiles (radius > 0) precedes every culled one
// (their depth key is 0xFFFFFFFF), so the ranks that cover a window of IPB consecutive
// instances number at most IPB.  A workgroup therefore (1) finds the first/last rank of its
// window with two binary searches in global memory, (2) stages those ranks' offsets and tile
// rects in LDS once, (3) resolves each of its instances with an 11-step search in LDS instead
// of a 20-step search through L2, and re-uses the rect instead of recomputing it per instance.
// Only the tile id (<= 16 bits here) is emitted as the sort key: the instance sort is stable
// on the tile bits and the depth order is already in the emission order.
constexpr int IPB = 2048;   // instances per workgroup

__global__ __launch_bounds__(256) void emit_tile_keys_kernel(
	int P, uint32_t L, const float2* __restrict__ means2D, const uint32_t* __restrict__ offsets,
	const int* __restrict__ radii, const uint32_t* __restrict__ perm, int gx, int gy,
	uint32_t* __restrict__ keys32, uint32_t* __restrict__ vals)
{
	__shared__ uint32_t s_off[IPB + 1];    // s_off[k] = offsets[r0 + k - 1] (exclusive start of rank r0+k)
	__shared__ uint32_t s_g[IPB];
	__shared__ uint32_t s_rect[IPB];       // x0 | y0 << 16
	__shared__ uint32_t s_w[IPB];          // rect width in tiles
	__shared__ int s_r0, s_r1;
	const uint32_t i0 = blockIdx.x * (uint32_t)IPB;
	const uint32_t i1 = (i0 + IPB < L ? i0 + IPB : L) - 1u;   // last instance of the window
	// waves 0 and 1 find the window's first / last rank: a 64-ary search (4 dependent loads for 1M
	// ranks instead of the 20 of a binary search -- this prologue is the kernel's critical path)
	if (threadIdx.x < 128) {
		const int lane = threadIdx.x & 63;
		const uint32_t target = threadIdx.x >= 64 ? i1 : i0;
		int lo = 0, hi = P - 1;   // smallest rank with offsets[rank] > target lies in [lo, hi]
		while (lo < hi) {
			const int span = hi - lo;                       // probes at lo + (l+1)*step - 1, the last one at hi
			const int step = (span + 64) / 64;
			int idx = lo + (lane + 1) * step - 1;
			idx = idx < hi ? idx : hi;
			const bool above = offsets[idx] > target;
			const unsigned long long m = __ballot(above);   // non-zero: offsets[hi] > target
			const int f = __builtin_ctzll(m);
			const int new_hi = lo + (f + 1) * step - 1;
			hi = new_hi < hi ? new_hi : hi;
			lo = f == 0 ? lo : lo + f * step;
		}
		if (lane == 0) {
			if (threadIdx.x >= 64) s_r1 = lo;
			else s_r0 = lo;
		}
	}
	__syncthreads();
	const int r0 = s_r0, nr = s_r1 - r0 + 1;   // nr <= IPB (every rank in the window has >= 1 tile)
	for (int k = threadIdx.x; k <= nr; k += 256) {
		const int r = r0 + k - 1;
		s_off[k] = r < 0 ? 0u : offsets[r];
	}
	for (int k = threadIdx.x; k < nr; k += 256) {
		const uint32_t g = perm[r0 + k];
		const float2 p = means2D[g];
		uint32_t x0, y0, x1, y1;
		get_rect(p.x, p.y, radii[g], gx, gy, x0, y0, x1, y1);
		s_g[k] = g;
		s_rect[k] = x0 | (y0 << 16);
		s_w[k] = x1 - x0;
	}
	__syncthreads();
#pragma unroll
	for (int it = 0; it < IPB / 256; it++) {
		const uint32_t i = i0 + it * 256 + threadIdx.x;
		if (i > i1) break;
		int lo = 0, hi = nr - 1;   // largest k with s_off[k] <= i
		while (lo < hi) {
			const int mid = (lo + hi + 1) >> 1;
			if (s_off[mid] <= i) lo = mid;
			else hi = mid - 1;
		}
		const uint32_t k = i - s_off[lo];
		const uint32_t rc = s_rect[lo];
		const uint32_t x0 = rc & 0xffffu, y0 = rc >> 16, w = s_w[lo];
		keys32[i] = (y0 + k / w) * (uint32_t)gx + (x0 + k % w);
		vals[i] = s_g[lo];
	}
}

void launch_emit_tile_keys(hipStream_t st, int P, uint32_t L, const float2* means2D,
			   const uint32_t* offsets, const int* radii, const uint32_t* perm, int gx,
			   int gy, uint32_t* keys32, uint32_t* vals)
{
	if (L == 0) return;
	hipLaunchKernelGGL(emit_tile_keys_kernel, dim3((L + IPB - 1) / IPB), dim3(256), 0, st, P, L,
			   means2D, offsets, radii, perm, gx, gy, keys32, vals);
}

#ifdef SGS_WITH_EXPERIMENTS
size_t sort32_temp_bytes(size_t L, int end_bit)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr,
					(uint32_t*)nullptr, (uint32_t*)nullptr, L, 0u, (unsigned)end_bit,
					(hipStream_t)0);
	return bytes;
}

hipError_t launch_sort32_pairs(hipStream_t st, void* temp, size_t temp_bytes, uint32_t* keys_in,
			       uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t L,
			       int end_bit)
{
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, L, 0u,
					 (unsigned)end_bit, st);
}
#endif

__global__ __launch_bounds__(256) void tile_ranges32_kernel(uint32_t L,
							     const uint32_t* __restrict__ tiles,
							     uint2* __restrict__ ranges)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= L) return;
	const uint32_t cur = tiles[i];
	if (i == 0) ranges[cur].x = 0;
	else {
		const uint32_t prev = tiles[i - 1];
		if (cur != prev) {
			ranges[prev].y = i;
			ranges[cur].x = i;
		}
	}
	if (i == L - 1) ranges[cur].y = L;
}

void launch_tile_ranges32(hipStream_t st, size_t L, const uint32_t* tiles, uint2* ranges, int ntiles)
{
	(void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)ntiles, st);
	if (L == 0) return;
	hipLaunchKernelGGL(tile_ranges32_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st,
			   (uint32_t)L, tiles, ranges);
}

} // namespace sgs
