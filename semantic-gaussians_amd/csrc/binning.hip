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
// MI355X notes.  The reference emits keys with one thread per Gaussian looping over its
// rect: lanes of a wave then write 64 unrelated 12-byte records per iteration and the
// wave runs as long as its largest Gaussian.  Here emission is instance-parallel: one lane
// per OUTPUT slot, which finds its Gaussian by a binary search over the (L2-resident)
// offsets array, so every store is a fully coalesced 512-B/256-B wave store and the work
// is perfectly balanced.  The plain scan and radix sort are rocPRIM library calls on the
// caller's stream (the reference uses CUB for the same two steps).
#include "sgs_kernels.h"
#include <cstring>   // rocPRIM's texture iterator calls host memset
#include <rocprim/rocprim.hpp>

namespace sgs {

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

// One lane per emitted (tile, Gaussian) instance.
__global__ __launch_bounds__(256) void duplicate_with_keys_kernel(
	int P, uint32_t L, const float2* __restrict__ means2D, const float* __restrict__ depths,
	const uint32_t* __restrict__ offsets, const int* __restrict__ radii, int gx, int gy,
	uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
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
	const int g = lo;
	const uint32_t base = (g == 0) ? 0u : offsets[g - 1];
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
				uint64_t* keys, uint32_t* vals, uint32_t L)
{
	if (L == 0) return;
	hipLaunchKernelGGL(duplicate_with_keys_kernel, dim3((L + 255u) / 256u), dim3(256), 0, st, P,
			   L, means2D, depths, offsets, radii, gx, gy, keys, vals);
}

size_t sort_temp_bytes(size_t L, int end_bit)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr,
					(uint32_t*)nullptr, (uint32_t*)nullptr, L, 0u,
					(unsigned)end_bit, (hipStream_t)0);
	return bytes;
}

hipError_t launch_sort_pairs(hipStream_t st, void* temp, size_t temp_bytes, uint64_t* keys_in,
			     uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t L,
			     int end_bit)
{
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, L, 0u,
					 (unsigned)end_bit, st);
}

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

} // namespace sgs
