// calib_counters.hip -- known-byte-count kernels in the accumulate sweep's access mix, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").
//   k_dma_read    N bytes by global_load_lds_dwordx4 (1 KB per wave-instruction), every byte once           -> reads only
//   k_vec_read    the same bytes by global_load_dwordx4                                                  -> reads only
//   k_nt_write    M bytes as `global_store_dword ... nt`, two complete 128-B lines per wave-instruction     -> writes only
//   k_mix         the sweep's mix: per workgroup 17 KB of LDS-DMA reads for every 10 KB of nt line stores   -> both
//   k_reread      a 64 MiB buffer read 16 times by LDS-DMA (1 GiB requested, 64 MiB unique: L2 / Infinity Cache)
// usage: calib_counters <kernel index 0..4>   (one kernel per process so that a --pmc pass sees only it, 20 launches)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr size_t GiB = (size_t)1 << 30;

__global__ __launch_bounds__(256) void k_dma_read(const float* __restrict__ src, size_t bytes_per_wg, float* sink)
{
	__shared__ float4 ring[4 * 4096 / 16 * 4];   // 4 stages x 16 KB
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ring;
	const char* p = (const char*)src + (size_t)blockIdx.x * bytes_per_wg + (size_t)wave * 4096 + (size_t)lane * 16;
	const int steps = (int)(bytes_per_wg / 16384);
	for (int it = 0; it < steps; it++) {
#pragma unroll
		for (int j = 0; j < 4; j++)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + j * 1024),
							 (__attribute__((address_space(3))) void*)(size_t)(lds0 + (uint32_t)(it & 3) * 16384u + (uint32_t)(wave * 4 + j) * 1024u), 16, 0, 0);
		p += 16384;
		__builtin_amdgcn_s_waitcnt(8 | (7 << 4) | (15 << 8));
	}
	__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
	__syncthreads();
	if (ring[threadIdx.x].x == 123.456f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_vec_read(const float4* __restrict__ src, size_t bytes_per_wg, float* sink)
{
	const float4* p = src + (size_t)blockIdx.x * (bytes_per_wg / 16) + threadIdx.x;
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
	const int steps = (int)(bytes_per_wg / 4096);
	for (int it = 0; it < steps; it++) {
		const float4 v = *p;
		acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
		p += 256;
	}
	if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = 1.f;
}

// each wave-instruction: 64 lanes x 4 B = two complete 128-B lines (lanes 0-31 one line, 32-63 the next), like the sweep
__global__ __launch_bounds__(256) void k_nt_write(float* __restrict__ dst, size_t bytes_per_wg)
{
	float* p = dst + (size_t)blockIdx.x * (bytes_per_wg / 4) + threadIdx.x;
	const int steps = (int)(bytes_per_wg / 1024);
	const float v = (float)threadIdx.x;
	for (int it = 0; it < steps; it++) {
		__builtin_nontemporal_store(v, p);
		p += 256;
	}
}

__global__ __launch_bounds__(256) void k_mix(const float* __restrict__ src, float* __restrict__ dst, size_t rd_per_wg, size_t wr_per_wg, float* sink)
{
	__shared__ float4 ring[4 * 4096 / 16 * 4];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ring;
	const char* p = (const char*)src + (size_t)blockIdx.x * rd_per_wg + (size_t)wave * 4096 + (size_t)lane * 16;
	float* q = dst + (size_t)blockIdx.x * (wr_per_wg / 4) + threadIdx.x;
	const int steps = (int)(rd_per_wg / 16384);
	const int stores_per_step = (int)(wr_per_wg / 1024 / steps);
	for (int it = 0; it < steps; it++) {
#pragma unroll
		for (int j = 0; j < 4; j++)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + j * 1024),
							 (__attribute__((address_space(3))) void*)(size_t)(lds0 + (uint32_t)(it & 3) * 16384u + (uint32_t)(wave * 4 + j) * 1024u), 16, 0, 0);
		p += 16384;
		for (int s = 0; s < stores_per_step; s++) {
			__builtin_nontemporal_store((float)lane, q);
			q += 256;
		}
	}
	__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
	__syncthreads();
	if (ring[threadIdx.x].x == 123.456f) sink[0] = 1.f;
}

int main(int argc, char** argv)
{
	const int which = argc > 1 ? atoi(argv[1]) : 0;
	float *src, *dst, *sink;
	CK(hipMalloc(&src, 2 * GiB));
	CK(hipMalloc(&dst, 3 * GiB));
	CK(hipMalloc(&sink, 256));
	CK(hipMemset(src, 0x3c, 2 * GiB));
	CK(hipMemset(dst, 0, 3 * GiB));
	CK(hipDeviceSynchronize());
	const int WG = 4096;
	double rd = 0, wr = 0;
	for (int rep = 0; rep < 20; rep++) {
		switch (which) {
		case 0: hipLaunchKernelGGL(k_dma_read, dim3(WG), dim3(256), 0, 0, src, GiB / WG, sink); rd = (double)GiB; break;
		case 1: hipLaunchKernelGGL(k_vec_read, dim3(WG), dim3(256), 0, 0, (const float4*)src, GiB / WG, sink); rd = (double)GiB; break;
		case 2: hipLaunchKernelGGL(k_nt_write, dim3(WG), dim3(256), 0, 0, dst, 2 * GiB / WG); wr = 2.0 * GiB; break;
		case 3: hipLaunchKernelGGL(k_mix, dim3(WG), dim3(256), 0, 0, src, dst, (GiB + GiB / 2) / WG, (2 * GiB + GiB / 2) / WG, sink); rd = 1.5 * GiB; wr = 2.5 * GiB; break;
		case 4:   // 16 passes over the first 64 MiB
			for (int pass = 0; pass < 16; pass++) hipLaunchKernelGGL(k_dma_read, dim3(WG), dim3(256), 0, 0, src, (GiB / 16) / WG, sink);
			rd = (double)GiB;
			break;
		}
	}
	CK(hipDeviceSynchronize());
	printf("kernel %d: per launch%s requested read %.0f bytes, written %.0f bytes\n", which, which == 4 ? " group of 16" : "", rd, wr);
	return 0;
}
