// repro_x16.hip -- standalone (no torch, no library) reproducer for round 2's finding: with the sweep
// kernel's matrix work issued as the gfx950 double-rate v_mfma_f32_32x32x16_bf16, waves of OTHER kernels resident on the
// same CUs sporadically received a wrong 256-byte beat of a global_load_dwordx4.
//
//   victim    : many small workgroups (64 threads, ~20 VGPRs, no LDS) streaming a read-only buffer with
//               global_load_dwordx4 and checking every word against the generating formula.
//   aggressor : the sweep's skeleton -- an LDS ring filled by global_load_lds_dwordx4 (4 KB per stage and wave), operands by
//               ds_read_b128, MFMAs into 128 accumulator registers, nontemporal dword stores now and then -- with the
//               matrix instruction, the DMA, the stores and the register footprint selectable.
//   The two run on two streams for a fixed time per configuration; the program prints the corrupted words per config.
//
// build: hipcc --offload-arch=gfx950 -O3 -o repro_x16 repro_x16.hip      run: ./repro_x16 [seconds per config]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint32_t pattern(uint32_t i) { return i * 2654435761u ^ (i >> 7) ^ 0x9E3779B9u; }

__global__ void fill_kernel(uint32_t* p, size_t n) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pattern((uint32_t)i);
}

// bad[0] = mismatching words, bad[1..8] = first few (index, got) pairs
__global__ __launch_bounds__(64) void victim_kernel(const u32x4* __restrict__ src, uint32_t n4, int iters, uint32_t salt, uint32_t* bad)
{
	uint32_t idx = (blockIdx.x * 64u + threadIdx.x + salt * 7919u) % n4;
	uint32_t nbad = 0;
	for (int it = 0; it < iters; it++) {
		const u32x4 v = src[idx];   // global_load_dwordx4
		const uint32_t b = idx * 4u;
		const bool ok = v.x == pattern(b) && v.y == pattern(b + 1) && v.z == pattern(b + 2) && v.w == pattern(b + 3);
		if (!ok) {
			nbad++;
			const uint32_t slot = atomicAdd(&bad[1], 1u);
			if (slot < 16) { bad[4 + 4 * slot] = idx; bad[5 + 4 * slot] = v.x; bad[6 + 4 * slot] = pattern(b); bad[7 + 4 * slot] = (uint32_t)it | (blockIdx.x << 12); }
		}
		idx += 64u * 1021u;
		if (idx >= n4) idx -= n4;
	}
	if (nbad) atomicAdd(&bad[0], nbad);
}

// MODE bits: 1 = LDS-DMA ring, 2 = x16 MFMA (else x8 pairs if bit 2), 4 = x8 MFMA, 8 = nt stores, 16 = claim all 256 registers
template <int MODE>
__global__ __launch_bounds__(256, 2) void aggressor_kernel(const float* __restrict__ src, size_t nfloat, float* __restrict__ sink, int iters)
{
	__shared__ float4 ring[4 * 16384 / 16];   // 4 stages x 16 KB
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ring;
	for (int i = threadIdx.x; i < 4 * 16384 / 16; i += 256) ring[i] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
	__syncthreads();
	f32x16 acc[8];
#pragma unroll
	for (int b = 0; b < 8; b++)
#pragma unroll
		for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
	size_t off = ((size_t)blockIdx.x * 65536 + (size_t)wave * 4096) % (nfloat - 65536 * 4);
	for (int it = 0; it < iters; it++) {
		const uint32_t st = lds0 + (uint32_t)(it & 3) * 16384u;
		if (MODE & 1) {
#pragma unroll
			for (int j = 0; j < 4; j++)
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + (size_t)j * 256 + lane * 4),
								 (__attribute__((address_space(3))) void*)(size_t)(st + (uint32_t)(wave * 4 + j) * 1024u), 16, 0, 0);
			off += 65536 * 2;
			if (off >= nfloat - 65536 * 4) off -= nfloat - 65536 * 4;
			__builtin_amdgcn_s_waitcnt((8) | (7 << 4) | (15 << 8));   // vmcnt(8): two bundles in flight
		}
		// operands from the stage filled two iterations ago
		const uint32_t rd = lds0 + (uint32_t)((it + 2) & 3) * 16384u + (uint32_t)lane * 16u;
		u32x4 a, b0, b1;
		asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
			     : "=&v"(a), "=&v"(b0), "=&v"(b1) : "v"(rd) : "memory");
		if (MODE & 2) {
#pragma unroll
			for (int b = 0; b < 8; b++) {
				acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b0), acc[b], 0, 0, 0);
				acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b1), __builtin_bit_cast(bf16x8, a), acc[b], 0, 0, 0);
			}
		} else if (MODE & 4) {
			const s16x4 a0 = __builtin_bit_cast(s16x4, __builtin_shufflevector(a, a, 0, 1)), a1 = __builtin_bit_cast(s16x4, __builtin_shufflevector(a, a, 2, 3));
			const s16x4 c0 = __builtin_bit_cast(s16x4, __builtin_shufflevector(b0, b0, 0, 1)), c1 = __builtin_bit_cast(s16x4, __builtin_shufflevector(b0, b0, 2, 3));
#pragma unroll
			for (int b = 0; b < 8; b++) {
				acc[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a0, c0, acc[b], 0, 0, 0);
				acc[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a1, c1, acc[b], 0, 0, 0);
				acc[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(c0, a1, acc[b], 0, 0, 0);
				acc[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(c1, a0, acc[b], 0, 0, 0);
			}
		}
		if ((MODE & 8) && (it & 15) == 15) {
			float* dst = sink + ((size_t)blockIdx.x * 256 + threadIdx.x) + (size_t)(it >> 4 & 63) * 16 * 1048576;
#pragma unroll
			for (int b = 0; b < 8; b++)
#pragma unroll
				for (int r = 0; r < 16; r++) __builtin_nontemporal_store(acc[b][r], dst + (size_t)(b * 16 + r) * 1048576 / 8);
		}
		__builtin_amdgcn_s_barrier();
	}
	__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
	float s = 0.f;
#pragma unroll
	for (int b = 0; b < 8; b++) s += acc[b][0] + acc[b][7];
	if (s == 123.456f) sink[0] = s;
	if (MODE & 16) asm volatile("" : : : "a151");   // accumulators move to a[0:127]; claiming up to a151 makes VGPR (104) + AGPR (152) = 256 = half a SIMD's file: no other wave fits beside two of these
}

struct Config { const char* name; int mode; };

template <int MODE> void launch_aggr(hipStream_t st, const float* src, size_t nf, float* sink, int iters) {
	hipLaunchKernelGGL(aggressor_kernel<MODE>, dim3(512), dim3(256), 0, st, src, nf, sink, iters);
}

int main(int argc, char** argv)
{
	const double secs = argc > 1 ? atof(argv[1]) : 10.0;
	const size_t NW = (size_t)1 << 28;   // victim buffer: 1 GiB of words
	uint32_t* vsrc; float* asrc; float* sink; uint32_t* bad;
	CK(hipMalloc(&vsrc, NW * 4));
	CK(hipMalloc(&asrc, NW * 4));
	CK(hipMalloc(&sink, (size_t)64 * 16 * 1048576 * 4 + 4096));
	CK(hipMalloc(&bad, 4096));
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, vsrc, NW);
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)asrc, NW);
	CK(hipDeviceSynchronize());
	hipStream_t sv, sa;
	CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
	const Config cfgs[] = {
		{"no aggressor", 0},
		{"ring + x8 MFMA + stores (what ships)", 1 | 4 | 8},
		{"ring + x16 MFMA + stores (round 2's form)", 1 | 2 | 8},
		{"ring + x16 MFMA", 1 | 2},
		{"x16 MFMA only (no DMA, no stores)", 2},
		{"ring only", 1},
		{"ring + x16 MFMA + stores, all 256 registers claimed", 1 | 2 | 8 | 16},
	};
	printf("%-56s %10s %12s %14s\n", "aggressor", "victims", "aggr launches", "corrupt words");
	for (const Config& c : cfgs) {
		CK(hipMemset(bad, 0, 4096));
		const auto t0 = std::chrono::steady_clock::now();
		long nv = 0, na = 0;
		uint32_t salt = 0;
		while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
			if (c.mode) {
				switch (c.mode) {
				case 1 | 4 | 8: launch_aggr<1 | 4 | 8>(sa, asrc, NW, sink, 2000); break;
				case 1 | 2 | 8: launch_aggr<1 | 2 | 8>(sa, asrc, NW, sink, 2000); break;
				case 1 | 2: launch_aggr<1 | 2>(sa, asrc, NW, sink, 2000); break;
				case 2: launch_aggr<2>(sa, asrc, NW, sink, 2000); break;
				case 1: launch_aggr<1>(sa, asrc, NW, sink, 2000); break;
				case 1 | 2 | 8 | 16: launch_aggr<1 | 2 | 8 | 16>(sa, asrc, NW, sink, 2000); break;
				}
				na++;
			}
			for (int k = 0; k < 8; k++) {
				hipLaunchKernelGGL(victim_kernel, dim3(16384), dim3(64), 0, sv, (const u32x4*)vsrc, (uint32_t)(NW / 4), 64, salt++, bad);
				nv++;
			}
			CK(hipStreamSynchronize(sv));
			if ((na & 7) == 0) CK(hipStreamSynchronize(sa));
		}
		CK(hipDeviceSynchronize());
		uint32_t h[72];
		CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
		printf("%-56s %10ld %12ld %14u\n", c.name, nv, na, h[0]);
		for (uint32_t i = 0; i < h[1] && i < 6; i++)
			printf("      word %u: got %08x want %08x (iter %u, block %u)\n", h[4 + 4 * i] * 4, h[5 + 4 * i], h[6 + 4 * i], h[7 + 4 * i] & 4095u, h[7 + 4 * i] >> 12);
		fflush(stdout);
	}
	return 0;
}
