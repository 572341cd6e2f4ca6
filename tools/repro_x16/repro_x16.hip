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

// A victim with no memory traffic inside its loop: the same 8-register mixing chain evaluated twice in disjoint registers
// (same operations, the second copy one instruction behind), compared at the end.  A VGPR write that is lost or damaged
// for some lanes -- by whatever shares the SIMD -- makes the two copies disagree.
__global__ __launch_bounds__(64) void victim_valu_kernel(int iters, uint32_t salt, uint32_t* bad)
{
	const uint32_t t = blockIdx.x * 64u + threadIdx.x + salt * 977u;
	float a[8], b[8];
#pragma unroll
	for (int i = 0; i < 8; i++) { a[i] = (float)((t * (2 * i + 3)) & 1023u) * 0.001f + 0.5f; b[i] = a[i]; }
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			a[i] = __builtin_fmaf(a[i], 0.999f, a[(i + 3) & 7] * 0.0007f);
			asm volatile("" : "+v"(a[i]));
			b[i] = __builtin_fmaf(b[i], 0.999f, b[(i + 3) & 7] * 0.0007f);
			asm volatile("" : "+v"(b[i]));
		}
	}
	uint32_t nbad = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) nbad += (__float_as_uint(a[i]) != __float_as_uint(b[i])) ? 1u : 0u;
	if (nbad) {
		atomicAdd(&bad[2], nbad);
		const uint32_t slot = atomicAdd(&bad[3], 1u);
		if (slot < 8) { bad[80 + 4 * slot] = threadIdx.x; bad[81 + 4 * slot] = __float_as_uint(a[0]); bad[82 + 4 * slot] = __float_as_uint(b[0]); bad[83 + 4 * slot] = blockIdx.x; }
	}
}

// The same with transcendental instructions (v_sqrt_f32 / v_rcp_f32 / v_exp_f32 run on the quarter-rate unit, 16 lanes
// per pass): the library's victim (its per-Gaussian preprocess) came out with radii that fit a wrong square root in one
// 16-lane row.
__global__ __launch_bounds__(64) void victim_trans_kernel(int iters, uint32_t salt, uint32_t* bad)
{
	const uint32_t t = blockIdx.x * 64u + threadIdx.x + salt * 977u;
	float a[4], b[4];
#pragma unroll
	for (int i = 0; i < 4; i++) { a[i] = (float)((t * (2 * i + 3)) & 1023u) * 0.001f + 0.5f; b[i] = a[i]; }
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 4; i++) {
			a[i] = __builtin_sqrtf(__builtin_fmaf(a[i], a[(i + 1) & 3], 0.37f)) + __builtin_amdgcn_rcpf(a[(i + 2) & 3] + 1.5f) + __builtin_amdgcn_exp2f(-a[(i + 3) & 3]) * 0.25f;
			asm volatile("" : "+v"(a[i]));
			b[i] = __builtin_sqrtf(__builtin_fmaf(b[i], b[(i + 1) & 3], 0.37f)) + __builtin_amdgcn_rcpf(b[(i + 2) & 3] + 1.5f) + __builtin_amdgcn_exp2f(-b[(i + 3) & 3]) * 0.25f;
			asm volatile("" : "+v"(b[i]));
		}
	}
	uint32_t nbad = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) nbad += (__float_as_uint(a[i]) != __float_as_uint(b[i])) ? 1u : 0u;
	if (nbad) {
		atomicAdd(&bad[112], nbad);
		const uint32_t slot = atomicAdd(&bad[113], 1u);
		if (slot < 3) { bad[116 + 4 * slot] = threadIdx.x; bad[117 + 4 * slot] = __float_as_uint(a[0]); bad[118 + 4 * slot] = __float_as_uint(b[0]); bad[119 + 4 * slot] = blockIdx.x; }
	}
}

// The same with PACKED fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32): the library's damaged quantity
// (the first component of the 3-D covariance) is the low half of a packed add in the victim kernel's code.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void victim_pk_kernel(int iters, uint32_t salt, uint32_t* bad)
{
	const uint32_t t = blockIdx.x * 64u + threadIdx.x + salt * 977u;
	f32x2 a[4], b[4];
#pragma unroll
	for (int i = 0; i < 4; i++) { a[i] = f32x2{(float)((t * (2 * i + 3)) & 1023u) * 0.001f + 0.5f, (float)((t * (2 * i + 5)) & 511u) * 0.002f + 0.25f}; b[i] = a[i]; }
	const f32x2 k0 = {0.999f, 0.998f}, k1 = {0.0007f, 0.0011f};
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 4; i++) {
			a[i] = __builtin_elementwise_fma(a[i], k0, a[(i + 1) & 3] * k1) + a[(i + 2) & 3].yx * k1;
			asm volatile("" : "+v"(a[i]));
			b[i] = __builtin_elementwise_fma(b[i], k0, b[(i + 1) & 3] * k1) + b[(i + 2) & 3].yx * k1;
			asm volatile("" : "+v"(b[i]));
		}
	}
	uint32_t nbad = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) nbad += (__float_as_uint(a[i].x) != __float_as_uint(b[i].x) ? 1u : 0u) + (__float_as_uint(a[i].y) != __float_as_uint(b[i].y) ? 1u : 0u);
	if (nbad) {
		atomicAdd(&bad[96], nbad);
		const uint32_t slot = atomicAdd(&bad[97], 1u);
		if (slot < 3) { bad[100 + 4 * slot] = threadIdx.x; bad[101 + 4 * slot] = __float_as_uint(a[0].x); bad[102 + 4 * slot] = __float_as_uint(b[0].x); bad[103 + 4 * slot] = blockIdx.x; }
	}
}

// Round 6: a victim that is NOTHING but packed-fp32 chains, each checked against the SAME arithmetic in scalar
// instructions (v_pk_fma_f32 and v_fma_f32 round identically per component), at the register footprint of the library's victim
// (preprocess: 96 VGPRs -- two such waves fit beside one aggressor wave on a SIMD), 256 threads per workgroup.
__global__ __launch_bounds__(256) void victim_pk_vs_scalar_kernel(int iters, uint32_t salt, uint32_t* bad)
{
	asm volatile("" : : : "v95");   // 96 registers per wave, like the library's victim
	const uint32_t t = blockIdx.x * 256u + threadIdx.x + salt * 977u;
	f32x2 a[6];
	float sx[6], sy[6];
#pragma unroll
	for (int i = 0; i < 6; i++) {
		a[i] = f32x2{(float)((t * (2 * i + 3)) & 1023u) * 0.001f + 0.5f, (float)((t * (2 * i + 5)) & 511u) * 0.002f + 0.25f};
		sx[i] = a[i].x;
		sy[i] = a[i].y;
	}
	const f32x2 k0 = {0.999f, 0.998f}, k1 = {0.0007f, 0.0011f};
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 6; i++) {
			const int j = (i + 1) % 6, l = (i + 2) % 6;
			// packed: a_i = fma(a_i, k0, a_j * k1) + a_l * k1   (v_pk_mul, v_pk_fma, v_pk_mul, v_pk_add)
			const f32x2 m1 = a[j] * k1;
			const f32x2 f1 = __builtin_elementwise_fma(a[i], k0, m1);
			const f32x2 m2 = a[l] * k1;
			f32x2 r = f1 + m2;
			asm volatile("" : "+v"(r));
			// scalar twin, component by component
			float rx = __builtin_fmaf(sx[i], 0.999f, sx[j] * 0.0007f) + sx[l] * 0.0007f;
			float ry = __builtin_fmaf(sy[i], 0.998f, sy[j] * 0.0011f) + sy[l] * 0.0011f;
			asm volatile("" : "+v"(rx), "+v"(ry));
			a[i] = r;
			sx[i] = rx;
			sy[i] = ry;
		}
	}
	uint32_t nbad = 0;
#pragma unroll
	for (int i = 0; i < 6; i++) nbad += (__float_as_uint(a[i].x) != __float_as_uint(sx[i]) ? 1u : 0u) + (__float_as_uint(a[i].y) != __float_as_uint(sy[i]) ? 1u : 0u);
	if (nbad) {
		atomicAdd(&bad[126], nbad);
		const uint32_t slot = atomicAdd(&bad[127], 1u);
		if (slot < 1) { bad[125] = threadIdx.x | (blockIdx.x << 8); }
	}
}

// The library's real victim arithmetic: the 3-D covariance of a Gaussian from its scale and quaternion
// (semantic-gaussians_amd/csrc/sgs_device.h, cov3d_from_scale_rot), which hipcc's SLP vectoriser turns into a chain of
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 with op_sel and neg modifiers.  Results are written out and compared, after
// the aggressor has stopped, with a second evaluation by the same kernel.
__device__ __forceinline__ void cov3d(float sx, float sy, float sz, float r, float x, float y, float z, float* cov)
{
	float R[3][3], M[3][3];
	R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
	R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
	R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
	const float s[3] = {sx, sy, sz};
#pragma unroll
	for (int k = 0; k < 3; k++)
#pragma unroll
		for (int i = 0; i < 3; i++) M[k][i] = s[k] * R[i][k];
#define SIG(i, j) (M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j])
	cov[0] = SIG(0, 0); cov[1] = SIG(1, 0); cov[2] = SIG(2, 0); cov[3] = SIG(1, 1); cov[4] = SIG(2, 1); cov[5] = SIG(2, 2);
#undef SIG
}
__global__ __launch_bounds__(256) void victim_cov_kernel(int n, const float* __restrict__ scales, const float* __restrict__ rots, float* __restrict__ out)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	float c[6];
	cov3d(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2], rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3], c);
#pragma unroll
	for (int k = 0; k < 6; k++) out[6 * (size_t)i + k] = c[k];
}
__global__ void compare_kernel(size_t n, const uint32_t* a, const uint32_t* b, uint32_t* bad)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		if (a[i] != b[i]) {
			atomicAdd(&bad[120], 1u);
			if (atomicAdd(&bad[121], 1u) < 2) { bad[122] = (uint32_t)i; bad[123] = a[i]; bad[124] = b[i]; }
		}
}

// MODE bits: 1 = LDS-DMA ring, 2 = x16 MFMA (else x8 pairs if bit 2), 4 = x8 MFMA, 8 = nt stores, 16 = claim all 256 registers
template <int MODE>
__global__ __launch_bounds__(256, 2) void aggressor_kernel(const float* __restrict__ src, size_t nfloat, float* __restrict__ sink, int iters)
{
	__shared__ float4 ring[((MODE & 128) ? 96 * 1024 : 4 * 16384) / 16];   // 4 stages x 16 KB (+ padding to 96 KB with bit 128: one workgroup per CU, i.e. ONE aggressor wave per SIMD)
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
		if (MODE & 32) {   // the shipped six-product stream: 24 back-to-back x16 MFMAs on two accumulator blocks in a[0:31]
			asm volatile(
				"s_nop 1\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %2, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %1, %2, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %1, %0, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %2, %0, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %2, %1, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %2, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %1, %2, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %1, %0, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %2, %0, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %2, %1, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %2, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %1, %2, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %1, %0, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %2, %0, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %2, %1, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %2, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %1, %2, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %1, %0, a[16:31]\n\t"
				"v_mfma_f32_32x32x16_bf16 a[0:15], %2, %0, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %2, %1, a[16:31]"
				: : "v"(a), "v"(b0), "v"(b1) : "a127", "memory");
		} else if (MODE & 64) {   // the same stream on the legacy x8 instruction (48 MFMAs, same duration)
			const uint64_t a0 = ((uint64_t)a.y << 32) | a.x, c0 = ((uint64_t)b0.y << 32) | b0.x, d0 = ((uint64_t)b1.y << 32) | b1.x;
#pragma unroll
			for (int r = 0; r < 8; r++)
				asm volatile(
					"v_mfma_f32_32x32x8_bf16 a[0:15], %0, %1, a[0:15]\n\tv_mfma_f32_32x32x8_bf16 a[16:31], %0, %2, a[16:31]\n\t"
					"v_mfma_f32_32x32x8_bf16 a[0:15], %1, %2, a[0:15]\n\tv_mfma_f32_32x32x8_bf16 a[16:31], %1, %0, a[16:31]\n\t"
					"v_mfma_f32_32x32x8_bf16 a[0:15], %2, %0, a[0:15]\n\tv_mfma_f32_32x32x8_bf16 a[16:31], %2, %1, a[16:31]"
					: : "v"(a0), "v"(c0), "v"(d0) : "a127", "memory");
		} else if (MODE & 2) {
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
	const int NCOV = 1 << 22;
	float *cscale, *crot, *cout, *cref;
	CK(hipMalloc(&cscale, (size_t)NCOV * 12)); CK(hipMalloc(&crot, (size_t)NCOV * 16));
	CK(hipMalloc(&cout, (size_t)NCOV * 24)); CK(hipMalloc(&cref, (size_t)NCOV * 24));
	{   // scales ~ 0.01 .. 0.03, quaternions of unit-ish norm, from the word pattern
		std::vector<float> hs((size_t)NCOV * 3), hr((size_t)NCOV * 4);
		for (size_t i = 0; i < hs.size(); i++) hs[i] = 0.01f + 0.02f * (float)(pattern((uint32_t)i) & 0xffff) / 65536.0f;
		for (size_t i = 0; i < hr.size(); i++) hr[i] = ((float)(pattern((uint32_t)(i * 7 + 1)) & 0xffff) / 32768.0f - 1.0f) * 0.6f;
		CK(hipMemcpy(cscale, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
		CK(hipMemcpy(crot, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
	}
	hipLaunchKernelGGL(victim_cov_kernel, dim3(NCOV / 256), dim3(256), 0, 0, NCOV, cscale, crot, cref);
	CK(hipDeviceSynchronize());
	hipStream_t sv, sa;
	const int ncus = argc > 2 ? atoi(argv[2]) : 0;   // > 0: BOTH streams confined to the same `ncus` compute units (victim and aggressor always share CUs)
	if (ncus > 0) {
		hipDeviceProp_t pr;
		CK(hipGetDeviceProperties(&pr, 0));
		std::vector<uint32_t> mask((pr.multiProcessorCount + 31) / 32, 0u);
		for (int b = 0; b < ncus && b < pr.multiProcessorCount; b++) mask[b >> 5] |= 1u << (b & 31);
		CK(hipExtStreamCreateWithCUMask(&sv, (uint32_t)mask.size(), mask.data()));
		CK(hipExtStreamCreateWithCUMask(&sa, (uint32_t)mask.size(), mask.data()));
		printf("both streams confined to compute units [0, %d) of %d\n", ncus, pr.multiProcessorCount);
	} else {
		CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
		CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
	}
	const bool r6 = argc > 3;   // round 6's short list: the bisect's shape (one dense-x16 wave per SIMD, foreign packed-fp32 waves beside it) + controls
	const Config cfgs6[] = {
		{"no aggressor", 0},
		{"ring + 24 dense x16 MFMAs per step, ONE workgroup per CU", 1 | 32 | 128},
		{"ring + 48 dense x8 MFMAs per step, ONE workgroup per CU (control)", 1 | 64 | 128},
		{"ring + 24 dense x16 MFMAs per step + stores, two workgroups per CU", 1 | 32 | 8},
		{"24 dense x16 MFMAs per step, no memory, ONE workgroup per CU", 32 | 128},
	};
	const Config cfgs5[] = {
		{"no aggressor", 0},
		{"ring + x8 MFMA + stores (what ships)", 1 | 4 | 8},
		{"ring + x16 MFMA + stores (round 2's form)", 1 | 2 | 8},
		{"ring + x16 MFMA", 1 | 2},
		{"x16 MFMA only (no DMA, no stores)", 2},
		{"ring only", 1},
		{"ring + x16 MFMA + stores, all 256 registers claimed", 1 | 2 | 8 | 16},
		{"24 dense x16 MFMAs per step on AGPRs, no memory at all", 32},
		{"ring + 24 dense x16 MFMAs per step on AGPRs", 1 | 32},
		{"ring + 48 dense x8 MFMAs per step on AGPRs", 1 | 64},
	};
	printf("%-56s %10s %12s %s\n", "aggressor", "victims", "aggr launches", "corrupt words (load victim / valu victim)");
	const Config* cfgs = r6 ? cfgs6 : cfgs5;
	const int ncfg = r6 ? (int)(sizeof(cfgs6) / sizeof(Config)) : (int)(sizeof(cfgs5) / sizeof(Config));
	for (int ci = 0; ci < ncfg; ci++) {
		const Config& c = cfgs[ci];
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
				case 32: launch_aggr<32>(sa, asrc, NW, sink, 2000); break;
				case 1 | 32: launch_aggr<1 | 32>(sa, asrc, NW, sink, 2000); break;
				case 1 | 64: launch_aggr<1 | 64>(sa, asrc, NW, sink, 2000); break;
				case 1 | 32 | 128: launch_aggr<1 | 32 | 128>(sa, asrc, NW, sink, 2000); break;
				case 1 | 64 | 128: launch_aggr<1 | 64 | 128>(sa, asrc, NW, sink, 2000); break;
				case 1 | 32 | 8: launch_aggr<1 | 32 | 8>(sa, asrc, NW, sink, 2000); break;
				case 32 | 128: launch_aggr<32 | 128>(sa, asrc, NW, sink, 2000); break;
				}
				na++;
			}
			for (int k = 0; k < 8; k++) {
				hipLaunchKernelGGL(victim_kernel, dim3(16384), dim3(64), 0, sv, (const u32x4*)vsrc, (uint32_t)(NW / 4), 64, salt++, bad);
				hipLaunchKernelGGL(victim_valu_kernel, dim3(16384), dim3(64), 0, sv, 256, salt++, bad);
				hipLaunchKernelGGL(victim_trans_kernel, dim3(16384), dim3(64), 0, sv, 96, salt++, bad);
				hipLaunchKernelGGL(victim_pk_kernel, dim3(16384), dim3(64), 0, sv, 256, salt++, bad);
				hipLaunchKernelGGL(victim_pk_vs_scalar_kernel, dim3(4096), dim3(256), 0, sv, 256, salt++, bad);
				hipLaunchKernelGGL(victim_cov_kernel, dim3(NCOV / 256), dim3(256), 0, sv, NCOV, cscale, crot, cout);
				hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, sv, (size_t)NCOV * 6, (const uint32_t*)cout, (const uint32_t*)cref, bad);
				nv++;
			}
			CK(hipStreamSynchronize(sv));
			if ((na & 7) == 0) CK(hipStreamSynchronize(sa));
		}
		CK(hipDeviceSynchronize());
		uint32_t h[128];
		CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
		printf("%-70s %8ld %8ld %6u loads %6u valu %6u trans %6u packed %6u packed-vs-scalar %6u cov3d\n", c.name, nv, na, h[0], h[2], h[112], h[96], h[126], h[120]);
		if (h[120]) printf("      cov3d word %u: got %08x want %08x\n", h[122], h[123], h[124]);
		for (uint32_t i = 0; i < h[97] && i < 3; i++)
			printf("      packed chain: lane %u block %u a0 %08x b0 %08x\n", h[100 + 4 * i], h[103 + 4 * i], h[101 + 4 * i], h[102 + 4 * i]);
		for (uint32_t i = 0; i < h[113] && i < 3; i++)
			printf("      trans chain: lane %u block %u a0 %08x b0 %08x\n", h[116 + 4 * i], h[119 + 4 * i], h[117 + 4 * i], h[118 + 4 * i]);
		for (uint32_t i = 0; i < h[3] && i < 4; i++)
			printf("      valu chain: lane %u block %u a0 %08x b0 %08x\n", h[80 + 4 * i], h[83 + 4 * i], h[81 + 4 * i], h[82 + 4 * i]);
		for (uint32_t i = 0; i < h[1] && i < 6; i++)
			printf("      word %u: got %08x want %08x (iter %u, block %u)\n", h[4 + 4 * i] * 4, h[5 + 4 * i], h[6 + 4 * i], h[7 + 4 * i] & 4095u, h[7 + 4 * i] >> 12);
		fflush(stdout);
	}
	return 0;
}
