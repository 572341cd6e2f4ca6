// ubench_ldsadd.hip -- rate of LDS reduction primitives on MI355X (round 5: the fused backward sums eight waves' partial
// dL/dF tiles in LDS).  512-thread workgroups, one per CU; every wave issues `iters` x 16 operations of 64 lanes x 4 B on
// conflict-free addresses (row pitch 40 floats, the two lane halves 4 rows apart).
//   mode 0: ds_add_f32 (no return)   1: ds_add_u32   2: ds_write_b32   3: ds_read_b32 + v_add + ds_write_b32 (wave-private rows)
//   4: ds_add_f32, all 8 waves on the SAME tile (the fused kernel's pattern)   5: ds_write_b128   6: ds_pk_add_f16?  (skipped)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters)
{
	__shared__ float s[8 * 128 * 10 + 128 * 40];   // 8 x 5 KB private + one shared 20 KB tile
	for (int i = threadIdx.x; i < 8 * 128 * 10 + 128 * 40; i += 512) s[i] = 0.f;
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
	float* tile = s + 8 * 128 * 10;
	float* priv = s + wave * 1280;
	float v = 1e-3f * lane;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
			float* p = (MODE == 4 ? tile + ((it & 3) * 32 + row) * 40 : priv + row * 40) + l31;
			if (MODE == 0 || MODE == 4) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else if (MODE == 1) (void)__hip_atomic_fetch_add((unsigned*)p, (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else if (MODE == 2) *(volatile float*)p = v;
			else if (MODE == 3) { const float o = *(volatile float*)p; *(volatile float*)p = o + v; }
			else if (MODE == 5) { *(float4*)(priv + (lane * 4) + (r & 3) * 256) = make_float4(v, v, v, v); asm volatile("" ::: "memory"); }
		}
	}
	__syncthreads();
	out[blockIdx.x * 512 + threadIdx.x] = s[threadIdx.x] + tile[threadIdx.x];
}

int main()
{
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int CUs = prop.multiProcessorCount;
	float* out; CK(hipMalloc(&out, 4 * 512 * 1024));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int iters = 2000;
	for (int mode = 0; mode < 6; mode++) {
		float best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			CK(hipEventRecord(e0));
			switch (mode) {
			case 0: k<0><<<CUs, 512>>>(out, iters); break;
			case 1: k<1><<<CUs, 512>>>(out, iters); break;
			case 2: k<2><<<CUs, 512>>>(out, iters); break;
			case 3: k<3><<<CUs, 512>>>(out, iters); break;
			case 4: k<4><<<CUs, 512>>>(out, iters); break;
			case 5: k<5><<<CUs, 512>>>(out, iters); break;
			}
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		const double ops = (double)iters * 16 * 8;   // wave instructions per CU
		printf("mode %d: %.3f ms, %.1f cycles per wave instruction per CU @2.1GHz\n", mode, best, best * 1e-3 * 2.1e9 / ops);
	}
	return 0;
}
