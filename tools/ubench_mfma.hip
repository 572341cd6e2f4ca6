// ubench_mfma.hip -- ceiling of the blend accumulate's MFMA pattern on MI355X:
// 8 independent v_mfma_f32_32x32x2_f32 accumulators per wave, operands from LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: register operands only  1: operands from LDS (ds_read_b32)  2: + select per operand
__global__ __launch_bounds__(256, 3) void k(float* out, int iters, int n)
{
	__shared__ float s_w[16 * 256 + 16 * 128];
	for (int i = threadIdx.x; i < 16 * 256 + 16 * 128; i += 256) s_w[i] = 1e-3f * i;
	__syncthreads();
	const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31, wave = threadIdx.x >> 6;
	f32x16 acc[8];
	for (int nb = 0; nb < 8; nb++) for (int r = 0; r < 16; r++) acc[nb][r] = 0.f;
	float a = lane * 1e-3f, b = 1.0f + lane * 1e-4f;
	for (int it = 0; it < iters; it++) {
		for (int e = 0; e < 16; e += 2) {
			const float* wr = s_w + (e + half) * 256 + l31;
			if (MODE >= 1) a = s_w[16 * 256 + (e + half) * 128 + wave * 32 + l31];
			const bool live = (e + half) < n;
#pragma unroll
			for (int nb = 0; nb < 8; nb++) {
				float bw = (MODE >= 1) ? wr[nb * 32] : b;
				if (MODE >= 2) bw = live ? bw : 0.f;
				acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw, acc[nb], 0, 0, 0);
			}
		}
	}
	float r = 0;
	for (int nb = 0; nb < 8; nb++) for (int q = 0; q < 16; q++) r += acc[nb][q];
	out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main()
{
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int CUs = prop.multiProcessorCount;
	float* out; CK(hipMalloc(&out, 4 * 256 * 8192));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int wgs = 1; wgs <= 3; wgs++)
		for (int mode = 0; mode < 3; mode++) {
			const int blocks = CUs * wgs, iters = 400;
			float best = 1e9;
			for (int rep = 0; rep < 4; rep++) {
				CK(hipEventRecord(e0));
				if (mode == 0) k<0><<<blocks, 256>>>(out, iters, 15);
				else if (mode == 1) k<1><<<blocks, 256>>>(out, iters, 15);
				else k<2><<<blocks, 256>>>(out, iters, 15);
				CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			const double mfmas = (double)blocks * 4 * iters * 8 * 8;
			printf("WGs/CU %d mode %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD @2.4GHz)\n", wgs, mode, best,
			       mfmas * 4096 / best / 1e9, best * 1e-3 * 2.4e9 / (mfmas / (CUs * 4)));
		}
	return 0;
}
