// ubench_fma.hip -- MI355X micro-benchmarks that decide the blend kernel's FMA / feed form.
//   A: v_fmac_f32, VGPR operands            B: v_fmac_f32, SGPR multiplicand
//   C: v_pk_fma_f32, VGPR pairs             D: v_pk_fma_f32, SGPR-pair multiplicand
//   E/F: scalar-load-fed FMA stream (s_load_dwordx16 -> v_pk_fma / v_fmac) over a table of
//        rows, L2-resident and HBM-sized
// Build: hipcc --offload-arch=gfx950 -O3 ubench_fma.hip -o ubench_fma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float s0, float s1)
{
	float a[32];
#pragma unroll
	for (int i = 0; i < 32; i++) a[i] = threadIdx.x * 1e-3f + i;
	float w = threadIdx.x * 1e-6f + 1.0f, w2 = w;
	float f0 = s0 + threadIdx.x * 1e-9f, f1 = s1;
	for (int it = 0; it < iters; it++) {
		if (MODE == 0) {
#pragma unroll
			for (int i = 0; i < 32; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(f0), "v"(w));
		} else if (MODE == 1) {
#pragma unroll
			for (int i = 0; i < 32; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(w));
		} else if (MODE == 2) {
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2* p = (f2*)a;
			f2 ff = {f0, f1}, ww = {w, w2};
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(ff), "v"(ww));
		} else if (MODE == 3) {
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2* p = (f2*)a;
			f2 ss = {s0, s1}, ww = {w, w2};
#pragma unroll
			for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(ss), "v"(ww));
		}
	}
	float r = 0;
#pragma unroll
	for (int i = 0; i < 32; i++) r += a[i];
	out[blockIdx.x * 256 + threadIdx.x] = r;
}

// scalar-load-fed stream: each wave walks `n` rows of CC floats chosen by ids[] (uniform per
// block), accumulating acc[c] += row[c] * w.  PK=1 lets the compiler SLP-pack into
// v_pk_fma_f32; PK=0 forces scalar v_fmac through an asm barrier.
template <int CC, int PK>
__global__ __launch_bounds__(256) void sfeed_kernel(const uint32_t* __restrict__ ids, const float* __restrict__ rows,
						     float* __restrict__ out, int n, int C)
{
	float acc[CC];
#pragma unroll
	for (int c = 0; c < CC; c++) acc[c] = 0.f;
	float w = 1.0f + threadIdx.x * 1e-7f;
	const uint32_t* myids = ids + (size_t)blockIdx.x * n;
	for (int j = 0; j < n; j++) {
		const uint32_t id = myids[j];
		const float* __restrict__ f = rows + (size_t)id * C;
		if (PK) {
#pragma unroll
			for (int c = 0; c < CC; c++) acc[c] = __builtin_fmaf(f[c], w, acc[c]);
		} else {
#pragma unroll
			for (int c = 0; c < CC; c++) { float fv = f[c]; asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[c]) : "s"(fv), "v"(w)); }
		}
	}
	float r = 0;
#pragma unroll
	for (int c = 0; c < CC; c++) r += acc[c];
	out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

// vector-broadcast-fed: same stream, features loaded with uniform-address global_load_dwordx4
template <int CC>
__global__ __launch_bounds__(256) void vfeed_kernel(const uint32_t* ids, const float* rows, float* out, int n, int C)
{
	float acc[CC];
#pragma unroll
	for (int c = 0; c < CC; c++) acc[c] = 0.f;
	float w = 1.0f + threadIdx.x * 1e-7f;
	const uint32_t* myids = ids + (size_t)blockIdx.x * n;
	for (int j = 0; j < n; j++) {
		uint32_t id = myids[j];
		const float4* f = (const float4*)(rows + (size_t)id * C);
#pragma unroll
		for (int c = 0; c < CC / 4; c++) {
			float4 v = f[c];
			acc[4 * c] = __builtin_fmaf(v.x, w, acc[4 * c]);
			acc[4 * c + 1] = __builtin_fmaf(v.y, w, acc[4 * c + 1]);
			acc[4 * c + 2] = __builtin_fmaf(v.z, w, acc[4 * c + 2]);
			acc[4 * c + 3] = __builtin_fmaf(v.w, w, acc[4 * c + 3]);
		}
	}
	float r = 0;
#pragma unroll
	for (int c = 0; c < CC; c++) r += acc[c];
	out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

// LDS-broadcast-fed with PX pixels per lane: rows staged in LDS once, then each wave reads
// ds_read_b128 broadcast and does PX FMAs per feature value.
template <int CC, int PX>
__global__ __launch_bounds__(256) void ldsfeed_kernel(const float* __restrict__ rows, float* __restrict__ out, int n)
{
	__shared__ float4 s_rows[64 * CC / 4];
	for (int i = threadIdx.x; i < 64 * CC / 4; i += 256) s_rows[i] = ((const float4*)rows)[i];
	__syncthreads();
	float acc[PX][CC];
#pragma unroll
	for (int p = 0; p < PX; p++)
#pragma unroll
		for (int c = 0; c < CC; c++) acc[p][c] = 0.f;
	float w[PX];
#pragma unroll
	for (int p = 0; p < PX; p++) w[p] = 1.0f + threadIdx.x * 1e-7f * (p + 1);
	for (int j = 0; j < n; j++) {
		const float4* f = s_rows + (j & 63) * (CC / 4);
#pragma unroll
		for (int c = 0; c < CC / 4; c++) {
			float4 v = f[c];
#pragma unroll
			for (int p = 0; p < PX; p++) {
				acc[p][4 * c] = __builtin_fmaf(v.x, w[p], acc[p][4 * c]);
				acc[p][4 * c + 1] = __builtin_fmaf(v.y, w[p], acc[p][4 * c + 1]);
				acc[p][4 * c + 2] = __builtin_fmaf(v.z, w[p], acc[p][4 * c + 2]);
				acc[p][4 * c + 3] = __builtin_fmaf(v.w, w[p], acc[p][4 * c + 3]);
			}
		}
	}
	float r = 0;
#pragma unroll
	for (int p = 0; p < PX; p++)
#pragma unroll
		for (int c = 0; c < CC; c++) r += acc[p][c];
	out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F>
static float time_ms(F launch, int reps = 5)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	launch();
	CK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int r = 0; r < reps; r++) {
		CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	return best;
}

int main()
{
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
	const int CUs = prop.multiProcessorCount;
	float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 65536));
	{
		const int iters = 20000;
		const char* names[4] = {"v_fmac_f32 vgpr", "v_fmac_f32 sgpr", "v_pk_fma_f32 vgpr", "v_pk_fma_f32 sgpr"};
		for (int wpc = 1; wpc <= 4; wpc *= 2) {
			const int blocks = CUs * wpc * 2;
			for (int m = 0; m < 4; m++) {
				float ms = 0;
				switch (m) {
				case 0: ms = time_ms([&] { fma_kernel<0><<<blocks, 256>>>(out, iters, 1.0001f, 0.9999f); }); break;
				case 1: ms = time_ms([&] { fma_kernel<1><<<blocks, 256>>>(out, iters, 1.0001f, 0.9999f); }); break;
				case 2: ms = time_ms([&] { fma_kernel<2><<<blocks, 256>>>(out, iters, 1.0001f, 0.9999f); }); break;
				case 3: ms = time_ms([&] { fma_kernel<3><<<blocks, 256>>>(out, iters, 1.0001f, 0.9999f); }); break;
				}
				double fmas = (double)blocks * 256 * iters * 32;
				printf("FMA %-20s blocks/CU %d: %.3f ms  %.2f TFLOP/s  (%.1f lane-FMA/clk/CU @2.4GHz)\n", names[m], wpc * 2, ms,
				       2 * fmas / ms / 1e9, fmas / (ms * 1e-3) / CUs / 2.4e9);
			}
		}
	}
	// scalar / vector / LDS feed
	{
		const int C = 512;
		for (int big = 0; big < 2; big++) {
			const size_t nrows = big ? 1000000 : 4096;   // 2 GB (HBM) vs 8 MB (L2/MALL)
			float* rows; CK(hipMalloc(&rows, nrows * C * sizeof(float)));
			CK(hipMemset(rows, 0, nrows * C * sizeof(float)));
			const int n = 256, blocks = CUs * 24;
			std::vector<uint32_t> h((size_t)blocks * n);
			uint64_t s = 88172645463325252ull;
			for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % nrows); }
			uint32_t* ids; CK(hipMalloc(&ids, h.size() * 4));
			CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
			auto report = [&](const char* nm, int CC, float ms) {
				double fmas = (double)blocks * 256 * n * CC;
				double bytes = (double)blocks * 4 /*waves*/ * n * CC * 4;   // per-wave fetch
				printf("%-28s rows=%-8zu CC=%-3d %.3f ms  %.2f TFLOP/s  %.1f lane-FMA/clk/CU  wave-fetch %.2f TB/s (block-unique %.2f TB/s)\n", nm, nrows, CC, ms,
				       2 * fmas / ms / 1e9, fmas / (ms * 1e-3) / CUs / 2.4e9, bytes / ms / 1e9, bytes / 4 / ms / 1e9);
			};
			report("sfeed pk CC=64", 64, time_ms([&] { sfeed_kernel<64, 1><<<blocks, 256>>>(ids, rows, out, n, C); }));
			report("sfeed fmac CC=64", 64, time_ms([&] { sfeed_kernel<64, 0><<<blocks, 256>>>(ids, rows, out, n, C); }));
			report("sfeed pk CC=128", 128, time_ms([&] { sfeed_kernel<128, 1><<<blocks, 256>>>(ids, rows, out, n, C); }));
			report("sfeed fmac CC=128", 128, time_ms([&] { sfeed_kernel<128, 0><<<blocks, 256>>>(ids, rows, out, n, C); }));
			report("sfeed pk CC=32", 32, time_ms([&] { sfeed_kernel<32, 1><<<blocks, 256>>>(ids, rows, out, n, C); }));
			report("vfeed CC=64", 64, time_ms([&] { vfeed_kernel<64><<<blocks, 256>>>(ids, rows, out, n, C); }));
			report("vfeed CC=128", 128, time_ms([&] { vfeed_kernel<128><<<blocks, 256>>>(ids, rows, out, n, C); }));
			CK(hipFree(rows)); CK(hipFree(ids));
		}
		float* rows; CK(hipMalloc(&rows, 64 * 128 * sizeof(float)));
		CK(hipMemset(rows, 0, 64 * 128 * sizeof(float)));
		const int n = 4096, blocks = CUs * 8;
		auto rep = [&](const char* nm, int CC, int PX, float ms) {
			double fmas = (double)blocks * 256 * n * CC * PX;
			printf("%-28s CC=%-3d PX=%d %.3f ms  %.2f TFLOP/s  %.1f lane-FMA/clk/CU\n", nm, CC, PX, ms, 2 * fmas / ms / 1e9, fmas / (ms * 1e-3) / CUs / 2.4e9);
		};
		rep("ldsfeed", 64, 1, time_ms([&] { ldsfeed_kernel<64, 1><<<blocks, 256>>>(rows, out, n); }));
		rep("ldsfeed", 64, 2, time_ms([&] { ldsfeed_kernel<64, 2><<<blocks, 256>>>(rows, out, n); }));
		rep("ldsfeed", 32, 4, time_ms([&] { ldsfeed_kernel<32, 4><<<blocks, 256>>>(rows, out, n); }));
		rep("ldsfeed", 128, 1, time_ms([&] { ldsfeed_kernel<128, 1><<<blocks, 256>>>(rows, out, n); }));
	}
	return 0;
}
