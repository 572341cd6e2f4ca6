// ubench_store_cus.hip -- round 6: is the store rate of the accumulate sweep a CHIP limit or a PER-COMPUTE-UNIT limit?
// The CU-partition scan (profiles/r06_cu_partition.txt) shows the sweep slowing in proportion to the compute units it loses, which "the burst is
// the chip's write rate"  does not predict.  This writes the same 2.57 GB with full-line stores (256 B per wave
// instruction, the sweep's pattern reduced to its essence) from a stream confined to n CUs, at two occupancies:
//   fat  : 8-wave workgroups that own their CU (128 KB of LDS), i.e. the sweep's occupancy -- 64 stores per wave in a burst, then a pause of
//          `gap` x the burst's issue time (the matrix phase), like the sweep's duty cycle
//   thin : 4-wave workgroups, as many as fit
// and reports TB/s and bytes per clock and CU (at 2.1 GHz).    hipcc --offload-arch=gfx950 -O3 -o ubench_store_cus ubench_store_cus.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr size_t BYTES = (size_t)512 * 968 * 1296 * 4;

template <bool FAT>
__global__ __launch_bounds__(FAT ? 512 : 256) void store_kernel(float* __restrict__ out, size_t floats_per_wg, int gap)
{
	extern __shared__ float lds[];
	const int nw = FAT ? 8 : 4;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	float* base = out + (size_t)blockIdx.x * floats_per_wg + (size_t)wave * (floats_per_wg / nw) + lane;
	const size_t n = floats_per_wg / nw / 64;   // store instructions of this wave
	float v = (float)blockIdx.x;
	if (FAT && gap < 0) lds[threadIdx.x] = v;   // (keeps the dynamic LDS alive)
	for (size_t i = 0; i < n; i += 64) {
#pragma unroll 16
		for (int j = 0; j < 64; j++)
			if (i + j < n) base[(i + j) * 64] = v;
		// the "matrix phase": gap x 64 x 8 cycles of dependent VALU work
		for (int g = 0; g < gap * 64; g++) {
			asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(v));
		}
	}
	if (v == 12345.6789f) out[0] = v;
}

int main(int argc, char** argv)
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	float* out;
	if (hipMalloc(&out, BYTES + (64 << 20)) != hipSuccess) return 1;
	hipDeviceProp_t pr;
	hipGetDeviceProperties(&pr, 0);
	const int ncu = pr.multiProcessorCount;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipFuncSetAttribute((const void*)store_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
	printf("%s: %d CUs, %.2f GB per launch\n", pr.name, ncu, BYTES * 1e-9);
	const int cus[] = {256, 224, 192, 128, 64, 32};
	for (int fat = 1; fat >= 0; fat--)
		for (int gap : {0, 1, 3})
			for (int n : cus) {
				if (n > ncu) continue;
				if (!fat && gap) continue;
				std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
				for (int b = ncu - n; b < ncu; b++) mask[b >> 5] |= 1u << (b & 31);
				hipStream_t st;
				if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("stream creation failed\n"); return 2; }
				const size_t per_wg = 128 * 1024 / 4 * (fat ? 2 : 1);   // floats: 256 KB per fat workgroup (a tile pair's share), 128 KB per thin one
				const int nwg = (int)(BYTES / 4 / per_wg);
				float best = 1e9f;
				for (int rep = 0; rep < 5; rep++) {
					hipEventRecord(e0, st);
					if (fat) hipLaunchKernelGGL(store_kernel<true>, dim3(nwg), dim3(512), 128 * 1024, st, out, per_wg, gap);
					else hipLaunchKernelGGL(store_kernel<false>, dim3(nwg), dim3(256), 0, st, out, per_wg, gap);
					hipEventRecord(e1, st);
					const hipError_t es = hipEventSynchronize(e1);
					if (es != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(es)); return 3; }
					float ms;
					hipEventElapsedTime(&ms, e0, e1);
					if (rep > 0 && ms < best) best = ms;
				}
				const double tbs = (double)nwg * per_wg * 4 / best * 1e-9;
				printf("%s gap %d  %3d CUs: %.3f ms  %.2f TB/s  %.1f B/clk/CU\n", fat ? "fat (1 WG of 8 waves per CU)" : "thin (4-wave WGs)         ", gap, n, best, tbs,
				       tbs * 1e12 / n / 2.1e9);
				hipStreamDestroy(st);
			}
	return 0;
}
