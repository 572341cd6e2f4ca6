// ubench_store.hip -- what does the (C,H,W) output write pattern of the blend epilogue cost?
// Every variant writes the same 512 x 968 x 1296 fp32 image (2.57 GB) from 4941 x 4 workgroups
// of 256 lanes (tile x 128 channels, wave = 64 channels x 128 pixels), only the
// lane -> address map differs.   hipcc --offload-arch=gfx950 -O3 -o ubench_store ubench_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int W = 1296, H = 968, C = 512, GX = 81, GY = 61, NT = GX * GY;

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, int per_xcd, int total,
							  const float4* __restrict__ src = nullptr, float* __restrict__ sink = nullptr)
{
	const int b = blockIdx.x;
	const int v = (b & 7) * per_xcd + (b >> 3);
	if (v >= total) return;
	int tile = v >> 2, chunk = v & 3;
	if (MODE == 5) {   // chunk-major inside each tile row: concurrent workgroups share the channel chunk
		const int row = v / (GX * 4), i = v - row * GX * 4;
		chunk = i / GX;
		tile = row * GX + (i - chunk * GX);
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int cgrp = wave & 1, pgrp = wave >> 1, half = lane >> 5, l31 = lane & 31;
	const size_t HW = (size_t)H * W;
	const int tx = tile % GX, ty = tile / GX;
	float val = (float)v;
	if (src) {   // mixed traffic: 64 KB of reads per workgroup (1.3 GB in all), hashed 4-KB runs, before the stores
		float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
		const size_t nruns = (size_t)1 << 18;   // 1 GiB source / 4 KB
#pragma unroll 4
		for (int i = 0; i < 16; i++) {
			const size_t run = ((size_t)v * 16 + i) * 2654435761ull % nruns;
			const float4 x = src[run * 256 + threadIdx.x];
			acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
		}
		val += (acc.x + acc.y + acc.z + acc.w) * 1e-30f;
		if (val == 12345.678f) sink[0] = val;
	}
	if (MODE == 0) {   // contiguous 128 KB per workgroup, 256 B per wave store
		float* base = out + (size_t)v * 32768 + wave * 8192 + lane;
#pragma unroll 8
		for (int i = 0; i < 128; i++) base[i * 64] = val;
	} else if (MODE == 1 || MODE == 3 || MODE == 5) {   // the kernel's pattern: 4 x 64-B segments per wave store
#pragma unroll
		for (int pb = 0; pb < 4; pb++) {
			const int qidx = pgrp * 128 + pb * 32 + l31;
			const int x = tx * 16 + (qidx & 15);
			const int y = ty * 16 + (qidx >> 6) * 4 + ((qidx & 63) >> 4);
			if (x < W && y < H) {
				const size_t pix = (size_t)y * W + x;
#pragma unroll
				for (int cb = 0; cb < 2; cb++)
#pragma unroll
					for (int r = 0; r < 16; r++) {
						const int c = chunk * 128 + cgrp * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
						if (MODE == 3) __builtin_nontemporal_store(val, &out[(size_t)c * HW + pix]);
						else out[(size_t)c * HW + pix] = val;
					}
			}
		}
	} else if (MODE == 2) {   // tile PAIR x 64 channels: 128-B runs (32 consecutive x), 2 channels per wave store
		const int pair = v >> 3, sub = v & 7;                  // 8 workgroups of 64 channels per pair
		const int ptx = (pair % ((GX + 1) / 2)) * 2, pty = pair / ((GX + 1) / 2);
		if (pty >= GY) return;
		// wave: 16 channels x 512 pixels -> 128 stores of (2 channels x 32 px)
		for (int row = 0; row < 16; row++) {
			const int y = pty * 16 + row, x = ptx * 16 + l31;
			if (x < W && y < H) {
#pragma unroll
				for (int k = 0; k < 8; k++) {
					const int c = sub * 64 + wave * 16 + k * 2 + half;
					out[(size_t)c * HW + (size_t)y * W + x] = val;
				}
			}
		}
	} else if (MODE >= 15 && MODE <= 20) {
		// staggered tile pairs x 64 channels: even rows x in [32k, 32k+32), odd rows x in [32k+16, 32k+48):
		// every 128-B line of the (pitch 5184) image is owned by ONE workgroup.
		// 15: wave 0/1 write the left/right 64-B halves of even rows, wave 2/3 of odd rows (same time)
		// 16: one wave store = 2 complete lines (waves 0,1: even rows, 4 rows each; 2,3: odd rows)
		const int npx = (GX + 1) / 2 + 1;
		int pair = v >> 3, sub = v & 7;
		if (MODE == 17) {   // x-major: the 8 channel groups of a tile row are 8 consecutive sweeps over its pairs
			const int row = v / (npx * 8), i = v - row * npx * 8;
			sub = i / npx;
			pair = row * npx + (i - sub * npx);
		}
		const int pk = pair % npx, pty = pair / npx;
		if (pty >= GY) return;
		if (MODE == 15) {
			const int parity = wave >> 1, side = wave & 1;
			for (int rr = 0; rr < 8; rr++) {
				const int y = pty * 16 + rr * 2 + parity;
				const int x = pk * 32 + parity * 16 - 32 + side * 16 + (lane & 15);
#pragma unroll
				for (int k = 0; k < 16; k++) {
					const int c = sub * 64 + k * 4 + (lane >> 4);
					if (x >= 0 && x < W && y < H) out[(size_t)c * HW + (size_t)y * W + x] = val;
				}
			}
		} else {
			const int parity = wave >> 1;
			for (int rr = 0; rr < 4; rr++) {
				const int y = pty * 16 + ((wave & 1) * 4 + rr) * 2 + parity;
				const int x = pk * 32 + parity * 16 - 32 + l31;
#pragma unroll
				for (int k = 0; k < 32; k++) {
					const int c = sub * 64 + k * 2 + half;
					if (x >= 0 && x < W && y < H) {
						float* p = &out[(size_t)c * HW + (size_t)y * W + x];
						if (MODE == 18) asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(val) : "memory");
						else if (MODE == 19) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(val) : "memory");
						else if (MODE == 20) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(val) : "memory");
						else *p = val;
					}
				}
			}
		}
	} else if (MODE == 11) {   // tile pairs with a 128-B-aligned pitch (1312 px): aligned full-line runs
		constexpr int WP = 1312;
		const int pair = v >> 3, sub = v & 7;
		const int ptx = (pair % ((GX + 1) / 2)) * 2, pty = pair / ((GX + 1) / 2);
		if (pty >= GY) return;
		for (int row = 0; row < 16; row++) {
			const int y = pty * 16 + row, x = ptx * 16 + l31;
			if (x < W && y < H) {
#pragma unroll
				for (int k = 0; k < 8; k++) {
					const int c = sub * 64 + wave * 16 + k * 2 + half;
					out[(size_t)c * H * WP + (size_t)y * WP + x] = val;
				}
			}
		}
	} else if (MODE >= 12 && MODE <= 14) {   // hashed runs of 64 / 128 / 256 B, one store = 4 / 2 / 1 runs
		constexpr int RUN = MODE == 12 ? 64 : (MODE == 13 ? 128 : 256);
		constexpr int LPR = RUN / 4;                    // lanes per run
		const size_t nslots = (size_t)C * H * W * 4 / RUN;
		for (int i = 0; i < 128; i++) {
			const size_t id = (((size_t)v * 4 + wave) * 128 + i) * (64 / LPR) + lane / LPR;
			const size_t slot = id * 2654435761ull % nslots;
			out[slot * LPR + (lane % LPR)] = val;
		}
	} else if (MODE >= 6) {   // runs of RUN bytes at hashed places: where is the DRAM locality knee?
		constexpr int RUN = MODE == 6 ? 512 : (MODE == 7 ? 1024 : (MODE == 8 ? 2048 : (MODE == 9 ? 4096 : 16384)));
		constexpr int NRUN = 131072 / RUN;             // runs per workgroup
		constexpr int PER = RUN / 256;                  // wave stores per run
		const size_t nslots = (size_t)C * H * W * 4 / RUN;
		for (int r = wave; r < NRUN; r += 4) {
			const size_t slot = ((size_t)v * NRUN + r) * 2654435761ull % nslots;
			float* base = out + slot * (RUN / 4) + lane;
#pragma unroll
			for (int i = 0; i < PER; i++) base[i * 64] = val;
		}
	} else if (MODE == 4) {   // one full image row segment of 4 tiles (256 B) per wave store
		const int quad = v >> 4, sub = v & 15;                 // 16 workgroups of 32 channels per 4 tiles
		const int qtx = (quad % ((GX + 3) / 4)) * 4, qty = quad / ((GX + 3) / 4);
		if (qty >= GY) return;
		for (int row = 0; row < 16; row++) {
			const int y = qty * 16 + row, x = qtx * 16 + lane;
			if (x < W && y < H) {
#pragma unroll
				for (int k = 0; k < 8; k++) {
					const int c = sub * 32 + wave * 8 + k;
					out[(size_t)c * HW + (size_t)y * W + x] = val;
				}
			}
		}
	}
}

int main()
{
	float* out;
	const size_t bytes = (size_t)C * H * W * 4;
	if (hipMalloc(&out, bytes + (128 << 20)) != hipSuccess) return 1;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const char* names[21] = {"contiguous 128 KB per WG", "blend epilogue pattern (4 x 64 B per store)",
				"tile pairs (2 x 128 B per store)", "epilogue pattern, nontemporal", "tile quads (256 B per store)", "epilogue pattern, chunk-major block order", "hashed 512-B runs", "hashed 1-KB runs", "hashed 2-KB runs", "hashed 4-KB runs", "hashed 16-KB runs", "tile pairs, pitch 1312 (aligned 128-B runs)", "hashed 64-B runs", "hashed 128-B runs", "hashed 256-B runs", "staggered pairs, half lines from 2 waves", "staggered pairs, full lines per store", "as 16, consecutive workgroups along x", "as 16, nt", "as 16, sc0 sc1", "as 16, sc1"};
	float4* src;
	float* sink;
	hipMalloc(&src, (size_t)1 << 30);
	hipMalloc(&sink, 64);
	hipMemset(src, 0, (size_t)1 << 30);
	for (int pass = 0; pass < 2; pass++)
	for (int mode = 0; mode < 21; mode++) {
		if (pass == 1 && mode != 0 && mode != 1 && mode < 16) continue;
		const int total = (mode >= 15) ? ((GX + 1) / 2 + 1) * GY * 8 : (mode == 2 || mode == 11) ? ((GX + 1) / 2) * GY * 8 : (mode == 4 ? ((GX + 3) / 4) * GY * 16 : NT * 4);
		const int per_xcd = (total + 7) / 8;
		float best = 1e9f;
		for (int rep = 0; rep < 6; rep++) {
			hipEventRecord(e0);
			switch (mode) {
			case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 3: hipLaunchKernelGGL(store_kernel<3>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 4: hipLaunchKernelGGL(store_kernel<4>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 5: hipLaunchKernelGGL(store_kernel<5>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 6: hipLaunchKernelGGL(store_kernel<6>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 7: hipLaunchKernelGGL(store_kernel<7>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 8: hipLaunchKernelGGL(store_kernel<8>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 9: hipLaunchKernelGGL(store_kernel<9>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 10: hipLaunchKernelGGL(store_kernel<10>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 11: hipLaunchKernelGGL(store_kernel<11>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 12: hipLaunchKernelGGL(store_kernel<12>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 13: hipLaunchKernelGGL(store_kernel<13>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 14: hipLaunchKernelGGL(store_kernel<14>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 15: hipLaunchKernelGGL(store_kernel<15>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 16: hipLaunchKernelGGL(store_kernel<16>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 17: hipLaunchKernelGGL(store_kernel<17>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 18: hipLaunchKernelGGL(store_kernel<18>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 19: hipLaunchKernelGGL(store_kernel<19>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			case 20: hipLaunchKernelGGL(store_kernel<20>, dim3(per_xcd * 8), dim3(256), 0, 0, out, per_xcd, total, pass ? src : nullptr, sink); break;
			}
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (rep > 0 && ms < best) best = ms;
		}
		printf("%s mode %d  %-46s %.3f ms  %.2f TB/s written%s\n", pass ? "R+W" : "W  ", mode, names[mode], best, bytes / best * 1e-9, pass ? " (+1.3 GB read)" : "");
	}
	return 0;
}
