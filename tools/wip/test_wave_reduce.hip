// standalone check of wave_sum8 (sgs_device.h): hipcc --offload-arch=gfx950 -O2 -I semantic-gaussians_amd/csrc tools/wip/test_wave_reduce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "sgs_device.h"
__global__ void k(const float* in, float* out, float* dbg)
{
	const int lane = threadIdx.x;
	float v[8];
	for (int c = 0; c < 8; c++) v[c] = in[c * 64 + lane];
	const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), false, false);
	dbg[lane] = __builtin_bit_cast(float, s[0]) + __builtin_bit_cast(float, s[1]);
	const float u = sgs::wave_sum8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
	dbg[64 + lane] = u;
	if ((lane & 7) == 0) out[sgs::wave_sum8_component(lane)] = u;
}
int main()
{
	float h[512], *d, *o, *g, r[8], dbg[128];
	for (int i = 0; i < 512; i++) h[i] = (float)((i * 37) % 101) - 50.f;
	(void)hipMalloc(&d, 2048); (void)hipMalloc(&o, 32); (void)hipMalloc(&g, 512);
	(void)hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
	k<<<1, 64>>>(d, o, g);
	(void)hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
	(void)hipMemcpy(dbg, g, 512, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int c = 0; c < 8; c++) {
		float s = 0;
		for (int l = 0; l < 64; l++) s += h[c * 64 + l];
		printf("comp %d: got %g want %g\n", c, r[c], s);
		bad += r[c] != s;
	}
	printf("fold32 lane0: got %g want %g ; lane 40: got %g want %g\n", dbg[0], h[0] + h[32], dbg[40], h[64 + 8] + h[64 + 40]);
	printf("u:"); for (int l = 0; l < 64; l++) printf(" %g", dbg[64 + l]); printf("\n");
	printf(bad ? "FAIL\n" : "OK\n");
	return bad;
}
