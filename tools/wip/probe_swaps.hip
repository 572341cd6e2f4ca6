#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
	const int lane = threadIdx.x;
	const unsigned a = lane, b = 100 + lane;
	const auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
	const auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
	out[0 * 64 + lane] = s32[0]; out[1 * 64 + lane] = s32[1];
	out[2 * 64 + lane] = s16[0]; out[3 * 64 + lane] = s16[1];
	out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x128, 0xF, 0xF, false);
	out[5 * 64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x141, 0xF, 0xF, false);
	out[6 * 64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x1B, 0xF, 0xF, false);
	out[7 * 64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0xB1, 0xF, 0xF, false);
}
int main()
{
	int *d, h[512];
	(void)hipMalloc(&d, 2048);
	k<<<1, 64>>>(d);
	(void)hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
	const char* names[8] = {"p32.0", "p32.1", "p16.0", "p16.1", "ror8", "hmir", "q3210", "q1032"};
	for (int r = 0; r < 8; r++) {
		printf("%s:", names[r]);
		for (int l = 0; l < 64; l++) printf(" %d", h[r * 64 + l]);
		printf("\n");
	}
	return 0;
}
