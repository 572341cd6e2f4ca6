// composite.hip -- the "over" composite of depth-ordered Gaussian-shard partials (BASELINE config 5):
//     out[c][px] = sum_s (prod_{s' < s} T_s'[px]) * A_s[c][px]  +  (prod_s T_s[px]) * bg[c]
// One launch reads every partial ONCE and writes the band once (the torch formulation re-reads and re-writes the
// (C, h, W) band per shard: 3 x (S - 1) + 2 passes).  Pure HBM streaming: 4 (S + 1) C h W bytes + the T planes.
#include "sgs_kernels.h"

namespace sgs {

namespace {

typedef float f32x4_ __attribute__((ext_vector_type(4)));

struct CompositeArgs {
	const float* A[SGS_MAX_SHARDS];   // (C, rows, W) each, rows `pitch_a` floats apart
	const float* T[SGS_MAX_SHARDS];   // (rows, W)
};

// thread = 4 consecutive pixels of one row; blockIdx.y = a group of CG channels
template <int CG>
__global__ __launch_bounds__(256) void composite_over_kernel(CompositeArgs a, int S, const float* __restrict__ bg,
							      float* __restrict__ out, float* __restrict__ t_out, int C, size_t npix)
{
	const size_t p4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
	if (p4 >= npix) return;
	const bool full = p4 + 4 <= npix;
	float pre[SGS_MAX_SHARDS][4];   // transmittance in front of shard s
	float acc_t[4] = {1.f, 1.f, 1.f, 1.f};
	for (int s = 0; s < S; s++) {
		float t[4];
		if (full) {
			const float4 v = *reinterpret_cast<const float4*>(a.T[s] + p4);
			t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
		} else {
			for (int i = 0; i < 4; i++) t[i] = p4 + i < npix ? a.T[s][p4 + i] : 1.f;
		}
		for (int i = 0; i < 4; i++) {
			pre[s][i] = acc_t[i];
			acc_t[i] *= t[i];
		}
	}
	const int c0 = blockIdx.y * CG;
	if (c0 == 0 && t_out) {
		for (int i = 0; i < 4; i++)
			if (p4 + i < npix) t_out[p4 + i] = acc_t[i];
	}
	for (int c = c0; c < c0 + CG && c < C; c++) {
		const size_t o = (size_t)c * npix + p4;
		float r[4];
		const float b = bg ? bg[c] : 0.f;
		// the torch reference order: A_0 + T_0 A_1 + (T_0 T_1) A_2 + ... + bg * T_total, left to right
		for (int i = 0; i < 4; i++) r[i] = 0.f;
		for (int s = 0; s < S; s++) {
			float v[4];
			if (full) {
				const f32x4_ q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(a.A[s] + o));
				v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
			} else {
				for (int i = 0; i < 4; i++) v[i] = p4 + i < npix ? a.A[s][o + i] : 0.f;
			}
			for (int i = 0; i < 4; i++) r[i] = s == 0 ? v[i] : r[i] + pre[s][i] * v[i];
		}
		if (bg)
			for (int i = 0; i < 4; i++) r[i] = r[i] + b * acc_t[i];
		if (full) {
			const f32x4_ w = {r[0], r[1], r[2], r[3]};
			__builtin_nontemporal_store(w, reinterpret_cast<f32x4_*>(out + o));
		} else {
			for (int i = 0; i < 4; i++)
				if (p4 + i < npix) out[o + i] = r[i];
		}
	}
}

} // namespace

hipError_t launch_composite_over(hipStream_t st, int S, const float* const* A, const float* const* T, const float* bg,
				 float* out, float* t_out, int C, size_t npix)
{
	CompositeArgs a;
	for (int s = 0; s < SGS_MAX_SHARDS; s++) {
		a.A[s] = s < S ? A[s] : nullptr;
		a.T[s] = s < S ? T[s] : nullptr;
	}
	constexpr int CG = 16;
	const dim3 grid((unsigned)((npix + 1023) / 1024), (unsigned)((C + CG - 1) / CG));
	hipLaunchKernelGGL(composite_over_kernel<CG>, grid, dim3(256), 0, st, a, S, bg, out, t_out, C, npix);
	return hipGetLastError();
}

} // namespace sgs
