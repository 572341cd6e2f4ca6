// blend_bwd.hip -- backward of the alpha-composite (runtime channel count) and of the
// per-Gaussian preprocess, for gfx950.
//
// Behaviour restated from CR/cuda_rasterizer/backward.cu:394-552 (renderCUDA<C> backward),
// :141-271 (computeCov2DCUDA), :341-391 (preprocessCUDA backward), :275-336 (computeCov3D
// backward), :20-136 (computeColorFromSH backward); see SURVEY.md A.5.
//
// The reference instantiates its blend backward on the compile-time NUM_CHANNELS = 3
// (config.h:15).  Here the channel count is a runtime argument; the colour state is
// processed in chunks of CC channels, one (tile, chunk) per workgroup.  Every geometry
// gradient is linear in dL/dalpha and dL/dalpha is a plain sum over channels, so each chunk
// adds its partial contribution through the same atomics with no cross-chunk ordering.
//
// MI355X notes:
//  * lane = pixel, wave = 16x4 strip (as in the forward); the back-to-front walk starts at
//    the strip's largest n_contrib, so whole batches behind it are skipped;
//  * gradients are reduced over the 64 lanes of the wave with DPP adds before ONE atomic
//    per (wave, Gaussian, component) -- the reference issues one atomic per pixel;
//  * `last_color` of the reference is folded into the recurrence
//    rec <- alpha*c + (1-alpha)*rec evaluated at the end of the step (same values, no
//    second C-sized register array).
#include "sgs_kernels.h"

namespace sgs {

struct StagedEntryB {
	float a2, b2, c2, o;
	float x, y;
	uint32_t id;
	float ca;   // conic.x
	float cb, cc;
	float pad0, pad1;
};

template <int CC>
__global__ __launch_bounds__(256) void blend_bwd_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
	const float* __restrict__ bg, const float2* __restrict__ means2D,
	const float4* __restrict__ conic_opacity, const float* __restrict__ colors,
	const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
	const float* __restrict__ dL_dpixels, float* __restrict__ dL_dmean2D,
	float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
	int W, int H, int C, int gx, int nchunks, const uint32_t* __restrict__ gate)
{
	if (gate && gate[1] == 0u) return;   // the work-list path did the job
	const int tile = blockIdx.x / nchunks;
	const int chunk = blockIdx.x - tile * nchunks;
	const int c0 = chunk * CC;
	const int cn = (C - c0) < CC ? (C - c0) : CC;
	const int tx = tile % gx, ty = tile / gx;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int px = tx * SGS_TILE + (lane & 15);
	const int py = ty * SGS_TILE + wave * 4 + (lane >> 4);
	const bool inside = px < W && py < H;
	const float pxf = (float)px, pyf = (float)py;
	const size_t HW = (size_t)H * W;
	const size_t pix = (size_t)py * W + px;
	const uint2 range = ranges[tile];

	__shared__ StagedEntryB s_e[256];
	__shared__ int s_max[4];

	const float T_final = inside ? final_Ts[pix] : 0.f;
	float T = T_final;
	const int last_contributor = inside ? (int)n_contrib[pix] : 0;

	float rec[CC], g[CC];
	float bg_dot = 0.f;
#pragma unroll
	for (int c = 0; c < CC; c++) {
		rec[c] = 0.f;
		g[c] = (inside && c < cn) ? dL_dpixels[(size_t)(c0 + c) * HW + pix] : 0.f;
	}
#pragma unroll
	for (int c = 0; c < CC; c++)
		if (c < cn) bg_dot += bg[c0 + c] * g[c];

	const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

	// strip-level and tile-level starting points of the back-to-front walk
	int wave_max = last_contributor;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const int o = __shfl_xor(wave_max, off);
		wave_max = o > wave_max ? o : wave_max;
	}
	if (lane == 0) s_max[wave] = wave_max;
	__syncthreads();
	int tile_max = s_max[0];
	tile_max = s_max[1] > tile_max ? s_max[1] : tile_max;
	tile_max = s_max[2] > tile_max ? s_max[2] : tile_max;
	tile_max = s_max[3] > tile_max ? s_max[3] : tile_max;

	// entries [0, tile_max) are walked from tile_max-1 down to 0 in batches of 256
	for (int hi = tile_max; hi > 0; hi -= 256) {
		const int n = hi < 256 ? hi : 256;   // this batch covers indices [hi-n, hi)
		__syncthreads();
		if ((int)threadIdx.x < n) {
			// slot k holds entry index hi-1-k (descending)
			const uint32_t id = point_list[range.x + (uint32_t)(hi - 1 - (int)threadIdx.x)];
			const float2 xy = means2D[id];
			const float4 co = conic_opacity[id];
			StagedEntryB e;
			e.a2 = -0.5f * co.x;
			e.b2 = -co.y;
			e.c2 = -0.5f * co.z;
			e.o = co.w;
			e.x = xy.x;
			e.y = xy.y;
			e.id = id;
			e.ca = co.x;
			e.cb = co.y;
			e.cc = co.z;
			e.pad0 = e.pad1 = 0.f;
			s_e[threadIdx.x] = e;
		}
		__syncthreads();
		for (int k = 0; k < n; k++) {
			const int idx = hi - 1 - k;          // 0-based list index of this entry
			if (idx >= wave_max) continue;       // nobody in this strip got that far
			const StagedEntryB e = s_e[k];
			const float dx = e.x - pxf, dy = e.y - pyf;
			const float power =
				__builtin_fmaf(e.b2 * dx, dy, __builtin_fmaf(e.c2 * dy, dy, (e.a2 * dx) * dx));
			const float G = expf_contract(power);
			const float alpha = fmin_(0.99f, e.o * G);
			const bool valid = inside && (idx < last_contributor) && !(power > 0.0f) &&
					   !(alpha < 1.0f / 255.0f);
			if (__ballot(valid) == 0ull) continue;
			const float oma = 1.f - alpha;
			if (valid) T = T / oma;
			const float wgt = valid ? alpha * T : 0.f;   // dchannel_dcolor
			const uint32_t id = __builtin_amdgcn_readfirstlane(e.id);
			const float* __restrict__ col = colors + (size_t)id * C + c0;
			float* __restrict__ dcol = dL_dcolors + (size_t)id * C + c0;
			float S = 0.f;
#pragma unroll
			for (int c = 0; c < CC; c++) {
				if (c < cn) {
					const float cv = col[c];
					// contribution to dL/dalpha uses the colour accumulated BEHIND this entry
					S += (cv - rec[c]) * g[c];
					const float gc = wave_sum(wgt * g[c]);
					if (lane == 0) atomicAdd(&dcol[c], gc);
					// fold this entry into the running "behind" colour for the next one
					if (valid) rec[c] = alpha * cv + oma * rec[c];
				}
			}
			float dL_dalpha = S * T;
			dL_dalpha += (-T_final / oma) * bg_dot;
			if (!valid) dL_dalpha = 0.f;
			const float dL_dG = e.o * dL_dalpha;
			const float Gv = valid ? G : 0.f;   // keeps inf/NaN of rejected lanes out of the sums
			const float gdx = Gv * dx, gdy = Gv * dy;
			const float dG_ddelx = -gdx * e.ca - gdy * e.cb;
			const float dG_ddely = -gdy * e.cc - gdx * e.cb;
			const float m0 = wave_sum(dL_dG * dG_ddelx * ddelx_dx);
			const float m1 = wave_sum(dL_dG * dG_ddely * ddely_dy);
			const float k0 = wave_sum(-0.5f * gdx * dx * dL_dG);
			const float k1 = wave_sum(-0.5f * gdx * dy * dL_dG);
			const float k3 = wave_sum(-0.5f * gdy * dy * dL_dG);
			const float op = wave_sum(Gv * dL_dalpha);
			if (lane == 0) {
				atomicAdd(&dL_dmean2D[3 * (size_t)id], m0);
				atomicAdd(&dL_dmean2D[3 * (size_t)id + 1], m1);
				atomicAdd(&dL_dconic[4 * (size_t)id], k0);
				atomicAdd(&dL_dconic[4 * (size_t)id + 1], k1);
				atomicAdd(&dL_dconic[4 * (size_t)id + 3], k3);
				atomicAdd(&dL_dopacity[id], op);
			}
		}
	}
}

hipError_t launch_blend_backward(hipStream_t st, const BlendBwdArgs& a, const uint32_t* gate)
{
	const int ntiles = a.gx * a.gy;
	if (ntiles == 0 || a.C == 0) return hipSuccess;
	if (a.C <= 4) {
		hipLaunchKernelGGL((blend_bwd_kernel<4>), dim3(ntiles), dim3(256), 0, st, a.ranges,
				   a.point_list, a.bg, a.means2D, a.conic_opacity, a.colors, a.final_T,
				   a.n_contrib, a.dL_dpix, a.dL_dmean2D, a.dL_dconic, a.dL_dopacity,
				   a.dL_dcolors, a.W, a.H, a.C, a.gx, 1, gate);
	} else {
		const int nch = (a.C + 31) / 32;
		hipLaunchKernelGGL((blend_bwd_kernel<32>), dim3(ntiles * nch), dim3(256), 0, st, a.ranges,
				   a.point_list, a.bg, a.means2D, a.conic_opacity, a.colors, a.final_T,
				   a.n_contrib, a.dL_dpix, a.dL_dmean2D, a.dL_dconic, a.dL_dopacity,
				   a.dL_dcolors, a.W, a.H, a.C, a.gx, nch, gate);
	}
	return hipGetLastError();
}

// -------------------------------------------------------------------------------------
// Per-Gaussian backward: conic -> cov2D -> cov3D & mean; mean2D -> mean3D; SH; scale/rot.
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
	int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii,
	const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
	const float* __restrict__ scales, const float* __restrict__ rotations, float mod,
	const float* __restrict__ cov3Ds, const float* __restrict__ view,
	const float* __restrict__ proj, float fx, float fy, float tanx, float tany,
	const float* __restrict__ campos, const float* __restrict__ dL_dmean2D,
	const float* __restrict__ dL_dconic, float* __restrict__ dL_dmeans,
	const float* __restrict__ dL_dcolor, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
	float* __restrict__ dL_dscale, float* __restrict__ dL_drot)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= P || !(radii[i] > 0)) return;
	const float mx = means3D[3 * (size_t)i], my = means3D[3 * (size_t)i + 1],
		    mz = means3D[3 * (size_t)i + 2];
	float cov3D[6];
#pragma unroll
	for (int k = 0; k < 6; k++) cov3D[k] = cov3Ds[6 * (size_t)i + k];

	float dmean[3];
	float dcov[6];
	// ---- computeCov2DCUDA (backward.cu:141-271)
	{
		const float dcx = dL_dconic[4 * (size_t)i], dcy = dL_dconic[4 * (size_t)i + 1],
			    dcz = dL_dconic[4 * (size_t)i + 3];
		const Cov2D c2 = cov2d_parts(mx, my, mz, fx, fy, tanx, tany, cov3D, view);
		const float limx = 1.3f * tanx, limy = 1.3f * tany;
		const float x_grad_mul = (c2.txtz < -limx || c2.txtz > limx) ? 0.f : 1.f;
		const float y_grad_mul = (c2.tytz < -limy || c2.tytz > limy) ? 0.f : 1.f;
		const float a = c2.a, b = c2.b, c = c2.c;
		const float denom = a * c - b * b;
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
		const float(*T)[3] = c2.T;
		if (denom2inv != 0) {
			dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
			dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
			dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
			dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
			dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
			dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
			dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
			dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
			dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
		} else {
#pragma unroll
			for (int k = 0; k < 6; k++) dcov[k] = 0;
		}
#pragma unroll
		for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = dcov[k];
		const float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]},
				       {cov3D[1], cov3D[3], cov3D[4]},
				       {cov3D[2], cov3D[4], cov3D[5]}};
		float dT[2][3];
#pragma unroll
		for (int k = 0; k < 3; k++) {
			dT[0][k] = 2 * (T[0][0] * V[k][0] + T[0][1] * V[k][1] + T[0][2] * V[k][2]) * dL_da +
				   (T[1][0] * V[k][0] + T[1][1] * V[k][1] + T[1][2] * V[k][2]) * dL_db;
			dT[1][k] = 2 * (T[1][0] * V[k][0] + T[1][1] * V[k][1] + T[1][2] * V[k][2]) * dL_dc +
				   (T[0][0] * V[k][0] + T[0][1] * V[k][1] + T[0][2] * V[k][2]) * dL_db;
		}
#define SGS_WG(i_, j_) view[4 * (j_) + (i_)]   // glm W[i][j]: column i, row j
		const float dJ00 = SGS_WG(0, 0) * dT[0][0] + SGS_WG(0, 1) * dT[0][1] + SGS_WG(0, 2) * dT[0][2];
		const float dJ02 = SGS_WG(2, 0) * dT[0][0] + SGS_WG(2, 1) * dT[0][1] + SGS_WG(2, 2) * dT[0][2];
		const float dJ11 = SGS_WG(1, 0) * dT[1][0] + SGS_WG(1, 1) * dT[1][1] + SGS_WG(1, 2) * dT[1][2];
		const float dJ12 = SGS_WG(2, 0) * dT[1][0] + SGS_WG(2, 1) * dT[1][1] + SGS_WG(2, 2) * dT[1][2];
#undef SGS_WG
		const float tz = 1.f / c2.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
		const float dtx = x_grad_mul * -fx * tz2 * dJ02;
		const float dty = y_grad_mul * -fy * tz2 * dJ12;
		const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * c2.t[0]) * tz3 * dJ02 +
				  (2 * fy * c2.t[1]) * tz3 * dJ12;
		dmean[0] = view[0] * dtx + view[1] * dty + view[2] * dtz;
		dmean[1] = view[4] * dtx + view[5] * dty + view[6] * dtz;
		dmean[2] = view[8] * dtx + view[9] * dty + view[10] * dtz;
	}
	// ---- preprocessCUDA backward (backward.cu:365-382)
	{
		const f4 mh = xf4x4(proj, mx, my, mz);
		const float m_w = 1.0f / (mh.w + 0.0000001f);
		const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
		const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
		const float gx_ = dL_dmean2D[3 * (size_t)i], gy_ = dL_dmean2D[3 * (size_t)i + 1];
		dmean[0] += (proj[0] * m_w - proj[3] * mul1) * gx_ + (proj[1] * m_w - proj[3] * mul2) * gy_;
		dmean[1] += (proj[4] * m_w - proj[7] * mul1) * gx_ + (proj[5] * m_w - proj[7] * mul2) * gy_;
		dmean[2] += (proj[8] * m_w - proj[11] * mul1) * gx_ + (proj[9] * m_w - proj[11] * mul2) * gy_;
	}
	if (shs) {
		// ---- computeColorFromSH backward (backward.cu:20-136)
		const float* __restrict__ sh = shs + (size_t)i * M * 3;
		float* __restrict__ dsh = dL_dsh + (size_t)i * M * 3;
		const float dox = mx - campos[0], doy = my - campos[1], doz = mz - campos[2];
		const float len = sqrtf(dox * dox + doy * doy + doz * doz);
		const float x = dox / len, y = doy / len, z = doz / len;
		float dRGB[3];
#pragma unroll
		for (int c = 0; c < 3; c++)
			dRGB[c] = dL_dcolor[3 * (size_t)i + c] * (clamped[3 * (size_t)i + c] ? 0.f : 1.f);
		float dRdx[3] = {0, 0, 0}, dRdy[3] = {0, 0, 0}, dRdz[3] = {0, 0, 0};
#define SGS_S(k, c) sh[3 * (k) + (c)]
#define SGS_DS(k, v)                                       \
	{                                                  \
		const float v_ = (v);                      \
		for (int c = 0; c < 3; c++) dsh[3 * (k) + c] = v_ * dRGB[c]; \
	}
		SGS_DS(0, SH_C0);
		if (D > 0) {
			SGS_DS(1, -SH_C1 * y);
			SGS_DS(2, SH_C1 * z);
			SGS_DS(3, -SH_C1 * x);
#pragma unroll
			for (int c = 0; c < 3; c++) {
				dRdx[c] = -SH_C1 * SGS_S(3, c);
				dRdy[c] = -SH_C1 * SGS_S(1, c);
				dRdz[c] = SH_C1 * SGS_S(2, c);
			}
			if (D > 1) {
				const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
				SGS_DS(4, SH_C2[0] * xy);
				SGS_DS(5, SH_C2[1] * yz);
				SGS_DS(6, SH_C2[2] * (2.f * zz - xx - yy));
				SGS_DS(7, SH_C2[3] * xz);
				SGS_DS(8, SH_C2[4] * (xx - yy));
#pragma unroll
				for (int c = 0; c < 3; c++) {
					dRdx[c] += SH_C2[0] * y * SGS_S(4, c) + SH_C2[2] * 2.f * -x * SGS_S(6, c) + SH_C2[3] * z * SGS_S(7, c) + SH_C2[4] * 2.f * x * SGS_S(8, c);
					dRdy[c] += SH_C2[0] * x * SGS_S(4, c) + SH_C2[1] * z * SGS_S(5, c) + SH_C2[2] * 2.f * -y * SGS_S(6, c) + SH_C2[4] * 2.f * -y * SGS_S(8, c);
					dRdz[c] += SH_C2[1] * y * SGS_S(5, c) + SH_C2[2] * 2.f * 2.f * z * SGS_S(6, c) + SH_C2[3] * x * SGS_S(7, c);
				}
				if (D > 2) {
					SGS_DS(9, SH_C3[0] * y * (3.f * xx - yy));
					SGS_DS(10, SH_C3[1] * xy * z);
					SGS_DS(11, SH_C3[2] * y * (4.f * zz - xx - yy));
					SGS_DS(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
					SGS_DS(13, SH_C3[4] * x * (4.f * zz - xx - yy));
					SGS_DS(14, SH_C3[5] * z * (xx - yy));
					SGS_DS(15, SH_C3[6] * x * (xx - 3.f * yy));
#pragma unroll
					for (int c = 0; c < 3; c++) {
						dRdx[c] += (SH_C3[0] * SGS_S(9, c) * 3.f * 2.f * xy + SH_C3[1] * SGS_S(10, c) * yz +
							    SH_C3[2] * SGS_S(11, c) * -2.f * xy + SH_C3[3] * SGS_S(12, c) * -3.f * 2.f * xz +
							    SH_C3[4] * SGS_S(13, c) * (-3.f * xx + 4.f * zz - yy) +
							    SH_C3[5] * SGS_S(14, c) * 2.f * xz + SH_C3[6] * SGS_S(15, c) * 3.f * (xx - yy));
						dRdy[c] += (SH_C3[0] * SGS_S(9, c) * 3.f * (xx - yy) + SH_C3[1] * SGS_S(10, c) * xz +
							    SH_C3[2] * SGS_S(11, c) * (-3.f * yy + 4.f * zz - xx) +
							    SH_C3[3] * SGS_S(12, c) * -3.f * 2.f * yz + SH_C3[4] * SGS_S(13, c) * -2.f * xy +
							    SH_C3[5] * SGS_S(14, c) * -2.f * yz + SH_C3[6] * SGS_S(15, c) * -3.f * 2.f * xy);
						dRdz[c] += (SH_C3[1] * SGS_S(10, c) * xy + SH_C3[2] * SGS_S(11, c) * 4.f * 2.f * yz +
							    SH_C3[3] * SGS_S(12, c) * 3.f * (2.f * zz - xx - yy) +
							    SH_C3[4] * SGS_S(13, c) * 4.f * 2.f * xz + SH_C3[5] * SGS_S(14, c) * (xx - yy));
					}
				}
			}
		}
#undef SGS_S
#undef SGS_DS
		const float ddx = dRdx[0] * dRGB[0] + dRdx[1] * dRGB[1] + dRdx[2] * dRGB[2];
		const float ddy = dRdy[0] * dRGB[0] + dRdy[1] * dRGB[1] + dRdy[2] * dRGB[2];
		const float ddz = dRdz[0] * dRGB[0] + dRdz[1] * dRGB[1] + dRdz[2] * dRGB[2];
		// dnormvdv (auxiliary.h:107-117)
		const float sum2 = dox * dox + doy * doy + doz * doz;
		const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
		dmean[0] += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
		dmean[1] += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
		dmean[2] += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
	}
#pragma unroll
	for (int k = 0; k < 3; k++) dL_dmeans[3 * (size_t)i + k] = dmean[k];

	if (scales) {
		// ---- computeCov3D backward (backward.cu:275-336)
		const float qr = rotations[4 * (size_t)i], qx = rotations[4 * (size_t)i + 1],
			    qy = rotations[4 * (size_t)i + 2], qz = rotations[4 * (size_t)i + 3];
		float R[3][3];
		rot_matrix(qr, qx, qy, qz, R);
		const float s[3] = {mod * scales[3 * (size_t)i], mod * scales[3 * (size_t)i + 1],
				    mod * scales[3 * (size_t)i + 2]};
		float Mm[3][3];
#pragma unroll
		for (int r = 0; r < 3; r++)
#pragma unroll
			for (int c = 0; c < 3; c++) Mm[r][c] = s[r] * R[c][r];
		const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
					{0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
					{0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
		float dM[3][3];
#pragma unroll
		for (int r = 0; r < 3; r++)
#pragma unroll
			for (int c = 0; c < 3; c++)
				dM[r][c] = (2.0f * Mm[r][0]) * dS[0][c] + (2.0f * Mm[r][1]) * dS[1][c] + (2.0f * Mm[r][2]) * dS[2][c];
#pragma unroll
		for (int k = 0; k < 3; k++)
			dL_dscale[3 * (size_t)i + k] = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
		float dMt[3][3];
#pragma unroll
		for (int a = 0; a < 3; a++)
#pragma unroll
			for (int b = 0; b < 3; b++) dMt[a][b] = dM[a][b] * s[a];
		const float r = qr, x = qx, y = qy, z = qz;
		dL_drot[4 * (size_t)i + 0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
		dL_drot[4 * (size_t)i + 1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
		dL_drot[4 * (size_t)i + 2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
		dL_drot[4 * (size_t)i + 3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
	}
}

void launch_preprocess_bwd(hipStream_t st, int P, int D, int M, const float* means3D,
			   const int* radii, const float* shs, const uint8_t* clamped,
			   const float* scales, const float* rotations, float mod,
			   const float* cov3Ds, const float* view, const float* proj, float fx,
			   float fy, float tanx, float tany, const float* campos,
			   const float* dL_dmean2D, const float* dL_dconic, float* dL_dmeans,
			   const float* dL_dcolor, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
			   float* dL_drot)
{
	hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, D, M,
			   means3D, radii, shs, clamped, scales, rotations, mod, cov3Ds, view, proj, fx,
			   fy, tanx, tany, campos, dL_dmean2D, dL_dconic, dL_dmeans, dL_dcolor, dL_dcov3D,
			   dL_dsh, dL_dscale, dL_drot);
}

} // namespace sgs
