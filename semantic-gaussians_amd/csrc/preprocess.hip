// preprocess.hip -- per-Gaussian forward preprocess and frustum marking (gfx950).
//
// One lane per Gaussian, 256-lane workgroups (4 wave64).  The work is ~100 flops and
// <= 123 B of traffic per Gaussian, i.e. HBM-trivial (0.1 GB at 1M Gaussians); the only
// requirement here is bit-exact integer outputs (radii, tiles_touched, depth bits), so
// the arithmetic follows sgs_device.h's contract with contraction disabled.
//
// Behaviour restated from CR/cuda_rasterizer/forward.cu:155-256 (preprocessCUDA),
// auxiliary.h:139-164 (in_frustum) and rasterizer_impl.cu:54-66 (checkFrustum).
#include "sgs_kernels.h"

namespace sgs {

// View-dependent colour of one Gaussian from its (M, 3) SH coefficients: colour_c = sum_n Y_n(d) sh[n][c] + 0.5 with d the
// unit vector from the camera centre to the Gaussian, negative results clamped to 0 and flagged for the backward (what
// forward.cu:20-71 computes).  Y_n comes from the generated monomial table (sgs_device.h sh_eval), summed n ascending: the
// oracle evaluates the same table in the same order, so the colours agree bit for bit.
template <int DEG>
__device__ __forceinline__ void sh_colour(float mx, float my, float mz, const float* __restrict__ campos,
					  const float* __restrict__ sh, float* __restrict__ colour,
					  uint8_t* __restrict__ clamped)
{
	constexpr int NB = (DEG + 1) * (DEG + 1);
	const float vx = mx - campos[0], vy = my - campos[1], vz = mz - campos[2];
	const float len = sqrtf(vx * vx + vy * vy + vz * vz);
	float px[4], py[4], pz[4];
	px[0] = py[0] = pz[0] = 1.f;
	px[1] = vx / len;
	py[1] = vy / len;
	pz[1] = vz / len;
#pragma unroll
	for (int e = 2; e < 4; e++) {
		px[e] = px[e - 1] * px[1];
		py[e] = py[e - 1] * py[1];
		pz[e] = pz[e - 1] * pz[1];
	}
	float Y[NB];
#pragma unroll
	for (int n = 0; n < NB; n++) Y[n] = sh_eval(n, 0, px, py, pz);
#pragma unroll
	for (int c = 0; c < 3; c++) {
		float r = 0.f;
#pragma unroll
		for (int n = 0; n < NB; n++) r += Y[n] * sh[3 * n + c];
		r += 0.5f;
		clamped[c] = (r < 0) ? 1 : 0;
		colour[c] = fmax_(r, 0.0f);
	}
}

__global__ __launch_bounds__(256) void preprocess_fwd_kernel(
	int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
	float mod, const float* __restrict__ rotations, const float* __restrict__ opacities,
	const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
	const float* __restrict__ colors_precomp, const float* __restrict__ view,
	const float* __restrict__ proj, const float* __restrict__ campos, int W, int H, float tanx,
	float tany, float fx, float fy, int gx, int gy, int prefiltered, int num_channels,
	int* __restrict__ radii, float2* __restrict__ means2D, float* __restrict__ depths,
	float* __restrict__ cov3Ds, float* __restrict__ rgb, uint8_t* __restrict__ clamped,
	float4* __restrict__ conic_opacity, uint32_t* __restrict__ tiles_touched,
	int* __restrict__ trap_flag, uint32_t* __restrict__ ds_cnt0, uint32_t* __restrict__ ds_gcnt0)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	uint32_t sort_key = 0xFFFFFFFFu;   // what depths[i] ends up holding
	if (i < P) do {
	radii[i] = 0;
	tiles_touched[i] = 0;
	// depth doubles as the presort key (binning mode 0): culled Gaussians sort last
	depths[i] = __uint_as_float(0xFFFFFFFFu);

	const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1],
		    pz = means3D[3 * (size_t)i + 2];
	const f4 ph = xf4x4(proj, px, py, pz);
	const f3 pv = xf4x3(view, px, py, pz);
	if (pv.z <= 0.2f) {
		// reference: printf + __trap() (auxiliary.h:156-160); here: flag, the host
		// turns it into SGS_ETRAP.
		if (prefiltered) atomicOr(trap_flag, 1);
		break;
	}
	const float pw = 1.0f / (ph.w + 0.0000001f);
	const float ppx = ph.x * pw, ppy = ph.y * pw;

	float cov3D[6];
	if (cov3D_precomp) {
#pragma unroll
		for (int k = 0; k < 6; k++) cov3D[k] = cov3D_precomp[6 * (size_t)i + k];
	} else {
		cov3d_from_scale_rot(scales[3 * (size_t)i], scales[3 * (size_t)i + 1],
				     scales[3 * (size_t)i + 2], mod, rotations[4 * (size_t)i],
				     rotations[4 * (size_t)i + 1], rotations[4 * (size_t)i + 2],
				     rotations[4 * (size_t)i + 3], cov3D);
#pragma unroll
		for (int k = 0; k < 6; k++) cov3Ds[6 * (size_t)i + k] = cov3D[k];
	}
	const Cov2D c2 = cov2d_parts(px, py, pz, fx, fy, tanx, tany, cov3D, view);
	const float det = c2.a * c2.c - c2.b * c2.b;
	if (det == 0.0f) break;
	// conic = inverse of the 2x2 covariance; screen extent = 3 sigma of its larger eigenvalue (half_trace +- root, the
	// root floored at sqrt(0.1) as the reference does), rounded up to whole pixels
	const float inv_det = 1.f / det;
	const float4 conic_o = make_float4(c2.c * inv_det, -c2.b * inv_det, c2.a * inv_det, opacities[i]);
	const float half_trace = 0.5f * (c2.a + c2.c);
	const float root = sqrtf(fmax_(0.1f, half_trace * half_trace - det));
	const float extent = ceilf(3.f * sqrtf(fmax_(half_trace + root, half_trace - root)));
	const float pix_x = ndc2pix(ppx, W), pix_y = ndc2pix(ppy, H);
	uint32_t x0, y0, x1, y1;
	get_rect(pix_x, pix_y, (int)extent, gx, gy, x0, y0, x1, y1);
	if ((x1 - x0) * (y1 - y0) == 0) break;

	if (!colors_precomp) {
		const float* sh = shs + (size_t)i * M * 3;
		float* col = rgb + (size_t)i * num_channels;
		uint8_t* cl = clamped + 3 * (size_t)i;
		switch (D) {
		case 0: sh_colour<0>(px, py, pz, campos, sh, col, cl); break;
		case 1: sh_colour<1>(px, py, pz, campos, sh, col, cl); break;
		case 2: sh_colour<2>(px, py, pz, campos, sh, col, cl); break;
		default: sh_colour<3>(px, py, pz, campos, sh, col, cl); break;
		}
	}

	depths[i] = pv.z;
	sort_key = __float_as_uint(pv.z);
	radii[i] = (int)extent;
	means2D[i] = make_float2(pix_x, pix_y);
	conic_opacity[i] = conic_o;
	tiles_touched[i] = (y1 - y0) * (x1 - x0);
	} while (0);

	// depth_sort.hip's pass-0 count matrices: keys per (tile of 4096 Gaussians, lowest key byte), aggregated per
	// workgroup in LDS (every Gaussian counts, the culled ones with their 0xFFFFFFFF key)
	if (ds_cnt0) {
		__shared__ uint32_t s_h[256];
		s_h[threadIdx.x] = 0u;
		__syncthreads();
		if (i < P) atomicAdd(&s_h[sort_key & 255u], 1u);
		__syncthreads();
		const uint32_t c = s_h[threadIdx.x];
		if (c) {
			const uint32_t tile = blockIdx.x / (DS_TILE / 256);
			atomicAdd(&ds_cnt0[(size_t)tile * 256 + threadIdx.x], c);
			atomicAdd(&ds_gcnt0[(size_t)(tile / DS_GRP) * 256 + threadIdx.x], c);
		}
	}
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P,
							    const float* __restrict__ means3D,
							    const float* __restrict__ view,
							    uint8_t* __restrict__ present)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= P) return;
	const f3 pv = xf4x3(view, means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1],
			    means3D[3 * (size_t)i + 2]);
	present[i] = !(pv.z <= 0.2f);
}

void launch_preprocess_fwd(hipStream_t st, int P, int D, int M, const float* means3D,
			   const float* scales, float mod, const float* rotations,
			   const float* opacities, const float* shs, const float* cov3D_precomp,
			   const float* colors_precomp, const float* view, const float* proj,
			   const float* campos, int W, int H, float tanx, float tany, float fx,
			   float fy, int gx, int gy, int prefiltered, int num_channels, int* radii,
			   float2* means2D, float* depths, float* cov3Ds, float* rgb,
			   uint8_t* clamped, float4* conic_opacity, uint32_t* tiles_touched,
			   int* trap_flag, uint32_t* ds_cnt0, uint32_t* ds_gcnt0)
{
	hipLaunchKernelGGL(preprocess_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, D, M,
			   means3D, scales, mod, rotations, opacities, shs, cov3D_precomp,
			   colors_precomp, view, proj, campos, W, H, tanx, tany, fx, fy, gx, gy,
			   prefiltered, num_channels, radii, means2D, depths, cov3Ds, rgb, clamped,
			   conic_opacity, tiles_touched, trap_flag, ds_cnt0, ds_gcnt0);
}

void launch_mark_visible(hipStream_t st, int P, const float* means3D, const float* view,
			 uint8_t* present)
{
	hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D,
			   view, present);
}

} // namespace sgs
