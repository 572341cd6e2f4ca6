// preprocess_bwd.hip -- per-Gaussian backward: from the blend's gradients (dL/dconic, dL/dmean2D in NDC
// units, dL/dcolour) to the raw inputs (mean, scale, quaternion, SH coefficients, or a precomputed Sigma3).
//
// Same function as CR/cuda_rasterizer/backward.cu:20-391 (computeColorFromSH / computeCov2DCUDA /
// computeCov3D / preprocessCUDA backward), derived here from the matrix calculus of the forward model and
// written with small dense-matrix helpers rather than expanded scalars.  With G_X = dL/dX ("every entry of X
// independent", so a symmetric X gets a symmetric G_X whose off-diagonals each carry half the parameter's
// derivative):
//
//   forward                               backward
//   t = Rw m + tv                         G_m  = Rw^T G_t
//   (u,v) = t.xy (or a clamped const)     G_t.xy gated off where the frustum clamp is active
//   J = [[fx/tz, 0, -fx u/tz^2],          G_u = -fx/tz^2 G_J02,  G_v = -fy/tz^2 G_J12,
//        [0, fy/tz, -fy v/tz^2]]          G_tz = -(fx G_J00 + fy G_J11)/tz^2 + 2 (fx u G_J02 + fy v G_J12)/tz^3
//   A = J Rw                              G_J  = G_A Rw^T
//   S2 = A V A^T + 0.3 I                  G_V  = A^T G_S A,   G_A = 2 G_S A V
//   K  = S2^-1 = adj(S2)/det              G_S  = -K G_K K = -adj G_K adj / det^2   (1/(det^2 + 1e-7) as the reference guards it)
//   V  = M M^T, M = R(q) diag(s)          G_M  = 2 G_V M,  G_s[k] = sum_i G_M[i][k] R[i][k],  G_R = G_M diag(s)
//   R(q) = I + 2 r [v]x + 2 [v]x^2        G_r  = 2 v.w,  G_v = 2 (r w + (G_R + G_R^T) v - 2 tr(G_R) v),  w = vee(G_R - G_R^T)
//   ndc = (Pm m).xy / ((Pm m).w + 1e-7)   G_m += sum_k g_k (Pm[k] - ndc_k Pm[3]) / w'
//   rgb = sum_n Y_n(d) sh_n + 0.5, d = (m - cam)/|m - cam|
//                                         G_sh_n = Y_n(d) G_rgb (masked where rgb was clamped at 0),
//                                         G_m += (I - d d^T)/|m - cam| * sum_n grad Y_n(d) (sh_n . G_rgb)
//
// Quirks of the reference kept on purpose (pinned by tests/test_ref_splat.py against float64 autograd):
// the frustum clamp stops the gradient of the clamped coordinate entirely (backward.cu:172-173); the returned
// scale gradient is with respect to mod*scale (backward.cu:311-318 never multiplies by `mod`); the quaternion is
// used as given.  The SH polynomials and their derivatives come from a generated monomial table
// (sh_poly_table.h, tools/gen_sh_table.py).
#include "sgs_kernels.h"
#include "sh_poly_table.h"

namespace sgs {

namespace {

struct Mat3 {
	float m[3][3];
};

__device__ __forceinline__ Mat3 sym_from6(const float* __restrict__ c, float off_scale)
{
	Mat3 V;
	V.m[0][0] = c[0];
	V.m[1][1] = c[3];
	V.m[2][2] = c[5];
	V.m[0][1] = V.m[1][0] = off_scale * c[1];
	V.m[0][2] = V.m[2][0] = off_scale * c[2];
	V.m[1][2] = V.m[2][1] = off_scale * c[4];
	return V;
}

// (sh_eval: sgs_device.h -- the forward colour uses the same table)
template <int DEG>
__device__ __forceinline__ void sh_backward(const float* __restrict__ sh, float* __restrict__ dsh, const float d[3],
					    const float grgb[3], float gdir[3])
{
	float px[4], py[4], pz[4];
	px[0] = py[0] = pz[0] = 1.f;
#pragma unroll
	for (int e = 1; e < 4; e++) {
		px[e] = px[e - 1] * d[0];
		py[e] = py[e - 1] * d[1];
		pz[e] = pz[e - 1] * d[2];
	}
	gdir[0] = gdir[1] = gdir[2] = 0.f;
#pragma unroll
	for (int n = 0; n < (DEG + 1) * (DEG + 1); n++) {
		const float y = sh_eval(n, 0, px, py, pz);
		float proj = 0.f;   // sh_n . G_rgb
#pragma unroll
		for (int c = 0; c < 3; c++) {
			dsh[3 * n + c] = y * grgb[c];
			proj += sh[3 * n + c] * grgb[c];
		}
#pragma unroll
		for (int a = 0; a < 3; a++) gdir[a] += sh_eval(n, 1 + a, px, py, pz) * proj;
	}
}

} // namespace

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
	if (i >= P || !(radii[i] > 0)) return;   // only rendered Gaussians (backward.cu:153,362)
	const size_t gi = (size_t)i;
	const float m[3] = {means3D[3 * gi], means3D[3 * gi + 1], means3D[3 * gi + 2]};
	const Mat3 V = sym_from6(cov3Ds + 6 * gi, 1.f);

	// world -> view rotation as a matrix acting on column vectors: Rw[k][c] = view[4 c + k]
	float Rw[3][3], t[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		t[k] = view[12 + k];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			Rw[k][c] = view[4 * c + k];
			t[k] += Rw[k][c] * m[c];
		}
	}
	// frustum clamp of the two lateral coordinates; `gate` = 1 where the coordinate is live
	const float lim[2] = {1.3f * tanx, 1.3f * tany};
	const float foc[2] = {fx, fy};
	const float itz = 1.f / t[2];
	float uv[2], gate[2];
#pragma unroll
	for (int a = 0; a < 2; a++) {
		const float ratio = t[a] * itz;
		gate[a] = (ratio < -lim[a] || ratio > lim[a]) ? 0.f : 1.f;
		uv[a] = fmin_(lim[a], fmax_(-lim[a], ratio)) * t[2];
	}
	// A = J Rw, row a of J is foc[a]/tz * e_a - foc[a] uv[a]/tz^2 * e_z
	float A[2][3];
#pragma unroll
	for (int a = 0; a < 2; a++) {
		const float ja = foc[a] * itz, jz = -foc[a] * uv[a] * itz * itz;
#pragma unroll
		for (int c = 0; c < 3; c++) A[a][c] = ja * Rw[a][c] + jz * Rw[2][c];
	}
	float AV[2][3];   // A V
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int c = 0; c < 3; c++) AV[a][c] = A[a][0] * V.m[0][c] + A[a][1] * V.m[1][c] + A[a][2] * V.m[2][c];
	float S[2][2];    // low-pass filtered 2D covariance
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int b = 0; b < 2; b++)
			S[a][b] = AV[a][0] * A[b][0] + AV[a][1] * A[b][1] + AV[a][2] * A[b][2] + (a == b ? 0.3f : 0.f);
	const float det = S[0][0] * S[1][1] - S[0][1] * S[0][1];
	const float rho = 1.0f / (det * det + 0.0000001f);

	// G_S = -rho * adj G_K adj with the blend's symmetric G_K (its off-diagonal slot is already the half)
	const float GK[2][2] = {{dL_dconic[4 * gi], dL_dconic[4 * gi + 1]}, {dL_dconic[4 * gi + 1], dL_dconic[4 * gi + 3]}};
	const float adj[2][2] = {{S[1][1], -S[0][1]}, {-S[0][1], S[0][0]}};
	float GS[2][2];
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int b = 0; b < 2; b++) {
			float acc = 0.f;
#pragma unroll
			for (int p = 0; p < 2; p++)
#pragma unroll
				for (int q = 0; q < 2; q++) acc += adj[a][p] * GK[p][q] * adj[q][b];
			GS[a][b] = -rho * acc;
		}

	// G_V = A^T G_S A; the 6-vector holds each symmetric pair once -> off-diagonals doubled
	float GSA[2][3];
#pragma unroll
	for (int a = 0; a < 2; a++)
#pragma unroll
		for (int c = 0; c < 3; c++) GSA[a][c] = GS[a][0] * A[0][c] + GS[a][1] * A[1][c];
	Mat3 GV;
#pragma unroll
	for (int r = 0; r < 3; r++)
#pragma unroll
		for (int c = 0; c < 3; c++) GV.m[r][c] = A[0][r] * GSA[0][c] + A[1][r] * GSA[1][c];
	{
		float* o = dL_dcov3D + 6 * gi;
		o[0] = GV.m[0][0];
		o[1] = 2.f * GV.m[0][1];
		o[2] = 2.f * GV.m[0][2];
		o[3] = GV.m[1][1];
		o[4] = 2.f * GV.m[1][2];
		o[5] = GV.m[2][2];
	}

	// G_A = 2 G_S (A V);  G_J = G_A Rw^T (only the four live entries of J matter)
	float GJ[2][3];
#pragma unroll
	for (int a = 0; a < 2; a++) {
		float GA[3];
#pragma unroll
		for (int c = 0; c < 3; c++) GA[c] = 2.f * (GS[a][0] * AV[0][c] + GS[a][1] * AV[1][c]);
#pragma unroll
		for (int k = 0; k < 3; k++) GJ[a][k] = GA[0] * Rw[k][0] + GA[1] * Rw[k][1] + GA[2] * Rw[k][2];
	}
	float Gt[3];
	Gt[2] = 0.f;
#pragma unroll
	for (int a = 0; a < 2; a++) {
		Gt[a] = gate[a] * (-foc[a] * itz * itz) * GJ[a][2];
		Gt[2] += -foc[a] * itz * itz * GJ[a][a] + 2.f * foc[a] * uv[a] * itz * itz * itz * GJ[a][2];
	}
	float Gm[3];
#pragma unroll
	for (int c = 0; c < 3; c++) Gm[c] = Rw[0][c] * Gt[0] + Rw[1][c] * Gt[1] + Rw[2][c] * Gt[2];

	// screen position: ndc_k = h_k / (h_w + 1e-7), h = Pm [m;1] with Pm[k][c] = proj[4 c + k]
	{
		float h[4];
#pragma unroll
		for (int k = 0; k < 4; k++) h[k] = proj[k] * m[0] + proj[4 + k] * m[1] + proj[8 + k] * m[2] + proj[12 + k];
		const float iw = 1.0f / (h[3] + 0.0000001f);
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const float g = dL_dmean2D[3 * gi + k] * iw, ndc = h[k] * iw;
#pragma unroll
			for (int c = 0; c < 3; c++) Gm[c] += g * (proj[4 * c + k] - ndc * proj[4 * c + 3]);
		}
	}

	if (shs) {
		float v[3], d[3], grgb[3], gdir[3];
#pragma unroll
		for (int a = 0; a < 3; a++) v[a] = m[a] - campos[a];
		const float len2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
		const float ilen = 1.0f / sqrtf(len2);
#pragma unroll
		for (int a = 0; a < 3; a++) {
			d[a] = v[a] * ilen;
			grgb[a] = clamped[3 * gi + a] ? 0.f : dL_dcolor[3 * gi + a];
		}
		const float* sh = shs + gi * M * 3;
		float* dsh = dL_dsh + gi * M * 3;
		switch (D) {
		case 0: sh_backward<0>(sh, dsh, d, grgb, gdir); break;
		case 1: sh_backward<1>(sh, dsh, d, grgb, gdir); break;
		case 2: sh_backward<2>(sh, dsh, d, grgb, gdir); break;
		default: sh_backward<3>(sh, dsh, d, grgb, gdir); break;
		}
		// through d = v/|v|: (I - d d^T)/|v|
		const float radial = d[0] * gdir[0] + d[1] * gdir[1] + d[2] * gdir[2];
#pragma unroll
		for (int a = 0; a < 3; a++) Gm[a] += (gdir[a] - radial * d[a]) * ilen;
	}
#pragma unroll
	for (int c = 0; c < 3; c++) dL_dmeans[3 * gi + c] = Gm[c];

	if (scales) {
		// V = M M^T with M = R diag(s): G_M = 2 G_V M (G_V with halved off-diagonals = the matrix gradient)
		const Mat3& GVm = GV;
		const float qr = rotations[4 * gi], qv[3] = {rotations[4 * gi + 1], rotations[4 * gi + 2], rotations[4 * gi + 3]};
		float R[3][3];
		rot_matrix(qr, qv[0], qv[1], qv[2], R);
		const float s[3] = {mod * scales[3 * gi], mod * scales[3 * gi + 1], mod * scales[3 * gi + 2]};
		float GR[3][3], Gs[3] = {0.f, 0.f, 0.f};
#pragma unroll
		for (int r = 0; r < 3; r++)
#pragma unroll
			for (int k = 0; k < 3; k++) {
				float gm = 0.f;   // G_M[r][k] = 2 sum_c G_V[r][c] R[c][k] s[k]
#pragma unroll
				for (int c = 0; c < 3; c++) gm += GVm.m[r][c] * R[c][k];
				gm *= 2.f * s[k];
				Gs[k] += gm * R[r][k];
				GR[r][k] = gm * s[k];
			}
#pragma unroll
		for (int k = 0; k < 3; k++) dL_dscale[3 * gi + k] = Gs[k];
		// quaternion: antisymmetric part w, symmetric part Sy, trace
		const float w[3] = {GR[2][1] - GR[1][2], GR[0][2] - GR[2][0], GR[1][0] - GR[0][1]};
		const float tr = GR[0][0] + GR[1][1] + GR[2][2];
		dL_drot[4 * gi] = 2.f * (qv[0] * w[0] + qv[1] * w[1] + qv[2] * w[2]);
#pragma unroll
		for (int a = 0; a < 3; a++) {
			float sv = 0.f;
#pragma unroll
			for (int b = 0; b < 3; b++) sv += (GR[a][b] + GR[b][a]) * qv[b];
			dL_drot[4 * gi + 1 + a] = 2.f * (qr * w[a] + sv - 2.f * tr * qv[a]);
		}
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
