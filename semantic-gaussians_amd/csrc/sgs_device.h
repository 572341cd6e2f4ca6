// sgs_device.h -- device-side arithmetic contract shared by all gfx950 kernels.
//
// Everything here is evaluated in fp32, left to right as written, with no implicit FMA
// contraction (the build passes -ffp-contract=off); fmaf() appears only where the
// contract (DESIGN.md "arithmetic contract") says "fma".  The CPU oracle restates the
// same formulas independently (oracle/sgs_oracle.c); integer outputs (radii, tiles
// touched, keys, sorted lists, n_contrib) must agree bit for bit.
//
// Reference semantics restated (not copied): CR/cuda_rasterizer/auxiliary.h:41-164,
// forward.cu:20-151 (CR = submodules/channel-rasterization of the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SGS_TILE 16
#define SGS_TILE_PX 256

#include "sh_poly_table.h"

namespace sgs {

// Streaming stores (the `nt` hint) for data that is written once and read by a LATER kernel: the work list's weight rows.
// Measured (profiles/r04_worklist_nt_stores.txt): the weights pre-pass 0.262 -> 0.229 ms at cfg3, the sweep that reads them unchanged.
typedef uint32_t sgs_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(uint4* p, uint4 v)
{
	__builtin_nontemporal_store(sgs_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<sgs_u32x4*>(p));
}
template <typename T> __device__ __forceinline__ void store_nt(T* p, T v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ float fmin_(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float fmax_(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

// out_k = m[k]x + m[4+k]y + m[8+k]z + m[12+k]   (auxiliary.h:58-77)
__device__ __forceinline__ f3 xf4x3(const float* __restrict__ m, float x, float y, float z)
{
	f3 o;
	o.x = m[0] * x + m[4] * y + m[8] * z + m[12];
	o.y = m[1] * x + m[5] * y + m[9] * z + m[13];
	o.z = m[2] * x + m[6] * y + m[10] * z + m[14];
	return o;
}
__device__ __forceinline__ f4 xf4x4(const float* __restrict__ m, float x, float y, float z)
{
	f4 o;
	o.x = m[0] * x + m[4] * y + m[8] * z + m[12];
	o.y = m[1] * x + m[5] * y + m[9] * z + m[13];
	o.z = m[2] * x + m[6] * y + m[10] * z + m[14];
	o.w = m[3] * x + m[7] * y + m[11] * z + m[15];
	return o;
}

// auxiliary.h:41-44 -- evaluated in double, rounded once to fp32.
__device__ __forceinline__ float ndc2pix(float v, int S)
{
	return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// auxiliary.h:46-56 -- fp32 arithmetic, C cast (truncation), clamp to [0, grid].
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy,
					 uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1)
{
	const float r = (float)max_radius;
	x0 = (uint32_t)imin_(gx, imax_(0, (int)((px - r) / (float)SGS_TILE)));
	y0 = (uint32_t)imin_(gy, imax_(0, (int)((py - r) / (float)SGS_TILE)));
	x1 = (uint32_t)imin_(gx, imax_(0, (int)((px + r + (float)(SGS_TILE - 1)) / (float)SGS_TILE)));
	y1 = (uint32_t)imin_(gy, imax_(0, (int)((py + r + (float)(SGS_TILE - 1)) / (float)SGS_TILE)));
}

// wave64 sum -> valid in every lane (DPP butterflies inside rows of 16, then readlane).
__device__ __forceinline__ float wave_sum(float v)
{
	// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
	v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
	v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
	v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
	v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
	// every lane of a 16-lane row now holds the row sum
	const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
	const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
	const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
	const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
	return (r0 + r1) + (r2 + r3);
}

// Eight wave64 sums at once, as a transposed reduction: 18 VALU instead of 8 x 11.  Each step halves the number of
// lanes a partial sum is spread over AND the number of live registers: v_permlane32_swap / v_permlane16_swap exchange
// halves / 16-lane rows between two registers so that one add folds two components at a time; from 8 lanes down, DPP.
// Result: lane l with (l & 7) == 0 holds the total of component wave_sum8_component(l) (other lanes: partial sums).
__device__ __forceinline__ int wave_sum8_component(int lane) { return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2); }
// (inline asm, not __builtin_amdgcn_permlane*_swap: hipcc 7.2 folds `s[0] + s[1]` of the builtin's result pair into `s[0] + s[0]`.  The s_nop in front covers
// the VALU-write -> permlane-read distance the compiler cannot see into the asm for, the one behind the permlane-write -> VALU-read distance; the swaps of
// one level are independent of each other and share ONE pair of them -- round 6: a pair per swap was 36 idle cycles of a ~340-cycle geometry step.)
__device__ __forceinline__ float wave_sum8_tail(float s01, float s23)
{
	// 16 -> 8 lanes: lanes with bit 3 clear keep s01, the others s23; the partner's value arrives by a row rotation by 8
	const bool up = (__lane_id() & 8) != 0;
	const float keep = up ? s23 : s01, give = up ? s01 : s23;
	float u = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0x128, 0xF, 0xF, false));
	// 8 -> 1 inside each group of 8 lanes: row_half_mirror, quad reverse [3,2,1,0] = 0x1B, quad_perm [1,0,3,2] = 0xB1
	u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x141, 0xF, 0xF, false));
	u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x1B, 0xF, 0xF, false));
	u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0xB1, 0xF, 0xF, false));
	return u;
}
__device__ __forceinline__ float wave_sum8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7)
{
	// lanes < 32: the sum of the first register's halves; >= 32: of the second's
	asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\ts_nop 1"
		     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
	float r0 = v0 + v1, r1 = v2 + v3, r2 = v4 + v5, r3 = v6 + v7;
	// rows 0 / 2: the first register's row pairs; rows 1 / 3: the second's
	asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
	return wave_sum8_tail(r0 + r1, r2 + r3);
}
// six sums (components 6 and 7 of wave_sum8 zero): one swap and two moves fewer
__device__ __forceinline__ float wave_sum6(float v0, float v1, float v2, float v3, float v4, float v5)
{
	asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\ts_nop 1"
		     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5));
	float r0 = v0 + v1, r1 = v2 + v3, r2 = v4 + v5, r3 = 0.f;
	asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
	return wave_sum8_tail(r0 + r1, r2 + r3);
}

// Blend exponential ("exp contract"): range reduction by the 1.5*2^23 magic
// add, two-term ln2, degree-5 Horner with explicit fma, exponent insertion by integer
// add.  <= 5 ulp from exp(); reproduced bit for bit by the oracle so that the
// alpha < 1/255 and T < 1e-4 decisions of the blend are identical on both sides.
__device__ __forceinline__ float expf_contract(float x)
{
	const float LOG2E = 1.44269504088896341f;
	const float LN2_HI = 0.693145751953125f;
	const float LN2_LO = 1.42860682030941723e-6f;
	const float MAGIC = 12582912.0f;
	x = fmax_(x, -87.0f);
	const float t = x * LOG2E;
	float nf = t + MAGIC;
	// keep the compiler from re-associating (t+M)-M
	asm volatile("" : "+v"(nf));
	const float n = nf - MAGIC;
	float r = __builtin_fmaf(n, -LN2_HI, x);
	r = __builtin_fmaf(n, -LN2_LO, r);
	float p = 0.008182921446859837f;
	p = __builtin_fmaf(p, r, 0.04184672236442566f);
	p = __builtin_fmaf(p, r, 0.16668450832366943f);
	p = __builtin_fmaf(p, r, 0.4999966621398926f);
	p = __builtin_fmaf(p, r, 1.0f);
	p = __builtin_fmaf(p, r, 1.0f);
	return __uint_as_float(__float_as_uint(p) + (__float_as_uint(nf) << 23));
}

// Rotation matrix of the quaternion (w, v) as given (not normalised, like the reference: forward.cu:127-137), written in
// cyclic form: with (i, j, k) an even permutation of (0, 1, 2)
//     R[i][i] = 1 - 2 (v_a^2 + v_b^2)   (a < b the two other axes),   R[i][j] = 2 (v_i v_j - w v_k),   R[j][i] = 2 (v_i v_j + w v_k).
// Each entry is the same fp32 expression tree as the oracle's (products commute exactly), which the bit-exact radii need.
__device__ __forceinline__ void rot_matrix(float w, float vx, float vy, float vz, float R[3][3])
{
	const float v[3] = {vx, vy, vz};
#pragma unroll
	for (int i = 0; i < 3; i++) {
		const int j = (i + 1) % 3, k = (i + 2) % 3;
		const int a = j < k ? j : k, b = j < k ? k : j;
		R[i][i] = 1.f - 2.f * (v[a] * v[a] + v[b] * v[b]);
		R[i][j] = 2.f * (v[i] * v[j] - w * v[k]);
		R[j][i] = 2.f * (v[i] * v[j] + w * v[k]);
	}
}

// Sigma = R diag(mod*s)^2 R^T, accumulated as sum_k M(k,i) M(k,j) with
// M(k,i) = (mod*s_k) * R(i,k), k ascending (forward.cu:118-151 through glm's
// column-major algebra).  cov = xx, xy, xz, yy, yz, zz.
__device__ __forceinline__ void cov3d_from_scale_rot(float sx, float sy, float sz, float mod,
						     float qr, float qx, float qy, float qz,
						     float cov[6])
{
	float R[3][3], M[3][3];
	rot_matrix(qr, qx, qy, qz, R);
	const float s[3] = {mod * sx, mod * sy, mod * sz};
#pragma unroll
	for (int k = 0; k < 3; k++)
#pragma unroll
		for (int i = 0; i < 3; i++) M[k][i] = s[k] * R[i][k];
#define SGS_SIG(i, j) (M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j])
	cov[0] = SGS_SIG(0, 0);
	cov[1] = SGS_SIG(1, 0);
	cov[2] = SGS_SIG(2, 0);
	cov[3] = SGS_SIG(1, 1);
	cov[4] = SGS_SIG(2, 1);
	cov[5] = SGS_SIG(2, 2);
#undef SGS_SIG
}

// EWA projection pieces shared by forward (forward.cu:74-113) and backward
// (backward.cu:163-196): clamped view-space mean t, T[i][j] = (J*Wr)(i,j) and the
// low-pass filtered 2D covariance (a,b,c).
struct Cov2D {
	float t[3];
	float txtz, tytz;
	float T[2][3];
	float a, b, c;
};
__device__ __forceinline__ Cov2D cov2d_parts(float mx, float my, float mz, float fx, float fy,
					     float tanx, float tany, const float cov3D[6],
					     const float* __restrict__ view)
{
	Cov2D o;
	const f3 cam = xf4x3(view, mx, my, mz);
	// screen-space guard band: the view-space mean is pulled back to 1.3 x the frustum's half extent per axis before the
	// Jacobian is formed (the unclamped ratios are kept: the backward gates on them)
	const float focal[2] = {fx, fy}, bound[2] = {1.3f * tanx, 1.3f * tany};
	const float ratio[2] = {cam.x / cam.z, cam.y / cam.z};
	const float depth = cam.z;
	float lat[2];   // clamped lateral coordinates
#pragma unroll
	for (int a = 0; a < 2; a++) lat[a] = fmin_(bound[a], fmax_(-bound[a], ratio[a])) * depth;
	o.t[0] = lat[0]; o.t[1] = lat[1]; o.t[2] = depth;
	o.txtz = ratio[0]; o.tytz = ratio[1];
	// Jacobian of the perspective map at that point: row a = (focal_a / z) e_a - (focal_a lat_a / z^2) e_z, times the
	// view rotation (column-major `view`: element (row r, col c) = view[4 c + r])
#pragma unroll
	for (int a = 0; a < 2; a++) {
		const float jd = focal[a] / depth, jz = -(focal[a] * lat[a]) / (depth * depth);
#pragma unroll
		for (int r = 0; r < 3; r++) o.T[a][r] = view[4 * r + a] * jd + view[4 * r + 2] * jz;
	}
	const float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]},
			       {cov3D[1], cov3D[3], cov3D[4]},
			       {cov3D[2], cov3D[4], cov3D[5]}};
	float X[2][3];   // T V
#pragma unroll
	for (int r = 0; r < 2; r++)
#pragma unroll
		for (int c = 0; c < 3; c++)
			X[r][c] = o.T[r][0] * V[c][0] + o.T[r][1] * V[c][1] + o.T[r][2] * V[c][2];
	o.a = X[0][0] * o.T[0][0] + X[0][1] * o.T[0][1] + X[0][2] * o.T[0][2];
	o.b = X[1][0] * o.T[0][0] + X[1][1] * o.T[0][1] + X[1][2] * o.T[0][2];
	o.c = X[1][0] * o.T[1][0] + X[1][1] * o.T[1][1] + X[1][2] * o.T[1][2];
	o.a += 0.3f;   // the one-pixel low-pass filter
	o.c += 0.3f;
	return o;
}

// Real spherical harmonics from the GENERATED monomial table (sh_poly_table.h, tools/gen_sh_table.py): value (q = 0) or
// partial derivative (q = 1..3) of basis function n at the unit vector whose powers are tabulated in px / py / pz.
// Term order = table order: acc += ((c x^i) y^j) z^k, left to right -- part of the arithmetic contract with the oracle
// (the forward colour must agree bit for bit).  n and q must be compile-time constants after unrolling.
__device__ __forceinline__ float sh_eval(int n, int q, const float px[4], const float py[4], const float pz[4])
{
	float acc = 0.f;
#pragma unroll
	for (int t = SH_RANGE[n][q][0]; t < SH_RANGE[n][q][1]; t++)
		acc += SH_TERM[t].c * px[SH_TERM[t].i] * py[SH_TERM[t].j] * pz[SH_TERM[t].k];
	return acc;
}

// Work-list chunk starts (blend_fwd_split.hip): chunk 0 of EVERY tile is pre-assigned (slots [tile * 128, tile * 128 + 128));
// only chunks >= 1 go through `table`, at index (range.x >> 7) + tile + c.  That index is collision free among
// non-empty tiles (consecutive lists satisfy floor((x+n)/128) + 1 >= floor(x/128) + ceil(n/128)); an EMPTY tile has
// range (0, 0) like the reference's, so its index would be `tile` and can coincide with an early tile's -- which is why
// chunk 0 never touches the table (round 3: an empty tile next to a short-offset tile rendered that tile's first entry).
// SGS_OPT_OUT_BANDS (include/sgs_raster.h): the feature map written BAND-major for an image-partitioned exchange between `nb` ranks -- band b holds the
// image rows [lo_b, hi_b), lo_b = min(H, 16 * (gy * b / nb)) (sgs_hip.dist.band_rows), as a contiguous (C, hi_b - lo_b, pitch) block; the bands
// follow each other in the buffer, so band b starts C * pitch * lo_b floats in.  For the tile row `ty`: its band's first tile row and row count.
__host__ __device__ __forceinline__ void sgs_band_of(int ty, int gy, int nb, int H, int& lo_tile, int& rows)
{
	int b = (int)(((long long)(ty + 1) * nb - 1) / gy);
	if (b < 0) b = 0;
	if (b > nb - 1) b = nb - 1;
	while (b > 0 && (long long)gy * b / nb > ty) b--;
	while (b < nb - 1 && (long long)gy * (b + 1) / nb <= ty) b++;
	lo_tile = (int)((long long)gy * b / nb);
	const int hi_tile = (int)((long long)gy * (b + 1) / nb);
	const int lo_row = 16 * lo_tile < H ? 16 * lo_tile : H, hi_row = 16 * hi_tile < H ? 16 * hi_tile : H;
	rows = hi_row - lo_row;
}

__device__ __forceinline__ uint32_t sgs_chunk_start(const uint32_t* __restrict__ table, uint32_t chunk_base, uint32_t tile, uint32_t ci)
{
	return ci == 0u ? tile * 128u : table[chunk_base + ci];
}

} // namespace sgs
