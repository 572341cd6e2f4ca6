// fusion_map.hip -- the 2-D -> 3-D fusion step on the device.
//
// Behaviour: PointCloudToImageMapper.compute_mapping (dataset/fusion_utils.py:30-78) and the per-view
// accumulation of fuse_one_scene (fusion.py:127-147).  The reference does both in NumPy on the host:
// per view it copies the rendered depth and all Gaussian centres to the CPU, projects them in float64,
// tests occlusion against the depth, gathers features[:, y, x] and copies the gathered (N, C) block back.
// Here: one kernel per view for the mapping (float64, the reference's operation order, so the integer
// pixel coordinates and the visibility bit are the reference's), one for the accumulation (a wave per
// visible point streams its pixel's C-vector from an (H, W, C) feature map into the (N, C) sums).
// HBM-bound integer / byte work: lane = point for the mapping (28 B in, 32 B out per point), lane = 4
// channels for the accumulation (coalesced 1 KB per wave and pass).
#include "sgs_kernels.h"

namespace sgs {

namespace {

// np.round(v).astype(int) as the reference's platform (x86-64) evaluates it: round half to even, and the
// "integer indefinite" value for NaN / inf / out of range.
__device__ __forceinline__ long long round_to_i64(double v)
{
	const double r = rint(v);
	if (!(fabs(r) < 9223372036854775808.0)) return (long long)0x8000000000000000ull;   // also NaN
	return (long long)r;
}

struct Projected {
	long long ui, vi;
	double z, dist;
	bool inside;
};

__device__ __forceinline__ Projected project_point(const float* __restrict__ coords, int i,
						    const double m[16], double fx, double fy, double cx, double cy,
						    int W, int H, int cut)
{
	const double x = (double)coords[3 * (size_t)i], y = (double)coords[3 * (size_t)i + 1],
		     z = (double)coords[3 * (size_t)i + 2];
	// (transform^T) (x, y, z, 1): four terms, left to right (fusion_utils.py:45)
	const double c0 = m[0] * x + m[4] * y + m[8] * z + m[12];
	const double c1 = m[1] * x + m[5] * y + m[9] * z + m[13];
	const double c2 = m[2] * x + m[6] * y + m[10] * z + m[14];
	Projected p;
	p.z = c2;
	p.ui = round_to_i64(c0 * fx / c2 + cx);
	p.vi = round_to_i64(c1 * fy / c2 + cy);
	const double du = (double)p.ui - (double)W / 2, dv = (double)p.vi - (double)H / 2;
	p.dist = sqrt(du * du + dv * dv);
	p.inside = p.ui >= cut && p.vi >= cut && p.ui < W - cut && p.vi < H - cut;
	return p;
}

// the view's world_view_transform as the reference holds it: 16 floats on the device (read uniformly)
struct Mat16 {
	double v[16];
	__device__ explicit Mat16(const float* __restrict__ p)
	{
#pragma unroll
		for (int k = 0; k < 16; k++) v[k] = (double)p[k];
	}
};

// depth_mode 2, pass 1: z-buffer of the points themselves ("surface", fusion_utils.py:58-62).  The
// reference loops over the points keeping the smallest z per pixel; a minimum is order independent, and
// positive doubles order like their bit patterns.
__global__ __launch_bounds__(256) void fusion_zbuffer_kernel(int N, const float* __restrict__ coords,
							      const float* __restrict__ wvt,
							      double fx, double fy, double cx, double cy, int W,
							      int H, int cut, unsigned long long* __restrict__ zbuf)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= N) return;
	const Mat16 m(wvt);
	const Projected p = project_point(coords, i, m.v, fx, fy, cx, cy, W, H, cut);
	if (p.inside && p.z > 0.2)
		atomicMin(&zbuf[(size_t)p.vi * W + (size_t)p.ui], (unsigned long long)__double_as_longlong(p.z));
}

// DEPTH: 0 none (front test), 1 float32 map (the rendered depth), 2 float64 z-buffer
template <int DEPTH>
__global__ __launch_bounds__(256) void fusion_mapping_kernel(int N, const float* __restrict__ coords,
							      const float* __restrict__ wvt,
							      double fx, double fy, double cx, double cy, int W,
							      int H, int cut, double vis_thres,
							      const void* __restrict__ depth,
							      long long* __restrict__ mapping, double* __restrict__ weight)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= N) return;
	const Mat16 m(wvt);
	const Projected p = project_point(coords, i, m.v, fx, fy, cx, cy, W, H, cut);
	bool vis = p.inside;
	if (DEPTH == 0) {
		vis = vis && p.z > 0.0;
	} else if (vis) {
		const size_t pix = (size_t)p.vi * W + (size_t)p.ui;
		// fusion_utils.py:65-67: `self.vis_thres * depth_cur` is evaluated in the depth map's OWN precision -- a
		// Python float times a float32 array is a float32 product (the rendered depth is float32), times the
		// float64 z-buffer of "surface" mode a float64 one -- and only then compared with the float64 |d - z|.
		double d, thr;
		if (DEPTH == 1) {
			const float df = static_cast<const float*>(depth)[pix];
			d = (double)df;
			thr = (double)((float)vis_thres * df);
		} else {
			d = static_cast<const double*>(depth)[pix];
			thr = vis_thres * d;
		}
		vis = fabs(d - p.z) <= thr;
	}
	mapping[3 * (size_t)i] = vis ? p.vi : 0;
	mapping[3 * (size_t)i + 1] = vis ? p.ui : 0;
	mapping[3 * (size_t)i + 2] = vis ? 1 : 0;
	weight[i] = exp(-p.dist / 10);
}

__global__ __launch_bounds__(256) void fill_u64_kernel(size_t n, unsigned long long v, unsigned long long* p)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) p[i] = v;
}

// One wave per point: feat_sum[i, :] += features[y, x, :], times[i] += 1 for visible points.
__global__ __launch_bounds__(256) void fusion_accumulate_kernel(int N, int C, const float* __restrict__ feat_hwc,
								 int W, const long long* __restrict__ mapping,
								 float* __restrict__ feat_sum, float* __restrict__ times)
{
	const int lane = threadIdx.x & 63;
	const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (i >= N) return;
	if (mapping[3 * (size_t)i + 2] == 0) return;
	const size_t pix = (size_t)mapping[3 * (size_t)i] * W + (size_t)mapping[3 * (size_t)i + 1];
	const float* __restrict__ src = feat_hwc + pix * C;
	float* __restrict__ dst = feat_sum + (size_t)i * C;
	if ((C & 3) == 0) {
		for (int c = lane * 4; c < C; c += 256) {
			const float4 f = *reinterpret_cast<const float4*>(src + c);
			float4 s = *reinterpret_cast<float4*>(dst + c);
			s.x += f.x;
			s.y += f.y;
			s.z += f.z;
			s.w += f.w;
			*reinterpret_cast<float4*>(dst + c) = s;
		}
	} else {
		for (int c = lane; c < C; c += 64) dst[c] += src[c];
	}
	if (lane == 0) times[i] += 1.0f;
}

} // namespace

hipError_t launch_fusion_mapping(hipStream_t st, int N, const float* coords, const float* wvt,
				 const double intr[4], int W, int H, int cut, double vis_thres, int depth_mode,
				 const float* depth, double* zbuf, long long* mapping, double* weight)
{
	if (N == 0) return hipSuccess;
	const dim3 grid((N + 255) / 256), block(256);
	if (depth_mode == 2) {
		const size_t npx = (size_t)W * H;
		const double far = 999999.0;
		unsigned long long bits;
		__builtin_memcpy(&bits, &far, 8);
		hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((npx + 255) / 256)), block, 0, st, npx, bits,
				   (unsigned long long*)zbuf);
		hipLaunchKernelGGL(fusion_zbuffer_kernel, grid, block, 0, st, N, coords, wvt, intr[0], intr[1], intr[2],
				   intr[3], W, H, cut, (unsigned long long*)zbuf);
		hipLaunchKernelGGL(fusion_mapping_kernel<2>, grid, block, 0, st, N, coords, wvt, intr[0], intr[1], intr[2],
				   intr[3], W, H, cut, vis_thres, (const void*)zbuf, mapping, weight);
	} else if (depth_mode == 1) {
		hipLaunchKernelGGL(fusion_mapping_kernel<1>, grid, block, 0, st, N, coords, wvt, intr[0], intr[1], intr[2],
				   intr[3], W, H, cut, vis_thres, (const void*)depth, mapping, weight);
	} else {
		hipLaunchKernelGGL(fusion_mapping_kernel<0>, grid, block, 0, st, N, coords, wvt, intr[0], intr[1], intr[2],
				   intr[3], W, H, cut, vis_thres, (const void*)nullptr, mapping, weight);
	}
	return hipGetLastError();
}

hipError_t launch_fusion_accumulate(hipStream_t st, int N, int C, const float* feat_hwc, int W,
				    const long long* mapping, float* feat_sum, float* times)
{
	if (N == 0 || C == 0) return hipSuccess;
	hipLaunchKernelGGL(fusion_accumulate_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, C, feat_hwc, W, mapping,
			   feat_sum, times);
	return hipGetLastError();
}

} // namespace sgs
