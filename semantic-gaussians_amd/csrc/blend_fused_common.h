// blend_fused_common.h -- pieces shared by the two single-kernel forward blends (blend_fused.hip: wave-autonomous,
// blend_fused_pc.hip: one producer wave + consumer waves per workgroup).  Internal to libsgs_hip.so.
#pragma once
#include "sgs_kernels.h"

namespace sgs {
namespace fused {

constexpr uint32_t F_BG_ID = 0xFFFFFFFFu;   // work-list id of the closing T * bg pseudo entry

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct QRec {   // one queued list entry (32 B)
	float a2, b2, c2, o;
	float x, y, thr;
	uint32_t id;
};

struct FusedArgs {
	const uint2* ranges;
	const uint32_t* point_list;
	const float2* means2D;
	const float4* conic_opacity;
	const float* features;
	const float* bg;
	float* out;
	float* final_T;
	uint32_t* n_contrib;
	int W, H, C, gx, gy;
	int seg, nseg, nchunks, per_xcd, total_items;
};

// f -> bf16 hi (round to nearest even) and bf16 lo = bf16(f - hi); two values packed per register
__device__ __forceinline__ void split2(float f0, float f1, short& h0, short& h1, short& l0, short& l1)
{
	const __bf16 a = (__bf16)f0, b = (__bf16)f1;
	const __bf16 c = (__bf16)(f0 - (float)a), d = (__bf16)(f1 - (float)b);
	h0 = __builtin_bit_cast(short, a);
	h1 = __builtin_bit_cast(short, b);
	l0 = __builtin_bit_cast(short, c);
	l1 = __builtin_bit_cast(short, d);
}

struct Frag {   // 4 k-values of one operand, split
	s16x4 hi, lo;
};

__device__ __forceinline__ Frag make_frag(const float f[4])
{
	Frag r;
	short h[4], l[4];
	split2(f[0], f[1], h[0], h[1], l[0], l[1]);
	split2(f[2], f[3], h[2], h[3], l[2], l[3]);
	r.hi = s16x4{h[0], h[1], h[2], h[3]};
	r.lo = s16x4{l[0], l[1], l[2], l[3]};
	return r;
}

// one 1-KB LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to lds_addr + lane * 16
__device__ __forceinline__ void dma16(const void* gptr, uint32_t lds_addr)
{
	asm volatile("s_mov_b32 m0, %0\n\t"
		     "s_nop 0\n\t"
		     "global_load_lds_dwordx4 %1, off"
		     :
		     : "s"(lds_addr), "v"(gptr)
		     : "memory", "m0");
}

typedef int v4i __attribute__((ext_vector_type(4)));

// Output stores go through a raw buffer descriptor over this workgroup's 128 channel planes:
//   buffer_store_dword data, voffset, srsrc, soffset offen nt
// soffset (SGPR) walks the channel plane, voffset is the lane's byte offset inside the plane pair -- and a lane that
// must not store (pixel outside the image) simply carries an out-of-range voffset: the hardware's bounds check drops
// it.  So image edges need no exec masking and no second code path.  `nt`: the image is written once.
constexpr uint32_t F_OOB = 0xFFFFFFFCu;
constexpr uint32_t F_NUM_RECORDS = 0xFFFFF000u;

// The plane offset walks 0,1,2,3, 8,9,10,11, 16,.. (the 32x32 MFMA's accumulator rows: register r of channel block cb
// is channel 32 cb + (r & 3) + 8 (r >> 2)) by scalar adds.  The adds are asm volatile on purpose: written as plain
// arithmetic the compiler tabulates all 128 offsets as loop invariants, runs out of SGPRs and spills them to VGPR lanes.
__device__ __forceinline__ void walk_plane(uint32_t& so, int r, uint32_t plane1, uint32_t plane5)
{
	if ((r & 3) == 3) asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(plane5) : "scc");
	else asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(plane1) : "scc");
}


} // namespace fused
} // namespace sgs
