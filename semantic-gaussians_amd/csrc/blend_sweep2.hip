// blend_sweep2.hip -- round 3: the accumulate half of the C >= 128 forward blend (CR/cuda_rasterizer/forward.cu:
// 262-375, the `C[ch] += features[ch] * alpha * T` line 355-356) in fp32-class arithmetic at the memory system's
// rate.  Same job as blend_accum_sweep_kernel (blend_fwd_split.hip): out[ch][px] = sum_k F[k][ch] * W[k][px] over
// the work list the weights pre-pass wrote, as tile-row sweeps that store complete 128-B lines.  What is new:
//
//  1. ARITHMETIC.  The weights arrive as fp32 rows (work-list format MODE 3) and the product runs either as
//       ARITH_EXACT : v_mfma_f32_32x32x2_f32 -- a k-ordered fp32 fma chain, bit-identical to the contract, or
//       ARITH_X6    : "f32-equivalent": F = F1 + F2 + F3 and W = W1 + W2 + W3 EXACTLY (three bf16 terms of 8
//                     significant bits each cover fp32's 24), the six products with i + j <= 4 on
//                     v_mfma_f32_32x32x8_bf16_1k, fp32 accumulate.  Every product is exact in fp32; what is dropped
//                     (F2 W3 + F3 W2 + F3 W3) is <= 2^-23 |F W| -- the size of the ONE rounding the reference's
//                     fp32 multiply-add makes per term.  The splits are 11 VALU per pair of values, done in
//                     registers (v_cvt_pk_bf16_f32 on both halves), 6/8 of the fp32 MFMA's matrix time.
//       ARITH_X6W   : experiment only: the same six products on the double-rate v_mfma_f32_32x32x16_bf16.
//
//  2. NO vmcnt IN THE LOOP.  gfx9 has ONE counter for loads, LDS-DMA and stores, ordered only within each kind, so
//     a counted wait for a DMA bundle is also a wait for every store issued before it: round 2's sweep had to
//     drain the ring ahead of a tile's store burst and the stores ahead of the next wait.  Here a bundle's arrival
//     is read off LDS itself: the last DMA instruction of a wave's bundle deposits the (wave-private) id words; the
//     wave overwrites them with a sentinel before it issues the bundle and polls them when it needs the bundle
//     (loads return in issue order, so "my last DMA landed" implies "all my DMAs of the bundle landed"; the
//     workgroup barrier that follows extends it to the other waves' parts).  Stores are never waited for until
//     the kernel ends.
//
//  3. STORES SPREAD OVER THE NEXT TILE.  A finished tile pair is 128 store instructions per wave.  Half of them
//     (pixel blocks 0, 1) go out at once -- that many fit the 6-bit counter without blocking -- and free the four
//     accumulator blocks the next left tile accumulates into; the other half (blocks 2, 3) are issued 16 per batch
//     behind the next tile's MFMAs.  The eight accumulator blocks therefore change roles from pair to pair
//     (mapping M = 0 / 1 below); every index is a compile-time constant of the instantiation.
//
// Workgroup = (tile-row segment, 128 channels, row parity), 4 waves = 4 channel groups of 32, wave tile = 32 channels
// x the 128 pixels of the parity (4 MFMA blocks); ring of 4 stages of 16 fp32 feature rows (8 KB) + 16 fp32 weight
// rows of this parity (8 KB) + ids (1 KB), three bundles in flight.
#include "sgs_kernels.h"
#include <type_traits>
#include <cstdlib>

namespace sgs {

namespace {

constexpr int AB = 16;                        // work-list entries per batch
constexpr uint32_t SGS_BG_ID = 0xFFFFFFFFu;   // work-list id of the closing T * bg pseudo entry
constexpr uint32_t S2_SENT = 0xFFFFFFFEu;     // "not landed yet" (never a Gaussian id: ids < 2^31)
constexpr int S2_SEGMAX = 96;
constexpr int S2_JMAX = 1024;
// ring geometry by weights format: fp32 rows (8 KB per batch and parity, 4 stages) | three bf16 terms (12 KB, 3 stages:
// two workgroups of 4 x 21 KB do not fit a CU's 160 KB).  Stage = features | this parity's weights | ids of the batch
// LA bundles on.
// KIND 0: fp32 weight rows, 4 stages; 1: pre-split weights (three bf16 terms), 3 stages; 2: fp32 weight rows, 3 stages + a
// 12-KB buffer the four waves split the batch's weights into (S2_X6C)
template <int KIND> struct RingCfg {
	static constexpr int NST = KIND == 0 ? 4 : 3, LA = NST - 1;
	static constexpr int WBYTES = KIND == 1 ? 12288 : 8192;
	static constexpr int STAGE = 8192 + WBYTES + 1024;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// accumulator blocks of the left / right tile's pixel block pb under mapping M
__host__ __device__ constexpr int LB(int M, int pb) { return M == 0 ? pb : (pb < 2 ? pb : pb + 2); }       // {0,1,2,3} | {0,1,4,5}
__host__ __device__ constexpr int RB(int M, int pb) { return M == 0 ? pb + 4 : (pb < 2 ? pb + 2 : pb + 4); } // {4,5,6,7} | {2,3,6,7}

// The accumulators live in the accumulator half of the register file under LITERAL names: block b = a[16 b : 16 b + 15],
// eight blocks = a[0:127].  hipcc never sees them as values (every statement that touches them names them and lists
// a127 as a clobber, which also makes the kernel descriptor allocate all 128), so there is nothing for it to copy,
// rotate or spill when a block changes its role from accumulator to store source from one tile pair to the next --
// as C++ values the same structure spilled 400-780 registers.  Audit after every edit (cdna_hip_programming.md 5.7
// item 4): .vgpr_spill_count 0, scratch 0, and no compiler-generated v_accvgpr_* outside ;;#ASMSTART / ;;#ASMEND.
#define S2_ACC "a127"

template <int BLK>
__device__ __forceinline__ void acc_zero()
{
	asm volatile(
		"v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0\n\t"
		"v_accvgpr_write_b32 a[%c4], 0\n\tv_accvgpr_write_b32 a[%c5], 0\n\tv_accvgpr_write_b32 a[%c6], 0\n\tv_accvgpr_write_b32 a[%c7], 0\n\t"
		"v_accvgpr_write_b32 a[%c8], 0\n\tv_accvgpr_write_b32 a[%c9], 0\n\tv_accvgpr_write_b32 a[%c10], 0\n\tv_accvgpr_write_b32 a[%c11], 0\n\t"
		"v_accvgpr_write_b32 a[%c12], 0\n\tv_accvgpr_write_b32 a[%c13], 0\n\tv_accvgpr_write_b32 a[%c14], 0\n\tv_accvgpr_write_b32 a[%c15], 0"
		: : "i"(BLK * 16), "i"(BLK * 16 + 1), "i"(BLK * 16 + 2), "i"(BLK * 16 + 3), "i"(BLK * 16 + 4), "i"(BLK * 16 + 5),
		    "i"(BLK * 16 + 6), "i"(BLK * 16 + 7), "i"(BLK * 16 + 8), "i"(BLK * 16 + 9), "i"(BLK * 16 + 10), "i"(BLK * 16 + 11),
		    "i"(BLK * 16 + 12), "i"(BLK * 16 + 13), "i"(BLK * 16 + 14), "i"(BLK * 16 + 15)
		: S2_ACC);
}

// a whole block into VGPRs (edge and segment-end stores only).  WAIT: the block may have been written by the MFMA
// issued just before (16-pass f32 form: 18 wait states from issue to a VALU read of D) -- 32 states of s_nop.
template <int BLK, bool WAIT>
__device__ __forceinline__ void acc_read(float (&v)[16])
{
	if (WAIT) asm volatile("s_nop 15\n\ts_nop 15" : : : "memory");
	asm volatile(
		"v_accvgpr_read_b32 %0, a[%c16]\n\tv_accvgpr_read_b32 %1, a[%c17]\n\tv_accvgpr_read_b32 %2, a[%c18]\n\tv_accvgpr_read_b32 %3, a[%c19]\n\t"
		"v_accvgpr_read_b32 %4, a[%c20]\n\tv_accvgpr_read_b32 %5, a[%c21]\n\tv_accvgpr_read_b32 %6, a[%c22]\n\tv_accvgpr_read_b32 %7, a[%c23]\n\t"
		"v_accvgpr_read_b32 %8, a[%c24]\n\tv_accvgpr_read_b32 %9, a[%c25]\n\tv_accvgpr_read_b32 %10, a[%c26]\n\tv_accvgpr_read_b32 %11, a[%c27]\n\t"
		"v_accvgpr_read_b32 %12, a[%c28]\n\tv_accvgpr_read_b32 %13, a[%c29]\n\tv_accvgpr_read_b32 %14, a[%c30]\n\tv_accvgpr_read_b32 %15, a[%c31]"
		: "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]),
		  "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15])
		: "i"(BLK * 16), "i"(BLK * 16 + 1), "i"(BLK * 16 + 2), "i"(BLK * 16 + 3), "i"(BLK * 16 + 4), "i"(BLK * 16 + 5),
		  "i"(BLK * 16 + 6), "i"(BLK * 16 + 7), "i"(BLK * 16 + 8), "i"(BLK * 16 + 9), "i"(BLK * 16 + 10), "i"(BLK * 16 + 11),
		  "i"(BLK * 16 + 12), "i"(BLK * 16 + 13), "i"(BLK * 16 + 14), "i"(BLK * 16 + 15));
}

__device__ __forceinline__ uint32_t pk_bf16(float a, float b)
{
	const f32x2 v = {a, b};
	return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

// (x0, x1) -> three packed bf16 pairs with x = t1 + t2 + t3 exactly (fp32 subtraction of a value's own rounding is exact)
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3)
{
	p1 = pk_bf16(x0, x1);
	const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
	p2 = pk_bf16(r0, r1);
	const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
	p3 = pk_bf16(s0, s1);
}

struct Op3 {   // one lane's 8 k-values of an operand as three bf16 terms: t[term][q] = the four values of MFMA q (k = 8 h + 4 q + i)
	u32x2 t[3][2];
};

__device__ __forceinline__ void split8(const float (&x)[8], Op3& o)
{
#pragma unroll
	for (int q = 0; q < 2; q++) {
		uint32_t a1, a2, a3, b1, b2, b3;
		split3(x[4 * q], x[4 * q + 1], a1, a2, a3);
		split3(x[4 * q + 2], x[4 * q + 3], b1, b2, b3);
		o.t[0][q] = u32x2{a1, b1};
		o.t[1][q] = u32x2{a2, b2};
		o.t[2][q] = u32x2{a3, b3};
	}
}

// The six products with i + j <= 4 of one pixel block into accumulator block BLK, smallest terms first (all land in one
// fp32 accumulator; the order only matters to the last bit).  Each product is two v_mfma_f32_32x32x8_bf16 over the two
// halves of the lane's 8 k-values (lane half h owns k = 8 h + 4 q + i in MFMA q: the pairing of A and B stays
// consistent).  s_nop 1: the operands may have been written by the VALU instruction just before (hipcc pads nothing
// for an asm); back-to-back MFMAs on one accumulator need no padding.
template <int BLK>
__device__ __forceinline__ void mfma_x6(const Op3& A, const Op3& B)
{
#define S2_M "v_mfma_f32_32x32x8_bf16 a[%c0:%c1], "
	asm volatile(
		"s_nop 1\n\t"
		S2_M "%6, %8, a[%c0:%c1]\n\t" S2_M "%7, %9, a[%c0:%c1]\n\t"        // A3 B1
		S2_M "%2, %12, a[%c0:%c1]\n\t" S2_M "%3, %13, a[%c0:%c1]\n\t"      // A1 B3
		S2_M "%4, %10, a[%c0:%c1]\n\t" S2_M "%5, %11, a[%c0:%c1]\n\t"      // A2 B2
		S2_M "%4, %8, a[%c0:%c1]\n\t" S2_M "%5, %9, a[%c0:%c1]\n\t"        // A2 B1
		S2_M "%2, %10, a[%c0:%c1]\n\t" S2_M "%3, %11, a[%c0:%c1]\n\t"      // A1 B2
		S2_M "%2, %8, a[%c0:%c1]\n\t" S2_M "%3, %9, a[%c0:%c1]"            // A1 B1
		: : "i"(BLK * 16), "i"(BLK * 16 + 15),
		    "v"(A.t[0][0]), "v"(A.t[0][1]), "v"(A.t[1][0]), "v"(A.t[1][1]), "v"(A.t[2][0]), "v"(A.t[2][1]),
		    "v"(B.t[0][0]), "v"(B.t[0][1]), "v"(B.t[1][0]), "v"(B.t[1][1]), "v"(B.t[2][0]), "v"(B.t[2][1])
		: S2_ACC);
#undef S2_M
}

// (experiment) the same six products on the double-rate v_mfma_f32_32x32x16_bf16: operand = the lane's 8 k-values
template <int BLK>
__device__ __forceinline__ void mfma_x6_wide(const Op3& A, const Op3& B)
{
	u32x4 a[3], b[3];
#pragma unroll
	for (int t = 0; t < 3; t++) {
		a[t] = u32x4{A.t[t][0].x, A.t[t][0].y, A.t[t][1].x, A.t[t][1].y};
		b[t] = u32x4{B.t[t][0].x, B.t[t][0].y, B.t[t][1].x, B.t[t][1].y};
	}
#define S2_M "v_mfma_f32_32x32x16_bf16 a[%c0:%c1], "
	asm volatile(
		"s_nop 1\n\t"
		S2_M "%4, %5, a[%c0:%c1]\n\t" S2_M "%2, %7, a[%c0:%c1]\n\t" S2_M "%3, %6, a[%c0:%c1]\n\t"
		S2_M "%3, %5, a[%c0:%c1]\n\t" S2_M "%2, %6, a[%c0:%c1]\n\t" S2_M "%2, %5, a[%c0:%c1]"
		: : "i"(BLK * 16), "i"(BLK * 16 + 15), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2])
		: S2_ACC);
#undef S2_M
}

// eight fp32 words, 512 B apart (one per work-list entry of this lane's k range), issued without a wait
#define S2_READ8(dst, addr)                                                                         \
	asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:512\n\tds_read_b32 %2, %8 offset:1024\n\t" \
		     "ds_read_b32 %3, %8 offset:1536\n\tds_read_b32 %4, %8 offset:2048\n\tds_read_b32 %5, %8 offset:2560\n\t" \
		     "ds_read_b32 %6, %8 offset:3072\n\tds_read_b32 %7, %8 offset:3584"                      \
		     : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7]) \
		     : "v"(addr) : "memory")
#define S2_WAIT8(dst)                                                                               \
	do {                                                                                            \
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]) : : "memory"); \
		__builtin_amdgcn_sched_barrier(0);                                                          \
	} while (0)

// one (A term, B term, k half) combination into TWO accumulator blocks: consecutive MFMAs never touch the same
// accumulator (a dependent 8-pass MFMA issues ~40 cycles after its predecessor, an independent one after 32)
template <int BX, int BY>
__device__ __forceinline__ void mfma_pair(u32x2 a, u32x2 bx, u32x2 by)
{
	asm volatile(
		"s_nop 1\n\t"
		"v_mfma_f32_32x32x8_bf16 a[%c0:%c1], %4, %5, a[%c0:%c1]\n\t"
		"v_mfma_f32_32x32x8_bf16 a[%c2:%c3], %4, %6, a[%c2:%c3]"
		: : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15), "v"(a), "v"(bx), "v"(by)
		: S2_ACC, "memory");
}

// One batch of 16 entries out of ring stage `st` (LDS byte address) into accumulator blocks B0..B3 (the tile's four
// pixel blocks), f32-equivalent arithmetic.  Lane (half h, l31) of channel group cg holds A = F[k = 8 h + j][cg * 32 +
// l31] and, per pixel block, B = W[k = 8 h + j][pb * 32 + l31], j = 0..7.  All LDS reads of the ring are inline asm (the
// compiler's waitcnt pass would put vmcnt(0) in front of a ds_read that may alias an LDS-DMA destination).
// Schedule (ILV): blocks in pairs; the second pair's weights are split three-ways in 11-VALU slices BETWEEN the first
// pair's MFMA statements (a slice is 22 cycles, an MFMA 32: the matrix pipe does not see them).
template <bool WIDE, bool ILV, int B0, int B1, int B2, int B3>
__device__ __forceinline__ void s2_compute_x6(uint32_t st, int cg, int half, int l31)
{
	const uint32_t fa = st + (uint32_t)((8 * half) * 128 + cg * 32 + l31) * 4u;
	const uint32_t wa = st + 8192u + (uint32_t)((8 * half) * 128 + l31) * 4u;   // + pb * 128
	if constexpr (ILV && !WIDE) {
		float f[8], w0[8], w1[8], w2[8], w3[8];
		// (lgkmcnt is 4 bits: at most 15 LDS operations are counted, so the reads go out 8 at a time behind a wait for
		// the group before last; LDS returns in order)
#define S2_WAIT8N(dst, n_)                                                                          \
	do {                                                                                            \
		asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]) : : "memory"); \
		__builtin_amdgcn_sched_barrier(0);                                                          \
	} while (0)
		S2_READ8(f, fa);
		S2_READ8(w0, wa);
		S2_WAIT8N(f, 8);
		Op3 A, X, Y, X2, Y2;
		split8(f, A);
		S2_READ8(w1, wa + 128u);
		S2_WAIT8N(w0, 8);
		split8(w0, X);
		S2_READ8(w2, wa + 256u);
		S2_WAIT8N(w1, 8);
		split8(w1, Y);
		S2_READ8(w3, wa + 384u);
		S2_WAIT8N(w2, 8);
		// first pair; slice i of the second pair's split behind statement i
		constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
		uint32_t x2[3][4], y2[3][4];
// (the empty asm statements pin the slice between the two MFMA statements around it: volatile asm keeps its order, the
// inputs become available at the first, the outputs are consumed by the second -- sched_barrier alone does not stop the
// instruction selector from sinking the whole split below the last MFMA)
#define S2_SLICE1(src_, dst_, k_)                                                                    \
	do {                                                                                             \
		asm volatile("" : "+v"(src_[2 * (k_)]), "+v"(src_[2 * (k_) + 1]));                            \
		split3(src_[2 * (k_)], src_[2 * (k_) + 1], dst_[0][k_], dst_[1][k_], dst_[2][k_]);             \
		asm volatile("" : "+v"(dst_[0][k_]), "+v"(dst_[1][k_]), "+v"(dst_[2][k_]));                    \
	} while (0)
#define S2_SLICE(i_)                                                                                 \
	do {                                                                                             \
		if ((i_) == 4) S2_WAIT8N(w3, 0);                                                             \
		if ((i_) < 4) S2_SLICE1(w2, x2, (i_));                                                       \
		else if ((i_) < 8) S2_SLICE1(w3, y2, (i_) - 4);                                              \
		__builtin_amdgcn_sched_barrier(0);                                                           \
	} while (0)
#pragma unroll
		for (int c = 0; c < 6; c++)
#pragma unroll
			for (int q = 0; q < 2; q++) {
				mfma_pair<B0, B1>(A.t[TA[c]][q], X.t[TB[c]][q], Y.t[TB[c]][q]);
				__builtin_amdgcn_sched_barrier(0);
				S2_SLICE(2 * c + q);
			}
#undef S2_SLICE
#undef S2_SLICE1
#undef S2_WAIT8N
#pragma unroll
		for (int t = 0; t < 3; t++)
#pragma unroll
			for (int q = 0; q < 2; q++) {
				X2.t[t][q] = u32x2{x2[t][2 * q], x2[t][2 * q + 1]};
				Y2.t[t][q] = u32x2{y2[t][2 * q], y2[t][2 * q + 1]};
			}
#pragma unroll
		for (int c = 0; c < 6; c++)
#pragma unroll
			for (int q = 0; q < 2; q++) mfma_pair<B2, B3>(A.t[TA[c]][q], X2.t[TB[c]][q], Y2.t[TB[c]][q]);
		return;
	}
	float f[8], w[8], wn[8];
	S2_READ8(f, fa);
	S2_READ8(w, wa);
	S2_READ8(wn, wa + 128u);
	asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
		     "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : : "memory");
	__builtin_amdgcn_sched_barrier(0);
	Op3 A, B, Bn;
	split8(f, A);
	split8(w, B);
#define S2_PB(pb_, BLK_)                                                                             \
	do {                                                                                             \
		if (pb_ < 3) {   /* the next block's weights landed during the previous block's products */ \
			S2_WAIT8(wn);                                                                            \
			split8(wn, Bn);                                                                          \
		}                                                                                            \
		if (WIDE) mfma_x6_wide<BLK_>(A, B);                                                          \
		else mfma_x6<BLK_>(A, B);                                                                    \
		if (pb_ < 2) S2_READ8(wn, wa + (uint32_t)(pb_ + 2) * 128u);                                   \
		if (pb_ < 3) B = Bn;                                                                         \
	} while (0)
	S2_PB(0, B0);
	S2_PB(1, B1);
	S2_PB(2, B2);
	S2_PB(3, B3);
#undef S2_PB
}

// all 24 MFMAs of a pixel-block pair in one statement: nothing (no compiler nop, no VALU) between them
template <int BX, int BY>
__device__ __forceinline__ void mfma_dense(const Op3& A, const u32x4 (&x)[3], const u32x4 (&y)[3])
{
	const u32x2 x0a = {x[0].x, x[0].y}, x0b = {x[0].z, x[0].w}, x1a = {x[1].x, x[1].y}, x1b = {x[1].z, x[1].w}, x2a = {x[2].x, x[2].y}, x2b = {x[2].z, x[2].w};
	const u32x2 y0a = {y[0].x, y[0].y}, y0b = {y[0].z, y[0].w}, y1a = {y[1].x, y[1].y}, y1b = {y[1].z, y[1].w}, y2a = {y[2].x, y[2].y}, y2b = {y[2].z, y[2].w};
#define S2_MX "v_mfma_f32_32x32x8_bf16 a[%c0:%c1], "
#define S2_MY "v_mfma_f32_32x32x8_bf16 a[%c2:%c3], "
#define S2_P(a_, bx_, by_) S2_MX a_ ", " bx_ ", a[%c0:%c1]\n\t" S2_MY a_ ", " by_ ", a[%c2:%c3]\n\t"
	// A: %4 %5 = term 1 (q 0, 1), %6 %7 = term 2, %8 %9 = term 3; X: %10 .. %15 likewise; Y: %16 .. %21
	asm volatile(
		"s_nop 1\n\t"
		S2_P("%8", "%10", "%16") S2_P("%9", "%11", "%17")      // A3 B1
		S2_P("%4", "%14", "%20") S2_P("%5", "%15", "%21")      // A1 B3
		S2_P("%6", "%12", "%18") S2_P("%7", "%13", "%19")      // A2 B2
		S2_P("%6", "%10", "%16") S2_P("%7", "%11", "%17")      // A2 B1
		S2_P("%4", "%12", "%18") S2_P("%5", "%13", "%19")      // A1 B2
		S2_P("%4", "%10", "%16") S2_P("%5", "%11", "%17")      // A1 B1
		"s_nop 0"
		: : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15),
		    "v"(A.t[0][0]), "v"(A.t[0][1]), "v"(A.t[1][0]), "v"(A.t[1][1]), "v"(A.t[2][0]), "v"(A.t[2][1]),
		    "v"(x0a), "v"(x0b), "v"(x1a), "v"(x1b), "v"(x2a), "v"(x2b),
		    "v"(y0a), "v"(y0b), "v"(y1a), "v"(y1b), "v"(y2a), "v"(y2b)
		: S2_ACC, "memory");
#undef S2_P
#undef S2_MX
#undef S2_MY
}

// the double-rate forms: one v_mfma_f32_32x32x16_bf16 per (A term, B term) and block
template <int BX, int BY>
__device__ __forceinline__ void mfma_pair_wide(u32x4 a, u32x4 bx, u32x4 by)
{
	asm volatile(
		"s_nop 1\n\t"
		"v_mfma_f32_32x32x16_bf16 a[%c0:%c1], %4, %5, a[%c0:%c1]\n\t"
		"v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %4, %6, a[%c2:%c3]"
		: : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15), "v"(a), "v"(bx), "v"(by)
		: S2_ACC, "memory");
}

template <int BX, int BY>
__device__ __forceinline__ void mfma_dense_wide(const u32x4 (&a)[3], const u32x4 (&x)[3], const u32x4 (&y)[3])
{
#define S2_P(a_, bx_, by_) "v_mfma_f32_32x32x16_bf16 a[%c0:%c1], " a_ ", " bx_ ", a[%c0:%c1]\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], " a_ ", " by_ ", a[%c2:%c3]\n\t"
	asm volatile(
		"s_nop 1\n\t"
		S2_P("%6", "%7", "%10")     // A3 B1
		S2_P("%4", "%9", "%12")     // A1 B3
		S2_P("%5", "%8", "%11")     // A2 B2
		S2_P("%5", "%7", "%10")     // A2 B1
		S2_P("%4", "%8", "%11")     // A1 B2
		S2_P("%4", "%7", "%10")     // A1 B1
		"s_nop 0"
		: : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15),
		    "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(y[0]), "v"(y[1]), "v"(y[2])
		: S2_ACC, "memory");
#undef S2_P
}

// (round 4 experiment, make X16=1) the same double-rate statements with ONE filler between consecutive MFMAs -- SP = 1:
// `s_nop 0`, SP = 2: a VALU move on a scratch register.  Round 3's evidence: the x16 sweep whose MFMAs were
// separated by VALU work never disturbed a neighbour (0 of 324 000), 12 back-to-back ones do (1 in 300-1 100): is it the
// DENSITY of the issue?  An MFMA holds the pipe for 32 cycles, so one filler in its shadow costs nothing.
template <int SP, int BX, int BY>
__device__ __forceinline__ void mfma_dense_wide_sp(const u32x4 (&a)[3], const u32x4 (&x)[3], const u32x4 (&y)[3])
{
	uint32_t scratch = 0u;
#define S2_F "%13"
#define S2_P(a_, bx_, by_)                                                                            \
	"v_mfma_f32_32x32x16_bf16 a[%c0:%c1], " a_ ", " bx_ ", a[%c0:%c1]\n\t" S2_FILL                  \
	"v_mfma_f32_32x32x16_bf16 a[%c2:%c3], " a_ ", " by_ ", a[%c2:%c3]\n\t" S2_FILL
	if constexpr (SP == 1) {
#define S2_FILL "s_nop 0\n\t"
		asm volatile("s_nop 1\n\t" S2_P("%6", "%7", "%10") S2_P("%4", "%9", "%12") S2_P("%5", "%8", "%11")
			     S2_P("%5", "%7", "%10") S2_P("%4", "%8", "%11") S2_P("%4", "%7", "%10") "s_nop 0"
			     : : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15),
				 "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(scratch)
			     : S2_ACC, "memory");
#undef S2_FILL
	} else {
#define S2_FILL "v_mov_b32 " S2_F ", " S2_F "\n\t"
		asm volatile("s_nop 1\n\t" S2_P("%6", "%7", "%10") S2_P("%4", "%9", "%12") S2_P("%5", "%8", "%11")
			     S2_P("%5", "%7", "%10") S2_P("%4", "%8", "%11") S2_P("%4", "%7", "%10") "s_nop 0"
			     : : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15),
				 "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(scratch)
			     : S2_ACC, "memory");
#undef S2_FILL
	}
#undef S2_P
#undef S2_F
}

template <int SP, int BX, int BY>
__device__ __forceinline__ void mfma_pair_wide_sp(u32x4 a, u32x4 bx, u32x4 by)
{
	uint32_t scratch = 0u;
	if constexpr (SP == 1)
		asm volatile("s_nop 1\n\t"
			     "v_mfma_f32_32x32x16_bf16 a[%c0:%c1], %4, %5, a[%c0:%c1]\n\ts_nop 0\n\t"
			     "v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %4, %6, a[%c2:%c3]\n\ts_nop 0"
			     : : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15), "v"(a), "v"(bx), "v"(by), "v"(scratch) : S2_ACC, "memory");
	else
		asm volatile("s_nop 1\n\t"
			     "v_mfma_f32_32x32x16_bf16 a[%c0:%c1], %4, %5, a[%c0:%c1]\n\tv_mov_b32 %7, %7\n\t"
			     "v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %4, %6, a[%c2:%c3]\n\tv_mov_b32 %7, %7"
			     : : "i"(BX * 16), "i"(BX * 16 + 15), "i"(BY * 16), "i"(BY * 16 + 15), "v"(a), "v"(bx), "v"(by), "v"(scratch) : S2_ACC, "memory");
}

// The batch with the weights already split by the weights pre-pass (work-list format MODE 4): stage = features (fp32,
// split here: 44 VALU) | [group h][term][128 px x 8 bf16].  A lane's B operand of a (term, pixel block) is ONE 16-byte
// read: its two halves are the operands of the two MFMAs over k = 8 h + 4 q + i.  Pixel blocks in pairs, consecutive
// MFMAs on different accumulators (a dependent 8-pass MFMA issues ~40 cycles after its predecessor, an independent one
// after 32); the second pair's operands land while the first pair multiplies.  The first pair is one dense statement;
// the NP pieces of the next bundle's DMA (`piece`) go between the second pair's statements, where their issue slots
// (~60 cycles each) hide behind 64 cycles of matrix work.
template <bool WIDE, int B0, int B1, int B2, int B3, int NP, typename F>
__device__ __forceinline__ void s2_compute_x6p(uint32_t st, uint32_t wb, int cg, int half, int l31, F piece)
{
	const uint32_t fa = st + (uint32_t)((8 * half) * 128 + cg * 32 + l31) * 4u;
	const uint32_t wa = wb + (uint32_t)half * 6144u + (uint32_t)l31 * 16u;   // + term * 2048 + pb * 512  (wb: the stage's weight area / the split buffer)
	float f[8];
	u32x4 x[3], y[3], x2[3], y2[3];
	S2_READ8(f, fa);
#define S2_RDB(dst_, pb_)                                                                            \
	asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6" \
		     : "=&v"(dst_[0]), "=&v"(dst_[1]), "=&v"(dst_[2]) : "v"(wa), "n"((pb_) * 512), "n"(2048 + (pb_) * 512), "n"(4096 + (pb_) * 512) : "memory")
#define S2_WTB(dst_, n_)                                                                             \
	do {                                                                                             \
		asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(dst_[0]), "+v"(dst_[1]), "+v"(dst_[2]) : : "memory"); \
		__builtin_amdgcn_sched_barrier(0);                                                           \
	} while (0)
	S2_RDB(x, 0);
	S2_RDB(y, 1);
	asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory");
	__builtin_amdgcn_sched_barrier(0);
	Op3 A;
	split8(f, A);
	S2_RDB(x2, 2);
	S2_RDB(y2, 3);
	S2_WTB(x, 9);
	S2_WTB(y, 6);
	constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
	if constexpr (WIDE) {   // the lane's 8 k-values of a term are one operand of the double-rate MFMA
		u32x4 a[3];
#pragma unroll
		for (int t = 0; t < 3; t++) a[t] = u32x4{A.t[t][0].x, A.t[t][0].y, A.t[t][1].x, A.t[t][1].y};
		mfma_dense_wide<B0, B1>(a, x, y);
		S2_WTB(x2, 3);
		S2_WTB(y2, 0);
#define S2_HALF2W(c_)                                                                                \
	do {                                                                                             \
		mfma_pair_wide<B2, B3>(a[TA[c_]], x2[TB[c_]], y2[TB[c_]]);                                   \
		if constexpr ((c_) < NP) piece(std::integral_constant<int, (c_)>{});                         \
	} while (0)
		S2_HALF2W(0); S2_HALF2W(1); S2_HALF2W(2); S2_HALF2W(3); S2_HALF2W(4); S2_HALF2W(5);
#undef S2_HALF2W
		return;
	}
	mfma_dense<B0, B1>(A, x, y);
	S2_WTB(x2, 3);
	S2_WTB(y2, 0);
#define S2_HALF2(c_)                                                                                 \
	do {                                                                                             \
		mfma_pair<B2, B3>(A.t[TA[c_]][0], u32x2{x2[TB[c_]].x, x2[TB[c_]].y}, u32x2{y2[TB[c_]].x, y2[TB[c_]].y}); \
		if constexpr ((c_) < NP) piece(std::integral_constant<int, (c_)>{});                         \
		mfma_pair<B2, B3>(A.t[TA[c_]][1], u32x2{x2[TB[c_]].z, x2[TB[c_]].w}, u32x2{y2[TB[c_]].z, y2[TB[c_]].w}); \
	} while (0)
	S2_HALF2(0); S2_HALF2(1); S2_HALF2(2); S2_HALF2(3); S2_HALF2(4); S2_HALF2(5);
#undef S2_HALF2
#undef S2_RDB
#undef S2_WTB
}

// S2_X6C: the batch's fp32 weights -> the three bf16 terms in the pre-split operand layout ([group of 8 entries][term]
// [128 px'][8 x bf16], 12 KB).  Wave w converts pixels 32 w .. 32 w + 31 and FETCHES exactly those (its two weight DMAs
// gather the 128-byte pieces [entry][32 px'] of the 16 entries into its own 2 KB of the stage), so the arrival of its own
// bundle is all it needs: the split runs before the step's barrier, which then covers "every bundle landed" and "the split
// buffer is complete" at once.  The buffer is double: a wave may be a step ahead of the slowest reader of the other half.
// The weights kernel hands over 1 KB per entry instead of 1.5 KB and does no splitting; the sweep splits every weight once
// per workgroup instead of once per wave (S2_X6).  Results are bit-identical to S2_X6P (the same split3).
__device__ __forceinline__ void s2_split_coop(uint32_t st, uint32_t split_a, int wave, int half, int l31)
{
	float w[8];
	const uint32_t ra = st + 8192u + (uint32_t)wave * 2048u + (uint32_t)(8 * half) * 128u + (uint32_t)l31 * 4u;
	asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:128\n\tds_read_b32 %2, %8 offset:256\n\t"
		     "ds_read_b32 %3, %8 offset:384\n\tds_read_b32 %4, %8 offset:512\n\tds_read_b32 %5, %8 offset:640\n\t"
		     "ds_read_b32 %6, %8 offset:768\n\tds_read_b32 %7, %8 offset:896"
		     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
		     : "v"(ra) : "memory");
	S2_WAIT8(w);
	Op3 X;
	split8(w, X);
	const uint32_t da = split_a + (uint32_t)half * 6144u + (uint32_t)(32 * wave + l31) * 16u;
	const u32x4 t0 = {X.t[0][0].x, X.t[0][0].y, X.t[0][1].x, X.t[0][1].y};
	const u32x4 t1 = {X.t[1][0].x, X.t[1][0].y, X.t[1][1].x, X.t[1][1].y};
	const u32x4 t2 = {X.t[2][0].x, X.t[2][0].y, X.t[2][1].x, X.t[2][1].y};
	asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:2048\n\tds_write_b128 %0, %3 offset:4096\n\ts_waitcnt lgkmcnt(0)"
		     : : "v"(da), "v"(t0), "v"(t1), "v"(t2) : "memory");
}

// The same batch as an exact k-ordered fp32 fma chain: one v_mfma_f32_32x32x2_f32 per pair of entries and pixel block
// (lane half h = entry 2 p + h); bit-identical to the contract (the closing T * bg entry is its final fma(T, bg, acc)).
template <int B0, int B1, int B2, int B3>
__device__ __forceinline__ void s2_compute_exact(uint32_t st, int cg, int half, int l31)
{
	const uint32_t fa = st + (uint32_t)(half * 128 + cg * 32 + l31) * 4u;      // + pair * 1024
	const uint32_t wa = st + 8192u + (uint32_t)(half * 128 + l31) * 4u;       // + pair * 1024 + pb * 128
	float a, b0, b1, b2, b3;
	asm volatile(
		"ds_read_b32 %0, %5\n\t"
		"ds_read_b32 %1, %6\n\t"
		"ds_read_b32 %2, %6 offset:128\n\t"
		"ds_read_b32 %3, %6 offset:256\n\t"
		"ds_read_b32 %4, %6 offset:384\n\t"
		"s_waitcnt lgkmcnt(0)"
		: "=&v"(a), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
		: "v"(fa), "v"(wa)
		: "memory");
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int p = 0; p < 8; p++) {
		float an = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
		if (p < 7) {
			const uint32_t fn = fa + (uint32_t)(p + 1) * 1024u, wn = wa + (uint32_t)(p + 1) * 1024u;
			asm volatile(
				"ds_read_b32 %0, %5\n\t"
				"ds_read_b32 %1, %6\n\t"
				"ds_read_b32 %2, %6 offset:128\n\t"
				"ds_read_b32 %3, %6 offset:256\n\t"
				"ds_read_b32 %4, %6 offset:384"
				: "=&v"(an), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
				: "v"(fn), "v"(wn)
				: "memory");
		}
		asm volatile(
			"v_mfma_f32_32x32x2_f32 a[%c0:%c1], %8, %9, a[%c0:%c1]\n\t"
			"v_mfma_f32_32x32x2_f32 a[%c2:%c3], %8, %10, a[%c2:%c3]\n\t"
			"v_mfma_f32_32x32x2_f32 a[%c4:%c5], %8, %11, a[%c4:%c5]\n\t"
			"v_mfma_f32_32x32x2_f32 a[%c6:%c7], %8, %12, a[%c6:%c7]"
			: : "i"(B0 * 16), "i"(B0 * 16 + 15), "i"(B1 * 16), "i"(B1 * 16 + 15), "i"(B2 * 16), "i"(B2 * 16 + 15),
			    "i"(B3 * 16), "i"(B3 * 16 + 15), "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
			: S2_ACC);
		if (p < 7) {
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an), "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3) : : "memory");
			__builtin_amdgcn_sched_barrier(0);
			a = an; b0 = n0; b1 = n1; b2 = n2; b3 = n3;
		}
	}
}

// ---- stores.  Register r of a block is channel plane (r & 3) + 8 (r >> 2) (+ 4 for lanes >= 32); lanes 0-15 / 16-31
// of a block are parity rows 2 pb / 2 pb + 1.  v_permlane16_swap of the left and the right tile's registers makes two
// registers of 32 consecutive pixels each (image rows y and y + 2): every store writes two complete 128-B lines.
// `global_store_dword voffset, data, s[base] nt`: the channel plane is a wave-uniform SGPR base, one 32-bit byte offset
// per lane and row; nt because the image is written once.  Four registers (the four planes r0 .. r0 + 3) per statement;
// s_nop 1: a VALU write (the accvgpr reads) -> v_permlane16_swap needs two wait states.
// SMODE (development ablations, wrong pixels, same bytes): 1 = the registers straight from the accumulator file (no
// v_accvgpr_read, no permlane: what do those cost?), 2 = the same 2 KB as two 16-byte-per-lane stores of whole lines
// (8 planes x one row each: is the store path bound by instructions or by bytes?)
template <int BL, int BR, int R0, int SMODE>
__device__ __forceinline__ void s2_store4(const float* ubase, uint32_t o0, uint32_t o1, uint64_t plane, uint32_t wdelta)
{
	static_assert((R0 & 3) == 0, "four consecutive planes");
	const uint64_t p0 = (uint64_t)ubase + (uint64_t)(8 * (R0 >> 2)) * plane, p1 = p0 + plane, p2 = p1 + plane, p3 = p2 + plane;
	if constexpr (SMODE == 3) {
		// Round 4 (ping-pong sweep): the same bytes as TWO global_store_dwordx4.  After the permlane16 swap lane l31 of register
		// k holds pixel xp + l31 of plane R0 + k; a 4 x 4 transpose inside every quad of lanes (two rounds of quad_perm DPP +
		// bit merge) makes lane qi of a quad hold four CONSECUTIVE pixels of plane R0 + qi, so one store instruction writes
		// eight complete 128-B lines and a tile pair is 32 store instructions instead of 128.  Why it matters here and did
		// not in the kernel above: the wave's loads, LDS-DMAs and stores share ONE 6-bit counter of outstanding operations;
		// a pair's 64 immediate stores on top of three bundles in flight run into it and the wave stalls for a memory
		// latency -- in the ping-pong workgroup the partner half then waits at the barrier too.
		// o0 / o1 here: byte offsets of [4 half + (lane & 3)][row][x pair + 4 ((lane & 31) >> 2)]; wdelta = the lane's masks
		// (bit 0: lane & 1, bit 1: lane & 2).  The masks are bit merges on purpose (as ?: the compiler branches, and a DPP move
		// under a partial exec mask cannot read the disabled lanes).
		uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
		asm volatile(
			"v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c12]\n\t"
			"v_accvgpr_read_b32 %2, a[%c9]\n\tv_accvgpr_read_b32 %3, a[%c13]\n\t"
			"v_accvgpr_read_b32 %4, a[%c10]\n\tv_accvgpr_read_b32 %5, a[%c14]\n\t"
			"v_accvgpr_read_b32 %6, a[%c11]\n\tv_accvgpr_read_b32 %7, a[%c15]\n\t"
			"s_nop 1\n\t"
			"v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
			"s_nop 1"
			: "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
			: "i"(BL * 16 + R0), "i"(BL * 16 + R0 + 1), "i"(BL * 16 + R0 + 2), "i"(BL * 16 + R0 + 3),
			  "i"(BR * 16 + R0), "i"(BR * 16 + R0 + 1), "i"(BR * 16 + R0 + 2), "i"(BR * 16 + R0 + 3));
		const uint32_t m1 = (wdelta & 1u) ? 0xFFFFFFFFu : 0u, m2 = (wdelta & 2u) ? 0xFFFFFFFFu : 0u;
		auto sel = [](uint32_t m, uint32_t a, uint32_t b) __attribute__((always_inline)) { return (a & m) | (b & ~m); };
		auto qx1 = [](uint32_t v) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); };
		auto qx2 = [](uint32_t v) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); };
#pragma unroll
		for (int w = 0; w < 2; w++) {   // row y (t0 t2 t4 t6) and row y + 2 (t1 t3 t5 t7)
			const uint32_t v0 = w ? t1 : t0, v1 = w ? t3 : t2, v2 = w ? t5 : t4, v3 = w ? t7 : t6;
			const uint32_t x0 = sel(m1, qx1(v1), v0), x1 = sel(m1, v1, qx1(v0));
			const uint32_t x2 = sel(m1, qx1(v3), v2), x3 = sel(m1, v3, qx1(v2));
			const uint32_t y0 = sel(m2, qx2(x2), x0), y2 = sel(m2, x2, qx2(x0));
			const uint32_t y1 = sel(m2, qx2(x3), x1), y3 = sel(m2, x3, qx2(x1));
			const u32x4 d = {y0, y1, y2, y3};
			// (s_nop: a store of more than 64 bits is still reading its data registers when the next VALU may overwrite them)
			asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(w ? o1 : o0), "v"(d), "s"(p0) : "memory");
		}
		return;
	}
	if constexpr (SMODE == 2) {
		asm volatile(
			"global_store_dwordx4 %2, a[%c0:%c0+3], %4 nt\n\t"
			"global_store_dwordx4 %3, a[%c1:%c1+3], %4 nt\n\ts_nop 1"
			: : "i"(BL * 16 + R0), "i"(BR * 16 + R0), "v"(o0 + wdelta), "v"(o1 + wdelta), "s"(p0) : "memory");
		return;
	}
	if constexpr (SMODE == 1) {
		asm volatile(
			"global_store_dword %8, a[%c0], %10 nt\n\tglobal_store_dword %9, a[%c4], %10 nt\n\t"
			"global_store_dword %8, a[%c1], %11 nt\n\tglobal_store_dword %9, a[%c5], %11 nt\n\t"
			"global_store_dword %8, a[%c2], %12 nt\n\tglobal_store_dword %9, a[%c6], %12 nt\n\t"
			"global_store_dword %8, a[%c3], %13 nt\n\tglobal_store_dword %9, a[%c7], %13 nt"
			: : "i"(BL * 16 + R0), "i"(BL * 16 + R0 + 1), "i"(BL * 16 + R0 + 2), "i"(BL * 16 + R0 + 3),
			    "i"(BR * 16 + R0), "i"(BR * 16 + R0 + 1), "i"(BR * 16 + R0 + 2), "i"(BR * 16 + R0 + 3),
			    "v"(o0), "v"(o1), "s"(p0), "s"(p1), "s"(p2), "s"(p3)
			: "memory");
		return;
	}
	uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
	asm volatile(
		"v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c12]\n\t"
		"v_accvgpr_read_b32 %2, a[%c9]\n\tv_accvgpr_read_b32 %3, a[%c13]\n\t"
		"v_accvgpr_read_b32 %4, a[%c10]\n\tv_accvgpr_read_b32 %5, a[%c14]\n\t"
		"v_accvgpr_read_b32 %6, a[%c11]\n\tv_accvgpr_read_b32 %7, a[%c15]\n\t"
		"s_nop 1\n\t"
		"v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
		"global_store_dword %16, %0, %18 nt\n\tglobal_store_dword %17, %1, %18 nt\n\t"
		"global_store_dword %16, %2, %19 nt\n\tglobal_store_dword %17, %3, %19 nt\n\t"
		"global_store_dword %16, %4, %20 nt\n\tglobal_store_dword %17, %5, %20 nt\n\t"
		"global_store_dword %16, %6, %21 nt\n\tglobal_store_dword %17, %7, %21 nt"
		: "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
		: "i"(BL * 16 + R0), "i"(BL * 16 + R0 + 1), "i"(BL * 16 + R0 + 2), "i"(BL * 16 + R0 + 3),
		  "i"(BR * 16 + R0), "i"(BR * 16 + R0 + 1), "i"(BR * 16 + R0 + 2), "i"(BR * 16 + R0 + 3),
		  "v"(o0), "v"(o1), "s"(p0), "s"(p1), "s"(p2), "s"(p3)
		: "memory");
}

template <int BL, int BR, int R0, int NR, int SMODE>
__device__ __forceinline__ void s2_store_rows(const float* ubase, uint32_t o0, uint32_t o1, uint64_t plane, uint32_t wdelta)
{
	s2_store4<BL, BR, R0, SMODE>(ubase, o0, o1, plane, wdelta);
	if constexpr (NR > 4) s2_store_rows<BL, BR, R0 + 4, NR - 4, SMODE>(ubase, o0, o1, plane, wdelta);
}

// a pair at the image edge: per-store predication.  bp = &out[c0 + 4 half][ty * 16 + g + 4 pb][xl0 + (lane & 31)]
template <int BL, int BR>
__device__ __forceinline__ void s2_store_pair_guarded(float* bp, size_t HW, int PW, bool ok0, bool ok1)
{
	float l[16], r[16];
	acc_read<BL, true>(l);
	acc_read<BR, false>(r);
#pragma unroll
	for (int i = 0; i < 16; i++) {
		const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(l[i]), __float_as_uint(r[i]), false, false);
		float* dst = bp + (size_t)((i & 3) + 8 * (i >> 2)) * HW;
		if (ok0) *dst = __uint_as_float(sw[0]);
		if (ok1) dst[2 * (size_t)PW] = __uint_as_float(sw[1]);
	}
}

// a half with no partner in this segment (segment ends): 64-B pieces.
// bp = &out[c0 + 4 half][ty * 16 + g + 2 ((lane >> 4) & 1) + 4 pb][x0 + (lane & 15)]
template <int BLK>
__device__ __forceinline__ void s2_store_single(float* bp, size_t HW, bool ok)
{
	float v[16];
	acc_read<BLK, true>(v);
#pragma unroll
	for (int r = 0; r < 16; r++)
		if (ok) bp[(size_t)((r & 3) + 8 * (r >> 2)) * HW] = v[r];
}

// NORM mode (the per-pixel L2 norm the reference's consumer divides by, eval_segmentation.py:155): the
// feature map is NOT written; every finished accumulator block adds, per pixel, the sum of the squares of this wave's 32
// channels into an (H, W) plane.  Block register r of lane l is channel (r & 3) + 8 (r >> 2) + 4 (l >> 5) of pixel
// l & 31: sixteen squares per lane, the two channel halves meet through ds_bpermute, lanes 0-31 issue the atomic.
template <int BLK>
__device__ __forceinline__ void s2_norm_block(float* plane, int PW, int W, int H, int x0, int yb, int l31, int lane)
{
	float v[16];
	acc_read<BLK, true>(v);
	float ss = 0.f;
#pragma unroll
	for (int r = 0; r < 16; r++) ss = __builtin_fmaf(v[r], v[r], ss);
	const float other = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane ^ 32) << 2), __builtin_bit_cast(int, ss)));
	ss += other;
	const int x = x0 + (l31 & 15), y = yb + 2 * ((l31 >> 4) & 1);
	if (lane < 32 && x < W && y < H) atomicAdd(plane + (size_t)y * PW + x, ss);
}

} // namespace

enum { S2_EXACT = 0, S2_X6 = 1, S2_X6W = 2, S2_X6S = 3, S2_X6P = 4, S2_X6PW = 5, S2_X6C = 6 };   // X6C: fp32 weights handed over, split ONCE per workgroup (each wave a quarter) into LDS   // X6PW: X6P on v_mfma_f32_32x32x16_bf16   // X6P: weights pre-split by the weights kernel   // X6S: the six products block by block (the first form; A/B)

// DBG (development ablations, 0 in production): 1 = no stores, 2 = no matrix work, 4 / 8 = store ablations (s2_store4).
template <int ARITH, int DBG>
__global__ __launch_bounds__(256, 2) void blend_accum_sweep2_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const char* __restrict__ wgt, const float* __restrict__ features,
	const float* __restrict__ bg, float* __restrict__ out, const uint32_t* __restrict__ counter,
	int W, int H, int C, int gx, int nchunks_c, int seg, int nseg, int per_xcd, int total_items, int PW,
	unsigned long long* __restrict__ trace, const uint32_t* __restrict__ order, int dealt)
{
	if (counter[1] != 0u) return;   // arena overflowed / frame aborted
#ifdef SGS_SWEEP_EXCLUSIVE
	// EXPERIMENT: claim accumulator registers up to a[151]: 104 VGPRs + 152 AGPRs = 256 = half a SIMD's file, so nothing
	// else fits on a SIMD beside two waves of this kernel
	if (ARITH == S2_X6PW) asm volatile("" : : : "a151");
#endif
	const int b = blockIdx.x;
	int chunk, g, rest;   // 128-channel chunk, row parity, segment (= ty * nseg + sg) of this workgroup
	if (dealt) {   // segments dealt to the XCDs in serpentine order of their rank (sweep_plan_kernel)
		const int x = b & 7, pos = b >> 3, sib = 2 * nchunks_c;
		const int m = pos / sib, w = pos - m * sib;
		const int k = 16 * (m >> 1) + ((m & 1) ? 15 - x : x);
		if (k >= total_items) return;
		chunk = w % nchunks_c;
		g = w / nchunks_c;
		rest = order ? (int)order[k] : k;
	} else {
		const int v = (b & 7) * per_xcd + (b >> 3);
		if (v >= total_items) return;
		chunk = v % nchunks_c;
		g = (v / nchunks_c) & 1;
		rest = v / (2 * nchunks_c);
	}
	const unsigned long long t_begin = trace ? wall_clock64() : 0ull;
	const int stagger = (PW & 31) == 16 ? 1 : 0;   // odd rows start 64 B into a line
	const int sg = rest % nseg, ty = rest / nseg;
	const int tx0 = sg * seg;   // even
	const int nt = (gx - tx0) < seg ? (gx - tx0) : seg;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int cg = wave;
	const int half = lane >> 5, l31 = lane & 31;
	const int cbase = chunk * 128;
	const int c0 = cbase + cg * 32;
	const size_t HW = (size_t)H * PW;

	constexpr bool PRE = ARITH == S2_X6P || ARITH == S2_X6PW;   // the ARENA holds three bf16 terms per weight
	constexpr bool COOP = ARITH == S2_X6C;
	constexpr int KIND = PRE ? 1 : (COOP ? 2 : 0);
	constexpr int S2_NST = RingCfg<KIND>::NST, S2_LA = RingCfg<KIND>::LA, S2_STAGE = RingCfg<KIND>::STAGE;
	__shared__ float4 s_ring[S2_NST * S2_STAGE / 16];
	__shared__ float4 s_split[COOP ? 2 * 12288 / 16 : 1];
	constexpr int JMAX = COOP ? 384 : S2_JMAX;   // (the double split buffer takes the room of 5 KB of table)
	__shared__ uint2 s_bt[JMAX];   // .x = first arena slot of the batch, .y = entries | tile in segment << 8 | last of tile << 16
	__shared__ uint32_t s_tot[S2_SEGMAX], s_cb[S2_SEGMAX], s_pref[S2_SEGMAX + 1];

	// ---- prologue: the segment's batches as one flat table (ordinary accesses: nothing is in flight yet)
	if ((int)threadIdx.x < nt) {
		const int tile = ty * gx + tx0 + threadIdx.x;
		s_tot[threadIdx.x] = nact[tile];   // >= 1: every tile ends with the T * bg pseudo entry
		s_cb[threadIdx.x] = (ranges[tile].x >> 7) + (uint32_t)tile;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t acc = 0;
		for (int t = 0; t < nt; t++) {
			s_pref[t] = acc;
			acc += (s_tot[t] + AB - 1) / AB;
		}
		s_pref[nt] = acc;
	}
	__syncthreads();
	const uint32_t J = s_pref[nt];
	auto fill_table = [&](uint32_t wbase) __attribute__((always_inline)) {
		uint32_t maxnb = 0;
		for (int t = 0; t < nt; t++) maxnb = max(maxnb, s_pref[t + 1] - s_pref[t]);
		for (int tb = 0; tb < nt; tb += 8)
			for (uint32_t qb = 0; qb < maxnb; qb += 32) {
				const int t = tb + (int)(threadIdx.x >> 5);
				const uint32_t q = qb + (threadIdx.x & 31);
				if (t < nt) {
					const uint32_t p0 = s_pref[t], nb = s_pref[t + 1] - p0;
					if (q < nb && p0 + q >= wbase && p0 + q < wbase + JMAX) {
						const uint32_t tot = s_tot[t], first = q * AB;
						const uint32_t slot = sgs_chunk_start(table, s_cb[t], (uint32_t)(ty * gx + tx0 + t), first >> 7) + (first & 127u);
						const uint32_t n = (tot - first) < (uint32_t)AB ? (tot - first) : (uint32_t)AB;
						s_bt[p0 + q - wbase] = make_uint2(slot, n | ((uint32_t)t << 8) | (q + 1 == nb ? 1u << 16 : 0u));
					}
				}
			}
		if (threadIdx.x < 2 * S2_LA && J + threadIdx.x >= wbase && J + threadIdx.x - wbase < JMAX)
			s_bt[J + threadIdx.x - wbase] = make_uint2((uint32_t)(ty * gx + tx0 + nt - 1) * 128u, 1u | ((uint32_t)(nt - 1) << 8));
	};
	fill_table(0);
	__syncthreads();

	const uint32_t ring = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_ring;
	const uint32_t bt_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_bt;
	const uint32_t split_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_split;
	const uint32_t sub = (uint32_t)(4 * wave + half);   // this lane fetches the feature rows of entries sub and sub + 2
	const uint32_t my_ids = 8192u + (uint32_t)RingCfg<KIND>::WBYTES + (uint32_t)wave * 256u;   // this wave's id words inside a stage
	// bundle = features + this parity's weights of the batch at `slot` into stage st, then the ids of the batch
	// (slot2, n2) into the wave's id words -- LAST, so that their arrival means the wave's whole bundle arrived.
	// The id words hold the sentinel at that point: a wave writes it right after it has read a stage's ids (one step
	// before the stage is refilled), the prologue for the first NST bundles.
	// The bundle is issued in NPIECE pieces (one DMA instruction each) so that the step can place them between its
	// MFMA statements.
	struct Bundle { uint32_t slot, id0, id1, slot2, n2, st; };
	constexpr int NPIECE = PRE ? 6 : 5;
	auto dma_piece = [&](auto I, const Bundle& bd) __attribute__((always_inline)) {
		constexpr int i = decltype(I)::value;
		if constexpr (i < 2) {
			const uint32_t id = i == 0 ? bd.id0 : bd.id1;
			const float* row = id == SGS_BG_ID ? bg : features + (size_t)id * C;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + cbase + l31 * 4),
							 (__attribute__((address_space(3))) void*)(size_t)(bd.st + (uint32_t)(4 * wave + 2 * i) * 512u), 16, 0, 0);
		} else if constexpr (i == NPIECE - 1) {
			const uint32_t li = (uint32_t)(lane & 15) < bd.n2 ? (uint32_t)(lane & 15) : bd.n2 - 1u;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(act_id + bd.slot2 + li),
							 (__attribute__((address_space(3))) void*)(size_t)(bd.st + my_ids), 4, 0, 0);
		} else if constexpr (PRE) {
			// three bf16 terms: per group of 8 entries [term][256 px'][8 x bf16]; this parity's 2 KB of each (group, term)
			// as two 1-KB pieces; piece p = 3 wave + i -> (group p / 6, term (p % 6) / 2, half p % 2)
			const int pc = 3 * wave + (i - 2);
			const char* wsrc = wgt + (size_t)((bd.slot >> 3) + (uint32_t)(pc / 6)) * 12288 + (size_t)((pc % 6) / 2) * 4096 +
					   (size_t)g * 2048 + (size_t)(pc % 2) * 1024 + (size_t)lane * 16;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
							 (__attribute__((address_space(3))) void*)(size_t)(bd.st + 8192u + (uint32_t)pc * 1024u), 16, 0, 0);
		} else if constexpr (COOP) {
			// fp32 weight rows of 1 KB per entry: the 128 bytes [32 wave .. 32 wave + 31] px' of this parity, entries
			// 8 (i - 2) + (lane >> 3), as 16-byte pieces -> this wave's [entry][32 px'] block of the stage
			const char* wsrc = wgt + (size_t)(bd.slot + 8 * (i - 2) + (lane >> 3)) * 1024 + (size_t)g * 512 + (size_t)wave * 128 + (size_t)(lane & 7) * 16;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
							 (__attribute__((address_space(3))) void*)(size_t)(bd.st + 8192u + (uint32_t)wave * 2048u + (uint32_t)(i - 2) * 1024u), 16, 0, 0);
		} else {
			// fp32 weight rows of 1 KB per entry: this parity's 512 B of entries 4 wave .. 4 wave + 3
			const char* wsrc = wgt + (size_t)(bd.slot + 4 * wave + (lane >> 5)) * 1024 + (size_t)g * 512 + (size_t)(lane & 31) * 16;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (i - 2) * 2048),
							 (__attribute__((address_space(3))) void*)(size_t)(bd.st + 8192u + (uint32_t)(wave * 2 + (i - 2)) * 1024u), 16, 0, 0);
		}
	};
	auto issue_all = [&](const Bundle& bd) __attribute__((always_inline)) {
		dma_piece(std::integral_constant<int, 0>{}, bd);
		dma_piece(std::integral_constant<int, 1>{}, bd);
		dma_piece(std::integral_constant<int, 2>{}, bd);
		dma_piece(std::integral_constant<int, 3>{}, bd);
		dma_piece(std::integral_constant<int, 4>{}, bd);
		if constexpr (NPIECE == 6) dma_piece(std::integral_constant<int, 5>{}, bd);
	};

	// the eight accumulator blocks a[0:127] (see acc_zero)
	acc_zero<0>(); acc_zero<1>(); acc_zero<2>(); acc_zero<3>(); acc_zero<4>(); acc_zero<5>(); acc_zero<6>(); acc_zero<7>();

	uint32_t wbase = 0;
#pragma unroll
	for (int k = 0; k < S2_NST; k++)   // every stage's id words start as "not landed"
		asm volatile("ds_write_b32 %0, %1" : : "v"(ring + (uint32_t)k * S2_STAGE + my_ids + (uint32_t)lane * 4u), "v"(S2_SENT) : "memory");
	asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
#pragma unroll
	for (int k = 0; k < S2_LA; k++) {   // prologue bundles 0 .. LA-1 (their feature ids by ordinary loads)
		const uint2 e = s_bt[k], e2 = s_bt[k + S2_LA];
		const uint32_t n = e.y & 255u;
		const uint32_t id0 = act_id[e.x + (sub < n ? sub : n - 1u)];
		const uint32_t id1 = act_id[e.x + (sub + 2u < n ? sub + 2u : n - 1u)];
		issue_all(Bundle{e.x, id0, id1, e2.x, e2.y & 255u, ring + (uint32_t)k * S2_STAGE});
	}
	uint32_t st0 = ring, stI = ring + S2_LA * S2_STAGE;   // stages of batch j and of bundle j + LA
	uint32_t j = 0;

	// store addressing of the pair whose blocks 2, 3 are still to be written (deferred), and how far that is
	const float* const ubase = out + (size_t)c0 * HW;
	const uint64_t plane = (uint64_t)HW * 4u;
	uint32_t d_o0 = 0u;    // byte offset of [4 half][y0 + 8][xp] (block 2's first row)
	int dprog = 4;         // chunks of 16 stores issued (4 = nothing pending)

	// ---- one batch: table entry, bundle arrival, barrier, next bundle; returns the entry word of batch j
	Bundle nb;   // the bundle this step issues (j + LA)
	uint32_t late = 0;   // (trace) steps of this wave whose bundle had not landed when the step began
	auto batch_head = [&]() __attribute__((always_inline)) -> uint32_t {
		if (j + 2 * S2_LA >= wbase + JMAX) {   // (uniform, long segments only) slide the table window
			__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
			__syncthreads();
			wbase = j;
			fill_table(wbase);
			__syncthreads();
		}
		// ONE LDS round trip: the three table words of this step, this lane's id word of bundle j (arrival check) and
		// the two feature-row ids this lane fetches for bundle j + LA (they sit in the same id words)
		const uint32_t a = bt_a + (j - wbase) * 8u;
		const uint32_t pa = st0 + my_ids + (uint32_t)lane * 4u, ia = st0 + my_ids + sub * 4u;
		uint32_t r0, r1, wv, id0, id1;
		uint64_t rd;
		asm volatile(
			"ds_read_b32 %0, %6 offset:4\n\t"
			"ds_read_b32 %1, %6 offset:%9\n\t"
			"ds_read_b64 %2, %6 offset:%10\n\t"
			"ds_read_b32 %3, %7\n\t"
			"ds_read_b32 %4, %8\n\t"
			"ds_read_b32 %5, %8 offset:8\n\t"
			"s_waitcnt lgkmcnt(0)"
			: "=&v"(r0), "=&v"(r1), "=&v"(rd), "=&v"(wv), "=&v"(id0), "=&v"(id1)
			: "v"(a), "v"(pa), "v"(ia), "n"(S2_LA * 8), "n"(2 * S2_LA * 8)
			: "memory");
		if (__builtin_amdgcn_ballot_w64(wv == S2_SENT) != 0ull) {   // not landed yet: poll (no vmcnt: stores may be outstanding in any number)
			if (trace) late++;   // (tools/sweep_trace.py: how often a step finds its bundle still in flight)
			int spins = 0;
			do {
				if (++spins > (1 << 22)) __builtin_trap();   // (a lost bundle must not hang the device)
				__builtin_amdgcn_s_sleep(1);
				asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %4 offset:8\n\ts_waitcnt lgkmcnt(0)"
					     : "=&v"(wv), "=&v"(id0), "=&v"(id1) : "v"(pa), "v"(ia) : "memory");
			} while (__builtin_amdgcn_ballot_w64(wv == S2_SENT) != 0ull);
		}
		// the ids are in registers: this stage's id words go back to "not landed" for its next bundle (issued next step)
		asm volatile("ds_write_b32 %0, %1" : : "v"(pa), "v"(S2_SENT) : "memory");
		const uint32_t e0y = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0);
		nb.slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)r1);
		nb.slot2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rd);
		nb.n2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rd >> 32)) & 255u;
		nb.id0 = id0;
		nb.id1 = id1;
		nb.st = stI;   // the stage batch j - 1 was computed from
		if constexpr (COOP && !(DBG & 2)) s2_split_coop(st0, split_a + (j & 1u) * 12288u, wave, half, l31);   // (this wave's bundle has landed)
		__builtin_amdgcn_s_barrier();
		return e0y;
	};
	auto batch_tail = [&]() __attribute__((always_inline)) {
		st0 = st0 + S2_STAGE == ring + S2_NST * S2_STAGE ? ring : st0 + S2_STAGE;
		stI = stI + S2_STAGE == ring + S2_NST * S2_STAGE ? ring : stI + S2_STAGE;
		j++;
	};

// the step's work behind the barrier: issue bundle j + LA, multiply batch j.  The pre-split path threads the DMA pieces
// between its MFMA statements; the others issue the bundle first.
#define S2_COMPUTE(b0_, b1_, b2_, b3_)                                                               \
	do {                                                                                             \
		if ((DBG & 2) || !(PRE || COOP)) issue_all(nb);                                              \
		if (!(DBG & 2)) {                                                                            \
			if (ARITH == S2_EXACT) s2_compute_exact<b0_, b1_, b2_, b3_>(st0, cg, half, l31);          \
			else if (PRE) s2_compute_x6p<ARITH == S2_X6PW, b0_, b1_, b2_, b3_, NPIECE>(st0, st0 + 8192u, cg, half, l31, [&](auto I) __attribute__((always_inline)) { dma_piece(I, nb); }); \
			else if (COOP) {                                                                         \
				s2_compute_x6p<false, b0_, b1_, b2_, b3_, NPIECE>(st0, split_a + (j & 1u) * 12288u, cg, half, l31, [&](auto I) __attribute__((always_inline)) { dma_piece(I, nb); }); \
			} else s2_compute_x6<ARITH == S2_X6W, ARITH == S2_X6, b0_, b1_, b2_, b3_>(st0, cg, half, l31); \
		}                                                                                            \
	} while (0)

	constexpr bool NORM = (DBG & 32) != 0;   // sum-of-squares plane instead of the feature map
	const bool skip_stores = (DBG & 1) != 0;
	constexpr int SMODE = (DBG & 4) ? 2 : ((DBG & 8) ? 1 : 0);
	const uint32_t wdelta = (((uint32_t)(lane >> 3) * (uint32_t)HW + 4u * (uint32_t)(lane & 7)) - ((uint32_t)(4 * half) * (uint32_t)HW + (uint32_t)l31)) * 4u;
// byte offset (from ubase) this lane's stores of a pair start at: row y_, the pair's left tile column txl_
#define S2_PAIR_OFF(y_, txl_) ((((uint32_t)(4 * half) * (uint32_t)HW + (uint32_t)((y_) * PW + (txl_) * SGS_TILE + l31)) * 4u))
// 16 stores of the deferred half of the pair finished under mapping MP_: chunk c = block 2 + (c >> 1), registers 8 (c & 1) ..
#define S2_DEFERRED_CHUNK(MP_)                                                                       \
	do {                                                                                             \
		if (dprog < 4) {                                                                             \
			const uint32_t o0 = d_o0 + (uint32_t)((dprog >> 1) * 4 * PW) * 4u, o1 = o0 + (uint32_t)(2 * PW) * 4u; \
			switch (dprog) {                                                                         \
			case 0: s2_store_rows<LB(MP_, 2), RB(MP_, 2), 0, 8, SMODE>(ubase, o0, o1, plane, wdelta); break;         \
			case 1: s2_store_rows<LB(MP_, 2), RB(MP_, 2), 8, 8, SMODE>(ubase, o0, o1, plane, wdelta);                \
				acc_zero<LB(MP_, 2)>(); acc_zero<RB(MP_, 2)>(); break;                               \
			case 2: s2_store_rows<LB(MP_, 3), RB(MP_, 3), 0, 8, SMODE>(ubase, o0, o1, plane, wdelta); break;         \
			default: s2_store_rows<LB(MP_, 3), RB(MP_, 3), 8, 8, SMODE>(ubase, o0, o1, plane, wdelta);               \
				acc_zero<LB(MP_, 3)>(); acc_zero<RB(MP_, 3)>(); break;                               \
			}                                                                                        \
			dprog++;                                                                                 \
		}                                                                                            \
	} while (0)

// a whole tile accumulated into the left blocks of mapping M_, the deferred stores of the previous pair (finished
// under the other mapping) riding along; tx_ = the tile's column afterwards
#define S2_LEFT_TILE(M_, tx_)                                                                        \
	do {                                                                                             \
		uint32_t e_;                                                                                 \
		do {                                                                                         \
			e_ = batch_head();                                                                       \
			S2_COMPUTE(LB(M_, 0), LB(M_, 1), LB(M_, 2), LB(M_, 3));                                  \
			S2_DEFERRED_CHUNK(1 - (M_));                                                             \
			batch_tail();                                                                            \
		} while ((e_ >> 16) == 0u);                                                                  \
		while (dprog < 4) S2_DEFERRED_CHUNK(1 - (M_));                                               \
		tx_ = tx0 + (int)((e_ >> 8) & 255u);                                                         \
	} while (0)
#define S2_RIGHT_TILE(M_, tx_)                                                                       \
	do {                                                                                             \
		uint32_t e_;                                                                                 \
		do {                                                                                         \
			e_ = batch_head();                                                                       \
			S2_COMPUTE(RB(M_, 0), RB(M_, 1), RB(M_, 2), RB(M_, 3));                                  \
			batch_tail();                                                                            \
		} while ((e_ >> 16) == 0u);                                                                  \
		tx_ = tx0 + (int)((e_ >> 8) & 255u);                                                         \
	} while (0)

	const int y0 = ty * SGS_TILE + g;   // first image row of this parity in the tile row
	const int hi = (l31 >> 4) & 1;

// the half rows of a tile with no partner in this segment (64-B pieces), blocks b0_..b3_ of tile column tx_
#define S2_NORM_TILE(b0_, b1_, b2_, b3_, tx_)                                                        \
	do {                                                                                             \
		s2_norm_block<b0_>(out, PW, W, H, (tx_) * SGS_TILE, y0, l31, lane);                          \
		s2_norm_block<b1_>(out, PW, W, H, (tx_) * SGS_TILE, y0 + 4, l31, lane);                      \
		s2_norm_block<b2_>(out, PW, W, H, (tx_) * SGS_TILE, y0 + 8, l31, lane);                      \
		s2_norm_block<b3_>(out, PW, W, H, (tx_) * SGS_TILE, y0 + 12, l31, lane);                     \
	} while (0)
#define S2_STORE_SINGLE(b0_, b1_, b2_, b3_, tx_)                                                     \
	do {                                                                                             \
		if (NORM) {                                                                                  \
			S2_NORM_TILE(b0_, b1_, b2_, b3_, tx_);                                                   \
		} else if (!skip_stores) {                                                                   \
			const int xs_ = (tx_) * SGS_TILE + (l31 & 15);                                           \
			float* p_ = out + (size_t)(c0 + 4 * half) * HW + (size_t)(y0 + 2 * hi) * PW + xs_;        \
			s2_store_single<b0_>(p_, HW, xs_ < W && y0 + 2 * hi < H);                                 \
			s2_store_single<b1_>(p_ + (size_t)4 * PW, HW, xs_ < W && y0 + 2 * hi + 4 < H);            \
			s2_store_single<b2_>(p_ + (size_t)8 * PW, HW, xs_ < W && y0 + 2 * hi + 8 < H);            \
			s2_store_single<b3_>(p_ + (size_t)12 * PW, HW, xs_ < W && y0 + 2 * hi + 12 < H);          \
		}                                                                                            \
		acc_zero<b0_>(); acc_zero<b1_>(); acc_zero<b2_>(); acc_zero<b3_>();                          \
	} while (0)

#define S2_ZERO_ALL() do { acc_zero<0>(); acc_zero<1>(); acc_zero<2>(); acc_zero<3>(); acc_zero<4>(); acc_zero<5>(); acc_zero<6>(); acc_zero<7>(); } while (0)

// pair (tx_ - 1, tx_) finished under mapping M_: blocks 0, 1 now, blocks 2, 3 deferred (interior) / everything now (edge).
// The blocks read here were last written at least two MFMA groups before the tile's last instruction (blocks 0, 1),
// the s_nop covers the rest of that latency.
#define S2_PAIR_DONE(M_, tx_)                                                                        \
	do {                                                                                             \
		const int xp_ = ((tx_) - 1) * SGS_TILE + l31;                                                \
		const bool inside_ = ((tx_) + 1) * SGS_TILE <= W && y0 + 14 < H;   /* (uniform) */           \
		if (NORM) {                                                                                  \
			S2_NORM_TILE(LB(M_, 0), LB(M_, 1), LB(M_, 2), LB(M_, 3), (tx_) - 1);                     \
			S2_NORM_TILE(RB(M_, 0), RB(M_, 1), RB(M_, 2), RB(M_, 3), tx_);                           \
			S2_ZERO_ALL();                                                                           \
		} else if (skip_stores) {                                                                    \
			S2_ZERO_ALL();                                                                           \
		} else if (inside_) {                                                                        \
			const uint32_t o0_ = S2_PAIR_OFF(y0, (tx_) - 1);                                          \
			const uint32_t o1_ = o0_ + (uint32_t)(2 * PW) * 4u;                                      \
			asm volatile("s_nop 15" : : : "memory");                                                 \
			s2_store_rows<LB(M_, 0), RB(M_, 0), 0, 16, SMODE>(ubase, o0_, o1_, plane, wdelta);                      \
			s2_store_rows<LB(M_, 1), RB(M_, 1), 0, 16, SMODE>(ubase, o0_ + (uint32_t)(4 * PW) * 4u, o1_ + (uint32_t)(4 * PW) * 4u, plane, wdelta); \
			acc_zero<LB(M_, 0)>(); acc_zero<RB(M_, 0)>(); acc_zero<LB(M_, 1)>(); acc_zero<RB(M_, 1)>(); \
			d_o0 = o0_ + (uint32_t)(8 * PW) * 4u;                                                    \
			dprog = 0;                                                                               \
		} else {                                                                                     \
			float* p_ = out + (size_t)(c0 + 4 * half) * HW + (size_t)y0 * PW + xp_;                   \
			s2_store_pair_guarded<LB(M_, 0), RB(M_, 0)>(p_, HW, PW, xp_ < W && y0 < H, xp_ < W && y0 + 2 < H); \
			s2_store_pair_guarded<LB(M_, 1), RB(M_, 1)>(p_ + (size_t)4 * PW, HW, PW, xp_ < W && y0 + 4 < H, xp_ < W && y0 + 6 < H); \
			s2_store_pair_guarded<LB(M_, 2), RB(M_, 2)>(p_ + (size_t)8 * PW, HW, PW, xp_ < W && y0 + 8 < H, xp_ < W && y0 + 10 < H); \
			s2_store_pair_guarded<LB(M_, 3), RB(M_, 3)>(p_ + (size_t)12 * PW, HW, PW, xp_ < W && y0 + 12 < H, xp_ < W && y0 + 14 < H); \
			S2_ZERO_ALL();                                                                           \
		}                                                                                            \
	} while (0)

	// ---- the sweep.  Even rows: even tiles are left halves; odd rows of a staggered pitch: odd tiles.
	int tx = tx0;
	if (J > 0 && ((tx0 + g * stagger) & 1) != 0) {   // the segment starts with a right half whose partner belongs to the previous segment
		S2_RIGHT_TILE(0, tx);
		S2_STORE_SINGLE(RB(0, 0), RB(0, 1), RB(0, 2), RB(0, 3), tx);
	}
	while (j < J) {
		S2_LEFT_TILE(0, tx);
		if (j >= J) { S2_STORE_SINGLE(LB(0, 0), LB(0, 1), LB(0, 2), LB(0, 3), tx); break; }
		S2_RIGHT_TILE(0, tx);
		S2_PAIR_DONE(0, tx);
		if (j >= J) { while (dprog < 4) S2_DEFERRED_CHUNK(0); break; }
		S2_LEFT_TILE(1, tx);
		if (j >= J) { S2_STORE_SINGLE(LB(1, 0), LB(1, 1), LB(1, 2), LB(1, 3), tx); break; }
		S2_RIGHT_TILE(1, tx);
		S2_PAIR_DONE(1, tx);
		if (j >= J) { while (dprog < 4) S2_DEFERRED_CHUNK(1); break; }
	}
#undef S2_COMPUTE
	__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // drain the dummy tail bundles before LDS is released
	if (trace && threadIdx.x == 0) {
		trace[4 * (size_t)b] = t_begin;
		trace[4 * (size_t)b + 1] = wall_clock64();
		trace[4 * (size_t)b + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
					   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
		trace[4 * (size_t)b + 3] = (unsigned long long)J | ((unsigned long long)nt << 32) | ((unsigned long long)late << 40);
	}
}


// ======================================================================================================================
// Round 4: the same sweep as ONE 8-wave workgroup for BOTH row parities of a segment -- "ping-pong" (blend variant sweep
// nibble 6; the default).  Why: with two independent 4-wave workgroups per CU (one per parity, the kernel above) the two
// waves that share a SIMD meet at a random phase: PMC showed the matrix pipe 59 % busy with 59 % of the wave-cycles in
// issue stalls, and one workgroup per CU alone ran only 24 % slower than two (profiles/r04_sweep_occupancy.txt) -- the
// second wave hides little.  Here the two waves of a SIMD are the two PARITIES of the same (segment, 32 channels) and
// their phases are made complementary by construction:
//
//     waves 0-3 (parity 0):  PREP(j)  |B|  MFMA(j)   |B|  PREP(j+1) |B|  MFMA(j+1) ...
//     waves 4-7 (parity 1):       |B|  PREP(j)   |B|  MFMA(j)   |B|  PREP(j+1) |B| ...
//
//   MFMA(j) = the 48 v_mfma_f32_32x32x8_bf16 of batch j, operands already in registers, with this wave's five DMA pieces of
//             bundle j + LA between the MFMA pairs of the second half (their issue hides under matrix time there);
//   PREP(j) = everything else: the step's table words, operand LDS reads + the feature split of batch j, the deferred /
//             pair stores.
// The code of the two halves is THE SAME loop (PREP, barrier, MFMA, barrier); the second half simply executes one extra
// s_barrier before it (and the first half one after it), i.e. it runs one barrier behind.  Every workgroup-wide barrier
// inside the common code therefore stays matched, and the only obligations are about data one half produces for both:
//   * ring stage of batch j: read by half 0 in [B2j, B2j+1), by half 1 in [B2j+1, B2j+2); refilled (bundle j + NST) by
//     DMA pieces issued in the MFMA phases of step j + 1, i.e. after B2j+3 -- never before the slower reader is done;
//   * arrival of bundle j + 1: half 0 reads it right after B2j+2, so EVERY wave checks its own pieces (the id words its
//     last DMA deposits, as above) before it arrives at B2j+2: half 0 at the end of MFMA(j), half 1 at the end of PREP(j);
//   * the batch-table window: refilled by half 0 only, between two barriers during which half 1 reads nothing.
// Each feature row is fetched ONCE for both parities (the kernel above fetched it once per parity: -0.4 GB of L2 -> LDS
// traffic per frame at cfg3) and the stage is 8 KB features + 2 x 12 KB pre-split weights + id words; 4 stages (139 KB of
// LDS, one workgroup per CU), three bundles in flight.  Waves 4-7 run at s_setprio 1 (the second-dispatched half loses
// the VALU arbitration otherwise: MI355X_MICROARCH.md "Two waves per SIMD", item 4).
// Stores, accumulator roles (mapping M), pairing of left / right halves, edge cases: exactly the kernel above -- per wave
// nothing changed but WHEN it does things.  Arithmetic: S2_X6P (pre-split weights, six products) only; results are
// bit-identical to variant 0x6E (same products, same order, same accumulators).
constexpr int S3_NST = 4, S3_LA = S3_NST - 1;
// stage = features | weights parity 0 | weights parity 1 | id words of 8 waves.  COOP (sweep nibble 5): the weights arrive as
// fp32 rows (8 KB per parity and batch instead of 12 KB of pre-split terms) and each wave splits the quarter it fetched into
// a per-parity split buffer ONE STEP AHEAD (s2_split_coop): the two barriers of a step make it visible to its three sister waves
constexpr int S3_WB(bool coop) { return coop ? 8192 : 12288; }
constexpr int S3_STAGE_OF(bool coop) { return 8192 + 2 * S3_WB(coop) + 8 * 256; }

// MM (make X16=1 only): 0 = the six products on v_mfma_f32_32x32x8_bf16 (the product); 1 = on the double-rate x16 MFMA, dense
// statements; 2 / 3 = x16 with one filler (s_nop 0 / a VALU move) between consecutive MFMAs -- the round-4 bisect of DESIGN.md 5.10
// FREE (sweep nibble 4, experiment): the same kernel WITHOUT the two barriers per step -- the halves run free of each other, bounded
// only by the data: a wave reads batch j when all eight waves' pieces of bundle j have landed (an arrival counter per ring stage,
// bumped by every wave when its own pieces are in), and refills a stage when all eight waves have read the batch it held (a
// consumption counter per stage).  Why: in lock step a half that is draining a tile pair's store burst keeps its partner at the
// barrier (15-20 % of the step); free-running, the partner works on until the ring stops it.
// STP (round 6, "store placement"; profiles/r06_sweep_phases.txt: the default form runs 1.02 ms with its stores and 0.70 without, and a finished
// pair's 64 immediate stores per wave are a burst during which the wave does nothing else -- the cost is the DRAIN of 128 KB per CU, paid in
// issue stalls): 0 = rounds 3-5 (blocks 0, 1 of both tiles at once when the pair completes, blocks 2, 3 in four chunks of 16 in the PREP phases
// of the next left tile's steps); 1 = only pixel block 0 at once; pixel block 1 leaves in the MATRIX phase of the next left tile's first step,
// behind its first twelve products -- for that the steps pair the pixel blocks (0, 2 | 1, 3) instead of (0, 1 | 2, 3), so that the first dense
// statement of the new tile only needs the two accumulator blocks the first 32 stores have freed (the per-block product order is untouched:
// bit-identical maps).  Sweep 1.017 -> 0.983 ms, order-balanced over three boxes (profiles/r06_sweep_store_placement*.txt).  Two more placements were built
// on this hook, measured and removed: the four deferred chunks in that matrix-phase slot as well (no store in a PREP phase at all): +2 %; the chunks behind
// the step's LAST products, in front of the arrival check's wait: +10 % (1.10 ms) -- stores at the end of a matrix phase delay the wave's own DMA pieces' arrival check.
template <int DBG, int MM = 0, bool COOP = false, bool FREE = false, int STP = 0>
__global__ __launch_bounds__(512, 2) void blend_accum_sweep3_kernel(
	const uint2* __restrict__ ranges, const uint32_t* __restrict__ table,
	const uint32_t* __restrict__ nact, const uint32_t* __restrict__ act_id,
	const char* __restrict__ wgt, const float* __restrict__ features,
	const float* __restrict__ bg, float* __restrict__ out_img, const uint32_t* __restrict__ counter,
	int W, int H_img, int C, int gx, int nchunks_c, int seg, int nseg, int per_xcd, int total_items, int PW,
	unsigned long long* __restrict__ trace, const uint32_t* __restrict__ order, int dealt, int tune, int bands)
{
	// (x16 forms) 144 AGPRs + 112 VGPRs = 256 registers per wave: two waves per SIMD fill its register file, so that NO foreign wave --
	// not even an 8-register fill kernel -- can be resident on this workgroup's compute unit between its first and its last matrix
	// instruction (all eight waves are resident from dispatch; the last matrix phase lies before the final barrier every wave passes)
	// (both halves of the count are pinned: the lock-step form happens to need 112 VGPRs, the free-running one 104 -- 248 would leave 16
	// registers per SIMD lane for a foreign wave; tests/test_code_object.py reads the counts back from the built library)
	if constexpr (MM != 0 && !COOP) asm volatile("" : : : "a143");
	if constexpr (MM != 0 && COOP) asm volatile("" : : : "a139");   // (experiment: the fp32 hand-over form needs 116 VGPRs: 140 + 116 = 256)
	if constexpr (MM != 0 && FREE) asm volatile("" : : : "v111");
	if (counter[1] != 0u) return;   // arena overflowed / frame aborted
	(void)tune;   // (tuning word, bits [19:16] of the blend variant: unused -- call E's placement / priority experiments are settled,
	// profiles/r04_sweep_dma_placement.txt: pieces issued in PREP cost 300-400 cycles each, per-phase priorities change nothing)
	const int b = blockIdx.x;
	int chunk, rest;   // 128-channel chunk, segment (= ty * nseg + sg) of this workgroup
	if (dealt) {   // segments dealt to the XCDs in serpentine order of their rank (sweep_plan_kernel)
		const int x = b & 7, pos = b >> 3, sib = nchunks_c;
		const int m = pos / sib;
		const int k = 16 * (m >> 1) + ((m & 1) ? 15 - x : x);
		if (k >= total_items) return;
		chunk = pos - m * sib;
		rest = order ? (int)order[k] : k;
	} else {
		const int v = (b & 7) * per_xcd + (b >> 3);
		if (v >= total_items) return;
		chunk = v % nchunks_c;
		rest = v / nchunks_c;
	}
	const unsigned long long t_begin = trace ? wall_clock64() : 0ull;
	const int stagger = (PW & 31) == 16 ? 1 : 0;   // odd rows start 64 B into a line
	const int sg = rest % nseg, ty = rest / nseg;
	// SGS_OPT_OUT_BANDS: this tile row's band is a (C, rows, PW) image of its own -- from here on `out` / `H` / `ty_out` are the band's base,
	// height and the tile row inside it; the lists are still the frame's (ty)
	float* __restrict__ out = out_img;
	int H = H_img, ty_out = ty;
	if (bands > 1) {
		int lo_tile, rows;
		sgs_band_of(ty, (H_img + SGS_TILE - 1) / SGS_TILE, bands, H_img, lo_tile, rows);
		out = out_img + (size_t)C * (size_t)PW * (size_t)(SGS_TILE * lo_tile);
		H = rows;
		ty_out = ty - lo_tile;
	}
	const int tx0 = sg * seg;   // even
	const int nt = (gx - tx0) < seg ? (gx - tx0) : seg;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int g = wave >> 2, cg = wave & 3;   // row parity (= half of the workgroup), channel group
	const int half = lane >> 5, l31 = lane & 31;
	const int cbase = chunk * 128;
	const int c0 = cbase + cg * 32;
	const size_t HW = (size_t)H * PW;

	constexpr int S3_STAGE = S3_STAGE_OF(COOP), WB = S3_WB(COOP);
	constexpr int NP = COOP ? 4 : 5;   // DMA pieces per wave and bundle: features | 3 pre-split / 2 fp32 weight pieces | ids
	__shared__ float4 s_ring[S3_NST * S3_STAGE / 16];
	__shared__ float4 s_split[COOP ? 4 * 12288 / 16 : 1];   // COOP: [parity][batch & 1] x 12 KB of split terms
	constexpr int JMAX = COOP ? 768 : S2_JMAX;   // (COOP: 163 KB of ring + split buffers + a 1024-batch window would not fit 160 KB)
	__shared__ uint2 s_bt[JMAX];   // .x = first arena slot of the batch, .y = entries | tile in segment << 8 | last of tile << 16
	__shared__ uint32_t s_tot[S2_SEGMAX], s_cb[S2_SEGMAX], s_pref[S2_SEGMAX + 1];
	__shared__ uint32_t s_flag[8];   // (FREE) [0..3] waves whose pieces of the stage's bundle have landed, [4..7] waves that have read the stage's batch (both cumulative)

	// ---- prologue: the segment's batches as one flat table (ordinary accesses: nothing is in flight yet)
	if (FREE && threadIdx.x < 8) s_flag[threadIdx.x] = 0u;
	if ((int)threadIdx.x < nt) {
		const int tile = ty * gx + tx0 + threadIdx.x;
		s_tot[threadIdx.x] = nact[tile];   // >= 1: every tile ends with the T * bg pseudo entry
		s_cb[threadIdx.x] = (ranges[tile].x >> 7) + (uint32_t)tile;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t acc = 0;
		for (int t = 0; t < nt; t++) {
			s_pref[t] = acc;
			acc += (s_tot[t] + AB - 1) / AB;
		}
		s_pref[nt] = acc;
	}
	__syncthreads();
	const uint32_t J = s_pref[nt];
	auto fill_table = [&](uint32_t wbase) __attribute__((always_inline)) {   // (the first half's 256 threads)
		if (threadIdx.x >= 256) return;
		uint32_t maxnb = 0;
		for (int t = 0; t < nt; t++) maxnb = max(maxnb, s_pref[t + 1] - s_pref[t]);
		for (int tb = 0; tb < nt; tb += 8)
			for (uint32_t qb = 0; qb < maxnb; qb += 32) {
				const int t = tb + (int)(threadIdx.x >> 5);
				const uint32_t q = qb + (threadIdx.x & 31);
				if (t < nt) {
					const uint32_t p0 = s_pref[t], nb = s_pref[t + 1] - p0;
					if (q < nb && p0 + q >= wbase && p0 + q < wbase + JMAX) {
						const uint32_t tot = s_tot[t], first = q * AB;
						const uint32_t slot = sgs_chunk_start(table, s_cb[t], (uint32_t)(ty * gx + tx0 + t), first >> 7) + (first & 127u);
						const uint32_t n = (tot - first) < (uint32_t)AB ? (tot - first) : (uint32_t)AB;
						s_bt[p0 + q - wbase] = make_uint2(slot, n | ((uint32_t)t << 8) | (q + 1 == nb ? 1u << 16 : 0u));
					}
				}
			}
		if (threadIdx.x < 2 * S3_LA && J + threadIdx.x >= wbase && J + threadIdx.x - wbase < JMAX)
			s_bt[J + threadIdx.x - wbase] = make_uint2((uint32_t)(ty * gx + tx0 + nt - 1) * 128u, 1u | ((uint32_t)(nt - 1) << 8));
	};
	fill_table(0);
	__syncthreads();

	const uint32_t ring = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_ring;
	const uint32_t bt_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_bt;
	const uint32_t sub = (uint32_t)(2 * wave + half);   // this lane fetches the feature row of entry sub (one 1-KB piece = two rows per wave)
	const uint32_t my_ids = 8192u + 2u * (uint32_t)WB + (uint32_t)wave * 256u;   // this wave's id words inside a stage
	const uint32_t split_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_split + (uint32_t)g * 2u * 12288u;   // this parity's two split buffers
	// bundle = features + both parities' weights of the batch at `slot` into stage st, then the ids of the batch
	// (slot2, n2) into the wave's id words -- LAST, so that their arrival means the wave's whole bundle arrived.
	struct Bundle { uint32_t slot, id0, slot2, n2, st; };
	// Pieces of a wave: 0 = two feature rows (entries 2 wave, 2 wave + 1: this chunk's 512 B of each; a 64-bit address per
	// lane), then the weight pieces, then (NP - 1) the ids.  Weight pieces, pre-split hand-over: the batch is per group of 8
	// entries [term][256 px'][8 x bf16] = 12 KB, a parity's half of a term = 2 KB = two 1-KB pieces, 24 per batch: pc = 3 wave
	// + (i - 1) -> parity pc / 12, then (group, term, half).  COOP: fp32 rows of 1 KB per entry; the wave fetches exactly what
	// it will split -- the 128 B [32 cg .. 32 cg + 31] px' of its parity for entries 8 (i - 1) + (lane >> 3), 16 B per lane.
	// Weight and id pieces are "uniform base + 32-bit lane offset": dma_src returns the base, dma_voff the lane's offset; the
	// SGPR-base form of the instruction moves half the address bytes.  (m0 = the LDS destination; the compiler sets m0 itself
	// in front of its own LDS-DMA builtin, it keeps nothing in it.)
	const uint32_t lane16 = COOP ? (uint32_t)(lane >> 3) * 1024u + (uint32_t)(lane & 7) * 16u : (uint32_t)lane * 16u;
	auto dma_src = [&](auto I, const Bundle& bd) __attribute__((always_inline)) -> const char* {
		constexpr int i = decltype(I)::value;
		if constexpr (i == 0) {
			const float* row = bd.id0 == SGS_BG_ID ? bg : features + (size_t)bd.id0 * C;
			return (const char*)(row + cbase + l31 * 4);
		} else if constexpr (i == NP - 1) {
			return (const char*)(act_id + bd.slot2);
		} else if constexpr (COOP) {
			return wgt + (size_t)(bd.slot + 8u * (uint32_t)(i - 1)) * 1024 + (size_t)g * 512 + (size_t)cg * 128;
		} else {
			const int pc = 3 * wave + (i - 1);
			const int par = pc / 12, p = pc - 12 * par;
			return wgt + (size_t)((bd.slot >> 3) + (uint32_t)(p / 6)) * 12288 + (size_t)((p % 6) / 2) * 4096 +
			       (size_t)par * 2048 + (size_t)(p % 2) * 1024;
		}
	};
	auto dma_go = [&](auto I, const char* src, uint32_t st, uint32_t voff) __attribute__((always_inline)) {
		constexpr int i = decltype(I)::value;
		if constexpr (i == 0)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
							 (__attribute__((address_space(3))) void*)(size_t)(st + (uint32_t)(2 * wave) * 512u), 16, 0, 0);
		else {
			const uint32_t ldst = i == NP - 1 ? st + my_ids
						  : (COOP ? st + 8192u + (uint32_t)g * 8192u + (uint32_t)cg * 2048u + (uint32_t)(i - 1) * 1024u
							  : st + 8192u + (uint32_t)(3 * wave + (i - 1)) * 1024u);
			if constexpr (i == NP - 1)
				asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(ldst), "v"(voff), "s"(src) : "memory", "m0");
			else
				asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(ldst), "v"(voff), "s"(src) : "memory", "m0");
		}
	};
	auto id_voff = [&](const Bundle& bd) __attribute__((always_inline)) -> uint32_t {
		return ((uint32_t)(lane & 15) < bd.n2 ? (uint32_t)(lane & 15) : bd.n2 - 1u) * 4u;
	};
	auto issue_all = [&](const Bundle& bd) __attribute__((always_inline)) {
		dma_go(std::integral_constant<int, 0>{}, dma_src(std::integral_constant<int, 0>{}, bd), bd.st, 0u);
		dma_go(std::integral_constant<int, 1>{}, dma_src(std::integral_constant<int, 1>{}, bd), bd.st, lane16);
		dma_go(std::integral_constant<int, 2>{}, dma_src(std::integral_constant<int, 2>{}, bd), bd.st, lane16);
		if constexpr (NP == 5) dma_go(std::integral_constant<int, 3>{}, dma_src(std::integral_constant<int, 3>{}, bd), bd.st, lane16);
		dma_go(std::integral_constant<int, NP - 1>{}, dma_src(std::integral_constant<int, NP - 1>{}, bd), bd.st, id_voff(bd));
	};

	// the eight accumulator blocks a[0:127] (see acc_zero)
	acc_zero<0>(); acc_zero<1>(); acc_zero<2>(); acc_zero<3>(); acc_zero<4>(); acc_zero<5>(); acc_zero<6>(); acc_zero<7>();

	uint32_t wbase = 0;
#pragma unroll
	for (int k = 0; k < S3_NST; k++)   // every stage's id words start as "not landed"
		asm volatile("ds_write_b32 %0, %1" : : "v"(ring + (uint32_t)k * S3_STAGE + my_ids + (uint32_t)lane * 4u), "v"(S2_SENT) : "memory");
	asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
#pragma unroll
	for (int k = 0; k < S3_LA; k++) {   // prologue bundles 0 .. LA-1 (their feature ids by ordinary loads)
		const uint2 ev = s_bt[k], e2v = s_bt[k + S3_LA];
		// (table words are workgroup-uniform: say so, the weight / id pieces want their bases in SGPRs)
		const uint32_t ex = (uint32_t)__builtin_amdgcn_readfirstlane((int)ev.x), ey = (uint32_t)__builtin_amdgcn_readfirstlane((int)ev.y);
		const uint32_t e2x = (uint32_t)__builtin_amdgcn_readfirstlane((int)e2v.x), e2y = (uint32_t)__builtin_amdgcn_readfirstlane((int)e2v.y);
		const uint32_t n = ey & 255u;
		const uint32_t id0 = act_id[ex + (sub < n ? sub : n - 1u)];
		issue_all(Bundle{ex, id0, e2x, e2y & 255u, ring + (uint32_t)k * S3_STAGE});
	}
	uint32_t st0 = ring, stI = ring + S3_LA * S3_STAGE;   // stages of batch j and of bundle j + LA (= the stage of batch j - 1)
	uint32_t j = 0;
	uint32_t late = 0;   // (trace) polls of this wave that found its pieces still in flight

	// (development, DBG & 4) phase clocks: shader cycles this wave spent, summed over its steps, in
	//   0 table words | 1 DMA issue | 2 deferred stores | 3 operand wait + feature split | 4 arrival check | 5 barrier 1 |
	//   6 the 48 MFMAs (issue) | 7 arrival check | 8 barrier 2 | 9 between steps (pair stores, tile bookkeeping)
	// written behind the workgroup trace (12 words per wave).  s_memtime is an SMEM access: reading it waits for lgkmcnt(0),
	// i.e. also for LDS reads in flight -- the stamps behind the operand reads see them land (a small distortion).
	constexpr bool PH = (DBG & 4) != 0;
	uint32_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint32_t ph_prev = PH ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
#define S3_STAMP(k_)                                                                                 \
	do {                                                                                             \
		if constexpr (PH) {                                                                          \
			const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();                            \
			ph[k_] += now_ - ph_prev;                                                                \
			ph_prev = now_;                                                                          \
		}                                                                                            \
	} while (0)

	// "my pieces of the bundle in stage st have landed" (the wave's id words no longer hold the sentinel); returns the id of
	// the feature row this lane fetches for the bundle LA ahead of it, and re-arms the words for the stage's next bundle
	// The check is split so that its LDS round trip can ride under other work: poll_issue sends the two reads, poll_finish
	// (after an s_waitcnt lgkmcnt(0) that the caller needs anyway, or its own) looks at them.
	auto poll_issue = [&](uint32_t st, uint32_t& wv, uint32_t& id0) __attribute__((always_inline)) {
		const uint32_t pa = st + my_ids + (uint32_t)lane * 4u, ia = st + my_ids + sub * 4u;
		asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(wv), "=&v"(id0) : "v"(pa), "v"(ia) : "memory");
	};
	auto poll_finish = [&](uint32_t st, uint32_t wv, uint32_t id0) __attribute__((always_inline)) -> uint32_t {
		const uint32_t pa = st + my_ids + (uint32_t)lane * 4u, ia = st + my_ids + sub * 4u;
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wv), "+v"(id0) : : "memory");
		if (__builtin_amdgcn_ballot_w64(wv == S2_SENT) != 0ull) {   // not landed yet: poll (no vmcnt: stores may be outstanding in any number)
			if (trace) late++;
			int spins = 0;
			do {
				if (++spins > (1 << 22)) __builtin_trap();   // (a lost bundle must not hang the device)
				__builtin_amdgcn_s_sleep(1);
				asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(wv), "=&v"(id0) : "v"(pa), "v"(ia) : "memory");
			} while (__builtin_amdgcn_ballot_w64(wv == S2_SENT) != 0ull);
		}
		asm volatile("ds_write_b32 %0, %1" : : "v"(pa), "v"(S2_SENT) : "memory");
		return id0;
	};
	auto poll = [&](uint32_t st) __attribute__((always_inline)) -> uint32_t {
		uint32_t wv, id0;
		poll_issue(st, wv, id0);
		return poll_finish(st, wv, id0);
	};
	const uint32_t flag_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)s_flag;
	// (FREE) wait until the counter at LDS address a has reached `target` (wave-uniform; bounded: a protocol error must not hang the device)
	auto flag_wait = [&](uint32_t a, uint32_t target) __attribute__((always_inline)) {
		uint32_t v;
		int spins = 0;
		for (;;) {
			asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
			if ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
			if (++spins > (1 << 22)) __builtin_trap();
			__builtin_amdgcn_s_sleep(1);
		}
	};
	// the same in two parts: the read goes out early (flag_peek), the check (flag_check) finds it landed -- only a counter that is
	// still short costs a poll loop
	auto flag_peek = [&](uint32_t a, uint32_t& v) __attribute__((always_inline)) {
		asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory");
	};
	auto flag_check = [&](uint32_t a, uint32_t v, uint32_t target) __attribute__((always_inline)) {
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
		if ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) < target) flag_wait(a, target);
	};
	// (FREE) this wave bumps the counter: the LDS unit executes a wave's instructions in order, so everything the wave read
	// from / saw in LDS before is behind it
	auto flag_add = [&](uint32_t a) __attribute__((always_inline)) {
		if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(a), "v"(1u) : "memory");
	};
	uint32_t nid = poll(ring);   // bundle 0: the ids of batch LA
	if constexpr (FREE) flag_add(flag_a);   // (this wave's pieces of bundle 0 are in)
	uint32_t nid1 = 0u;          // (COOP) the ids one batch further on: the arrival check runs a step earlier there
	if constexpr (COOP) {
		nid1 = poll(ring + S3_STAGE);   // bundle 1
		if (!(DBG & 2)) s2_split_coop(ring + (uint32_t)g * 8192u, split_a, cg, half, l31);   // batch 0's weights -> split buffer 0
	}
	if constexpr (!FREE) {
		__builtin_amdgcn_s_barrier();   // every wave's pieces of bundle 0 have landed (COOP: and batch 0's split terms are complete)
		if (g) {   // the second half runs one barrier behind the first
			__builtin_amdgcn_s_setprio(1);
			__builtin_amdgcn_s_barrier();
		}
	} else if (g) {
		__builtin_amdgcn_s_setprio(1);
	}

	// store addressing of the pair whose blocks 2, 3 are still to be written (deferred), and how far that is
	const float* const ubase = out + (size_t)c0 * HW;
	const uint64_t plane = (uint64_t)HW * 4u;
	uint32_t d_o0 = 0u;    // byte offset of [4 half][y0 + 8][xp] (block 2's first row)
	int dprog = 4;         // chunks of 16 stores issued (4 = nothing pending)
	uint32_t pb_o0 = 0u;   // (STP) byte offset of [4 half][y0 + 4][xp]: pixel block 1 of the finished pair, still to be written
	bool pend_b = false;   // (STP) ... and whether it is (uniform)

	Bundle nb;   // the bundle this step issues (j + LA)
	// the step's table words: entry word of batch j (returned), slots of the bundle to issue
	auto step_head = [&]() __attribute__((always_inline)) -> uint32_t {
		if (j + 2 * S3_LA >= wbase + JMAX) {   // (uniform, long segments only) slide the table window: both halves pass the
			// two barriers one phase apart; the first half refills while the second is between its phases
			__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
			__builtin_amdgcn_s_barrier();
			wbase = j;
			fill_table(wbase);
			asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
			__builtin_amdgcn_s_barrier();
		}
		const uint32_t a = bt_a + (j - wbase) * 8u;
		uint32_t r0, r1;
		uint64_t rd;
		asm volatile(
			"ds_read_b32 %0, %3 offset:4\n\t"
			"ds_read_b32 %1, %3 offset:%4\n\t"
			"ds_read_b64 %2, %3 offset:%5\n\t"
			"s_waitcnt lgkmcnt(0)"
			: "=&v"(r0), "=&v"(r1), "=&v"(rd)
			: "v"(a), "n"(S3_LA * 8), "n"(2 * S3_LA * 8)
			: "memory");
		const uint32_t e0y = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0);
		nb.slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)r1);
		nb.slot2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rd);
		nb.n2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rd >> 32)) & 255u;
		nb.id0 = nid;
		nb.st = stI;   // the stage batch j - 1 was computed from: both halves have its operands in registers
		return e0y;
	};
	auto step_tail = [&]() __attribute__((always_inline)) {
		st0 = st0 + S3_STAGE == ring + S3_NST * S3_STAGE ? ring : st0 + S3_STAGE;
		stI = stI + S3_STAGE == ring + S3_NST * S3_STAGE ? ring : stI + S3_STAGE;
		j++;
	};

	const bool skip_stores = (DBG & 1) != 0;
	// DBG & 8 (experiment): transposed 16-byte stores (s2_store4, SMODE 3) -- a tile pair = 32 store instructions instead of 128.
	// Measured (call G, profiles/r04_sweep_wide_stores.txt): 6 % SLOWER (1.26-1.34 vs 1.18-1.22 ms), and the cycles between
	// steps did not move: what a pair's store burst costs is its BYTES (16 KB per wave through a CU's store path at ~7 B/clk),
	// not its instruction count or the wave's 6-bit counter of outstanding operations.  The 4-byte stores stay.
	constexpr int SMODE = (DBG & 8) ? 3 : 0;
	const uint32_t wdelta = SMODE == 3 ? (uint32_t)(lane & 3) : 0u;   // (SMODE 3: the lane's transpose masks)
#undef S2_PAIR_OFF
#define S2_PAIR_OFF(y_, txl_)                                                                        \
	(SMODE == 3 ? (((uint32_t)(4 * half + (lane & 3)) * (uint32_t)HW + (uint32_t)((y_) * PW + (txl_) * SGS_TILE + 4 * (l31 >> 2))) * 4u) \
		    : (((uint32_t)(4 * half) * (uint32_t)HW + (uint32_t)((y_) * PW + (txl_) * SGS_TILE + l31)) * 4u))
#define S3_RDB(dst_, pb_)                                                                            \
	asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6" \
		     : "=&v"(dst_[0]), "=&v"(dst_[1]), "=&v"(dst_[2]) : "v"(wa_), "n"((pb_) * 512), "n"(2048 + (pb_) * 512), "n"(4096 + (pb_) * 512) : "memory")
// second pixel-block pair of the batch: the same (A term, B term, k half) order as mfma_dense, as MFMA pairs with this wave's
// five DMA pieces of bundle j + LA between them.  Measured (tools/sweep_phases.py, profiles/r04_sweep_phases.txt): issued
// in the PREP phase right behind the operand ds_reads a piece cost ~275 cycles of issue (1 370 per step, the longest item
// of the phase); between MFMA pairs, with no LDS read in flight, its slot hides under the 64 cycles of matrix work.
#define S3_HALF2(bx_, by_, c_)                                                                       \
	do {                                                                                             \
		constexpr int TA_[6] = {2, 0, 1, 1, 0, 0}, TB_[6] = {0, 2, 1, 0, 1, 0};   /* smallest terms first */ \
		mfma_pair<bx_, by_>(A_.t[TA_[c_]][0], u32x2{x2_[TB_[c_]].x, x2_[TB_[c_]].y}, u32x2{y2_[TB_[c_]].x, y2_[TB_[c_]].y}); \
		if constexpr ((c_) == 0) dma_go(std::integral_constant<int, 0>{}, da0_, dst_, 0u);           \
		if constexpr ((c_) == 1) dma_go(std::integral_constant<int, 1>{}, da1_, dst_, lane16);       \
		if constexpr ((c_) == 2) dma_go(std::integral_constant<int, 2>{}, da2_, dst_, lane16);       \
		if constexpr ((c_) == 3 && NP == 5) dma_go(std::integral_constant<int, 3>{}, da3_, dst_, lane16); \
		if constexpr ((c_) == NP - 1) dma_go(std::integral_constant<int, NP - 1>{}, da4_, dst_, io4_); \
		mfma_pair<bx_, by_>(A_.t[TA_[c_]][1], u32x2{x2_[TB_[c_]].z, x2_[TB_[c_]].w}, u32x2{y2_[TB_[c_]].z, y2_[TB_[c_]].w}); \
	} while (0)
#define S3_HALF2W(bx_, by_, c_)                                                                      \
	do {                                                                                             \
		constexpr int TA_[6] = {2, 0, 1, 1, 0, 0}, TB_[6] = {0, 2, 1, 0, 1, 0};                      \
		if constexpr (MM == 1) mfma_pair_wide<bx_, by_>(aw_[TA_[c_]], x2_[TB_[c_]], y2_[TB_[c_]]);   \
		else mfma_pair_wide_sp<(MM > 1 ? MM - 1 : 1), bx_, by_>(aw_[TA_[c_]], x2_[TB_[c_]], y2_[TB_[c_]]); \
		if constexpr ((c_) == 0) dma_go(std::integral_constant<int, 0>{}, da0_, dst_, 0u);           \
		if constexpr ((c_) == 1) dma_go(std::integral_constant<int, 1>{}, da1_, dst_, lane16);       \
		if constexpr ((c_) == 2) dma_go(std::integral_constant<int, 2>{}, da2_, dst_, lane16);       \
		if constexpr ((c_) == 3 && NP == 5) dma_go(std::integral_constant<int, 3>{}, da3_, dst_, lane16); \
		if constexpr ((c_) == NP - 1) dma_go(std::integral_constant<int, NP - 1>{}, da4_, dst_, io4_); \
	} while (0)
// One step = batch j into accumulator blocks b0_..b3_.  PREP: table words; operand reads of batch j go out first (stage j
// landed before the barrier this phase began with); the DMA pieces of bundle j + LA and the deferred stores (DEF_) are
// issued while they land; the feature split; [second half: arrival check of bundle j + 1]; barrier; MFMA: 48 products;
// [first half: arrival check of bundle j + 1]; barrier.
#define S3_STEP(b0_, b1_, b2_, b3_, DEF_, MID_)                                                      \
	do {                                                                                             \
		S3_STAMP(9);                                                                                 \
		uint32_t fa_v_ = 0u, fc_v_ = 0u;                                                             \
		if constexpr (FREE) flag_peek(flag_a + (j & 3u) * 4u, fa_v_);   /* (rides with the table words) */ \
		e_ = step_head();                                                                            \
		S3_STAMP(0);                                                                                 \
		/* the source addresses of this wave's five DMA pieces: computed HERE (64-bit VALU chains), issued between the MFMA  \
		   pairs, where only the m0 write and the load itself remain */                                \
		const char* da0_ = dma_src(std::integral_constant<int, 0>{}, nb);   /* per lane: two feature rows */ \
		const char* da1_ = dma_src(std::integral_constant<int, 1>{}, nb);   /* wave-uniform bases (SGPRs): */ \
		const char* da2_ = dma_src(std::integral_constant<int, 2>{}, nb);   /* the lane offset is lane16 */ \
		const char* da3_ = dma_src(std::integral_constant<int, (NP == 5 ? 3 : 2)>{}, nb);             \
		const char* da4_ = (const char*)(act_id + nb.slot2);                                          \
		uint32_t io4_ = id_voff(nb);                                                                 \
		const uint32_t dst_ = nb.st;                                                                 \
		asm volatile("" : "+v"(da0_), "+s"(da1_), "+s"(da2_), "+s"(da3_), "+s"(da4_), "+v"(io4_));    \
		float f_[8];                                                                                 \
		u32x4 x_[3], y_[3], x2_[3], y2_[3];                                                          \
		Op3 A_;                                                                                      \
		const uint32_t st1_ = st0 + S3_STAGE == ring + S3_NST * S3_STAGE ? ring : st0 + S3_STAGE;     \
		const uint32_t st2_ = st1_ + S3_STAGE == ring + S3_NST * S3_STAGE ? ring : st1_ + S3_STAGE;   \
		const uint32_t stn_ = COOP ? st2_ : st1_;   /* the stage whose arrival this step checks (COOP: one step earlier) */ \
		uint32_t pw_ = 0u, pid_ = 0u;                                                                \
		const uint32_t fa_ = st0 + (uint32_t)((8 * half) * 128 + cg * 32 + l31) * 4u;                \
		const uint32_t wa_ = (COOP ? split_a + (j & 1u) * 12288u : st0 + 8192u + (uint32_t)g * 12288u) + (uint32_t)half * 6144u + (uint32_t)l31 * 16u; \
		if (DBG & 2) { if constexpr (FREE) flag_wait(flag_a + 16u + ((j + 3u) & 3u) * 4u, 8u * ((j + 3u) >> 2)); issue_all(nb); } \
		S3_STAMP(1);                                                                                 \
		DEF_;   /* (before the operand reads: the transposes need registers the operands would occupy) */ \
		S3_STAMP(2);                                                                                 \
		if constexpr (FREE) flag_check(flag_a + (j & 3u) * 4u, fa_v_, 8u * ((j >> 2) + 1u));   /* every wave's pieces of bundle j are in */ \
		if constexpr (COOP && !(DBG & 2))   /* batch j + 1's weights (this wave's own pieces, checked a step ago) -> the other split buffer */ \
			s2_split_coop(st1_ + (uint32_t)g * 8192u, split_a + ((j + 1u) & 1u) * 12288u, cg, half, l31); \
		if (!(DBG & 2)) {                                                                            \
			S2_READ8(f_, fa_);                                                                       \
			S3_RDB(x_, 0);                                                                           \
			S3_RDB(y_, (STP ? 2 : 1));   /* (STP: the dense statement takes pixel blocks 0 and 2) */  \
		}                                                                                            \
		if (!(DBG & 2)) {                                                                            \
			asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f_[0]), "+v"(f_[1]), "+v"(f_[2]), "+v"(f_[3]), "+v"(f_[4]), "+v"(f_[5]), "+v"(f_[6]), "+v"(f_[7]) : : "memory"); \
			__builtin_amdgcn_sched_barrier(0);                                                       \
			split8(f_, A_);                                                                          \
			S3_RDB(x2_, (STP ? 1 : 2));                                                              \
			S3_RDB(y2_, 3);                                                                          \
			if (g && !FREE) poll_issue(stn_, pw_, pid_);   /* (second half) the arrival check's reads ride along */ \
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x_[0]), "+v"(x_[1]), "+v"(x_[2]), "+v"(y_[0]), "+v"(y_[1]), "+v"(y_[2]), \
				     "+v"(x2_[0]), "+v"(x2_[1]), "+v"(x2_[2]), "+v"(y2_[0]), "+v"(y2_[1]), "+v"(y2_[2]) : : "memory"); \
			__builtin_amdgcn_sched_barrier(0);                                                       \
		}                                                                                            \
		if constexpr (FREE) flag_add(flag_a + 16u + (j & 3u) * 4u);   /* this wave has read batch j out of its stage */ \
		S3_STAMP(3);                                                                                 \
		if (g && !FREE) {                                                                            \
			if (DBG & 2) poll_issue(stn_, pw_, pid_);                                                \
			const uint32_t got_ = poll_finish(stn_, pw_, pid_);                                      \
			if constexpr (COOP) { nid = nid1; nid1 = got_; } else nid = got_;                         \
		}                                                                                            \
		S3_STAMP(4);                                                                                 \
		if constexpr (!FREE) __builtin_amdgcn_s_barrier();                                           \
		S3_STAMP(5);                                                                                 \
		if (!(DBG & 2)) {                                                                            \
			if constexpr (MM == 0) {                                                                 \
				if constexpr (FREE) flag_peek(flag_a + 16u + ((j + 3u) & 3u) * 4u, fc_v_);            \
				mfma_dense<b0_, b1_>(A_, x_, y_);                                                    \
				/* (FREE) the stage bundle j + LA goes into held batch j - 1: every wave must have read it */ \
				if constexpr (FREE) flag_check(flag_a + 16u + ((j + 3u) & 3u) * 4u, fc_v_, 8u * ((j + 3u) >> 2)); \
				MID_;                                                                                \
				S3_HALF2(b2_, b3_, 0); S3_HALF2(b2_, b3_, 1); S3_HALF2(b2_, b3_, 2);                 \
				S3_HALF2(b2_, b3_, 3); S3_HALF2(b2_, b3_, 4);                                        \
				if (!g || FREE) poll_issue(stn_, pw_, pid_);   /* (first half) under the last four MFMAs */ \
				S3_HALF2(b2_, b3_, 5);                                                               \
			} else {   /* (make X16=1) the double-rate forms */                                     \
				u32x4 aw_[3];                                                                        \
				for (int t_ = 0; t_ < 3; t_++) aw_[t_] = u32x4{A_.t[t_][0].x, A_.t[t_][0].y, A_.t[t_][1].x, A_.t[t_][1].y}; \
				if constexpr (FREE) flag_peek(flag_a + 16u + ((j + 3u) & 3u) * 4u, fc_v_);            \
				if constexpr (MM == 1) mfma_dense_wide<b0_, b1_>(aw_, x_, y_);                        \
				else mfma_dense_wide_sp<MM - 1, b0_, b1_>(aw_, x_, y_);                               \
				if constexpr (FREE) flag_check(flag_a + 16u + ((j + 3u) & 3u) * 4u, fc_v_, 8u * ((j + 3u) >> 2)); \
				MID_;                                                                                \
				S3_HALF2W(b2_, b3_, 0); S3_HALF2W(b2_, b3_, 1); S3_HALF2W(b2_, b3_, 2);              \
				S3_HALF2W(b2_, b3_, 3); S3_HALF2W(b2_, b3_, 4);                                      \
				if (!g || FREE) poll_issue(stn_, pw_, pid_);                                         \
				S3_HALF2W(b2_, b3_, 5);                                                              \
			}                                                                                        \
		} else {                                                                                     \
			MID_;                                                                                    \
			if (!g || FREE) poll_issue(stn_, pw_, pid_);                                             \
		}                                                                                            \
		S3_STAMP(6);                                                                                 \
		if (!g || FREE) {                                                                            \
			const uint32_t got_ = poll_finish(stn_, pw_, pid_);                                      \
			if constexpr (COOP) { nid = nid1; nid1 = got_; } else nid = got_;                         \
			if constexpr (FREE) flag_add(flag_a + ((j + 1u) & 3u) * 4u);   /* this wave's pieces of bundle j + 1 are in */ \
		}                                                                                            \
		S3_STAMP(7);                                                                                 \
		if constexpr (!FREE) __builtin_amdgcn_s_barrier();                                           \
		S3_STAMP(8);                                                                                 \
		step_tail();                                                                                 \
	} while (0)

// (STP) pixel block 1 of the pair that finished under mapping MP_: its 32 stores, then the two blocks are the new tile's (pixel blocks 1 and 3's)
#define S3_PART_B(MP_)                                                                               \
	do {                                                                                             \
		if (pend_b) {                                                                                \
			s2_store_rows<LB(MP_, 1), RB(MP_, 1), 0, 16, SMODE>(ubase, pb_o0, pb_o0 + (uint32_t)(2 * PW) * 4u, plane, wdelta); \
			acc_zero<LB(MP_, 1)>(); acc_zero<RB(MP_, 1)>();                                          \
			asm volatile("s_nop 3" : : : "memory");   /* (the products that follow accumulate into these blocks) */ \
			pend_b = false;                                                                          \
		}                                                                                            \
	} while (0)
#define S3_LEFT_TILE(M_, tx_)                                                                        \
	do {                                                                                             \
		uint32_t e_;                                                                                 \
		do {                                                                                         \
			if constexpr (STP == 0) S3_STEP(LB(M_, 0), LB(M_, 1), LB(M_, 2), LB(M_, 3), S2_DEFERRED_CHUNK(1 - (M_)), (void)0); \
			else S3_STEP(LB(M_, 0), LB(M_, 2), LB(M_, 1), LB(M_, 3), S2_DEFERRED_CHUNK(1 - (M_)), S3_PART_B(1 - (M_))); \
		} while ((e_ >> 16) == 0u);                                                                  \
		if constexpr (STP != 0) S3_PART_B(1 - (M_));   /* (cannot be pending: every tile has a step; kept for the proof) */ \
		while (dprog < 4) S2_DEFERRED_CHUNK(1 - (M_));                                               \
		tx_ = tx0 + (int)((e_ >> 8) & 255u);                                                         \
	} while (0)
#define S3_RIGHT_TILE(M_, tx_)                                                                       \
	do {                                                                                             \
		uint32_t e_;                                                                                 \
		do {                                                                                         \
			if constexpr (STP == 0) S3_STEP(RB(M_, 0), RB(M_, 1), RB(M_, 2), RB(M_, 3), (void)0, (void)0); \
			else S3_STEP(RB(M_, 0), RB(M_, 2), RB(M_, 1), RB(M_, 3), (void)0, (void)0);               \
		} while ((e_ >> 16) == 0u);                                                                  \
		tx_ = tx0 + (int)((e_ >> 8) & 255u);                                                         \
	} while (0)
// (STP) a pair finished under mapping M_: pixel block 0 of both tiles now, pixel block 1 pending (S3_PART_B), blocks 2, 3 deferred as ever;
// at the image's edge everything at once, as S2_PAIR_DONE does
#define S3_PAIR_DONE(M_, tx_)                                                                        \
	do {                                                                                             \
		const bool inside_ = ((tx_) + 1) * SGS_TILE <= W && y0 + 14 < H;   /* (uniform) */           \
		if (STP == 0 || skip_stores || !inside_) {                                                   \
			S2_PAIR_DONE(M_, tx_);                                                                   \
		} else {                                                                                     \
			const uint32_t o0_ = S2_PAIR_OFF(y0, (tx_) - 1);                                          \
			const uint32_t o1_ = o0_ + (uint32_t)(2 * PW) * 4u;                                      \
			asm volatile("s_nop 15" : : : "memory");                                                 \
			s2_store_rows<LB(M_, 0), RB(M_, 0), 0, 16, SMODE>(ubase, o0_, o1_, plane, wdelta);        \
			acc_zero<LB(M_, 0)>(); acc_zero<RB(M_, 0)>();                                            \
			pb_o0 = o0_ + (uint32_t)(4 * PW) * 4u;                                                   \
			pend_b = true;                                                                           \
			d_o0 = o0_ + (uint32_t)(8 * PW) * 4u;                                                    \
			dprog = 0;                                                                               \
		}                                                                                            \
	} while (0)

	constexpr bool NORM = false;
	const int y0 = ty_out * SGS_TILE + g;   // first image row of this parity in the tile row (inside its band, SGS_OPT_OUT_BANDS)
	const int hi = (l31 >> 4) & 1;

	// ---- the sweep.  Even rows: even tiles are left halves; odd rows of a staggered pitch: odd tiles.  (The two halves of
	// the workgroup may therefore be in different branches of this code at the same step; every step has two barriers.)
	int tx = tx0;
	if (J > 0 && ((tx0 + g * stagger) & 1) != 0) {   // the segment starts with a right half whose partner belongs to the previous segment
		S3_RIGHT_TILE(0, tx);
		S2_STORE_SINGLE(RB(0, 0), RB(0, 1), RB(0, 2), RB(0, 3), tx);
	}
	while (j < J) {
		S3_LEFT_TILE(0, tx);
		if (j >= J) { S2_STORE_SINGLE(LB(0, 0), LB(0, 1), LB(0, 2), LB(0, 3), tx); break; }
		S3_RIGHT_TILE(0, tx);
		S3_PAIR_DONE(0, tx);
		if (j >= J) { S3_PART_B(0); while (dprog < 4) S2_DEFERRED_CHUNK(0); break; }
		S3_LEFT_TILE(1, tx);
		if (j >= J) { S2_STORE_SINGLE(LB(1, 0), LB(1, 1), LB(1, 2), LB(1, 3), tx); break; }
		S3_RIGHT_TILE(1, tx);
		S3_PAIR_DONE(1, tx);
		if (j >= J) { S3_PART_B(1); while (dprog < 4) S2_DEFERRED_CHUNK(1); break; }
	}
	if (!g && !FREE) __builtin_amdgcn_s_barrier();   // (the first half's counterpart of the second half's extra barrier)
	__builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // drain the dummy tail bundles before LDS is released
	if (trace && threadIdx.x == 0) {
		trace[4 * (size_t)b] = t_begin;
		trace[4 * (size_t)b + 1] = wall_clock64();
		trace[4 * (size_t)b + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
					   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
		trace[4 * (size_t)b + 3] = (unsigned long long)J | ((unsigned long long)nt << 32) | ((unsigned long long)late << 40);
	}
	if constexpr (PH) {
		if (trace && lane == 0) {
			unsigned long long* q = trace + 4 * (4096 + 8192) + 12 * ((size_t)b * 8 + wave);
#pragma unroll
			for (int k = 0; k < 10; k++) q[k] = ph[k];
			q[10] = J;
			q[11] = (unsigned long long)wave;
		}
	}
#undef S2_PAIR_OFF
#undef S3_STAMP
#undef S3_HALF2
#undef S3_HALF2W
#undef S3_STEP
#undef S3_LEFT_TILE
#undef S3_RIGHT_TILE
#undef S3_PART_B
#undef S3_PAIR_DONE
#undef S3_RDB
}

__global__ void norm_plane_background_kernel(float* __restrict__ plane, size_t n, const float* __restrict__ bg, int C)
{
	float ss = 0.f;
	for (int c = 0; c < C; c++) ss = __builtin_fmaf(bg[c], bg[c], ss);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) plane[i] = ss;
}

hipError_t launch_norm_plane_background(hipStream_t st, float* plane, size_t n, const float* bg, int C)
{
	hipLaunchKernelGGL(norm_plane_background_kernel, dim3(1024), dim3(256), 0, st, plane, n, bg, C);
	return hipGetLastError();
}

// ADVICE r5 / DESIGN.md 5.10: a kernel that issues the double-rate MFMA must own its compute unit -- 8 waves x 256 registers = the CU's whole
// register file, more than half of its LDS.  The build checks the code object (csrc/check_code_object.py, run by `make`); this is the same
// question put to the LOADED code object on the device it will run on, once per process: whatever the runtime reports for the kernel that is
// about to be launched.  false -> the caller takes the x8 form (same bits, 10 % slower) and says so once on stderr.
bool x16_kernel_owns_cu(const void* fn, const char* name)
{
	hipFuncAttributes at;
	if (hipFuncGetAttributes(&at, fn) != hipSuccess) {
		(void)hipGetLastError();
		fprintf(stderr, "libsgs_hip: cannot read the attributes of %s: its x16 MFMA form is not used\n", name);
		return false;
	}
	const bool ok = at.numRegs == 256 && at.maxThreadsPerBlock == 512 && at.sharedSizeBytes > 80u * 1024u && at.localSizeBytes == 0;
	if (!ok)
		fprintf(stderr, "libsgs_hip: %s does not own its compute unit (registers %d, threads %d, LDS %zu B, scratch %zu B; needs 256 / 512 / > 80 KB / 0): "
				"its x16 MFMA form is not used (DESIGN.md 5.10)\n", name, at.numRegs, at.maxThreadsPerBlock, (size_t)at.sharedSizeBytes, (size_t)at.localSizeBytes);
	return ok;
}

int sweep3_x16_ownership()   // 1: both x16 ping-pong sweeps own their CU; 0: they do not (the x8 form runs instead)
{
	static const int own = (x16_kernel_owns_cu((const void*)&blend_accum_sweep3_kernel<0, 1, false, true>, "blend_accum_sweep3_kernel<0, 1, false, true>") &&
				x16_kernel_owns_cu((const void*)&blend_accum_sweep3_kernel<0, 1, false, true, 1>, "blend_accum_sweep3_kernel<0, 1, false, true, 1>") &&
				x16_kernel_owns_cu((const void*)&blend_accum_sweep3_kernel<0, 1>, "blend_accum_sweep3_kernel<0, 1>")) ? 1 : 0;
	return own;
}

hipError_t launch_accum_sweep3(hipStream_t st, int dbg, const BlendFwdArgs& a, const uint32_t* table,
			       const uint32_t* nbatches, const uint32_t* act_id, const char* wgt, const uint32_t* counter,
			       int nc, int seg, int nseg, int pxcd, int items, unsigned long long* trace,
			       const uint32_t* order, int dealt, int tune, int form, int stp)
{
	if (tune == 1 && !sweep3_x16_ownership()) {   // (said once on stderr; sgs_x16_cu_ownership() reports it)
		tune = 0;   // the same products on v_mfma_f32_32x32x8_bf16: bit-identical maps
		if (form == 2) form = 0;   // (the product library's x8 sweep is the lock-step form)
		stp = 0;
	}
	if (stp != 0 && !(tune == 1 && form == 2)) return hipErrorInvalidValue;   // (the store placements exist for the default form only)
	const bool coop = form == 1;   // (form: 0 = lock step, 1 = fp32 hand-over, 2 = free-running halves)
#define S3_LAUNCH(D_)                                                                                \
	hipLaunchKernelGGL((blend_accum_sweep3_kernel<D_>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands)
	// bits [19:16] of the variant: 1 = the sweep on the double-rate v_mfma_f32_32x32x16_bf16, dense (round 5: THE DEFAULT -- the workgroup
	// owns its compute unit); (make X16=1) 2 = x16 + s_nop filler, 3 = x16 + VALU filler
#define S3_LAUNCH_MM(M_)                                                                             \
	hipLaunchKernelGGL((blend_accum_sweep3_kernel<0, M_>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands)
	if (tune == 1 && dbg == 0 && form == 0) {
		S3_LAUNCH_MM(1);
		return hipGetLastError();
	}
#ifdef SGS_WITH_X16
	if (tune == 2 || tune == 3) {
		if (tune == 2) S3_LAUNCH_MM(2);
		else S3_LAUNCH_MM(3);
		return hipGetLastError();
	}
#endif
	// the free-running halves (flags instead of barriers, nibble 4) on the x16 MFMA: round 5's default
	if (tune == 1 && form == 2 && dbg == 0) {   // stp: where a finished pair's stores are issued (template argument STP of the kernel)
#define S3_LAUNCH_FS(S_)                                                                             \
	hipLaunchKernelGGL((blend_accum_sweep3_kernel<0, 1, false, true, S_>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands)
		if (stp == 1) S3_LAUNCH_FS(1);
		else if (stp == 0) S3_LAUNCH_FS(0);
		else return hipErrorInvalidValue;
#undef S3_LAUNCH_FS
		return hipGetLastError();
	}
#undef S3_LAUNCH_MM
#ifdef SGS_WITH_EXPERIMENTS   // round 6: round 4's fp32 hand-over (the sweep splits the weights one step ahead, sweep nibble 5) on the x16 MFMA, lock step
	if (tune == 1 && form == 1 && dbg == 0) {
		hipLaunchKernelGGL((blend_accum_sweep3_kernel<0, 1, true>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table,
				   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg,
				   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands);
		return hipGetLastError();
	}
#endif
#ifdef SGS_WITH_EXPERIMENTS   // round 6: the ablations / phase clocks of the DEFAULT form (free-running halves on x16), for profiles/r06_sweep_phases.txt
	if (tune == 1 && form == 2 && (dbg == 1 || dbg == 2 || dbg == 3 || dbg == 4)) {
#define S3_LAUNCH_FX(D_)                                                                             \
	if (stp == 1) hipLaunchKernelGGL((blend_accum_sweep3_kernel<D_, 1, false, true, 1>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands);                                \
	else hipLaunchKernelGGL((blend_accum_sweep3_kernel<D_, 1, false, true>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands)
		if (dbg == 1) S3_LAUNCH_FX(1);
		else if (dbg == 2) S3_LAUNCH_FX(2);
		else if (dbg == 3) S3_LAUNCH_FX(3);
		else S3_LAUNCH_FX(4);
#undef S3_LAUNCH_FX
		return hipGetLastError();
	}
#endif
	if (tune != 0) return hipErrorInvalidValue;
#ifndef SGS_WITH_EXPERIMENTS   // the product library holds ONE ping-pong sweep; the forms below are make EXPERIMENTS=1
	(void)coop;
	if (form != 0 || dbg != 0) return hipErrorInvalidValue;
	S3_LAUNCH(0);
	return hipGetLastError();
#else
	if (coop) {   // sweep nibble 5: fp32 weights handed over, split by the sweep one step ahead
#define S3_LAUNCH_C(D_)                                                                              \
	hipLaunchKernelGGL((blend_accum_sweep3_kernel<D_, 0, true>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands)
		if (dbg == 1) S3_LAUNCH_C(1);
		else if (dbg == 2) S3_LAUNCH_C(2);
		else if (dbg == 4) S3_LAUNCH_C(4);
		else S3_LAUNCH_C(0);
#undef S3_LAUNCH_C
		return hipGetLastError();
	}
	if (form == 2) {   // sweep nibble 4: free-running halves (flags instead of barriers)
#define S3_LAUNCH_F(D_)                                                                              \
	hipLaunchKernelGGL((blend_accum_sweep3_kernel<D_, 0, false, true>), dim3(pxcd * 8), dim3(512), 0, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt, tune, a.bands)
		if (dbg == 1) S3_LAUNCH_F(1);
		else if (dbg == 2) S3_LAUNCH_F(2);
		else if (dbg == 4) S3_LAUNCH_F(4);
		else S3_LAUNCH_F(0);
#undef S3_LAUNCH_F
		return hipGetLastError();
	}
	if (dbg == 1) S3_LAUNCH(1);        // (development ablations) no stores
	else if (dbg == 2) S3_LAUNCH(2);   // no matrix work
	else if (dbg == 3) S3_LAUNCH(3);   // ring only
	else if (dbg == 4) S3_LAUNCH(4);   // phase clocks (tools/sweep_phases.py)
	else if (dbg == 8) S3_LAUNCH(8);   // transposed 16-byte stores (experiment: slower)
	else S3_LAUNCH(0);
	return hipGetLastError();
#endif
#undef S3_LAUNCH
}

hipError_t launch_accum_sweep2(hipStream_t st, int arith, int dbg, const BlendFwdArgs& a, const uint32_t* table,
			       const uint32_t* nbatches, const uint32_t* act_id, const char* wgt, const uint32_t* counter,
			       int nc, int seg, int nseg, int pxcd, int items, unsigned long long* trace,
			       const uint32_t* order, int dealt)
{
	// (development) SGS_DEBUG_SWEEP_DYNLDS=<bytes>: unused dynamic LDS on top of the kernel's own, to pin the sweep at ONE
	// workgroup per CU for occupancy experiments (is a stage of the kernel bound by per-workgroup latency or by the chip?)
	static const int dyn_lds = getenv("SGS_DEBUG_SWEEP_DYNLDS") ? atoi(getenv("SGS_DEBUG_SWEEP_DYNLDS")) : 0;
#define S2_LAUNCH(A_, D_)                                                                            \
	hipLaunchKernelGGL((blend_accum_sweep2_kernel<A_, D_>), dim3(pxcd * 8), dim3(256), dyn_lds, st, a.ranges, table, \
			   nbatches, act_id, wgt, a.features, a.bg, a.out, counter, a.W, a.H, a.C, a.gx, nc, seg, nseg, \
			   pxcd, items, a.pitch, trace, order, dealt)
#ifndef SGS_WITH_EXPERIMENTS   // the product library: the exact fp32 sweep (variant 15) and the norm-plane epilogues of N1; the rest is make EXPERIMENTS=1
	if (arith == S2_EXACT && dbg == 0) S2_LAUNCH(S2_EXACT, 0);
	else if (arith == S2_EXACT && dbg == 32) S2_LAUNCH(S2_EXACT, 32);
	else if (arith == S2_X6P && dbg == 32) S2_LAUNCH(S2_X6P, 32);
	else return hipErrorInvalidValue;
	return hipGetLastError();
#else
	if (arith == S2_EXACT) {
		if (dbg == 1) S2_LAUNCH(S2_EXACT, 1);
		else if (dbg == 32) S2_LAUNCH(S2_EXACT, 32);
		else if (dbg == 2) S2_LAUNCH(S2_EXACT, 2);
		else S2_LAUNCH(S2_EXACT, 0);
	} else if (arith == S2_X6W || arith == S2_X6PW) {
		// The double-rate v_mfma_f32_32x32x16_bf16 builds are experiments (forwards running beside them come out
		// with damaged packed-fp32 results on some boxes; round 4: a library GEMM beside the same victim does not do that, so
		// the trigger is in these kernels).  They are not in the product library: `make X16=1` builds them for the reproducers.
#ifdef SGS_WITH_X16
		if (arith == S2_X6W) S2_LAUNCH(S2_X6W, 0);
		else if (dbg == 1) S2_LAUNCH(S2_X6PW, 1);
		else if (dbg == 2) S2_LAUNCH(S2_X6PW, 2);
		else if (dbg == 4) S2_LAUNCH(S2_X6PW, 4);
		else if (dbg == 8) S2_LAUNCH(S2_X6PW, 8);
		else S2_LAUNCH(S2_X6PW, 0);
#else
		return hipErrorInvalidValue;
#endif
	} else if (arith == S2_X6P) {
		if (dbg == 1) S2_LAUNCH(S2_X6P, 1);
		else if (dbg == 2) S2_LAUNCH(S2_X6P, 2);
		else if (dbg == 3) S2_LAUNCH(S2_X6P, 3);
		else if (dbg == 4) S2_LAUNCH(S2_X6P, 4);
		else if (dbg == 8) S2_LAUNCH(S2_X6P, 8);
		else if (dbg == 6) S2_LAUNCH(S2_X6P, 6);
		else if (dbg == 10) S2_LAUNCH(S2_X6P, 10);
		else if (dbg == 32) S2_LAUNCH(S2_X6P, 32);
		else S2_LAUNCH(S2_X6P, 0);
	} else if (arith == S2_X6C) {
		if (dbg == 1) S2_LAUNCH(S2_X6C, 1);
		else if (dbg == 2) S2_LAUNCH(S2_X6C, 2);
		else if (dbg == 3) S2_LAUNCH(S2_X6C, 3);
		else S2_LAUNCH(S2_X6C, 0);
	} else if (arith == S2_X6S) {
		if (dbg == 1) S2_LAUNCH(S2_X6S, 1);
		else S2_LAUNCH(S2_X6S, 0);
	} else {
		if (dbg == 1) S2_LAUNCH(S2_X6, 1);
		else if (dbg == 2) S2_LAUNCH(S2_X6, 2);
		else if (dbg == 3) S2_LAUNCH(S2_X6, 3);
		else S2_LAUNCH(S2_X6, 0);
	}
	return hipGetLastError();
#endif
#undef S2_LAUNCH
}

} // namespace sgs
