// blend_fused_pc.hip -- the C >= 128 forward blend as ONE kernel, weights computed ONCE per pixel strip and shared
// through LDS by all channel waves of the workgroup (chain redundancy R = 1).
//
// Same function as the reference's renderCUDA (CR/cuda_rasterizer/forward.cu:262-375).  Round 1 split it into a
// weights pre-pass and a streaming accumulate with a 1.04 GB round trip through HBM in between
// (blend_fwd_split.hip); blend_fused.hip showed that a wave which does everything itself (one wave per SIMD, chain
// recomputed per 128-channel chunk) is issue-bound (3.97 ms at cfg3).  Here the roles are split INSIDE a workgroup:
//
//   workgroup = (tile-row segment, ONE strip of 64 pixels, up to 512 channels) = 1 producer wave + C / 64 consumer waves
//   (9 waves at C = 512, three per SIMD at most: 168 registers each).
//
//   producer wave   walks the tiles' depth-sorted lists (lane = entry, 64 at a time, prefetched across tile
//                   boundaries), rejects entries that cannot reach the strip, runs the alpha / transmittance chain
//                   with lane = pixel in the contract's arithmetic (bit-identical weights, n_contrib, final_T) and
//                   publishes batches of 16 active entries: fp32 weights [16][64], ids, a meta word.  One batch per
//                   step.  It also writes final_T / n_contrib.
//   consumer waves  64 channels x 64 pixels each (four 32 x 32 MFMA blocks): fetch their 256-B slice of the batch's
//                   16 feature rows by LDS-DMA into a private 3-stage ring (issued two steps before use), multiply
//                   (split-bf16 x3 products or exact fp32 MFMA), keep a finished left tile in a second accumulator
//                   set and store complete 128-B lines after the right tile (v_permlane16_swap pairing, buffer
//                   stores whose out-of-range offsets are the edge guard).
//   one s_barrier per step hands batch k from the producer to the consumers (consumed at step k + 3).
//
// The chain costs ~30 VALU per (entry, pixel) ONCE per strip (not once per channel chunk), on a wave of its own that
// co-issues with the consumers' MFMAs; nothing but the features (once per strip from L2) and the output crosses HBM.
//
// vmcnt discipline (consumers): LDS-DMA and stores are inline asm; vmcnt orders loads among loads and stores among
// stores only, so a counted wait is safe up to the number of YOUNGER LOADS.  The wait before a batch is
// vmcnt(4 x younger bundles); a tile's stores are issued after the next bundle has been waited for, and the step that
// consumes it skips its wait, so the stores have a step to drain before anything waits behind them.
#include "blend_fused_common.h"

namespace sgs {

using namespace fused;

namespace {

constexpr int PB = 16;                    // list entries per batch
constexpr int P_NST = 3;                  // feature ring stages per consumer wave
constexpr int P_LEAD = 3;                 // a batch produced in step k is consumed in step k + 3
constexpr int P_NW = 4;                   // weight / id / meta slots in flight
constexpr int P_WROWS = PB + 1;           // + one dummy row for entries no pixel takes
constexpr int P_WBUF = P_WROWS * 64 * 4;  // fp32 weights [17][64]
constexpr int P_QCAP = 80;                // candidate queue (ring)
constexpr int P_CSTAGE = PB * 64 * 4;     // 16 entries x 64 channels fp32 (one consumer, one stage)
constexpr int P_NDMA = 4;                 // LDS-DMA instructions per consumer per batch (4 entries each)
constexpr int P_MAXCONS = 8;
constexpr int P_OFF_WBUF = 0;
constexpr int P_OFF_IDS = P_OFF_WBUF + P_NW * P_WBUF;
constexpr int P_OFF_META = P_OFF_IDS + P_NW * P_WROWS * 4;      // P_NW meta words + the batch total
constexpr int P_OFF_QREC = (P_OFF_META + (P_NW + 1) * 4 + 127) & ~127;
constexpr int P_OFF_QIDX = P_OFF_QREC + P_QCAP * 32;
constexpr int P_OFF_RING = (P_OFF_QIDX + P_QCAP * 4 + 1023) & ~1023;
constexpr uint32_t M_LAST = 1u << 30;     // meta: the batch completes its tile (low 16 bits: tile column)

__host__ __device__ constexpr int pc_lds_bytes(int ncons) { return P_OFF_RING + ncons * P_NST * P_CSTAGE; }
static_assert(pc_lds_bytes(P_MAXCONS) <= 160 * 1024, "LDS budget");

typedef f32x16 Acc2[2][2];   // [channel block of 32][pixel block of 32]

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void acc2_zero(Acc2& S)
{
	const s16x4 z = {0, 0, 0, 0};
	const f32x16 c0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int cb = 0; cb < 2; cb++)
#pragma unroll
		for (int pb = 0; pb < 2; pb++) S[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(z, z, c0, 0, 0, 0);
}

// A completed pair (see blend_fused.hip): L = left tile's accumulators, R = right tile's; offs[pb][row] are the lanes'
// byte offsets inside the wave's 64 channel planes, or F_OOB for lanes / rows that must not be written.
__device__ __forceinline__ void store_pair2(const Acc2& L, const Acc2& R, v4i rsrc, const uint32_t offs[2][2],
					    uint32_t plane_bytes)
{
	const uint32_t plane5 = 5u * plane_bytes;
#pragma unroll
	for (int pb = 0; pb < 2; pb++) {
		uint32_t so = 0;
		asm volatile("s_mov_b32 %0, 0" : "=s"(so));
#pragma unroll
		for (int cb = 0; cb < 2; cb++)
#pragma unroll
			for (int r = 0; r < 16; r++) {
				// (plain reads: an "a" constraint would make the compiler split the 168 registers 84 / 84 between
				// the VGPR and accumulator files and spill the 128 accumulators)
				const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(L[cb][pb][r]),
										 __float_as_uint(R[cb][pb][r]), false, false);
				asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt"
					     :
					     : "v"(sw[0]), "v"(offs[pb][0]), "s"(rsrc), "s"(so)
					     : "memory");
				asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt"
					     :
					     : "v"(sw[1]), "v"(offs[pb][1]), "s"(rsrc), "s"(so)
					     : "memory");
				walk_plane(so, r, plane_bytes, plane5);
			}
	}
}

} // namespace

template <bool EXACT>
__global__ __launch_bounds__(64 * (P_MAXCONS + 1), 3) void blend_fused_pc_kernel(const FusedArgs a, const int ncons,
										     const int chunk_c, const int dbg)
{
	extern __shared__ __attribute__((aligned(1024))) char p_smem[];
	const int b = blockIdx.x;
	const int v = (b & 7) * a.per_xcd + (b >> 3);
	if (v >= a.total_items) return;
	const int chunk = v % a.nchunks;
	int rest = v / a.nchunks;
	const int strip = rest & 3;
	rest >>= 2;
	const int sg = rest % a.nseg, ty = rest / a.nseg;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool is_prod = wave == ncons;
	const int g = strip & 1, hh = strip >> 1;
	const int half = lane >> 5, l31 = lane & 31;
	const int W = a.W, H = a.H, C = a.C, gx = a.gx;
	const size_t HW = (size_t)H * W;
	const int cbase = chunk * chunk_c;
	const int stagger = (W & 31) == 16 ? 1 : 0;
	const int sh = g * stagger;
	int tlo = sg * a.seg - sh;
	if (tlo < 0) tlo = 0;
	int thi = sg == a.nseg - 1 ? gx : (sg + 1) * a.seg - sh;
	if (thi > gx) thi = gx;

	float* const wbuf = reinterpret_cast<float*>(p_smem + P_OFF_WBUF);
	uint32_t* const idl = reinterpret_cast<uint32_t*>(p_smem + P_OFF_IDS);
	volatile uint32_t* const meta = reinterpret_cast<volatile uint32_t*>(p_smem + P_OFF_META);   // [P_NW] + total
	QRec* const qrec = reinterpret_cast<QRec*>(p_smem + P_OFF_QREC);
	uint32_t* const qidx = reinterpret_cast<uint32_t*>(p_smem + P_OFF_QIDX);
	if (threadIdx.x == 0) meta[P_NW] = tlo >= thi ? 0u : 0xFFFFFFFFu;   // number of batches, once known
	lds_barrier();

	// ================================================================== producer state (wave `ncons`)
	const int row_i = 2 * (lane >> 5) + ((lane >> 4) & 1);
	const int py = ty * SGS_TILE + g + 2 * (4 * hh + row_i);
	const float pyf = (float)py;
	const float ylo = (float)(ty * SGS_TILE + g + 8 * hh), yhi = ylo + 6.f;
	int p_tile = tlo;
	bool p_fin = p_tile >= thi;
	uint32_t p_r0 = 0, p_r1 = 0, p_next = 0;
	float T = 1.f;
	uint32_t last = 0;
	bool done = true, all_done = true;
	float pxf = 0.f;
	bool inside = false;
	int q_head = 0, q_tail = 0, q_cnt = 0;
	uint32_t A_id = 0, B_id = 0, C_id = 0, N_id = 0, N_id1 = 0;
	float2 A_xy = make_float2(0.f, 0.f), B_xy = A_xy, N_xy = A_xy;
	float4 A_co = make_float4(0.f, 0.f, 0.f, 0.f), B_co = A_co, N_co = A_co;
	uint32_t n_r0 = 0, n_r1 = 0;
	bool n_gathered = false;

	auto tile_range = [&](int t, uint32_t& r0, uint32_t& r1) __attribute__((always_inline)) {
		if (t < thi) {
			const uint2 r = a.ranges[ty * gx + t];
			r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.x);
			r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.y);
		} else {
			r0 = r1 = 0;
		}
	};
	auto load_ids = [&](uint32_t r0, uint32_t r1, uint32_t first) __attribute__((always_inline)) -> uint32_t {
		const uint32_t pos = r0 + first + (uint32_t)lane;
		return pos < r1 ? a.point_list[pos] : 0u;
	};
	auto begin_tile = [&]() __attribute__((always_inline)) {
		const int px = p_tile * SGS_TILE + (lane & 15);
		pxf = (float)px;
		inside = px < W && py < H;
		T = 1.f;
		last = 0;
		done = !inside;
		all_done = __ballot(!done) == 0ull;
		q_head = q_tail = q_cnt = 0;
	};
	if (is_prod && !p_fin) {
		tile_range(p_tile, p_r0, p_r1);
		A_id = load_ids(p_r0, p_r1, 0);
		B_id = load_ids(p_r0, p_r1, 64);
		C_id = load_ids(p_r0, p_r1, 128);
		A_xy = a.means2D[A_id];
		A_co = a.conic_opacity[A_id];
		B_xy = a.means2D[B_id];
		B_co = a.conic_opacity[B_id];
		tile_range(p_tile + 1, n_r0, n_r1);
		N_id = load_ids(n_r0, n_r1, 0);
		N_id1 = load_ids(n_r0, n_r1, 64);
		begin_tile();
	}
	// chunk A (lane = entry) through the strip-level rejection into the queue, then advance the prefetch
	auto refill = [&]() __attribute__((always_inline)) {
		const uint32_t n_list = p_r1 - p_r0;
		const bool valid = p_next + (uint32_t)lane < n_list;
		QRec e;
		e.a2 = -0.5f * A_co.x;
		e.b2 = -A_co.y;
		e.c2 = -0.5f * A_co.z;
		e.o = A_co.w;
		e.x = A_xy.x;
		e.y = A_xy.y;
		e.id = A_id;
		// alpha = o exp(power) >= 1/255 needs power >= ln(1 / (255 o)); 1 % below it a pixel provably fails the
		// alpha test (contract exp <= 5 ulp, __logf ~1e-6)
		e.thr = __logf(1.0f / (255.0f * A_co.w)) - 0.01f;
		bool keep = valid;
		if (valid && e.a2 < 0.f && e.c2 < 0.f && 4.f * e.a2 * e.c2 - e.b2 * e.b2 > 0.f) {
			// exact maximum of the concave quadratic form over the strip's pixel box
			const float x0 = (float)(p_tile * SGS_TILE);
			const float dxl = e.x - (x0 + 15.f) - 0.01f, dxh = e.x - x0 + 0.01f;
			const float dyl = e.y - yhi - 0.01f, dyh = e.y - ylo + 0.01f;
			if (!(dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f)) {
				float qmax = -__builtin_inff();
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const float ex = k ? dxh : dxl;
					const float sy = fmin_(fmax_(-e.b2 * ex / (2.f * e.c2), dyl), dyh);
					qmax = fmax_(qmax, e.a2 * ex * ex + e.b2 * ex * sy + e.c2 * sy * sy);
					const float ey = k ? dyh : dyl;
					const float sx = fmin_(fmax_(-e.b2 * ey / (2.f * e.a2), dxl), dxh);
					qmax = fmax_(qmax, e.a2 * sx * sx + e.b2 * sx * ey + e.c2 * ey * ey);
				}
				keep = !(qmax < e.thr - 0.01f);
			}
		}
		const uint64_t km = __ballot(keep);
		if (keep) {
			int slot = q_tail + __builtin_popcountll(km & ((1ull << lane) - 1ull));
			if (slot >= P_QCAP) slot -= P_QCAP;
			qrec[slot] = e;
			qidx[slot] = p_next + (uint32_t)lane + 1u;
		}
		const int nk = __builtin_popcountll(km);
		q_tail += nk;
		if (q_tail >= P_QCAP) q_tail -= P_QCAP;
		q_cnt += nk;
		p_next += 64u;
		A_id = B_id;
		A_xy = B_xy;
		A_co = B_co;
		B_id = C_id;
		B_xy = a.means2D[B_id];
		B_co = a.conic_opacity[B_id];
		C_id = load_ids(p_r0, p_r1, p_next + 128u);
		if (!n_gathered) {
			N_xy = a.means2D[N_id];
			N_co = a.conic_opacity[N_id];
			n_gathered = true;
		}
	};

	// ================================================================== consumer state (waves 0 .. ncons - 1)
	Acc2 SA, SB;   // left tiles accumulate into SA, right tiles into SB
	if (!is_prod) {
		acc2_zero(SA);
		acc2_zero(SB);
	}
	bool has_pending = false;
	int fin_tile = -1;
	int skip_wait = 0;
	const int c0 = cbase + 64 * wave;   // this consumer's first channel
	char* const ring = p_smem + P_OFF_RING + (is_prod ? 0 : wave) * (P_NST * P_CSTAGE);
	const uint32_t ring_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)ring;
	const uint32_t plane_bytes = (uint32_t)HW * 4u;
	const uint64_t obase = (uint64_t)(a.out + (size_t)(is_prod ? cbase : c0) * HW);
	const v4i rsrc = {__builtin_amdgcn_readfirstlane((int)(uint32_t)obase),
			  __builtin_amdgcn_readfirstlane((int)(uint32_t)((obase >> 32) & 0xFFFFu)), (int)F_NUM_RECORDS, 0x00020000};
	auto finish_tile = [&](int tx) __attribute__((always_inline)) {
		const bool is_left = ((tx + sh) & 1) == 0;
		if (is_left && tx != thi - 1) {
			has_pending = true;
			return;
		}
		const bool lvalid = is_left || has_pending, rvalid = !is_left;
		const int ybase = ty * SGS_TILE + g + 8 * hh;
		const uint32_t hoff = (uint32_t)(4 * half) * plane_bytes;
		const int x = (is_left ? tx : tx - 1) * SGS_TILE + l31;
		const bool lane_ok = (l31 < 16 ? lvalid : rvalid) && x < W;
		uint32_t offs[2][2];
#pragma unroll
		for (int pb = 0; pb < 2; pb++)
#pragma unroll
			for (int rw = 0; rw < 2; rw++) {
				const int y = ybase + 4 * pb + 2 * rw;
				offs[pb][rw] = (lane_ok && y < H) ? hoff + (uint32_t)(y * W + x) * 4u : F_OOB;
			}
		if (!(dbg & 1)) store_pair2(SA, SB, rsrc, offs, plane_bytes);
		acc2_zero(SA);
		acc2_zero(SB);
		has_pending = false;
	};

	uint32_t nprod = tlo >= thi ? 0u : 0xFFFFFFFFu;
	uint32_t sk = 0;   // k mod P_NW
	for (uint32_t k = 0; k < 400000u; k++) {   // (bounded: a logic error must not hang the GPU)
		if (k >= (uint32_t)P_LEAD && k - (uint32_t)P_LEAD >= nprod) break;
		if (is_prod) {
			// ---------------------------------------------------------------- produce batch k
			if (!p_fin) {
				float* wout = wbuf + (size_t)sk * (P_WBUF / 4);
				uint32_t* iout = idl + sk * P_WROWS;
				int na = 0;
				bool closed = false;
				const int cur_tile = p_tile;
				bool cur_last = false;
				while (!closed) {
					const bool exhausted = q_cnt == 0 && p_next >= p_r1 - p_r0;
					if (all_done || exhausted) break;
					if (q_cnt == 0) {
						refill();
						continue;
					}
					const QRec e = qrec[q_head];
					const uint32_t idx1 = qidx[q_head];
					q_head = q_head + 1 == P_QCAP ? 0 : q_head + 1;
					q_cnt--;
					const float dx = e.x - pxf, dy = e.y - pyf;
					const float power = __builtin_fmaf(e.b2 * dx, dy, __builtin_fmaf(e.c2 * dy, dy, (e.a2 * dx) * dx));
					const bool cand0 = !done && !(power > 0.0f) && !(power < e.thr);
					if (__ballot(cand0) == 0ull) continue;
					const float alpha = fmin_(0.99f, e.o * expf_contract(power));
					const float test_T = T * (1.0f - alpha);
					const bool cand = cand0 && !(alpha < 1.0f / 255.0f);
					const bool stop = cand && (test_T < 0.0001f);
					const bool take = cand && !stop;
					done = done || stop;
					const float w = take ? alpha * T : 0.f;
					T = take ? test_T : T;
					last = take ? idx1 : last;
					all_done = __ballot(!done) == 0ull;
					if (__ballot(take) != 0ull) {
						wout[na * 64 + lane] = w;
						iout[na] = e.id;
						na++;
						closed = na == PB;
					}
				}
				if (!closed) {
					// tile finished: closing T * bg entry, zero padding, per-pixel outputs, next tile
					wout[na * 64 + lane] = inside ? T : 0.f;
					iout[na] = F_BG_ID;
					na++;
					for (; na < PB; na++) {
						wout[na * 64 + lane] = 0.f;
						iout[na] = F_BG_ID;
					}
					if (chunk == 0 && inside) {
						const size_t pix = (size_t)py * W + (size_t)(p_tile * SGS_TILE + (lane & 15));
						a.final_T[pix] = T;
						a.n_contrib[pix] = last;
					}
					cur_last = true;
					p_tile++;
					if (p_tile >= thi) {
						p_fin = true;
						meta[P_NW] = k + 1;
					} else {
						p_r0 = n_r0;
						p_r1 = n_r1;
						p_next = 0;
						A_id = N_id;
						if (n_gathered) {
							A_xy = N_xy;
							A_co = N_co;
						} else {
							A_xy = a.means2D[A_id];
							A_co = a.conic_opacity[A_id];
						}
						B_id = N_id1;
						B_xy = a.means2D[B_id];
						B_co = a.conic_opacity[B_id];
						C_id = load_ids(p_r0, p_r1, 128);
						tile_range(p_tile + 1, n_r0, n_r1);
						N_id = load_ids(n_r0, n_r1, 0);
						N_id1 = load_ids(n_r0, n_r1, 64);
						n_gathered = false;
						begin_tile();
					}
				}
				meta[sk] = (uint32_t)cur_tile | (cur_last ? M_LAST : 0u);
			}
		} else {
			// ---------------------------------------------------------------- consume
			// [A] the feature rows of batch k - 1 (published by the previous step's barrier): 4 x 1 KB, lanes
			//     16 i' .. 16 i' + 15 fetch entry 4 i + i' (256 B = this wave's 64 channels)
			const bool have1 = k >= 1u && k - 1u < nprod && !(dbg & 4);
			if (have1) {
				const uint32_t s1 = (sk + P_NW - 1) % P_NW, st1 = (k - 1u) % P_NST;
				const uint32_t* ids = idl + s1 * P_WROWS;
				const uint32_t dst = ring_a + st1 * P_CSTAGE;
#pragma unroll
				for (int i = 0; i < P_NDMA; i++) {
					const uint32_t id = ids[4 * i + (lane >> 4)];
					const float* row = id == F_BG_ID ? a.bg : a.features + (size_t)id * C;
					dma16(row + c0 + (lane & 15) * 4, dst + (uint32_t)i * 1024u);
				}
			}
			// [B] batch j = k - 3
			if (k >= (uint32_t)P_LEAD) {
				const uint32_t j = k - (uint32_t)P_LEAD;
				const uint32_t sj = (sk + P_NW - P_LEAD) % P_NW;
				// younger loads than bundle j: bundle j + 1 (if it exists) and bundle j + 2 = k - 1 (if it exists)
				if (skip_wait > 0) {
					skip_wait--;
				} else {
					const int younger = (j + 1u < nprod ? 1 : 0) + (have1 ? 1 : 0);
					if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * P_NDMA) : "memory");
					else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(P_NDMA) : "memory");
					else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				}
				// the tile completed by the previous batch: stores (after the NEXT bundle has landed as well, so that
				// the next step need not wait behind them)
				if (fin_tile >= 0) {
					const bool is_left = ((fin_tile + sh) & 1) == 0;
					if (!(is_left && fin_tile != thi - 1)) {
						if (have1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(P_NDMA) : "memory");
						else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
						skip_wait = 1;
					}
					finish_tile(fin_tile);
					fin_tile = -1;
				}
				const uint32_t mj = meta[sj];
				const int tj = (int)(mj & 0xFFFFu);
				const bool into_a = ((tj + sh) & 1) == 0;
				const float* stg = reinterpret_cast<const float*>(ring + (j % P_NST) * P_CSTAGE);
				const float* wb = wbuf + (size_t)sj * (P_WBUF / 4);
				if (dbg & 2) {
				} else if (!EXACT) {
					// entries 8 half + 4 kg + 0..3; A = features [entry][64 ch], B = weights [entry][64 px]
#define SGS_PC_BATCH(S_)                                                                                       \
	_Pragma("unroll") for (int kg = 0; kg < 2; kg++) {                                                     \
		const int e0 = 8 * half + 4 * kg;                                                              \
		Frag fb[2];                                                                                    \
		_Pragma("unroll") for (int pb = 0; pb < 2; pb++) {                                             \
			const float* wp = wb + e0 * 64 + 32 * pb + l31;                                        \
			const float rb[4] = {wp[0], wp[64], wp[128], wp[192]};                                 \
			fb[pb] = make_frag(rb);                                                                \
		}                                                                                              \
		_Pragma("unroll") for (int cb = 0; cb < 2; cb++) {                                             \
			const float* fp = stg + e0 * 64 + 32 * cb + l31;                                       \
			const float ra[4] = {fp[0], fp[64], fp[128], fp[192]};                                 \
			const Frag fa = make_frag(ra);                                                         \
			_Pragma("unroll") for (int pb = 0; pb < 2; pb++) {                                     \
				S_[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(fa.lo, fb[pb].hi, S_[cb][pb], 0, 0, 0); \
				S_[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(fa.hi, fb[pb].lo, S_[cb][pb], 0, 0, 0); \
				S_[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(fa.hi, fb[pb].hi, S_[cb][pb], 0, 0, 0); \
			}                                                                                      \
		}                                                                                              \
	}
					if (into_a) {
						SGS_PC_BATCH(SA)
					} else {
						SGS_PC_BATCH(SB)
					}
#undef SGS_PC_BATCH
				} else {
					// exact: entry pair pr, lane half h holds entry 2 pr + h: fma chains in list order
#define SGS_PC_BATCH(S_)                                                                                       \
	_Pragma("unroll") for (int pr = 0; pr < 8; pr++) {                                                     \
		const int e = 2 * pr + half;                                                                   \
		const float b0 = wb[e * 64 + l31], b1 = wb[e * 64 + 32 + l31];                                 \
		const float a0 = stg[e * 64 + l31], a1 = stg[e * 64 + 32 + l31];                               \
		S_[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, S_[0][0], 0, 0, 0);                    \
		S_[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, S_[0][1], 0, 0, 0);                    \
		S_[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, S_[1][0], 0, 0, 0);                    \
		S_[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, S_[1][1], 0, 0, 0);                    \
	}
					if (into_a) {
						SGS_PC_BATCH(SA)
					} else {
						SGS_PC_BATCH(SB)
					}
#undef SGS_PC_BATCH
				}
				if (mj & M_LAST) fin_tile = tj;
			}
		}
		lds_barrier();
		nprod = meta[P_NW];
		sk = sk + 1 == P_NW ? 0 : sk + 1;
	}
	if (!is_prod) {
		if (fin_tile >= 0) finish_tile(fin_tile);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
}

bool blend_forward_fused_pc_eligible(const BlendFwdArgs& a)
{
	return a.C >= 128 && a.C % 64 == 0 && !a.out_depth && a.gx > 0 && a.gy > 0 &&
	       (size_t)64 * a.W * a.H * 4 < (1ull << 32);
}

hipError_t launch_blend_forward_fused_pc(hipStream_t st, const BlendFwdArgs& a, bool exact, int seg_tiles, int dbg)
{
	static bool attr_set[2] = {false, false};
	FusedArgs f;
	f.ranges = a.ranges;
	f.point_list = a.point_list;
	f.means2D = a.means2D;
	f.conic_opacity = a.conic_opacity;
	f.features = a.features;
	f.bg = a.bg;
	f.out = a.out;
	f.final_T = a.final_T;
	f.n_contrib = a.n_contrib;
	f.W = a.W;
	f.H = a.H;
	f.C = a.C;
	f.gx = a.gx;
	f.gy = a.gy;
	// channel chunk per workgroup: the largest multiple of 64 that divides C and is <= 512 (8 consumer waves)
	int ncons = P_MAXCONS;
	while (ncons > 1 && (a.C % (64 * ncons)) != 0) ncons--;
	const int chunk_c = 64 * ncons;
	f.nchunks = a.C / chunk_c;
	int seg = seg_tiles > 0 ? seg_tiles : 16;
	while (seg > 4 && (long long)a.gy * ((a.gx + seg - 1) / seg) * 4 * f.nchunks < 2048) seg /= 2;
	seg = (seg + 1) & ~1;
	f.seg = seg;
	f.nseg = (a.gx + seg - 1) / seg;
	f.total_items = a.gy * f.nseg * 4 * f.nchunks;
	f.per_xcd = (f.total_items + 7) / 8;
	const size_t lds_bytes = (size_t)pc_lds_bytes(ncons);
	const int which = exact ? 1 : 0;
	if (!attr_set[which]) {
		hipError_t e = exact ? hipFuncSetAttribute(reinterpret_cast<const void*>(&blend_fused_pc_kernel<true>),
							   hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(P_MAXCONS))
				     : hipFuncSetAttribute(reinterpret_cast<const void*>(&blend_fused_pc_kernel<false>),
							   hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(P_MAXCONS));
		if (e != hipSuccess) return e;
		attr_set[which] = true;
	}
	const dim3 block(64 * (ncons + 1));
	if (exact)
		hipLaunchKernelGGL(blend_fused_pc_kernel<true>, dim3(f.per_xcd * 8), block, lds_bytes, st, f, ncons, chunk_c, dbg);
	else
		hipLaunchKernelGGL(blend_fused_pc_kernel<false>, dim3(f.per_xcd * 8), block, lds_bytes, st, f, ncons, chunk_c, dbg);
	return hipGetLastError();
}

} // namespace sgs
