// blend_fused.hip -- the C >= 128 forward blend as ONE kernel with no intermediate in HBM.
//
// What the reference's renderCUDA does per pixel (CR/cuda_rasterizer/forward.cu:262-375: walk the tile's
// depth-sorted list, power / alpha / the three skips / transmittance, accumulate C channels, add T * bg, write
// final_T and n_contrib) is done here per WAVE, for a strip of 64 pixels and 128 channels, entirely out of
// registers and the wave's private slice of LDS.  Round 1 ran this as two kernels (blend_fwd_split.hip: a weights
// pre-pass that wrote 1 KB of blend weights per active list entry, then a streaming accumulate that read them
// back: 1.04 GB of HBM round trip per cfg3 frame and 0.31 ms of a latency-bound pre-pass).  Here the weights
// never leave the CU.
//
// Decomposition.  Workgroup = 4 waves = the 4 strips of (tile-row segment, 128-channel chunk); a strip is
// the 16 x 4 pixels of one row parity g and half h of a 16 x 16 tile (rows y = g + 2 (4 h + i), i = 0..3).
// The waves of a workgroup share nothing but the launch: there is no s_barrier in this kernel.  Each wave
//   * walks its tiles' lists itself (lane = list entry, 64 at a time, two chunks prefetched, also across the
//     tile boundary), rejects entries whose Gaussian cannot reach any pixel of the strip (exact maximum of the
//     quadratic form over the strip's pixel box) and queues the survivors in LDS;
//   * runs the alpha / transmittance chain with lane = pixel, in the contract's arithmetic (bit-identical
//     weights, n_contrib and final_T on every path), and appends the entries some pixel takes to a batch of 16:
//     fp32 weights [16][64] and the 16 Gaussian ids, both in LDS;
//   * fetches the batch's feature rows (its 512-B column slice of each) by LDS-DMA into a 3-stage ring,
//     two batches ahead of their use;
//   * multiplies: out[128 ch][64 px] += F^T W as 8 MFMA blocks of 32 x 32, default split-bf16 x 3 products
//     (DESIGN.md 5.2), or exact fp32 MFMA (v_mfma_f32_32x32x2_f32, an fma chain in list order: the contract's
//     bits).  The matrix work of batch j is cut into 16 slices issued between the chain evaluations of batch
//     j + 2, so the matrix pipe runs under the VALU work of the same wave;
//   * keeps a finished LEFT tile's accumulators in a second register set and merges them with the RIGHT
//     neighbour's by v_permlane16_swap so that every store writes complete 128-B lines (the round-1 finding:
//     partial lines cost 40 % of the store rate).
// 256 accumulator registers + the rest: one wave per SIMD (launch_bounds(256, 1)); latency is hidden by the
// DMA ring, the prefetched list chunks and the asynchronous matrix pipe rather than by other waves.
//
// Every wave computes the weights of its strip for its own 128 channels, so the chain is evaluated C / 128
// times per pixel (4 x at C = 512) -- ~25 VALU per (entry, pixel) against 2 x 128 FMAs' worth of matrix work; the
// list walk costs 28 B per entry from L2.  The T * bg term is a list entry (id SGS_BG_ID, weights = final T,
// feature row = the background vector), as in round 1.
//
// vmcnt discipline: LDS-DMA and the fast-path stores are inline asm (invisible to the compiler's waitcnt pass,
// which would otherwise drain the DMA before every LDS read); the only explicit wait is one
// s_waitcnt vmcnt(8 (LA - 1)) at the top of a step.  All other memory operations are ordinary C++: the
// compiler's own waits for them can only over-wait (in-order counter), never under-wait.
#include "blend_fused_common.h"

namespace sgs {

using namespace fused;

namespace {

constexpr int FB = 16;                 // list entries per batch
constexpr int F_NST = 3;               // feature ring stages
constexpr int F_LA = F_NST - 1;        // batches of look-ahead (bundles in flight)
constexpr int F_STAGE = FB * 512;      // 16 entries x 128 channels fp32
constexpr int F_WROWS = FB + 1;        // + one dummy row that absorbs the writes of entries no pixel takes
constexpr int F_WBUF = F_WROWS * 64 * 4;   // 16 entries x 64 pixels fp32 (+ dummy)
constexpr int F_QCAP = 80;             // candidate queue (ring): refilled with <= 64 when <= 16 are left
constexpr int F_NDMA = 8;              // LDS-DMA instructions per bundle (2 entries each)
constexpr int F_OFF_RING = 0;
constexpr int F_OFF_WBUF = F_OFF_RING + F_NST * F_STAGE;
constexpr int F_OFF_QREC = F_OFF_WBUF + F_NST * F_WBUF;
constexpr int F_OFF_QIDX = F_OFF_QREC + F_QCAP * 32;
constexpr int F_OFF_IDS = F_OFF_QIDX + F_QCAP * 4;
constexpr int F_PER_WAVE = ((F_OFF_IDS + F_NST * F_WROWS * 4) + 127) & ~127;
constexpr int F_NSLICE = 16;

static_assert(4 * F_PER_WAVE <= 160 * 1024, "LDS budget");

typedef f32x16 AccSet[4][2];   // [channel block of 32][pixel block of 32]

// A completed pair: L = left tile's accumulators, R = right tile's.  v_permlane16_swap exchanges the 16-lane half
// rows: lanes then hold complete 32-pixel lines of strip row 2 pb (first result) and 2 pb + 1 (second), and every
// store instruction writes two whole 128-B lines.  offs[pb][row]: lane byte offsets (or F_OOB).
__device__ __forceinline__ void store_pair(const AccSet& L, const AccSet& R, v4i rsrc, const uint32_t offs[2][2],
					   uint32_t plane_bytes)
{
	const uint32_t plane5 = 5u * plane_bytes;
#pragma unroll
	for (int pb = 0; pb < 2; pb++) {
		uint32_t so = 0;
		asm volatile("s_mov_b32 %0, 0" : "=s"(so));
#pragma unroll
		for (int cb = 0; cb < 4; cb++)
#pragma unroll
			for (int r = 0; r < 16; r++) {
				uint32_t x, y;   // explicit, ordered copies out of the accumulator file keep the live range at two registers
				asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(L[cb][pb][r]));
				asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(R[cb][pb][r]));
				const auto sw = __builtin_amdgcn_permlane16_swap(x, y, false, false);
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

// Zero a set with eight MFMAs whose C operand is the inline constant 0 (0 x 0 + 0): the result is born in the
// accumulator file.  (Element-wise zeroing goes through VGPR temporaries and accumulator-sized phis.)
__device__ __forceinline__ void acc_zero(AccSet& S)
{
	const s16x4 z = {0, 0, 0, 0};
	const f32x16 c0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int cb = 0; cb < 4; cb++)
#pragma unroll
		for (int pb = 0; pb < 2; pb++) S[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(z, z, c0, 0, 0, 0);
}

} // namespace

template <bool EXACT>
__global__ __launch_bounds__(256, 1) void blend_fused_kernel(const FusedArgs a)
{
	extern __shared__ __attribute__((aligned(128))) char f_smem[];
	const int b = blockIdx.x;
	const int v = (b & 7) * a.per_xcd + (b >> 3);
	if (v >= a.total_items) return;
	const int chunk = v % a.nchunks;
	const int rest = v / a.nchunks;
	const int sg = rest % a.nseg, ty = rest / a.nseg;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int g = wave & 1, hh = wave >> 1;
	const int half = lane >> 5, l31 = lane & 31;
	const int W = a.W, H = a.H, C = a.C, gx = a.gx;
	const size_t HW = (size_t)H * W;
	const int cbase = chunk * 128;
	const int stagger = (W & 31) == 16 ? 1 : 0;
	const int sh = g * stagger;
	// this wave's tiles: [tlo, thi) of tile row ty (segments of g = 1 strips start one tile early when the
	// rows of that parity begin half a line into a 128-B line, so that pairs are never split)
	int tlo = sg * a.seg - sh;
	if (tlo < 0) tlo = 0;
	int thi = sg == a.nseg - 1 ? gx : (sg + 1) * a.seg - sh;
	if (thi > gx) thi = gx;

	char* lds = f_smem + wave * F_PER_WAVE;
	const uint32_t lds_a = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds;
	float* wbuf = reinterpret_cast<float*>(lds + F_OFF_WBUF);
	QRec* qrec = reinterpret_cast<QRec*>(lds + F_OFF_QREC);
	uint32_t* qidx = reinterpret_cast<uint32_t*>(lds + F_OFF_QIDX);
	uint32_t* idl = reinterpret_cast<uint32_t*>(lds + F_OFF_IDS);

	// chain lane = pixel: block pb = lane >> 5, row 2 pb + ((lane >> 4) & 1) of the strip, column lane & 15
	const int row_i = 2 * (lane >> 5) + ((lane >> 4) & 1);
	const int py = ty * SGS_TILE + g + 2 * (4 * hh + row_i);
	const float pyf = (float)py;
	const float ylo = (float)(ty * SGS_TILE + g + 8 * hh), yhi = ylo + 6.f;   // the strip's rows span

	// ------------------------------------------------------------------ producer state
	int p_tile = tlo;            // tile being walked
	bool p_fin = p_tile >= thi;  // no more tiles
	uint32_t p_r0 = 0, p_r1 = 0; // its list range
	uint32_t p_next = 0;         // list position of the chunk in `A` (first entry)
	float T = 1.f;
	uint32_t last = 0;
	bool done = true;
	bool all_done = true;        // (uniform) every pixel of the strip is finished
	float pxf = 0.f;
	bool inside = false;
	int q_head = 0, q_tail = 0, q_cnt = 0;
	// list chunks in registers (lane = entry): A = current, B = next (gather in flight), Cid = ids after next,
	// N = the NEXT tile's first chunk, Nid1 = ids of its second chunk
	uint32_t A_id = 0, B_id = 0, C_id = 0, N_id = 0, N_id1 = 0;
	float2 A_xy = make_float2(0.f, 0.f), B_xy = A_xy, N_xy = A_xy;
	float4 A_co = make_float4(0.f, 0.f, 0.f, 0.f), B_co = A_co, N_co = A_co;
	uint32_t n_r0 = 0, n_r1 = 0;   // the next tile's range
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
		return pos < r1 ? a.point_list[pos] : 0u;   // (id 0 is a valid address; entries past the end are masked later)
	};
	// start of a tile: per-pixel state
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

	if (!p_fin) {   // prologue: first tile's chunks synchronously, the next tile's asynchronously
		tile_range(p_tile, p_r0, p_r1);
		A_id = load_ids(p_r0, p_r1, 0);
		B_id = load_ids(p_r0, p_r1, 64);
		C_id = load_ids(p_r0, p_r1, 128);
		A_xy = a.means2D[A_id];
		A_co = a.conic_opacity[A_id];
		B_xy = a.means2D[B_id];
		B_co = a.conic_opacity[B_id];
		p_next = 0;
		tile_range(p_tile + 1, n_r0, n_r1);
		N_id = load_ids(n_r0, n_r1, 0);
		N_id1 = load_ids(n_r0, n_r1, 64);
		n_gathered = false;
		begin_tile();
	}

	// ---- refill: chunk A (lane = entry) through the strip-level rejection into the queue, then advance the prefetch
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
			if (slot >= F_QCAP) slot -= F_QCAP;
			qrec[slot] = e;
			qidx[slot] = p_next + (uint32_t)lane + 1u;
		}
		const int nk = __builtin_popcountll(km);
		q_tail += nk;
		if (q_tail >= F_QCAP) q_tail -= F_QCAP;
		q_cnt += nk;
		// advance: A <- B, gather B with C's ids, request the ids after that
		p_next += 64u;
		A_id = B_id;
		A_xy = B_xy;
		A_co = B_co;
		B_id = C_id;
		B_xy = a.means2D[B_id];
		B_co = a.conic_opacity[B_id];
		C_id = load_ids(p_r0, p_r1, p_next + 128u);
		if (!n_gathered) {   // the next tile's first chunk: its ids arrived long ago
			N_xy = a.means2D[N_id];
			N_co = a.conic_opacity[N_id];
			n_gathered = true;
		}
	};

	// ------------------------------------------------------------------ consumer state
	// left tiles accumulate into SA, right tiles into SB (a line of 32 pixels = a left and a right tile; which
	// tiles are "left" alternates with the row parity when W % 32 == 16); no copies between the sets
	AccSet SA, SB;
	acc_zero(SA);
	acc_zero(SB);
	bool has_pending = false;   // SA holds a finished left tile waiting for its right neighbour
	// meta of the batches in flight (shift register: batch produced k steps ago): its tile, and whether it
	// completes the tile
	int m_tile[F_LA + 1];
	bool m_last[F_LA + 1];
#pragma unroll
	for (int k = 0; k <= F_LA; k++) {
		m_tile[k] = 0;
		m_last[k] = false;
	}
	int fin_tile = -1;     // tile completed by the previous step's batch (stored at the top of this step)
	uint32_t nprod = p_fin ? 0u : 0xFFFFFFFFu;   // total batches once the producer is finished
	uint32_t sp = 0, sj = (uint32_t)(F_NST - F_LA) % F_NST;   // ring slots of batch P (= step) and of batch j = step - LA

	// descriptor over out[cbase .. cbase + 128)[.][.]; plane stride in bytes fits 32 bits (host checks 128 H W 4 < 2^32)
	const uint32_t plane_bytes = (uint32_t)HW * 4u;
	const uint64_t obase = (uint64_t)(a.out + (size_t)cbase * HW);
	const v4i rsrc = {__builtin_amdgcn_readfirstlane((int)(uint32_t)obase),
			  __builtin_amdgcn_readfirstlane((int)(uint32_t)((obase >> 32) & 0xFFFFu)),   // stride 0
			  (int)F_NUM_RECORDS, 0x00020000};
	// A finished tile.  A left tile waits in SA for its right neighbour (unless it is the last of the range); a right
	// tile completes the line pair.  ONE store path serves every case -- pair, left tile alone, right tile alone,
	// image edges: lanes that must not store get an out-of-range offset.  (Separate code paths per case, each
	// zeroing "its" set, make accumulator-sized phis the register allocator answers with spills.)
	auto finish_tile = [&](int tx) __attribute__((always_inline)) {
		const bool is_left = ((tx + sh) & 1) == 0;
		if (is_left && tx != thi - 1) {
			has_pending = true;
			return;
		}
		const bool lvalid = is_left || has_pending, rvalid = !is_left;
		const int ybase = ty * SGS_TILE + g + 8 * hh;   // image row of strip row 0
		const uint32_t hoff = (uint32_t)(4 * half) * plane_bytes;   // lanes >= 32 hold channel + 4
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
		store_pair(SA, SB, rsrc, offs, plane_bytes);
		acc_zero(SA);
		acc_zero(SB);
		has_pending = false;
	};

	for (uint32_t step = 0; step < 400000u; step++) {   // (bounded: a logic error must not hang the GPU)
		const bool producing = !p_fin;
		const bool consuming = step >= (uint32_t)F_LA;
		if (consuming && step - (uint32_t)F_LA >= nprod) break;
#pragma unroll
		for (int k = F_LA; k > 0; k--) {
			m_tile[k] = m_tile[k - 1];
			m_last[k] = m_last[k - 1];
		}
		m_last[0] = false;
		// everything but the newest LA - 1 bundles has landed: bundle j and every list prefetch of earlier steps
		asm volatile("s_waitcnt vmcnt(%0)" : : "n"(F_NDMA * (F_LA - 1)) : "memory");

		// ---- stores of the tile completed by the previous step's batch
		if (fin_tile >= 0) {
			finish_tile(fin_tile);
			fin_tile = -1;
		}

		// ---- the queue holds >= 16 candidates (or the tile's list is used up) before the static section
		if (producing)
			while (q_cnt <= 16 && p_next < p_r1 - p_r0) refill();

		int na = 0;
		bool closed = !producing;
		float* wout = wbuf + (size_t)sp * (F_WBUF / 4);
		uint32_t* iout = idl + sp * F_WROWS;
		const int cur_tile = p_tile;
		bool cur_last = false;

		// ---- one list entry through the alpha / transmittance chain, lane = pixel, branch-free: a position with
		// nothing to do (queue empty, batch complete, strip finished) runs on a masked dummy.  Entries no pixel
		// takes land in the dummy row and do not advance `na`.
#define SGS_CHAIN_STEP()                                                                                          \
	{                                                                                                         \
		const bool adv = !closed && q_cnt > 0 && !all_done;                                               \
		const QRec e = qrec[q_head];                                                                      \
		const uint32_t idx1 = qidx[q_head];                                                               \
		const float dx = e.x - pxf, dy = e.y - pyf;                                                       \
		const float power = __builtin_fmaf(e.b2 * dx, dy, __builtin_fmaf(e.c2 * dy, dy, (e.a2 * dx) * dx)); \
		const bool cand0 = adv && !done && !(power > 0.0f) && !(power < e.thr);                           \
		const float alpha = fmin_(0.99f, e.o * expf_contract(power));                                     \
		const float test_T = T * (1.0f - alpha);                                                          \
		const bool cand = cand0 && !(alpha < 1.0f / 255.0f);                                              \
		const bool stop = cand && (test_T < 0.0001f);                                                     \
		const bool take = cand && !stop;                                                                  \
		done = done || stop;                                                                              \
		const float w = take ? alpha * T : 0.f;                                                           \
		T = take ? test_T : T;                                                                            \
		last = take ? idx1 : last;                                                                        \
		const bool any = __ballot(take) != 0ull;                                                          \
		const int row = any ? na : FB;                                                                    \
		wout[row * 64 + lane] = w;                                                                        \
		iout[row] = e.id;                                                                                 \
		na += any ? 1 : 0;                                                                                \
		closed = closed || na == FB;                                                                      \
		q_head = adv ? (q_head + 1 == F_QCAP ? 0 : q_head + 1) : q_head;                                  \
		q_cnt -= adv ? 1 : 0;                                                                             \
		all_done = __ballot(!done) == 0ull;                                                               \
	}

		// ---- batch j's matrix work in 16 static positions, each behind one chain evaluation of batch P
		if (consuming) {
			const char* stg = lds + F_OFF_RING + sj * F_STAGE;
			const float* wb = wbuf + (size_t)sj * (F_WBUF / 4);
			const bool into_a = ((m_tile[F_LA] + sh) & 1) == 0;
			if (!EXACT) {
				// position k: entries 8 half + 4 (k >> 3) + 0..3, channel block (k >> 1) & 3, pixel block k & 1;
				// raw operands are read one position ahead
				float ra[4], rb[2][4];
				Frag fa, fb[2];
#define SGS_READ_OPS(K_)                                                                                      \
	{                                                                                                     \
		constexpr int kg_ = (K_) >> 3, cb_ = ((K_) >> 1) & 3, pb_ = (K_) & 1;                         \
		const int e0_ = 8 * half + 4 * kg_;                                                           \
		if (pb_ == 0) {                                                                               \
			const float* fp_ = reinterpret_cast<const float*>(stg) + e0_ * 128 + 32 * cb_ + l31;  \
			ra[0] = fp_[0]; ra[1] = fp_[128]; ra[2] = fp_[256]; ra[3] = fp_[384];                 \
		}                                                                                             \
		if (cb_ == 0) {                                                                               \
			const float* wp_ = wb + e0_ * 64 + 32 * pb_ + l31;                                    \
			rb[pb_][0] = wp_[0]; rb[pb_][1] = wp_[64]; rb[pb_][2] = wp_[128]; rb[pb_][3] = wp_[192]; \
		}                                                                                             \
	}
#define SGS_POSITION(S_, K_)                                                                                  \
	{                                                                                                     \
		asm volatile("; position " #K_ " of " #S_);   /* keeps the two role copies from being merged into selects */ \
		constexpr int cb_ = ((K_) >> 1) & 3, pb_ = (K_) & 1;                                          \
		if (pb_ == 0) fa = make_frag(ra);                                                             \
		if (cb_ == 0) fb[pb_] = make_frag(rb[pb_]);                                                   \
		if ((K_) + 1 < F_NSLICE) SGS_READ_OPS(((K_) + 1) & 15)                                        \
		S_[cb_][pb_] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(fa.lo, fb[pb_].hi, S_[cb_][pb_], 0, 0, 0); \
		SGS_CHAIN_STEP()                                                                              \
		S_[cb_][pb_] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(fa.hi, fb[pb_].lo, S_[cb_][pb_], 0, 0, 0); \
		S_[cb_][pb_] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(fa.hi, fb[pb_].hi, S_[cb_][pb_], 0, 0, 0); \
	}
#define SGS_ALL_POSITIONS(S_)                                                                                 \
	SGS_POSITION(S_, 0) SGS_POSITION(S_, 1) SGS_POSITION(S_, 2) SGS_POSITION(S_, 3)                       \
	SGS_POSITION(S_, 4) SGS_POSITION(S_, 5) SGS_POSITION(S_, 6) SGS_POSITION(S_, 7)                       \
	SGS_POSITION(S_, 8) SGS_POSITION(S_, 9) SGS_POSITION(S_, 10) SGS_POSITION(S_, 11)                     \
	SGS_POSITION(S_, 12) SGS_POSITION(S_, 13) SGS_POSITION(S_, 14) SGS_POSITION(S_, 15)
				SGS_READ_OPS(0)
				if (into_a) {
					SGS_ALL_POSITIONS(SA)
				} else {
					SGS_ALL_POSITIONS(SB)
				}
#undef SGS_ALL_POSITIONS
#undef SGS_POSITION
#undef SGS_READ_OPS
			} else {
				// exact: position k = entry pair k >> 1 (lane half h holds entry 2 (k >> 1) + h), channel blocks
				// 2 (k & 1), 2 (k & 1) + 1, both pixel blocks: four v_mfma_f32_32x32x2_f32, fma chains in list order
				float xa[2], xb[2], kb[2];
#define SGS_READ_OPS(K_)                                                                                      \
	{                                                                                                     \
		constexpr int pr_ = (K_) >> 1, cbh_ = (K_) & 1;                                               \
		const int e_ = 2 * pr_ + half;                                                                \
		const float* fp_ = reinterpret_cast<const float*>(stg) + e_ * 128 + 64 * cbh_ + l31;          \
		xa[0] = fp_[0]; xa[1] = fp_[32];                                                              \
		if (cbh_ == 0) {                                                                              \
			const float* wp_ = wb + e_ * 64 + l31;                                                \
			xb[0] = wp_[0]; xb[1] = wp_[32];                                                      \
		}                                                                                             \
	}
#define SGS_POSITION(S_, K_)                                                                                  \
	{                                                                                                     \
		asm volatile("; position " #K_ " of " #S_);                                                   \
		constexpr int c0_ = 2 * ((K_) & 1);                                                           \
		const float a0_ = xa[0], a1_ = xa[1];                                                         \
		if (((K_) & 1) == 0) { kb[0] = xb[0]; kb[1] = xb[1]; }                                        \
		if ((K_) + 1 < F_NSLICE) SGS_READ_OPS(((K_) + 1) & 15)                                        \
		S_[c0_][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, kb[0], S_[c0_][0], 0, 0, 0);           \
		S_[c0_][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, kb[1], S_[c0_][1], 0, 0, 0);           \
		SGS_CHAIN_STEP()                                                                              \
		S_[c0_ + 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, kb[0], S_[c0_ + 1][0], 0, 0, 0);   \
		S_[c0_ + 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, kb[1], S_[c0_ + 1][1], 0, 0, 0);   \
	}
#define SGS_ALL_POSITIONS(S_)                                                                                 \
	SGS_POSITION(S_, 0) SGS_POSITION(S_, 1) SGS_POSITION(S_, 2) SGS_POSITION(S_, 3)                       \
	SGS_POSITION(S_, 4) SGS_POSITION(S_, 5) SGS_POSITION(S_, 6) SGS_POSITION(S_, 7)                       \
	SGS_POSITION(S_, 8) SGS_POSITION(S_, 9) SGS_POSITION(S_, 10) SGS_POSITION(S_, 11)                     \
	SGS_POSITION(S_, 12) SGS_POSITION(S_, 13) SGS_POSITION(S_, 14) SGS_POSITION(S_, 15)
				SGS_READ_OPS(0)
				if (into_a) {
					SGS_ALL_POSITIONS(SA)
				} else {
					SGS_ALL_POSITIONS(SB)
				}
#undef SGS_ALL_POSITIONS
#undef SGS_POSITION
#undef SGS_READ_OPS
			}
		}

		// ---- the rest of batch P: more candidates (pipeline fill, or entries nobody took), refills, and the end of the tile
		if (producing) {
			while (!closed) {
				const bool exhausted = q_cnt == 0 && p_next >= p_r1 - p_r0;
				if (all_done || exhausted) break;
				if (q_cnt == 0) refill();
				else SGS_CHAIN_STEP()
			}
			if (!closed) {
				// ---- tile finished: closing T * bg entry, zero padding, per-pixel outputs
				wout[na * 64 + lane] = inside ? T : 0.f;
				iout[na] = F_BG_ID;
				na++;
				for (; na < FB; na++) {
					wout[na * 64 + lane] = 0.f;
					iout[na] = F_BG_ID;
				}
				if (chunk == 0 && inside) {
					const size_t pix = (size_t)py * W + (size_t)(p_tile * SGS_TILE + (lane & 15));
					a.final_T[pix] = T;
					a.n_contrib[pix] = last;
				}
				cur_last = true;
				// ---- switch to the next tile
				p_tile++;
				if (p_tile >= thi) {
					p_fin = true;
					nprod = step + 1;
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
			// ---- batch P is complete: its feature rows by LDS-DMA (8 x 1 KB: lanes 0-31 entry 2 i, lanes 32-63 entry 2 i + 1)
			const uint32_t dst = lds_a + F_OFF_RING + sp * F_STAGE;
#pragma unroll
			for (int i = 0; i < F_NDMA; i++) {
				const uint32_t id = iout[2 * i + half];
				const float* row = id == F_BG_ID ? a.bg : a.features + (size_t)id * C;
				dma16(row + cbase + l31 * 4, dst + (uint32_t)i * 1024u);
			}
			m_tile[0] = cur_tile;
			m_last[0] = cur_last;
		}
#undef SGS_CHAIN_STEP
		if (consuming && m_last[F_LA]) fin_tile = m_tile[F_LA];
		sp = sp + 1 == F_NST ? 0 : sp + 1;
		sj = sj + 1 == F_NST ? 0 : sj + 1;
	}

	// ---- the last tile of the range (its batch was consumed by the final step)
	if (fin_tile >= 0) finish_tile(fin_tile);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

bool blend_forward_fused_eligible(const BlendFwdArgs& a)
{
	return a.C >= 128 && a.C % 128 == 0 && !a.out_depth && a.gx > 0 && a.gy > 0;
}

hipError_t launch_blend_forward_fused(hipStream_t st, const BlendFwdArgs& a, bool exact, int seg_tiles)
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
	f.nchunks = a.C / 128;
	// segment length (tiles per wave sweep): short enough for >= ~8 workgroups per CU in the launch, long enough to
	// amortise the pipeline fill; even, so that pairs stay inside a segment
	int seg = seg_tiles > 0 ? seg_tiles : 16;
	while (seg > 4 && (long long)a.gy * ((a.gx + seg - 1) / seg) * f.nchunks < 2048) seg /= 2;
	seg = (seg + 1) & ~1;
	f.seg = seg;
	f.nseg = (a.gx + seg - 1) / seg;
	f.total_items = a.gy * f.nseg * f.nchunks;
	f.per_xcd = (f.total_items + 7) / 8;
	const size_t lds_bytes = 4 * (size_t)F_PER_WAVE;
	const int which = exact ? 1 : 0;
	if (!attr_set[which]) {
		hipError_t e = exact ? hipFuncSetAttribute(reinterpret_cast<const void*>(&blend_fused_kernel<true>),
							   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
				     : hipFuncSetAttribute(reinterpret_cast<const void*>(&blend_fused_kernel<false>),
							   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		if (e != hipSuccess) return e;
		attr_set[which] = true;
	}
	if (exact)
		hipLaunchKernelGGL(blend_fused_kernel<true>, dim3(f.per_xcd * 8), dim3(256), lds_bytes, st, f);
	else
		hipLaunchKernelGGL(blend_fused_kernel<false>, dim3(f.per_xcd * 8), dim3(256), lds_bytes, st, f);
	return hipGetLastError();
}

} // namespace sgs
