// depth_sort.hip -- the depth presort of the Gaussians (binning modes 0 and 2): perm[r] = index of the Gaussian
// of depth rank r, ties in index order.  Same result as the stable 32-bit radix sort of (depth bits, index) the
// reference's 64-bit instance sort implies (CR/cuda_rasterizer/rasterizer_impl.cu:297-307 sorts tile << 32 | depth
// with the Gaussian index as the stable tie-break).
//
// Why not the library sort.  rocPRIM's onesweep on 1M keys is 5 kernels + 9 buffer fills = 0.19 ms of a 1.7 ms
// frame, all of it launch / latency bound (profiles/r02h_kernel_stats_views1.txt).  This one is seven small kernels
// (round 6: FOUR -- the passes count their own digits and chain on flagged count words, see CHAIN below)
// and no fill of its own, built for this size class (P ~ 10^5 .. 10^7):
//   * least-significant-digit radix sort, 8-bit digits, a workgroup per tile of 4096 keys;
//   * where a workgroup's keys go needs, per digit, the number of keys with that digit in all EARLIER tiles.  There is
//     no look-back chain and no spinning: the count matrix cnt[tile][digit] of a pass is written by a counting
//     kernel in front of it (pass 0's by the kernel that produces the keys, preprocess.hip).  A second matrix per
//     group of 32 tiles keeps the column sums short: a workgroup reads <= groups + 31 rows of 1 KB.
//   * inside a tile: a wave owns 1024 consecutive keys, 64 per step; the keys of a step that share a digit are found
//     with eight ballots (no match_any on gfx9), their rank is a popcount, one lane per digit advances the wave's
//     LDS counter; the tile is reordered through LDS so that the scatter writes runs of consecutive addresses.
// Deterministic (atomics only add counts), stable, no inter-workgroup waiting.
#include "sgs_kernels.h"
#include <cstdlib>

namespace sgs {

namespace {

constexpr int DS_ITEMS = DS_TILE / 256;   // keys per thread of the counting kernels (256 threads)
constexpr int DS_WAVES_DEFAULT = 16;
constexpr bool DS_CHAIN_DEFAULT = true;   // (SGS_DS_CHAIN=0: rounds 2-5's form with its three counting kernels)

// exclusive scan of one value per thread over the FIRST 256 threads of the workgroup (one per digit); *total = sum.  Every thread of the
// workgroup calls it (two barriers); the result is meaningless for threads >= 256.
__device__ __forceinline__ uint32_t wg_scan256(uint32_t v, uint32_t* s_tmp, uint32_t* total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t u = (uint32_t)__shfl_up((int)incl, o);
		if (lane >= o) incl += u;
	}
	if (lane == 63 && wave < 4) s_tmp[wave] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		const uint32_t x = s_tmp[w];
		if (w < wave) base += x;
		tot += x;
	}
	__syncthreads();   // s_tmp may be reused
	if (total) *total = tot;
	return base + incl - v;
}

} // namespace

// PASS 0 reads the keys where preprocess left them and takes the element's index as its value; the last pass writes
// only the values (= perm).
// SPAN (last pass, binning mode 0): the rank's record and span counts for binning_rows.hip are written here, where
// (rank, Gaussian) is known, instead of by a kernel of their own (span_counts_kernel).
// NW = waves per workgroup (round 6).  A tile is 4096 keys whatever NW is; with 4 waves (rounds 2-5) a compute unit holds ONE workgroup of four
// waves that walks through five barrier-separated phases of dependent latencies -- 17 us per pass for 8 MB of traffic.  16 waves x 4 keys per
// lane shorten every phase (4 ballot rounds instead of 16, a quarter of the LDS reorder and of the scatter per wave) at the price of a 16-row wave prefix.
// CHAIN (round 6, SGS_DS_CHAIN=1): the three counting kernels are gone.  Pass 0 (its own matrices still come from preprocess) also builds the GLOBAL
// histograms of digits 1 .. 3 -- they do not depend on the order the later passes will see the keys in -- and a pass p >= 1 counts its own tile's digits
// (the ballot ranking has them anyway), publishes the row as FLAG | count words (and n << 24 | sum into the group row) and reads the rows of the tiles
// in front of it as they appear.  A single 32-bit word carries flag and value, so no fence is involved (device-scope relaxed atomics: an XCD's L2 is
// not coherent with the other seven); a workgroup takes its tile from a ticket, so whatever it waits for belongs to a workgroup that is already running.
constexpr uint32_t DS_FLAG = 0x80000000u;
__device__ __forceinline__ uint32_t ds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int PASS, bool SPAN, int NW, bool CHAIN>
__global__ __launch_bounds__(64 * NW) void depth_sort_pass_kernel(
	int P, int groups, const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
	uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t* cnt,
	uint32_t* gcnt, DepthSortSpanOut so, uint32_t* chain, const uint32_t* __restrict__ sgcnt, int sgroups)
{
	constexpr int SHIFT = 8 * PASS;
	constexpr int NT = 64 * NW, ITEMS = DS_TILE / NT;
	constexpr bool WAITS = CHAIN && PASS > 0;   // this pass makes its own count matrix and waits for the rows in front of its tile
	const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
	int k = blockIdx.x;
	if constexpr (WAITS) {
		__shared__ int s_ticket;
		if (t == 0) s_ticket = (int)__hip_atomic_fetch_add(&chain[3 * 256 + PASS - 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
		k = s_ticket;
	}
	__shared__ uint32_t s_hist[NW][256];  // per wave: keys per digit, then the wave's start inside the tile's digit run
	__shared__ uint32_t s_base[256];      // output position of the tile's first key of each digit
	__shared__ uint32_t s_excl[256];      // start of each digit's run inside the reordered tile
	__shared__ uint32_t s_tmp[4];
	__shared__ uint32_t s_keys[DS_TILE], s_vals[DS_TILE];

	for (int q = t; q < NW * 256; q += NT) (&s_hist[0][0])[q] = 0u;

	// ---- this tile's keys: element (wave, i, lane) = index  k * 4096 + wave * (4096 / NW) + i * 64 + lane
	const uint32_t first = (uint32_t)k * DS_TILE + (uint32_t)wave * (DS_TILE / NW) + (uint32_t)lane;
	uint32_t key[ITEMS], val[ITEMS];
#pragma unroll
	for (int i = 0; i < ITEMS; i++) {
		const uint32_t idx = first + 64u * i;
		const bool ok = idx < (uint32_t)P;
		key[i] = ok ? keys_in[idx] : 0xFFFFFFFFu;
		val[i] = PASS == 0 ? idx : (ok ? vals_in[idx] : 0u);
	}

	// ---- where the tile's keys of digit d start in the output: all keys with a smaller digit, plus the keys of
	// digit d in earlier tiles (whole groups from gcnt, the tiles of this tile's own group from cnt)
	if constexpr (!WAITS) {
		const int grp = k / DS_GRP;
		uint32_t tot = 0, pre = 0;
		if (t < 256) {
			if (sgcnt) {   // three levels (more than 64 groups): <= sgroups + 31 + 31 rows instead of groups + 31
				const int sg = grp / DS_GRP;
				for (int q = 0; q < sgroups; q++) {
					const uint32_t c = sgcnt[(size_t)q * 256 + t];
					tot += c;
					pre += q < sg ? c : 0u;
				}
				for (int g = sg * DS_GRP; g < grp; g++) pre += gcnt[(size_t)g * 256 + t];
			} else {
				for (int g = 0; g < groups; g++) {
					const uint32_t c = gcnt[(size_t)g * 256 + t];
					tot += c;
					pre += g < grp ? c : 0u;
				}
			}
			for (int kk = grp * DS_GRP; kk < k; kk++) pre += cnt[(size_t)kk * 256 + t];
		}
		const uint32_t start = wg_scan256(tot, s_tmp, nullptr);
		if (t < 256) s_base[t] = start + pre;
	}
	if constexpr (CHAIN && PASS == 0) {   // the global histograms of digits 1 .. 3, for the passes behind this one
		__shared__ uint32_t s_gh[3][256];
		for (int q = t; q < 3 * 256; q += NT) (&s_gh[0][0])[q] = 0u;
		__syncthreads();
#pragma unroll
		for (int i = 0; i < ITEMS; i++)
			if (first + 64u * i < (uint32_t)P) {
				atomicAdd(&s_gh[0][(key[i] >> 8) & 255u], 1u);
				atomicAdd(&s_gh[1][(key[i] >> 16) & 255u], 1u);
				atomicAdd(&s_gh[2][key[i] >> 24], 1u);
			}
		__syncthreads();
		for (int q = t; q < 3 * 256; q += NT) {
			const uint32_t c = (&s_gh[0][0])[q];
			if (c) atomicAdd(&chain[q], c);
		}
	}
	__syncthreads();   // s_hist zeroed (wg_scan256 has barriers too; this one is for clarity)

	// ---- rank of every key among the keys of its wave with the same digit (wave order = (i, lane))
	uint32_t rank[ITEMS];
	const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
	for (int i = 0; i < ITEMS; i++) {
		const bool ok = first + 64u * i < (uint32_t)P;
		const uint32_t d = (key[i] >> SHIFT) & 255u;
		unsigned long long m = __ballot(ok);
#pragma unroll
		for (int b = 0; b < 8; b++) {
			const unsigned long long bal = __ballot((d >> b) & 1u);
			m &= ((d >> b) & 1u) ? bal : ~bal;
		}
		// (lanes that are not ok: m is meaningless for them and they do not take part)
		const uint32_t before = (uint32_t)__popcll(m & below);
		uint32_t old = 0;
		if (ok && before == 0u) {   // first lane of its digit: advance the wave's counter
			old = s_hist[wave][d];
			s_hist[wave][d] = old + (uint32_t)__popcll(m);
		}
		const int leader = ok ? (int)__builtin_ctzll(m) : lane;
		old = (uint32_t)__shfl((int)old, leader);
		rank[i] = old + before;
	}
	__syncthreads();

	// ---- per digit: the waves' starts inside the digit's run, the run's start inside the tile
	{
		uint32_t run = 0;
		if (t < 256) {
#pragma unroll
			for (int w = 0; w < NW; w++) {
				const uint32_t hv = s_hist[w][t];
				s_hist[w][t] = run;
				run += hv;
			}
		}
		if constexpr (WAITS) {   // publish this tile's row first: the tiles behind it are waiting for it
			if (t < 256) {
				__hip_atomic_store(&cnt[(size_t)k * 256 + t], DS_FLAG | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_fetch_add(&gcnt[(size_t)(k / DS_GRP) * 256 + t], (1u << 24) | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		const uint32_t ex = wg_scan256(run, s_tmp, nullptr);
		if (t < 256) s_excl[t] = ex;
		if constexpr (WAITS) {
			const uint32_t gh = t < 256 ? chain[(size_t)(PASS - 1) * 256 + t] : 0u;   // (complete: pass 0's kernel has ended)
			const uint32_t start = wg_scan256(gh, s_tmp, nullptr);
			if (t < 256) {
				const int grp = k / DS_GRP;
				uint32_t pre = 0;
				// whole groups in front (each holds DS_GRP tiles: complete when its contribution count says so), then the tiles of this group
				const int nitems = grp + (k - grp * DS_GRP);
				for (int b = 0; b < nitems; b += 8) {
					uint32_t w[8];
#pragma unroll
					for (int j = 0; j < 8; j++) {
						const int it = b + j;
						const uint32_t* ptr = it < grp ? &gcnt[(size_t)it * 256 + t] : &cnt[(size_t)(grp * DS_GRP + (it - grp)) * 256 + t];
						w[j] = it < nitems ? ds_ld(ptr) : 0u;
					}
#pragma unroll
					for (int j = 0; j < 8; j++) {
						const int it = b + j;
						if (it >= nitems) continue;
						const bool is_grp = it < grp;
						const uint32_t* ptr = is_grp ? &gcnt[(size_t)it * 256 + t] : &cnt[(size_t)(grp * DS_GRP + (it - grp)) * 256 + t];
						int spins = 0;
						while (is_grp ? (w[j] >> 24) != (uint32_t)DS_GRP : (w[j] & DS_FLAG) == 0u) {
							if (++spins > (1 << 24)) __builtin_trap();   // (a protocol error must not hang the device)
							__builtin_amdgcn_s_sleep(1);
							w[j] = ds_ld(ptr);
						}
						pre += is_grp ? (w[j] & 0xFFFFFFu) : (w[j] & ~DS_FLAG);
					}
				}
				s_base[t] = start + pre;
			}
		}
	}
	__syncthreads();

	// ---- reorder the tile in LDS
#pragma unroll
	for (int i = 0; i < ITEMS; i++) {
		if (first + 64u * i < (uint32_t)P) {
			const uint32_t d = (key[i] >> SHIFT) & 255u;
			const uint32_t lp = s_excl[d] + s_hist[wave][d] + rank[i];
			s_keys[lp] = key[i];
			s_vals[lp] = val[i];
		}
	}
	__syncthreads();

	// ---- scatter: runs of consecutive output positions
	const uint32_t nvalid = (uint32_t)P - (uint32_t)k * DS_TILE < (uint32_t)DS_TILE ? (uint32_t)P - (uint32_t)k * DS_TILE
										 : (uint32_t)DS_TILE;
	unsigned long long csum = 0;   // (SPAN) this thread's share of sum(counts64) = major instances << 32 | instances
#pragma unroll
	for (int i = 0; i < ITEMS; i++) {
		const uint32_t j = (uint32_t)t + (uint32_t)NT * i;
		const bool ok = j < nvalid;
		uint32_t kv = 0, pos = 0;
		if (ok) {
			kv = s_keys[j];
			const uint32_t d = (kv >> SHIFT) & 255u;
			pos = s_base[d] + (j - s_excl[d]);
			if (PASS < 3) keys_out[pos] = kv;
			const uint32_t g = s_vals[j];
			vals_out[pos] = g;
			if (SPAN) {   // binning_rows.hip span_counts_kernel, rank = pos
				uint4 rec = make_uint4(0u, 0u, 0u, 0u);
				uint64_t c64 = 0;
				const int rad = so.radii[g];
				if (rad > 0) {
					const float2 pm = so.means2D[g];
					uint32_t x0, y0, x1, y1;
					get_rect(pm.x, pm.y, rad, so.gx, so.gy, x0, y0, x1, y1);
					const uint32_t lo = so.major_x ? x0 : y0, hi = so.major_x ? x1 : y1;
					const uint32_t payload = so.major_x ? (y0 | (y1 << 16)) : (x0 | (x1 << 16));
					if (hi > lo) {
						rec = make_uint4(g, lo | (hi << 16), payload, 0u);
						const uint32_t nmaj = hi - lo, nmin = (payload >> 16) - (payload & 0xffffu);
						c64 = ((uint64_t)nmaj << 32) | (uint64_t)(nmaj * nmin);
					}
				}
				so.rrec[pos] = rec;
				csum += c64;   // (the per-rank counts themselves have no reader: only their total)
			}
		}
	}
	if (SPAN && so.stage_a_tab && k == 0 && t == 0) {   // stage A's one-segment tables (binning_rows.hip seg_tables_kernel, single)
		so.stage_a_tab[0] = 0u;
		so.stage_a_tab[1] = (uint32_t)P;
		so.stage_a_tab[2] = 0u;
		so.stage_a_tab[3] = so.stage_a_chunks;
		so.stage_a_tab[4] = 0u;
		so.stage_a_tab[5] = so.stage_a_groups;
	}
	if (SPAN) {   // only the TOTAL of counts64 is ever needed (num_rendered and the major-instance count): one atomic
		// per workgroup instead of a scan of the array
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) {
			const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)csum, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(csum >> 32), o);
			csum += ((unsigned long long)hi << 32) | lo;
		}
		__shared__ unsigned long long s_sum[NW];
		if (lane == 0) s_sum[wave] = csum;
		__syncthreads();
		if (t == 0) {
			unsigned long long tot = 0;
#pragma unroll
			for (int w = 0; w < NW; w++) tot += s_sum[w];
			const unsigned long long before = atomicAdd(so.total, tot);
			if (so.cc_done) {
				// the count record (capi.hip count_check_kernel) by whichever workgroup is last: its ticket is taken only when its own add has
				// RETURNED (the data dependence on `before`), so the last ticket sees every workgroup's add in the total it then reads
				const uint32_t mine = __hip_atomic_fetch_add(so.cc_done, 1u + (uint32_t)(before & 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (mine == gridDim.x - 1u) {
					const unsigned long long rl = __hip_atomic_load(so.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const unsigned long long L = rl & 0xffffffffull, R = rl >> 32;
					const uint32_t trap = (uint32_t)*so.cc_trap;
					const uint32_t abort = (L > (unsigned long long)so.cc_L_cap || R > (unsigned long long)so.cc_R_cap || trap != 0u) ? 1u : 0u;
					so.cc_rec[0] = (uint32_t)L;
					so.cc_rec[1] = (uint32_t)R;
					so.cc_rec[2] = trap;
					so.cc_rec[3] = abort;
					so.cc_host[0] = (uint32_t)L;
					so.cc_host[1] = (uint32_t)R;
					so.cc_host[2] = trap;
					so.cc_host[3] = abort;
					__threadfence_system();
				}
			}
		}
	}
}

// cnt[tile][d] / gcnt[tile / 32][d] of one pass from the keys in the order that pass reads them: a workgroup per tile,
// LDS histogram, the tile's row by plain stores, the group's row by 256 coalesced atomics.  (Counting the next pass's
// digits from inside the scatter of the previous pass was tried first: 2M scattered device-scope atomics cost 0.2 -
// 2.3 ms per pass, against 16 us for the pass itself.)
template <int SHIFT>
__global__ __launch_bounds__(256) void depth_sort_count_kernel(int P, const uint32_t* __restrict__ keys,
								uint32_t* __restrict__ cnt, uint32_t* __restrict__ gcnt,
								uint32_t* __restrict__ sgcnt)
{
	__shared__ uint32_t s_h[256];
	const int k = blockIdx.x, t = threadIdx.x;
	s_h[t] = 0u;
	__syncthreads();
#pragma unroll
	for (int i = 0; i < DS_ITEMS; i++) {
		const uint32_t idx = (uint32_t)k * DS_TILE + (uint32_t)t + 256u * i;
		if (idx < (uint32_t)P) atomicAdd(&s_h[(keys[idx] >> SHIFT) & 255u], 1u);
	}
	__syncthreads();
	const uint32_t c = s_h[t];
	cnt[(size_t)k * 256 + t] = c;
	if (c) atomicAdd(&gcnt[(size_t)(k / DS_GRP) * 256 + t], c);
	if (c && sgcnt) atomicAdd(&sgcnt[(size_t)(k / (DS_GRP * DS_GRP)) * 256 + t], c);
}

// pass 0's super-group rows from its group rows (those come from the kernel that made the keys): sgcnt0[q][d] = sum of <= 32 group rows
__global__ __launch_bounds__(256) void depth_sort_sg0_kernel(int groups, const uint32_t* __restrict__ gcnt0, uint32_t* __restrict__ sgcnt0)
{
	const int q = blockIdx.x, t = threadIdx.x;
	uint32_t s = 0;
	for (int g = q * DS_GRP; g < (q + 1) * DS_GRP && g < groups; g++) s += gcnt0[(size_t)g * 256 + t];
	sgcnt0[(size_t)q * 256 + t] = s;
}

// pass 0's count matrices from an array of keys (the forward gets them from preprocess.hip; this is for
// sgs_debug_depth_sort and the tests)
__global__ __launch_bounds__(256) void depth_sort_count0_kernel(int P, const uint32_t* __restrict__ keys,
								 uint32_t* __restrict__ cnt0, uint32_t* __restrict__ gcnt0)
{
	__shared__ uint32_t s_h[256];
	const int i = blockIdx.x * 256 + threadIdx.x;
	s_h[threadIdx.x] = 0u;
	__syncthreads();
	if (i < P) atomicAdd(&s_h[keys[i] & 255u], 1u);
	__syncthreads();
	const uint32_t c = s_h[threadIdx.x];
	if (c) {
		const uint32_t tile = blockIdx.x / (DS_TILE / 256);
		atomicAdd(&cnt0[(size_t)tile * 256 + threadIdx.x], c);
		atomicAdd(&gcnt0[(size_t)(tile / DS_GRP) * 256 + threadIdx.x], c);
	}
}

hipError_t launch_depth_sort_standalone(hipStream_t st, int P, const DepthSortLayout& lay, char* scratch,
					const uint32_t* keys, uint32_t* perm)
{
	if (P <= 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(scratch + lay.counts, 0, lay.counts_bytes, st);
	if (e != hipSuccess) return e;
	uint32_t* cnt0 = (uint32_t*)(scratch + lay.counts);
	hipLaunchKernelGGL(depth_sort_count0_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, keys, cnt0,
			   cnt0 + (size_t)lay.tiles * 256);
	return launch_depth_sort(st, P, lay, scratch, keys, perm, nullptr);
}

void depth_sort_layout(int P, DepthSortLayout* lay)
{
	const int tiles = (P + DS_TILE - 1) / DS_TILE;
	const int groups = (tiles + DS_GRP - 1) / DS_GRP;
	size_t off = 0;
	auto take = [&](size_t bytes) { off = (off + 127) & ~(size_t)127; const size_t o = off; off += bytes; return o; };
	lay->tiles = tiles;
	lay->groups = groups;
	lay->counts = take((size_t)4 * ((size_t)tiles + groups) * 256 * 4);   // 4 passes x (tile rows | group rows)
	lay->chain = take((size_t)(3 * 256 + 4) * 4);                         // ghist of digits 1 .. 3 | tickets of passes 1 .. 3 (cleared with the matrices)
	lay->sgroups = (groups + DS_GRP - 1) / DS_GRP;
	lay->sg = take((size_t)4 * (size_t)lay->sgroups * 256 * 4);           // 4 passes x super-group rows (used past 64 groups)
	lay->counts_bytes = off - lay->counts;
	lay->keys[0] = take((size_t)P * 4);
	lay->keys[1] = take((size_t)P * 4);
	lay->vals[0] = take((size_t)P * 4);
	lay->vals[1] = take((size_t)P * 4);
	lay->total = (off + 127) & ~(size_t)127;
}

// scratch: depth_sort_layout bytes, its `counts` region zeroed before the kernel that fills pass 0's matrices ran
hipError_t launch_depth_sort(hipStream_t st, int P, const DepthSortLayout& lay, char* scratch,
			     const uint32_t* depth_bits, uint32_t* perm, const DepthSortSpanOut* span)
{
	if (P <= 0) return hipSuccess;
	const size_t rows = (size_t)lay.tiles + lay.groups;
	uint32_t* cnt[4];
	uint32_t* gcnt[4];
	for (int p = 0; p < 4; p++) {
		cnt[p] = (uint32_t*)(scratch + lay.counts) + (size_t)p * rows * 256;
		gcnt[p] = cnt[p] + (size_t)lay.tiles * 256;
	}
	uint32_t* kA = (uint32_t*)(scratch + lay.keys[0]);
	uint32_t* kB = (uint32_t*)(scratch + lay.keys[1]);
	uint32_t* vA = (uint32_t*)(scratch + lay.vals[0]);
	uint32_t* vB = (uint32_t*)(scratch + lay.vals[1]);
	const dim3 grid(lay.tiles), block(256);
	const DepthSortSpanOut none{};
	// waves per workgroup of the passes: SGS_DS_WAVES = 4 (rounds 2-5) | 8 | 16 (read once); default in DS_WAVES_DEFAULT
	static const int nw = [] { const char* e = getenv("SGS_DS_WAVES"); const int v = e ? atoi(e) : DS_WAVES_DEFAULT; return (v == 4 || v == 8 || v == 16) ? v : DS_WAVES_DEFAULT; }();
	// SGS_DS_CHAIN=1 (read once; round 6): the passes count their own digits and wait for the rows in front of their tile -- no counting kernels
	static const bool chain_on = [] { const char* e = getenv("SGS_DS_CHAIN"); return e ? (*e && *e != '0') : DS_CHAIN_DEFAULT; }();
	uint32_t* chain = (uint32_t*)(scratch + lay.chain);
	// (a chained tile sums <= groups + 31 flagged rows with device-scope loads, eight at a time: past 64 groups -- 8.4 M keys -- that walk is longer than
	// a counting kernel; BASELINE config 5's 50 M keys, 382 groups, keep the three counting kernels)
	const bool use_chain = chain_on && lay.groups <= 64;
	// past 64 groups: a third level of count rows (round 6: 50 M keys 9.0 -> ? ms, profiles/r06_depth_sort_sizes.txt)
	const bool three = !use_chain && lay.groups > 64;
	uint32_t* sg[4];
	for (int p = 0; p < 4; p++) sg[p] = three ? (uint32_t*)(scratch + lay.sg) + (size_t)p * lay.sgroups * 256 : nullptr;
	if (three) hipLaunchKernelGGL(depth_sort_sg0_kernel, dim3(lay.sgroups), dim3(256), 0, st, lay.groups, gcnt[0], sg[0]);
#define DS_PASS(PASS_, SPAN_, KI_, VI_, KO_, VO_, C_, G_, SO_)                                                                        \
	do {                                                                                                                               \
		if (use_chain) hipLaunchKernelGGL((depth_sort_pass_kernel<PASS_, SPAN_, 16, true>), grid, dim3(1024), 0, st, P, lay.groups, KI_, VI_, KO_, VO_, C_, G_, SO_, chain, sg[PASS_], lay.sgroups); \
		else if (nw == 16) hipLaunchKernelGGL((depth_sort_pass_kernel<PASS_, SPAN_, 16, false>), grid, dim3(1024), 0, st, P, lay.groups, KI_, VI_, KO_, VO_, C_, G_, SO_, chain, sg[PASS_], lay.sgroups); \
		else if (nw == 8) hipLaunchKernelGGL((depth_sort_pass_kernel<PASS_, SPAN_, 8, false>), grid, dim3(512), 0, st, P, lay.groups, KI_, VI_, KO_, VO_, C_, G_, SO_, chain, sg[PASS_], lay.sgroups); \
		else hipLaunchKernelGGL((depth_sort_pass_kernel<PASS_, SPAN_, 4, false>), grid, dim3(256), 0, st, P, lay.groups, KI_, VI_, KO_, VO_, C_, G_, SO_, chain, sg[PASS_], lay.sgroups); \
	} while (0)
	DS_PASS(0, false, depth_bits, (const uint32_t*)nullptr, kA, vA, cnt[0], gcnt[0], none);
	if (!use_chain) hipLaunchKernelGGL(depth_sort_count_kernel<8>, grid, block, 0, st, P, kA, cnt[1], gcnt[1], sg[1]);
	DS_PASS(1, false, kA, vA, kB, vB, cnt[1], gcnt[1], none);
	if (!use_chain) hipLaunchKernelGGL(depth_sort_count_kernel<16>, grid, block, 0, st, P, kB, cnt[2], gcnt[2], sg[2]);
	DS_PASS(2, false, kB, vB, kA, vA, cnt[2], gcnt[2], none);
	if (!use_chain) hipLaunchKernelGGL(depth_sort_count_kernel<24>, grid, block, 0, st, P, kA, cnt[3], gcnt[3], sg[3]);
	if (span) DS_PASS(3, true, kA, vA, (uint32_t*)nullptr, perm, cnt[3], gcnt[3], *span);
	else DS_PASS(3, false, kA, vA, (uint32_t*)nullptr, perm, cnt[3], gcnt[3], none);
#undef DS_PASS
	return hipGetLastError();
}

} // namespace sgs
