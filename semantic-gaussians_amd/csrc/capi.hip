// capi.hip -- extern "C" entry points of libsgs_hip.so (see include/sgs_raster.h) and the
// host-side orchestration that the reference keeps in
// CR/cuda_rasterizer/rasterizer_impl.cu:141-441 (Rasterizer::{markVisible,forward,backward})
// and SK/simple_knn.cu:186-220 (SimpleKNN::knn).
//
// Differences from the reference that are deliberate (DESIGN.md "host orchestration"):
//  * every launch, scan, sort, memset and copy goes to the caller's stream (the reference
//    uses the legacy null stream) -- required for one-process-per-GPU view sharding and for
//    overlapping views on several streams;
//  * the image-state `ranges` array is sized by the tile count, not H*W entries;
//  * prefiltered-but-culled is reported as SGS_ETRAP instead of a device trap;
//  * per-stage hipEvent timing is available for bench.py (off by default).
#include "../../include/sgs_raster.h"
#include "sgs_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
// Process-wide DEFAULTS of the tuning options (sgs_set_*): what a stream gets when it has no override of its
// own.  Every other piece of mutable state lives in a StreamCtx (below).
std::atomic<int> g_default_opt[SGS_OPT_COUNT] = {};
float g_stage_ms[7] = {0, 0, 0, 0, 0, 0, 0};   // last resolved stage times (guarded by g_ev_mu)

// State of one (device, stream): option overrides, the adaptive work-list capacity of the split blend with its
// pinned feedback words, the backward's own capacity feedback, counters.  A forward / backward only ever touches
// the context of the stream it was called with, so concurrent callers on different streams (or devices) share
// nothing; two calls on the SAME stream from different host threads are serialised by the context's mutex.
struct StreamCtx {
	std::mutex mu;
	int opt[SGS_OPT_COUNT];
	// forward (split blend work list)
	uint32_t arena_hint = 0;
	uint32_t* usage_host = nullptr;   // pinned {slots requested, overflow flag} of this stream's last split forward
	hipEvent_t usage_ev = nullptr;    // recorded behind the copy into usage_host
	bool usage_pending = false;
	// device words [0] = tile count, [1 ..] = the tiles ordered by the work-list length of this stream's previous frame
	// (BlendFwdArgs::tile_order: the weights pre-pass of the next frame takes its tiles longest-first)
	uint32_t* tile_order = nullptr;
	size_t tile_order_cap = 0;
	// backward (work-list MFMA path)
	uint32_t bwd_hint = 0;
	uint32_t* bwd_usage_host = nullptr;
	hipEvent_t bwd_ev = nullptr;
	bool bwd_pending = false;
	// deferred-count forward (SGS_OPT_DEFER_COUNT): capacity guesses for num_rendered / the major-instance count,
	// the pinned record {num_rendered, major instances, trap flag, abort flag} of the last deferred forward
	uint32_t L_hint = 0, R_hint = 0;
	uint32_t* count_host = nullptr;
	hipEvent_t count_ev = nullptr;
	bool count_pending = false;
	bool any_forward = false;   // a forward has been issued on this (device, stream)
	// sgs_stream_set_front: the forward's front end (preprocess -> depth sort -> span partitions) is enqueued on `front` (a stream confined to a
	// small compute-unit partition), the blend on the context's own stream; ev_in orders the front end behind the stream's earlier work (the
	// previous frame's blend reads the buffers the front end overwrites), ev_front the blend behind the front end
	hipStream_t front = nullptr;
	hipEvent_t ev_in = nullptr, ev_front = nullptr;
	int last_num_rendered = 0;
	uint64_t stat[SGS_STAT_COUNT];
	StreamCtx()
	{
		for (int& o : opt) o = -1;
		for (uint64_t& v : stat) v = 0;
	}
	~StreamCtx()
	{
		if (usage_host) (void)hipHostFree(usage_host);
		if (tile_order) (void)hipFree(tile_order);
		if (bwd_usage_host) (void)hipHostFree(bwd_usage_host);
		if (count_host) (void)hipHostFree(count_host);
		if (count_ev) (void)hipEventDestroy(count_ev);
		if (usage_ev) (void)hipEventDestroy(usage_ev);
		if (bwd_ev) (void)hipEventDestroy(bwd_ev);
		if (ev_in) (void)hipEventDestroy(ev_in);
		if (ev_front) (void)hipEventDestroy(ev_front);
	}
	int option(int which) const { return opt[which] >= 0 ? opt[which] : g_default_opt[which].load(); }
	// pinned word pair + event, created on first use (under `mu`)
	bool ensure(uint32_t*& words, hipEvent_t& ev)
	{
		if (!words) {
			if (hipHostMalloc((void**)&words, 16, hipHostMallocDefault) != hipSuccess) {
				words = nullptr;
				(void)hipGetLastError();
				return false;
			}
			words[0] = words[1] = words[2] = words[3] = 0;
		}
		if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
			ev = nullptr;
			(void)hipGetLastError();
			return false;
		}
		return true;
	}
};

std::mutex g_ctx_mu;
std::map<std::pair<int, void*>, std::shared_ptr<StreamCtx>> g_ctx;

// Shared ownership: a call holds its context for its whole duration, so sgs_stream_release() from another thread
// cannot free a context that a forward / backward / result call is still using (it only drops the map's reference).
std::shared_ptr<StreamCtx> ctx_of(void* stream)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) dev = 0;
	std::lock_guard<std::mutex> lk(g_ctx_mu);
	auto& slot = g_ctx[std::make_pair(dev, stream)];
	if (!slot) slot = std::make_shared<StreamCtx>();
	return slot;
}

// The backward's work-list scratch: a PRIVATE stream-ordered pool per device.  The default pool is left alone
// (other libraries in the host process use it); this one keeps up to 4 GiB resident across synchronisations so
// that the scratch of iteration n + 1 is a pool hit instead of a driver allocation.
std::mutex g_pool_mu;
std::map<int, hipMemPool_t> g_pool;
hipMemPool_t scratch_pool()
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return nullptr;
	std::lock_guard<std::mutex> lk(g_pool_mu);
	auto it = g_pool.find(dev);
	if (it != g_pool.end()) return it->second;
	hipMemPoolProps props;
	memset(&props, 0, sizeof(props));
	props.allocType = hipMemAllocationTypePinned;
	props.handleTypes = hipMemHandleTypeNone;
	props.location.type = hipMemLocationTypeDevice;
	props.location.id = dev;
	hipMemPool_t pool = nullptr;
	if (hipMemPoolCreate(&pool, &props) != hipSuccess) {
		(void)hipGetLastError();
		pool = nullptr;
	} else {
		// memory this pool keeps across synchronisations is invisible to the host framework's allocator (torch's
		// empty_cache() cannot reclaim it): SGS_BWD_POOL_RELEASE_MB bounds it (default 4096; 0 = give everything
		// back at every synchronisation, one driver allocation per backward)
		uint64_t threshold = 4ull << 30;
		if (const char* env = getenv("SGS_BWD_POOL_RELEASE_MB")) threshold = (uint64_t)strtoull(env, nullptr, 10) << 20;
		(void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold);
	}
	g_pool[dev] = pool;
	return pool;
}

int fail(int code, const std::string& msg)
{
	g_err = msg;
	return code;
}

int fail_hip(hipError_t e, const char* what)
{
	g_err = std::string(what) + ": " + hipGetErrorString(e);
	return SGS_EHIP;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }

// 128-B aligned carving of an opaque chunk (the reference's `obtain`,
// CR/cuda_rasterizer/rasterizer_impl.h:21-27).
struct Carver {
	size_t off = 0;
	size_t take(size_t bytes)
	{
		off = align_up(off, 128);
		const size_t o = off;
		off += bytes;
		return o;
	}
};

struct GeomLayout {
	sgs_geometry_layout pub;
	size_t scan_temp, scan_temp_bytes, trap_flag, count_rec, total;
	size_t totals64, stage_a_tab, cc_done;   // inside trap_flag's 128-byte block (geom_layout)
	size_t ds;                     // depth_sort.hip scratch, directly behind trap_flag's 128 bytes (one memset clears both)
	sgs::DepthSortLayout ds_lay;
	// depth presort of the Gaussians (binning modes 0 and 2)
	size_t perm, counts_sorted;
	// mode 0: rows | tiles counts of the ranked Gaussians and their inclusive scan
	size_t counts64, rrec;
};

GeomLayout geom_layout(int P)
{
	GeomLayout g;
	Carver c;
	const size_t p = (size_t)P;
	g.pub.depths = c.take(p * 4);
	g.pub.clamped = c.take(p * 3);
	g.pub.radii = c.take(p * 4);
	g.pub.means2D = c.take(p * 8);
	g.pub.cov3D = c.take(p * 24);
	g.pub.conic_opacity = c.take(p * 16);
	g.pub.rgb = c.take(p * 12);
	g.pub.tiles_touched = c.take(p * 4);
	g.pub.point_offsets = c.take(p * 4);
	g.scan_temp_bytes = sgs::scan_temp_bytes(P);
	g.scan_temp = c.take(g.scan_temp_bytes);
	g.count_rec = c.take(16);   // deferred-count forward: {num_rendered, major instances, trap, abort}
	g.trap_flag = c.take(128);   // one 128-byte block, cleared by one memset: [0] trap word | [64] 64-bit totals (major instances << 32 | instances) | [80] stage A's six table words
	g.totals64 = g.trap_flag + 64;
	g.stage_a_tab = g.trap_flag + 80;
	g.cc_done = g.trap_flag + 104;        // (round 6) workgroups of the sort's last pass that have added their counts
	static_assert(80 + 6 * 4 <= 104 && 104 + 4 <= 128 && 64 + 8 <= 80, "the trap block's fields overlap");
	sgs::depth_sort_layout(P, &g.ds_lay);
	g.ds = c.take(g.ds_lay.total);   // (128-aligned: starts right behind trap_flag; its count matrices come first)
	g.perm = c.take(p * 4);
	g.counts_sorted = c.take(p * 4);
	g.counts64 = c.take(p * 8);
	g.rrec = c.take(p * 16);
	g.total = align_up(c.off, 128) + 128;
	g.pub.total = g.total;
	return g;
}

struct BinLayout {
	sgs_binning_layout pub;
	size_t sort_temp, sort_temp_bytes, arena, total;
	sgs::SplitArena arena_lay;
	// mode 0 row builder scratch (forward only; 0 when not requested)
	size_t rowtab, cmat, gtot, tilelen;
};

// sort_room: reserve the radix sort's temporary storage (binning modes 1 / 2 sort the instances; mode 0 -- the default, span
// partitions -- never does: 0.4 GB per frame at cfg3 that round 3 carried for nothing).  The four public arrays come first, so
// their offsets (what sgs_binning_layout_of reports and the backward uses) do not depend on it.
BinLayout bin_layout(size_t L, int sort_bits, uint32_t arena_capacity = 0, int ntiles = 0, uint32_t R = 0,
		     int gx = 0, int gy = 0, int P = 0, bool sort_room = true)
{
	BinLayout b;
	Carver c;
	b.pub.keys_unsorted = c.take(L * 8);
	b.pub.vals_unsorted = c.take(L * 4);
	b.pub.keys_sorted = c.take(L * 8);
	b.pub.point_list = c.take(L * 4);
	b.sort_temp_bytes = 0;
	if (L && sort_room) {   // room for either sorting mode's radix sort
		const size_t t64 = sgs::sort_temp_bytes(L, 0, sort_bits), t32 = sgs::sort32_temp_bytes(L, sort_bits - 32);
		b.sort_temp_bytes = t64 > t32 ? t64 : t32;
	}
	b.sort_temp = c.take(b.sort_temp_bytes);
	b.arena = 0;
	if (arena_capacity) {
		const size_t bytes = sgs::split_arena_bytes(arena_capacity, L, ntiles, &b.arena_lay);
		b.arena = c.take(bytes);
	}
	b.rowtab = b.cmat = b.gtot = b.tilelen = 0;
	if (R) {
		size_t w0, w1, w2, w3;
		sgs::row_binning_scratch(P, R, gx, gy, &w0, &w1, &w2, &w3);
		b.rowtab = c.take(w0 * 4);
		b.cmat = c.take(w1 * 4);
		b.gtot = c.take(w2 * 4);
		b.tilelen = c.take(w3 * 4);
	}
	b.total = align_up(c.off, 128) + 128;
	b.pub.total = b.total;
	return b;
}

sgs_image_layout img_layout(int W, int H)
{
	sgs_image_layout im;
	Carver c;
	const size_t n = (size_t)W * H;
	const size_t tiles = (size_t)((W + SGS_TILE - 1) / SGS_TILE) * ((H + SGS_TILE - 1) / SGS_TILE);
	im.accum_alpha = c.take(n * 4);
	im.n_contrib = c.take(n * 4);
	im.ranges = c.take(tiles * 8);
	im.total = align_up(c.off, 128) + 128;
	return im;
}

// CR/cuda_rasterizer/rasterizer_impl.cu:35-50
uint32_t higher_msb(uint32_t n)
{
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step;
		else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

// the caller's chunk may be unaligned (torch guarantees 512 B, others may not)
inline char* align_ptr(char* p) { return (char*)align_up((size_t)p, 128); }

// Per-stage device timing with hipEvents on the caller's stream.
//   mode 1: resolve at the end of the call (one extra host sync per forward)
//   mode 2: deferred -- events are parked and resolved by sgs_get_stage_ms(), so the timed
//           region of bench.py carries no extra synchronisation
struct EventSet {
	hipEvent_t ev[8];
	int n;
};
std::mutex g_ev_mu, g_ms_mu;
std::vector<EventSet> g_parked;

struct StageTimer {
	int mode;
	hipStream_t st;
	EventSet es;
	StageTimer(int m, hipStream_t s) : mode(m), st(s)
	{
		es.n = 0;
		if (mode)
			// (timing only: nobody reads memory behind these events, so their records need no system-scope fence -- the eight of a frame
			// cost 37 instead of 47 us of launch gaps, gpurun_out r05ad)
			for (auto& e : es.ev) (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
	}
	void mark()
	{
		if (mode && es.n < 8) (void)hipEventRecord(es.ev[es.n++], st);
	}
	void on(hipStream_t s) { st = s; }   // (a frame split over two streams: every mark goes to the stream its stage runs on)
	static void resolve(EventSet& s, float* ms)
	{
		(void)hipEventSynchronize(s.ev[s.n - 1]);
		for (int i = 0; i < 7; i++) {
			ms[i] = 0.f;
			if (i + 1 < s.n) (void)hipEventElapsedTime(&ms[i], s.ev[i], s.ev[i + 1]);
		}
		for (auto& e : s.ev) (void)hipEventDestroy(e);
	}
	void finish()
	{
		if (!mode) return;
		if (mode == 2) {
			std::lock_guard<std::mutex> lk(g_ev_mu);
			g_parked.push_back(es);
		} else {
			std::lock_guard<std::mutex> lk(g_ms_mu);
			resolve(es, g_stage_ms);
		}
		mode = 0;
	}
	~StageTimer()
	{
		if (mode)   // early error return: release the events
			for (auto& e : es.ev) (void)hipEventDestroy(e);
	}
};

// Deferred-count forward: the instance counts stay on the device.  rec = {num_rendered, major instances, trap flag, abort}; abort != 0 (a count
// exceeds the capacity the buffers were sized for, or the prefiltered trap fired) makes every later kernel of the frame exit.  Round 6: the record --
// on the device and in the stream's pinned host words -- is written by the LAST workgroup of the depth sort's last pass (depth_sort.hip, DepthSortSpanOut::cc_*),
// which is where the totals are added up; rounds 2-5 had a one-thread kernel for it (one more launch on the frame's critical path).

// capacity guess for a count: 1.25 x what the last frame needed, 64k granularity (stable buffer sizes)
inline uint32_t grow_hint(uint32_t hint, uint32_t used)
{
	uint64_t want = (uint64_t)used + used / 4 + 1;
	want = (want + 0xffffull) & ~0xffffull;
	if (want > 0x7fffffffull) want = 0x7fffffffull;
	if ((uint64_t)hint >= want && (uint64_t)hint <= 2 * want) return hint;   // still fits, not wasteful: keep
	return (uint32_t)want;
}

// debug != 0 (the reference's CHECK_CUDA, CR/cuda_rasterizer/auxiliary.h:166-173): launch errors are checked after every
// stage, and the stream is synchronised and checked at the END of the call -- an asynchronous fault is reported by
// that call, attributed to "forward" / "backward".  The reference synchronises after EVERY stage; its own render_chn
// passes debug=True unconditionally (model/renderer.py:182), so that behaviour would put ~8 host round trips into
// every production frame (+0.15 ms at the headline size).  SGS_DEBUG_SYNC_EVERY_STAGE=1 restores it for fault hunting.
static const bool g_sync_every_stage = [] { const char* e = getenv("SGS_DEBUG_SYNC_EVERY_STAGE"); return e && *e && *e != '0'; }();
#define SGS_CHECK_STAGE(what)                                                             \
	do {                                                                              \
		hipError_t e_ = hipGetLastError();                                        \
		if (e_ == hipSuccess && debug && g_sync_every_stage) e_ = hipStreamSynchronize(cur_st); \
		if (e_ != hipSuccess) return fail_hip(e_, what);                          \
	} while (0)

} // namespace

extern "C" {

int sgs_abi_version(void) { return SGS_ABI_VERSION; }
const char* sgs_last_error(void) { return g_err.c_str(); }

int sgs_set_blend_variant(int variant) { return g_default_opt[SGS_OPT_BLEND_VARIANT].exchange(variant); }
int sgs_set_stage_timing(int enable) { return g_default_opt[SGS_OPT_STAGE_TIMING].exchange(enable); }
int sgs_set_binning_mode(int mode) { return g_default_opt[SGS_OPT_BINNING_MODE].exchange(mode); }
long long sgs_debug_depth_sort(int P, const unsigned* keys, unsigned* perm, void* scratch, void* stream)
{
	if (P < 0) return fail(SGS_EINVAL, "bad size");
	sgs::DepthSortLayout lay;
	sgs::depth_sort_layout(P, &lay);
	if (!keys || !perm || !scratch) return (long long)lay.total + 128;   // size query
	const hipError_t e = sgs::launch_depth_sort_standalone((hipStream_t)stream, P, lay, align_ptr((char*)scratch), keys, perm);
	if (e != hipSuccess) return fail_hip(e, "depth sort");
	return 0;
}
void sgs_debug_set_sweep_trace(void* device_words) { sgs::set_sweep_trace(device_words); }
int sgs_set_backward_mode(int mode) { return g_default_opt[SGS_OPT_BACKWARD_MODE].exchange(mode); }
int sgs_build_flags(void)
{
	int f = 0;
#ifdef SGS_WITH_FUSED
	f |= 1;
#endif
#ifdef SGS_WITH_X16
	f |= 2;
#endif
#ifdef SGS_WITH_EXPERIMENTS
	f |= 4;
#endif
	return f;
}

int sgs_stream_set_option(void* stream, int option, int value)
{
	if (option < 0 || option >= SGS_OPT_COUNT) return fail(SGS_EINVAL, "unknown option");
	const std::shared_ptr<StreamCtx> c = ctx_of(stream);
	std::lock_guard<std::mutex> lk(c->mu);
	const int prev = c->opt[option];
	c->opt[option] = value < 0 ? -1 : value;
	return prev < 0 ? 0x7fffffff : prev;   // (0x7fffffff: there was no override)
}

int sgs_forward_result(void* stream, int wait, int* num_rendered)
{
	const std::shared_ptr<StreamCtx> c = ctx_of(stream);
	std::lock_guard<std::mutex> lk(c->mu);
	if (!c->any_forward)   // wrong stream, or the wrong current device: contexts are keyed by (hipGetDevice(), stream)
		return fail(SGS_EINVAL, "sgs_forward_result: no forward has been issued on this (current device, stream)");
	if (!c->count_pending) {   // the last forward on this stream was an ordinary (blocking) one
		if (num_rendered) *num_rendered = c->last_num_rendered;
		return 0;
	}
	hipError_t e = wait ? hipEventSynchronize(c->count_ev) : hipEventQuery(c->count_ev);
	if (e == hipErrorNotReady) {
		(void)hipGetLastError();
		return SGS_ENOTREADY;
	}
	if (e != hipSuccess) return fail_hip(e, "deferred count");
	const uint32_t L = c->count_host[0], R = c->count_host[1], trap = c->count_host[2], abort = c->count_host[3];
	c->count_pending = false;
	c->L_hint = grow_hint(c->L_hint, L);
	c->R_hint = grow_hint(c->R_hint, R);
	c->last_num_rendered = (int)L;
	if (num_rendered) *num_rendered = (int)L;
	if (trap) return fail(SGS_ETRAP, "Point is filtered although prefiltered is set. This shouldn't happen!");
	if (abort) {
		c->stat[SGS_STAT_DEFERRED_RETRIES]++;
		return fail(SGS_ERETRY, "deferred-count forward: the frame needs more room than the capacity guess; render it again");
	}
	return 0;
}

int sgs_stream_get_stat(void* stream, int stat, uint64_t* out)
{
	if (stat < 0 || stat >= SGS_STAT_COUNT || !out) return fail(SGS_EINVAL, "unknown statistic");
	const std::shared_ptr<StreamCtx> c = ctx_of(stream);
	std::lock_guard<std::mutex> lk(c->mu);
	*out = c->stat[stat];
	return 0;
}

int sgs_stream_release(void* stream)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) dev = 0;
	std::shared_ptr<StreamCtx> c;
	{
		std::lock_guard<std::mutex> lk(g_ctx_mu);
		auto it = g_ctx.find(std::make_pair(dev, stream));
		if (it == g_ctx.end()) return 0;
		c = it->second;
		g_ctx.erase(it);
	}
	// calls in flight on other threads keep their own reference; kernels already enqueued on the stream may still
	// write the pinned feedback words, so the stream is drained before the last reference (and with it the pinned
	// memory) goes away
	std::lock_guard<std::mutex> lk(c->mu);
	(void)hipStreamSynchronize((hipStream_t)stream);
	if (c->front) (void)hipStreamSynchronize(c->front);   // (a deferred count record is written from the front stream)
	(void)hipGetLastError();
	return 1;
}

// ---- compute-unit partitions (round 6; DESIGN.md 7.0: a 256-register, 8-wave sweep workgroup can only start on a completely EMPTY compute
// unit, so the short latency-bound kernels of other views' front ends, resident on all 256 CUs, hold it off chip-wide)
int sgs_device_cu_count(void)
{
	int dev = 0, n = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
		(void)hipGetLastError();
		return fail(SGS_EHIP, "cannot read the device's compute-unit count");
	}
	return n;
}

int sgs_stream_create_cu_range(int cu_first, int cu_count, void** stream_out)
{
	if (!stream_out) return fail(SGS_EINVAL, "null argument");
	*stream_out = nullptr;
	const int ncu = sgs_device_cu_count();
	if (ncu < 0) return ncu;
	if (cu_first < 0 || cu_count <= 0 || cu_first + cu_count > ncu) return fail(SGS_EINVAL, "compute-unit range outside the device");
	std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
	for (int b = cu_first; b < cu_first + cu_count; b++) mask[(size_t)b >> 5] |= 1u << (b & 31);
	hipStream_t s = nullptr;
	const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
	if (e != hipSuccess) return fail_hip(e, "hipExtStreamCreateWithCUMask");
	*stream_out = (void*)s;
	return 0;
}

int sgs_stream_destroy(void* stream)
{
	if (!stream) return fail(SGS_EINVAL, "the null stream cannot be destroyed");
	(void)sgs_stream_release(stream);   // (drains it and frees its context, if any)
	const hipError_t e = hipStreamDestroy((hipStream_t)stream);
	if (e != hipSuccess) return fail_hip(e, "hipStreamDestroy");
	return 0;
}

int sgs_stream_set_front(void* stream, void* front_stream)
{
	const std::shared_ptr<StreamCtx> c = ctx_of(stream);
	std::lock_guard<std::mutex> lk(c->mu);
	c->front = (front_stream == stream) ? nullptr : (hipStream_t)front_stream;
	return 0;
}

int sgs_x16_cu_ownership(void) { return (sgs::sweep3_x16_ownership() ? 1 : 0) | (sgs::bwd_fused_x16_ownership() ? 2 : 0); }

int sgs_get_stage_ms(float* ms7)
{
	std::vector<EventSet> parked;
	{
		std::lock_guard<std::mutex> lk(g_ev_mu);
		parked.swap(g_parked);
	}
	std::lock_guard<std::mutex> lk2(g_ms_mu);
	if (!parked.empty()) {   // deferred mode: mean over the parked forward calls
		double acc[7] = {0, 0, 0, 0, 0, 0, 0};
		for (auto& s : parked) {
			float ms[7];
			StageTimer::resolve(s, ms);
			for (int i = 0; i < 7; i++) acc[i] += ms[i];
		}
		for (int i = 0; i < 7; i++) g_stage_ms[i] = (float)(acc[i] / (double)parked.size());
	}
	for (int i = 0; i < 7; i++) ms7[i] = g_stage_ms[i];
	return (int)parked.size();
}

int sgs_geometry_layout_of(int P, sgs_geometry_layout* out)
{
	if (P < 0 || !out) return fail(SGS_EINVAL, "bad argument");
	*out = geom_layout(P).pub;
	return 0;
}
int sgs_binning_layout_of(int num_rendered, sgs_binning_layout* out)
{
	if (num_rendered < 0 || !out) return fail(SGS_EINVAL, "bad argument");
	// sort temp size does not depend on the number of key bits for the layout's public part
	*out = bin_layout((size_t)num_rendered, 64).pub;
	return 0;
}
int sgs_image_layout_of(int width, int height, sgs_image_layout* out)
{
	if (width < 0 || height < 0 || !out) return fail(SGS_EINVAL, "bad argument");
	*out = img_layout(width, height);
	return 0;
}
int sgs_sort_bits(int width, int height)
{
	const uint32_t gx = (width + SGS_TILE - 1) / SGS_TILE, gy = (height + SGS_TILE - 1) / SGS_TILE;
	return 32 + (int)higher_msb(gx * gy);
}

int sgs_rasterize_forward(sgs_alloc_fn geometry_buffer, void* geometry_user,
			  sgs_alloc_fn binning_buffer, void* binning_user,
			  sgs_alloc_fn image_buffer, void* image_user, int P, int D, int M,
			  const float* background, int width, int height, const float* means3D,
			  const float* shs, const float* colors_precomp, const float* opacities,
			  const float* scales, float scale_modifier, const float* rotations,
			  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
			  const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
			  int num_channels, float* out_color, float* out_depth, int* radii, int debug,
			  void* stream)
{
	hipStream_t st = (hipStream_t)stream;
	// The one-shot stream options (output pitch, norm plane) belong to THIS call whatever becomes of it: they are consumed
	// before the first early return -- bad arguments, P == 0 -- so that a call that does nothing cannot leave them armed
	// for the next forward on the stream (a stale pitch on a contiguous buffer would be an out-of-bounds write, a stale
	// norm-plane flag would render a plane where a feature map is expected).
	const std::shared_ptr<StreamCtx> cx_owner = ctx_of(stream);
	StreamCtx* const cx = cx_owner.get();
	std::lock_guard<std::mutex> ctx_lock(cx->mu);
	const int out_pitch_opt = cx->option(SGS_OPT_OUT_PITCH);
	cx->opt[SGS_OPT_OUT_PITCH] = -1;
	const bool norm_plane = cx->option(SGS_OPT_NORM_PLANE) > 0;   // out_color is ONE (H, pitch) plane that receives sum_c out[c]^2
	cx->opt[SGS_OPT_NORM_PLANE] = -1;
	const int out_bands = cx->opt[SGS_OPT_OUT_BANDS];   // (one shot, stream only: there is no process default for it)
	cx->opt[SGS_OPT_OUT_BANDS] = -1;
	if (P < 0 || width <= 0 || height <= 0 || num_channels <= 0)
		return fail(SGS_EINVAL, "bad sizes");
	if (!geometry_buffer || !binning_buffer || !image_buffer || !out_color)
		return fail(SGS_EINVAL, "null buffer callback / output");
	if (num_channels != 3 && colors_precomp == nullptr)   // rasterizer_impl.cu:243-246
		return fail(SGS_EINVAL, "For non-RGB, provide precomputed Gaussian colors!");
	if (out_depth && num_channels != 3)
		return fail(SGS_EINVAL, "the RGB-D variant renders exactly 3 channels");
	if (P == 0) return 0;   // caller returns zeros (rasterize_points.cu:85-120)
	if (!means3D || !opacities || !viewmatrix || !projmatrix || !background)
		return fail(SGS_EINVAL, "null required input");
	if (!cov3D_precomp && (!scales || !rotations))
		return fail(SGS_EINVAL, "need scales+rotations or cov3D_precomp");
	if (!colors_precomp && (!shs || !cam_pos)) return fail(SGS_EINVAL, "need shs+campos or colors_precomp");

	const float focal_y = height / (2.0f * tan_fovy);   // rasterizer_impl.cu:223-224
	const float focal_x = width / (2.0f * tan_fovx);
	const int gx = (width + SGS_TILE - 1) / SGS_TILE, gy = (height + SGS_TILE - 1) / SGS_TILE;
	const int ntiles = gx * gy;

	cx->stat[SGS_STAT_FORWARDS]++;
	cx->any_forward = true;
	if (norm_plane && (num_channels % 128 != 0 || out_depth))
		return fail(SGS_EINVAL, "SGS_OPT_NORM_PLANE needs a multiple of 128 channels and no depth plane");
	// Two-stream frame (sgs_stream_set_front): everything up to and including the span partitions goes to `fs`, the blend stays on `st`.
	hipStream_t fs = st;
	if (cx->front && cx->front != st) {
		if (!cx->ev_in && hipEventCreateWithFlags(&cx->ev_in, hipEventDisableTiming) != hipSuccess) cx->ev_in = nullptr;
		if (!cx->ev_front && hipEventCreateWithFlags(&cx->ev_front, hipEventDisableTiming) != hipSuccess) cx->ev_front = nullptr;
		if (cx->ev_in && cx->ev_front && hipEventRecord(cx->ev_in, st) == hipSuccess && hipStreamWaitEvent(cx->front, cx->ev_in, 0) == hipSuccess)
			fs = cx->front;
		else
			(void)hipGetLastError();   // (no events: the frame runs on one stream, as without a front stream)
	}
	StageTimer tm(cx->option(SGS_OPT_STAGE_TIMING), fs);
	hipStream_t cur_st = fs;   // (SGS_CHECK_STAGE: the stream the stage just enqueued ran on)

	const GeomLayout gl = geom_layout(P);
	char* gchunk = (char*)geometry_buffer(geometry_user, gl.total);
	if (!gchunk) return fail(SGS_EALLOC, "geometry buffer allocation failed");
	gchunk = align_ptr(gchunk);
	const sgs_image_layout il = img_layout(width, height);
	char* ichunk = (char*)image_buffer(image_user, il.total);
	if (!ichunk) return fail(SGS_EALLOC, "image buffer allocation failed");
	ichunk = align_ptr(ichunk);

	float* depths = (float*)(gchunk + gl.pub.depths);
	uint8_t* clamped = (uint8_t*)(gchunk + gl.pub.clamped);
	int* radii_int = (int*)(gchunk + gl.pub.radii);
	float2* means2D = (float2*)(gchunk + gl.pub.means2D);
	float* cov3D = (float*)(gchunk + gl.pub.cov3D);
	float4* conic_opacity = (float4*)(gchunk + gl.pub.conic_opacity);
	float* rgb = (float*)(gchunk + gl.pub.rgb);
	uint32_t* tiles_touched = (uint32_t*)(gchunk + gl.pub.tiles_touched);
	uint32_t* point_offsets = (uint32_t*)(gchunk + gl.pub.point_offsets);
	int* trap_flag = (int*)(gchunk + gl.trap_flag);
	if (radii == nullptr) radii = radii_int;

	// binning mode 0: sort the P Gaussians by depth bits and emit instances in that order, so
	// that the big instance sort only has to be stable on the tile bits (binning.hip)
	const int bmode = cx->option(SGS_OPT_BINNING_MODE);
#ifndef SGS_WITH_EXPERIMENTS   // (modes 1 / 2 run on rocPRIM's scan and radix sort: make EXPERIMENTS=1, csrc/binning.hip)
	if (bmode == 1 || bmode == 2)
		return fail(SGS_EINVAL, "binning modes 1 / 2 (the reference's order of operations / round 1's tile-key sort, on the library scan and sort) are not in this build (make EXPERIMENTS=1)");
	if (gx > 2048 || gy > 2048)
		return fail(SGS_EINVAL, "an image axis longer than 32768 pixels needs the tile-key sort of binning mode 2, which is not in this build (make EXPERIMENTS=1)");
#endif
	const bool presort = bmode == 0 || bmode == 2 || bmode == 3;   // (3: round 2's A/B mode with the library sort -- removed, an alias of 0)
	const bool own_sort = presort;   // depth_sort.hip
	// one clear: the trap flag and, right behind it, the depth sort's count matrices (filled by preprocess)
	hipError_t e = hipMemsetAsync(trap_flag, 0, own_sort ? 128 + gl.ds_lay.counts_bytes : 4, fs);
	if (e != hipSuccess) return fail_hip(e, "memset");
	uint32_t* ds_cnt0 = own_sort ? (uint32_t*)(gchunk + gl.ds + gl.ds_lay.counts) : nullptr;
	uint32_t* ds_gcnt0 = own_sort ? ds_cnt0 + (size_t)gl.ds_lay.tiles * 256 : nullptr;

	tm.mark();
	sgs::launch_preprocess_fwd(fs, P, D, M, means3D, scales, scale_modifier, rotations, opacities,
				   shs, cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos,
				   width, height, tan_fovx, tan_fovy, focal_x, focal_y, gx, gy,
				   prefiltered, num_channels, radii, means2D, depths, cov3D, rgb, clamped,
				   conic_opacity, tiles_touched, trap_flag, ds_cnt0, ds_gcnt0);
	SGS_CHECK_STAGE("preprocess");
	tm.mark();
	// mode 0: per-tile lists from span partitions (binning_rows.hip); its per-wave bin tables live in LDS,
	// so absurdly long grid axes (> 32k pixels) take the mode-2 path
	const bool rows = (bmode == 0 || bmode == 3) && gx <= 2048 && gy <= 2048;   // span partitions (binning_rows.hip)
	uint32_t* perm = presort ? (uint32_t*)(gchunk + gl.perm) : nullptr;
	// (the deferred-count decision: made HERE, in front of the sort, because the sort's last pass writes the count record itself -- round 6)
	const uint64_t* totals64 = (const uint64_t*)(gchunk + gl.totals64);   // (rows)
	const int defer_opt = cx->option(SGS_OPT_DEFER_COUNT);
	if (cx->count_pending && cx->count_ev && hipEventQuery(cx->count_ev) == hipSuccess) {
		// a deferred frame nobody asked about: still learn from it
		cx->L_hint = grow_hint(cx->L_hint, cx->count_host[0]);
		cx->R_hint = grow_hint(cx->R_hint, cx->count_host[1]);
		cx->count_pending = false;
	}
	(void)hipGetLastError();
	const bool defer = defer_opt > 0 && rows && !(debug && g_sync_every_stage) && (defer_opt == 2 || (cx->L_hint > 0 && cx->R_hint > 0)) &&
			   cx->ensure(cx->count_host, cx->count_ev);
	const uint32_t defer_L = defer ? (defer_opt == 2 ? 4096u : cx->L_hint) : 0u;   // (2: tests -- a capacity no real frame fits, exercises the abort)
	const uint32_t defer_R = defer ? (defer_opt == 2 ? 4096u : cx->R_hint) : 0u;
	if (presort) {
		sgs::DepthSortSpanOut span{radii, means2D, gx, gy, gx >= gy, (uint64_t*)(gchunk + gl.counts64),
					   (uint4*)(gchunk + gl.rrec), (unsigned long long*)(gchunk + gl.totals64),
					   (uint32_t*)(gchunk + gl.stage_a_tab), 0u, 0u,
					   nullptr, 0u, 0u, nullptr, nullptr, nullptr};
		if (defer) {   // (defer implies rows, i.e. the span form of the last pass)
			span.cc_done = (uint32_t*)(gchunk + gl.cc_done);
			span.cc_L_cap = defer_L;
			span.cc_R_cap = defer_R;
			span.cc_trap = trap_flag;
			span.cc_rec = (uint32_t*)(gchunk + gl.count_rec);
			span.cc_host = cx->count_host;
		}
		sgs::row_binning_stage_a_counts(P, &span.stage_a_chunks, &span.stage_a_groups);
		e = sgs::launch_depth_sort(fs, P, gl.ds_lay, gchunk + gl.ds, (const uint32_t*)depths, perm, rows ? &span : nullptr);
		if (e != hipSuccess) return fail_hip(e, "gaussian depth sort");
		if (rows) {
			// the sort's last pass has written the span counts and their total (behind the trap flag) -- no scan
		} else {
			uint32_t* counts_sorted = (uint32_t*)(gchunk + gl.counts_sorted);
			sgs::launch_gather_counts(fs, P, perm, tiles_touched, counts_sorted);
			e = sgs::launch_inclusive_scan(fs, gchunk + gl.scan_temp, gl.scan_temp_bytes, counts_sorted,
						       point_offsets, P);
		}
	} else {
		e = sgs::launch_inclusive_scan(fs, gchunk + gl.scan_temp, gl.scan_temp_bytes, tiles_touched,
					       point_offsets, P);
	}
	if (e != hipSuccess) return fail_hip(e, "inclusive scan");
	SGS_CHECK_STAGE("inclusive scan");

	// The instance counts.  Default: the one blocking read-back of the forward (rasterizer_impl.cu:283).
	// SGS_OPT_DEFER_COUNT (binning mode 0; not with debug + SGS_DEBUG_SYNC_EVERY_STAGE): nothing is read back.  The buffers and grids are sized
	// from this stream's capacity guesses (1.25 x what its previous frame needed), the true counts are checked
	// against them on the device (by the sort's last workgroup: a frame that does not fit aborts itself), copied to a pinned
	// record and reported by sgs_forward_result().  The host never waits for the GPU inside the call, so one host
	// thread can keep several streams fed.  The return value is then the CAPACITY the binning buffer was laid out
	// for, not num_rendered -- such a forward cannot be handed to sgs_rasterize_backward.
	// sum over the Gaussians of (major instances << 32 | instances): the scan's last element, or the own sort's total
	uint32_t L = 0, Rrows = 0;
	const uint32_t* abort_word = nullptr;
	if (defer) {
		L = defer_L;
		Rrows = defer_R;
		uint32_t* rec = (uint32_t*)(gchunk + gl.count_rec);
		// (the record itself was written by the sort's last pass)
		e = hipEventRecord(cx->count_ev, fs);
		if (e != hipSuccess) return fail_hip(e, "deferred count record");
		cx->count_pending = true;
		cx->stat[SGS_STAT_DEFERRED_FORWARDS]++;
		abort_word = rec + 3;
	} else {
		int host_vals[2] = {0, 0};
		uint64_t host_rl = 0;   // mode 0: row instances << 32 | num_rendered
		if (own_sort && rows) {   // the trap flag and the totals share one 128-byte block: ONE copy (each is a kernel)
			uint64_t blk[9];
			e = hipMemcpyAsync(blk, trap_flag, sizeof(blk), hipMemcpyDeviceToHost, fs);
			if (e == hipSuccess) e = hipStreamSynchronize(fs);
			host_vals[1] = (int)(uint32_t)blk[0];
			host_rl = blk[8];
		} else {
			if (rows) e = hipMemcpyAsync(&host_rl, totals64, 8, hipMemcpyDeviceToHost, fs);
			else e = hipMemcpyAsync(&host_vals[0], point_offsets + (P - 1), 4, hipMemcpyDeviceToHost, fs);
			if (e == hipSuccess) e = hipMemcpyAsync(&host_vals[1], trap_flag, 4, hipMemcpyDeviceToHost, fs);
			if (e == hipSuccess) e = hipStreamSynchronize(fs);
		}
		if (e != hipSuccess) return fail_hip(e, "num_rendered read-back");
		if (host_vals[1] != 0)
			return fail(SGS_ETRAP, "Point is filtered although prefiltered is set. This shouldn't happen!");
		L = rows ? (uint32_t)(host_rl & 0xffffffffull) : (uint32_t)host_vals[0];
		Rrows = rows ? (uint32_t)(host_rl >> 32) : 0u;
		if (rows && (host_rl & 0xffffffffull) > 0x7fffffffull) return fail(SGS_EINVAL, "num_rendered exceeds 2^31-1");
		if (L > 0x7fffffffu) return fail(SGS_EINVAL, "num_rendered exceeds 2^31-1");
		if (rows) {
			cx->L_hint = grow_hint(cx->L_hint, L);
			cx->R_hint = grow_hint(cx->R_hint, Rrows);
		}
		cx->count_pending = false;
		cx->last_num_rendered = (int)L;
	}
	tm.mark();

	const int sort_bits = 32 + (int)higher_msb((uint32_t)ntiles);
	// split blend (weights pre-pass + streaming accumulate) for the 128-channel-aligned part
	const int variant = cx->option(SGS_OPT_BLEND_VARIANT);
	// variants 32 / 33: the fused single-kernel blend (split-bf16 / exact fp32), bits [11:8] = segment length / 2
	// 32-35: the experimental single-kernel blends (contiguous output only)
#ifdef SGS_WITH_FUSED   // (make FUSED=1: the two single-kernel experiments of round 2, DESIGN.md 5.6 -- evidence, not product)
	const bool want_fused = !defer && (variant & 0xff) >= 32 && (variant & 0xff) <= 35 &&
				(out_pitch_opt <= 0 || out_pitch_opt == width);
#else
	const bool want_fused = false;
	if ((variant & 0xff) >= 32 && (variant & 0xff) <= 35 && variant < 0x100)
		return fail(SGS_EINVAL, "blend variants 32-35 (fused single-kernel experiments) are not in this build (make FUSED=1)");
#endif
#ifndef SGS_WITH_X16   // (make X16=1: the double-rate-MFMA experiments, DESIGN.md 5.10 -- reproducers, not product)
	// (the x16 forms whose workgroups leave room for foreign waves on their compute unit -- round 2's and round 3's sweeps -- and the
	// filler forms of the ping-pong sweep; the DENSE x16 ping-pong sweep, 0x1 << 16 | ..6, ships: it is the default)
	if (variant >= 16 && ((variant & 15) == 12 || (variant & 15) == 15 || ((variant & 15) == 8 && ((variant >> 8) & 15) == 8) ||
			      (((variant & 15) == 6 || (variant & 15) == 4) && ((variant >> 16) & 15) > 1)))
		return fail(SGS_EINVAL, "blend variants on v_mfma_f32_32x32x16_bf16 other than the ping-pong sweep (sweep nibble 12 / 15, 0x8.8, 0x2..3 << 16 | ..6) are not in this build (make X16=1)");
#endif
#ifndef SGS_WITH_EXPERIMENTS   // (make EXPERIMENTS=1: the development forms, csrc/Makefile)
	{
		// what ships: 0 (default) / 6 (single-kernel px4 form, also the gated fallback) / 14 (round 2's two-term sweep) / 15 (exact fp32), and the
		// word form  sweep nibble {0, 8: two-term | 6: ping-pong | 11: exact} | segment length [7:4] | workgroup order [13:12] |
		// [19:16] 1 = the ping-pong sweep on the x16 MFMA (nibble 4: free running, the default; 6: lock step), 0 = on x8  -- nothing else
		const int nib = variant & 15;
		const bool plain = variant == 0 || variant == 6 || variant == 14 || variant == 15;
		const int tune = (variant >> 16) & 15;
		const bool word = variant >= 16 && (variant & ~0x3F30FF) == 0 && (((variant >> 20) & 3) == 0 || (((variant >> 20) & 3) == 1 && (variant & 15) == 4 && tune == 1)) &&   /* [21:20]: store placement of the free-running x16 sweep */
				  ((tune == 0 && (nib == 0 || nib == 6 || nib == 8 || nib == 11)) ||
				   (tune == 1 && (nib == 6 || nib == 4)));   // bits [19:16] = 1: the ping-pong sweep on x16, lock step (6) / free running (4: what 0 selects)
		if (!plain && !word && !want_fused)
			return fail(SGS_EINVAL, "this blend variant is a development form that is not in this build (make EXPERIMENTS=1)");
	}
#endif
	const bool use_split = !want_fused && (variant == 0 || variant == 14 || variant == 15 || variant >= 16) && !out_depth && num_channels >= 128 && L > 0;
	uint32_t arena_cap = 0;
	uint64_t arena_max = 0;
	if (use_split) {
		// feedback from THIS stream's previous split forward: the copy into the pinned words was enqueued on this
		// stream before the synchronisation a few lines up, so it has completed (the event says so)
		uint32_t hint = cx->arena_hint;
		if (cx->usage_pending && cx->usage_ev && hipEventQuery(cx->usage_ev) == hipSuccess) {
			const uint32_t used = cx->usage_host[0], ovf = cx->usage_host[1];
			if (ovf) {
				hint = used + used / 2;            // `used` counts every request, also the refused ones
				cx->stat[SGS_STAT_FWD_OVERFLOWS]++;
			} else if (used) {
				hint = hint > used + used / 4 ? hint : used + used / 4;
			}
			cx->usage_pending = false;
		}
		(void)hipGetLastError();
		// floor: every tile's implicit first chunk (128 slots) + a quarter on top for the tiles that need more (round 3: a half
		// -- at cfg3 92 % of the tiles stay inside their first chunk; an overflow costs one fallback frame and grows the hint)
		if (hint < (uint32_t)ntiles * 160u) hint = (uint32_t)ntiles * 160u;
		hint = (hint + 0xffffu) & ~0xffffu;   // 64k-slot granularity keeps the buffer size stable
		cx->arena_hint = hint;
		cx->stat[SGS_STAT_ARENA_SLOTS] = hint;
		// a tile can never need more than its list length rounded up to whole chunks
		arena_max = (uint64_t)L + 128ull * (uint64_t)ntiles;
		arena_cap = (uint64_t)hint < arena_max ? hint : (uint32_t)arena_max;
	}
	const BinLayout bl = bin_layout(L, sort_bits, arena_cap, ntiles, Rrows, gx, gy, P, !rows);
	char* bchunk = (char*)binning_buffer(binning_user, bl.total);
	if (!bchunk) return fail(SGS_EALLOC, "binning buffer allocation failed");
	bchunk = align_ptr(bchunk);
	uint64_t* keys_u = (uint64_t*)(bchunk + bl.pub.keys_unsorted);
	uint32_t* vals_u = (uint32_t*)(bchunk + bl.pub.vals_unsorted);
	uint64_t* keys_s = (uint64_t*)(bchunk + bl.pub.keys_sorted);
	uint32_t* point_list = (uint32_t*)(bchunk + bl.pub.point_list);

	uint2* ranges = (uint2*)(ichunk + il.ranges);
	bool counter_reset_done = false;
	if (rows) {
		// mode 0: two span partitions, no instance sort (binning_rows.hip).  The 8-B-per-instance
		// keys_unsorted area holds the major instances (8 B each, R <= L).
		tm.mark();   // (no separate emission stage)
		e = sgs::launch_row_binning(fs, P, Rrows, gx, gy, (const uint4*)(gchunk + gl.rrec), (uint2*)keys_u,
					    (uint32_t*)(bchunk + bl.rowtab), (uint32_t*)(bchunk + bl.cmat),
					    (uint32_t*)(bchunk + bl.gtot), (uint32_t*)(bchunk + bl.tilelen), ranges, point_list,
					    abort_word, use_split ? (uint32_t*)(bchunk + bl.arena + bl.arena_lay.counter) : nullptr,
					    (uint32_t)ntiles * 128u, (const uint32_t*)(gchunk + gl.stage_a_tab));
		counter_reset_done = use_split && Rrows != 0;   // (launch_row_binning with R == 0 is just a memset)
		if (e != hipSuccess) return fail_hip(e, "row binning");
		SGS_CHECK_STAGE("row binning");
		tm.mark();
		tm.mark();   // (ranges come out of the same pass)
	} else if (presort) {
		// mode 2: 32-bit tile keys (depth order is already in the emission order).  The
		// 8-B-per-instance "keys_unsorted" area holds the unsorted and the sorted tile ids.
		uint32_t* tiles_u = (uint32_t*)keys_u;
		uint32_t* tiles_s = tiles_u + L;
		sgs::launch_emit_tile_keys(fs, P, L, means2D, point_offsets, radii, perm, gx, gy, tiles_u,
					   vals_u);
		SGS_CHECK_STAGE("emit tile keys");
		tm.mark();
		if (L > 0) {
			e = sgs::launch_sort32_pairs(fs, bchunk + bl.sort_temp, bl.sort_temp_bytes, tiles_u,
						     tiles_s, vals_u, point_list, L, sort_bits - 32);
			if (e != hipSuccess) return fail_hip(e, "radix sort");
		}
		SGS_CHECK_STAGE("radix sort");
		tm.mark();
		sgs::launch_tile_ranges32(fs, L, tiles_s, ranges, ntiles);
		SGS_CHECK_STAGE("identifyTileRanges");
		tm.mark();
	} else {
		sgs::launch_duplicate_with_keys(fs, P, means2D, depths, point_offsets, radii, gx, gy, keys_u,
						vals_u, L, nullptr);
		SGS_CHECK_STAGE("duplicateWithKeys");
		tm.mark();
		if (L > 0) {
			e = sgs::launch_sort_pairs(fs, bchunk + bl.sort_temp, bl.sort_temp_bytes, keys_u, keys_s,
						   vals_u, point_list, L, 0, sort_bits);
			if (e != hipSuccess) return fail_hip(e, "radix sort");
		}
		SGS_CHECK_STAGE("radix sort");
		tm.mark();
		sgs::launch_tile_ranges(fs, L, keys_s, ranges, ntiles);
		SGS_CHECK_STAGE("identifyTileRanges");
		tm.mark();
	}

	if (fs != st) {   // the blend waits for the lists; from here on every launch and mark is on the context's own stream
		e = hipEventRecord(cx->ev_front, fs);
		if (e == hipSuccess) e = hipStreamWaitEvent(st, cx->ev_front, 0);
		if (e != hipSuccess) return fail_hip(e, "front-end hand-over");
		tm.on(st);
	}
	cur_st = st;

	sgs::BlendFwdArgs a;
	a.ranges = ranges;
	a.point_list = point_list;
	a.W = width;
	a.H = height;
	a.C = num_channels;
	a.gx = gx;
	a.gy = gy;
	a.means2D = means2D;
	a.features = colors_precomp ? colors_precomp : rgb;
	a.conic_opacity = conic_opacity;
	a.depths = depths;
	a.bg = background;
	a.final_T = (float*)(ichunk + il.accum_alpha);
	a.n_contrib = (uint32_t*)(ichunk + il.n_contrib);
	a.out = out_color;
	a.out_depth = out_depth;
	a.pitch = out_pitch_opt > 0 ? out_pitch_opt : width;
	a.abort = abort_word;
	a.usage_host = nullptr;
	a.counter_reset_done = counter_reset_done;
	a.norm_plane = norm_plane;
	a.bands = out_bands > 1 ? out_bands : 0;
	if (a.bands) {
		const int nibv = variant >= 16 ? (variant & 15) : -1;
		if ((num_channels & 127) || out_depth || norm_plane || a.bands > 4096 ||
		    !(variant == 0 || variant == 6 || nibv == 4 || nibv == 6))
			return fail(SGS_EINVAL, "SGS_OPT_OUT_BANDS needs num_channels % 128 == 0, no depth plane, no norm plane and the default / ping-pong sweep");
	}
	if (a.pitch < width) return fail(SGS_EINVAL, "output pitch smaller than the image width");
#ifdef SGS_WITH_FUSED
	if (want_fused && (variant & 0xff) >= 34 && sgs::blend_forward_fused_pc_eligible(a)) {
		tm.mark();
		e = sgs::launch_blend_forward_fused_pc(st, a, (variant & 0xff) == 35, ((variant >> 8) & 15) * 2, (variant >> 12) & 15);
	} else if (want_fused && sgs::blend_forward_fused_eligible(a) && (size_t)128 * width * height * 4 < (1ull << 32)) {
		tm.mark();
		e = sgs::launch_blend_forward_fused(st, a, (variant & 0xff) == 33, ((variant >> 8) & 15) * 2);
	} else
#endif
	if (norm_plane && (!use_split || want_fused)) {
		// nothing rendered (L == 0) -> every pixel is the background: sum_c bg[c]^2; any other reason is a variant
		// that has no norm epilogue
		if (L > 0) return fail(SGS_EINVAL, "SGS_OPT_NORM_PLANE needs the default blend (variants 0 / 15)");
		tm.mark();
		e = sgs::launch_norm_plane_background(st, out_color, (size_t)height * a.pitch, background, num_channels);
	} else if (use_split) {
		char* arena = bchunk + bl.arena;
		// accumulate kernel (low nibble of the sweep word, blend_fwd_split.hip): default 6 = six bf16 products of the exact
		// three-term splits on the x8 MFMA ("f32-equivalent") in round 4's ping-pong sweep (blend_sweep2.hip; 14 = the same
		// products in round 3's two-workgroups-per-CU sweep, bit-identical); variant 15 = 11 = fp32-input MFMA, bit-identical to
		// the contract; variant 14 = 8 = round 2's two-term split (three products, 3 * 2^-16 per term: the fastest, not fp32-class).
		// SGS_DEFAULT_SWEEP=14 (environment, read once) restores round 3's kernel as the default for A/B runs.
		// Round 5: the default is the ping-pong sweep on the double-rate v_mfma_f32_32x32x16_bf16 (word 0x10006: sweep nibble 6, bits [19:16]
		// = 1).  DESIGN.md 5.10: dense x16 issue damages packed-fp32 results of FOREIGN waves resident on the same compute unit (0 events
		// in 180 000 forwards once victim and aggressor are confined to disjoint CU masks, 557 with shared CUs) -- and the ping-pong
		// workgroup owns its CU outright: 8 waves x 256 registers, 139 KB of LDS, every wave resident from before the first to after the
		// last matrix instruction.  SGS_DEFAULT_SWEEP=6 (environment, read once) restores the x8 form, =14 round 3's kernel (EXPERIMENTS).
		// The default of the default: its FREE-RUNNING form (word 0x10004: the two halves of the workgroup hand the ring stages over through LDS
		// counters instead of two barriers per step) -- on x8 that form lost (the lock step kept the halves' matrix phases apart);
		// with the products at half the time it wins: sweep 1.05 -> 1.01 ms alone, equal with four views in flight, bit-identical maps.
		// Round 6: ... with store placement 1 (word 0x110004, bits [21:20] = 1: only pixel block 0 of a finished tile pair is stored at once, pixel block
		// 1 leaves behind the first twelve products of the next tile's first matrix phase -- bit-identical maps, sweep -3 %, profiles/r06_sweep_store_placement.txt).
		// SGS_DEFAULT_SWEEP = 4: round 5's default (0x10004), 16: lock step on x16, 6: lock step on x8, 14: round 3's kernel (EXPERIMENTS).
		static const int default_sweep = [] {
			const char* e = getenv("SGS_DEFAULT_SWEEP");
			if (!e) return 0x110004;
			const int v = atoi(e);
			return v == 14 ? 14 : (v == 16 ? 0x10006 : (v == 4 ? 0x10004 : 6));
		}();
		const int split_word = variant >= 16 ? variant : (variant == 15 ? 11 : (variant == 14 ? 8 : default_sweep));
		if (a.bands && (split_word & 15) != 4 && (split_word & 15) != 6)   // (the word the default resolves to: SGS_DEFAULT_SWEEP=14 selects a kernel without bands)
			return fail(SGS_EINVAL, "SGS_OPT_OUT_BANDS needs the ping-pong sweep (the default)");
		if (norm_plane) {
			if ((split_word & 15) != 14 && (split_word & 15) != 11 && (split_word & 15) != 6 && (split_word & 15) != 4)
				return fail(SGS_EINVAL, "SGS_OPT_NORM_PLANE needs the default blend (variants 0 / 15)");
			e = hipMemsetAsync(out_color, 0, (size_t)height * a.pitch * sizeof(float), st);
			if (e != hipSuccess) return fail_hip(e, "memset (norm plane)");
		}
		struct MarkCtx { StageTimer* t; } mctx{&tm};
		// the stream's tile-order feedback buffer: stream-ordered allocations on `st` (hipMalloc / hipFree inside a
		// forward synchronise the whole device -- every other stream's frames in flight stall -- and are illegal under stream
		// capture), grown in powers of two so that a growing grid reallocates log(n) times; the old block is freed in stream
		// order behind the frames that still read it.  A failure is counted and costs only the schedule.  SGS_NO_TILE_ORDER=1: A/B switch.
		static const bool no_tile_order = getenv("SGS_NO_TILE_ORDER") && atoi(getenv("SGS_NO_TILE_ORDER")) != 0;
		if (!no_tile_order && cx->tile_order_cap < (size_t)ntiles) {
			if (cx->tile_order) (void)hipFreeAsync(cx->tile_order, st);
			cx->tile_order = nullptr;
			cx->tile_order_cap = 0;
			size_t want = 8192;
			while (want < (size_t)ntiles) want *= 2;
			if (hipMallocAsync((void**)&cx->tile_order, (want + 1) * 4, st) == hipSuccess && cx->tile_order &&
			    hipMemsetAsync(cx->tile_order, 0, 4, st) == hipSuccess)
				cx->tile_order_cap = want;
			else {
				(void)hipGetLastError();
				if (cx->tile_order) (void)hipFreeAsync(cx->tile_order, st);
				cx->tile_order = nullptr;
				cx->stat[SGS_STAT_TILE_ORDER_ALLOC_FAILURES]++;
			}
		}
		a.tile_order = no_tile_order ? nullptr : cx->tile_order;
		const bool can_report = cx->ensure(cx->usage_host, cx->usage_ev);
		a.usage_host = can_report ? cx->usage_host : nullptr;
		bool usage_reported = false;
		e = sgs::launch_blend_forward_split(st, a, arena, bl.arena_lay, [](void* u) { static_cast<MarkCtx*>(u)->t->mark(); }, &mctx, split_word, &usage_reported);
		if (e != hipSuccess) return fail_hip(e, "blend forward (split)");
		const uint32_t* counter = (const uint32_t*)(arena + bl.arena_lay.counter);
		const int c_split = (num_channels / 128) * 128;
		if ((uint64_t)arena_cap < arena_max)   // overflow possible: gated single-kernel fallback
			e = sgs::launch_blend_forward(st, a, 0, counter, 0);
		if (e == hipSuccess && c_split < num_channels)
			e = sgs::launch_blend_forward(st, a, 0, nullptr, c_split);
		if (e == hipSuccess && can_report) {
			if (!usage_reported) e = hipMemcpyAsync(cx->usage_host, counter, 8, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess) e = hipEventRecord(cx->usage_ev, st);
			cx->usage_pending = e == hipSuccess;
		}
	} else {
		tm.mark();   // (no weights pre-pass on this path)
		e = sgs::launch_blend_forward(st, a, (variant >= 1 && variant <= 5) ? variant : 0);   // 1-5: single-kernel forms (make EXPERIMENTS=1); everything else: px4 + remainder
	}
	if (e != hipSuccess) return fail_hip(e, "blend forward");
	SGS_CHECK_STAGE("blend forward");
	if (debug && !g_sync_every_stage) {   // one synchronisation for the whole call (see SGS_CHECK_STAGE); also of a deferred frame
		const hipError_t es = hipStreamSynchronize(st);
		if (es != hipSuccess) return fail_hip(es, "forward (debug: asynchronous error; SGS_DEBUG_SYNC_EVERY_STAGE=1 names the stage)");
	}
	tm.mark();
	tm.finish();
	return (int)L;
}

int sgs_rasterize_backward(int P, int D, int M, int R, const float* background, int width,
			   int height, const float* means3D, const float* shs,
			   const float* colors_precomp, const float* scales, float scale_modifier,
			   const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
			   const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
			   const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
			   const float* dL_dpix, int num_channels, float* dL_dmean2D, float* dL_dconic,
			   float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
			   float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream)
{
	hipStream_t st = (hipStream_t)stream;
	const hipStream_t cur_st = st;
	if (P < 0 || R < 0 || width <= 0 || height <= 0 || num_channels <= 0)
		return fail(SGS_EINVAL, "bad sizes");
	if (P == 0) return 0;
	if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer))
		return fail(SGS_EINVAL, "null state buffer");
	if (!dL_dpix || !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_dmean3D ||
	    !dL_dcov3D || !dL_dscale || !dL_drot)
		return fail(SGS_EINVAL, "null gradient buffer");
	if (shs && num_channels != 3) return fail(SGS_EINVAL, "SH colours imply 3 channels");

	const std::shared_ptr<StreamCtx> cx_owner = ctx_of(stream);
	StreamCtx* const cx = cx_owner.get();
	std::lock_guard<std::mutex> ctx_lock(cx->mu);
	const int gx = (width + SGS_TILE - 1) / SGS_TILE, gy = (height + SGS_TILE - 1) / SGS_TILE;
	const GeomLayout gl = geom_layout(P);
	const sgs_image_layout il = img_layout(width, height);
	const BinLayout bl = bin_layout((size_t)R, 64);
	char* gchunk = align_ptr(geom_buffer);
	char* ichunk = align_ptr(image_buffer);
	char* bchunk = binning_buffer ? align_ptr(binning_buffer) : nullptr;
	if (radii == nullptr) radii = (const int*)(gchunk + gl.pub.radii);

	const float focal_y = height / (2.0f * tan_fovy);
	const float focal_x = width / (2.0f * tan_fovx);

	sgs::BlendBwdArgs a;
	a.ranges = (const uint2*)(ichunk + il.ranges);
	a.point_list = bchunk ? (const uint32_t*)(bchunk + bl.pub.point_list) : nullptr;
	a.W = width;
	a.H = height;
	a.C = num_channels;
	a.gx = gx;
	a.gy = gy;
	a.bg = background;
	a.means2D = (const float2*)(gchunk + gl.pub.means2D);
	a.conic_opacity = (const float4*)(gchunk + gl.pub.conic_opacity);
	a.colors = colors_precomp ? colors_precomp : (const float*)(gchunk + gl.pub.rgb);
	a.final_T = (const float*)(ichunk + il.accum_alpha);
	a.n_contrib = (const uint32_t*)(ichunk + il.n_contrib);
	a.dL_dpix = dL_dpix;
	a.dL_dmean2D = dL_dmean2D;
	a.dL_dconic = dL_dconic;
	a.dL_dopacity = dL_dopacity;
	a.dL_dcolors = dL_dcolor;
	{   // the order this stream's last forward left (normally the forward of this very frame): a schedule, never a result
		static const bool no_tile_order = getenv("SGS_NO_TILE_ORDER") && atoi(getenv("SGS_NO_TILE_ORDER")) != 0;
		a.tile_order = (!no_tile_order && cx->tile_order && cx->tile_order_cap >= (size_t)gx * gy) ? cx->tile_order : nullptr;
	}
	// SGS_OPT_BWD_CLEARS_DCOLOR: dL_dcolor arrives uninitialised; it is cleared by the work-list pre-pass where that
	// path runs, by a memset otherwise
	const size_t dcolor_floats = (size_t)P * (size_t)num_channels;
	bool dcolor_dirty = cx->option(SGS_OPT_BWD_CLEARS_DCOLOR) > 0;
	if (R > 0) {
		hipError_t e = hipSuccess;
		bool done = false;
		const int bw_mode = cx->option(SGS_OPT_BACKWARD_MODE);
#ifndef SGS_WITH_EXPERIMENTS
		if (bw_mode == 4 || bw_mode == 5) return fail(SGS_EINVAL, "backward modes 4 / 5 (rounds 2-4's two-kernel form) are not in this build (make EXPERIMENTS=1)");
#endif
		if (bw_mode != 1 && sgs::blend_backward_mfma_eligible(a)) {
			// the forward's work list again, in stream-ordered scratch.  Its capacity adapts to what THIS
			// stream's previous backward used (the forward's hint is only the starting point: below 128
			// channels the forward never builds a work list).
			const int ntiles = gx * gy;
			uint32_t hint = cx->bwd_hint > cx->arena_hint ? cx->bwd_hint : cx->arena_hint;
			if (cx->bwd_pending && cx->bwd_ev && hipEventQuery(cx->bwd_ev) == hipSuccess) {
				const uint32_t used = cx->bwd_usage_host[0], ovf = cx->bwd_usage_host[1];
				if (ovf) {
					hint = used + used / 2;
					cx->stat[SGS_STAT_BWD_OVERFLOWS]++;   // that backward ran on the per-chunk fallback
				} else if (used && hint < used + used / 4) {
					hint = used + used / 4;
				}
				cx->bwd_pending = false;
			}
			(void)hipGetLastError();
			if (hint < (uint32_t)ntiles * 192u) hint = (uint32_t)ntiles * 192u;
			hint = (hint + 0xffffu) & ~0xffffu;
			cx->bwd_hint = hint;
			const uint64_t cap_max = (uint64_t)R + 128ull * (uint64_t)ntiles;
			uint32_t cap = (uint64_t)hint < cap_max ? hint : (uint32_t)cap_max;
			if (bw_mode == 2) cap = 128u * (uint32_t)((ntiles + 1) / 2);   // (tests: guaranteed overflow -> gated fallback)
			sgs::SplitArena lay;
			const size_t bytes = sgs::split_arena_bytes(cap, (size_t)R, ntiles, &lay, 1024);   // fp32 weight rows
			void* scratch = nullptr;
			hipMemPool_t pool = scratch_pool();
			hipError_t ea = pool ? hipMallocFromPoolAsync(&scratch, bytes + 128, pool, st)
					     : hipMallocAsync(&scratch, bytes + 128, st);
			if (ea == hipSuccess && scratch) {
				char* arena = align_ptr((char*)scratch);
				const bool fold = dcolor_dirty && (dcolor_floats & 3) == 0 && ((uintptr_t)dL_dcolor & 15u) == 0;
				if (dcolor_dirty && !fold) (void)hipMemsetAsync(dL_dcolor, 0, dcolor_floats * 4, st);
				dcolor_dirty = false;
				// modes 0 / 2 / 3: the fused kernel (one read of the gradient); 4 / 5: rounds 2-4's two kernels (split / fp32 products)
				e = sgs::launch_blend_backward_mfma(st, a, arena, lay, bw_mode == 3 || bw_mode == 5, fold ? dcolor_floats : 0,
								    bw_mode == 4 || bw_mode == 5);
				if (e == hipSuccess && cx->ensure(cx->bwd_usage_host, cx->bwd_ev)) {
					if (hipMemcpyAsync(cx->bwd_usage_host, arena + lay.counter, 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
					    hipEventRecord(cx->bwd_ev, st) == hipSuccess)
						cx->bwd_pending = true;
				}
				const hipError_t e2 = hipFreeAsync(scratch, st);
				if (e == hipSuccess) e = e2;
				done = true;
			} else {   // no scratch: this backward runs on the per-chunk kernel (counted: it is ~25x slower at C = 512)
				(void)hipGetLastError();
				cx->stat[SGS_STAT_BWD_POOL_FALLBACKS]++;
			}
		}
		if (!done) {
			if (dcolor_dirty) {
				e = hipMemsetAsync(dL_dcolor, 0, dcolor_floats * 4, st);
				if (e != hipSuccess) return fail_hip(e, "memset");
				dcolor_dirty = false;
			}
			e = sgs::launch_blend_backward(st, a);
		}
		if (e != hipSuccess) return fail_hip(e, "blend backward");
	}
	if (dcolor_dirty && dcolor_floats) {   // (nothing was rendered)
		const hipError_t e = hipMemsetAsync(dL_dcolor, 0, dcolor_floats * 4, st);
		if (e != hipSuccess) return fail_hip(e, "memset");
	}
	SGS_CHECK_STAGE("blend backward");

	const float* cov3D_ptr = cov3D_precomp ? cov3D_precomp : (const float*)(gchunk + gl.pub.cov3D);
	sgs::launch_preprocess_bwd(st, P, D, M, means3D, radii, shs,
				   (const uint8_t*)(gchunk + gl.pub.clamped), scales, rotations,
				   scale_modifier, cov3D_ptr, viewmatrix, projmatrix, focal_x, focal_y,
				   tan_fovx, tan_fovy, campos, dL_dmean2D, dL_dconic, dL_dmean3D, dL_dcolor,
				   dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
	SGS_CHECK_STAGE("preprocess backward");
	if (debug && !g_sync_every_stage) {
		const hipError_t es = hipStreamSynchronize(st);
		if (es != hipSuccess) return fail_hip(es, "backward (debug: asynchronous error; SGS_DEBUG_SYNC_EVERY_STAGE=1 names the stage)");
	}
	return 0;
}

int sgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
		     uint8_t* present, void* stream)
{
	(void)projmatrix;
	if (P < 0) return fail(SGS_EINVAL, "bad sizes");
	if (P == 0) return 0;
	if (!means3D || !viewmatrix || !present) return fail(SGS_EINVAL, "null argument");
	sgs::launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail_hip(e, "markVisible");
	return 0;
}

int sgs_fusion_compute_mapping(int N, const float* coords, const float* world_view_transform,
			       const double* intrinsics4, int image_w, int image_h, int cut_bound,
			       double vis_thres, int depth_mode, const float* depth, double* zbuf,
			       long long* mapping, double* weight, void* stream)
{
	if (N < 0 || image_w <= 0 || image_h <= 0 || cut_bound < 0) return fail(SGS_EINVAL, "bad sizes");
	if (N == 0) return 0;
	if (!coords || !world_view_transform || !intrinsics4 || !mapping || !weight)
		return fail(SGS_EINVAL, "null argument");
	if (depth_mode < 0 || depth_mode > 2) return fail(SGS_EINVAL, "depth_mode must be 0 (none), 1 (map) or 2 (surface)");
	if (depth_mode == 1 && !depth) return fail(SGS_EINVAL, "depth_mode 1 needs the (H,W) depth map");
	if (depth_mode == 2 && !zbuf) return fail(SGS_EINVAL, "depth_mode 2 needs an (H,W) float64 scratch");
	hipError_t e = sgs::launch_fusion_mapping((hipStream_t)stream, N, coords, world_view_transform, intrinsics4,
						  image_w, image_h, cut_bound, vis_thres, depth_mode, depth, zbuf,
						  mapping, weight);
	if (e != hipSuccess) return fail_hip(e, "fusion mapping");
	return 0;
}

int sgs_fusion_accumulate(int N, int C, const float* features_hwc, int image_w, int image_h,
			  const long long* mapping, float* feat_sum, float* times, void* stream)
{
	if (N < 0 || C < 0 || image_w <= 0 || image_h <= 0) return fail(SGS_EINVAL, "bad sizes");
	if (N == 0 || C == 0) return 0;
	if (!features_hwc || !mapping || !feat_sum || !times) return fail(SGS_EINVAL, "null argument");
	if ((C & 3) == 0 && ((((uintptr_t)features_hwc) | ((uintptr_t)feat_sum)) & 15u))
		return fail(SGS_EINVAL, "features and sums must be 16-byte aligned");
	hipError_t e = sgs::launch_fusion_accumulate((hipStream_t)stream, N, C, features_hwc, image_w, mapping,
						     feat_sum, times);
	if (e != hipSuccess) return fail_hip(e, "fusion accumulate");
	return 0;
}

int sgs_composite_over(int num_shards, const float* const* partial_A, const float* const* partial_T, const float* background,
		       float* out, float* T_out, int num_channels, int rows, int width, void* stream)
{
	if (num_shards < 1 || num_shards > sgs::SGS_MAX_SHARDS) return fail(SGS_EINVAL, "1 .. 16 shards");
	if (num_channels < 0 || rows < 0 || width < 0) return fail(SGS_EINVAL, "bad sizes");
	if (!partial_A || !partial_T || !out) return fail(SGS_EINVAL, "null argument");
	const size_t npix = (size_t)rows * (size_t)width;
	if (npix == 0 || num_channels == 0) return 0;
	uintptr_t align = (uintptr_t)out | (uintptr_t)(npix * 4);
	for (int s = 0; s < num_shards; s++) {
		if (!partial_A[s] || !partial_T[s]) return fail(SGS_EINVAL, "null partial");
		align |= (uintptr_t)partial_A[s] | (uintptr_t)partial_T[s];
	}
	if (align & 15u) return fail(SGS_EINVAL, "partials, output and rows * width * 4 must be 16-byte aligned");
	const hipError_t e = sgs::launch_composite_over((hipStream_t)stream, num_shards, partial_A, partial_T, background, out, T_out,
							num_channels, npix);
	if (e != hipSuccess) return fail_hip(e, "composite");
	return 0;
}

int sgs_knn_mean_dist2(int P, const float* points, float* meanDists, sgs_alloc_fn scratch,
		       void* scratch_user, void* stream)
{
	if (P < 0) return fail(SGS_EINVAL, "bad sizes");
	if (P == 0) return 0;
	if (!points || !meanDists || !scratch) return fail(SGS_EINVAL, "null argument");
	const size_t bytes = sgs::knn_scratch_bytes(P);
	char* s = (char*)scratch(scratch_user, bytes + 128);
	if (!s) return fail(SGS_EALLOC, "scratch allocation failed");
	hipError_t e = sgs::launch_knn((hipStream_t)stream, P, points, meanDists, align_ptr(s), bytes);
	if (e != hipSuccess) return fail_hip(e, "knn");
	return 0;
}

int sgs_debug_sorted_keys(int P, int num_rendered, int width, int height, const char* geom_buffer,
			  char* binning_buffer, const char* image_buffer, void* stream)
{
	if (P < 0 || num_rendered < 0 || width <= 0 || height <= 0) return fail(SGS_EINVAL, "bad sizes");
	if (num_rendered == 0) return 0;
	if (!geom_buffer || !binning_buffer || !image_buffer) return fail(SGS_EINVAL, "null state buffer");
	const GeomLayout gl = geom_layout(P);
	const BinLayout bl = bin_layout((size_t)num_rendered, 64);
	const sgs_image_layout il = img_layout(width, height);
	const char* gchunk = align_ptr(const_cast<char*>(geom_buffer));
	char* bchunk = align_ptr(binning_buffer);
	const char* ichunk = align_ptr(const_cast<char*>(image_buffer));
	const int ntiles = ((width + SGS_TILE - 1) / SGS_TILE) * ((height + SGS_TILE - 1) / SGS_TILE);
	// tile of position i = the tile whose range holds it (valid for every binning mode)
	sgs::launch_reconstruct_keys_ranges((hipStream_t)stream, ntiles, (const uint2*)(ichunk + il.ranges),
					    (const uint32_t*)(bchunk + bl.pub.point_list),
					    (const float*)(gchunk + gl.pub.depths),
					    (uint64_t*)(bchunk + bl.pub.keys_sorted));
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail_hip(e, "reconstruct keys");
	return 0;
}

int sgs_debug_expf(int n, const float* in, float* out, void* stream)
{
	if (n < 0 || (n > 0 && (!in || !out))) return fail(SGS_EINVAL, "bad argument");
	sgs::launch_debug_expf((hipStream_t)stream, n, in, out);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return fail_hip(e, "debug_expf");
	return 0;
}

} // extern "C"
