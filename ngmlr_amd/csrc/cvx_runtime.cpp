/*
 * cvx_runtime.cpp -- host side of libcvxalign.so: the C ABI of include/cvx_align.h
 * over the gfx950 kernels of cvx_kernels.hip.
 *
 * One cvx_context = one device + one HIP stream (ngmlr creates one aligner per worker
 * thread, reference src/CS.cpp:418; contexts are independent, a context is not
 * re-entrant -- same contract as the reference's ConvexAlignFast instances).
 *
 * cvx_batch_run():
 *   plan_kernel -> (readback, host assigns kernel class / arena offsets) ->
 *   fill_ring_kernel per class -> backtrack_kernel -> (readback n_ops) -> compact_ops_kernel
 *
 * There is no CPU compute path in this file: without a device every entry point
 * returns CVX_ERR_NO_DEVICE.
 */
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cvx_align.h"
#include "cvx_host_logic.h"
#include "cvx_launch.h"
#include "cvx_types.h"

using namespace cvx;

namespace {

thread_local std::string g_err;

void set_err(const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_err = buf;
}

#define HIP_TRY(expr)                                                              \
	do {                                                                           \
		hipError_t e_ = (expr);                                                    \
		if (e_ != hipSuccess) {                                                    \
			set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return (e_ == hipErrorOutOfMemory) ? CVX_ERR_OOM : CVX_ERR_HIP;        \
		}                                                                          \
	} while (0)

template <typename T>
struct DevBuf {
	T *p = nullptr;
	size_t cap = 0; /* elements */
	int ensure(size_t n) {
		if (n <= cap) return CVX_OK;
		if (p) { (void) hipFree(p); p = nullptr; cap = 0; }
		size_t want = n + n / 8 + 64;
		hipError_t e = hipMalloc((void **) &p, want * sizeof(T));
		if (e != hipSuccess) {
			set_err("hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
			p = nullptr;
			return CVX_ERR_OOM;
		}
		cap = want;
		return CVX_OK;
	}
	void release() {
		if (p) (void) hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

}  // namespace

static const int kAuxStreams = 4;

struct cvx_context {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t aux[kAuxStreams] = {nullptr, nullptr, nullptr, nullptr};  /* concurrent fill classes */
	ScoreParams sp;
	uint64_t max_matrix_mb = 10000;
	int num_cus = 256;
	int tune_min_slots = 0;   /* tuning knob (env CVX_TUNE_MIN_M): smallest M*NW a tile may use */
	int tune_late_min = kLateMinGroups;  /* test knob (env CVX_TUNE_LATE_MIN): groups of the exactly tracked tail (huge: exact everywhere) */
	int tune_force_wrap = 0;  /* test knob (env CVX_TUNE_FORCE_WRAP16): route every tile to the int16-run kernels */
	/* pinned (page-locked) upload staging, grown on demand and reused by every upload on this
	 * handle: sequences and corridor rows are packed here by several host threads and go to the
	 * device as two DMA copies */
	void *stage_seq = nullptr, *stage_rows = nullptr;
	size_t stage_seq_cap = 0, stage_rows_cap = 0;
	/* freed batches keep their device arenas and wait here for the next upload (at most
	 * kPoolBatches): hipMalloc / hipFree of multi-GB arenas per call are slow, and hipFree
	 * synchronises the whole device, which would serialise handles that work side by side */
	std::vector<struct cvx_batch_s *> pool;
};
static const size_t kPoolBatches = 2;

namespace {

/* grow-only pinned buffer; false if the host cannot pin that much (caller falls back to pageable) */
bool ensure_pinned(void **p, size_t *cap, size_t need) {
	if (need <= *cap) return true;
	if (*p) { (void) hipHostFree(*p); *p = nullptr; *cap = 0; }
	const size_t want = need + need / 8 + 4096;
	if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); *p = nullptr; return false; }
	*cap = want;
	return true;
}

}  // namespace

struct cvx_batch_s {
	int n = 0;
	std::vector<TileIn> tin;
	std::vector<TilePlan> plan;
	std::vector<TileRun> trun;
	std::vector<TileOut> tout;
	std::vector<uint64_t> dst_off;
	std::vector<int32_t> lists;
	uint64_t ops_total = 0;
	bool ran = false;

	DevBuf<uint8_t> d_seq;
	DevBuf<RowDesc> d_rows;
	DevBuf<TileIn> d_tin;
	DevBuf<TilePlan> d_plan;
	DevBuf<TileRun> d_trun;
	DevBuf<TileOut> d_tout;
	DevBuf<uint32_t> d_dirs;
	DevBuf<int32_t> d_regions;
	DevBuf<int32_t> d_lists;
	DevBuf<int32_t> d_heads;
	DevBuf<uint64_t> d_dstoff;
	DevBuf<uint32_t> d_dense;
	DevBuf<uint8_t> d_gscratch;      /* slot state of tiles taken by the catch-all kernel */
	DevBuf<uint64_t> d_gscratch_off;

	hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	std::vector<hipEvent_t> lev;          /* 3 events per fill class: start, two-phase pass done, exact pass done */
	std::vector<cvx_launch_info> launches;
	cvx_timing timing;

	void release() {
		d_seq.release(); d_rows.release(); d_tin.release(); d_plan.release(); d_trun.release();
		d_tout.release(); d_dirs.release(); d_regions.release(); d_lists.release();
		d_heads.release(); d_dstoff.release(); d_dense.release(); d_gscratch.release(); d_gscratch_off.release();
		for (auto &e : ev) if (e) { (void) hipEventDestroy(e); e = nullptr; }
		for (auto &e : lev) if (e) (void) hipEventDestroy(e);
		lev.clear();
	}
};

extern "C" {

const char *cvx_last_error(void) { return g_err.c_str(); }

int cvx_abi_version(void) { return CVX_ABI_VERSION; }

int cvx_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int cvx_create(int device_id, const cvx_params *p, uint64_t max_matrix_mb, cvx_handle *out) {
	if (!p || !out) { set_err("cvx_create: NULL argument"); return CVX_ERR_ARG; }
	*out = nullptr;
	/* The device kernels implement the scalar recurrence (reference
	 * src/ConvexAlignFast.cpp:606-774).  The SSE path the reference actually runs is
	 * identical to it only while opening a gap directly off the other gap type can
	 * never win: gap_open + gap_ext_min < mismatch.  Outside
	 * that regime we refuse rather than compute something the reference would not. */
	const bool sane = p->match > 0.0f && p->mismatch < 0.0f && p->gap_open < 0.0f &&
			p->gap_extend < 0.0f && p->gap_extend_min < 0.0f && p->gap_decay >= 0.0f &&
			p->gap_extend <= p->gap_extend_min &&
			(p->gap_open + p->gap_extend_min) < p->mismatch - 0.25f;
	if (!sane) {
		set_err("cvx_create: scoring (%g,%g,%g,%g,%g,%g) outside the supported regime "
				"(need match>0, penalties<0, gap_extend<=gap_extend_min, "
				"gap_open+gap_extend_min < mismatch-0.25)",
				p->match, p->mismatch, p->gap_open, p->gap_extend, p->gap_extend_min, p->gap_decay);
		return CVX_ERR_PARAMS;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		set_err("cvx_create: no HIP device available (the HIP path has no CPU fallback)");
		return CVX_ERR_NO_DEVICE;
	}
	if (device_id < 0 || device_id >= ndev) {
		set_err("cvx_create: device %d out of range (%d devices)", device_id, ndev);
		return CVX_ERR_NO_DEVICE;
	}
	HIP_TRY(hipSetDevice(device_id));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device_id));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		set_err("cvx_create: device %d is %s; this library carries gfx950 code only", device_id, prop.gcnArchName);
		return CVX_ERR_NO_DEVICE;
	}
	cvx_context *c = new (std::nothrow) cvx_context();
	if (!c) return CVX_ERR_OOM;
	c->device = device_id;
	c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	c->sp.mat = p->match; c->sp.mis = p->mismatch; c->sp.go = p->gap_open;
	c->sp.ge = p->gap_extend; c->sp.gem = p->gap_extend_min; c->sp.decay = p->gap_decay;
	c->max_matrix_mb = max_matrix_mb ? max_matrix_mb : 10000;
	if (const char *e = getenv("CVX_TUNE_MIN_M")) c->tune_min_slots = atoi(e);
	if (const char *e = getenv("CVX_TUNE_FORCE_WRAP16")) c->tune_force_wrap = atoi(e);
	if (const char *e = getenv("CVX_TUNE_LATE_MIN")) c->tune_late_min = std::max(1, atoi(e));
	hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	for (int i = 0; i < kAuxStreams && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
	if (e != hipSuccess) {
		set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
		delete c;
		return CVX_ERR_HIP;
	}
	*out = c;
	return CVX_OK;
}

void cvx_destroy(cvx_handle h) {
	if (!h) return;
	(void) hipSetDevice(h->device);
	if (h->stream) (void) hipStreamDestroy(h->stream);
	for (auto &a : h->aux) if (a) (void) hipStreamDestroy(a);
	for (cvx_batch_s *b : h->pool) { b->release(); delete b; }
	h->pool.clear();
	if (h->stage_seq) (void) hipHostFree(h->stage_seq);
	if (h->stage_rows) (void) hipHostFree(h->stage_rows);
	delete h;
}

int cvx_batch_upload(cvx_handle h, int32_t n, const cvx_tile *tiles, cvx_batch *out) {
	if (!h || !out || n < 0 || (n > 0 && !tiles)) { set_err("cvx_batch_upload: bad argument"); return CVX_ERR_ARG; }
	*out = nullptr;
	HIP_TRY(hipSetDevice(h->device));

	UploadLayout L;
	std::vector<TileIn> tin;
	int bad = -1;
	int lrc;
	try {
		lrc = upload_layout(n, tiles, tin, L, &bad);
	} catch (const std::bad_alloc &) {
		set_err("cvx_batch_upload: host allocation failed");
		return CVX_ERR_OOM;
	}
	if (lrc == kLayoutMalformed) { set_err("cvx_batch_upload: tile %d malformed", bad); return CVX_ERR_ARG; }
	if (lrc == kLayoutTooLarge) {
		set_err("cvx_batch_upload: %llu sequence bytes exceed one batch (4 GiB); split the batch",
				(unsigned long long) L.seq_total);
		return CVX_ERR_ARG;
	}
	const uint64_t seq_total = L.seq_total, n_rows = L.n_rows;
	const std::vector<uint64_t> &wprefix = L.wprefix;

	cvx_batch_s *b;
	if (!h->pool.empty()) {
		b = h->pool.back();
		h->pool.pop_back();
	} else {
		b = new (std::nothrow) cvx_batch_s();
	}
	if (!b) return CVX_ERR_OOM;
	b->n = n;
	b->ran = false;
	b->ops_total = 0;
	memset(&b->timing, 0, sizeof(b->timing));
	int rc = CVX_OK;
	try {
		b->tin.swap(tin);
		const size_t rows_bytes = (size_t) std::max<uint64_t>(n_rows, 1) * sizeof(RowDesc);
		std::vector<uint8_t> pg_seq;       /* pageable fallbacks when pinning fails */
		std::vector<RowDesc> pg_rows;
		uint8_t *hseq;
		RowDesc *hrows;
		if (ensure_pinned(&h->stage_seq, &h->stage_seq_cap, (size_t) seq_total) &&
				ensure_pinned(&h->stage_rows, &h->stage_rows_cap, rows_bytes)) {
			hseq = static_cast<uint8_t *>(h->stage_seq);
			hrows = static_cast<RowDesc *>(h->stage_rows);
		} else {
			pg_seq.resize((size_t) seq_total);
			pg_rows.resize(rows_bytes / sizeof(RowDesc));
			hseq = pg_seq.data();
			hrows = pg_rows.data();
		}
		upload_zero_pads(L, hseq);
		int threads = (int) std::thread::hardware_concurrency();
		threads = std::max(1, std::min(threads, 16));
		if (wprefix[(size_t) n] < (8u << 20)) threads = 1;      /* not worth a thread below ~8 MB */
		auto pack = [&](int begin, int end) { upload_pack(begin, end, tiles, b->tin, hseq, hrows); };
		if ((rc = b->d_seq.ensure((size_t) seq_total)) == CVX_OK &&
				(rc = b->d_rows.ensure(rows_bytes / sizeof(RowDesc))) == CVX_OK &&
				(rc = b->d_tin.ensure((size_t) std::max(n, 1))) == CVX_OK &&
				(rc = b->d_plan.ensure((size_t) std::max(n, 1))) == CVX_OK &&
				(rc = b->d_trun.ensure((size_t) std::max(n, 1))) == CVX_OK &&
				(rc = b->d_tout.ensure((size_t) std::max(n, 1))) == CVX_OK &&
				(rc = b->d_dstoff.ensure((size_t) std::max(n, 1))) == CVX_OK &&
				(rc = b->d_lists.ensure((size_t) std::max(n, 1))) == CVX_OK &&
				(rc = b->d_heads.ensure(64)) == CVX_OK) {
			/* pack and copy in a few pieces, so that the DMA of one piece runs under the packing
			 * of the next (pieces are contiguous in both arenas) */
			hipStream_t st = h->stream;
			hipError_t e = hipSuccess;
			const int pieces = threads > 1 ? 4 : 1;
			int t0 = 0;
			uint64_t seq_done = 0;                   /* bytes of hseq already on their way */
			for (int pc = 1; pc <= pieces && e == hipSuccess; ++pc) {
				int t1 = n;
				if (pc < pieces) {
					const uint64_t target = wprefix[(size_t) n] / (uint64_t) pieces * (uint64_t) pc;
					t1 = (int) (std::upper_bound(wprefix.begin(), wprefix.end(), target) - wprefix.begin());
					t1 = std::min(std::max(t1, t0), n);
				}
				if (t1 > t0) {
					std::vector<uint64_t> wp((size_t) (t1 - t0) + 1);
					for (int i = t0; i <= t1; ++i) wp[(size_t) (i - t0)] = wprefix[(size_t) i] - wprefix[(size_t) t0];
					parallel_ranges(t1 - t0, wp, threads, [&](int bg, int en) { pack(t0 + bg, t0 + en); });
				}
				const uint64_t seq_end = (t1 == n) ? seq_total : (uint64_t) b->tin[(size_t) t1].ref_off;
				if (seq_end > seq_done)
					e = hipMemcpyAsync(b->d_seq.p + seq_done, hseq + seq_done, (size_t) (seq_end - seq_done), hipMemcpyHostToDevice, st);
				seq_done = seq_end;
				const uint64_t r0 = (t0 < n) ? b->tin[(size_t) t0].row_off : n_rows;
				const uint64_t r1 = (t1 < n) ? b->tin[(size_t) t1].row_off : n_rows;
				if (e == hipSuccess && r1 > r0)
					e = hipMemcpyAsync(b->d_rows.p + r0, hrows + r0, (size_t) (r1 - r0) * sizeof(RowDesc), hipMemcpyHostToDevice, st);
				t0 = t1;
			}
			if (e == hipSuccess && n) e = hipMemcpyAsync(b->d_tin.p, b->tin.data(), (size_t) n * sizeof(TileIn), hipMemcpyHostToDevice, st);
			if (e == hipSuccess) e = hipStreamSynchronize(st);    /* the staging buffers are reused by the next upload */
			for (auto &ev : b->ev) if (e == hipSuccess && !ev) e = hipEventCreate(&ev);
			if (e != hipSuccess) { set_err("upload copy failed: %s", hipGetErrorString(e)); rc = CVX_ERR_HIP; }
		}
	} catch (const std::bad_alloc &) {
		set_err("cvx_batch_upload: host allocation failed");
		rc = CVX_ERR_OOM;
	}
	if (rc != CVX_OK) { b->release(); delete b; return rc; }
	*out = b;
	return CVX_OK;
}

static float ev_ms(hipEvent_t a, hipEvent_t b) {
	float ms = 0.0f;
	if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0f;
	return ms;
}

int cvx_batch_run(cvx_handle h, cvx_batch b) {
	if (!h || !b) { set_err("cvx_batch_run: NULL argument"); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	const int n = b->n;
	hipStream_t st = h->stream;
	b->ran = false;
	b->ops_total = 0;
	memset(&b->timing, 0, sizeof(b->timing));
	if (n == 0) { b->ran = true; return CVX_OK; }

	HIP_TRY(hipEventRecord(b->ev[0], st));
	HIP_TRY(launch_plan(b->d_rows.p, b->d_tin.p, b->d_plan.p, n, h->max_matrix_mb, st));
	b->plan.resize((size_t) n);
	HIP_TRY(hipMemcpyAsync(b->plan.data(), b->d_plan.p, (size_t) n * sizeof(TilePlan), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));

	/* ---- host planning: kernel class, arena offsets, work lists (cvx_host_logic.h) */
	HostPlan hp;
	host_plan(n, b->plan.data(), b->tin.data(), h->tune_min_slots, h->tune_force_wrap, hp);
	b->trun.swap(hp.trun);
	b->tout.swap(hp.tout);
	std::vector<std::vector<int32_t>> &cls = hp.cls;
	std::vector<int32_t> &generic = hp.generic;
	const uint64_t dir_dwords = hp.dir_dwords, ops_ints = hp.ops_ints, cells = hp.cells, active = hp.active;
	const int n_fast = hp.n_fast;
	int rc;
	if ((rc = b->d_dirs.ensure((size_t) dir_dwords + 64)) != CVX_OK) return rc;
	if ((rc = b->d_regions.ensure((size_t) ops_ints + 64)) != CVX_OK) return rc;

	b->lists.clear();
	std::vector<int> seg_begin(cls.size(), 0);
	for (size_t c = 0; c < cls.size(); ++c) {
		auto &v = cls[c];      /* already in LPT order */
		seg_begin[c] = (int) b->lists.size();
		b->lists.insert(b->lists.end(), v.begin(), v.end());
	}
	const int generic_begin = (int) b->lists.size();
	b->lists.insert(b->lists.end(), generic.begin(), generic.end());
	std::vector<uint64_t> goff(generic.size() + 1, 0);
	for (size_t g = 0; g < generic.size(); ++g)
		goff[g + 1] = goff[g] + (uint64_t) generic_scratch_bytes(b->trun[(size_t) generic[g]].ring);
	if (!generic.empty()) {
		if ((rc = b->d_gscratch.ensure((size_t) goff.back() + 256)) != CVX_OK) return rc;
		if ((rc = b->d_gscratch_off.ensure(goff.size())) != CVX_OK) return rc;
		HIP_TRY(hipMemcpyAsync(b->d_gscratch_off.p, goff.data(), goff.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
	}
	HIP_TRY(hipMemcpyAsync(b->d_trun.p, b->trun.data(), (size_t) n * sizeof(TileRun), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(b->d_tout.p, b->tout.data(), (size_t) n * sizeof(TileOut), hipMemcpyHostToDevice, st));
	if (!b->lists.empty())
		HIP_TRY(hipMemcpyAsync(b->d_lists.p, b->lists.data(), b->lists.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemsetAsync(b->d_heads.p, 0, 64 * sizeof(int32_t), st));
	HIP_TRY(hipEventRecord(b->ev[1], st));

	/* ---- forward fill: one launch per populated kernel class, classes run concurrently
	 * on separate streams (a sparsely populated class would otherwise serialise a whole
	 * tile latency behind the big one); widest rings first, they have the longest tiles */
	int launches = 0;
	b->launches.clear();
	for (int cc = (int) cls.size() - 1; cc >= 0; --cc) {
		const size_t c = (size_t) cc;
		if (cls[c].empty()) continue;
		const KernelClass &kc = kClasses[c / 2];
		while (b->lev.size() < (size_t) (launches + 1) * 3) {
			hipEvent_t e;
			HIP_TRY(hipEventCreate(&e));
			b->lev.push_back(e);
		}
		cvx_launch_info li;
		memset(&li, 0, sizeof(li));
		li.slots_per_lane = kc.m; li.waves = kc.nw; li.wrap16 = (int) (c & 1); li.n_tiles = (int) cls[c].size();
		for (int32_t ti : cls[c]) {
			const TilePlan &p = b->plan[(size_t) ti];
			const TileIn &in = b->tin[(size_t) ti];
			li.cells += p.cells; li.active_cells += p.active;
			li.alg_bytes += p.cells + 6ull * (uint64_t) in.H + 2ull * (uint64_t) in.W;
			li.read_bases += (uint64_t) in.H;
		}
		b->launches.push_back(li);
		hipStream_t ls = h->aux[launches % kAuxStreams];
		HIP_TRY(hipStreamWaitEvent(ls, b->ev[1], 0));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 3], ls));
		FillArgs a;
		a.seq = b->d_seq.p;
		a.rows = reinterpret_cast<const RowDesc2 *>(b->d_rows.p);
		a.tin = b->d_tin.p;
		a.trun = b->d_trun.p;
		a.tout = b->d_tout.p;
		a.dirs = b->d_dirs.p;
		a.list = b->d_lists.p + seg_begin[c];
		a.list_n = (int) cls[c].size();
		a.late_min_groups = h->tune_late_min;
		a.redo_count = b->d_heads.p;
		a.ops = b->d_regions.p;
		a.sp = h->sp;
		HIP_TRY(launch_fill(kc.m, kc.nw, (c & 1) != 0, false, a, ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 3 + 1], ls));
		/* exact-tracking pass over the tiles the two-phase pass flagged (usually none) */
		HIP_TRY(launch_fill(kc.m, kc.nw, (c & 1) != 0, true, a, ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 3 + 2], ls));
		launches++;
	}
	if (!generic.empty()) {
		while (b->lev.size() < (size_t) (launches + 1) * 3) {
			hipEvent_t e;
			HIP_TRY(hipEventCreate(&e));
			b->lev.push_back(e);
		}
		cvx_launch_info li;
		memset(&li, 0, sizeof(li));
		li.slots_per_lane = 0; li.waves = 16; li.wrap16 = 1; li.n_tiles = (int) generic.size();
		for (int32_t ti : generic) {
			const TilePlan &p = b->plan[(size_t) ti];
			const TileIn &in = b->tin[(size_t) ti];
			li.cells += p.cells; li.active_cells += p.active;
			li.alg_bytes += p.cells + 6ull * (uint64_t) in.H + 2ull * (uint64_t) in.W;
			li.read_bases += (uint64_t) in.H;
		}
		b->launches.push_back(li);
		hipStream_t ls = h->aux[launches % kAuxStreams];
		HIP_TRY(hipStreamWaitEvent(ls, b->ev[1], 0));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 3], ls));
		FillArgs a;
		a.seq = b->d_seq.p;
		a.rows = reinterpret_cast<const RowDesc2 *>(b->d_rows.p);
		a.tin = b->d_tin.p;
		a.trun = b->d_trun.p;
		a.tout = b->d_tout.p;
		a.dirs = b->d_dirs.p;
		a.list = b->d_lists.p + generic_begin;
		a.list_n = (int) generic.size();
		a.late_min_groups = h->tune_late_min;
		a.redo_count = b->d_heads.p;
		a.ops = b->d_regions.p;
		a.sp = h->sp;
		HIP_TRY(launch_fill_generic(a, b->d_gscratch.p, b->d_gscratch_off.p, ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 3 + 1], ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 3 + 2], ls));
		launches++;
	}
	for (int i = 0; i < launches; ++i) HIP_TRY(hipStreamWaitEvent(st, b->lev[(size_t) i * 3 + 2], 0));
	HIP_TRY(hipEventRecord(b->ev[2], st));

	/* ---- backtrack + ops compaction */
	BacktrackArgs ba;
	ba.seq = b->d_seq.p;
	ba.rows = reinterpret_cast<const RowDesc2 *>(b->d_rows.p);
	ba.tin = b->d_tin.p;
	ba.trun = b->d_trun.p;
	ba.tout = b->d_tout.p;
	ba.dirs = b->d_dirs.p;
	ba.ops = b->d_regions.p;
	ba.n_tiles = n;
	HIP_TRY(launch_backtrack(ba, st));
	HIP_TRY(hipMemcpyAsync(b->tout.data(), b->d_tout.p, (size_t) n * sizeof(TileOut), hipMemcpyDeviceToHost, st));
	int32_t redone = 0;
	HIP_TRY(hipMemcpyAsync(&redone, b->d_heads.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	b->dst_off.assign((size_t) n, 0);
	uint64_t total = 0;
	for (int i = 0; i < n; ++i) {
		b->dst_off[(size_t) i] = total;
		if (b->tout[(size_t) i].status == 0) total += (uint64_t) b->tout[(size_t) i].n_ops;
	}
	b->ops_total = total;
	if ((rc = b->d_dense.ensure((size_t) total + 64)) != CVX_OK) return rc;
	HIP_TRY(hipMemcpyAsync(b->d_dstoff.p, b->dst_off.data(), (size_t) n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
	HIP_TRY(launch_compact(b->d_regions.p, b->d_trun.p, b->d_tout.p, b->d_dstoff.p, b->d_dense.p, n, st));
	HIP_TRY(hipEventRecord(b->ev[3], st));
	HIP_TRY(hipStreamSynchronize(st));

	for (int i = 0; i < launches; ++i) b->launches[(size_t) i].ms = ev_ms(b->lev[(size_t) i * 3], b->lev[(size_t) i * 3 + 1]);
	b->timing.plan_ms = ev_ms(b->ev[0], b->ev[1]);
	b->timing.fill_ms = ev_ms(b->ev[1], b->ev[2]);
	b->timing.backtrack_ms = ev_ms(b->ev[2], b->ev[3]);
	b->timing.total_ms = ev_ms(b->ev[0], b->ev[3]);
	b->timing.cells = cells;
	b->timing.active_cells = active;
	b->timing.dir_bytes = dir_dwords * 4;
	b->timing.n_fill_launches = launches;
	b->timing.n_tiles_fast = n_fast;
	b->timing.n_tiles_redone = redone;
	b->ran = true;
	return CVX_OK;
}

int cvx_batch_timing(cvx_batch b, cvx_timing *t) {
	if (!b || !t) { set_err("cvx_batch_timing: NULL argument"); return CVX_ERR_ARG; }
	*t = b->timing;
	return CVX_OK;
}

int cvx_batch_launch_info(cvx_batch b, int32_t i, cvx_launch_info *info) {
	if (!b || !info || !b->ran || i < 0 || (size_t) i >= b->launches.size()) { set_err("cvx_batch_launch_info: bad index"); return CVX_ERR_ARG; }
	*info = b->launches[(size_t) i];
	return CVX_OK;
}

int cvx_batch_ops_total(cvx_batch b, uint64_t *n_ops) {
	if (!b || !n_ops || !b->ran) { set_err("cvx_batch_ops_total: batch not run"); return CVX_ERR_ARG; }
	*n_ops = b->ops_total;
	return CVX_OK;
}

int cvx_batch_download(cvx_handle h, cvx_batch b, cvx_result *results, uint32_t *ops_arena,
		uint64_t ops_capacity, uint64_t *ops_used) {
	if (!h || !b || !b->ran || (b->n > 0 && !results)) { set_err("cvx_batch_download: bad argument / batch not run"); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	if (ops_used) *ops_used = b->ops_total;
	for (int i = 0; i < b->n; ++i) {
		const TileOut &o = b->tout[(size_t) i];
		cvx_result &r = results[i];
		memset(&r, 0, sizeof(r));
		r.score = o.score;
		r.status = o.status;
		r.best_ref_index = o.best_x;
		r.best_read_index = o.best_y;
		r.ref_position = o.ref_position;
		r.qstart = o.qstart;
		r.qend = o.qend;
		r.n_ops = o.status == 0 ? o.n_ops : 0;
		r.ops_begin = b->dst_off[(size_t) i];
		r.cells = b->plan[(size_t) i].cells;
	}
	if (b->ops_total > ops_capacity) {
		set_err("cvx_batch_download: ops arena too small (%llu needed, %llu given)",
				(unsigned long long) b->ops_total, (unsigned long long) ops_capacity);
		return CVX_ERR_CAPACITY;
	}
	if (b->ops_total) {
		if (!ops_arena) { set_err("cvx_batch_download: NULL ops arena"); return CVX_ERR_ARG; }
		HIP_TRY(hipMemcpy(ops_arena, b->d_dense.p, (size_t) b->ops_total * sizeof(uint32_t), hipMemcpyDeviceToHost));
	}
	return CVX_OK;
}

void cvx_batch_free(cvx_handle h, cvx_batch b) {
	if (!b) return;
	if (h) (void) hipSetDevice(h->device);
	if (h && h->pool.size() < kPoolBatches) {
		b->ran = false;
		h->pool.push_back(b);      /* arenas and events stay allocated for the next upload */
		return;
	}
	b->release();
	delete b;
}

int cvx_score_batch(cvx_handle h, int32_t n, const char *const *refs, const char *const *qrys, float *scores) {
	if (!h || n < 0 || (n > 0 && (!refs || !qrys || !scores))) { set_err("cvx_score_batch: bad argument"); return CVX_ERR_ARG; }
	if (n == 0) return CVX_OK;
	HIP_TRY(hipSetDevice(h->device));
	std::vector<ScorePair> pairs((size_t) n);
	uint64_t bytes = 0, rows = 0;
	for (int i = 0; i < n; ++i) {
		if (!refs[i] || !qrys[i]) { set_err("cvx_score_batch: NULL sequence %d", i); return CVX_ERR_ARG; }
		const size_t rl = strlen(refs[i]) + 1, ql = strlen(qrys[i]) + 1;
		ScorePair &p = pairs[(size_t) i];
		p.ref_off = bytes; bytes += rl;
		p.qry_off = bytes; bytes += ql;
		p.ref_len = (int32_t) std::min<size_t>(rl, 0x7fffffff);
		p.qry_len = (int32_t) std::min<size_t>(ql, 0x7fffffff);
		p.scratch_off = rows;
		if (rl < 100000 && ql < 100000) rows += 2 * (uint64_t) rl;
	}
	std::vector<uint8_t> hseq((size_t) bytes + 16);
	for (int i = 0; i < n; ++i) {
		memcpy(&hseq[(size_t) pairs[(size_t) i].ref_off], refs[i], (size_t) pairs[(size_t) i].ref_len);
		memcpy(&hseq[(size_t) pairs[(size_t) i].qry_off], qrys[i], (size_t) pairs[(size_t) i].qry_len);
	}
	DevBuf<uint8_t> d_seq; DevBuf<ScorePair> d_pairs; DevBuf<int32_t> d_rows; DevBuf<float> d_out;
	int rc;
	if ((rc = d_seq.ensure(hseq.size())) != CVX_OK || (rc = d_pairs.ensure((size_t) n)) != CVX_OK ||
			(rc = d_rows.ensure((size_t) rows + 64)) != CVX_OK || (rc = d_out.ensure((size_t) n)) != CVX_OK) {
		d_seq.release(); d_pairs.release(); d_rows.release(); d_out.release();
		return rc;
	}
	hipError_t e = hipMemcpyAsync(d_seq.p, hseq.data(), hseq.size(), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_pairs.p, pairs.data(), (size_t) n * sizeof(ScorePair), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = launch_score(d_seq.p, d_pairs.p, d_rows.p, d_out.p, n, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(scores, d_out.p, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	d_seq.release(); d_pairs.release(); d_rows.release(); d_out.release();
	if (e != hipSuccess) { set_err("cvx_score_batch: %s", hipGetErrorString(e)); return CVX_ERR_HIP; }
	return CVX_OK;
}

int cvx_align_batch(cvx_handle h, int32_t n, const cvx_tile *tiles, cvx_result *results,
		uint32_t *ops_arena, uint64_t ops_capacity, uint64_t *ops_used) {
	cvx_batch b = nullptr;
	int rc = cvx_batch_upload(h, n, tiles, &b);
	if (rc != CVX_OK) return rc;
	rc = cvx_batch_run(h, b);
	if (rc == CVX_OK) rc = cvx_batch_download(h, b, results, ops_arena, ops_capacity, ops_used);
	cvx_batch_free(h, b);
	return rc;
}

}  /* extern "C" */
