/*
 * cvx_runtime.cpp -- host side of libcvxalign.so: the C ABI of include/cvx_align.h
 * over the gfx950 kernels of cvx_kernels.hip / cvx_generic.hip / cvx_score.hip.
 *
 * One cvx_context = one device (ngmlr creates one aligner per worker thread, reference
 * src/CS.cpp:418; contexts are independent, a context is not re-entrant -- same contract as
 * the reference's ConvexAlignFast instances).  A batch of tiles moves through four stages:
 *
 *   upload   host threads pack sequences + corridor rows into the batch's own pinned staging,
 *            piece by piece, each piece's DMA running under the packing of the next  (stream `io`)
 *   plan     plan_kernel, plan records back to pinned memory                          (stream `io`)
 *   compute  host: kernel class / arena offsets / LPT lists from the plan records;
 *            fill_ring_kernel per class (+ exact redo pass)                   (streams `main` + `aux`)
 *            backtrack_kernel, finalize_kernel (device-side prefix sums and result records),
 *            compact_ops_kernel, result records back to pinned memory               (stream `main`;
 *            optionally `post`, beside the next batch's fills: measured to gain nothing, see stage_compute)
 *   finish   dense ops back to pinned memory                                          (stream `io`)
 *
 * The streaming entry points (cvx_submit / cvx_wait / cvx_job_release) keep several batches in
 * flight on one handle: the upload and plan of batch k+1 and the download of batch k-1 run
 * under the kernels of batch k, and the only host waits are on events of work that was queued a
 * whole batch earlier.  The staged entry points (cvx_batch_*) and cvx_align_batch run the same
 * stages back to back.
 *
 * There is no CPU compute path in this file: without a device every entry point returns
 * CVX_ERR_NO_DEVICE.
 */
#include <hip/hip_runtime_api.h>
#include <dirent.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cvx_align.h"
#include "cvx_host_logic.h"
#include "cvx_index_build.h"
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
#define RC_TRY(expr) do { int rc_ = (expr); if (rc_ != CVX_OK) return rc_; } while (0)

/* no C++ exception may cross the C ABI (std::bad_alloc from the host-side vectors) */
#define ABI_GUARD_BEGIN try {
#define ABI_GUARD_END                                                              \
	} catch (const std::bad_alloc &) {                                             \
		set_err("host allocation failed");                                         \
		return CVX_ERR_OOM;                                                        \
	} catch (...) {                                                                \
		set_err("unexpected C++ exception");                                       \
		return CVX_ERR_HIP;                                                        \
	}

/* Buffers that grow while a handle streams jobs of changing size (ngmlr's pipeline: launches of 50 to 4 000 tiles on five job
 * slots).  Two things made cvx_submit take 15-50 ms per call there (profiles/r06_e2e_submit_trace.txt, `upload stage`):
 *   - hipFree / hipHostFree wait for everything in flight on the device.  An outgrown block is therefore not freed on the spot
 *     but parked (defer_release) and given back when its handle has nothing in flight, or at cvx_destroy;
 *   - every slot climbed to the size of the largest launch by itself.  Slots of one handle share a high-water mark per buffer
 *     (bind): the first growth of a slot goes straight to the largest capacity any of them has had. */
struct DeferredFrees {
	std::mutex mtx;
	std::vector<void *> dev, host;
	void drain() {
		std::vector<void *> d, h;
		{
			std::lock_guard<std::mutex> lk(mtx);
			d.swap(dev);
			h.swap(host);
		}
		for (void *p : d) (void) hipFree(p);
		for (void *p : h) (void) hipHostFree(p);
	}
	bool empty() { std::lock_guard<std::mutex> lk(mtx); return dev.empty() && host.empty(); }
};
DeferredFrees g_deferred;

/* the streams the service handles of a process share, per device (cvx_create_ex; never destroyed: they die with the process) */
static const int kServiceStreamsMax = 16, kMaxServiceDevices = 64;
struct ServiceStreams { hipStream_t st[kServiceStreamsMax] = {nullptr}; unsigned next = 0; };
ServiceStreams g_service_streams[kMaxServiceDevices];
std::mutex g_service_mtx;

template <typename T>
struct DevBuf {
	T *p = nullptr;
	size_t cap = 0; /* elements */
	size_t *hwm = nullptr;      /* shared by the same buffer of every job slot of the handle (bytes) */
	int ensure(size_t n) {
		if (n <= cap) return CVX_OK;
		const size_t old = cap;
		if (p) {
			std::lock_guard<std::mutex> lk(g_deferred.mtx);
			g_deferred.dev.push_back(p);
			p = nullptr; cap = 0;
		}
		const size_t asked = n + n / 8 + 64;
		size_t want = asked;
		if (want < 2 * old) want = 2 * old;      /* a buffer that has to grow at least doubles; the first allocation stays close to what was asked for */
		if (hwm && want * sizeof(T) < *hwm && *hwm / 16 <= asked * sizeof(T)) want = (*hwm + sizeof(T) - 1) / sizeof(T);      /* (never more than 16 x what was asked for) */
		hipError_t e = hipMalloc((void **) &p, want * sizeof(T));
		if (e != hipSuccess && want > asked) {      /* the generous size does not fit: give back what is parked, then what was asked for may */
			(void) hipGetLastError();
			g_deferred.drain();
			want = asked;
			e = hipMalloc((void **) &p, want * sizeof(T));
		}
		if (e != hipSuccess) {
			set_err("hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
			p = nullptr;
			return CVX_ERR_OOM;
		}
		cap = want;
		if (hwm && cap * sizeof(T) > *hwm) *hwm = cap * sizeof(T);
		return CVX_OK;
	}
	void release() {
		if (p) (void) hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

/* grow-only page-locked host buffer */
struct PinBuf {
	void *p = nullptr;
	size_t cap = 0; /* bytes */
	size_t *hwm = nullptr;
	int ensure(size_t bytes) {
		if (bytes <= cap) return CVX_OK;
		const size_t old = cap;
		if (p) {
			std::lock_guard<std::mutex> lk(g_deferred.mtx);
			g_deferred.host.push_back(p);
			p = nullptr; cap = 0;
		}
		const size_t asked = bytes + bytes / 8 + 4096;
		size_t want = asked;
		if (want < 2 * old) want = 2 * old;                     /* (see DevBuf::ensure) */
		if (hwm && want < *hwm && *hwm / 16 <= asked) want = *hwm;
		hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
		if (e != hipSuccess && want > asked) {
			(void) hipGetLastError();
			g_deferred.drain();
			want = asked;
			e = hipHostMalloc(&p, want, hipHostMallocDefault);
		}
		if (e != hipSuccess) {
			(void) hipGetLastError();
			set_err("hipHostMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
			p = nullptr;
			return CVX_ERR_OOM;
		}
		cap = want;
		if (hwm && cap > *hwm) *hwm = cap;
		return CVX_OK;
	}
	void release() {
		if (p) (void) hipHostFree(p);
		p = nullptr;
		cap = 0;
	}
	template <typename T> T *as() const { return static_cast<T *>(p); }
};

/* page-locked blocks handed out by cvx_host_alloc: sequences found inside one travel without packing */
std::mutex g_pin_mtx;
std::vector<std::pair<const char *, size_t>> g_pin_blocks;

bool in_pinned_block(const void *p, uint64_t bytes) {
	const char *c = static_cast<const char *>(p);
	std::lock_guard<std::mutex> lk(g_pin_mtx);
	for (const auto &b : g_pin_blocks)
		if (c >= b.first && c + bytes <= b.first + b.second) return true;
	return false;
}

}  // namespace

/* Four streams per handle (io, main, post, one aux) and not more: the ROCm runtime multiplexes the streams of a process onto
 * (by default) four hardware queues, and two streams that land on the same queue serialise --
 * measured: with seven streams the 18-tile M = 4 launch ran alone for 11 ms in front of the
 * 24 558-tile M = 3 launch instead of beside it. */
static const int kAuxStreams = 2;      /* (round 5: a batch has up to four fill classes -- chained, gangs, M = 4, M = 3 -- and each wants a stream of its own) */
static const size_t kPoolBatches = 8;      /* (4 until the text stage of a finished job got its own thread: a launch in the fill, one uploading, two in the text stage and those the workers still copy from) */

enum BatchState { kFailed = -1, kEmpty = 0, kUploaded = 1, kPlanned = 2, kComputed = 3, kFinished = 4 };

struct cvx_batch_s {
	int n = 0;
	int state = kEmpty;
	bool in_flight = false;          /* submitted through the streaming API and not yet released */
	hipStream_t s_run = nullptr;     /* the `main` stream of the set that carried this batch's kernels */
	int fail_rc = CVX_OK;            /* state == kFailed: what went wrong with THIS job (cvx_wait returns it) */
	std::string fail_msg;
	uint64_t seq_total = 0, n_rows = 0, n_rowsx = 0;
	uint64_t zero_copy_bytes = 0;    /* sequence bytes of the last upload that travelled straight from the caller's page-locked arena */
	uint64_t ops_total = 0;          /* valid after the compute stage has been waited for */
	uint64_t dense_cap = 0;
	bool have_ops = false;           /* dense ops are in h_ops */

	/* host side, page-locked */
	PinBuf h_zero;                   /* zeros for the pads around blocks that travel straight from the caller's arena */
	size_t zero_cap = 0;
	PinBuf h_seq, h_delta, h_rsrc, h_rowsx, h_tin;     /* upload staging (owned by the batch: no wait before reuse by another batch):
	                                                    * sequences, one step byte per corridor row, RowSrc[n], rows that need the verbatim form */
	std::vector<RowDesc> chain_rows;                   /* scratch: expanded rows of a tile being chained */
	PinBuf h_plan;                   /* TilePlan[n] */
	PinBuf h_trun, h_tout, h_lists, h_goff;   /* what the host planning produces */
	PinBuf h_res;                    /* ResultRec[n] + BatchSummary */
	PinBuf h_ops;                    /* dense ops */

	DevBuf<uint8_t> d_seq;
	DevBuf<RowDesc> d_rows;          /* the expanded rows arena (expand_rows_kernel) */
	DevBuf<uint8_t> d_delta;
	DevBuf<RowSrc> d_rsrc;
	DevBuf<RowDesc> d_rowsx;
	DevBuf<TileIn> d_tin;
	DevBuf<TilePlan> d_plan;
	DevBuf<TileRun> d_trun;
	DevBuf<TileOut> d_tout;
	DevBuf<uint32_t> d_dirs;
	DevBuf<int32_t> d_regions;
	DevBuf<int32_t> d_lists;
	DevBuf<int32_t> d_counters;
	DevBuf<uint64_t> d_dstoff;
	DevBuf<uint32_t> d_dense;
	DevBuf<uint8_t> d_res;           /* ResultRec[n] + BatchSummary */
	DevBuf<uint8_t> d_gscratch;      /* slot state of tiles taken by the catch-all kernel */
	DevBuf<uint64_t> d_gscratch_off;
	/* chained tiles (row blocks) */
	/* device-side text stage (cvx_job_text) */
	PinBuf h_ext, h_trec, h_toff, h_text;
	DevBuf<int32_t> d_ext;
	DevBuf<TextRec> d_trec;
	DevBuf<unsigned long long> d_tlen;   /* lengths, offsets, total */
	DevBuf<uint8_t> d_text;
	bool text_done = false;          /* d_trec holds this job's text records (cvx_job_text ran) */
	DevBuf<unsigned long long> d_nmoff;  /* nmPerPosition: entry counts, offsets, total of the requested tile range */
	DevBuf<int32_t> d_nm;            /* ... and the triples */
	PinBuf h_nmoff;
	PinBuf h_nm;                     /* the triples on the host (cvx_job_nm_profile_resident) */
	hipEvent_t ev_nm0 = nullptr, ev_nm1 = nullptr;
	PinBuf h_win;                    /* WindowDesc[n]: reference windows decoded on the device (cvx_submit_windows) */
	PinBuf h_refs;                   /* ... and the decoded windows back on the host (cvx_job_window_refs): the reference part of d_seq */
	uint64_t refs_base = 0;          /* offset of that part in the sequence arena */
	bool have_refs = false;
	DevBuf<WindowDesc> d_win;
	PinBuf h_chain;                  /* ChainTask[] of all chain classes, ChainBlk[], tile lists */
	DevBuf<uint8_t> d_chain;
	DevBuf<BoundaryRec> d_bnd;
	uint32_t bnd_epoch = 0;          /* tag of the records the latest launch wrote (0: buffer freshly zeroed) */
	DevBuf<ChainOut> d_chain_out;

	hipEvent_t ev_bt0 = nullptr, ev_bt1 = nullptr;   /* fork / join of the long-read backtrack launch */
	hipEvent_t ev_in = nullptr;      /* upload + plan records on the host */
	hipEvent_t ev_res = nullptr;     /* result records on the host */
	hipEvent_t ev_ops = nullptr;     /* dense ops on the host */
	hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   /* timing: plan begin/end, fills done, all done, fills may start */
	std::vector<hipEvent_t> lev;     /* 4 events per fill class: start, two-phase pass done, exact pass done, the class's own backtrack done */
	std::vector<cvx_launch_info> launches;
	cvx_timing timing;

	const TileIn *tin() const { return h_tin.as<TileIn>(); }
	const TilePlan *plan() const { return h_plan.as<TilePlan>(); }
	const ResultRec *res() const { return h_res.as<ResultRec>(); }
	const BatchSummary *summary() const { return reinterpret_cast<const BatchSummary *>(h_res.as<uint8_t>() + (size_t) n * sizeof(ResultRec)); }

	/* the buffers whose size follows the job's tile count or cells: the same buffer of every slot of a handle shares one mark */
	static const int kSharedMarks = 32;
	void bind(size_t *marks) {
		int k = 0;
		PinBuf *pins[] = { &h_seq, &h_delta, &h_rsrc, &h_tin, &h_plan, &h_trun, &h_tout, &h_lists, &h_res, &h_ops, &h_chain };
		for (PinBuf *b_ : pins) b_->hwm = &marks[k++];
		d_seq.hwm = &marks[k++]; d_rows.hwm = &marks[k++]; d_delta.hwm = &marks[k++]; d_rsrc.hwm = &marks[k++]; d_tin.hwm = &marks[k++];
		d_plan.hwm = &marks[k++]; d_trun.hwm = &marks[k++]; d_tout.hwm = &marks[k++]; d_dirs.hwm = &marks[k++]; d_regions.hwm = &marks[k++];
		d_lists.hwm = &marks[k++]; d_dstoff.hwm = &marks[k++]; d_dense.hwm = &marks[k++]; d_res.hwm = &marks[k++];
		d_chain.hwm = &marks[k++]; d_bnd.hwm = &marks[k++]; d_chain_out.hwm = &marks[k++];
		/* the big buffers of the text stage (strings, profile triples: 125 kB per 10 kb alignment); written by the one thread that
		 * runs a handle's text stages -- never the thread that submits, and these entries are nobody else's */
		h_text.hwm = &marks[k++]; h_nm.hwm = &marks[k++]; d_text.hwm = &marks[k++]; d_nm.hwm = &marks[k++];
		static_assert(11 + 17 + 4 <= kSharedMarks, "marks");
	}
	int make_events() {
		if (!ev_in) HIP_TRY(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
		if (!ev_res) HIP_TRY(hipEventCreateWithFlags(&ev_res, hipEventDisableTiming));
		if (!ev_ops) HIP_TRY(hipEventCreateWithFlags(&ev_ops, hipEventDisableTiming));
		if (!ev_bt0) HIP_TRY(hipEventCreateWithFlags(&ev_bt0, hipEventDisableTiming));
		if (!ev_bt1) HIP_TRY(hipEventCreateWithFlags(&ev_bt1, hipEventDisableTiming));
		for (auto &e : ev) if (!e) HIP_TRY(hipEventCreate(&e));
		return CVX_OK;
	}
	void release() {
		h_zero.release(); zero_cap = 0;
		h_seq.release(); h_delta.release(); h_rsrc.release(); h_rowsx.release(); h_tin.release();
		d_delta.release(); d_rsrc.release(); d_rowsx.release(); h_plan.release(); h_trun.release(); h_tout.release();
		h_lists.release(); h_goff.release(); h_res.release(); h_ops.release();
		d_seq.release(); d_rows.release(); d_tin.release(); d_plan.release(); d_trun.release();
		d_tout.release(); d_dirs.release(); d_regions.release(); d_lists.release();
		d_counters.release(); d_dstoff.release(); d_dense.release(); d_res.release();
		d_gscratch.release(); d_gscratch_off.release();
		h_win.release(); d_win.release(); h_refs.release();
		h_ext.release(); h_trec.release(); h_toff.release(); h_text.release();
		d_ext.release(); d_trec.release(); d_tlen.release(); d_text.release();
		d_nmoff.release(); d_nm.release(); h_nmoff.release(); h_nm.release();
		if (ev_nm0) { (void) hipEventDestroy(ev_nm0); ev_nm0 = nullptr; }
		if (ev_nm1) { (void) hipEventDestroy(ev_nm1); ev_nm1 = nullptr; }
		h_chain.release(); d_chain.release(); d_bnd.release(); d_chain_out.release();
		if (ev_in) { (void) hipEventDestroy(ev_in); ev_in = nullptr; }
		if (ev_res) { (void) hipEventDestroy(ev_res); ev_res = nullptr; }
		if (ev_ops) { (void) hipEventDestroy(ev_ops); ev_ops = nullptr; }
		if (ev_bt0) { (void) hipEventDestroy(ev_bt0); ev_bt0 = nullptr; }
		if (ev_bt1) { (void) hipEventDestroy(ev_bt1); ev_bt1 = nullptr; }
		for (auto &e : ev) if (e) { (void) hipEventDestroy(e); e = nullptr; }
		for (auto &e : lev) if (e) (void) hipEventDestroy(e);
		lev.clear();
	}
};

/* Per-handle state of the candidate search: staging and device buffers live as long as the handle, so that the steady
 * state of a caller that searches batch after batch (ngmlr's CS threads: a few hundred sub-reads per call) allocates
 * nothing, copies through page-locked memory and sleeps on a blocking event while the device works. */
struct cvx_search_state {
	PinBuf h_seq, h_meta, h_out;       /* reads; offsets + lengths + list offsets + work list + dense begins; results coming back */
	DevBuf<uint8_t> d_seq;
	DevBuf<uint64_t> d_off, d_listoff, d_begin, d_srcoff;
	PinBuf h_srcoff;
	DevBuf<int32_t> d_len, d_ncand, d_work, d_miss;
	DevBuf<unsigned long long> d_events;
	DevBuf<float> d_maxhit, d_scores;
	DevBuf<uint32_t> d_rlist, d_undo;
	DevBuf<unsigned int> d_ticket;
	size_t tables_clean_words = 0;     /* this much of d_keys is known to hold nothing but empty slots (search_wave_hbm_kernel's tables) */
	DevBuf<SearchCandidate> d_cand, d_dense;
	DevBuf<uint64_t> d_keys;
	hipEvent_t done = nullptr;
	/* kernel time of the last call (cvx_stage_kernel_ms): an event pair around every kernel launch of the call, summed when it ends */
	std::vector<int32_t> attempts;     /* per read of the last call: table sizes tried (cvx_search_last_attempts) */
	std::vector<hipEvent_t> kev;
	size_t kev_used = 0;
	float kernel_ms = 0.0f;
	int kmark(hipStream_t st) {
		if (kev_used == kev.size()) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); kev.push_back(e); }
		HIP_TRY(hipEventRecord(kev[kev_used++], st));
		return CVX_OK;
	}
	void release() {
		for (hipEvent_t e : kev) (void) hipEventDestroy(e);
		kev.clear(); kev_used = 0;
		h_seq.release(); h_meta.release(); h_out.release();
		d_seq.release(); d_off.release(); d_listoff.release(); d_begin.release(); d_srcoff.release(); h_srcoff.release(); d_len.release(); d_ncand.release(); d_work.release(); d_miss.release();
		d_events.release(); d_maxhit.release(); d_scores.release(); d_rlist.release(); d_undo.release(); d_ticket.release(); d_cand.release(); d_dense.release(); d_keys.release();
		if (done) { (void) hipEventDestroy(done); done = nullptr; }
	}
};

namespace {
/* everything queued on `st` so far is done; the calling thread sleeps meanwhile (hipStreamSynchronize spins) */
int search_wait(cvx_search_state *ss, hipStream_t st) {
	if (!ss->done) HIP_TRY(hipEventCreateWithFlags(&ss->done, hipEventBlockingSync | hipEventDisableTiming));
	HIP_TRY(hipEventRecord(ss->done, st));
	HIP_TRY(hipEventSynchronize(ss->done));
	return CVX_OK;
}
}  // namespace

void cvx_search_state_free(cvx_search_state *ss) {
	if (!ss) return;
	ss->release();
	delete ss;
}

struct cvx_context {
	int device = 0;
	hipStream_t s_io = nullptr;      /* uploads, plan, downloads (high priority: its short kernels and copies
	                                  * must not queue behind the fill's workgroups) */
	hipStream_t s_main = nullptr;    /* forward fills (and the small input copies in front of them) */
	hipStream_t s_text = nullptr;    /* the text stage of FINISHED jobs (cvx_job_text, nm profile): its own stream, never queued behind the
	                                  * fills of the batches that follow */
	hipStream_t s_post = nullptr;    /* backtrack, finalize, compaction, result download: runs beside the NEXT batch's fill */
	hipStream_t aux[kAuxStreams] = {nullptr};  /* concurrent fill classes */
	bool service = false;                      /* cvx_create_ex(CVX_CREATE_SERVICE): short search / scoring / decode calls only */
	bool owns_main = true;                     /* false: s_main is one of the process's shared service streams */
	std::mutex streams_mtx;                    /* ensure_streams */
	/* a second set (main, post, aux) for small streaming jobs (experiment, off by default: see single_lane): the jobs of a
	 * batching driver (tens of tiles each) are latency-bound chains of small kernels on an otherwise empty device;
	 * consecutive ones can alternate between the two sets */
	hipStream_t s_main2 = nullptr, s_post2 = nullptr, aux2[kAuxStreams] = {nullptr};
	unsigned small_jobs = 0;         /* small streaming jobs queued so far (their parity picks the set) */
	ScoreParams sp;
	uint64_t max_matrix_mb = 10000;
	int num_cus = 256;
	int pack_threads = 16;
	int tune_min_slots = 0;   /* tuning knob (env CVX_TUNE_MIN_M): smallest M*NW a tile may use */
	int tune_late_min = kLateMinGroups;  /* test knob (env CVX_TUNE_LATE_MIN): groups of the exactly tracked tail (huge: exact everywhere) */
	int tune_late_shift = kLateShift;    /* tuning knob (env CVX_TUNE_LATE_SHIFT): the exactly tracked tail is groups >> this (at least tune_late_min) */
	int tune_max_slots = 0;   /* tuning knob (env CVX_TUNE_MAX_M): largest whole-tile ring class; wider tiles are chained */
	int tune_long_need = 0;   /* tuning knob (env CVX_TUNE_LONG_NEED): see PlanTuning */
	int tune_gang_prio = 0;   /* tuning knob (env CVX_TUNE_GANG_PRIO = 0 / 1): gang tiles at raised wave priority (measured: 26.9 against 24.0 ms, a loss) */
	int tune_gangs = 0;       /* tuning knob (env CVX_TUNE_GANGS = 0 / 1): rings of 384 / 576 slots as gangs of two / three waves instead of chained row
	                           * blocks.  Off: measured, the gangs lose -- the ONT mix's retry tiles alone 24.0 ms against 21.4 ms chained, the whole mix
	                           * equal within noise (profiles/r05_gang_ab.txt): short tiles leave a 576-slot ring idle through its ramps (60 % slot use
	                           * against 85 % for 64-row blocks), which eats the 1.6 x cheaper cell update, and three waves in lock step on three SIMDs
	                           * wait for the slowest of them every step */
	int tune_long_steps = 0, tune_small_batch = 0;   /* tuning knobs (env CVX_TUNE_LONG_STEPS / CVX_TUNE_SMALL_BATCH): see PlanTuning */
	bool single_lane = true;  /* experiment (env CVX_TUNE_TWO_LANES=1 clears it): small streaming jobs alternate between two stream sets.
	                           * Measured with the batching dispatcher at four launches in flight: no gain -- 20 000 reads 25.1 s against
	                           * 21.2 s, launches get smaller and each still lasts as long as its slowest tile -- so it stays off. */
	int tune_exact_steps = kExactDirectSteps;   /* tuning knob (env CVX_TUNE_EXACT_STEPS, 0 = off): tiles of this many steps go straight to the exact fill */
	int tune_wide_prio = 1;   /* tuning knob (env CVX_TUNE_WIDE_PRIO = 0 / 1 / 2: off, priority 1, priority 2): the widest ring class of a batch of several one priority notch up */
	int tune_chain_prio = -1; /* tuning knob (env CVX_TUNE_CHAIN_PRIO = 0 / 1): wave priority of chained blocks; -1 = the default (raised) */
	int tune_chain_lds_kb = 48; /* tuning knob (env CVX_TUNE_CHAIN_LDS_KB): LDS per CU the residency cap of a chained class may hold while ring classes
	                           * of the same batch run beside it (0 = no limit: round 5's rule) */
	int tune_chain_m = 0;     /* test knob (env CVX_TUNE_CHAIN_M): row-block height class (1, 2, 4) of chained tiles */
	int tune_force_wrap = 0;  /* test knob (env CVX_TUNE_FORCE_WRAP16): route every tile to the int16-run kernels */
	int tune_pen_table = 1;   /* tuning knob (env CVX_TUNE_PEN_TABLE = 0 / 1): convex penalty from the LDS table in the two-phase float-score fills */
	int test_fail_compute = 0; /* test knob (env CVX_TUNE_FAIL_COMPUTE = k): the k-th compute stage of this handle fails (error-path tests) */
	int bt_group = 0;          /* lanes per tile in the backtrack: 0 = auto (by the number of tiles walked together: 64 / 32 / 16 / 8, and 32 for
	                            * the much-longer-than-average reads of a bulk walked at 8 or 16), 8 / 16 / 32 = that many for all, 64 = the
	                            * one-wave-per-tile walk, -1 = round 4's rule (64 below 4 096 tiles, else 8 + 32) (env CVX_TUNE_BT_GROUP) */
	bool overlap_post = false; /* tuning knob (env CVX_TUNE_OVERLAP_POST): backtrack/finalize/compaction of batch k on their own stream, beside the fills of batch k+1 */
	bool sse_variant = false; /* scoring outside the regime where the reference's SSE path equals the scalar recurrence:
	                           * every tile goes to the catch-all kernel's SSE-variant instantiation */
	/* freed batches keep their device arenas and pinned staging and wait here for the next upload
	 * (at most kPoolBatches): hipMalloc / hipFree of multi-GB arenas per call are slow, and hipFree
	 * synchronises the whole device, which would serialise handles that work side by side */
	size_t marks[cvx_batch_s::kSharedMarks] = {0};   /* high-water marks of the streaming slots' buffers (cvx_batch_s::bind) */
	std::vector<cvx_batch_s *> pool;
	std::vector<cvx_batch_s *> pending;      /* streaming jobs whose compute stage is not queued yet */
	std::vector<cvx_batch_s *> live;         /* every streaming job the caller has not released yet (cvx_destroy frees what is left) */
	/* sub-read scoring (cvx_score_batch): persistent staging and device buffers */
	PinBuf sc_hseq, sc_hpairs, sc_hout;
	DevBuf<uint8_t> sc_seq;
	DevBuf<ScorePair> sc_pairs;
	DevBuf<int32_t> sc_rows;
	DevBuf<float> sc_out;
	hipEvent_t sc_ev0 = nullptr, sc_ev1 = nullptr, sc_done = nullptr;
	float sc_kernel_ms = 0.0f;
	float decode_kernel_ms = 0.0f;     /* decode_windows_kernel of the last cvx_genome_decode (cvx_stage_kernel_ms) */
	bool score_no_diag = false;   /* test knob (env CVX_TUNE_SCORE_NO_DIAG): always the row-by-row kernels */
	struct cvx_search_state *search = nullptr;   /* candidate search (cvx_search_batch): persistent staging and device buffers */
};

struct cvx_genome_s {            /* an encoded reference genome resident in HBM (cvx_genome.hip) */
	int device = 0;
	uint64_t n_nibbles = 0;
	int32_t n_starts = 0;
	DevBuf<uint8_t> d_bin;
	DevBuf<uint64_t> d_starts;
};

namespace {

cvx_batch_s *acquire_batch(cvx_context *h) {
	if (!h->pool.empty()) {
		cvx_batch_s *b = h->pool.back();
		h->pool.pop_back();
		return b;
	}
	return new (std::nothrow) cvx_batch_s();
}

void recycle_batch(cvx_context *h, cvx_batch_s *b) {
	const bool failed = b->state == kFailed;
	b->state = kEmpty;
	b->in_flight = false;
	b->have_ops = false;
	b->have_refs = false;
	b->text_done = false;
	b->fail_rc = CVX_OK;
	b->fail_msg.clear();
	if (h) {
		auto it = std::find(h->live.begin(), h->live.end(), b);
		if (it != h->live.end()) h->live.erase(it);
	}
	/* (a job that failed half-way keeps nothing worth pooling: its arenas go back to the allocator) */
	if (h && !failed && h->pool.size() < kPoolBatches) { h->pool.push_back(b); return; }
	b->release();
	delete b;
}

/* something failed after work was queued: nothing may still be running on the arenas when they
 * go back to the allocator (or to the pool) */
void discard_batch(cvx_context *h, cvx_batch_s *b) {
	(void) hipDeviceSynchronize();
	if (h) {
		auto it = std::find(h->pending.begin(), h->pending.end(), b);
		if (it != h->pending.end()) h->pending.erase(it);
		it = std::find(h->live.begin(), h->live.end(), b);
		if (it != h->live.end()) h->live.erase(it);
	}
	b->release();
	delete b;
}

/* A streaming job whose stage failed stays alive (the caller still holds its handle) and remembers why:
 * cvx_wait on it returns this code, cvx_job_release frees it.  Whatever was queued for it is drained first. */
int fail_job(cvx_batch_s *b, int rc) {
	(void) hipDeviceSynchronize();
	b->state = kFailed;
	b->fail_rc = rc;
	b->fail_msg = g_err;
	return rc;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
	float ms = 0.0f;
	if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void) hipGetLastError(); return 0.0f; }
	return ms;
}

/* ---- stage 1: pack into pinned staging and copy to the device, piece by piece (stream `io`) */
/* the streams of an aligning handle beside `main` (cvx_create makes only that one) */
int ensure_streams(cvx_context *c) {
	if (c->s_io) return CVX_OK;
	/* (a handle is not re-entrant, but two threads making a handle's first alignment-side calls at once -- stage_upload and
	 * cvx_nm_profile_ops both come here -- must not create the set twice; and a creation that fails half-way leaves what exists in
	 * place: the next call creates only what is missing -- ADVICE r5) */
	std::lock_guard<std::mutex> lk(c->streams_mtx);
	if (c->s_io) return CVX_OK;
	int prio_lo = 0, prio_hi = 0;
	(void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);     /* numerically lower = higher priority */
	hipError_t e = hipSuccess;
	auto make = [&](hipStream_t *st) { if (e == hipSuccess && *st == nullptr) e = hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
	make(&c->s_post);
	/* the text stage of finished jobs runs beside the fills of the jobs behind them: at normal priority its stream shared a
	 * hardware queue with one of the fill classes and every one of its waits sat out a whole fill kernel (38 ms per launch of
	 * 500 tiles inside ngmlr, profiles/r06_e2e_device_decode.txt); the high-priority queues carry only uploads and plans
	 * (CVX_TEXT_STREAM_PRIO=0: as before) */
	{
		const char *tp = getenv("CVX_TEXT_STREAM_PRIO");
		if (e == hipSuccess && c->s_text == nullptr && !(tp && atoi(tp) == 0)) e = hipStreamCreateWithPriority(&c->s_text, hipStreamNonBlocking, prio_hi);
	}
	make(&c->s_text);
	for (int i = 0; i < kAuxStreams; ++i) make(&c->aux[i]);
	make(&c->s_main2);
	make(&c->s_post2);
	for (int i = 0; i < kAuxStreams; ++i) make(&c->aux2[i]);
	if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->s_io, hipStreamNonBlocking, prio_hi);      /* last: its existence says "all of them" */
	if (e != hipSuccess) { set_err("hipStreamCreate failed: %s", hipGetErrorString(e)); return CVX_ERR_HIP; }
	return CVX_OK;
}

int stage_upload(cvx_context *h, cvx_batch_s *b, int32_t n, const cvx_tile *tiles,
		const cvx_genome_s *genome = nullptr, const uint64_t *ref_position = nullptr) {
	RC_TRY(ensure_streams(h));
	UploadLayout L;
	std::vector<TileIn> tin;
	int bad = -1;
	const bool windows = genome != nullptr;
	const int lrc = upload_layout(n, tiles, tin, L, &bad, windows);
	if (lrc == kLayoutMalformed) { set_err("tile %d malformed", bad); return CVX_ERR_ARG; }
	if (lrc == kLayoutTooLarge) {
		set_err("%llu sequence bytes exceed one batch (4 GiB); split the batch", (unsigned long long) L.seq_total);
		return CVX_ERR_ARG;
	}
	b->n = n;
	b->state = kEmpty;
	b->have_ops = false;
	b->have_refs = false;
	b->text_done = false;
	b->ops_total = 0;
	b->seq_total = L.seq_total;
	b->n_rows = L.n_rows;
	memset(&b->timing, 0, sizeof(b->timing));
	RC_TRY(b->make_events());
	const size_t n1 = (size_t) std::max(n, 1);
	const size_t rows1 = (size_t) std::max<uint64_t>(L.arena_rows, 1);      /* closed-form corridors own no rows (RowView, cvx_types.h) */
	RC_TRY(b->h_delta.ensure((size_t) L.delta_total + 256));
	RC_TRY(b->h_rsrc.ensure(n1 * sizeof(RowSrc)));
	RC_TRY(b->d_delta.ensure((size_t) L.delta_total + 256));
	RC_TRY(b->d_rsrc.ensure(n1));
	RC_TRY(b->h_tin.ensure(n1 * sizeof(TileIn)));
	RC_TRY(b->h_plan.ensure(n1 * sizeof(TilePlan)));
	RC_TRY(b->h_res.ensure(n1 * sizeof(ResultRec) + sizeof(BatchSummary)));
	RC_TRY(b->d_seq.ensure((size_t) L.seq_total + 256));
	RC_TRY(b->d_rows.ensure(rows1));
	RC_TRY(b->d_tin.ensure(n1));
	RC_TRY(b->d_plan.ensure(n1));
	RC_TRY(b->d_trun.ensure(n1));
	RC_TRY(b->d_tout.ensure(n1));
	RC_TRY(b->d_dstoff.ensure(n1));
	RC_TRY(b->d_lists.ensure(2 * n1));             /* fill lists + backtrack order */
	if (b->d_counters.cap < 64) {
		RC_TRY(b->d_counters.ensure(64));
		HIP_TRY(hipMemset(b->d_counters.p, 0, b->d_counters.cap * sizeof(int32_t)));
	}
	RC_TRY(b->d_res.ensure(n1 * sizeof(ResultRec) + sizeof(BatchSummary)));
	if (n) memcpy(b->h_tin.p, tin.data(), (size_t) n * sizeof(TileIn));

	/* a block of sequences that already lies back to back in page-locked memory is not packed: the
	 * device pulls it out of the caller's arena (the job's staging then holds only what the host wrote) */
	const bool zc_qry = n > 0 && L.qry_contig && L.qry_bytes > 0 && in_pinned_block(tiles[0].qry, L.qry_bytes + 4);
	const bool zc_ref = n > 0 && !windows && L.ref_contig && L.ref_bytes > 0 && in_pinned_block(tiles[0].ref, L.ref_bytes + 4);
	b->zero_copy_bytes = (zc_qry ? L.qry_bytes : 0) + (zc_ref ? L.ref_bytes : 0);
	const bool pack_seq = !(zc_qry && (zc_ref || windows));      /* anything left for the host to copy? */
	if (pack_seq) RC_TRY(b->h_seq.ensure((size_t) L.seq_total + 256));
	uint8_t *hseq = b->h_seq.as<uint8_t>();
	uint8_t *hdelta = b->h_delta.as<uint8_t>();
	std::vector<RowOverflow> overflow;
	hipStream_t st = h->s_io;
	/* the three pads: uploaded with the packed blocks, or cleared on the device around the blocks that travel as they are
	 * (queued behind those copies: a copy rounded up to whole dwords may spill a few bytes into the pad that follows) */
	if (pack_seq) upload_zero_pads(L, hseq);
	if (zc_qry || zc_ref) {
		/* Pads around blocks that travel as they are: copied from a page-locked block of zeros, whole 256-byte
		 * units (SDMA engines; a memset would be a kernel that has to find wave slots beside the fill), queued
		 * BEFORE the blocks, which then overwrite the few bytes of overlap.  A block's own copy is rounded up to
		 * whole dwords: up to three bytes of whatever follows it in the caller's arena land in the pad behind it --
		 * pads only have to be readable (every cell outside a tile is forced to the empty element), not zero. */
		const uint64_t zmax = L.pad + 1024;
		RC_TRY(b->h_zero.ensure((size_t) zmax));
		if (b->zero_cap != b->h_zero.cap) { memset(b->h_zero.p, 0, b->h_zero.cap); b->zero_cap = b->h_zero.cap; }
		auto zero_range = [&](uint64_t lo, uint64_t hi) -> int {      /* [lo, hi) widened to 256-byte units, inside the arena */
			lo = lo / 256 * 256;
			hi = std::min<uint64_t>((hi + 255) / 256 * 256, (L.seq_total + 255) / 256 * 256);
			for (uint64_t at = lo; at < hi; at += zmax / 256 * 256) {
				const uint64_t len = std::min<uint64_t>(hi - at, zmax / 256 * 256);
				HIP_TRY(hipMemcpyAsync(b->d_seq.p + at, b->h_zero.p, (size_t) len, hipMemcpyHostToDevice, st));
			}
			return CVX_OK;
		};
		if (zc_qry) {
			RC_TRY(zero_range(0, L.qry_base));
			RC_TRY(zero_range(L.qry_base + L.qry_bytes, L.ref_base));
			HIP_TRY(hipMemcpyAsync(b->d_seq.p + L.qry_base, tiles[0].qry, (size_t) ((L.qry_bytes + 3) / 4 * 4), hipMemcpyHostToDevice, st));
		}
		if (zc_ref) {
			RC_TRY(zero_range(L.ref_base + L.ref_bytes, L.seq_total));
			HIP_TRY(hipMemcpyAsync(b->d_seq.p + L.ref_base, tiles[0].ref, (size_t) ((L.ref_bytes + 3) / 4 * 4), hipMemcpyHostToDevice, st));
		}
	}
	/* what the host still moves per tile decides whether packing is worth threads and pieces */
	uint64_t pack_work = L.delta_total * 9ull;
	if (!zc_qry) pack_work += L.qry_bytes;
	if (!zc_ref && !windows) pack_work += L.ref_bytes;
	const std::vector<uint64_t> &wprefix = L.wprefix;
	int threads = std::max(1, std::min(h->pack_threads, PackPool::get().size()));
	if (pack_work < (8u << 20)) threads = 1;      /* not worth a thread below ~8 MB */
	const int pieces = threads > 1 ? 8 : 1;
	int t0 = 0;
	/* bytes of the two blocks of hseq / of hdelta already on their way (block A = [pad][reads][pad], block B = [references][pad]) */
	uint64_t a_done = 0, b_done = L.ref_base, delta_done = 0;
	const uint64_t a_end_all = L.ref_base, b_end_all = (L.seq_total + 255) / 256 * 256;
	for (int pc = 1; pc <= pieces; ++pc) {
		int t1 = n;
		if (pc < pieces) {
			const uint64_t target = wprefix[(size_t) n] / (uint64_t) pieces * (uint64_t) pc;
			t1 = (int) (std::upper_bound(wprefix.begin(), wprefix.end(), target) - wprefix.begin());
			t1 = std::min(std::max(t1, t0), n);
		}
		if (t1 > t0 && pack_work > 0) {
			std::vector<uint64_t> wp((size_t) (t1 - t0) + 1);
			for (int i = t0; i <= t1; ++i) wp[(size_t) (i - t0)] = wprefix[(size_t) i] - wprefix[(size_t) t0];
			const int base = t0;
			/* every packing range of the piece collects the rows of its misfits in its own list */
			const size_t first = overflow.size();
			overflow.resize(first + (size_t) threads + 1);
			std::atomic<int> slot(0);
			parallel_ranges(t1 - t0, wp, threads, [&](int bg, int en) {
				upload_pack(base + bg, base + en, tiles, tin, hseq, hdelta, L.rsrc, overflow[first + (size_t) slot.fetch_add(1)], !zc_qry, !zc_ref && !windows);
			});
		}
		/* Copy boundaries are multiples of 256 bytes: a host-to-device copy whose address or size is
		 * not dword-aligned is not handed to the SDMA engines but to a blit kernel that pulls the bytes
		 * over PCIe with compute units the fill needs.  The bytes below the rounded-down end are all
		 * packed (tiles are laid out in order inside either block); the remainder travels with the next
		 * piece, the last piece runs to the aligned end. */
		if (!zc_qry) {
			const uint64_t a_end = (t1 == n) ? a_end_all : (uint64_t) tin[(size_t) t1].qry_off / 256 * 256;
			if (a_end > a_done) HIP_TRY(hipMemcpyAsync(b->d_seq.p + a_done, hseq + a_done, (size_t) (a_end - a_done), hipMemcpyHostToDevice, st));
			a_done = std::max(a_done, a_end);
		}
		if (!zc_ref && !windows) {
			const uint64_t b_end = (t1 == n) ? b_end_all : (uint64_t) tin[(size_t) t1].ref_off / 256 * 256;
			if (b_end > b_done) HIP_TRY(hipMemcpyAsync(b->d_seq.p + b_done, hseq + b_done, (size_t) (b_end - b_done), hipMemcpyHostToDevice, st));
			b_done = std::max(b_done, b_end);
		}
		if (L.delta_total) {
			uint64_t del_end = L.delta_total;
			if (t1 < n) {        /* first tile at or after t1 whose rows travel as steps (src_off = step-stream offset until the misfits are renumbered below) */
				int q = t1;
				while (q < n && L.rsrc[(size_t) q].fmt != kRowsDelta8 && L.rsrc[(size_t) q].fmt != kRowsExplicit) q++;
				if (q < n) del_end = L.rsrc[(size_t) q].src_off;
			}
			del_end = (t1 == n) ? (del_end + 255) / 256 * 256 : del_end / 256 * 256;
			if (del_end > delta_done)
				HIP_TRY(hipMemcpyAsync(b->d_delta.p + delta_done, hdelta + delta_done, (size_t) (del_end - delta_done), hipMemcpyHostToDevice, st));
			delta_done = std::max(delta_done, del_end);
		}
		t0 = t1;
	}
	if (n) HIP_TRY(hipMemcpyAsync(b->d_tin.p, b->h_tin.p, (size_t) n * sizeof(TileIn), hipMemcpyHostToDevice, st));
	if (n) {
		/* the misfits' rows (none in any corridor the reference builds), then the rows arena on the device */
		uint64_t n_x = 0;
		for (const RowOverflow &o : overflow) n_x += o.rows.size();
		b->n_rowsx = n_x;
		if (n_x) {
			RC_TRY(b->h_rowsx.ensure((size_t) n_x * sizeof(RowDesc)));
			RC_TRY(b->d_rowsx.ensure((size_t) n_x));
			RowDesc *hx = b->h_rowsx.as<RowDesc>();
			uint64_t at = 0;
			for (const RowOverflow &o : overflow) {
				uint64_t r = 0;
				for (int32_t ti : o.tiles) {
					L.rsrc[(size_t) ti].src_off = at + r;
					r += (uint64_t) tin[(size_t) ti].H;
				}
				if (!o.rows.empty()) memcpy(hx + at, o.rows.data(), o.rows.size() * sizeof(RowDesc));
				at += o.rows.size();
			}
			HIP_TRY(hipMemcpyAsync(b->d_rowsx.p, hx, (size_t) n_x * sizeof(RowDesc), hipMemcpyHostToDevice, st));
		}
		memcpy(b->h_rsrc.p, L.rsrc.data(), (size_t) n * sizeof(RowSrc));
		HIP_TRY(hipMemcpyAsync(b->d_rsrc.p, b->h_rsrc.p, (size_t) n * sizeof(RowSrc), hipMemcpyHostToDevice, st));
		/* rows arena: only for the tiles whose corridors came as arrays; closed forms are evaluated where they are needed */
		if (L.arena_rows) HIP_TRY(launch_expand_rows(b->d_rsrc.p, b->d_tin.p, b->d_delta.p, b->d_rowsx.p, b->d_rows.p, n, false, st));
	}
	if (windows && n) {
		/* the references: decoded from the resident genome straight into the arena (and its last pad cleared) */
		RC_TRY(b->h_win.ensure((size_t) n * sizeof(WindowDesc)));
		RC_TRY(b->d_win.ensure((size_t) n));
		WindowDesc *hw = b->h_win.as<WindowDesc>();
		for (int i = 0; i < n; ++i) {
			hw[i].position = ref_position[i];
			hw[i].dst_off = tin[(size_t) i].ref_off;
			hw[i].n_chars = tiles[i].ref_len;
		}
		HIP_TRY(hipMemcpyAsync(b->d_win.p, hw, (size_t) n * sizeof(WindowDesc), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemsetAsync(b->d_seq.p + L.ref_base + L.ref_bytes, 0, (size_t) (L.seq_total - L.ref_base - L.ref_bytes), st));
		HIP_TRY(launch_decode_windows(genome->d_bin.p, genome->d_starts.p, genome->n_starts, b->d_win.p, n, b->d_seq.p, st));
		/* the decoded characters back to the host, 1 byte per reference base under everything that follows: a caller whose
		 * text stage runs on the host (MD needs the reference base of every mismatch and deletion) reads them there
		 * (cvx_job_window_refs) instead of decoding the window a second time on a core */
		RC_TRY(b->h_refs.ensure((size_t) L.ref_bytes + 64));
		HIP_TRY(hipMemcpyAsync(b->h_refs.p, b->d_seq.p + L.ref_base, (size_t) L.ref_bytes, hipMemcpyDeviceToHost, st));
		b->refs_base = L.ref_base;
		b->have_refs = true;
	}
	b->state = kUploaded;
	return CVX_OK;
}

/* ---- stage 2: corridor analysis on `st`, records back to the host */
int stage_plan(cvx_context *h, cvx_batch_s *b, hipStream_t st) {
	const int n = b->n;
	HIP_TRY(hipEventRecord(b->ev[0], st));
	if (n) {
		HIP_TRY(launch_plan(b->d_rows.p, b->d_rsrc.p, b->d_tin.p, b->d_plan.p, n, b->n_rows / (uint64_t) n, h->max_matrix_mb, st));
		HIP_TRY(hipMemcpyAsync(b->h_plan.p, b->d_plan.p, (size_t) n * sizeof(TilePlan), hipMemcpyDeviceToHost, st));
	}
	HIP_TRY(hipEventRecord(b->ev[1], st));
	HIP_TRY(hipEventRecord(b->ev_in, st));
	b->state = kPlanned;
	return CVX_OK;
}

/* ---- stage 3: host planning, then every kernel of the batch on `main` (+ aux); nothing waits */
int stage_compute(cvx_context *h, cvx_batch_s *b, bool streaming = false) {
	const int n = b->n;
	if (h->test_fail_compute > 0 && --h->test_fail_compute == 0) {      /* test knob: this job fails before anything is queued for it */
		set_err("stage_compute: failure injected by CVX_TUNE_FAIL_COMPUTE");
		return CVX_ERR_OOM;
	}
	/* which stream set carries this batch's kernels (see cvx_context::s_main2) */
	const bool second = streaming && n > 0 && n < kSmallJobTiles && !h->single_lane && ((h->small_jobs++ & 1u) != 0u);
	hipStream_t const S_main = second ? h->s_main2 : h->s_main;
	hipStream_t const S_post = second ? h->s_post2 : h->s_post;
	hipStream_t const *S_aux = second ? h->aux2 : h->aux;
	b->s_run = S_main;
	hipStream_t st = S_main;
	HIP_TRY(hipEventSynchronize(b->ev_in));        /* queued a whole batch ago in the streaming case */
	b->launches.clear();
	b->ops_total = 0;
	b->have_ops = false;
	if (n == 0) {
		HIP_TRY(hipEventRecord(b->ev[4], st));
		hipStream_t ps = h->overlap_post ? S_post : S_main;
		HIP_TRY(hipStreamWaitEvent(ps, b->ev[4], 0));
		HIP_TRY(hipEventRecord(b->ev[2], ps));
		HIP_TRY(hipEventRecord(b->ev[3], ps));
		HIP_TRY(hipEventRecord(b->ev_res, ps));
		b->state = kComputed;
		return CVX_OK;
	}

	/* host planning: kernel class, arena offsets, work lists (cvx_host_logic.h) */
	HostPlan hp;
	PlanTuning tune;
	tune.min_slots = h->tune_min_slots; tune.max_slots = h->tune_max_slots; tune.force_wrap = h->tune_force_wrap; tune.chain_m = h->tune_chain_m;
	tune.force_generic = h->sse_variant ? 1 : 0;
	tune.long_steps = h->tune_long_steps; tune.small_batch = h->tune_small_batch; tune.long_need = h->tune_long_need;
	tune.no_gangs = h->tune_gangs ? 0 : 1;
	/* (a tile that gets chained needs its rows on the host: rebuilt from the step stream the batch still owns) */
	const RowSrc *rsrc = b->h_rsrc.as<RowSrc>();
	host_plan_rows(n, b->plan(), b->tin(), [&](int i, std::vector<RowDesc> &tmp) -> const RowDesc * {
		tmp.resize((size_t) std::max(b->tin()[(size_t) i].H, 1));
		expand_rows_host(rsrc[(size_t) i], b->tin()[(size_t) i].H, b->h_delta.as<uint8_t>(), b->h_rowsx.as<RowDesc>(), tmp.data());
		return tmp.data();
	}, true, tune, hp);
	std::vector<std::vector<int32_t>> &cls = hp.cls;
	std::vector<int32_t> &generic = hp.generic;
	RC_TRY(b->d_dirs.ensure((size_t) hp.dir_dwords + 64));
	RC_TRY(b->d_regions.ensure((size_t) hp.ops_ints + 64));
	/* dense ops arena: an alignment of H read bases has far fewer than H run-length ops (about
	 * 0.3 H at 15 % error); if a batch ever needs more, finalize reports it and stage_ops
	 * compacts again into a larger arena */
	b->dense_cap = std::min<uint64_t>(hp.ops_ints, hp.ops_ints / 3 + 64ull * (uint64_t) n);
	RC_TRY(b->d_dense.ensure((size_t) b->dense_cap + 64));
	b->dense_cap = std::max<uint64_t>(b->dense_cap, b->d_dense.cap - 64);

	RC_TRY(b->h_trun.ensure((size_t) n * sizeof(TileRun)));
	RC_TRY(b->h_tout.ensure((size_t) n * sizeof(TileOut)));
	RC_TRY(b->h_lists.ensure((size_t) 2 * n * sizeof(int32_t) + 64));
	memcpy(b->h_trun.p, hp.trun.data(), (size_t) n * sizeof(TileRun));
	memcpy(b->h_tout.p, hp.tout.data(), (size_t) n * sizeof(TileOut));
	int32_t *lists = b->h_lists.as<int32_t>();
	size_t n_listed = 0;
	std::vector<int> seg_begin(cls.size(), 0);
	/* Very long tiles go straight to the exact-tracking instantiation.  The two-phase pass saves three half-rate ops per
	 * cell (~12 %) but a tile whose best cell is not in its last anti-diagonals -- a local alignment: an inverted segment,
	 * a read that does not reach its end -- is redone from step 0, and for a 100 kb tile that second pass is another
	 * ~100 ms on one wave behind everything else (C5 mix: 77 ms of a 200 ms fill).  Tiles of kExactDirectSteps steps and
	 * more are flagged kPadRedo up front and listed first; the exact launch over that prefix does them once.  Only while
	 * the class cannot fill the device several times over (then 12 % of throughput would cost more than the tail). */
	std::vector<int> n_direct(cls.size(), 0);
	for (size_t c = 0; c < cls.size(); ++c) {
		seg_begin[c] = (int) n_listed;      /* already in LPT order */
		if (h->tune_exact_steps > 0 && !cls[c].empty()) {
			int nl = 0;
			for (int32_t ti : cls[c]) if (hp.trun[(size_t) ti].nsteps >= h->tune_exact_steps) nl++;
			if (nl > 0 && nl <= kExactDirectMaxTiles) {
				std::stable_partition(cls[c].begin(), cls[c].end(), [&](int32_t ti) { return hp.trun[(size_t) ti].nsteps >= h->tune_exact_steps; });
				TileOut *ho = b->h_tout.as<TileOut>();
				for (int q = 0; q < nl; ++q) ho[(size_t) cls[c][(size_t) q]].pad = kPadRedo;
				n_direct[c] = nl;
			}
		}
		if (!cls[c].empty()) memcpy(lists + n_listed, cls[c].data(), cls[c].size() * sizeof(int32_t));
		n_listed += cls[c].size();
	}
	const int generic_begin = (int) n_listed;
	if (!generic.empty()) memcpy(lists + n_listed, generic.data(), generic.size() * sizeof(int32_t));
	n_listed += generic.size();
	/* behind the fill lists: every computed tile once, longest read first (counting sort on H / 32) --
	 * the order in which the backtrack takes them, four to a wave: the walk of a tile is a serial
	 * chain of ~H / 7 probes, so the long ones must start first and share their wave with their like.
	 * One segment per fill launch, in launch order (chained classes, whole-tile classes from the widest ring down, the
	 * catch-all kernel): a batch of several classes walks each class right behind its own fill, on that fill's stream,
	 * while the other classes still fill; a batch of one class has one segment = the whole list. */
	const size_t bt_begin = n_listed;
	std::vector<const std::vector<int32_t> *> launch_tiles;
	for (size_t c = 0; c < hp.chain_tasks.size(); ++c) if (!hp.chain_tasks[c].empty()) launch_tiles.push_back(&hp.chain_tiles[c]);
	for (int cc = (int) cls.size() - 1; cc >= 0; --cc) if (!cls[(size_t) cc].empty()) launch_tiles.push_back(&cls[(size_t) cc]);
	if (!generic.empty()) launch_tiles.push_back(&generic);
	std::vector<std::pair<size_t, int>> bt_seg;      /* (offset in lists, tiles) per launch */
	{
		const TileIn *tin = b->tin();
		constexpr int kBuckets = 4096;
		auto bucket = [&](int32_t ti) { const int k = tin[(size_t) ti].H >> 5; return kBuckets - 1 - (k < kBuckets ? k : kBuckets - 1); };
		std::vector<int32_t> count((size_t) kBuckets + 1);
		for (const std::vector<int32_t> *v : launch_tiles) {
			std::fill(count.begin(), count.end(), 0);
			for (int32_t ti : *v) if (!hp.trun[(size_t) ti].skip) count[(size_t) bucket(ti) + 1]++;
			for (int k = 0; k < kBuckets; ++k) count[(size_t) k + 1] += count[(size_t) k];
			const size_t at = n_listed;
			const int m = count[(size_t) kBuckets];
			/* (stable: inside a bucket of equally long reads the class's own order, most cells first) */
			for (int32_t ti : *v) if (!hp.trun[(size_t) ti].skip) lists[at + (size_t) count[(size_t) bucket(ti)]++] = ti;
			n_listed += (size_t) m;
			bt_seg.emplace_back(at, m);
		}
	}
	const int n_walk = (int) (n_listed - bt_begin);
	if (!generic.empty()) {
		RC_TRY(b->h_goff.ensure((generic.size() + 1) * sizeof(uint64_t)));
		uint64_t *goff = b->h_goff.as<uint64_t>();
		goff[0] = 0;
		for (size_t g = 0; g < generic.size(); ++g)
			goff[g + 1] = goff[g] + (uint64_t) generic_scratch_bytes(hp.trun[(size_t) generic[g]].ring);
		RC_TRY(b->d_gscratch.ensure((size_t) goff[generic.size()] + 256));
		RC_TRY(b->d_gscratch_off.ensure(generic.size() + 1));
		HIP_TRY(hipMemcpyAsync(b->d_gscratch_off.p, goff, (generic.size() + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
	}
	/* chained tiles: tasks of every chain class, block table and tile lists in one upload */
	size_t chain_task_off[kNumChainClasses * 2] = {0}, chain_tile_off[kNumChainClasses * 2] = {0};
	size_t chain_blk_off = 0, chain_bytes = 0;
	if (hp.n_chained) {
		for (size_t c = 0; c < hp.chain_tasks.size(); ++c) { chain_task_off[c] = chain_bytes; chain_bytes += hp.chain_tasks[c].size() * sizeof(ChainTask); }
		chain_blk_off = chain_bytes; chain_bytes += hp.chain_blk.size() * sizeof(ChainBlk);
		for (size_t c = 0; c < hp.chain_tiles.size(); ++c) { chain_tile_off[c] = chain_bytes; chain_bytes += (hp.chain_tiles[c].size() * sizeof(int32_t) + 7) / 8 * 8; }
		RC_TRY(b->h_chain.ensure(chain_bytes));
		uint8_t *hc = b->h_chain.as<uint8_t>();
		for (size_t c = 0; c < hp.chain_tasks.size(); ++c)
			if (!hp.chain_tasks[c].empty()) memcpy(hc + chain_task_off[c], hp.chain_tasks[c].data(), hp.chain_tasks[c].size() * sizeof(ChainTask));
		memcpy(hc + chain_blk_off, hp.chain_blk.data(), hp.chain_blk.size() * sizeof(ChainBlk));
		for (size_t c = 0; c < hp.chain_tiles.size(); ++c)
			if (!hp.chain_tiles[c].empty()) memcpy(hc + chain_tile_off[c], hp.chain_tiles[c].data(), hp.chain_tiles[c].size() * sizeof(int32_t));
		RC_TRY(b->d_chain.ensure(chain_bytes));
		RC_TRY(b->d_chain_out.ensure(hp.chain_blk.size()));
		{
			/* boundary records validate themselves by the launch epoch in their upper bits: a buffer starts out zeroed
			 * (epoch 0 = never written) and is zeroed again when the epochs wrap */
			const size_t had = b->d_bnd.cap;
			RC_TRY(b->d_bnd.ensure((size_t) hp.bnd_recs + 64));
			if (b->d_bnd.cap != had || b->bnd_epoch >= kBndEpochMax) {
				HIP_TRY(hipMemsetAsync(b->d_bnd.p, 0, b->d_bnd.cap * sizeof(BoundaryRec), st));
				b->bnd_epoch = 0;
			}
			b->bnd_epoch += 1;
		}
		HIP_TRY(hipMemcpyAsync(b->d_chain.p, hc, chain_bytes, hipMemcpyHostToDevice, st));
	}
	HIP_TRY(hipMemcpyAsync(b->d_trun.p, b->h_trun.p, (size_t) n * sizeof(TileRun), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(b->d_tout.p, b->h_tout.p, (size_t) n * sizeof(TileOut), hipMemcpyHostToDevice, st));
	if (n_listed) HIP_TRY(hipMemcpyAsync(b->d_lists.p, lists, n_listed * sizeof(int32_t), hipMemcpyHostToDevice, st));
	/* (the batch's counters are zero: cleared when the arena was allocated and again by finalize_kernel,
	 * their last reader -- a memset here would be a tiny kernel that has to find a free wave slot among
	 * the previous batch's 24 576 backtrack waves before this batch's fills may start: measured 7 ms) */
	HIP_TRY(hipEventRecord(b->ev[4], st));        /* inputs of the fills are in place */

	/* forward fill: one launch per populated kernel class (+ its exact redo pass), classes run
	 * concurrently on separate streams (a sparsely populated class would otherwise serialise a
	 * whole tile latency behind the big one); widest rings first, they have the longest tiles */
	auto fill_args = [&](const int32_t *list, int list_n) {
		FillArgs a;
		a.seq = b->d_seq.p;
		a.rows = reinterpret_cast<const RowDesc2 *>(b->d_rows.p);
		a.rsrc = b->d_rsrc.p;
		a.tin = b->d_tin.p;
		a.trun = b->d_trun.p;
		a.tout = b->d_tout.p;
		a.dirs = b->d_dirs.p;
		a.list = list;
		a.list_n = list_n;
		a.redo_count = b->d_counters.p;
		a.late_min_groups = h->tune_late_min;
		a.late_shift = h->tune_late_shift;
		a.pen_table = h->tune_pen_table;
		a.tasks = nullptr; a.chain_ticket = nullptr; a.bnd = nullptr; a.chain_out = nullptr; a.bnd_epoch = 0; a.chain_prio = 0;
		a.ops = b->d_regions.p;
		a.sp = h->sp;
		return a;
	};
	auto launch_stats = [&](const std::vector<int32_t> &v, int m, int nw, int wrap, int kind) {
		cvx_launch_info li;
		memset(&li, 0, sizeof(li));
		li.slots_per_lane = m; li.waves = nw; li.wrap16 = wrap; li.n_tiles = (int) v.size(); li.kind = kind;
		for (int32_t ti : v) {
			const TilePlan &p = b->plan()[(size_t) ti];
			const TileIn &in = b->tin()[(size_t) ti];
			li.cells += p.cells; li.active_cells += p.active;
			li.alg_bytes += p.cells + 6ull * (uint64_t) in.H + 2ull * (uint64_t) in.W;
			li.read_bases += (uint64_t) in.H;
		}
		b->launches.push_back(li);
	};
	/* backtrack, device-side result records + prefix sums, ops compaction */
	BacktrackArgs ba;
	ba.chain_blk = hp.n_chained ? reinterpret_cast<const ChainBlk *>(b->d_chain.p + chain_blk_off) : nullptr;
	ba.seq = b->d_seq.p;
	ba.rows = reinterpret_cast<const RowDesc2 *>(b->d_rows.p);
	ba.rsrc = b->d_rsrc.p;
	ba.tin = b->d_tin.p;
	ba.trun = b->d_trun.p;
	ba.tout = b->d_tout.p;
	ba.dirs = b->d_dirs.p;
	ba.ops = b->d_regions.p;
	ba.n_tiles = n;
	/* the walk of lists[at, at + count) (longest read first) on `ws`.  Few tiles in the batch: the walk is latency-bound and
	 * 64 probing lanes per tile take the long diagonal runs in a quarter of the probes; many tiles: it is issue-bound and
	 * several tiles share a wave -- eight for the bulk; the few reads much longer than the rest (a latency-bound tail, a
	 * serial chain of H / 7 probes each) get 32 lanes per tile, beside the bulk on `side` when `fork` (one walk for the whole
	 * batch), in front of it on the same stream otherwise (measured: PacBio 5.4 -> 4.9 ms with 8 lanes, ONT mix 8.2 -> 6.6 with 32) */
	auto walk_list = [&](size_t at, int count, hipStream_t ws, hipStream_t side, bool fork) -> int {
		if (count <= 0) return CVX_OK;
		/* Lanes per tile by the number of tiles walked together (round 5): a walk is a serial chain of probes per tile, and
		 * what hides a probe's latency is other waves -- so few tiles get many lanes each (a probe then covers 64 / 32 / 16
		 * path columns of a diagonal run instead of 8) until the walk has about six waves per SIMD, and only beyond that
		 * is it issue-bound and eight lanes per tile the cheapest.  Measured against round 4's rule (one wave per tile below
		 * 4 096 tiles, eight lanes from there on): C5 mix, 4 096 tiles of 100 kb, walk 39.5 ms at 8 lanes = 512 waves on 1 024
		 * SIMDs, 25.9 at 16, 21.6 at 32, 19.4 at 64; ONT mix at 49 152 tiles in one walk 7.6 ms at 8, 6.0 at 16; the PacBio
		 * bench (49 152 tiles): the walk alone 7.7 ms at 8 and 8.4 at 16, the pipelined step 114.4-114.9 against 113.9-114.1 ms
		 * (profiles/r05_ab_bt_group.txt). */
		const int auto_group = count <= 6144 ? 64 : count <= 12288 ? 32 : count <= 49152 ? 16 : 8;
		if (h->bt_group < 0 ? n_walk < 4096 : (h->bt_group == 0 && auto_group == 64)) {
			HIP_TRY(launch_backtrack(ba, b->d_lists.p + at, count, 64, ws));
		} else if (h->bt_group > 0) {
			HIP_TRY(launch_backtrack(ba, b->d_lists.p + at, count, h->bt_group, ws));
		} else if (h->bt_group == 0 && auto_group >= 32) {
			HIP_TRY(launch_backtrack(ba, b->d_lists.p + at, count, auto_group, ws));
		} else {
			const int bulk = h->bt_group < 0 ? 8 : auto_group;      /* 8 or 16; the much-longer-than-average reads at 32 */
			const TileIn *tin = b->tin();
			const uint64_t mean_h = b->n_rows / (uint64_t) std::max(n, 1);
			int n_long = 0;
			while (n_long < count && (uint64_t) tin[(size_t) lists[at + (size_t) n_long]].H > 3 * mean_h) n_long++;
			if (n_long > 0 && fork) {
				HIP_TRY(hipEventRecord(b->ev_bt0, ws));
				HIP_TRY(hipStreamWaitEvent(side, b->ev_bt0, 0));
				HIP_TRY(launch_backtrack(ba, b->d_lists.p + at, n_long, 32, side));
				HIP_TRY(hipEventRecord(b->ev_bt1, side));
			} else if (n_long > 0) {
				HIP_TRY(launch_backtrack(ba, b->d_lists.p + at, n_long, 32, ws));
			}
			HIP_TRY(launch_backtrack(ba, b->d_lists.p + at + n_long, count - n_long, bulk, ws));
			if (n_long > 0 && fork) HIP_TRY(hipStreamWaitEvent(ws, b->ev_bt1, 0));
		}
		return CVX_OK;
	};
	/* Several fill classes (ONT mix: chained retries, M = 4, M = 3; C5): each class is walked right behind its own fill on
	 * that fill's stream.  The launch of such a batch lasts as long as its longest dependency chain, and while the last
	 * chains finish on a few waves the device has issue slots to spare: the other classes' walks run there instead of
	 * behind everything (CVX_TUNE_BT_PER_CLASS=0: one walk behind all fills, as a batch of one class has it anyway). */
	static const bool bt_per_class_env = !(getenv("CVX_TUNE_BT_PER_CLASS") && atoi(getenv("CVX_TUNE_BT_PER_CLASS")) == 0);
	/* Only where the walk is issue-bound (>= 4096 tiles; measured, r04c: ONT mix 60 000 tiles 57.2 -> 56.0 ms, 24 000 tiles
	 * 35.3 -> 34.2, C5 mix 6 144 tiles 397 -> 369 ms); a small batch's one-wave-per-tile walks are latency-bound chains that
	 * gain nothing from starting early and cost the fills still running (C5 mix 2 048 tiles: 183.6 -> 187.2 ms). */
	const bool per_class = bt_per_class_env && launch_tiles.size() > 1 && !h->overlap_post && n_walk >= 4096;
	int launches = 0;
	/* fill launches go round-robin over the two aux streams and the main stream itself (which has
	 * nothing else to do until they are all done): three classes side by side */
	/* (the `post` stream carries a fill class too unless the post-fill overlap experiment owns it) */
	hipStream_t fill_streams[kAuxStreams + 2];
	int n_fill_streams = 0;
	/* (order: first side stream, post, main -- the assignment of rounds 2-4 for up to three classes -- and the second side stream
	 * only for a fourth class, i.e. with gangs.  Which class rides on which stream is not neutral: with the three whole-tile
	 * classes of the C5 mix on side / side / post instead of side / post / main the same batch takes 181 or 236 ms depending on the
	 * handle, on side / post / main 194 every time: gpurun_out r05u, profiles/r05_fill_stream_order.txt) */
	fill_streams[n_fill_streams++] = S_aux[0];
	if (!h->overlap_post) fill_streams[n_fill_streams++] = S_post;
	fill_streams[n_fill_streams++] = st;
	for (int i = 1; i < kAuxStreams; ++i) fill_streams[n_fill_streams++] = S_aux[i];
	auto begin_launch = [&](hipStream_t ls) -> int {
		while (b->lev.size() < (size_t) (launches + 1) * 4) {
			hipEvent_t e;
			HIP_TRY(hipEventCreate(&e));
			b->lev.push_back(e);
		}
		HIP_TRY(hipStreamWaitEvent(ls, b->ev[4], 0));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4], ls));
		return CVX_OK;
	};
	/* closes launch number `launches` on its stream: the class's own walk (per_class), then the event everything after waits for */
	auto end_launch = [&](hipStream_t ls) -> int {
		if (per_class) RC_TRY(walk_list(bt_seg[(size_t) launches].first, bt_seg[(size_t) launches].second, ls, ls, false));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4 + 3], ls));
		launches++;
		return CVX_OK;
	};
	/* chained tiles first (their dependency chains are the longest thing in a batch): the row-block
	 * tasks of a class, then the per-tile reduction of the block results */
	for (size_t c = 0; c < hp.chain_tasks.size(); ++c) {
		if (hp.chain_tasks[c].empty()) continue;
		const int m = kChainClasses[c / 2];
		launch_stats(hp.chain_tiles[c], m, (int) hp.chain_tasks[c].size(), (int) (c & 1), CVX_LAUNCH_CHAINED);     /* `waves` = row-block tasks */
		hipStream_t ls = fill_streams[launches % n_fill_streams];
		RC_TRY(begin_launch(ls));
		FillArgs a = fill_args(nullptr, (int) hp.chain_tasks[c].size());
		a.tasks = reinterpret_cast<const ChainTask *>(b->d_chain.p + chain_task_off[c]);
		a.chain_ticket = b->d_counters.p + 8 + (int) c;
		a.bnd = b->d_bnd.p;
		a.bnd_epoch = b->bnd_epoch;
		a.chain_prio = (h->tune_chain_prio >= 0) ? h->tune_chain_prio : 1;
		a.chain_out = b->d_chain_out.p;
		/* tasks are dispatched in order, long before their turn; resident tasks beyond the ones that can
		 * actually run only poll.  Unused dynamic LDS caps the residency at ~1.5x the blocks that are
		 * live at one time (need / rows-per-block per tile, + slack). */
		uint64_t live = 0;
		for (int32_t ti : hp.chain_tiles[c]) live += (uint64_t) b->plan()[(size_t) ti].need / (uint64_t) (64 * m + kChainChunk) + 2;
		const uint64_t resident = std::min<uint64_t>(8192, std::max<uint64_t>(768, live + live / 2 + 256));
		const size_t per_cu = (size_t) ((resident + (uint64_t) h->num_cus - 1) / (uint64_t) h->num_cus);
		size_t pad_lds = per_cu >= 32 ? 0 : (size_t) (160 * 1024) / per_cu - 4096;
		pad_lds = std::min<size_t>(pad_lds, 60 * 1024) / 256 * 256;
		/* The padding is LDS the ring classes of the same batch cannot use: a handful of chained retries among a thousand whole
		 * tiles (ngmlr's own launches: 10-40 chained tiles, resident = 768 tasks = 3 per CU at 50 KB each) held 150 of a CU's
		 * 160 KB for their 7 ms, and the M = 3 / M = 4 classes -- 4-5 KB per wave -- crawled until they were gone: a launch's
		 * fill was the SUM of the chained class and the widest ring class (18.6 = 7.0 + 11.7 ms, profiles/r06_e2e_launch_trace.txt).
		 * Beside ring classes the cap may hold tune_chain_lds_kb per CU; more tasks than can run then sit in their back-off sleep. */
		bool rings_beside = false;
		for (size_t rc_ = 0; rc_ < cls.size(); ++rc_) rings_beside = rings_beside || !cls[rc_].empty();
		if (rings_beside && h->tune_chain_lds_kb > 0 && per_cu > 0 && per_cu < 32) {
			const size_t budget = (size_t) h->tune_chain_lds_kb * 1024 / per_cu;
			const size_t capped = budget > 4096 ? (budget - 4096) / 256 * 256 : 0;
			pad_lds = std::min(pad_lds, capped);
		}
		HIP_TRY(launch_fill(m, 1, (c & 1) != 0, 2, a, pad_lds, ls));
		HIP_TRY(launch_chain_reduce(reinterpret_cast<const int32_t *>(b->d_chain.p + chain_tile_off[c]), (int) hp.chain_tiles[c].size(),
				b->d_trun.p, b->d_chain_out.p, b->d_tout.p, ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4 + 1], ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4 + 2], ls));
		RC_TRY(end_launch(ls));
	}
	int ring_classes = 0;
	for (size_t c = 0; c < cls.size(); ++c) ring_classes += cls[c].empty() ? 0 : 1;
	bool widest = true;
	for (int cc = (int) cls.size() - 1; cc >= 0; --cc) {
		const size_t c = (size_t) cc;
		if (cls[c].empty()) continue;
		const KernelClass &kc = kClasses[c / 2];
		const int wide_prio = (h->tune_wide_prio && widest && ring_classes > 1) ? h->tune_wide_prio : 0;
		widest = false;
		launch_stats(cls[c], kc.m, kc.gang, (int) (c & 1), kc.gang > 1 ? CVX_LAUNCH_GANG : CVX_LAUNCH_WHOLE);      /* `waves` = waves per tile (a gang's size) */
		hipStream_t ls = fill_streams[launches % n_fill_streams];
		RC_TRY(begin_launch(ls));
		if (n_direct[c] > 0) {
			/* the very long tiles of the class, exact from the first step (flagged above), before everything else */
			FillArgs ad = fill_args(b->d_lists.p + seg_begin[c], n_direct[c]);
			ad.chain_prio = kc.gang > 1 ? h->tune_gang_prio : wide_prio;
			HIP_TRY(launch_fill(kc.m, kc.gang, (c & 1) != 0, 1, ad, 0, ls));
		}
		FillArgs a = fill_args(b->d_lists.p + seg_begin[c] + n_direct[c], (int) cls[c].size() - n_direct[c]);
		a.chain_prio = kc.gang > 1 ? h->tune_gang_prio : wide_prio;
		if (a.list_n > 0) HIP_TRY(launch_fill(kc.m, kc.gang, (c & 1) != 0, 0, a, 0, ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4 + 1], ls));
		/* exact-tracking pass over the tiles the two-phase pass flagged (usually none) */
		if (a.list_n > 0) HIP_TRY(launch_fill(kc.m, kc.gang, (c & 1) != 0, 1, a, 0, ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4 + 2], ls));
		RC_TRY(end_launch(ls));
	}
	if (!generic.empty()) {
		launch_stats(generic, 0, 16, 1, CVX_LAUNCH_CATCH_ALL);
		hipStream_t ls = fill_streams[launches % n_fill_streams];
		RC_TRY(begin_launch(ls));
		const FillArgs a = fill_args(b->d_lists.p + generic_begin, (int) generic.size());
		HIP_TRY(launch_fill_generic(a, h->sse_variant, b->d_gscratch.p, b->d_gscratch_off.p, ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4 + 1], ls));
		HIP_TRY(hipEventRecord(b->lev[(size_t) launches * 4 + 2], ls));
		RC_TRY(end_launch(ls));
	}
	/* Everything after the fills CAN run on its own stream, so that `main` goes straight on to the next
	 * batch's fills while this batch's backtrack (one wave per tile) and small kernels run beside them. */
	/* Measured (profiles/r02_timeline.txt): the overlap only moves time around -- fill and backtrack
	 * are bound by the same issue slots, the step takes fill + backtrack either way (76 ms for 24 576
	 * PacBio tiles), and the fill's own launch stretches from 66 to 76 ms.  It therefore stays OFF by
	 * default (post == main, stages back to back, clean per-kernel timings); CVX_TUNE_OVERLAP_POST=1
	 * turns it on. */
	st = h->overlap_post ? S_post : S_main;
	HIP_TRY(hipStreamWaitEvent(st, b->ev[4], 0));  /* also orders `post` behind the input copies when no fill was launched */
	for (int i = 0; i < launches; ++i) HIP_TRY(hipStreamWaitEvent(st, b->lev[(size_t) i * 4 + 3], 0));
	HIP_TRY(hipEventRecord(b->ev[2], st));

	if (!per_class) {
		RC_TRY(walk_list(bt_begin, n_walk, st, S_aux[0], true));
	}
	ResultRec *d_rec = reinterpret_cast<ResultRec *>(b->d_res.p);
	BatchSummary *d_sum = reinterpret_cast<BatchSummary *>(b->d_res.p + (size_t) n * sizeof(ResultRec));
	HIP_TRY(launch_finalize(b->d_tout.p, b->d_plan.p, b->d_dstoff.p, d_rec, d_sum, b->d_counters.p, n, b->dense_cap, st));
	HIP_TRY(launch_compact(b->d_regions.p, b->d_trun.p, b->d_tout.p, b->d_dstoff.p, b->d_dense.p, n, b->dense_cap, st));
	HIP_TRY(hipEventRecord(b->ev[3], st));
	HIP_TRY(hipMemcpyAsync(b->h_res.p, b->d_res.p, (size_t) n * sizeof(ResultRec) + sizeof(BatchSummary), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipEventRecord(b->ev_res, st));

	b->timing.cells = hp.cells;
	b->timing.active_cells = hp.active;
	b->timing.dir_bytes = hp.dir_dwords * 4;
	b->timing.n_fill_launches = launches;
	b->timing.n_tiles_fast = hp.n_fast;
	b->timing.n_tiles_chained = hp.n_chained;
	b->state = kComputed;
	return CVX_OK;
}

/* ---- wait for the result records; timing of the batch */
int stage_results(cvx_context *h, cvx_batch_s *b) {
	(void) h;
	if (b->state < kComputed) { set_err("internal: results requested from a batch whose kernels were never queued (state %d)", b->state); return CVX_ERR_ARG; }
	HIP_TRY(hipEventSynchronize(b->ev_res));
	if (b->n == 0) { b->ops_total = 0; b->state = kFinished; return CVX_OK; }
	const BatchSummary *s = b->summary();
	b->ops_total = s->ops_total;
	const int launches = b->timing.n_fill_launches;
	for (int i = 0; i < launches; ++i) b->launches[(size_t) i].ms = ev_ms(b->lev[(size_t) i * 4], b->lev[(size_t) i * 4 + 1]);
	b->timing.plan_ms = ev_ms(b->ev[0], b->ev[1]);
	/* fill = until the last fill class has finished its exact pass; backtrack = what is left of the compute stage (when every
	 * class is walked behind its own fill, the walks of the early classes lie inside `fill`: the two still add up) */
	float fill_end = 0.0f;
	for (int i = 0; i < launches; ++i) fill_end = std::max(fill_end, ev_ms(b->ev[4], b->lev[(size_t) i * 4 + 2]));
	if (launches == 0) fill_end = ev_ms(b->ev[4], b->ev[2]);
	b->timing.fill_ms = fill_end;
	b->timing.backtrack_ms = std::max(0.0f, ev_ms(b->ev[4], b->ev[3]) - fill_end);
	b->timing.total_ms = b->timing.plan_ms + ev_ms(b->ev[4], b->ev[3]);
	b->timing.n_tiles_redone = s->n_redone;
	b->timing.chain_task_ticks = s->chain_task_ticks;
	b->timing.chain_poll_ticks = s->chain_poll_ticks;
	b->state = kFinished;
	return CVX_OK;
}

/* ---- stage 4: dense ops to pinned host memory (stream `io`) */
int stage_ops(cvx_context *h, cvx_batch_s *b) {
	if (b->have_ops) return CVX_OK;
	if (b->ops_total > b->dense_cap) {
		/* rare: more ops than the arena was sized for -- grow it and compact again */
		const uint64_t cap = b->ops_total;
		RC_TRY(b->d_dense.ensure((size_t) cap + 64));
		b->dense_cap = b->d_dense.cap - 64;
		hipStream_t ps = b->s_run ? b->s_run : h->s_main;      /* the batch's own stream set: behind everything it queued */
		HIP_TRY(launch_compact(b->d_regions.p, b->d_trun.p, b->d_tout.p, b->d_dstoff.p, b->d_dense.p, b->n, b->dense_cap, ps));
		HIP_TRY(hipStreamSynchronize(ps));
	}
	if (b->ops_total) {
		RC_TRY(b->h_ops.ensure((size_t) b->ops_total * sizeof(uint32_t)));
		/* The wait below is for the whole io stream, on purpose: it already carries the upload and corridor analysis of the
		 * job submitted last, and returning only when those are done paces the caller -- it submits its next batch one
		 * step later, so that exactly one corridor analysis runs beside each fill (beside a fill it takes most of the
		 * fill's duration; two of them queued under one fill finish late and the next fill starts late: measured 150
		 * instead of 124 ms per step with an event right behind the copy).  CVX_TUNE_OPS_EVENT=1 selects the event. */
		HIP_TRY(hipMemcpyAsync(b->h_ops.p, b->d_dense.p, (size_t) b->ops_total * sizeof(uint32_t), hipMemcpyDeviceToHost, h->s_io));
		/* Round 4 tried to drop the pacing for closed-form batches (their corridor analysis reads nothing: 2.3 ms alone): over
		 * 3 steps of 24 576 tiles the event-only form measured the same, over the driver's 20 steps of 49 152 tiles it costs
		 * 146.5 instead of 118.7 ms per step (gpurun_out/r04g/pacing.txt) -- the analysis is starved beside a fill whatever it
		 * reads (70-100 ms), and two of them queued under one fill still delay the fill after next.  The stream wait stays;
		 * CVX_TUNE_OPS_EVENT=1 selects the event. */
		static const bool ops_event = getenv("CVX_TUNE_OPS_EVENT") && atoi(getenv("CVX_TUNE_OPS_EVENT")) != 0;
		if (ops_event) {
			HIP_TRY(hipEventRecord(b->ev_ops, h->s_io));
			HIP_TRY(hipEventSynchronize(b->ev_ops));
		} else {
			HIP_TRY(hipStreamSynchronize(h->s_io));
		}
	}
	b->have_ops = true;
	return CVX_OK;
}

/* queue the compute stage of every submitted batch whose plan records have arrived (all of them,
 * up to `upto`, when `block`): keeps the device one batch ahead of the host */
int pump(cvx_context *h, bool block, const cvx_batch_s *upto) {
	while (!h->pending.empty()) {
		cvx_batch_s *b = h->pending.front();
		if (!block) {
			hipError_t q = hipEventQuery(b->ev_in);
			if (q == hipErrorNotReady) { (void) hipGetLastError(); break; }
			if (q != hipSuccess) { set_err("hipEventQuery: %s", hipGetErrorString(q)); h->pending.erase(h->pending.begin()); (void) fail_job(b, CVX_ERR_HIP); continue; }
		}
		h->pending.erase(h->pending.begin());
		/* a failure (say, the direction arena of a multi-GB batch does not fit beside the batches in flight) belongs
		 * to THIS job: it is recorded on it and reported by its own cvx_wait, never against another job's call */
		const int rc = stage_compute(h, b, true);
		if (rc != CVX_OK) (void) fail_job(b, rc);
		if (upto && b == upto) break;
	}
	return CVX_OK;
}

}  // namespace

/* the process's pack threads for the host-only text code (cvx_sam.cpp) */
namespace cvx {
void pack_pool_run(int n_tasks, const std::function<void(int)> &fn) { PackPool::get().run(n_tasks, fn); }
}

/* Before the HIP runtime reads its settings (it does at the first HIP call of the process, which for ngmlr and for bench.py is
 * one of ours): sixteen hardware queues per device instead of the runtime's four.  A handle runs its fill classes side by side
 * on three streams beside the upload stream, and a process holds several handles (ngmlr: the aligner, sixteen search handles,
 * the scoring plugin's); with four hardware queues the streams of a later handle share queues, and two fill classes that
 * share a queue run one after the other -- measured (gpurun_out/r05b): the ONT mix 45 ms per batch on the first handle of a
 * process, 59 ms on every later one (C5 mix: 195 / 288 ms); with eight queues every handle gets the 45 / 195.  Sixteen, not
 * eight, because of ngmlr: there the aligner's fill streams are created when fifty streams of the search and scoring handles
 * exist, and with eight queues two of its classes still landed on one -- the launch trace of the pipeline
 * (CVX_LAUNCH_TRACE=1, profiles/r05_e2e_launch_trace.txt) showed a launch's fill lasting the SUM of its classes (M = 4 10.1 ms +
 * M = 3 7.7 ms = 19 ms of a 20.5 ms launch); with sixteen: 10.7 of 12.2 ms, mapping 3.0 -> 2.6 s for 20 000 reads
 * (profiles/r05_e2e_hw_queues.txt).  The bench configurations do not care (8 against 16: C5 6 834 - 6 887 both, ONT 8 550 - 8 970
 * both); thirty-two oversubscribe the hardware (launches 53 ms in flight).  Not overridden when the user has set
 * GPU_MAX_HW_QUEUES. */
static int g_hwq_set_by_library = 0;            /* 1: the constructor exported GPU_MAX_HW_QUEUES itself (the user had not) */
static std::atomic<int> g_blocking_sync[64];    /* per device: 0 not decided, 1 applied, 2 refused by the runtime, 3 CVX_WAIT=spin */
static int g_runtime_up_at_load = 0;             /* 1: the process had /dev/kfd open when this library was loaded -- a HIP / HSA runtime was already
                                                  * initialised, so the variable exported below cannot reach it any more */
__attribute__((constructor)) static void cvx_process_settings() {
	g_hwq_set_by_library = getenv("GPU_MAX_HW_QUEUES") == nullptr;
	if (DIR *d = opendir("/proc/self/fd")) {
		while (struct dirent *e = readdir(d)) {
			char path[320], target[64];
			snprintf(path, sizeof(path), "/proc/self/fd/%s", e->d_name);
			const ssize_t n = readlink(path, target, sizeof(target) - 1);
			if (n > 0) { target[n] = 0; if (strcmp(target, "/dev/kfd") == 0) { g_runtime_up_at_load = 1; break; } }
		}
		closedir(d);
	}
	setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

extern "C" {

const char *cvx_last_error(void) { return g_err.c_str(); }

int cvx_abi_version(void) { return CVX_ABI_VERSION; }

#ifndef CVX_BUILD_ID
#define CVX_BUILD_ID "unknown"
#endif
const char *cvx_build_id(void) { return CVX_BUILD_ID; }
#ifndef CVX_FILL_ID
#define CVX_FILL_ID "unknown"
#endif
#ifndef CVX_SEARCH_ID
#define CVX_SEARCH_ID "unknown"
#endif
const char *cvx_source_id(const char *family) {
	if (family && strcmp(family, "fill") == 0) return CVX_FILL_ID;
	if (family && strcmp(family, "search") == 0) return CVX_SEARCH_ID;
	return CVX_BUILD_ID;
}

int cvx_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int cvx_device_synchronize(int device_id) {
	HIP_TRY(hipSetDevice(device_id));
	HIP_TRY(hipDeviceSynchronize());
	return CVX_OK;
}

int cvx_create(int device_id, const cvx_params *p, uint64_t max_matrix_mb, cvx_handle *out) {
	return cvx_create_ex(device_id, p, max_matrix_mb, 0u, out);
}

int cvx_runtime_regime(int device_id, cvx_regime *out) {
	if (!out || device_id < 0) { set_err("cvx_runtime_regime: bad argument"); return CVX_ERR_ARG; }
	memset(out, 0, sizeof(*out));
	const char *q = getenv("GPU_MAX_HW_QUEUES");
	out->hw_queues_env = q ? atoi(q) : 0;
	out->hw_queues_set_by_library = g_hwq_set_by_library;
	const int b = g_blocking_sync[device_id & 63].load();
	out->blocking_sync = b == 1 ? 1 : (b == 0 ? -1 : 0);
	out->blocking_sync_why = b;
	int shared = 4;
	if (const char *e2 = getenv("CVX_SERVICE_STREAMS")) shared = atoi(e2) > 0 ? std::min(atoi(e2), kServiceStreamsMax) : 0;
	out->service_streams = shared;
	out->runtime_up_at_load = g_runtime_up_at_load;
	return CVX_OK;
}

int cvx_create_ex(int device_id, const cvx_params *p, uint64_t max_matrix_mb, uint32_t flags, cvx_handle *out) {
	ABI_GUARD_BEGIN
	if (!p || !out) { set_err("cvx_create: NULL argument"); return CVX_ERR_ARG; }
	*out = nullptr;
	/* The ring kernels implement the scalar recurrence (reference src/ConvexAlignFast.cpp:606-774)
	 * and use its sign structure (match > 0 > every penalty, extension never dearer than its floor).
	 * The SSE path the reference actually runs is identical to that recurrence only while opening a
	 * gap directly off the other gap type can never win: gap_open + gap_ext_min < mismatch (SURVEY.md
	 * Appendix A; the 0.25 keeps float rounding out of the argument).  Any other scoring -- the
	 * reference accepts whatever --match/--mismatch/--gap-* the user passes -- is served by the
	 * catch-all kernel's SSE-variant instantiation, which restates fwdFillMatrixSSESimple cell by cell
	 * (cvx_generic.hip): slower per cell, same results as the reference. */
	const bool finite = std::isfinite(p->match) && std::isfinite(p->mismatch) && std::isfinite(p->gap_open) &&
			std::isfinite(p->gap_extend) && std::isfinite(p->gap_extend_min) && std::isfinite(p->gap_decay);
	if (!finite) {
		set_err("cvx_create: scoring (%g,%g,%g,%g,%g,%g) is not finite",
				p->match, p->mismatch, p->gap_open, p->gap_extend, p->gap_extend_min, p->gap_decay);
		return CVX_ERR_PARAMS;
	}
	const bool fast_regime = p->match > 0.0f && p->mismatch < 0.0f && p->gap_open < 0.0f &&
			p->gap_extend < 0.0f && p->gap_extend_min < 0.0f && p->gap_decay >= 0.0f &&
			p->gap_extend <= p->gap_extend_min &&
			(p->gap_open + p->gap_extend_min) < p->mismatch - 0.25f;
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
	/* How a host thread waits for this device.  The runtime's default is to spin: hipStreamSynchronize and hipEventSynchronize
	 * burn a core for as long as they wait -- an event created with hipEventBlockingSync included (tools/wait_probe.hip,
	 * profiles/r04_wait_probe.txt: 50.0 ms of thread CPU per 50 ms wait in every form).  Inside ngmlr that is a core per CS
	 * thread parked in a search or scoring call and one for the dispatcher, on a host that needs its cores for the reads'
	 * own code.  With the device in blocking-sync mode the same waits cost 0.5 ms of CPU per 50 ms and return 0.04 ms later.
	 * CVX_WAIT=spin keeps the runtime's default. */
	{
		/* once per device and process, by whichever thread gets here first (the others wait in call_once): the flag is a
		 * process-wide property of the device, documented as such in cvx_align.h; a runtime that refuses it (a context that is
		 * already active with another policy) is reported, the waits then spin */
		static std::once_flag once[64];
		std::call_once(once[device_id & 63], [&] {
			if (g_runtime_up_at_load && g_hwq_set_by_library)
				fprintf(stderr, "cvx_create: a HIP runtime was already initialised when libcvxalign.so was loaded: GPU_MAX_HW_QUEUES=16 came too late for it -- "
						"export it before the process starts, or mixed launches run their fill classes one after the other (cvx_runtime_regime)\n");
			const char *w = getenv("CVX_WAIT");
			if (w && strcmp(w, "spin") == 0) { g_blocking_sync[device_id & 63] = 3; return; }
			const hipError_t fe = hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
			g_blocking_sync[device_id & 63] = fe == hipSuccess ? 1 : 2;
			if (fe != hipSuccess) {
				(void) hipGetLastError();
				fprintf(stderr, "cvx_create: device %d stays in its current scheduling mode (%s): host threads that wait for it will spin\n", device_id, hipGetErrorString(fe));
			}
		});
	}
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
	c->sse_variant = !fast_regime;
	if (const char *e = getenv("CVX_TUNE_BT_GROUP")) c->bt_group = atoi(e);
	if (const char *e = getenv("CVX_TUNE_OVERLAP_POST")) c->overlap_post = atoi(e) != 0;
	if (const char *e = getenv("CVX_TUNE_SSE_VARIANT")) c->sse_variant = c->sse_variant || atoi(e) != 0;   /* test knob */
	c->pack_threads = PackPool::get().size();      /* the process's shared pack threads (CVX_PACK_THREADS) */
	if (const char *e = getenv("CVX_TUNE_MIN_M")) c->tune_min_slots = atoi(e);
	if (const char *e = getenv("CVX_TUNE_FORCE_WRAP16")) c->tune_force_wrap = atoi(e);
	if (const char *e = getenv("CVX_TUNE_PEN_TABLE")) c->tune_pen_table = atoi(e) != 0;
	{
		/* the table form needs the penalty min(gem, gext + run * decay) to be constant from run kPenClamp on.  With decay >= 0 the sum
		 * never shrinks as the run grows (binary32 multiply and add are monotone), so that is the case exactly when the penalty of run
		 * kPenClamp has already reached gem, or decay is 0 (it is from run 27 on with the default scoring); anything else -- a tiny
		 * decay -- keeps the arithmetic form */
		volatile float prod = (float) kPenClamp * p->gap_decay;
		volatile float at_clamp = p->gap_extend + prod;
		const bool reached = !(at_clamp < p->gap_extend_min);
		if (!(p->gap_decay >= 0.0f) || !(reached || p->gap_decay == 0.0f)) c->tune_pen_table = 0;
	}
	if (const char *e = getenv("CVX_TUNE_FAIL_COMPUTE")) c->test_fail_compute = atoi(e);
	if (const char *e = getenv("CVX_TUNE_SCORE_NO_DIAG")) c->score_no_diag = atoi(e) != 0;
	if (const char *e = getenv("CVX_TUNE_MAX_M")) c->tune_max_slots = atoi(e);
	if (const char *e = getenv("CVX_TUNE_CHAIN_M")) c->tune_chain_m = atoi(e);
	if (const char *e = getenv("CVX_TUNE_CHAIN_PRIO")) c->tune_chain_prio = atoi(e) != 0;
	if (const char *e = getenv("CVX_TUNE_CHAIN_LDS_KB")) c->tune_chain_lds_kb = atoi(e) > 0 ? atoi(e) : 0;
	if (const char *e = getenv("CVX_TUNE_WIDE_PRIO")) c->tune_wide_prio = atoi(e);
	if (const char *e = getenv("CVX_TUNE_EXACT_STEPS")) c->tune_exact_steps = atoi(e);
	if (const char *e = getenv("CVX_TUNE_TWO_LANES")) c->single_lane = atoi(e) == 0;
	if (const char *e = getenv("CVX_TUNE_LONG_STEPS")) c->tune_long_steps = atoi(e);
	if (const char *e = getenv("CVX_TUNE_SMALL_BATCH")) c->tune_small_batch = atoi(e);
	if (const char *e = getenv("CVX_TUNE_LONG_NEED")) c->tune_long_need = atoi(e);
	if (const char *e = getenv("CVX_TUNE_GANGS")) c->tune_gangs = atoi(e) != 0;
	if (const char *e = getenv("CVX_TUNE_GANG_PRIO")) c->tune_gang_prio = atoi(e) != 0;
	if (const char *e = getenv("CVX_TUNE_LATE_MIN")) c->tune_late_min = std::max(1, atoi(e));
	if (const char *e = getenv("CVX_TUNE_LATE_SHIFT")) c->tune_late_shift = std::min(16, std::max(0, atoi(e)));
	/* One stream now; the seven others an aligning handle uses (upload, post, text, the fill classes' side streams and the second
	 * set for small jobs) are created by its first alignment call (ensure_streams).  ngmlr holds 32 handles that only ever
	 * score or search on `main`: 224 streams that were created at start-up and destroyed at exit for nothing
	 * (0.8 s between the last alignment and the process's exit, profiles/r04_timeline_e2e.txt). */
	/* A SERVICE handle (CVX_CREATE_SERVICE: ngmlr's 16 searching and 32 scoring handles) makes short, latency-bound calls from a
	 * thread that blocks in them: its stream is created at the device's highest priority, which on this runtime also means a
	 * hardware queue out of another pool than the aligning handle's fill streams' -- 50 normal streams on 16 hardware queues put
	 * fill classes behind each other and scoring kernels behind 10 ms fills (profiles/r06_e2e_launch_trace.txt). */
	hipError_t e = hipSuccess;
	c->service = (flags & CVX_CREATE_SERVICE) != 0;
	const char *sp = getenv("CVX_SERVICE_PRIO");
	int shared = 4;      /* CVX_SERVICE_STREAMS: streams per device the service handles of a process share (0 = one of its own per handle) */
	if (const char *e2 = getenv("CVX_SERVICE_STREAMS")) shared = atoi(e2) > 0 ? std::min(atoi(e2), kServiceStreamsMax) : 0;
	if (c->service && shared > 0) {
		/* ngmlr holds 32 service handles beside the aligner's ten streams, and the runtime spreads a process's streams over
		 * sixteen hardware queues: two fill classes that land on one queue run one after the other (a launch's fill was the SUM
		 * of its classes again with -t 32, profiles/r06_e2e_launch_trace.txt), a 0.3 ms scoring kernel waits behind a 10 ms fill.
		 * The service handles therefore take turns on a few shared streams -- their calls are short, every call waits on its own
		 * event -- and the process stays below sixteen streams: one hardware queue each. */
		std::lock_guard<std::mutex> lk(g_service_mtx);
		ServiceStreams &ss = g_service_streams[device_id % kMaxServiceDevices];
		const int k = ss.next++ % shared;
		if (ss.st[k] == nullptr) e = hipStreamCreateWithFlags(&ss.st[k], hipStreamNonBlocking);
		c->s_main = ss.st[k];
		c->owns_main = false;
	} else if (c->service && sp && atoi(sp) != 0) {      /* (measured: 20 000 reads map in 3.07 s with it against 2.33 s without, search calls 7.0 against 3.4 ms -- off unless asked for) */
		int prio_lo = 0, prio_hi = 0;
		(void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);     /* numerically lower = higher priority */
		e = hipStreamCreateWithPriority(&c->s_main, hipStreamNonBlocking, prio_hi);
	} else {
		e = hipStreamCreateWithFlags(&c->s_main, hipStreamNonBlocking);
	}
	if (e != hipSuccess) {
		set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
		cvx_destroy(c);
		return CVX_ERR_HIP;
	}
	*out = c;
	return CVX_OK;
	ABI_GUARD_END
}

void cvx_destroy(cvx_handle h) {
	if (!h) return;
	(void) hipSetDevice(h->device);
	(void) hipDeviceSynchronize();
	g_deferred.drain();
	if (h->s_main && h->owns_main) (void) hipStreamDestroy(h->s_main);      /* (a service handle's stream belongs to the process) */
	if (h->s_post) (void) hipStreamDestroy(h->s_post);
	if (h->s_text) (void) hipStreamDestroy(h->s_text);
	if (h->s_io) (void) hipStreamDestroy(h->s_io);
	for (auto &a : h->aux) if (a) (void) hipStreamDestroy(a);
	if (h->s_main2) (void) hipStreamDestroy(h->s_main2);
	if (h->s_post2) (void) hipStreamDestroy(h->s_post2);
	for (auto &a : h->aux2) if (a) (void) hipStreamDestroy(a);
	for (cvx_batch_s *b : h->pool) { b->release(); delete b; }
	h->pool.clear();
	/* jobs the caller never released (submitted, maybe waited for): the device is idle, free them too --
	 * their handles are dead from here on, like everything else that belonged to this context */
	for (cvx_batch_s *b : h->live) { b->release(); delete b; }
	h->live.clear();
	h->pending.clear();
	h->sc_hseq.release(); h->sc_hpairs.release(); h->sc_hout.release();
	h->sc_seq.release(); h->sc_pairs.release(); h->sc_rows.release(); h->sc_out.release();
	if (h->sc_ev0) (void) hipEventDestroy(h->sc_ev0);
	if (h->sc_done) (void) hipEventDestroy(h->sc_done);
	if (h->sc_ev1) (void) hipEventDestroy(h->sc_ev1);
	cvx_search_state_free(h->search);
	delete h;
}

/* ------------------------------------------------------------------ staged form */

int cvx_batch_upload(cvx_handle h, int32_t n, const cvx_tile *tiles, cvx_batch *out) {
	ABI_GUARD_BEGIN
	if (!h || !out || n < 0 || (n > 0 && !tiles)) { set_err("cvx_batch_upload: bad argument"); return CVX_ERR_ARG; }
	*out = nullptr;
	HIP_TRY(hipSetDevice(h->device));
	cvx_batch_s *b = acquire_batch(h);
	if (!b) return CVX_ERR_OOM;
	int rc = stage_upload(h, b, n, tiles);
	if (rc == CVX_OK) {
		hipError_t e = hipStreamSynchronize(h->s_io);     /* inputs resident when this returns */
		if (e != hipSuccess) { set_err("upload copy failed: %s", hipGetErrorString(e)); rc = CVX_ERR_HIP; }
	}
	if (rc != CVX_OK) { discard_batch(h, b); return rc; }
	*out = b;
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_batch_run(cvx_handle h, cvx_batch b) {
	ABI_GUARD_BEGIN
	if (!h || !b || b->state < kUploaded) { set_err("cvx_batch_run: NULL argument / batch not uploaded"); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	RC_TRY(stage_plan(h, b, h->s_main));
	RC_TRY(stage_compute(h, b));
	RC_TRY(stage_results(h, b));
	HIP_TRY(hipStreamSynchronize(h->s_main));
	HIP_TRY(hipStreamSynchronize(h->s_post));
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_batch_timing(cvx_batch b, cvx_timing *t) {
	if (!b || !t) { set_err("cvx_batch_timing: NULL argument"); return CVX_ERR_ARG; }
	*t = b->timing;
	return CVX_OK;
}

int cvx_batch_launch_info(cvx_batch b, int32_t i, cvx_launch_info *info) {
	if (!b || !info || b->state < kFinished || i < 0 || (size_t) i >= b->launches.size()) { set_err("cvx_batch_launch_info: bad index"); return CVX_ERR_ARG; }
	*info = b->launches[(size_t) i];
	return CVX_OK;
}

int cvx_batch_ops_total(cvx_batch b, uint64_t *n_ops) {
	if (!b || !n_ops || b->state < kFinished) { set_err("cvx_batch_ops_total: batch not run"); return CVX_ERR_ARG; }
	*n_ops = b->ops_total;
	return CVX_OK;
}

int cvx_batch_download(cvx_handle h, cvx_batch b, cvx_result *results, uint32_t *ops_arena,
		uint64_t ops_capacity, uint64_t *ops_used) {
	ABI_GUARD_BEGIN
	if (!h || !b || b->state < kFinished || (b->n > 0 && !results)) { set_err("cvx_batch_download: bad argument / batch not run"); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	if (ops_used) *ops_used = b->ops_total;
	static_assert(sizeof(cvx_result) == sizeof(ResultRec), "ResultRec mirrors cvx_result");
	if (b->n) memcpy(results, b->res(), (size_t) b->n * sizeof(cvx_result));
	if (b->ops_total > ops_capacity) {
		set_err("cvx_batch_download: ops arena too small (%llu needed, %llu given)",
				(unsigned long long) b->ops_total, (unsigned long long) ops_capacity);
		return CVX_ERR_CAPACITY;
	}
	if (b->ops_total) {
		if (!ops_arena) { set_err("cvx_batch_download: NULL ops arena"); return CVX_ERR_ARG; }
		RC_TRY(stage_ops(h, b));
		memcpy(ops_arena, b->h_ops.p, (size_t) b->ops_total * sizeof(uint32_t));
	}
	return CVX_OK;
	ABI_GUARD_END
}

void cvx_batch_free(cvx_handle h, cvx_batch b) {
	if (!b) return;
	if (h) (void) hipSetDevice(h->device);
	recycle_batch(h, b);
}

int cvx_align_batch(cvx_handle h, int32_t n, const cvx_tile *tiles, cvx_result *results,
		uint32_t *ops_arena, uint64_t ops_capacity, uint64_t *ops_used) {
	cvx_job j = nullptr;
	int rc = cvx_submit(h, n, tiles, &j);
	if (rc != CVX_OK) return rc;
	const cvx_result *res = nullptr;
	const uint32_t *ops = nullptr;
	uint64_t n_ops = 0;
	rc = cvx_wait(h, j, &res, &ops, &n_ops);
	if (rc == CVX_OK) {
		if (ops_used) *ops_used = n_ops;
		if (n && results) memcpy(results, res, (size_t) n * sizeof(cvx_result));
		if (n && !results) { set_err("cvx_align_batch: NULL results"); rc = CVX_ERR_ARG; }
		else if (n_ops > ops_capacity) {
			set_err("cvx_align_batch: ops arena too small (%llu needed, %llu given)",
					(unsigned long long) n_ops, (unsigned long long) ops_capacity);
			rc = CVX_ERR_CAPACITY;
		} else if (n_ops) {
			if (!ops_arena) { set_err("cvx_align_batch: NULL ops arena"); rc = CVX_ERR_ARG; }
			else memcpy(ops_arena, ops, (size_t) n_ops * sizeof(uint32_t));
		}
		cvx_job_release(h, j);
	}
	return rc;
}

/* ------------------------------------------------------------------ streaming form */

static int submit_common(cvx_handle h, int32_t n, const cvx_tile *tiles, const cvx_genome_s *genome, const uint64_t *ref_position, cvx_job *out) {
	ABI_GUARD_BEGIN
	if (!h || !out || n < 0 || (n > 0 && !tiles) || (genome && n > 0 && !ref_position)) { set_err("cvx_submit: bad argument"); return CVX_ERR_ARG; }
	*out = nullptr;
	if (genome && genome->device != h->device) { set_err("cvx_submit_windows: the genome lives on device %d, the handle on %d", genome->device, h->device); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	/* first hand the device whatever is ready to run, then spend host time on packing (a job that fails there keeps
	 * its own error; this call reports only what happens to the batch being submitted) */
	static const bool trace = getenv("CVX_SUBMIT_TRACE") != nullptr;      /* where a slow cvx_submit spends its time (stderr, calls over 5 ms) */
	const auto t0 = std::chrono::steady_clock::now();
	if (h->live.empty() && !g_deferred.empty()) g_deferred.drain();      /* nothing of this handle is in flight: outgrown blocks go back now */
	(void) pump(h, false, nullptr);
	const auto t1 = std::chrono::steady_clock::now();
	cvx_batch_s *b = acquire_batch(h);
	if (!b) return CVX_ERR_OOM;
	b->bind(h->marks);
	const auto t2 = std::chrono::steady_clock::now();
	int rc = stage_upload(h, b, n, tiles, genome, ref_position);
	const auto t3 = std::chrono::steady_clock::now();
	if (rc == CVX_OK) rc = stage_plan(h, b, h->s_io);
	if (rc != CVX_OK) { discard_batch(h, b); return rc; }
	b->in_flight = true;
	h->pending.push_back(b);
	h->live.push_back(b);
	const auto t4 = std::chrono::steady_clock::now();
	(void) pump(h, false, nullptr);
	if (trace) {
		const auto t5 = std::chrono::steady_clock::now();
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
		if (ms(t0, t5) > 5.0) fprintf(stderr, "cvx_submit: %d tiles in %.2f ms: pump %.2f, batch slot %.2f, upload stage %.2f, plan stage %.2f, pump %.2f\n", n, ms(t0, t5), ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5));
	}
	*out = b;
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_submit(cvx_handle h, int32_t n, const cvx_tile *tiles, cvx_job *out) {
	return submit_common(h, n, tiles, nullptr, nullptr, out);
}

int cvx_submit_windows(cvx_handle h, cvx_genome g, int32_t n, const cvx_tile *tiles, const uint64_t *ref_position, cvx_job *out) {
	if (!g) { set_err("cvx_submit_windows: NULL genome"); return CVX_ERR_ARG; }
	return submit_common(h, n, tiles, g, ref_position, out);
}

int cvx_wait(cvx_handle h, cvx_job j, const cvx_result **results, const uint32_t **ops, uint64_t *n_ops) {
	ABI_GUARD_BEGIN
	if (!h || !j || !j->in_flight) { set_err("cvx_wait: not a submitted job"); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	/* A job that failed (now or in an earlier call) stays valid until cvx_job_release and keeps answering with its
	 * own error; nothing of another job is ever returned in its place. */
	if (j->state != kFailed && j->state < kComputed) (void) pump(h, true, j);
	if (j->state != kFailed && j->state < kFinished) { const int rc = stage_results(h, j); if (rc != CVX_OK) (void) fail_job(j, rc); }
	if (j->state != kFailed) { const int rc = stage_ops(h, j); if (rc != CVX_OK) (void) fail_job(j, rc); }
	if (j->state == kFailed) { g_err = j->fail_msg; return j->fail_rc; }
	(void) pump(h, false, nullptr);      /* later jobs whose inputs have arrived meanwhile */
	if (results) *results = reinterpret_cast<const cvx_result *>(j->res());
	if (ops) *ops = j->h_ops.as<uint32_t>();
	if (n_ops) *n_ops = j->ops_total;
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_job_window_refs(cvx_handle h, cvx_job j, const char **refs) {
	ABI_GUARD_BEGIN
	if (!h || !j || !refs || j->state < kFinished) { set_err("cvx_job_window_refs: job not finished (call cvx_wait first)"); return CVX_ERR_ARG; }
	if (!j->have_refs) { set_err("cvx_job_window_refs: not a job of cvx_submit_windows"); return CVX_ERR_ARG; }
	const char *base = j->h_refs.as<char>();
	const TileIn *tin = j->tin();
	for (int i = 0; i < j->n; ++i) refs[i] = base + (tin[i].ref_off - j->refs_base);
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_job_poll(cvx_handle h, cvx_job j, int32_t *done) {
	ABI_GUARD_BEGIN
	if (!h || !j || !j->in_flight || !done) { set_err("cvx_job_poll: not a submitted job"); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	(void) pump(h, false, nullptr);          /* queue the kernels of whatever has its corridor plans back */
	*done = 0;
	if (j->state == kFailed || j->state >= kFinished) { *done = 1; return CVX_OK; }
	if (j->state >= kComputed) {
		hipError_t q = hipEventQuery(j->ev_res);
		if (q == hipSuccess) *done = 1;
		else if (q == hipErrorNotReady) (void) hipGetLastError();
		else { set_err("hipEventQuery: %s", hipGetErrorString(q)); return CVX_ERR_HIP; }
	}
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_job_timing(cvx_job j, cvx_timing *t) { return cvx_batch_timing(j, t); }
int cvx_job_launch_info(cvx_job j, int32_t i, cvx_launch_info *info) { return cvx_batch_launch_info(j, i, info); }

void cvx_job_release(cvx_handle h, cvx_job j) {
	if (!j) return;
	if (h) {
		(void) hipSetDevice(h->device);
		auto it = std::find(h->pending.begin(), h->pending.end(), j);
		if (it != h->pending.end()) h->pending.erase(it);
		if ((j->state >= kPlanned && j->state < kFinished) || j->state == kFailed) (void) hipDeviceSynchronize();   /* released without waiting */
	}
	recycle_batch(h, j);
}

/* ------------------------------------------------------------------ page-locked arenas, corridor rows, host-side probe */

int cvx_host_alloc(uint64_t bytes, void **out) {
	ABI_GUARD_BEGIN
	if (!out) { set_err("cvx_host_alloc: NULL argument"); return CVX_ERR_ARG; }
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void) hipGetLastError(); set_err("cvx_host_alloc: no HIP device"); return CVX_ERR_NO_DEVICE; }
	void *p = nullptr;
	const size_t want = (size_t) bytes + 4096;        /* slack: block copies are rounded up to whole dwords */
	hipError_t e = hipHostMalloc(&p, want, hipHostMallocPortable);
	if (e != hipSuccess) { (void) hipGetLastError(); set_err("hipHostMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e)); return CVX_ERR_OOM; }
	{
		std::lock_guard<std::mutex> lk(g_pin_mtx);
		g_pin_blocks.emplace_back(static_cast<const char *>(p), want);
	}
	*out = p;
	return CVX_OK;
	ABI_GUARD_END
}

void cvx_host_free(void *p) {
	if (!p) return;
	{
		std::lock_guard<std::mutex> lk(g_pin_mtx);
		for (size_t i = 0; i < g_pin_blocks.size(); ++i)
			if (g_pin_blocks[i].first == p) { g_pin_blocks.erase(g_pin_blocks.begin() + (long) i); break; }
	}
	(void) hipHostFree(p);
}

int cvx_corridor_rows(cvx_handle h, const cvx_tile *tile, int32_t *offset, int32_t *length) {
	ABI_GUARD_BEGIN
	if (!h || !tile || tile->qry_len < 0 || (tile->qry_len > 0 && (!offset || !length))) { set_err("cvx_corridor_rows: bad argument"); return CVX_ERR_ARG; }
	const int H = tile->qry_len;
	if (H == 0) return CVX_OK;
	if (tile->corridor_kind == CVX_CORRIDOR_ROWS) {
		if (!tile->row_offset || !tile->row_length || (tile->row_stride_bytes & 3) || tile->row_stride_bytes < 4) { set_err("cvx_corridor_rows: bad row arrays"); return CVX_ERR_ARG; }
		for (int y = 0; y < H; ++y) {
			memcpy(&offset[y], (const char *) tile->row_offset + (size_t) y * (size_t) tile->row_stride_bytes, 4);
			memcpy(&length[y], (const char *) tile->row_length + (size_t) y * (size_t) tile->row_stride_bytes, 4);
		}
		return CVX_OK;
	}
	/* the closed forms are evaluated as the product evaluates them: on the device, by the function every kernel uses for
	 * the rows of such a tile (affine_row_offset; here through expand_rows_kernel, which writes them out) */
	cvx_tile t = *tile;
	static const char dummy[1] = {0};
	t.ref = t.qry = dummy;      /* only the corridor matters here */
	t.ref_len = 0;
	std::vector<TileIn> tin;
	UploadLayout L;
	int bad = -1;
	{
		cvx_tile probe = t;      /* validate the descriptor (its bounds depend on the number of rows); no sequence is read */
		if (upload_layout(1, &probe, tin, L, &bad, false) != kLayoutOk) { set_err("cvx_corridor_rows: malformed corridor descriptor"); return CVX_ERR_ARG; }
	}
	HIP_TRY(hipSetDevice(h->device));
	RowSrc rs = L.rsrc[0];
	TileIn ti;
	memset(&ti, 0, sizeof(ti));
	ti.H = H;
	DevBuf<RowSrc> d_rs;
	DevBuf<TileIn> d_ti;
	DevBuf<RowDesc> d_rows;
	int rc = d_rs.ensure(1);
	if (rc == CVX_OK) rc = d_ti.ensure(1);
	if (rc == CVX_OK) rc = d_rows.ensure((size_t) H);
	std::vector<RowDesc> rows((size_t) H);
	hipError_t e = hipSuccess;
	if (rc == CVX_OK) {
		e = hipMemcpy(d_rs.p, &rs, sizeof(rs), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(d_ti.p, &ti, sizeof(ti), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = launch_expand_rows(d_rs.p, d_ti.p, nullptr, nullptr, d_rows.p, 1, true, h->s_main);
		if (e == hipSuccess) e = hipStreamSynchronize(h->s_main);
		if (e == hipSuccess) e = hipMemcpy(rows.data(), d_rows.p, (size_t) H * sizeof(RowDesc), hipMemcpyDeviceToHost);
	}
	d_rs.release(); d_ti.release(); d_rows.release();
	if (rc != CVX_OK) return rc;
	if (e != hipSuccess) { set_err("cvx_corridor_rows: %s", hipGetErrorString(e)); return CVX_ERR_HIP; }
	for (int y = 0; y < H; ++y) { offset[y] = rows[(size_t) y].off; length[y] = rows[(size_t) y].len; }
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_pack_probe(int32_t n, const cvx_tile *tiles, int32_t iters, int32_t assume_page_locked, double *ms_per_iter, uint64_t *bytes_touched) {
	ABI_GUARD_BEGIN
	if (n < 0 || (n > 0 && !tiles) || iters <= 0 || !ms_per_iter) { set_err("cvx_pack_probe: bad argument"); return CVX_ERR_ARG; }
	/* what stage_upload does on the host, minus every HIP call: layout, packing into (ordinary) staging on the
	 * process's pack threads.  Nothing is aligned -- this measures the submit side, it computes nothing. */
	std::vector<uint8_t> hseq, hdelta;
	uint64_t touched = 0;
	const auto c0 = std::chrono::steady_clock::now();
	for (int it = 0; it < iters; ++it) {
		UploadLayout L;
		std::vector<TileIn> tin;
		int bad = -1;
		const int lrc = upload_layout(n, tiles, tin, L, &bad, false);
		if (lrc != kLayoutOk) { set_err("cvx_pack_probe: tile %d malformed / batch too large", bad); return CVX_ERR_ARG; }
		const bool zc_qry = assume_page_locked && L.qry_contig, zc_ref = assume_page_locked && L.ref_contig;
		if (hseq.size() < L.seq_total + 256) hseq.resize((size_t) L.seq_total + 256);
		if (hdelta.size() < L.delta_total + 256) hdelta.resize((size_t) L.delta_total + 256);
		uint64_t pack_work = L.delta_total * 9ull + (zc_qry ? 0 : L.qry_bytes) + (zc_ref ? 0 : L.ref_bytes);
		int threads = PackPool::get().size();
		if (pack_work < (8u << 20)) threads = 1;
		std::vector<RowOverflow> overflow((size_t) threads + 1);
		std::atomic<int> slot(0);
		if (pack_work > 0)
			parallel_ranges(n, L.wprefix, threads, [&](int bg, int en) {
				upload_pack(bg, en, tiles, tin, hseq.data(), hdelta.data(), L.rsrc, overflow[(size_t) slot.fetch_add(1)], !zc_qry, !zc_ref);
			});
		touched = 2 * ((zc_qry ? 0 : L.qry_bytes) + (zc_ref ? 0 : L.ref_bytes)) + L.delta_total * 9ull + (uint64_t) n * (sizeof(TileIn) + sizeof(RowSrc) + sizeof(cvx_tile));
	}
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
	*ms_per_iter = dt * 1e3 / iters;
	if (bytes_touched) *bytes_touched = touched;
	return CVX_OK;
	ABI_GUARD_END
}

/* ------------------------------------------------------------------ device-side text stage */

int cvx_job_text(cvx_handle h, cvx_job j, const int32_t *ext_qstart, const int32_t *ext_qend,
		cvx_alignment_text *out, uint64_t *text_off, const char **text, uint64_t *text_bytes) {
	ABI_GUARD_BEGIN
	static_assert(sizeof(TextRec) == sizeof(cvx_alignment_text), "TextRec mirrors cvx_alignment_text");
	if (!h || !j || j->state < kFinished) { set_err("cvx_job_text: job not finished (call cvx_wait first)"); return CVX_ERR_ARG; }
	const int n = j->n;
	if (n > 0 && (!out || !text_off || !text)) { set_err("cvx_job_text: NULL output"); return CVX_ERR_ARG; }
	if (text_bytes) *text_bytes = 0;
	if (n == 0) { if (text) *text = ""; return CVX_OK; }
	HIP_TRY(hipSetDevice(h->device));
	hipStream_t st = h->s_text;
	const size_t n1 = (size_t) n;
	TextArgs a;
	a.ext_qstart = a.ext_qend = nullptr;
	if (ext_qstart || ext_qend) {
		RC_TRY(j->h_ext.ensure(2 * n1 * sizeof(int32_t)));
		RC_TRY(j->d_ext.ensure(2 * n1));
		int32_t *he = j->h_ext.as<int32_t>();
		for (int i = 0; i < n; ++i) { he[i] = ext_qstart ? ext_qstart[i] : 0; he[n1 + (size_t) i] = ext_qend ? ext_qend[i] : 0; }
		HIP_TRY(hipMemcpyAsync(j->d_ext.p, he, 2 * n1 * sizeof(int32_t), hipMemcpyHostToDevice, st));
		a.ext_qstart = j->d_ext.p;
		a.ext_qend = j->d_ext.p + n1;
	}
	RC_TRY(j->d_trec.ensure(n1));
	RC_TRY(j->d_tlen.ensure(2 * n1 + 8));
	RC_TRY(j->h_trec.ensure(n1 * sizeof(TextRec)));
	RC_TRY(j->h_toff.ensure((n1 + 1) * sizeof(unsigned long long)));
	a.seq = j->d_seq.p;
	a.tin = j->d_tin.p;
	a.trun = j->d_trun.p;
	a.tout = j->d_tout.p;
	a.ops = j->d_regions.p;
	a.recs = j->d_trec.p;
	a.text_len = j->d_tlen.p;
	a.text_off = j->d_tlen.p + n1;
	a.text_total = j->d_tlen.p + 2 * n1;
	a.text = nullptr;
	a.n_tiles = n;
	/* pass 1: lengths, fields, offsets; the total comes back with the offsets */
	HIP_TRY(launch_text_size(a, st));
	unsigned long long *hoff = j->h_toff.as<unsigned long long>();
	HIP_TRY(hipMemcpyAsync(hoff, a.text_off, (n1 + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(j->h_trec.p, j->d_trec.p, n1 * sizeof(TextRec), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	const unsigned long long total = hoff[n1];
	/* pass 2: the strings */
	RC_TRY(j->d_text.ensure((size_t) total + 64));
	RC_TRY(j->h_text.ensure((size_t) total + 64));
	a.text = j->d_text.p;
	HIP_TRY(launch_text_write(a, st));
	HIP_TRY(hipMemcpyAsync(j->h_text.p, j->d_text.p, (size_t) ((total + 255) / 256 * 256 <= j->d_text.cap ? (total + 255) / 256 * 256 : total), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	memcpy(out, j->h_trec.p, n1 * sizeof(TextRec));
	for (int i = 0; i < n; ++i) text_off[i] = hoff[i];
	*text = j->h_text.as<char>();
	if (text_bytes) *text_bytes = total;
	j->text_done = true;
	return CVX_OK;
	ABI_GUARD_END
}

static int nm_profile_common(cvx_handle h, cvx_job j, int32_t first, int32_t count, uint64_t *entry_off,
		int32_t *triples, uint64_t cap_entries, double *kernel_ms, bool sizes_only, const int32_t **resident = nullptr);

/* ABI 9: cvx_job_text + cvx_job_nm_profile_resident of the whole job in two round trips to the device instead of four (sizes
 * of both, then strings and triples of both): what a dispatcher runs per finished launch */
int cvx_job_text_all(cvx_handle h, cvx_job j, const int32_t *ext_qstart, const int32_t *ext_qend,
		cvx_alignment_text *out, uint64_t *text_off, const char **text, uint64_t *text_bytes,
		uint64_t *nm_entry_off, const int32_t **triples) {
	ABI_GUARD_BEGIN
	if (!h || !j || j->state < kFinished) { set_err("cvx_job_text_all: job not finished (call cvx_wait first)"); return CVX_ERR_ARG; }
	const int n = j->n;
	if (!text || !triples || (n > 0 && (!out || !text_off || !nm_entry_off))) { set_err("cvx_job_text_all: NULL output"); return CVX_ERR_ARG; }
	if (text_bytes) *text_bytes = 0;
	*triples = nullptr;
	if (n == 0) { *text = ""; return CVX_OK; }
	HIP_TRY(hipSetDevice(h->device));
	hipStream_t st = h->s_text;
	const size_t n1 = (size_t) n;
	TextArgs a;
	memset(&a, 0, sizeof(a));
	if (ext_qstart || ext_qend) {
		RC_TRY(j->h_ext.ensure(2 * n1 * sizeof(int32_t)));
		RC_TRY(j->d_ext.ensure(2 * n1));
		int32_t *he = j->h_ext.as<int32_t>();
		for (int i = 0; i < n; ++i) { he[i] = ext_qstart ? ext_qstart[i] : 0; he[n1 + (size_t) i] = ext_qend ? ext_qend[i] : 0; }
		HIP_TRY(hipMemcpyAsync(j->d_ext.p, he, 2 * n1 * sizeof(int32_t), hipMemcpyHostToDevice, st));
		a.ext_qstart = j->d_ext.p;
		a.ext_qend = j->d_ext.p + n1;
	}
	RC_TRY(j->d_trec.ensure(n1));
	RC_TRY(j->d_tlen.ensure(2 * n1 + 8));
	RC_TRY(j->h_trec.ensure(n1 * sizeof(TextRec)));
	RC_TRY(j->h_toff.ensure((n1 + 1) * sizeof(unsigned long long)));
	RC_TRY(j->d_nmoff.ensure(2 * n1 + 8));
	RC_TRY(j->h_nmoff.ensure((n1 + 1) * sizeof(unsigned long long)));
	a.seq = j->d_seq.p; a.tin = j->d_tin.p; a.trun = j->d_trun.p; a.tout = j->d_tout.p; a.ops = j->d_regions.p;
	a.recs = j->d_trec.p;
	a.text_len = j->d_tlen.p; a.text_off = j->d_tlen.p + n1; a.text_total = j->d_tlen.p + 2 * n1;
	a.text = nullptr;
	a.n_tiles = n;
	/* round trip 1: lengths, fields and offsets of the strings; entry counts and offsets of the profile */
	HIP_TRY(launch_text_size(a, st));
	unsigned long long *d_len = j->d_nmoff.p, *d_off = j->d_nmoff.p + n1;
	HIP_TRY(launch_nm_offsets(a, 0, n, d_len, d_off, d_off + n1, st));
	unsigned long long *hoff = j->h_toff.as<unsigned long long>(), *hnm = j->h_nmoff.as<unsigned long long>();
	HIP_TRY(hipMemcpyAsync(hoff, a.text_off, (n1 + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hnm, d_off, (n1 + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(j->h_trec.p, j->d_trec.p, n1 * sizeof(TextRec), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	const unsigned long long total = hoff[n1], entries = hnm[n1];
	/* round trip 2: the strings and the triples */
	RC_TRY(j->d_text.ensure((size_t) total + 64));
	RC_TRY(j->h_text.ensure((size_t) total + 64));
	RC_TRY(j->d_nm.ensure(3 * (size_t) entries + 16));
	RC_TRY(j->h_nm.ensure((3 * (size_t) entries + 16) * sizeof(int32_t)));
	a.text = j->d_text.p;
	HIP_TRY(launch_text_write(a, st));
	HIP_TRY(hipMemcpyAsync(j->h_text.p, j->d_text.p, (size_t) ((total + 255) / 256 * 256 <= j->d_text.cap ? (total + 255) / 256 * 256 : total), hipMemcpyDeviceToHost, st));
	HIP_TRY(launch_nm_profile(a, 0, n, d_off, j->d_nm.p, st));
	if (entries > 0) HIP_TRY(hipMemcpyAsync(j->h_nm.p, j->d_nm.p, 3 * (size_t) entries * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	memcpy(out, j->h_trec.p, n1 * sizeof(TextRec));
	for (int i = 0; i < n; ++i) text_off[i] = hoff[i];
	for (int i = 0; i <= n; ++i) nm_entry_off[i] = hnm[i];
	*text = j->h_text.as<char>();
	*triples = j->h_nm.as<int32_t>();
	if (text_bytes) *text_bytes = total;
	j->text_done = true;
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_job_nm_profile(cvx_handle h, cvx_job j, int32_t first, int32_t count, uint64_t *entry_off,
		int32_t *triples, uint64_t cap_entries, double *kernel_ms) {
	return nm_profile_common(h, j, first, count, entry_off, triples, cap_entries, kernel_ms, false);
}

int cvx_job_nm_profile_resident(cvx_handle h, cvx_job j, int32_t first, int32_t count, uint64_t *entry_off,
		const int32_t **triples, double *kernel_ms) {
	if (!triples) { set_err("cvx_job_nm_profile_resident: NULL argument"); return CVX_ERR_ARG; }
	*triples = nullptr;
	return nm_profile_common(h, j, first, count, entry_off, nullptr, 0, kernel_ms, false, triples);
}

int cvx_job_nm_sizes(cvx_handle h, cvx_job j, int32_t first, int32_t count, uint64_t *entry_off) {
	return nm_profile_common(h, j, first, count, entry_off, nullptr, 0, nullptr, true);
}

static int nm_profile_common(cvx_handle h, cvx_job j, int32_t first, int32_t count, uint64_t *entry_off,
		int32_t *triples, uint64_t cap_entries, double *kernel_ms, bool sizes_only, const int32_t **resident) {
	ABI_GUARD_BEGIN
	if (kernel_ms) *kernel_ms = 0.0;
	if (!h || !j || j->state < kFinished) { set_err("cvx_job_nm_profile: job not finished (call cvx_wait first)"); return CVX_ERR_ARG; }
	if (!j->text_done) { set_err("cvx_job_nm_profile: call cvx_job_text first (it counts the entries)"); return CVX_ERR_ARG; }
	if (first < 0 || count < 0 || (int64_t) first + count > j->n || (count > 0 && !entry_off)) { set_err("cvx_job_nm_profile: bad tile range / NULL offsets"); return CVX_ERR_ARG; }
	if (count == 0) return CVX_OK;
	HIP_TRY(hipSetDevice(h->device));
	hipStream_t st = h->s_text;
	const size_t c1 = (size_t) count;
	RC_TRY(j->d_nmoff.ensure(2 * c1 + 8));
	RC_TRY(j->h_nmoff.ensure((c1 + 1) * sizeof(unsigned long long)));
	if (!j->ev_nm0) HIP_TRY(hipEventCreate(&j->ev_nm0));
	if (!j->ev_nm1) HIP_TRY(hipEventCreate(&j->ev_nm1));
	TextArgs a;
	memset(&a, 0, sizeof(a));
	a.seq = j->d_seq.p; a.tin = j->d_tin.p; a.trun = j->d_trun.p; a.tout = j->d_tout.p; a.ops = j->d_regions.p;
	a.recs = j->d_trec.p;
	a.n_tiles = j->n;
	unsigned long long *d_len = j->d_nmoff.p, *d_off = j->d_nmoff.p + c1;      /* the total lands right behind the offsets */
	HIP_TRY(launch_nm_offsets(a, first, count, d_len, d_off, d_off + c1, st));
	unsigned long long *hoff = j->h_nmoff.as<unsigned long long>();
	HIP_TRY(hipMemcpyAsync(hoff, d_off, (c1 + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	for (size_t i = 0; i <= c1; ++i) entry_off[i] = hoff[i];
	const unsigned long long total = hoff[c1];
	if (sizes_only) return CVX_OK;      /* the entry offsets alone: no 12-byte-per-column arena, no profile kernel */
	if (triples && total > cap_entries) { set_err("cvx_job_nm_profile: %llu entries, room for %llu", total, (unsigned long long) cap_entries); return CVX_ERR_CAPACITY; }
	RC_TRY(j->d_nm.ensure(3 * (size_t) total + 16));
	if (resident) {
		RC_TRY(j->h_nm.ensure((3 * (size_t) total + 16) * sizeof(int32_t)));
		triples = j->h_nm.as<int32_t>();
		*resident = triples;
	}
	HIP_TRY(hipEventRecord(j->ev_nm0, st));
	HIP_TRY(launch_nm_profile(a, first, count, d_off, j->d_nm.p, st));
	HIP_TRY(hipEventRecord(j->ev_nm1, st));
	if (triples && total > 0) HIP_TRY(hipMemcpyAsync(triples, j->d_nm.p, 3 * (size_t) total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (kernel_ms) { float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, j->ev_nm0, j->ev_nm1)); *kernel_ms = ms; }
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_nm_profile_ops(cvx_handle h, int32_t n, const cvx_result *results, const uint32_t *ops_arena, uint64_t ops_total,
		uint64_t *entry_off, int32_t *triples, uint64_t cap_entries) {
	ABI_GUARD_BEGIN
	if (!h || n < 0 || (n > 0 && (!results || !entry_off)) || (ops_total > 0 && !ops_arena)) { set_err("cvx_nm_profile_ops: bad argument"); return CVX_ERR_ARG; }
	if (n == 0) return CVX_OK;
	HIP_TRY(hipSetDevice(h->device));
	RC_TRY(ensure_streams(h));
	const size_t n1 = (size_t) n;
	std::vector<TileOut> tout(n1);
	std::vector<TileRun> trun(n1);
	std::vector<unsigned long long> off(n1 + 1, 0ull);
	for (size_t i = 0; i < n1; ++i) {
		const cvx_result &r = results[i];
		memset(&tout[i], 0, sizeof(TileOut));
		memset(&trun[i], 0, sizeof(TileRun));
		tout[i].status = r.status;
		tout[i].qstart = r.qstart;
		unsigned long long cnt = 0;
		if (r.status == CVX_TILE_OK) {
			if (r.n_ops < 0 || r.ops_begin + (uint64_t) r.n_ops > ops_total) { set_err("cvx_nm_profile_ops: ops of tile %zu outside the arena", i); return CVX_ERR_ARG; }
			tout[i].n_ops = r.n_ops;
			trun[i].ops_off = r.ops_begin;
			/* entries per op, as addPosition admits them (src/ConvexAlignFast.cpp:76-98) */
			long long pr = 0, pq = r.qstart;
			for (int k = 0; k < r.n_ops; ++k) {
				const uint32_t w = ops_arena[r.ops_begin + (uint64_t) k];
				const long long len = (long long) (w >> 4);
				const int type = (int) (w & 15u);
				if (type == CVX_OP_EQ || type == CVX_OP_X) {
					const long long mn = pr < pq ? pr : pq, skip = 17 - mn > 0 ? 17 - mn : 0;
					cnt += (unsigned long long) (len - skip > 0 ? len - skip : 0);
					pr += len; pq += len;
				} else if (type == CVX_OP_D) {
					if (pq > 16) { const long long skip = 17 - pr > 0 ? 17 - pr : 0; cnt += (unsigned long long) (len - skip > 0 ? len - skip : 0); }
					pr += len;
				} else if (type == CVX_OP_I) {
					pq += len;
				} else { set_err("cvx_nm_profile_ops: op code %d in tile %zu", type, i); return CVX_ERR_ARG; }
			}
		}
		off[i + 1] = off[i] + cnt;
	}
	for (size_t i = 0; i <= n1; ++i) entry_off[i] = off[i];
	const unsigned long long total = off[n1];
	if (!triples) return CVX_OK;                      /* sizes only */
	if (total > cap_entries) { set_err("cvx_nm_profile_ops: %llu entries, room for %llu", total, (unsigned long long) cap_entries); return CVX_ERR_CAPACITY; }
	DevBuf<int32_t> d_ops, d_tri;
	DevBuf<TileOut> d_tout;
	DevBuf<TileRun> d_trun;
	DevBuf<unsigned long long> d_off;
	int rc = CVX_OK;
	auto body = [&]() -> int {
		RC_TRY(d_ops.ensure((size_t) ops_total + 16));
		RC_TRY(d_tri.ensure(3 * (size_t) total + 16));
		RC_TRY(d_tout.ensure(n1));
		RC_TRY(d_trun.ensure(n1));
		RC_TRY(d_off.ensure(n1 + 1));
		hipStream_t st = h->s_text;
		if (ops_total) HIP_TRY(hipMemcpyAsync(d_ops.p, ops_arena, (size_t) ops_total * sizeof(uint32_t), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(d_tout.p, tout.data(), n1 * sizeof(TileOut), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(d_trun.p, trun.data(), n1 * sizeof(TileRun), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), (n1 + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
		TextArgs a;
		memset(&a, 0, sizeof(a));
		a.tout = d_tout.p; a.trun = d_trun.p; a.ops = d_ops.p; a.n_tiles = n;
		HIP_TRY(launch_nm_profile(a, 0, n, d_off.p, d_tri.p, st));
		if (total) HIP_TRY(hipMemcpyAsync(triples, d_tri.p, 3 * (size_t) total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		return CVX_OK;
	};
	rc = body();
	d_ops.release(); d_tri.release(); d_tout.release(); d_trun.release(); d_off.release();
	return rc;
	ABI_GUARD_END
}

/* ------------------------------------------------------------------ resident genome (decode half) */

int cvx_genome_upload(cvx_handle h, const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table, int32_t n_starts, cvx_genome *out) {
	ABI_GUARD_BEGIN
	if (!h || !out || !bin_ref || !start_table || n_starts < 2 || n_nibbles < 2) { set_err("cvx_genome_upload: bad argument"); return CVX_ERR_ARG; }
	*out = nullptr;
	for (int32_t i = 1; i < n_starts; ++i)
		if (start_table[i] <= start_table[i - 1]) { set_err("cvx_genome_upload: start table not ascending at %d", i); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(h->device));
	cvx_genome_s *g = new (std::nothrow) cvx_genome_s();
	if (!g) return CVX_ERR_OOM;
	g->device = h->device;
	g->n_nibbles = n_nibbles;
	g->n_starts = n_starts;
	/* a window may start or end in the last spacer: decode() reads up to a byte past the last nibble pair */
	const size_t bytes = (size_t) ((n_nibbles + 1) / 2);
	int rc = g->d_bin.ensure(bytes + 64);
	if (rc == CVX_OK) rc = g->d_starts.ensure((size_t) n_starts);
	if (rc != CVX_OK) { g->d_bin.release(); g->d_starts.release(); delete g; return rc; }
	hipError_t e = hipMemset(g->d_bin.p, 0x44, g->d_bin.cap);        /* beyond the genome: N */
	if (e == hipSuccess) e = hipMemcpy(g->d_bin.p, bin_ref, bytes, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(g->d_starts.p, start_table, (size_t) n_starts * sizeof(uint64_t), hipMemcpyHostToDevice);
	if (e != hipSuccess) {
		set_err("cvx_genome_upload: %s", hipGetErrorString(e));
		g->d_bin.release(); g->d_starts.release(); delete g;
		return CVX_ERR_HIP;
	}
	*out = g;
	return CVX_OK;
	ABI_GUARD_END
}

void cvx_genome_free(cvx_handle h, cvx_genome g) {
	if (!g) return;
	if (h) { (void) hipSetDevice(h->device); (void) hipDeviceSynchronize(); }
	g->d_bin.release();
	g->d_starts.release();
	delete g;
}

int cvx_genome_decode(cvx_handle h, cvx_genome g, int32_t n, const uint64_t *position, const int32_t *length,
		const uint64_t *out_offset, char *out) {
	ABI_GUARD_BEGIN
	if (!h || !g || n < 0 || (n > 0 && (!position || !length || !out_offset || !out))) { set_err("cvx_genome_decode: bad argument"); return CVX_ERR_ARG; }
	if (n == 0) return CVX_OK;
	HIP_TRY(hipSetDevice(h->device));
	std::vector<WindowDesc> win((size_t) n);
	uint64_t total = 0;
	for (int i = 0; i < n; ++i) {
		if (length[i] < 1) { set_err("cvx_genome_decode: window %d has length %d", i, length[i]); return CVX_ERR_ARG; }
		win[(size_t) i].position = position[i];
		win[(size_t) i].dst_off = total;
		win[(size_t) i].n_chars = (int64_t) length[i] - 1;           /* the reference's last byte is the NUL */
		total += (uint64_t) length[i];
	}
	DevBuf<uint8_t> d_out;
	DevBuf<WindowDesc> d_win;
	RC_TRY(d_out.ensure((size_t) total + 64));
	int rc = d_win.ensure((size_t) n);
	if (rc != CVX_OK) { d_out.release(); return rc; }
	std::vector<uint8_t> host((size_t) total);
	hipStream_t st = h->s_main;
	hipError_t e = hipMemsetAsync(d_out.p, 0, (size_t) total, st);   /* the NUL that ends every window */
	if (e == hipSuccess) e = hipMemcpyAsync(d_win.p, win.data(), (size_t) n * sizeof(WindowDesc), hipMemcpyHostToDevice, st);
	hipEvent_t k0 = nullptr, k1 = nullptr;      /* the kernel alone, on its own stream */
	if (e == hipSuccess) e = hipEventCreate(&k0);
	if (e == hipSuccess) e = hipEventCreate(&k1);
	if (e == hipSuccess) e = hipEventRecord(k0, st);
	if (e == hipSuccess) e = launch_decode_windows(g->d_bin.p, g->d_starts.p, g->n_starts, d_win.p, n, d_out.p, st);
	if (e == hipSuccess) e = hipEventRecord(k1, st);
	if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d_out.p, (size_t) total, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e == hipSuccess) h->decode_kernel_ms = ev_ms(k0, k1);
	if (k0) (void) hipEventDestroy(k0);
	if (k1) (void) hipEventDestroy(k1);
	d_out.release();
	d_win.release();
	if (e != hipSuccess) { set_err("cvx_genome_decode: %s", hipGetErrorString(e)); return CVX_ERR_HIP; }
	for (int i = 0; i < n; ++i) memcpy(out + out_offset[i], host.data() + win[(size_t) i].dst_off, (size_t) length[i]);
	return CVX_OK;
	ABI_GUARD_END
}

/* ------------------------------------------------------------------ candidate search (search half) */

struct cvx_index_s {             /* one unit of ngmlr's CompactPrefixTable, resident in HBM (cvx_search.hip) */
	int device = 0;
	int32_t k = 0;
	uint64_t n_index = 0;        /* 4^k + 2 */
	uint64_t unit_offset = 0;
	uint32_t n_locs = 0;
	DevBuf<uint2> d_rows;            /* (row start, row length | used << 31) per prefix: SearchArgs::rows */
	DevBuf<uint32_t> d_locs;
};

/* a table cvx_index_build_device(CVX_INDEX_KEEP_RESIDENT) left on its device: the next cvx_index_upload of these very host
 * arrays on that device takes it over instead of reading 4^k records on the host and copying a gigabyte back up */
struct ResidentTable {
	std::mutex mtx;
	int device = -1;
	int32_t k = 0;
	const void *index = nullptr;
	const uint32_t *locs = nullptr;
	uint64_t n_locs = 0;
	void *d_rows = nullptr;
	uint32_t *d_locs = nullptr;
	void drop() {      /* (mtx held) */
		if (device >= 0 && (d_rows || d_locs)) { (void) hipSetDevice(device); if (d_rows) (void) hipFree(d_rows); if (d_locs) (void) hipFree(d_locs); }
		device = -1; d_rows = nullptr; d_locs = nullptr; index = nullptr; locs = nullptr;
	}
} g_resident_table;

int cvx_index_upload(cvx_handle h, int32_t k, const void *index, const uint32_t *locs, uint32_t n_locs, uint64_t unit_offset, cvx_index *out) {
	ABI_GUARD_BEGIN
	if (!h || !out || !index || (n_locs > 0 && !locs) || k < 4 || k > 15) { set_err("cvx_index_upload: bad argument (kmer_len 4..15)"); return CVX_ERR_ARG; }
	*out = nullptr;
	HIP_TRY(hipSetDevice(h->device));
	const uint64_t n_index = (1ull << (2 * k)) + 2ull;
	{
		std::lock_guard<std::mutex> lk(g_resident_table.mtx);
		ResidentTable &r = g_resident_table;
		if (r.device == h->device && r.k == k && r.index == index && r.locs == locs && r.n_locs == n_locs && r.d_rows && r.d_locs) {
			cvx_index_s *ix = new (std::nothrow) cvx_index_s();
			if (!ix) return CVX_ERR_OOM;
			ix->device = h->device; ix->k = k; ix->n_index = n_index; ix->unit_offset = unit_offset; ix->n_locs = n_locs;
			ix->d_rows.p = static_cast<uint2 *>(r.d_rows); ix->d_rows.cap = (size_t) n_index - 1;
			ix->d_locs.p = r.d_locs; ix->d_locs.cap = (size_t) n_locs + 1;
			r.device = -1; r.d_rows = nullptr; r.d_locs = nullptr; r.index = nullptr; r.locs = nullptr;
			*out = ix;
			return CVX_OK;
		}
	}
	/* the 5-byte Index records (uint m_TabIndex; char m_RevCompIndex, #pragma pack(1): src/PrefixTable.h:15-31) as one 8-byte
	 * record per prefix: where GetRefEntry's row starts (m_TabIndex - 1), how long it is (the next record's m_TabIndex - this
	 * one's, PrefixTable.cpp:476-532) and Index::used() */
	const uint8_t *p = static_cast<const uint8_t *>(index);
	auto tab_at = [&](uint64_t i) { uint32_t t; memcpy(&t, p + 5 * i, 4); return t; };
	std::vector<uint2> rows;
	try { rows.resize((size_t) n_index - 1); } catch (const std::bad_alloc &) { set_err("cvx_index_upload: out of host memory"); return CVX_ERR_OOM; }
	for (uint64_t i = 0; i + 1 < n_index; ++i) {
		const uint32_t t0 = tab_at(i), t1 = tab_at(i + 1);
		const bool used = p[5 * i + 4] != 0;                  /* Index::used() */
		if (used && (t0 == 0 || t1 < t0 || (uint64_t) t1 - 1 > n_locs || t1 - t0 > 0x7FFFFFFFu)) {
			set_err("cvx_index_upload: index entry %llu points outside the %u locations", (unsigned long long) i, n_locs);
			return CVX_ERR_ARG;
		}
		rows[(size_t) i] = make_uint2(used ? t0 - 1u : 0u, used ? ((t1 - t0) | 0x80000000u) : 0u);
	}
	cvx_index_s *ix = new (std::nothrow) cvx_index_s();
	if (!ix) return CVX_ERR_OOM;
	ix->device = h->device; ix->k = k; ix->n_index = n_index; ix->unit_offset = unit_offset; ix->n_locs = n_locs;
	int rc = ix->d_rows.ensure(rows.size());
	if (rc == CVX_OK) rc = ix->d_locs.ensure((size_t) n_locs + 1);
	hipError_t e = hipSuccess;
	if (rc == CVX_OK) {
		e = hipMemcpy(ix->d_rows.p, rows.data(), rows.size() * sizeof(uint2), hipMemcpyHostToDevice);
		if (e == hipSuccess && n_locs) e = hipMemcpy(ix->d_locs.p, locs, (size_t) n_locs * 4, hipMemcpyHostToDevice);
	}
	if (rc != CVX_OK || e != hipSuccess) {
		if (e != hipSuccess) { set_err("cvx_index_upload: %s", hipGetErrorString(e)); rc = CVX_ERR_HIP; }
		ix->d_rows.release(); ix->d_locs.release(); delete ix;
		return rc;
	}
	*out = ix;
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_index_build_device(int32_t device_id, const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table, const uint64_t *seq_lengths, int32_t n_seqs,
		int32_t kmer_len, int32_t ref_skip, int32_t bin_shift, void *ref_table_index, uint32_t *ref_table, uint64_t ref_table_capacity,
		uint64_t *n_locations, uint32_t flags) {
	ABI_GUARD_BEGIN
	if (!bin_ref || !start_table || !seq_lengths || n_seqs <= 0 || kmer_len < 4 || kmer_len > 15 || ref_skip < 0 || bin_shift < 0 || bin_shift > 30 ||
			!ref_table_index || !n_locations) { set_err("cvx_index_build_device: bad argument (kmer_len 4..15, bin_shift 0..30)"); return CVX_ERR_ARG; }
	for (int32_t s = 0; s < n_seqs; ++s) {
		if ((start_table[s] & 1ull) || start_table[s] + seq_lengths[s] > n_nibbles) { set_err("cvx_index_build_device: sequence %d does not start on a byte or leaves the genome", s); return CVX_ERR_ARG; }
		if (s > 0 && start_table[s] < start_table[s - 1]) { set_err("cvx_index_build_device: sequences out of order (a row lists its locations in walk order)"); return CVX_ERR_ARG; }
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void) hipGetLastError(); set_err("cvx_index_build_device: no HIP device"); return CVX_ERR_NO_DEVICE; }
	if (device_id < 0 || device_id >= ndev) { set_err("cvx_index_build_device: device %d of %d", device_id, ndev); return CVX_ERR_ARG; }
	HIP_TRY(hipSetDevice(device_id));
	hipStream_t st = nullptr;
	HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	char err[320];
	err[0] = 0;
	const bool keep = (flags & CVX_INDEX_KEEP_RESIDENT) != 0u;
	void *d_rows = nullptr;
	uint32_t *d_locs = nullptr;
	const int rc = index_build_device(bin_ref, n_nibbles, start_table, seq_lengths, n_seqs, kmer_len, ref_skip, bin_shift, ref_table_index, ref_table,
			ref_table_capacity, n_locations, st, err, sizeof(err), keep ? &d_rows : nullptr, keep ? &d_locs : nullptr);
	(void) hipStreamSynchronize(st);
	(void) hipStreamDestroy(st);
	if (rc != CVX_OK) set_err("cvx_index_build_device: %s", err);
	if (keep) {
		std::lock_guard<std::mutex> lk(g_resident_table.mtx);
		g_resident_table.drop();                                  /* one table waits at a time */
		if (rc == CVX_OK && d_rows && d_locs) {
			g_resident_table.device = device_id; g_resident_table.k = kmer_len; g_resident_table.index = ref_table_index; g_resident_table.locs = ref_table;
			g_resident_table.n_locs = *n_locations; g_resident_table.d_rows = d_rows; g_resident_table.d_locs = d_locs;
		}
	}
	return rc;
	ABI_GUARD_END
}

void cvx_index_free(cvx_handle h, cvx_index ix) {
	if (!ix) return;
	if (h) { (void) hipSetDevice(h->device); (void) hipDeviceSynchronize(); }
	ix->d_rows.release(); ix->d_locs.release();
	delete ix;
}

} /* extern "C" */

/* the reads either as n NUL-terminated strings (seqs / lens) or back to back in one block (arena / offsets: read i =
 * arena[offsets[i] .. offsets[i + 1] - 1) with a NUL at offsets[i + 1] - 1) */
static int search_common(cvx_handle h, cvx_index ix, int32_t n, const char *const *seqs, const int32_t *lens, const uint8_t *arena, const uint64_t *offsets,
		float sensitivity, float min_hits, int32_t bin_shift, int32_t first_bits,
		int32_t *n_candidates, uint64_t *cand_begin, cvx_candidate *cands, uint64_t cand_capacity, uint64_t *cand_used,
		float *max_hit, int32_t *kmer_misses) {
	ABI_GUARD_BEGIN
	static_assert(sizeof(SearchCandidate) == sizeof(cvx_candidate), "SearchCandidate mirrors cvx_candidate");
	/* bin_shift >= 1: the "bin is listed" flag lives in bit 63 of the vote table's key, and with a shift of 0 a vote near the
	 * start of the genome (location - offset in the read < 0) would set that bit itself; ngmlr's bin size is 4 */
	if (!h || !ix || n < 0 || (n > 0 && ((!arena && (!seqs || !lens)) || (arena && !offsets) || !n_candidates || !cand_begin)) || bin_shift < 1 || bin_shift > 30 ||
			(first_bits != 0 && (first_bits < 8 || first_bits > 20))) {
		set_err("cvx_search_batch: bad argument (bin_shift 1..30, first_bits 0 or 8..20)"); return CVX_ERR_ARG;
	}
	if (ix->device != h->device) { set_err("cvx_search_batch: the index lives on device %d, the handle on %d", ix->device, h->device); return CVX_ERR_ARG; }
	if (cand_used) *cand_used = 0;
	if (n == 0) return CVX_OK;
	HIP_TRY(hipSetDevice(h->device));
	if (!h->search) { h->search = new (std::nothrow) cvx_search_state(); if (!h->search) return CVX_ERR_OOM; }
	cvx_search_state *ss = h->search;
	hipStream_t st = h->s_main;
	/* the reads, a NUL behind each (the N-run scan of PrefixIteration relies on the terminator), in page-locked staging */
	uint64_t bytes = 0;
	if (arena) {
		for (int i = 0; i < n; ++i) {
			if (offsets[i + 1] <= offsets[i] || offsets[i + 1] - offsets[i] > 0x7FFFFFFFull || arena[offsets[i + 1] - 1] != 0) {
				set_err("cvx_search_batch_arena: read %d is not [offsets[i], offsets[i + 1] - 1) followed by a NUL", i); return CVX_ERR_ARG;
			}
		}
		bytes = offsets[n] - offsets[0];
	} else {
		for (int i = 0; i < n; ++i) {
			if (!seqs[i] || lens[i] < 0) { set_err("cvx_search_batch: bad read %d", i); return CVX_ERR_ARG; }
			bytes += (uint64_t) lens[i] + 1;
		}
	}
	const size_t n1 = (size_t) n;
	/* a block that lies in memory from cvx_host_alloc travels as it is: the host touches no base (the 64 bytes behind the last read
	 * that the kernels may read are cleared on the device) */
	const uint8_t *src = arena ? arena + offsets[0] : nullptr;
	const bool zero_copy = arena && in_pinned_block(reinterpret_cast<const char *>(src), bytes + 4);
	if (!zero_copy) RC_TRY(ss->h_seq.ensure((size_t) bytes + 64));
	/* meta: [off u64 n][list_off u64 n][begin u64 n][len i32 n][work i32 n] */
	RC_TRY(ss->h_meta.ensure(n1 * (8 + 8 + 8 + 4 + 4) + 64));
	/* out: [events u64 n][ncand i32 n][miss i32 n][maxhit f32 n] (the dense candidates get their own buffer below) */
	RC_TRY(ss->h_out.ensure(n1 * (8 + 4 + 4 + 4) + 64));
	uint8_t *hseq = zero_copy ? nullptr : ss->h_seq.as<uint8_t>();
	uint64_t *h_off = ss->h_meta.as<uint64_t>(), *h_listoff = h_off + n1, *h_begin = h_listoff + n1;
	int32_t *h_len = reinterpret_cast<int32_t *>(h_begin + n1), *h_work = h_len + n1;
	unsigned long long *h_events = ss->h_out.as<unsigned long long>();
	int32_t *h_ncand = reinterpret_cast<int32_t *>(h_events + n1), *h_miss = h_ncand + n1;
	float *h_maxhit = reinterpret_cast<float *>(h_miss + n1);
	if (arena) {
		for (int i = 0; i < n; ++i) { h_off[i] = offsets[i] - offsets[0]; h_len[i] = (int32_t) (offsets[i + 1] - offsets[i] - 1); }
		if (!zero_copy) { memcpy(hseq, src, (size_t) bytes); memset(hseq + bytes, 0, 64); }
	} else {
		uint64_t at = 0;
		for (int i = 0; i < n; ++i) {
			h_off[i] = at;
			memcpy(hseq + at, seqs[i], (size_t) lens[i]);
			hseq[at + (uint64_t) lens[i]] = 0;
			at += (uint64_t) lens[i] + 1;
			h_len[i] = lens[i];
		}
		memset(hseq + at, 0, 64);
	}
	RC_TRY(ss->d_seq.ensure((size_t) bytes + 64));
	RC_TRY(ss->d_off.ensure(n1)); RC_TRY(ss->d_listoff.ensure(n1)); RC_TRY(ss->d_begin.ensure(n1));
	RC_TRY(ss->d_len.ensure(n1)); RC_TRY(ss->d_ncand.ensure(n1)); RC_TRY(ss->d_work.ensure(n1)); RC_TRY(ss->d_miss.ensure(n1));
	RC_TRY(ss->d_events.ensure(n1)); RC_TRY(ss->d_maxhit.ensure(n1));
	SearchArgs a;
	memset(&a, 0, sizeof(a));
	a.rows = ix->d_rows.p; a.locs = ix->d_locs.p; a.unit_offset = ix->unit_offset; a.k = ix->k;
	a.seq = ss->d_seq.p; a.seq_off = ss->d_off.p; a.seq_len = ss->d_len.p; a.n = n;
	a.events = ss->d_events.p; a.list_off = ss->d_listoff.p; a.n_cand = ss->d_ncand.p; a.max_hit = ss->d_maxhit.p; a.kmer_misses = ss->d_miss.p;
	a.sensitivity = sensitivity; a.min_hits = min_hits; a.bin_shift = bin_shift;
	if (zero_copy) {
		HIP_TRY(hipMemsetAsync(ss->d_seq.p + bytes / 4 * 4, 0, 68, st));      /* what lies behind the block, first: the copy below is rounded up to whole dwords */
		HIP_TRY(hipMemcpyAsync(ss->d_seq.p, src, (size_t) ((bytes + 3) / 4 * 4), hipMemcpyHostToDevice, st));
	} else {
		HIP_TRY(hipMemcpyAsync(ss->d_seq.p, hseq, (size_t) ((bytes + 63) / 4 * 4), hipMemcpyHostToDevice, st));
	}
	HIP_TRY(hipMemcpyAsync(ss->d_off.p, h_off, n1 * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(ss->d_len.p, h_len, n1 * 4, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemsetAsync(ss->d_miss.p, 0, n1 * 4, st));
	HIP_TRY(hipMemsetAsync(ss->d_maxhit.p, 0, n1 * 4, st));
	/* A read starts on the wave-per-read kernel with its vote map in LDS (cvx_search.hip), in the smallest map that certainly holds
	 * it: a call of CVX_TUNE_SEARCH_CLASSIFY (2 048) reads or more counts every read's votes first (one cheap kernel and one round
	 * trip) and sorts the reads into launches by map size -- 6 KB of LDS for a read of <= 384 votes, 24 KB for <= 1 536, 48 KB
	 * above -- so that as many reads as possible share a CU; a smaller call takes the 2^11-slot map for every read without
	 * counting (its reads do not fill the device anyway).  A read that does not fit its map after all (more than 3 072 bins; a bin
	 * beyond 32 bits; longer than the sequence buffer) is redone -- the same attempt, then the rest of the ladder -- by the same
	 * kernel over the real table in HBM, which needs the vote count to size its lists.  Candidates of read i lie at d_cand +
	 * src_off[i]: two per bin its map can hold (or per vote, when that is fewer) for an LDS-map read, the sparse region behind all
	 * of those (two slots per vote) for the others.  CVX_TUNE_SEARCH_WAVE=2 sends every read to the HBM-table form, =0 to the
	 * lane-per-read kernel (one serial chain of votes per read; an independent implementation: tests, A/B);
	 * CVX_TUNE_SEARCH_LOG2=9..12 forces one map size (tests: a small one sends most reads through the fall-back). */
	const char *wave_env = getenv("CVX_TUNE_SEARCH_WAVE");      /* (read per call: the parity tests switch it inside one process) */
	const int wave_mode = wave_env ? atoi(wave_env) : 1;
	const bool use_wave = wave_mode == 1;
	const bool lane_serial = wave_mode == 0;
	const char *cls_env = getenv("CVX_TUNE_SEARCH_CLASSIFY"), *log2_env = getenv("CVX_TUNE_SEARCH_LOG2"), *slot8_env = getenv("CVX_TUNE_SEARCH_SLOT8");
	const int classify_min = cls_env ? atoi(cls_env) : 2048;
	const int forced_log2 = log2_env ? std::min(kSearchWaveLog2Max, std::max(kSearchWaveLog2Min, atoi(log2_env))) : 0;
	/* (a table with three locations per k-mer and more -- a genome of 600 Mbp and up at ngmlr's defaults -- gives a 256-base read
	 * 1 500 votes and more: the one-size map of a small call would not hold it, so its reads are counted and sorted as well) */
	const bool big_table = (uint64_t) ix->n_locs >= 3ull * (ix->n_index - 2ull);
	const bool classify = use_wave && !forced_log2 && (n >= classify_min || big_table);
	bool counted = false;
	uint64_t total = 0;             /* votes of the reads that took the HBM-table form so far: their rList / candidate regions */
	std::vector<uint8_t> has_region(n1, 0);
	ss->kev_used = 0;
	ss->kernel_ms = 0.0f;
	auto count_votes = [&]() -> int {      /* votes per read (all reads: the pass is cheap): map sizes, sizes of the HBM kernel's lists */
		RC_TRY(ss->kmark(st));
		HIP_TRY(launch_search_count(a, st));
		RC_TRY(ss->kmark(st));
		HIP_TRY(hipMemcpyAsync(h_events, ss->d_events.p, n1 * 8, hipMemcpyDeviceToHost, st));
		RC_TRY(search_wait(ss, st));
		counted = true;
		return CVX_OK;
	};
	if (classify) RC_TRY(count_votes());
	std::vector<uint8_t> map_log2(n1, 0);      /* 0: not an LDS-map read */
	RC_TRY(ss->d_srcoff.ensure(n1));
	RC_TRY(ss->h_srcoff.ensure(n1 * 8));
	uint64_t *h_srcoff = ss->h_srcoff.as<uint64_t>();
	uint64_t fixed_total = 0;
	for (int i = 0; i < n; ++i) {
		h_srcoff[i] = fixed_total;
		if (!use_wave || h_len[i] + 1 > kSearchWaveSeq) continue;
		const int l = forced_log2 ? forced_log2 : classify ? search_wave_log2(h_events[i]) : kSearchWaveLog2Default;
		map_log2[(size_t) i] = (uint8_t) l;
		uint64_t bins = (uint64_t) search_wave_entries(l);
		if (classify && h_events[i] < bins) bins = h_events[i];
		fixed_total += 2 * bins;
	}
	RC_TRY(ss->d_cand.ensure((size_t) fixed_total + 64));
	HIP_TRY(hipMemcpyAsync(ss->d_srcoff.p, h_srcoff, n1 * 8, hipMemcpyHostToDevice, st));
	/* Regions only for the reads that really go to the HBM-table form, handed out when a read first gets there and kept for its
	 * retries (sizing them by the votes of all n reads cost a repeat-rich call gigabytes per handle).  The
	 * candidate arena grows without losing the lists already in it. */
	auto give_regions = [&](const std::vector<int32_t> &reads) -> int {
		bool grew = false;
		for (int32_t i : reads) {
			if (has_region[(size_t) i]) continue;
			has_region[(size_t) i] = 1;
			h_listoff[i] = total;
			total += h_events[i];
			grew = true;
		}
		if (!grew) return CVX_OK;
		RC_TRY(ss->d_rlist.ensure((size_t) total + 64));
		if (ss->d_cand.cap < (size_t) (fixed_total + 2 * total) + 64) {
			DevBuf<SearchCandidate> bigger;
			RC_TRY(bigger.ensure((size_t) (fixed_total + 2 * total) + (size_t) total / 2 + 64));
			hipError_t e = hipMemcpyAsync(bigger.p, ss->d_cand.p, ss->d_cand.cap * sizeof(SearchCandidate), hipMemcpyDeviceToDevice, st);
			int rc = CVX_OK;
			if (e != hipSuccess) { set_err("cvx_search_batch: %s", hipGetErrorString(e)); rc = CVX_ERR_HIP; }
			if (rc == CVX_OK) rc = search_wait(ss, st);
			if (rc != CVX_OK) { bigger.release(); return rc; }      /* (this buffer leaked on the error paths) */
			ss->d_cand.release();
			ss->d_cand = bigger;
		}
		HIP_TRY(hipMemcpyAsync(ss->d_listoff.p, h_listoff, n1 * 8, hipMemcpyHostToDevice, st));
		return CVX_OK;
	};
	/* the reference's retry ladder (CS.cpp:345-394): the current table size with a probe budget of a third of the table, then
	 * that + 2, + 3, ... up to 2^20 entries with 0.777; first_bits = CS::c_SrchTableBitLen (16 when the thread starts; the
	 * reference adapts it per batch, CS.cpp:482-489 -- the size only decides WHEN an attempt runs out of budget, a successful
	 * attempt returns the same list at every size). */
	const int bits0 = first_bits ? first_bits : 16;
	const bool trace = getenv("CVX_SEARCH_TRACE") != nullptr;      /* one line per call on stderr: who ran where, for how long (read per call: tests switch it on inside a process) */
	const std::chrono::steady_clock::time_point tr0 = std::chrono::steady_clock::now();
	std::string tr;
	auto tr_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(); };
	std::vector<int32_t> work_wave, work_hbm;
	for (int i = 0; i < n; ++i) (map_log2[(size_t) i] ? work_wave : work_hbm).push_back(i);
	ss->attempts.assign(n1, 0);
	for (int attempt = 0; !work_wave.empty() || !work_hbm.empty(); ++attempt) {
		const int bits = attempt == 0 ? bits0 : bits0 + 1 + attempt;
		if (bits > 20) break;
		for (int32_t i : work_wave) ss->attempts[(size_t) i] += 1;
		for (int32_t i : work_hbm) ss->attempts[(size_t) i] += 1;
		a.bits = bits;
		a.hpoc_factor = attempt == 0 ? 0.333f : 0.777f;
		std::vector<int32_t> next_wave, next_hbm, hbm_now(work_hbm);
		if (!work_wave.empty()) {
			/* one launch per map size, back to back (the work list holds the sizes' reads one after the other) */
			size_t at = 0;
			a.cand = ss->d_cand.p; a.cand_off = ss->d_srcoff.p;
			for (int l = kSearchWaveLog2Min; l <= kSearchWaveLog2Max; ++l) {
				const size_t first = at;
				int longest = 0;
				for (int32_t i : work_wave) if (map_log2[(size_t) i] == l) { h_work[at++] = i; longest = std::max(longest, h_len[i]); }
				if (at == first) continue;
				HIP_TRY(hipMemcpyAsync(ss->d_work.p + first, h_work + first, (at - first) * 4, hipMemcpyHostToDevice, st));
				a.work = ss->d_work.p + first;
				a.n_work = (int32_t) (at - first);
				RC_TRY(ss->kmark(st));
				/* the 8-byte map where it applies: the reference's default table size or below, sub-reads (a count of 255 is out of
				 * reach of a read's 256 k-mers unless a row repeats inside a bin), not the largest map; CVX_TUNE_SEARCH_SLOT8=0 / 1 */
				const bool slot8 = slot8_env ? atoi(slot8_env) != 0 && bits <= 16 && l < kSearchWaveLog2Max : (bits <= 16 && l < kSearchWaveLog2Max && longest <= 268);
				HIP_TRY(launch_search_wave(a, l, (longest + 65 + 63) / 64 * 64, slot8, st));
				RC_TRY(ss->kmark(st));
				if (trace) { char b[96]; snprintf(b, sizeof(b), " [2^%d-slot maps: %zu reads]", l, at - first); tr += b; }
			}
			a.work = ss->d_work.p;
			HIP_TRY(hipMemcpyAsync(h_ncand, ss->d_ncand.p, n1 * 4, hipMemcpyDeviceToHost, st));
			RC_TRY(search_wait(ss, st));
			for (int32_t i : work_wave) {
				if (h_ncand[i] == kSearchNeedsHbm) hbm_now.push_back(i);
				else if (h_ncand[i] < 0) next_wave.push_back(i);
			}
			if (trace) { char b[160]; snprintf(b, sizeof(b), " [bits %d wave %zu -> %zu to hbm, %zu retry, at %.2f ms]", bits, work_wave.size(), hbm_now.size() - work_hbm.size(), next_wave.size(), tr_ms()); tr += b; }
		}
		if (!hbm_now.empty()) {
			a.work = ss->d_work.p;
			if (!counted) RC_TRY(count_votes());
			RC_TRY(give_regions(hbm_now));
			for (int32_t i : hbm_now) h_srcoff[i] = fixed_total + 2 * h_listoff[i];
			const size_t per_read = (size_t) 1 << bits;
			a.rlist = ss->d_rlist.p; a.cand = ss->d_cand.p + fixed_total;
			if (!lane_serial) {
				/* the wave kernel's tables belong to the waves: as many as can be resident (or as 8 GB hold), each 2^bits entries of 16
				 * bytes, empty between launches -- a wave frees the slots a read opened when the read is done -- so they are cleared
				 * when they are allocated, when the table size changes and after a launch that failed, not per read */
				RC_TRY(ss->d_undo.ensure((size_t) total + 64));
				a.undo = ss->d_undo.p;
				const size_t n_tables = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hbm_now.size(), 8192), ((size_t) 8 << 30) / (per_read * 16)));
				const size_t need_words = n_tables * per_read * 2;
				if (ss->d_keys.cap < need_words) { RC_TRY(ss->d_keys.ensure(need_words)); ss->tables_clean_words = 0; }
				if (ss->tables_clean_words < need_words) {
					HIP_TRY(hipMemsetAsync(ss->d_keys.p, 0xFF, ss->d_keys.cap * 8, st));      /* (scores of a free slot are never read) */
					ss->tables_clean_words = ss->d_keys.cap;
				}
				RC_TRY(ss->d_ticket.ensure(16));
				HIP_TRY(hipMemsetAsync(ss->d_ticket.p, 0, 4, st));
				a.keys = ss->d_keys.p;
				memcpy(h_work, hbm_now.data(), hbm_now.size() * 4);
				HIP_TRY(hipMemcpyAsync(ss->d_work.p, h_work, hbm_now.size() * 4, hipMemcpyHostToDevice, st));
				a.n_work = (int32_t) hbm_now.size();
				ss->tables_clean_words = 0;                     /* until the launch is known to have ended */
				RC_TRY(ss->kmark(st));
				HIP_TRY(launch_search_wave_hbm(a, (int) n_tables, ss->d_ticket.p, st));
				RC_TRY(ss->kmark(st));
			} else {
			const size_t chunk = std::max<size_t>(1, std::min<size_t>(hbm_now.size(), ((size_t) 8 << 30) / (per_read * 16)));
			ss->tables_clean_words = 0;                         /* the lane-per-read kernel leaves its tables as they are */
			RC_TRY(ss->d_keys.ensure(chunk * per_read));
			RC_TRY(ss->d_scores.ensure(chunk * per_read * 2));
			a.keys = ss->d_keys.p; a.scores = ss->d_scores.p;
			for (size_t w0 = 0; w0 < hbm_now.size(); w0 += chunk) {
				const size_t m = std::min(chunk, hbm_now.size() - w0);
				memcpy(h_work, hbm_now.data() + w0, m * 4);
				HIP_TRY(hipMemcpyAsync(ss->d_work.p, h_work, m * 4, hipMemcpyHostToDevice, st));
				HIP_TRY(hipMemsetAsync(ss->d_keys.p, 0xFF, m * per_read * 8, st));          /* every slot empty */
				a.n_work = (int32_t) m;
				RC_TRY(ss->kmark(st));
				HIP_TRY(launch_search(a, st));
				RC_TRY(ss->kmark(st));
				if (w0 + chunk < hbm_now.size()) RC_TRY(search_wait(ss, st));      /* (the work list is reused by the next chunk) */
			}
			}
			HIP_TRY(hipMemcpyAsync(h_ncand, ss->d_ncand.p, n1 * 4, hipMemcpyDeviceToHost, st));
			RC_TRY(search_wait(ss, st));
			if (!lane_serial) ss->tables_clean_words = ss->d_keys.cap;      /* the launch ended: every wave left its table empty */
			for (int32_t i : hbm_now) if (h_ncand[i] < 0) next_hbm.push_back(i);
			if (trace) { char b[160]; snprintf(b, sizeof(b), " [bits %d hbm %zu (%llu votes in the call) -> %zu retry, at %.2f ms]", bits, hbm_now.size(), (unsigned long long) total, next_hbm.size(), tr_ms()); tr += b; }
		}
		work_wave.swap(next_wave);
		work_hbm.swap(next_hbm);
	}
	/* (a read that left the ladder without a list keeps -1; one still flagged for the HBM kernel cannot remain: it was redone in its attempt) */
	/* dense candidate list in read order, compacted on the device: only the entries come back (the sparse arena is
	 * two slots per vote) */
	uint64_t need = 0;
	for (int i = 0; i < n; ++i) { h_begin[i] = need; cand_begin[i] = need; n_candidates[i] = h_ncand[i]; if (h_ncand[i] > 0) need += (uint64_t) h_ncand[i]; }
	if (cand_used) *cand_used = need;
	if (max_hit || kmer_misses) {
		HIP_TRY(hipMemcpyAsync(h_maxhit, ss->d_maxhit.p, n1 * 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(h_miss, ss->d_miss.p, n1 * 4, hipMemcpyDeviceToHost, st));
	}
	if (need > cand_capacity || (need > 0 && !cands)) {
		if (max_hit || kmer_misses) RC_TRY(search_wait(ss, st));
		set_err("cvx_search_batch: candidate arena too small (%llu needed, %llu given)", (unsigned long long) need, (unsigned long long) cand_capacity);
		return CVX_ERR_CAPACITY;
	}
	if (need) {
		RC_TRY(ss->d_dense.ensure((size_t) need + 64));
		HIP_TRY(hipMemcpyAsync(ss->d_begin.p, h_begin, n1 * 8, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(ss->d_srcoff.p, h_srcoff, n1 * 8, hipMemcpyHostToDevice, st));
		RC_TRY(ss->kmark(st));
		HIP_TRY(launch_search_compact(ss->d_cand.p, ss->d_srcoff.p, ss->d_ncand.p, ss->d_begin.p, ss->d_dense.p, n, st));
		RC_TRY(ss->kmark(st));
		/* (straight into the caller's memory: pageable unless it came from cvx_host_alloc, then the copy is staged by the runtime) */
		HIP_TRY(hipMemcpyAsync(cands, ss->d_dense.p, (size_t) need * sizeof(SearchCandidate), hipMemcpyDeviceToHost, st));
	}
	RC_TRY(search_wait(ss, st));
	for (size_t q = 0; q + 1 < ss->kev_used; q += 2) ss->kernel_ms += ev_ms(ss->kev[q], ss->kev[q + 1]);
	if (trace) fprintf(stderr, "cvx_search_batch: %d reads, %llu bases, %llu candidates, %.2f ms:%s\n", n, (unsigned long long) bytes - (unsigned long long) n, (unsigned long long) need, tr_ms(), tr.c_str());
	if (max_hit) memcpy(max_hit, h_maxhit, n1 * 4);
	if (kmer_misses) memcpy(kmer_misses, h_miss, n1 * 4);
	return CVX_OK;
	ABI_GUARD_END
}

extern "C" {

int cvx_search_batch_ex(cvx_handle h, cvx_index ix, int32_t n, const char *const *seqs, const int32_t *lens,
		float sensitivity, float min_hits, int32_t bin_shift, int32_t first_bits,
		int32_t *n_candidates, uint64_t *cand_begin, cvx_candidate *cands, uint64_t cand_capacity, uint64_t *cand_used,
		float *max_hit, int32_t *kmer_misses) {
	return search_common(h, ix, n, seqs, lens, nullptr, nullptr, sensitivity, min_hits, bin_shift, first_bits, n_candidates, cand_begin, cands, cand_capacity, cand_used, max_hit, kmer_misses);
}

int cvx_search_batch_arena(cvx_handle h, cvx_index ix, int32_t n, const uint8_t *arena, const uint64_t *offsets,
		float sensitivity, float min_hits, int32_t bin_shift, int32_t first_bits,
		int32_t *n_candidates, uint64_t *cand_begin, cvx_candidate *cands, uint64_t cand_capacity, uint64_t *cand_used,
		float *max_hit, int32_t *kmer_misses) {
	if (n > 0 && (!arena || !offsets)) { set_err("cvx_search_batch_arena: NULL arena"); return CVX_ERR_ARG; }
	return search_common(h, ix, n, nullptr, nullptr, arena ? arena : reinterpret_cast<const uint8_t *>(""), offsets, sensitivity, min_hits, bin_shift, first_bits, n_candidates, cand_begin, cands, cand_capacity, cand_used, max_hit, kmer_misses);
}

int cvx_search_last_attempts(cvx_handle h, int32_t n, int32_t *attempts) {
	if (!h || n < 0 || (n > 0 && !attempts)) { set_err("cvx_search_last_attempts: bad argument"); return CVX_ERR_ARG; }
	if (!h->search || (size_t) n > h->search->attempts.size()) { set_err("cvx_search_last_attempts: the handle's last search had fewer than %d reads", n); return CVX_ERR_ARG; }
	if (n) memcpy(attempts, h->search->attempts.data(), (size_t) n * sizeof(int32_t));
	return CVX_OK;
}

int cvx_search_batch(cvx_handle h, cvx_index ix, int32_t n, const char *const *seqs, const int32_t *lens,
		float sensitivity, float min_hits, int32_t bin_shift,
		int32_t *n_candidates, uint64_t *cand_begin, cvx_candidate *cands, uint64_t cand_capacity, uint64_t *cand_used) {
	return cvx_search_batch_ex(h, ix, n, seqs, lens, sensitivity, min_hits, bin_shift, 0, n_candidates, cand_begin, cands, cand_capacity, cand_used, nullptr, nullptr);
}

/* ------------------------------------------------------------------ sub-read scoring */

int cvx_score_batch(cvx_handle h, int32_t n, const char *const *refs, const char *const *qrys, float *scores) {
	ABI_GUARD_BEGIN
	if (!h || n < 0 || (n > 0 && (!refs || !qrys || !scores))) { set_err("cvx_score_batch: bad argument"); return CVX_ERR_ARG; }
	if (n == 0) return CVX_OK;
	HIP_TRY(hipSetDevice(h->device));
	/* persistent pinned staging and device buffers: no allocation on the steady-state path */
	RC_TRY(h->sc_hpairs.ensure((size_t) n * sizeof(ScorePair)));
	RC_TRY(h->sc_hout.ensure((size_t) n * sizeof(float)));
	ScorePair *pairs = h->sc_hpairs.as<ScorePair>();
	uint64_t bytes = 0, rows = 0;
	size_t max_rl = 0, max_ql = 0;
	for (int i = 0; i < n; ++i) {
		if (!refs[i] || !qrys[i]) { set_err("cvx_score_batch: NULL sequence %d", i); return CVX_ERR_ARG; }
		const size_t rl = strlen(refs[i]) + 1, ql = strlen(qrys[i]) + 1;
		max_rl = std::max(max_rl, rl);
		max_ql = std::max(max_ql, ql);
		ScorePair &p = pairs[i];
		p.ref_off = bytes; bytes += rl;
		p.qry_off = bytes; bytes += ql;
		p.ref_len = (int32_t) std::min<size_t>(rl, 0x7fffffff);
		p.qry_len = (int32_t) std::min<size_t>(ql, 0x7fffffff);
		p.scratch_off = rows;
		if (rl < 100000 && ql < 100000) rows += 2 * (uint64_t) rl;
	}
	RC_TRY(h->sc_hseq.ensure((size_t) bytes + 256));
	uint8_t *hseq = h->sc_hseq.as<uint8_t>();
	for (int i = 0; i < n; ++i) {
		memcpy(hseq + pairs[i].ref_off, refs[i], (size_t) pairs[i].ref_len);
		memcpy(hseq + pairs[i].qry_off, qrys[i], (size_t) pairs[i].qry_len);
	}
	RC_TRY(h->sc_seq.ensure((size_t) bytes + 256));
	RC_TRY(h->sc_pairs.ensure((size_t) n));
	RC_TRY(h->sc_rows.ensure((size_t) rows + 64));
	RC_TRY(h->sc_out.ensure((size_t) n));
	hipStream_t st = h->s_main;
	HIP_TRY(hipMemcpyAsync(h->sc_seq.p, hseq, (size_t) ((bytes + 255) / 256 * 256), hipMemcpyHostToDevice, st));   /* dword-aligned size: SDMA, not a blit kernel */
	HIP_TRY(hipMemcpyAsync(h->sc_pairs.p, pairs, (size_t) n * sizeof(ScorePair), hipMemcpyHostToDevice, st));
	/* the kernel alone, on the stream it runs on (cvx_score_kernel_ms: the device-resident rate beside the marshalled one) */
	if (!h->sc_ev0) { HIP_TRY(hipEventCreate(&h->sc_ev0)); HIP_TRY(hipEventCreate(&h->sc_ev1)); }
	HIP_TRY(hipEventRecord(h->sc_ev0, st));
	if (max_ql <= 512 && max_rl <= 2048 && !h->score_no_diag)
		HIP_TRY(launch_score_diag(h->sc_seq.p, h->sc_pairs.p, h->sc_out.p, n, st));      /* the batched shape (256-base sub-read x ~300-base window) */
	else
		HIP_TRY(launch_score(h->sc_seq.p, h->sc_pairs.p, h->sc_rows.p, h->sc_out.p, n, (int) std::min<size_t>(max_rl, 0x7fffffff), st));
	HIP_TRY(hipEventRecord(h->sc_ev1, st));
	HIP_TRY(hipMemcpyAsync(h->sc_hout.p, h->sc_out.p, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, st));
	/* wait on a blocking event, not by spinning on the stream: in ngmlr dozens of worker threads sit in this call at the
	 * same time on a host that needs its cores for the stages that stayed on the CPU */
	if (!h->sc_done) HIP_TRY(hipEventCreateWithFlags(&h->sc_done, hipEventBlockingSync | hipEventDisableTiming));
	HIP_TRY(hipEventRecord(h->sc_done, st));
	HIP_TRY(hipEventSynchronize(h->sc_done));
	memcpy(scores, h->sc_hout.p, (size_t) n * sizeof(float));
	h->sc_kernel_ms = ev_ms(h->sc_ev0, h->sc_ev1);
	return CVX_OK;
	ABI_GUARD_END
}

int cvx_stage_kernel_ms(cvx_handle h, int32_t stage, float *ms) {
	if (!h || !ms) { set_err("cvx_stage_kernel_ms: NULL argument"); return CVX_ERR_ARG; }
	switch (stage) {
	case CVX_STAGE_SCORE: *ms = h->sc_kernel_ms; return CVX_OK;
	case CVX_STAGE_DECODE: *ms = h->decode_kernel_ms; return CVX_OK;
	case CVX_STAGE_SEARCH: *ms = h->search ? h->search->kernel_ms : 0.0f; return CVX_OK;
	default: set_err("cvx_stage_kernel_ms: unknown stage %d", stage); return CVX_ERR_ARG;
	}
}

int cvx_score_kernel_ms(cvx_handle h, float *ms) {
	if (!h || !ms) { set_err("cvx_score_kernel_ms: NULL argument"); return CVX_ERR_ARG; }
	*ms = h->sc_kernel_ms;
	return CVX_OK;
}

}  /* extern "C" */
