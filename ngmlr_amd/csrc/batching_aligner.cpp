/* batching_aligner.cpp -- see batching_aligner.h */
#include "batching_aligner.h"
#include "cvx_pcsample.h"
#include "service_device.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <pthread.h>

#ifdef CVX_IN_NGMLR_TREE
/* Inside ngmlr the first device call of the run used to be made by the first CS thread, after the index had been built:
 * the HIP runtime's initialisation, the first handle and the first use of every kernel (code objects are loaded on first
 * launch) -- several hundred ms -- sat at the head of the mapping phase.  A thread started when the binary is loaded does all
 * of that while ngmlr parses its arguments and builds or reads its index: it creates a handle, aligns one tiny tile, scores one
 * pair and goes away.  CVX_PREWARM=0 switches it off. */
namespace {
struct Prewarm {
	std::thread t;
	/* (joined when the binary's statics go, i.e. before the HIP runtime's own: a run that ends at once -- `--help` -- must not
	 * tear the runtime down under this thread) */
	std::chrono::steady_clock::time_point const loaded = std::chrono::steady_clock::now();
	~Prewarm() {
		if (t.joinable()) t.join();
		if (getenv("CVX_TIMELINE")) fprintf(stderr, "cvx timeline: the binary's statics go %.2f s after the library was loaded\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - loaded).count());
	}
	Prewarm() {
		const char * e = getenv("CVX_PREWARM");
		if (e && atoi(e) == 0) return;
		t = std::thread([this] {
			cvx_params p = { 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f };
			cvx_handle h = 0;
			if (cvx_create(0, &p, 0, &h) != CVX_OK) return;      /* no device: the run will say so itself */
			struct Done { Prewarm * w; ~Done() { if (getenv("CVX_TIMELINE")) fprintf(stderr, "cvx timeline: device warm-up done %.2f s after the library was loaded\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - w->loaded).count()); } } done{this};
			static const char ref[] = "ACGTTGCAAGGCTTAACCGGTTAAGGCCTTGACCATGGTACCAGTCAGTCGATCGATTGCA";
			static const char qry[] = "ACGTTGCAAGGCTTAACCGGTTAAGGCCTTGACCATGGTACCAGTCAGTCGATCGATTGCA";
			int32_t off[60], len[60];
			for (int y = 0; y < 60; ++y) { off[y] = y - 16; len[y] = 32; }
			cvx_tile t;
			memset(&t, 0, sizeof(t));
			t.ref = ref; t.qry = qry; t.ref_len = 60; t.qry_len = 60;
			t.row_offset = off; t.row_length = len; t.row_stride_bytes = 4; t.corridor_kind = CVX_CORRIDOR_ROWS;
			cvx_result r;
			uint32_t ops[256];
			uint64_t used = 0;
			(void) cvx_align_batch(h, 1, &t, &r, ops, 256, &used);
			const char * rr[1] = { ref };
			const char * qq[1] = { qry };
			float sc = 0.0f;
			(void) cvx_score_batch(h, 1, rr, qq, &sc);
			cvx_destroy(h);
		});
	}
} g_prewarm;
}
#endif

namespace Convex {

BatchingAligner::BatchingAligner(ConvexAlignHip * be, int nWorkers, int maxB, int tmoUs) :
		backend(be), workers(nWorkers >= 0 ? nWorkers : 1), parked(0),
		maxBatch(maxB > 0 ? maxB : 1), timeoutUs(tmoUs), stop(false), launches(0), requests(0), maxInFlight(0),
		target(0), holdUs(20000), feedActive(true), textLaunches(0), textNs(0), launchTrace(false), deviceText(false), parkedNs(0), finishNs(0), busyNs(0), maxFlight(2), emaServiceUs(0.0), leadUs(3000), textPending(0), textStop(false), maxTextAhead(4) {
	if (const char * e = getenv("CVX_BATCH_TARGET")) target = atoi(e) > 0 ? atoi(e) : 0;
	if (const char * e = getenv("CVX_BATCH_HOLD_US")) holdUs = atoi(e) > 0 ? atoi(e) : 0;
	if (const char * e = getenv("CVX_BATCH_LEAD_US")) leadUs = atoi(e);      /* < 0: the plain timeout rule while a launch runs */
	/* launches in flight: the upload and corridor analysis of the second run under the kernels of the first.  More (tried
	 * with 4, and with the runtime's two stream sets) only makes the launches smaller: 20 000 reads 25.1 s against 21.2 s */
	if (const char * e = getenv("CVX_BATCH_INFLIGHT")) maxFlight = atoi(e) > 0 ? atoi(e) : 1;
	if (const char * e = getenv("CVX_DEVICE_TEXT")) deviceText = atoi(e) != 0;
	if (const char * e = getenv("CVX_LAUNCH_TRACE")) launchTrace = atoi(e) != 0;
	if (const char * e = getenv("CVX_TEXT_AHEAD")) maxTextAhead = atoi(e) > 0 ? atoi(e) : 1;
	dispatcher = std::thread([this] { dispatchLoop(); });
	if (deviceText) textThread = std::thread([this] { textLoop(); });
}

BatchingAligner::~BatchingAligner() {
	{
		std::lock_guard<std::mutex> lk(mtx);
		stop = true;
	}
	cvDispatch.notify_all();
	dispatcher.join();
	if (textThread.joinable()) {
		{
			std::lock_guard<std::mutex> lk(mtx);
			textStop = true;
		}
		cvText.notify_all();
		textThread.join();
	}
}

/* with mtx held */
void BatchingAligner::completeLaunch(Launch * l) {
	for (size_t i = 0; i < l->reqs.size(); ++i) {
		Request * r = l->reqs[i];
		r->failed = l->failed;
		r->result = l->failed ? 0 : &l->results[i];
		r->done = true;
		/* under the lock: the request lives on its worker's stack until that worker has seen `done` */
		if (r->fiber) FiberApi::Wake(r->fiber);      /* a read on a user-level context: runnable on its carrier (cvx_fiber.h) */
		else r->cv.notify_one();
	}
}

/* CVX_DEVICE_TEXT=1: CIGAR / MD / profile of finished launches on the device, one launch at a time in the order they finished */
void BatchingAligner::textLoop() {
	pthread_setname_np(pthread_self(), "cvx-text");
	std::unique_lock<std::mutex> lk(mtx);
	for (;;) {
		while (textQueue.empty() && !textStop) cvText.wait(lk);
		if (textQueue.empty()) break;
		Launch * l = textQueue.front();
		textQueue.pop_front();
		lk.unlock();
		try {
			std::vector<ConvexAlignHip::Tile const *> tiles(l->reqs.size());
			for (size_t i = 0; i < tiles.size(); ++i) tiles[i] = &l->reqs[i]->tile;
			l->text = new ConvexAlignHip::JobText();
			std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
			backend->Text(l->job, tiles.data(), (int) tiles.size(), *l->text);
			long long const ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
			lk.lock();
			textNs += ns;
			textLaunches += 1;
			if (launchTrace) fprintf(stderr, "cvx text stage: %d tiles in %.2f ms, %d launches behind it\n", (int) l->reqs.size(), (double) ns * 1e-6, (int) textQueue.size());
		} catch (...) {
			lk.lock();
			l->failed = true;
		}
		completeLaunch(l);
		textPending -= 1;
		cvDispatch.notify_one();
	}
}

void BatchingAligner::WorkerDone() {
	{
		std::lock_guard<std::mutex> lk(mtx);
		workers -= 1;
	}
	cvDispatch.notify_all();      /* the remaining workers may all be parked now */
}

void BatchingAligner::SetBatchTarget(int targetRequests, int holdMicroseconds) {
	std::lock_guard<std::mutex> lk(mtx);
	target = targetRequests > 0 ? targetRequests : 0;
	holdUs = holdMicroseconds > 0 ? holdMicroseconds : 0;
}

void BatchingAligner::SetFeedActive(bool active) {
	{
		std::lock_guard<std::mutex> lk(mtx);
		feedActive = active;
	}
	cvDispatch.notify_all();
}

void BatchingAligner::WorkerJoined() {
	std::lock_guard<std::mutex> lk(mtx);
	workers += 1;
}

/* called with the lock held */
bool BatchingAligner::shouldCut(bool deviceIdle) const {
	if (queue.empty()) return false;
	if ((int) queue.size() >= maxBatch) return true;
	std::chrono::steady_clock::time_point const now = std::chrono::steady_clock::now();
	/* nobody left who could add to the launch -- decisive only without a batch target: with many more contexts than cores
	 * (align_pool.h) the contexts that hold a read are parked nearly all the time (their host stages are short), while idle
	 * contexts and the CS threads are about to bring more; measured with this rule in front: 27-31 tiles per launch
	 * whatever the target (gpurun_out r04a) */
	if (target <= 0 && parked >= workers) return true;
	if (target > 0 && (int) queue.size() < target) {
		/* Many more contexts than cores (align_pool.h): the host stages, not the device, bound the throughput, a launch
		 * lasts about as long as its slowest tile whatever it carries, and every launch costs the dispatcher, the pack
		 * threads and `its` workers' wake-ups the same -- so a request may wait for company while other contexts keep
		 * the cores busy.  Bounded by holdUs -- and over at once when nobody is left to add to the launch: every worker that
		 * holds a read is parked and the pool has no producer and no queued read any more (the tail of a run, sparse input;
		 * each of a read's dependent intervals paid the full hold there: ADVICE r4). */
		if (!feedActive && parked >= workers) return true;
		return now - oldest >= std::chrono::microseconds(holdUs);
	}
	if (deviceIdle) return true;                             /* nothing to overlap with: latency first */
	if (emaServiceUs > 0.0 && leadUs >= 0) {
		/* A launch is running and the next one cannot start its kernels before that one is done: whatever arrives until
		 * shortly before then travels for free.  Cut `leadUs` (the time an upload and a corridor analysis take) before the
		 * running launch is expected to end -- a launch lasts about as long as its slowest tile whatever it carries, so
		 * tiles per launch is the throughput of the whole device path (20 000 reads: 18 -> 30 tiles per launch). */
		double const waitUs = emaServiceUs - (double) leadUs;
		return now - frontSince >= std::chrono::microseconds((long long) (waitUs > 0.0 ? waitUs : 0.0));
	}
	if (timeoutUs > 0 && now - oldest >= std::chrono::microseconds(timeoutUs)) return true;
	return false;
}

/* The one thread that owns the device handle.  Up to maxFlight (two) launches in flight: the upload and corridor analysis
 * of the younger run under the kernels of the older; requests that arrive meanwhile form the launch after that. */
void BatchingAligner::dispatchLoop() {
	pcsample::arm_this_thread(3);
	pthread_setname_np(pthread_self(), "cvx-dispatch");      /* (thread names: tools/e2e_rates.py splits the process's CPU time by them) */
	std::unique_lock<std::mutex> lk(mtx);
	for (;;) {
		/* buffers of launches whose workers have all finished writing */
		while (!retired.empty()) {
			Launch * l = retired.back();
			retired.pop_back();
			lk.unlock();
			if (l->job) backend->Release(l->job);
			delete l->text;
			delete l;
			lk.lock();
		}
		if (stop && queue.empty() && inFlight.empty() && textPending == 0) break;
		/* (with the text stage on its own thread a launch stays alive until that stage has run: the dispatcher does not cut
		 * further ahead than four launches whose text is still to be written -- their job slots hold the launch's arenas; two
		 * left the device idle whenever a text stage had to grow its buffers: a launch in flight 61 % of the time) */
		bool const canSubmit = (int) inFlight.size() < maxFlight && textPending < maxTextAhead;
		if (canSubmit && shouldCut(inFlight.empty())) {
			Launch * l = new Launch();
			l->job = 0; l->results = 0; l->ops = 0; l->failed = false; l->text = 0; l->submitMs = l->waitMs = 0.0;
			size_t const take = std::min(queue.size(), (size_t) maxBatch);
			l->oldestAt = oldest;
			l->cutAt = std::chrono::steady_clock::now();
			l->reqs.assign(queue.begin(), queue.begin() + (long) take);
			queue.erase(queue.begin(), queue.begin() + (long) take);
			if (!queue.empty()) oldest = std::chrono::steady_clock::now();
			l->unfinished = (int) l->reqs.size();
			for (Request * r : l->reqs) r->launch = l;
			launches += 1;
			lk.unlock();
			std::vector<ConvexAlignHip::Tile> tiles(l->reqs.size());
			for (size_t i = 0; i < tiles.size(); ++i) tiles[i] = l->reqs[i]->tile;
			try {
				l->job = backend->Submit(tiles.data(), (int) tiles.size());
			} catch (...) {
				l->failed = true;
			}
			l->submitMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - l->cutAt).count();
			lk.lock();
			if (inFlight.empty()) { busySince = std::chrono::steady_clock::now(); frontSince = busySince; }
			inFlight.push_back(l);
			if ((long) inFlight.size() > maxInFlight) maxInFlight = (long) inFlight.size();
			continue;                                        /* maybe a second launch right away */
		}
		if (!inFlight.empty()) {
			/* Block in Wait only when nothing else can happen meanwhile (two launches in flight, or nobody left to
			 * add requests).  Otherwise poll the oldest launch at a few kHz: requests that arrive while it runs are cut
			 * into the second launch -- whose upload and corridor analysis then run under its kernels -- as soon as
			 * one of the rules fires. */
			Launch * l = inFlight.front();
			bool const block = l->failed || (int) inFlight.size() >= maxFlight || parked >= workers;
			if (!block) {
				lk.unlock();
				bool const done = backend->Poll(l->job);
				lk.lock();
				if (!done) {
					cvDispatch.wait_for(lk, std::chrono::microseconds(200));
					continue;
				}
			}
			inFlight.pop_front();
			lk.unlock();
			if (!l->failed) {
				try {
					std::chrono::steady_clock::time_point const tw = std::chrono::steady_clock::now();
					backend->Wait(l->job, &l->results, &l->ops);
					if (launchTrace) {
						std::chrono::steady_clock::time_point const t = std::chrono::steady_clock::now();
						l->waitMs = std::chrono::duration<double, std::milli>(t - tw).count();
						fprintf(stderr, "cvx dispatcher: submit %.2f ms, wait entered %.2f ms after the cut, blocked in it %.2f ms; ", l->submitMs, std::chrono::duration<double, std::milli>(tw - l->cutAt).count(), l->waitMs);
						backend->Trace(l->job, (int) l->reqs.size(), std::chrono::duration<double, std::milli>(t - l->cutAt).count(),
								std::chrono::duration<double, std::milli>(l->cutAt - l->oldestAt).count());
					}
				} catch (...) {
					l->failed = true;
				}
			}
			lk.lock();
			{
				/* service time of the launch that just ended: from the moment it was the oldest one in flight */
				std::chrono::steady_clock::time_point const t = std::chrono::steady_clock::now();
				double const us = (double) std::chrono::duration_cast<std::chrono::microseconds>(t - frontSince).count();
				if (!l->failed) emaServiceUs = emaServiceUs > 0.0 ? 0.7 * emaServiceUs + 0.3 * us : us;
				frontSince = t;
				if (inFlight.empty()) busyNs += std::chrono::duration_cast<std::chrono::nanoseconds>(t - busySince).count();
			}
			if (!deviceText && !l->failed) {
				/* a launch of windows decoded on the device (window_decode_binding.inc): its workers' host text stage reads the
				 * characters that came back with the results (they are parked: nobody else touches their tiles) */
				std::vector<ConvexAlignHip::Tile *> tp(l->reqs.size());
				for (size_t i = 0; i < tp.size(); ++i) tp[i] = &l->reqs[i]->tile;
				lk.unlock();
				bool refsFailed = false;
				try {
					(void) backend->WindowRefs(l->job, tp.data(), (int) tp.size());
				} catch (...) {
					refsFailed = true;
				}
				lk.lock();
				if (refsFailed) l->failed = true;
			}
			if (deviceText && !l->failed) {
				/* its text stage runs on the text thread, under the kernels of the launches behind it */
				textQueue.push_back(l);
				textPending += 1;
				cvText.notify_one();
			} else {
				completeLaunch(l);
			}
			continue;
		}
		/* idle: nothing in flight, nothing to cut yet */
		if (!queue.empty() && target > 0 && (int) queue.size() < target) cvDispatch.wait_until(lk, oldest + std::chrono::microseconds(holdUs));
		else if (!queue.empty() && timeoutUs > 0) cvDispatch.wait_until(lk, oldest + std::chrono::microseconds(timeoutUs));
		else cvDispatch.wait(lk);
	}
}

int BatchingAligner::SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
		char const * const refSeq, char const * const qrySeq, Align & result,
		int const externalQStart, int const externalQEnd, void * extData) {
	(void) mode; (void) extData;
	Request req;
	req.tile.corridor = corridor; req.tile.corridorHeight = corridorHeight;
	req.tile.refSeq = refSeq; req.tile.qrySeq = qrySeq; req.tile.result = &result;
	req.tile.externalQStart = externalQStart; req.tile.externalQEnd = externalQEnd; req.tile.ret = -1; req.tile.failed = false;
	req.launch = 0; req.result = 0; req.done = false; req.failed = false;
	req.fiber = FiberApi::Current();
	ConvexAlignHip::Prepare(req.tile);       /* in the caller's thread; throws like the reference for a malformed call */

	std::unique_lock<std::mutex> lk(mtx);
	if (queue.empty()) oldest = std::chrono::steady_clock::now();
	queue.push_back(&req);
	requests += 1;
	parked += 1;
	cvDispatch.notify_one();
	std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
	if (req.fiber) {
		/* the read gives its carrier thread back while its tile is in flight; the dispatcher's Wake makes it runnable there */
		while (!req.done) {
			lk.unlock();
			FiberApi::Park();
			lk.lock();
		}
	} else {
		while (!req.done) req.cv.wait(lk);
	}
	std::chrono::steady_clock::time_point const t1 = std::chrono::steady_clock::now();
	parkedNs += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
	parked -= 1;
	Launch * l = req.launch;
	bool const failed = req.failed;
	cvx_result const * r = req.result;
	uint32_t const * ops = l->ops;
	lk.unlock();

	/* the text stage of this request, in this worker's thread, out of the launch's buffers */
	bool threw = failed;
	if (!failed) {
		try {
			if (l->text) backend->FinishText(req.tile, *r, *l->text, (int) (r - l->results));
			else backend->Finish(req.tile, *r, ops);
		} catch (...) {
			threw = true;
		}
	}
	lk.lock();
	finishNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
	if (--l->unfinished == 0) {
		retired.push_back(l);
		cvDispatch.notify_one();
	}
	lk.unlock();
	if (threw) throw 1;
	return req.tile.ret;
}

/* ------------------------------------------------------------------ SharedAligner */

namespace {
/* one (backend, batching aligner) pair per device; workers are dealt round-robin over the devices in the
 * order they construct their aligner, so ngmlr's -t N reads shard over every MI355X of the node (all tiles
 * of a read stay with its worker, hence on one device) */
const int kMaxDevices = 64;
std::mutex g_sharedMtx;
ConvexAlignHip * g_backend[kMaxDevices] = {0};
BatchingAligner * g_shared[kMaxDevices] = {0};
int g_deviceUsers[kMaxDevices] = {0};
int g_users = 0;
long g_joined = 0;
long g_lastLaunches = 0, g_lastRequests = 0;
double g_lastParked = 0.0, g_lastFinish = 0.0, g_lastBusy = 0.0, g_lastTextSeconds = 0.0;
long g_lastTextLaunches = 0;
std::chrono::steady_clock::time_point g_firstJoin;
std::chrono::steady_clock::time_point const g_loaded = std::chrono::steady_clock::now();      /* ~ process start */
bool g_poolAccounting = false;                       /* under g_sharedMtx */
int g_poolTarget = 256, g_poolHoldUs = 30000;        /* the batch target a dispatcher gets under the pool (align_pool.cpp sets it with the accounting) */
bool g_feedActive = true;                            /* under g_sharedMtx: handed to dispatchers created later */
thread_local BatchingAligner * tl_dispatcher = 0;    /* the dispatcher of the SharedAligner this thread constructed (pool accounting) */
/* ... or this user-level context: a fiber's aligner front travels with the fiber, not with its carrier thread */
BatchingAligner * & currentDispatcher() {
	Fiber * const f = FiberApi::Current();
	return f ? reinterpret_cast<BatchingAligner * &>(FiberApi::Local(f, 0)) : tl_dispatcher;
}
}

void SharedAligner::UsePoolAccounting(bool on, int batchTarget, int holdMicroseconds) {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	g_poolAccounting = on;
	g_poolTarget = batchTarget;
	g_poolHoldUs = holdMicroseconds;
}
void SharedAligner::SetFeedActive(bool active) {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	g_feedActive = active;
	for (int d = 0; d < kMaxDevices; ++d) if (g_shared[d]) g_shared[d]->SetFeedActive(active);
}
void SharedAligner::ThreadBegin() { if (BatchingAligner * d = currentDispatcher()) d->WorkerJoined(); }
void SharedAligner::ThreadEnd() { if (BatchingAligner * d = currentDispatcher()) d->WorkerDone(); }

SharedAligner::SharedAligner(int const stdOutMode, float const match, float const mismatch, float const gapOpen,
		float const gapExtend, float const gapExtendMin, float const gapDecay, int const deviceId) : shared(0), device(0), perRead(false) {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	perRead = g_poolAccounting;
	/* CVX_DEVICES=k: only the first k devices.  CVX_ALIAS_DEVICES=k: deal the workers over k LOGICAL devices (own backend, own
	 * dispatcher each) that all live on the physical devices present -- the multi-device path of this class on a one-GPU box
	 * (tests; not a scaling measurement).  One rule for the whole pipeline: service_device.h */
	int nDev = 0, nPhysical = 0;
	DeviceLayout(nDev, nPhysical);
	if (nDev > kMaxDevices) nDev = kMaxDevices;
	/* deviceId >= 0 pins the worker; the default spreads them */
	device = (deviceId >= 0 && nDev > 0) ? deviceId % nDev : (nDev > 0 ? (int) (g_joined % nDev) : 0);
	if (g_joined == 0) g_firstJoin = std::chrono::steady_clock::now();
	g_joined += 1;
	if (g_shared[device] == 0) {
		int maxBatch = 4096, timeoutUs = 2000;
		if (const char * e = getenv("CVX_BATCH_MAX")) maxBatch = atoi(e);
		if (const char * e = getenv("CVX_BATCH_TIMEOUT_US")) timeoutUs = atoi(e);
		g_backend[device] = new ConvexAlignHip(stdOutMode, match, mismatch, gapOpen, gapExtend, gapExtendMin, gapDecay, nPhysical > 0 ? device % nPhysical : device);   /* throws without a usable device */
		g_shared[device] = new BatchingAligner(g_backend[device], 0, maxBatch, timeoutUs);   /* workers join one by one */
		/* with alignment contexts off the CS threads (align_pool.h) a launch waits for 256 tiles, 30 ms at most: hundreds of
		 * contexts hide that wait, and the device sees a few large launches instead of many small ones (CVX_BATCH_TARGET /
		 * CVX_BATCH_HOLD_US override) */
		g_shared[device]->SetFeedActive(g_feedActive);
		if (perRead && !getenv("CVX_BATCH_TARGET")) g_shared[device]->SetBatchTarget(g_poolTarget, getenv("CVX_BATCH_HOLD_US") ? atoi(getenv("CVX_BATCH_HOLD_US")) : g_poolHoldUs);
	}
	if (perRead) currentDispatcher() = g_shared[device];
	else g_shared[device]->WorkerJoined();
	g_deviceUsers[device] += 1;
	g_users += 1;
	shared = g_shared[device];
}

SharedAligner::~SharedAligner() {
	if (perRead) { if (currentDispatcher() == shared) currentDispatcher() = 0; }
	else shared->WorkerDone();   /* (outside the process-wide lock: it only touches this device's aligner) */
	std::lock_guard<std::mutex> g(g_sharedMtx);
	g_deviceUsers[device] -= 1;
	g_users -= 1;
	if (g_deviceUsers[device] == 0) {
		g_lastLaunches += g_shared[device]->Launches();
		g_lastRequests += g_shared[device]->Requests();
		g_lastParked += g_shared[device]->ParkedSeconds();
		g_lastFinish += g_shared[device]->FinishSeconds();
		g_lastBusy += g_shared[device]->BusySeconds();
		g_lastTextLaunches += g_shared[device]->TextLaunches();
		g_lastTextSeconds += g_shared[device]->TextSeconds();
		delete g_shared[device]; g_shared[device] = 0;
		delete g_backend[device]; g_backend[device] = 0;
	}
	if (g_users == 0) {
		fprintf(stderr, "SharedAligner: %ld alignments in %ld device launches (%.1f per launch)\n", g_lastRequests,
				g_lastLaunches, g_lastLaunches ? (double) g_lastRequests / (double) g_lastLaunches : 0.0);
		double const wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_firstJoin).count();
		fprintf(stderr, "SharedAligner: %ld workers over %.2f s: %.1f %% of their time parked in SingleAlign (%.2f ms per alignment), "
				"%.1f %% in their text stage (%.2f CPU-s, %.0f us per alignment); a launch was in flight %.1f %% of the time\n", g_joined, wall,
				100.0 * g_lastParked / (wall * (double) g_joined), g_lastRequests ? 1e3 * g_lastParked / (double) g_lastRequests : 0.0,
				100.0 * g_lastFinish / (wall * (double) g_joined), g_lastFinish, g_lastRequests ? 1e6 * g_lastFinish / (double) g_lastRequests : 0.0, 100.0 * g_lastBusy / wall);
		{
			long prepared = 0, closedForm = 0;
			ConvexAlignHip::CorridorStats(prepared, closedForm);
			fprintf(stderr, "SharedAligner: %ld of %ld corridors travelled as closed forms (cvx_corridor_fit), the rest as row arrays\n", closedForm, prepared);
		}
		{
			long wl = 0, wt = 0, ml = 0;
			ConvexAlignHip::WindowStats(wl, wt, ml);
			if (wl + ml > 0) fprintf(stderr, "SharedAligner: %ld tiles in %ld launches took their reference as windows of the genome in HBM (cvx_submit_windows; CVX_DEVICE_DECODE=0 turns that off), %ld mixed launches materialised theirs\n", wt, wl, ml);
		}
		if (g_lastTextLaunches > 0) fprintf(stderr, "SharedAligner: text stage on the device for %ld launches (cvx_job_text + cvx_job_nm_profile), %.3f s of the dispatchers' time\n", g_lastTextLaunches, g_lastTextSeconds);
		fprintf(stderr, "SharedAligner: library loaded at 0, first worker joined at %.2f s, last one left at %.2f s\n",
				std::chrono::duration<double>(g_firstJoin - g_loaded).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - g_loaded).count());
	}
}

int SharedAligner::SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
		char const * const refSeq, char const * const qrySeq, Align & result,
		int const externalQStart, int const externalQEnd, void * extData) {
	return shared->SingleAlign(mode, corridor, corridorHeight, refSeq, qrySeq, result, externalQStart, externalQEnd, extData);
}

long SharedAligner::Launches() {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	long n = g_lastLaunches;
	for (int d = 0; d < kMaxDevices; ++d) if (g_shared[d]) n += g_shared[d]->Launches();
	return n;
}
int SharedAligner::ActiveDevices() {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	int n = 0;
	for (int d = 0; d < kMaxDevices; ++d) if (g_shared[d]) n += 1;
	return n;
}
long SharedAligner::Requests() {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	long n = g_lastRequests;
	for (int d = 0; d < kMaxDevices; ++d) if (g_shared[d]) n += g_shared[d]->Requests();
	return n;
}

}  // namespace Convex
