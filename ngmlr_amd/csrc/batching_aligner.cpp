/* batching_aligner.cpp -- see batching_aligner.h */
#include "batching_aligner.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace Convex {

BatchingAligner::BatchingAligner(ConvexAlignHip * be, int nWorkers, int maxB, int tmoUs) :
		backend(be), workers(nWorkers >= 0 ? nWorkers : 1), parked(0), leaderActive(false),
		maxBatch(maxB > 0 ? maxB : 1), timeoutUs(tmoUs), launches(0), requests(0) {
}

BatchingAligner::~BatchingAligner() {
}

void BatchingAligner::WorkerDone() {
	std::unique_lock<std::mutex> lk(mtx);
	workers -= 1;
	/* the remaining workers may all be parked now */
	if (!leaderActive && !queue.empty() && parked >= workers) flushLocked(lk);
}

void BatchingAligner::WorkerJoined() {
	std::unique_lock<std::mutex> lk(mtx);
	workers += 1;
}

/* called with the lock held by a parked thread: run everything that is queued */
void BatchingAligner::flushLocked(std::unique_lock<std::mutex> & lk) {
	leaderActive = true;
	std::vector<Request *> batch;
	batch.swap(queue);
	lk.unlock();

	std::vector<ConvexAlignHip::Tile> tiles(batch.size());
	for (size_t i = 0; i < batch.size(); ++i) { tiles[i] = batch[i]->tile; batch[i]->queued = false; }
	bool failed = false;
	try {
		backend->AlignTiles(tiles.data(), (int) tiles.size());
	} catch (...) {
		failed = true;   /* every waiter rethrows in its own thread, like the reference's hard errors */
	}

	lk.lock();
	for (size_t i = 0; i < batch.size(); ++i) {
		batch[i]->tile.ret = tiles[i].ret;
		batch[i]->failed = failed;
		batch[i]->done = true;
	}
	launches += 1;
	leaderActive = false;
	cv.notify_all();
}

int BatchingAligner::SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
		char const * const refSeq, char const * const qrySeq, Align & result,
		int const externalQStart, int const externalQEnd, void * extData) {
	(void) mode; (void) extData;
	Request req;
	req.tile.corridor = corridor; req.tile.corridorHeight = corridorHeight;
	req.tile.refSeq = refSeq; req.tile.qrySeq = qrySeq; req.tile.result = &result;
	req.tile.externalQStart = externalQStart; req.tile.externalQEnd = externalQEnd; req.tile.ret = -1;
	req.done = false; req.failed = false;

	std::unique_lock<std::mutex> lk(mtx);
	queue.push_back(&req);
	req.queued = true;
	requests += 1;
	parked += 1;
	std::chrono::microseconds const patience(timeoutUs > 0 ? timeoutUs : 1000000);
	std::chrono::steady_clock::time_point deadline = std::chrono::steady_clock::now() + patience;
	while (!req.done) {
		bool const expired = timeoutUs > 0 && std::chrono::steady_clock::now() >= deadline;
		bool const mine = !leaderActive && !queue.empty() &&
				((int) queue.size() >= maxBatch || parked >= workers || expired);
		if (mine) {
			flushLocked(lk);      /* may or may not contain my own request */
			continue;
		}
		/* The timed wait only makes sense while this request still sits in the queue and nobody is
		 * flushing: once a leader has taken it (or is busy with an earlier batch) the only event to
		 * wait for is the leader's notify_all -- a deadline that has already passed would turn
		 * wait_until into a spin on the mutex the leader needs. */
		if (timeoutUs > 0 && req.queued && !leaderActive) {
			if (expired) deadline = std::chrono::steady_clock::now() + patience;   /* re-arm */
			cv.wait_until(lk, deadline);
		} else {
			cv.wait(lk);
		}
	}
	parked -= 1;
	lk.unlock();
	if (req.failed) throw 1;
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
}

SharedAligner::SharedAligner(int const stdOutMode, float const match, float const mismatch, float const gapOpen,
		float const gapExtend, float const gapExtendMin, float const gapDecay, int const deviceId) : shared(0), device(0) {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	int nDev = cvx_device_count();
	if (const char * e = getenv("CVX_DEVICES")) nDev = atoi(e) < nDev ? atoi(e) : nDev;      /* use only the first k devices */
	if (nDev > kMaxDevices) nDev = kMaxDevices;
	/* deviceId >= 0 pins the worker; the default spreads them */
	device = (deviceId >= 0 && nDev > 0) ? deviceId % nDev : (nDev > 0 ? (int) (g_joined % nDev) : 0);
	g_joined += 1;
	if (g_shared[device] == 0) {
		int maxBatch = 4096, timeoutUs = 2000;
		if (const char * e = getenv("CVX_BATCH_MAX")) maxBatch = atoi(e);
		if (const char * e = getenv("CVX_BATCH_TIMEOUT_US")) timeoutUs = atoi(e);
		g_backend[device] = new ConvexAlignHip(stdOutMode, match, mismatch, gapOpen, gapExtend, gapExtendMin, gapDecay, device);   /* throws without a usable device */
		g_shared[device] = new BatchingAligner(g_backend[device], 0, maxBatch, timeoutUs);   /* workers join one by one */
	}
	g_shared[device]->WorkerJoined();
	g_deviceUsers[device] += 1;
	g_users += 1;
	shared = g_shared[device];
}

SharedAligner::~SharedAligner() {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	shared->WorkerDone();
	g_deviceUsers[device] -= 1;
	g_users -= 1;
	if (g_deviceUsers[device] == 0) {
		g_lastLaunches += g_shared[device]->Launches();
		g_lastRequests += g_shared[device]->Requests();
		delete g_shared[device]; g_shared[device] = 0;
		delete g_backend[device]; g_backend[device] = 0;
	}
	if (g_users == 0) {
		fprintf(stderr, "SharedAligner: %ld alignments in %ld device launches (%.1f per launch)\n", g_lastRequests,
				g_lastLaunches, g_lastLaunches ? (double) g_lastRequests / (double) g_lastLaunches : 0.0);
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
long SharedAligner::Requests() {
	std::lock_guard<std::mutex> g(g_sharedMtx);
	long n = g_lastRequests;
	for (int d = 0; d < kMaxDevices; ++d) if (g_shared[d]) n += g_shared[d]->Requests();
	return n;
}

}  // namespace Convex
