/* batching_aligner.cpp -- see batching_aligner.h */
#include "batching_aligner.h"

#include <chrono>

namespace Convex {

BatchingAligner::BatchingAligner(ConvexAlignHip * be, int nWorkers, int maxB, int tmoUs) :
		backend(be), workers(nWorkers > 0 ? nWorkers : 1), parked(0), leaderActive(false),
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

/* called with the lock held by a parked thread: run everything that is queued */
void BatchingAligner::flushLocked(std::unique_lock<std::mutex> & lk) {
	leaderActive = true;
	std::vector<Request *> batch;
	batch.swap(queue);
	lk.unlock();

	std::vector<ConvexAlignHip::Tile> tiles(batch.size());
	for (size_t i = 0; i < batch.size(); ++i) tiles[i] = batch[i]->tile;
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
	requests += 1;
	parked += 1;
	std::chrono::steady_clock::time_point const deadline =
			std::chrono::steady_clock::now() + std::chrono::microseconds(timeoutUs > 0 ? timeoutUs : 1000000);
	while (!req.done) {
		bool const mine = !leaderActive && !queue.empty() &&
				((int) queue.size() >= maxBatch || parked >= workers ||
				 (timeoutUs > 0 && std::chrono::steady_clock::now() >= deadline));
		if (mine) {
			flushLocked(lk);      /* may or may not contain my own request */
			continue;
		}
		if (timeoutUs > 0) cv.wait_until(lk, deadline);
		else cv.wait(lk);
	}
	parked -= 1;
	lk.unlock();
	if (req.failed) throw 1;
	return req.tile.ret;
}

}  // namespace Convex
