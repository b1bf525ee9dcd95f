/*
 * batching_aligner.h -- many reads in flight behind the single-call IAlignment surface.
 *
 * ngmlr calls SingleAlign once per interval from each worker thread and needs the result
 * before it can trim the next interval (reference src/AlignmentBuffer.cpp:3361-3406,
 * :291-425), so one worker can never keep a GPU busy.  BatchingAligner is an IAlignment
 * that many worker threads share: each SingleAlign parks its request in a queue; when the
 * queue is full, or every registered worker is parked, or a deadline passes, ONE of the
 * parked threads takes the whole queue to ConvexAlignHip::AlignTiles (one device launch)
 * and wakes the others.  No extra thread, no change to the callers.
 *
 * Usage in ngmlr: one BatchingAligner per device, handed to every AlignmentBuffer in place
 * of its private aligner; `workers` = number of CS threads (-t).
 */
#ifndef BATCHING_ALIGNER_H
#define BATCHING_ALIGNER_H

#include <condition_variable>
#include <mutex>
#include <vector>

#include "convex_align_hip.h"

namespace Convex {

class BatchingAligner: public IAlignment {
public:
	/* backend is owned by the caller.  maxBatch: flush when this many requests wait;
	 * timeoutUs: flush after the oldest request waited this long (0 = only the other rules). */
	BatchingAligner(ConvexAlignHip * backend, int workers, int maxBatch = 4096, int timeoutUs = 2000);
	virtual ~BatchingAligner();

	virtual int GetScoreBatchSize() const { return 0; }
	virtual int GetAlignBatchSize() const { return 0; }
	virtual int BatchScore(int const, int const, char const * const * const, char const * const * const,
			float * const, void *) { throw "Not implemented"; }
	virtual int BatchAlign(int const, int const, char const * const * const, char const * const * const,
			Align * const, void *) { throw "Not implemented"; }
	virtual int SingleAlign(int const, int const, char const * const, char const * const, Align &, void *) {
		throw "Not implemented";
	}
	virtual int SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
			char const * const refSeq, char const * const qrySeq, Align & result,
			int const externalQStart, int const externalQEnd, void * extData);

	/* a worker that stops calling (end of input) must say so, or the "everybody is parked"
	 * rule would wait for it */
	void WorkerDone();
	/* a worker that joins after construction (SharedAligner: one per AlignmentBuffer) */
	void WorkerJoined();

	/* statistics */
	long Launches() const { return launches; }
	long Requests() const { return requests; }

private:
	struct Request {
		ConvexAlignHip::Tile tile;
		bool done;
		bool failed;
		bool queued;           /* still in the queue (no leader has taken it yet) */
	};
	ConvexAlignHip * backend;
	std::mutex mtx;
	std::condition_variable cv;
	std::vector<Request *> queue;
	int workers;
	int parked;
	bool leaderActive;
	int maxBatch;
	int timeoutUs;
	long launches, requests;

	void flushLocked(std::unique_lock<std::mutex> & lk);
};

/*
 * SharedAligner -- the form ngmlr's pipeline takes without any other change: every
 * AlignmentBuffer (one per CS worker thread, reference src/CS.cpp:412-418) constructs "its"
 * aligner at src/AlignmentBuffer.h:355 and deletes it in its destructor (:375).  Constructing a
 * SharedAligner there instead gives each worker a thin IAlignment whose SingleAlign parks in ONE
 * BatchingAligner per device (one ConvexAlignHip each): with `-t N` up to N tiles travel per device
 * launch, and on a node with several MI355X the workers -- hence the reads -- are dealt round-robin over
 * the devices (CVX_DEVICES=k limits that to the first k).  The first SharedAligner of a device creates
 * its pair, the last one to go deletes it; the very last one prints the launch statistics.  Scoring parameters are those of the
 * first construction (every worker passes the same Config values).
 */
class SharedAligner: public IAlignment {
public:
	SharedAligner(int const stdOutMode, float const match, float const mismatch, float const gapOpen,
			float const gapExtend, float const gapExtendMin, float const gapDecay, int const deviceId = -1 /* -1: workers are dealt over all devices */);
	virtual ~SharedAligner();

	virtual int GetScoreBatchSize() const { return 0; }
	virtual int GetAlignBatchSize() const { return 0; }
	virtual int BatchScore(int const, int const, char const * const * const, char const * const * const,
			float * const, void *) { throw "Not implemented"; }
	virtual int BatchAlign(int const, int const, char const * const * const, char const * const * const,
			Align * const, void *) { throw "Not implemented"; }
	virtual int SingleAlign(int const, int const, char const * const, char const * const, Align &, void *) {
		throw "Not implemented";
	}
	virtual int SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
			char const * const refSeq, char const * const qrySeq, Align & result,
			int const externalQStart, int const externalQEnd, void * extData);

	/* statistics of the process-wide aligner (0 when none exists) */
	static long Launches();
	static long Requests();

private:
	BatchingAligner * shared;
	int device;
};

}  // namespace Convex

#endif
