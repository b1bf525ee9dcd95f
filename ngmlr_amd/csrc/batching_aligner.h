/*
 * batching_aligner.h -- many reads in flight behind the single-call IAlignment surface.
 *
 * ngmlr calls SingleAlign once per interval from each worker thread and needs the result
 * before it can trim the next interval (reference src/AlignmentBuffer.cpp:3361-3406,
 * :291-425), so one worker can never keep a GPU busy.  BatchingAligner is an IAlignment
 * that many worker threads share.  Each SingleAlign prepares its request (string lengths,
 * the offsetInMatrix side effect) in the caller's thread, parks it in a queue and sleeps.
 * ONE dispatcher thread per BatchingAligner -- the only thread that talks to the device
 * handle -- cuts the queue into launches and keeps up to two of them in flight through the
 * streaming ABI (cvx_submit / cvx_wait): while launch k runs, the requests that arrive form
 * launch k+1, whose upload and corridor analysis overlap launch k's kernels.  When a launch
 * has finished, the dispatcher hands every parked worker its result record and wakes it; the
 * worker writes its own CIGAR / MD / NM into its own Align (the text stage runs on as many
 * threads as there are workers), and the last one returns the launch's buffers.
 *
 * A launch is cut when   the queue holds maxBatch requests,
 *                   or   every registered worker is parked (nobody else can add to it),
 *                   or   (only with a batch target, CVX_BATCH_TARGET > 0: many more alignment contexts than cores,
 *                        align_pool.h) -- while fewer than `target` requests wait, only once the oldest has waited
 *                        holdUs (CVX_BATCH_HOLD_US), or at once when every registered worker is parked and nobody is left
 *                        to feed the contexts (SetFeedActive(false)); from `target` requests on the rules below apply,
 *                   or   the oldest request has waited timeoutUs (while no launch has been timed yet), resp. the launch
 *                        that is running is expected to end within leadUs (the time an upload + corridor analysis
 *                        take: what arrives before that travels for free),
 *                   or   the device is idle and something is queued (latency before batch size
 *                        when there is nothing to overlap with).
 *
 * Errors stay with their request: a hard error of one tile (corridor no kernel covers, CIGAR
 * that does not fit) is thrown in that worker's thread only -- the reference's caller drops
 * exactly that alignment (src/AlignmentBuffer.cpp:454-463); only a failure of a whole launch
 * fails all of its requests.
 *
 * Usage in ngmlr: one BatchingAligner per device, handed to every AlignmentBuffer in place
 * of its private aligner (SharedAligner below does that without touching the pipeline).
 */
#ifndef BATCHING_ALIGNER_H
#define BATCHING_ALIGNER_H

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "convex_align_hip.h"
#include "cvx_fiber.h"

namespace Convex {

class BatchingAligner: public IAlignment {
public:
	/* backend is owned by the caller.  maxBatch: cut a launch when this many requests wait;
	 * timeoutUs: cut one when the oldest request has waited this long (0 = only the other rules). */
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
	/* batch target of the dispatcher (see the rules above; 0 = none); before the first request */
	void SetBatchTarget(int targetRequests, int holdMicroseconds);
	/* whether anybody outside the registered workers may still bring requests (align_pool.h: CS threads attached, or reads
	 * queued for a context).  While that is so a launch below the batch target waits for company up to holdUs; once it is
	 * not, "every registered worker is parked" cuts the launch at once (tail of a run, sparse input: ADVICE r4) */
	void SetFeedActive(bool active);

	/* statistics */
	long Launches() const { return launches; }
	long Requests() const { return requests; }
	long MaxInFlight() const { return maxInFlight; }
	/* seconds all workers together spent parked in SingleAlign (queue + launch + wake-up), seconds they spent in their own
	 * text stage, and seconds the dispatcher had at least one launch in flight: with the wall time and the worker count
	 * these say whether the pipeline around the aligner or the device path is the limit */
	double ParkedSeconds() const { return parkedNs * 1e-9; }
	double FinishSeconds() const { return finishNs * 1e-9; }
	double BusySeconds() const { return busyNs * 1e-9; }
	long TextLaunches() const { return textLaunches; }        /* launches whose text stage ran on the device (CVX_DEVICE_TEXT=1) */
	double TextSeconds() const { return textNs * 1e-9; }      /* the dispatcher's time in those calls */

private:
	struct Launch;
	struct Request {
		ConvexAlignHip::Tile tile;
		Launch * launch;                   /* set when the dispatcher has taken the request */
		cvx_result const * result;         /* set when the launch is done */
		bool done;
		bool failed;                       /* the whole launch failed */
		Fiber * fiber;                     /* the caller runs on a user-level context (cvx_fiber.h): woken with FiberApi::Wake, not through cv */
		std::condition_variable cv;        /* its own worker sleeps here: a finished launch wakes its own workers only (with
		                                    * hundreds of contexts parked, one shared condition woke them all per launch) */
	};
	struct Launch {
		cvx_job job;
		std::vector<Request *> reqs;
		cvx_result const * results;
		uint32_t const * ops;
		std::chrono::steady_clock::time_point cutAt, oldestAt;      /* CVX_LAUNCH_TRACE: when the launch was cut, when its oldest request arrived */
		double submitMs, waitMs;            /* CVX_LAUNCH_TRACE: the dispatcher's time inside Submit (tile table + cvx_submit) and blocked in Wait */
		int unfinished;                    /* workers still writing their text out of this launch's buffers */
		bool failed;
		ConvexAlignHip::JobText * text;    /* CVX_DEVICE_TEXT=1: the launch's text stage ran on the device (else 0) */
	};
	ConvexAlignHip * backend;
	std::mutex mtx;
	std::condition_variable cvDispatch;    /* something to do for the dispatcher */
	std::vector<Request *> queue;
	std::chrono::steady_clock::time_point oldest;
	std::deque<Launch *> inFlight;         /* submitted, not yet waited for (dispatcher only) */
	std::vector<Launch *> retired;         /* all workers done: buffers to give back (by the dispatcher) */
	int workers;
	int parked;
	int maxBatch;
	int timeoutUs;
	int target;                                 /* batch target (CVX_BATCH_TARGET, 0 = none) and how long the oldest request may */
	int holdUs;                                 /* wait for it (CVX_BATCH_HOLD_US) */
	bool feedActive;                            /* see SetFeedActive (true until told otherwise) */
	long textLaunches; long long textNs;
	bool launchTrace;                           /* CVX_LAUNCH_TRACE=1: one stderr line per finished launch */
	bool deviceText;                            /* CVX_DEVICE_TEXT=1: cvx_job_text + cvx_job_nm_profile per launch instead of one
	                                             * host text stage per request on the workers (measured: DESIGN.md 6) */
	bool stop;
	long launches, requests, maxInFlight;
	long long parkedNs, finishNs, busyNs;       /* under mtx */
	int maxFlight;                              /* launches the dispatcher keeps in flight (CVX_BATCH_INFLIGHT, default 2) */
	double emaServiceUs;                        /* how long a launch lasts once it is the oldest in flight (moving average) */
	int leadUs;                                 /* cut the next launch this long before the running one should end (CVX_BATCH_LEAD_US) */
	std::chrono::steady_clock::time_point frontSince;   /* when the oldest launch in flight became the oldest */
	std::chrono::steady_clock::time_point busySince;
	std::thread dispatcher;
	/* The text stage of a finished launch (CVX_DEVICE_TEXT=1) off the dispatcher's critical path: its own thread on the
	 * handle's text stream, under the kernels of the launches that follow.  cvx_job_text /
	 * cvx_job_nm_profile touch only the finished job's own buffers and that stream; the dispatcher keeps cutting, submitting
	 * and waiting meanwhile.  A launch's requests are woken by whichever thread completes it. */
	std::thread textThread;
	std::deque<Launch *> textQueue;        /* waited for, text stage not yet run (under mtx) */
	std::condition_variable cvText;
	int textPending;                       /* launches in textQueue or in the text thread's hands */
	bool textStop;
	int maxTextAhead;                      /* CVX_TEXT_AHEAD (4): launches the dispatcher may have handed to the text thread before it stops cutting new ones */

	void dispatchLoop();
	void textLoop();
	void completeLaunch(Launch * l);       /* with mtx held: hands every request of the launch its result and wakes it */
	bool shouldCut(bool deviceIdle) const;
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
	/* logical devices that have a dispatcher right now (CVX_ALIAS_DEVICES puts several on one physical device) */
	static int ActiveDevices();

	/* Pool accounting (align_pool.h): with K >> cores alignment contexts most aligner fronts belong to threads that hold
	 * no read, and "every registered worker is parked" must count only those that do.  After UsePoolAccounting(true) a
	 * SharedAligner no longer registers for its lifetime; the thread that constructed it registers between ThreadBegin()
	 * and ThreadEnd() (no-ops for a thread without such an aligner). */
	static void UsePoolAccounting(bool on, int batchTarget = 256, int holdMicroseconds = 30000);      /* + the dispatchers' batch target under the pool */
	static void ThreadBegin();
	static void ThreadEnd();
	/* the pool's producers are gone and its queue is empty (false) / a producer attached (true): passed on to every dispatcher */
	static void SetFeedActive(bool active);

private:
	BatchingAligner * shared;
	int device;
	bool perRead;      /* registers per read (pool accounting), not for its lifetime */
};

}  // namespace Convex

#endif
