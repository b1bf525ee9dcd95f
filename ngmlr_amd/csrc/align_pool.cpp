/* align_pool.cpp -- see align_pool.h.  Compiled only inside ngmlr's tree (tools/build_ngmlr_hip.sh). */
#include "align_pool.h"
#include "cvx_pcsample.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <cstring>

#include "AlignmentBuffer.h"
#include "NGM.h"
#include "batching_aligner.h"
#include "cvx_fiber.h"

namespace Convex {

namespace {

struct Item {
	ReadGroup * group;      /* long read: processLongReadLIS */
	MappedRead * read;      /* short read: processShortRead */
};

struct Pool {
	std::mutex mtx;
	std::condition_variable cvWork;      /* contexts: something is queued, or stop */
	std::condition_variable cvSpace;     /* producers: the queue has room again */
	std::condition_variable cvDrained;   /* Detach: queue empty and nobody busy */
	std::deque<Item> queue;
	std::vector<std::thread> threads;
	int maxContexts, queueLimit;
	int idle, busy;
	bool stop;
	bool feedDone;          /* no producer is left (drainAndStop has begun): what is queued is all there will ever be */
	long failed;            /* reads whose long-read stage threw (reported by the last Detach, which then throws) */
	/* statistics */
	long items, maxQueued, maxBusy;
	long long producerBlockedNs, busyNs;
	std::chrono::steady_clock::time_point born;

	Pool() : maxContexts(512), queueLimit(0), idle(0), busy(0), stop(false), feedDone(false), failed(0), items(0), maxQueued(0), maxBusy(0),
			producerBlockedNs(0), busyNs(0), born(std::chrono::steady_clock::now()) {
		if (const char * e = getenv("CVX_POOL_CONTEXTS")) maxContexts = atoi(e) > 0 ? atoi(e) : 1;
		queueLimit = 2 * maxContexts;
		if (const char * e = getenv("CVX_POOL_QUEUE")) queueLimit = atoi(e) > 0 ? atoi(e) : 1;
	}

	void contextMain() {
		pthread_setname_np(pthread_self(), "cvx-context");
		/* what CS::DoRun does for its own thread (reference src/CS.cpp:414-419): the constructor writes the SAM
		 * prolog once, under NGM's output lock */
		NGM.AquireOutputLock();
		AlignmentBuffer * buffer = new AlignmentBuffer(Config.getOutputFile());
		NGM.ReleaseOutputLock();
		std::unique_lock<std::mutex> lk(mtx);
		for (;;) {
			while (queue.empty() && !stop) {
				idle += 1;
				cvWork.wait(lk);
				idle -= 1;
			}
			if (queue.empty()) break;      /* stop, and nothing left */
			Item const it = queue.front();
			queue.pop_front();
			/* the last queued read has been taken and nobody will bring another: from here on a dispatcher must not hold a
			 * launch back "for company" once every context that holds a read is parked (batching_aligner.cpp, shouldCut) */
			if (feedDone && queue.empty()) SharedAligner::SetFeedActive(false);
			busy += 1;
			if (busy > maxBusy) maxBusy = busy;
			lk.unlock();
			cvSpace.notify_one();
			std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
			/* this thread counts as a worker of its device's dispatcher only while it holds a read: "every worker is
			 * parked" then means every context that could still add a tile to the launch */
			SharedAligner::ThreadBegin();
			bool threw = false;
			try {
				if (it.group != 0) buffer->processLongReadLIS(it.group);
				else buffer->processShortRead(it.read);
			} catch (...) {
				/* the reference loses the CS thread here and the run fails (NGMTask::Run logs and rethrows); a context finishes
				 * the other reads first, then the last Detach fails the run just as loudly (it used to exit 0) */
				fprintf(stderr, "AlignPool: exception while processing a read\n");
				threw = true;
			}
			SharedAligner::ThreadEnd();
			lk.lock();
			if (threw) failed += 1;
			busyNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
			busy -= 1;
			if (busy == 0 && queue.empty()) cvDrained.notify_all();
		}
		lk.unlock();
		delete buffer;       /* ~SAMWriter flushes this context's records (src/SAMWriter.h:20-22) */
	}

	void submit(Item const it) {
		std::unique_lock<std::mutex> lk(mtx);
		if ((int) queue.size() >= queueLimit) {
			std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
			while ((int) queue.size() >= queueLimit) cvSpace.wait(lk);
			producerBlockedNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
		}
		queue.push_back(it);
		items += 1;
		if ((long) queue.size() > maxQueued) maxQueued = (long) queue.size();
		if (idle == 0 && (int) threads.size() < maxContexts) {
			/* contexts are created on demand: a run of a few reads never pays for K of them */
			threads.emplace_back([this] { contextMain(); });
		} else {
			cvWork.notify_one();
		}
	}

	/* -> reads that failed */
	long drainAndStop() {
		{
			std::unique_lock<std::mutex> lk(mtx);
			feedDone = true;
			if (queue.empty()) SharedAligner::SetFeedActive(false);
			while (!(queue.empty() && busy == 0)) cvDrained.wait(lk);
			stop = true;
		}
		double const drained = std::chrono::duration<double>(std::chrono::steady_clock::now() - born).count();
		cvWork.notify_all();
		for (std::thread & t : threads) t.join();
		double const wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - born).count();
		if (getenv("CVX_TIMELINE")) fprintf(stderr, "cvx timeline: AlignPool drained %.2f s after its first producer, contexts joined (SAM buffers flushed) at %.2f s\n", drained, wall);
		fprintf(stderr, "AlignPool: %ld reads on %zu contexts (limit %d) over %.2f s: at most %ld reads in flight, %ld queued; "
				"contexts held a read %.1f %% of their time, CS threads waited %.2f s for room in the queue\n",
				items, threads.size(), maxContexts, wall, maxBusy, maxQueued,
				threads.empty() ? 0.0 : 100.0 * (double) busyNs * 1e-9 / (wall * (double) threads.size()), (double) producerBlockedNs * 1e-9);
		return failed;
	}
};

/* The same pool on user-level contexts (cvx_fiber.h; the default, CVX_POOL_FIBERS=0 selects the pthread form above): a
 * context is a fiber with its own AlignmentBuffer, min(16, cores) carrier threads run whichever of them hold a read that is
 * not waiting for the device.  4 096 reads in flight by default (CVX_POOL_CONTEXTS), created on demand. */
struct FiberContexts {
	FiberPool * pool;
	std::atomic<long> failed;
	int maxContexts, carriers, queueLimit;
	std::chrono::steady_clock::time_point born;

	static void run(void * user, void ** slot, void * itemPtr) {
		FiberContexts * const self = (FiberContexts *) user;
		Item * const it = (Item *) itemPtr;
		if (*slot == 0) {
			/* What CS::DoRun does for its own thread (reference src/CS.cpp:414-419).  The output lock there protects ONE thing in
			 * the constructor: the SAM prolog, written by whichever AlignmentBuffer is built first (`if (first) { WriteProlog();
			 * first = false; }`, src/AlignmentBuffer.h:318-321).  The first context of the process is built under the lock like
			 * every CS thread's own buffer; by the time it is done the prolog is out -- every producer built its buffer under
			 * the lock before it submitted anything -- and the thousands that follow need not queue behind each other and behind
			 * every SAM flush for a constructor that allocates (2 700 of them in a 1.2 s run, CVX_POOL_LOCKED_CTOR=1 restores it). */
			static std::atomic<bool> prologOut(false);
			static bool const lockedCtor = [] { const char * e = getenv("CVX_POOL_LOCKED_CTOR"); return e && atoi(e) != 0; }();
			if (lockedCtor || !prologOut.load(std::memory_order_acquire)) {
				NGM.AquireOutputLock();
				*slot = new AlignmentBuffer(Config.getOutputFile());
				NGM.ReleaseOutputLock();
				prologOut.store(true, std::memory_order_release);
			} else {
				*slot = new AlignmentBuffer(Config.getOutputFile());
			}
		}
		AlignmentBuffer * const buffer = (AlignmentBuffer *) *slot;
		SharedAligner::ThreadBegin();      /* counts as a worker of its dispatcher only while it holds a read */
		try {
			if (it->group != 0) buffer->processLongReadLIS(it->group);
			else buffer->processShortRead(it->read);
		} catch (...) {
			fprintf(stderr, "AlignPool: exception while processing a read\n");
			self->failed += 1;
		}
		SharedAligner::ThreadEnd();
		delete it;
	}
	static void destroySlot(void *, void * slot) { delete (AlignmentBuffer *) slot; }      /* ~SAMWriter flushes this context's records */
	static void carrierStart(void *, int) { pthread_setname_np(pthread_self(), "cvx-context"); pcsample::arm_this_thread(1); }
	static void lastItemTaken(void *) { SharedAligner::SetFeedActive(false); }

	FiberContexts() : pool(0), failed(0), maxContexts(4096), carriers(16), queueLimit(0), born(std::chrono::steady_clock::now()) {
		if (const char * e = getenv("CVX_POOL_CONTEXTS")) maxContexts = atoi(e) > 0 ? atoi(e) : 1;
		unsigned const hw = std::thread::hardware_concurrency();
		if (hw > 0 && (int) hw < carriers) carriers = (int) hw;
		/* A container's CPU quota stalls EVERY thread of the process once a period's budget is spent: 16 carriers + 32 CS threads on
		 * a 16-core quota were throttled for 15 s of thread time in a 1.7 s run, and the dispatcher's cvx_submit with them
		 * (profiles/r06_e2e_thread_matrix.txt).  The carriers take five eighths of what the cgroup allows -- at steady state the
		 * contexts need twice the CPU of the CS threads (80 000 reads: 22.4 against 11.4 CPU-s, ten carriers 28 500 reads/s against
		 * eight's 22 500, profiles/r06_e2e_steady_state.txt); the CS threads (-t) are ngmlr's. */
		{
			double cores = 0.0;
			if (FILE * f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
				char q[64];
				double period = 0.0;
				if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) cores = atof(q) / period;
				fclose(f);
			} else if (FILE * g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
				double quota = -1.0, period = 0.0;
				if (fscanf(g, "%lf", &quota) != 1) quota = -1.0;
				fclose(g);
				if (FILE * p2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(p2, "%lf", &period) != 1) period = 0.0; fclose(p2); }
				if (quota > 0.0 && period > 0.0) cores = quota / period;
			}
			if (cores >= 1.0) {
				int const share = (int) (cores * 0.625 + 0.5);
				carriers = std::min(carriers, std::max(2, share));
			}
		}
		if (const char * e = getenv("CVX_POOL_CARRIERS")) carriers = atoi(e) > 0 ? atoi(e) : 1;
		if (carriers > maxContexts) carriers = maxContexts;
		queueLimit = maxContexts / 2 > 512 ? maxContexts / 2 : 512;
		if (const char * e = getenv("CVX_POOL_QUEUE")) queueLimit = atoi(e) > 0 ? atoi(e) : 1;
		size_t stackKb = 512;
		if (const char * e = getenv("CVX_POOL_STACK_KB")) stackKb = atoi(e) >= 64 ? (size_t) atoi(e) : 64;
		FiberPool::Callbacks cb;
		cb.user = this;
		cb.run = &FiberContexts::run;
		cb.destroySlot = &FiberContexts::destroySlot;
		cb.carrierStart = &FiberContexts::carrierStart;
		cb.lastItemTaken = &FiberContexts::lastItemTaken;
		pool = new FiberPool(carriers, maxContexts, stackKb * 1024, queueLimit, cb);
	}
	void submit(Item const it) { pool->Submit(new Item(it)); }
	long drainAndStop() {
		pool->CloseFeed();
		pool->DrainAndStop();
		FiberPool::Stats const s = pool->GetStats();
		double const wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - born).count();
		fprintf(stderr, "AlignPool: %ld reads on %ld user-level contexts (limit %d) over %d carrier threads in %.2f s: at most %ld reads in flight, %ld queued; "
				"%ld parks; a context held a read %.2f ms on average, the carriers ran read code %.1f %% of their time (%.2f CPU-s), CS threads waited %.2f s for room in the queue\n",
				s.items, s.fibers, maxContexts, s.carriers, wall, s.maxInFlight, s.maxQueued, s.parks,
				s.items ? 1e3 * s.holdingSeconds / (double) s.items : 0.0, 100.0 * s.runningSeconds / (wall * (double) s.carriers), s.runningSeconds,
				s.producerBlockedSeconds);
		delete pool;
		pool = 0;
		return failed.load();
	}
};

struct AnyPool {
	Pool * threads;
	FiberContexts * fibers;
	AnyPool() : threads(0), fibers(0) {
		const char * e = getenv("CVX_POOL_FIBERS");
		if (e && atoi(e) == 0) threads = new Pool();
		else fibers = new FiberContexts();
	}
	~AnyPool() { delete threads; delete fibers; }
	void submit(Item const it) { if (fibers) fibers->submit(it); else threads->submit(it); }
	long drainAndStop() { return fibers ? fibers->drainAndStop() : threads->drainAndStop(); }
};

std::atomic<long long> g_inputWaitNs(0), g_inputHeldNs(0);
std::atomic<long> g_inputBatches(0), g_inputReads(0);

std::mutex g_poolMtx;
AnyPool * g_pool = 0;
int g_producers = 0;

bool fibersWanted() {
	const char * e = getenv("CVX_POOL_FIBERS");
	return !(e && atoi(e) == 0);
}

AnyPool * poolForSubmit() {
	std::lock_guard<std::mutex> g(g_poolMtx);
	if (g_pool == 0) g_pool = new AnyPool();      /* a producer that never attached (not a reference call path) still works */
	return g_pool;
}

}  // namespace

void AlignPool::Attach() {
	pthread_setname_np(pthread_self(), "ngm-cs");      /* the calling CS thread */
	pcsample::arm_this_thread(2);
	std::lock_guard<std::mutex> g(g_poolMtx);
	/* the aligner fronts built from now on (the CS threads' own, the contexts') register with their dispatcher per
	 * read (ThreadBegin / ThreadEnd), not for their lifetime */
	/* thousands of user-level contexts: a launch waits for 2 048 tiles, 10 ms at most; 512 threads: 256 tiles, 30 ms */
	if (fibersWanted()) SharedAligner::UsePoolAccounting(true, 2048, 10000);
	else SharedAligner::UsePoolAccounting(true);
	if (g_pool == 0) g_pool = new AnyPool();
	g_producers += 1;
	SharedAligner::SetFeedActive(true);
}

void AlignPool::Detach() {
	AnyPool * last = 0;
	{
		std::lock_guard<std::mutex> g(g_poolMtx);
		g_producers -= 1;
		if (g_producers == 0) {
			/* nobody can submit any more (a CS thread that starts later simply opens a new pool) */
			last = g_pool;
			g_pool = 0;
		}
	}
	if (last != 0) {
		long const failed = last->drainAndStop();
		delete last;
		if (g_inputBatches.load() > 0) fprintf(stderr, "AlignPool: ngmlr's input lock (parse + split of the reads, one thread at a time): held %.2f s for %ld reads in %ld batches (%.1f us per read), "
				"CS threads waited %.2f s for it\n", (double) g_inputHeldNs.load() * 1e-9, g_inputReads.load(), g_inputBatches.load(),
				g_inputReads.load() ? (double) g_inputHeldNs.load() * 1e-3 / (double) g_inputReads.load() : 0.0, (double) g_inputWaitNs.load() * 1e-9);
		if (failed > 0) {
			fprintf(stderr, "AlignPool: %ld read(s) failed in their long-read stage: the run is incomplete\n", failed);
			throw "AlignPool: reads failed";      /* into NGMTask::Run of the last CS thread: logged and rethrown, as for a CS thread's own exception */
		}
	}
}

long long AlignPool::ProbeNow() {
	return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void AlignPool::InputLockTimes(long long beforeLock, long long locked, long long beforeUnlock, int reads) {
	g_inputWaitNs += locked - beforeLock;
	g_inputHeldNs += beforeUnlock - locked;
	g_inputBatches += 1;
	g_inputReads += reads;
}

void AlignPool::Submit(ReadGroup * group) {
	Item it; it.group = group; it.read = 0;
	poolForSubmit()->submit(it);
}

void AlignPool::SubmitShort(MappedRead * read) {
	Item it; it.group = 0; it.read = read;
	poolForSubmit()->submit(it);
}

}  // namespace Convex
