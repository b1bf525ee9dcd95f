/*
 * parking_cpu_aligner.h -- test-only IAlignment for oracle/_ref/ngmlr_pool_parked (tools/build_ngmlr_hip.sh): the
 * reference's own Convex::ConvexAlignFast, but every SingleAlign first gives its caller's user-level context back to the
 * carrier thread (FiberApi::Park) and is woken by a "dispatcher" thread a little later -- the control flow of
 * SharedAligner::SingleAlign on the device path (ngmlr_amd/csrc/batching_aligner.cpp), with the CPU aligner as the
 * compute.  tests/test_pool_cpu.py runs ngmlr's whole long-read stage through it: thousands of park / resume cycles in
 * the middle of processLongReadLIS, SAM identical to the unmodified reference, no GPU.  Compiled only inside ngmlr's tree.
 */
#ifndef PARKING_CPU_ALIGNER_H
#define PARKING_CPU_ALIGNER_H

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

/* (ConvexAlignFast.h comes from AlignmentBuffer.h right in front of this header; its include guard does not survive a second inclusion) */
#include "cvx_fiber.h"

namespace Convex {

class ParkingCpuAligner: public IAlignment {
	struct Waker {
		std::mutex mtx;
		std::condition_variable cv;
		std::vector<Fiber *> waiting;
		bool stop;
		long parks, wakes, largest;
		std::thread th;
		Waker() : stop(false), parks(0), wakes(0), largest(0) { th = std::thread([this] { loop(); }); }
		~Waker() {
			{ std::lock_guard<std::mutex> lk(mtx); stop = true; }
			cv.notify_all();
			th.join();
			fprintf(stderr, "ParkingCpuAligner: %ld parks, %ld wakes, at most %ld contexts woken at once\n", parks, wakes, largest);
		}
		void loop() {
			std::unique_lock<std::mutex> lk(mtx);
			for (;;) {
				if (waiting.empty()) { if (stop) return; cv.wait(lk); continue; }
				lk.unlock();
				std::this_thread::sleep_for(std::chrono::microseconds(400));      /* a "launch": company arrives meanwhile */
				lk.lock();
				std::vector<Fiber *> batch;
				batch.swap(waiting);
				if ((long) batch.size() > largest) largest = (long) batch.size();
				wakes += (long) batch.size();
				for (Fiber * f : batch) FiberApi::Wake(f);
			}
		}
		void park() {
			Fiber * const f = FiberApi::Current();
			if (f == 0) return;      /* the pthread form of the pool: nothing to give back */
			{ std::lock_guard<std::mutex> lk(mtx); waiting.push_back(f); parks += 1; }
			cv.notify_one();
			FiberApi::Park();
		}
	};
	static Waker & waker() { static Waker w; return w; }
	ConvexAlignFast inner;
public:
	ParkingCpuAligner(int const stdOutMode, float const match, float const mismatch, float const gapOpen, float const gapExtend,
			float const gapExtendMin, float const gapDecay) : inner(stdOutMode, match, mismatch, gapOpen, gapExtend, gapExtendMin, gapDecay) { (void) waker(); }
	virtual int GetScoreBatchSize() const { return inner.GetScoreBatchSize(); }
	virtual int GetAlignBatchSize() const { return inner.GetAlignBatchSize(); }
	virtual int BatchScore(int const mode, int const batchSize, char const * const * const refSeqList, char const * const * const qrySeqList,
			float * const results, void * extData) { return inner.BatchScore(mode, batchSize, refSeqList, qrySeqList, results, extData); }
	virtual int BatchAlign(int const mode, int const batchSize, char const * const * const refSeqList, char const * const * const qrySeqList,
			Align * const results, void * extData) { return inner.BatchAlign(mode, batchSize, refSeqList, qrySeqList, results, extData); }
	virtual int SingleAlign(int const mode, int const corridor, char const * const refSeq, char const * const qrySeq, Align & result, void * extData) {
		return inner.SingleAlign(mode, corridor, refSeq, qrySeq, result, extData);
	}
	virtual int SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight, char const * const refSeq, char const * const qrySeq,
			Align & result, int const externalQStart, int const externalQEnd, void * extData) {
		waker().park();
		return inner.SingleAlign(mode, corridor, corridorHeight, refSeq, qrySeq, result, externalQStart, externalQEnd, extData);
	}
};

}  // namespace Convex

#endif
