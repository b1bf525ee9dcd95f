/*
 * align_pool.h -- reads in flight decoupled from ngmlr's CS threads (second half).
 *
 * ngmlr runs a read's whole long-read stage -- LIS over the scored sub-reads, interval building, one
 * SingleAlign per interval (each waits for the one before: reference src/AlignmentBuffer.cpp:3361-3406),
 * SV checks, SAM record -- synchronously on the CS thread that finished scoring the read's last sub-read
 * (`out->processLongReadLIS(group)`, reference src/ScoreBuffer.cpp:152-159 and src/CS.cpp:293-297).  One read
 * in flight per `-t` thread, so a device launch can never carry more tiles than there are CS threads, and
 * `-t` is bounded by the host's cores and by ~430 MB of search tables per CS thread (src/CS.cpp:422-426).
 *
 * AlignPool takes that call off the CS thread: the group is queued and one of K >> t "alignment contexts"
 * runs it.  A context is a thread with its own AlignmentBuffer (own aligner front, own SAM buffer, none of
 * the CS thread's search tables), i.e. exactly the object the reference gives each CS thread
 * (src/CS.cpp:412-419) -- the read's code path, its data and the thread-safety assumptions (one
 * AlignmentBuffer per thread, shared state behind NGM's own locks and atomics) are the reference's own.
 * Contexts spend their time parked in SharedAligner::SingleAlign (batching_aligner.h), so K of them cost K
 * stacks, not K cores; the CS threads stay at the host's real core count and keep searching and scoring.
 * With K = 512 (CVX_POOL_CONTEXTS; they are created on demand) a launch carries hundreds of tiles instead of ~20.
 *
 * Binding (applied by tools/build_ngmlr_hip.sh to its /tmp copy of the reference, shown in INTEGRATION.md):
 *   src/CS.cpp  CS::DoRun      AlignPool::Attach() before the thread builds its own buffers,
 *                              AlignPool::Detach() after its read loop (the last CS thread drains the pool,
 *                              deletes the contexts -- their SAM buffers flush -- and only then returns)
 *   src/CS.cpp:296, src/ScoreBuffer.cpp:155        out->processLongReadLIS(g)  ->  AlignPool::Submit(g)
 *   src/ScoreBuffer.cpp:159, :283                  out->processShortRead(r)    ->  AlignPool::SubmitShort(r)
 * Output order changes (it already depends on the thread schedule in the reference); the set of SAM records
 * does not: tests/test_gpu_e2e.py compares sorted records with the unmodified reference's.
 *
 * Only meaningful inside ngmlr's tree (CVX_IN_NGMLR_TREE): it drives the reference's own AlignmentBuffer.
 */
#ifndef CVX_ALIGN_POOL_H
#define CVX_ALIGN_POOL_H

#include "MappedRead.h"

namespace Convex {

class AlignPool {
public:
	/* a CS thread starts / ends producing (src/CS.cpp CS::DoRun) */
	static void Attach();
	static void Detach();
	/* AlignmentBuffer::processLongReadLIS(group) / processShortRead(read) on a pool context; blocks while
	 * the queue is full (CVX_POOL_QUEUE, default 2 x contexts) */
	static void Submit(ReadGroup * group);
	static void SubmitShort(MappedRead * read);
	/* Measurement only: how long ngmlr's CS threads wait for and hold the ONE lock under which reads are parsed and split into
	 * sub-reads (_NGM::GetNextReadBatch, reference src/NGM.cpp:190-244: kseq_read, the per-base copy, ~40 sub-read objects per
	 * 10 kb read) -- the serial stage of the pipeline that is not the device's.  Printed with the pool's statistics. */
	static long long ProbeNow();
	static void InputLockTimes(long long beforeLock, long long locked, long long beforeUnlock, int reads);
};

}  // namespace Convex

#endif
