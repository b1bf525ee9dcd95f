/*
 * cvx_fiber.h -- reads in flight that are not OS threads.
 *
 * ngmlr keeps one read per worker and a read's interval loop is sequential: every SingleAlign waits for the one before
 * (reference src/AlignmentBuffer.cpp:3361-3406, retry loop :291-425, realign :1551-1776).  Round 4's AlignPool ran
 * processLongReadLIS on K pthread "alignment contexts" parked inside SharedAligner::SingleAlign; K = 512 threads gave
 * launches of 222-256 tiles on a device that holds 7 000 waves, and K = 1 024 lost to the host's CPU quota (thread
 * creation, wake-ups, the scheduler).  A FiberPool keeps the reference's code path untouched -- a read still runs
 * processLongReadLIS from top to bottom on its own stack with its own AlignmentBuffer -- but the stack belongs to a
 * user-level context: where the pthread form slept on a condition variable, the fiber switches back to its carrier
 * thread (Fiber::Park), the carrier runs another read's host stage, and the dispatcher marks the fiber runnable when
 * its launch has come back (Fiber::Wake).  min(16, cores) carriers run thousands of reads in flight; a parked read
 * costs its touched stack pages and its AlignmentBuffer, no kernel object.
 *
 * Fibers are PINNED to the carrier that created them: thread-local addresses the compiler cached across a call stay
 * valid, glibc's per-thread state (errno, malloc arenas, locale) never migrates, and Park / Wake need no
 * cross-thread hand-over protocol -- a fiber that is woken before it has finished parking is simply found runnable by
 * its own carrier the moment it switches out.
 *
 * Generic on purpose (items are void *, the per-fiber context object is a slot the run callback fills): the runtime
 * is tested on CPU without ngmlr (tests/cpp/fiber_test.cpp); align_pool.cpp supplies the callbacks that drive the
 * reference's AlignmentBuffer.
 */
#ifndef CVX_FIBER_H
#define CVX_FIBER_H

#include <stddef.h>

namespace Convex {

struct Fiber;      /* opaque */

class FiberPool {
public:
	struct Callbacks {
		void * user;
		/* one item on a fiber.  *slot is that fiber's own context object: 0 on the fiber's first item (run creates it),
		 * kept for the fiber's lifetime.  May call Fiber::Park() any number of times.  Must not throw. */
		void (*run)(void * user, void ** slot, void * item);
		/* at shutdown, for every fiber whose slot is set, on that fiber's carrier thread (not on the fiber's stack) */
		void (*destroySlot)(void * user, void * slot);
		/* first thing a carrier thread does (may be 0) */
		void (*carrierStart)(void * user, int index);
		/* a carrier has just taken the last queued item after CloseFeed() (may be 0; called without the pool's lock) */
		void (*lastItemTaken)(void * user);
	};
	struct Stats {
		long items, fibers, maxInFlight, maxQueued, parks;
		double producerBlockedSeconds;      /* Submit callers waiting for room in the queue, summed */
		double holdingSeconds;              /* fibers between taking an item and finishing it, summed (parked time included) */
		double runningSeconds;              /* carriers executing fiber code, summed */
		int carriers;
	};

	/* carriers: OS threads; maxFibers: reads in flight at most (dealt evenly over the carriers, created on demand);
	 * stackBytes per fiber (rounded up to pages, plus one guard page); queueLimit: Submit blocks at this many waiting items */
	FiberPool(int carriers, int maxFibers, size_t stackBytes, int queueLimit, Callbacks const & cb);
	~FiberPool();                      /* DrainAndStop() first */

	void Submit(void * item);          /* any thread */
	void CloseFeed();                  /* no Submit will follow (enables lastItemTaken) */
	void DrainAndStop();               /* waits until every item has run, destroys the slots, joins the carriers */
	Stats GetStats() const;

	struct Impl;
private:
	Impl * impl;
	FiberPool(FiberPool const &);
	FiberPool & operator=(FiberPool const &);
};

/* For blocking code that may run on a fiber (batching_aligner.cpp). */
struct FiberApi {
	static Fiber * Current();                   /* 0 on a plain thread */
	/* Switch back to the carrier; returns after a Wake(this fiber).  One Wake pairs with one Park; a Wake that arrives
	 * first makes the next Park return at once. */
	static void Park();
	static void Wake(Fiber * f);                /* any thread */
	/* fiber-local storage: kSlots pointers per fiber, 0-initialised (a thread_local that travels with the read).  In use: slot 0 the
	 * read's dispatcher (batching_aligner.cpp), slots 1-3 the note of its placeholder window (DeviceWindows, convex_align_hip.cpp) */
	enum { kSlots = 4 };
	static void *& Local(Fiber * f, int slot);
};

}  // namespace Convex

#endif
