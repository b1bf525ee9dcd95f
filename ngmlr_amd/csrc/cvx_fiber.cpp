/* cvx_fiber.cpp -- see cvx_fiber.h.  Host-only; ucontext for the switch (two sigprocmask calls per switch: a read parks
 * once or twice in its life, the cost is nowhere). */
#include "cvx_fiber.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

namespace Convex {

namespace {
typedef std::chrono::steady_clock Clock;
struct Carrier;
}

struct Fiber {
	ucontext_t ctx;
	Carrier * carrier;
	FiberPool::Impl * pool;
	void * map;                /* guard page + stack */
	size_t mapBytes;
	void * slot;               /* the user's per-fiber context object */
	void * item;               /* what it is running */
	void * local[FiberApi::kSlots];
	enum State { FREE, RUNNING, PARKED, RUNNABLE } state;      /* under the pool's lock */
	bool pendingWake;          /* under the pool's lock: a Wake arrived while the fiber was not parked */
	bool finished;             /* written by the fiber right before it switches back: its item is done */
	Clock::time_point tookAt;
};

namespace {
struct Carrier {
	int index;
	int quota;                 /* fibers this carrier may create */
	std::thread thread;
	ucontext_t sched;
	std::vector<Fiber *> all, freeList;
	std::deque<Fiber *> runnable;
	std::condition_variable cv;
	bool sleeping;
	Carrier() : index(0), quota(1), sleeping(false) {}
};
thread_local Fiber * tl_current = 0;
}

struct FiberPool::Impl {
	mutable std::mutex mtx;              /* ONE lock: the queue, every carrier's lists, every fiber's state.  Held for list operations only */
	std::condition_variable cvSpace, cvDrained;
	std::deque<void *> queue;
	std::vector<Carrier *> carriers;
	Callbacks cb;
	size_t stackBytes;
	int queueLimit;
	int inFlight;
	unsigned nextCarrier;
	bool stop, feedClosed, joined;
	Stats st;
	long long producerBlockedNs, holdingNs, runningNs;

	Fiber * createFiber(Carrier * c);
	void carrierMain(Carrier * c);
	static void fiberEntry();
};

void FiberPool::Impl::fiberEntry() {
	Fiber * const f = tl_current;
	for (;;) {
		f->pool->cb.run(f->pool->cb.user, &f->slot, f->item);
		f->finished = true;
		swapcontext(&f->ctx, &f->carrier->sched);
	}
}

Fiber * FiberPool::Impl::createFiber(Carrier * c) {
	size_t const page = (size_t) sysconf(_SC_PAGESIZE);
	size_t const stack = (stackBytes + page - 1) / page * page;
	Fiber * f = new Fiber();
	f->carrier = c;
	f->pool = this;
	f->mapBytes = stack + page;
	/* untouched pages cost nothing: a read that never goes deep keeps a few pages resident */
	f->map = mmap(0, f->mapBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
	if (f->map == MAP_FAILED) {
		perror("FiberPool: mmap of a fiber stack");
		abort();
	}
	mprotect(f->map, page, PROT_NONE);      /* an overflow faults instead of running into the neighbour */
	f->slot = 0;
	f->item = 0;
	for (int i = 0; i < FiberApi::kSlots; ++i) f->local[i] = 0;
	f->state = Fiber::FREE;
	f->pendingWake = false;
	f->finished = false;
	getcontext(&f->ctx);
	f->ctx.uc_stack.ss_sp = (char *) f->map + page;
	f->ctx.uc_stack.ss_size = stack;
	f->ctx.uc_link = 0;
	makecontext(&f->ctx, (void (*)()) &Impl::fiberEntry, 0);
	return f;
}

void FiberPool::Impl::carrierMain(Carrier * c) {
	if (cb.carrierStart) cb.carrierStart(cb.user, c->index);
	std::unique_lock<std::mutex> lk(mtx);
	for (;;) {
		Fiber * f = 0;
		bool tookLast = false;
		if (!c->runnable.empty()) {
			/* reads whose launch has come back go first: they finish, and free their fiber for the queue */
			f = c->runnable.front();
			c->runnable.pop_front();
		} else if (!queue.empty() && (!c->freeList.empty() || (int) c->all.size() < c->quota)) {
			void * const item = queue.front();
			queue.pop_front();
			tookLast = feedClosed && queue.empty();
			inFlight += 1;
			if ((long) inFlight > st.maxInFlight) st.maxInFlight = inFlight;
			if (!c->freeList.empty()) {
				f = c->freeList.back();
				c->freeList.pop_back();
			} else {
				lk.unlock();
				f = createFiber(c);
				lk.lock();
				c->all.push_back(f);
				st.fibers += 1;
			}
			f->item = item;
			f->tookAt = Clock::now();
			cvSpace.notify_one();
		}
		if (f != 0) {
			f->state = Fiber::RUNNING;
			lk.unlock();
			if (tookLast && cb.lastItemTaken) cb.lastItemTaken(cb.user);
			Clock::time_point const t0 = Clock::now();
			tl_current = f;
			swapcontext(&c->sched, &f->ctx);
			tl_current = 0;
			Clock::time_point const t1 = Clock::now();
			lk.lock();
			runningNs += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
			if (f->finished) {
				f->finished = false;
				f->state = Fiber::FREE;
				f->item = 0;
				c->freeList.push_back(f);
				holdingNs += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - f->tookAt).count();
				inFlight -= 1;
				if (inFlight == 0 && queue.empty()) cvDrained.notify_all();
			} else {
				/* it parked.  A Wake that came while it was still on its way out made it runnable already */
				st.parks += 1;
				if (f->pendingWake) {
					f->pendingWake = false;
					f->state = Fiber::RUNNABLE;
					c->runnable.push_back(f);
				} else {
					f->state = Fiber::PARKED;
				}
			}
			continue;
		}
		if (stop) break;
		c->sleeping = true;
		c->cv.wait(lk);
		c->sleeping = false;
	}
	lk.unlock();
	/* the run is over: every fiber is FREE.  Their context objects go on this thread (AlignPool: ~AlignmentBuffer flushes
	 * the context's SAM records) */
	for (Fiber * f : c->all) {
		if (f->slot != 0 && cb.destroySlot) cb.destroySlot(cb.user, f->slot);
		munmap(f->map, f->mapBytes);
		delete f;
	}
	c->all.clear();
	c->freeList.clear();
}

FiberPool::FiberPool(int nCarriers, int maxFibers, size_t stackBytes, int queueLimit, Callbacks const & cb) : impl(new Impl()) {
	if (nCarriers < 1) nCarriers = 1;
	if (maxFibers < nCarriers) maxFibers = nCarriers;
	impl->cb = cb;
	impl->stackBytes = stackBytes < 65536 ? 65536 : stackBytes;
	impl->queueLimit = queueLimit > 0 ? queueLimit : 1;
	impl->inFlight = 0;
	impl->nextCarrier = 0;
	impl->stop = impl->feedClosed = impl->joined = false;
	impl->producerBlockedNs = impl->holdingNs = impl->runningNs = 0;
	impl->st = Stats();
	impl->st.carriers = nCarriers;
	for (int i = 0; i < nCarriers; ++i) {
		Carrier * c = new Carrier();
		c->index = i;
		c->quota = maxFibers / nCarriers + (i < maxFibers % nCarriers ? 1 : 0);
		impl->carriers.push_back(c);
	}
	/* (threads start once every Carrier exists: Submit walks the vector) */
	for (Carrier * c : impl->carriers) {
		Impl * const im = impl;
		c->thread = std::thread([im, c] { im->carrierMain(c); });
	}
}

FiberPool::~FiberPool() {
	if (!impl->joined) DrainAndStop();
	for (Carrier * c : impl->carriers) delete c;
	delete impl;
}

void FiberPool::Submit(void * item) {
	std::unique_lock<std::mutex> lk(impl->mtx);
	if ((int) impl->queue.size() >= impl->queueLimit) {
		Clock::time_point const t0 = Clock::now();
		while ((int) impl->queue.size() >= impl->queueLimit) impl->cvSpace.wait(lk);
		impl->producerBlockedNs += std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count();
	}
	impl->queue.push_back(item);
	impl->st.items += 1;
	if ((long) impl->queue.size() > impl->st.maxQueued) impl->st.maxQueued = (long) impl->queue.size();
	/* wake ONE sleeping carrier that can take it (a free fiber, or room for a new one); carriers that are running look
	 * at the queue after their current slice by themselves */
	size_t const n = impl->carriers.size();
	for (size_t k = 0; k < n; ++k) {
		Carrier * c = impl->carriers[(impl->nextCarrier + k) % n];
		if (c->sleeping && (!c->freeList.empty() || (int) c->all.size() < c->quota)) {
			c->sleeping = false;      /* the next Submit picks another one */
			c->cv.notify_one();
			impl->nextCarrier = (unsigned) ((impl->nextCarrier + k + 1) % n);
			break;
		}
	}
}

void FiberPool::CloseFeed() {
	bool emptyNow;
	{
		std::lock_guard<std::mutex> lk(impl->mtx);
		impl->feedClosed = true;
		emptyNow = impl->queue.empty();
	}
	if (emptyNow && impl->cb.lastItemTaken) impl->cb.lastItemTaken(impl->cb.user);
}

void FiberPool::DrainAndStop() {
	{
		std::unique_lock<std::mutex> lk(impl->mtx);
		if (impl->joined) return;
		while (!(impl->queue.empty() && impl->inFlight == 0)) impl->cvDrained.wait(lk);
		impl->stop = true;
		for (Carrier * c : impl->carriers) c->cv.notify_one();
	}
	for (Carrier * c : impl->carriers) c->thread.join();
	impl->joined = true;
}

FiberPool::Stats FiberPool::GetStats() const {
	std::lock_guard<std::mutex> lk(impl->mtx);
	Stats s = impl->st;
	s.producerBlockedSeconds = (double) impl->producerBlockedNs * 1e-9;
	s.holdingSeconds = (double) impl->holdingNs * 1e-9;
	s.runningSeconds = (double) impl->runningNs * 1e-9;
	return s;
}

Fiber * FiberApi::Current() { return tl_current; }

void FiberApi::Park() {
	Fiber * const f = tl_current;
	if (f == 0) {
		fprintf(stderr, "FiberApi::Park called outside a fiber\n");
		abort();
	}
	swapcontext(&f->ctx, &f->carrier->sched);
}

void FiberApi::Wake(Fiber * f) {
	std::lock_guard<std::mutex> lk(f->pool->mtx);
	if (f->state == Fiber::PARKED) {
		f->state = Fiber::RUNNABLE;
		Carrier * const c = f->carrier;
		c->runnable.push_back(f);
		if (c->sleeping) {
			c->sleeping = false;
			c->cv.notify_one();
		}
	} else {
		f->pendingWake = true;      /* still on its way out (or not parked yet): its carrier requeues it when it arrives */
	}
}

void *& FiberApi::Local(Fiber * f, int slot) { return f->local[slot]; }

}  // namespace Convex
