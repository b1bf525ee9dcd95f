/*
 * fiber_test.cpp -- the user-level context runtime (ngmlr_amd/csrc/cvx_fiber.{h,cpp}) without a device and without ngmlr.
 *
 * The shape it is built for (ngmlr_amd/csrc/align_pool.cpp, batching_aligner.cpp): producers submit reads; a read runs on
 * a fiber, and several times in its life it hands a request to ONE dispatcher thread and parks; the dispatcher collects
 * requests into launches and wakes every fiber of a finished launch.  Here a "read" is a little arithmetic chain whose
 * every step goes through the dispatcher, so a lost wake-up, a double resume, a fiber resumed on the wrong stack or a
 * clobbered fiber-local shows up as a wrong sum, a hang (the caller runs this under a timeout) or a crash.
 *
 *   fiber_test <carriers> <fibers> <items> <parks per item> [immediate]
 * "immediate": the dispatcher wakes a request the moment it sees it -- the Wake-before-Park race on every request.
 * Prints one line "ok ..." and exits 0, or says what went wrong and exits 1.
 */
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "cvx_fiber.h"

using namespace Convex;

namespace {

struct Request {
	Fiber * fiber;
	long in, out;
	bool done;
};

struct Dispatcher {
	std::mutex mtx;
	std::condition_variable cv;
	std::vector<Request *> queue;
	bool stop = false, immediate = false;
	long launches = 0, served = 0, largest = 0;
	std::thread th;

	void loop() {
		std::unique_lock<std::mutex> lk(mtx);
		for (;;) {
			if (queue.empty()) {
				if (stop) break;
				cv.wait(lk);
				continue;
			}
			if (!immediate) {
				/* let company arrive, as the real dispatcher does */
				lk.unlock();
				std::this_thread::sleep_for(std::chrono::microseconds(300));
				lk.lock();
			}
			std::vector<Request *> batch;
			batch.swap(queue);
			launches += 1;
			served += (long) batch.size();
			if ((long) batch.size() > largest) largest = (long) batch.size();
			for (Request * r : batch) {
				r->out = r->in * 3 + 1;
				r->done = true;
				FiberApi::Wake(r->fiber);      /* under the lock, like BatchingAligner::dispatchLoop */
			}
		}
	}

	long call(long v) {      /* on a fiber: BatchingAligner::SingleAlign's shape */
		Request req;
		req.fiber = FiberApi::Current();
		req.in = v; req.out = 0; req.done = false;
		std::unique_lock<std::mutex> lk(mtx);
		queue.push_back(&req);
		cv.notify_one();
		while (!req.done) {
			lk.unlock();
			FiberApi::Park();
			lk.lock();
		}
		return req.out;
	}
};

struct Item { long seed; int parks; long result; };

struct Env {
	Dispatcher disp;
	std::atomic<long> slotsMade{0}, slotsFreed{0}, ran{0}, localErrors{0}, lastTaken{0};
};

struct Slot { long uses; char pad[4096]; };

void runItem(void * user, void ** slot, void * itemPtr) {
	Env * env = (Env *) user;
	Item * it = (Item *) itemPtr;
	if (*slot == 0) { *slot = new Slot(); ((Slot *) *slot)->uses = 0; env->slotsMade += 1; }
	Slot * s = (Slot *) *slot;
	s->uses += 1;
	Fiber * self = FiberApi::Current();
	/* a fiber-local must survive every park, whatever ran on the carrier meanwhile */
	FiberApi::Local(self, 1) = (void *) it;
	volatile char deep[8192];      /* touch some stack below the frame */
	memset((void *) deep, (int) (it->seed & 0x7f), sizeof(deep));
	long v = it->seed;
	for (int p = 0; p < it->parks; ++p) {
		v = env->disp.call(v) % 1000003;
		if (FiberApi::Current() != self || FiberApi::Local(self, 1) != (void *) it) env->localErrors += 1;
	}
	if (deep[100] != (char) (it->seed & 0x7f)) env->localErrors += 1;
	it->result = v;
	env->ran += 1;
}
void destroySlot(void * user, void * slot) { ((Env *) user)->slotsFreed += 1; delete (Slot *) slot; }
void lastTaken(void * user) { ((Env *) user)->lastTaken += 1; }

}  // namespace

int main(int argc, char ** argv) {
	int const carriers = argc > 1 ? atoi(argv[1]) : 4;
	int const fibers = argc > 2 ? atoi(argv[2]) : 512;
	int const items = argc > 3 ? atoi(argv[3]) : 20000;
	int const parks = argc > 4 ? atoi(argv[4]) : 3;
	Env env;
	env.disp.immediate = argc > 5 && strcmp(argv[5], "immediate") == 0;
	env.disp.th = std::thread([&env] { env.disp.loop(); });

	FiberPool::Callbacks cb;
	cb.user = &env;
	cb.run = &runItem;
	cb.destroySlot = &destroySlot;
	cb.carrierStart = 0;
	cb.lastItemTaken = &lastTaken;
	std::vector<Item> work((size_t) items);
	for (int i = 0; i < items; ++i) { work[(size_t) i].seed = 17 + 31L * i; work[(size_t) i].parks = parks + (i % 3 == 0 ? 1 : 0) - (i % 7 == 0 && parks > 0 ? 1 : 0); work[(size_t) i].result = -1; }
	FiberPool::Stats st;
	{
		FiberPool pool(carriers, fibers, 256 * 1024, fibers / 2 > 4 ? fibers / 2 : 4, cb);
		/* four producers, like ngmlr's CS threads */
		std::vector<std::thread> prod;
		for (int t = 0; t < 4; ++t) prod.emplace_back([&, t] { for (int i = t; i < items; i += 4) pool.Submit(&work[(size_t) i]); });
		for (std::thread & t : prod) t.join();
		pool.CloseFeed();
		pool.DrainAndStop();
		st = pool.GetStats();
	}
	{
		std::lock_guard<std::mutex> lk(env.disp.mtx);
		env.disp.stop = true;
	}
	env.disp.cv.notify_all();
	env.disp.th.join();

	long bad = 0, totalParks = 0;
	for (int i = 0; i < items; ++i) {
		Item const & it = work[(size_t) i];
		long v = it.seed;
		for (int p = 0; p < it.parks; ++p) v = (v * 3 + 1) % 1000003;
		if (v != it.result) bad += 1;
		totalParks += it.parks;
	}
	bool ok = bad == 0 && env.ran == items && env.localErrors == 0 && env.slotsMade == env.slotsFreed && env.slotsMade <= fibers
			&& st.items == items && st.fibers == env.slotsMade && st.maxInFlight <= fibers && st.parks == totalParks && env.lastTaken >= 1
			&& env.disp.served == totalParks;
	printf("%s: %d items on %ld fibers (limit %d) over %d carriers: %ld wrong, %ld fiber-local errors, %ld parks (expected %ld), at most %ld in flight, %ld queued, "
			"%ld dispatcher launches (largest %ld), slots made / freed %ld / %ld\n", ok ? "ok" : "FAILED", items, st.fibers, fibers, carriers, bad, env.localErrors.load(),
			st.parks, totalParks, st.maxInFlight, st.maxQueued, env.disp.launches, env.disp.largest, env.slotsMade.load(), env.slotsFreed.load());
	return ok ? 0 : 1;
}
