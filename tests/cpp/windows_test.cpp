/*
 * windows_test.cpp -- Convex::DeviceWindows' note of a placeholder window (ngmlr_amd/csrc/convex_align_hip.{h,cpp}) without a
 * device and without ngmlr: what window_decode_binding.inc leaves for ConvexAlignHip::Prepare must travel with the READ --
 * fiber-local while the read parks and other reads run on its carrier thread, thread-local on plain worker threads.
 *
 *   windows_test <carriers> <fibers> <items>
 * Every item notes its own (buffer, position, length), parks a few times (woken by a helper thread, like a launch's end),
 * and must find exactly its own note afterwards; a foreign buffer must not be recognised.  "ok ..." / exit 0, or exit 1.
 */
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "convex_align_hip.h"
#include "cvx_fiber.h"

using namespace Convex;

namespace {
struct Waker {
	std::mutex mtx;
	std::condition_variable cv;
	std::vector<Fiber *> queue;
	bool stop = false;
	void loop() {
		std::unique_lock<std::mutex> lk(mtx);
		for (;;) {
			if (queue.empty()) { if (stop) break; cv.wait(lk); continue; }
			std::vector<Fiber *> b;
			b.swap(queue);
			lk.unlock();
			for (Fiber * f : b) FiberApi::Wake(f);
			lk.lock();
		}
	}
	void parkOnce() {
		{ std::lock_guard<std::mutex> lk(mtx); queue.push_back(FiberApi::Current()); }
		cv.notify_one();
		FiberApi::Park();
	}
};
struct Env { Waker waker; std::atomic<long> errors{0}, ran{0}; };

bool check(char * buf, unsigned long long position, int length, char * foreign) {
	unsigned long long p = 0;
	int l = 0;
	if (!DeviceWindows::Lookup(buf, p, l) || p != position || l != length) return false;
	if (DeviceWindows::Lookup(foreign, p, l) || DeviceWindows::Lookup(0, p, l)) return false;
	if ((int) strlen(buf) != length - 1) return false;
	for (int i = 0; i < length - 1; ++i) if (buf[i] != 'x') return false;
	return true;
}

void runItem(void * user, void ** slot, void * itemPtr) {
	(void) slot;
	Env * env = (Env *) user;
	long const id = (long) (size_t) itemPtr;
	int const length = 2 + (int) (id % 700);
	unsigned long long const position = 1000ull + (unsigned long long) id * 7919ull + (1ull << 33);      /* beyond 32 bits */
	std::vector<char> buf((size_t) length + 100, '?'), other(16, 'x');
	DeviceWindows::Placeholder(buf.data(), position, length);
	bool ok = check(buf.data(), position, length, other.data());
	for (int p = 0; p < 3 && ok; ++p) {
		env->waker.parkOnce();      /* other reads note their windows on this carrier meanwhile */
		ok = check(buf.data(), position, length, other.data());
	}
	if (buf[(size_t) length] != '?') ok = false;      /* nothing written behind the window's NUL */
	if (!ok) env->errors += 1;
	env->ran += 1;
}
void destroySlot(void *, void *) {}
void lastTaken(void *) {}
}

int main(int argc, char ** argv) {
	int const carriers = argc > 1 ? atoi(argv[1]) : 4;
	int const fibers = argc > 2 ? atoi(argv[2]) : 256;
	int const items = argc > 3 ? atoi(argv[3]) : 20000;
	if (!DeviceWindows::Enabled()) { printf("DeviceWindows off (CVX_DEVICE_DECODE=0)\n"); return 1; }
	if (DeviceWindows::HaveGenome()) { printf("a genome before SetGenome\n"); return 1; }
	Env env;
	std::thread wk([&env] { env.waker.loop(); });
	{
		FiberPool::Callbacks cb;
		cb.user = &env; cb.run = &runItem; cb.destroySlot = &destroySlot; cb.carrierStart = 0; cb.lastItemTaken = &lastTaken;
		FiberPool pool(carriers, fibers, 256 * 1024, fibers, cb);
		for (int i = 0; i < items; ++i) pool.Submit((void *) (size_t) (i + 1));
		pool.CloseFeed();
		pool.DrainAndStop();
	}
	{ std::lock_guard<std::mutex> lk(env.waker.mtx); env.waker.stop = true; }
	env.waker.cv.notify_all();
	wk.join();
	/* plain worker threads: the note is thread-local */
	std::atomic<long> threadErrors{0};
	std::vector<std::thread> ths;
	for (int t = 0; t < 8; ++t) ths.emplace_back([&threadErrors, t] {
		std::vector<char> other(16, 'x');
		for (int k = 0; k < 2000; ++k) {
			int const length = 2 + (t * 131 + k) % 500;
			std::vector<char> buf((size_t) length + 100);
			unsigned long long const position = 5000ull + (unsigned long long) t * 1000003ull + (unsigned long long) k;
			DeviceWindows::Placeholder(buf.data(), position, length);
			std::this_thread::yield();
			if (!check(buf.data(), position, length, other.data())) threadErrors += 1;
		}
	});
	for (std::thread & t : ths) t.join();
	unsigned long long const starts[2] = { 1000ull, 5000ull };
	unsigned char const bin[4] = { 0x44, 0x44, 0x44, 0x44 };
	DeviceWindows::SetGenome(bin, 8, starts, 2);
	bool const have = DeviceWindows::HaveGenome();
	if (env.errors.load() || threadErrors.load() || env.ran.load() != items || !have) {
		printf("FAILED: %ld fiber items wrong, %ld thread items wrong, %ld of %d ran, genome %d\n", env.errors.load(), threadErrors.load(), env.ran.load(), items, (int) have);
		return 1;
	}
	printf("ok: %d reads on %d fibers over %d carriers and 16000 on 8 threads found their own window note\n", items, fibers, carriers);
	return 0;
}
