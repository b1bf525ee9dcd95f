/*
 * cvx_pcsample.h -- a per-thread program-counter sampler for the ngmlr-side bindings (development aid, off unless
 * CVX_PC_SAMPLE=<file> is set).  perf and gdb are not part of every image this runs in; the pipeline's ceiling is the host's
 * CPU-seconds per read, and "which functions" is a question a histogram of sampled PCs answers: every
 * participating thread arms a CPU-time timer (CLOCK_THREAD_CPUTIME_ID, one SIGPROF per millisecond of its own CPU time,
 * delivered to that very thread), the handler stores the interrupted PC and a small thread-class tag, and at exit the samples and
 * /proc/self/maps go to the file; tools/pcsample_report.py turns them into per-function shares with nm.
 */
#pragma once
#include <atomic>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <ucontext.h>
#include <unistd.h>

namespace Convex {
namespace pcsample {

struct State {
	std::atomic<uint64_t> n;
	uint64_t cap;
	uint64_t *pcs;            /* pc | (class << 56) */
	const char *path;
};
inline State &state() { static State s = { {0}, 0, 0, 0 }; return s; }
inline int &thread_class() { static thread_local int c = 0; return c; }

inline void on_prof(int, siginfo_t *, void *ctx) {
	State &s = state();
	if (!s.pcs) return;
	const uint64_t at = s.n.fetch_add(1, std::memory_order_relaxed);
	if (at >= s.cap) return;
	const ucontext_t *uc = static_cast<const ucontext_t *>(ctx);
	s.pcs[at] = ((uint64_t) uc->uc_mcontext.gregs[REG_RIP] & 0x00FFFFFFFFFFFFFFull) | ((uint64_t) thread_class() << 56);
}

inline void dump() {
	State &s = state();
	if (!s.pcs || !s.path) return;
	FILE *f = fopen(s.path, "wb");
	if (!f) return;
	uint64_t n = s.n.load();
	if (n > s.cap) n = s.cap;
	fwrite(&n, 8, 1, f);
	fwrite(s.pcs, 8, (size_t) n, f);
	if (FILE *m = fopen("/proc/self/maps", "r")) {
		char line[512];
		while (fgets(line, sizeof(line), m)) fputs(line, f);
		fclose(m);
	}
	fclose(f);
}

/* call once from every thread that is to be sampled; cls: 1 carrier / context, 2 CS thread, 3 dispatcher, 4 other */
inline void arm_this_thread(int cls) {
	static std::atomic<int> init(0);
	const char *path = getenv("CVX_PC_SAMPLE");
	if (!path || !*path) return;
	State &s = state();
	int expected = 0;
	if (init.compare_exchange_strong(expected, 1)) {
		s.cap = 8u << 20;
		s.pcs = static_cast<uint64_t *>(calloc((size_t) s.cap, 8));
		s.path = strdup(path);
		struct sigaction sa;
		memset(&sa, 0, sizeof(sa));
		sa.sa_sigaction = &on_prof;
		sa.sa_flags = SA_SIGINFO | SA_RESTART;
		sigaction(SIGPROF, &sa, 0);
		atexit(&dump);
		init.store(2);
	}
	while (init.load() != 2) { }
	static thread_local bool armed = false;
	if (armed) return;
	armed = true;
	thread_class() = cls;
	struct sigevent sev;
	memset(&sev, 0, sizeof(sev));
	sev.sigev_notify = SIGEV_THREAD_ID;
	sev.sigev_signo = SIGPROF;
	sev._sigev_un._tid = (pid_t) syscall(SYS_gettid);
	timer_t t;
	if (timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &t) != 0) return;
	struct itimerspec its;
	its.it_interval.tv_sec = 0; its.it_interval.tv_nsec = 1000000;
	its.it_value = its.it_interval;
	timer_settime(t, 0, &its, 0);
}

}  // namespace pcsample
}  // namespace Convex
