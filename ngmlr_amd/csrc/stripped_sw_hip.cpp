/* stripped_sw_hip.cpp -- see stripped_sw_hip.h */
#include "stripped_sw_hip.h"
#include "service_device.h"

#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <chrono>
#include <mutex>

namespace {
const int kMaxDevices = 64;
/* handles per device: a scoring call is synchronous (strings in, scores out) and mostly latency -- copy, a short kernel,
 * copy -- so one handle behind one mutex serialises 64 workers on ~0.3 ms round trips while the device idles.  Workers
 * are dealt round-robin over kLanes handles (own streams, own staging); calls on different lanes overlap. */
const int kLanes = 16;
std::mutex g_mtx[kMaxDevices][kLanes];
std::mutex g_tableMtx;
cvx_handle g_handle[kMaxDevices][kLanes] = {{0}};
int g_users[kMaxDevices] = {0};
long g_joined[kMaxDevices] = {0};
/* statistics, printed when the last scorer of a device goes (like SharedAligner's line) */
struct DevStats { std::atomic<long> calls{0}, pairs{0}, single{0}; std::atomic<long long> ns{0}, ctorNs{0}; };
DevStats g_st[kMaxDevices];      /* per logical device: the line of a device says what ran THERE */
std::chrono::steady_clock::time_point const g_loaded = std::chrono::steady_clock::now();      /* ~ process start */
double g_firstCtorBegin = -1.0, g_firstCtorEnd = -1.0;
double since_load() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - g_loaded).count(); }
}

StrippedSWHip::StrippedSWHip(int const deviceId) : device(deviceId >= 0 && deviceId < kMaxDevices ? deviceId : Convex::ServiceDeviceOfThisThread()), lane(0) {
	std::chrono::steady_clock::time_point const c0 = std::chrono::steady_clock::now();
	std::lock_guard<std::mutex> g(g_tableMtx);
	if (g_firstCtorBegin < 0.0) g_firstCtorBegin = since_load();
	lane = (int) (g_joined[device]++ % kLanes);
	if (g_handle[device][lane] == 0) {
		/* the scoring kernel has fixed weights; the handle only needs a valid scoring triple */
		cvx_params p = { 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f };
		if (cvx_create_ex(Convex::PhysicalDeviceOf(device), &p, 0, CVX_CREATE_SERVICE, &g_handle[device][lane]) != CVX_OK) {
			fprintf(stderr, "StrippedSWHip: %s\n", cvx_last_error());
			g_handle[device][lane] = 0;
			throw "StrippedSWHip: no usable MI355X";
		}
	}
	g_users[device] += 1;
	if (g_firstCtorEnd < 0.0) g_firstCtorEnd = since_load();
	g_st[device].ctorNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c0).count();
}

StrippedSWHip::~StrippedSWHip() {
	std::lock_guard<std::mutex> g(g_tableMtx);
	if (--g_users[device] == 0) {
		std::chrono::steady_clock::time_point const d0 = std::chrono::steady_clock::now();
		for (int l = 0; l < kLanes; ++l) {
			std::lock_guard<std::mutex> d(g_mtx[device][l]);
			if (g_handle[device][l]) cvx_destroy(g_handle[device][l]);
			g_handle[device][l] = 0;
		}
		g_joined[device] = 0;
		DevStats & st = g_st[device];
		int nl = 0, np = 0;
		Convex::DeviceLayout(nl, np);
		char where[64] = "";
		if (nl > 1) snprintf(where, sizeof(where), " on device %d (physical %d)", device, Convex::PhysicalDeviceOf(device));
		fprintf(stderr, "StrippedSWHip: %ld scoring calls%s (%ld of them single pairs), %ld pairs, %.2f s inside the calls summed over the workers "
				"(%.3f ms per call), %.2f s constructing, %.2f s destroying the handles\n", st.calls.load(), where, st.single.load(), st.pairs.load(),
				st.ns.load() * 1e-9, st.calls.load() ? st.ns.load() * 1e-6 / (double) st.calls.load() : 0.0, st.ctorNs.load() * 1e-9,
				std::chrono::duration<double>(std::chrono::steady_clock::now() - d0).count());
		st.calls = 0; st.pairs = 0; st.single = 0; st.ns = 0; st.ctorNs = 0;
		fprintf(stderr, "StrippedSWHip: library loaded at 0, first scorer constructed %.2f - %.2f s, last one gone at %.2f s\n", g_firstCtorBegin, g_firstCtorEnd, since_load());
	}
}

int StrippedSWHip::BatchScore(int const mode, int const batchSize, char const * const * const refSeqList,
		char const * const * const qrySeqList, float * const results, void * extData) {
	(void) mode; (void) extData;
	std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
	std::lock_guard<std::mutex> d(g_mtx[device][lane]);
	DevStats & st = g_st[device];
	st.calls += 1; st.pairs += batchSize; if (batchSize == 1) st.single += 1;
	struct Stop { std::chrono::steady_clock::time_point t; DevStats * s; ~Stop() { s->ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); } } stop{t0, &st};
	if (cvx_score_batch(g_handle[device][lane], batchSize, refSeqList, qrySeqList, results) != CVX_OK) {
		fprintf(stderr, "StrippedSWHip: %s\n", cvx_last_error());
		throw 1;
	}
	return batchSize;
}

int StrippedSWHip::SingleScore(int const mode, int const corridor, char const * const refSeq,
		char const * const qrySeq, float & result, void * extData) {
	(void) corridor;
	char const * r[1] = { refSeq };
	char const * q[1] = { qrySeq };
	float s = -1.0f;
	BatchScore(mode, 1, r, q, &s, extData);
	result = s;
	return s == -1.0f ? 0 : 1;     /* the reference returns 0 when a sequence is too long */
}
