/* candidate_search_hip.cpp -- see candidate_search_hip.h.  Host-only C++ over the C ABI. */
#include "candidate_search_hip.h"
#include "service_device.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace Convex {

namespace {
/* handles per device: a call is synchronous (reads in, lists out) and a chain of short kernels with host round trips in
 * between, so one handle behind one mutex would serialise the CS threads; a device's threads are dealt round-robin over its
 * kLanes handles (a handle's persistent vote tables are ~0.4 GB for a batch of 400 sub-reads) */
const int kLanes = 16;
struct PerDevice {
	std::mutex mtx;                     /* creation of this device's table copy and handles */
	std::mutex laneMtx[kLanes];
	cvx_handle handle[kLanes] = {0};
	cvx_index index = 0;
	bool ready = false;
	std::atomic<int> nextLane{0};
	std::atomic<long> calls{0}, reads{0}, lists{0};
	std::atomic<long long> ns{0};
};
std::mutex g_mtx;                       /* creation / shutdown of the instance */
PerDevice g_dev[kMaxLogicalDevices];
CandidateSearchHip * g_instance = 0;
thread_local int tl_lane = -1;
}

CandidateSearchHip * CandidateSearchHip::Get(int kmerLength, void const * refTableIndex, uint32_t const * refTable, uint32_t nLocations,
		uint64_t unitOffset) {
	std::lock_guard<std::mutex> g(g_mtx);
	if (g_instance != 0) return g_instance;
	int nl = 0, np = 0;
	DeviceLayout(nl, np);
	if (nl <= 0) {
		/* no silent host path: a binary built with the device search fails loudly without its device */
		fprintf(stderr, "CandidateSearchHip: no usable MI355X\n");
		throw "CandidateSearchHip: no usable MI355X";
	}
	CandidateSearchHip * s = new CandidateSearchHip();
	s->kmerLength = kmerLength; s->refTableIndex = refTableIndex; s->refTable = refTable; s->nLocations = nLocations; s->unitOffset = unitOffset;
	g_instance = s;
	return s;
}

/* this device's handles and its copy of the k-mer table, on the first search of one of its threads (throws like Get) */
static void prepare_device(int logical, int kmerLength, void const * refTableIndex, uint32_t const * refTable, uint32_t nLocations, uint64_t unitOffset) {
	PerDevice & d = g_dev[logical];
	std::lock_guard<std::mutex> g(d.mtx);
	if (d.ready) return;
	int const physical = PhysicalDeviceOf(logical);
	cvx_params p = { 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f };      /* the search has no scoring; a handle needs a valid set */
	for (int l = 0; l < kLanes; ++l) {
		if (cvx_create_ex(physical, &p, 0, CVX_CREATE_SERVICE, &d.handle[l]) != CVX_OK) {
			fprintf(stderr, "CandidateSearchHip: %s\n", cvx_last_error());
			for (int q = 0; q < l; ++q) { cvx_destroy(d.handle[q]); d.handle[q] = 0; }
			throw "CandidateSearchHip: no usable MI355X";
		}
	}
	if (cvx_index_upload(d.handle[0], kmerLength, refTableIndex, refTable, nLocations, unitOffset, &d.index) != CVX_OK) {
		fprintf(stderr, "CandidateSearchHip: %s\n", cvx_last_error());
		for (int q = 0; q < kLanes; ++q) { cvx_destroy(d.handle[q]); d.handle[q] = 0; }
		throw "CandidateSearchHip: the k-mer table could not be put on the device";
	}
	d.ready = true;
}

void CandidateSearchHip::Shutdown() {
	std::lock_guard<std::mutex> g(g_mtx);
	if (g_instance == 0) return;
	std::chrono::steady_clock::time_point const s0 = std::chrono::steady_clock::now();
	long calls = 0, reads = 0, lists = 0;
	long long ns = 0;
	int used = 0;
	for (int dv = 0; dv < kMaxLogicalDevices; ++dv) {
		PerDevice & d = g_dev[dv];
		std::lock_guard<std::mutex> gd(d.mtx);
		if (!d.ready) continue;
		used += 1;
		for (int l = 0; l < kLanes; ++l) d.laneMtx[l].lock();
		/* (leaving the scoring and search handles to the end of the process instead saves nothing: measured, the runtime's own
		 * teardown at exit then takes the 0.5 s these calls take, profiles/r04_timeline_e2e.txt) */
		cvx_index_free(d.handle[0], d.index);
		d.index = 0;
		for (int l = 0; l < kLanes; ++l) { cvx_destroy(d.handle[l]); d.handle[l] = 0; }
		for (int l = 0; l < kLanes; ++l) d.laneMtx[l].unlock();
		d.ready = false;
		calls += d.calls.load(); reads += d.reads.load(); lists += d.lists.load(); ns += d.ns.load();
	}
	fprintf(stderr, "CandidateSearchHip: %ld search calls, %ld reads (%.0f per call), %ld candidates, %.2f s inside the calls summed over the threads "
			"(%.3f ms per call, %.1f us per read)\n", calls, reads, calls ? (double) reads / (double) calls : 0.0,
			lists, (double) ns * 1e-9, calls ? (double) ns * 1e-6 / (double) calls : 0.0, reads ? (double) ns * 1e-3 / (double) reads : 0.0);
	if (used > 1) {
		/* one line per device: the k-mer table was resident on each of them, every CS thread searched on its own device */
		for (int dv = 0; dv < kMaxLogicalDevices; ++dv) {
			PerDevice & d = g_dev[dv];
			if (d.calls.load() == 0) continue;
			fprintf(stderr, "CandidateSearchHip: device %d (physical %d): %ld search calls, %ld reads, %ld candidates\n", dv, PhysicalDeviceOf(dv), d.calls.load(), d.reads.load(), d.lists.load());
			d.calls = 0; d.reads = 0; d.lists = 0; d.ns = 0;
		}
	} else {
		for (int dv = 0; dv < kMaxLogicalDevices; ++dv) { g_dev[dv].calls = 0; g_dev[dv].reads = 0; g_dev[dv].lists = 0; g_dev[dv].ns = 0; }
	}
	if (getenv("CVX_TIMELINE")) fprintf(stderr, "cvx timeline: CandidateSearchHip freed its index and handles in %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count());
	delete g_instance;
	g_instance = 0;
}

void CandidateSearchHip::Search(Batch & b, float sensitivity, float minKmerHits, int binShift, int firstTableBits) {
	size_t const n = b.seqs.size();
	b.nCand.assign(n, 0); b.begin.assign(n, 0); b.maxHit.assign(n, 0.0f); b.kmerMisses.assign(n, 0); b.attempts.assign(n, 1);
	if (n == 0) return;
	int const dv = ServiceDeviceOfThisThread();
	PerDevice & d = g_dev[dv];
	if (!d.ready) prepare_device(dv, kmerLength, refTableIndex, refTable, nLocations, unitOffset);
	if (tl_lane < 0) tl_lane = d.nextLane.fetch_add(1) % kLanes;
	std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
	std::lock_guard<std::mutex> lane(d.laneMtx[tl_lane]);
	if (b.cands.size() < 4096) b.cands.resize(4096);
	uint64_t used = 0;
	for (int attempt = 0; ; ++attempt) {
		int const rc = cvx_search_batch_ex(d.handle[tl_lane], d.index, (int32_t) n, b.seqs.data(), b.lens.data(), sensitivity, minKmerHits, binShift,
				firstTableBits, b.nCand.data(), b.begin.data(), b.cands.data(), (uint64_t) b.cands.size(), &used, b.maxHit.data(), b.kmerMisses.data());
		if (rc == CVX_ERR_CAPACITY && attempt == 0 && used > b.cands.size()) {      /* more candidates than the buffer holds: once more with room */
			b.cands.resize((size_t) used + used / 4 + 64);
			continue;
		}
		if (rc != CVX_OK) {
			fprintf(stderr, "CandidateSearchHip: %s\n", cvx_last_error());
			throw 1;
		}
		(void) cvx_search_last_attempts(d.handle[tl_lane], (int32_t) n, b.attempts.data());
		break;
	}
	d.calls += 1; d.reads += (long) n; d.lists += (long) used;
	d.ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace Convex
