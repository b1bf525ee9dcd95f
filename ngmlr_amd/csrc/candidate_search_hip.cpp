/* candidate_search_hip.cpp -- see candidate_search_hip.h.  Host-only C++ over the C ABI. */
#include "candidate_search_hip.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace Convex {

namespace {
/* handles per device: a call is synchronous (reads in, lists out) and a chain of short kernels with host round trips in
 * between, so one handle behind one mutex would serialise the CS threads; they are dealt round-robin over kLanes handles
 * (a handle's persistent vote tables are ~0.4 GB for a batch of 400 sub-reads) */
const int kLanes = 16;
std::mutex g_mtx;                       /* creation / shutdown */
std::mutex g_laneMtx[kLanes];
cvx_handle g_handle[kLanes] = {0};
CandidateSearchHip * g_instance = 0;
std::atomic<int> g_nextLane(0);
std::atomic<long> g_calls(0), g_reads(0), g_lists(0);
std::atomic<long long> g_ns(0);
thread_local int tl_lane = -1;
}

CandidateSearchHip * CandidateSearchHip::Get(int kmerLength, void const * refTableIndex, uint32_t const * refTable, uint32_t nLocations,
		uint64_t unitOffset, int deviceId) {
	std::lock_guard<std::mutex> g(g_mtx);
	if (g_instance != 0) return g_instance;
	cvx_params p = { 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f };      /* the search has no scoring; a handle needs a valid set */
	for (int l = 0; l < kLanes; ++l) {
		if (cvx_create_ex(deviceId, &p, 0, CVX_CREATE_SERVICE, &g_handle[l]) != CVX_OK) {
			/* no silent host path: a binary built with the device search fails loudly without its device */
			fprintf(stderr, "CandidateSearchHip: %s\n", cvx_last_error());
			for (int q = 0; q < l; ++q) { cvx_destroy(g_handle[q]); g_handle[q] = 0; }
			throw "CandidateSearchHip: no usable MI355X";
		}
	}
	CandidateSearchHip * s = new CandidateSearchHip();
	s->device = deviceId;
	if (cvx_index_upload(g_handle[0], kmerLength, refTableIndex, refTable, nLocations, unitOffset, &s->index) != CVX_OK) {
		fprintf(stderr, "CandidateSearchHip: %s\n", cvx_last_error());
		for (int q = 0; q < kLanes; ++q) { cvx_destroy(g_handle[q]); g_handle[q] = 0; }
		delete s;
		throw "CandidateSearchHip: the k-mer table could not be put on the device";
	}
	g_instance = s;
	return s;
}

void CandidateSearchHip::Shutdown() {
	std::lock_guard<std::mutex> g(g_mtx);
	if (g_instance == 0) return;
	for (int l = 0; l < kLanes; ++l) g_laneMtx[l].lock();
	std::chrono::steady_clock::time_point const s0 = std::chrono::steady_clock::now();
	/* (leaving the 32 scoring and search handles to the end of the process instead saves nothing: measured, the runtime's own
	 * teardown at exit then takes the 0.5 s these calls take, profiles/r04_timeline_e2e.txt) */
	cvx_index_free(g_handle[0], g_instance->index);
	for (int l = 0; l < kLanes; ++l) { cvx_destroy(g_handle[l]); g_handle[l] = 0; }
	for (int l = 0; l < kLanes; ++l) g_laneMtx[l].unlock();
	fprintf(stderr, "CandidateSearchHip: %ld search calls, %ld reads (%.0f per call), %ld candidates, %.2f s inside the calls summed over the threads "
			"(%.3f ms per call, %.1f us per read)\n", g_calls.load(), g_reads.load(), g_calls.load() ? (double) g_reads.load() / (double) g_calls.load() : 0.0,
			g_lists.load(), (double) g_ns.load() * 1e-9, g_calls.load() ? (double) g_ns.load() * 1e-6 / (double) g_calls.load() : 0.0,
			g_reads.load() ? (double) g_ns.load() * 1e-3 / (double) g_reads.load() : 0.0);
	if (getenv("CVX_TIMELINE")) fprintf(stderr, "cvx timeline: CandidateSearchHip freed its index and handles in %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count());
	delete g_instance;
	g_instance = 0;
}

void CandidateSearchHip::Search(Batch & b, float sensitivity, float minKmerHits, int binShift, int firstTableBits) {
	size_t const n = b.seqs.size();
	b.nCand.assign(n, 0); b.begin.assign(n, 0); b.maxHit.assign(n, 0.0f); b.kmerMisses.assign(n, 0); b.attempts.assign(n, 1);
	if (n == 0) return;
	if (tl_lane < 0) tl_lane = g_nextLane.fetch_add(1) % kLanes;
	std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
	std::lock_guard<std::mutex> lane(g_laneMtx[tl_lane]);
	if (b.cands.size() < 4096) b.cands.resize(4096);
	uint64_t used = 0;
	for (int attempt = 0; ; ++attempt) {
		int const rc = cvx_search_batch_ex(g_handle[tl_lane], index, (int32_t) n, b.seqs.data(), b.lens.data(), sensitivity, minKmerHits, binShift,
				firstTableBits, b.nCand.data(), b.begin.data(), b.cands.data(), (uint64_t) b.cands.size(), &used, b.maxHit.data(), b.kmerMisses.data());
		if (rc == CVX_ERR_CAPACITY && attempt == 0 && used > b.cands.size()) {      /* more candidates than the buffer holds: once more with room */
			b.cands.resize((size_t) used + used / 4 + 64);
			continue;
		}
		if (rc != CVX_OK) {
			fprintf(stderr, "CandidateSearchHip: %s\n", cvx_last_error());
			throw 1;
		}
		(void) cvx_search_last_attempts(g_handle[tl_lane], (int32_t) n, b.attempts.data());
		break;
	}
	g_calls += 1; g_reads += (long) n; g_lists += (long) used;
	g_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace Convex
