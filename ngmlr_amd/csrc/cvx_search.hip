/*
 * cvx_search.hip -- ngmlr's candidate search on gfx950 (search half): the k-mer vote of a batch of
 * (sub-)reads over the reference's k-mer table resident in HBM.
 *
 * Behavioural contract, reference file:line (all under src/):
 *   CS::RunRead              CS.cpp:324-398        per read: fresh vote table, PrefixIteration, CollectResultsStd; on
 *                                                  overflow of the probe budget the search is repeated with a larger table
 *   CS::PrefixIteration      CSstatic.cpp:23-73    every 13-mer, 2 bits per base as (c >> 1) & 3; windows holding 'N' are skipped
 *   CS::PrefixSearch         CS.cpp:57-99          forward row of the k-mer, reverse row of its reverse complement
 *   CompactPrefixTable::GetRefEntry  PrefixTable.cpp:476-532 (revComp :45-59)
 *   CS::AddLocationStd       CS.cpp:101-149        vote table keyed by bin, growing threshold, rList in order of qualification
 *   CS::CollectResultsStd    CS.cpp:219-268        LocationScore list: rList order, forward score before reverse score
 *
 * The list a read gets depends on the ORDER of its votes (a bin enters rList when one of its scores reaches the
 * threshold of that moment, and the threshold grows with the votes), and what follows the search sorts candidates with
 * an unstable sort: a drop-in has to produce the list in the reference's order.  The vote is therefore kept serial per
 * read -- one LANE owns a read, its vote table (same multiplicative hash, same linear probing, same probe budget, so that
 * the overflow / retry behaviour is the reference's too) and its rList -- and the device's width goes into the batch:
 * hundreds of thousands of reads in flight hide the dependent loads of each.  Integer and float32 arithmetic as in the
 * reference (votes are +1.0f, the threshold is maxHitNumber * sensitivity in float32).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvx_launch.h"

namespace cvx {

static const uint64_t kEmptyKey = ~0ull;
static const uint64_t kListed = 1ull << 63;          /* the reference's state bit 0x80000000: the bin is in rList */

__device__ __forceinline__ uint64_t rev_comp13(uint64_t prefix, const int k) {
	/* PrefixTable.cpp:45-59: complement (A<->T is ^2 on the (c >> 1) & 3 code: A 0, C 1, T 2, G 3), then reverse the 2-bit groups */
	const int bits = 2 * k;
	uint64_t c = (prefix ^ 0xAAAAAAAAull) & ((1ull << bits) - 1ull);
	uint64_t r = 0;
	for (int i = 0; i < k; ++i) { r = (r << 2) | (c & 3ull); c >>= 2; }
	return r;
}

/* events (votes) of every read: the size of its rList / candidate regions */
__global__ void __launch_bounds__(64)
search_count_kernel(const SearchArgs a) {
	const int i = blockIdx.x * 64 + threadIdx.x;
	if (i >= a.n) return;
	const uint8_t *seq = a.seq + a.seq_off[i];
	long long length = a.seq_len[i];
	const int K = a.k;
	const uint64_t mask = (1ull << (2 * K)) - 1ull;
	unsigned long long events = 0;
	for (;;) {
		if (length < K) break;
		if (*seq == 'N') {
			long long n_skip = 1;
			while (seq[n_skip] == 'N') ++n_skip;
			seq += n_skip;
			if (n_skip >= length - K) break;
			length -= n_skip;
		}
		uint64_t prefix = 0;
		bool restart = false;
		for (long long q = 0; q < K - 1; ++q) {
			const int ch = seq[q];
			if (ch == 'N') { seq += q + 1; length -= q + 1; restart = true; break; }
			prefix = (prefix << 2) | (uint64_t) ((ch >> 1) & 3);
		}
		if (restart) continue;
		for (long long q = K - 1; q < length; ++q) {
			const int ch = seq[q];
			if (ch == 'N') { seq += q + 1; length -= q + 1; restart = true; break; }
			prefix = ((prefix << 2) | (uint64_t) ((ch >> 1) & 3)) & mask;
			const uint64_t rc = rev_comp13(prefix, K);
			if (a.used[prefix]) events += a.tab[prefix + 1] - a.tab[prefix];
			if (a.used[rc]) events += a.tab[rc + 1] - a.tab[rc];
		}
		if (!restart) break;
	}
	a.events[i] = events;
}

/* one lane = one CS::RunRead attempt with a table of 2^bits entries */
__global__ void __launch_bounds__(64)
search_kernel(const SearchArgs a) {
	const int q = blockIdx.x * 64 + threadIdx.x;
	if (q >= a.n_work) return;
	const int i = a.work ? a.work[q] : q;                  /* read index (retries run on a list) */
	const int bits = a.bits;
	const uint32_t size = 1u << bits;
	uint64_t *keys = a.keys + (size_t) q * size;
	float2 *fr = reinterpret_cast<float2 *>(a.scores) + (size_t) q * size;
	uint32_t *rlist = a.rlist + a.list_off[i];
	const uint8_t *seq = a.seq + a.seq_off[i];
	long long length = a.seq_len[i];
	const int read_len = a.seq_len[i];
	const int K = a.k;
	const uint64_t mask = (1ull << (2 * K)) - 1ull;
	long long hpoc = (long long) ((float) size * a.hpoc_factor);       /* CS.cpp:352 / :379 */
	float max_hit = 0.0f, thresh = 0.0f;
	int rlen = 0;
	bool overflow = false;
	unsigned long long offset = 0;
	int misses = 0;        /* k-mers neither whose row nor whose reverse complement's row is in the table: kCount, CS.cpp:67-69 */

	auto add = [&](const uint64_t bin, const bool reverse) {           /* CS.cpp:101-149 */
		uint32_t e = (uint32_t) ((bin * 11400714819323199488ull) >> (64 - bits));
		uint64_t key = keys[e];
		while (key != kEmptyKey && (key & ~kListed) != bin) {
			if (++e >= size) e = 0;
			if (--hpoc == 0) { overflow = true; return; }
			key = keys[e];
		}
		float score = 1.0f;
		if (key == kEmptyKey) {
			key = bin;
			fr[e] = reverse ? make_float2(0.0f, 1.0f) : make_float2(1.0f, 0.0f);
		} else {
			float2 v = fr[e];
			if (reverse) { v.y += 1.0f; score = v.y; } else { v.x += 1.0f; score = v.x; }
			fr[e] = v;
		}
		if (score > max_hit) { max_hit = score; thresh = max_hit * a.sensitivity; }
		if (!(key & kListed) && score >= thresh) {
			key |= kListed;
			rlist[rlen++] = e;
		}
		keys[e] = key;
	};

	for (; !overflow;) {                                                /* CSstatic.cpp:23-73 */
		if (length < K) break;
		if (*seq == 'N') {
			long long n_skip = 1;
			while (seq[n_skip] == 'N') ++n_skip;
			seq += n_skip;
			if (n_skip >= length - K) break;
			length -= n_skip;
			offset += (unsigned long long) n_skip;
		}
		uint64_t prefix = 0;
		bool restart = false;
		for (long long p = 0; p < K - 1; ++p) {
			const int ch = seq[p];
			if (ch == 'N') { seq += p + 1; length -= p + 1; offset += (unsigned long long) (p + 1); restart = true; break; }
			prefix = (prefix << 2) | (uint64_t) ((ch >> 1) & 3);
		}
		if (restart) continue;
		for (long long p = K - 1; p < length && !overflow; ++p) {
			const int ch = seq[p];
			if (ch == 'N') { seq += p + 1; length -= p + 1; offset += (unsigned long long) (p + 1); restart = true; break; }
			prefix = ((prefix << 2) | (uint64_t) ((ch >> 1) & 3)) & mask;
			const unsigned long long pos = offset + (unsigned long long) p + 1ull - (unsigned long long) K;
			/* CS.cpp:57-99 over GetRefEntry (PrefixTable.cpp:476-532): forward row, then the reverse complement's row */
			const uint64_t rcp = rev_comp13(prefix, K);
			const bool use_f = a.used[prefix] != 0, use_r = a.used[rcp] != 0;
			if (!use_f && !use_r) misses += 1;       /* entries[0].refTotal == 0 (PrefixTable.cpp:489-525), counted before the votes */
			for (int rev = 0; rev < 2 && !overflow; ++rev) {
				const uint64_t pr = rev ? rcp : prefix;
				if (!(rev ? use_r : use_f)) continue;
				const uint32_t start = a.tab[pr] - 1u, nloc = a.tab[pr + 1] - 1u - start;
				const unsigned long long corr = rev ? (unsigned long long) read_len - (pos + (unsigned long long) K) : pos;
				for (uint32_t j = 0; j < nloc && !overflow; ++j) {
					const unsigned long long loc = (unsigned long long) a.locs[start + j] + a.unit_offset;
					add((loc - corr) >> a.bin_shift, rev != 0);
				}
			}
		}
		if (!restart) break;
	}

	/* kCount is reset per read, not per attempt (CS.cpp:338): the k-mers an overflowed attempt visited stay counted */
	if (a.kmer_misses) a.kmer_misses[i] += misses;
	if (overflow) { a.n_cand[i] = -1; return; }
	/* CollectResultsStd, CS.cpp:219-268 */
	const float thr = a.min_hits > thresh ? a.min_hits : thresh;
	const unsigned long long half = a.bin_shift > 0 ? 1ull << (a.bin_shift - 1) : 0ull;
	SearchCandidate *out = a.cand + 2ull * a.list_off[i];
	int n = 0;
	for (int r = 0; r < rlen; ++r) {
		const uint32_t e = rlist[r];
		const unsigned long long bin = keys[e] & ~kListed;
		const float2 v = fr[e];
		if (v.x >= thr) { SearchCandidate c; c.location = (bin << a.bin_shift) + half; c.score = v.x; c.reverse = 0; out[n++] = c; }
		if (v.y >= thr) { SearchCandidate c; c.location = (bin << a.bin_shift) + half; c.score = v.y; c.reverse = 1; out[n++] = c; }
	}
	a.n_cand[i] = n;
	a.max_hit[i] = max_hit;
}

/* dense[dst_begin[i] ...) = the n_cand[i] candidates of read i (its region of the sparse arena starts at 2 * list_off[i]) */
__global__ void __launch_bounds__(64)
search_compact_kernel(const SearchCandidate *sparse, const uint64_t *list_off, const int32_t *n_cand, const uint64_t *dst_begin,
		SearchCandidate *dense, int n) {
	const int i = blockIdx.x;
	if (i >= n) return;
	const int m = n_cand[i];
	const SearchCandidate *src = sparse + 2ull * list_off[i];
	SearchCandidate *dst = dense + dst_begin[i];
	for (int q = threadIdx.x; q < m; q += 64) dst[q] = src[q];
}

hipError_t launch_search_compact(const SearchCandidate *sparse, const uint64_t *list_off, const int32_t *n_cand, const uint64_t *dst_begin,
		SearchCandidate *dense, int n, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(search_compact_kernel, dim3(n), dim3(64), 0, st, sparse, list_off, n_cand, dst_begin, dense, n);
	return hipGetLastError();
}

hipError_t launch_search_count(const SearchArgs &a, hipStream_t st) {
	if (a.n <= 0) return hipSuccess;
	hipLaunchKernelGGL(search_count_kernel, dim3((a.n + 63) / 64), dim3(64), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_search(const SearchArgs &a, hipStream_t st) {
	if (a.n_work <= 0) return hipSuccess;
	hipLaunchKernelGGL(search_kernel, dim3((a.n_work + 63) / 64), dim3(64), 0, st, a);
	return hipGetLastError();
}

}  // namespace cvx
