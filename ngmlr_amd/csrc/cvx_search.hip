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

/*
 * search_wave_kernel -- one WAVE per read (round 4).  The lane-per-read kernel above keeps a read's vote table in HBM and
 * walks the read's votes as one chain of dependent global loads and stores (~1 us per link on a lane that has the memory
 * system almost to itself): a batch of a few hundred sub-reads -- what one of ngmlr's CS threads hands over per call -- is
 * seven waves chasing pointers.  Here a read owns a wave and its vote table lives in LDS:
 *
 *   - the reference's table of 2^bits entries is needed only for its COLLISION BEHAVIOUR (which probe step opens which
 *     entry, when the probe budget runs out): the table stays virtual, and the few hundred slots a 256-bp sub-read really
 *     occupies are kept in an LDS map  virtual slot -> (bin, forward score, reverse score, listed);  "is virtual slot e
 *     occupied, and by which bin" is one lookup in that map, so the probe sequence, the budget and with them the overflow
 *     / retry behaviour are the reference's, step by step (CS.cpp:101-149);
 *   - the k-mer walk stays serial (CSstatic.cpp:23-73 with its N rules) but only collects (position, prefix) pairs, 64 at
 *     a time; the table rows of those 64 k-mers and of their reverse complements are then looked up by 64 lanes at once
 *     (one memory round trip instead of 4 x 64 dependent ones), and the locations of a row are fetched 64 per load;
 *   - the votes are cast in the reference's order by the whole wave in lock step (every lane computes the same vote: the
 *     control flow is uniform, the LDS accesses are broadcasts), a vote costs a few LDS round trips instead of HBM ones.
 *
 * A read with more bins than the LDS map holds (kSearchWaveSlots / 2; a read whose k-mers are all over the genome) is
 * flagged kSearchNeedsHbm and redone by search_kernel.  Same outputs as search_kernel; candidates at cand + cand_off[i].
 */
namespace {
const int kSearchWaveSeq = 4096;        /* longest read (with its NUL) the wave kernel takes; longer ones go to search_kernel */
struct WaveTable {
	uint32_t slot[kSearchWaveSlots];      /* virtual slot of the entry | listed << 31; 0xFFFFFFFF: free */
	uint32_t bin_lo[kSearchWaveSlots], bin_hi[kSearchWaveSlots];
	float f[kSearchWaveSlots], r[kSearchWaveSlots];
	uint16_t rlist[kSearchWaveSlots / 2];
	uint32_t c_prefix_lo[64], c_prefix_hi[64];      /* the chunk: k-mers in walk order */
	uint32_t c_pos[64];
	uint32_t c_start[2][64], c_n[2][64];          /* their table rows, forward / reverse complement */
	uint8_t seq[kSearchWaveSeq + 64];             /* the read and its NUL: the serial walk reads LDS, not HBM */
};
}

__global__ void __launch_bounds__(64)
search_wave_kernel(const SearchArgs a) {
	__shared__ WaveTable T;
	const int q = blockIdx.x;
	if (q >= a.n_work) return;
	const int lane = threadIdx.x;
	const int i = a.work ? a.work[q] : q;
	const int bits = a.bits;
	const uint32_t size = 1u << bits;
	const uint8_t *gseq = a.seq + a.seq_off[i];
	long long length = a.seq_len[i];
	const int read_len = a.seq_len[i];
	const int K = a.k;
	const uint64_t mask = (1ull << (2 * K)) - 1ull;
	long long hpoc = (long long) ((float) size * a.hpoc_factor);
	float max_hit = 0.0f, thresh = 0.0f;
	int rlen = 0, entries = 0, misses = 0;
	bool overflow = false, too_many = false;
	unsigned long long offset = 0;
	if (read_len + 1 > kSearchWaveSeq) { if (lane == 0) a.n_cand[i] = kSearchNeedsHbm; return; }

	for (int s = lane; s < kSearchWaveSlots; s += 64) T.slot[s] = 0xFFFFFFFFu;
	for (int s = lane; s < read_len + 64; s += 64) T.seq[s] = s < read_len ? gseq[s] : (uint8_t) 0;      /* coalesced; NULs behind the read */
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	int sb = 0;       /* the walk's cursor into T.seq (the reference's moving `sequence` pointer) */

	/* every lane runs this with the same arguments (uniform control flow); lane 0 writes */
	auto vote = [&](const uint64_t bin, const bool reverse) {
		uint32_t e = (uint32_t) ((bin * 11400714819323199488ull) >> (64 - bits));
		const uint32_t blo = (uint32_t) bin, bhi = (uint32_t) (bin >> 32);
		int h;
		uint32_t sv;
		for (;;) {
			/* is virtual slot e taken?  (multiplicative hash of e into the LDS map, linear probing there) */
			h = (int) ((e * 2654435761u) >> (32 - 11));
			static_assert(kSearchWaveSlots == 2048, "the map's hash keeps 11 bits");
			for (;;) {
				sv = T.slot[h];
				if (sv == 0xFFFFFFFFu || (sv & 0x7FFFFFFFu) == e) break;
				h = (h + 1) & (kSearchWaveSlots - 1);
			}
			if (sv == 0xFFFFFFFFu) break;                                   /* free in the virtual table: a new entry goes here */
			if (T.bin_lo[h] == blo && T.bin_hi[h] == bhi) break;            /* this bin's entry */
			if (++e >= size) e = 0;                                         /* CS.cpp:107-114 */
			if (--hpoc == 0) { overflow = true; return; }
		}
		float score = 1.0f;
		bool listed = false;
		if (sv == 0xFFFFFFFFu) {
			if (entries >= kSearchWaveSlots / 2) { too_many = true; return; }
			entries += 1;
			if (lane == 0) { T.bin_lo[h] = blo; T.bin_hi[h] = bhi; T.f[h] = reverse ? 0.0f : 1.0f; T.r[h] = reverse ? 1.0f : 0.0f; }
		} else {
			listed = (sv >> 31) != 0u;
			if (reverse) { score = T.r[h] + 1.0f; if (lane == 0) T.r[h] = score; }
			else { score = T.f[h] + 1.0f; if (lane == 0) T.f[h] = score; }
		}
		if (score > max_hit) { max_hit = score; thresh = max_hit * a.sensitivity; }
		if (!listed && score >= thresh) {
			listed = true;
			if (lane == 0) T.rlist[rlen] = (uint16_t) h;
			rlen += 1;
		}
		if (lane == 0) T.slot[h] = e | (listed ? 0x80000000u : 0u);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	};

	/* the votes of the chunk's first `cn` k-mers, in order */
	auto cast_chunk = [&](const int cn) {
		/* table rows of all its k-mers, both orientations: one round trip for the wave */
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      /* lane 0's chunk entries -> every lane */
		if (lane < cn) {
			const uint64_t pr = ((uint64_t) T.c_prefix_hi[lane] << 32) | T.c_prefix_lo[lane];
			const uint64_t rc = rev_comp13(pr, K);
			const bool uf = a.used[pr] != 0, ur = a.used[rc] != 0;
			uint32_t sf = 0, nf = 0, sr = 0, nr = 0;
			if (uf) { sf = a.tab[pr] - 1u; nf = a.tab[pr + 1] - 1u - sf; }
			if (ur) { sr = a.tab[rc] - 1u; nr = a.tab[rc + 1] - 1u - sr; }
			T.c_start[0][lane] = sf; T.c_n[0][lane] = uf ? nf : 0xFFFFFFFFu;      /* 0xFFFFFFFF: row not in the table */
			T.c_start[1][lane] = sr; T.c_n[1][lane] = ur ? nr : 0xFFFFFFFFu;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
		for (int c = 0; c < cn && !overflow && !too_many; ++c) {
			const unsigned long long pos = T.c_pos[c];
			const uint32_t n0 = T.c_n[0][c], n1 = T.c_n[1][c];
			if (n0 == 0xFFFFFFFFu && n1 == 0xFFFFFFFFu) misses += 1;          /* kCount, counted before the k-mer's votes */
			for (int rev = 0; rev < 2 && !overflow && !too_many; ++rev) {
				const uint32_t nloc = rev ? n1 : n0;
				if (nloc == 0xFFFFFFFFu) continue;
				const uint32_t start = T.c_start[rev][c];
				const unsigned long long corr = rev ? (unsigned long long) read_len - (pos + (unsigned long long) K) : pos;
				for (uint32_t j0 = 0; j0 < nloc && !overflow && !too_many; j0 += 64) {
					const uint32_t m = nloc - j0 < 64u ? nloc - j0 : 64u;
					const uint32_t mine = (uint32_t) lane < m ? a.locs[start + j0 + (uint32_t) lane] : 0u;      /* 64 locations per load */
					for (uint32_t j = 0; j < m && !overflow && !too_many; ++j) {
						const unsigned long long loc = (unsigned long long) (uint32_t) __shfl((int) mine, (int) j, 64) + a.unit_offset;
						vote((loc - corr) >> a.bin_shift, rev != 0);
					}
				}
			}
		}
	};

	/* CSstatic.cpp:23-73: the walk, collecting k-mers 64 at a time */
	int cn = 0;
	auto push = [&](const uint64_t prefix, const unsigned long long pos) {
		if (lane == 0) { T.c_prefix_lo[cn] = (uint32_t) prefix; T.c_prefix_hi[cn] = (uint32_t) (prefix >> 32); T.c_pos[cn] = (uint32_t) pos; }
		cn += 1;
		if (cn == 64) { cast_chunk(64); cn = 0; }
	};
	for (; !overflow && !too_many;) {
		if (length < K) break;
		if (T.seq[sb] == 'N') {
			int n_skip = 1;
			while (T.seq[sb + n_skip] == 'N') ++n_skip;
			sb += n_skip;
			if (n_skip >= length - K) break;
			length -= n_skip;
			offset += (unsigned long long) n_skip;
		}
		uint64_t prefix = 0;
		bool restart = false;
		for (int p = 0; p < K - 1; ++p) {
			const int ch = T.seq[sb + p];
			if (ch == 'N') { sb += p + 1; length -= p + 1; offset += (unsigned long long) (p + 1); restart = true; break; }
			prefix = (prefix << 2) | (uint64_t) ((ch >> 1) & 3);
		}
		if (restart) continue;
		for (int p = K - 1; p < length && !overflow && !too_many; ++p) {
			const int ch = T.seq[sb + p];
			if (ch == 'N') { sb += p + 1; length -= p + 1; offset += (unsigned long long) (p + 1); restart = true; break; }
			prefix = ((prefix << 2) | (uint64_t) ((ch >> 1) & 3)) & mask;
			push(prefix, offset + (unsigned long long) p + 1ull - (unsigned long long) K);
		}
		if (!restart) break;
	}
	if (cn > 0 && !overflow && !too_many) cast_chunk(cn);

	if (too_many) { if (lane == 0) a.n_cand[i] = kSearchNeedsHbm; return; }      /* redone by search_kernel: nothing of this attempt counts */
	if (lane == 0 && a.kmer_misses) a.kmer_misses[i] += misses;
	if (overflow) { if (lane == 0) a.n_cand[i] = -1; return; }
	/* CollectResultsStd, CS.cpp:219-268: rList order, forward before reverse; 64 list entries per pass */
	const float thr = a.min_hits > thresh ? a.min_hits : thresh;
	const unsigned long long half = a.bin_shift > 0 ? 1ull << (a.bin_shift - 1) : 0ull;
	SearchCandidate *out = a.cand + a.cand_off[i];
	int n = 0;
	for (int r0 = 0; r0 < rlen; r0 += 64) {
		const int r = r0 + lane;
		int cnt = 0;
		float vf = 0.0f, vr = 0.0f;
		unsigned long long bin = 0;
		if (r < rlen) {
			const int h = T.rlist[r];
			vf = T.f[h]; vr = T.r[h];
			bin = ((unsigned long long) T.bin_hi[h] << 32) | T.bin_lo[h];
			cnt = (vf >= thr ? 1 : 0) + (vr >= thr ? 1 : 0);
		}
		int incl = cnt;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int u = __shfl_up(incl, d, 64);
			if (lane >= d) incl += u;
		}
		int at = n + incl - cnt;
		if (vf >= thr && r < rlen) { SearchCandidate c; c.location = (bin << a.bin_shift) + half; c.score = vf; c.reverse = 0; out[at++] = c; }
		if (vr >= thr && r < rlen) { SearchCandidate c; c.location = (bin << a.bin_shift) + half; c.score = vr; c.reverse = 1; out[at++] = c; }
		n += __shfl(incl, 63, 64);
	}
	if (lane == 0) { a.n_cand[i] = n; a.max_hit[i] = max_hit; }
}

/* dense[dst_begin[i] ...) = the n_cand[i] candidates of read i, which lie at sparse + src_off[i] */
__global__ void __launch_bounds__(64)
search_compact_kernel(const SearchCandidate *sparse, const uint64_t *src_off, const int32_t *n_cand, const uint64_t *dst_begin,
		SearchCandidate *dense, int n) {
	const int i = blockIdx.x;
	if (i >= n) return;
	const int m = n_cand[i];
	const SearchCandidate *src = sparse + src_off[i];
	SearchCandidate *dst = dense + dst_begin[i];
	for (int q = threadIdx.x; q < m; q += 64) dst[q] = src[q];
}

hipError_t launch_search_compact(const SearchCandidate *sparse, const uint64_t *src_off, const int32_t *n_cand, const uint64_t *dst_begin,
		SearchCandidate *dense, int n, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(search_compact_kernel, dim3(n), dim3(64), 0, st, sparse, src_off, n_cand, dst_begin, dense, n);
	return hipGetLastError();
}

hipError_t launch_search_wave(const SearchArgs &a, hipStream_t st) {
	if (a.n_work <= 0) return hipSuccess;
	hipLaunchKernelGGL(search_wave_kernel, dim3(a.n_work), dim3(64), 0, st, a);
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
