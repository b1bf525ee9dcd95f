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
 * an unstable sort: a drop-in has to produce the list in the reference's order, with the reference's vote table (same
 * multiplicative hash, same linear probing, same probe budget, so that the overflow / retry behaviour is the reference's
 * too).  Integer and float32 arithmetic as in the reference (votes are +1.0f, the threshold is maxHitNumber * sensitivity
 * in float32).  Two implementations of that contract:
 *   search_wave_kernel / search_wave_hbm_kernel   the product path: a read owns a wave, which casts 64 consecutive votes at
 *       once with the sequential semantics reproduced by construction (further down);
 *   search_kernel                                  round 3's form and the independent check (CVX_TUNE_SEARCH_WAVE=0): a read
 *       owns a LANE, which casts its votes one by one over its own table in HBM -- the device's width goes into the batch.
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

/* CSstatic.cpp:23-73 in closed form: the windows [q, q + K), q < kmer_windows(), that hold no 'N' are the k-mers the reference
 * visits, in order of q.  (The reference restarts its scan behind every 'N', which changes nothing about the windows that follow,
 * with one exception: at the top of its loop -- at the start of the read, or after a restart that lands on another 'N' -- it
 * skips the run of n_skip 'N's and STOPS when n_skip >= length - K, where a plain scan would stop at n_skip > length - K: the
 * read's last window is lost when exactly K bases follow a run of 'N's that is two or more long or starts the read; a lone 'N'
 * inside the read restarts without that test.)  tests/test_search_cpu.py::test_closed_form_of_the_kmer_walk pins this form
 * against the checker's serial walk; search_kernel below keeps the serial walk. */
template <class CHAR_AT>
__device__ __forceinline__ int kmer_windows(const int read_len, const int K, CHAR_AT char_at) {
	const int n_all = read_len - K + 1;
	if (n_all >= 2) {
		const int q = n_all - 1;
		if (char_at(q - 1) == 'N' && (q == 1 || char_at(q - 2) == 'N')) return n_all - 1;
	}
	return n_all;
}
/* the k-mer of window q (2 bits per base as (c >> 1) & 3); false when the window holds an 'N' */
template <class CHAR_AT>
__device__ __forceinline__ bool kmer_at(const int q, const int K, CHAR_AT char_at, uint64_t &prefix) {
	bool ok = true;
	prefix = 0;
	for (int j = 0; j < K; ++j) {
		const int ch = char_at(q + j);
		ok = ok && ch != 'N';
		prefix = (prefix << 2) | (uint64_t) ((ch >> 1) & 3);
	}
	return ok;
}

/* events (votes) of every read: which LDS map it gets, the size of its rList / candidate regions.  One wave per read, a lane per
 * window (round 6: a lane per read walked its 256 bases as a chain of dependent loads, 1.15 ms per 100 000 sub-reads). */
__global__ void __launch_bounds__(256)
search_count_kernel(const SearchArgs a) {
	const int i = blockIdx.x * 4 + (int) (threadIdx.x >> 6);
	const int lane = (int) (threadIdx.x & 63u);
	if (i >= a.n) return;
	const uint8_t *seq = a.seq + a.seq_off[i];
	const int read_len = a.seq_len[i];
	const int K = a.k;
	auto char_at = [&](const int p) -> int { return p <= read_len ? seq[p] : 0; };      /* (the NUL behind the read is there) */
	const int n_win = kmer_windows(read_len, K, char_at);
	unsigned long long events = 0;
	for (int q0 = 0; q0 < n_win; q0 += 64) {
		const int q = q0 + lane;
		uint64_t prefix = 0;
		if (q < n_win && kmer_at(q, K, char_at, prefix)) {
			const uint2 rf = a.rows[prefix], rr = a.rows[rev_comp13(prefix, K)];
			if (rf.y >> 31) events += rf.y & 0x7FFFFFFFu;
			if (rr.y >> 31) events += rr.y & 0x7FFFFFFFu;
		}
	}
	for (int d = 32; d >= 1; d >>= 1) {
		const uint32_t lo = (uint32_t) __shfl_down((int) (uint32_t) events, d, 64), hi = (uint32_t) __shfl_down((int) (uint32_t) (events >> 32), d, 64);
		events += ((unsigned long long) hi << 32) | lo;
	}
	if (lane == 0) a.events[i] = events;
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
			const uint2 row_f = a.rows[prefix], row_r = a.rows[rcp];
			const bool use_f = (row_f.y >> 31) != 0u, use_r = (row_r.y >> 31) != 0u;
			if (!use_f && !use_r) misses += 1;       /* entries[0].refTotal == 0 (PrefixTable.cpp:489-525), counted before the votes */
			for (int rev = 0; rev < 2 && !overflow; ++rev) {
				const uint64_t pr = rev ? rcp : prefix;
				if (!(rev ? use_r : use_f)) continue;
				const uint32_t start = rev ? row_r.x : row_f.x, nloc = (rev ? row_r.y : row_f.y) & 0x7FFFFFFFu;
				(void) pr;
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
 * search_vote_read (search_wave_kernel, search_wave_hbm_kernel) -- one WAVE per read, the votes cast 64 at a time (round 4).
 *
 * The lane-per-read kernel above walks a read's votes as one chain of dependent loads and stores (~1 us per link): fine for
 * a sub-read with a few hundred votes when there are thousands of them in flight, hopeless for the sub-reads a real genome
 * produces beside those -- one inside a repeat family or a microsatellite casts 10^4..10^5 votes (tools/e2e_rates.py
 * --synthetic-rep: 100-900 ms for a call of 400 sub-reads, profiles/r04_e2e_rep.txt).  Here a read owns a wave, and a BATCH
 * of 64 consecutive votes -- in the reference's order, across table rows and k-mers -- is cast by the 64 lanes at once with
 * the sequential semantics of CS.cpp:101-149 reproduced by construction:
 *
 *   - probing: every lane probes the table as it stood before the batch.  A vote's probe path consists of entries that were
 *     occupied before the batch and ends at its own bin's entry or at the first free slot, so an earlier vote of the same
 *     batch can change it in one way only: by opening a new entry in exactly that free slot.  Same slot, same bin: the later
 *     vote joins the entry (a duplicate, below).  Same slot, different bin: a HAZARD -- the batch's new entries are taken
 *     back and the batch is cast vote by vote (cast_serial; rare: the table of 2^16+ slots holds a few hundred entries);
 *   - the probe budget (hpoc) is a sum over the votes; a batch that could exhaust it is cast serially as well, so the vote
 *     at which the attempt overflows -- and with it kCount, the k-mers visited until then -- is the reference's;
 *   - scores: votes of the batch for the same entry and orientation are ranked in lane order, vote j scores base + rank + 1
 *     (float32 additions of 1.0f: exact);
 *   - maxHitNumber before vote j is a prefix maximum over the lanes, the threshold of that moment maxHit * sensitivity in
 *     float32; an entry enters rList at the first vote (in lane order) whose score reaches its threshold, and the order of
 *     rList is the lane order of those votes (ballot + popcount).
 *
 * The k-mer walk stays serial (CSstatic.cpp:23-73 with its N rules) and collects (position, prefix) pairs 64 at a time; the
 * table rows of those k-mers and their reverse complements are looked up by 64 lanes at once, a prefix sum over the 128 row
 * lengths numbers the chunk's votes, and lane t of a batch finds its row by binary search and loads its location: one
 * memory round trip per 64 votes.
 *
 * Two tables behind the same code (template parameter):
 *   LdsVotes  the reference's table of 2^bits entries is needed only for its COLLISION BEHAVIOUR (which probe step opens
 *             which entry, when the budget runs out): it stays virtual, and the slots a sub-read really occupies live in an
 *             LDS map  virtual slot -> (bin, forward score, reverse score, listed) of 2^9 .. 2^12 slots, sized per read by
 *             the host from the read's vote count.  A read with more bins than its map holds (3/4 of the slots), a bin
 *             beyond 32 bits, a score beyond 16, or longer than kSearchWaveSeq, is flagged kSearchNeedsHbm and redone with
 *   HbmVotes  the real table in HBM (keys / scores of search_kernel's layout), any number of bins, any read length.
 * Same outputs as search_kernel: candidates at cand + cand_off[i] (LdsVotes) or cand + 2 * list_off[i] (HbmVotes).
 */
namespace {
const uint32_t kFreeSlot = 0xFFFFFFFFu;
enum { kProbeFree = 0, kProbeMatch = 1, kProbeOther = 2 };

struct ChunkRows {                      /* the chunk being cast: the k-mers at 64 consecutive window positions (LDS, both table forms) */
	uint32_t c_start[2][64];                       /* where their table rows start, forward / reverse complement */
	uint32_t row_first[129];                       /* votes before row 2 c + rev, in casting order; [128] = all of them */
	uint16_t miss_upto[64];                        /* k-mers 0..c with neither row in the table (kCount) */
	uint8_t row_at[64];                            /* a batch's rows by the vote they start at */
	uint8_t scr[1024];                             /* duplicate detection inside a batch */
};

/* Inclusive scans over the wave on the DPP path (round 6; six ds_bpermute round trips each until then): row_shr 1 / 2 / 4 / 8 inside a
 * row of 16 lanes -- a lane whose source lies outside its row takes the identity --, then lane 15 of a row into the next row
 * (row_bcast:15, rows 1 and 3) and lane 31 into rows 2 and 3 (row_bcast:31).  Every lane of the wave has to be active. */
/* A wave is its own workgroup here and its LDS instructions execute in order: what one lane wrote is there for the next LDS
 * instruction of any lane.  Only the compiler has to be kept from moving accesses across; a workgroup-scope fence would also
 * wait for every global load in flight (the next batch's locations). */
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
#define CVX_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v, const int lane) {
	(void) lane;
	v += (uint32_t) CVX_DPP((int) v, 0x111, 0xf); v += (uint32_t) CVX_DPP((int) v, 0x112, 0xf);
	v += (uint32_t) CVX_DPP((int) v, 0x114, 0xf); v += (uint32_t) CVX_DPP((int) v, 0x118, 0xf);
	v += (uint32_t) CVX_DPP((int) v, 0x142, 0xa); v += (uint32_t) CVX_DPP((int) v, 0x143, 0xc);
	return v;
}
/* (v >= 0: the identity is 0.0f) */
__device__ __forceinline__ float wave_incl_max(float v, const int lane) {
	(void) lane;
#define CVX_DPP_FMAX(ctrl, rows) { const float u = __int_as_float(CVX_DPP(__float_as_int(v), ctrl, rows)); v = u > v ? u : v; }
	CVX_DPP_FMAX(0x111, 0xf) CVX_DPP_FMAX(0x112, 0xf) CVX_DPP_FMAX(0x114, 0xf) CVX_DPP_FMAX(0x118, 0xf) CVX_DPP_FMAX(0x142, 0xa) CVX_DPP_FMAX(0x143, 0xc)
#undef CVX_DPP_FMAX
	return v;
}
__device__ __forceinline__ uint32_t wave_incl_umax(uint32_t v) {
#define CVX_DPP_UMAX(ctrl, rows) { const uint32_t u = (uint32_t) CVX_DPP((int) v, ctrl, rows); v = u > v ? u : v; }
	CVX_DPP_UMAX(0x111, 0xf) CVX_DPP_UMAX(0x112, 0xf) CVX_DPP_UMAX(0x114, 0xf) CVX_DPP_UMAX(0x118, 0xf) CVX_DPP_UMAX(0x142, 0xa) CVX_DPP_UMAX(0x143, 0xc)
#undef CVX_DPP_UMAX
	return v;
}
/* lane `src` (the same for every lane) of a register: v_readlane, no LDS round trip */
__device__ __forceinline__ int lane_of(const int v, const int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_of(const float v, const int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ uint64_t lane_of(const uint64_t v, const int src) {
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, src), hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), src);
	return ((uint64_t) hi << 32) | lo;
}

struct LdsVotes {
	static const bool kLds = true;
	/* The map of one read, carved out of the workgroup's dynamic LDS (round 6): `slots` = 2^log2s entries of 12 bytes --
	 *   slot[h]  virtual slot of the entry | listed << 31 (kFreeSlot: free)
	 *   bin[h]   the bin, 32 bits: a vote whose bin does not fit sends the read to the table in HBM (a location before the start of
	 *            the unit, i.e. a negative difference; no genome has 2^36 bases per unit)
	 *   fr[h]    forward score in the low half, reverse score in the high half, as integers: votes are +1.0f, so a score IS a count; a
	 *            count that would pass 65535 sends the read to HBM as well
	 * -- at most 3/4 full, behind them rList (16-bit map positions) and the read.  Round 5's map was 2 048 slots of 20 bytes at most
	 * half full, 47 KB whatever the read: three reads per CU, and a sub-read of a 512 Mbp genome (1 200-1 450 bins) never fitted.
	 * The host picks log2s per read from its vote count (search_common): 6 KB for a read with <= 384 votes, 24 KB for one with
	 * <= 1 536. */
	uint32_t *slot, *bin, *fr;
	uint16_t *rlist;
	uint8_t *seq;
	int log2s, cap;
	uint32_t smask;
	int entries;
	static __host__ __device__ size_t bytes(const int log2s, const int seq_cap) {
		const size_t S = (size_t) 1 << log2s;
		return S * 12 + (S * 3 / 4) * 2 + (size_t) seq_cap;
	}
	__device__ void carve(uint32_t *base, const int log2s_) {
		log2s = log2s_;
		const uint32_t S = 1u << log2s;
		smask = S - 1u; cap = (int) (S * 3u / 4u);
		slot = base; bin = base + S; fr = base + 2u * S;
		rlist = reinterpret_cast<uint16_t *>(base + 3u * S);
		seq = reinterpret_cast<uint8_t *>(rlist + cap);
		entries = 0;
	}
	__device__ void clear(const int lane) { for (uint32_t h = (uint32_t) lane; h <= smask; h += 64u) slot[h] = kFreeSlot; }
	__device__ void fence() const { lds_fence(); }
	__device__ int char_at(const int p) const { return seq[p]; }
	/* the map's own probe sequence of key e: double hashing (an odd step walks all 2^log2s slots) -- the lanes of a batch probe at
	 * once and the wave waits for the longest chain, which under linear probing at 60 % load is several times the mean */
	__device__ uint32_t first_slot(const uint32_t e) const { return (e * 2654435761u) >> (32 - log2s); }
	__device__ uint32_t slot_step(const uint32_t e) const { return ((e * 0x85EBCA6Bu) >> (32 - log2s)) | 1u; }
	__device__ static bool fits(const uint64_t b) { return (b >> 32) == 0ull; }
	/* what the virtual table holds at slot e, as far as a vote for `bin` cares */
	__device__ int find(const uint32_t e, const uint64_t b, int &idx, bool &listed) const {
		uint32_t h = first_slot(e);
		const uint32_t step = slot_step(e);
		uint32_t sv;
		for (;;) {      /* two slots of the sequence per round trip */
			const uint32_t h1 = (h + step) & smask;
			const uint32_t s0 = slot[h], s1 = slot[h1];
			if (s0 == kFreeSlot || (s0 & 0x7FFFFFFFu) == e) { sv = s0; break; }
			if (s1 == kFreeSlot || (s1 & 0x7FFFFFFFu) == e) { sv = s1; h = h1; break; }
			h = (h1 + step) & smask;
		}
		if (sv == kFreeSlot) { idx = (int) h; return kProbeFree; }      /* (where claim() starts) */
		if (bin[h] == (uint32_t) b) { idx = (int) h; listed = (sv >> 31) != 0u; return kProbeMatch; }
		return kProbeOther;
	}
	/* open (or join, when another lane of the batch just opened it) the entry of virtual slot e; idx: the free map slot find()
	 * ended at -- everything before it on e's probe sequence is taken by other keys and stays so */
	__device__ void claim(const uint32_t e, const uint64_t b, int &idx, bool &creator, bool &hazard) {
		uint32_t h = (uint32_t) idx;
		const uint32_t step = slot_step(e);
		for (;;) {
			const uint32_t old = atomicCAS(&slot[h], kFreeSlot, e);
			if (old == kFreeSlot) { creator = true; bin[h] = (uint32_t) b; fr[h] = 0u; break; }
			if ((old & 0x7FFFFFFFu) == e) break;
			h = (h + step) & smask;
		}
		idx = (int) h;
		hazard = false;         /* decided by verify() once the creators' bins are visible */
	}
	__device__ bool verify(const int idx, const uint64_t b) const { return bin[idx] == (uint32_t) b; }
	__device__ void unclaim(const int idx, const uint32_t e) { (void) e; slot[idx] = kFreeSlot; }
	__device__ float score(const int idx, const bool rev) const { return (float) reinterpret_cast<const uint16_t *>(fr)[2 * idx + (rev ? 1 : 0)]; }
	__device__ static bool score_fits(const float s) { return s <= 65535.0f; }
	__device__ void set_score(const int idx, const bool rev, const float s) { reinterpret_cast<uint16_t *>(fr)[2 * idx + (rev ? 1 : 0)] = (uint16_t) (uint32_t) s; }
	__device__ void set_listed(const int idx, const uint32_t e) { slot[idx] = e | 0x80000000u; }
	__device__ void list_put(const int pos, const int idx) { rlist[pos] = (uint16_t) idx; }
	__device__ int list_at(const int pos) const { return rlist[pos]; }
	__device__ uint64_t bin_of(const int idx) const { return (uint64_t) bin[idx]; }
	__device__ float2 scores_of(const int idx) const { const uint32_t v = fr[idx]; return make_float2((float) (v & 0xFFFFu), (float) (v >> 16)); }
	__device__ bool room_for(const int n_new) const { return entries + n_new <= cap; }
	__device__ bool list_room(const int n_listed) const { (void) n_listed; return true; }      /* (rList holds `cap` positions) */
	__device__ void remember(const int at, const uint32_t e) { (void) at; (void) e; }
	/* after `done` of `total` windows: at this rate the read ends with 5/4 of what the map holds or more -- give up now instead of
	 * at the entry that does not fit (a sub-read of a 2 Gbp genome casts 5 000 votes into as many bins: it was cast to 59 % in
	 * LDS before the map was full, then all over again over the table in HBM).  Only ever a question for the largest map: the
	 * smaller ones are chosen to hold the read's vote count.  A read given up wrongly is merely slower. */
	__device__ bool hopeless(const int done, const int total) const {
		return done * 8 >= total && (long long) entries * total * 4 > (long long) cap * done * 5;
	}
};

/* The same map in 8 bytes per slot, for the case that fills the device at the scale of a genome: a first attempt at the
 * reference's default table size (virtual slots of 16 bits) on a sub-read of at most 268 bases (a bin can hardly collect more
 * than 255 votes of one orientation).  { bin (31 bits) | listed << 31; 0xFFFFFFFF: free } { virtual slot | forward count << 16 |
 * reverse count << 24 }; rList holds a quarter of the slots.  Whatever does not fit -- a bin of 31 bits and more, a count beyond
 * 255, more listed bins than rList holds -- sends the read to the table in HBM, as with the 12-byte map.  2 048 slots are 16 KB
 * where the 12-byte form takes 24: eight reads on a CU instead of five. */
struct LdsVotes8 {
	static const bool kLds = true;
	uint2 *sl;
	uint16_t *rlist;
	uint8_t *seq;
	int log2s, cap, list_cap;
	uint32_t smask;
	int entries;
	static __host__ __device__ size_t bytes(const int log2s, const int seq_cap) {
		const size_t S = (size_t) 1 << log2s;
		return S * 8 + (S / 4) * 2 + (size_t) seq_cap;
	}
	__device__ void carve(uint32_t *base, const int log2s_) {
		log2s = log2s_;
		const uint32_t S = 1u << log2s;
		smask = S - 1u; cap = (int) (S * 3u / 4u); list_cap = (int) (S / 4u);
		sl = reinterpret_cast<uint2 *>(base);
		rlist = reinterpret_cast<uint16_t *>(base + 2u * S);
		seq = reinterpret_cast<uint8_t *>(rlist + list_cap);
		entries = 0;
	}
	__device__ void clear(const int lane) { for (uint32_t h = (uint32_t) lane; h <= smask; h += 64u) sl[h] = make_uint2(kFreeSlot, 0u); }
	__device__ void fence() const { lds_fence(); }
	__device__ int char_at(const int p) const { return seq[p]; }
	__device__ uint32_t first_slot(const uint32_t e) const { return (e * 2654435761u) >> (32 - log2s); }
	__device__ uint32_t slot_step(const uint32_t e) const { return ((e * 0x85EBCA6Bu) >> (32 - log2s)) | 1u; }
	__device__ static bool fits(const uint64_t b) { return b < 0x7FFFFFFFull; }
	__device__ int find(const uint32_t e, const uint64_t b, int &idx, bool &listed) const {
		uint32_t h = first_slot(e);
		const uint32_t step = slot_step(e);
		uint2 v;
		for (;;) {      /* two slots of the sequence per round trip */
			const uint32_t h1 = (h + step) & smask;
			const uint2 v0 = sl[h], v1 = sl[h1];
			if (v0.x == kFreeSlot || (v0.y & 0xFFFFu) == e) { v = v0; break; }
			if (v1.x == kFreeSlot || (v1.y & 0xFFFFu) == e) { v = v1; h = h1; break; }
			h = (h1 + step) & smask;
		}
		idx = (int) h;
		if (v.x == kFreeSlot) return kProbeFree;      /* (where claim() starts) */
		if ((v.x & 0x7FFFFFFFu) == (uint32_t) b) { listed = (v.x >> 31) != 0u; return kProbeMatch; }
		return kProbeOther;
	}
	/* The claim is ONE 64-bit compare-and-swap of the whole slot, (free, 0) -> (bin, virtual slot): what it returns tells a loser
	 * whose slot this has become.  (Claiming the bin word alone and naming the virtual slot in a second store leaves the losers
	 * to read that store -- and nothing keeps the compiler from laying the losers' path out in front of the winners'.) */
	__device__ void claim(const uint32_t e, const uint64_t b, int &idx, bool &creator, bool &hazard) {
		uint32_t h = (uint32_t) idx;
		const uint32_t step = slot_step(e);
		const unsigned long long free64 = (unsigned long long) kFreeSlot, mine = ((unsigned long long) e << 32) | (unsigned long long) (uint32_t) b;
		for (;;) {
			const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&sl[h]), free64, mine);
			if (old == free64) { creator = true; break; }
			if ((uint32_t) (old >> 32 & 0xFFFFull) == e) break;
			h = (h + step) & smask;
		}
		idx = (int) h;
		hazard = false;         /* decided by verify() */
	}
	__device__ bool verify(const int idx, const uint64_t b) const { return (sl[idx].x & 0x7FFFFFFFu) == (uint32_t) b; }
	__device__ void unclaim(const int idx, const uint32_t e) { (void) e; sl[idx] = make_uint2(kFreeSlot, 0u); }      /* (a free slot is (free, 0): the claim compares both words) */
	__device__ float score(const int idx, const bool rev) const { return (float) reinterpret_cast<const uint8_t *>(&sl[idx])[rev ? 7 : 6]; }
	__device__ static bool score_fits(const float s) { return s <= 255.0f; }
	__device__ void set_score(const int idx, const bool rev, const float s) { reinterpret_cast<uint8_t *>(&sl[idx])[rev ? 7 : 6] = (uint8_t) (uint32_t) s; }
	__device__ void set_listed(const int idx, const uint32_t e) { (void) e; sl[idx].x |= 0x80000000u; }      /* (one lane per entry lists it) */
	__device__ void list_put(const int pos, const int idx) { rlist[pos] = (uint16_t) idx; }
	__device__ int list_at(const int pos) const { return rlist[pos]; }
	__device__ uint64_t bin_of(const int idx) const { return (uint64_t) (sl[idx].x & 0x7FFFFFFFu); }
	__device__ float2 scores_of(const int idx) const { const uint32_t v = sl[idx].y; return make_float2((float) ((v >> 16) & 0xFFu), (float) (v >> 24)); }
	__device__ bool room_for(const int n_new) const { return entries + n_new <= cap; }
	__device__ bool list_room(const int n_listed) const { return n_listed <= list_cap; }
	__device__ void remember(const int at, const uint32_t e) { (void) at; (void) e; }
	__device__ bool hopeless(const int done, const int total) const { (void) done; (void) total; return false; }      /* (only ever used below the largest map) */
};

struct HbmVotes {
	static const bool kLds = false;
	/* The reference's table itself, 2^bits entries of 16 bytes: { bin | kListed (kEmptyKey: free), forward score, reverse score }
	 * -- key and scores in ONE 64-byte sector (round 6; two arrays until then: over a 2 Gbp genome, where every sub-read casts
	 * 5 000 votes into 5 000 bins and none fits an LDS map, the kernel ran at the device's random-access rate, and half of its
	 * sectors were the second array and the 1 MB memset per read).  The table belongs to the wave, not to the read: the wave takes
	 * reads off a ticket counter, remembers the slots it opens (undo[]) and frees exactly those when the read is done. */
	unsigned long long *tab;  /* entry e at tab[2 e], tab[2 e + 1] = (forward bits) | (reverse bits) << 32 */
	uint32_t *rlist;
	uint32_t *undo;
	const uint8_t *gseq;     /* the read, seq_bytes of it readable (its NUL included) */
	int seq_bytes;
	int entries;
	unsigned long long last_sc;      /* scores of the entry this lane's last find() matched */
	/* lanes of one wave share the table through L2: every access is an agent-scope atomic (no stale L1 lines), so ordering them
	 * needs no more than the wave's own memory operations completing -- a workgroup fence; an agent-scope one would write L2 back */
	__device__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
	__device__ int char_at(const int p) const { return p < seq_bytes ? gseq[p] : 0; }
	__device__ int find(const uint32_t e, const uint64_t bin, int &idx, bool &listed) {
		/* (both halves of the entry in flight at once: one sector) */
		const uint64_t key = __hip_atomic_load(&tab[2 * (size_t) e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint64_t sc = __hip_atomic_load(&tab[2 * (size_t) e + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (key == kEmptyKey) { last_sc = 0ull; return kProbeFree; }      /* (whoever opens it starts from 0 / 0) */
		if ((key & ~kListed) == bin) { idx = (int) e; listed = (key & kListed) != 0ull; last_sc = sc; return kProbeMatch; }
		return kProbeOther;
	}
	__device__ void claim(const uint32_t e, const uint64_t bin, int &idx, bool &creator, bool &hazard) {
		unsigned long long expected = kEmptyKey;
		const bool won = __hip_atomic_compare_exchange_strong(&tab[2 * (size_t) e], &expected, (unsigned long long) bin, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		idx = (int) e;
		creator = won;
		hazard = !won && (expected & ~kListed) != bin;
		if (won) __hip_atomic_store(&tab[2 * (size_t) e + 1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__device__ bool verify(const int idx, const uint64_t bin) const { (void) idx; (void) bin; return true; }
	__device__ void unclaim(const int idx, const uint32_t e) { (void) idx; __hip_atomic_store(&tab[2 * (size_t) e], (unsigned long long) kEmptyKey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	__device__ void remember(const int at, const uint32_t e) { undo[at] = e; }
	/* the score the lane's last find() saw (0 for a slot that was free then: opened in this very batch) */
	__device__ float score(const int idx, const bool rev) const { (void) idx; return __uint_as_float((uint32_t) (rev ? last_sc >> 32 : last_sc)); }
	__device__ void set_score(const int idx, const bool rev, const float s) {
		__hip_atomic_store(reinterpret_cast<uint32_t *>(&tab[2 * (size_t) (uint32_t) idx + 1]) + (rev ? 1 : 0), __float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__device__ void set_listed(const int idx, const uint32_t e) { (void) e; __hip_atomic_fetch_or(&tab[2 * (size_t) (uint32_t) idx], (unsigned long long) kListed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	__device__ void list_put(const int pos, const int idx) { rlist[pos] = (uint32_t) idx; }
	__device__ int list_at(const int pos) const { return (int) __hip_atomic_load(&rlist[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	__device__ uint64_t bin_of(const int idx) const { return __hip_atomic_load(&tab[2 * (size_t) (uint32_t) idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~kListed; }
	__device__ float2 scores_of(const int idx) const {
		const unsigned long long sc = __hip_atomic_load(&tab[2 * (size_t) (uint32_t) idx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return make_float2(__uint_as_float((uint32_t) sc), __uint_as_float((uint32_t) (sc >> 32)));
	}
	__device__ bool room_for(const int n_new) const { (void) n_new; return true; }
	__device__ bool list_room(const int n_listed) const { (void) n_listed; return true; }
	__device__ static bool fits(const uint64_t b) { (void) b; return true; }
	__device__ static bool score_fits(const float s) { (void) s; return true; }
	__device__ bool hopeless(const int done, const int total) const { (void) done; (void) total; return false; }
	/* the read is done (or its attempt ran out of budget): the slots it opened are free again */
	__device__ void cleanup(const int lane) {
		fence();
		for (int j = lane; j < entries; j += 64) __hip_atomic_store(&tab[2 * (size_t) undo[j]], (unsigned long long) kEmptyKey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		fence();
		entries = 0;
	}
};

/* the running state of a read's vote (CS::RunRead's locals, CS.cpp:324-398) */
struct VoteState {
	long long hpoc;
	float max_hit, thresh;
	int rlen, misses;
	bool overflow, too_many;
};

template <class TABLE>
__device__ void search_vote_read(const SearchArgs &a, TABLE &tb, ChunkRows &C, const int i, const int lane, SearchCandidate *out) {
	const int bits = a.bits;
	const uint32_t size = 1u << bits;
	const int read_len = a.seq_len[i];
	const int K = a.k;
	const uint64_t lt = (1ull << lane) - 1ull, gt = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
	VoteState S;
	S.hpoc = (long long) ((float) size * a.hpoc_factor);       /* CS.cpp:352 / :379 */
	S.max_hit = 0.0f; S.thresh = 0.0f; S.rlen = 0; S.misses = 0; S.overflow = false; S.too_many = false;

	/* one vote, every lane with the same arguments (uniform control flow); lane 0 writes.  CS.cpp:101-149 */
	auto vote_serial = [&](const uint64_t bin, const bool reverse) {
		uint32_t e = (uint32_t) ((bin * 11400714819323199488ull) >> (64 - bits));
		int idx = 0;
		bool listed = false;
		int st;
		for (;;) {
			st = tb.find(e, bin, idx, listed);
			if (st != kProbeOther) break;
			if (++e >= size) e = 0;                                         /* CS.cpp:107-114 */
			if (--S.hpoc == 0) { S.overflow = true; return; }
		}
		float score = 1.0f;
		if (st == kProbeFree) {
			if (!tb.room_for(1)) { S.too_many = true; return; }
			bool creator = false, hazard = false;
			if (lane == 0) { tb.claim(e, bin, idx, creator, hazard); tb.remember(tb.entries, e); }
			tb.fence();
			idx = lane_of(idx, 0);
			tb.entries += 1;
			listed = false;
		} else {
			score = tb.score(idx, reverse) + 1.0f;
			if (!TABLE::score_fits(score)) { S.too_many = true; return; }
		}
		if (lane == 0) tb.set_score(idx, reverse, score);
		if (score > S.max_hit) { S.max_hit = score; S.thresh = S.max_hit * a.sensitivity; }
		if (!listed && score >= S.thresh) {
			if (!tb.list_room(S.rlen + 1)) { S.too_many = true; return; }
			if (lane == 0) { tb.list_put(S.rlen, idx); tb.set_listed(idx, e); }
			S.rlen += 1;
		}
		tb.fence();
	};

	/* the batch's votes one by one (lane j holds vote j: bin, orientation, k-mer index in the chunk) */
	auto cast_serial = [&](const int m, const uint64_t bin, const bool rev, const int c) {
		for (int j = 0; j < m; ++j) {
			vote_serial(lane_of(bin, j), lane_of((int) rev, j) != 0);
			if (S.overflow) { S.misses += C.miss_upto[lane_of(c, j)]; return; }
			if (S.too_many) return;
		}
	};

	/* <= 64 consecutive votes at once; `active` lanes hold them in casting order */
	auto cast_batch = [&](const bool active, const uint64_t bin, const bool rev, const int c) {
		const int m = __popcll(__ballot(active));
		/* probe the table as it stands */
		uint32_t e = (uint32_t) ((bin * 11400714819323199488ull) >> (64 - bits));
		uint32_t steps = 0;
		int idx = 0, st = kProbeOther;
		bool listed = false;
		if (active) {
			for (;;) {
				st = tb.find(e, bin, idx, listed);
				if (st != kProbeOther) break;
				if (++e >= size) e = 0;
				if ((long long) ++steps >= S.hpoc) break;               /* this vote alone exhausts the budget */
			}
		}
		const uint32_t total_steps = (uint32_t) lane_of((int) wave_incl_sum(active ? steps : 0u, lane), 63);
		const int n_free = __popcll(__ballot(active && st == kProbeFree));
		if ((long long) total_steps >= S.hpoc || !tb.room_for(n_free)) { cast_serial(m, bin, rev, c); return; }
		/* new entries; two votes of the batch that open the same slot for different bins cannot be ordered here */
		bool creator = false, hazard = false;
		if (active && st == kProbeFree) tb.claim(e, bin, idx, creator, hazard);
		tb.fence();
		if (active && st == kProbeFree && !creator && !tb.verify(idx, bin)) hazard = true;
		if (__ballot(hazard) != 0ull) {
			if (creator) tb.unclaim(idx, e);
			tb.fence();
			cast_serial(m, bin, rev, c);
			return;
		}
		S.hpoc -= (long long) total_steps;
		const uint64_t created = __ballot(creator);
		if (creator) tb.remember(tb.entries + __popcll(created & lt), e);
		tb.entries += __popcll(created);
		/* votes of the batch for the same entry: ranked in lane order per orientation, listed once */
		uint64_t grp_h = 1ull << lane, grp_hr = 1ull << lane;
		if (active) C.scr[(uint32_t) idx & 1023u] = (uint8_t) lane;      /* (1 024 buckets: 64 lanes over 256 met by chance eight times a batch, each a trip of the loop below) */
		lds_fence();
		uint64_t todo = __ballot(active && C.scr[(uint32_t) idx & 1023u] != (uint8_t) lane);
		while (todo != 0ull) {
			const int l = __ffsll((unsigned long long) todo) - 1;
			const int kh = lane_of(idx, l);
			const uint64_t same_h = __ballot(active && idx == kh);
			const uint64_t same_f = __ballot(active && idx == kh && !rev);
			if (active && idx == kh) { grp_h = same_h; grp_hr = rev ? (same_h & ~same_f) : same_f; }
			todo &= ~same_h;
		}
		float s = 0.0f;
		if (active) s = tb.score(idx, rev) + (float) (__popcll(grp_hr & lt) + 1);
		if (TABLE::kLds && __ballot(active && !TABLE::score_fits(s)) != 0ull) { S.too_many = true; return; }      /* (the attempt is discarded: what it already changed does not matter) */
		const float pm = wave_incl_max(s, lane);
		const float mh = pm > S.max_hit ? pm : S.max_hit;                 /* maxHitNumber right after this vote */
		const bool qual = active && s >= mh * a.sensitivity;
		const uint64_t qb = __ballot(qual);
		const bool lister = qual && !listed && (qb & grp_h & lt) == 0ull;
		const uint64_t lb = __ballot(lister);
		if (TABLE::kLds && !tb.list_room(S.rlen + __popcll(lb))) { S.too_many = true; return; }
		if (lister) { tb.list_put(S.rlen + __popcll(lb & lt), idx); tb.set_listed(idx, e); }
		S.rlen += __popcll(lb);
		if (active && (grp_hr & gt) == 0ull) tb.set_score(idx, rev, s);
		const float last = lane_of(pm, 63);
		if (last > S.max_hit) { S.max_hit = last; S.thresh = S.max_hit * a.sensitivity; }
		tb.fence();
	};

	/* CSstatic.cpp:23-73, the walk, 64 window positions at a time (round 6; until then one k-mer per trip of a serial loop: 30 % of a
	 * read's time): kmer_windows() / kmer_at() above. */
	auto char_at = [&](const int p) -> int { return tb.char_at(p); };
	const int n_win = kmer_windows(read_len, K, char_at);
	/* lane t of the chunk that starts at window q0: its k-mer and the table rows of the k-mer and of its reverse complement (one
	 * 8-byte record each: start, length | used << 31).  The loads are issued here, a chunk ahead of their use. */
	struct Kmers { bool ok; uint2 rf, rr; };
	auto lookup = [&](const int q0) {
		Kmers k;
		const int q = q0 + lane;
		uint64_t prefix = 0;
		k.ok = q < n_win && kmer_at(q, K, char_at, prefix);
		/* (unconditional loads -- a lane without a k-mer reads record 0 and drops it: a load inside a branch makes the compiler wait
		 * for ALL loads in flight where the branches join, and these are meant to stay in flight for a whole chunk) */
		if (!k.ok) prefix = 0;
		const uint2 rf = a.rows[prefix], rr = a.rows[rev_comp13(prefix, K)];
		k.rf = k.ok ? rf : make_uint2(0u, 0u); k.rr = k.ok ? rr : make_uint2(0u, 0u);
		return k;
	};

	/* the votes of the chunk's k-mers, in order (lane c = the k-mer at window q0 + c; a window with an 'N' has no rows and is no miss) */
	auto cast_chunk = [&](const Kmers &k, const int q0) {
		const bool uf = (k.rf.y >> 31) != 0u, ur = (k.rr.y >> 31) != 0u;
		const uint32_t n0 = uf ? k.rf.y & 0x7FFFFFFFu : 0u, n1 = ur ? k.rr.y & 0x7FFFFFFFu : 0u;
		const bool miss = k.ok && !uf && !ur;                  /* entries[0].refTotal == 0 (PrefixTable.cpp:489-525): kCount, CS.cpp:67-69 */
		C.c_start[0][lane] = uf ? k.rf.x : 0u; C.c_start[1][lane] = ur ? k.rr.x : 0u;
		const uint32_t incl = wave_incl_sum(n0 + n1, lane);
		const uint32_t first_f = incl - n0 - n1, first_r = incl - n1;      /* votes before this lane's forward / reverse row */
		C.row_first[2 * lane] = first_f;
		C.row_first[2 * lane + 1] = first_r;
		C.miss_upto[lane] = (uint16_t) wave_incl_sum(miss ? 1u : 0u, lane);
		lds_fence();
		const uint32_t votes = (uint32_t) lane_of((int) incl, 63);
		/* lane t of the batch that starts at vote v0: its table row -- the last non-empty row that starts at or before vote v0 + t --
		 * and its location.  The rows that start inside the batch drop their number at the vote they start with (non-empty rows
		 * start at different votes), a running maximum carries it over the votes behind, `carry` (the row of the vote before the
		 * batch) over the votes in front of the first: four LDS instructions and a DPP scan where a binary search over the 128 row
		 * starts made seven dependent LDS round trips.  The location's load is issued here and waited for where the bin is
		 * computed, one batch later: the round trip to HBM runs under the casting of the batch before. */
		struct Fetched { bool active, rev; int c; uint32_t loc; };
		uint32_t carry = 0u;
		auto fetch = [&](const uint32_t v0) {
			Fetched f;
			const uint32_t v = v0 + (uint32_t) lane;
			C.row_at[lane] = 0;
			lds_fence();
			if (n0 != 0u && first_f - v0 < 64u) C.row_at[first_f - v0] = (uint8_t) (2 * lane);          /* (unsigned: a row from before v0 wraps far beyond 64) */
			if (n1 != 0u && first_r - v0 < 64u) C.row_at[first_r - v0] = (uint8_t) (2 * lane + 1);
			lds_fence();
			uint32_t row = wave_incl_umax((uint32_t) C.row_at[lane]);
			row = row > carry ? row : carry;
			carry = (uint32_t) lane_of((int) row, 63);
			f.active = v < votes; f.rev = (row & 1u) != 0u; f.c = (int) (row >> 1);
			uint32_t at = C.c_start[row & 1u][f.c] + (v - C.row_first[row]);
			if (!f.active) at = 0u;
			f.loc = a.locs[at];      /* (unconditional, as in lookup()) */
			return f;
		};
		Fetched cur = fetch(0u);
		for (uint32_t v0 = 0; v0 < votes && !S.overflow && !S.too_many; v0 += 64) {
			const Fetched nxt = fetch(v0 + 64u);      /* (behind the last batch: no lane active) */
			uint64_t bin = 0;
			if (cur.active) {
				const unsigned long long loc = (unsigned long long) cur.loc + a.unit_offset;
				const unsigned long long pos = (unsigned long long) (q0 + cur.c);
				const unsigned long long corr = cur.rev ? (unsigned long long) read_len - (pos + (unsigned long long) K) : pos;
				bin = (loc - corr) >> a.bin_shift;
			}
			if (TABLE::kLds && __ballot(cur.active && !TABLE::fits(bin)) != 0ull) { S.too_many = true; break; }
			cast_batch(cur.active, bin, cur.rev, cur.c);
			cur = nxt;
		}
		if (!S.overflow && !S.too_many) S.misses += C.miss_upto[63];
	};

	Kmers kc = lookup(0);
	for (int q0 = 0; q0 < n_win && !S.overflow && !S.too_many; q0 += 64) {
		const Kmers kn = lookup(q0 + 64);      /* (behind the last chunk: no lane with a k-mer) */
		cast_chunk(kc, q0);
		kc = kn;
		if (tb.hopeless(q0 + 64, n_win)) S.too_many = true;
	}

	if (S.too_many) { if (lane == 0) a.n_cand[i] = kSearchNeedsHbm; return; }      /* redone over a table in HBM: nothing of this attempt counts */
	/* kCount is reset per read, not per attempt (CS.cpp:338): the k-mers an overflowed attempt visited stay counted */
	if (lane == 0 && a.kmer_misses) a.kmer_misses[i] += S.misses;
	if (S.overflow) { if (lane == 0) a.n_cand[i] = -1; return; }
	/* CollectResultsStd, CS.cpp:219-268: rList order, forward before reverse; 64 list entries per pass */
	const float thr = a.min_hits > S.thresh ? a.min_hits : S.thresh;
	const unsigned long long half = a.bin_shift > 0 ? 1ull << (a.bin_shift - 1) : 0ull;
	int n = 0;
	for (int r0 = 0; r0 < S.rlen; r0 += 64) {
		const int r = r0 + lane;
		int cnt = 0;
		float vf = 0.0f, vr = 0.0f;
		unsigned long long bin = 0;
		if (r < S.rlen) {
			const int h = tb.list_at(r);
			const float2 sc = tb.scores_of(h);
			vf = sc.x; vr = sc.y;
			bin = tb.bin_of(h);
			cnt = (vf >= thr ? 1 : 0) + (vr >= thr ? 1 : 0);
		}
		const int incl = (int) wave_incl_sum((uint32_t) cnt, lane);
		int at = n + incl - cnt;
		if (vf >= thr && r < S.rlen) { SearchCandidate cd; cd.location = (bin << a.bin_shift) + half; cd.score = vf; cd.reverse = 0; out[at++] = cd; }
		if (vr >= thr && r < S.rlen) { SearchCandidate cd; cd.location = (bin << a.bin_shift) + half; cd.score = vr; cd.reverse = 1; out[at++] = cd; }
		n += lane_of(incl, 63);
	}
	if (lane == 0) { a.n_cand[i] = n; a.max_hit[i] = S.max_hit; }
}
}  // namespace

template <class MAP>
__global__ void __launch_bounds__(64)
search_wave_kernel(const SearchArgs a, const int log2s, const int seq_cap) {
	extern __shared__ __attribute__((aligned(16))) uint32_t map_lds[];      /* (the 8-byte map reads its slots as 64-bit words) */
	__shared__ ChunkRows C;
	const int q = blockIdx.x;
	if (q >= a.n_work) return;
	const int lane = threadIdx.x;
	const int i = a.work ? a.work[q] : q;
	const int read_len = a.seq_len[i];
	if (read_len + 65 > seq_cap) { if (lane == 0) a.n_cand[i] = kSearchNeedsHbm; return; }      /* (the host sized seq_cap for the launch's longest read: cannot happen) */
	const uint8_t *gseq = a.seq + a.seq_off[i];
	MAP tb;
	tb.carve(map_lds, log2s);
	tb.clear(lane);
	for (int s = lane; s < read_len + 64; s += 64) tb.seq[s] = s < read_len ? gseq[s] : (uint8_t) 0;      /* coalesced; NULs behind the read */
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	search_vote_read(a, tb, C, i, lane, a.cand + a.cand_off[i]);
}

/* the same over the real table in HBM: block b owns table b of the launch (clean when the launch starts, clean when it ends)
 * and takes reads off the ticket counter until none is left */
__global__ void __launch_bounds__(64)
search_wave_hbm_kernel(const SearchArgs a, unsigned int *ticket) {
	__shared__ ChunkRows C;
	const int lane = threadIdx.x;
	const size_t size = (size_t) 1 << a.bits;
	HbmVotes tb;
	tb.tab = reinterpret_cast<unsigned long long *>(a.keys) + 2 * (size_t) blockIdx.x * size;
	tb.entries = 0; tb.last_sc = 0ull;
	for (;;) {
		unsigned int q = 0;
		if (lane == 0) q = atomicAdd(ticket, 1u);
		q = (unsigned int) lane_of((int) q, 0);
		if (q >= (unsigned int) a.n_work) break;
		const int i = a.work ? a.work[q] : (int) q;
		tb.seq_bytes = a.seq_len[i] + 1;
		tb.rlist = a.rlist + a.list_off[i];
		tb.undo = a.undo + a.list_off[i];
		tb.gseq = a.seq + a.seq_off[i];
		search_vote_read(a, tb, C, i, lane, a.cand + 2ull * a.list_off[i]);
		tb.cleanup(lane);
	}
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

size_t search_wave_lds_bytes(const int log2s, const int seq_cap, const bool slot8) { return slot8 ? LdsVotes8::bytes(log2s, seq_cap) : LdsVotes::bytes(log2s, seq_cap); }

hipError_t launch_search_wave(const SearchArgs &a, const int log2s, const int seq_cap, const bool slot8, hipStream_t st) {
	if (a.n_work <= 0) return hipSuccess;
	if (log2s < kSearchWaveLog2Min || log2s > kSearchWaveLog2Max || seq_cap < 65 || seq_cap > kSearchWaveSeq + 64 || (seq_cap & 3)) return hipErrorInvalidValue;
	if (slot8) {
		if (a.bits > 16 || log2s == kSearchWaveLog2Max) return hipErrorInvalidValue;      /* 16-bit virtual slots; the largest map keeps the 12-byte form (its give-up rule) */
		hipLaunchKernelGGL(search_wave_kernel<LdsVotes8>, dim3(a.n_work), dim3(64), LdsVotes8::bytes(log2s, seq_cap), st, a, log2s, seq_cap);
	} else {
		hipLaunchKernelGGL(search_wave_kernel<LdsVotes>, dim3(a.n_work), dim3(64), LdsVotes::bytes(log2s, seq_cap), st, a, log2s, seq_cap);
	}
	return hipGetLastError();
}

hipError_t launch_search_wave_hbm(const SearchArgs &a, int n_tables, unsigned int *ticket, hipStream_t st) {
	if (a.n_work <= 0) return hipSuccess;
	if (n_tables <= 0 || !ticket) return hipErrorInvalidValue;
	hipLaunchKernelGGL(search_wave_hbm_kernel, dim3(a.n_work < n_tables ? a.n_work : n_tables), dim3(64), 0, st, a, ticket);
	return hipGetLastError();
}

hipError_t launch_search_count(const SearchArgs &a, hipStream_t st) {
	if (a.n <= 0) return hipSuccess;
	hipLaunchKernelGGL(search_count_kernel, dim3((a.n + 3) / 4), dim3(256), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_search(const SearchArgs &a, hipStream_t st) {
	if (a.n_work <= 0) return hipSuccess;
	hipLaunchKernelGGL(search_kernel, dim3((a.n_work + 63) / 64), dim3(64), 0, st, a);
	return hipGetLastError();
}

}  // namespace cvx
