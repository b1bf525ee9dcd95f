/*
 * cs_oracle.c -- CPU restatement of ngmlr's candidate search for one (sub-)read (search half).
 *
 * TEST INFRASTRUCTURE ONLY: nothing in the product path may include, link or call this file; tests/ use it as
 * the checker of the device kernel (cvx_search.hip).
 *
 * Parity status: PINNED -- tests/test_search_cpu.py checks it against every candidate-search call recorded from
 * the unmodified reference on its own test_3 reads (tools/make_golden_cs.sh: 5 663 sub-reads, LocationScore lists
 * in the reference's own order, maxHitNumber, the threshold, and kmer_misses = kCount as the reference left it -- CS.cpp:26,
 * 67-69,221-224,338: summed over the attempts of the retry ladder, which is why the recording also holds the table size of
 * each read's first attempt (the reference adapts it per batch, CS.cpp:482-489: 2^8 .. 2^16 on test_3).  kCount is a global
 * that all CS threads increment without synchronisation; the recording is a -t 1 run, where it is the read's own count, and
 * that is the meaning restated here).  The reference's CS class itself cannot be
 * compiled on its own (it drags NGM, the task system, the read providers and the output writers along), so there
 * is no oracle/_ref build of it; the recorded calls are the anchor.
 *
 * What is restated, with the reference lines it follows (all under /root/reference/src):
 *   CS::PrefixIteration      CSstatic.cpp:23-73     every 13-mer of the read, 2 bits per base as (c >> 1) & 3, a
 *                                                   window that holds an 'N' is skipped by restarting behind it --
 *                                                   including the quirk that a run of N at the start of a (re)started
 *                                                   stretch ends the walk when 13 or fewer characters follow it
 *   CS::PrefixSearch         CS.cpp:57-99           per k-mer: the table row of the k-mer (forward hits) and of its
 *                                                   reverse complement (reverse hits); every location, moved back by the
 *                                                   k-mer's offset in the read (forward: pos, reverse: readLength - (pos +
 *                                                   13)), is binned (>> binShift) and voted for
 *   CompactPrefixTable::GetRefEntry  PrefixTable.cpp:476-532, revComp :45-59
 *   CS::AddLocationStd       CS.cpp:101-149         open-addressing vote table keyed by bin (multiplicative hash,
 *                                                   linear probing, a budget of probe steps -- `hpoc` -- whose
 *                                                   exhaustion throws), one forward and one reverse score per bin;
 *                                                   maxHitNumber / currentThresh = maxHitNumber * sensitivity grow as
 *                                                   votes arrive, and a bin enters rList the first time one of its
 *                                                   scores reaches the threshold OF THAT MOMENT: the list order
 *                                                   depends on the order of the votes
 *   CS::CollectResultsStd    CS.cpp:219-268         threshold = max(minKmerHits, currentThresh); rList in order, forward
 *                                                   score first; location = (bin << binShift) + (1 << (binShift - 1))
 *   CS::RunRead              CS.cpp:324-398         table of 2^16 entries with hpoc = 0.333 * size; on overflow the
 *                                                   search restarts with 2^18, 2^19, 2^20 entries and hpoc = 0.777 * size;
 *                                                   a read that overflows all of them gets no candidates
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	int k;
	uint64_t unit_offset;
	uint32_t n_index;        /* 4^k + 1 */
	uint32_t *tab;           /* m_TabIndex of every prefix (running sum + 1), n_index + 1 entries */
	uint8_t *used;           /* m_RevCompIndex != 0 */
	uint32_t *locs;
	uint32_t n_locs;
} cs_table;

typedef struct { uint64_t loc; float f, r; uint32_t state; } cs_entry;

/* the compact form the recorder writes: the used prefixes (ascending) with their slot counts */
void *cs_table_create(int k, uint64_t unit_offset, uint32_t n_used, const uint32_t *prefix, const uint32_t *cnt,
		const uint32_t *locs, uint32_t n_locs) {
	cs_table *t = (cs_table *) calloc(1, sizeof(cs_table));
	t->k = k;
	t->unit_offset = unit_offset;
	t->n_index = (1u << (2 * k)) + 1u;
	t->tab = (uint32_t *) malloc(((size_t) t->n_index + 1) * 4);
	t->used = (uint8_t *) calloc((size_t) t->n_index + 1, 1);
	t->locs = (uint32_t *) malloc(((size_t) n_locs + 1) * 4);
	memcpy(t->locs, locs, (size_t) n_locs * 4);
	t->n_locs = n_locs;
	/* createRefTableIndex, PrefixTable.cpp:277-326: m_TabIndex = next + 1 for every prefix, next += freq for the used ones */
	uint32_t next = 0, u = 0;
	for (uint32_t p = 0; p <= t->n_index; ++p) {
		t->tab[p] = next + 1;
		if (u < n_used && prefix[u] == p) { t->used[p] = 1; next += cnt[u]; ++u; }
	}
	return t;
}

/* the table as ngmlr holds it (and cvx_index_upload takes it): 4^k + 2 packed 5-byte Index records (uint m_TabIndex; char
 * m_RevCompIndex, src/PrefixTable.h:15-31) and the locations */
void *cs_table_create_raw(int k, uint64_t unit_offset, const uint8_t *index5, const uint32_t *locs, uint32_t n_locs) {
	cs_table *t = (cs_table *) calloc(1, sizeof(cs_table));
	t->k = k;
	t->unit_offset = unit_offset;
	t->n_index = (1u << (2 * k)) + 1u;
	t->tab = (uint32_t *) malloc(((size_t) t->n_index + 1) * 4);
	t->used = (uint8_t *) calloc((size_t) t->n_index + 1, 1);
	t->locs = (uint32_t *) malloc(((size_t) n_locs + 1) * 4);
	memcpy(t->locs, locs, (size_t) n_locs * 4);
	t->n_locs = n_locs;
	for (uint32_t p = 0; p <= t->n_index; ++p) {
		memcpy(&t->tab[p], index5 + 5 * (size_t) p, 4);
		t->used[p] = index5[5 * (size_t) p + 4] != 0;
	}
	return t;
}

void cs_table_destroy(void *h) {
	cs_table *t = (cs_table *) h;
	if (!t) return;
	free(t->tab); free(t->used); free(t->locs); free(t);
}

/* PrefixTable.cpp:45-59 */
static const unsigned char kReverse16[16] = { 0x00, 0x04, 0x08, 0x0C, 0x01, 0x05, 0x09, 0x0D, 0x02, 0x06, 0x0A, 0x0E, 0x03, 0x07, 0x0B, 0x0F };
static uint64_t rev_comp(uint64_t prefix, int k) {
	const int bits = 2 * k, shift = 32 - bits;
	const uint64_t mask = ((uint64_t) 1 << bits) - 1;
	uint64_t c = (prefix ^ 0xAAAAAAAAull) & mask;
	c <<= shift;
	return ((uint64_t) kReverse16[c & 0x0f] << 28) | ((uint64_t) kReverse16[(c >> 4) & 0x0f] << 24) | ((uint64_t) kReverse16[(c >> 8) & 0x0f] << 20)
			| ((uint64_t) kReverse16[(c >> 12) & 0x0f] << 16) | ((uint64_t) kReverse16[(c >> 16) & 0x0f] << 12) | ((uint64_t) kReverse16[(c >> 20) & 0x0f] << 8)
			| ((uint64_t) kReverse16[(c >> 24) & 0x0f] << 4) | (uint64_t) kReverse16[(c >> 28) & 0x0f];
}

typedef struct {
	const cs_table *t;
	cs_entry *table;
	uint32_t *rlist;
	int bits, rlist_len;
	uint32_t state;
	long hpoc;
	float max_hit, thresh, sens;
	int bin_shift, read_len;
	int overflow;
	int misses;              /* kCount: k-mers found in neither orientation (CS.cpp:67-69) */
} cs_run;

/* CS.cpp:101-149 */
static void add_location(cs_run *c, uint64_t bin, int reverse) {
	const uint32_t len = 1u << c->bits;
	uint32_t e = (uint32_t) ((bin * 11400714819323199488ull) >> (64 - c->bits));
	int is_cur;
	while ((is_cur = ((c->table[e].state & 0x7FFFFFFFu) == c->state)) && c->table[e].loc != bin) {
		if (++e >= len) e = 0;
		if (--c->hpoc == 0) { c->overflow = 1; return; }
	}
	float score = 1.0f;
	if (!is_cur) {
		c->table[e].loc = bin;
		c->table[e].state = c->state & 0x7FFFFFFFu;
		c->table[e].f = reverse ? 0.0f : 1.0f;
		c->table[e].r = reverse ? 1.0f : 0.0f;
	} else if (reverse) score = (c->table[e].r += 1.0f);
	else score = (c->table[e].f += 1.0f);
	if (score > c->max_hit) { c->max_hit = score; c->thresh = c->max_hit * c->sens; }
	if (!(c->table[e].state & 0x80000000u) && score >= c->thresh) {
		c->table[e].state |= 0x80000000u;
		c->rlist[c->rlist_len++] = e;
	}
}

/* CS.cpp:57-99 + PrefixTable.cpp:476-532 (one table unit) */
static void prefix_search(cs_run *c, uint64_t prefix, uint64_t pos) {
	const cs_table *t = c->t;
	const uint64_t pr[2] = { prefix, rev_comp(prefix, t->k) };
	/* CS.cpp:67-69: `cur->refTotal == 0` of the forward entry = neither the k-mer's row nor its reverse complement's is in
	 * the table (PrefixTable.cpp:489-525: the forward entry's refTotal is the sum of both rows) */
	if (!t->used[pr[0]] && !t->used[pr[1]]) c->misses += 1;
	for (int rev = 0; rev < 2 && !c->overflow; ++rev) {
		const uint64_t p = pr[rev];
		if (!t->used[p]) continue;
		const uint32_t start = t->tab[p] - 1, n = t->tab[p + 1] - 1 - start;
		const uint64_t corr = rev ? (uint64_t) c->read_len - (pos + (uint64_t) t->k) : pos;
		for (uint32_t i = 0; i < n && !c->overflow; ++i) {
			const uint64_t loc = (uint64_t) t->locs[start + i] + t->unit_offset;
			add_location(c, (loc - corr) >> c->bin_shift, rev);
		}
	}
}

/* CSstatic.cpp:23-73 (prefixskip 0: every k-mer of the read), iterative form of the tail recursion */
static void prefix_iteration(cs_run *c, const char *seq, uint64_t length) {
	const int K = c->t->k;
	const uint64_t mask = ((uint64_t) 1 << (2 * K)) - 1;
	uint64_t offset = 0;
	for (;;) {
		if (length < (uint64_t) K) return;
		if (*seq == 'N') {
			uint64_t n_skip = 1;
			while (seq[n_skip] == 'N') ++n_skip;
			seq += n_skip;
			if (n_skip >= length - (uint64_t) K) return;
			length -= n_skip;
			offset += n_skip;
		}
		uint64_t prefix = 0;
		int restart = 0;
		for (uint64_t i = 0; i < (uint64_t) K - 1; ++i) {
			const char ch = seq[i];
			if (ch == 'N') { seq += i + 1; length -= i + 1; offset += i + 1; restart = 1; break; }
			prefix = (prefix << 2) | (uint64_t) ((ch >> 1) & 3);
		}
		if (restart) continue;
		for (uint64_t i = (uint64_t) K - 1; i < length; ++i) {
			const char ch = seq[i];
			if (ch == 'N') { seq += i + 1; length -= i + 1; offset += i + 1; restart = 1; break; }
			prefix = ((prefix << 2) | (uint64_t) ((ch >> 1) & 3)) & mask;
			prefix_search(c, prefix, offset + i + 1 - (uint64_t) K);
			if (c->overflow) return;
		}
		if (!restart) return;
	}
}

/* One CS::RunRead (CS.cpp:324-398): returns the number of LocationScore entries (CollectResultsStd, :219-268) written to
 * out_* in list order, -1 when every table size overflowed ("too many candidates": the read gets none).  seq must be
 * NUL-terminated (the N-run scan relies on it, like the reference). */
int cs_search_ex(void *h, const char *seq, int len, float sensitivity, float min_hits, int bin_shift, int first_bits,
		uint64_t *out_loc, float *out_score, int32_t *out_rev, int cap, float *max_hit, float *thresh, int32_t *rlist_len, int32_t *table_bits,
		int32_t *kmer_misses);

int cs_search(void *h, const char *seq, int len, float sensitivity, float min_hits, int bin_shift,
		uint64_t *out_loc, float *out_score, int32_t *out_rev, int cap, float *max_hit, float *thresh, int32_t *rlist_len, int32_t *table_bits) {
	return cs_search_ex(h, seq, len, sensitivity, min_hits, bin_shift, 16, out_loc, out_score, out_rev, cap, max_hit, thresh, rlist_len, table_bits, 0);
}

/* The same with the table size of the first attempt as a parameter (CS::c_SrchTableBitLen, 16 when a CS thread starts and
 * adapted per batch, CS.cpp:482-489; the retries then take first_bits + 2, + 3, ... up to 20, CS.cpp:363-394) and kCount:
 * the k-mers found in neither orientation, reset per read and NOT per attempt (CS.cpp:338), so the k-mers an overflowed
 * attempt visited before it gave up stay counted. */
int cs_search_ex(void *h, const char *seq, int len, float sensitivity, float min_hits, int bin_shift, int first_bits,
		uint64_t *out_loc, float *out_score, int32_t *out_rev, int cap, float *max_hit, float *thresh, int32_t *rlist_len, int32_t *table_bits,
		int32_t *kmer_misses) {
	const cs_table *t = (const cs_table *) h;
	int misses = 0;
	if (kmer_misses) *kmer_misses = 0;
	for (int attempt = 0; ; ++attempt) {
		cs_run c;
		memset(&c, 0, sizeof(c));
		c.t = t;
		c.bits = attempt == 0 ? first_bits : first_bits + 1 + attempt;
		if (c.bits > 20) break;
		const uint32_t size = 1u << c.bits;
		c.table = (cs_entry *) malloc((size_t) size * sizeof(cs_entry));
		c.rlist = (uint32_t *) malloc((size_t) size * 4);
		for (uint32_t i = 0; i < size; ++i) { c.table[i].loc = 9223372036854775808ull; c.table[i].state = 0xFFFFFFFFu; }
		c.state = 1;
		c.hpoc = (long) ((float) size * (attempt == 0 ? 0.333f : 0.777f));
		c.sens = sensitivity;
		c.bin_shift = bin_shift;
		c.read_len = len;
		prefix_iteration(&c, seq, (uint64_t) len);
		misses += c.misses;
		if (kmer_misses) *kmer_misses = misses;
		int n = -1;
		if (!c.overflow) {
			const float thr = min_hits > c.thresh ? min_hits : c.thresh;
			const uint64_t half = bin_shift > 0 ? (uint64_t) 1 << (bin_shift - 1) : 0;
			n = 0;
			for (int i = 0; i < c.rlist_len; ++i) {
				const cs_entry e = c.table[c.rlist[i]];
				if (e.f >= thr) { if (n < cap) { out_loc[n] = (e.loc << bin_shift) + half; out_score[n] = e.f; out_rev[n] = 0; } ++n; }
				if (e.r >= thr) { if (n < cap) { out_loc[n] = (e.loc << bin_shift) + half; out_score[n] = e.r; out_rev[n] = 1; } ++n; }
			}
			if (max_hit) *max_hit = c.max_hit;
			if (thresh) *thresh = thr;
			if (rlist_len) *rlist_len = c.rlist_len;
			if (table_bits) *table_bits = c.bits;
		}
		free(c.table); free(c.rlist);
		if (n >= 0) return n;
	}
	return -1;
}
