/* cvx_launch.h -- host-callable launchers of the gfx950 kernels (cvx_kernels.hip). */
#ifndef CVX_LAUNCH_H
#define CVX_LAUNCH_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "cvx_types.h"

namespace cvx {

/* m in {1,2,3,4}.  mode (cvx_kernels.hip FillMode): 0 two-phase best-cell tracking (flags
 * undecided tiles), 1 the exact pass that redoes the flagged tiles of the same list (launch both, in
 * this order, on one stream), 2 chained row blocks (a.tasks, list_n = number of tasks). */
/* gang > 1 (2 or 3; m = 3, float runs, modes 0 / 1): that many waves share the tile's ring (cvx_kernels.hip, GANG) */
hipError_t launch_fill(int m, int gang, bool wrap, int mode, const FillArgs &a, size_t pad_lds, hipStream_t st);
hipError_t launch_chain_reduce(const int32_t *tiles, int n_tiles, const TileRun *trun, const ChainOut *cout, TileOut *tout, hipStream_t st);
/* sub-read scoring (cvx_score.hip) */
struct ScorePair {
	uint64_t ref_off, qry_off;   /* byte offsets in the sequence buffer (strings keep their NUL) */
	uint64_t scratch_off;        /* int offset of this pair's two DP rows */
	int32_t ref_len, qry_len;    /* strlen + 1 */
};
/* max_ref_len: longest reference string of the batch incl. its NUL (<= 512: rows in registers, four pairs per workgroup) */
hipError_t launch_score(const uint8_t *seq, const ScorePair *pairs, int32_t *scratch, float *out, int n, int max_ref_len, hipStream_t st);
/* every pair has qry_len <= 512 and ref_len <= 2048 (NULs included): diagonal kernel, no serial dependency (cvx_score.hip) */
hipError_t launch_score_diag(const uint8_t *seq, const ScorePair *pairs, float *out, int n, hipStream_t st);

/* candidate search (cvx_search.hip): k-mer vote of a batch of reads over the resident k-mer table */
struct SearchCandidate {         /* same layout as cvx_candidate (include/cvx_align.h) */
	uint64_t location;
	float score;
	int32_t reverse;
};
struct SearchArgs {
	/* the table: ONE 8-byte record per prefix (round 6) -- x = where its row starts in RefTable (m_TabIndex - 1), y = the row's length
	 * with Index::used() in bit 31 -- instead of m_TabIndex[p], m_TabIndex[p + 1] and a byte of flags in two arrays: a k-mer's look-up
	 * is one sector of HBM, not two (the look-ups were half of the search's traffic at genome scale); then RefTable */
	const uint2 *rows;
	const uint32_t *locs;
	unsigned long long unit_offset;
	int32_t k;
	/* the reads: bytes with a NUL behind each */
	const uint8_t *seq;
	const uint64_t *seq_off;
	const int32_t *seq_len;
	int32_t n;
	/* per read */
	unsigned long long *events;      /* votes it will cast (search_count_kernel) */
	const uint64_t *list_off;        /* exclusive prefix sum of events: rList region; candidates at twice that */
	uint32_t *rlist;
	uint32_t *undo;                  /* search_wave_hbm_kernel: the table slots a read opened, same regions as rlist */
	SearchCandidate *cand;
	int32_t *n_cand;                 /* -1: the probe budget ran out at this table size */
	float *max_hit;
	int32_t *kmer_misses;            /* += k-mers found in neither orientation (kCount, CS.cpp:67-69); NULL: not wanted */
	/* this attempt */
	const int32_t *work;             /* read indices (NULL: all n) */
	int32_t n_work;
	int32_t bits;                    /* vote table of 2^bits entries per read in flight */
	float hpoc_factor;               /* 0.333 first attempt, 0.777 on retries (CS.cpp:352, :379) */
	uint64_t *keys;
	float *scores;                   /* two floats per table entry: forward, reverse */
	float sensitivity, min_hits;
	int32_t bin_shift;
	/* search_wave_kernel (one wave per read, vote table in LDS): candidates of read i at cand + cand_off[i] (room for two per
	 * entry the read can have: search_wave_cand()); n_cand[i] = kSearchNeedsHbm when the read does not fit the LDS table */
	const uint64_t *cand_off;
};
/* the LDS vote map of a read has 2^log2s slots and holds 3/4 of that many bins (cvx_search.hip, LdsVotes) */
static const int kSearchWaveLog2Min = 9, kSearchWaveLog2Max = 12, kSearchWaveLog2Default = 11;
static const int kSearchWaveSeq = 4096;              /* longest read (with its NUL) the LDS form takes */
static inline int search_wave_entries(const int log2s) { return (3 << log2s) / 4; }
/* the smallest map that certainly holds a read casting `votes` votes (a bin takes at least one); the largest for more than that */
static inline int search_wave_log2(const unsigned long long votes) {
	for (int l = kSearchWaveLog2Min; l < kSearchWaveLog2Max; ++l) if (votes <= (unsigned long long) search_wave_entries(l)) return l;
	return kSearchWaveLog2Max;
}
static const int kSearchNeedsHbm = -2;
size_t search_wave_lds_bytes(int log2s, int seq_cap, bool slot8);
hipError_t launch_search_count(const SearchArgs &a, hipStream_t st);
hipError_t launch_search(const SearchArgs &a, hipStream_t st);
/* seq_cap: bytes of LDS for the read, >= the launch's longest read + 65, a multiple of 4 */
/* slot8: the 8-byte map (a.bits <= 16, counts up to 255, rList of a quarter of the slots; not for the largest map) */
hipError_t launch_search_wave(const SearchArgs &a, int log2s, int seq_cap, bool slot8, hipStream_t st);
/* a wave per read over the real table in HBM: n_tables tables of 2^bits 16-byte entries at a.keys, every one empty when the
 * launch starts and when it ends; *ticket = 0 */
hipError_t launch_search_wave_hbm(const SearchArgs &a, int n_tables, unsigned int *ticket, hipStream_t st);
/* dense[dst_begin[i] ...) = the n_cand[i] candidates of read i, which lie at sparse + src_off[i] */
hipError_t launch_search_compact(const SearchCandidate *sparse, const uint64_t *src_off, const int32_t *n_cand, const uint64_t *dst_begin,
		SearchCandidate *dense, int n, hipStream_t st);

/* reference windows from the 4-bit genome resident in HBM (cvx_genome.hip) */
hipError_t launch_decode_windows(const uint8_t *bin, const uint64_t *starts, int n_starts,
		const WindowDesc *win, int n, uint8_t *dst, hipStream_t st);

/* device-side text stage (cvx_text.hip): lengths + fields + offsets, then the strings */
hipError_t launch_text_size(const TextArgs &a, hipStream_t st);
hipError_t launch_text_write(const TextArgs &a, hipStream_t st);
/* nmPerPosition of the tiles [first, first + count): entry counts (TextRec::nm_count, after launch_text_size) -> offsets
 * (count values + the total behind them), then the triples */
hipError_t launch_nm_offsets(const TextArgs &a, int first, int count, unsigned long long *len, unsigned long long *off, unsigned long long *total, hipStream_t st);
hipError_t launch_nm_profile(const TextArgs &a, int first, int count, const unsigned long long *off, int32_t *triples, hipStream_t st);

/* catch-all kernel (cvx_generic.hip): any ring size, state in a global scratch */
size_t generic_scratch_bytes(int ring);
/* sse_variant: the reference's SSE-path semantics for scoring outside the scalar-equivalent regime */
hipError_t launch_fill_generic(const FillArgs &a, bool sse_variant, uint8_t *scratch, const uint64_t *scratch_off, hipStream_t st);
/* the rows arena from the one-byte-per-row step stream (+ verbatim rows of the tiles that do not fit it) */
/* closed_forms: also write out the rows of closed-form corridors (cvx_corridor_rows; the product evaluates those in registers) */
hipError_t launch_expand_rows(const RowSrc *rsrc, const TileIn *tin, const uint8_t *delta, const RowDesc *rowsx, RowDesc *rows,
		int n_tiles, bool closed_forms, hipStream_t st);
/* rows_per_tile: mean read rows per tile of the batch (picks 256 threads or one wave per tile) */
hipError_t launch_plan(const RowDesc *rows, const RowSrc *rsrc, const TileIn *tin, TilePlan *plan, int n_tiles, uint64_t rows_per_tile,
		unsigned long long max_matrix_mb, hipStream_t st);
/* group 8 / 16 / 32: that many lanes per tile, 64 / group tiles per wave (16 is reached through CVX_TUNE_BT_GROUP only);
 * anything else: one wave per tile.  Tiles are taken from order[0, n_order) (longest read first); order == NULL (one wave
 * per tile only): tile b for block b over all a.n_tiles */
hipError_t launch_backtrack(const BacktrackArgs &a, const int32_t *order, int n_order, int group, hipStream_t st);
hipError_t launch_finalize(const TileOut *tout, const TilePlan *plan, uint64_t *dst_off, ResultRec *res,
		BatchSummary *sum, int32_t *counters, int n_tiles, uint64_t dense_cap, hipStream_t st);
hipError_t launch_compact(const int32_t *regions, const TileRun *trun, const TileOut *tout,
		const uint64_t *dst_off, uint32_t *dense, int n_tiles, uint64_t dense_cap, hipStream_t st);

}  // namespace cvx
#endif
