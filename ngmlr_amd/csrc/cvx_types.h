/*
 * cvx_types.h -- structures shared by the gfx950 kernels (cvx_kernels.hip) and the
 * host runtime (cvx_runtime.cpp).  Internal; the public boundary is include/cvx_align.h.
 *
 * HBM layout of one uploaded batch (all arenas are single hipMalloc blocks):
 *
 *   seq   : bytes.  [pad][tile0 ref][tile0 qry][tile1 ref]... [pad]; a tile addresses
 *           its sequences by 32-bit byte offsets, so one batch holds < 4 GiB of bases.
 *           The pads (>= max(H+W)+RING_MAX) keep the speculative reference-character
 *           loads of idle ring slots inside the allocation.
 *   rows  : int2 (offset, length) per read row = the caller's CorridorLine[] without
 *           the 8 bytes of offsetInMatrix (reference src/IAlignment.h:29-33).
 *   dirs  : 2-bit backtracking codes, anti-diagonal-major (see below).
 *   ops   : per-tile regions the backtrack writes run-length ops into, then a dense
 *           arena the host downloads.
 *
 * Direction matrix layout.  The reference keeps 1 byte per corridor cell, row-major
 * (directionMatrix, src/AlignmentMatrixFast.h:261).  The device fill walks the
 * corridor by anti-diagonals r = x + y, holding read row y in ring slot s = y mod N
 * (slot s = thread * M + j).  A cell's direction (stop / outside, insertion = up,
 * deletion = left, diagonal -- EQ or X is re-derived from the sequences) is kept as two
 * bit-plane words per slot per 32 steps: for step t = r - r0
 *
 *      dword  dir_off + ((t >> 5) * N + s) * 2 + plane ,   bit 31 - (t & 31)
 *      plane 0 = "gap" (I or D), plane 1 = "consumes a read base" (I or diagonal)
 *      (1,0) deletion, (1,1) insertion, (0,1) diagonal, (0,0) stop
 *
 * The fill shifts the per-step lane masks of the recurrence into the words with one
 * add-with-carry each; a lane stores its 2*M words every 32 steps (coalesced), and the
 * backtrack finds both planes of a cell in one 8-byte load, neighbouring path cells in
 * neighbouring words.  Only the *layout* differs from the reference; the algorithmic
 * byte count used for the roofline stays 1 byte per cell.
 */
#ifndef CVX_TYPES_H
#define CVX_TYPES_H

#include <stdint.h>

namespace cvx {

struct RowDesc2 { int32_t x, y; };  /* (offset, length); same layout as HIP's int2 */
struct RowDesc { int32_t off, len; }; /* the same pair under the host's names */

/* A finished row's slot is cleared one step after its last cell (the row below still
 * reads that cell) and handed to row y+N at the next 4-step group boundary, so row
 * y+N-1 must not start earlier than 4 steps after row y ended. */
static const int kSwitchMargin = 4;
static const int kRingMax = 4096;      /* largest ring any fill kernel provides */
/* two-phase best-cell tracking of the ring fill (cvx_kernels.hip): the last
 * max(kLateMinGroups, groups >> kLateShift) four-step groups are tracked exactly; TileOut::pad == kPadRedo
 * marks a tile whose best cell may lie before them (redone by the exact instantiation) */
static const int kLateMinGroups = 128;
static const int kLateShift = 5;       /* exactly tracked tail = max(kLateMinGroups, groups >> kLateShift): a 10 kb tile tracks its last 161 groups
                                        * = 644 steps, two corridor widths (round 3: groups / 8; 109.1 -> 107.9 ms per 49 120 tiles, still no tile redone) */
static const int kPadRedo = 2;
/* fill_ring_kernel<.., TAB>: entries of the LDS penalty table, and the run from which the run registers are clamped (the
 * penalty must be constant from kPenClamp on: checked by the host per scoring; a run grows by at most four between two clamps) */
static const int kPenEntries = 64;
static const int kPenClamp = kPenEntries - 8;
static const int kGangDepth = 8;       /* fill_ring_kernel<.., G > 1>: steps of lane-boundary records a wave keeps for its successor (> G) */
static const int kChainChunk = 16;     /* chained row blocks: steps per boundary hand-off (multiple of 4, power of two, <= 64) */

struct ScoreParams {
	float mat, mis, go, ge, gem, decay;
};

/* How a tile's corridor rows travel to the device (expand_rows_kernel rebuilds the int2 rows arena):
 * every corridor the reference builds has one width for all rows and row offsets that move by a
 * few columns per row, so a row is one signed byte -- the offset's step from the row above -- instead
 * of eight; anything else (a width that changes, a step outside -128..127) goes verbatim. */
enum RowFormat { kRowsDelta8 = 0, kRowsExplicit = 1, kRowsAffine = 2, kRowsConst = 3 };
struct RowSrc {
	uint64_t src_off;      /* kRowsDelta8: byte offset of the tile's H step bytes (byte 0 unused) in the delta stream;
	                        * kRowsExplicit: index of its first RowDesc in the explicit-rows buffer */
	int32_t off0;          /* offset of row 0 (kRowsDelta8), of every row (kRowsConst) */
	int32_t width;         /* the common row length (all but kRowsExplicit) */
	int32_t fmt;
	/* kRowsAffine (cvx_tile.corridor_kind == CVX_CORRIDOR_AFFINE): offset[y] = (int) (((float) y - d) / k - right),
	 * the closed form of the reference's corridor builders (src/AlignmentBuffer.cpp:107-127, 178-191, 68-82);
	 * nothing of such a tile's rows crosses PCIe */
	float k, d, right;
};

/* The closed form, shared by the device kernel and the host's chain planning: binary32 throughout, every
 * operation rounded on its own (the files that include this are compiled with -ffp-contract=off), the
 * divide correctly rounded on both sides, C truncation. */
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int32_t affine_row_offset(int y, float d, float k, float right) {
	const float a = (float) y - d;
	const float q = a / k;
	const float r = q - right;
	return (int32_t) r;
}

#if defined(__HIPCC__)
/* The rows of one tile wherever they live (round 4): a closed-form corridor is evaluated in registers by every kernel
 * that needs a row -- plan, fill (row records), backtrack, catch-all -- so such a tile has no slice of the rows arena
 * at all (no expand_rows_kernel pass, no 8 bytes per read row in HBM); tiles whose rows travelled as arrays read their
 * slice as before.  fmt is a property of the tile, hence uniform wherever one wave works on one tile. */
struct RowView {
	const RowDesc2 *rows;  /* slice of the rows arena (kRowsDelta8 / kRowsExplicit tiles), first row of the view */
	int32_t fmt;           /* kRowsAffine / kRowsConst: closed form; anything else: rows[y] */
	int32_t width;
	int32_t y0;            /* row index of the view's row 0 inside the tile (chained row blocks) */
	float k, d, right;     /* kRowsConst: the bits of `right` are the constant offset */
};
__device__ __forceinline__ RowView row_view(const RowSrc &rs, const RowDesc2 *arena, const uint64_t row_off, const int y0 = 0) {
	RowView v;
	v.rows = arena + row_off + y0;
	v.fmt = rs.fmt; v.width = rs.width; v.y0 = y0;
	v.k = rs.k; v.d = rs.d;
	v.right = rs.fmt == kRowsConst ? __int_as_float(rs.off0) : rs.right;
	return v;
}
__device__ __forceinline__ RowDesc2 row_at(const RowView &v, const int y) {
	RowDesc2 r;
	if (v.fmt == kRowsAffine || v.fmt == kRowsConst) {
		r.x = v.fmt == kRowsConst ? __float_as_int(v.right) : affine_row_offset(y + v.y0, v.d, v.k, v.right);
		r.y = v.width;
	} else {
		r = v.rows[y];
	}
	return r;
}
#endif

struct TileIn {            /* written by the host at upload */
	uint32_t ref_off;      /* byte offset of ref[0] in the seq arena */
	uint32_t qry_off;      /* byte offset of qry[0] */
	int32_t W, H;
	uint64_t row_off;      /* index of row 0 in the rows arena (int2 units); closed-form corridors own no rows there */
	uint64_t reserved;
};

enum PlanFlags {
	kPlanIrregular = 1,    /* row start/end anti-diagonals not monotone: no ring kernel */
	kPlanEmpty = 2,        /* no cell inside [0,W) */
	kPlanTooLarge = 4,     /* exceeds maxMatrixSizeMB */
	kPlanWrap16 = 8        /* a row or column can carry a gap run past SHRT_MAX */
};

struct TilePlan {          /* written by plan_kernel, read back by the host */
	int32_t r0;            /* first anti-diagonal with a cell */
	int32_t rend;          /* one past the last */
	int32_t need;          /* ring slots needed */
	int32_t flags;
	uint64_t cells;        /* sum of row_length (reference matrixSize) */
	uint64_t active;       /* cells inside [0,W) */
};

struct TileRun {           /* written by the host after planning */
	uint64_t dir_off;      /* dword offset in the dirs arena (chained tiles: see ChainBlk) */
	uint64_t ops_off;      /* int offset of this tile's ops region */
	int32_t ring;          /* N (chained tiles: rows per block) */
	int32_t ops_cap;
	int32_t r0;            /* origin of the step index t = x + y - r0 */
	int32_t nsteps;
	int32_t skip;          /* != 0: tile not computed (status preset in TileOut) */
	int32_t mnw;           /* M of the fill kernel class that owns the tile (0: catch-all kernel) */
	int32_t chain_blk0;    /* chained tiles: index of the tile's first ChainBlk / ChainTask, else -1 */
	int32_t chain_nblk;    /* number of row blocks */
};

/* Chained tiles (cvx_kernels.hip, kFillChain): block g holds read rows [g * ring, (g + 1) * ring). */
struct ChainBlk {          /* where the backtrack finds a block's direction words */
	uint64_t dir_off;      /* uint2 offset of the block's region in the dirs arena */
	int32_t tblk0;         /* (first step of the block - TileRun::r0) >> 5 */
	int32_t nblk32;        /* 32-step word rows in the region */
};

struct ChainTask {         /* one row block = one wave's work */
	int32_t tile;
	int32_t y0;            /* first read row */
	int32_t rows;          /* rows in the block (<= ring) */
	int32_t r0;            /* first step; (r0 - TileRun::r0) is a multiple of 32 */
	int32_t nsteps;
	int32_t blk;           /* index of this block's ChainBlk / ChainOut */
	int32_t prev;          /* block index of the block above, -1 for the first block */
	int32_t bnd_lo;        /* the row above this block: first column and number of cells inside [0, W) */
	int32_t bnd_len;
	int32_t has_next;      /* a block below consumes this block's last row */
	uint64_t dir_off;      /* dword offset of the block's region */
	uint64_t bnd_in_off;   /* BoundaryRec offset of the stream written by `prev` */
	uint64_t bnd_out_off;  /* BoundaryRec offset of this block's own stream */
};

struct BoundaryRec {       /* what a cell of a block's last row offers the row below: ONE 8-byte word, stored and loaded
                            * atomically, that says by itself whether it has been written in this launch */
	uint32_t s_bits;       /* the cell's score (float bits) */
	uint32_t meta;         /* [15:0] gap-run register (float kernels: run + 1 as an integer; wrap kernels: the int16 run),
	                        * [16] the cell is an insertion, [31:17] launch epoch (kBndEpochMax values, never 0).
	                        * The up candidate V is not stored: it is a function of (score, run, insertion) */
};
static const uint32_t kBndEpochMax = 32767;

struct BoundaryVal {       /* a decoded BoundaryRec in LDS */
	float V, S;            /* up candidate, score */
	uint32_t run;          /* gap-run register (float bits, or the int16 run of the wrap kernels) */
	uint32_t is_ins;
};

struct ChainOut {          /* best cell of one block, tile coordinates */
	float score;           /* -1: no positive score in the block */
	int32_t best_x, best_y;
	int32_t failed;        /* the wave gave up waiting for its predecessor */
};

struct TileOut {
	float score;
	int32_t status;
	int32_t best_x, best_y;
	int32_t ref_position, qstart, qend;
	int32_t n_ops;
	int32_t ops_first;     /* index of the first op inside the tile's region */
	int32_t pad;           /* 0 filled, 1 backtracked, kPadRedo: needs the exact-tracking fill pass */
};

struct ResultRec {         /* same layout as cvx_result (include/cvx_align.h), written by finalize_kernel */
	float score;
	int32_t status;
	int32_t best_x, best_y;
	int32_t ref_position, qstart, qend;
	int32_t n_ops;
	uint64_t ops_begin;
	uint64_t cells;
};

struct BatchSummary {      /* follows the result records in the same buffer */
	uint64_t ops_total;    /* ops of all valid tiles */
	uint64_t dense_cap;    /* capacity the compaction ran with (ops_total > dense_cap: compact again) */
	int32_t n_valid;
	int32_t n_redone;      /* tiles that needed the exact-tracking fill pass */
	/* chained row blocks: s_memtime ticks (a constant ~100 MHz clock) the tasks of the launch ran for in total, and the part
	 * of that spent polling for a boundary record their predecessor had not written yet (cvx_timing.chain_*) */
	uint64_t chain_task_ticks, chain_poll_ticks;
};
static const int kCtrChainTicks = 16;  /* int32 index of the two uint64 tick counters inside the batch's counter block */

struct WindowDesc {        /* one reference window to decode from the resident genome (cvx_genome.hip) */
	uint64_t position;     /* first base, concatenated-genome coordinates (ngmlr's onRefStart) */
	uint64_t dst_off;      /* byte offset of the window's first character in the destination buffer */
	int64_t n_chars;       /* characters to write = the reference call's sequenceLength - 1 (its NUL is not stored) */
};

struct TextRec {           /* same layout as cvx_alignment_text (include/cvx_align.h), written by text_kernel */
	int32_t ret;
	float score;
	int32_t position_offset, qstart, qend, nm;
	float identity;
	int32_t alignment_length, cigar_op_count, sv_type;
	int32_t first_ref, first_read, last_ref, last_read;
	int32_t nm_count, cigar_len, md_len;
};

struct TextArgs {          /* device-side text stage (cvx_text.hip) */
	const uint8_t *seq;
	const TileIn *tin;
	const TileRun *trun;
	const TileOut *tout;
	const int32_t *ops;            /* per-tile op regions */
	const int32_t *ext_qstart;     /* per tile, or NULL for zeros */
	const int32_t *ext_qend;
	TextRec *recs;
	unsigned long long *text_len;  /* bytes of tile t's two strings incl. their NULs */
	unsigned long long *text_off;  /* exclusive prefix sum of text_len */
	unsigned long long *text_total;
	uint8_t *text;                 /* [cigar NUL md NUL] per tile, dense */
	int32_t n_tiles;
};

struct FillArgs {
	const uint8_t *seq;
	const RowDesc2 *rows;
	const RowSrc *rsrc;    /* per tile: which form its rows have (RowView) */
	const TileIn *tin;
	const TileRun *trun;
	TileOut *tout;
	uint32_t *dirs;
	const int32_t *list;   /* tile indices of this kernel class, largest first */
	int32_t list_n;        /* = grid size: one workgroup per tile */
	int32_t *redo_count;   /* statistics: tiles redone by the exact pass */
	/* kFillChain only */
	const ChainTask *tasks;  /* list_n tasks in dependency order */
	int32_t *chain_ticket;   /* zeroed before the launch */
	BoundaryRec *bnd;
	uint32_t bnd_epoch;      /* tag of the boundary records written by this launch */
	int32_t chain_prio;      /* != 0: chained blocks run at raised wave priority */
	ChainOut *chain_out;     /* per block */
	int32_t late_min_groups; /* exactly tracked tail, in 4-step groups (kLateMinGroups; a test knob raises it) */
	int32_t late_shift;      /* ... or groups >> late_shift of them if that is more (kLateShift) */
	int32_t pen_table;       /* != 0: the two-phase float-score launches read the convex penalty from an LDS table (TAB instantiation) */
	int32_t *ops;          /* per-tile op regions */
	ScoreParams sp;
};

struct BacktrackArgs {
	const ChainBlk *chain_blk;
	const uint8_t *seq;
	const RowDesc2 *rows;
	const RowSrc *rsrc;
	const TileIn *tin;
	const TileRun *trun;
	TileOut *tout;
	const uint32_t *dirs;
	int32_t *ops;          /* region arena */
	int32_t n_tiles;
};

}  // namespace cvx

#endif
