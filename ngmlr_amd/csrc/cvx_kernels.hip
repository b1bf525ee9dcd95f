/*
 * cvx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for ngmlr's
 * convex-gap banded Smith-Waterman.  gfx950 only: no other target, no dual paths.
 *
 *   plan_kernel       per-tile corridor analysis (AlignmentMatrixFast::prepare,
 *                     reference src/AlignmentMatrixFast.cpp:30-60, plus what the
 *                     anti-diagonal schedule needs: ring size, first/last diagonal)
 *   fill_ring_kernel  forward fill (fwdFillMatrixSSESimple == scalar recurrence of
 *                     reference src/ConvexAlignFast.cpp:606-774), anti-diagonal
 *                     wavefront, one wave per tile (per row block for chained tiles)
 *   backtrack_kernel  revBacktrack + validPath (src/ConvexAlignFast.cpp:335-432,
 *                     src/AlignmentMatrixFast.cpp:213-220)
 *   compact_ops_kernel  gathers the per-tile op regions into one dense arena
 *
 * Parallel scheme of the fill (see DESIGN.md for the derivation).  Cells on one
 * anti-diagonal r = x + y are independent: (x,y) needs left (x-1,y) and up (x,y-1)
 * from r-1 and diag (x-1,y-1) from r-2.  A wave keeps read ROWS in a ring of
 * N = 64*M slots, row y in slot y mod N, M consecutive slots per lane.  Per step
 * every slot advances its row by one column, so "left" is the slot's own previous
 * value (a register), "up" is the previous slot's value (a register for M-1 of the M
 * slots, one DPP wave_ror:1 for the lane boundary) and "diag" is the up value the
 * slot saw one step earlier.  Row state never touches LDS or HBM; the only HBM
 * traffic is one reference character per cell (L1/L2 resident) in, and the 2-bit
 * direction codes out (two bit-plane words per slot per 32 steps, 2N dwords).
 * The recurrence's priority chain runs on 64-bit lane masks in SGPRs (SALU), so the
 * VALU only sees the float adds/max/compares and a few selects per cell.
 *
 * Floating point: scores are IEEE binary32, every * and + rounded separately as in
 * the reference's scalar and SSE code (compile with -ffp-contract=off).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvx_types.h"
#include "cvx_launch.h"

namespace cvx {

#define CVX_DEV __device__ __forceinline__

/* lane i <- lane (i-1) mod 64 : DPP wave_ror:1 (GFX9 DPP_WF_RR1 = 0x13C) */
CVX_DEV int rot1_i(int v) {
	return __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, true);
}
CVX_DEV float rot1_f(float v) {
	return __int_as_float(rot1_i(__float_as_int(v)));
}

/* ------------------------------------------------------------------ plan */

/* gs(y): anti-diagonal of the first cell of row y; ge(y): one past the last.
 * Row y covers x in [max(0,off), min(off+len, W))  (src/ConvexAlignFast.cpp:948-950). */
CVX_DEV void row_span(const int2 ol, int W, int y, int &gs, int &ge) {
	long long lo = ol.x > 0 ? ol.x : 0;
	long long hi = (long long) ol.x + (long long) ol.y;
	if (hi > W) hi = W;
	if (hi < lo) hi = lo;
	gs = (int) (lo + y);
	ge = (int) (hi + y);
}

/* TPT threads per tile, 256 / TPT tiles per workgroup: 256 for batches of long reads, 64 (a wave
 * per tile) for batches of short ones, where a 256-thread group per 150-row tile was mostly idle
 * threads and workgroup launches (short-read config: 1.7 of 8.9 ms per 100 000 tiles) */
template <int TPT>
__global__ void __launch_bounds__(256)
plan_kernel(const int2 *rows, const RowSrc *rsrc, const TileIn *tin, TilePlan *plan, int n_tiles, unsigned long long max_matrix_mb) {
	constexpr int TPB = 256 / TPT;                 /* tiles per block */
	const int sub = threadIdx.x / TPT;             /* tile slot inside the block */
	const int ltid = threadIdx.x % TPT;
	const int t = blockIdx.x * TPB + sub;
	const bool live = t < n_tiles;
	const TileIn ti = tin[live ? t : 0];
	/* rows of a closed-form corridor are evaluated in registers (RowView): the analysis of such a tile reads nothing
	 * but its 32-byte description -- it used to read 8 H bytes a handful of times (3.5 ms per 49 152 PacBio tiles, and
	 * most of a fill's duration when it ran beside one) */
	const RowView rv = row_view(rsrc[live ? t : 0], reinterpret_cast<const RowDesc2 *>(rows), ti.row_off);
	auto row = [&](const int y) { const RowDesc2 q = row_at(rv, y); return make_int2(q.x, q.y); };
	const int H = live ? ti.H : 0, W = ti.W;

	__shared__ unsigned long long s_cells[TPB], s_active[TPB];
	__shared__ int s_need[TPB], s_flags[TPB], s_maxlen[TPB], s_rend[TPB], s_r0[TPB];
	if (ltid == 0) { s_cells[sub] = 0; s_active[sub] = 0; s_need[sub] = 1; s_flags[sub] = 0; s_maxlen[sub] = 0; s_rend[sub] = -0x7fffffff; s_r0[sub] = 0x7fffffff; }
	__syncthreads();

	unsigned long long cells = 0, active = 0;
	int need = 1, flags = 0, maxlen = 0, rendmax = -0x7fffffff, r0min = 0x7fffffff;
	for (int y = ltid; y < H; y += TPT) {
		const int2 ol = row(y);
		int gs, ge;
		row_span(ol, W, y, gs, ge);
		cells += (unsigned long long) (long long) ol.y;
		active += (unsigned long long) (ge - gs);
		if (ol.y > maxlen) maxlen = ol.y;
		if (ge > rendmax) rendmax = ge;
		if (gs < r0min) r0min = gs;
		if (y > 0) {
			int pgs, pge;
			row_span(row(y - 1), W, y - 1, pgs, pge);
			if (gs <= pgs) flags |= kPlanIrregular;  /* ring schedule needs increasing row starts */
			/* ... and rows that end in order: the fill hands slots over in row order (its staged row
			 * records are overwritten on that assumption).  True for every corridor the reference builds
			 * (one width, offsets that never decrease); a corridor whose rows shrink goes to the catch-all kernel. */
			if (ge < pge) flags |= kPlanIrregular;
		}
		/* first row y' > y that starts at or after ge + margin (gs is increasing): gallop out from
		 * a guess -- row starts advance by about two anti-diagonals per row in a sloped corridor --
		 * then bisect; a handful of row reads instead of log2(H) */
		const int lim = ge + kSwitchMargin;
		auto starts_before = [&](int yy) {       /* gs(yy) < lim */
			int mgs, mge;
			row_span(row(yy), W, yy, mgs, mge);
			return mgs < lim;
		};
		int lo = y + 1, hi = H;                  /* rows < lo start before lim, rows >= hi do not */
		int g = y + 1 + ((lim - gs) >> 1);
		g = g < lo ? lo : g;
		if (g < hi) {
			if (starts_before(g)) {
				lo = g + 1;
				for (int step = 1; lo < hi; step <<= 1) {
					const int p = (lo + step - 1 < hi) ? lo + step - 1 : hi - 1;
					if (starts_before(p)) lo = p + 1; else { hi = p; break; }
				}
			} else {
				hi = g;
				for (int step = 1; lo < hi; step <<= 1) {
					const int p = (hi - step > lo) ? hi - step : lo;
					if (starts_before(p)) { lo = p + 1; break; } else hi = p;
				}
			}
		}
		while (lo < hi) {
			const int mid = (lo + hi) >> 1;
			if (starts_before(mid)) lo = mid + 1; else hi = mid;
		}
		const int n = lo - y + 1;
		if (n > need) need = n;
		if (H > 32767) {
			/* How long can a gap run get in this corridor?  A deletion run stays inside one row (<= its length, tracked
			 * above); an insertion run stays inside one column, i.e. inside the consecutive rows that contain it: for the
			 * last column of row y those are the rows up to the first one that starts at or behind hi(y) (row starts do not
			 * decrease in a regular corridor).  Only when such a stretch exceeds SHRT_MAX can the reference's `short
			 * indelRun` wrap (src/AlignmentMatrixFast.h:43) and the int16-emulating kernels are needed -- a 100 kb read on
			 * a 350-column corridor never gets there, and the float-run kernels are three times as fast. */
			long long lo_y = ol.x > 0 ? ol.x : 0;
			long long hi_y = (long long) ol.x + (long long) ol.y;
			if (hi_y > W) hi_y = W;
			if (hi_y < lo_y) hi_y = lo_y;
			int a = y + 1, b = H;                    /* rows < a start before hi(y), rows >= b do not */
			while (a < b) {
				const int mid = (a + b) >> 1;
				const int2 om = row(mid);
				const long long lo_m = om.x > 0 ? om.x : 0;
				if (lo_m < hi_y) a = mid + 1; else b = mid;
			}
			if (a - y > maxlen) maxlen = a - y;      /* folded into the same maximum: either kind of run past 32767 needs the wrap kernels */
		}
	}
	if (H > 0) {
		atomicAdd(&s_cells[sub], cells);
		atomicAdd(&s_active[sub], active);
		atomicMax(&s_need[sub], need);
		atomicOr(&s_flags[sub], flags);
		atomicMax(&s_maxlen[sub], maxlen);
		atomicMax(&s_rend[sub], rendmax);
		atomicMin(&s_r0[sub], r0min);
	}
	__syncthreads();

	if (ltid == 0 && live) {
		TilePlan p;
		p.cells = s_cells[sub];
		p.active = s_active[sub];
		p.need = s_need[sub];
		int f = s_flags[sub];
		int r0 = 0, rend = 0;
		if (H > 0) { r0 = s_r0[sub]; rend = s_rend[sub]; }   /* first / one-past-last anti-diagonal with a cell */
		if (H <= 0 || s_active[sub] == 0) f |= kPlanEmpty;
		/* src/AlignmentMatrixFast.cpp:45: (ulong)(matrixSize / 1000.0f / 1000.0f) < maxMatrixSizeMB */
		const float mb = (float) s_cells[sub] / 1000.0f / 1000.0f;
		if (!((unsigned long long) mb < max_matrix_mb)) f |= kPlanTooLarge;
		/* longest possible deletion (row length) or insertion (column extent, rows above) run; the column bound needs
		 * row starts that do not decrease, so an irregular corridor that tall keeps the old rule */
		if (s_maxlen[sub] > 32767 || (H > 32767 && (f & kPlanIrregular))) f |= kPlanWrap16;
		p.r0 = r0;
		p.rend = rend;
		p.flags = f;
		plan[t] = p;
	}
}

/* ------------------------------------------------------------------ rows */

/* Rebuilds the (offset, length) rows arena from what travelled over PCIe (RowSrc, cvx_types.h): one
 * signed step byte per row -- a running sum per tile: wave scans plus a carry, 256 rows per pass -- or
 * the rows verbatim for the tiles that do not fit that form.  The result is exactly the caller's
 * CorridorLine[] minus offsetInMatrix; everything downstream reads this arena as before. */
__global__ void __launch_bounds__(64)
expand_rows_kernel(const RowSrc *rsrc, const TileIn *tin, const uint8_t *delta, const int2 *rowsx, int2 *rows, int n_tiles, int closed_forms) {
	/* one wave per tile, 64 rows per pass, the running offset in a register: no LDS, no barrier -- the
	 * kernel runs on the upload stream beside the previous batch's fill, where a workgroup barrier per
	 * 256 rows cost it tens of ms (measured: 46 ms average in the pipelined bench, 1.6 ms alone) */
	const int t = blockIdx.x;
	if (t >= n_tiles) return;
	const int lane = threadIdx.x;
	const RowSrc rs = rsrc[t];
	const TileIn ti = tin[t];
	const int H = ti.H;
	int2 *out = rows + ti.row_off;
	if (rs.fmt == kRowsExplicit) {
		const int2 *src = rowsx + rs.src_off;
		for (int y = lane; y < H; y += 64) out[y] = src[y];
		return;
	}
	if ((rs.fmt == kRowsAffine || rs.fmt == kRowsConst) && !closed_forms) return;      /* evaluated in registers wherever a row is needed (RowView) */
	if (rs.fmt == kRowsAffine) {
		/* the reference's corridor builders in closed form (cvx_types.h affine_row_offset; src/AlignmentBuffer.cpp:107-127,
		 * 178-191, 68-82): binary32 subtract, correctly rounded divide, subtract, truncation -- row by row, no carried state */
		for (int y = lane; y < H; y += 64) out[y] = make_int2(affine_row_offset(y, rs.d, rs.k, rs.right), rs.width);
		return;
	}
	if (rs.fmt == kRowsConst) {
		for (int y = lane; y < H; y += 64) out[y] = make_int2(rs.off0, rs.width);
		return;
	}
	const int8_t *d = reinterpret_cast<const int8_t *>(delta + rs.src_off);
	int carry = rs.off0;
	for (int y0 = 0; y0 < H; y0 += 64) {
		const int y = y0 + lane;
		int v = (y < H && y > 0) ? (int) d[y] : 0;
#pragma unroll
		for (int k = 1; k < 64; k <<= 1) {
			const int u = __shfl_up(v, k, 64);
			if (lane >= k) v += u;
		}
		if (y < H) out[y] = make_int2(carry + v, rs.width);
		carry += __builtin_amdgcn_readlane(v, 63);
	}
}

/* ------------------------------------------------------------------ fill */

typedef unsigned long long u64;

/* per-lane predicate <-> wave-uniform 64-bit lane mask (SGPR pair) */
CVX_DEV u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
/* A wave-uniform value that is needed in a rare path of a long loop (row staging, the direction flush): handed through an
 * "s" constraint at its use, so that it lives in a scalar register across the loop instead of in a vector register the
 * allocator then spills to scratch and reloads inside the loop (round 5: 26 spilled dwords, +16 GB of HBM traffic per launch) */
CVX_DEV int in_sgpr(int v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); return v; }
CVX_DEV float in_sgpr(float v) { return __int_as_float(in_sgpr(__float_as_int(v))); }
CVX_DEV unsigned in_sgpr(unsigned v) { return (unsigned) in_sgpr((int) v); }
CVX_DEV u64 in_sgpr(u64 u) { return ((u64) in_sgpr((unsigned) (u >> 32)) << 32) | (u64) in_sgpr((unsigned) u); }
template <typename T> CVX_DEV T *in_sgpr(T *p) {
	const u64 u = (u64) p;
	const unsigned lo = (unsigned) in_sgpr((int) (unsigned) u), hi = (unsigned) in_sgpr((int) (unsigned) (u >> 32));
	return (T *) (((u64) hi << 32) | (u64) lo);
}
CVX_DEV bool lanes(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
/* mask of lane i <- mask of lane (i-1) mod 64 : the SALU twin of wave_ror:1 */
CVX_DEV u64 rot1_m(u64 m) { return (m << 1) | (m >> 63); }
/* acc = (acc << 1) | (lane's bit of m): one v_addc_co_u32 with the mask as carry-in.
 * m must come from SALU mask logic (it does: planes are s_or_b64 results). */
#ifndef CVX_FILL_ADDC_SGPR
#define CVX_FILL_ADDC_SGPR 1
#endif
CVX_DEV unsigned shl1_in(unsigned acc, u64 m) {
#if CVX_FILL_ADDC_SGPR
	/* the (unused) carry-out goes to an SGPR pair of the compiler's choice, not to VCC: in that form the
	 * instruction issues beside a full-rate VALU op like any other half-rate one; with VCC as its
	 * destination it does not (profiles/r03_ubench_pipes.txt, kinds 15-18) */
	u64 carry_out;
	asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(acc), "=s"(carry_out) : "s"(m));
#else
	asm volatile("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(acc) : "s"(m) : "vcc");
#endif
	return acc;
}

/* ------------------------------------------------------------------ backtrack */

/* validPath, src/AlignmentMatrixFast.cpp:213-220: float arithmetic, int truncation,
 * no contraction.  A cell (x, row) is valid iff minC < x < maxC. */
CVX_DEV void valid_bounds(const int off, const int width, int &minC, int &maxC) {
	minC = (int) ((float) off + 0.1f * (float) width);
	maxC = (int) ((float) (minC + width) - 0.1f * (float) width);
}

/* direction code of bit `bit` of a plane pair: plane 0 = gap (I or D), plane 1 = the cell
 * consumes a read base on the way back (I or diagonal) */
CVX_DEV unsigned plane_code(const unsigned wx, const unsigned wy, const int bit) {
	const unsigned px = (wx >> bit) & 1u, py = (wy >> bit) & 1u;
	return px ? (py ? 1u : 2u) : (py ? 3u : 0u);
}

/*
 * revBacktrack (src/ConvexAlignFast.cpp:335-432) by ONE wave, run-skipping.  The path is
 * a chain of runs (diagonal runs broken by short gaps); instead of one dependent load per
 * cell, the 64 lanes probe the next 64 cells along the current direction at once (lane i
 * looks at the i-th cell back), a ballot finds how far the run goes, and the walk jumps to
 * its end.  Every probe is four coalesced loads (corridor rows, plane words, both sequences).
 * The gap that ends a diagonal run is then resolved from the SAME probe whenever the words
 * already in registers cover it (a deletion run lies in one lane's word, an insertion run in
 * the words of the following lanes), so a 10-kb PacBio tile costs ~1 500 probes instead of
 * ~20 000 dependent steps.  The walk state is wave-uniform; lane 0 writes the run-length ops.
 * `o` carries the argmax in (score, best_x, best_y) and returns the FwdResults.
 */
template <bool CHAINED>
CVX_DEV void backtrack_walk(const int lane, const int H, const int N, const int r0, const int ops_cap,
		const RowView &rows, const uint2 *dirs, const ChainBlk *blk, const uint8_t *ref, const uint8_t *qry, int *ops, TileOut &o) {
	/* make the walk's state provably wave-uniform so that it runs on scalar branches */
	const int best_x = __builtin_amdgcn_readfirstlane(o.best_x);
	const int best_y = __builtin_amdgcn_readfirstlane(o.best_y);
	/* src/ConvexAlignFast.cpp:338 */
	if (best_y <= 0) { o.status = 1; return; }

	const int qend = (H - best_y) - 1;
	int idx = ops_cap - 1;
	int elem = 4;          /* CIGAR_S: the trailing clip is tracked but not stored */
	int elem_len = qend;
	int consumed = qend;
	int x = best_x, y = best_y;
	int s = y % N;         /* ring slot of row y */
	int status = 0;
	unsigned want = 3u;    /* direction of the run being followed (first probe: a guess) */

	/* revBacktrack's run-length bookkeeping (src/ConvexAlignFast.cpp:395-403) */
	auto emit = [&](int cur, int n) {
		if (n <= 0) return;
		if (cur == elem) {
			elem_len += n;
		} else {
			if (elem != 4) { if (lane == 0) ops[idx] = (elem_len << 4) | elem; idx -= 1; }
			elem = cur;
			elem_len = n;
		}
	};

	/* every probe either consumes a cell or fixes the direction, so x + y + 2 probes always
	 * suffice; the cap turns corrupted direction data into an invalid tile, never a hang */
	int budget = 2 * (best_x + best_y) + 8;
	for (;;) {
		if (--budget < 0) { status = 3; break; }
		const int dx = (want != 1u) ? 1 : 0, dy = (want != 2u) ? 1 : 0;
		const int cx = x - lane * dx, cy = y - lane * dy;
		const bool inside = (cx >= 0 && cy >= 0);
		const int lx = cx > 0 ? cx : 0, ly = cy > 0 ? cy : 0;
		int sl = s - lane * dy;
		if (sl < 0) sl += N;
		const int tt = cx + cy - r0;
		const int ttc = tt > 0 ? tt : 0;
		const RowDesc2 ol = row_at(rows, ly);
		uint2 w;
		if (CHAINED) {
			/* row block of the probed cell -> its own region of direction words (N is a power of two here) */
			const int gb = ly / N;
			const ChainBlk cb = blk[gb];
			int wr = (ttc >> 5) - cb.tblk0;
			wr = wr < 0 ? 0 : (wr >= cb.nblk32 ? cb.nblk32 - 1 : wr);      /* cells outside their row are masked below */
			w = dirs[cb.dir_off + (size_t) wr * N + (size_t) (ly - gb * N)];
		} else {
			w = dirs[(size_t) (ttc >> 5) * N + sl];
		}
		const int rc = ref[lx], qc = qry[ly];
		/* used by every probe on purpose: keeps all four loads in one round trip */
		const u64 eqm = ballot(rc == qc);
		/* getDirection, src/AlignmentMatrixFast.cpp:185-195: outside -> STOP */
		const bool in_row = inside && tt >= 0 && cx >= ol.x && cx < ol.x + ol.y;
		unsigned code = plane_code(w.x, w.y, 31 - (ttc & 31));
		if (!in_row) code = 0u;
		int minC, maxC;
		valid_bounds(ol.x, ol.y, minC, maxC);

		const u64 run = ballot(code == want);
		const int L = (~run == 0ull) ? 64 : __builtin_ctzll(~run);   /* cells of this run */
		const u64 low = (L == 64) ? ~0ull : ((1ull << L) - 1ull);
		/* every visited cell must pass validPath before the move (:368-373) */
		const u64 vp = ballot(cx > minC && cx < maxC);
		if ((~vp & low) != 0ull) { status = 2; break; }

		if (want == 3u) {
			int pos = 0;
			while (pos < L) {
				const int isq = (int) ((eqm >> pos) & 1ull);
				const u64 m = (isq ? ~eqm : eqm) >> pos;
				int rl = (m == 0ull) ? 64 - pos : __builtin_ctzll(m);
				if (rl > L - pos) rl = L - pos;
				emit(isq ? 7 : 8, rl);
				pos += rl;
			}
			x -= L; y -= L; consumed += L;
		} else if (want == 1u) {
			emit(1, L);
			y -= L; consumed += L;
		} else {
			emit(2, L);
			x -= L;
		}
		if (want != 2u) { s -= L; if (s < 0) s += N; }
		if (L == 64) continue;
		const unsigned nxt = (unsigned) __builtin_amdgcn_readlane((int) code, L);
		if (nxt == 0u) break;     /* CIGAR_STOP (or outside the matrix) */
		if (want != 3u) { want = nxt; continue; }

		/* A diagonal run ended in a gap at (x, y) = lane L's cell.  Follow the gap run inside the
		 * words this probe already holds; whatever they do not cover goes to the next probe. */
		const int tp = x + y - r0;                       /* step index of (x, y), >= 0 inside a row */
		if (nxt == 2u) {
			/* deletion run: same row, earlier steps = higher bits of lane L's word */
			const unsigned wx = (unsigned) __builtin_amdgcn_readlane((int) w.x, L);
			const unsigned wy = (unsigned) __builtin_amdgcn_readlane((int) w.y, L);
			const int row_lo = max(__builtin_amdgcn_readlane(ol.x, L), 0);
			const int b0 = 31 - (tp & 31);
			const unsigned dm = (wx & ~wy) >> b0;        /* bit k: cell (x - k, y) is D; bit 0 is set */
			int Ld = (~dm == 0u) ? 32 : __builtin_ctz(~dm);   /* <= 32 - b0: the shift filled in zeros */
			const int in_word = 32 - b0;
			const int in_rowc = x - row_lo + 1;          /* cells down to the start of the row */
			if (Ld > in_rowc) Ld = in_rowc;
			const int rminC = __builtin_amdgcn_readlane(minC, L), rmaxC = __builtin_amdgcn_readlane(maxC, L);
			if (!(x - (Ld - 1) > rminC && x < rmaxC)) { status = 2; break; }
			emit(2, Ld);
			x -= Ld;
			if (Ld == in_rowc) break;                    /* next cell is left of the row: STOP */
			if (Ld == in_word) { want = 2u; continue; }  /* the run may go on in the previous word */
			const unsigned c2 = plane_code(wx, wy, b0 + Ld);
			if (c2 == 0u) break;
			want = c2;
		} else {
			/* insertion run: same column, cell k is in lane L+k's row and slot; it is step tp - k,
			 * which that lane's word covers unless a 32-step boundary lies in between */
			const int k = lane - L;
			const int tn = tp - k;
			const bool have = (k >= 0) && (cy >= 0) && (tn >= 0) && ((tn >> 5) == (ttc >> 5));
			const bool col_in = (x >= ol.x) && (x < ol.x + ol.y);
			unsigned c2 = plane_code(w.x, w.y, 31 - (tn & 31));
			if (!col_in) c2 = 0u;
			const u64 im = ballot(have && c2 == 1u) >> L;       /* bit k: cell (x, y - k) is I; bit 0 is set */
			const int Li = (~im == 0ull) ? 64 : __builtin_ctzll(~im);   /* <= 64 - L */
			const u64 ilow = (Li == 64) ? ~0ull : ((1ull << Li) - 1ull);
			const u64 vpi = ballot(x > minC && x < maxC) >> L;
			if ((~vpi & ilow) != 0ull) { status = 2; break; }
			emit(1, Li);
			y -= Li; consumed += Li;
			s -= Li; if (s < 0) s += N;
			const int e = L + Li;
			want = 1u;                                   /* default: let the next probe look again */
			if (e < 64 && ((ballot(have) >> e) & 1ull) != 0ull) {
				const unsigned c3 = (unsigned) __builtin_amdgcn_readlane((int) c2, e);
				if (c3 == 0u) break;
				want = c3;
			}
		}
	}
	if (status == 0) {
		if (elem != 4) { if (lane == 0) ops[idx] = (elem_len << 4) | elem; idx -= 1; }
		consumed += (y + 1);
		o.ref_position = x + 1;
		o.qstart = y + 1;
		o.qend = qend;
		o.ops_first = idx + 1;
		o.n_ops = ops_cap - 1 - idx;
		if (H != consumed) status = 3;
	}
	o.status = status;
}

/* first cell of the tile in (y, x) order: *fy = -1 when no row has a cell inside [0, W).
 * Rare path (a tile without any positive score), run by backtrack_kernel. */
CVX_DEV void first_cell(const RowView &rows, int H, int W, int lane, int *fy, int *fx) {
	*fy = -1;
	*fx = 0;
	for (int y0 = 0; y0 < H; y0 += 64) {
		const int yy = y0 + lane;
		bool has = false;
		int lo_i = 0;
		if (yy < H) {
			const RowDesc2 ol = row_at(rows, yy);
			long long lo = ol.x > 0 ? ol.x : 0;
			long long hi = (long long) ol.x + (long long) ol.y;
			if (hi > W) hi = W;
			has = hi > lo;
			lo_i = (int) lo;
		}
		const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
		if (m != 0ull) {
			const int l = __builtin_ctzll(m);
			*fy = y0 + l;
			*fx = __builtin_amdgcn_readlane(lo_i, l);
			return;
		}
	}
}


template <bool WRAP> struct RunT { typedef float type; };
template <> struct RunT<true> { typedef int type; };
template <bool B> struct BoolTag { static constexpr bool value = B; };

/* Occupancy target.  The M = 3 kernel (corridors of 310-370 columns, i.e. almost every
 * PacBio/ONT tile) is measured ~14 % faster at 6 waves/SIMD (80 VGPRs, the few spilled values are
 * tile constants outside the step loop) than at the 5 the allocator picks by itself; the other
 * classes keep the default.  CVX_FILL_WAVES_PER_EU overrides for A/B runs. */
#ifndef CVX_FILL_WAVES_PER_EU
#define CVX_FILL_WAVES_PER_EU 6
#endif
#ifndef CVX_FILL_WAVES_M4
#define CVX_FILL_WAVES_M4 1
#endif
#ifndef CVX_FILL_WAVES_TAB
#define CVX_FILL_WAVES_TAB 7
#endif
/* (round 5: the two-phase M = 3 instantiation with the penalty table -- the PacBio launch -- fits seven waves per SIMD: 72 VGPRs,
 * 4 KB of LDS per wave; 104.95 -> 101.7 ms per 49 108 tiles.  The exact and the gang instantiations keep six.) */
#define CVX_FILL_OCC(M, SEVEN) __attribute__((amdgpu_waves_per_eu( \
		(M) == 3 ? ((SEVEN) ? CVX_FILL_WAVES_TAB : CVX_FILL_WAVES_PER_EU) : ((M) == 4 ? CVX_FILL_WAVES_M4 : 1), \
		(M) == 3 ? ((SEVEN) ? CVX_FILL_WAVES_TAB : CVX_FILL_WAVES_PER_EU) : ((M) == 4 && CVX_FILL_WAVES_M4 > 1 ? CVX_FILL_WAVES_M4 : 8))))

/* TAB instantiation: 1 = the penalty read of a cell is consumed one step later, where the cell's offers to its two
 * consumers (V, Hc) are first needed -- the LDS round trip then has most of a step to come back; 0 = consumed at once */
#ifndef CVX_FILL_TAB_LAZY
#define CVX_FILL_TAB_LAZY 1
#endif

/* gang: s_sleep argument of a wave that waits for its neighbour's record (x 64 clocks; 0 = spin) */
#ifndef CVX_GANG_SLEEP
#define CVX_GANG_SLEEP 1
#endif

/* instruction-order experiments on the cell update (0: leave it to the compiler) */
#ifndef CVX_FILL_SCHED
#define CVX_FILL_SCHED 0
#endif

#ifndef CVX_FILL_PRIO
#define CVX_FILL_PRIO 1
#endif

/* What a cell that extends a gap offers: src/ConvexAlignFast.cpp:669-675, E = (score == 0) ? 0 : score + pen with
 * pen = min(gap_ext_min, gap_ext + run * decay).  score >= 0 and pen < 0, so this is max(score + pen, score * -2^100):
 * -0 for score 0 (compares equal to the reference's +0 and never reaches an output), score + pen otherwise.  One
 * function for the cell update and for the consumer of a chained block's boundary records, which rebuilds the up
 * candidate from (score, run) with exactly these operations. */
CVX_DEV float gap_extend_value(const float sc, const float runf, const float gem, const float gext, const float decay) {
	const float pen = fminf(gem, gext + runf * decay);
	return fmaxf(sc + pen, sc * -0x1p100f);
}

enum FillMode { kFillTwoPhase = 0, kFillExact = 1, kFillChain = 2 };

/*
 * One wave per tile: block b takes tile list[b].  The list is in LPT order and the hardware
 * dispatches workgroups in index order as wave slots free up, which is the work queue a
 * persistent kernel would build by hand -- without the atomic cursor and without a tile loop
 * around the step loop.
 *
 * Best-cell tracking (src/ConvexAlignFast.cpp:758-763: first strict maximum in (y, x) order)
 * is two-phase.  Exact (score, step, row) tracking costs three half-rate VALU ops per cell; the
 * best cell of an alignment that reaches the end of the read lies in the last few anti-
 * diagonals, so the kFillTwoPhase instantiation only keeps a per-lane running maximum
 * (2 ops per M cells) up to the last `late` groups and tracks exactly from there on.  The
 * late result is the tile's answer iff it strictly beats every earlier score; otherwise the
 * tile is flagged (TileOut::pad = kPadRedo) and the kFillExact instantiation, launched right
 * behind on the same stream over the same list, redoes just the flagged tiles with exact
 * tracking from the first step.
 *
 * kFillChain: corridors with more live rows than the widest ring (need > 256: the retry loop's
 * widened corridors up to 8192 columns, full-matrix inversion tiles) are cut into blocks of N
 * consecutive read rows; block g is an ordinary ring tile whose first row takes its "up" inputs
 * from the boundary stream that block g-1 writes while it computes its last row.  Blocks of one
 * tile run concurrently on different CUs, each a few hundred steps behind its predecessor
 * (about need / N of them at a time), so a wide tile is spread over many CUs instead of living
 * in one workgroup.  A wave takes the next task from a ticket counter (tasks are listed in
 * dependency order, so the producer of anything a wave waits for is always running or done);
 * the boundary is a stream of self-validating 8-byte records (BoundaryRec: score, run, insertion bit,
 * launch epoch) written with device-scope atomic stores as the last row advances and read kChainChunk
 * steps ahead of their use -- no counter, no release fence in the producer, no round trip to memory on
 * the consumer's critical path while the producer is ahead (it starts 2 N anti-diagonals earlier).
 */
template <int M, bool WRAP, int MODE, bool TAB = false, int G = 1>
__global__ void __launch_bounds__(64 * G) CVX_FILL_OCC(M, TAB && G == 1)
fill_ring_kernel(const FillArgs a) {
	/* G > 1: a GANG of G waves shares one ring of N = 64 M G slots -- wave w holds the slots [64 M w, 64 M (w + 1)), i.e.
	 * every G-th stretch of 64 M consecutive read rows.  Inside a wave nothing changes; the lane boundary between the last
	 * lane of wave w and the first lane of wave w + 1 (and from the last wave back to the first: the ring) goes through one
	 * self-validating 8-byte record per step in LDS instead of the DPP rotate.  The waves of a gang are never more than
	 * G - 1 steps apart (each needs its predecessor's record of the step before), so they run in lock step on their own
	 * SIMDs: a corridor with 257-576 live rows -- the retry loop's doubled corridors, src/AlignmentBuffer.cpp:291-294 -- is
	 * a whole tile on a ring again (M = 3 per wave: the cheapest cell update there is) instead of 64-row blocks chained
	 * through L2 at 1.6 x the instructions per cell. */
	constexpr int N = 64 * M * G;
	constexpr int NW = 64 * M;             /* slots of one wave */
	constexpr bool EXACT = (MODE != kFillTwoPhase);
	constexpr bool CHAIN = (MODE == kFillChain);
	constexpr bool GANG = G > 1;
	static_assert(!GANG || (!WRAP && !CHAIN), "gangs serve whole tiles with float runs");
	static_assert(!TAB || (!WRAP && MODE == kFillTwoPhase), "the penalty table serves the two-phase float-score instantiation only");
	/* gap run: float (exact small ints), int16-emulating int, or (TAB) the byte address 4 * run of the run's penalty in s_pen */
	typedef typename RunT<WRAP || TAB>::type run_t;
	const int tid = threadIdx.x;           /* = ring slot / M of the thread's first slot */
	const int lane = GANG ? (tid & 63) : tid;
	const int wv = GANG ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;      /* wave inside the gang */
	const float go = a.sp.go;
	const float gext = a.sp.ge, gem = a.sp.gem, decay = a.sp.decay;
	/* keep match / mismatch in VGPRs: v_cndmask cannot take two SGPR values plus a mask */
	float vmat = a.sp.mat, vmis = a.sp.mis;
	asm volatile("" : "+v"(vmat), "+v"(vmis));

	/* The row a slot takes over next is a 16-byte record in LDS, written 16*M rows at a time by the whole
	 * wave long before the hand-over (stage_rows below): the hand-over itself is one ds_read_b128 and two
	 * adds for the few lanes whose row just ended, instead of two dependent trips to HBM (corridor row, then
	 * reference characters) and ~25 VALU instructions executed by the whole wave for one or two lanes --
	 * round 2's counters had the wave parked on s_waitcnt for a quarter of its time, most of it here.
	 * The row of a slot's best cell (changes only at a hand-over) lives in LDS too, to keep VGPRs for occupancy. */
	__shared__ int4 s_rec[CHAIN ? 1 : M * G][64];      /* (gang: wave w's records at [w * M + ...]) */
	__shared__ int s_besty[M][64 * G];
	/* gang: what the last slot of wave w offers the first slot of wave w + 1, one record per step in a ring of kGangDepth
	 * (BoundaryRec layout: score bits | run16 | insertion bit | 15-bit step tag), and the waves' partial results at the end */
	__shared__ u64 s_gx[GANG ? G : 1][GANG ? kGangDepth : 1];
	__shared__ float s_gred[GANG ? G : 1][4];
	__shared__ int s_gfail[GANG ? G : 1];
	__shared__ BoundaryVal s_bnd[CHAIN ? kChainChunk : 1];      /* the predecessor's boundary records of the current chunk of steps */
	/* TAB: the convex penalty min(gem, gext + run * decay) (src/ConvexAlignFast.cpp:672-674) takes 28 distinct values under
	 * every preset; entry `run` of this table holds it, computed once per wave with the very operations the arithmetic form
	 * uses (binary32 multiply, add, min, each rounded on its own).  The run register of a slot is then the entry's byte
	 * address and the cell update reads its penalty with one ds_read_b32 -- the LDS pipe is otherwise idle in the step
	 * loop -- instead of v_mul + v_add + v_min.  The penalty is constant from some run on (27 with the default scoring:
	 * gext + run * decay has reached gem); the host enables this form only when that run is below kPenClamp
	 * (FillArgs::pen_table), and the run registers are clamped to kPenClamp at every group end -- a run register only ever
	 * selects a penalty, so the clamp changes nothing, and the table needs kPenClamp + 5 entries whatever the corridor
	 * (gap runs through zero-score cells are as long as a row is wide). */
	__shared__ float s_pen[TAB ? kPenEntries : 1];

	int t;                          /* tile */
	int task_id = 0, y0 = 0;        /* chain: task index, first read row of the block */
	u64 chain_t0 = 0ull, chain_polled = 0ull;      /* chain: s_memtime at the task's start; ticks spent polling for boundary records */
	ChainTask ct;
	if (CHAIN) {
		int tk = 0;
		if (lane == 0) tk = atomicAdd(a.chain_ticket, 1);
		task_id = __builtin_amdgcn_readfirstlane(tk);
		if (task_id >= a.list_n) return;
		ct = a.tasks[task_id];
		t = ct.tile;
		y0 = ct.y0;
		/* A chained tile is a dependency chain through all of its blocks and, beside thousands of whole tiles (ONT
		 * mix: 10 % retries at twice the width among 54 000 short tiles; C5: the corridors widened to 2 048 / 8 192
		 * columns), the long pole of the launch: its waves go first on their SIMDs and the whole-tile classes fill in
		 * (ONT, 60 000 tiles: 7 660-7 770 -> 8 320-8 450 Gbp/h; C5 mix, 2 048 tiles: 2 860-2 930 -> 3 140).  The host
		 * can switch it off (CVX_TUNE_CHAIN_PRIO=0). */
		if (a.chain_prio) __builtin_amdgcn_s_setprio(3);
		chain_t0 = __builtin_amdgcn_s_memtime();
	} else {
		t = a.list[blockIdx.x];
		/* the widest ring class of a batch of several (FillArgs::chain_prio, set by the host): its waves pay the most per step, its
		 * tiles are the launch's long pole beside the narrower classes' -- one notch above those */
		if (!GANG && a.chain_prio) { if (a.chain_prio >= 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#if CVX_FILL_PRIO
		/* The list is longest-first and a tile is a serial chain of steps: in a batch of uneven tiles (ONT mix: median
		 * 1.3 kb, up to 20 kb) the launch lasts as long as its longest tiles take at a sixth of a SIMD.  The first
		 * sixteenth of the list runs at raised wave priority: those waves proceed at nearly a whole SIMD's pace, the short
		 * tiles fill in behind them.  (A batch of equal tiles -- the PacBio bench -- is unaffected.) */
		if (blockIdx.x < (unsigned) (a.list_n >> 4)) __builtin_amdgcn_s_setprio(2);
#endif
		/* a gang's tiles are the widest corridors of the batch -- the retry loop's second and third attempts, twice and three
		 * times the steps per row -- and, like the chained blocks, the long pole of a mixed launch: raised priority for all of them
		 * (FillArgs::chain_prio; CVX_TUNE_GANG_PRIO=0 switches it off) */
		if (GANG && a.chain_prio) __builtin_amdgcn_s_setprio(2);
		if (MODE == kFillExact) {
			if (a.tout[t].pad != kPadRedo) return;   /* block-uniform */
			if (tid == 0) atomicAdd(a.redo_count, 1);
		}
	}
	/* The tile's constants are wave-uniform, but the records they come from are fetched with vector loads (nothing tells the
	 * compiler the tables are not written meanwhile), and what is derived from them -- 64-bit addresses above all -- then sits in
	 * vector registers for the whole step loop, to be spilled and reloaded in its rare paths (round 5: 26 dwords, +16 GB of
	 * scratch traffic per launch).  Through readfirstlane once, here, they are scalar for good. */
	TileIn ti = a.tin[t];
	ti.ref_off = in_sgpr(ti.ref_off); ti.qry_off = in_sgpr(ti.qry_off); ti.W = in_sgpr(ti.W); ti.H = in_sgpr(ti.H);
	ti.row_off = in_sgpr((u64) ti.row_off);
	TileRun tr = a.trun[t];
	tr.dir_off = in_sgpr((u64) tr.dir_off); tr.r0 = in_sgpr(tr.r0); tr.nsteps = in_sgpr(tr.nsteps);
	RowView rv = row_view(a.rsrc[t], a.rows, ti.row_off, y0);      /* wave-uniform: closed forms are evaluated in make_rec */
	rv.rows = in_sgpr(rv.rows); rv.fmt = in_sgpr(rv.fmt); rv.width = in_sgpr(rv.width);
	rv.k = in_sgpr(rv.k); rv.d = in_sgpr(rv.d); rv.right = in_sgpr(rv.right);
	const uint8_t *seq = a.seq;
	const int H = CHAIN ? ct.rows : ti.H, W = ti.W;     /* rows of this task */
	const unsigned qry_off = ti.qry_off + (unsigned) y0;
	const int r0 = CHAIN ? ct.r0 : tr.r0;
	const int nsteps = CHAIN ? ct.nsteps : tr.nsteps;
	uint32_t *dirs = in_sgpr(a.dirs + (CHAIN ? ct.dir_off : tr.dir_off));
	/* per-slot state in VGPRs (static indexing only).  A slot that is not inside its
	 * row's range holds the reference's empty element (score 0, run 0, STOP:
	 * src/AlignmentMatrixFast.h:49-53), i.e. S = 0, runs = 0, V = Hc = gap_open;
	 * the update below produces exactly that for inactive lanes by itself. */
	float S[M];        /* score of the slot's latest cell                            */
	float Hc[M];       /* left candidate that cell offers to the next column         */
	float V[M];        /* up candidate it offers to the next row                     */
	float dg[M];       /* diagonal score for the slot's next cell                    */
	run_t drun[M];     /* deletion run of the latest cell: the run itself (0 unless D) in the int16   */
	run_t irun[M];     /* kernels, run + 1 in the float ones (read only through mD / mI); same for I  */
	int cnt[M];        /* next column index inside the row (negative: not started)   */
	int len[M];        /* row length after clipping to [0,W)                         */
	int qch[M];        /* read character of the row                                  */
	unsigned xa[M];    /* seq-arena offset of the next reference dword to prefetch   */
	unsigned cwn[M];   /* reference characters of the NEXT 4-step group              */
	float penp[M];     /* TAB (lazy form): the penalty an extension of the slot's latest cell pays, on its way from LDS */
	constexpr bool LAZY = TAB && (CVX_FILL_TAB_LAZY != 0);
	float best[M];
	int best_r[M];
	float lbest = 0.0f;          /* early phase: running maximum of this lane's cells */
	unsigned accA[M], accB[M];   /* direction bit-planes of the current 32-step block */
	/* per-slot lane masks in SGPRs */
	u64 mD[M];         /* latest cell is a deletion (run > 0)  */
	u64 mI[M];         /* latest cell is an insertion          */

	/* Row record: what a slot needs to take row yy (block-local index) over at any later step rnext:
	 *   x = first anti-diagonal of the row (cnt = rnext - x is the column index inside the row, < 0 before it starts)
	 *   y = row length after clipping to [0, W)          z = read character of the row
	 *   w = arena offset of the reference character of anti-diagonal 0 in this row (= ref_off - row index);
	 *       also identifies the row: yy = ref_base - w.
	 * Rows at and beyond H (the ring outlives the tile) get a record that never starts. */
	const unsigned ref_base = ti.ref_off - (unsigned) y0;
	auto make_rec = [&](const int yy) {
		int4 rec;
		rec.w = (int) (ref_base - (unsigned) yy);
		if (yy < H) {
			const RowDesc2 ol = row_at(rv, yy);
			const long long Ws = (long long) W;
			long long lo = ol.x > 0 ? ol.x : 0;
			long long hi = (long long) ol.x + (long long) ol.y;
			if (hi > Ws) hi = Ws;
			if (hi < lo) hi = lo;
			rec.x = yy + y0 + (int) lo;
			rec.y = (int) (hi - lo);
			rec.z = seq[qry_off + (unsigned) yy];
		} else {
			rec.x = yy + y0 + (1 << 30);
			rec.y = 0;
			rec.z = 0;
		}
		return rec;
	};
	/* slot j takes the row of `rec` over; rnext = index of the next step.  Invariant between groups:
	 * xa[j] = rec.w + r + 4, the address of the characters of the group after next. */
	auto take_row = [&](const int j, const int4 rec, const int rnext, const bool fetch_now) {
		cnt[j] = rnext - rec.x;
		len[j] = rec.y;
		qch[j] = rec.z;
		xa[j] = (unsigned) rec.w + (unsigned) rnext + 4u;
		/* The characters of the group starting at rnext: whatever cwn[j] holds will do when the row starts
		 * no earlier than the group after (cells outside a row are forced to the empty element whatever
		 * they compare); the regular prefetch at the top of the next group then picks the row up.  Only a
		 * ring without slack hands a slot over less than a group before its row starts. */
		if (fetch_now || cnt[j] > -4) cwn[j] = *reinterpret_cast<const unsigned *>(seq + (xa[j] - 4u));
	};
	/* the wave writes the records of rows [Y, Y + 16 M) to their slots (sY = the first one's slot inside the wave, wave-uniform) */
	constexpr int kStage = 16 * M;
	auto stage_rows = [&](const int Y, const int sY) {
		if (lane < kStage) {
			const int4 rec = make_rec(Y + lane);
			const int sl = sY + lane;
			s_rec[CHAIN ? 0 : wv * M + sl % M][sl / M] = rec;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      /* one wave: LDS write -> read order across lanes */
	};

	if (TAB) {
		for (int i = lane; i < kPenEntries; i += 64) s_pen[i] = fminf(gem, gext + (float) i * decay);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	}
#pragma unroll
	for (int j = 0; j < M; ++j) {
		s_besty[j][tid] = 0;
		S[j] = 0.0f; Hc[j] = go; V[j] = go; dg[j] = 0.0f;
		drun[j] = 0; irun[j] = 0;
		best[j] = 0.0f; best_r[j] = 0;
		penp[j] = gem;
		accA[j] = accB[j] = 0u;
		mD[j] = 0; mI[j] = 0;
		cwn[j] = 0u;
		take_row(j, make_rec(tid * M + j), r0, true);
	}
	/* rows [N, 2N - 16 M) are staged up front; from then on the hand-over of every row that is a multiple
	 * of 16 M (slot 0 of lanes 0, 16, 32, 48) stages the 16 M rows that end one ring further on: their slots
	 * were all handed over before it (rows end in order), and the first of them is needed only when the row
	 * 16 M above the triggering one ends, N - 16 M row ends later.  (A gang's wave stages its own stretches of 64 M rows:
	 * three of the four chunks of its next stretch up front, then one chunk per trigger, a stretch every N rows.) */
	int stage_next = N + wv * NW, stage_slot = 0;
	auto stage_advance = [&]() {
		stage_next += kStage;
		stage_slot += kStage;
		if (stage_slot >= NW) { stage_slot = 0; stage_next += N - NW; }
	};
	if (!CHAIN) {
		for (int c = 0; c < NW / kStage - 1; ++c) { stage_rows(stage_next, stage_slot); stage_advance(); }
	}

	int gang_failed = 0;
	if (GANG) {
		/* no record is valid yet (tag 0x7fff is the one of step 32767, by when every entry has long been rewritten) */
		if (lane < kGangDepth) s_gx[wv][lane] = ~0ull;
		__syncthreads();
	}
	const int gang_pred = GANG ? (wv + G - 1) % G : 0;      /* the wave whose last slot holds the row above this wave's first */

	const int ngroups = (nsteps + 3) >> 2;
	int late = ngroups >> a.late_shift;
	if (late < a.late_min_groups) late = a.late_min_groups;
	const int gswitch = (EXACT || late >= ngroups) ? 0 : ngroups - late;   /* first exactly tracked group */
	int r = r0;

	/* chain: where this block's last row lives (it feeds the next block) and what has been published */
	const int out_slot = CHAIN ? (ct.rows - 1) : 0;
	const int out_lane = out_slot / M, out_j = out_slot % M;
	u64 *bnd_out = CHAIN ? reinterpret_cast<u64 *>(a.bnd + ct.bnd_out_off) : nullptr;
	u64 *bnd_in = CHAIN ? reinterpret_cast<u64 *>(a.bnd + ct.bnd_in_off) : nullptr;
	const unsigned epoch = a.bnd_epoch;
	int chain_failed = 0;
	BoundaryVal bcur;               /* boundary record of the next step */
	bcur.V = go; bcur.S = 0.0f; bcur.run = 0u; bcur.is_ins = 0u;
	u64 bpre = 0ull;                /* this lane's record of the NEXT chunk, requested one chunk early (epoch 0: not valid) */

	/* one 4-step group; TRACK: exact best-cell tracking (else only the lane maximum) */
	auto group = [&](auto track_tag, const int g) {
		constexpr bool TRACK = decltype(track_tag)::value;
		if (CHAIN && (g & (kChainChunk / 4 - 1)) == 0) {
			/* boundary records of the next kChainChunk steps: record i belongs to column lo + i of the row above this
			 * block.  Lane l < kChainChunk owns the record of step r + l; it asked for it one chunk ago. */
			const int x = (r - y0) + lane;                     /* column of the first row's cell at step r + lane */
			const int idx = x - ct.bnd_lo;
			const bool mine = ct.prev >= 0 && lane < kChainChunk && idx >= 0 && idx < ct.bnd_len;
			u64 q = bpre;
			bool ok = !mine || (unsigned) (q >> 49) == epoch;
			int spins = 0;
			const bool must_poll = !chain_failed && ballot(!ok) != 0ull;
			const u64 poll_t0 = must_poll ? __builtin_amdgcn_s_memtime() : 0ull;      /* (statistics: cvx_timing.chain_poll_ticks) */
			while (!chain_failed && ballot(!ok) != 0ull) {     /* (wave-uniform) the producer has not got there yet */
				if (!ok) {
					q = __hip_atomic_load(bnd_in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					ok = (unsigned) (q >> 49) == epoch;
				}
				if (ballot(!ok) == 0ull) break;
				if (++spins > (1 << 21)) { chain_failed = 1; break; }     /* seconds: never hang the device */
				/* back off: a task that was dispatched long before its turn must not flood the fabric with polls */
				if (spins < 8) __builtin_amdgcn_s_sleep(2);
				else if (spins < 64) __builtin_amdgcn_s_sleep(32);
				else __builtin_amdgcn_s_sleep(127);
			}
			if (must_poll) chain_polled += __builtin_amdgcn_s_memtime() - poll_t0;
			BoundaryVal br;
			br.V = go; br.S = 0.0f; br.run = 0u; br.is_ins = 0u;     /* outside the row above: the empty element */
			if (mine && ok) {
				const unsigned meta = (unsigned) (q >> 32);
				const float sc = __uint_as_float((unsigned) q);
				const unsigned run16 = meta & 0xffffu;
				const bool ins = (meta >> 16) & 1u;
				/* the run register as the producer's slot held it, and the run its gap penalty was computed from */
				const float runf = WRAP ? (float) (int) (short) run16 : (float) run16 - 1.0f;
				br.S = sc;
				br.run = WRAP ? (unsigned) (int) (short) run16 : __float_as_uint((float) run16);
				br.is_ins = ins ? 1u : 0u;
				br.V = ins ? gap_extend_value(sc, runf, gem, gext, decay) : sc + go;
			}
			if (lane < kChainChunk) s_bnd[lane] = br;
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      /* one wave: LDS write -> read order */
			bcur = s_bnd[0];
			/* ask for the next chunk's records now: they are on their way while this chunk computes */
			const int idxn = idx + kChainChunk;
			bpre = 0ull;
			if (ct.prev >= 0 && lane < kChainChunk && idxn >= 0 && idxn < ct.bnd_len)
				bpre = __hip_atomic_load(bnd_in + idxn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		/* this group's reference characters were fetched one group ago */
		unsigned cw[M];
#pragma unroll
		for (int j = 0; j < M; ++j) {
			cw[j] = cwn[j];
			cwn[j] = *reinterpret_cast<const unsigned *>(seq + xa[j]);
			xa[j] += 4u;
		}
		/* flush the previous 32-step block of direction words HERE, right after the
		 * wait for this group's characters: gfx9 counts loads and stores in one vmcnt,
		 * so a store issued just before that wait would be waited for in full */
		if (g != 0 && (g & 7) == 0) {
			/* scalar base of the block + a 32-bit lane offset (a per-lane 64-bit pointer kept across the loop was spilled) */
			uint32_t *d = dirs + (size_t) ((g >> 3) - 1) * (N * 2);
			const unsigned dl = (unsigned) tid * (M * 2);
#pragma unroll
			for (int j = 0; j < M; ++j) { d[dl + 2 * j] = accA[j]; d[dl + 2 * j + 1] = accB[j]; }
		}

#pragma unroll
		for (int i = 0; i < 4; ++i) {
			/* lane boundary: previous lane's last slot, values of step r-1 */
			/* (lazy TAB form: what a slot's latest cell offers -- E to an extension, O to an opening -- is worked out here, one
			 * step after the cell, from its score, the penalty that has come back from LDS meanwhile and its two masks) */
			float p_E[M], p_O[M];
			auto offers = [&](const int j) {
				p_E[j] = fmaxf(S[j] + penp[j], S[j] * -0x1p100f);
				p_O[j] = S[j] + go;
			};
			if (LAZY) offers(M - 1);
			float uV0 = LAZY ? rot1_f(lanes(mI[M - 1]) ? p_E[M - 1] : p_O[M - 1]) : rot1_f(V[M - 1]);
			float uS0 = rot1_f(S[M - 1]);
			run_t uI0;
			if (WRAP || TAB) uI0 = (run_t) rot1_i((int) irun[M - 1]);
			else uI0 = (run_t) rot1_f((float) irun[M - 1]);
			u64 mIu0 = rot1_m(mI[M - 1]);
			u64 gq = 0ull;
			if (GANG && r != r0) gq = __hip_atomic_load(&s_gx[gang_pred][(r - r0 - 1) & (kGangDepth - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			/* gang: the first slot of the wave has the last slot of the wave before it above it.  The record of the step before
			 * was asked for at the top of this step and is looked at only here, in front of slot 0, the last slot of the step --
			 * the waves of a gang run in lock step, it has been there for most of a step (a spin otherwise, bounded) */
			auto gang_take = [&]() {
				float sc = 0.0f, vv = go;
				unsigned run16 = 0u, ins = 0u;
				if (r != r0) {
					const unsigned want_tag = (unsigned) (r - r0 - 1) & 0x7fffu;
					int spins = 0;
					while (!gang_failed && (((unsigned) __builtin_amdgcn_readfirstlane((int) (gq >> 32))) >> 17) != want_tag) {
						if (++spins > (1 << 22)) { gang_failed = 1; break; }      /* never hang the device */
#if CVX_GANG_SLEEP > 0
						__builtin_amdgcn_s_sleep(CVX_GANG_SLEEP);
#endif
						gq = __hip_atomic_load(&s_gx[gang_pred][(r - r0 - 1) & (kGangDepth - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
					const unsigned meta = (unsigned) __builtin_amdgcn_readfirstlane((int) (gq >> 32));
					sc = __uint_as_float((unsigned) __builtin_amdgcn_readfirstlane((int) gq));
					run16 = meta & 0xffffu;
					ins = (meta >> 16) & 1u;
					/* the up candidate is a function of (score, run, insertion): rebuilt with the cell update's own operations */
					vv = ins ? gap_extend_value(sc, (float) run16 - 1.0f, gem, gext, decay) : sc + go;
				}
				if (lane == 0) {
					uV0 = vv;
					uS0 = sc;
					uI0 = TAB ? (run_t) (int) (run16 << 2) : (run_t) (float) run16;
				}
				mIu0 = (mIu0 & ~1ull) | (u64) ins;
			};
			/* ... and offers its own last slot's new cell to the wave after it: lane 63, right after that slot (the first of the step) */
			auto gang_give = [&]() {
				const unsigned run16 = TAB ? (((unsigned) (int) irun[M - 1]) >> 2) & 0xffffu : ((unsigned) (int) (float) irun[M - 1]) & 0xffffu;
				const unsigned meta = run16 | (unsigned) (((mI[M - 1] >> 63) & 1ull) << 16) | (((unsigned) (r - r0) & 0x7fffu) << 17);
				if (lane == 63)
					__hip_atomic_store(&s_gx[wv][(r - r0) & (kGangDepth - 1)], ((u64) meta << 32) | (u64) __float_as_uint(S[M - 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			};
			if (CHAIN) {
				/* the block's first row (lane 0, slot 0) has the previous block's last row above it */
				/* (the record was fetched from LDS one step ago: its latency is off the step's critical path) */
				const BoundaryVal br = bcur;
				bcur = s_bnd[((r - r0) + 1) & (kChainChunk - 1)];
				if (lane == 0) {
					uV0 = br.V;
					uS0 = br.S;
					uI0 = WRAP ? (run_t) (int) br.run : (run_t) __uint_as_float(br.run);
				}
				const int bit = __builtin_amdgcn_readfirstlane((int) br.is_ins);
				mIu0 = (mIu0 & ~1ull) | (u64) (bit & 1);
			}

			/* The cell update in three phases per slot: (1) candidates, maximum and the equality /
			 * activity lane masks; (2) the priority chain on the masks (SALU); (3) the new slot state.
			 * Slot j reads slot j-1's values of the previous step: phase 3 runs in descending j. */
			float p_lc[M], p_dc[M], p_uc[M], p_mx[M];
			u64 p_eL[M], p_eU[M], p_eG[M], p_act[M], p_isDl[M], p_isIu[M];
			u64 p_nD[M], p_nI[M], p_gap[M], p_cread[M];
			auto phase1 = [&](const int j) {
				float uV = (j > 0) ? V[j > 0 ? j - 1 : 0] : uV0;
				const run_t uI = (j > 0) ? irun[j > 0 ? j - 1 : 0] : uI0;
				const u64 mIu = (j > 0) ? mI[j > 0 ? j - 1 : 0] : mIu0;
				float lcv = Hc[j];
				if (LAZY) {
					if (j > 0) {
						offers(j > 0 ? j - 1 : 0);
						uV = lanes(mIu) ? p_E[j > 0 ? j - 1 : 0] : p_O[j > 0 ? j - 1 : 0];
					}
					lcv = lanes(mD[j]) ? p_E[j] : p_O[j];
				}
				const int refc = (int) ((cw[j] >> (8 * i)) & 0xffu);
				const bool eq = (refc == qch[j]);
				const float diag_cell = dg[j] + (eq ? vmat : vmis);
				float lc = lcv, dc = diag_cell, uc = uV;
				const float mx = fmaxf(fmaxf(fmaxf(lc, dc), uc), 0.0f);
				p_lc[j] = lc; p_dc[j] = dc; p_uc[j] = uc; p_mx[j] = mx;
				p_eL[j] = ballot(mx == lc);
				p_eU[j] = ballot(mx == uc);
				p_eG[j] = ballot(mx == dc);
				p_act[j] = ballot((unsigned) cnt[j] < (unsigned) len[j]);
				p_isDl[j] = WRAP ? ballot(drun[j] > 0) : mD[j];
				p_isIu[j] = WRAP ? ballot(uI > 0) : mIu;
			};
			auto phase2 = [&](const int j) {
				const u64 eL = p_eL[j], eU = p_eU[j], eG = p_eG[j], act = p_act[j], isDl = p_isDl[j], isIu = p_isIu[j];
				/* priority: del-extend > ins-extend > diag > del-open > ins-open > stop
				 * (src/ConvexAlignFast.cpp:703-738), on lane masks; nothing fires on a
				 * lane that is outside its row */
				const u64 c2 = isIu & eU;
				const u64 nD = eL & (isDl | ~(c2 | eG)) & act;
				const u64 nI = ~nD & eU & (isIu | ~eG) & act;
				p_nD[j] = nD; p_nI[j] = nI;
				p_gap[j] = nD | nI;
				/* plane 1 = nI | nG with nG = eG & ~gap & act; the act term is dropped: direction
				 * bits of cells outside a row are never read (backtrack_walk masks them) */
				p_cread[j] = nI | (eG & ~nD);
			};
			auto phase3 = [&](const int j) {
				const float uS = (j > 0) ? S[j > 0 ? j - 1 : 0] : uS0;
				const run_t uI = (j > 0) ? irun[j > 0 ? j - 1 : 0] : uI0;
				const u64 nD = p_nD[j], nI = p_nI[j], isDl = p_isDl[j], isIu = p_isIu[j];
				const float mx = p_mx[j];
				/* outside the row the new "cell" is the empty element: score 0 */
				const float sc = lanes(p_act[j]) ? mx : 0.0f;
				run_t nd, ni;
				float runf = 0.0f;
				float E;
				if (TAB) {
					/* the registers hold 4 * (run + 1), the table address of the penalty a cell that extends this one pays
					 * for; as in the float form they are only ever read through the masks (isDl, isIu) */
					const u64 extD = nD & isDl, extI = nI & isIu;
					const int t1 = lanes(extI) ? (int) uI : 4;
					const int ra = lanes(extD) ? (int) drun[j] : t1;
					nd = (run_t) (ra + 4);
					ni = nd;
					const float pen = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_pen) + ra);
					if (LAZY) penp[j] = pen;
					E = fmaxf(sc + pen, sc * -0x1p100f);      /* gap_extend_value with the penalty looked up (dead code in the lazy form) */
				} else if (WRAP) {
					/* indelRun is a short in the reference (src/AlignmentMatrixFast.h:43) */
					nd = lanes(nD) ? (lanes(isDl) ? (run_t) (short) ((int) drun[j] + 1) : (run_t) 1) : (run_t) 0;
					ni = lanes(nI) ? (lanes(isIu) ? (run_t) (short) ((int) uI + 1) : (run_t) 1) : (run_t) 0;
					runf = (float) (lanes(nD) ? nd : ni);
				} else {
					/* One run register per cell: the run of a gap cell (plus one), anything otherwise.  It is only
					 * ever read through the masks "left cell was D" / "up cell was I" (isDl, isIu), so
					 * nothing needs zeroing: an extension continues the run of the cell it extends, an
					 * opening starts at 1 (src/ConvexAlignFast.cpp:655-668,703-738). */
					/* The register holds run + 1, the run of a cell that extends this one, so an extension
					 * is one select and needs no "either extension" mask. */
					const u64 extD = nD & isDl, extI = nI & isIu;
					const float t1 = lanes(extI) ? (float) uI : 1.0f;
					runf = lanes(extD) ? (float) drun[j] : t1;
					nd = (run_t) (runf + 1.0f);
					ni = nd;
				}
				if (!TAB) E = gap_extend_value(sc, runf, gem, gext, decay);
				const float O = sc + go;

				dg[j] = uS;
				S[j] = sc;
				drun[j] = nd;
				irun[j] = ni;
				if (!LAZY) {
					V[j] = lanes(nI) ? E : O;
					Hc[j] = lanes(nD) ? E : O;
				}
				if (TRACK) {
					const u64 better = ballot(sc > best[j]);   /* sc is 0 outside the row, best >= 0 */
					best[j] = lanes(better) ? mx : best[j];
					best_r[j] = lanes(better) ? r : best_r[j];
				} else if (i >= 2) {
					/* early phase: the running maximum samples steps 2 and 3 of every group only.  A cell of
					 * step 0 or 1 scores at most `match` more than its best predecessor (left / up cost, the
					 * diagonal adds at most `match`), and its predecessors lie in sampled steps (or in step 0,
					 * bounded the same way), so every untracked score is <= lbest + match; the acceptance test
					 * at the end carries that slack. */
					lbest = fmaxf(lbest, sc);
				}
				mD[j] = nD;
				mI[j] = nI;
				cnt[j] += 1;
				accA[j] = shl1_in(accA[j], p_gap[j]);       /* plane 0: I or D */
				accB[j] = shl1_in(accB[j], p_cread[j]);     /* plane 1: I or diagonal */
			};
#if CVX_FILL_SCHED == 1
			/* all candidates / compares first, then all mask logic, then all state updates */
#pragma unroll
			for (int j = M - 1; j >= 0; --j) phase1(j);
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int j = M - 1; j >= 0; --j) phase2(j);
			__builtin_amdgcn_sched_barrier(0);
#elif CVX_FILL_SCHED == 2
			/* software pipeline over the slots: the mask logic of slot j runs beside the compares of slot j-1 */
#pragma unroll
			for (int q = M; q >= -1; --q) {
				if (q < M && q >= 0) phase1(q);
				if (q + 1 < M && q + 1 >= 0) phase2(q + 1);
				__builtin_amdgcn_sched_barrier(0);
			}
#endif
#pragma unroll
			for (int j = M - 1; j >= 0; --j) {
#if CVX_FILL_SCHED == 0 || CVX_FILL_SCHED == 3
				if (GANG && j == 0) gang_take();
				phase1(j); phase2(j);
#endif
				phase3(j);
				if (GANG && j == M - 1) gang_give();
#if CVX_FILL_SCHED == 3
				__builtin_amdgcn_sched_barrier(0);
#endif
				if (CHAIN && j == out_j && ct.has_next) {
					/* the last row's new cell goes to the boundary stream (record index = its column in the row) */
					if (lane == out_lane && (unsigned) (cnt[j] - 1) < (unsigned) len[j]) {
						const unsigned run16 = WRAP ? ((unsigned) (int) irun[j] & 0xffffu) : ((unsigned) (int) (float) irun[j] & 0xffffu);
						const unsigned meta = run16 | (unsigned) (((mI[j] >> out_lane) & 1ull) << 16) | (epoch << 17);
						__hip_atomic_store(bnd_out + (cnt[j] - 1), ((u64) meta << 32) | (u64) __float_as_uint(S[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
			}
			r += 1;
		}
		/* hand finished slots to their next row (y + N): a row's last cell is consumed
		 * by the row below one step after it was computed, so wait for cnt > len
		 * (a chained block has at most N rows: nothing is ever handed on) */
		/* (wave-uniform: a row that is a multiple of 16 M is handed over in this group) */
		const bool stage_now = !CHAIN && (ballot(cnt[0] > len[0]) & 0x0001000100010001ull) != 0ull;
#pragma unroll
		for (int j = 0; j < M; ++j) {
			if (!CHAIN && cnt[j] > len[j]) {     /* finished a real row (slots beyond the tile count up from -2^30) */
				if (TRACK) {
					const int yy = (int) (ref_base - (xa[j] - (unsigned) r - 4u));
					if (best_r[j] >= r - cnt[j]) s_besty[j][tid] = yy;
				}
				take_row(j, s_rec[CHAIN ? 0 : wv * M + j][lane], r, false);
			}
		}
		if (TAB) {
			/* (one register per slot: the deletion and the insertion run of a cell share it in this form) */
#pragma unroll
			for (int j = 0; j < M; ++j) { const int c = min((int) drun[j], 4 * kPenClamp); drun[j] = (run_t) c; irun[j] = (run_t) c; }
		}
		if (stage_now) {
			stage_rows(stage_next, stage_slot);
			stage_advance();
		}
	};

	for (int g = 0; g < gswitch; ++g) group(BoolTag<false>(), g);
	for (int g = gswitch; g < ngroups; ++g) group(BoolTag<true>(), g);

	if (ngroups > 0) {
		/* last block (complete or partial): left-align so that step (t & 31) sits at
		 * bit 31 - (t & 31) */
		const int done = ((ngroups - 1) & 7) + 1;     /* groups in the last block */
		const int sh = 32 - 4 * done;
		uint32_t *d = dirs + ((size_t) ((ngroups - 1) >> 3) * N + (size_t) tid * M) * 2;
#pragma unroll
		for (int j = 0; j < M; ++j) { d[2 * j] = sh ? accA[j] << sh : accA[j]; d[2 * j + 1] = sh ? accB[j] << sh : accB[j]; }
	}

	/* argmax with the reference's tie-break: first strict maximum in (y, x) order
	 * (src/ConvexAlignFast.cpp:758-763 / :1165-1170) among the exactly tracked cells */
	float b = -1.0f;
	int by = 0x7fffffff, bx = 0x7fffffff;
#pragma unroll
	for (int j = 0; j < M; ++j) {
		int vy = s_besty[j][tid];
		const int ycur = (int) (ref_base - (xa[j] - (unsigned) r - 4u));      /* the row the slot holds now */
		if (ycur < H && best_r[j] >= r - cnt[j]) vy = ycur;
		const float v = best[j];
		const int vx = best_r[j] - vy;
		if (v > 0.0f) {     /* best[] starts at 0: a slot that never saw a positive score has no candidate */
			if (v > b || (v == b && (vy < by || (vy == by && vx < bx)))) { b = v; by = vy; bx = vx; }
		}
	}
	float be = lbest;       /* maximum over the cells that were not tracked exactly */
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const float ob = __shfl_xor(b, off, 64);
		const int oy = __shfl_xor(by, off, 64);
		const int ox = __shfl_xor(bx, off, 64);
		if (ob > b || (ob == b && (oy < by || (oy == by && ox < bx)))) { b = ob; by = oy; bx = ox; }
		if (!EXACT) be = fmaxf(be, __shfl_xor(be, off, 64));
	}
	if (CHAIN) {
		if (lane == 0) {
			ChainOut co;
			co.score = b;
			co.best_y = (b > -1.0f) ? by + y0 : 0;
			co.best_x = (b > -1.0f) ? bx - y0 : 0;      /* bx was computed against block-local rows */
			co.failed = chain_failed;
			a.chain_out[ct.blk] = co;
			unsigned long long *ticks = reinterpret_cast<unsigned long long *>(a.redo_count + kCtrChainTicks);
			atomicAdd(ticks, (unsigned long long) (__builtin_amdgcn_s_memtime() - chain_t0));
			if (chain_polled) atomicAdd(ticks + 1, (unsigned long long) chain_polled);
		}
		return;
	}
	if (GANG) {
		/* the waves' partial results, combined by wave 0 in wave order with the same tie-break */
		if (lane == 0) {
			s_gred[wv][0] = b; s_gred[wv][1] = __int_as_float(by); s_gred[wv][2] = __int_as_float(bx);
			s_gred[wv][3] = EXACT ? 0.0f : be;
			s_gfail[wv] = gang_failed;
		}
		__syncthreads();
		if (wv != 0) return;
#pragma unroll
		for (int w = 1; w < G; ++w) {
			const float ob = s_gred[w][0];
			const int oy = __float_as_int(s_gred[w][1]), ox = __float_as_int(s_gred[w][2]);
			if (ob > b || (ob == b && (oy < by || (oy == by && ox < bx)))) { b = ob; by = oy; bx = ox; }
			if (!EXACT) be = fmaxf(be, s_gred[w][3]);
		}
	}
	if (lane == 0) {
		/* b == -1: no positive score among the tracked cells.  If there is none anywhere either,
		 * the reference (curr_max starts at -1) takes the first cell in (y, x) order, score 0;
		 * backtrack_kernel resolves that rare case. */
		TileOut o;
		o.score = b;
		o.status = 0;
		o.best_x = (b > -1.0f) ? bx : 0;
		o.best_y = (b > -1.0f) ? by : 0;
		o.ref_position = 0; o.qstart = 0; o.qend = 0; o.n_ops = 0; o.ops_first = 0;
		/* pad = 0: filled, not backtracked yet; kPadRedo: an untracked cell scored at least as much
		 * as the best tracked one, so the first strict maximum is not known -> exact pass */
		/* be under-estimates the early maximum by at most `match` (steps 0 and 1 of a group are not sampled);
		 * 2 * match + 1 also covers the rounding of the float adds behind that bound */
		o.pad = (!EXACT && gswitch > 0 && !(b > be + 2.0f * a.sp.mat + 1.0f)) ? kPadRedo : 0;
		if (GANG) {
			/* a wave that gave up waiting for its neighbour's record: CVX_TILE_UNSUPPORTED, loud, never a hang or a wrong answer */
			bool failed = gang_failed != 0;
			for (int w = 1; w < G; ++w) failed = failed || s_gfail[w] != 0;
			if (failed) { o.status = -1; o.pad = 0; }
		}
		a.tout[t] = o;
	}
}

/* A tile record read by a whole wave: lane i fetches dword i, fields come back as SGPRs
 * through v_readlane: three VGPRs instead of one per field. */
CVX_DEV int rec_load(const void *rec, const int n_dwords, const int lane) {
	return lane < n_dwords ? reinterpret_cast<const int *>(rec)[lane] : 0;
}
CVX_DEV int fld(const int w, const int k) { return __builtin_amdgcn_readlane(w, k); }
CVX_DEV unsigned long long fld64(const int w, const int k) {
	return ((unsigned long long) (unsigned) fld(w, k + 1) << 32) | (unsigned) fld(w, k);
}

/* backtrack of one tile by one wave (nothing to do for skipped, invalid or already walked tiles) */
CVX_DEV void walk_tile(const BacktrackArgs &a, const int t, const int lane) {
	static_assert(sizeof(TileRun) == 48 && sizeof(TileIn) == 32 && sizeof(TileOut) == 40, "record layout");
	const int wr = rec_load(a.trun + t, 12, lane);   /* dir_off 0-1, ops_off 2-3, ring 4, ops_cap 5, r0 6, nsteps 7, skip 8, mnw 9, chain_blk0 10 */
	const int wo = rec_load(a.tout + t, 10, lane);   /* score 0, status 1, best_x 2, best_y 3, ..., pad 9 */
	if (fld(wr, 8) != 0) return;
	TileOut o;
	o.score = __int_as_float(fld(wo, 0));
	o.status = fld(wo, 1);
	o.best_x = fld(wo, 2); o.best_y = fld(wo, 3);
	o.ref_position = 0; o.qstart = 0; o.qend = 0; o.n_ops = 0; o.ops_first = 0;
	o.pad = fld(wo, 9);
	if (o.status != 0 || o.pad != 0) return;
	const int wi = rec_load(a.tin + t, 8, lane);     /* ref_off 0, qry_off 1, W 2, H 3, row_off 4-5 */
	const int H = fld(wi, 3), W = fld(wi, 2);
	const RowView rows = row_view(a.rsrc[t], a.rows, fld64(wi, 4));
	if (!(o.score > 0.0f)) {
		/* no positive score: the reference's best cell is the first cell in (y, x) order with
		 * score 0 (curr_max starts at -1, src/ConvexAlignFast.cpp:758-763) */
		int fy, fx;
		first_cell(rows, H, W, lane, &fy, &fx);
		if (fy < 0) {
			o.score = -1.0f; o.status = 5; o.pad = 1;
			if (lane == 0) a.tout[t] = o;
			return;
		}
		o.score = 0.0f; o.best_x = fx; o.best_y = fy;
	}
	const int cb0 = fld(wr, 10);
	if (cb0 >= 0)
		backtrack_walk<true>(lane, H, fld(wr, 4), fld(wr, 6), fld(wr, 5), rows,
				reinterpret_cast<const uint2 *>(a.dirs), a.chain_blk + cb0,
				a.seq + (unsigned) fld(wi, 0), a.seq + (unsigned) fld(wi, 1),
				a.ops + fld64(wr, 2), o);
	else
		backtrack_walk<false>(lane, H, fld(wr, 4), fld(wr, 6), fld(wr, 5), rows,
				reinterpret_cast<const uint2 *>(a.dirs + fld64(wr, 0)), nullptr,
				a.seq + (unsigned) fld(wi, 0), a.seq + (unsigned) fld(wi, 1),
				a.ops + fld64(wr, 2), o);
	o.pad = 1;
	if (lane == 0) a.tout[t] = o;
}

/* ------------------------------------------------------------------ backtrack, G lanes per tile */

/*
 * The same walk as backtrack_walk with G (= 16) lanes per tile and four tiles per wave.  The
 * one-wave-per-tile walk keeps its state wave-uniform and therefore lives on the scalar unit: ~140
 * SALU instructions per probe, and one scalar issue slot per SIMD every ~4.3 cycles makes that the
 * bound (SQ counters: the scalar unit 78 % busy, VALU 30 %).  Diagonal runs between two gaps are
 * ~7 cells long at 15 % error, so 64 probing lanes are mostly idle anyway.  Here the walk state is
 * group-uniform in VGPRs, predicates replace the scalar branches, four tiles share every
 * instruction, and the EQ / X sub-runs of a diagonal run are written by their own lanes instead of a
 * scalar loop.  Probe geometry, run skipping, validPath and the op encoding are those of
 * backtrack_walk (reference src/ConvexAlignFast.cpp:335-432, src/AlignmentMatrixFast.cpp:213-220).
 */
template <int G>
struct Group {
	int gl;        /* lane inside the group */
	int base;      /* first lane of the group */
	CVX_DEV unsigned ballot(bool p) const {
		const u64 b = __builtin_amdgcn_ballot_w64(p);
		return (unsigned) (b >> base) & (G == 32 ? 0xffffffffu : ((1u << (G & 31)) - 1u));
	}
	CVX_DEV int bcast(int v, int l) const { return __shfl(v, base + l, 64); }
};

template <int G>
CVX_DEV void backtrack_walk_grp(const Group<G> g, const bool has_tile, const bool chained, const int H, const int N, const int r0, const int ops_cap,
		const RowView &rows, const uint2 *dirs, const ChainBlk *blk, const uint8_t *ref, const uint8_t *qry, int *ops, TileOut &o) {
	/* `chained` is a property of the group's tile (a wave may carry both kinds): dirs is the tile's own
	 * region for whole tiles and the arena for chained ones, whose blocks carry their offsets */
	const int gl = g.gl;
	const int best_x = o.best_x, best_y = o.best_y;
	bool act = has_tile;
	if (has_tile && best_y <= 0) { o.status = 1; act = false; }      /* src/ConvexAlignFast.cpp:338 */
	const bool walked = act;

	const int qend = (H - best_y) - 1;
	int idx = ops_cap - 1;
	int elem = 4;          /* CIGAR_S: the trailing clip is tracked but not stored */
	int elem_len = qend;
	int consumed = qend;
	int x = best_x, y = best_y;
	int s = act ? y % N : 0;
	int status = 0;
	unsigned want = 3u;
	int budget = 2 * (best_x + best_y) + 8;

	auto emit = [&](int cur, int n) {      /* revBacktrack's run-length bookkeeping (:395-403) */
		if (n <= 0) return;
		if (cur == elem) {
			elem_len += n;
		} else {
			if (elem != 4) { if (gl == 0) ops[idx] = (elem_len << 4) | elem; idx -= 1; }
			elem = cur;
			elem_len = n;
		}
	};
	int c_gb = -1;             /* chained tiles: block whose record c_cb holds (per lane) */
	ChainBlk c_cb;
	c_cb.dir_off = 0; c_cb.tblk0 = 0; c_cb.nblk32 = 1;

	while (__builtin_amdgcn_ballot_w64(act) != 0ull) {
		if (act) {
			bool go_on = true;
			if (--budget < 0) { status = 3; go_on = false; }
			else {
				const int dx = (want != 1u) ? 1 : 0, dy = (want != 2u) ? 1 : 0;
				const int cx = x - gl * dx, cy = y - gl * dy;
				const bool inside = (cx >= 0 && cy >= 0);
				const int lx = cx > 0 ? cx : 0, ly = cy > 0 ? cy : 0;
				int sl = s - gl * dy;
				if (sl < 0) sl += N;
				const int tt = cx + cy - r0;
				const int ttc = tt > 0 ? tt : 0;
				const RowDesc2 ol = row_at(rows, ly);
				size_t widx = (size_t) (ttc >> 5) * N + sl;
				if (chained) {
					/* the block record of this lane's row: kept from the previous probe while the row stays in
					 * the same 64-row block, so that most probes cost one memory round trip, not two */
					const int gb = ly / N;
					if (gb != c_gb) { c_cb = blk[gb]; c_gb = gb; }
					int wr = (ttc >> 5) - c_cb.tblk0;
					wr = wr < 0 ? 0 : (wr >= c_cb.nblk32 ? c_cb.nblk32 - 1 : wr);
					widx = c_cb.dir_off + (size_t) wr * N + (size_t) (ly - gb * N);
				}
				const uint2 w = dirs[widx];
				const int rc = ref[lx], qc = qry[ly];
				const unsigned eqm = g.ballot(rc == qc);
				const bool in_row = inside && tt >= 0 && cx >= ol.x && cx < ol.x + ol.y;    /* getDirection: outside -> STOP */
				unsigned code = plane_code(w.x, w.y, 31 - (ttc & 31));
				if (!in_row) code = 0u;
				int minC, maxC;
				valid_bounds(ol.x, ol.y, minC, maxC);

				const unsigned full = (G == 32) ? 0xffffffffu : ((1u << (G & 31)) - 1u);
				const unsigned run = g.ballot(code == want);
				const int L = (run == full) ? G : __builtin_ctz(~run);
				const unsigned low = (L == G) ? full : ((1u << L) - 1u);
				const unsigned vp = g.ballot(cx > minC && cx < maxC);      /* validPath before every move (:368-373) */
				if ((~vp & low) != 0u) { status = 2; go_on = false; }
				else {
					if (want == 3u) {
						if (L > 0) {
							/* EQ / X sub-runs of the diagonal run, lane-parallel: sub-run j (cells from the
							 * current one backwards) goes to ops[idx - w0 - j]; the last one stays pending */
							const unsigned e = eqm & low;
							const unsigned starts = (((e ^ (e << 1)) & low) & ~1u) | 1u;
							const int k = __builtin_popcount(starts);
							const int op0 = (e & 1u) ? 7 : 8;
							if (k == 1) {
								emit(op0, L);
							} else {
								const unsigned after0 = starts & ~1u;
								int len0 = __builtin_ctz(after0);
								int w0 = 0;
								if (op0 == elem) len0 += elem_len;
								else if (elem != 4) { if (gl == 0) ops[idx] = (elem_len << 4) | elem; w0 = 1; }
								const bool is_start = gl < L && ((starts >> gl) & 1u) != 0u;
								const int j = __builtin_popcount(starts & ((1u << gl) - 1u));
								const unsigned rest = (starts >> gl) >> 1;
								int mylen = rest ? __builtin_ctz(rest) + 1 : L - gl;
								if (gl == 0) mylen = len0;
								const int myop = ((e >> gl) & 1u) ? 7 : 8;
								if (is_start && j < k - 1) ops[idx - w0 - j] = (mylen << 4) | myop;
								idx -= w0 + (k - 1);
								const int lastpos = 31 - __builtin_clz(starts);
								elem = ((e >> lastpos) & 1u) ? 7 : 8;
								elem_len = L - lastpos;
							}
						}
						x -= L; y -= L; consumed += L;
					} else if (want == 1u) {
						emit(1, L);
						y -= L; consumed += L;
					} else {
						emit(2, L);
						x -= L;
					}
					if (want != 2u) { s -= L; if (s < 0) s += N; }
					if (L < G) {
						const unsigned nxt = (unsigned) g.bcast((int) code, L);
						if (nxt == 0u) go_on = false;                       /* CIGAR_STOP (or outside the matrix) */
						else if (want != 3u) want = nxt;
						else {
							/* a diagonal run ended in a gap at (x, y) = lane L's cell: follow the gap run inside
							 * the words this probe already holds */
							const int tp = x + y - r0;
							if (nxt == 2u) {
								const unsigned wx = (unsigned) g.bcast((int) w.x, L);
								const unsigned wy = (unsigned) g.bcast((int) w.y, L);
								const int row_lo = max(g.bcast(ol.x, L), 0);
								const int b0 = 31 - (tp & 31);
								const unsigned dm = (wx & ~wy) >> b0;
								int Ld = (~dm == 0u) ? 32 : __builtin_ctz(~dm);
								const int in_word = 32 - b0;
								const int in_rowc = x - row_lo + 1;
								if (Ld > in_rowc) Ld = in_rowc;
								const int rminC = g.bcast(minC, L), rmaxC = g.bcast(maxC, L);
								if (!(x - (Ld - 1) > rminC && x < rmaxC)) { status = 2; go_on = false; }
								else {
									emit(2, Ld);
									x -= Ld;
									if (Ld == in_rowc) go_on = false;            /* next cell is left of the row: STOP */
									else if (Ld == in_word) want = 2u;           /* the run may go on in the previous word */
									else {
										const unsigned c2 = plane_code(wx, wy, b0 + Ld);
										if (c2 == 0u) go_on = false; else want = c2;
									}
								}
							} else {
								const int k = gl - L;
								const int tn = tp - k;
								const bool have = (k >= 0) && (cy >= 0) && (tn >= 0) && ((tn >> 5) == (ttc >> 5));
								const bool col_in = (x >= ol.x) && (x < ol.x + ol.y);
								unsigned c2 = plane_code(w.x, w.y, 31 - (tn & 31));
								if (!col_in) c2 = 0u;
								const unsigned im = g.ballot(have && c2 == 1u) >> L;      /* bit k: cell (x, y - k) is I; bit 0 is set */
								const int Li = (~im == 0u) ? 32 : __builtin_ctz(~im);       /* <= G - L */
								const unsigned ilow = (Li >= 32) ? 0xffffffffu : ((1u << Li) - 1u);
								const unsigned vpi = g.ballot(x > minC && x < maxC) >> L;
								if ((~vpi & ilow) != 0u) { status = 2; go_on = false; }
								else {
									emit(1, Li);
									y -= Li; consumed += Li;
									s -= Li; if (s < 0) s += N;
									const int en = L + Li;
									want = 1u;                                   /* default: let the next probe look again */
									const unsigned hv = g.ballot(have);
									const int c3i = g.bcast((int) c2, en < G ? en : 0);
									if (en < G && ((hv >> en) & 1u) != 0u) {
										if (c3i == 0) go_on = false; else want = (unsigned) c3i;
									}
								}
							}
						}
					}
				}
			}
			if (!go_on) act = false;
		}
	}
	if (walked) {
		if (status == 0) {
			if (elem != 4) { if (gl == 0) ops[idx] = (elem_len << 4) | elem; idx -= 1; }
			consumed += (y + 1);
			o.ref_position = x + 1;
			o.qstart = y + 1;
			o.qend = qend;
			o.ops_first = idx + 1;
			o.n_ops = ops_cap - 1 - idx;
			if (H != consumed) status = 3;
		}
		o.status = status;
	}
}

/* G lanes per tile: block b walks tiles order[b * (64 / G) ...] (largest first: the four tiles of a
 * wave have paths of similar length) */
template <int G>
__global__ void __launch_bounds__(64)
backtrack_grp_kernel(const BacktrackArgs a, const int32_t *order, const int n_order) {
	const int lane = threadIdx.x;
	Group<G> g;
	g.gl = lane & (G - 1);
	g.base = lane & ~(G - 1);
	const int q = blockIdx.x * (64 / G) + lane / G;
	bool has = q < n_order;
	const int t = has ? order[q] : 0;
	TileOut o;
	o.score = -1.0f; o.status = 0; o.best_x = 0; o.best_y = 0;
	o.ref_position = 0; o.qstart = 0; o.qend = 0; o.n_ops = 0; o.ops_first = 0; o.pad = 0;
	TileRun tr;
	TileIn ti;
	tr.skip = 1; tr.ring = 64; tr.r0 = 0; tr.ops_cap = 8; tr.dir_off = 0; tr.ops_off = 0; tr.chain_blk0 = -1;
	ti.H = 0; ti.W = 0; ti.row_off = 0; ti.ref_off = 0; ti.qry_off = 0;
	if (has) {
		tr = a.trun[t];
		o = a.tout[t];
		ti = a.tin[t];
		if (tr.skip != 0 || o.status != 0 || o.pad != 0) has = false;
	}
	const int H = ti.H, W = ti.W;
	/* (a group's tile may be of either kind: per-lane fmt; a batch is all of one kind in practice and the branch in row_at uniform) */
	const RowView rows = row_view(a.rsrc[t], a.rows, ti.row_off);
	bool dead = false;       /* no cell at all: status 5 */
	if (__builtin_amdgcn_ballot_w64(has && !(o.score > 0.0f)) != 0ull) {
		/* no positive score: the reference's best cell is the first cell in (y, x) order with score 0
		 * (curr_max starts at -1, src/ConvexAlignFast.cpp:758-763) */
		if (has && !(o.score > 0.0f)) {
			int fy = -1, fx = 0;
			bool searching = true;
			for (int y0 = 0; __builtin_amdgcn_ballot_w64(searching && y0 < H) != 0ull; y0 += G) {
				if (searching && y0 < H) {
					const int yy = y0 + g.gl;
					bool hc = false;
					int lo_i = 0;
					if (yy < H) {
						const RowDesc2 ol = row_at(rows, yy);
						long long lo = ol.x > 0 ? ol.x : 0;
						long long hi = (long long) ol.x + (long long) ol.y;
						if (hi > W) hi = W;
						hc = hi > lo;
						lo_i = (int) lo;
					}
					const unsigned m = g.ballot(hc);
					const int l = m ? __builtin_ctz(m) : 0;
					const int fxl = g.bcast(lo_i, l);
					if (m != 0u) { fy = y0 + l; fx = fxl; searching = false; }
				}
			}
			if (fy < 0) { dead = true; has = false; }
			else { o.score = 0.0f; o.best_x = fx; o.best_y = fy; }
		}
	}
	const bool chained = tr.chain_blk0 >= 0;
	int *ops = a.ops + tr.ops_off;
	const uint8_t *ref = a.seq + ti.ref_off, *qry = a.seq + ti.qry_off;
	const uint2 *dirs = reinterpret_cast<const uint2 *>(chained ? a.dirs : a.dirs + tr.dir_off);
	backtrack_walk_grp<G>(g, has, chained, H, tr.ring, tr.r0, tr.ops_cap, rows, dirs,
			a.chain_blk + (chained ? tr.chain_blk0 : 0), ref, qry, ops, o);
	if (dead) { o.score = -1.0f; o.status = 5; }
	if ((has || dead) && g.gl == 0) { o.pad = 1; a.tout[t] = o; }
}

/* one wave per tile, after every fill launch of the batch has finished.  (Walkers running
 * BESIDE the fill, in the two wave slots per SIMD it leaves free, were tried: the fill slowed
 * down by as much as the backtrack took -- both phases are bound by instruction issue,
 * DESIGN.md 5.) */
__global__ void __launch_bounds__(64)
backtrack_kernel(const BacktrackArgs a, const int32_t *order, const int n_order) {
	/* block b walks tile order[b] (longest read first; a class's own tiles when a batch walks class by class), or tile b */
	if ((int) blockIdx.x >= n_order) return;
	const int t = order ? order[blockIdx.x] : (int) blockIdx.x;
	walk_tile(a, t, threadIdx.x);
}

/*
 * finalize_kernel -- what the host used to do between backtrack and compaction, on the device:
 * exclusive prefix sum of the op counts of the valid tiles (= each tile's slice of the dense ops
 * arena), the caller-facing result records (cvx_result layout) and the batch summary.  One
 * workgroup; a thread takes a contiguous chunk of tiles.
 */
__global__ void __launch_bounds__(256)
finalize_kernel(const TileOut *tout, const TilePlan *plan, uint64_t *dst_off, ResultRec *res,
		BatchSummary *sum, int32_t *counters, int n_tiles, unsigned long long dense_cap) {
	/* 256 threads = one wave per SIMD: a workgroup that finds room on a CU the next batch's fill
	 * already occupies (a 1024-thread group waited ~50 ms for sixteen free wave slots on one CU) */
	constexpr int T = 256;
	__shared__ unsigned long long s_part[T];
	const int tid = threadIdx.x;
	const int per = (n_tiles + T - 1) / T;
	const int t0 = min(n_tiles, tid * per), t1 = min(n_tiles, t0 + per);
	unsigned long long mine = 0;
	for (int t = t0; t < t1; ++t) {
		const TileOut o = tout[t];
		if (o.status == 0 && o.n_ops > 0) mine += (unsigned long long) o.n_ops;
	}
	s_part[tid] = mine;
	__syncthreads();
	/* Hillis-Steele inclusive scan over the partial sums */
	for (int d = 1; d < T; d <<= 1) {
		const unsigned long long v = (tid >= d) ? s_part[tid - d] : 0ull;
		__syncthreads();
		s_part[tid] += v;
		__syncthreads();
	}
	unsigned long long off = s_part[tid] - mine;
	int n_valid = 0;
	for (int t = t0; t < t1; ++t) {
		const TileOut o = tout[t];
		ResultRec r;
		r.score = o.score;
		r.status = o.status;
		r.best_x = o.best_x; r.best_y = o.best_y;
		r.ref_position = o.ref_position; r.qstart = o.qstart; r.qend = o.qend;
		r.n_ops = (o.status == 0) ? o.n_ops : 0;
		r.ops_begin = off;
		r.cells = plan[t].cells;
		res[t] = r;
		dst_off[t] = off;
		if (o.status == 0) { n_valid += 1; if (o.n_ops > 0) off += (unsigned long long) o.n_ops; }
	}
	__shared__ int s_valid;
	if (tid == 0) s_valid = 0;
	__syncthreads();
	if (n_valid) atomicAdd(&s_valid, n_valid);
	__syncthreads();
	if (tid == T - 1) {
		BatchSummary b;
		b.ops_total = s_part[T - 1];
		b.dense_cap = dense_cap;
		b.n_valid = s_valid;
		b.n_redone = counters ? counters[0] : 0;
		b.chain_task_ticks = counters ? reinterpret_cast<const unsigned long long *>(counters + kCtrChainTicks)[0] : 0ull;
		b.chain_poll_ticks = counters ? reinterpret_cast<const unsigned long long *>(counters + kCtrChainTicks)[1] : 0ull;
		*sum = b;
	}
	/* last reader of the batch's counters (redo statistics, chain tickets): leave them zeroed for the
	 * batch's next run (after the summary above has read the redo count) */
	__syncthreads();
	if (counters && tid < 64) counters[tid] = 0;
}

/* dense[dst_off[t] .. +n_ops) = region of tile t (tiles that do not fit the arena are skipped:
 * the host sees ops_total > dense_cap in the summary, grows the arena and compacts again) */
__global__ void __launch_bounds__(256)
compact_ops_kernel(const int32_t *regions, const TileRun *trun, const TileOut *tout,
		const uint64_t *dst_off, uint32_t *dense, int n_tiles, unsigned long long dense_cap) {
	const int t = blockIdx.x;
	if (t >= n_tiles) return;
	const TileOut o = tout[t];
	if (o.status != 0 || o.n_ops <= 0) return;
	const unsigned long long d0 = dst_off[t];
	if (d0 + (unsigned long long) o.n_ops > dense_cap) return;
	const int32_t *src = regions + trun[t].ops_off + o.ops_first;
	uint32_t *dst = dense + d0;
	for (int i = threadIdx.x; i < o.n_ops; i += blockDim.x) dst[i] = (uint32_t) src[i];
}

/* ------------------------------------------------------------------ launchers */

/* best cell of a chained tile = first strict maximum over its row blocks in block order (blocks
 * are in row order and each block's own best is already its first strict maximum in (y, x) order) */
__global__ void __launch_bounds__(64)
chain_reduce_kernel(const int32_t *tiles, int n_tiles, const TileRun *trun, const ChainOut *cout, TileOut *tout) {
	const int q = blockIdx.x;
	if (q >= n_tiles) return;
	const int t = tiles[q];
	const TileRun tr = trun[t];
	const int lane = threadIdx.x;
	float b = -1.0f;
	int by = 0x7fffffff, bx = 0x7fffffff, failed = 0;
	for (int g = lane; g < tr.chain_nblk; g += 64) {
		const ChainOut c = cout[tr.chain_blk0 + g];
		failed |= c.failed;
		if (c.score > b) { b = c.score; by = c.best_y; bx = c.best_x; }      /* g ascending: earlier blocks win ties */
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const float ob = __shfl_xor(b, off, 64);
		const int oy = __shfl_xor(by, off, 64);
		const int ox = __shfl_xor(bx, off, 64);
		failed |= __shfl_xor(failed, off, 64);
		if (ob > b || (ob == b && (oy < by || (oy == by && ox < bx)))) { b = ob; by = oy; bx = ox; }
	}
	if (lane == 0) {
		TileOut o;
		o.score = b;
		o.status = failed ? -1 : 0;          /* CVX_TILE_UNSUPPORTED: loud, never silent */
		o.best_x = (b > -1.0f) ? bx : 0;
		o.best_y = (b > -1.0f) ? by : 0;
		o.ref_position = 0; o.qstart = 0; o.qend = 0; o.n_ops = 0; o.ops_first = 0;
		o.pad = 0;
		tout[t] = o;
	}
}

/* gangs: G waves per tile on one ring of 64 * 3 * G slots (two-phase with the penalty table when the scoring allows, else
 * arithmetic; and the exact pass) */
template <int G>
static hipError_t launch_fill_gang(const FillArgs &a, int mode, hipStream_t st) {
	if (mode == kFillExact) hipLaunchKernelGGL((fill_ring_kernel<3, false, kFillExact, false, G>), dim3(a.list_n), dim3(64 * G), 0, st, a);
	else if (a.pen_table) hipLaunchKernelGGL((fill_ring_kernel<3, false, kFillTwoPhase, true, G>), dim3(a.list_n), dim3(64 * G), 0, st, a);
	else hipLaunchKernelGGL((fill_ring_kernel<3, false, kFillTwoPhase, false, G>), dim3(a.list_n), dim3(64 * G), 0, st, a);
	return hipGetLastError();
}

template <int M, bool WRAP>
static hipError_t launch_fill_t(const FillArgs &a, int mode, size_t pad_lds, hipStream_t st) {
	/* one wave per tile of the list (per task for chained tiles; pad_lds = unused dynamic LDS that
	 * caps how many waiting tasks are resident) */
	if (mode == kFillChain) {
		if constexpr (M == 1 || M == 2 || M == 4) hipLaunchKernelGGL((fill_ring_kernel<M, WRAP, kFillChain>), dim3(a.list_n), dim3(64), pad_lds, st, a);
		else return hipErrorInvalidValue;
	} else if (mode == kFillExact) hipLaunchKernelGGL((fill_ring_kernel<M, WRAP, kFillExact>), dim3(a.list_n), dim3(64), 0, st, a);
	else if (!WRAP && a.pen_table) hipLaunchKernelGGL((fill_ring_kernel<M, false, kFillTwoPhase, true>), dim3(a.list_n), dim3(64), 0, st, a);
	else hipLaunchKernelGGL((fill_ring_kernel<M, WRAP, kFillTwoPhase>), dim3(a.list_n), dim3(64), 0, st, a);
	return hipGetLastError();
}

template <int M>
static hipError_t launch_fill_w(const FillArgs &a, bool wrap, int mode, size_t pad_lds, hipStream_t st) {
	return wrap ? launch_fill_t<M, true>(a, mode, pad_lds, st) : launch_fill_t<M, false>(a, mode, pad_lds, st);
}

hipError_t launch_fill(int m, int gang, bool wrap, int mode, const FillArgs &a, size_t pad_lds, hipStream_t st) {
	if (a.list_n <= 0) return hipSuccess;
	if (gang > 1) {
		if (m != 3 || wrap || mode == kFillChain) return hipErrorInvalidValue;
		return gang == 2 ? launch_fill_gang<2>(a, mode, st) : gang == 3 ? launch_fill_gang<3>(a, mode, st) : hipErrorInvalidValue;
	}
	switch (m) {
	case 1: return launch_fill_w<1>(a, wrap, mode, pad_lds, st);
	case 2: return launch_fill_w<2>(a, wrap, mode, pad_lds, st);
	case 3: return launch_fill_w<3>(a, wrap, mode, pad_lds, st);
	case 4: return launch_fill_w<4>(a, wrap, mode, pad_lds, st);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_chain_reduce(const int32_t *tiles, int n_tiles, const TileRun *trun, const ChainOut *cout, TileOut *tout, hipStream_t st) {
	if (n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(chain_reduce_kernel, dim3(n_tiles), dim3(64), 0, st, tiles, n_tiles, trun, cout, tout);
	return hipGetLastError();
}

hipError_t launch_expand_rows(const RowSrc *rsrc, const TileIn *tin, const uint8_t *delta, const RowDesc *rowsx, RowDesc *rows,
		int n_tiles, bool closed_forms, hipStream_t st) {
	if (n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(expand_rows_kernel, dim3(n_tiles), dim3(64), 0, st, rsrc, tin, delta,
			reinterpret_cast<const int2 *>(rowsx), reinterpret_cast<int2 *>(rows), n_tiles, closed_forms ? 1 : 0);
	return hipGetLastError();
}

hipError_t launch_plan(const RowDesc *rows, const RowSrc *rsrc, const TileIn *tin, TilePlan *plan, int n_tiles, uint64_t rows_per_tile,
		unsigned long long max_matrix_mb, hipStream_t st) {
	if (n_tiles <= 0) return hipSuccess;
	if (rows_per_tile >= 1024)
		hipLaunchKernelGGL(plan_kernel<256>, dim3(n_tiles), dim3(256), 0, st,
				reinterpret_cast<const int2 *>(rows), rsrc, tin, plan, n_tiles, max_matrix_mb);
	else
		hipLaunchKernelGGL(plan_kernel<64>, dim3((n_tiles + 3) / 4), dim3(256), 0, st,
				reinterpret_cast<const int2 *>(rows), rsrc, tin, plan, n_tiles, max_matrix_mb);
	return hipGetLastError();
}

hipError_t launch_backtrack(const BacktrackArgs &a, const int32_t *order, int n_order, int group, hipStream_t st) {
	if (a.n_tiles <= 0) return hipSuccess;
	if (group == 16 && order != nullptr) {
		if (n_order <= 0) return hipSuccess;
		hipLaunchKernelGGL(backtrack_grp_kernel<16>, dim3((n_order + 3) / 4), dim3(64), 0, st, a, order, n_order);   /* four tiles per wave */
	} else if (group == 4 && order != nullptr) {
		if (n_order <= 0) return hipSuccess;
		hipLaunchKernelGGL(backtrack_grp_kernel<4>, dim3((n_order + 15) / 16), dim3(64), 0, st, a, order, n_order);  /* sixteen tiles per wave (CVX_TUNE_BT_GROUP=4 only) */
	} else if (group == 8 && order != nullptr) {
		if (n_order <= 0) return hipSuccess;
		hipLaunchKernelGGL(backtrack_grp_kernel<8>, dim3((n_order + 7) / 8), dim3(64), 0, st, a, order, n_order);    /* eight tiles per wave */
	} else if (group == 32 && order != nullptr) {
		if (n_order <= 0) return hipSuccess;
		hipLaunchKernelGGL(backtrack_grp_kernel<32>, dim3((n_order + 1) / 2), dim3(64), 0, st, a, order, n_order);   /* two tiles per wave */
	} else {
		const int nb = order != nullptr ? n_order : a.n_tiles;
		if (nb <= 0) return hipSuccess;
		hipLaunchKernelGGL(backtrack_kernel, dim3(nb), dim3(64), 0, st, a, order, nb);   /* one wave per tile */
	}
	return hipGetLastError();
}

hipError_t launch_finalize(const TileOut *tout, const TilePlan *plan, uint64_t *dst_off, ResultRec *res,
		BatchSummary *sum, int32_t *counters, int n_tiles, uint64_t dense_cap, hipStream_t st) {
	hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, tout, plan, dst_off, res, sum, counters,
			n_tiles, (unsigned long long) dense_cap);
	return hipGetLastError();
}

hipError_t launch_compact(const int32_t *regions, const TileRun *trun, const TileOut *tout,
		const uint64_t *dst_off, uint32_t *dense, int n_tiles, uint64_t dense_cap, hipStream_t st) {
	if (n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(compact_ops_kernel, dim3(n_tiles), dim3(256), 0, st, regions, trun, tout, dst_off, dense, n_tiles,
			(unsigned long long) dense_cap);
	return hipGetLastError();
}

}  // namespace cvx
