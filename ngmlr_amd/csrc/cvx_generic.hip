/*
 * cvx_generic.hip -- the catch-all forward fill (gfx950).
 *
 * fill_ring_kernel (cvx_kernels.hip) keeps row state in registers and therefore has a
 * largest ring (4096 live rows, corridors up to ~8 k wide) and needs increasing row
 * starts.  Everything else the reference would accept -- corridors the retry loop has
 * widened past that (reference src/AlignmentBuffer.cpp:291-294: corridor*multiplier up
 * to 2*refSeqLen), or an arbitrary caller-made CorridorLine[] -- goes through this kernel:
 * the same anti-diagonal ring schedule, same recurrence (src/ConvexAlignFast.cpp:651-763),
 * same direction layout, but slot state lives in a per-tile global-memory scratch (served
 * by L1/L2), one workgroup per tile, one barrier per anti-diagonal.  It is slow per cell
 * and exists for completeness: no input is ever computed on the CPU.
 *
 * Ring size Ng: >= the plan's `need` for corridors with increasing row starts (slots are
 * re-bound to row y+Ng as in the fast kernel); for irregular corridors Ng >= H, i.e. one
 * slot per row and nothing is ever re-bound, which removes every ordering assumption.
 * indelRun is kept as the reference's short (wrap included).
 *
 * SSE = true is the second job of this kernel: scoring parameters outside the regime in which
 * the reference's SSE path equals the scalar recurrence (gap_open + gap_ext_min >= mismatch,
 * SURVEY.md Appendix A).  There the reference's results are those of fwdFillMatrixSSESimple
 * itself (src/ConvexAlignFast.cpp:914-1287), restated here cell by cell:
 *   - cells before the row's last 12 follow the SSE rules: the up/diag candidates first, "ins-extend"
 *     whenever the cell above carries a run (whatever its direction, :1084-1091), then the serial
 *     left fix-up with `left.indelRun > 0` untested for direction (:1127-1141);
 *   - the last <= 12 cells of every row are recomputed by the scalar rules, starting from the SSE
 *     value left of them (:1179-1277); rows below see the recomputed values;
 *   - the running maximum sees the SSE value of every cell the 4-wide loop covered (which overlaps
 *     the recomputed tail by 8..11 cells) and then the recomputed values, in that order.
 * A slot therefore carries two left-to-right chains, A (SSE rules, cells [0, sse_n)) and B (scalar
 * rules, cells [len - 12, len)), and publishes A before the tail and B inside it.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvx_types.h"
#include "cvx_launch.h"

namespace cvx {

namespace {

struct Slot {              /* private per-slot state, one struct per slot in scratch */
	float Hc, dg, best;
	int drun, cnt, len, qch, y, best_r, best_y;
	unsigned accA, accB;
	/* SSE variant only: chain A = the 4-wide loop's value of the previous cell (score, the left
	 * candidate it offers, its run and whether it is a deletion), and the best recomputed (chain B)
	 * cell of the current row, merged into best at the end of the row */
	float aS, aHc, bbest;
	int arun, aisD, bbest_r;
};

struct Pub {               /* what a slot shows its lower neighbour, double buffered */
	float S, V;
	int irun;              /* run if the cell is an insertion, else 0 */
	int run;               /* run of the cell whatever its direction (SSE variant: `up.indelRun > 0`) */
};

}  // namespace

template <bool SSE>
__global__ void __launch_bounds__(1024)
fill_generic_kernel(const FillArgs a, uint8_t *scratch, const uint64_t *scratch_off) {
	const int qi = blockIdx.x;
	if (qi >= a.list_n) return;
	const int t = a.list[qi];
	const TileIn ti = a.tin[t];
	const TileRun tr = a.trun[t];
	const RowView rows = row_view(a.rsrc[t], a.rows, ti.row_off);      /* closed-form corridors are evaluated in registers */
	const uint8_t *ref = a.seq + ti.ref_off;
	const uint8_t *qry = a.seq + ti.qry_off;
	const int H = ti.H, W = ti.W;
	const int Ng = tr.ring;
	uint32_t *dirs = a.dirs + tr.dir_off;
	const float mat = a.sp.mat, mis = a.sp.mis, go = a.sp.go;
	const float gext = a.sp.ge, gem = a.sp.gem, decay = a.sp.decay;
	const int tid = threadIdx.x, T = blockDim.x;

	uint8_t *base = scratch + scratch_off[qi];
	Slot *slot = reinterpret_cast<Slot *>(base);
	Pub *pub0 = reinterpret_cast<Pub *>(base + (size_t) Ng * sizeof(Slot));
	Pub *pub1 = pub0 + Ng;

	auto bind = [&](Slot &s, int yy, int rnext) {
		s.y = yy;
		if (yy < H) {
			const RowDesc2 ol = row_at(rows, yy);
			long long lo = ol.x > 0 ? ol.x : 0;
			long long hi = (long long) ol.x + (long long) ol.y;
			if (hi > W) hi = W;
			if (hi < lo) hi = lo;
			s.cnt = rnext - (yy + (int) lo);
			s.len = (int) (hi - lo);
			s.qch = qry[yy];
		} else {
			s.cnt = -(1 << 30);
			s.len = 0;
			s.qch = 0;
		}
	};

	for (int s = tid; s < Ng; s += T) {
		Slot st;
		st.Hc = go; st.dg = 0.0f; st.best = -1.0f;
		st.drun = 0; st.best_r = 0; st.best_y = 0; st.accA = 0u; st.accB = 0u;
		st.aS = 0.0f; st.aHc = go; st.arun = 0; st.aisD = 0; st.bbest = -1.0f; st.bbest_r = 0;
		bind(st, s, tr.r0);
		slot[s] = st;
		Pub p; p.S = 0.0f; p.V = go; p.irun = 0; p.run = 0;
		pub0[s] = p;
		pub1[s] = p;
	}
	__syncthreads();

	int r = tr.r0;
	for (int step = 0; step < tr.nsteps; ++step, ++r) {
		const Pub *prev = (step & 1) ? pub1 : pub0;
		Pub *next = (step & 1) ? pub0 : pub1;
		const bool flush = ((step & 31) == 31) || (step == tr.nsteps - 1);
		for (int s = tid; s < Ng; s += T) {
			Slot st = slot[s];
			const Pub up = prev[s == 0 ? Ng - 1 : s - 1];
			const Pub me = prev[s];
			const bool act = (unsigned) st.cnt < (unsigned) st.len;
			Pub out;
			unsigned code = 0u;
			if (act && !SSE) {
				const int x = r - st.y;
				const bool eq = ((int) ref[x] == st.qch);
				const float diag_cell = st.dg + (eq ? mat : mis);
				const float up_cell = up.V;
				const float left_cell = st.Hc;
				const float mx = fmaxf(fmaxf(fmaxf(left_cell, diag_cell), up_cell), 0.0f);
				const bool isDl = st.drun > 0, isIu = up.irun > 0;
				int nd = 0, ni = 0;
				if (isDl && mx == left_cell) { nd = (int) (short) (st.drun + 1); code = 2u; }
				else if (isIu && mx == up_cell) { ni = (int) (short) (up.irun + 1); code = 1u; }
				else if (mx == diag_cell) { code = 3u; }
				else if (mx == left_cell) { nd = 1; code = 2u; }
				else if (mx == up_cell) { ni = 1; code = 1u; }
				const int run = (code == 2u) ? nd : ni;
				const float pen = fminf(gem, gext + (float) run * decay);
				const float E = (mx == 0.0f) ? 0.0f : mx + pen;
				const float O = mx + go;
				out.S = mx;
				out.irun = ni;
				out.run = run;
				out.V = (code == 1u) ? E : O;
				st.Hc = (code == 2u) ? E : O;
				st.drun = nd;
				if (mx > st.best) { st.best = mx; st.best_r = r; st.best_y = st.y; }
			} else if (act) {
				/* the reference's SSE path, cell by cell (see the header) */
				const int x = r - st.y;
				const int c = st.cnt;                                    /* column index inside the row */
				const int sse_n = st.len > 4 ? ((st.len - 4 + 3) / 4) * 4 : 0;   /* cells the 4-wide loop covers (:949) */
				const int tail0 = st.len > 12 ? st.len - 12 : 0;        /* first recomputed cell (:1179) */
				const bool eq = ((int) ref[x] == st.qch);
				const float diag_cell = st.dg + (eq ? mat : mis);
				const float up_cell = up.V;                              /* (:1008-1016): extension only off an insertion */
				/* what the cell passes on: score, direction code, run; E / O as in the scalar kernel */
				float fS = 0.0f; unsigned fcode = 0u; int frun = 0;
				if (c < sse_n) {
					/* chain A.  left = chain A's previous cell (the empty element at the row start) */
					const float aS_l = (c == 0) ? 0.0f : st.aS;
					const float aHc_l = (c == 0) ? go : st.aHc;
					const int arun_l = (c == 0) ? 0 : st.arun;
					const float t = (up_cell > diag_cell) ? up_cell : diag_cell;       /* _mm_max_ps (:1040) */
					float sc = (0.0f > t) ? 0.0f : t;
					unsigned code_a = 0u; int run_a = 0;
					if (sc == up_cell) { code_a = 1u; run_a = 1; }                     /* :1049-1055 */
					if (sc == diag_cell) { code_a = 3u; run_a = 0; }                   /* :1068-1076 */
					if (up.run > 0 && sc == up_cell) { code_a = 1u; run_a = (int) (short) (int) ((float) up.run + 1.0f); }   /* :1084-1091 */
					const float left_cell = aHc_l;
					if (left_cell >= sc) {                                             /* :1127-1141 */
						if (arun_l > 0) { sc = left_cell; code_a = 2u; run_a = (int) (short) (arun_l + 1); }
						else if (left_cell > sc || code_a == 0u || (code_a == 1u && !(up.run > 0))) { sc = left_cell; code_a = 2u; run_a = 1; }
					}
					(void) aS_l;
					if (sc > st.best) { st.best = sc; st.best_r = r; st.best_y = st.y; }   /* :1165-1170 */
					const float pen_a = fminf(gem, gext + (float) run_a * decay);
					st.aS = sc;
					st.aHc = (code_a == 2u) ? ((sc == 0.0f) ? 0.0f : sc + pen_a) : sc + go;
					st.arun = run_a;
					st.aisD = (code_a == 2u);
					fS = sc; fcode = code_a; frun = run_a;
				}
				if (c >= tail0) {
					/* chain B: the scalar rules; its first cell takes chain A's cell to the left (or the
					 * empty element at the row start), afterwards its own */
					/* st.Hc / st.drun: chain B's previous cell, or -- at the first recomputed cell -- chain A's
					 * cell to its left (mirrored below on every step before the tail), or the empty element */
					const float left_cell = st.Hc;
					const int del_run = st.drun;
					const float mx = fmaxf(fmaxf(fmaxf(left_cell, diag_cell), up_cell), 0.0f);
					const bool isDl = del_run > 0, isIu = up.irun > 0;
					int nd = 0, ni = 0; unsigned code_b = 0u;
					if (isDl && mx == left_cell) { nd = (int) (short) (del_run + 1); code_b = 2u; }
					else if (isIu && mx == up_cell) { ni = (int) (short) (up.irun + 1); code_b = 1u; }
					else if (mx == diag_cell) { code_b = 3u; }
					else if (mx == left_cell) { nd = 1; code_b = 2u; }
					else if (mx == up_cell) { ni = 1; code_b = 1u; }
					const int run_b = (code_b == 2u) ? nd : ni;
					const float pen_b = fminf(gem, gext + (float) run_b * decay);
					st.Hc = (code_b == 2u) ? ((mx == 0.0f) ? 0.0f : mx + pen_b) : mx + go;
					st.drun = nd;
					if (mx > st.bbest) { st.bbest = mx; st.bbest_r = r; }
					fS = mx; fcode = code_b; frun = run_b;
				} else {
					/* before the tail: chain B's "previous cell" is chain A's cell (what cell tail0 reads) */
					st.Hc = st.aHc;
					st.drun = st.aisD ? st.arun : 0;
				}
				code = fcode;
				const float pen = fminf(gem, gext + (float) frun * decay);
				out.S = fS;
				out.run = frun;
				out.irun = (fcode == 1u) ? frun : 0;
				out.V = (fcode == 1u) ? ((fS == 0.0f) ? 0.0f : fS + pen) : fS + go;
			} else {
				/* outside the row: the empty element (src/AlignmentMatrixFast.h:49-53) */
				out.S = 0.0f; out.V = go; out.irun = 0; out.run = 0;
				st.Hc = go; st.drun = 0;
				if (SSE) { st.aS = 0.0f; st.aHc = go; st.arun = 0; st.aisD = 0; }
			}
			(void) me;
			st.dg = up.S;
			st.cnt += 1;
			st.accA = (st.accA << 1) | ((code == 1u || code == 2u) ? 1u : 0u);   /* plane 0: I or D */
			st.accB = (st.accB << 1) | ((code & 1u));                            /* plane 1: I or diagonal */
			if (flush) {
				const int done = (step & 31) + 1;
				uint32_t *d = dirs + ((size_t) (step >> 5) * Ng + s) * 2;
				d[0] = st.accA << (32 - done);
				d[1] = st.accB << (32 - done);
			}
			/* a finished row is cleared by the `else` branch above on the step after its
			 * last cell; hand the slot to row y+Ng once that has happened */
			if (SSE && st.cnt >= st.len && st.bbest > -1.0f) {
				/* end of the row: the recomputed cells come after every SSE-loop cell of the row in the
				 * reference's order, so they only win with a strictly larger score */
				if (st.bbest > st.best) { st.best = st.bbest; st.best_r = st.bbest_r; st.best_y = st.y; }
				st.bbest = -1.0f;
			}
			if (st.cnt > st.len && st.cnt < (1 << 29)) bind(st, st.y + Ng, r + 1);
			slot[s] = st;
			next[s] = out;
		}
		__syncthreads();
	}

	/* argmax, first strict maximum in (y, x) order (src/ConvexAlignFast.cpp:758-763) */
	__shared__ float s_b[1024];
	__shared__ int s_y[1024], s_x[1024];
	float b = -1.0f;
	int by = 0x7fffffff, bx = 0x7fffffff;
	for (int s = tid; s < Ng; s += T) {
		const Slot st = slot[s];
		if (st.best > -1.0f) {
			const int vy = st.best_y, vx = st.best_r - st.best_y;
			if (st.best > b || (st.best == b && (vy < by || (vy == by && vx < bx)))) { b = st.best; by = vy; bx = vx; }
		}
	}
	s_b[tid] = b; s_y[tid] = by; s_x[tid] = bx;
	__syncthreads();
	for (int off = T >> 1; off >= 1; off >>= 1) {
		if (tid < off) {
			const float ob = s_b[tid + off];
			const int oy = s_y[tid + off], ox = s_x[tid + off];
			if (ob > s_b[tid] || (ob == s_b[tid] && (oy < s_y[tid] || (oy == s_y[tid] && ox < s_x[tid])))) {
				s_b[tid] = ob; s_y[tid] = oy; s_x[tid] = ox;
			}
		}
		__syncthreads();
	}
	if (tid == 0) {
		TileOut o;
		b = s_b[0];
		o.score = b;
		o.status = (b > -1.0f) ? 0 : 5;
		o.best_x = (b > -1.0f) ? s_x[0] : 0;
		o.best_y = (b > -1.0f) ? s_y[0] : 0;
		o.ref_position = 0; o.qstart = 0; o.qend = 0; o.n_ops = 0; o.ops_first = 0; o.pad = 0;
		a.tout[t] = o;
	}
}

size_t generic_scratch_bytes(int ring) {
	return (size_t) ring * (sizeof(Slot) + 2 * sizeof(Pub)) + 256;
}

hipError_t launch_fill_generic(const FillArgs &a, bool sse_variant, uint8_t *scratch, const uint64_t *scratch_off, hipStream_t st) {
	if (a.list_n <= 0) return hipSuccess;
	if (sse_variant) hipLaunchKernelGGL(fill_generic_kernel<true>, dim3(a.list_n), dim3(1024), 0, st, a, scratch, scratch_off);
	else hipLaunchKernelGGL(fill_generic_kernel<false>, dim3(a.list_n), dim3(1024), 0, st, a, scratch, scratch_off);
	return hipGetLastError();
}

}  // namespace cvx
