/*
 * cvx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for ngmlr's
 * convex-gap banded Smith-Waterman.  gfx950 only: no other target, no dual paths.
 *
 *   plan_kernel       per-tile corridor analysis (AlignmentMatrixFast::prepare,
 *                     reference src/AlignmentMatrixFast.cpp:30-60, plus what the
 *                     anti-diagonal schedule needs: ring size, first/last diagonal)
 *   fill_ring_kernel  forward fill (fwdFillMatrixSSESimple == scalar recurrence of
 *                     reference src/ConvexAlignFast.cpp:606-774), anti-diagonal
 *                     wavefront, one wave (or NW lock-stepped waves) per tile
 *   backtrack_kernel  revBacktrack + validPath (src/ConvexAlignFast.cpp:335-432,
 *                     src/AlignmentMatrixFast.cpp:213-220)
 *   compact_ops_kernel  gathers the per-tile op regions into one dense arena
 *
 * Parallel scheme of the fill (see DESIGN.md for the derivation).  Cells on one
 * anti-diagonal r = x + y are independent: (x,y) needs left (x-1,y) and up (x,y-1)
 * from r-1 and diag (x-1,y-1) from r-2.  A wave keeps read ROWS in a ring of
 * N = 64*M*NW slots, row y in slot y mod N, M consecutive slots per lane.  Per step
 * every slot advances its row by one column, so "left" is the slot's own previous
 * value (a register), "up" is the previous slot's value (a register for M-1 of the M
 * slots, one DPP wave_ror:1 for the lane boundary) and "diag" is the up value the
 * slot saw one step earlier.  Row state never touches LDS or HBM; the only HBM
 * traffic is one reference character per cell (L1/L2 resident) in, and the 2-bit
 * direction codes out, N contiguous dwords per 16 steps.
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
	return __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, false);
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

__global__ void __launch_bounds__(256)
plan_kernel(const int2 *rows, const TileIn *tin, TilePlan *plan, int n_tiles, unsigned long long max_matrix_mb) {
	const int t = blockIdx.x;
	if (t >= n_tiles) return;
	const TileIn ti = tin[t];
	const int2 *r = rows + ti.row_off;
	const int H = ti.H, W = ti.W;

	__shared__ unsigned long long s_cells, s_active;
	__shared__ int s_need, s_flags, s_maxlen, s_rend;
	if (threadIdx.x == 0) { s_cells = 0; s_active = 0; s_need = 1; s_flags = 0; s_maxlen = 0; s_rend = -0x7fffffff; }
	__syncthreads();

	unsigned long long cells = 0, active = 0;
	int need = 1, flags = 0, maxlen = 0, rendmax = -0x7fffffff;
	for (int y = threadIdx.x; y < H; y += blockDim.x) {
		const int2 ol = r[y];
		int gs, ge;
		row_span(ol, W, y, gs, ge);
		cells += (unsigned long long) (long long) ol.y;
		active += (unsigned long long) (ge - gs);
		if (ol.y > maxlen) maxlen = ol.y;
		if (ge > rendmax) rendmax = ge;
		if (y > 0) {
			int pgs, pge;
			row_span(r[y - 1], W, y - 1, pgs, pge);
			if (gs <= pgs) flags |= kPlanIrregular;  /* ring schedule needs increasing row starts */
		}
		/* first row y' > y that starts at or after ge + margin (gs is increasing) */
		const int lim = ge + kSwitchMargin;
		int lo = y + 1, hi = H;
		while (lo < hi) {
			const int mid = (lo + hi) >> 1;
			int mgs, mge;
			row_span(r[mid], W, mid, mgs, mge);
			if (mgs >= lim) hi = mid; else lo = mid + 1;
		}
		const int n = lo - y + 1;
		if (n > need) need = n;
	}
	atomicAdd(&s_cells, cells);
	atomicAdd(&s_active, active);
	atomicMax(&s_need, need);
	atomicOr(&s_flags, flags);
	atomicMax(&s_maxlen, maxlen);
	atomicMax(&s_rend, rendmax);
	__syncthreads();

	if (threadIdx.x == 0) {
		TilePlan p;
		p.cells = s_cells;
		p.active = s_active;
		p.need = s_need;
		int f = s_flags;
		int r0 = 0, rend = 0;
		if (H > 0) {
			int gs, ge;
			row_span(r[0], W, 0, gs, ge);
			r0 = gs;
			rend = s_rend;
		}
		if (H <= 0 || s_active == 0) f |= kPlanEmpty;
		/* src/AlignmentMatrixFast.cpp:45: (ulong)(matrixSize / 1000.0f / 1000.0f) < maxMatrixSizeMB */
		const float mb = (float) s_cells / 1000.0f / 1000.0f;
		if (!((unsigned long long) mb < max_matrix_mb)) f |= kPlanTooLarge;
		if (H > 32767 || s_maxlen > 32767) f |= kPlanWrap16;
		p.r0 = r0;
		p.rend = rend;
		p.flags = f;
		plan[t] = p;
	}
}

/* ------------------------------------------------------------------ fill */

template <int M, int NW, bool WRAP>
struct Ring {
	static constexpr int N = 64 * M * NW;

	/* per-slot state, all in VGPRs (static indexing only) */
	float S[M];      /* score of the slot's latest cell (0 while the row is idle) */
	float Hc[M];     /* left candidate that cell offers to the next column          */
	float V[M];      /* up candidate it offers to the next row                      */
	float dg[M];     /* diagonal score for the slot's next cell                     */
	int drun[M];     /* deletion run of the latest cell (0 unless direction D)      */
	int irun[M];     /* insertion run (0 unless direction I)                        */
	int cnt[M];      /* next column index inside the row (negative: not started)    */
	int len[M];      /* row length after clipping to [0,W)                          */
	int qch[M];      /* read character of the row                                   */
	int y[M];        /* read row held by the slot                                   */
	unsigned xa[M];  /* seq-arena offset of the reference character for the group   */
	unsigned dacc[M];/* 2-bit direction codes of the current 16-step block          */
	float best[M];
	int best_r[M], best_y[M];
};

template <int M, int NW, bool WRAP>
__global__ void __launch_bounds__(64 * NW)
fill_ring_kernel(const FillArgs a) {
	constexpr int N = 64 * M * NW;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const float mat = a.sp.mat, mis = a.sp.mis, go = a.sp.go;
	const float gext = a.sp.ge, gem = a.sp.gem, decay = a.sp.decay;

	__shared__ int s_tile;
	__shared__ float s_xf[2][NW > 1 ? NW : 1][2];
	__shared__ int s_xi[2][NW > 1 ? NW : 1];
	__shared__ float s_rbest[NW > 1 ? NW : 1];
	__shared__ int s_ry[NW > 1 ? NW : 1], s_rx[NW > 1 ? NW : 1];

	for (;;) {
		int qi;
		if (NW == 1) {
			qi = 0;
			if (lane == 0) qi = atomicAdd(a.queue_head, 1);
			qi = __builtin_amdgcn_readfirstlane(qi);
		} else {
			__syncthreads();
			if (tid == 0) s_tile = atomicAdd(a.queue_head, 1);
			__syncthreads();
			qi = s_tile;
		}
		if (qi >= a.list_n) break;
		const int t = a.list[qi];
		const TileIn ti = a.tin[t];
		const TileRun tr = a.trun[t];
		const int2 *rows = reinterpret_cast<const int2 *>(a.rows) + ti.row_off;
		const uint8_t *seq = a.seq;
		const int H = ti.H, W = ti.W;
		uint32_t *dirs = a.dirs + tr.dir_off;

		Ring<M, NW, WRAP> s;

		/* (re)bind slot j to its row s.y[j]; rnext = index of the next step */
		auto bind_row = [&](int j, int rnext) {
			const int yy = s.y[j];
			if (yy < H) {
				const int2 ol = rows[yy];
				long long lo = ol.x > 0 ? ol.x : 0;
				long long hi = (long long) ol.x + (long long) ol.y;
				if (hi > W) hi = W;
				if (hi < lo) hi = lo;
				s.cnt[j] = rnext - (yy + (int) lo);
				s.len[j] = (int) (hi - lo);
				s.qch[j] = seq[ti.qry_off + (unsigned) yy];
				s.xa[j] = ti.ref_off + (unsigned) (rnext - yy);
			} else {
				s.cnt[j] = -(1 << 30);
				s.len[j] = 0;
				s.qch[j] = 0;
				s.xa[j] = ti.ref_off;
			}
		};

#pragma unroll
		for (int j = 0; j < M; ++j) {
			s.y[j] = tid * M + j;
			s.S[j] = 0.0f; s.Hc[j] = go; s.V[j] = go; s.dg[j] = 0.0f;
			s.drun[j] = 0; s.irun[j] = 0; s.dacc[j] = 0u;
			s.best[j] = -1.0f; s.best_r[j] = 0; s.best_y[j] = 0;
			bind_row(j, tr.r0);
		}

		const int ngroups = (tr.nsteps + 3) >> 2;
		int r = tr.r0;
		for (int g = 0; g < ngroups; ++g) {
			/* one unaligned dword = the 4 reference characters of this group */
			unsigned cw[M];
#pragma unroll
			for (int j = 0; j < M; ++j) {
				cw[j] = *reinterpret_cast<const unsigned *>(seq + s.xa[j]);
				s.xa[j] += 4u;
			}
			const int shbase = (g & 3) * 8;

#pragma unroll
			for (int i = 0; i < 4; ++i) {
				/* the lane boundary: previous lane's last slot, values of step r-1 */
				float uV0 = rot1_f(s.V[M - 1]);
				float uS0 = rot1_f(s.S[M - 1]);
				int uI0 = rot1_i(s.irun[M - 1]);
				if (NW > 1) {
					const int par = (r & 1);
					if (lane == 63) {
						s_xf[par][wave][0] = s.V[M - 1];
						s_xf[par][wave][1] = s.S[M - 1];
						s_xi[par][wave] = s.irun[M - 1];
					}
					__syncthreads();
					if (lane == 0) {
						const int pw = (wave + NW - 1) % NW;
						uV0 = s_xf[par][pw][0];
						uS0 = s_xf[par][pw][1];
						uI0 = s_xi[par][pw];
					}
				}
				const int sh = shbase + 2 * i;

#pragma unroll
				for (int j = M - 1; j >= 0; --j) {
					const float uV = (j > 0) ? s.V[j > 0 ? j - 1 : 0] : uV0;
					const float uS = (j > 0) ? s.S[j > 0 ? j - 1 : 0] : uS0;
					const int uI = (j > 0) ? s.irun[j > 0 ? j - 1 : 0] : uI0;

					const int refc = (int) ((cw[j] >> (8 * i)) & 0xffu);
					const bool eq = (refc == s.qch[j]);
					const float diag_cell = s.dg[j] + (eq ? mat : mis);
					const float up_cell = uV;
					const float left_cell = s.Hc[j];
					const float mx = fmaxf(fmaxf(fmaxf(left_cell, 0.0f), diag_cell), up_cell);

					const bool eL = (mx == left_cell);
					const bool eU = (mx == up_cell);
					const bool eG = (mx == diag_cell);
					const bool isDl = s.drun[j] > 0;
					const bool isIu = uI > 0;
					/* priority: del-extend > ins-extend > diag > del-open > ins-open > stop
					 * (src/ConvexAlignFast.cpp:703-738) */
					const bool c2 = isIu && eU;
					const bool newD = eL && (isDl || !(c2 || eG));
					const bool newI = !newD && eU && (isIu || !eG);
					const bool newG = !newD && !newI && eG;
					int nd, ni;
					if (WRAP) {
						/* indelRun is a short in the reference (src/AlignmentMatrixFast.h:43) */
						nd = newD ? (isDl ? (int) (short) (s.drun[j] + 1) : 1) : 0;
						ni = newI ? (isIu ? (int) (short) (uI + 1) : 1) : 0;
					} else {
						nd = newD ? s.drun[j] + 1 : 0;
						ni = newI ? uI + 1 : 0;
					}
					const unsigned code = newD ? 2u : (newI ? 1u : (newG ? 3u : 0u));

					const bool act = (unsigned) s.cnt[j] < (unsigned) s.len[j];
					s.dg[j] = uS;
					if (act) {
						const int run = WRAP ? (newD ? nd : ni) : (nd | ni);
						const float pen = fminf(gem, gext + (float) run * decay);
						const float E = (mx == 0.0f) ? 0.0f : mx + pen;
						const float O = mx + go;
						s.S[j] = mx;
						s.drun[j] = nd;
						s.irun[j] = ni;
						s.V[j] = newI ? E : O;
						s.Hc[j] = newD ? E : O;
						if (mx > s.best[j]) { s.best[j] = mx; s.best_r[j] = r; }
						s.dacc[j] |= code << sh;
					}
					if (s.cnt[j] == s.len[j]) {
						/* first step after the row's last cell.  The slot below (processed
						 * earlier in this step) has just consumed that last cell as its "up"
						 * and keeps its score as next step's "diag"; from now on this slot is
						 * outside the corridor (empty element, src/AlignmentMatrixFast.h:49-53). */
						s.S[j] = 0.0f; s.drun[j] = 0; s.irun[j] = 0;
						s.V[j] = go; s.Hc[j] = go;
					}
					s.cnt[j] += 1;
				}
				r += 1;
			}

			/* hand finished slots to their next row (y + N) */
#pragma unroll
			for (int j = 0; j < M; ++j) {
				if (s.cnt[j] > s.len[j] && s.y[j] < H) {   /* ended AND already reset */
					if (s.best_r[j] >= r - s.cnt[j]) s.best_y[j] = s.y[j];
					s.y[j] += N;
					bind_row(j, r);
				}
			}

			if ((g & 3) == 3) {
				uint32_t *d = dirs + (size_t) (g >> 2) * N + (size_t) tid * M;
#pragma unroll
				for (int j = 0; j < M; ++j) { d[j] = s.dacc[j]; s.dacc[j] = 0u; }
			}
		}
		if ((ngroups & 3) != 0) {
			uint32_t *d = dirs + (size_t) (ngroups >> 2) * N + (size_t) tid * M;
#pragma unroll
			for (int j = 0; j < M; ++j) d[j] = s.dacc[j];
		}

		/* argmax with the reference's tie-break: first strict maximum in (y, x) order
		 * (src/ConvexAlignFast.cpp:758-763 / :1165-1170) */
		float b = -1.0f;
		int by = 0x7fffffff, bx = 0x7fffffff;
#pragma unroll
		for (int j = 0; j < M; ++j) {
			if (s.y[j] < H && s.best_r[j] >= r - s.cnt[j]) s.best_y[j] = s.y[j];
			const float v = s.best[j];
			const int vy = s.best_y[j];
			const int vx = s.best_r[j] - s.best_y[j];
			if (v > -1.0f) {
				if (v > b || (v == b && (vy < by || (vy == by && vx < bx)))) { b = v; by = vy; bx = vx; }
			}
		}
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			const float ob = __shfl_xor(b, off, 64);
			const int oy = __shfl_xor(by, off, 64);
			const int ox = __shfl_xor(bx, off, 64);
			if (ob > b || (ob == b && (oy < by || (oy == by && ox < bx)))) { b = ob; by = oy; bx = ox; }
		}
		if (NW > 1) {
			__syncthreads();
			if (lane == 0) { s_rbest[wave] = b; s_ry[wave] = by; s_rx[wave] = bx; }
			__syncthreads();
			if (tid == 0) {
				for (int w = 1; w < NW; ++w) {
					const float ob = s_rbest[w];
					const int oy = s_ry[w], ox = s_rx[w];
					if (ob > b || (ob == b && (oy < by || (oy == by && ox < bx)))) { b = ob; by = oy; bx = ox; }
				}
			}
		}
		if (tid == 0) {
			TileOut o;
			o.score = b;
			o.status = (b > -1.0f) ? 0 : 5;
			o.best_x = (b > -1.0f) ? bx : 0;
			o.best_y = (b > -1.0f) ? by : 0;
			o.ref_position = 0; o.qstart = 0; o.qend = 0; o.n_ops = 0; o.ops_first = 0; o.pad = 0;
			a.tout[t] = o;
		}
	}
}

/* ------------------------------------------------------------------ backtrack */

/* validPath, src/AlignmentMatrixFast.cpp:213-220: float arithmetic, int truncation,
 * no contraction. */
CVX_DEV bool valid_path(const int2 ol, int x) {
	const int width = ol.y;
	const int minC = (int) ((float) ol.x + 0.1f * (float) width);
	const int maxC = (int) ((float) (minC + width) - 0.1f * (float) width);
	return x > minC && x < maxC;
}

__global__ void __launch_bounds__(64)
backtrack_kernel(const BacktrackArgs a) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= a.n_tiles) return;
	const TileRun tr = a.trun[t];
	if (tr.skip) return;
	const TileIn ti = a.tin[t];
	TileOut o = a.tout[t];
	if (o.status != 0) return;
	const int H = ti.H;
	const int N = tr.ring;
	const int2 *rows = reinterpret_cast<const int2 *>(a.rows) + ti.row_off;
	const uint32_t *dirs = a.dirs + tr.dir_off;
	const uint8_t *ref = a.seq + ti.ref_off;
	const uint8_t *qry = a.seq + ti.qry_off;
	int *ops = a.ops + tr.ops_off;

	/* src/ConvexAlignFast.cpp:338 */
	if (o.best_y <= 0) { o.status = 1; a.tout[t] = o; return; }

	const int qend = (H - o.best_y) - 1;
	int idx = tr.ops_cap - 1;
	int elem = 4;          /* CIGAR_S: the trailing clip is tracked but not stored */
	int elem_len = qend;
	int consumed = qend;
	int x = o.best_x, y = o.best_y;
	int status = 0;
	for (;;) {
		/* getDirection, src/AlignmentMatrixFast.cpp:185-195 */
		if (y < 0 || x < 0) break;
		const int2 ol = rows[y];
		if (x < ol.x || x >= ol.x + ol.y) break;
		const int tt = x + y - tr.r0;
		if (tt < 0) break;
		const uint32_t w = dirs[(size_t) (tt >> 4) * N + (y % N)];
		const unsigned code = (w >> (2 * (tt & 15))) & 3u;
		if (code == 0u) break;
		if (!valid_path(ol, x)) { status = 2; break; }
		int cur;
		if (code == 3u) {
			cur = (ref[x] == qry[y]) ? 7 : 8;
			x -= 1; y -= 1; consumed += 1;
		} else if (code == 1u) {
			cur = 1; y -= 1; consumed += 1;
		} else {
			cur = 2; x -= 1;
		}
		if (cur == elem) {
			elem_len += 1;
		} else {
			if (elem != 4) ops[idx--] = (elem_len << 4) | elem;
			elem = cur;
			elem_len = 1;
		}
	}
	if (status == 0) {
		if (elem != 4) ops[idx--] = (elem_len << 4) | elem;
		consumed += (y + 1);
		o.ref_position = x + 1;
		o.qstart = y + 1;
		o.qend = qend;
		o.ops_first = idx + 1;
		o.n_ops = tr.ops_cap - 1 - idx;
		if (H != consumed) status = 3;
	}
	o.status = status;
	a.tout[t] = o;
}

/* dense[dst_off[t] .. +n_ops) = region of tile t */
__global__ void __launch_bounds__(256)
compact_ops_kernel(const int32_t *regions, const TileRun *trun, const TileOut *tout,
		const uint64_t *dst_off, uint32_t *dense, int n_tiles) {
	const int t = blockIdx.x;
	if (t >= n_tiles) return;
	const TileOut o = tout[t];
	if (o.status != 0 || o.n_ops <= 0) return;
	const int32_t *src = regions + trun[t].ops_off + o.ops_first;
	uint32_t *dst = dense + dst_off[t];
	for (int i = threadIdx.x; i < o.n_ops; i += blockDim.x) dst[i] = (uint32_t) src[i];
}

/* ------------------------------------------------------------------ launchers */

template <int M, int NW, bool WRAP>
static hipError_t launch_fill_t(const FillArgs &a, int grid, hipStream_t st) {
	hipLaunchKernelGGL((fill_ring_kernel<M, NW, WRAP>), dim3(grid), dim3(64 * NW), 0, st, a);
	return hipGetLastError();
}

template <int M, int NW>
static hipError_t launch_fill_w(const FillArgs &a, bool wrap, int grid, hipStream_t st) {
	return wrap ? launch_fill_t<M, NW, true>(a, grid, st) : launch_fill_t<M, NW, false>(a, grid, st);
}

hipError_t launch_fill(int m, int nw, bool wrap, const FillArgs &a, int grid, hipStream_t st) {
	if (nw == 1) {
		switch (m) {
		case 1: return launch_fill_w<1, 1>(a, wrap, grid, st);
		case 2: return launch_fill_w<2, 1>(a, wrap, grid, st);
		case 3: return launch_fill_w<3, 1>(a, wrap, grid, st);
		case 4: return launch_fill_w<4, 1>(a, wrap, grid, st);
		case 5: return launch_fill_w<5, 1>(a, wrap, grid, st);
		case 6: return launch_fill_w<6, 1>(a, wrap, grid, st);
		case 8: return launch_fill_w<8, 1>(a, wrap, grid, st);
		default: return hipErrorInvalidValue;
		}
	}
	if (m != 4) return hipErrorInvalidValue;
	switch (nw) {
	case 2: return launch_fill_w<4, 2>(a, wrap, grid, st);
	case 4: return launch_fill_w<4, 4>(a, wrap, grid, st);
	case 8: return launch_fill_w<4, 8>(a, wrap, grid, st);
	case 16: return launch_fill_w<4, 16>(a, wrap, grid, st);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_plan(const RowDesc *rows, const TileIn *tin, TilePlan *plan, int n_tiles,
		unsigned long long max_matrix_mb, hipStream_t st) {
	if (n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(plan_kernel, dim3(n_tiles), dim3(256), 0, st,
			reinterpret_cast<const int2 *>(rows), tin, plan, n_tiles, max_matrix_mb);
	return hipGetLastError();
}

hipError_t launch_backtrack(const BacktrackArgs &a, hipStream_t st) {
	if (a.n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(backtrack_kernel, dim3((a.n_tiles + 63) / 64), dim3(64), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_compact(const int32_t *regions, const TileRun *trun, const TileOut *tout,
		const uint64_t *dst_off, uint32_t *dense, int n_tiles, hipStream_t st) {
	if (n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(compact_ops_kernel, dim3(n_tiles), dim3(256), 0, st, regions, trun, tout, dst_off, dense, n_tiles);
	return hipGetLastError();
}

}  // namespace cvx
