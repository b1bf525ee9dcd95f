/*
 * cvx_corridor.cpp -- cvx_corridor_fit: the closed form behind a CorridorLine[].
 *
 * ngmlr's binding can only hand over what SingleAlign receives: the rows its corridor builders wrote
 * (reference src/AlignmentBuffer.cpp:68-197).  Every one of those builders evaluates one expression per row,
 *     offset[y] = (int) (((float) y - d) / k - right),   length[y] = width          (or a constant offset),
 * with k = qryLen * 1.0f / refLen (:117, :141) -- both lengths are SingleAlign arguments --, d = width / 2.0f, right = 0
 * for the endpoints corridor (:118-124), d = 0 and an unknown `corridorRight` for the anchors corridor (:178-191),
 * k = 1 for the linear ones (:68-82).  A tile that travels as (kind, k, d, right, offset, width) needs no row array
 * packed, uploaded, expanded or read by any kernel (cvx_tile.corridor_kind), so the shim asks this function
 * for the form and sends the rows only when there is none.
 *
 * A form is returned only after EVERY row has been compared with the expression the device evaluates
 * (cvx::affine_row_offset, this translation unit compiled like the device code: -ffp-contract=off, no fast-math,
 * correctly rounded divide): a recognised tile is bit-identical input, not an approximation.
 * The unknown `right`: trunc(fl(q_y - r)) falls monotonically as r grows, so every row bounds r from both sides; the
 * real-number intersection of those bounds gives a first candidate, the rows that still disagree (rounding of the
 * subtraction at |q| ~ 10^4) say on which side, and a bisection over the float ordering ends it -- any float inside
 * the feasible interval generates identical rows, which one the reference held does not matter.
 */
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <vector>

#include <immintrin.h>

#include "cvx_align.h"
#include "cvx_types.h"
#include "cvx_host_logic.h"

namespace {

/* floats in their total order as integers (for the bisection) */
inline int32_t ordOf(float f) {
	int32_t i;
	memcpy(&i, &f, 4);
	return i < 0 ? (int32_t) (0x80000000u - (uint32_t) i) : i;
}
inline float floatOf(int32_t o) {
	int32_t i = o < 0 ? (int32_t) (0x80000000u - (uint32_t) o) : o;
	float f;
	memcpy(&f, &i, 4);
	return f;
}

/* first row whose offset the form does not reproduce (-1: none); *dir > 0: the form's value is too large there (right too small) */
inline int firstMismatch(const int32_t *off, int H, float d, float k, float right, int *dir) {
	for (int y = 0; y < H; ++y) {
		const int32_t got = cvx::affine_row_offset(y, d, k, right);
		if (got != off[y]) {
			if (dir) *dir = got > off[y] ? 1 : -1;
			return y;
		}
	}
	return -1;
}

/* The two passes over all rows of the anchors corridor, scalar and AVX2 (the binding runs this once per SingleAlign, 10 000
 * rows for a 10 kb read; divps / subps / cvttps2dq round exactly like their scalar forms, so both give the same answers).
 *   quotientsAndBounds  q[y] = (float) y / k, and the real-number bounds of `right` (see cvx_corridor_fit)
 *   verifyRight         first row with (int) (q[y] - right) != off[y], or -1 */
void quotientsAndBoundsScalar(const int32_t *off, int H, float k, float *q, double *loOut, double *hiOut) {
	double lo = -1.0e30, hi = 1.0e30;
	for (int y = 0; y < H; ++y) {
		const float qf = ((float) y - 0.0f) / k;
		q[y] = qf;
		const int32_t o = off[y];
		const double t = (double) qf - (double) o;
		const double a = t - (o >= 0 ? 1.0 : 0.0), b = t + (o <= 0 ? 1.0 : 0.0);
		if (a > lo) lo = a;
		if (b < hi) hi = b;
	}
	*loOut = lo; *hiOut = hi;
}

int verifyRightScalar(const int32_t *off, int H, const float *q, float right, int *dir) {
	for (int y = 0; y < H; ++y) {
		const int32_t got = (int32_t) (q[y] - right);      /* affine_row_offset(y, 0, k, right) with its quotient from the table */
		if (got != off[y]) { *dir = got > off[y] ? 1 : -1; return y; }
	}
	return -1;
}

__attribute__((target("avx2"))) void quotientsAndBoundsAvx2(const int32_t *off, int H, float k, float *q, double *loOut, double *hiOut) {
	const __m256 kv = _mm256_set1_ps(k);
	__m256i yv = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
	const __m256i eight = _mm256_set1_epi32(8);
	__m256d lo0 = _mm256_set1_pd(-1.0e30), lo1 = lo0, hi0 = _mm256_set1_pd(1.0e30), hi1 = hi0;
	const __m256d one = _mm256_set1_pd(1.0);
	const __m128i zero = _mm_setzero_si128();
	int y = 0;
	for (; y + 8 <= H; y += 8) {
		const __m256 qf = _mm256_div_ps(_mm256_cvtepi32_ps(yv), kv);
		_mm256_storeu_ps(q + y, qf);
		const __m256i o = _mm256_loadu_si256((const __m256i *) (off + y));
		const __m128i oL = _mm256_castsi256_si128(o), oH = _mm256_extracti128_si256(o, 1);
		const __m256d tL = _mm256_sub_pd(_mm256_cvtps_pd(_mm256_castps256_ps128(qf)), _mm256_cvtepi32_pd(oL));
		const __m256d tH = _mm256_sub_pd(_mm256_cvtps_pd(_mm256_extractf128_ps(qf, 1)), _mm256_cvtepi32_pd(oH));
		/* a = t - (o >= 0), b = t + (o <= 0): masks from 32-bit compares widened to 64-bit lanes */
		const __m256d geL = _mm256_castsi256_pd(_mm256_cvtepi32_epi64(_mm_xor_si128(_mm_cmpgt_epi32(zero, oL), _mm_set1_epi32(-1))));
		const __m256d geH = _mm256_castsi256_pd(_mm256_cvtepi32_epi64(_mm_xor_si128(_mm_cmpgt_epi32(zero, oH), _mm_set1_epi32(-1))));
		const __m256d leL = _mm256_castsi256_pd(_mm256_cvtepi32_epi64(_mm_xor_si128(_mm_cmpgt_epi32(oL, zero), _mm_set1_epi32(-1))));
		const __m256d leH = _mm256_castsi256_pd(_mm256_cvtepi32_epi64(_mm_xor_si128(_mm_cmpgt_epi32(oH, zero), _mm_set1_epi32(-1))));
		lo0 = _mm256_max_pd(lo0, _mm256_sub_pd(tL, _mm256_and_pd(geL, one)));
		lo1 = _mm256_max_pd(lo1, _mm256_sub_pd(tH, _mm256_and_pd(geH, one)));
		hi0 = _mm256_min_pd(hi0, _mm256_add_pd(tL, _mm256_and_pd(leL, one)));
		hi1 = _mm256_min_pd(hi1, _mm256_add_pd(tH, _mm256_and_pd(leH, one)));
		yv = _mm256_add_epi32(yv, eight);
	}
	double l[4], h[4];
	_mm256_storeu_pd(l, _mm256_max_pd(lo0, lo1));
	_mm256_storeu_pd(h, _mm256_min_pd(hi0, hi1));
	double lo = l[0], hi = h[0];
	for (int i = 1; i < 4; ++i) { if (l[i] > lo) lo = l[i]; if (h[i] < hi) hi = h[i]; }
	for (; y < H; ++y) {
		const float qf = ((float) y - 0.0f) / k;
		q[y] = qf;
		const int32_t o = off[y];
		const double t = (double) qf - (double) o;
		const double a = t - (o >= 0 ? 1.0 : 0.0), b = t + (o <= 0 ? 1.0 : 0.0);
		if (a > lo) lo = a;
		if (b < hi) hi = b;
	}
	*loOut = lo; *hiOut = hi;
}

__attribute__((target("avx2"))) int verifyRightAvx2(const int32_t *off, int H, const float *q, float right, int *dir) {
	const __m256 rv = _mm256_set1_ps(right);
	int y = 0;
	for (; y + 8 <= H; y += 8) {
		const __m256i got = _mm256_cvttps_epi32(_mm256_sub_ps(_mm256_loadu_ps(q + y), rv));
		const __m256i want = _mm256_loadu_si256((const __m256i *) (off + y));
		if (_mm256_movemask_epi8(_mm256_cmpeq_epi32(got, want)) != -1) break;
	}
	for (; y < H; ++y) {
		const int32_t got = (int32_t) (q[y] - right);
		if (got != off[y]) { *dir = got > off[y] ? 1 : -1; return y; }
	}
	return -1;
}

bool const g_avx2 = __builtin_cpu_supports("avx2") && getenv("CVX_CORRIDOR_NO_AVX2") == 0;      /* (the variable: the scalar form under test) */

void setRows(cvx_tile *f) {
	f->corridor_kind = CVX_CORRIDOR_ROWS;
	f->corridor_k = f->corridor_d = f->corridor_right = 0.0f;
	f->corridor_offset = f->corridor_width = 0;
}

}  // namespace

extern "C" int cvx_corridor_fit(const int32_t *row_offset, const int32_t *row_length, int32_t row_stride_bytes, int32_t n_rows,
		int32_t ref_len, int32_t qry_len, cvx_tile *form) {
	if (form == 0) return CVX_ERR_ARG;
	setRows(form);
	if (n_rows < 0 || (n_rows > 0 && (row_offset == 0 || row_length == 0)) || (row_stride_bytes & 3) || row_stride_bytes < 4) return CVX_ERR_ARG;
	const int H = n_rows;
	if (H == 0) return CVX_OK;
	const int32_t width = row_length[0];
	const int32_t off0 = row_offset[0];
	if (width < 0) return CVX_OK;
	/* one pass over the caller's (strided) rows: the offsets into a contiguous scratch, and whether all lengths agree, all
	 * offsets agree, all offsets run down the diagonal -- branch-free, the answers are looked at afterwards */
	static thread_local std::vector<int32_t> offBuf;
	static thread_local std::vector<float> qBuf;
	offBuf.resize((size_t) H + 8);
	int32_t *off = offBuf.data();
	uint32_t lenDiff = 0, offDiff = 0, diagDiff = 0;
	{
		const char *po = (const char *) row_offset, *pl = (const char *) row_length;
		const size_t st = (size_t) row_stride_bytes;
		for (int y = 0; y < H; ++y) {
			const int32_t o = *(const int32_t *) (po + (size_t) y * st);
			const int32_t l = *(const int32_t *) (pl + (size_t) y * st);
			off[y] = o;
			lenDiff |= (uint32_t) (l ^ width);
			offDiff |= (uint32_t) (o ^ off0);
			diagDiff |= (uint32_t) o - (uint32_t) y - (uint32_t) off0;
		}
	}
	if (lenDiff != 0) return CVX_OK;                      /* rows of different lengths: no builder's */
	if (offDiff == 0) {      /* getCorridorFull (:84-105), or a one-row tile */
		form->corridor_kind = CVX_CORRIDOR_CONST;
		form->corridor_offset = off0;
		form->corridor_width = width;
		return CVX_OK;
	}
	if (diagDiff == 0) {     /* getCorridorLinear / getCorridorOriginal (:52-82): i - width / 2, exact in binary32 below 2^24 */
		if (off0 > -(1 << 23) && off0 < (1 << 23) && H < (1 << 23)) {
			const float d = (float) -off0;
			if (cvx::affine_form_ok(1.0f, d, 0.0f, H) && firstMismatch(off, H, d, 1.0f, 0.0f, 0) < 0) {
				form->corridor_kind = CVX_CORRIDOR_AFFINE;
				form->corridor_k = 1.0f; form->corridor_d = d; form->corridor_right = 0.0f;
				form->corridor_width = width;
			}
		}
		return CVX_OK;
	}
	if (ref_len <= 0 || qry_len <= 0) return CVX_OK;
	const float k = (float) qry_len * 1.0f / (float) ref_len;
	if (!(k > 0.0f)) return CVX_OK;
	/* getCorridorEndpoints (:107-127) */
	{
		const float d = (float) width / 2.0f;
		if (cvx::affine_form_ok(k, d, 0.0f, H) && firstMismatch(off, H, d, k, 0.0f, 0) < 0) {
			form->corridor_kind = CVX_CORRIDOR_AFFINE;
			form->corridor_k = k; form->corridor_d = d; form->corridor_right = 0.0f;
			form->corridor_width = width;
			return CVX_OK;
		}
	}
	/* getCorridorEndpointsWithAnchors (:178-191): d = 0, right unknown.  Bounds of `right` over the reals, with t = q - o:
	 *   o > 0:  o <= q - r < o + 1     o < 0:  o - 1 < q - r <= o     o == 0:  -1 < q - r < 1
	 *   i.e. r in (t - [o >= 0], t + [o <= 0]) up to which ends are closed */
	qBuf.resize((size_t) H + 8);
	float *q = qBuf.data();
	double lo, hi;
	if (g_avx2) quotientsAndBoundsAvx2(off, H, k, q, &lo, &hi);
	else quotientsAndBoundsScalar(off, H, k, q, &lo, &hi);
	/* the subtraction is rounded at |q|'s magnitude: allow the intersection to be empty by a few of those ulps */
	const double slack = 8.0 * std::ldexp(1.0, -23) * ((double) H / (double) k + std::fabs(lo) + 1.0);
	if (!(lo <= hi + slack) || !(std::fabs(lo) < 1.0e9) || !(std::fabs(hi) < 1.0e9)) return CVX_OK;
	int32_t oLo = ordOf((float) (lo - slack)) - 2, oHi = ordOf((float) (hi + slack)) + 2;
	float cand = (float) (0.5 * (lo + hi));
	for (int it = 0; it < 64 && oLo <= oHi; ++it) {
		int dir = 0;
		if (!cvx::affine_form_ok(k, 0.0f, cand, H)) break;
		if ((g_avx2 ? verifyRightAvx2(off, H, q, cand, &dir) : verifyRightScalar(off, H, q, cand, &dir)) < 0) {
			form->corridor_kind = CVX_CORRIDOR_AFFINE;
			form->corridor_k = k; form->corridor_d = 0.0f; form->corridor_right = cand;
			form->corridor_width = width;
			return CVX_OK;
		}
		const int32_t oc = ordOf(cand);
		if (dir > 0) oLo = oc + 1;                 /* value too large somewhere: right must grow */
		else oHi = oc - 1;
		if (oLo > oHi) break;                      /* rows on both sides disagree: not this builder's corridor */
		cand = floatOf(oLo + (int32_t) (((int64_t) oHi - (int64_t) oLo) / 2));
	}
	return CVX_OK;                                 /* the caller's rows travel as they are */
}

/* The same for a whole tile table on n_threads host threads (0 = all): every tile that carries rows (CVX_CORRIDOR_ROWS) gets its
 * closed form written in place when there is one -- what a binding does per SingleAlign (ConvexAlignHip::Prepare) done for a
 * batch that arrived as rows (bench.py's `binding_input_form`). */
#include <atomic>
#include <thread>

extern "C" int cvx_corridor_fit_batch(int32_t n_tiles, cvx_tile *tiles, int32_t n_threads, int32_t *n_fitted) {
	if (n_tiles < 0 || (n_tiles > 0 && tiles == 0)) return CVX_ERR_ARG;
	int nt = n_threads > 0 ? n_threads : (int) std::thread::hardware_concurrency();
	if (nt < 1) nt = 1;
	if (nt > (n_tiles + 63) / 64) nt = (n_tiles + 63) / 64 > 0 ? (n_tiles + 63) / 64 : 1;
	std::atomic<int> next(0), fitted(0), err(CVX_OK);
	auto work = [&]() {
		int mine = 0;
		for (;;) {
			const int b = next.fetch_add(64);
			if (b >= n_tiles) break;
			const int e = b + 64 < n_tiles ? b + 64 : n_tiles;
			for (int i = b; i < e; ++i) {
				cvx_tile &t = tiles[i];
				if (t.corridor_kind != CVX_CORRIDOR_ROWS) continue;
				const int rc = cvx_corridor_fit(t.row_offset, t.row_length, t.row_stride_bytes, t.qry_len, t.ref_len, t.qry_len, &t);
				if (rc != CVX_OK) { int ok = CVX_OK; err.compare_exchange_strong(ok, rc); }
				else if (t.corridor_kind != CVX_CORRIDOR_ROWS) mine += 1;
			}
		}
		fitted += mine;
	};
	std::vector<std::thread> th;
	for (int t = 1; t < nt; ++t) th.emplace_back(work);
	work();
	for (auto &t : th) t.join();
	if (n_fitted) *n_fitted = fitted.load();
	return err.load();
}
