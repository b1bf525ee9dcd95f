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
#include <cstring>
#include <stdint.h>
#include <vector>

#include "cvx_align.h"
#include "cvx_types.h"
#include "cvx_host_logic.h"

namespace {

inline int32_t rowOffset(const int32_t *base, int32_t strideBytes, int y) {
	return *(const int32_t *) ((const char *) base + (size_t) y * (size_t) strideBytes);
}

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
inline int firstMismatch(const int32_t *off, int32_t stride, int H, float d, float k, float right, int *dir) {
	for (int y = 0; y < H; ++y) {
		const int32_t got = cvx::affine_row_offset(y, d, k, right);
		const int32_t want = rowOffset(off, stride, y);
		if (got != want) {
			if (dir) *dir = got > want ? 1 : -1;
			return y;
		}
	}
	return -1;
}

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
	bool sameOff = true, sameDiag = true;
	for (int y = 0; y < H; ++y) {
		if (rowOffset(row_length, row_stride_bytes, y) != width) return CVX_OK;      /* rows of different lengths: no builder's */
		const int32_t o = rowOffset(row_offset, row_stride_bytes, y);
		sameOff = sameOff && o == off0;
		sameDiag = sameDiag && (int64_t) o - y == (int64_t) off0;
	}
	if (sameOff) {      /* getCorridorFull (:84-105), or a one-row tile */
		form->corridor_kind = CVX_CORRIDOR_CONST;
		form->corridor_offset = off0;
		form->corridor_width = width;
		return CVX_OK;
	}
	if (sameDiag) {     /* getCorridorLinear / getCorridorOriginal (:52-82): i - width / 2, exact in binary32 below 2^24 */
		if (off0 > -(1 << 23) && off0 < (1 << 23) && H < (1 << 23)) {
			const float d = (float) -off0;
			if (cvx::affine_form_ok(1.0f, d, 0.0f, H) && firstMismatch(row_offset, row_stride_bytes, H, d, 1.0f, 0.0f, 0) < 0) {
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
		if (cvx::affine_form_ok(k, d, 0.0f, H) && firstMismatch(row_offset, row_stride_bytes, H, d, k, 0.0f, 0) < 0) {
			form->corridor_kind = CVX_CORRIDOR_AFFINE;
			form->corridor_k = k; form->corridor_d = d; form->corridor_right = 0.0f;
			form->corridor_width = width;
			return CVX_OK;
		}
	}
	/* getCorridorEndpointsWithAnchors (:178-191): d = 0, right unknown.  Bounds of `right` over the reals:
	 *   o > 0:  o <= q - r < o + 1     o < 0:  o - 1 < q - r <= o     o == 0:  -1 < q - r < 1 */
	double lo = -1.0e30, hi = 1.0e30;
	static thread_local std::vector<float> qOf;      /* (float) y / k per row: the divide is paid once, the candidates below reuse it */
	qOf.resize((size_t) H);
	for (int y = 0; y < H; ++y) {
		const float qf = ((float) y - 0.0f) / k;
		qOf[(size_t) y] = qf;
		const double q = (double) qf;
		const int32_t o = rowOffset(row_offset, row_stride_bytes, y);
		double a, b;
		if (o > 0) { a = q - (double) o - 1.0; b = q - (double) o; }
		else if (o < 0) { a = q - (double) o; b = q - (double) o + 1.0; }
		else { a = q - 1.0; b = q + 1.0; }
		if (a > lo) lo = a;
		if (b < hi) hi = b;
	}
	/* the subtraction is rounded at |q|'s magnitude: allow the intersection to be empty by a few of those ulps */
	const double slack = 8.0 * std::ldexp(1.0, -23) * ((double) H / (double) k + std::fabs(lo) + 1.0);
	if (!(lo <= hi + slack) || !(std::fabs(lo) < 1.0e9) || !(std::fabs(hi) < 1.0e9)) return CVX_OK;
	int32_t oLo = ordOf((float) (lo - slack)) - 2, oHi = ordOf((float) (hi + slack)) + 2;
	float cand = (float) (0.5 * (lo + hi));
	for (int it = 0; it < 64 && oLo <= oHi; ++it) {
		int dir = 0;
		bool same = cvx::affine_form_ok(k, 0.0f, cand, H);
		if (same) {
			/* affine_row_offset(y, 0, k, cand) with its quotient taken from the table: the same three binary32 operations */
			for (int y = 0; y < H; ++y) {
				const int32_t got = (int32_t) (qOf[(size_t) y] - cand);
				const int32_t want = rowOffset(row_offset, row_stride_bytes, y);
				if (got != want) { dir = got > want ? 1 : -1; same = false; break; }
			}
		}
		if (same) {
			form->corridor_kind = CVX_CORRIDOR_AFFINE;
			form->corridor_k = k; form->corridor_d = 0.0f; form->corridor_right = cand;
			form->corridor_width = width;
			return CVX_OK;
		}
		if (dir == 0) break;                       /* the form itself is not acceptable */
		const int32_t oc = ordOf(cand);
		if (dir > 0) oLo = oc + 1;                 /* value too large somewhere: right must grow */
		else oHi = oc - 1;
		if (oLo > oHi) break;                      /* rows on both sides disagree: not this builder's corridor */
		cand = floatOf(oLo + (int32_t) (((int64_t) oHi - (int64_t) oLo) / 2));
	}
	return CVX_OK;                                 /* the caller's rows travel as they are */
}
