/*
 * score_oracle.c -- plain-C restatement of what StrippedSW::BatchScore / SingleScore
 * compute (reference src/StrippedSW.cpp:118-203 over lib/.../ssw.c).  TEST
 * INFRASTRUCTURE ONLY (see score_abi.h); pinned against oracle/_ref in
 * tests/test_score_oracle_cpu.py.
 *
 * What the reference does, line by line:
 *  - both strings are taken WITH their terminating NUL (read_len = strlen + 1,
 *    src/StrippedSW.cpp:131-132) and mapped through nt_table (:108-114): A/C/G/T (either
 *    case) -> 0..3, U/u -> 0, everything else (N, x, NUL) -> 4;
 *  - the 5x5 matrix is +1 on the ACGT diagonal, -1 off it, 0 in row/column 4
 *    (src/StrippedSW.h:20-36);
 *  - gap_open = gap_extension = -1 are passed to ssw_align's `const uint8_t weight_gapO/E`
 *    parameters (ssw.c:997-998) and therefore arrive as 255: a gap costs 255 per base;
 *  - ssw_align runs the 8-bit kernel and, when its (biased) maximum reaches 255, the 16-bit
 *    kernel (ssw.c:1018-1031); only score1 is used.  Below 255 no cell can afford a gap, at or
 *    above it the 16-bit kernel's Farrar recurrence is the textbook one for the maximum (its
 *    lazy-F shortcut only forbids an insertion adjacent to a deletion, which a -1 mismatch
 *    always beats at this gap cost).
 *  => score = max over the matrix of
 *       H[i][j] = max(0, H[i-1][j-1] + s(i,j), H[i-1][j] - 255, H[i][j-1] - 255).
 *  - sequences of maxSeqLen (100000) or more score -1.0f (:134-135).
 * Exact while the score stays below 32767 (the 16-bit kernel saturates there).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "score_abi.h"

static int code_of(unsigned char c) {
	switch (c) {
	case 'A': case 'a': case 'U': case 'u': return 0;   /* nt_table[85] = nt_table[117] = 0, src/StrippedSW.cpp:114-116 */
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

static float score_pair(const char *ref, const char *qry) {
	const int ref_len = (int) strlen(ref) + 1, read_len = (int) strlen(qry) + 1;
	if (read_len >= 100000 || ref_len >= 100000) return -1.0f;
	int32_t *prev = (int32_t *) calloc((size_t) ref_len + 1, sizeof(int32_t));
	int32_t *cur = (int32_t *) calloc((size_t) ref_len + 1, sizeof(int32_t));
	int32_t best = 0;
	for (int i = 0; i < read_len; ++i) {
		const int a = code_of((unsigned char) qry[i]);
		int32_t left = 0;
		for (int j = 0; j < ref_len; ++j) {
			const int b = code_of((unsigned char) ref[j]);
			const int s = (a == 4 || b == 4) ? 0 : (a == b ? 1 : -1);
			int32_t h = (j > 0 ? prev[j - 1] : 0) + s;
			if (prev[j] - 255 > h) h = prev[j] - 255;
			if (left - 255 > h) h = left - 255;
			if (h < 0) h = 0;
			cur[j] = h;
			left = h;
			if (h > best) best = h;
		}
		int32_t *t = prev; prev = cur; cur = t;
	}
	free(prev); free(cur);
	return (float) best;
}

void *score_oracle_create(void) { return malloc(1); }
void score_oracle_destroy(void *h) { free(h); }
const char *score_oracle_kind(void) { return "port"; }
int score_oracle_batch(void *h, int n, const char *const *refs, const char *const *qrys, float *out) {
	(void) h;
	for (int i = 0; i < n; ++i) out[i] = score_pair(refs[i], qrys[i]);
	return n;
}
