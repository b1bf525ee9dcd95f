/*
 * cvx_score.hip -- sub-read scoring on gfx950: what
 * StrippedSW::BatchScore / SingleScore compute (reference src/StrippedSW.cpp:118-203 over
 * the vendored striped Smith-Waterman, lib/Complete-Striped-Smith-Waterman-Library/src/ssw.c).
 *
 * Semantics (derivation in oracle/score_oracle.c): both strings are taken with their
 * terminating NUL, letters map to A/C/G/T = 0..3 and anything else to 4, the matrix is
 * +1 / -1 with zeros for code 4, and because gap_open = gap_extension = -1 reach ssw as
 * uint8_t 255, the score is the maximum of
 *     H[i][j] = max(0, H[i-1][j-1] + s, H[i-1][j] - 255, H[i][j-1] - 255).
 *
 * One wave per (reference window, read) pair, row by row over the read.  The serial
 * left dependency of a row is a max-plus scan: with a_k = max(0, diag + s, up - 255),
 * H[j] = max_{k<=j}(a_k + 255 k) - 255 j, i.e. an inclusive prefix MAX of a_k + 255 k --
 * done with 6 DPP/shuffle steps per 64 columns plus a carry between 64-column chunks.
 * Integer arithmetic, exact.  Rows live in a small global scratch (L1/L2 resident).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvx_launch.h"

namespace cvx {

__device__ __forceinline__ int nt_code(int c) {
	c |= 0x20;                       /* nt_table is case-insensitive; U/u map to 0 like A (src/StrippedSW.cpp:111-116) */
	return (c == 'a' || c == 'u') ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : c == 't' ? 3 : 4;
}

__global__ void __launch_bounds__(64)
score_kernel(const uint8_t *seq, const ScorePair *pairs, int32_t *scratch, float *out, int n) {
	const int p = blockIdx.x;
	if (p >= n) return;
	const int lane = threadIdx.x;
	const ScorePair pr = pairs[p];
	/* lengths include the terminating NUL, as in the reference (:131-132) */
	const int R = pr.ref_len, Q = pr.qry_len;
	if (R >= 100000 || Q >= 100000) {      /* maxSeqLen, src/StrippedSW.h:88 */
		if (lane == 0) out[p] = -1.0f;
		return;
	}
	const uint8_t *ref = seq + pr.ref_off;
	const uint8_t *qry = seq + pr.qry_off;
	int32_t *prev = scratch + pr.scratch_off;
	int32_t *cur = prev + R;
	for (int j = lane; j < R; j += 64) prev[j] = 0;
	__syncthreads();

	int best = 0;
	for (int i = 0; i < Q; ++i) {
		const int qc = nt_code(qry[i]);
		int carry = -0x40000000;
		for (int c0 = 0; c0 < R; c0 += 64) {
			const int j = c0 + lane;
			const bool valid = j < R;
			int key = -0x40000000;
			if (valid) {
				const int up = prev[j];
				const int dg = j > 0 ? prev[j - 1] : 0;
				const int rc = nt_code(ref[j]);
				const int s = (qc == 4 || rc == 4) ? 0 : (qc == rc ? 1 : -1);
				int a = dg + s;
				a = max(a, up - 255);
				a = max(a, 0);
				key = a + 255 * j;
			}
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const int t = __shfl_up(key, d, 64);
				if (lane >= d) key = max(key, t);
			}
			key = max(key, carry);
			if (valid) {
				const int h = key - 255 * j;
				cur[j] = h;
				best = max(best, h);
			}
			carry = __builtin_amdgcn_readlane(key, 63);
		}
		int32_t *t = prev; prev = cur; cur = t;
		__syncthreads();
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
	if (lane == 0) out[p] = (float) best;
}

hipError_t launch_score(const uint8_t *seq, const ScorePair *pairs, int32_t *scratch, float *out, int n, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(score_kernel, dim3(n), dim3(64), 0, st, seq, pairs, scratch, out, n);
	return hipGetLastError();
}

}  // namespace cvx
