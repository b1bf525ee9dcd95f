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
 * Integer arithmetic, exact.
 *
 * score_reg_kernel<KC> is the kernel for the shape the reference actually batches (1024 pairs of a
 * 256-base sub-read against a ~300-base window, src/ScoreBuffer.cpp:87-168): the previous DP row
 * lives in registers -- lane l holds columns l, l + 64, ... (KC <= 8 chunks, windows up to 512
 * columns) --, the diagonal input is a one-lane shuffle, nothing touches memory inside the row
 * loop, and four pairs share a workgroup.  score_kernel keeps rows in a global scratch and takes
 * whatever is longer (the inversion checks on kb-long sequences).
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

template <int KC>
__global__ void __launch_bounds__(256)
score_reg_kernel(const uint8_t *seq, const ScorePair *pairs, float *out, int n) {
	const int lane = threadIdx.x & 63;
	const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (p >= n) return;                                  /* wave-uniform */
	const ScorePair pr = pairs[p];
	const int R = pr.ref_len, Q = pr.qry_len;            /* lengths include the NUL (:131-132) */
	if (R >= 100000 || Q >= 100000) {                    /* maxSeqLen, src/StrippedSW.h:88 */
		if (lane == 0) out[p] = -1.0f;
		return;
	}
	const uint8_t *ref = seq + pr.ref_off;
	const uint8_t *qry = seq + pr.qry_off;
	int rcode[KC], prev[KC];
#pragma unroll
	for (int k = 0; k < KC; ++k) {
		const int j = lane + 64 * k;
		rcode[k] = j < R ? nt_code(ref[j]) : 4;
		prev[k] = 0;
	}
	int best = 0;
	for (int i = 0; i < Q; ++i) {
		const int qc = nt_code(qry[i]);
		int carry = -0x40000000;                         /* max over the columns so far of a + 255 * column */
		int prev_last = 0;                               /* H[i-1][64k - 1]: the diagonal input of a chunk's first column */
#pragma unroll
		for (int k = 0; k < KC; ++k) {
			const int j = lane + 64 * k;
			if (64 * k < R) {                            /* wave-uniform */
				const bool valid = j < R;
				const int up = prev[k];
				int dg = __shfl_up(up, 1, 64);
				if (lane == 0) dg = prev_last;           /* column 0 of the matrix: 0 */
				prev_last = __builtin_amdgcn_readlane(up, 63);
				const int s = (qc == 4 || rcode[k] == 4) ? 0 : (qc == rcode[k] ? 1 : -1);
				int a = max(max(dg + s, up - 255), 0);
				int key = valid ? a + 255 * j : -0x40000000;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const int t = __shfl_up(key, d, 64);
					if (lane >= d) key = max(key, t);
				}
				key = max(key, carry);
				carry = __builtin_amdgcn_readlane(key, 63);
				const int h = valid ? key - 255 * j : 0;
				prev[k] = h;
				best = max(best, h);
			}
		}
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
	if (lane == 0) out[p] = (float) best;
}

/*
 * score_diag_kernel -- the batched shape without any serial dependency.  A gap costs 255 per base, so an alignment
 * with g >= 1 gaps scores sum(segments) - 255 g (or less), where the ungapped segments together consume at most
 * min(Q, R) bases of the shorter sequence: with min(Q, R) <= 511 the segments other than the best one total at most
 * 511 g / (g + 1) <= 255 g (g = 1: the smaller of two segments is <= 255), hence no gapped path beats the best
 * ungapped one -- the matrix maximum is the maximum over all diagonals of the best contiguous run (Kadane:
 * h = max(0, h + s)), which is what H[i][j] = max(0, H[i-1][j-1] + s, ...) computes along a diagonal when the gap
 * terms never win.  Every lane walks its own diagonal, 64 diagonals at a time, both sequences as codes in LDS; no
 * shuffles, no row state.  Exact for the same reason the 8-bit ssw kernel's answer is (scores <= 511 here).
 * One wave per pair, four pairs per workgroup; windows up to kDiagMaxRef columns.
 */
static const int kDiagMaxQry = 512, kDiagMaxRef = 2048;

__global__ void __launch_bounds__(256)
score_diag_kernel(const uint8_t *seq, const ScorePair *pairs, float *out, int n) {
	__shared__ uint8_t s_q[4][kDiagMaxQry];
	__shared__ uint8_t s_r[4][kDiagMaxRef + 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int p = blockIdx.x * 4 + w;
	const bool live = p < n;                             /* wave-uniform */
	ScorePair pr;
	pr.ref_off = pr.qry_off = 0; pr.ref_len = pr.qry_len = 0; pr.scratch_off = 0;
	if (live) pr = pairs[p];
	const int R = pr.ref_len, Q = pr.qry_len;            /* lengths include the NUL (:131-132) */
	for (int j = lane; j < R; j += 64) s_r[w][j] = (uint8_t) nt_code(seq[pr.ref_off + (unsigned) j]);
	for (int i = lane; i < Q; i += 64) s_q[w][i] = (uint8_t) nt_code(seq[pr.qry_off + (unsigned) i]);
	__syncthreads();
	if (!live) return;
	int best = 0;
	for (int d0 = -(Q - 1); d0 < R; d0 += 64) {
		const int d = d0 + lane;                         /* this lane's diagonal: column j = i + d */
		const int i_lo = d < 0 ? -d : 0;
		const int i_hi = d < R ? min(Q, R - d) : 0;      /* rows [i_lo, i_hi) */
		const int w_lo = max(0, -(d0 + 63)), w_hi = min(Q, R - d0);       /* the wave's union of row ranges */
		int h = 0;
		for (int i = w_lo; i < w_hi; ++i) {
			const int qc = s_q[w][i];
			const bool act = i >= i_lo && i < i_hi;
			const int rc = act ? (int) s_r[w][i + d] : 4;
			const int s = (qc == 4 || rc == 4) ? 0 : (qc == rc ? 1 : -1);
			h = act ? max(h + s, 0) : 0;
			best = max(best, h);
		}
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
	if (lane == 0) out[p] = (float) best;
}

hipError_t launch_score_diag(const uint8_t *seq, const ScorePair *pairs, float *out, int n, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(score_diag_kernel, dim3((n + 3) / 4), dim3(256), 0, st, seq, pairs, out, n);
	return hipGetLastError();
}

hipError_t launch_score(const uint8_t *seq, const ScorePair *pairs, int32_t *scratch, float *out, int n, int max_ref_len, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	const int blocks = (n + 3) / 4;
	if (max_ref_len <= 64 * 5) hipLaunchKernelGGL(score_reg_kernel<5>, dim3(blocks), dim3(256), 0, st, seq, pairs, out, n);
	else if (max_ref_len <= 64 * 8) hipLaunchKernelGGL(score_reg_kernel<8>, dim3(blocks), dim3(256), 0, st, seq, pairs, out, n);
	else hipLaunchKernelGGL(score_kernel, dim3(n), dim3(64), 0, st, seq, pairs, scratch, out, n);
	return hipGetLastError();
}

}  // namespace cvx
