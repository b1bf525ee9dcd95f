/*
 * cvx_genome.hip -- reference windows decoded on the device from a 4-bit genome resident in HBM
 * (gfx950; SURVEY.md 8 f4, decode half).
 *
 * ngmlr keeps its reference as two bases per byte (A 0, T 1, G 2, C 3, N 4; high nibble first,
 * src/SequenceProvider.cpp:76-113, :333-386) and expands a window to chars on the host for every
 * alignment (DecodeRefSequenceExact, :493-565, called with corridor 0 from
 * src/AlignmentBuffer.cpp:215).  With the encoded genome resident in HBM a tile only carries
 * (position, length): the window is written straight into the batch's sequence arena, and the
 * decoded characters -- 10 % of the upload -- never cross PCIe.
 *
 * decode_windows_kernel restates DecodeRefSequenceExact(sequence, position, length, 0):
 *   memset 'x'; chromosome of the position (getChrStart :157-180: first start above it, or the next
 *   one when the position lies in the 1000 N in front of it); the end clipped to the chromosome's
 *   end; a start in front of the chromosome skips to its first base; decode() then writes
 *   (start & 1) + 2 * ((end - start + 1) / 2) characters, character k = the nibble of position
 *   start + k.  Everything else stays 'x' (which never matches anything in the DP).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvx_types.h"
#include "cvx_launch.h"

namespace cvx {

__device__ __forceinline__ uint8_t dec4_char(unsigned v) {
	/* dec4, src/SequenceProvider.cpp:90-104 (values above 4 throw there; a valid genome has none) */
	return v == 0u ? 'A' : v == 1u ? 'T' : v == 2u ? 'G' : v == 3u ? 'C' : v == 4u ? 'N' : '?';
}

__global__ void __launch_bounds__(256)
decode_windows_kernel(const uint8_t *bin, const unsigned long long *starts, int n_starts,
		const WindowDesc *win, int n, uint8_t *dst) {
	const int wi = blockIdx.x;
	if (wi >= n) return;
	const WindowDesc w = win[wi];
	const long long n_chars = w.n_chars;                 /* characters to produce (the caller's NUL is not stored) */
	const long long seq_len = w.n_chars + 1;             /* sequenceLength of the reference's call */
	const unsigned long long sp = w.position;
	/* getChrStart: upper_bound over the start table */
	int lo = 0, hi = n_starts;                           /* first index with starts[i] > sp */
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (starts[mid] > sp) hi = mid; else lo = mid + 1;
	}
	int up = lo;
	if (up < n_starts && starts[up] - sp < 1000ull) up += 1;
	bool go = (up > 0 && up < n_starts);
	unsigned long long dstart = sp, n_dec = 0;
	long long off = 0;
	if (go) {
		const unsigned long long chr_start = starts[up - 1], chr_end = starts[up] - 1000ull;
		const unsigned long long end_pos = sp + (unsigned long long) seq_len;
		unsigned long long dend = end_pos;
		if (end_pos > chr_end) dend -= end_pos - chr_end;
		if (dstart < chr_start) {
			if (dend > chr_start) { off = (long long) (chr_start - dstart); dstart = chr_start; }
			else go = false;
		}
		if (go) n_dec = (dstart & 1ull) + 2ull * ((dend - dstart + 1ull) / 2ull);
	}
	uint8_t *out = dst + w.dst_off;
	for (long long i = threadIdx.x; i < n_chars; i += blockDim.x) {
		uint8_t c = 'x';
		const long long k = i - off;
		if (go && k >= 0 && (unsigned long long) k < n_dec) {
			const unsigned long long p = dstart + (unsigned long long) k;
			const unsigned b = bin[p >> 1];
			c = dec4_char((p & 1ull) ? (b & 0xFu) : (b >> 4));
		}
		out[i] = c;
	}
}

hipError_t launch_decode_windows(const uint8_t *bin, const uint64_t *starts, int n_starts,
		const WindowDesc *win, int n, uint8_t *dst, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(decode_windows_kernel, dim3(n), dim3(256), 0, st, bin,
			reinterpret_cast<const unsigned long long *>(starts), n_starts, win, n, dst);
	return hipGetLastError();
}

}  // namespace cvx
