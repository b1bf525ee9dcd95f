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
	/* one WAVE per window, four windows per workgroup (round 6: a 4.5 kb window is 285 sixteen-character pieces -- a workgroup of 256
	 * lanes did one piece per lane and spent its life starting up) */
	const int wi = blockIdx.x * 4 + (int) (threadIdx.x >> 6);
	const int lane = (int) (threadIdx.x & 63u);
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
	/* Sixteen characters per lane and trip (round 6; a byte per lane -- 64-byte stores per wave instruction -- reached 1.6 TB/s of the
	 * 1.5 B per character this kernel moves): the lane owns one 16-byte-ALIGNED piece of the destination, reads the
	 * 8-9 genome bytes behind it with one unaligned 8-byte load (+ one byte when the piece starts on a low nibble), expands them
	 * through a table packed into a 64-bit constant, and stores the piece with one dwordx4; the window's ragged head and tail (its
	 * address in the arena is whatever the tile layout gave it) go byte by byte. */
	const unsigned long long kLut = 0x3F3F3F4E43475441ull;      /* 'A' 'T' 'G' 'C' 'N' '?' '?' '?' : dec4, src/SequenceProvider.cpp:90-104 */
	const long long head = (long long) ((16u - (unsigned) ((uintptr_t) out & 15u)) & 15u);      /* characters in front of the first aligned piece */
	const long long n_pieces = n_chars > head ? (n_chars - head + 15) / 16 : 0;
	auto char_at = [&](const long long i) -> uint8_t {      /* the kernel's definition of character i (used for the ragged ends) */
		const long long k = i - off;
		if (!(go && k >= 0 && (unsigned long long) k < n_dec)) return (uint8_t) 'x';
		const unsigned long long p = dstart + (unsigned long long) k;
		const unsigned b = bin[p >> 1];
		return dec4_char((p & 1ull) ? (b & 0xFu) : (b >> 4));
	};
	for (long long i = lane; i < (head < n_chars ? head : n_chars); i += 64) out[i] = char_at(i);
	for (long long pc = lane; pc < n_pieces; pc += 64) {
		const long long i0 = head + 16 * pc;
		const long long k0 = i0 - off;
		const bool whole = i0 + 16 <= n_chars;
		const bool inside = go && k0 >= 0 && (unsigned long long) (k0 + 16) <= n_dec;      /* all sixteen are decoded characters */
		if (whole && inside) {
			const unsigned long long p0 = dstart + (unsigned long long) k0;
			const uint8_t *src = bin + (p0 >> 1);
			unsigned long long lo8;
			__builtin_memcpy(&lo8, src, 8);                  /* nibbles p0 & ~1 .. + 15, high nibble of a byte first */
			unsigned extra = 0;
			if (p0 & 1ull) extra = src[8];
			uint32_t wv[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				uint32_t word = 0;
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const unsigned t = (unsigned) (4 * q + c) + (unsigned) (p0 & 1ull);      /* nibble index from the first loaded byte */
					const unsigned byte = t < 16u ? (unsigned) ((lo8 >> (8u * (t >> 1))) & 0xFFull) : extra;
					unsigned nib = (t & 1u) ? (byte & 0xFu) : (byte >> 4);
					nib = nib < 5u ? nib : 5u;
					word |= (uint32_t) ((kLut >> (8u * nib)) & 0xFFull) << (8 * c);
				}
				wv[q] = word;
			}
			*reinterpret_cast<uint4 *>(out + i0) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
		} else {
			const long long i1 = i0 + 16 < n_chars ? i0 + 16 : n_chars;
			for (long long i = i0; i < i1; ++i) out[i] = char_at(i);
		}
	}
}

hipError_t launch_decode_windows(const uint8_t *bin, const uint64_t *starts, int n_starts,
		const WindowDesc *win, int n, uint8_t *dst, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(decode_windows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, bin,
			reinterpret_cast<const unsigned long long *>(starts), n_starts, win, n, dst);
	return hipGetLastError();
}

}  // namespace cvx
