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

/* four nibble values, one per byte (0..15) -> four characters: v_perm_b32 as an eight-entry byte table (A T G C | N ? ? ?), a second
 * one that turns bit 3 of a value into a byte mask (selector 12 reads 0x00, 13 reads 0xFF) for the values the table does not hold */
__device__ __forceinline__ uint32_t lut4(const uint32_t x) {
	const uint32_t c = __builtin_amdgcn_perm(0x3F3F3F4Eu, 0x43475441u, x & 0x07070707u);
	const uint32_t m = __builtin_amdgcn_perm(0u, 0u, ((x >> 3) & 0x01010101u) | 0x0C0C0C0Cu);
	return (c & ~m) | (0x3F3F3F3Fu & m);
}
/* eight nibbles, the first in bits 31:28 -> their characters in memory order */
__device__ __forceinline__ void expand8(const uint32_t d, uint32_t &o0, uint32_t &o1) {
	const uint32_t ch = lut4((d >> 4) & 0x0F0F0F0Fu), cl = lut4(d & 0x0F0F0F0Fu);      /* nibbles 0 2 4 6 / 1 3 5 7, the first in byte 3 */
	o0 = __builtin_amdgcn_perm(ch, cl, 0x02060307u);
	o1 = __builtin_amdgcn_perm(ch, cl, 0x00040105u);
}

__global__ void __launch_bounds__(256)
decode_windows_kernel(const uint8_t * __restrict__ bin, const unsigned long long * __restrict__ starts, int n_starts,
		const WindowDesc * __restrict__ win, int n, uint8_t * __restrict__ dst) {
	/* one WAVE per window, four windows per workgroup (round 6: a 4.5 kb window is 285 sixteen-character pieces -- a workgroup of 256
	 * lanes did one piece per lane and spent its life starting up).  The window index is wave-uniform and says so (readfirstlane): the
	 * descriptor and the search over the start table are scalar loads, off the vector memory pipe */
	const int wi = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4u + (threadIdx.x >> 6)));
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
	 * four at a time through v_perm_b32 used as a byte table (lut4: 40 instructions per piece where shifts and selects per character
	 * were 240 -- the kernel was as much instruction- as latency-bound), and stores the piece with one dwordx4; the window's ragged head and tail (its
	 * address in the arena is whatever the tile layout gave it) go byte by byte. */
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
	/* Four pieces per lane in flight (late round 6): the genome bytes of a lane's next four pieces are requested before the first
	 * is expanded -- a wave's 4.5 trips over a window were 4.5 dependent round trips to HBM (load, expand, store, next load), now
	 * two.  Pointers are __restrict__: the stores into the arena cannot alias the genome. */
	constexpr int U = 4;
	for (long long base = 0; base < n_pieces; base += 64 * U) {
		unsigned long long lo8[U];
		unsigned extra[U], odd[U];
		long long at[U];
		bool fast[U], valid[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const long long pc = base + 64 * u + lane;
			const long long i0 = head + 16 * pc;
			const long long k0 = i0 - off;
			valid[u] = pc < n_pieces;
			at[u] = i0;
			/* all sixteen characters lie in the window and are decoded characters */
			fast[u] = valid[u] && i0 + 16 <= n_chars && go && k0 >= 0 && (unsigned long long) (k0 + 16) <= n_dec;
			lo8[u] = 0ull; extra[u] = 0u; odd[u] = 0u;
			if (fast[u]) {
				const unsigned long long p0 = dstart + (unsigned long long) k0;
				const uint8_t *src = bin + (p0 >> 1);
				__builtin_memcpy(&lo8[u], src, 8);               /* nibbles p0 & ~1 .. + 15, high nibble of a byte first */
				odd[u] = (unsigned) (p0 & 1ull);
				if (odd[u]) extra[u] = src[8];
			}
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			if (fast[u]) {
				/* the sixteen nibbles as one big-endian number (the first in bits 63:60); a piece that starts on a low nibble
				 * drops the first and takes the seventeenth from the ninth byte */
				unsigned long long nb = __builtin_bswap64(lo8[u]);
				if (odd[u]) nb = (nb << 4) | (unsigned long long) (extra[u] >> 4);
				uint32_t wv[4];
				expand8((uint32_t) (nb >> 32), wv[0], wv[1]);
				expand8((uint32_t) nb, wv[2], wv[3]);
				*reinterpret_cast<uint4 *>(out + at[u]) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
			} else if (valid[u]) {
				const long long i1 = at[u] + 16 < n_chars ? at[u] + 16 : n_chars;
				for (long long i = at[u]; i < i1; ++i) out[i] = char_at(i);
			}
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
