/*
 * cvx_genome_host.cpp -- host half of the resident-genome entry points of include/cvx_align.h:
 * the 4-bit encoding ngmlr keeps its reference in.  No device code, no HIP.
 *
 * Follows _SequenceProvider::Init (reference src/SequenceProvider.cpp:333-386; enc4 :76-89) and the
 * refStartPos table of :415-424; checked byte for byte against the binRef the unmodified reference
 * builds from its own test genomes (tests/test_decode_cpu.py, fixtures from tools/make_golden.sh).
 */
#include <cstdint>

#include "cvx_align.h"

namespace {

const int kMinRefSeqLen = 10;          /* SequenceProvider.h:79: shorter sequences are skipped */
const int kSpacerBytes = 500;          /* 1000 N in front of the first sequence and after every one */

inline unsigned code_of(char c) {      /* enc4: case-insensitive, everything but ACGT is N */
	switch (c) {
	case 'A': case 'a': return 0u;
	case 'T': case 't': return 1u;
	case 'G': case 'g': return 2u;
	case 'C': case 'c': return 3u;
	default: return 4u;
	}
}

}  // namespace

extern "C" {

uint64_t cvx_genome_encoded_bytes(int32_t n, const uint64_t *lengths) {
	uint64_t b = kSpacerBytes;
	for (int32_t i = 0; i < n; ++i)
		if (lengths[i] > (uint64_t) kMinRefSeqLen) b += (lengths[i] + 1) / 2 + kSpacerBytes;
	return b;
}

int cvx_genome_encode(int32_t n, const char *const *seqs, const uint64_t *lengths, uint8_t *bin_ref,
		uint64_t *n_nibbles, uint64_t *start_table, int32_t *n_starts) {
	if (n < 0 || (n > 0 && (!seqs || !lengths)) || !bin_ref || !n_nibbles || !start_table || !n_starts) return CVX_ERR_ARG;
	const uint8_t spacer = (uint8_t) ((4u << 4) | 4u);
	uint64_t at = 0;
	for (int q = 0; q < kSpacerBytes; ++q) bin_ref[at++] = spacer;
	int32_t kept = 0;
	uint64_t last_len = 0;
	for (int32_t i = 0; i < n; ++i) {
		const uint64_t len = lengths[i];
		if (len <= (uint64_t) kMinRefSeqLen) continue;
		if (!seqs[i]) return CVX_ERR_ARG;
		start_table[kept++] = at * 2;                    /* first base of the sequence, in nibbles */
		last_len = len;
		const char *s = seqs[i];
		const uint64_t pairs = len / 2;
		for (uint64_t k = 0; k < pairs; ++k) bin_ref[at++] = (uint8_t) ((code_of(s[2 * k]) << 4) | code_of(s[2 * k + 1]));
		if (len & 1) bin_ref[at++] = (uint8_t) ((code_of(s[len - 1]) << 4) | 4u);
		for (int q = 0; q < kSpacerBytes; ++q) bin_ref[at++] = spacer;
	}
	*n_nibbles = at * 2;
	if (kept == 0) { *n_starts = 0; return CVX_ERR_ARG; }
	start_table[kept] = start_table[kept - 1] + last_len + 1000;   /* upper bound for positions on the last sequence */
	*n_starts = kept + 1;
	return CVX_OK;
}

}  /* extern "C" */
