/*
 * decode_oracle.c -- plain-C restatement of how ngmlr keeps its reference genome and hands a
 * window of it to the aligner (decode half).  TEST INFRASTRUCTURE ONLY: nothing
 * in ngmlr_amd/, include/ or the timed part of bench.py links or calls it.
 *
 * Pinned: tools/make_golden.sh records, from the unmodified reference running its own test data,
 * the encoded genome (binRef + refStartPos) and every window DecodeRefSequenceExact produced for
 * an alignment (tests/golden/decode_test_*.npz; all 979 windows of test_3 under
 * oracle/_ref/golden_full/).  tests/test_decode_cpu.py checks both functions below against them.
 *
 *   decode_oracle_encode   _SequenceProvider::Init, reference src/SequenceProvider.cpp:333-386
 *                          (enc4 :76-89) and the refStartPos table of :415-424
 *   decode_oracle_window   _SequenceProvider::DecodeRefSequenceExact :493-565 over decode :475-490
 *                          (dec4 :90-113) and getChrStart :157-180
 */
#include <ctype.h>
#include <stdint.h>
#include <string.h>

static int enc4(char c) {                       /* :76-89 */
	switch (toupper((unsigned char) c)) {
	case 'A': return 0;
	case 'T': return 1;
	case 'G': return 2;
	case 'C': return 3;
	}
	return 4;
}

static char dec4(int c) {                       /* :90-104; anything else throws in the reference */
	switch (c) {
	case 0: return 'A';
	case 1: return 'T';
	case 2: return 'G';
	case 3: return 'C';
	case 4: return 'N';
	}
	return '?';
}

/* bytes decode_oracle_encode will write (what the reference's loop produces; its allocation is a
 * little larger, :321) */
uint64_t decode_oracle_encoded_bytes(int n, const uint64_t *len) {
	uint64_t b = 500;
	for (int i = 0; i < n; ++i) if (len[i] > 10) b += (len[i] + 1) / 2 + 500;   /* minRefSeqLen = 10, SequenceProvider.h:79 */
	return b;
}

/* returns the number of sequences kept; starts gets kept + 1 entries (the last one is the
 * artificial upper bound of :423-424); *nibbles = binRefIndex after :386 */
int decode_oracle_encode(int n, const char *const *seq, const uint64_t *len, uint8_t *bin, uint64_t *nibbles, uint64_t *starts) {
	uint64_t idx = 0;
	const int sp = enc4('N');
	for (int i = 0; i < 500; ++i) bin[idx++] = (uint8_t) ((sp << 4) | sp);          /* :337-342 */
	int kept = 0;
	uint64_t last_len = 0;
	for (int i = 0; i < n; ++i) {
		if (!(len[i] > 10)) continue;                                                 /* :348 */
		starts[kept++] = idx * 2;                                                     /* SeqStart, :349 */
		last_len = len[i];
		const char *r = seq[i];
		for (uint64_t k = 0; k < len[i] / 2 * 2; k += 2) bin[idx++] = (uint8_t) ((enc4(r[k]) << 4) | enc4(r[k + 1]));   /* :358-362 */
		if (len[i] & 1) bin[idx++] = (uint8_t) ((enc4(r[len[i] - 1]) << 4) | sp);   /* :363-367 */
		for (int q = 0; q < 500; ++q) bin[idx++] = (uint8_t) ((sp << 4) | sp);      /* :369-374 */
	}
	*nibbles = idx * 2;
	if (kept) starts[kept] = starts[kept - 1] + last_len + 1000;                      /* :424 */
	return kept;
}

static char nibble_at(const uint8_t *bin, uint64_t p) {     /* decode(), :475-490: output char k is the nibble of position start + k */
	const uint8_t b = bin[p >> 1];
	return dec4((p & 1) ? (b & 0xF) : (b >> 4));
}

/* DecodeRefSequenceExact(sequence, startPosition, sequenceLength, corridor = 0) -- the only form the
 * aligner's caller uses (src/AlignmentBuffer.cpp:215).  starts: n_starts entries incl. the upper bound. */
int decode_oracle_window(const uint8_t *bin, const uint64_t *starts, int n_starts, uint64_t startPosition, int64_t sequenceLength, char *sequence) {
	if (sequenceLength <= 0) return 0;
	if (startPosition >= starts[n_starts - 1]) return 0;            /* :496 GetConcatRefLen guard (approximated by the upper bound) */
	memset(sequence, 'x', (size_t) sequenceLength);                 /* :501 */
	/* getChrStart, :157-180: first start > position; inside the 1000 N in front of it -> the next chromosome */
	int up = 0;
	while (up < n_starts && !(starts[up] > startPosition)) up++;
	if (up < n_starts && starts[up] - startPosition < 1000) up++;
	if (up <= 0 || up >= n_starts) { sequence[sequenceLength - 1] = '\0'; return 1; }
	const uint64_t chr_start = starts[up - 1], chr_end = starts[up] - 1000;
	uint64_t decodeStart = startPosition;                            /* halfCorridor = 0 */
	const uint64_t endPosition = startPosition + (uint64_t) sequenceLength;
	uint64_t decodeEnd = endPosition;
	if (endPosition > chr_end) decodeEnd -= endPosition - chr_end;   /* :514-519 */
	int64_t off = 0;
	int go = 1;
	if (decodeStart < chr_start) {                                   /* :530-541 (the :521 branch needs halfCorridor > start) */
		if (decodeEnd > chr_start) { off = (int64_t) (chr_start - decodeStart); decodeStart = chr_start; }
		else go = 0;
	}
	if (go) {
		const uint64_t n = (decodeStart & 1) + 2 * ((decodeEnd - decodeStart + 1) / 2);   /* chars decode() writes */
		for (uint64_t k = 0; k < n; ++k)
			if (off + (int64_t) k < sequenceLength) sequence[off + (int64_t) k] = nibble_at(bin, decodeStart + k);
	}
	sequence[sequenceLength - 1] = '\0';                             /* :562 */
	return 1;
}
