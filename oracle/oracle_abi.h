/*
 * oracle_abi.h -- C ABI shared by the two CPU checkers under oracle/.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ngmlr_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Two libraries export exactly these symbols:
 *   oracle/_ref/libcvx_oracle_ref.so   -- the reference's own ConvexAlignFast
 *        (compiled from /root/reference/src by path, see oracle/Makefile)
 *   oracle/libcvx_oracle_port.so       -- plain-C restatement (oracle/convex_oracle.c)
 *
 * One call == one ConvexAlignFast::SingleAlign (reference src/ConvexAlignFast.cpp:452-559)
 * on buffers allocated the way AlignmentBuffer::computeAlignment does
 * (reference src/AlignmentBuffer.cpp:271-278).
 */
#ifndef CVX_ORACLE_ABI_H
#define CVX_ORACLE_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int32_t ret;              /* SingleAlign return value (finalCigarLength or -1) */
	float score;              /* Align::Score */
	int32_t position_offset;  /* Align::PositionOffset */
	int32_t qstart, qend;     /* Align::QStart / QEnd */
	int32_t nm;               /* Align::NM */
	float identity;           /* Align::Identity */
	int32_t alignment_length; /* Align::alignmentLength */
	int32_t cigar_op_count;   /* Align::cigarOpCount */
	int32_t sv_type;          /* Align::svType (N-clip flags) */
	int32_t first_ref, first_read, last_ref, last_read; /* firstPosition / lastPosition */
	int32_t nm_count;         /* PositionNM triples written (= min(alignment_length, capacity)) */
	int32_t cigar_len, md_len;/* strlen of the two text buffers */
} oracle_align_out;

/* params: match, mismatch, gapOpen, gapExtend(max), gapExtendMin, gapDecay
 * (ctor order of ConvexAlignFast, reference src/ConvexAlignFast.cpp:29-43). */
void *oracle_create(const float params[6]);
void oracle_destroy(void *h);

/* ref/qry are NUL-terminated.  row_offset/row_length have `height` (= qry_len)
 * entries.  cigar/md are caller buffers of text_cap bytes each (4*qry_len+64 is
 * always enough for CIGAR; MD is truncated to text_cap-1 if it ever grows past
 * it, md_len still reports the full length).  nm_triples receives
 * (refPos, readPos, nm) int32 triples, capacity nm_cap triples.
 * Returns 0, or -1 if the reference threw. */
int oracle_align(void *h, const char *ref, const char *qry,
		const int32_t *row_offset, const int32_t *row_length, int32_t height,
		int32_t ext_qstart, int32_t ext_qend, oracle_align_out *out,
		char *cigar, char *md, int32_t text_cap, int32_t *nm_triples, int32_t nm_cap);

/* Many alignments on `n_threads` host threads inside ONE call (the CPU-baseline leg of bench.py: python threads
 * around oracle_align share the interpreter lock for their buffer handling and stop scaling at ~16 threads).
 * Tile i: ref[i] / qry[i] need not be NUL-terminated (ref_len / qry_len given; each thread makes the terminated
 * copies the reference's strlen needs), rows at row_offset[i] / row_length[i] (qry_len[i] entries).  One aligner
 * instance per thread, tiles dealt round-robin.  outs[i] as oracle_align fills it; cigar / md of tile i go to
 * text + text_off[i] and text + text_off[i] + text_cap[i] (two buffers of text_cap[i] bytes each).
 * busy_seconds[t]: time thread t spent inside SingleAlign.  Returns the number of tiles whose call threw.
 * Only the reference build exports this symbol. */
int oracle_align_many(const float params[6], int32_t n_threads, int32_t n,
		const char *const *ref, const int32_t *ref_len, const char *const *qry, const int32_t *qry_len,
		const int32_t *const *row_offset, const int32_t *const *row_length,
		oracle_align_out *outs, char *text, const uint64_t *text_off, const int32_t *text_cap, double *busy_seconds);

/* Which implementation is this: "reference" or "port". */
const char *oracle_kind(void);

#ifdef __cplusplus
}
#endif
#endif
