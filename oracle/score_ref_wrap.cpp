/*
 * score_ref_wrap.cpp -- the REFERENCE's StrippedSW (over its vendored ssw.c) behind
 * score_abi.h.  Test infrastructure only; built by oracle/Makefile from the reference
 * sources where they lie, output in oracle/_ref/ (git-ignored).
 */
#include "StrippedSW.h"
#include "score_abi.h"

ILog const * _log = 0;          /* only StrippedSW::SingleAlign logs; it is never called here */
IConfig * _config = 0;

extern "C" {
void *score_oracle_create(void) { return new StrippedSW(); }
void score_oracle_destroy(void *h) { delete static_cast<StrippedSW *>(h); }
const char *score_oracle_kind(void) { return "reference"; }
int score_oracle_batch(void *h, int n, const char *const *refs, const char *const *qrys, float *out) {
	return static_cast<StrippedSW *>(h)->BatchScore(0, n, refs, qrys, out, 0);
}
}
