/*
 * score_abi.h -- C ABI of the CPU checkers for the sub-read SCORING path
 * (StrippedSW::BatchScore / SingleScore over ssw.c).
 * TEST INFRASTRUCTURE ONLY, same rules as oracle_abi.h.
 *
 *   oracle/_ref/libscore_oracle_ref.so  reference src/StrippedSW.cpp +
 *        lib/Complete-Striped-Smith-Waterman-Library/src/ssw.c compiled where they lie
 *   oracle/libscore_oracle_port.so      plain-C restatement (oracle/score_oracle.c)
 */
#ifndef CVX_SCORE_ABI_H
#define CVX_SCORE_ABI_H
#ifdef __cplusplus
extern "C" {
#endif
void *score_oracle_create(void);
void score_oracle_destroy(void *h);
/* StrippedSW::BatchScore (reference src/StrippedSW.cpp:118-160): NUL-terminated strings */
int score_oracle_batch(void *h, int n, const char *const *refs, const char *const *qrys, float *out);
const char *score_oracle_kind(void);
#ifdef __cplusplus
}
#endif
#endif
