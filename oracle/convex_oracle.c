/*
 * convex_oracle.c -- plain-C CPU restatement of ngmlr's convex-gap banded
 * Smith-Waterman hot path (ConvexAlignFast over AlignmentMatrixFast).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path;
 * it is never linked into, called from, or used as a fallback by the product
 * (ngmlr_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it (see oracle_abi.h).
 *
 * Parity pinning: the reference ships no golden CIGAR/score vectors for this
 * path, so this restatement is pinned against
 *   (1) oracle/_ref -- the reference's own ConvexAlignFast compiled from
 *       /root/reference/src (tests/test_oracle_cpu.py, thousands of seeded tiles),
 *   (2) tests/golden/ -- tiles recorded from the unmodified reference binary on its
 *       own test data plus the SURVEY Appendix D known answers.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/).  The arithmetic is IEEE binary32 with every * and +
 * rounded separately (build with -ffp-contract=off, see oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "oracle_abi.h"

/* direction codes, src/AlignmentMatrixFast.h:15-24 */
enum { OP_I = 1, OP_D = 2, OP_S = 4, OP_EQ = 7, OP_X = 8, OP_STOP = 10 };

/* src/AlignmentMatrixFast.h:34-54 (score, short indelRun, char direction) */
typedef struct {
	float score;
	int16_t run;
	int8_t dir;
} cell_t;

typedef struct {
	float mat, mis, go_read, go_ref, gext, gext_min, gdecay;
	uint64_t max_matrix_mb; /* src/IConfig.h:47 (10000) */
	int use_spec_fill;      /* 0: restate the SSE path that runs; 1: scalar spec */
} oracle_t;

static const cell_t EMPTY = { 0.0f, 0, OP_STOP };

static inline float fminf_std(float a, float b) { /* std::min(a, b) */
	return (b < a) ? b : a;
}
static inline float fmaxf_std(float a, float b) { /* std::max(a, b) */
	return (a < b) ? b : a;
}

/* Working state of one alignment (AlignmentMatrixFast members). */
typedef struct {
	int W, H;
	const int32_t *off, *len;
	uint64_t *row_base;     /* offsetInMatrix, src/AlignmentMatrixFast.cpp:39-44 */
	int8_t *dirs;           /* directionMatrix, 1 byte / cell */
	cell_t *cur, *last;     /* rolling rows */
	int cur_off, cur_len, last_off, last_len;
	int have_last;
} matrix_t;

/* src/AlignmentMatrixFast.h:74-89 getElementUp */
static inline cell_t get_up(const matrix_t *m, int x, int y) {
	if (y < 0 || x < 0) return EMPTY;
	if (x < m->last_off || x >= m->last_off + m->last_len) return EMPTY;
	return m->last[x - m->last_off];
}

/* src/AlignmentMatrixFast.h:97-111 getElementCurr */
static inline cell_t get_cur(const matrix_t *m, int x, int y) {
	if (y < 0 || x < 0) return EMPTY;
	if (x < m->cur_off || x >= m->cur_off + m->cur_len) return EMPTY;
	return m->cur[x - m->cur_off];
}

/* src/AlignmentMatrixFast.cpp:185-195 getDirection (read side) */
static inline int get_dir(const matrix_t *m, int x, int y) {
	if (y < 0 || y > m->H - 1 || x < 0) return OP_STOP;
	if (x < m->off[y] || x >= m->off[y] + m->len[y]) return OP_STOP;
	return m->dirs[m->row_base[y] + (uint64_t) (x - m->off[y])];
}

static inline void set_dir(matrix_t *m, int x, int y, int d) {
	m->dirs[m->row_base[y] + (uint64_t) (x - m->off[y])] = (int8_t) d;
}

/* src/AlignmentMatrixFast.cpp:197-211 prepareLine */
static void prepare_line(matrix_t *m, int y) {
	cell_t *tmp = m->last;
	m->last = m->cur;
	m->last_off = m->cur_off;
	m->last_len = m->cur_len;
	m->cur = tmp;
	m->cur_off = m->off[y];
	m->cur_len = m->len[y];
}

typedef struct {
	int best_ref_index, best_read_index;
	int qend, qstart, ref_position;
	int alignment_offset;
	float curr_max;
} fwd_t;

/* convex extension penalty, src/ConvexAlignFast.cpp:672-674 */
static inline float ext_pen(const oracle_t *o, int run) {
	return fminf_std(o->gext_min, o->gext + (float) run * o->gdecay);
}

/* The scalar cell update: src/ConvexAlignFast.cpp:651-763 (== :1182-1275). */
static inline void cell_spec(const oracle_t *o, matrix_t *m, fwd_t *f,
		const char *ref, char qc, int x, int y) {
	float diag_score = get_up(m, x - 1, y - 1).score;
	cell_t up = get_up(m, x, y - 1);
	cell_t left = get_cur(m, x - 1, y);

	int eq = (qc == ref[x]);
	float diag_cell = diag_score + (eq ? o->mat : o->mis);
	float up_cell, left_cell;
	int ins_run = 0, del_run = 0;

	if (up.dir == OP_I) {
		ins_run = up.run;
		up_cell = (up.score == 0) ? 0 : up.score + ext_pen(o, ins_run);
	} else {
		up_cell = up.score + o->go_read;
	}
	if (left.dir == OP_D) {
		del_run = left.run;
		left_cell = (left.score == 0) ? 0 : left.score + ext_pen(o, del_run);
	} else {
		left_cell = left.score + o->go_ref;
	}

	float max_cell = 0;
	max_cell = fmaxf_std(left_cell, max_cell);
	max_cell = fmaxf_std(diag_cell, max_cell);
	max_cell = fmaxf_std(up_cell, max_cell);

	cell_t *c = &m->cur[x - m->cur_off];
	if (del_run > 0 && max_cell == left_cell) {
		c->score = max_cell; c->dir = OP_D; c->run = (int16_t) (del_run + 1);
	} else if (ins_run > 0 && max_cell == up_cell) {
		c->score = max_cell; c->dir = OP_I; c->run = (int16_t) (ins_run + 1);
	} else if (max_cell == diag_cell) {
		c->score = max_cell; c->dir = eq ? OP_EQ : OP_X; c->run = 0;
	} else if (max_cell == left_cell) {
		c->score = max_cell; c->dir = OP_D; c->run = 1;
	} else if (max_cell == up_cell) {
		c->score = max_cell; c->dir = OP_I; c->run = 1;
	} else {
		c->score = 0; c->dir = OP_STOP; c->run = 0;
	}
	set_dir(m, x, y, c->dir);

	if (max_cell > f->curr_max) { /* src/ConvexAlignFast.cpp:758-763 */
		f->curr_max = max_cell;
		f->best_ref_index = x;
		f->best_read_index = y;
	}
}

/* Scalar specification of the forward fill: src/ConvexAlignFast.cpp:606-774. */
static float fill_spec(const oracle_t *o, matrix_t *m, fwd_t *f, const char *ref, const char *qry) {
	f->curr_max = -1.0f;
	for (int y = 0; y < m->H; ++y) {
		prepare_line(m, y);
		int x0 = m->off[y];
		for (int x = x0; x < x0 + m->len[y]; ++x) {
			if (x >= m->W || x < 0) continue;
			cell_spec(o, m, f, ref, qry[y], x, y);
		}
	}
	f->qend = (m->H - f->best_read_index) - 1;
	if (m->H == 0) f->best_read_index = f->best_ref_index = 0;
	return f->curr_max;
}

/*
 * The forward fill that actually runs: fwdFillMatrixSSESimple,
 * src/ConvexAlignFast.cpp:914-1287, restated lane by lane.  Per row: blocks of
 * SIMD_LEVEL=4 cells take their up/diag candidates "vector style" (:950-1101)
 * with the relaxed ins-extend test `up.indelRun > 0` (:1084), then a serial
 * left fix-up with the relaxed `left.indelRun > 0` test (:1103-1174); the last
 * <= 12 cells of every row are then recomputed by the scalar update (:1179-1277).
 */
static float fill_sse_semantics(const oracle_t *o, matrix_t *m, fwd_t *f, const char *ref, const char *qry) {
	enum { SIMD = 4 };
	f->curr_max = -1.0f;
	for (int y = 0; y < m->H; ++y) {
		prepare_line(m, y);
		int xOffset = m->off[y];
		char qc = qry[y];
		int xMax = xOffset + m->len[y];
		if (m->W < xMax) xMax = m->W;
		int xStart = xOffset > 0 ? xOffset : 0;

		for (int x = xStart; x < xMax - SIMD; x += SIMD) {
			float score_t[SIMD], up_run_f[SIMD];
			int dir_t[SIMD];
			float run_t[SIMD];
			for (int j = 0; j < SIMD; ++j) {
				/* protected and unprotected loads (:966-993) return the same cells */
				cell_t up = get_up(m, x + j, y - 1);
				float diag_score = get_up(m, x + j - 1, y - 1).score;
				int eq = ((float) qc == (float) ref[x + j]);
				float diag_cell = diag_score + (eq ? o->mat : o->mis);
				float up_run = (float) up.run;
				float up_cell;
				if (up.dir == OP_I) { /* :1008-1016 */
					up_cell = (up.score == 0.0f) ? 0.0f
							: up.score + fminf_std(o->gext_min, o->gext + up_run * o->gdecay);
				} else {
					up_cell = up.score + o->go_read;
				}
				/* _mm_max_ps(a,b) = a > b ? a : b  (:1040) */
				float t = (up_cell > diag_cell) ? up_cell : diag_cell;
				float max_cell = (0.0f > t) ? 0.0f : t;
				int dir = OP_STOP;
				float run = 0.0f;
				if (max_cell == up_cell) { dir = OP_I; run = 1.0f; }              /* :1049-1055 */
				if (max_cell == diag_cell) { dir = eq ? OP_EQ : OP_X; run = 0.0f; } /* :1068-1076 */
				if (up_run > 0.0f && max_cell == up_cell) { dir = OP_I; run = up_run + 1.0f; } /* :1084-1091 */
				score_t[j] = max_cell; dir_t[j] = dir; run_t[j] = run; up_run_f[j] = up_run;
			}
			cell_t left = get_cur(m, x - 1, y);
			for (int j = 0; j < SIMD; ++j) { /* :1104-1174 */
				float left_cell;
				if (left.dir == OP_D) {
					left_cell = (left.score == 0) ? 0 : left.score + ext_pen(o, left.run);
				} else {
					left_cell = left.score + o->go_ref;
				}
				cell_t *c = &m->cur[x + j - m->cur_off];
				c->score = score_t[j];
				c->dir = (int8_t) dir_t[j];
				c->run = (int16_t) (int32_t) run_t[j];
				int d = dir_t[j];
				if (left_cell >= c->score) {
					if (left.run > 0) {
						c->score = left_cell; d = OP_D; c->dir = OP_D; c->run = (int16_t) (left.run + 1);
					} else if (left_cell > c->score || d == OP_STOP || (d == OP_I && up_run_f[j] <= 0)) {
						c->score = left_cell; d = OP_D; c->dir = OP_D; c->run = 1;
					}
				}
				set_dir(m, x + j, y, d);
				if (c->score > f->curr_max) {
					f->curr_max = c->score;
					f->best_ref_index = x + j;
					f->best_read_index = y;
				}
				left = *c;
			}
		}

		int xt = xMax - SIMD - 8;
		if (xt < xStart) xt = xStart;
		for (int x = xt; x < xMax; ++x) { /* :1179-1277 */
			cell_spec(o, m, f, ref, qc, x, y);
		}
	}
	f->qend = (m->H - f->best_read_index) - 1;
	if (m->H == 0) f->best_read_index = f->best_ref_index = 0;
	return f->curr_max;
}

/* src/AlignmentMatrixFast.cpp:213-220 validPath (float arithmetic, int truncation) */
static int valid_path(const matrix_t *m, int x, int y) {
	int width = m->len[y];
	int minCorridor = (int) ((float) m->off[y] + 0.1f * (float) width);
	int maxCorridor = (int) ((float) (minCorridor + width) - 0.1f * (float) width);
	return x > minCorridor && x < maxCorridor;
}

/* src/ConvexAlignFast.cpp:335-432 revBacktrack.  bc has bc_len ints. */
static int rev_backtrack(const matrix_t *m, fwd_t *f, int *bc, int bc_len, int *hard_error) {
	if (f->best_read_index <= 0) return 0;
	int valid = 1;
	int idx = bc_len - 1;
	int elem = OP_S;
	int elem_len = f->qend;
	int cigar_len = f->qend;
	int x = f->best_ref_index, y = f->best_read_index;
	int cur;
	while ((cur = get_dir(m, x, y)) != OP_STOP) {
		if (!valid_path(m, x, y)) return 0;
		if (cur == OP_X || cur == OP_EQ) { y -= 1; x -= 1; cigar_len += 1; }
		else if (cur == OP_I) { y -= 1; cigar_len += 1; }
		else if (cur == OP_D) { x -= 1; }
		else return 0;
		if (cur == elem) {
			elem_len += 1;
		} else {
			bc[idx--] = (elem_len << 4 | elem);
			elem = cur;
			elem_len = 1;
		}
		if (idx < 0) { *hard_error = 1; return 0; } /* reference: throw 1 (:405-408) */
	}
	bc[idx--] = (elem_len << 4 | elem);
	bc[idx--] = ((y + 1) << 4 | OP_S);
	cigar_len += (y + 1);
	f->ref_position = x + 1;
	f->qstart = y + 1;
	f->alignment_offset = idx + 1;
	if (m->H != cigar_len) valid = 0;
	return valid;
}

/* src/ConvexAlignFast.cpp:21-27 */
static int popcount32(uint32_t i) {
	i = i - ((i >> 1) & 0x55555555);
	i = (i & 0x33333333) + ((i >> 2) & 0x33333333);
	return (int) ((((i + (i >> 4)) & 0x0F0F0F0F) * 0x01010101) >> 24);
}

typedef struct {
	char *cigar; int cigar_cap;
	char *md; int md_cap;        /* grows (checkMdBufferLength, :100-110) */
	int32_t *nm; int nm_len;     /* PositionNM triples, grows (:76-98) */
	int nm_index;
} text_t;

static void md_reserve(text_t *t, int md_offset, int min_diff) {
	if (md_offset > (int) (t->md_cap * 0.9f) || min_diff > (t->md_cap - md_offset)) {
		int ncap = t->md_cap * 2;
		while (min_diff > ncap - md_offset) ncap *= 2; /* the reference doubles once; never short in practice */
		t->md = (char *) realloc(t->md, (size_t) ncap + 64);
		t->md_cap = ncap;
	}
}

/* src/ConvexAlignFast.cpp:76-98 addPosition */
static void add_position(text_t *t, int posInRef, int posInRead, int Yi) {
	if (posInRead > 16 && posInRef > 16) {
		if (t->nm_index >= t->nm_len) {
			int nlen = t->nm_len * 2;
			t->nm = (int32_t *) realloc(t->nm, sizeof(int32_t) * 3 * (size_t) nlen);
			memset(t->nm + 3 * t->nm_len, 0, sizeof(int32_t) * 3 * (size_t) (nlen - t->nm_len));
			t->nm_len = nlen;
		}
		t->nm[3 * t->nm_index + 0] = posInRef - 16;
		t->nm[3 * t->nm_index + 1] = posInRead - 16;
		t->nm[3 * t->nm_index + 2] = Yi;
		t->nm_index += 1;
	}
}

/* src/ConvexAlignFast.cpp:112-333 convertCigar.  refSeq already advanced to ref_position. */
static int convert_cigar(const char *refSeq, const int *bc, int bc_len, const fwd_t *f,
		int extQStart, int extQEnd, oracle_align_out *out, text_t *t, int *hard_error) {
	uint32_t buffer = 0;
	int posInRef = 0, posInRead = 0;
	int cigarOpCount = 0;
	int exactAlignmentLength = 0;
	int finalCigarLength = 0;
	int cigar_offset = 0, md_offset = 0;
	int idx = f->alignment_offset;

	out->sv_type = 0;
	out->qstart = (bc[idx] >> 4) + extQStart;
	if (out->qstart > 0) {
		cigar_offset += sprintf(t->cigar + cigar_offset, "%d%c", out->qstart, 'S');
		cigarOpCount += 1;
		finalCigarLength += out->qstart;
	}
	posInRead = bc[idx] >> 4;
	out->first_ref = posInRef;
	out->first_read = posInRead;

	int matches = 0, alignmentLength = 0;
	int cigar_m_length = 0, md_eq_length = 0, ref_index = 0;
	const uint32_t maxIndelLength = 1;
	int Yi = 0;

	for (int j = idx + 1; j < bc_len - 1; ++j) {
		int op = bc[j] & 15;
		int oplen = bc[j] >> 4;
		alignmentLength += oplen;
		switch (op) {
		case OP_X:
			cigar_m_length += oplen;
			for (int k = 0; k < oplen; ++k) {
				md_reserve(t, md_offset, 100);
				md_offset += sprintf(t->md + md_offset, "%d", md_eq_length);
				md_eq_length = 0;
				md_offset += sprintf(t->md + md_offset, "%c", refSeq[ref_index++]);
				buffer = buffer << 1;
				buffer = buffer | 1;
				Yi = popcount32(buffer);
				add_position(t, posInRef++, posInRead++, Yi);
			}
			exactAlignmentLength += oplen;
			break;
		case OP_EQ:
			cigar_m_length += oplen;
			md_eq_length += oplen;
			matches += oplen;
			for (int k = 0; k < oplen; ++k) {
				buffer = buffer << 1;
				Yi = popcount32(buffer);
				add_position(t, posInRef++, posInRead++, Yi);
			}
			ref_index += oplen;
			exactAlignmentLength += oplen;
			break;
		case OP_D:
			if (cigar_m_length > 0) {
				cigar_offset += sprintf(t->cigar + cigar_offset, "%d%c", cigar_m_length, 'M');
				cigarOpCount += 1;
				finalCigarLength += cigar_m_length;
				cigar_m_length = 0;
			}
			cigar_offset += sprintf(t->cigar + cigar_offset, "%d%c", oplen, 'D');
			cigarOpCount += 1;
			md_reserve(t, md_offset, 100 + oplen);
			md_offset += sprintf(t->md + md_offset, "%d", md_eq_length);
			md_eq_length = 0;
			t->md[md_offset++] = '^';
			for (int k = 0; k < oplen; ++k) {
				t->md[md_offset++] = refSeq[ref_index++];
				buffer = buffer << 1;
				if ((uint32_t) k < maxIndelLength) {
					buffer = buffer | 1;
					Yi = (Yi + 1 > 0) ? Yi + 1 : 0;
				}
				add_position(t, posInRef++, posInRead, Yi);
			}
			exactAlignmentLength += oplen;
			break;
		case OP_I:
			if (cigar_m_length > 0) {
				cigar_offset += sprintf(t->cigar + cigar_offset, "%d%c", cigar_m_length, 'M');
				cigarOpCount += 1;
				finalCigarLength += cigar_m_length;
				cigar_m_length = 0;
			}
			cigar_offset += sprintf(t->cigar + cigar_offset, "%d%c", oplen, 'I');
			cigarOpCount += 1;
			finalCigarLength += oplen;
			for (int k = 0; k < oplen; ++k) {
				buffer = buffer << 1;
				if ((uint32_t) k < maxIndelLength) {
					buffer = buffer | 1;
					Yi = (Yi + 1 > 0) ? Yi + 1 : 0;
				}
				posInRead += 1;
			}
			exactAlignmentLength += oplen;
			break;
		default:
			*hard_error = 1; /* reference: throw 1 (:274-276) */
			return -1;
		}
	}

	md_reserve(t, md_offset, 100);
	md_offset += sprintf(t->md + md_offset, "%d", md_eq_length);
	if (cigar_m_length > 0) {
		cigar_offset += sprintf(t->cigar + cigar_offset, "%d%c", cigar_m_length, 'M');
		cigarOpCount += 1;
		finalCigarLength += cigar_m_length;
		cigar_m_length = 0;
	}
	out->qend = (bc[bc_len - 1] >> 4) + extQEnd;
	if (out->qend > 0) {
		cigar_offset += sprintf(t->cigar + cigar_offset, "%d%c", out->qend, 'S');
		cigarOpCount += 1;
	}
	finalCigarLength += out->qend;

	out->identity = matches * 1.0f / alignmentLength;
	t->cigar[cigar_offset] = '\0';
	t->md[md_offset] = '\0';
	out->nm = alignmentLength - matches;
	out->alignment_length = exactAlignmentLength;
	out->last_ref = posInRef;
	out->last_read = posInRead;
	out->cigar_op_count = cigarOpCount;
	out->cigar_len = cigar_offset;
	out->md_len = md_offset;
	return finalCigarLength;
}

void *oracle_create(const float p[6]) {
	oracle_t *o = (oracle_t *) calloc(1, sizeof(oracle_t));
	o->mat = p[0]; o->mis = p[1];
	o->go_read = p[2]; o->go_ref = p[2]; /* src/ConvexAlignFast.cpp:39-40 */
	o->gext = p[3]; o->gext_min = p[4]; o->gdecay = p[5];
	o->max_matrix_mb = 10000;
	o->use_spec_fill = 0;
	return o;
}

void oracle_destroy(void *h) { free(h); }

const char *oracle_kind(void) { return "port"; }

/* Port-only knob for tests: 1 = scalar spec fill (:606-774), 0 = SSE-path semantics. */
void oracle_port_set_spec_fill(void *h, int on) { ((oracle_t *) h)->use_spec_fill = on; }

/* Port-only: forward results of the last call (for kernel-level parity checks). */
static __thread int g_last_best_x, g_last_best_y, g_last_ref_position, g_last_qstart, g_last_qend;
static __thread float g_last_fill_score;      /* curr_max of the last call's forward fill, valid path or not */
float oracle_port_last_fill_score(void) { return g_last_fill_score; }
static __thread int *g_last_ops; static __thread int g_last_nops;
/* Port-only: run-length ops (len<<4|op, forward order, clips excluded) of the last valid call. */
int oracle_port_last_ops(int32_t *out, int32_t cap) {
	int n = g_last_nops < cap ? g_last_nops : cap;
	if (out && n > 0) memcpy(out, g_last_ops, sizeof(int) * (size_t) n);
	return g_last_nops;
}
void oracle_port_last_fwd(int32_t out[5]) {
	out[0] = g_last_best_x; out[1] = g_last_best_y; out[2] = g_last_ref_position;
	out[3] = g_last_qstart; out[4] = g_last_qend;
}

/* src/ConvexAlignFast.cpp:452-559 SingleAlign */
int oracle_align(void *h, const char *ref, const char *qry,
		const int32_t *row_offset, const int32_t *row_length, int32_t height,
		int32_t ext_qstart, int32_t ext_qend, oracle_align_out *out,
		char *cigar, char *md, int32_t text_cap, int32_t *nm_triples, int32_t nm_cap) {
	const oracle_t *o = (const oracle_t *) h;
	int rc = 0;
	memset(out, 0, sizeof(*out));
	out->score = -1.0f;
	out->ret = -1;
	if (text_cap > 0) { cigar[0] = '\0'; md[0] = '\0'; }

	matrix_t m;
	memset(&m, 0, sizeof(m));
	m.W = (int) strlen(ref);
	m.H = (int) strlen(qry);
	m.off = row_offset;
	m.len = row_length;

	/* AlignmentMatrixFast::prepare, src/AlignmentMatrixFast.cpp:30-60 */
	m.row_base = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (height > 0 ? height : 1));
	uint64_t matrixSize = 0;
	int maxLen = 0;
	for (int i = 0; i < height; ++i) {
		m.row_base[i] = matrixSize;
		matrixSize += (uint64_t) (int64_t) row_length[i];
		if (row_length[i] > maxLen) maxLen = row_length[i];
	}
	int allocated = ((uint64_t) ((float) matrixSize / 1000.0f / 1000.0f) < o->max_matrix_mb);
	if (allocated) {
		m.dirs = (int8_t *) malloc((size_t) matrixSize + 16);
		m.cur = (cell_t *) malloc(sizeof(cell_t) * (size_t) (maxLen + 1));
		m.last = (cell_t *) malloc(sizeof(cell_t) * (size_t) (maxLen + 1));
		for (int i = 0; i <= maxLen; ++i) { m.cur[i] = EMPTY; m.last[i] = EMPTY; }

		fwd_t f;
		memset(&f, 0, sizeof(f));
		float score = o->use_spec_fill ? fill_spec(o, &m, &f, ref, qry)
				: fill_sse_semantics(o, &m, &f, ref, qry);

		g_last_fill_score = score;
		int bc_len = 200000; /* defaultMaxBinaryCigarLength, :36 */
		if (bc_len < m.H) bc_len = m.H + 1; /* :480-485 */
		int *bc = (int *) malloc(sizeof(int) * (size_t) bc_len);
		int hard = 0;
		int valid = rev_backtrack(&m, &f, bc, bc_len, &hard);
		g_last_best_x = f.best_ref_index; g_last_best_y = f.best_read_index;
		g_last_ref_position = f.ref_position; g_last_qstart = f.qstart; g_last_qend = f.qend;
		if (hard) rc = -1;
		g_last_nops = 0;
		if (valid) {
			int nops = (bc_len - 1) - (f.alignment_offset + 1);
			g_last_ops = (int *) realloc(g_last_ops, sizeof(int) * (size_t) (nops > 0 ? nops : 1));
			memcpy(g_last_ops, bc + f.alignment_offset + 1, sizeof(int) * (size_t) (nops > 0 ? nops : 0));
			g_last_nops = nops;
			text_t t;
			t.cigar_cap = m.H * 4 + 64;
			t.cigar = (char *) malloc((size_t) t.cigar_cap + 64);
			t.md_cap = m.H * 4 > 128 ? m.H * 4 : 128;
			t.md = (char *) malloc((size_t) t.md_cap + 64);
			t.md[0] = '\0';
			t.nm_len = (m.H + 1) * 2; /* src/AlignmentBuffer.cpp:277 */
			t.nm = (int32_t *) calloc((size_t) t.nm_len * 3, sizeof(int32_t));
			t.nm_index = 0;
			int hard2 = 0;
			int fin = convert_cigar(ref + f.ref_position, bc, bc_len, &f, ext_qstart, ext_qend, out, &t, &hard2);
			if (hard2) {
				rc = -1;
			} else {
				out->ret = fin;
				out->position_offset = f.ref_position;
				out->score = score;
				/* N-clip flags, :493-528 (both branches set 0x1; tests for 'X') */
				int W = m.W;
				int nCount = 0, probeCount = 0;
				int lo = f.ref_position - 100; if (lo < 0) lo = 0;
				for (int k = f.ref_position; k > lo; --k) { if (ref[k] == 'X') nCount++; probeCount++; }
				if (nCount > (probeCount * 0.8f)) out->sv_type |= 0x1;
				nCount = 0; probeCount = 0;
				int hi = out->last_ref + 100;
				if (hi > W - f.ref_position) hi = W - f.ref_position;
				for (int k = out->last_ref; k < hi; ++k) { if (ref[f.ref_position + k] == 'X') nCount++; probeCount++; }
				if (nCount > (probeCount * 0.8f)) out->sv_type |= 0x1;

				int cl = out->cigar_len < text_cap - 1 ? out->cigar_len : text_cap - 1;
				int ml = out->md_len < text_cap - 1 ? out->md_len : text_cap - 1;
				if (text_cap > 0) {
					memcpy(cigar, t.cigar, (size_t) cl); cigar[cl] = '\0';
					memcpy(md, t.md, (size_t) ml); md[ml] = '\0';
				}
				int n = out->alignment_length < t.nm_len ? out->alignment_length : t.nm_len;
				if (n > nm_cap) n = nm_cap;
				if (nm_triples) memcpy(nm_triples, t.nm, sizeof(int32_t) * 3 * (size_t) n); else n = 0;
				out->nm_count = n;
			}
			free(t.cigar); free(t.md); free(t.nm);
		}
		free(bc);
		free(m.dirs); free(m.cur); free(m.last);
	}
	free(m.row_base);
	return rc;
}
