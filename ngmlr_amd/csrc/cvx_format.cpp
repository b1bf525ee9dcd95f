/*
 * cvx_format.cpp -- host text stage of the hot path: run-length ops -> SAM CIGAR,
 * MD, NM, identity, per-position mismatch profile and the N-clip flags.
 *
 * Behavioural contract = reference src/ConvexAlignFast.cpp:112-333 (convertCigar),
 * :76-98 (addPosition), :21-27 (32-bit window popcount), :493-528 (N-clip flags).
 * Written against the forward-ordered op list the device backtrack emits instead of
 * the reference's back-filled binaryCigar buffer.  Pure host code; no HIP, no oracle.
 */
#include <emmintrin.h>
#include <cstdint>
#include <cstring>

#include "cvx_align.h"

namespace {

/* bounded text sink: keeps counting past the capacity so the caller can resize */
struct Sink {
	char *p;
	int cap;
	int n;
	void ch(char c) {
		if (n < cap - 1) p[n] = c;
		++n;
	}
	void num(int v) {
		char tmp[12];
		int k = 0;
		unsigned u = v < 0 ? (unsigned) (-(long long) v) : (unsigned) v;
		do { tmp[k++] = (char) ('0' + u % 10); u /= 10; } while (u);
		if (v < 0) ch('-');
		while (k) ch(tmp[--k]);
	}
	void finish() {
		if (cap > 0) p[n < cap - 1 ? n : cap - 1] = '\0';
	}
};

struct NmSink {
	int32_t *p;
	int cap;
	int n;
	/* one entry per reference-consuming column once both coordinates passed 16 */
	void add(int posInRef, int posInRead, int yi) {
		if (posInRead > 16 && posInRef > 16) {
			if (p && n < cap) {
				p[3 * n + 0] = posInRef - 16;
				p[3 * n + 1] = posInRead - 16;
				p[3 * n + 2] = yi;
			}
			++n;
		}
	}
	/* `count` consecutive match columns starting at (posInRef, posInRead), all with the same yi */
	void add_run(int posInRef, int posInRead, int count, int yi) {
		int j0 = 17 - posInRead;
		if (17 - posInRef > j0) j0 = 17 - posInRef;
		if (j0 < 0) j0 = 0;
		if (j0 >= count) return;
		if (p) {
			int32_t *o = p + 3 * (size_t) n;
			const int room = cap - n;
			const int m = count - j0 < room ? count - j0 : (room > 0 ? room : 0);
			int a = posInRef + j0 - 16, b = posInRead + j0 - 16;
			int j = 0;
			/* four (ref, read, yi) triples = three 16-byte stores (SSE2: every x86-64 has it) */
			__m128i v0 = _mm_set_epi32(a + 1, yi, b, a), v1 = _mm_set_epi32(b + 2, a + 2, yi, b + 1), v2 = _mm_set_epi32(yi, b + 3, a + 3, yi);
			const __m128i s0 = _mm_set_epi32(4, 0, 4, 4), s1 = _mm_set_epi32(4, 4, 0, 4), s2 = _mm_set_epi32(0, 4, 4, 0);
			for (; j + 4 <= m; j += 4) {
				_mm_storeu_si128(reinterpret_cast<__m128i *>(o), v0);
				_mm_storeu_si128(reinterpret_cast<__m128i *>(o + 4), v1);
				_mm_storeu_si128(reinterpret_cast<__m128i *>(o + 8), v2);
				v0 = _mm_add_epi32(v0, s0); v1 = _mm_add_epi32(v1, s1); v2 = _mm_add_epi32(v2, s2);
				o += 12;
			}
			a += j; b += j;
			for (; j < m; ++j) { o[0] = a++; o[1] = b++; o[2] = yi; o += 3; }
		}
		n += count - j0;
	}
};

inline int popcnt(uint32_t v) { return __builtin_popcount(v); }

}  // namespace

extern "C" int cvx_format_alignment(const cvx_result *r, const uint32_t *ops_arena,
		const char *ref, int32_t ref_len, int32_t qry_len,
		int32_t ext_qstart, int32_t ext_qend,
		char *cigar, int32_t cigar_cap, char *md, int32_t md_cap,
		int32_t *nm_triples, int32_t nm_cap, cvx_alignment_text *out) {
	(void) qry_len;
	if (!r || !out || cigar_cap < 0 || md_cap < 0 || (cigar_cap > 0 && !cigar) || (md_cap > 0 && !md))
		return CVX_ERR_ARG;
	memset(out, 0, sizeof(*out));
	out->ret = -1;
	out->score = -1.0f;
	if (cigar_cap > 0) cigar[0] = '\0';
	if (md_cap > 0) md[0] = '\0';
	if (r->status != CVX_TILE_OK) return CVX_OK;
	if (r->n_ops > 0 && !ops_arena) return CVX_ERR_ARG;
	if (!ref && ref_len > 0) return CVX_ERR_ARG;

	const uint32_t *ops = ops_arena + r->ops_begin;
	const char *rs = ref + r->ref_position; /* convertCigar sees refSeq + ref_position */

	Sink cg{cigar, cigar_cap, 0};
	Sink mdz{md, md_cap, 0};
	NmSink nm{nm_triples, nm_cap, 0};

	int n_cigar_ops = 0;
	int final_len = 0;
	const int qstart = r->qstart + ext_qstart;
	if (qstart > 0) { cg.num(qstart); cg.ch('S'); ++n_cigar_ops; final_len += qstart; }

	int pos_ref = 0, pos_read = r->qstart;
	out->first_ref = 0;
	out->first_read = pos_read;

	uint32_t window = 0;  /* last 32 alignment columns, 1 = mismatch or gap opening */
	int yi = 0;
	int m_run = 0;        /* pending M (= and X merged) */
	int eq_run = 0;       /* pending MD match count */
	int matches = 0, columns = 0;
	int ri = 0;           /* index into rs */

	for (int k = 0; k < r->n_ops; ++k) {
		const int op = (int) (ops[k] & 15u);
		const int len = (int) (ops[k] >> 4);
		columns += len;
		if (op == CVX_OP_EQ) {
			m_run += len; eq_run += len; matches += len;
			/* (round 6: 9 % of the alignment contexts' CPU time was this loop with a library call per column for the
			 * popcount -- tools/pcsample_report.py; after at most 32 matches the window is empty and stays so) */
			int i = 0;
			for (; i < len && window != 0u; ++i) {
				window <<= 1;
				yi = popcnt(window);
				nm.add(pos_ref++, pos_read++, yi);
			}
			if (i < len) {
				yi = 0;
				nm.add_run(pos_ref, pos_read, len - i, 0);
				pos_ref += len - i; pos_read += len - i;
			}
			ri += len;
		} else if (op == CVX_OP_X) {
			m_run += len;
			for (int i = 0; i < len; ++i) {
				mdz.num(eq_run); eq_run = 0;
				mdz.ch(rs[ri++]);
				window = (window << 1) | 1u;
				yi = popcnt(window);
				nm.add(pos_ref++, pos_read++, yi);
			}
		} else if (op == CVX_OP_D) {
			if (m_run > 0) { cg.num(m_run); cg.ch('M'); ++n_cigar_ops; final_len += m_run; m_run = 0; }
			cg.num(len); cg.ch('D'); ++n_cigar_ops;
			mdz.num(eq_run); eq_run = 0;
			mdz.ch('^');
			for (int i = 0; i < len; ++i) {
				mdz.ch(rs[ri++]);
				window <<= 1;
				if (i < 1) { window |= 1u; yi = (yi + 1 > 0) ? yi + 1 : 0; }
				nm.add(pos_ref++, pos_read, yi);
			}
		} else if (op == CVX_OP_I) {
			if (m_run > 0) { cg.num(m_run); cg.ch('M'); ++n_cigar_ops; final_len += m_run; m_run = 0; }
			cg.num(len); cg.ch('I'); ++n_cigar_ops; final_len += len;
			for (int i = 0; i < len; ++i) {
				window <<= 1;
				if (i < 1) { window |= 1u; yi = (yi + 1 > 0) ? yi + 1 : 0; }
				pos_read += 1;
			}
		} else {
			return CVX_ERR_ARG; /* reference: "Invalid cigar string", throw 1 */
		}
	}
	mdz.num(eq_run);
	if (m_run > 0) { cg.num(m_run); cg.ch('M'); ++n_cigar_ops; final_len += m_run; }
	const int qend = r->qend + ext_qend;
	if (qend > 0) { cg.num(qend); cg.ch('S'); ++n_cigar_ops; }
	final_len += qend;
	cg.finish();
	mdz.finish();

	out->ret = final_len;
	out->score = r->score;
	out->position_offset = r->ref_position;
	out->qstart = qstart;
	out->qend = qend;
	out->nm = columns - matches;
	out->identity = matches * 1.0f / columns;
	out->alignment_length = columns;
	out->cigar_op_count = n_cigar_ops;
	out->last_ref = pos_ref;
	out->last_read = pos_read;
	out->nm_count = nm.n;
	out->cigar_len = cg.n;
	out->md_len = mdz.n;

	/* N-clip flags: both tests set bit 0x1 and look for 'X' (never produced by the
	 * decoder, which emits 'N'/'x') -- reproduced as is, src/ConvexAlignFast.cpp:493-528 */
	int sv = 0;
	{
		int n_count = 0, probes = 0;
		int lo = r->ref_position - 100;
		if (lo < 0) lo = 0;
		for (int k = r->ref_position; k > lo; --k) { if (k < ref_len && ref[k] == 'X') ++n_count; ++probes; }
		if (n_count > probes * 0.8f) sv |= 0x1;
		n_count = 0; probes = 0;
		int hi = pos_ref + 100;
		if (hi > ref_len - r->ref_position) hi = ref_len - r->ref_position;
		for (int k = pos_ref; k < hi; ++k) { if (rs[k] == 'X') ++n_count; ++probes; }
		if (n_count > probes * 0.8f) sv |= 0x1;
	}
	out->sv_type = sv;
	return CVX_OK;
}

/* ---- batch form: a parallel-for over tiles (dynamic chunks of 16) ---- */
#include <atomic>
#include <thread>
#include <vector>

extern "C" int cvx_format_batch(int32_t n, const cvx_result *results, const uint32_t *ops_arena,
		const cvx_tile *tiles, const cvx_text_buffers *bufs, cvx_alignment_text *out,
		int32_t n_threads) {
	if (n < 0 || (n > 0 && (!results || !tiles || !bufs || !out))) return CVX_ERR_ARG;
	if (n == 0) return CVX_OK;
	int nt = n_threads > 0 ? n_threads : (int) std::thread::hardware_concurrency();
	if (nt < 1) nt = 1;
	if (nt > (n + 15) / 16) nt = (n + 15) / 16;
	std::atomic<int> next(0), err(CVX_OK);
	auto work = [&]() {
		for (;;) {
			const int b = next.fetch_add(16);
			if (b >= n) break;
			const int e = b + 16 < n ? b + 16 : n;
			for (int i = b; i < e; ++i) {
				const cvx_text_buffers &tb = bufs[i];
				const int rc = cvx_format_alignment(&results[i], ops_arena, tiles[i].ref, tiles[i].ref_len,
						tiles[i].qry_len, tb.ext_qstart, tb.ext_qend, tb.cigar, tb.cigar_cap, tb.md, tb.md_cap,
						tb.nm_triples, tb.nm_cap, &out[i]);
				if (rc != CVX_OK) { int ok = CVX_OK; err.compare_exchange_strong(ok, rc); }
			}
		}
	};
	std::vector<std::thread> th;
	for (int t = 1; t < nt; ++t) th.emplace_back(work);
	work();
	for (auto &t : th) t.join();
	return err.load();
}
