/*
 * cvx_sam.cpp -- SAM record assembly: the text of one alignment record from the fields
 * ngmlr's writer holds, for one record or for a batch of them on the process's pack threads.
 *
 * Behavioural contract = reference src/SAMWriter.cpp:87-224 (DoWriteReadGeneric) and :308-357
 * (DoWriteUnmappedReadGeneric): field order, the tags AS NM XI XS XE XR MD SV SA QS QE CV, the
 * "<read length>S" + CG:B:I form for CIGARs of 65 536 operations and more, hard clipping, the read
 * group, and the reference's in-place reversal of the quality string for reverse-strand records
 * (:104-106 -- it reverses the caller's buffer every time such a record is written, so a read with two
 * reverse-strand records prints its second one with the qualities forward again; reproduced).
 * The reference prints field by field through vsprintf into a 100 MB buffer per writer
 * (src/GenericReadWriter.h:40-58); this is a single pass over a bounded sink, digits by hand, libc only
 * for the two floating-point tags (XI: %g, CV: %f -- the same libc the reference prints them with).
 * Pure host code; no HIP, no oracle.
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <unordered_map>
#include <vector>

#include "cvx_align.h"

namespace cvx { void pack_pool_run(int n_tasks, const std::function<void(int)> &fn); }

namespace {

const int kReportOffset = 1;      /* src/SAMWriter.cpp:19 */

/* bounded sink that keeps counting past its capacity (the caller learns the need) */
struct Out {
	char *p;
	uint64_t cap, n;
	void ch(char c) { if (n < cap) p[n] = c; ++n; }
	/* m bytes at once (round 6: a record is 20 kB of SEQ and QUAL; a byte at a time with a capacity test each was 5 % of the
	 * alignment contexts' CPU time, tools/pcsample_report.py) */
	void put(const char *s, uint64_t m) {
		if (n < cap) { const uint64_t room = cap - n; memcpy(p + n, s, (size_t) (m < room ? m : room)); }
		n += m;
	}
	void str(const char *s) { put(s, strlen(s)); }
	void mem(const char *s, int64_t len) {            /* "%.*s": stops at a NUL like printf does */
		if (len > 0) put(s, strnlen(s, (size_t) len));
	}
	/* s[len - 1], s[len - 2], ..., s[0] */
	void rev(const char *s, int64_t len) {
		if (len <= 0) return;
		if (n < cap) { const uint64_t room = cap - n; const int64_t m = (uint64_t) len < room ? len : (int64_t) room; char *d = p + n; for (int64_t i = 0; i < m; ++i) d[i] = s[len - 1 - i]; }
		n += (uint64_t) len;
	}
	void u64(uint64_t u) {
		char tmp[24];
		int k = 0;
		do { tmp[k++] = (char) ('0' + u % 10); u /= 10; } while (u);
		while (k) ch(tmp[--k]);
	}
	void i32(int32_t v) {
		if (v < 0) { ch('-'); u64((uint64_t) (-(int64_t) v)); } else u64((uint64_t) v);
	}
	void u32(uint32_t v) { u64(v); }
};

int64_t bounded_len(const char *s, int64_t lim) {     /* what "%.*s" prints of s */
	int64_t n = 0;
	while (n < lim && s[n]) ++n;
	return n;
}

/* qual_reversed: print the quality string back to front (the state the reference's in-place reversal has
 * reached for this record); the caller's buffer is not touched here */
void record(const cvx_sam_record &r, bool qual_reversed, Out &o) {
	int flags = r.flags;
	if (!r.primary) flags |= 0x800;
	if (r.reverse) flags |= 0x10;
	const int clipped_len = r.read_length - r.qstart - r.qend;
	o.str(r.read_name); o.ch('\t');
	o.i32(flags); o.ch('\t');
	o.mem(r.ref_name, r.ref_name_len); o.ch('\t');
	o.u32(r.location + (uint32_t) kReportOffset); o.ch('\t');
	o.i32(r.mq); o.ch('\t');
	const bool long_cigar = r.bam_cigar_fix && !r.skip && r.cigar_op_count >= 0x10000;
	if (long_cigar) {
		o.i32(r.hard_clip ? clipped_len : r.read_length); o.ch('S'); o.ch('\t');
	} else {
		o.str(r.cigar); o.ch('\t');
	}
	o.str(r.mate_ref_name); o.ch('\t');
	o.u32((uint32_t) (r.mate_location + kReportOffset)); o.ch('\t');
	o.i32(r.template_length); o.ch('\t');
	if (r.hard_clip) o.mem(r.seq + r.qstart, clipped_len); else o.mem(r.seq, r.read_length);
	o.ch('\t');
	if (r.qual != 0) {
		/* the reversal covers [0, read_length) of the whole string; the clip is taken from the result */
		const int64_t from = r.hard_clip ? r.qstart : 0, want = r.hard_clip ? clipped_len : r.read_length;
		if (!qual_reversed) {
			o.mem(r.qual + from, want);
		} else {
			/* character i of the reversed string is qual[read_length - 1 - i]; "%.*s" would stop at a NUL of the reversed string */
			int64_t hi = (int64_t) r.read_length - 1 - from;              /* first source character */
			int64_t lo = (int64_t) r.read_length - from - want;           /* last one */
			if (lo < 0) lo = 0;
			if (hi >= lo) {
				if (const void *z = memrchr(r.qual + lo, 0, (size_t) (hi - lo + 1))) lo = (const char *) z - r.qual + 1;      /* stops at a NUL */
				o.rev(r.qual + lo, hi - lo + 1);
			}
		}
		o.ch('\t');
	} else {
		o.str("*\t");
	}
	if (r.rg_id != 0) { o.str("RG:Z:"); o.str(r.rg_id); o.ch('\t'); }
	o.str("AS:i:"); o.i32((int) r.score); o.ch('\t');
	o.str("NM:i:"); o.i32(r.nm); o.ch('\t');
	{
		const float identity = std::round(r.identity * 10000.0f) / 10000.0f;
		char tmp[64];
		snprintf(tmp, sizeof(tmp), "%g", identity);
		o.str("XI:f:"); o.str(tmp); o.ch('\t');
	}
	o.str("XS:i:0\t");
	o.str("XE:i:"); o.i32((int) r.score); o.ch('\t');
	o.str("XR:i:"); o.i32(clipped_len); o.ch('\t');
	o.str("MD:Z:"); o.str(r.md); o.ch('\t');
	if (r.sv_type > -1) { o.str("SV:i:"); o.i32(r.sv_type); o.ch('\t'); }
	if (r.n_others > 0) {
		o.str("SA:Z:");
		for (int i = 0; i < r.n_others; ++i) {
			const cvx_sam_other &a = r.others[i];
			o.mem(a.ref_name, a.ref_name_len); o.ch(',');
			o.i32((int32_t) (a.location + (uint32_t) kReportOffset)); o.ch(',');       /* "%d" of an unsigned sum */
			o.ch(a.reverse ? '-' : '+'); o.ch(',');
			o.str(a.cigar); o.ch(',');
			o.i32(a.mq); o.ch(',');
			o.i32(a.nm); o.ch(';');
		}
		o.ch('\t');
	}
	o.str("QS:i:"); o.i32(r.qstart); o.ch('\t');
	o.str("QE:i:"); o.i32(r.read_length - r.qend); o.ch('\t');
	{
		const int clipped = r.qstart + r.qend;
		const float covered = (r.read_length - clipped) * 100.0f / r.read_length;
		char tmp[64];
		snprintf(tmp, sizeof(tmp), "%f", covered);
		o.str("CV:f:"); o.str(tmp);
	}
	if (long_cigar) {
		/* the real CIGAR as BAM words in the CG tag: len << 4 | op */
		o.str("\tCG:B:I");
		const char *p = r.cigar;
		for (int i = 0; i < r.cigar_op_count; ++i) {
			long len = 0;
			while (*p == ' ' || *p == '\t') ++p;                      /* strtol skips blanks; a CIGAR has none */
			bool neg = false;
			if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
			while (*p >= '0' && *p <= '9') { len = len * 10 + (*p - '0'); ++p; }
			if (neg) len = -len;
			int op = 0;
			switch (*p) {
			case 'M': op = 0; break; case 'I': op = 1; break; case 'D': op = 2; break; case 'N': op = 3; break;
			case 'S': op = 4; break; case 'H': op = 5; break; case '=': op = 7; break; case 'X': op = 8; break;
			default: op = 0; break;
			}
			if (*p) ++p;
			const unsigned word = (unsigned) len << 4 | (unsigned) op;
			o.ch(','); o.i32((int32_t) word);                         /* "%d" */
		}
	}
	o.ch('\n');
}

void unmapped(const cvx_sam_unmapped &r, Out &o) {
	o.str(r.read_name); o.ch('\t');
	o.i32(r.flags | 0x4); o.ch('\t');
	if (r.ref_name) { o.mem(r.ref_name, r.ref_name_len); o.ch('\t'); } else o.str("*\t");
	o.i32(r.location + kReportOffset); o.ch('\t');
	o.str("0\t*\t");
	o.ch(r.mate_ref); o.ch('\t');
	o.i32(r.mate_location + kReportOffset); o.ch('\t');
	o.i32(r.template_length); o.ch('\t');
	o.mem(r.seq, r.read_length); o.ch('\t');
	if (r.qual != 0) o.mem(r.qual, r.read_length); else o.ch('*');
	if (r.rg_id != 0) { o.str("\tRG:Z:"); o.str(r.rg_id); }
	o.ch('\n');
}

bool reverses_qual(const cvx_sam_record &r) { return r.reverse && r.qual != 0 && r.qual[0] != '\0'; }

void reverse_in_place(char *q, int n) {
	for (int i = 0, j = n - 1; i < j; ++i, --j) { const char c = q[i]; q[i] = q[j]; q[j] = c; }
}

}  // namespace

extern "C" int cvx_sam_record_text(cvx_sam_record *r, char *out, uint64_t cap, uint64_t *len) {
	if (!r || !len || (cap > 0 && !out) || !r->read_name || !r->seq || !r->cigar || !r->md || !r->mate_ref_name ||
			(r->ref_name_len > 0 && !r->ref_name) || r->read_length <= 0 || (r->n_others > 0 && !r->others)) return CVX_ERR_ARG;
	/* the reference reverses first and prints afterwards; a call that does not fit must not leave the buffer reversed
	 * twice after the retry, so the reversal is applied only when the record went out */
	Out o{out, cap, 0};
	record(*r, reverses_qual(*r), o);
	*len = o.n;
	if (o.n > cap) return CVX_ERR_CAPACITY;
	if (reverses_qual(*r)) reverse_in_place(r->qual, r->read_length);
	return CVX_OK;
}

extern "C" int cvx_sam_unmapped_text(const cvx_sam_unmapped *r, char *out, uint64_t cap, uint64_t *len) {
	if (!r || !len || (cap > 0 && !out) || !r->read_name || !r->seq || r->read_length < 0) return CVX_ERR_ARG;
	Out o{out, cap, 0};
	unmapped(*r, o);
	*len = o.n;
	return o.n > cap ? CVX_ERR_CAPACITY : CVX_OK;
}

extern "C" int cvx_sam_batch(int32_t n, cvx_sam_record *recs, char *out, uint64_t cap, uint64_t *offsets) {
	if (n < 0 || (n > 0 && (!recs || !offsets)) || (cap > 0 && !out)) return CVX_ERR_ARG;
	if (n == 0) { if (offsets) offsets[0] = 0; return CVX_OK; }
	for (int i = 0; i < n; ++i) {
		const cvx_sam_record &r = recs[i];
		if (!r.read_name || !r.seq || !r.cigar || !r.md || !r.mate_ref_name || (r.ref_name_len > 0 && !r.ref_name) ||
				r.read_length <= 0 || (r.n_others > 0 && !r.others)) return CVX_ERR_ARG;
	}
	/* the state of every quality string at the time its record is written, as the reference's one-by-one in-place
	 * reversal would have it: parity of the reverse-strand records of the same buffer up to and including this one */
	std::vector<uint8_t> rev((size_t) n, 0);
	std::unordered_map<const char *, uint8_t> parity;
	for (int i = 0; i < n; ++i) {
		const cvx_sam_record &r = recs[i];
		if (!r.qual) continue;
		uint8_t &p = parity[r.qual];
		if (reverses_qual(r)) p ^= 1;
		rev[(size_t) i] = p;
	}
	/* pass 1: lengths; pass 2: text.  Records are independent; chunks of 64 per task. */
	const int chunk = 64, n_tasks = (n + chunk - 1) / chunk;
	cvx::pack_pool_run(n_tasks, [&](int t) {
		const int b = t * chunk, e = b + chunk < n ? b + chunk : n;
		for (int i = b; i < e; ++i) {
			Out o{0, 0, 0};
			record(recs[i], rev[(size_t) i] != 0, o);
			offsets[i + 1] = o.n;
		}
	});
	offsets[0] = 0;
	for (int i = 0; i < n; ++i) offsets[i + 1] += offsets[i];
	if (offsets[n] > cap) return CVX_ERR_CAPACITY;       /* offsets[n] = the need */
	cvx::pack_pool_run(n_tasks, [&](int t) {
		const int b = t * chunk, e = b + chunk < n ? b + chunk : n;
		for (int i = b; i < e; ++i) {
			Out o{out + offsets[i], offsets[i + 1] - offsets[i], 0};
			record(recs[i], rev[(size_t) i] != 0, o);
		}
	});
	/* leave every caller buffer in the state the reference would leave it in */
	for (auto &kv : parity) {
		if (!kv.second) continue;
		for (int i = 0; i < n; ++i) if (recs[i].qual == kv.first) { reverse_in_place(recs[i].qual, recs[i].read_length); break; }
	}
	return CVX_OK;
}
