/*
 * ref_wrap.cpp -- exposes the REFERENCE's own Convex::ConvexAlignFast through
 * oracle_abi.h.  Test infrastructure only (see oracle_abi.h).
 *
 * Built by oracle/Makefile together with the reference sources *where they lie*
 * (/root/reference/src/ConvexAlignFast.cpp, AlignmentMatrixFast.cpp); nothing is
 * copied into this repository and the output goes to oracle/_ref/ (git-ignored).
 */
#include <chrono>
#include <cstring>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "ConvexAlignFast.h"   /* -I/root/reference/src */
#include "IConfig.h"

#include "oracle_abi.h"

/* The one global the two reference translation units need (src/IConfig.h:361). */
IConfig *_config = new IConfig();

extern "C" {

void *oracle_create(const float p[6]) {
	return new Convex::ConvexAlignFast(0, p[0], p[1], p[2], p[3], p[4], p[5]);
}

void oracle_destroy(void *h) {
	delete static_cast<Convex::ConvexAlignFast *>(h);
}

const char *oracle_kind(void) {
	return "reference";
}

int oracle_align(void *h, const char *ref, const char *qry,
		const int32_t *row_offset, const int32_t *row_length, int32_t height,
		int32_t ext_qstart, int32_t ext_qend, oracle_align_out *out,
		char *cigar, char *md, int32_t text_cap, int32_t *nm_triples, int32_t nm_cap) {
	Convex::ConvexAlignFast *aligner = static_cast<Convex::ConvexAlignFast *>(h);
	int const readLength = (int) strlen(qry);

	CorridorLine *lines = new CorridorLine[height > 0 ? height : 1];
	for (int i = 0; i < height; ++i) {
		lines[i].offset = row_offset[i];
		lines[i].length = row_length[i];
		lines[i].offsetInMatrix = 0;
	}

	/* Buffers as the one caller allocates them (src/AlignmentBuffer.cpp:271-278). */
	Align a;
	a.maxBufferLength = readLength * 4;
	a.maxMdBufferLength = readLength * 4;
	a.pBuffer1 = new char[a.maxBufferLength + 16];
	a.pBuffer2 = new char[a.maxMdBufferLength + 16];
	a.pBuffer1[0] = '\0';
	a.pBuffer2[0] = '\0';
	a.nmPerPostionLength = (readLength + 1) * 2;
	a.nmPerPosition = new PositionNM[a.nmPerPostionLength];

	int rc = 0;
	int ret = -1;
	try {
		ret = aligner->SingleAlign(0, lines, height, ref, qry, a, ext_qstart, ext_qend, 0);
	} catch (...) {
		rc = -1;
	}

	memset(out, 0, sizeof(*out));
	out->ret = ret;
	out->score = a.Score;
	out->position_offset = a.PositionOffset;
	out->qstart = a.QStart;
	out->qend = a.QEnd;
	out->nm = a.NM;
	out->identity = a.Identity;
	out->alignment_length = a.alignmentLength;
	out->cigar_op_count = a.cigarOpCount;
	out->sv_type = a.svType;
	out->first_ref = a.firstPosition.refPosition;
	out->first_read = a.firstPosition.readPosition;
	out->last_ref = a.lastPosition.refPosition;
	out->last_read = a.lastPosition.readPosition;
	out->cigar_len = (int) strlen(a.pBuffer1);
	out->md_len = (int) strlen(a.pBuffer2);
	if (text_cap > 0) {
		int cl = out->cigar_len < text_cap - 1 ? out->cigar_len : text_cap - 1;
		int ml = out->md_len < text_cap - 1 ? out->md_len : text_cap - 1;
		memcpy(cigar, a.pBuffer1, cl); cigar[cl] = '\0';
		memcpy(md, a.pBuffer2, ml); md[ml] = '\0';
	}
	int n = 0;
	if (ret >= 0 && nm_triples != 0) {
		n = a.alignmentLength < a.nmPerPostionLength ? a.alignmentLength : a.nmPerPostionLength;
		if (n > nm_cap) n = nm_cap;
		for (int i = 0; i < n; ++i) {
			nm_triples[3 * i + 0] = a.nmPerPosition[i].refPosition;
			nm_triples[3 * i + 1] = a.nmPerPosition[i].readPosition;
			nm_triples[3 * i + 2] = a.nmPerPosition[i].nm;
		}
	}
	out->nm_count = n;

	a.clearBuffer();
	a.clearNmPerPosition();
	delete[] lines;
	return rc;
}


int oracle_align_many(const float params[6], int32_t n_threads, int32_t n,
		const char *const *ref, const int32_t *ref_len, const char *const *qry, const int32_t *qry_len,
		const int32_t *const *row_offset, const int32_t *const *row_length,
		oracle_align_out *outs, char *text, const uint64_t *text_off, const int32_t *text_cap, double *busy_seconds) {
	if (n_threads < 1) n_threads = 1;
	std::vector<int> threw((size_t) n_threads, 0);
	auto work = [&](int t) {
		void *h = oracle_create(params);
		std::string r, q;
		double busy = 0.0;
		for (int i = t; i < n; i += n_threads) {
			r.assign(ref[i], (size_t) ref_len[i]);
			q.assign(qry[i], (size_t) qry_len[i]);
			char *cig = text + text_off[i];
			char *md = cig + text_cap[i];
			auto t0 = std::chrono::steady_clock::now();
			int rc = oracle_align(h, r.c_str(), q.c_str(), row_offset[i], row_length[i], qry_len[i], 0, 0, &outs[i],
					cig, md, text_cap[i], 0, 0);
			busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (rc != 0) threw[(size_t) t]++;
		}
		if (busy_seconds) busy_seconds[t] = busy;
		oracle_destroy(h);
	};
	std::vector<std::thread> th;
	for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
	work(0);
	for (auto &x : th) x.join();
	int bad = 0;
	for (int v : threw) bad += v;
	return bad;
}

}
