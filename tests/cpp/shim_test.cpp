/*
 * shim_test.cpp -- drives Convex::ConvexAlignHip (the IAlignment drop-in) exactly the way
 * AlignmentBuffer::computeAlignment drives the reference aligner (buffers allocated as in
 * reference src/AlignmentBuffer.cpp:271-278) on recorded tiles and compares every field
 * of Align with the expected values in the record file (written by the CPU oracle in
 * tests/test_gpu_shim.py; same record layout as tools/ref_recorder/recording_aligner.h).
 * Exit code 0 = all tiles identical.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <unistd.h>
#include <string>
#include <vector>

#include "convex_align_hip.h"
#include "stripped_sw_hip.h"

static bool rd(FILE * f, void * p, size_t n) { return fread(p, 1, n, f) == n; }

/* a crash inside the library shows where (there is no debugger on the GPU boxes) */
static void on_crash(int sig) {
	/* async-signal-safe only: the heap may be the thing that is broken */
	static char const msg[] = "shim_test: fatal signal, backtrace follows\n";
	(void) !write(2, msg, sizeof(msg) - 1);
	static void * frames[64];
	int const n = backtrace(frames, 64);
	backtrace_symbols_fd(frames, n, 2);
	_exit(128 + sig);
}

int main(int argc, char ** argv) {
	signal(SIGSEGV, on_crash);
	signal(SIGABRT, on_crash);
	if (argc < 2) { fprintf(stderr, "usage: shim_test records.bin [batch]\n"); return 2; }
	bool const batch = argc > 2;
	FILE * f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	IAlignment * aligner = new Convex::ConvexAlignHip(0, 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f);

	struct Rec {
		std::string ref, qry, cigar, md;
		std::vector<CorridorLine> lines;
		int32_t eqs, eqe, ret;
		uint32_t score_bits, ident_bits;
		int32_t fields[11];
		std::vector<int32_t> nm;
		Align * align;
	};
	std::vector<Rec *> recs;
	int32_t hdr[6];
	while (rd(f, hdr, sizeof(hdr))) {
		if (hdr[0] != 0x43565854) { fprintf(stderr, "bad magic\n"); return 2; }
		Rec * r = new Rec();
		r->ref.resize(hdr[1]); r->qry.resize(hdr[2]);
		rd(f, &r->ref[0], hdr[1]); rd(f, &r->qry[0], hdr[2]);
		std::vector<int32_t> off(hdr[3]), len(hdr[3]);
		rd(f, off.data(), 4 * hdr[3]); rd(f, len.data(), 4 * hdr[3]);
		r->lines.resize(hdr[3]);
		for (int i = 0; i < hdr[3]; ++i) { r->lines[i].offset = off[i]; r->lines[i].length = len[i]; r->lines[i].offsetInMatrix = 0; }
		r->eqs = hdr[4]; r->eqe = hdr[5];
		rd(f, &r->ret, 4); rd(f, &r->score_bits, 4); rd(f, r->fields, 44); rd(f, &r->ident_bits, 4);
		int32_t cl, ml, n;
		rd(f, &cl, 4); rd(f, &ml, 4);
		r->cigar.resize(cl); r->md.resize(ml);
		rd(f, &r->cigar[0], cl); rd(f, &r->md[0], ml);
		rd(f, &n, 4);
		r->nm.resize(3 * n);
		rd(f, r->nm.data(), 12 * n);
		recs.push_back(r);
	}
	fclose(f);

	std::vector<Convex::ConvexAlignHip::Tile> tiles(recs.size());
	for (size_t i = 0; i < recs.size(); ++i) {
		Rec & r = *recs[i];
		int const readLength = (int) r.qry.size();
		Align * a = new Align();
		a->maxBufferLength = readLength * 4;
		a->maxMdBufferLength = readLength * 4;
		a->pBuffer1 = new char[a->maxBufferLength + 16];
		a->pBuffer2 = new char[a->maxMdBufferLength + 16];
		a->pBuffer1[0] = '\0'; a->pBuffer2[0] = '\0';
		a->nmPerPostionLength = (readLength + 1) * 2;
		a->nmPerPosition = new PositionNM[a->nmPerPostionLength];
		a->svType = 1234;
		r.align = a;
		Convex::ConvexAlignHip::Tile & t = tiles[i];
		t.corridor = r.lines.data(); t.corridorHeight = (int) r.lines.size();
		t.refSeq = r.ref.c_str(); t.qrySeq = r.qry.c_str(); t.result = a;
		t.externalQStart = r.eqs; t.externalQEnd = r.eqe; t.ret = -2;
	}
	if (batch) {
		static_cast<Convex::ConvexAlignHip *>(aligner)->AlignTiles(tiles.data(), (int) tiles.size());
	} else {
		bool const verbose = getenv("SHIM_VERBOSE") != 0;
		for (size_t i = 0; i < tiles.size(); ++i) {
			if (verbose) fprintf(stderr, "tile %zu H %d W %zu\n", i, tiles[i].corridorHeight, strlen(tiles[i].refSeq));
			tiles[i].ret = aligner->SingleAlign(0, tiles[i].corridor, tiles[i].corridorHeight, tiles[i].refSeq,
					tiles[i].qrySeq, *tiles[i].result, tiles[i].externalQStart, tiles[i].externalQEnd, 0);
		}
	}
	int bad = 0, valid = 0;
	for (size_t i = 0; i < recs.size(); ++i) {
		Rec & r = *recs[i];
		Align & a = *r.align;
		int const ret = tiles[i].ret;
		char const * why = 0;
		if (r.ret < 0) {
			if (ret != -1 || a.Score != -1.0f) why = "expected invalid";
		} else {
			valid++;
			uint32_t sb, ib;
			memcpy(&sb, &a.Score, 4); memcpy(&ib, &a.Identity, 4);
			int32_t got[11] = { a.PositionOffset, a.QStart, a.QEnd, a.NM, a.alignmentLength, a.cigarOpCount, a.svType,
					a.firstPosition.refPosition, a.firstPosition.readPosition, a.lastPosition.refPosition, a.lastPosition.readPosition };
			if (ret != r.ret) why = "ret";
			else if (sb != r.score_bits) why = "score bits";
			else if (ib != r.ident_bits) why = "identity bits";
			else if (memcmp(got, r.fields, sizeof(got)) != 0) why = "Align fields";
			else if (r.cigar != a.pBuffer1) why = "CIGAR";
			else if (r.md != a.pBuffer2) why = "MD";
			else {
				int const n = (int) r.nm.size() / 3;
				for (int k = 0; k < n && !why; ++k) {
					if (k >= a.nmPerPostionLength || a.nmPerPosition[k].refPosition != r.nm[3 * k] ||
							a.nmPerPosition[k].readPosition != r.nm[3 * k + 1] || a.nmPerPosition[k].nm != r.nm[3 * k + 2]) why = "nmPerPosition";
				}
			}
			/* prepare()'s side effect on the caller's corridor */
			unsigned long acc = 0;
			for (size_t y = 0; y < r.lines.size() && !why; ++y) {
				if (r.lines[y].offsetInMatrix != acc) why = "offsetInMatrix";
				acc += (unsigned long) r.lines[y].length;
			}
		}
		if (why) { bad++; fprintf(stderr, "tile %zu: %s (ret %d vs %d)\n", i, why, ret, r.ret); }
	}
	/* scoring calls of the same plugin surface: a perfect 40-mer scores 40,
	 * an N column scores 0, too-long input scores -1 with return value 0 */
	{
		StrippedSWHip sw(0);
		IAlignment * scorer = &sw;
		char const * refs[2] = { "TTTTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTGGGG", "ACGTNACGT" };
		char const * qrys[2] = { "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT", "ACGTAACGT" };
		float sc[2] = { 0, 0 };
		if (scorer->BatchScore(0, 2, refs, qrys, sc, 0) != 2 || sc[0] != 40.0f || sc[1] != 8.0f) {
			fprintf(stderr, "scoring shim: %f %f\n", sc[0], sc[1]);
			bad++;
		}
		float one = 0;
		if (scorer->SingleScore(0, 0, refs[0], qrys[0], one, 0) != 1 || one != 40.0f) bad++;
	}
	printf("shim_test: %zu tiles, %d valid, %d mismatches (%s)\n", recs.size(), valid, bad, batch ? "AlignTiles" : "SingleAlign");
	{
		/* how many of the CorridorLine[] arrays Prepare() recognised as one of the reference builders' closed forms */
		long prepared = 0, closedForm = 0;
		Convex::ConvexAlignHip::CorridorStats(prepared, closedForm);
		printf("shim_test: %ld of %ld corridors travelled as closed forms\n", closedForm, prepared);
	}
	delete aligner;
	return bad ? 1 : 0;
}
