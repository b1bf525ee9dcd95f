/*
 * batching_test.cpp -- T worker threads, each aligning its share of the recorded tiles
 * through ONE shared Convex::BatchingAligner with plain blocking SingleAlign calls (the
 * way ngmlr's CS threads would); results must equal the expected values in the record
 * file, and the number of device launches must be far below the number of requests.
 */
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <string>
#include <thread>
#include <vector>

#include "batching_aligner.h"

static bool rd(FILE * f, void * p, size_t n) { return fread(p, 1, n, f) == n; }

struct Rec {
	std::string ref, qry, cigar, md;
	std::vector<CorridorLine> lines;
	int32_t eqs, eqe, ret;
	uint32_t score_bits;
	int got_ret;
	bool threw;
	Align * align;
};

int main(int argc, char ** argv) {
	if (argc < 3) { fprintf(stderr, "usage: batching_test records.bin threads [shared | handles]\n"); return 2; }
	int const T = atoi(argv[2]);
	/* "shared":  every thread constructs its own Convex::SharedAligner, as ngmlr's workers do (src/AlignmentBuffer.h:355); with
	 *            CVX_ALIAS_DEVICES=2 in the environment they are dealt over two logical devices -- two backends, two dispatchers
	 *            -- on the one physical device: the multi-device path of the class on a one-GPU box
	 * "handles": T threads, each with its OWN ConvexAlignHip handle, aligning their shares concurrently through AlignTiles
	 *            (what bench.py --gpus N does with one handle and one host thread per device) */
	std::string const mode = argc > 3 ? argv[3] : "";
	FILE * f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	std::vector<Rec *> recs;
	int32_t hdr[6];
	while (rd(f, hdr, sizeof(hdr))) {
		Rec * r = new Rec();
		r->ref.resize(hdr[1]); r->qry.resize(hdr[2]);
		rd(f, &r->ref[0], hdr[1]); rd(f, &r->qry[0], hdr[2]);
		std::vector<int32_t> off(hdr[3]), len(hdr[3]);
		rd(f, off.data(), 4 * hdr[3]); rd(f, len.data(), 4 * hdr[3]);
		r->lines.resize(hdr[3]);
		for (int i = 0; i < hdr[3]; ++i) { r->lines[i].offset = off[i]; r->lines[i].length = len[i]; r->lines[i].offsetInMatrix = 0; }
		r->eqs = hdr[4]; r->eqe = hdr[5];
		int32_t fields[11]; uint32_t ident; int32_t cl, ml, n;
		rd(f, &r->ret, 4); rd(f, &r->score_bits, 4); rd(f, fields, 44); rd(f, &ident, 4);
		rd(f, &cl, 4); rd(f, &ml, 4);
		r->cigar.resize(cl); r->md.resize(ml);
		rd(f, &r->cigar[0], cl); rd(f, &r->md[0], ml);
		rd(f, &n, 4);
		std::vector<int32_t> nm(3 * n);
		rd(f, nm.data(), 12 * n);
		int const readLength = (int) r->qry.size();
		Align * a = new Align();
		a->maxBufferLength = readLength * 4; a->maxMdBufferLength = readLength * 4;
		a->pBuffer1 = new char[a->maxBufferLength + 16]; a->pBuffer2 = new char[a->maxMdBufferLength + 16];
		a->pBuffer1[0] = '\0'; a->pBuffer2[0] = '\0';
		a->nmPerPostionLength = (readLength + 1) * 2;
		a->nmPerPosition = new PositionNM[a->nmPerPostionLength];
		r->align = a; r->got_ret = -2; r->threw = false;
		recs.push_back(r);
	}
	fclose(f);

	/* one request that must fail ALONE: a CIGAR buffer far too small for its alignment is a hard error of that
	 * tile (the reference throws, its caller drops that one alignment, src/AlignmentBuffer.cpp:454-463) */
	size_t poisoned = recs.size();
	for (size_t i = 0; i < recs.size(); ++i) if (recs[i]->ret >= 0 && recs[i]->cigar.size() > 16) { poisoned = i; break; }
	if (poisoned < recs.size()) recs[poisoned]->align->maxBufferLength = 8;

	std::vector<std::thread> th;
	if (mode == "shared" || mode == "handles") {
		if (poisoned < recs.size()) recs[poisoned]->align->maxBufferLength = (int) recs[poisoned]->qry.size() * 4;   /* (no poisoned request here) */
		std::atomic<int> devicesSeen(0);
		for (int w = 0; w < T; ++w) {
			th.emplace_back([&, w]() {
				if (mode == "shared") {
					Convex::SharedAligner mine(0, 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f);
					int const d = Convex::SharedAligner::ActiveDevices();
					int seen = devicesSeen.load();
					while (d > seen && !devicesSeen.compare_exchange_weak(seen, d)) { }
					for (size_t i = (size_t) w; i < recs.size(); i += (size_t) T) {
						Rec & r = *recs[i];
						try {
							r.got_ret = mine.SingleAlign(0, r.lines.data(), (int) r.lines.size(), r.ref.c_str(), r.qry.c_str(), *r.align, r.eqs, r.eqe, 0);
						} catch (...) { r.threw = true; }
					}
				} else {
					Convex::ConvexAlignHip mine(0, 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f);
					std::vector<Convex::ConvexAlignHip::Tile> tiles;
					std::vector<size_t> which;
					for (size_t i = (size_t) w; i < recs.size(); i += (size_t) T) {
						Rec & r = *recs[i];
						Convex::ConvexAlignHip::Tile t;
						t.corridor = r.lines.data(); t.corridorHeight = (int) r.lines.size(); t.refSeq = r.ref.c_str(); t.qrySeq = r.qry.c_str();
						t.result = r.align; t.externalQStart = r.eqs; t.externalQEnd = r.eqe; t.ret = -1; t.failed = false; t.refLen = t.qryLen = 0;
						tiles.push_back(t); which.push_back(i);
					}
					/* three launches per handle so that the handles' uploads, kernels and downloads interleave */
					size_t const third = (tiles.size() + 2) / 3;
					for (size_t b = 0; b < tiles.size(); b += third) {
						size_t const e = b + third < tiles.size() ? b + third : tiles.size();
						try {
							mine.AlignTiles(tiles.data() + b, (int) (e - b));
						} catch (...) { for (size_t i = b; i < e; ++i) recs[which[i]]->threw = true; }
					}
					for (size_t i = 0; i < tiles.size(); ++i) { recs[which[i]]->got_ret = tiles[i].ret; if (tiles[i].failed) recs[which[i]]->threw = true; }
				}
			});
		}
		for (auto & t : th) t.join();
		int bad = 0;
		for (size_t i = 0; i < recs.size(); ++i) {
			Rec & r = *recs[i];
			uint32_t sb; memcpy(&sb, &r.align->Score, 4);
			bool ok = (r.ret < 0) ? (r.got_ret == -1) : (r.got_ret == r.ret && sb == r.score_bits && r.cigar == r.align->pBuffer1 && r.md == r.align->pBuffer2);
			if (r.threw) ok = false;
			if (!ok) { bad++; fprintf(stderr, "tile %zu differs (ret %d vs %d)\n", i, r.got_ret, r.ret); }
		}
		printf("batching_test %s: %zu requests from %d threads, %d logical devices in use, %d mismatches\n", mode.c_str(), recs.size(), T,
				mode == "shared" ? devicesSeen.load() : T, bad);
		return bad ? 1 : 0;
	}

	Convex::ConvexAlignHip backend(0, 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f);
	Convex::BatchingAligner shared(&backend, T, 256, 5000);
	IAlignment * aligner = &shared;
	for (int w = 0; w < T; ++w) {
		th.emplace_back([&, w]() {
			for (size_t i = (size_t) w; i < recs.size(); i += (size_t) T) {
				Rec & r = *recs[i];
				try {
					r.got_ret = aligner->SingleAlign(0, r.lines.data(), (int) r.lines.size(), r.ref.c_str(), r.qry.c_str(),
							*r.align, r.eqs, r.eqe, 0);
				} catch (...) {
					r.threw = true;
				}
			}
			shared.WorkerDone();
		});
	}
	for (auto & t : th) t.join();
	int bad = 0;
	for (size_t i = 0; i < recs.size(); ++i) {
		Rec & r = *recs[i];
		uint32_t sb; memcpy(&sb, &r.align->Score, 4);
		bool ok = (r.ret < 0) ? (r.got_ret == -1) : (r.got_ret == r.ret && sb == r.score_bits && r.cigar == r.align->pBuffer1 && r.md == r.align->pBuffer2);
		if (i == poisoned) ok = r.threw;             /* its own hard error, in its own thread */
		else if (r.threw) ok = false;                /* nobody else may be dragged along */
		if (!ok) { bad++; fprintf(stderr, "tile %zu differs (ret %d vs %d)\n", i, r.got_ret, r.ret); }
	}
	printf("batching_test: %zu requests from %d threads in %ld launches (up to %ld in flight), %d mismatches, 1 request failed alone as it must\n",
			recs.size(), T, shared.Launches(), shared.MaxInFlight(), bad);
	if (shared.Launches() * 2 > (long) recs.size() && recs.size() > 64) { fprintf(stderr, "not batching\n"); return 1; }
	return bad ? 1 : 0;
}
