// CPU test of the device-independent host logic (ngmlr_amd/csrc/cvx_host_logic.h): upload layout and
// packing (ragged, empty and strided tiles, multi-threaded == single-threaded), kernel-class choice,
// arena offsets and the LPT work lists.  Built with plain g++ by tests/test_host_logic_cpu.py.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "cvx_host_logic.h"

using namespace cvx;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

struct CorridorLine16 { int32_t offset, length; uint64_t offsetInMatrix; };   // the reference's 16-byte row

int main() {
	std::mt19937 rng(7);
	// ---------------------------------------------------------------- layout + packing
	const int n = 200;
	std::vector<std::string> refs(n), qrys(n);
	std::vector<std::vector<int32_t>> off(n), len(n);
	std::vector<std::vector<CorridorLine16>> lines(n);
	std::vector<cvx_tile> tiles(n);
	uint64_t n_closed_rows = 0;
	for (int i = 0; i < n; ++i) {
		const int W = (i % 17 == 0) ? 0 : (int) (rng() % 5000);
		const int H = (i % 13 == 0) ? 0 : (int) (rng() % 4000);
		refs[i].resize(W); qrys[i].resize(H);
		for (auto &c : refs[i]) c = "ACGTN"[rng() % 5];
		for (auto &c : qrys[i]) c = "ACGT"[rng() % 4];
		off[i].resize(H); len[i].resize(H); lines[i].resize(H);
		for (int y = 0; y < H; ++y) {
			// (one tile has offsets next to INT32_MIN / INT32_MAX: the 32-bit difference of the vector path would wrap)
			// two thirds of the tiles look like the reference's corridors (one width, offsets creeping along);
			// the others change width from row to row or jump by more than a byte
			off[i][y] = (int32_t) (y - 150 + (int) (rng() % 7)) + ((i % 11 == 5 && y > H / 2) ? 500 : 0) - ((i % 7 == 3) ? 2 * y : 0);
			len[i][y] = (i % 3 == 0) ? (int32_t) (300 + rng() % 70) : 340;
			if (i == 44) off[i][y] = (y & 1) ? INT32_MAX - y : INT32_MIN + y;
			lines[i][y] = {off[i][y], len[i][y], 0xdeadbeefull};
		}
		cvx_tile &t = tiles[i];
		memset(&t, 0, sizeof(t));
		t.ref = refs[i].data(); t.qry = qrys[i].data();
		t.ref_len = W; t.qry_len = H;
		if (i % 9 == 4 || i % 10 == 7) {
			// closed-form corridor: no row arrays; the rows the device would generate replace off / len for the checks below
			n_closed_rows += (uint64_t) H;
			t.corridor_width = 309 + i % 60;
			if (i % 9 == 4) {
				t.corridor_kind = CVX_CORRIDOR_AFFINE;
				t.corridor_k = (float) std::max(H, 1) * 1.0f / (float) std::max(W, 1);
				t.corridor_d = (i & 2) ? (float) t.corridor_width / 2.0f : 0.0f;
				t.corridor_right = (i & 2) ? 0.0f : 156.16f + (float) (i % 5);
			} else {
				t.corridor_kind = CVX_CORRIDOR_CONST;
				t.corridor_offset = (int) (W * -0.2);
			}
			for (int y = 0; y < H; ++y) {
				off[i][y] = t.corridor_kind == CVX_CORRIDOR_CONST ? t.corridor_offset
						: (int32_t) (((float) y - t.corridor_d) / t.corridor_k - t.corridor_right);
				len[i][y] = t.corridor_width;
			}
		} else if (i & 1) {       // CorridorLine[] passed directly, stride 16
			t.row_offset = H ? &lines[i][0].offset : nullptr;
			t.row_length = H ? &lines[i][0].length : nullptr;
			t.row_stride_bytes = 16;
		} else {
			t.row_offset = off[i].data(); t.row_length = len[i].data();
			t.row_stride_bytes = 4;
		}
	}
	UploadLayout L;
	std::vector<TileIn> tin;
	int bad = -1;
	CHECK(upload_layout(n, tiles.data(), tin, L, &bad) == kLayoutOk);
	CHECK(L.pad >= (uint64_t) kRingMax + 256);
	// arena = [pad][every read][pad][every reference][pad]: two blocks in tile order
	uint64_t qo = L.qry_base, so = L.ref_base, ro = 0;
	CHECK(L.qry_base == L.pad && L.pad % 256 == 0 && L.ref_base % 256 == 0 && L.ref_base >= L.qry_base + L.qry_bytes + L.pad);
	for (int i = 0; i < n; ++i) {
		CHECK(tin[i].ref_off == so); so += tiles[i].ref_len;
		CHECK(tin[i].qry_off == qo); qo += tiles[i].qry_len;
		// only corridors that came as row arrays own a slice of the rows arena (closed forms are evaluated in registers)
		if (tiles[i].corridor_kind == CVX_CORRIDOR_ROWS) { CHECK(tin[i].row_off == ro); ro += tiles[i].qry_len; }
		else CHECK(tin[i].row_off == 0);
		CHECK(tin[i].H == tiles[i].qry_len && tin[i].W == tiles[i].ref_len);
	}
	CHECK(qo == L.qry_base + L.qry_bytes && so == L.ref_base + L.ref_bytes && so + L.pad + 64 == L.seq_total && ro == L.arena_rows);
	CHECK(L.arena_rows == L.n_rows - n_closed_rows);
	CHECK(!L.qry_contig && !L.ref_contig);                                           // every tile has its own std::string
	CHECK(L.wprefix.size() == (size_t) n + 1 && L.wprefix[0] == 0);

	CHECK(L.rsrc.size() == (size_t) n && L.delta_total >= L.n_rows - n_closed_rows);
	std::vector<uint8_t> a(L.seq_total, 0xAA), b(L.seq_total, 0x55);
	std::vector<uint8_t> da(L.delta_total + 4, 0x11), db(L.delta_total + 4, 0x22);
	std::vector<RowSrc> sa = L.rsrc, sb = L.rsrc;
	upload_zero_pads(L, a.data());
	upload_zero_pads(L, b.data());
	RowOverflow oa;
	upload_pack(0, n, tiles.data(), tin, a.data(), da.data(), sa, oa);              // one thread
	std::vector<RowOverflow> ob(8);
	std::atomic<int> oslot{0};
	parallel_ranges(n, L.wprefix, 7, [&](int bg, int en) { upload_pack(bg, en, tiles.data(), tin, b.data(), db.data(), sb, ob[oslot++]); });
	CHECK(a == b);                                                                   // every byte defined, same result
	for (uint64_t k = 0; k < L.pad; ++k) if (a[k] != 0 || a[L.seq_total - 1 - k] != 0) { CHECK(!"pads zeroed"); break; }
	// the misfits' rows get their place in the verbatim buffer (what stage_upload does after the parallel phase)
	auto place = [&](std::vector<RowOverflow> &lists, std::vector<RowSrc> &rs, std::vector<RowDesc> &rowsx) {
		uint64_t at = 0;
		for (RowOverflow &o : lists) {
			uint64_t r = 0;
			for (int32_t ti : o.tiles) { rs[ti].src_off = at + r; r += (uint64_t) tin[ti].H; }
			rowsx.insert(rowsx.end(), o.rows.begin(), o.rows.end());
			at += o.rows.size();
		}
	};
	std::vector<RowDesc> xa, xb;
	std::vector<RowOverflow> la(1, oa);
	place(la, sa, xa);
	place(ob, sb, xb);
	int n_explicit = 0, n_delta = 0, n_closed = 0;
	for (int i = 0; i < n; ++i) {
		CHECK(memcmp(a.data() + tin[i].ref_off, refs[i].data(), refs[i].size()) == 0);
		CHECK(memcmp(a.data() + tin[i].qry_off, qrys[i].data(), qrys[i].size()) == 0);
		CHECK(sa[i].fmt == sb[i].fmt);
		if (tiles[i].corridor_kind != CVX_CORRIDOR_ROWS) { CHECK(sa[i].fmt == (tiles[i].corridor_kind == CVX_CORRIDOR_AFFINE ? kRowsAffine : kRowsConst)); n_closed++; }
		else (sa[i].fmt == kRowsExplicit ? n_explicit : n_delta) += 1;
		const int H = tiles[i].qry_len;
		std::vector<RowDesc> ra((size_t) H + 1), rb((size_t) H + 1);
		expand_rows_host(sa[i], H, da.data(), xa.data(), ra.data());                 // = expand_rows_kernel
		expand_rows_host(sb[i], H, db.data(), xb.data(), rb.data());
		for (int y = 0; y < H; ++y) {
			if (ra[y].off != off[i][y] || ra[y].len != len[i][y] || rb[y].off != off[i][y] || rb[y].len != len[i][y]) { CHECK(!"rows survive the one-byte form"); break; }
		}
	}
	CHECK(n_delta > 0 && n_explicit > 0 && n_closed > 10);                           // every form exercised
	// ranges handed to the threads tile [0, n) exactly once
	{
		std::vector<int> seen(n, 0);
		std::vector<std::pair<int, int>> got(64, {-1, -1});
		std::atomic<int> slot{0};
		parallel_ranges(n, L.wprefix, 16, [&](int bg, int en) { got[slot++] = {bg, en}; });
		for (auto &g : got) if (g.first >= 0) for (int i = g.first; i < g.second; ++i) seen[i]++;
		for (int i = 0; i < n; ++i) if (seen[i] != 1) { CHECK(!"range cover"); break; }
	}
	// malformed input is refused with the index of the offender
	{
		std::vector<cvx_tile> t2(tiles.begin(), tiles.begin() + 5);
		t2[3].row_stride_bytes = 6;
		CHECK(upload_layout(5, t2.data(), tin, L, &bad) == kLayoutMalformed && bad == 3);
		t2[3].row_stride_bytes = 4; t2[2].qry_len = 10; t2[2].row_offset = nullptr;
		CHECK(upload_layout(5, t2.data(), tin, L, &bad) == kLayoutMalformed && bad == 2);
		CHECK(upload_layout(0, nullptr, tin, L, &bad) == kLayoutOk && L.n_rows == 0 && L.seq_total == L.ref_base + L.pad + 64);
		// a malformed closed form, and sequences that do lie back to back (one arena, tile order)
		std::vector<cvx_tile> t3(tiles.begin(), tiles.begin() + 3);
		t3[1].corridor_kind = CVX_CORRIDOR_AFFINE; t3[1].corridor_k = 0.0f; t3[1].corridor_width = 300;
		CHECK(upload_layout(3, t3.data(), tin, L, &bad) == kLayoutMalformed && bad == 1);
		// ... and forms whose (float -> int) row offset would leave the int32 range or is not finite: refused, never
		// evaluated (undefined on the host, saturating on the device)
		t3[1].corridor_k = 1.0f; t3[1].corridor_d = 0.0f; t3[1].corridor_right = 1.0f;
		CHECK(upload_layout(3, t3.data(), tin, L, &bad) == kLayoutOk);
		t3[1].corridor_right = 3.0e9f;
		CHECK(upload_layout(3, t3.data(), tin, L, &bad) == kLayoutMalformed && bad == 1);
		t3[1].corridor_right = 0.0f; t3[1].corridor_d = std::numeric_limits<float>::infinity();
		CHECK(upload_layout(3, t3.data(), tin, L, &bad) == kLayoutMalformed && bad == 1);
		t3[1].corridor_d = 0.0f; t3[1].corridor_k = 1.0e-9f;       // (H - d) / k beyond 2^31 for any real tile height
		CHECK(upload_layout(3, t3.data(), tin, L, &bad) == (t3[1].qry_len >= 3 ? kLayoutMalformed : kLayoutOk));
		std::string arena_q, arena_r;
		for (int i = 0; i < 3; ++i) { arena_q += qrys[i]; arena_r += refs[i]; }
		size_t aq = 0, ar = 0;
		for (int i = 0; i < 3; ++i) { t3[i] = tiles[i]; t3[i].qry = arena_q.data() + aq; t3[i].ref = arena_r.data() + ar; aq += qrys[i].size(); ar += refs[i].size(); }
		CHECK(upload_layout(3, t3.data(), tin, L, &bad) == kLayoutOk && L.qry_contig && L.ref_contig);
		// nothing is copied for a block that travels from the caller's page-locked arena
		std::vector<uint8_t> z(L.seq_total, 0xEE), dz(L.delta_total + 4, 0);
		RowOverflow oz;
		upload_pack(0, 3, t3.data(), tin, z.data(), dz.data(), L.rsrc, oz, false, false);
		bool untouched = true;
		for (uint8_t c : z) if (c != 0xEE) untouched = false;
		CHECK(untouched);
	}
	// the pool: many callers at once, every task exactly once
	{
		std::vector<std::atomic<int>> hits(64 * 50);
		for (auto &h : hits) h = 0;
		std::vector<std::thread> callers;
		for (int c = 0; c < 8; ++c) callers.emplace_back([&, c] { for (int r = 0; r < 8; ++r) PackPool::get().run(50, [&](int i) { hits[(size_t) ((c * 8 + r) * 50 + i)]++; }); });
		for (auto &t : callers) t.join();
		bool once = true;
		for (auto &h : hits) if (h != 1) once = false;
		CHECK(once);
	}

	// ---------------------------------------------------------------- host planning
	const int m = 3000;
	std::vector<TilePlan> plan(m);
	std::vector<TileIn> pin(m);
	for (int i = 0; i < m; ++i) {
		TilePlan &p = plan[i];
		p.r0 = (int) (rng() % 50); p.rend = p.r0 + 100 + (int) (rng() % 20000);
		p.need = 1 + (int) (rng() % 5000);
		p.flags = 0;
		if (i % 97 == 0) p.flags |= kPlanEmpty;
		if (i % 89 == 0) p.flags |= kPlanTooLarge;
		if (i % 53 == 0) p.flags |= kPlanIrregular;
		if (i % 31 == 0) p.flags |= kPlanWrap16;
		p.active = (i % 7 == 0) ? 12345 : rng() % 100000000ull;     // ties on purpose
		p.cells = p.active + rng() % 1000;
		pin[i].H = 10 + (int) (rng() % 20000); pin[i].W = 10 + (int) (rng() % 20000);
	}
	HostPlan hp;
	host_plan(m, plan.data(), pin.data(), nullptr, PlanTuning(), hp);
	uint64_t dir = 0, ops = 0;
	std::vector<int> where(m, -1);
	for (size_t c = 0; c < hp.cls.size(); ++c) for (int32_t t : hp.cls[c]) { CHECK(where[t] == -1); where[t] = (int) c; }
	for (int32_t t : hp.generic) { CHECK(where[t] == -1); where[t] = 1000; }
	for (int i = 0; i < m; ++i) {
		const TilePlan &p = plan[i];
		const TileRun &r = hp.trun[i];
		if (p.flags & kPlanTooLarge) { CHECK(r.skip && hp.tout[i].status == CVX_TILE_TOO_LARGE && where[i] == -1); continue; }
		if (p.flags & kPlanEmpty) { CHECK(r.skip && hp.tout[i].status == CVX_TILE_EMPTY && where[i] == -1); continue; }
		CHECK(!r.skip && hp.tout[i].status == 0 && hp.tout[i].score == -1.0f);
		CHECK(r.dir_off == dir && r.ops_off == ops);                 // arenas are dense, in tile order
		CHECK(r.nsteps == p.rend - p.r0 && r.r0 == p.r0 && r.ops_cap == pin[i].H + pin[i].W + 8);
		dir += (uint64_t) ((r.nsteps + 31) / 32) * (uint64_t) r.ring * 2ull;
		ops += (uint64_t) r.ops_cap;
		/* (gangs of waves -- rings of 384 / 576 slots -- take float-run tiles only: an int16-run tile's widest ring is one wave's 256) */
		int widest = 0;
		for (int c = 0; c < kNumClasses; ++c) if (kClasses[c].gang == 1 || !(p.flags & kPlanWrap16)) widest = std::max(widest, kClasses[c].ring());
		if ((p.flags & kPlanIrregular) || p.need > widest) {    /* (no rows given: no chaining) */
			CHECK(where[i] == 1000 && r.mnw == 0 && r.ring % 64 == 0);
			CHECK(r.ring >= ((p.flags & kPlanIrregular) ? pin[i].H : p.need));
		} else {
			CHECK(where[i] >= 0 && where[i] < 1000);
			const KernelClass &kc = kClasses[where[i] / 2];
			CHECK(kc.ring() == r.ring && kc.ring() >= p.need && r.mnw == kc.m && r.chain_blk0 == -1);
			if (where[i] / 2 > 0) CHECK(kClasses[where[i] / 2 - 1].ring() < p.need);   // smallest class that fits
			CHECK((where[i] & 1) == ((p.flags & kPlanWrap16) ? 1 : 0));
		}
	}
	CHECK(dir == hp.dir_dwords && ops == hp.ops_ints);
	for (auto &v : hp.cls)           // LPT: most cells first, index breaks ties
		for (size_t q = 1; q < v.size(); ++q) {
			const uint64_t x = plan[v[q - 1]].active, y = plan[v[q]].active;
			if (!(x > y || (x == y && v[q - 1] < v[q]))) { CHECK(!"LPT order"); break; }
		}
	// the packed-key sort equals the comparator sort, also when it has to fall back
	{
		std::vector<int32_t> v1, v2;
		for (int i = 0; i < m; ++i) { v1.push_back(i); v2.push_back(i); }
		plan[5].active = 1ull << 50;                                      // does not fit a packed key
		lpt_sort(v1, plan.data());
		std::sort(v2.begin(), v2.end(), [&](int32_t x, int32_t y) {
			const uint64_t ax = plan[x].active, ay = plan[y].active; return ax != ay ? ax > ay : x < y; });
		CHECK(v1 == v2 && v1[0] == 5);
	}
	// tuning knobs: a floor on the ring class, forced int16-run kernels
	{ PlanTuning tn; tn.min_slots = 4; tn.force_wrap = 1; host_plan(m, plan.data(), pin.data(), nullptr, tn, hp); }
	for (size_t c = 0; c < hp.cls.size(); ++c) {
		if (!hp.cls[c].empty()) CHECK((c & 1) == 1 && kClasses[c / 2].m >= 4);
	}
	// ---------------------------------------------------------------- chained tiles (row blocks)
	{
		const int H = 3001, W = 3300, w = 2100;            // slope-1 band, ~1050 live rows: no ring holds it
		std::vector<RowDesc> rows((size_t) H + 10);
		for (int y = 0; y < H; ++y) { rows[(size_t) y + 5].off = y - w / 2; rows[(size_t) y + 5].len = w; }
		TilePlan p; memset(&p, 0, sizeof(p));
		p.r0 = 0; p.rend = (H - 1) + std::min(W, (H - 1) - w / 2 + w); p.need = 1054; p.cells = (uint64_t) H * w; p.active = p.cells;
		TileIn in; memset(&in, 0, sizeof(in));
		in.H = H; in.W = W; in.row_off = 5;
		HostPlan hc;
		host_plan(1, &p, &in, rows.data(), PlanTuning(), hc);
		CHECK(hc.n_chained == 1 && hc.generic.empty() && hc.n_fast == 0);
		const TileRun &r = hc.trun[0];
		const int cc = chain_class_for(p.need, true);
		const int N = 64 * kChainClasses[cc];
		CHECK(r.ring == N && r.chain_blk0 == 0 && r.chain_nblk == (H + N - 1) / N && r.mnw == kChainClasses[cc]);
		const std::vector<ChainTask> &tk = hc.chain_tasks[(size_t) cc * 2];
		CHECK((int) tk.size() == r.chain_nblk && (int) hc.chain_blk.size() == r.chain_nblk);
		uint64_t dir = 0, bnd = 0;
		for (int g = 0; g < (int) tk.size(); ++g) {
			const ChainTask &t = tk[(size_t) g];
			CHECK(t.tile == 0 && t.y0 == g * N && t.rows == std::min(N, H - g * N) && t.blk == g && t.prev == g - 1);
			CHECK(((t.r0 - p.r0) & 31) == 0 && t.has_next == (g + 1 < (int) tk.size() ? 1 : 0));
			// every cell of the block lies inside [r0, r0 + nsteps)
			for (int y = t.y0; y < t.y0 + t.rows; ++y) {
				const int lo = std::max(0, y - w / 2), hi = std::min(W, y - w / 2 + w);
				if (hi > lo) CHECK(lo + y >= t.r0 && hi + y <= t.r0 + t.nsteps && lo + y - t.r0 < 32 + 2 * N);
			}
			CHECK(t.dir_off == dir && hc.chain_blk[(size_t) g].dir_off * 2 == dir && hc.chain_blk[(size_t) g].tblk0 == (t.r0 - p.r0) / 32);
			dir += (uint64_t) hc.chain_blk[(size_t) g].nblk32 * N * 2;
			if (g > 0) {
				const int yb = t.y0 - 1;
				CHECK(t.bnd_lo == std::max(0, yb - w / 2) && t.bnd_len == std::min(W, yb - w / 2 + w) - t.bnd_lo && t.bnd_in_off == tk[(size_t) g - 1].bnd_out_off);
			}
			CHECK(t.bnd_out_off == bnd);
			if (t.has_next) { const int yl = t.y0 + t.rows - 1; bnd += (uint64_t) (std::min(W, yl - w / 2 + w) - std::max(0, yl - w / 2)); }
		}
		CHECK(dir == hc.dir_dwords && bnd == hc.bnd_recs);
		// two wide tiles: tasks interleave by block index, every block after the one above it
		const int nblk1 = r.chain_nblk;
		TilePlan p2[2] = {p, p}; TileIn in2[2] = {in, in};
		host_plan(2, p2, in2, rows.data(), PlanTuning(), hc);
		const std::vector<ChainTask> &t2 = hc.chain_tasks[(size_t) cc * 2];
		CHECK((int) t2.size() == 2 * nblk1);
		std::vector<int> seen(hc.chain_blk.size(), 0);
		for (const ChainTask &t : t2) { CHECK(t.prev < 0 || seen[(size_t) t.prev]); seen[(size_t) t.blk] = 1; }
		CHECK(t2[0].tile == 0 && t2[1].tile == 1 && t2[2].tile == 0 && t2[2].y0 == N);
	}
	// ---------------------------------------------------------------- a long narrow tile: chained only in a batch too small to fill the device
	{
		const int H = 21000, W = 21500, w = 340;
		std::vector<RowDesc> rows((size_t) H);
		for (int y = 0; y < H; ++y) { rows[(size_t) y].off = y - w / 2; rows[(size_t) y].len = w; }
		TilePlan p; memset(&p, 0, sizeof(p));
		p.r0 = 0; p.rend = (H - 1) + std::min(W, (H - 1) - w / 2 + w); p.need = 175; p.cells = (uint64_t) H * w; p.active = p.cells;
		TileIn in; memset(&in, 0, sizeof(in));
		in.H = H; in.W = W;
		HostPlan hl;
		host_plan(1, &p, &in, rows.data(), PlanTuning(), hl);                       // 42 000 steps >= kLongTileSteps, one tile: chained
		CHECK(p.rend - p.r0 >= kLongTileSteps && hl.n_chained == 1 && hl.n_fast == 0 && hl.trun[0].mnw == 1);
		{ PlanTuning tn; tn.long_steps = 50000; host_plan(1, &p, &in, rows.data(), tn, hl); }
		CHECK(hl.n_chained == 0 && hl.n_fast == 1 && hl.trun[0].mnw == 3);          // below the (raised) threshold: an M = 3 ring
		{ PlanTuning tn; tn.small_batch = 1; host_plan(1, &p, &in, rows.data(), tn, hl); }
		CHECK(hl.n_chained == 0 && hl.n_fast == 1);                                 // not a small batch any more: whole tiles
		{ PlanTuning tn; tn.chain_m = 2; host_plan(1, &p, &in, rows.data(), tn, hl); }
		CHECK(hl.n_chained == 1 && hl.trun[0].mnw == 2 && hl.trun[0].ring == 128);   // forced block height
		CHECK(chain_class_for(175, true) == 0 && chain_class_for(4000, false) == 0); // 64-row blocks for every batch
	}
	printf(fails ? "host_logic_test: %d FAILED\n" : "host_logic_test: ok\n", fails);
	return fails ? 1 : 0;
}
