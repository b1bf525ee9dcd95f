/*
 * cvx_host_logic.h -- the device-independent half of the host runtime (cvx_runtime.cpp):
 * how a batch of cvx_tile is laid out and packed for upload, and how the corridor plans that
 * come back from plan_kernel are turned into kernel classes, arena offsets and work lists.
 * Header-only and free of HIP so that the CPU suite can exercise it (tests/cpp/host_logic_test.cpp).
 */
#ifndef CVX_HOST_LOGIC_H
#define CVX_HOST_LOGIC_H

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "cvx_align.h"
#include "cvx_types.h"

namespace cvx {

/* ------------------------------------------------------------------ kernel classes */

struct KernelClass {
	int m, nw;                       /* slots per lane, waves per tile */
	int ring() const { return 64 * m * nw; }
};

/* fill_ring_kernel instantiations the runtime dispatches to, smallest ring first */
static const KernelClass kClasses[] = {
	{1, 1}, {2, 1}, {3, 1}, {4, 1}, {5, 1}, {6, 1}, {8, 1}, {4, 4}, {4, 8}, {4, 16},
};
static const int kNumClasses = (int) (sizeof(kClasses) / sizeof(kClasses[0]));

/* ------------------------------------------------------------------ host threads */

/* fn(begin, end) over [0, n) on up to `threads` host threads; ranges balanced by the work
 * prefix sums (prefix[i] = work of items < i, n + 1 entries) */
template <typename F>
void parallel_ranges(int n, const std::vector<uint64_t> &prefix, int threads, F fn) {
	if (threads <= 1 || n < 2 * threads) { fn(0, n); return; }
	const uint64_t total = prefix[(size_t) n];
	std::vector<std::thread> th;
	int begin = 0;
	for (int k = 1; k <= threads && begin < n; ++k) {
		int end = n;
		if (k < threads) {
			const uint64_t target = total / (uint64_t) threads * (uint64_t) k;
			end = (int) (std::upper_bound(prefix.begin(), prefix.begin() + n + 1, target) - prefix.begin());
			if (end <= begin) end = begin + 1;
			if (end > n) end = n;
		}
		th.emplace_back(fn, begin, end);
		begin = end;
	}
	for (auto &t : th) t.join();
}

/* ------------------------------------------------------------------ upload layout + packing */

struct UploadLayout {
	uint64_t pad = 0;               /* zeroed bytes before the first and after the last sequence */
	uint64_t seq_total = 0;         /* bytes of the seq arena */
	uint64_t n_rows = 0;            /* entries of the rows arena */
	std::vector<uint64_t> wprefix;  /* packing work per tile (bytes moved), prefix sums */
};

enum { kLayoutOk = 0, kLayoutMalformed = 1, kLayoutTooLarge = 2 };

/* Validates the tiles and assigns arena offsets (TileIn).  On kLayoutMalformed *bad is the
 * offending tile; kLayoutTooLarge: more than 4 GiB of bases (32-bit sequence offsets). */
inline int upload_layout(int n, const cvx_tile *tiles, std::vector<TileIn> &tin, UploadLayout &L, int *bad) {
	uint64_t seq_bytes = 0, n_rows = 0;
	int64_t max_hw = 0;
	for (int i = 0; i < n; ++i) {
		const cvx_tile &t = tiles[i];
		if (t.ref_len < 0 || t.qry_len < 0 || (t.ref_len > 0 && !t.ref) || (t.qry_len > 0 && !t.qry) ||
				(t.qry_len > 0 && (!t.row_offset || !t.row_length)) || (t.row_stride_bytes & 3) || t.row_stride_bytes < 4) {
			if (bad) *bad = i;
			return kLayoutMalformed;
		}
		seq_bytes += (uint64_t) t.ref_len + (uint64_t) t.qry_len;
		n_rows += (uint64_t) t.qry_len;
		max_hw = std::max<int64_t>(max_hw, (int64_t) t.ref_len + t.qry_len);
	}
	L.pad = (uint64_t) max_hw + kRingMax + 256;
	L.seq_total = seq_bytes + 2 * L.pad + 64;
	L.n_rows = n_rows;
	if (L.seq_total >= 0xFFFF0000ull) return kLayoutTooLarge;
	tin.resize((size_t) n);
	L.wprefix.assign((size_t) n + 1, 0);
	uint64_t so = L.pad, ro = 0;
	for (int i = 0; i < n; ++i) {
		const cvx_tile &t = tiles[i];
		TileIn &ti = tin[(size_t) i];
		ti.ref_off = (uint32_t) so;
		so += (uint64_t) t.ref_len;
		ti.qry_off = (uint32_t) so;
		so += (uint64_t) t.qry_len;
		ti.W = t.ref_len;
		ti.H = t.qry_len;
		ti.row_off = ro;
		ti.reserved = 0;
		ro += (uint64_t) t.qry_len;
		L.wprefix[(size_t) i + 1] = L.wprefix[(size_t) i] + (uint64_t) t.ref_len + 9ull * (uint64_t) t.qry_len + 64;
	}
	return kLayoutOk;
}

/* the kernels prefetch a little past either end of a tile: both pads must be defined */
inline void upload_zero_pads(const UploadLayout &L, uint8_t *hseq) {
	memset(hseq, 0, (size_t) L.pad);
	memset(hseq + (size_t) (L.seq_total - L.pad - 64), 0, (size_t) L.pad + 64);
}

/* copies tiles [begin, end) into the staging arenas (callable from several threads at once) */
inline void upload_pack(int begin, int end, const cvx_tile *tiles, const std::vector<TileIn> &tin,
		uint8_t *hseq, RowDesc *hrows) {
	for (int i = begin; i < end; ++i) {
		const cvx_tile &t = tiles[i];
		const TileIn &ti = tin[(size_t) i];
		if (t.ref_len) memcpy(hseq + ti.ref_off, t.ref, (size_t) t.ref_len);
		if (t.qry_len) memcpy(hseq + ti.qry_off, t.qry, (size_t) t.qry_len);
		RowDesc *dst = hrows + ti.row_off;
		const char *po = (const char *) t.row_offset;
		const char *pl = (const char *) t.row_length;
		const size_t stride = (size_t) t.row_stride_bytes;
		for (int y = 0; y < t.qry_len; ++y) {
			RowDesc rd;
			memcpy(&rd.off, po + (size_t) y * stride, 4);
			memcpy(&rd.len, pl + (size_t) y * stride, 4);
			dst[y] = rd;
		}
	}
}

/* ------------------------------------------------------------------ host planning */

struct HostPlan {
	std::vector<TileRun> trun;
	std::vector<TileOut> tout;
	std::vector<std::vector<int32_t>> cls;   /* work list per kernel class x {float runs, int16 runs} */
	std::vector<int32_t> generic;            /* tiles of the catch-all kernel */
	uint64_t dir_dwords = 0, ops_ints = 0, cells = 0, active = 0;
	int n_fast = 0;                          /* tiles taken by single-wave ring kernels */
};

/* Longest processing time first (most cells first, index as tie-break): the persistent waves
 * pull from the front.  Sorted as packed 64-bit keys when they fit (always, below a million
 * tiles of less than 2^43 cells), several times faster than comparing through the plan array. */
inline void lpt_sort(std::vector<int32_t> &v, const TilePlan *plan) {
	bool packed = v.size() < (1u << 20);
	if (packed) {
		std::vector<uint64_t> keys(v.size());
		for (size_t q = 0; q < v.size() && packed; ++q) {
			const uint64_t a = plan[(size_t) v[q]].active;
			if (a >= (1ull << 43) || (uint32_t) v[q] >= (1u << 20)) packed = false;
			keys[q] = (((1ull << 43) - 1 - a) << 20) | (uint64_t) (uint32_t) v[q];
		}
		if (packed) {
			std::sort(keys.begin(), keys.end());
			for (size_t q = 0; q < v.size(); ++q) v[q] = (int32_t) (keys[q] & ((1u << 20) - 1));
			return;
		}
	}
	std::sort(v.begin(), v.end(), [&](int32_t x, int32_t y) {
		const uint64_t ax = plan[(size_t) x].active, ay = plan[(size_t) y].active;
		return ax != ay ? ax > ay : x < y;
	});
}

/* Kernel class, arena offsets and (sorted) work lists of every tile from its corridor plan.
 * tune_min_slots / tune_force_wrap: the CVX_TUNE_* knobs of the runtime (0 = off). */
inline void host_plan(int n, const TilePlan *plan, const TileIn *tin, int tune_min_slots, int tune_force_wrap, HostPlan &hp) {
	hp.trun.assign((size_t) n, TileRun());
	hp.tout.assign((size_t) n, TileOut());
	hp.cls.assign((size_t) kNumClasses * 2, std::vector<int32_t>());
	hp.generic.clear();
	hp.dir_dwords = hp.ops_ints = hp.cells = hp.active = 0;
	hp.n_fast = 0;
	for (int i = 0; i < n; ++i) {
		const TilePlan &p = plan[(size_t) i];
		TileRun &r = hp.trun[(size_t) i];
		TileOut &o = hp.tout[(size_t) i];
		memset(&r, 0, sizeof(r));
		memset(&o, 0, sizeof(o));
		o.score = -1.0f;
		hp.cells += p.cells;
		r.skip = 1;
		if (p.flags & kPlanTooLarge) { o.status = CVX_TILE_TOO_LARGE; continue; }
		if (p.flags & kPlanEmpty) { o.status = CVX_TILE_EMPTY; continue; }
		int k = -1;
		if (!(p.flags & kPlanIrregular)) {
			for (int c = 0; c < kNumClasses; ++c)
				if (kClasses[c].ring() >= p.need && kClasses[c].m * kClasses[c].nw >= tune_min_slots) { k = c; break; }
		}
		int64_t ring;
		if (k < 0) {
			/* catch-all kernel: ring = need (regular, too wide for registers) or one slot per
			 * row PLUS ONE (irregular row starts): slot 0 reads its "up" neighbour from the last
			 * slot of the ring, which must therefore never hold a row -- with a ring of exactly H
			 * slots row 0 would see row H-1's live cells instead of the empty element
			 * (getElement(x, -1), src/AlignmentMatrixFast.h:74-111) */
			const int64_t want = (p.flags & kPlanIrregular) ? (int64_t) tin[(size_t) i].H + 1 : (int64_t) p.need;
			ring = ((want > 0 ? want : 1) + 63) / 64 * 64;
			const uint64_t dd = (uint64_t) ((p.rend - p.r0 + 31) / 32) * (uint64_t) ring * 2ull;
			if (ring > (1 << 30) || dd > (4ull << 30)) { o.status = CVX_TILE_UNSUPPORTED; continue; }  /* > 16 GiB of codes */
		} else {
			ring = kClasses[k].ring();
		}
		r.skip = 0;
		r.ring = (int32_t) ring;
		r.r0 = p.r0;
		r.nsteps = p.rend - p.r0;
		r.dir_off = hp.dir_dwords;
		hp.dir_dwords += (uint64_t) ((r.nsteps + 31) / 32) * (uint64_t) r.ring * 2ull;
		r.mnw = k < 0 ? 0 : (kClasses[k].m | (kClasses[k].nw << 8));
		r.ops_cap = tin[(size_t) i].H + tin[(size_t) i].W + 8;
		r.ops_off = hp.ops_ints;
		hp.ops_ints += (uint64_t) r.ops_cap;
		hp.active += p.active;
		if (k < 0) { hp.generic.push_back(i); continue; }
		if (kClasses[k].nw == 1) hp.n_fast++;
		hp.cls[(size_t) k * 2 + (((p.flags & kPlanWrap16) || tune_force_wrap) ? 1 : 0)].push_back(i);
	}
	for (auto &v : hp.cls) lpt_sort(v, plan);
}

}  // namespace cvx

#endif
