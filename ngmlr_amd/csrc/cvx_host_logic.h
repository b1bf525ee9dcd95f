/*
 * cvx_host_logic.h -- the device-independent half of the host runtime (cvx_runtime.cpp):
 * how a batch of cvx_tile is laid out and packed for upload, and how the corridor plans that
 * come back from plan_kernel are turned into kernel classes, arena offsets and work lists.
 * Header-only and free of HIP so that the CPU suite can exercise it (tests/cpp/host_logic_test.cpp).
 */
#ifndef CVX_HOST_LOGIC_H
#define CVX_HOST_LOGIC_H

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>

#include "cvx_align.h"
#include "cvx_types.h"

namespace cvx {

/* ------------------------------------------------------------------ kernel classes */

struct KernelClass {
	int m;                           /* slots per lane */
	int gang;                        /* waves per tile (1; 2 or 3: a gang of waves on one ring, float runs only) */
	int ring() const { return 64 * m * gang; }
};

/* fill_ring_kernel instantiations the runtime dispatches whole tiles to, smallest ring first.
 * Rings of more than 256 slots need so many registers (M = 6: 133, M = 8: 170 VGPRs) that only two
 * or three waves fit a SIMD; such tiles are chained instead (ONT mix, 60 000 tiles: 5 950 -> 7 180
 * Gbp/h when everything above M = 4 goes to 64-row blocks). */
/* Round 5: rings of 384 and 576 slots as GANGS of two / three M = 3 waves in one workgroup (the lane boundary between the
 * waves goes through LDS, cvx_kernels.hip): the retry loop's doubled corridors (257-576 live rows) are whole tiles again. */
static const KernelClass kClasses[] = { {1, 1}, {2, 1}, {3, 1}, {4, 1}, {3, 2}, {3, 3} };
static const int kNumClasses = (int) (sizeof(kClasses) / sizeof(kClasses[0]));

/* Corridors with more live rows than the widest single-wave ring are cut into row blocks that run
 * as a dependency chain (kFillChain).  Smaller blocks = more blocks of a tile in flight at once
 * (about need / rows-per-block): aim for 16 or more. */
static const int kChainClasses[] = {1, 2, 4};          /* M of the chain kernels: 64, 128, 256 rows per block */
static const int kNumChainClasses = 3;
/* A batch with fewer whole-tile waves than this cannot fill the device by tile parallelism alone:
 * its chains use the smallest blocks (a tile's time is then about (2 H + w) single-wave steps,
 * whatever its width), and its very long tiles are chained even if a ring would hold them. */
static const int kSmallBatchTiles = 2048;
static const int kLongTileSteps = 32768;
/* whole tiles of at least this many steps are filled by the exact-tracking instantiation right away (cvx_runtime.cpp,
 * stage_compute), as long as their class holds at most kExactDirectMaxTiles of them */
static const int kExactDirectSteps = 65536;
static const int kExactDirectMaxTiles = 4096;
/* streaming jobs smaller than this alternate between the runtime's two stream sets (cvx_runtime.cpp, stage_compute) */
static const int kSmallJobTiles = 2048;
inline int chain_class_for(int need, bool small_batch) {
	/* measured (C5 mix, 96 tiles): 64-row blocks 100 ms, 128-row 117 ms, 256-row 174 ms -- a tile's
	 * time is its block count times the lag between neighbouring blocks, and a step of a 64-row block
	 * is the shortest.  Batches that fill the device were tried with 128-row blocks (fewer instructions
	 * per cell): ONT mix, 60 000 tiles, 7 600-7 650 Gbp/h against 7 660-7 770 with 64-row blocks, and
	 * with the blocks at raised wave priority (FillArgs::chain_prio) 64 rows win clearly (8 320-8 450
	 * against 7 580-8 040).  The larger classes stay selectable (CVX_TUNE_CHAIN_M). */
	(void) need; (void) small_batch;
	return 0;
}

/* ------------------------------------------------------------------ host threads */

/* The process's pack threads: one persistent pool shared by every handle (eight handles on an 8-device node
 * used to spawn 24 fresh threads each per call -- 192 threads on a host that behaves like ~32 cores).  Size:
 * CVX_PACK_THREADS, else min(hardware threads, 16); the calling thread works too, so a pool of size 1 has no
 * worker threads at all.  Handles that pack at the same time share the workers task by task. */
class PackPool {
public:
	static PackPool &get() {
		static PackPool pool;
		return pool;
	}
	int size() const { return n_threads_; }
	/* runs fn(0) ... fn(n_tasks - 1), returns when all are done */
	void run(int n_tasks, const std::function<void(int)> &fn) {
		if (n_tasks <= 0) return;
		if (n_tasks == 1 || n_threads_ <= 1) { for (int i = 0; i < n_tasks; ++i) fn(i); return; }
		Group g;
		g.left = n_tasks;
		{
			std::lock_guard<std::mutex> lk(m_);
			for (int i = 0; i < n_tasks; ++i) q_.push_back(Task{&g, &fn, i});
		}
		cv_.notify_all();
		/* the caller takes tasks as well (its own or another caller's: whatever is at the front) */
		for (;;) {
			Task t;
			{
				std::unique_lock<std::mutex> lk(m_);
				if (g.left == 0) break;
				if (q_.empty()) { done_.wait(lk, [&] { return g.left == 0 || !q_.empty(); }); continue; }
				t = q_.front();
				q_.pop_front();
			}
			execute(t);
		}
	}
private:
	struct Group { int left = 0; };
	struct Task { Group *g; const std::function<void(int)> *fn; int i; };
	PackPool() {
		int hw = (int) std::thread::hardware_concurrency();
		n_threads_ = std::max(1, std::min(hw > 0 ? hw : 1, 16));
		if (const char *e = getenv("CVX_PACK_THREADS")) n_threads_ = std::max(1, atoi(e));
		for (int i = 1; i < n_threads_; ++i) workers_.emplace_back([this] { loop(); });
	}
	~PackPool() {
		{ std::lock_guard<std::mutex> lk(m_); stop_ = true; }
		cv_.notify_all();
		for (auto &t : workers_) t.join();
	}
	void execute(const Task &t) {
		(*t.fn)(t.i);
		bool last;
		{ std::lock_guard<std::mutex> lk(m_); last = (--t.g->left == 0); }
		if (last) done_.notify_all();
	}
	void loop() {
		pthread_setname_np(pthread_self(), "cvx-pack");
		for (;;) {
			Task t;
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
				if (stop_ && q_.empty()) return;
				t = q_.front();
				q_.pop_front();
			}
			execute(t);
			done_.notify_all();      /* a caller waiting for work to help with */
		}
	}
	std::mutex m_;
	std::condition_variable cv_, done_;
	std::deque<Task> q_;
	std::vector<std::thread> workers_;
	bool stop_ = false;
	int n_threads_ = 1;
};

/* fn(begin, end) over [0, n) in up to `threads` ranges balanced by the work prefix sums
 * (prefix[i] = work of items < i, n + 1 entries), on the process's pack threads */
template <typename F>
void parallel_ranges(int n, const std::vector<uint64_t> &prefix, int threads, F fn) {
	if (threads <= 1 || n < 2 * threads) { fn(0, n); return; }
	const uint64_t total = prefix[(size_t) n];
	std::vector<std::pair<int, int>> ranges;
	int begin = 0;
	for (int k = 1; k <= threads && begin < n; ++k) {
		int end = n;
		if (k < threads) {
			const uint64_t target = total / (uint64_t) threads * (uint64_t) k;
			end = (int) (std::upper_bound(prefix.begin(), prefix.begin() + n + 1, target) - prefix.begin());
			if (end <= begin) end = begin + 1;
			if (end > n) end = n;
		}
		ranges.emplace_back(begin, end);
		begin = end;
	}
	PackPool::get().run((int) ranges.size(), [&](int i) { fn(ranges[(size_t) i].first, ranges[(size_t) i].second); });
}

/* ------------------------------------------------------------------ upload layout + packing */

struct UploadLayout {
	uint64_t pad = 0;               /* defined bytes before the first, between the two blocks and after the last sequence (multiple of 256) */
	uint64_t seq_total = 0;         /* bytes of the seq arena: [pad][every read][pad][every reference][pad] */
	uint64_t qry_base = 0, qry_bytes = 0;   /* arena offset / size of the block of reads */
	uint64_t ref_base = 0, ref_bytes = 0;   /* ... of the block of references (ref_base is a multiple of 256) */
	uint64_t n_rows = 0;            /* read rows of all tiles */
	uint64_t arena_rows = 0;        /* entries of the rows arena: the rows of the tiles whose corridors came as arrays (closed forms own none) */
	uint64_t delta_total = 0;       /* bytes of the row-step stream (H bytes, 4-byte aligned, per tile whose rows come as arrays) */
	std::vector<RowSrc> rsrc;       /* per tile: where its rows come from (src_off = offset into the step stream until packed) */
	bool windows = false;           /* references decoded on the device from the resident genome: their block is not uploaded */
	/* the caller's reads (references) lie back to back in tile order: the block can travel as it is, without
	 * packing, when the runtime finds it in page-locked memory (cvx_host_alloc) */
	bool qry_contig = false, ref_contig = false;
	std::vector<uint64_t> wprefix;  /* packing work per tile (bytes moved if everything is packed), prefix sums */
};

enum { kLayoutOk = 0, kLayoutMalformed = 1, kLayoutTooLarge = 2 };

/* A closed form the device and the host evaluate alike: finite parameters, a positive finite slope, and every row's
 * offset ((float) y - d) / k - right inside the int32 range with room to spare -- the float -> int conversion of
 * affine_row_offset is undefined behaviour on the host and saturating on the device beyond it.  The
 * expression is monotone in y for k > 0, so rows 0 and H bound all of them. */
inline bool affine_form_ok(float k, float d, float right, int32_t H) {
	if (!(k > 0.0f && k < 3.0e38f) || !std::isfinite(d) || !std::isfinite(right)) return false;
	const double lo = (0.0 - (double) d) / (double) k - (double) right;
	const double hi = ((double) (H > 0 ? H : 0) - (double) d) / (double) k - (double) right;
	return std::fabs(lo) < 2.0e9 && std::fabs(hi) < 2.0e9;
}

/* Validates the tiles and assigns arena offsets (TileIn).  On kLayoutMalformed *bad is the
 * offending tile; kLayoutTooLarge: more than 4 GiB of bases (32-bit sequence offsets). */
inline int upload_layout(int n, const cvx_tile *tiles, std::vector<TileIn> &tin, UploadLayout &L, int *bad, bool windows = false) {
	uint64_t ref_bytes = 0, n_rows = 0, qry_bytes = 0;
	int64_t max_hw = 0;
	bool qc = n > 0, rc = n > 0 && !windows;
	for (int i = 0; i < n; ++i) {
		const cvx_tile &t = tiles[i];
		const bool rows_given = t.corridor_kind == CVX_CORRIDOR_ROWS;
		if (t.ref_len < 0 || t.qry_len < 0 || (t.ref_len > 0 && !t.ref && !windows) || (t.qry_len > 0 && !t.qry) ||
				(rows_given && t.qry_len > 0 && (!t.row_offset || !t.row_length)) ||
				(rows_given && ((t.row_stride_bytes & 3) || t.row_stride_bytes < 4)) ||
				t.corridor_kind < CVX_CORRIDOR_ROWS || t.corridor_kind > CVX_CORRIDOR_CONST ||
				(t.corridor_kind == CVX_CORRIDOR_AFFINE && !affine_form_ok(t.corridor_k, t.corridor_d, t.corridor_right, t.qry_len)) ||
				(!rows_given && t.corridor_width < 0)) {
			if (bad) *bad = i;
			return kLayoutMalformed;
		}
		if (i > 0) {
			if (tiles[i - 1].qry + tiles[i - 1].qry_len != t.qry) qc = false;
			if (!windows && tiles[i - 1].ref + tiles[i - 1].ref_len != t.ref) rc = false;
		}
		ref_bytes += (uint64_t) t.ref_len;
		qry_bytes += (uint64_t) t.qry_len;
		n_rows += (uint64_t) t.qry_len;
		max_hw = std::max<int64_t>(max_hw, (int64_t) t.ref_len + t.qry_len);
	}
	L.pad = ((uint64_t) max_hw + kRingMax + 256 + 255) / 256 * 256;
	L.qry_base = L.pad;
	L.qry_bytes = qry_bytes;
	L.ref_base = (L.qry_base + qry_bytes + L.pad + 255) / 256 * 256;
	L.ref_bytes = ref_bytes;
	L.seq_total = L.ref_base + ref_bytes + L.pad + 64;
	L.n_rows = n_rows;
	L.windows = windows;
	L.qry_contig = qc;
	L.ref_contig = rc;
	if (L.seq_total >= 0xFFFF0000ull) return kLayoutTooLarge;
	tin.resize((size_t) n);
	L.rsrc.assign((size_t) n, RowSrc());
	L.wprefix.assign((size_t) n + 1, 0);
	uint64_t qo = L.qry_base, ro = 0, dof = 0, rso = L.ref_base;
	for (int i = 0; i < n; ++i) {
		const cvx_tile &t = tiles[i];
		TileIn &ti = tin[(size_t) i];
		ti.ref_off = (uint32_t) rso;
		rso += (uint64_t) t.ref_len;
		ti.qry_off = (uint32_t) qo;
		qo += (uint64_t) t.qry_len;
		ti.W = t.ref_len;
		ti.H = t.qry_len;
		ti.row_off = 0;
		ti.reserved = 0;
		RowSrc &rs = L.rsrc[(size_t) i];
		memset(&rs, 0, sizeof(rs));
		uint64_t row_work = 0;
		if (t.corridor_kind == CVX_CORRIDOR_AFFINE) {
			rs.fmt = kRowsAffine; rs.width = t.corridor_width; rs.k = t.corridor_k; rs.d = t.corridor_d; rs.right = t.corridor_right;
		} else if (t.corridor_kind == CVX_CORRIDOR_CONST) {
			rs.fmt = kRowsConst; rs.width = t.corridor_width; rs.off0 = t.corridor_offset;
		} else {
			rs.src_off = dof; rs.fmt = kRowsDelta8;
			ti.row_off = ro;                       /* only corridors that came as arrays own a slice of the rows arena */
			ro += (uint64_t) t.qry_len;
			dof += ((uint64_t) t.qry_len + 3) / 4 * 4;
			row_work = 9ull * (uint64_t) t.qry_len;
		}
		L.wprefix[(size_t) i + 1] = L.wprefix[(size_t) i] + (windows ? 0 : (uint64_t) t.ref_len) + (uint64_t) t.qry_len + row_work + 64;
	}
	L.delta_total = dof;
	L.arena_rows = ro;
	return kLayoutOk;
}

/* the kernels prefetch a little past either end of a tile: all three pads must be defined (host staging form) */
inline void upload_zero_pads(const UploadLayout &L, uint8_t *hseq) {
	memset(hseq, 0, (size_t) L.qry_base);
	memset(hseq + (size_t) (L.qry_base + L.qry_bytes), 0, (size_t) (L.ref_base - L.qry_base - L.qry_bytes));
	if (!L.windows) memset(hseq + (size_t) (L.ref_base + L.ref_bytes), 0, (size_t) (L.seq_total - L.ref_base - L.ref_bytes));
}

/* Step bytes of rows [1, H) of a tile whose (offset, length) arrays are plain int32 arrays (stride 4), eight
 * rows per iteration.  Returns the first row it did not finish (H when all rows fit the one-byte form; a row
 * whose width differs or whose step is outside -128..127 stops it, and the scalar loop decides). */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) inline int pack_steps_avx2(const int32_t *off, const int32_t *len, int H, int32_t w0, int8_t *dst) {
	const __m256i vw = _mm256_set1_epi32(w0);
	const __m256i lo = _mm256_set1_epi32(-129), hi = _mm256_set1_epi32(128);
	const __m256i big = _mm256_set1_epi32(1 << 30), nbig = _mm256_set1_epi32(-(1 << 30));
	const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
			0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
	int y = 1;
	for (; y + 8 <= H; y += 8) {
		const __m256i cur = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(off + y));
		const __m256i prv = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(off + y - 1));
		const __m256i l = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(len + y));
		/* with both offsets inside (-2^30, 2^30) the 32-bit difference cannot wrap (anything else: scalar loop) */
		const __m256i d = _mm256_sub_epi32(cur, prv);
		const __m256i inr = _mm256_and_si256(_mm256_and_si256(_mm256_cmpgt_epi32(cur, nbig), _mm256_cmpgt_epi32(big, cur)),
				_mm256_and_si256(_mm256_cmpgt_epi32(prv, nbig), _mm256_cmpgt_epi32(big, prv)));
		const __m256i ok = _mm256_and_si256(_mm256_and_si256(_mm256_and_si256(_mm256_cmpgt_epi32(d, lo), _mm256_cmpgt_epi32(hi, d)), _mm256_cmpeq_epi32(l, vw)), inr);
		if (_mm256_movemask_epi8(ok) != -1) break;
		const __m256i b = _mm256_shuffle_epi8(d, pick);
		const uint32_t b0 = (uint32_t) _mm256_extract_epi32(b, 0), b1 = (uint32_t) _mm256_extract_epi32(b, 4);
		memcpy(dst + y, &b0, 4);
		memcpy(dst + y + 4, &b1, 4);
	}
	return y;
}
#endif

/* Rows of the tiles that do not fit the one-byte form, collected by one packing thread. */
struct RowOverflow {
	std::vector<int32_t> tiles;        /* tile indices, in packing order */
	std::vector<RowDesc> rows;         /* their rows, back to back */
};

/* copies tiles [begin, end) into the staging arenas (callable from several threads at once).  Rows that
 * come as arrays go to the step stream `hdelta` (one byte per row: offset[y] - offset[y-1]; rsrc[i] gets row
 * 0's offset and the common width); a tile whose width changes or whose offset jumps by more than a byte is
 * marked kRowsExplicit and its rows are appended to `ovf` instead.  Closed-form corridors have nothing to
 * pack.  copy_qry / copy_ref = false: that block travels straight from the caller's page-locked memory. */
inline void upload_pack(int begin, int end, const cvx_tile *tiles, const std::vector<TileIn> &tin,
		uint8_t *hseq, uint8_t *hdelta, std::vector<RowSrc> &rsrc, RowOverflow &ovf, bool copy_qry = true, bool copy_ref = true) {
	for (int i = begin; i < end; ++i) {
		const cvx_tile &t = tiles[i];
		const TileIn &ti = tin[(size_t) i];
		if (t.ref_len && copy_ref) memcpy(hseq + ti.ref_off, t.ref, (size_t) t.ref_len);
		if (t.qry_len && copy_qry) memcpy(hseq + ti.qry_off, t.qry, (size_t) t.qry_len);
		RowSrc &rs = rsrc[(size_t) i];
		if (rs.fmt != kRowsDelta8) continue;       /* closed form: generated on the device */
		const char *po = (const char *) t.row_offset;
		const char *pl = (const char *) t.row_length;
		const size_t stride = (size_t) t.row_stride_bytes;
		const int H = t.qry_len;
		if (H <= 0) continue;
		int8_t *dst = reinterpret_cast<int8_t *>(hdelta + rs.src_off);
		int32_t prev, w0;
		memcpy(&prev, po, 4);
		memcpy(&w0, pl, 4);
		rs.off0 = prev;
		rs.width = w0;
		dst[0] = 0;
		bool fits = true;
		int y_from = 1;
#if defined(__x86_64__)
		/* plain int32 arrays with in-range first offset: eight rows per iteration (the scalar loop below finishes
		 * the tail and is the judge of any row the vector loop stopped at) */
		static const bool have_avx2 = __builtin_cpu_supports("avx2");
		if (have_avx2 && stride == 4 && H > 16) {
			y_from = pack_steps_avx2(reinterpret_cast<const int32_t *>(po), reinterpret_cast<const int32_t *>(pl), H, w0, dst);
			if (y_from > 1) memcpy(&prev, po + (size_t) (y_from - 1) * stride, 4);
		}
#endif
		for (int y = y_from; y < H; ++y) {
			int32_t o, l;
			memcpy(&o, po + (size_t) y * stride, 4);
			memcpy(&l, pl + (size_t) y * stride, 4);
			const int64_t d = (int64_t) o - (int64_t) prev;
			if (l != w0 || d < -128 || d > 127) { fits = false; break; }
			dst[y] = (int8_t) d;
			prev = o;
		}
		if (!fits) {
			rs.fmt = kRowsExplicit;
			ovf.tiles.push_back(i);
			const size_t at = ovf.rows.size();
			ovf.rows.resize(at + (size_t) H);
			for (int y = 0; y < H; ++y) {
				RowDesc rd;
				memcpy(&rd.off, po + (size_t) y * stride, 4);
				memcpy(&rd.len, pl + (size_t) y * stride, 4);
				ovf.rows[at + (size_t) y] = rd;
			}
		}
	}
}

/* what expand_rows_kernel does on the device, for the host's own needs (chain planning, tests):
 * the (offset, length) rows of tile i.  hrowsx: the explicit-rows buffer (rsrc[i].src_off indexes it). */
inline void expand_rows_host(const RowSrc &rs, int H, const uint8_t *hdelta, const RowDesc *hrowsx, RowDesc *out) {
	if (rs.fmt == kRowsExplicit) {
		if (H > 0) memcpy(out, hrowsx + rs.src_off, (size_t) H * sizeof(RowDesc));
		return;
	}
	if (rs.fmt == kRowsAffine || rs.fmt == kRowsConst) {
		for (int y = 0; y < H; ++y) {
			out[y].off = rs.fmt == kRowsConst ? rs.off0 : affine_row_offset(y, rs.d, rs.k, rs.right);
			out[y].len = rs.width;
		}
		return;
	}
	const int8_t *d = reinterpret_cast<const int8_t *>(hdelta + rs.src_off);
	int32_t o = rs.off0;
	for (int y = 0; y < H; ++y) {
		if (y > 0) o += (int32_t) d[y];
		out[y].off = o;
		out[y].len = rs.width;
	}
}

/* ------------------------------------------------------------------ host planning */

struct HostPlan {
	std::vector<TileRun> trun;
	std::vector<TileOut> tout;
	std::vector<std::vector<int32_t>> cls;   /* work list per kernel class x {float runs, int16 runs} */
	std::vector<int32_t> generic;            /* tiles of the catch-all kernel */
	/* chained tiles (row blocks): per chain class x {float runs, int16 runs} the tasks in launch
	 * order, the tiles, and the per-block tables shared by all classes */
	std::vector<std::vector<ChainTask>> chain_tasks;
	std::vector<std::vector<int32_t>> chain_tiles;
	std::vector<ChainBlk> chain_blk;
	uint64_t bnd_recs = 0;                   /* BoundaryRec entries */
	uint64_t dir_dwords = 0, ops_ints = 0, cells = 0, active = 0;
	int n_fast = 0;                          /* tiles taken whole by one wave */
	int n_chained = 0;
};

/* Row blocks of one chained tile: fills hp.chain_blk and returns the tile's tasks (block order).
 * rows = the tile's (offset, length) rows as uploaded. */
inline void plan_chain_tile(int tile, int m, const TilePlan &p, const TileIn &in, const RowDesc *rows,
		HostPlan &hp, TileRun &r, std::vector<ChainTask> &out) {
	const int N = 64 * m, H = in.H, W = in.W;
	const int nblk = (H + N - 1) / N;
	r.ring = N;
	r.r0 = p.r0;
	r.nsteps = p.rend - p.r0;
	r.dir_off = 0;
	r.mnw = m;
	r.chain_blk0 = (int32_t) hp.chain_blk.size();
	r.chain_nblk = nblk;
	auto span = [&](int y, long long &lo, long long &hi) {
		lo = rows[y].off > 0 ? rows[y].off : 0;
		hi = (long long) rows[y].off + (long long) rows[y].len;
		if (hi > W) hi = W;
		if (hi < lo) hi = lo;
	};
	uint64_t prev_out = 0;
	for (int g = 0; g < nblk; ++g) {
		const int y0 = g * N, y1 = std::min(H, y0 + N);
		long long gs_min = 0x7fffffff, ge_max = -0x7fffffff;
		for (int y = y0; y < y1; ++y) {
			long long lo, hi;
			span(y, lo, hi);
			if (hi > lo) { gs_min = std::min(gs_min, lo + y); ge_max = std::max(ge_max, hi + y); }
		}
		ChainTask t;
		memset(&t, 0, sizeof(t));
		t.tile = tile;
		t.y0 = y0;
		t.rows = y1 - y0;
		if (ge_max > gs_min) {
			/* planes of all blocks share the tile's bit phase: a block begins on a multiple of 32 steps behind the tile's first.
			 * A block below another begins at least ONE step before its first cell: the diagonal input of that cell -- the score
			 * of the row above one column to the left -- reaches the slot as the `up` score of the step before (round 5: a block
			 * whose first cell fell exactly on such a multiple took 0 for it; one block in 32, and visible only where the best
			 * path runs down the corridor's first column, i.e. in tiles validPath rejects -- found on the C5 mix, where the raw
			 * fill score of 6 of 4 096 chained tiles differed from the ring kernels' and the reference's) */
			const long long lead = g > 0 ? 1 : 0;
			t.r0 = p.r0 + (int) ((gs_min - lead - p.r0) & ~31ll);
			t.nsteps = (int) (ge_max - t.r0);
		} else {
			t.r0 = p.r0;
			t.nsteps = 0;
		}
		t.blk = r.chain_blk0 + g;
		t.prev = g > 0 ? t.blk - 1 : -1;
		if (g > 0) {
			long long lo, hi;
			span(y0 - 1, lo, hi);
			t.bnd_lo = (int32_t) lo;
			t.bnd_len = (int32_t) (hi - lo);
			t.bnd_in_off = prev_out;
		}
		t.has_next = (g + 1 < nblk) ? 1 : 0;
		{
			long long lo, hi;
			span(y1 - 1, lo, hi);
			t.bnd_out_off = hp.bnd_recs;
			prev_out = hp.bnd_recs;
			if (t.has_next) hp.bnd_recs += (uint64_t) (hi - lo);
		}
		const uint64_t nblk32 = (uint64_t) ((t.nsteps + 31) / 32);
		t.dir_off = hp.dir_dwords;
		ChainBlk cb;
		cb.dir_off = hp.dir_dwords / 2;
		cb.tblk0 = (t.r0 - p.r0) >> 5;
		cb.nblk32 = (int32_t) std::max<uint64_t>(nblk32, 1);
		hp.dir_dwords += std::max<uint64_t>(nblk32, 1) * (uint64_t) N * 2ull;
		hp.chain_blk.push_back(cb);
		out.push_back(t);
	}
}

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
 * tune: the CVX_TUNE_* knobs of the runtime. */
/* tuning / test knobs of the runtime (CVX_TUNE_* environment variables; all 0 = off) */
struct PlanTuning {
	int min_slots = 0;     /* smallest M a whole tile may use */
	int max_slots = 0;     /* largest M a whole tile may use (wider tiles are chained) */
	int force_wrap = 0;    /* route every tile to the int16-run kernels */
	int chain_m = 0;       /* 1, 2 or 4: force the row-block height class of chained tiles */
	int force_generic = 0; /* every tile to the catch-all kernel (scoring that needs its SSE-variant instantiation) */
	int long_steps = 0;    /* > 0: replaces kLongTileSteps (a tile of a small batch with at least this many steps is chained) */
	int small_batch = 0;   /* > 0: replaces kSmallBatchTiles */
	int long_need = 0;     /* > 0: replaces the 128 live rows from which a very long tile of a small batch is chained */
	int no_gangs = 0;      /* != 0: no gang classes (corridors with more than 256 live rows are chained, as before round 5) */
};

/* rows_of(i, tmp) -> the (offset, length) rows of tile i (may fill and return tmp), or an empty function /
 * nullptr result when the host has none (then nothing is chained). */
template <typename RowsOf>
inline void host_plan_rows(int n, const TilePlan *plan, const TileIn *tin, RowsOf rows_of, bool have_rows, const PlanTuning &tune, HostPlan &hp);

inline void host_plan(int n, const TilePlan *plan, const TileIn *tin, const RowDesc *rows, const PlanTuning &tune, HostPlan &hp) {
	/* rows: one arena indexed by TileIn::row_off (tests), or NULL */
	host_plan_rows(n, plan, tin, [&](int i, std::vector<RowDesc> &) { return rows + tin[(size_t) i].row_off; }, rows != nullptr, tune, hp);
}

template <typename RowsOf>
inline void host_plan_rows(int n, const TilePlan *plan, const TileIn *tin, RowsOf rows_of, bool have_rows, const PlanTuning &tune, HostPlan &hp) {
	const int tune_min_slots = tune.min_slots, tune_force_wrap = tune.force_wrap;
	hp.trun.assign((size_t) n, TileRun());
	hp.tout.assign((size_t) n, TileOut());
	hp.cls.assign((size_t) kNumClasses * 2, std::vector<int32_t>());
	hp.generic.clear();
	hp.chain_tasks.assign((size_t) kNumChainClasses * 2, std::vector<ChainTask>());
	hp.chain_tiles.assign((size_t) kNumChainClasses * 2, std::vector<int32_t>());
	hp.chain_blk.clear();
	hp.bnd_recs = 0;
	hp.dir_dwords = hp.ops_ints = hp.cells = hp.active = 0;
	hp.n_fast = 0;
	hp.n_chained = 0;
	std::vector<std::vector<std::vector<ChainTask>>> per_tile((size_t) kNumChainClasses * 2);
	int n_work = 0;
	for (int i = 0; i < n; ++i) if (!(plan[(size_t) i].flags & (kPlanTooLarge | kPlanEmpty))) n_work++;
	const bool small_batch = n_work < (tune.small_batch > 0 ? tune.small_batch : kSmallBatchTiles);
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
		const bool wrap = (p.flags & kPlanWrap16) || tune_force_wrap;
		const bool regular = !(p.flags & kPlanIrregular);
		int k = -1;
		if (regular && !tune.force_generic) {
			for (int c = 0; c < kNumClasses; ++c)
				if (kClasses[c].ring() >= p.need && kClasses[c].m >= tune_min_slots &&
						(tune.max_slots <= 0 || kClasses[c].m <= tune.max_slots) &&
						(kClasses[c].gang == 1 || (!wrap && !tune.no_gangs))) { k = c; break; }
		}
		r.skip = 0;
		r.chain_blk0 = -1;
		r.chain_nblk = 0;
		r.ops_cap = tin[(size_t) i].H + tin[(size_t) i].W + 8;
		r.ops_off = hp.ops_ints;
		hp.ops_ints += (uint64_t) r.ops_cap;
		hp.active += p.active;
		const bool long_tile = small_batch && p.need >= (tune.long_need > 0 ? tune.long_need : 128) && (p.rend - p.r0) >= (tune.long_steps > 0 ? tune.long_steps : kLongTileSteps);
		if ((k < 0 || long_tile) && regular && have_rows && tune_min_slots == 0 && !tune.force_generic) {
			/* more live rows than any ring (or one very long tile in a batch too small to fill the
			 * device with whole tiles): row blocks chained through boundary streams */
			int cc = chain_class_for(p.need, small_batch);
			for (int q = 0; q < kNumChainClasses; ++q) if (kChainClasses[q] == tune.chain_m) cc = q;
			const size_t slot = (size_t) cc * 2 + (wrap ? 1 : 0);
			per_tile[slot].emplace_back();
			std::vector<RowDesc> tmp_rows;
			plan_chain_tile(i, kChainClasses[cc], p, tin[(size_t) i], rows_of(i, tmp_rows), hp, r, per_tile[slot].back());
			hp.chain_tiles[slot].push_back(i);
			hp.n_chained++;
			continue;
		}
		int64_t ring;
		if (k < 0) {
			/* catch-all kernel, irregular row starts: one slot per row PLUS ONE: slot 0 reads its "up"
			 * neighbour from the last slot of the ring, which must therefore never hold a row -- with a
			 * ring of exactly H slots row 0 would see row H-1's live cells instead of the empty element
			 * (getElement(x, -1), src/AlignmentMatrixFast.h:74-111) */
			const int64_t want = regular ? (int64_t) p.need : (int64_t) tin[(size_t) i].H + 1;
			ring = ((want > 0 ? want : 1) + 63) / 64 * 64;
			const uint64_t dd = (uint64_t) ((p.rend - p.r0 + 31) / 32) * (uint64_t) ring * 2ull;
			if (ring > (1 << 30) || dd > (4ull << 30)) {     /* > 16 GiB of codes for one irregular tile */
				o.status = CVX_TILE_UNSUPPORTED;
				r.skip = 1;
				hp.active -= p.active;
				continue;
			}
		} else {
			ring = kClasses[k].ring();
		}
		r.ring = (int32_t) ring;
		r.r0 = p.r0;
		r.nsteps = p.rend - p.r0;
		r.dir_off = hp.dir_dwords;
		hp.dir_dwords += (uint64_t) ((r.nsteps + 31) / 32) * (uint64_t) r.ring * 2ull;
		r.mnw = k < 0 ? 0 : kClasses[k].m;
		if (k < 0) { hp.generic.push_back(i); continue; }
		hp.n_fast++;
		hp.cls[(size_t) k * 2 + (wrap ? 1 : 0)].push_back(i);
	}
	/* chain tasks in launch order: block index major, tile minor -- a wave takes tasks in this order,
	 * so all tiles advance together and the block above any task has always been taken before it */
	for (size_t slot = 0; slot < per_tile.size(); ++slot) {
		size_t deepest = 0;
		for (auto &v : per_tile[slot]) deepest = std::max(deepest, v.size());
		for (size_t g = 0; g < deepest; ++g)
			for (auto &v : per_tile[slot])
				if (g < v.size()) hp.chain_tasks[slot].push_back(v[g]);
	}
	for (auto &v : hp.cls) lpt_sort(v, plan);
}

}  // namespace cvx

#endif
