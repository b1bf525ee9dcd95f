/*
 * convex_align_hip.cpp -- see convex_align_hip.h.  Host-only C++; all device work is
 * behind include/cvx_align.h.  No CPU alignment path exists here: if the device
 * library reports an error the call throws, exactly like a hard error in the reference.
 */
#include "convex_align_hip.h"
#include "cvx_fiber.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef CVX_IN_NGMLR_TREE
#include "IConfig.h"       /* Config.getMaxMatrixSizeMB(), as ConvexAlignFast's ctor reads it (src/ConvexAlignFast.cpp:49) */
#endif

namespace Convex {

namespace {
std::atomic<long> g_prepared(0), g_closedForm(0);
bool const g_fitCorridors = [] { const char * e = getenv("CVX_CORRIDOR_FIT"); return !(e && atoi(e) == 0); }();
}

/* ------------------------------------------------------------------ DeviceWindows */
namespace {
struct WindowNote { char const * buf; unsigned long long position; long length; };
thread_local WindowNote tl_window = { 0, 0, 0 };
void const * g_binRef = 0;
unsigned long long g_nNibbles = 0;
std::vector<uint64_t> g_startTable;
std::atomic<long> g_windowLaunches(0), g_windowTiles(0), g_mixedLaunches(0);
/* on wherever the binding announced a genome (DeviceWindows::SetGenome: only oracle/_ref/ngmlr_hip_all carries it); CVX_DEVICE_DECODE=0 keeps
 * the reference's decode on the worker's core */
bool const g_deviceDecode = [] { const char * e = getenv("CVX_DEVICE_DECODE"); return !(e && atoi(e) == 0); }();
}

bool DeviceWindows::Enabled() { return g_deviceDecode; }
bool DeviceWindows::HaveGenome() { return g_binRef != 0; }

void DeviceWindows::SetGenome(void const * binRef, unsigned long long nNibbles, unsigned long long const * startTable, int nStarts) {
	g_startTable.assign(startTable, startTable + (nStarts > 0 ? nStarts : 0));
	g_nNibbles = nNibbles;
	g_binRef = binRef;
}

void DeviceWindows::Placeholder(char * buf, unsigned long long position, int length) {
	memset(buf, 'x', (size_t) (length - 1));      /* what the reference's decode starts from (src/SequenceProvider.cpp:501) */
	buf[length - 1] = '\0';
	if (Fiber * const f = FiberApi::Current()) {
		/* slots 1-3 of the read's context (0 is the dispatcher: batching_aligner.cpp) */
		FiberApi::Local(f, 1) = buf;
		FiberApi::Local(f, 2) = reinterpret_cast<void *>((uintptr_t) position);
		FiberApi::Local(f, 3) = reinterpret_cast<void *>((uintptr_t) length);
	} else {
		tl_window.buf = buf; tl_window.position = position; tl_window.length = length;
	}
}

bool DeviceWindows::Lookup(char const * buf, unsigned long long & position, int & length) {
	if (buf == 0) return false;
	if (Fiber * const f = FiberApi::Current()) {
		if (FiberApi::Local(f, 1) != buf) return false;
		position = (unsigned long long) reinterpret_cast<uintptr_t>(FiberApi::Local(f, 2));
		length = (int) reinterpret_cast<uintptr_t>(FiberApi::Local(f, 3));
		return true;
	}
	if (tl_window.buf != buf) return false;
	position = tl_window.position;
	length = (int) tl_window.length;
	return true;
}

void ConvexAlignHip::WindowStats(long & windowLaunches, long & windowTiles, long & mixedLaunches) {
	windowLaunches = g_windowLaunches.load();
	windowTiles = g_windowTiles.load();
	mixedLaunches = g_mixedLaunches.load();
}

void ConvexAlignHip::CorridorStats(long & prepared, long & closedForm) {
	prepared = g_prepared.load();
	closedForm = g_closedForm.load();
}

ConvexAlignHip::ConvexAlignHip(int const stdOutMode, float const match, float const mismatch,
		float const gapOpen, float const gapExtend, float const gapExtendMin, float const gapDecay,
		int const deviceId, unsigned long const maxMatrixSizeMB) : handle(0), genome(0) {
	(void) stdOutMode;
	cvx_params p;
	p.match = match; p.mismatch = mismatch; p.gap_open = gapOpen;
	p.gap_extend = gapExtend; p.gap_extend_min = gapExtendMin; p.gap_decay = gapDecay;
	maxMatrixMB = maxMatrixSizeMB;
#ifdef CVX_IN_NGMLR_TREE
	if (maxMatrixMB == 0) maxMatrixMB = (unsigned long) Config.getMaxMatrixSizeMB();
#endif
	if (maxMatrixMB == 0) maxMatrixMB = 10000;      /* IConfig's default (src/IConfig.h:47) */
	if (cvx_create(deviceId, &p, (uint64_t) maxMatrixMB, &handle) != CVX_OK) {
		fprintf(stderr, "ConvexAlignHip: %s\n", cvx_last_error());
		throw "ConvexAlignHip: no usable MI355X / unsupported scoring";
	}
}

ConvexAlignHip::~ConvexAlignHip() {
	if (genome) cvx_genome_free(handle, genome);
	genome = 0;
	cvx_destroy(handle);
	handle = 0;
}

/* the reference's convex aligners answer 0 to both (src/ConvexAlignFast.cpp:434-439) */
int ConvexAlignHip::GetScoreBatchSize() const { return 0; }
int ConvexAlignHip::GetAlignBatchSize() const { return 0; }

int ConvexAlignHip::BatchScore(int const, int const, char const * const * const,
		char const * const * const, float * const, void *) {
	throw "Not implemented";
}

int ConvexAlignHip::BatchAlign(int const, int const, char const * const * const,
		char const * const * const, Align * const, void *) {
	throw "Not implemented";   /* no corridor in this signature; use AlignTiles() */
}

int ConvexAlignHip::SingleAlign(int const, int const, char const * const, char const * const,
		Align &, void *) {
	fprintf(stderr, "SingleAlign not implemented");
	throw "Not implemented";
}

int ConvexAlignHip::SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
		char const * const refSeq, char const * const qrySeq, Align & result,
		int const externalQStart, int const externalQEnd, void * extData) {
	(void) mode; (void) extData;
	Tile t;
	t.corridor = corridor; t.corridorHeight = corridorHeight;
	t.refSeq = refSeq; t.qrySeq = qrySeq; t.result = &result;
	t.externalQStart = externalQStart; t.externalQEnd = externalQEnd; t.ret = -1; t.failed = false;
	AlignTiles(&t, 1);
	if (t.failed) throw 1;
	return t.ret;
}

void ConvexAlignHip::Prepare(Tile & t) {
	Align & a = *t.result;
	a.svType = 0;               /* the reference reads it as a debug id, then clears it */
	a.Score = -1.0f;
	t.ret = -1;
	t.failed = false;
	t.window = false;
	t.refPosition = 0;
	int windowLength = 0;
	if (g_deviceDecode && g_binRef != 0 && DeviceWindows::Lookup(t.refSeq, t.refPosition, windowLength)) {
		t.window = true;                        /* the binding's placeholder: the string DecodeRefSequenceExact leaves has windowLength - 1 characters */
		t.refLen = windowLength - 1;
	} else {
		t.refLen = (int) strlen(t.refSeq);
	}
	t.qryLen = (int) strlen(t.qrySeq);
	if (t.corridorHeight != t.qryLen) {
		/* every reference caller passes corridorHeight == strlen(qry); anything else
		 * indexes the corridor out of bounds in the reference itself */
		fprintf(stderr, "ConvexAlignHip: corridorHeight %d != read length %d\n", t.corridorHeight, t.qryLen);
		throw 1;
	}
	/* caller-visible side effect of AlignmentMatrixFast::prepare (src/AlignmentMatrixFast.cpp:39-44) */
	unsigned long acc = 0;
	for (int y = 0; y < t.corridorHeight; ++y) {
		t.corridor[y].offsetInMatrix = acc;
		acc += (unsigned long) (long) t.corridor[y].length;
	}
	/* The rows came out of one of ngmlr's corridor builders (src/AlignmentBuffer.cpp:68-197): recover its closed form -- in
	 * the caller's thread, every row verified against the expression the device evaluates -- so that the launch carries
	 * 32 bytes for this tile instead of 8 per read row through the pack threads, PCIe and the rows arena */
	t.corridorKind = CVX_CORRIDOR_ROWS;
	t.corridorK = t.corridorD = t.corridorRight = 0.0f;
	t.corridorOffset = t.corridorWidth = 0;
	g_prepared.fetch_add(1, std::memory_order_relaxed);
	if (g_fitCorridors && t.corridorHeight > 0) {
		cvx_tile form;
		if (cvx_corridor_fit(&t.corridor[0].offset, &t.corridor[0].length, (int32_t) sizeof(CorridorLine), t.corridorHeight, t.refLen, t.qryLen, &form) == CVX_OK
				&& form.corridor_kind != CVX_CORRIDOR_ROWS) {
			t.corridorKind = form.corridor_kind;
			t.corridorK = form.corridor_k; t.corridorD = form.corridor_d; t.corridorRight = form.corridor_right;
			t.corridorOffset = form.corridor_offset; t.corridorWidth = form.corridor_width;
			g_closedForm.fetch_add(1, std::memory_order_relaxed);
		}
	}
}

cvx_job ConvexAlignHip::Submit(Tile const * tiles, int n) {
	packed.resize((size_t) (n > 0 ? n : 1));
	for (int i = 0; i < n; ++i) {
		Tile const & t = tiles[i];
		cvx_tile & c = packed[(size_t) i];
		c.ref = t.refSeq;
		c.qry = t.qrySeq;
		c.ref_len = t.refLen;
		c.qry_len = t.qryLen;
		c.row_offset = &t.corridor[0].offset;
		c.row_length = &t.corridor[0].length;
		c.row_stride_bytes = (int32_t) sizeof(CorridorLine);
		/* ngmlr hands over the rows its corridor builders produced (src/AlignmentBuffer.cpp:68-197); Prepare() has recovered
		 * the builder's closed form where there is one (the arrays are then ignored by the library) */
		c.corridor_kind = t.corridorKind;
		c.corridor_k = t.corridorK; c.corridor_d = t.corridorD; c.corridor_right = t.corridorRight;
		c.corridor_offset = t.corridorOffset; c.corridor_width = t.corridorWidth;
		c.reserved = 0;
	}
	int nWindows = 0;
	for (int i = 0; i < n; ++i) nWindows += tiles[i].window ? 1 : 0;
	cvx_job job = 0;
	int rc;
	if (nWindows > 0 && genome == 0) {
		if (cvx_genome_upload(handle, (uint8_t const *) g_binRef, g_nNibbles, g_startTable.data(), (int32_t) g_startTable.size(), &genome) != CVX_OK) {
			fprintf(stderr, "ConvexAlignHip: %s\n", cvx_last_error());
			throw 1;
		}
	}
	if (nWindows == n && n > 0) {
		/* every reference of the launch is a window of the resident genome: (position, length) per tile, decoded on the device
		 * straight into the launch's sequence arena (the reference: src/AlignmentBuffer.cpp:199-223 on the worker's core) */
		positions.resize((size_t) n);
		for (int i = 0; i < n; ++i) { positions[(size_t) i] = tiles[i].refPosition; packed[(size_t) i].ref = 0; }
		rc = cvx_submit_windows(handle, genome, n, packed.data(), (uint64_t const *) positions.data(), &job);
		g_windowLaunches.fetch_add(1, std::memory_order_relaxed);
		g_windowTiles.fetch_add(n, std::memory_order_relaxed);
	} else {
		if (nWindows > 0) materialiseWindows(tiles, n);      /* a launch that mixes both forms (no ngmlr path builds one) */
		rc = cvx_submit(handle, n, packed.data(), &job);
	}
	if (rc != CVX_OK) {
		fprintf(stderr, "ConvexAlignHip: %s\n", cvx_last_error());
		throw 1;
	}
	return job;
}

/* the windows among the tiles decoded into their callers' placeholder buffers (which extractReferenceSequenceForAlignment
 * allocated writable, window length + 100 bytes), so that the launch can travel as characters */
void ConvexAlignHip::materialiseWindows(Tile const * tiles, int n) {
	std::vector<uint64_t> pos, off;
	std::vector<int32_t> len;
	std::vector<int> who;
	uint64_t total = 0;
	for (int i = 0; i < n; ++i) if (tiles[i].window) {
		who.push_back(i);
		pos.push_back(tiles[i].refPosition);
		len.push_back(tiles[i].refLen + 1);
		off.push_back(total);
		total += (uint64_t) tiles[i].refLen + 1;
	}
	std::vector<char> out((size_t) total + 8);
	if (cvx_genome_decode(handle, genome, (int32_t) who.size(), pos.data(), len.data(), off.data(), out.data()) != CVX_OK) {
		fprintf(stderr, "ConvexAlignHip: %s\n", cvx_last_error());
		throw 1;
	}
	for (size_t k = 0; k < who.size(); ++k) memcpy(const_cast<char *>(tiles[who[k]].refSeq), out.data() + off[k], (size_t) len[k]);
	g_mixedLaunches.fetch_add(1, std::memory_order_relaxed);
}

bool ConvexAlignHip::Poll(cvx_job job) {
	int32_t done = 0;
	if (cvx_job_poll(handle, job, &done) != CVX_OK) return true;      /* let Wait report it */
	return done != 0;
}

void ConvexAlignHip::Wait(cvx_job job, cvx_result const ** results, uint32_t const ** ops) {
	uint64_t nOps = 0;
	if (cvx_wait(handle, job, results, ops, &nOps) != CVX_OK) {
		fprintf(stderr, "ConvexAlignHip: %s\n", cvx_last_error());
		throw 1;
	}
}

bool ConvexAlignHip::WindowRefs(cvx_job job, Tile * const * tiles, int n) {
	int nWindows = 0;
	for (int i = 0; i < n; ++i) nWindows += tiles[i]->window ? 1 : 0;
	if (nWindows == 0 || n <= 0) return false;
	if (nWindows < n) return false;      /* a mixed launch travelled as characters: its windows were materialised in the callers' buffers (Submit) */
	std::vector<char const *> refs((size_t) n);
	if (cvx_job_window_refs(handle, job, refs.data()) != CVX_OK) {
		/* the placeholders must never reach the text stage: fail the launch like any other error of it */
		fprintf(stderr, "ConvexAlignHip: %s\n", cvx_last_error());
		throw 1;
	}
	for (int i = 0; i < n; ++i) tiles[i]->refSeq = refs[(size_t) i];
	return true;
}

void ConvexAlignHip::Trace(cvx_job job, int nTiles, double serviceMs, double waitedMs) const {
	cvx_timing t;
	if (cvx_job_timing(job, &t) != CVX_OK) return;
	char cls[256];
	int at = 0;
	cls[0] = '\0';
	for (int i = 0; i < t.n_fill_launches && at < 200; ++i) {
		cvx_launch_info li;
		if (cvx_job_launch_info(job, i, &li) != CVX_OK) break;
		at += snprintf(cls + at, sizeof(cls) - (size_t) at, " M%d%s x%d %.2f ms", li.slots_per_lane, li.kind == CVX_LAUNCH_CHAINED ? "c" : (li.kind == CVX_LAUNCH_GANG ? "g" : (li.kind == CVX_LAUNCH_CATCH_ALL ? "x" : "")), li.n_tiles, li.ms);
	}
	fprintf(stderr, "cvx launch: %d tiles, %.1f M cells, oldest request waited %.1f ms, in flight %.1f ms: plan %.2f fill %.2f walk %.2f device total %.2f ms;%s%s\n",
			nTiles, (double) t.cells * 1e-6, waitedMs, serviceMs, t.plan_ms, t.fill_ms, t.backtrack_ms, t.total_ms, cls, t.n_tiles_redone ? " (redo)" : "");
}

void ConvexAlignHip::Release(cvx_job job) {
	cvx_job_release(handle, job);
}

void ConvexAlignHip::AlignTiles(Tile * tiles, int n) {
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) Prepare(tiles[i]);
	cvx_job job = Submit(tiles, n);
	cvx_result const * res = 0;
	uint32_t const * ops = 0;
	try {
		Wait(job, &res, &ops);
	} catch (...) {
		Release(job);
		throw;
	}
	/* windows decoded on the device: the host text stage reads the characters that came back with the results */
	std::vector<char const *> callers((size_t) n);
	std::vector<Tile *> ptrs((size_t) n);
	for (int i = 0; i < n; ++i) { callers[(size_t) i] = tiles[i].refSeq; ptrs[(size_t) i] = &tiles[i]; }
	try {
		(void) WindowRefs(job, ptrs.data(), n);
	} catch (...) {
		Release(job);
		throw;
	}
	for (int i = 0; i < n; ++i) {
		try {
			Finish(tiles[i], res[i], ops);
		} catch (...) {
			/* this tile's own hard error: the caller drops this alignment and no other (src/AlignmentBuffer.cpp:454-463) */
			tiles[i].failed = true;
			tiles[i].ret = -1;
			tiles[i].result->Score = -1.0f;
		}
	}
	for (int i = 0; i < n; ++i) tiles[i].refSeq = callers[(size_t) i];
	Release(job);
}

/* convertCigar + flags into the caller's Align (src/ConvexAlignFast.cpp:488-539) */
void ConvexAlignHip::Finish(Tile & t, cvx_result const & r, uint32_t const * ops) const {
	Align & a = *t.result;
	int const refLen = t.refLen, qryLen = t.qryLen;
	if (noAlignment(t, r)) return;
	cvx_alignment_text txt;
	for (;;) {
		int rc = cvx_format_alignment(&r, ops, t.refSeq, refLen, qryLen, t.externalQStart,
				t.externalQEnd, a.pBuffer1, a.maxBufferLength, a.pBuffer2, a.maxMdBufferLength,
				(int32_t *) a.nmPerPosition, a.nmPerPostionLength, &txt);
		if (rc != CVX_OK) throw 1;
		bool again = false;
		if (txt.md_len >= a.maxMdBufferLength) {       /* checkMdBufferLength: grow with new[] */
			int cap = a.maxMdBufferLength > 0 ? a.maxMdBufferLength : 64;
			while (cap <= txt.md_len) cap *= 2;
			delete[] a.pBuffer2;
			a.pBuffer2 = new char[cap];
			a.maxMdBufferLength = cap;
			again = true;
		}
		if (txt.nm_count > a.nmPerPostionLength) {      /* addPosition: grow with new[] */
			int cap = a.nmPerPostionLength > 0 ? a.nmPerPostionLength : 64;
			while (cap < txt.nm_count) cap *= 2;
			delete[] a.nmPerPosition;
			a.nmPerPosition = new PositionNM[cap];
			a.nmPerPostionLength = cap;
			again = true;
		}
		if (!again) break;
	}
	if (txt.cigar_len > a.maxBufferLength) {
		fprintf(stderr, "CIGAR/MD buffer not long enough (%d %d > %d %d). Please report this!\n",
				txt.cigar_len, txt.md_len, a.maxBufferLength, a.maxMdBufferLength);
		throw 1;
	}
	fillAlign(t, txt);
}

void ConvexAlignHip::fillAlign(Tile & t, cvx_alignment_text const & txt) const {
	Align & a = *t.result;
	a.QStart = txt.qstart;
	a.QEnd = txt.qend;
	a.firstPosition.refPosition = txt.first_ref;
	a.firstPosition.readPosition = txt.first_read;
	a.lastPosition.refPosition = txt.last_ref;
	a.lastPosition.readPosition = txt.last_read;
	a.Identity = txt.identity;
	a.NM = txt.nm;
	a.alignmentLength = txt.alignment_length;
	a.cigarOpCount = txt.cigar_op_count;
	a.PositionOffset = txt.position_offset;
	a.Score = txt.score;
	a.svType = txt.sv_type;
	t.ret = txt.ret;
}

/* true: the tile has no alignment and the Align says so already (the cases Finish handles before it formats anything) */
bool ConvexAlignHip::noAlignment(Tile & t, cvx_result const & r) const {
	Align & a = *t.result;
	if (r.status == CVX_TILE_UNSUPPORTED) {
		fprintf(stderr, "ConvexAlignHip: corridor shape not covered by any device kernel\n");
		throw 1;
	}
	if (r.status == CVX_TILE_OK) return false;
	if (r.status == CVX_TILE_TOO_LARGE) {
		/* the reference's message (src/AlignmentMatrixFast.cpp:56), same float arithmetic for the size */
		fprintf(stderr, "Warning: Couldn't allocate alignment matrix. Required memory (%llu) > max matrix size (%lu)\n\n",
				(long long) ((float) r.cells / 1000.0f / 1000.0f), maxMatrixMB);
	} else if (a.pBuffer2 != 0) {
		a.pBuffer2[0] = '\0';
	}
	a.Score = -1.0f;
	t.ret = -1;
	return true;
}

void ConvexAlignHip::Text(cvx_job job, Tile const * const * tiles, int n, JobText & jt) {
	size_t const n1 = (size_t) (n > 0 ? n : 0);
	jt.out.resize(n1 + 1);
	jt.textOff.resize(n1 + 1);
	jt.nmOff.assign(n1 + 1, 0);
	jt.text = "";
	if (n <= 0) return;
	std::vector<int32_t> ext(2 * n1);
	for (size_t i = 0; i < n1; ++i) { ext[i] = tiles[i]->externalQStart; ext[n1 + i] = tiles[i]->externalQEnd; }
	uint64_t bytes = 0;
	jt.nm = 0;
	int const rc = cvx_job_text_all(handle, job, ext.data(), ext.data() + n1, jt.out.data(), jt.textOff.data(), &jt.text, &bytes, jt.nmOff.data(), &jt.nm);
	if (rc != CVX_OK) {
		fprintf(stderr, "ConvexAlignHip: %s\n", cvx_last_error());
		throw 1;
	}
}

/* what Finish leaves in the caller's Align, copied out of the job's device-made text (same growth rules for pBuffer2 and
 * nmPerPosition, same hard error for a CIGAR beyond the caller's buffer) */
void ConvexAlignHip::FinishText(Tile & t, cvx_result const & r, JobText const & jt, int index) const {
	if (noAlignment(t, r)) return;
	Align & a = *t.result;
	cvx_alignment_text const & txt = jt.out[(size_t) index];
	if (txt.md_len >= a.maxMdBufferLength) {       /* checkMdBufferLength: grow with new[] */
		int cap = a.maxMdBufferLength > 0 ? a.maxMdBufferLength : 64;
		while (cap <= txt.md_len) cap *= 2;
		delete[] a.pBuffer2;
		a.pBuffer2 = new char[cap];
		a.maxMdBufferLength = cap;
	}
	if (txt.nm_count > a.nmPerPostionLength) {      /* addPosition: grow with new[] */
		int cap = a.nmPerPostionLength > 0 ? a.nmPerPostionLength : 64;
		while (cap < txt.nm_count) cap *= 2;
		delete[] a.nmPerPosition;
		a.nmPerPosition = new PositionNM[cap];
		a.nmPerPostionLength = cap;
	}
	if (txt.cigar_len > a.maxBufferLength) {
		fprintf(stderr, "CIGAR/MD buffer not long enough (%d %d > %d %d). Please report this!\n",
				txt.cigar_len, txt.md_len, a.maxBufferLength, a.maxMdBufferLength);
		throw 1;
	}
	char const * src = jt.text + jt.textOff[(size_t) index];
	memcpy(a.pBuffer1, src, (size_t) (txt.cigar_len + 1 <= a.maxBufferLength ? txt.cigar_len + 1 : a.maxBufferLength));
	memcpy(a.pBuffer2, src + txt.cigar_len + 1, (size_t) txt.md_len + 1);
	uint64_t const e0 = jt.nmOff[(size_t) index], e1 = jt.nmOff[(size_t) index + 1];
	if ((uint64_t) txt.nm_count != e1 - e0) {
		fprintf(stderr, "ConvexAlignHip: nmPerPosition count of the device (%llu) differs from the text stage's (%d)\n", (unsigned long long) (e1 - e0), txt.nm_count);
		throw 1;
	}
	if (e1 > e0) memcpy((void *) a.nmPerPosition, jt.nm + 3 * e0, (size_t) (e1 - e0) * 3 * sizeof(int32_t));
	fillAlign(t, txt);
}

}  // namespace Convex
