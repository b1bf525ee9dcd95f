/*
 * convex_align_hip.h -- the drop-in: an IAlignment implementation that runs ngmlr's
 * convex-gap alignment on an MI355X through the C ABI of include/cvx_align.h.
 *
 * It takes the place of Convex::ConvexAlignFast at the one construction site
 * (reference src/AlignmentBuffer.h:355-363) and keeps its contract:
 *   - same constructor arguments (stdOutMode + six scoring floats), plus a device id;
 *   - SingleAlign(mode, CorridorLine*, ...) has the same argument meaning, side effects
 *     (offsetInMatrix written into the caller's lines, pBuffer2 / nmPerPosition possibly
 *     re-allocated with new[], align.svType consumed as read id then reset) and the same
 *     error convention: -1 / Score -1.0f for "no valid alignment", `throw 1` for hard
 *     errors (reference src/ConvexAlignFast.cpp:452-559);
 *   - BatchScore / corridor-less SingleAlign throw like the reference's do.
 * Extra: AlignTiles() -- many corridor alignments in one launch (the shape the
 * reference's BatchAlign slot lacks a corridor argument for) -- and its split form
 * Submit() / Wait() / Finish() / Release(), which keeps several launches in flight
 * (batching_aligner.h drives it from one dispatcher thread per device).
 */
#ifndef CONVEX_ALIGN_HIP_H
#define CONVEX_ALIGN_HIP_H

#include <stdint.h>
#include <vector>

#include "ngmlr_abi.h"
#include "cvx_align.h"

namespace Convex {

class ConvexAlignHip: public IAlignment {
public:
	ConvexAlignHip(int const stdOutMode, float const match, float const mismatch, float const gapOpen,
			float const gapExtend, float const gapExtendMin, float const gapDecay, int const deviceId = 0,
			unsigned long const maxMatrixSizeMB = 0 /* Config.getMaxMatrixSizeMB(); 0 = the reference's default 10000 */);
	virtual ~ConvexAlignHip();

	virtual int GetScoreBatchSize() const;
	virtual int GetAlignBatchSize() const;

	virtual int BatchScore(int const mode, int const batchSize, char const * const * const refSeqList,
			char const * const * const qrySeqList, float * const results, void * extData);
	virtual int BatchAlign(int const mode, int const batchSize, char const * const * const refSeqList,
			char const * const * const qrySeqList, Align * const results, void * extData);
	virtual int SingleAlign(int const mode, int const corridor, char const * const refSeq,
			char const * const qrySeq, Align & result, void * extData);
	virtual int SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
			char const * const refSeq, char const * const qrySeq, Align & result,
			int const externalQStart, int const externalQEnd, void * extData);

	/* One request of AlignTiles(): exactly the arguments of the corridor SingleAlign. */
	struct Tile {
		CorridorLine * corridor;
		int corridorHeight;
		char const * refSeq;
		char const * qrySeq;
		Align * result;
		int externalQStart;
		int externalQEnd;
		int ret;               /* out: what SingleAlign would have returned */
		bool failed;           /* out (AlignTiles): this tile hit a hard error -- the reference would have thrown for it alone */
		int refLen, qryLen;    /* filled by Prepare() */
		/* filled by Prepare(): the closed form behind `corridor` when cvx_corridor_fit finds one (every row verified), else
		 * CVX_CORRIDOR_ROWS -- Submit then sends 32 bytes instead of the tile's row arrays */
		int32_t corridorKind;
		float corridorK, corridorD, corridorRight;
		int32_t corridorOffset, corridorWidth;
		/* filled by Prepare(): refSeq is a placeholder for the window of the resident genome at refPosition (DeviceWindows
		 * below) -- the launch carries (position, length) and the device decodes; Finish needs WindowRefs() first */
		bool window;
		unsigned long long refPosition;
	};
	/* n independent corridor alignments in one device launch.  A hard error that belongs to one tile (a corridor no
	 * kernel covers, a CIGAR that does not fit the caller's buffer) marks that tile `failed` and leaves the others
	 * alone -- the reference's caller drops exactly one alignment in that case (src/AlignmentBuffer.cpp:454-463);
	 * a failure of the launch itself throws. */
	void AlignTiles(Tile * tiles, int n);

	/* The same in stages, for a driver that keeps launches in flight.  Prepare / Finish may run on any thread (they
	 * touch only the tile and the caller's Align); Submit / Wait / Release belong to the ONE thread that owns this
	 * aligner's device handle.
	 *   Prepare  strlen + height check + the caller-visible offsetInMatrix side effect (throws like SingleAlign) + the
	 *            corridor's closed form (cvx_corridor_fit; CVX_CORRIDOR_FIT=0 sends the rows as round 5 did)
	 *   Submit   queues a launch for n prepared tiles, returns its job (throws on a hard error of the launch)
	 *   Poll     non-blocking "is it done"; Wait  blocks until the job is done: result records and run-length ops, valid until Release
	 *   Finish   convertCigar + flags of ONE tile into its Align (throws 1 for that tile's hard errors)
	 *   Release  gives the job's buffers back */
	static void Prepare(Tile & t);
	cvx_job Submit(Tile const * tiles, int n);
	bool Poll(cvx_job job);       /* true: Wait would not block (also keeps queued launches moving) */
	void Wait(cvx_job job, cvx_result const ** results, uint32_t const ** ops);
	void Finish(Tile & t, cvx_result const & r, uint32_t const * ops) const;
	void Release(cvx_job job);
	/* one line on stderr about a finished job: tiles, device stages, fill classes (CVX_LAUNCH_TRACE=1 in the dispatcher) */
	void Trace(cvx_job job, int nTiles, double serviceMs, double waitedMs) const;

	/* The text stage of a whole finished job on the device instead of one Finish per tile on the workers' cores
	 * (cvx_job_text + cvx_job_nm_profile: CIGAR, MD, the scalar fields and nmPerPosition from the ops and sequences
	 * still resident in HBM).  Text belongs to the device thread, after Wait and before Release; FinishText may run
	 * on any thread and only copies: the same Align, byte for byte, as Finish (tests/test_gpu_e2e.py, CVX_DEVICE_TEXT=1). */
	struct JobText {
		std::vector<cvx_alignment_text> out;
		std::vector<uint64_t> textOff, nmOff;
		int32_t const * nm;                /* (refPosition, readPosition, nm) triples of all tiles, in the job's page-locked memory */
		char const * text;                 /* the job's page-locked text buffer */
	};
	void Text(cvx_job job, Tile const * const * tiles, int n, JobText & jt);
	/* after Wait, for a launch that travelled through cvx_submit_windows: tiles[i]->refSeq is redirected to the characters
	 * the device decoded (cvx_job_window_refs: the job's page-locked memory, valid until Release), so that Finish -- the host
	 * text stage, which reads the reference base of every mismatch and deletion -- works on them.  false: not a window launch; throws 1 when a window launch cannot hand its characters back */
	bool WindowRefs(cvx_job job, Tile * const * tiles, int n);
	/* launches that travelled as windows of the resident genome (cvx_submit_windows), and launches that mixed windows with
	 * decoded references (their windows were materialised with cvx_genome_decode first) */
	static void WindowStats(long & windowLaunches, long & windowTiles, long & mixedLaunches);
	/* tiles Prepare() has seen in this process, and how many of them travelled as a closed form */
	static void CorridorStats(long & prepared, long & closedForm);
	void FinishText(Tile & t, cvx_result const & r, JobText const & jt, int index) const;

private:
	void fillAlign(Tile & t, cvx_alignment_text const & txt) const;
	bool noAlignment(Tile & t, cvx_result const & r) const;
	cvx_handle handle;
	unsigned long maxMatrixMB;
	std::vector<cvx_tile> packed;
	cvx_genome genome;                     /* DeviceWindows' genome on this aligner's device (uploaded by the first launch that carries a window) */
	std::vector<unsigned long long> positions;
	std::vector<char const *> refPtrs;
	void materialiseWindows(Tile const * tiles, int n);
};

/*
 * Reference windows decoded on the device inside ngmlr's worker flow (decode half).
 *
 * ngmlr expands the reference window of every alignment from its 4-bit genome on the worker's core
 * (extractReferenceSequenceForAlignment -> DecodeRefSequenceExact, reference src/AlignmentBuffer.cpp:199-223,
 * src/SequenceProvider.cpp:493-565) and hands SingleAlign the characters.  Between that call and SingleAlign the
 * caller only measures the string (strlen in the corridor builders, :112, :135).  Unless CVX_DEVICE_DECODE=0, the
 * binding (window_decode_binding.inc) allocates the same buffer, fills it with a placeholder of the same length and
 * notes (buffer, position, length) for the calling context; Prepare() recognises the buffer, the launch travels
 * through cvx_submit_windows, and the text stage -- on the host (default) or on the device (CVX_DEVICE_TEXT=1) -- reads
 * what the device decoded: the windows come back with the launch's results (cvx_job_window_refs, one byte per base) resp.
 * are read where the fill read them.  No core expands a window.  The note travels with the read:
 * fiber-local under the pool's user-level contexts, thread-local on a plain worker thread.
 */
struct DeviceWindows {
	static bool Enabled();                 /* unless CVX_DEVICE_DECODE=0 */
	/* ngmlr's encoded genome as _SequenceProvider::Init leaves it (binRef, binRefIndex nibbles, refStartPos with its upper
	 * bound): the pointers must stay valid; each aligner uploads it to its own device at its first window launch */
	static void SetGenome(void const * binRef, unsigned long long nNibbles, unsigned long long const * startTable, int nStarts);
	static bool HaveGenome();
	/* buf[0 .. length) stands for DecodeRefSequenceExact(buf, position, length, 0): length - 1 characters and a NUL */
	static void Placeholder(char * buf, unsigned long long position, int length);
	static bool Lookup(char const * buf, unsigned long long & position, int & length);
};

}  // namespace Convex

#endif
