/*
 * cvx_align.h -- C ABI of the MI355X-native convex-gap banded Smith-Waterman
 * aligner (libcvxalign.so).  This is the drop-in boundary for ngmlr's hot path:
 * everything ngmlr's IAlignment plugin surface needs for
 *
 *     Convex::ConvexAlignFast::SingleAlign(mode, CorridorLine*, corridorHeight,
 *                                          refSeq, qrySeq, Align&, externalQStart,
 *                                          externalQEnd, extData)
 *     (reference src/IAlignment.h:227-232, src/ConvexAlignFast.cpp:452-559)
 *
 * expressed as plain C: opaque handle, plain pointers and sizes, int return codes,
 * no C++ types and no exceptions across the boundary.  A "tile" below is exactly
 * one such SingleAlign call; the library takes thousands of them per launch.
 *
 * Entry point                 replaces (reference file:line)
 * --------------------------- -----------------------------------------------------
 * cvx_create / cvx_destroy    ConvexAlignFast ctor/dtor      src/ConvexAlignFast.cpp:29-64
 *                             (pfCreateAlignment(gpu_id)     src/IAlignment.h:249-250)
 * cvx_align_batch             N x SingleAlign steps 1-4      src/ConvexAlignFast.cpp:463-487
 *                             = AlignmentMatrixFast::prepare src/AlignmentMatrixFast.cpp:30-60
 *                             + fwdFillMatrixSSESimple       src/ConvexAlignFast.cpp:914-1287
 *                             + revBacktrack / validPath     src/ConvexAlignFast.cpp:335-432
 *                             (the BatchAlign slot the reference leaves unimplemented,
 *                              src/ConvexAlignFast.cpp:441-450)
 * cvx_batch_* (staged form)   the same, with inputs resident in HBM between calls
 * cvx_submit / cvx_wait       the same as a stream of batches: several in flight per handle,
 *                             upload / kernels / download of neighbouring batches overlapped
 *                             (what a batching host driver needs, SURVEY.md 8 f1;
 *                              reference src/AlignmentBuffer.cpp:3361-3406)
 * cvx_tile.corridor_kind      the corridor builders' closed forms, rows generated on the device
 *                             src/AlignmentBuffer.cpp:68-197 (a8: the builders stay in ngmlr, their output need not travel)
 * cvx_format_alignment        convertCigar + N-clip flags    src/ConvexAlignFast.cpp:112-333,493-528
 * cvx_job_text                the same for a whole finished job, on the device (next-row f3)
 * cvx_job_nm_profile          addPosition / nmPerPosition    src/ConvexAlignFast.cpp:76-98,186-269, on the device
 * cvx_sam_record_text / cvx_sam_batch  SAMWriter::DoWriteReadGeneric  src/SAMWriter.cpp:87-224 (f3: SAM record assembly)
 * cvx_sam_unmapped_text       SAMWriter::DoWriteUnmappedReadGeneric  src/SAMWriter.cpp:308-357
 * cvx_score_batch             StrippedSW::BatchScore/SingleScore  src/StrippedSW.cpp:118-203 (next-row f2)
 * cvx_genome_* / cvx_submit_windows  SequenceProvider's 4-bit genome + DecodeRefSequenceExact
 *                             src/SequenceProvider.cpp:333-386,475-565 (next-row f4, decode half)
 * cvx_job_window_refs         the windows cvx_submit_windows decoded, back on the host for a text stage there
 *                             (replaces extractReferenceSequenceForAlignment's decode, src/AlignmentBuffer.cpp:199-223)
 * cvx_job_text_all / cvx_job_nm_profile_resident  cvx_job_text + cvx_job_nm_profile per finished launch, profile in page-locked memory
 * cvx_index_upload / cvx_search_batch  CS::RunRead's k-mer vote over the CompactPrefixTable
 *                             src/CS.cpp:57-149,219-268,324-398, src/CSstatic.cpp:23-73, src/PrefixTable.cpp:476-532 (f4, search half)
 *
 * The binding a maintainer adds on the ngmlr side is in INTEGRATION.md
 * (ngmlr_amd/csrc/convex_align_hip.{h,cpp}: an IAlignment subclass over this ABI).
 *
 * There is no CPU fallback: every compute entry point fails with CVX_ERR_NO_DEVICE
 * when no gfx950 device is usable.
 */
#ifndef CVX_ALIGN_H
#define CVX_ALIGN_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVX_ABI_VERSION 9

/* return codes */
enum {
	CVX_OK = 0,
	CVX_ERR_NO_DEVICE = -1,      /* no HIP device / wrong device id */
	CVX_ERR_PARAMS = -2,         /* scoring parameters that are not finite (any finite set is accepted: scoring outside the regime in
	                              * which the reference's SSE path equals its scalar recurrence runs on the catch-all kernel's SSE variant) */
	CVX_ERR_ARG = -3,            /* NULL pointer, negative size, height != strlen(qry) ... */
	CVX_ERR_OOM = -4,            /* device or host allocation failed */
	CVX_ERR_HIP = -5,            /* a HIP call failed; see cvx_last_error() */
	CVX_ERR_CAPACITY = -6        /* caller's ops arena too small; *ops_used holds the need */
};

/* per-tile status in cvx_result.status */
enum {
	CVX_TILE_OK = 0,             /* valid alignment (reference: return value >= 0) */
	CVX_TILE_INVALID_ROW0 = 1,   /* best cell in read row 0          (:338) */
	CVX_TILE_INVALID_EDGE = 2,   /* path touched outer 10% of corridor (validPath) */
	CVX_TILE_INVALID_LENGTH = 3, /* consumed read length != H         (:424-428) */
	CVX_TILE_TOO_LARGE = 4,      /* > maxMatrixSizeMB (AlignmentMatrixFast.cpp:45,55-57) */
	CVX_TILE_EMPTY = 5,          /* H == 0 or no cell inside [0,W) */
	CVX_TILE_UNSUPPORTED = -1    /* corridor shape no device kernel covers (loud, never silent) */
};

/* operation codes inside the ops arena: (length << 4) | op, forward order.
 * Same numbering as the reference's directionMatrix (src/AlignmentMatrixFast.h:15-24). */
enum { CVX_OP_I = 1, CVX_OP_D = 2, CVX_OP_EQ = 7, CVX_OP_X = 8 };

/* Scoring, ctor order of ConvexAlignFast (src/ConvexAlignFast.cpp:29-43):
 * match > 0, the rest < 0 except gap_decay >= 0.  Defaults: 2,-5,-5,-5,-1,0.15
 * (src/IConfig.h:50-55). */
typedef struct {
	float match;
	float mismatch;
	float gap_open;      /* gap_open_read == gap_open_ref in the reference */
	float gap_extend;    /* --gap-extend-max */
	float gap_extend_min;
	float gap_decay;
} cvx_params;

/* One SingleAlign call.  ref/qry need not be NUL-terminated (lengths are explicit;
 * the reference takes strlen).  Row y of the corridor covers reference columns
 * [offset[y], offset[y]+length[y]) clipped to [0, ref_len).
 *
 * corridor_kind says where the rows come from:
 *   CVX_CORRIDOR_ROWS    the caller's arrays: row_offset / row_length are read with a byte stride so
 *                        that &CorridorLine[0].offset / &CorridorLine[0].length with stride
 *                        sizeof(CorridorLine)=16 can be passed directly (src/IAlignment.h:29-33).
 *   CVX_CORRIDOR_AFFINE  the closed form of the reference's own corridor builders: every row has
 *                        corridor_width columns and
 *                            offset[y] = (int) (((float) y - corridor_d) / corridor_k - corridor_right)
 *                        evaluated in binary32 exactly as the reference does (float subtract, correctly
 *                        rounded divide, float subtract, truncation), on the device -- no row array is
 *                        read, packed or uploaded:
 *                          getCorridorEndpointsWithAnchors  src/AlignmentBuffer.cpp:178-191
 *                              k = qryLen * 1.0f / refLen, d = 0, right = corridorRight, width = (int)(left + right)
 *                          getCorridorEndpoints             src/AlignmentBuffer.cpp:107-127
 *                              k = qryLen * 1.0f / refLen, d = width / 2.0f, right = 0
 *                          getCorridorLinear                src/AlignmentBuffer.cpp:68-82
 *                              k = 1, d = width / 2 (integer division), right = 0      (exact below 2^24 rows)
 *   CVX_CORRIDOR_CONST   every row is (corridor_offset, corridor_width):
 *                          getCorridorFull                  src/AlignmentBuffer.cpp:84-105
 * row_offset / row_length / row_stride_bytes are ignored for the closed forms. */
enum { CVX_CORRIDOR_ROWS = 0, CVX_CORRIDOR_AFFINE = 1, CVX_CORRIDOR_CONST = 2 };

typedef struct {
	const char *ref;
	const char *qry;
	const int32_t *row_offset;
	const int32_t *row_length;
	int32_t ref_len;
	int32_t qry_len;          /* = corridor height; every caller passes them equal */
	int32_t row_stride_bytes; /* 4 for packed int32 arrays, 16 for CorridorLine[] */
	int32_t corridor_kind;    /* CVX_CORRIDOR_* (0 = the row arrays) */
	float corridor_k;
	float corridor_d;
	float corridor_right;
	int32_t corridor_offset;
	int32_t corridor_width;
	int32_t reserved;
} cvx_tile;

/* ABI 7: the closed form behind a caller's rows (host only, no device call; any thread).  ngmlr's binding receives the
 * CorridorLine[] its builders wrote (src/AlignmentBuffer.cpp:68-197) and cannot know which builder made them; this
 * recovers (corridor_kind, k, d, right, offset, width) from the rows and the two sequence lengths -- k = qry_len * 1.0f /
 * ref_len as at :117 / :141, d = width / 2.0f (endpoints), d = 0 and a fitted `right` (anchors), k = 1 (linear), a constant
 * (full) -- and writes them into *form, whose other fields it leaves alone.  A form is returned only after every one of the
 * n_rows rows has been compared with the expression the device evaluates; rows no form reproduces (or rows of different
 * lengths) leave corridor_kind = CVX_CORRIDOR_ROWS.  row_offset / row_length are read with row_stride_bytes like
 * cvx_tile's.  Returns CVX_OK (whatever the kind) or CVX_ERR_ARG. */
int cvx_corridor_fit(const int32_t *row_offset, const int32_t *row_length, int32_t row_stride_bytes, int32_t n_rows,
		int32_t ref_len, int32_t qry_len, cvx_tile *form);
/* The same for a tile table, in place, on n_threads host threads (0 = all): every tile with corridor_kind == CVX_CORRIDOR_ROWS
 * gets its closed form when it has one; *n_fitted (may be NULL) = how many did. */
int cvx_corridor_fit_batch(int32_t n_tiles, cvx_tile *tiles, int32_t n_threads, int32_t *n_fitted);

/* What the forward fill + backtrack leave behind (FwdResults, src/ConvexAlignFast.h:69-77). */
typedef struct {
	float score;            /* curr_max of the fill (Align::Score when status == 0) */
	int32_t status;         /* CVX_TILE_* */
	int32_t best_ref_index; /* argmax cell, first in (y,x) order */
	int32_t best_read_index;
	int32_t ref_position;   /* Align::PositionOffset */
	int32_t qstart;         /* leading soft clip (without externalQStart) */
	int32_t qend;           /* trailing soft clip (without externalQEnd) */
	int32_t n_ops;
	uint64_t ops_begin;     /* index of the first op in the ops arena */
	uint64_t cells;         /* sum of row_length = bytes of the reference's directionMatrix */
} cvx_result;

typedef struct cvx_context *cvx_handle;
typedef struct cvx_batch_s *cvx_batch;
typedef struct cvx_batch_s *cvx_job;      /* a batch travelling through the streaming form */

/* Timing of the last cvx_batch_run, measured with HIP events on the library's stream. */
typedef struct {
	float plan_ms;      /* corridor analysis kernel + plan readback */
	float fill_ms;      /* all forward-fill launches */
	float backtrack_ms; /* backtrack + ops compaction */
	float total_ms;     /* first launch -> last kernel done */
	uint64_t cells;     /* corridor cells (sum of row_length) in the batch */
	uint64_t active_cells; /* cells inside [0,W): what the fill really computes */
	uint64_t dir_bytes; /* bytes of direction codes written to HBM */
	int32_t n_fill_launches;
	int32_t n_tiles_fast; /* tiles taken whole by one wave */
	int32_t n_tiles_redone; /* tiles whose best cell was not in the exactly tracked tail (second fill pass) */
	int32_t n_tiles_chained; /* wide tiles cut into chained row blocks */
	/* ABI 6: chained row blocks -- ticks of the device's constant clock (s_memtime) their waves ran for, summed over the tasks,
	 * and the part of it spent waiting for boundary records of the block above (chain_poll_ticks / chain_task_ticks = the share
	 * of a chained wave's life in which it holds a wave slot without issuing the recurrence) */
	uint64_t chain_task_ticks, chain_poll_ticks;
} cvx_timing;

enum { CVX_LAUNCH_WHOLE = 0, CVX_LAUNCH_GANG = 1, CVX_LAUNCH_CHAINED = 2, CVX_LAUNCH_CATCH_ALL = 3 };
/* One forward-fill launch of the last cvx_batch_run (HIP-event timed on the stream). */
typedef struct {
	int32_t slots_per_lane;  /* M */
	int32_t waves;           /* 1; for chained tiles the number of row-block tasks of the launch */
	int32_t wrap16;          /* int16 gap-run wrap emulation compiled in */
	int32_t n_tiles;
	float ms;                /* kernel duration */
	int32_t kind;            /* ABI 7 (in what was padding): CVX_LAUNCH_WHOLE one wave per tile, CVX_LAUNCH_GANG `waves` waves on one ring per
	                          * tile, CVX_LAUNCH_CHAINED `waves` row-block tasks in the launch, CVX_LAUNCH_CATCH_ALL the catch-all kernel */
	uint64_t cells;          /* sum of row_length over the launch's tiles */
	uint64_t active_cells;   /* cells inside [0,W) */
	uint64_t alg_bytes;      /* algorithmic bytes: sum over tiles of C + 6H + 2W */
	uint64_t read_bases;     /* sum of qry_len */
} cvx_launch_info;

const char *cvx_last_error(void);
int cvx_abi_version(void);
int cvx_device_count(void);
/* identifies the build of the device kernels (a hash of their source): profiles/ records it with the counters it
 * collects, and bench.py reports counter-derived figures only for the build they were collected on */
const char *cvx_build_id(void);
/* the same for one kernel family's sources alone -- "fill" (cvx_kernels.hip), "search" (cvx_search.hip), each with the shared
 * headers; anything else: cvx_build_id().  Counters collected for a kernel (profiles/r*_pmc.json) stay valid while its own id does. */
const char *cvx_source_id(const char *family);
/* Blocks until everything queued on `device_id` (by any handle of this process) has finished:
 * hipDeviceSynchronize behind the C ABI, for callers that bracket a timed region. */
int cvx_device_synchronize(int device_id);

/* max_matrix_mb: IConfig::maxMatrixSizeMB (src/IConfig.h:47), 0 -> 10000.
 *
 * Process-wide side effects, stated here because they reach beyond the handle:
 *   - the first cvx_create on a device puts that device into hipDeviceScheduleBlockingSync mode (host threads that wait for
 *     it sleep instead of spinning; this also governs the host application's own HIP waits on that device).  CVX_WAIT=spin
 *     leaves the runtime's default; a runtime that refuses the flag is reported on stderr.
 *   - loading the library sets GPU_MAX_HW_QUEUES=16 in the environment unless the variable is already set (the HIP runtime
 *     reads it at its first call): the streams of several handles in one process must not share hardware queues. */
int cvx_create(int device_id, const cvx_params *params, uint64_t max_matrix_mb, cvx_handle *out);
/* ABI 7: the same with flags.  CVX_CREATE_SERVICE: the handle will only make the short calls (cvx_score_batch, cvx_search_batch*,
 * cvx_genome_decode) from a thread that blocks in them -- its stream gets the device's highest priority, so that a 0.3 ms scoring
 * kernel does not queue behind a 10 ms fill of an aligning handle in the same process.  MEASURED AND OFF: inside ngmlr the run got
 * slower with it (profiles/r06_e2e_service_prio.txt); the flag is recorded in the handle, the priority applies only with CVX_SERVICE_PRIO=1. */
enum { CVX_CREATE_SERVICE = 1 };
/* ABI 7: which of the process-wide settings above this process really got (a host that initialised HIP before
 * the library was loaded gets neither, and its mixed launches run their fill classes one after the other).
 *   hw_queues_env             GPU_MAX_HW_QUEUES as the process's environment has it now (0: unset)
 *   hw_queues_set_by_library  1: the library exported it when it was loaded; 0: the user's value was left alone.  (The runtime
 *                             reads the variable at ITS first call: a value exported after that -- HIP initialised before this
 *                             library was loaded -- has no effect; runtime_up_at_load says whether that was the case, and the first
 *                             cvx_create then says so on stderr.)
 *   blocking_sync             1: hipDeviceScheduleBlockingSync applied to the device by the first cvx_create; 0: not in effect
 *                             (blocking_sync_why: 2 the runtime refused -- a context was already active --, 3 CVX_WAIT=spin);
 *                             -1: no handle has been created on the device yet
 *   service_streams           streams the CVX_CREATE_SERVICE handles of the process share per device (0: one each) */
typedef struct {
	int32_t hw_queues_env, hw_queues_set_by_library;
	int32_t blocking_sync, blocking_sync_why;
	int32_t service_streams;
	int32_t runtime_up_at_load;      /* 1: the process already had /dev/kfd open when the library was loaded (a runtime that cannot see the variable any more) */
	int32_t reserved[2];
} cvx_regime;
int cvx_runtime_regime(int device_id, cvx_regime *out);
int cvx_create_ex(int device_id, const cvx_params *params, uint64_t max_matrix_mb, uint32_t flags, cvx_handle *out);
void cvx_destroy(cvx_handle h);

/* Synchronous convenience form: upload, run, download.  ops_arena receives the
 * run-length ops of all valid tiles back to back; ops_used is set even on
 * CVX_ERR_CAPACITY so the caller can retry. */
int cvx_align_batch(cvx_handle h, int32_t n_tiles, const cvx_tile *tiles, cvx_result *results,
		uint32_t *ops_arena, uint64_t ops_capacity, uint64_t *ops_used);

/* Staged form: inputs stay resident in HBM, run may be repeated (bench / pipelining). */
int cvx_batch_upload(cvx_handle h, int32_t n_tiles, const cvx_tile *tiles, cvx_batch *out);
int cvx_batch_run(cvx_handle h, cvx_batch b);            /* enqueue + wait */
int cvx_batch_timing(cvx_batch b, cvx_timing *t);
int cvx_batch_ops_total(cvx_batch b, uint64_t *n_ops);
/* i in [0, timing.n_fill_launches) */
int cvx_batch_launch_info(cvx_batch b, int32_t i, cvx_launch_info *info);
int cvx_batch_download(cvx_handle h, cvx_batch b, cvx_result *results,
		uint32_t *ops_arena, uint64_t ops_capacity, uint64_t *ops_used);
void cvx_batch_free(cvx_handle h, cvx_batch b);

/* Streaming form: host buffers in, results out, several batches in flight on one handle.
 *
 *   cvx_submit  packs the tiles into the job's pinned staging (the caller's buffers are free again
 *               when it returns), queues the upload and the corridor analysis and returns; it also
 *               queues the kernels of earlier jobs whose inputs have arrived.  Nothing waits.
 *   cvx_wait    blocks until the job is done; *results / *ops point into page-locked memory owned by
 *               the job (cvx_result[n_tiles], the dense ops arena) and stay valid until
 *               cvx_job_release.  Jobs complete in submission order.
 *   cvx_job_release  returns the job's arenas to the handle (they are reused by the next submit).
 *
 * Errors belong to their job: if something fails while a job's kernels are queued (say a multi-GB arena does not
 * fit beside the batches in flight), that job's cvx_wait -- and only it -- returns the error, every time it is
 * called; the job handle stays valid until cvx_job_release.  cvx_submit reports only what happens to the batch
 * being submitted.  cvx_destroy frees whatever jobs were never released (their handles die with the context).
 *
 * With two or three jobs in flight (submit k+1, then wait k-1) the upload of batch k+1 and the
 * download of batch k-1 run under the kernels of batch k.  A handle is not re-entrant: one host
 * thread per handle (one handle per device and thread, like the reference's aligner instances). */
int cvx_submit(cvx_handle h, int32_t n_tiles, const cvx_tile *tiles, cvx_job *out);
int cvx_wait(cvx_handle h, cvx_job job, const cvx_result **results, const uint32_t **ops, uint64_t *n_ops);
/* Non-blocking: *done = 1 when cvx_wait on the job would return without waiting for the device (finished or failed).
 * Also hands the device the kernels of any job whose corridor analysis has come back meanwhile -- a driver that polls
 * instead of blocking in cvx_wait (batching_aligner.cpp) keeps the pipeline moving with it. */
int cvx_job_poll(cvx_handle h, cvx_job job, int32_t *done);
int cvx_job_timing(cvx_job job, cvx_timing *t);                        /* after cvx_wait */
int cvx_job_launch_info(cvx_job job, int32_t i, cvx_launch_info *info); /* after cvx_wait */
void cvx_job_release(cvx_handle h, cvx_job job);

/* Page-locked host memory for sequences.  When all reads of a batch lie back to back (tile i+1's qry
 * starts where tile i's ends) inside memory from cvx_host_alloc, and so do all its references, cvx_submit
 * does not touch a single base: the device pulls both blocks straight out of the caller's arena.  Such
 * memory must stay unchanged until cvx_wait has returned for every job that references it (ordinary
 * memory is copied into the job's own staging before cvx_submit returns, as before). */
int cvx_host_alloc(uint64_t bytes, void **out);
void cvx_host_free(void *p);

/* The corridor rows the device uses for a tile (its closed form evaluated by the device kernel, or the
 * caller's arrays as they arrive there): offset[qry_len], length[qry_len].  A checking aid -- the
 * reference's builders restated on the device must reproduce their arrays bit for bit. */
int cvx_corridor_rows(cvx_handle h, const cvx_tile *tile, int32_t *offset, int32_t *length);

/* Host-side cost of cvx_submit without a device: lays the batch out and packs it into ordinary memory
 * `iters` times on the calling thread (plus the shared pack threads), no HIP call, nothing computed.
 * assume_page_locked != 0: blocks of sequences that lie back to back count as page-locked (they would travel
 * without packing).  *ms_per_iter = wall time per pass; *bytes_touched = bytes the host read + wrote per pass.  For sizing
 * the host side of an N-device node on a box without devices (tools/host_scale.py). */
int cvx_pack_probe(int32_t n_tiles, const cvx_tile *tiles, int32_t iters, int32_t assume_page_locked, double *ms_per_iter, uint64_t *bytes_touched);

/* Reference genome resident in HBM (decode half).
 *
 * ngmlr keeps its reference 4 bits per base (src/SequenceProvider.cpp:76-113, :333-386) and expands a
 * window to chars on the host for every alignment (DecodeRefSequenceExact :493-565, called with
 * corridor 0 from src/AlignmentBuffer.cpp:199-223).  With the encoded genome on the device a tile
 * carries (position, length) instead of decoded characters.
 *
 *   cvx_genome_encode   what _SequenceProvider::Init builds (binRef + the refStartPos table of
 *                       :415-424) from n sequences: A 0, T 1, G 2, C 3, anything else 4 (case-
 *                       insensitive), two bases per byte, high nibble first; 1000 N in front, after
 *                       every sequence its pad nibble (odd lengths) and 1000 N; sequences of 10 bases
 *                       or fewer are skipped (SequenceProvider.h:79).  bin_ref: cvx_genome_encoded_bytes()
 *                       bytes; start_table: room for n + 1 entries, *n_starts of them are written
 *                       (kept sequences + the upper bound).  Host only.
 *   cvx_genome_upload   puts an encoded genome (ours or ngmlr's own binRef / refStartPos) into HBM.
 *   cvx_genome_decode   DecodeRefSequenceExact(out + out_offset[i], position[i], length[i], 0) for n
 *                       windows on the device, results back in host memory (length[i] bytes each, the
 *                       last one NUL, as the reference writes them).
 *   cvx_submit_windows  cvx_submit for tiles whose reference is a window of the resident genome:
 *                       tiles[i].ref is ignored, the reference of tile i is the first tiles[i].ref_len
 *                       characters of the window at ref_position[i] (= the string the reference's
 *                       caller builds with length ref_len + 1), decoded on the device straight into
 *                       the batch's sequence arena.  Everything else as cvx_submit. */
typedef struct cvx_genome_s *cvx_genome;
uint64_t cvx_genome_encoded_bytes(int32_t n, const uint64_t *lengths);
int cvx_genome_encode(int32_t n, const char *const *seqs, const uint64_t *lengths, uint8_t *bin_ref,
		uint64_t *n_nibbles, uint64_t *start_table, int32_t *n_starts);
int cvx_genome_upload(cvx_handle h, const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table,
		int32_t n_starts, cvx_genome *out);
void cvx_genome_free(cvx_handle h, cvx_genome g);
int cvx_genome_decode(cvx_handle h, cvx_genome g, int32_t n, const uint64_t *position, const int32_t *length,
		const uint64_t *out_offset, char *out);
int cvx_submit_windows(cvx_handle h, cvx_genome g, int32_t n_tiles, const cvx_tile *tiles,
		const uint64_t *ref_position, cvx_job *out);
/* ABI 9: the windows a job of cvx_submit_windows decoded, back on the host: refs[i] = the ref_len characters of tile i's
 * reference (page-locked memory of the job, valid after cvx_wait until cvx_job_release; not NUL-terminated).  They come
 * back with the job's other results at one byte per base; a caller that writes MD on the host (cvx_format_alignment)
 * reads the reference bases of mismatches and deletions there and never decodes a window itself. */
int cvx_job_window_refs(cvx_handle h, cvx_job job, const char **refs);

/* Candidate search (search half): the k-mer vote of CS::RunRead (src/CS.cpp:324-398: PrefixIteration
 * src/CSstatic.cpp:23-73, PrefixSearch / AddLocationStd src/CS.cpp:57-149, CollectResultsStd :219-268) for a batch of
 * (sub-)reads over the reference's k-mer table resident in HBM.
 *
 *   cvx_index_upload   puts one unit of ngmlr's CompactPrefixTable into HBM as it lies in host memory: `ref_table_index` =
 *                      TableUnit::RefTableIndex, 4^kmer_len + 2 entries of the packed 5-byte Index (uint m_TabIndex; char
 *                      m_RevCompIndex: src/PrefixTable.h:15-31), `ref_table` = TableUnit::RefTable (uint32 locations,
 *                      src/IRefProvider.h:10-16), `unit_offset` = TableUnit::Offset.  What GetRefEntry reads
 *                      (src/PrefixTable.cpp:476-532) and nothing else.  Genomes of one table unit (< 4 Gbp).
 *   cvx_search_batch   n reads (NUL-terminated strings; lens[i] = MappedRead::length): for read i the LocationScore list
 *                      CollectResultsStd hands to AllocScores -- same entries, same order (a bin enters the list when one of its
 *                      scores first reaches the growing threshold, so the order depends on the order of the votes, which is
 *                      kept) -- in cands[cand_begin[i] .. + n_candidates[i]); n_candidates[i] = -1 where the reference gives up
 *                      ("too many candidates": every table size of the retry ladder 2^16 / 2^18 / 2^19 / 2^20 overflowed its
 *                      probe budget).  sensitivity = Config.getSensitivity() (0.8), min_kmer_hits = getMinKmerHits() (0),
 *                      bin_shift = getBinSize() (4; 1..30).  CVX_ERR_CAPACITY with *cand_used = the need when cands is too small.
 *   cvx_search_batch_ex  the same with what CS::RunRead leaves behind beside the list (ABI 5): first_bits = the vote-table size
 *                      of the first attempt, CS::c_SrchTableBitLen (0 = 16, the value a CS thread starts with; the ladder is then
 *                      first_bits + 2, + 3, ... up to 20 as at src/CS.cpp:363-394 -- the size only decides when an attempt
 *                      runs out of its probe budget, a successful attempt returns the same list at every size);
 *                      max_hit[i] = maxHitNumber of the successful attempt (MappedRead::s, src/CS.cpp:226; 0 when the read
 *                      got no list); kmer_misses[i] = kCount, the k-mers of the read found in the table in neither
 *                      orientation, summed over the attempts as the reference sums them (src/CS.cpp:67-69, :221-224: more
 *                      than 0.9 (length - k + 1) of them zero the read's mapping quality).  Either may be NULL.  The handle
 *                      keeps its staging and device buffers between calls (no allocation in the steady state) and the
 *                      calling thread sleeps while the device works: one handle per searching thread. */
typedef struct cvx_index_s *cvx_index;
typedef struct {
	uint64_t location;       /* LocationScore::Location.m_Location = ResolveBin(bin) */
	float score;             /* LocationScore::Score.f: votes of that strand */
	int32_t reverse;         /* Location.isReverse() */
} cvx_candidate;
int cvx_index_upload(cvx_handle h, int32_t kmer_len, const void *ref_table_index, const uint32_t *ref_table, uint32_t n_locations,
		uint64_t unit_offset, cvx_index *out);
void cvx_index_free(cvx_handle h, cvx_index ix);
/* One table unit built on the host from an encoded genome (cvx_genome_encode's output or ngmlr's own binRef): what
 * CompactPrefixTable::CreateTable leaves behind (src/PrefixTable.cpp:265-352, :372-463), byte for byte -- for callers without
 * an ngmlr at hand and for measurements on genome-sized tables (ngmlr itself builds and caches its table; production feeds
 * that to cvx_index_upload).  start_table[i] / seq_lengths[i]: first nibble and number of bases of the i-th kept sequence;
 * ref_skip = --kmer-skip (2), bin_shift = --bin-size (4).  ref_table_index: room for 4^kmer_len + 2 records of 5 bytes;
 * ref_table: room for ref_table_capacity locations -- CVX_ERR_CAPACITY with *n_locations = the need when that is too few
 * (the index records are complete by then).  Host only: the walks one thread per sequence, the passes over the 4^k records on
 * min(hardware threads, 16) threads (CVX_INDEX_THREADS).  Bound inside ngmlr at CompactPrefixTable::CreateTable
 * (ngmlr_amd/csrc/index_build_binding.inc): a 2 Mbp reference 2.4 -> 0.36 s, table file byte-identical. */
int cvx_index_build(const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table, const uint64_t *seq_lengths, int32_t n_seqs,
		int32_t kmer_len, int32_t ref_skip, int32_t bin_shift, void *ref_table_index, uint32_t *ref_table, uint64_t ref_table_capacity,
		uint64_t *n_locations);
/* The same table built on device `device_id` (ABI 8; cvx_index.hip): arguments, outputs (host buffers) and every byte as
 * cvx_index_build.  The reference's serial walk is local -- which windows a sequence offers depends on the 'N's around them,
 * which of those it keeps on the two sampled k-mers before -- so it runs a window per thread: sampled windows compacted in
 * walk order, the drop rule against the neighbours in that list, a histogram, row starts by a scan, and the rows by a stable
 * radix sort of (k-mer, position) (scan and sort: rocPRIM).  512 Mbp: 3.8-4.1 s on eight host threads -> 0.30-0.35 s including the
 * copies in and out (1.3 GB); CVX_ERR_NO_DEVICE without a device (the host builder above is the alternative, not a silent substitute).
 * Device memory while it runs: the genome (0.5 B per base), ~17 B per sampled window (position, k-mer, flag, the sort's second
 * buffers) plus the sort's scratch, ~21 B per possible k-mer (4^k counters, weights, row starts, records): 512 Mbp at k = 13 ~6 GB,
 * 2 Gbp ~18 GB.
 * flags: CVX_INDEX_KEEP_RESIDENT leaves the table on the device in the form the search reads; the next cvx_index_upload of these
 * very arrays (same pointers, same count) on that device takes it over instead of converting 4^k records on the host and copying
 * a gigabyte back up.  One table waits at a time; a later build, or the process's end, drops it. */
#define CVX_INDEX_KEEP_RESIDENT 1u
int cvx_index_build_device(int32_t device_id, const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table, const uint64_t *seq_lengths,
		int32_t n_seqs, int32_t kmer_len, int32_t ref_skip, int32_t bin_shift, void *ref_table_index, uint32_t *ref_table,
		uint64_t ref_table_capacity, uint64_t *n_locations, uint32_t flags);
int cvx_search_batch(cvx_handle h, cvx_index ix, int32_t n, const char *const *seqs, const int32_t *lens,
		float sensitivity, float min_kmer_hits, int32_t bin_shift,
		int32_t *n_candidates, uint64_t *cand_begin, cvx_candidate *cands, uint64_t cand_capacity, uint64_t *cand_used);
int cvx_search_batch_ex(cvx_handle h, cvx_index ix, int32_t n, const char *const *seqs, const int32_t *lens,
		float sensitivity, float min_kmer_hits, int32_t bin_shift, int32_t first_bits,
		int32_t *n_candidates, uint64_t *cand_begin, cvx_candidate *cands, uint64_t cand_capacity, uint64_t *cand_used,
		float *max_hit, int32_t *kmer_misses);
/* ABI 7: the same for reads that already lie back to back in one block: read i = arena[offsets[i] .. offsets[i + 1] - 1), with its
 * terminating NUL at offsets[i + 1] - 1 (offsets has n + 1 entries).  No per-read pointer chasing on the host, one copy of the block
 * -- and none at all when the block lies in memory from cvx_host_alloc: the device pulls it as it is (a caller that keeps its
 * reads resident: 100 000 sub-reads per call at genome scale spent 4.6 x the kernels' time marshalling strings).
 * Outputs as for cvx_search_batch_ex; `cands` in memory from cvx_host_alloc comes back without a staging copy. */
int cvx_search_batch_arena(cvx_handle h, cvx_index ix, int32_t n, const uint8_t *arena, const uint64_t *offsets,
		float sensitivity, float min_kmer_hits, int32_t bin_shift, int32_t first_bits,
		int32_t *n_candidates, uint64_t *cand_begin, cvx_candidate *cands, uint64_t cand_capacity, uint64_t *cand_used,
		float *max_hit, int32_t *kmer_misses);

/* attempts[i] = vote-table sizes tried for read i in the handle's last cvx_search_batch(_ex) call (ABI 6): 1 = the first attempt
 * produced the list; more = the first attempt ran out of its probe budget -- the event CS::RunRead counts in m_Overflows
 * (src/CS.cpp:359-364), which CS::DoRun's per-batch adaptation of the table size reads (src/CS.cpp:482-489). */
int cvx_search_last_attempts(cvx_handle h, int32_t n, int32_t *attempts);

/* Sub-read scoring: what StrippedSW::BatchScore / SingleScore return
 * (reference src/StrippedSW.cpp:118-203 over ssw.c): refs/qrys are NUL-terminated strings,
 * scores[i] receives the alignment score (match +1, mismatch -1, N 0, gap 255 per base -- the
 * reference passes -1 into ssw's uint8_t gap weights) or -1.0f for sequences of 100000
 * characters or more.  Exact for scores below 32767. */
int cvx_score_batch(cvx_handle h, int32_t n, const char *const *refs, const char *const *qrys, float *scores);
/* duration of the scoring kernel of the handle's last cvx_score_batch (HIP events on its stream): the device-resident
 * rate, beside the rate of the whole call (strings in host memory in, scores out) */
int cvx_score_kernel_ms(cvx_handle h, float *ms);
/* The same for every next-row stage (ABI 6): device time of the kernels of the handle's last call of that stage, from HIP
 * events on the stream they ran on -- CVX_STAGE_SCORE: cvx_score_batch; CVX_STAGE_DECODE: decode_windows_kernel of
 * cvx_genome_decode; CVX_STAGE_SEARCH: every kernel of cvx_search_batch(_ex) (count, vote batches of each attempt of the
 * ladder, compaction) summed, without the host round trips between them. */
enum { CVX_STAGE_SCORE = 0, CVX_STAGE_DECODE = 1, CVX_STAGE_SEARCH = 2 };
int cvx_stage_kernel_ms(cvx_handle h, int32_t stage, float *ms);

/* Host-side text stage (convertCigar, src/ConvexAlignFast.cpp:112-333, and the
 * N-clip flags of :493-528).  Pure host code, no device needed. */
typedef struct {
	int32_t ret;              /* SingleAlign return value: QStart + sum(M) + sum(I) + QEnd, or -1 */
	float score;              /* Align::Score (-1.0f when ret < 0) */
	int32_t position_offset;
	int32_t qstart, qend;     /* including externalQStart / externalQEnd */
	int32_t nm;
	float identity;
	int32_t alignment_length;
	int32_t cigar_op_count;
	int32_t sv_type;
	int32_t first_ref, first_read, last_ref, last_read;
	int32_t nm_count;         /* PositionNM triples written */
	int32_t cigar_len, md_len;/* full lengths (text is truncated to the given capacity) */
} cvx_alignment_text;

int cvx_format_alignment(const cvx_result *r, const uint32_t *ops_arena,
		const char *ref, int32_t ref_len, int32_t qry_len,
		int32_t ext_qstart, int32_t ext_qend,
		char *cigar, int32_t cigar_cap, char *md, int32_t md_cap,
		int32_t *nm_triples, int32_t nm_cap, cvx_alignment_text *out);

/* The same for a whole batch on `n_threads` host threads (once the fill is on
 * the GPU, text generation is the serial tail).  tiles[i] supplies ref/ref_len/qry_len;
 * bufs[i] the caller's per-tile output buffers (any pointer may be NULL with capacity 0).
 * n_threads <= 0 picks the hardware concurrency.  Returns the first error, else CVX_OK. */
typedef struct {
	char *cigar;
	char *md;
	int32_t *nm_triples;
	int32_t cigar_cap, md_cap, nm_cap;
	int32_t ext_qstart, ext_qend;
} cvx_text_buffers;

int cvx_format_batch(int32_t n, const cvx_result *results, const uint32_t *ops_arena,
		const cvx_tile *tiles, const cvx_text_buffers *bufs, cvx_alignment_text *out,
		int32_t n_threads);

/* The same on the device: CIGAR, MD and the scalar fields of every tile of a finished
 * job, from the ops and sequences still resident in HBM -- one launch instead of a host core per ~70 Mbp/s.
 * After cvx_wait and before cvx_job_release.  ext_qstart / ext_qend: externalQStart / externalQEnd per
 * tile (NULL = zeros).  out[i] = what cvx_format_alignment would return for tile i (nm_count = the number
 * of profile entries the host form would write; the profile itself stays with the host form).  *text
 * points at page-locked memory owned by the job, valid until cvx_job_release: tile i's CIGAR is the
 * NUL-terminated string at (*text)[text_off[i]], its MD follows that NUL. */
int cvx_job_text(cvx_handle h, cvx_job job, const int32_t *ext_qstart, const int32_t *ext_qend,
		cvx_alignment_text *out, uint64_t *text_off, const char **text, uint64_t *text_bytes);

/* nmPerPosition (reference src/ConvexAlignFast.cpp:76-98,186-269: one (refPosition, readPosition, nm) triple per
 * EQ / X / D column once both positions passed 16; read by detectMisalignment, src/AlignmentBuffer.cpp:1320) of the
 * tiles [first, first + count) of a finished job, computed on the device from the resident ops.  After cvx_job_text
 * (which counts the entries: cvx_alignment_text.nm_count) and before cvx_job_release.  entry_off[count + 1]: tile
 * first + i owns the triples [entry_off[i], entry_off[i + 1]).  triples: room for cap_entries triples of three
 * int32 (CVX_ERR_CAPACITY, with entry_off filled in, when that is too little) -- or NULL: the profile stays in HBM
 * (a consumer on the device; measurement).  A range, not the whole job, because the profile is 12 bytes per column:
 * several GB for a full batch of long reads.  *kernel_ms (may be NULL): the kernel's own duration. */
int cvx_job_nm_profile(cvx_handle h, cvx_job job, int32_t first, int32_t count, uint64_t *entry_off,
		int32_t *triples, uint64_t cap_entries, double *kernel_ms);
/* ABI 9: the same with the triples left in page-locked memory the job owns (*triples, valid until cvx_job_release): one DMA
 * at PCIe speed instead of a staged copy into the caller's pageable array -- the profile is 125 kB per 10 kb alignment, and
 * a dispatcher that runs the text stage per launch (CVX_DEVICE_TEXT=1) spent most of that stage in the copy. */
int cvx_job_nm_profile_resident(cvx_handle h, cvx_job job, int32_t first, int32_t count, uint64_t *entry_off,
		const int32_t **triples, double *kernel_ms);
/* ABI 9: cvx_job_text and cvx_job_nm_profile_resident of the whole job in one call -- two round trips to the device (the sizes
 * of both, then the strings and triples of both) instead of four.  Outputs as in those two calls; nm_entry_off[n_tiles + 1].
 * Threads: the text-stage calls (cvx_job_text, cvx_job_nm_profile*, this one) touch only the finished job they are given and
 * the handle's text stream, so ONE other thread may run them while the handle's own thread submits and waits for other jobs
 * (BatchingAligner's text thread does); two text-stage calls on one handle at a time are not supported. */
int cvx_job_text_all(cvx_handle h, cvx_job job, const int32_t *ext_qstart, const int32_t *ext_qend,
		cvx_alignment_text *out, uint64_t *text_off, const char **text, uint64_t *text_bytes,
		uint64_t *nm_entry_off, const int32_t **triples);
/* the entry offsets alone (entry_off[0 .. count], as above): what a caller needs to size `triples` -- no profile is
 * computed and nothing is allocated for it */
int cvx_job_nm_sizes(cvx_handle h, cvx_job job, int32_t first, int32_t count, uint64_t *entry_off);

/* The same kernel over op lists the caller holds (someone who kept cvx_result + ops and released the job): tile i's
 * ops are ops_arena[results[i].ops_begin ... + n_ops), ops_total = ints in the arena.  entry_off[n + 1] as above;
 * triples == NULL: sizes only. */
int cvx_nm_profile_ops(cvx_handle h, int32_t n, const cvx_result *results, const uint32_t *ops_arena, uint64_t ops_total,
		uint64_t *entry_off, int32_t *triples, uint64_t cap_entries);

/* ---- SAM record assembly (reference src/SAMWriter.cpp:87-224, :308-357) ----
 *
 * The text of one SAM record from the fields ngmlr's writer holds (MappedRead, its LocationScore and Align), byte
 * for byte what SAMWriter::DoWriteReadGeneric prints: mandatory fields, then RG AS NM XI XS XE XR MD SV SA QS QE CV,
 * "<read length>S" + CG:B:I when bam_cigar_fix is set and the CIGAR has 65 536 operations or more, hard clipping.
 * Positions are the reference's 0-based Location.m_Location; the record prints + 1 (report_offset, :19). */
typedef struct {              /* another alignment of the same read: one SA:Z entry (:183-204) */
	const char *ref_name;     /* SequenceProvider.GetRefName(Location.getrefId(), len) */
	int32_t ref_name_len;
	uint32_t location;        /* Location.m_Location */
	int32_t reverse;          /* Location.isReverse() */
	const char *cigar;        /* Align::pBuffer1 */
	int32_t mq, nm;           /* Align::MQ, Align::NM */
} cvx_sam_other;

typedef struct {
	const char *read_name;    /* MappedRead::name */
	const char *seq;          /* MappedRead::Seq, or RevSeq for a reverse-strand record (:101-108) */
	char *qual;               /* MappedRead::qlty or NULL ("*").  Reversed IN PLACE over [0, read_length) when the record is
	                           * on the reverse strand and the string is not empty -- every time such a record is written, as
	                           * the reference does (:104-106) */
	int32_t read_length;      /* MappedRead::length */
	int32_t flags;            /* the caller's flags; 0x800 (not primary) and 0x10 (reverse) are added here (:97-108) */
	int32_t primary;          /* Align::primary */
	int32_t reverse;
	const char *ref_name;
	int32_t ref_name_len;
	uint32_t location;
	int32_t mq;               /* Align::MQ */
	const char *cigar;        /* Align::pBuffer1 */
	const char *md;           /* Align::pBuffer2 */
	int32_t cigar_op_count;   /* Align::cigarOpCount */
	const char *mate_ref_name;/* pRefName: "*", "=" or a name */
	int32_t mate_location;    /* pLoc (-1 for none) */
	int32_t template_length;  /* pDist */
	float score;              /* Scores[scoreID].Score.f: AS and XE print (int) score */
	int32_t nm;               /* Align::NM */
	float identity;           /* Align::Identity: XI prints round(identity * 10000) / 10000 with %g */
	int32_t qstart, qend;     /* Align::QStart / QEnd */
	int32_t sv_type;          /* Align::svType: SV:i printed when > -1 */
	int32_t n_others;         /* the read's other alignments that are not skipped, in order (i != scoreID && !Alignments[i].skip) */
	const cvx_sam_other *others;
	const char *rg_id;        /* Config.getRgId() or NULL */
	int32_t hard_clip;        /* Config.getHardClip() */
	int32_t bam_cigar_fix;    /* Config.getBamCigarFix() */
	int32_t skip;             /* Align::skip */
} cvx_sam_record;

typedef struct {              /* DoWriteUnmappedReadGeneric (:308-357); flags |= 0x4 */
	const char *read_name, *seq;
	const char *qual;         /* NULL: "*" */
	int32_t read_length, flags;
	const char *ref_name;     /* NULL: "*" (refId -1) */
	int32_t ref_name_len;
	int32_t location;         /* loc: printed + 1 */
	char mate_ref;            /* pRefName, one character */
	int32_t mate_location, template_length;
	const char *rg_id;
} cvx_sam_unmapped;

/* One record (the trailing '\n' included, no NUL).  *len = bytes of the record; CVX_ERR_CAPACITY when cap is smaller
 * (nothing is reversed then, the call can be repeated). */
int cvx_sam_record_text(cvx_sam_record *r, char *out, uint64_t cap, uint64_t *len);
int cvx_sam_unmapped_text(const cvx_sam_unmapped *r, char *out, uint64_t cap, uint64_t *len);

/* n records on the process's pack threads: record i is out[offsets[i] ... offsets[i + 1]).  Same bytes as n calls of
 * cvx_sam_record_text in order -- including the quality strings' in-place reversal when several records share one
 * (two records of one read).  CVX_ERR_CAPACITY with offsets[n] = the need when cap is too small (nothing reversed). */
int cvx_sam_batch(int32_t n, cvx_sam_record *recs, char *out, uint64_t cap, uint64_t *offsets);

#ifdef __cplusplus
}
#endif
#endif
