"""ctypes binding of include/cvx_align.h (libcvxalign.so).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C ngmlr_amd/csrc``.
Loading fails loudly when it is missing: there is no Python or CPU fallback for the
compute entry points.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVX_LIB") or os.path.join(HERE, "libcvxalign.so")   # CVX_LIB: A/B builds while tuning

CVX_OK = 0
STAGE_SCORE, STAGE_DECODE, STAGE_SEARCH = 0, 1, 2      # cvx_stage_kernel_ms
ERR_NAMES = {0: "CVX_OK", -1: "CVX_ERR_NO_DEVICE", -2: "CVX_ERR_PARAMS", -3: "CVX_ERR_ARG",
             -4: "CVX_ERR_OOM", -5: "CVX_ERR_HIP", -6: "CVX_ERR_CAPACITY"}
TILE_STATUS = {0: "ok", 1: "invalid-row0", 2: "invalid-edge", 3: "invalid-length", 4: "too-large",
               5: "empty", -1: "unsupported"}

EXPORTS = ("cvx_last_error", "cvx_abi_version", "cvx_source_id", "cvx_device_count", "cvx_device_synchronize", "cvx_create", "cvx_destroy",
           "cvx_align_batch", "cvx_batch_upload", "cvx_batch_run", "cvx_batch_timing",
           "cvx_batch_ops_total", "cvx_batch_launch_info", "cvx_batch_download", "cvx_batch_free",
           "cvx_submit", "cvx_wait", "cvx_job_timing", "cvx_job_launch_info", "cvx_job_release",
           "cvx_format_alignment", "cvx_format_batch", "cvx_score_batch",
           "cvx_genome_encoded_bytes", "cvx_genome_encode", "cvx_genome_upload", "cvx_genome_free",
           "cvx_genome_decode", "cvx_submit_windows", "cvx_job_text",
           "cvx_host_alloc", "cvx_host_free", "cvx_corridor_rows", "cvx_pack_probe", "cvx_build_id", "cvx_job_poll", "cvx_score_kernel_ms",
           "cvx_index_upload", "cvx_index_free", "cvx_search_batch", "cvx_search_batch_ex", "cvx_job_nm_profile", "cvx_job_nm_profile_resident", "cvx_job_text_all", "cvx_job_window_refs", "cvx_job_nm_sizes", "cvx_nm_profile_ops",
           "cvx_sam_record_text", "cvx_sam_unmapped_text", "cvx_sam_batch", "cvx_stage_kernel_ms", "cvx_search_last_attempts", "cvx_index_build", "cvx_index_build_device",
           "cvx_corridor_fit", "cvx_corridor_fit_batch", "cvx_create_ex", "cvx_runtime_regime", "cvx_search_batch_arena")


class CvxParams(C.Structure):
    _fields_ = [("match", C.c_float), ("mismatch", C.c_float), ("gap_open", C.c_float),
                ("gap_extend", C.c_float), ("gap_extend_min", C.c_float), ("gap_decay", C.c_float)]


class CvxTile(C.Structure):
    _fields_ = [("ref", C.c_char_p), ("qry", C.c_char_p), ("row_offset", C.c_void_p),
                ("row_length", C.c_void_p), ("ref_len", C.c_int32), ("qry_len", C.c_int32),
                ("row_stride_bytes", C.c_int32), ("corridor_kind", C.c_int32),
                ("corridor_k", C.c_float), ("corridor_d", C.c_float), ("corridor_right", C.c_float),
                ("corridor_offset", C.c_int32), ("corridor_width", C.c_int32), ("reserved", C.c_int32)]


class CvxRegime(C.Structure):
    _fields_ = [("hw_queues_env", C.c_int32), ("hw_queues_set_by_library", C.c_int32), ("blocking_sync", C.c_int32),
                ("blocking_sync_why", C.c_int32), ("service_streams", C.c_int32), ("runtime_up_at_load", C.c_int32), ("reserved", C.c_int32 * 2)]


CORRIDOR_ROWS, CORRIDOR_AFFINE, CORRIDOR_CONST = 0, 1, 2
assert C.sizeof(CvxTile) == 72


class CvxResult(C.Structure):
    _fields_ = [("score", C.c_float), ("status", C.c_int32), ("best_ref_index", C.c_int32),
                ("best_read_index", C.c_int32), ("ref_position", C.c_int32), ("qstart", C.c_int32),
                ("qend", C.c_int32), ("n_ops", C.c_int32), ("ops_begin", C.c_uint64),
                ("cells", C.c_uint64)]


class CvxTiming(C.Structure):
    _fields_ = [("plan_ms", C.c_float), ("fill_ms", C.c_float), ("backtrack_ms", C.c_float),
                ("total_ms", C.c_float), ("cells", C.c_uint64), ("active_cells", C.c_uint64),
                ("dir_bytes", C.c_uint64), ("n_fill_launches", C.c_int32), ("n_tiles_fast", C.c_int32),
                ("n_tiles_redone", C.c_int32), ("n_tiles_chained", C.c_int32),
                ("chain_task_ticks", C.c_uint64), ("chain_poll_ticks", C.c_uint64)]


class CvxLaunchInfo(C.Structure):
    _fields_ = [("slots_per_lane", C.c_int32), ("waves", C.c_int32), ("wrap16", C.c_int32),
                ("n_tiles", C.c_int32), ("ms", C.c_float), ("kind", C.c_int32), ("cells", C.c_uint64),
                ("active_cells", C.c_uint64), ("alg_bytes", C.c_uint64), ("read_bases", C.c_uint64)]


class CvxAlignmentText(C.Structure):
    _fields_ = [("ret", C.c_int32), ("score", C.c_float), ("position_offset", C.c_int32),
                ("qstart", C.c_int32), ("qend", C.c_int32), ("nm", C.c_int32), ("identity", C.c_float),
                ("alignment_length", C.c_int32), ("cigar_op_count", C.c_int32), ("sv_type", C.c_int32),
                ("first_ref", C.c_int32), ("first_read", C.c_int32), ("last_ref", C.c_int32),
                ("last_read", C.c_int32), ("nm_count", C.c_int32), ("cigar_len", C.c_int32),
                ("md_len", C.c_int32)]


class CvxSamOther(C.Structure):
    _fields_ = [("ref_name", C.c_char_p), ("ref_name_len", C.c_int32), ("location", C.c_uint32), ("reverse", C.c_int32),
                ("cigar", C.c_char_p), ("mq", C.c_int32), ("nm", C.c_int32)]


class CvxSamRecord(C.Structure):
    _fields_ = [("read_name", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_void_p), ("read_length", C.c_int32),
                ("flags", C.c_int32), ("primary", C.c_int32), ("reverse", C.c_int32), ("ref_name", C.c_char_p),
                ("ref_name_len", C.c_int32), ("location", C.c_uint32), ("mq", C.c_int32), ("cigar", C.c_char_p),
                ("md", C.c_char_p), ("cigar_op_count", C.c_int32), ("mate_ref_name", C.c_char_p),
                ("mate_location", C.c_int32), ("template_length", C.c_int32), ("score", C.c_float), ("nm", C.c_int32),
                ("identity", C.c_float), ("qstart", C.c_int32), ("qend", C.c_int32), ("sv_type", C.c_int32),
                ("n_others", C.c_int32), ("others", C.POINTER(CvxSamOther)), ("rg_id", C.c_char_p),
                ("hard_clip", C.c_int32), ("bam_cigar_fix", C.c_int32), ("skip", C.c_int32)]


class CvxSamUnmapped(C.Structure):
    _fields_ = [("read_name", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_char_p), ("read_length", C.c_int32),
                ("flags", C.c_int32), ("ref_name", C.c_char_p), ("ref_name_len", C.c_int32), ("location", C.c_int32),
                ("mate_ref", C.c_char), ("mate_location", C.c_int32), ("template_length", C.c_int32), ("rg_id", C.c_char_p)]


class CvxTextBuffers(C.Structure):
    _fields_ = [("cigar", C.c_void_p), ("md", C.c_void_p), ("nm_triples", C.c_void_p),
                ("cigar_cap", C.c_int32), ("md_cap", C.c_int32), ("nm_cap", C.c_int32),
                ("ext_qstart", C.c_int32), ("ext_qend", C.c_int32)]


class CvxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, str(code)), msg))
        self.code = code


_libs = {}


def load(path: str = None) -> C.CDLL:
    """Load libcvxalign.so (raises if it has not been built).  `path`: another build of the
    library (A/B runs while tuning, tools/ab_fill.py); default the in-tree one / $CVX_LIB."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP path has no fallback)" % path)
    lib = C.CDLL(path)
    lib.cvx_last_error.restype = C.c_char_p
    lib.cvx_abi_version.restype = C.c_int
    lib.cvx_build_id.restype = C.c_char_p
    lib.cvx_source_id.restype = C.c_char_p
    lib.cvx_source_id.argtypes = [C.c_char_p]
    lib.cvx_device_count.restype = C.c_int
    lib.cvx_device_synchronize.argtypes = [C.c_int]
    lib.cvx_create.argtypes = [C.c_int, C.POINTER(CvxParams), C.c_uint64, C.POINTER(C.c_void_p)]
    lib.cvx_destroy.argtypes = [C.c_void_p]
    lib.cvx_destroy.restype = None
    lib.cvx_align_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CvxTile), C.POINTER(CvxResult),
                                    C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.cvx_batch_upload.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CvxTile), C.POINTER(C.c_void_p)]
    lib.cvx_batch_run.argtypes = [C.c_void_p, C.c_void_p]
    lib.cvx_batch_timing.argtypes = [C.c_void_p, C.POINTER(CvxTiming)]
    lib.cvx_batch_launch_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CvxLaunchInfo)]
    lib.cvx_batch_ops_total.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.cvx_batch_download.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CvxResult), C.c_void_p,
                                       C.c_uint64, C.POINTER(C.c_uint64)]
    lib.cvx_batch_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.cvx_batch_free.restype = None
    lib.cvx_submit.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CvxTile), C.POINTER(C.c_void_p)]
    lib.cvx_wait.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(CvxResult)), C.POINTER(C.POINTER(C.c_uint32)),
                             C.POINTER(C.c_uint64)]
    lib.cvx_job_poll.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.cvx_job_timing.argtypes = [C.c_void_p, C.POINTER(CvxTiming)]
    lib.cvx_job_launch_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CvxLaunchInfo)]
    lib.cvx_job_release.argtypes = [C.c_void_p, C.c_void_p]
    lib.cvx_job_release.restype = None
    lib.cvx_format_batch.argtypes = [C.c_int32, C.POINTER(CvxResult), C.c_void_p, C.POINTER(CvxTile),
                                     C.POINTER(CvxTextBuffers), C.POINTER(CvxAlignmentText), C.c_int32]
    lib.cvx_genome_encoded_bytes.restype = C.c_uint64
    lib.cvx_genome_encoded_bytes.argtypes = [C.c_int32, C.c_void_p]
    lib.cvx_genome_encode.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                      C.c_void_p, C.POINTER(C.c_int32)]
    lib.cvx_genome_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.cvx_genome_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.cvx_genome_free.restype = None
    lib.cvx_genome_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cvx_submit_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(CvxTile), C.c_void_p, C.POINTER(C.c_void_p)]
    lib.cvx_job_text.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CvxAlignmentText), C.c_void_p,
                                 C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    lib.cvx_job_nm_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
    lib.cvx_job_nm_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.cvx_job_nm_profile_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double)]
    lib.cvx_job_window_refs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cvx_job_text_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CvxAlignmentText), C.c_void_p,
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_void_p)]
    lib.cvx_nm_profile_ops.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.cvx_sam_record_text.argtypes = [C.POINTER(CvxSamRecord), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.cvx_sam_unmapped_text.argtypes = [C.POINTER(CvxSamUnmapped), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.cvx_sam_batch.argtypes = [C.c_int32, C.POINTER(CvxSamRecord), C.c_void_p, C.c_uint64, C.c_void_p]
    lib.cvx_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
    lib.cvx_host_free.argtypes = [C.c_void_p]
    lib.cvx_host_free.restype = None
    lib.cvx_corridor_rows.argtypes = [C.c_void_p, C.POINTER(CvxTile), C.c_void_p, C.c_void_p]
    lib.cvx_pack_probe.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.cvx_index_upload.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.cvx_index_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.cvx_index_free.restype = None
    lib.cvx_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_float, C.c_float, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.cvx_search_batch_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
    lib.cvx_search_batch_arena.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
    lib.cvx_score_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.cvx_stage_kernel_ms.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]
    lib.cvx_search_last_attempts.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.cvx_index_build_device.argtypes = [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32]
    lib.cvx_index_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.cvx_runtime_regime.argtypes = [C.c_int, C.POINTER(CvxRegime)]
    lib.cvx_create_ex.argtypes = [C.c_int, C.POINTER(CvxParams), C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.cvx_corridor_fit_batch.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.cvx_corridor_fit.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CvxTile)]
    lib.cvx_score_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p]
    lib.cvx_format_alignment.argtypes = [C.POINTER(CvxResult), C.c_void_p, C.c_char_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int32,
                                         C.c_char_p, C.c_int32, C.c_void_p, C.c_int32,
                                         C.POINTER(CvxAlignmentText)]
    _libs[path] = lib
    return lib


def check(rc: int) -> None:
    if rc != CVX_OK:
        raise CvxError(rc, load().cvx_last_error().decode(errors="replace"))


def corridor_fit(row_offset, row_length, ref_len, qry_len, stride_bytes=4):
    """cvx_corridor_fit on int32 arrays (packed, or the offset / length columns of a CorridorLine[] with stride 16) ->
    (kind, k, d, right, offset, width)."""
    import numpy as np
    lib = load()
    off = np.ascontiguousarray(row_offset, dtype=np.int32) if stride_bytes == 4 else row_offset
    ln = np.ascontiguousarray(row_length, dtype=np.int32) if stride_bytes == 4 else row_length
    n = len(off) if stride_bytes == 4 else qry_len
    form = CvxTile()
    rc = lib.cvx_corridor_fit(off.ctypes.data, ln.ctypes.data, stride_bytes, n, ref_len, qry_len, C.byref(form))
    if rc != CVX_OK:
        raise RuntimeError("cvx_corridor_fit: %s" % ERR_NAMES.get(rc, rc))
    return (form.corridor_kind, form.corridor_k, form.corridor_d, form.corridor_right, form.corridor_offset, form.corridor_width)
