"""Host-side mirror of the reference's aligner surface for the convex-gap hot path.

``ConvexAlignHip`` plays the role of ``Convex::ConvexAlignFast`` behind ngmlr's
``IAlignment`` interface (reference src/IAlignment.h:211-247): constructed with the
same six scoring floats (src/AlignmentBuffer.h:345-363), ``single_align`` has the
argument meaning and error behaviour of the corridor ``SingleAlign`` overload
(returns -1 and Score -1.0 for "no valid alignment"), and ``batch_align`` is the
``BatchAlign`` slot the reference leaves unimplemented (src/ConvexAlignFast.cpp:441-450).
All compute goes through the C ABI of include/cvx_align.h; the C++ twin of this class
(the actual drop-in) is ngmlr_amd/csrc/convex_align_hip.{h,cpp}.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import capi

DEFAULT_SCORING = dict(match=2.0, mismatch=-5.0, gap_open=-5.0, gap_extend=-5.0,
                       gap_extend_min=-1.0, gap_decay=0.15)  # src/IConfig.h:50-55


class ConvexAlignHip:
    def __init__(self, device: int = 0, max_matrix_mb: int = 10000, lib_path: str = None, **scoring):
        self.lib = capi.load(lib_path)
        sc = dict(DEFAULT_SCORING)
        sc.update(scoring)
        self.params = capi.CvxParams(sc["match"], sc["mismatch"], sc["gap_open"], sc["gap_extend"],
                                     sc["gap_extend_min"], sc["gap_decay"])
        self.h = C.c_void_p()
        capi.check(self.lib.cvx_create(device, C.byref(self.params), max_matrix_mb, C.byref(self.h)))

    def stage_kernel_ms(self, stage: int) -> float:
        """cvx_stage_kernel_ms: device time (HIP events) of the kernels of the handle's last call of a next-row stage
        (capi.STAGE_SCORE / STAGE_DECODE / STAGE_SEARCH)."""
        ms = C.c_float()
        capi.check(self.lib.cvx_stage_kernel_ms(self.h, stage, C.byref(ms)))
        return float(ms.value)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.cvx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # IAlignment::GetAlignBatchSize(): the reference's convex aligner answers 0
    # ("no batching"); this backend takes any batch.
    def get_align_batch_size(self) -> int:
        return 1 << 20

    # ------------------------------------------------------------------ staged API
    def _pack(self, tiles: Sequence, closed_form: bool = False):
        """cvx_tile[] for per-tile objects.  closed_form: tiles that carry a corridor descriptor (Tile.desc) hand
        over that instead of their row arrays."""
        n = len(tiles)
        arr = (capi.CvxTile * max(n, 1))()
        keep = []
        for i, t in enumerate(tiles):
            arr[i].ref = t.ref
            arr[i].qry = t.qry
            arr[i].ref_len = len(t.ref)
            arr[i].qry_len = len(t.qry)
            desc = getattr(t, "desc", None) if closed_form else None
            if desc is not None:
                (arr[i].corridor_kind, arr[i].corridor_k, arr[i].corridor_d, arr[i].corridor_right,
                 arr[i].corridor_offset, arr[i].corridor_width) = desc
                keep.append((t.ref, t.qry))
                continue
            off = np.ascontiguousarray(t.row_offset, dtype=np.int32)
            ln = np.ascontiguousarray(t.row_length, dtype=np.int32)
            keep.append((off, ln, t.ref, t.qry))
            arr[i].row_offset = off.ctypes.data
            arr[i].row_length = ln.ctypes.data
            arr[i].row_stride_bytes = 4
        return arr, keep

    def corridor_rows(self, tile):
        """cvx_corridor_rows: the (offset, length) rows the DEVICE derives from the tile's closed form."""
        arr, keep = self._pack([tile], closed_form=True)
        H = len(tile.qry)
        off = np.zeros(max(H, 1), dtype=np.int32)
        ln = np.zeros(max(H, 1), dtype=np.int32)
        capi.check(self.lib.cvx_corridor_rows(self.h, arr, off.ctypes.data, ln.ctypes.data))
        return off[:H], ln[:H]

    def upload(self, tiles: Sequence, closed_form: bool = False) -> "DeviceBatch":
        arr, keep = self._pack(tiles, closed_form)
        b = C.c_void_p()
        capi.check(self.lib.cvx_batch_upload(self.h, len(tiles), arr, C.byref(b)))
        return DeviceBatch(self, b, list(tiles))

    def timed_host_path(self, tiles: Sequence) -> dict:
        """Host buffers in, results out: what cvx_align_batch does, with the stages timed
        separately (seconds of the C calls only; ctypes packing of the tile table excluded)."""
        import time
        arr, keep = self._pack(tiles)
        b = C.c_void_p()
        t0 = time.perf_counter()
        capi.check(self.lib.cvx_batch_upload(self.h, len(tiles), arr, C.byref(b)))
        t1 = time.perf_counter()
        batch = DeviceBatch(self, b, list(tiles))
        try:
            batch.run()
            t2 = time.perf_counter()
            batch.download()
            t3 = time.perf_counter()
        finally:
            batch.free()
        t4 = time.perf_counter()
        return {"upload_s": t1 - t0, "run_s": t2 - t1, "download_s": t3 - t2, "free_s": t4 - t3, "total_s": t4 - t0}

    # ------------------------------------------------------------------ streaming API
    def submit(self, tiles) -> "Job":
        """cvx_submit: host buffers in, nothing waits.  `tiles`: a synth.TileSet (its table points
        into the flat arrays) or a sequence of Tile objects."""
        if hasattr(tiles, "table"):
            tab = tiles.table()
            arr = tab.ctypes.data_as(C.POINTER(capi.CvxTile))
            keep = (tiles, tab)
            n = len(tab)
        else:
            arr, keep = self._pack(tiles)
            n = len(tiles)
        j = C.c_void_p()
        capi.check(self.lib.cvx_submit(self.h, n, arr, C.byref(j)))
        return Job(self, j, n, keep)

    # ------------------------------------------------------------------ reference-shaped API
    def batch_align(self, tiles: Sequence, want_nm: bool = True, closed_form: bool = False) -> List[dict]:
        """N x SingleAlign: returns one Align-like dict per tile (keys = the Align fields)."""
        batch = self.upload(tiles, closed_form)
        try:
            batch.run()
            return batch.alignments(want_nm=want_nm)
        finally:
            batch.free()

    def single_align(self, tile, want_nm: bool = True) -> dict:
        return self.batch_align([tile], want_nm=want_nm)[0]


RESULT_DTYPE = np.dtype([("score", np.float32), ("status", np.int32), ("best_ref_index", np.int32),
                         ("best_read_index", np.int32), ("ref_position", np.int32), ("qstart", np.int32),
                         ("qend", np.int32), ("n_ops", np.int32), ("ops_begin", np.uint64), ("cells", np.uint64)])
assert RESULT_DTYPE.itemsize == C.sizeof(capi.CvxResult)


class Job:
    """One batch travelling through the streaming form (cvx_submit / cvx_wait / cvx_job_release)."""

    def __init__(self, aligner: "ConvexAlignHip", handle, n: int, keep):
        self.al, self.j, self.n, self._keep = aligner, handle, n, keep
        self.results = None
        self.ops = None

    def wait(self):
        """Blocks until the job is done.  Returns (results, ops): numpy views of the job's pinned
        result records (RESULT_DTYPE) and dense ops arena, valid until release()."""
        res = C.POINTER(capi.CvxResult)()
        ops = C.POINTER(C.c_uint32)()
        n_ops = C.c_uint64()
        capi.check(self.al.lib.cvx_wait(self.al.h, self.j, C.byref(res), C.byref(ops), C.byref(n_ops)))
        self.res_ptr = res
        self.results = (np.ctypeslib.as_array(C.cast(res, C.POINTER(C.c_uint8)), shape=(self.n * RESULT_DTYPE.itemsize,))
                        .view(RESULT_DTYPE) if self.n else np.zeros(0, RESULT_DTYPE))
        self.ops = (np.ctypeslib.as_array(ops, shape=(int(n_ops.value),)) if n_ops.value else np.zeros(0, np.uint32))
        return self.results, self.ops

    def timing(self) -> capi.CvxTiming:
        t = capi.CvxTiming()
        capi.check(self.al.lib.cvx_job_timing(self.j, C.byref(t)))
        return t

    def launches(self) -> list:
        t = self.timing()
        out = []
        for i in range(t.n_fill_launches):
            li = capi.CvxLaunchInfo()
            capi.check(self.al.lib.cvx_job_launch_info(self.j, i, C.byref(li)))
            out.append({k: getattr(li, k) for k, _ in capi.CvxLaunchInfo._fields_})
        return out

    def text_raw(self, ext_qstart=None, ext_qend=None):
        """cvx_job_text: the device-side text stage of every tile of this finished job.
        -> (cvx_alignment_text array, text offsets uint64[n], text bytes)"""
        n = self.n
        out = (capi.CvxAlignmentText * max(n, 1))()
        off = np.zeros(max(n, 1), dtype=np.uint64)
        text = C.c_char_p()
        nbytes = C.c_uint64()
        eq = np.ascontiguousarray(ext_qstart, dtype=np.int32) if ext_qstart is not None else None
        ee = np.ascontiguousarray(ext_qend, dtype=np.int32) if ext_qend is not None else None
        lib = self.al.lib
        lib.cvx_job_text.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(capi.CvxAlignmentText),
                                     C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        tp = C.c_void_p()
        capi.check(lib.cvx_job_text(self.al.h, self.j, eq.ctypes.data if eq is not None else None,
                                    ee.ctypes.data if ee is not None else None, out, off.ctypes.data,
                                    C.byref(tp), C.byref(nbytes)))
        buf = (C.c_char * int(nbytes.value)).from_address(tp.value) if nbytes.value else None
        return out, off, buf

    def text(self, ext_qstart=None, ext_qend=None) -> List[dict]:
        """-> one dict per tile with the cvx_alignment_text fields + 'cigar' / 'md' (device-side text stage)."""
        out, off, buf = self.text_raw(ext_qstart, ext_qend)
        raw = buf.raw if buf is not None else b""
        res = []
        for i in range(self.n):
            t = out[i]
            d = {f: getattr(t, f) for f, _ in capi.CvxAlignmentText._fields_}
            d["score_bits"] = int(np.float32(t.score).view(np.uint32))
            o = int(off[i])
            d["cigar"] = raw[o:o + t.cigar_len].decode()
            d["md"] = raw[o + t.cigar_len + 1:o + t.cigar_len + 1 + t.md_len].decode()
            res.append(d)
        return res

    def nm_profile(self, first: int = 0, count: Optional[int] = None, to_host: bool = True):
        """cvx_job_nm_profile (after text()/text_raw()): nmPerPosition of the tiles [first, first + count) from the device.
        -> (entry offsets uint64[count + 1], triples int32[entries, 3] or None when to_host is False, kernel ms)"""
        count = self.n - first if count is None else count
        off = np.zeros(count + 1, dtype=np.uint64)
        ms = C.c_double()
        lib = self.al.lib
        if not to_host:      # the profile stays in HBM (a NULL buffer), for a consumer on the device
            capi.check(lib.cvx_job_nm_profile(self.al.h, self.j, first, count, off.ctypes.data, None, 0, C.byref(ms)))
            return off, None, ms.value
        # sizes first (the offsets alone: no profile kernel, no arena), then the one real call
        capi.check(lib.cvx_job_nm_sizes(self.al.h, self.j, first, count, off.ctypes.data))
        tri = np.zeros((int(off[count]), 3), dtype=np.int32)
        capi.check(lib.cvx_job_nm_profile(self.al.h, self.j, first, count, off.ctypes.data, tri.ctypes.data, int(off[count]), C.byref(ms)))
        return off, tri, ms.value

    def text_all(self, ext_qstart=None, ext_qend=None):
        """cvx_job_text_all (ABI 9): text() and nm_profile_resident() of the whole job in one call.
        -> (list of dicts as text(), entry offsets uint64[n + 1], triples int32[entries, 3])"""
        n = self.n
        out = (capi.CvxAlignmentText * max(n, 1))()
        off = np.zeros(max(n, 1), dtype=np.uint64)
        nmoff = np.zeros(n + 1, dtype=np.uint64)
        nbytes = C.c_uint64()
        eq = np.ascontiguousarray(ext_qstart, dtype=np.int32) if ext_qstart is not None else None
        ee = np.ascontiguousarray(ext_qend, dtype=np.int32) if ext_qend is not None else None
        lib = self.al.lib
        lib.cvx_job_text_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(capi.CvxAlignmentText),
                                         C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_void_p)]
        tp, np_ = C.c_void_p(), C.c_void_p()
        capi.check(lib.cvx_job_text_all(self.al.h, self.j, eq.ctypes.data if eq is not None else None,
                                        ee.ctypes.data if ee is not None else None, out, off.ctypes.data,
                                        C.byref(tp), C.byref(nbytes), nmoff.ctypes.data, C.byref(np_)))
        raw = (C.c_char * int(nbytes.value)).from_address(tp.value).raw if nbytes.value else b""
        res = []
        for i in range(n):
            t = out[i]
            d = {f: getattr(t, f) for f, _ in capi.CvxAlignmentText._fields_}
            d["score_bits"] = int(np.float32(t.score).view(np.uint32))
            o = int(off[i])
            d["cigar"] = raw[o:o + t.cigar_len].decode()
            d["md"] = raw[o + t.cigar_len + 1:o + t.cigar_len + 1 + t.md_len].decode()
            res.append(d)
        e = int(nmoff[n])
        tri = np.ctypeslib.as_array(C.cast(np_, C.POINTER(C.c_int32)), shape=(e * 3,)).reshape(e, 3).copy() if e else np.zeros((0, 3), dtype=np.int32)
        return res, nmoff, tri

    def nm_profile_resident(self, first: int = 0, count: Optional[int] = None):
        """cvx_job_nm_profile_resident (ABI 9): the same triples left in the job's page-locked memory (copied out here).
        -> (entry offsets uint64[count + 1], triples int32[entries, 3], kernel ms)"""
        count = self.n - first if count is None else count
        off = np.zeros(count + 1, dtype=np.uint64)
        ms = C.c_double()
        ptr = C.c_void_p()
        capi.check(self.al.lib.cvx_job_nm_profile_resident(self.al.h, self.j, first, count, off.ctypes.data, C.byref(ptr), C.byref(ms)))
        n = int(off[count])
        tri = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n * 3,)).reshape(n, 3).copy() if n else np.zeros((0, 3), dtype=np.int32)
        return off, tri, ms.value

    def release(self) -> None:
        if self.j:
            self.al.lib.cvx_job_release(self.al.h, self.j)
            self.j = None
            self.results = self.ops = None


def format_tileset(lib, tileset, idx, results, ops, n_threads: int = 0):
    """Host text stage (cvx_format_batch, all host threads) for tiles `idx` of a TileSet whose
    result records / ops arena are `results` / `ops` (numpy, RESULT_DTYPE).  Returns a list of
    dicts with the text-level Align fields (ret, score bits, CIGAR, MD, ...), no NM profile."""
    idx = np.asarray(idx, dtype=np.int64)
    n = len(idx)
    tab = np.ascontiguousarray(tileset.table()[idx])
    res = np.ascontiguousarray(results[idx])
    H = tileset.H[idx]
    caps = (4 * H + 64).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(caps)])
    cig = np.zeros(int(offs[-1]), dtype=np.uint8)
    md = np.zeros(int(offs[-1]), dtype=np.uint8)
    bufs = (capi.CvxTextBuffers * max(n, 1))()
    for k in range(n):
        bufs[k].cigar = cig.ctypes.data + int(offs[k])
        bufs[k].md = md.ctypes.data + int(offs[k])
        bufs[k].nm_triples = None
        bufs[k].cigar_cap = int(caps[k])
        bufs[k].md_cap = int(caps[k])
        bufs[k].nm_cap = 0
    out = (capi.CvxAlignmentText * max(n, 1))()
    capi.check(lib.cvx_format_batch(n, res.ctypes.data_as(C.POINTER(capi.CvxResult)), ops.ctypes.data,
                                    tab.ctypes.data_as(C.POINTER(capi.CvxTile)), bufs, out, n_threads))
    texts = []
    for k in range(n):
        t = out[k]
        d = {f: getattr(t, f) for f, _ in capi.CvxAlignmentText._fields_}
        d["score_bits"] = int(np.float32(t.score).view(np.uint32))
        o = int(offs[k])
        d["cigar"] = cig[o:o + t.cigar_len].tobytes().decode() if t.cigar_len < caps[k] else None
        d["md"] = md[o:o + t.md_len].tobytes().decode() if t.md_len < caps[k] else None
        texts.append(d)
    return texts


def encode_genome(lib, seqs: Sequence[bytes]):
    """cvx_genome_encode: ngmlr's 4-bit reference encoding (host only).  -> (binref uint8[], nibbles, start table uint64[])"""
    n = len(seqs)
    lens = np.array([len(x) for x in seqs], dtype=np.uint64)
    arr = (C.c_char_p * max(n, 1))(*seqs)
    binref = np.zeros(int(lib.cvx_genome_encoded_bytes(n, lens.ctypes.data)), dtype=np.uint8)
    starts = np.zeros(n + 1, dtype=np.uint64)
    nib = C.c_uint64()
    ns = C.c_int32()
    capi.check(lib.cvx_genome_encode(n, arr, lens.ctypes.data, binref.ctypes.data, C.byref(nib), starts.ctypes.data, C.byref(ns)))
    return binref, int(nib.value), starts[:ns.value].copy()


class Genome:
    """An encoded reference genome resident in HBM (decode half): windows are decoded on the
    device -- the counterpart of SequenceProvider.DecodeRefSequenceExact (reference src/SequenceProvider.cpp:493-565)."""

    def __init__(self, aligner: ConvexAlignHip, binref: np.ndarray, nibbles: int, starts: np.ndarray):
        self.al = aligner
        self.g = C.c_void_p()
        b = np.ascontiguousarray(binref, dtype=np.uint8)
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        capi.check(aligner.lib.cvx_genome_upload(aligner.h, b.ctypes.data, int(nibbles), st.ctypes.data, len(st), C.byref(self.g)))

    def decode(self, positions, lengths) -> List[bytes]:
        """DecodeRefSequenceExact(position, length, corridor 0) for every window, `length` bytes each (the last a NUL)."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(ln.astype(np.int64))]).astype(np.uint64)
        out = np.zeros(int(off[-1]) + 8, dtype=np.uint8)
        capi.check(self.al.lib.cvx_genome_decode(self.al.h, self.g, len(pos), pos.ctypes.data, ln.ctypes.data, off.ctypes.data, out.ctypes.data))
        return [out[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(pos))]

    def submit(self, tiles: Sequence, ref_positions) -> "Job":
        """cvx_submit_windows: the tiles' references are windows of this genome (tile.ref is not uploaded)."""
        arr, keep = self.al._pack(tiles)
        pos = np.ascontiguousarray(ref_positions, dtype=np.uint64)
        j = C.c_void_p()
        capi.check(self.al.lib.cvx_submit_windows(self.al.h, self.g, len(tiles), arr, pos.ctypes.data, C.byref(j)))
        return Job(self.al, j, len(tiles), (keep, pos))

    def free(self) -> None:
        if self.g:
            self.al.lib.cvx_genome_free(self.al.h, self.g)
            self.g = None


CANDIDATE_DTYPE = np.dtype([("location", np.uint64), ("score", np.float32), ("reverse", np.int32)])


class KmerIndex:
    """One unit of ngmlr's CompactPrefixTable resident in HBM, and CS::RunRead's candidate search over it for batches of
    (sub-)reads (search half; reference src/CS.cpp:57-149, 219-268, 324-398)."""

    def __init__(self, aligner: ConvexAlignHip, kmer_len: int, index_records: np.ndarray, locations: np.ndarray, unit_offset: int = 0):
        self.al = aligner
        self.ix = C.c_void_p()
        idx = np.ascontiguousarray(index_records)
        assert idx.dtype.itemsize == 5 and len(idx) == (1 << (2 * kmer_len)) + 2, "Index[4^k + 2], 5 packed bytes each"
        loc = np.ascontiguousarray(locations, dtype=np.uint32)
        capi.check(aligner.lib.cvx_index_upload(aligner.h, kmer_len, idx.ctypes.data, loc.ctypes.data, len(loc), unit_offset, C.byref(self.ix)))

    def search(self, reads: Sequence[bytes], sensitivity: float = 0.8, min_kmer_hits: float = 0.0, bin_shift: int = 4,
               first_bits: int = 0, extras: bool = False):
        """-> list (one per read) of CANDIDATE_DTYPE arrays in the reference's list order; None where the reference gives up.
        extras: also (max_hit float32[n], kmer_misses int32[n]) -- MappedRead::s and kCount of CS::RunRead."""
        n = len(reads)
        max_hit = np.zeros(max(n, 1), dtype=np.float32)
        misses = np.zeros(max(n, 1), dtype=np.int32)
        arr = (C.c_char_p * max(n, 1))(*reads)
        lens = np.array([len(r) for r in reads], dtype=np.int32)
        ncand = np.zeros(max(n, 1), dtype=np.int32)
        begin = np.zeros(max(n, 1), dtype=np.uint64)
        used = C.c_uint64()
        cap = 1 << 16
        while True:
            cands = np.zeros(cap, dtype=CANDIDATE_DTYPE)
            rc = self.al.lib.cvx_search_batch_ex(self.al.h, self.ix, n, arr, lens.ctypes.data, sensitivity, min_kmer_hits, bin_shift, first_bits,
                                                ncand.ctypes.data, begin.ctypes.data, cands.ctypes.data, cap, C.byref(used),
                                                max_hit.ctypes.data, misses.ctypes.data)
            if rc == -6 and used.value > cap:
                cap = int(used.value) + 64
                continue
            capi.check(rc)
            break
        lists = [None if ncand[i] < 0 else cands[int(begin[i]):int(begin[i]) + int(ncand[i])].copy() for i in range(n)]
        return (lists, max_hit[:n], misses[:n]) if extras else lists

    @staticmethod
    def make_arena(reads: Sequence[bytes], lib=None):
        """The reads back to back with a NUL behind each (cvx_search_batch_arena's input form) -> (arena uint8[], offsets uint64[n + 1],
        pinned handle or None).  lib: put the block into page-locked memory from cvx_host_alloc (the device then pulls it as it is)."""
        lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
        offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
        np.cumsum(lens + 1, out=offsets[1:])
        total = int(offsets[-1])
        blob = b"\0".join(reads) + b"\0" if len(reads) else b""
        pinned = None
        if lib is not None and total:
            p = C.c_void_p()
            if lib.cvx_host_alloc(total + 64, C.byref(p)) == 0:
                arena = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(total + 64,))
                arena[:total] = np.frombuffer(blob, dtype=np.uint8)
                arena[total:] = 0
                pinned = (lib, p)
                return arena, offsets, pinned
        arena = np.frombuffer(blob + b"\0" * 64, dtype=np.uint8).copy()
        return arena, offsets, pinned

    def search_arena(self, arena: np.ndarray, offsets: np.ndarray, sensitivity: float = 0.8, min_kmer_hits: float = 0.0, bin_shift: int = 4,
                     first_bits: int = 0, cands: Optional[np.ndarray] = None):
        """cvx_search_batch_arena: reads that lie back to back -> flat outputs (n_cand int32[n], begin uint64[n], cands CANDIDATE_DTYPE[used],
        max_hit float32[n], kmer_misses int32[n]); read i's list = cands[begin[i] : begin[i] + n_cand[i]] (n_cand < 0: the reference gives up)."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        max_hit = np.zeros(max(n, 1), dtype=np.float32)
        misses = np.zeros(max(n, 1), dtype=np.int32)
        ncand = np.zeros(max(n, 1), dtype=np.int32)
        begin = np.zeros(max(n, 1), dtype=np.uint64)
        used = C.c_uint64()
        if cands is None:
            cands = np.zeros(1 << 16, dtype=CANDIDATE_DTYPE)
        while True:
            rc = self.al.lib.cvx_search_batch_arena(self.al.h, self.ix, n, arena.ctypes.data, offsets.ctypes.data, sensitivity, min_kmer_hits, bin_shift, first_bits,
                                                   ncand.ctypes.data, begin.ctypes.data, cands.ctypes.data, len(cands), C.byref(used),
                                                   max_hit.ctypes.data, misses.ctypes.data)
            if rc == -6 and used.value > len(cands):
                cands = np.zeros(int(used.value) + int(used.value) // 8 + 64, dtype=CANDIDATE_DTYPE)
                continue
            capi.check(rc)
            break
        return ncand[:n], begin[:n], cands[:int(used.value)], max_hit[:n], misses[:n]

    def free(self) -> None:
        if self.ix:
            self.al.lib.cvx_index_free(self.al.h, self.ix)
            self.ix = None


class StrippedSWHip:
    """Mirror of the reference's StrippedSW for the scoring calls (src/StrippedSW.cpp:118-203):
    batch_score == BatchScore, single_score == SingleScore; alignment calls are not part of
    this backend.  GetScoreBatchSize() is 1024 in the reference (src/StrippedSW.h:53-55)."""

    def __init__(self, device: int = 0):
        self._al = ConvexAlignHip(device=device)
        self.lib = self._al.lib

    def get_score_batch_size(self) -> int:
        return 1024

    def batch_score(self, refs: Sequence[bytes], qrys: Sequence[bytes]) -> np.ndarray:
        n = len(refs)
        r = (C.c_char_p * max(n, 1))(*refs)
        q = (C.c_char_p * max(n, 1))(*qrys)
        out = np.zeros(max(n, 1), dtype=np.float32)
        capi.check(self.lib.cvx_score_batch(self._al.h, n, r, q, out.ctypes.data))
        return out[:n]

    def single_score(self, ref: bytes, qry: bytes) -> float:
        return float(self.batch_score([ref], [qry])[0])

    def close(self) -> None:
        self._al.close()


class DeviceBatch:
    """Tiles resident in HBM; run() may be repeated (bench) before download."""

    def __init__(self, aligner: ConvexAlignHip, handle, tiles):
        self.al = aligner
        self.b = handle
        self.tiles = tiles
        self.results = None
        self.ops = None

    def run(self) -> capi.CvxTiming:
        capi.check(self.al.lib.cvx_batch_run(self.al.h, self.b))
        t = capi.CvxTiming()
        capi.check(self.al.lib.cvx_batch_timing(self.b, C.byref(t)))
        return t

    def launches(self) -> list:
        t = capi.CvxTiming()
        capi.check(self.al.lib.cvx_batch_timing(self.b, C.byref(t)))
        out = []
        for i in range(t.n_fill_launches):
            li = capi.CvxLaunchInfo()
            capi.check(self.al.lib.cvx_batch_launch_info(self.b, i, C.byref(li)))
            out.append({k: getattr(li, k) for k, _ in capi.CvxLaunchInfo._fields_})
        return out

    def download(self):
        n = len(self.tiles)
        total = C.c_uint64()
        capi.check(self.al.lib.cvx_batch_ops_total(self.b, C.byref(total)))
        res = (capi.CvxResult * max(n, 1))()
        ops = np.zeros(max(int(total.value), 1), dtype=np.uint32)
        used = C.c_uint64()
        capi.check(self.al.lib.cvx_batch_download(self.al.h, self.b, res, ops.ctypes.data,
                                                  len(ops), C.byref(used)))
        self.results = res
        self.ops = ops
        return res, ops

    def alignments(self, want_nm: bool = True) -> List[dict]:
        if self.results is None:
            self.download()
        return [format_alignment(self.al.lib, self.results[i], self.ops, t, want_nm)
                for i, t in enumerate(self.tiles)]

    def format_batch(self, n_threads: int = 0, want_nm: bool = True):
        """Host text stage for the whole batch in one threaded C call (cvx_format_batch).
        Returns (seconds, texts, cigar_buffers, md_buffers)."""
        import time
        if self.results is None:
            self.download()
        n = len(self.tiles)
        arr, keep = self.al._pack(self.tiles)
        bufs = (capi.CvxTextBuffers * max(n, 1))()
        store = []
        for i, t in enumerate(self.tiles):
            cap = 4 * t.H + 64
            cig = C.create_string_buffer(cap)
            md = C.create_string_buffer(cap)
            nm = np.zeros((2 * (t.H + 1) + 16, 3), dtype=np.int32) if want_nm else None
            store.append((cig, md, nm))
            bufs[i].cigar = C.addressof(cig)
            bufs[i].md = C.addressof(md)
            bufs[i].nm_triples = nm.ctypes.data if want_nm else None
            bufs[i].cigar_cap = cap
            bufs[i].md_cap = cap
            bufs[i].nm_cap = len(nm) if want_nm else 0
            bufs[i].ext_qstart = t.ext_qstart
            bufs[i].ext_qend = t.ext_qend
        out = (capi.CvxAlignmentText * max(n, 1))()
        t0 = time.perf_counter()
        capi.check(self.al.lib.cvx_format_batch(n, self.results, self.ops.ctypes.data, arr, bufs, out, n_threads))
        dt = time.perf_counter() - t0
        return dt, out, store

    def free(self) -> None:
        if self.b:
            self.al.lib.cvx_batch_free(self.al.h, self.b)
            self.b = None


def format_alignment(lib, res: capi.CvxResult, ops: np.ndarray, tile, want_nm: bool = True) -> dict:
    """Host text stage (cvx_format_alignment) -> dict with the Align fields."""
    H, W = len(tile.qry), len(tile.ref)
    cap = 4 * H + 64
    txt = capi.CvxAlignmentText()
    for _ in range(3):
        cig = C.create_string_buffer(cap)
        md = C.create_string_buffer(cap)
        nm_cap = 2 * (H + 1) + W + 16
        nm = np.zeros((nm_cap, 3), dtype=np.int32)
        capi.check(lib.cvx_format_alignment(C.byref(res), ops.ctypes.data, tile.ref, W, H,
                                            tile.ext_qstart, tile.ext_qend, cig, cap, md, cap,
                                            nm.ctypes.data if want_nm else None, nm_cap, C.byref(txt)))
        if txt.cigar_len < cap and txt.md_len < cap:
            break
        cap = max(txt.cigar_len, txt.md_len) + 64
    d = {k: getattr(txt, k) for k, _ in capi.CvxAlignmentText._fields_}
    d["score_bits"] = int(np.float32(txt.score).view(np.uint32))
    d["cigar"] = cig.value.decode()
    d["md"] = md.value.decode()
    # What the reference's consumer reads: detectMisalignment walks nmPerPosition[i] for
    # i < alignmentLength (src/AlignmentBuffer.cpp:1320-1321), although only txt.nm_count triples were
    # written (positions > 16, none for insertion columns); the rest of the caller's buffer is zero here.
    # The buffer it walks is the caller's: (read length + 1) * 2 entries (src/AlignmentBuffer.cpp:277), doubled by addPosition
    # while the written triples do not fit (src/ConvexAlignFast.cpp:79-92; ConvexAlignHip::Finish grows it the same way) -- an
    # alignment with more columns than that (long deletions under a scoring with cheap gaps) is read up to the buffer's end.
    align_cap = 2 * (H + 1)
    while align_cap < txt.nm_count:
        align_cap *= 2
    n = min(txt.alignment_length, align_cap, nm_cap) if (txt.ret >= 0 and want_nm) else 0
    d["nm_per_position"] = nm[:n].copy()
    d["nm_count"] = txt.nm_count      # triples actually written
    d["status"] = res.status
    d["fwd_score_bits"] = int(np.float32(res.score).view(np.uint32))
    d["best_x"] = res.best_ref_index
    d["best_y"] = res.best_read_index
    d["rc"] = 0
    return d
