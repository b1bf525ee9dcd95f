"""ctypes loader for the CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package ``ngmlr_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libcvx_oracle_port.so")
REF_SO = os.path.join(HERE, "_ref", "libcvx_oracle_ref.so")

DEFAULT_PARAMS = (2.0, -5.0, -5.0, -5.0, -1.0, 0.15)  # src/IConfig.h:50-55


class OracleOut(C.Structure):
    _fields_ = [("ret", C.c_int32), ("score", C.c_float), ("position_offset", C.c_int32),
                ("qstart", C.c_int32), ("qend", C.c_int32), ("nm", C.c_int32),
                ("identity", C.c_float), ("alignment_length", C.c_int32),
                ("cigar_op_count", C.c_int32), ("sv_type", C.c_int32),
                ("first_ref", C.c_int32), ("first_read", C.c_int32),
                ("last_ref", C.c_int32), ("last_read", C.c_int32),
                ("nm_count", C.c_int32), ("cigar_len", C.c_int32), ("md_len", C.c_int32)]


def build(which: str = "all") -> None:
    """Compile the checkers (gcc/g++ only).  `ref` is skipped when /root/reference is absent."""
    subprocess.run(["make", "-s", "-C", HERE, which], check=True)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class Oracle:
    """One aligner instance of either checker.  kind: 'port' | 'reference'."""

    def __init__(self, kind: str = "port", params=DEFAULT_PARAMS):
        path = PORT_SO if kind == "port" else REF_SO
        if not os.path.exists(path):
            if kind == "port":
                build("port")
            else:
                raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.oracle_create.restype = C.c_void_p
        self.lib.oracle_create.argtypes = [C.POINTER(C.c_float)]
        self.lib.oracle_destroy.argtypes = [C.c_void_p]
        self.lib.oracle_kind.restype = C.c_char_p
        self.lib.oracle_align.restype = C.c_int
        self.lib.oracle_align.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_int32, C.c_int32, C.POINTER(OracleOut),
                                          C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32]
        self.kind = self.lib.oracle_kind().decode()
        assert self.kind == kind, (self.kind, kind)
        p = (C.c_float * 6)(*params)
        self.h = C.c_void_p(self.lib.oracle_create(p))
        if kind == "port":
            self.lib.oracle_port_set_spec_fill.argtypes = [C.c_void_p, C.c_int]
            self.lib.oracle_port_last_fwd.argtypes = [C.c_void_p]
            self.lib.oracle_port_last_ops.argtypes = [C.c_void_p, C.c_int32]
            self.lib.oracle_port_last_ops.restype = C.c_int

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_spec_fill(self, on: bool) -> None:
        self.lib.oracle_port_set_spec_fill(self.h, int(on))

    def last_fwd(self):
        a = (C.c_int32 * 5)()
        self.lib.oracle_port_last_fwd(a)
        return dict(best_x=a[0], best_y=a[1], ref_position=a[2], qstart=a[3], qend=a[4])

    def last_fill_score_bits(self) -> int:
        """bits of the forward fill's curr_max of the last align() (port only) -- also for tiles validPath rejects"""
        self.lib.oracle_port_last_fill_score.restype = C.c_float
        return int(np.float32(self.lib.oracle_port_last_fill_score()).view(np.uint32))

    def last_ops(self) -> np.ndarray:
        n = self.lib.oracle_port_last_ops(None, 0)
        a = np.zeros(max(n, 1), dtype=np.int32)
        self.lib.oracle_port_last_ops(a.ctypes.data, n)
        return a[:n].astype(np.uint32)

    def align(self, tile, want_nm: bool = True) -> dict:
        H = len(tile.qry)
        off = np.ascontiguousarray(tile.row_offset, dtype=np.int32)
        ln = np.ascontiguousarray(tile.row_length, dtype=np.int32)
        assert len(off) == H and len(ln) == H
        cap = 4 * H + 4 * len(tile.ref) + 256
        cig = C.create_string_buffer(cap)
        md = C.create_string_buffer(cap)
        nm_cap = 2 * (H + 1) + len(tile.ref) + 16
        nm = np.zeros((nm_cap, 3), dtype=np.int32)
        out = OracleOut()
        rc = self.lib.oracle_align(self.h, tile.ref, tile.qry, off.ctypes.data, ln.ctypes.data, H,
                                   tile.ext_qstart, tile.ext_qend, C.byref(out), cig, md, cap,
                                   nm.ctypes.data if want_nm else None, nm_cap)
        d = {k: getattr(out, k) for k, _ in OracleOut._fields_}
        d["rc"] = rc
        d["score_bits"] = int(np.float32(out.score).view(np.uint32))
        d["cigar"] = cig.value.decode()
        d["md"] = md.value.decode()
        d["nm_per_position"] = nm[:out.nm_count].copy()
        return d


COMPARE_KEYS = ("ret", "score_bits", "position_offset", "qstart", "qend", "nm", "alignment_length",
                "cigar_op_count", "sv_type", "first_ref", "first_read", "last_ref", "last_read",
                "cigar", "md")


def same_alignment(a: dict, b: dict, keys=COMPARE_KEYS) -> Optional[str]:
    """None if equal on every field the reference's caller can observe, else the first difference."""
    if a["ret"] < 0 and b["ret"] < 0:
        return None  # both invalid: the caller only sees -1 / Score -1
    for k in keys:
        if a[k] != b[k]:
            return "%s: %r != %r" % (k, a[k] if len(str(a[k])) < 80 else str(a[k])[:80], b[k] if len(str(b[k])) < 80 else str(b[k])[:80])
    if np.float32(a["identity"]).view(np.uint32) != np.float32(b["identity"]).view(np.uint32):
        return "identity"
    if a["nm_per_position"].shape != b["nm_per_position"].shape or not np.array_equal(a["nm_per_position"], b["nm_per_position"]):
        return "nm_per_position"
    return None


# ----------------------------------------------------------------- scoring path
SCORE_PORT_SO = os.path.join(HERE, "libscore_oracle_port.so")
SCORE_REF_SO = os.path.join(HERE, "_ref", "libscore_oracle_ref.so")


def have_score_ref() -> bool:
    return os.path.exists(SCORE_REF_SO)


class ScoreOracle:
    """StrippedSW::BatchScore checker.  kind: 'port' | 'reference'."""

    def __init__(self, kind: str = "port"):
        path = SCORE_PORT_SO if kind == "port" else SCORE_REF_SO
        if not os.path.exists(path):
            if kind == "port":
                build("port")
            else:
                raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.score_oracle_create.restype = C.c_void_p
        self.lib.score_oracle_destroy.argtypes = [C.c_void_p]
        self.lib.score_oracle_kind.restype = C.c_char_p
        self.lib.score_oracle_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p]
        assert self.lib.score_oracle_kind().decode() == kind
        self.h = C.c_void_p(self.lib.score_oracle_create())

    def scores(self, refs, qrys) -> np.ndarray:
        n = len(refs)
        r = (C.c_char_p * n)(*refs)
        q = (C.c_char_p * n)(*qrys)
        out = np.zeros(n, dtype=np.float32)
        self.lib.score_oracle_batch(self.h, n, r, q, out.ctypes.data)
        return out


DECODE_SO = os.path.join(HERE, "libdecode_oracle_port.so")


class DecodeOracle:
    """CPU restatement of ngmlr's genome encoding and DecodeRefSequenceExact (oracle/decode_oracle.c)."""

    def __init__(self):
        if not os.path.exists(DECODE_SO):
            build("port")
        self.lib = C.CDLL(DECODE_SO)
        self.lib.decode_oracle_encoded_bytes.restype = C.c_uint64
        self.lib.decode_oracle_encoded_bytes.argtypes = [C.c_int, C.c_void_p]
        self.lib.decode_oracle_encode.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.decode_oracle_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_void_p]

    def encode(self, seqs):
        """-> (binref uint8[], nibbles, starts uint64[kept + 1])"""
        n = len(seqs)
        lens = np.array([len(x) for x in seqs], dtype=np.uint64)
        arr = (C.c_char_p * max(n, 1))(*seqs)
        nb = int(self.lib.decode_oracle_encoded_bytes(n, lens.ctypes.data))
        binref = np.zeros(nb, dtype=np.uint8)
        starts = np.zeros(n + 1, dtype=np.uint64)
        nib = C.c_uint64()
        kept = self.lib.decode_oracle_encode(n, arr, lens.ctypes.data, binref.ctypes.data, C.byref(nib), starts.ctypes.data)
        return binref, int(nib.value), starts[:kept + 1].copy()

    def window(self, binref, starts, pos: int, length: int) -> bytes:
        out = np.zeros(length + 8, dtype=np.uint8)
        b = np.ascontiguousarray(binref, dtype=np.uint8)
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        self.lib.decode_oracle_window(b.ctypes.data, st.ctypes.data, len(st), int(pos), int(length), out.ctypes.data)
        return out[:length].tobytes()


# --------------------------------------------------------------------------- candidate search (search half)

CS_SO = os.path.join(HERE, "libcs_oracle_port.so")


class SearchFixture:
    """A recorded k-mer table + candidate-search calls (tools/make_golden_cs.sh -> tests/golden/cs_test_3.npz)."""

    def __init__(self, path):
        z = np.load(path)
        self.k = int(z["k"])
        self.unit_offset = int(z["unit_offset"])
        self.prefix, self.cnt, self.rc, self.locs = z["prefix"], z["cnt"], z["rc"], z["locs"]
        off = np.concatenate([[0], np.cumsum(z["seq_len"].astype(np.int64))])
        raw = z["seqs"].tobytes()
        self.seqs = [raw[int(off[i]):int(off[i + 1])] for i in range(len(z["seq_len"]))]
        so = np.concatenate([[0], np.cumsum(z["n_scores"].astype(np.int64))])
        self.want = [(z["loc"][int(so[i]):int(so[i + 1])], z["score"][int(so[i]):int(so[i + 1])], z["rev"][int(so[i]):int(so[i + 1])])
                     for i in range(len(self.seqs))]
        self.max_hit, self.thresh, self.rlist_len = z["max_hit"], z["thresh"], z["rlist_len"]
        # kCount as the reference left it (summed over the attempts of the retry ladder) and the table size its first attempt ran with
        self.kmer_misses, self.first_bits = z["kmer_misses"], z["first_bits"]

    def index_arrays(self):
        """The table as ngmlr holds it (src/PrefixTable.h:17-31 Index, packed 5 bytes: uint m_TabIndex, char m_RevCompIndex;
        src/IRefProvider.h:10-16 Location): (index bytes [(4^k + 2) * 5], locations uint32[])."""
        n = (1 << (2 * self.k)) + 2
        cnt_full = np.zeros(n, dtype=np.int64)
        cnt_full[self.prefix] = self.cnt
        tab = (1 + np.concatenate([[0], np.cumsum(cnt_full)[:-1]])).astype(np.uint32)
        rc = np.zeros(n, dtype=np.int8)
        rc[self.prefix] = self.rc
        idx = np.zeros(n, dtype=np.dtype([("tab", "<u4"), ("rc", "i1")]))
        idx["tab"], idx["rc"] = tab, rc
        assert idx.dtype.itemsize == 5
        return idx, np.ascontiguousarray(self.locs, dtype=np.uint32)


class SearchOracle:
    """cs_oracle.c: one CS::RunRead (reference src/CS.cpp:324-398) per call."""

    def __init__(self, fx: "SearchFixture" = None, raw=None):
        """fx: a recorded fixture (compact table form); raw = (k, unit_offset, index bytes [(4^k + 2) * 5], locations uint32[]): the
        table as ngmlr holds it."""
        if not os.path.exists(CS_SO):
            build("port")
        self.lib = C.CDLL(CS_SO)
        self.lib.cs_table_create.restype = C.c_void_p
        self.lib.cs_table_create.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        self.lib.cs_table_destroy.argtypes = [C.c_void_p]
        self.lib.cs_search.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self.lib.cs_search_ex.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        if raw is not None:
            k, unit_offset, idx5, locs = raw
            self.lib.cs_table_create_raw.restype = C.c_void_p
            self.lib.cs_table_create_raw.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
            i5 = np.ascontiguousarray(idx5).view(np.uint8)
            assert len(i5) == ((1 << (2 * k)) + 2) * 5
            l = np.ascontiguousarray(locs, dtype=np.uint32)
            self.t = self.lib.cs_table_create_raw(k, unit_offset, i5.ctypes.data, l.ctypes.data, len(l))
            return
        p = np.ascontiguousarray(fx.prefix, dtype=np.uint32)
        c = np.ascontiguousarray(fx.cnt, dtype=np.uint32)
        l = np.ascontiguousarray(fx.locs, dtype=np.uint32)
        self.t = self.lib.cs_table_create(fx.k, fx.unit_offset, len(p), p.ctypes.data, c.ctypes.data, l.ctypes.data, len(l))

    def search(self, seq: bytes, sensitivity=0.8, min_hits=0.0, bin_shift=4, cap=4096, first_bits=16):
        loc = np.zeros(cap, dtype=np.uint64)
        sc = np.zeros(cap, dtype=np.float32)
        rev = np.zeros(cap, dtype=np.int32)
        mh, th, rl, tb, km = C.c_float(), C.c_float(), C.c_int32(), C.c_int32(), C.c_int32()
        n = self.lib.cs_search_ex(self.t, seq, len(seq), sensitivity, min_hits, bin_shift, first_bits, loc.ctypes.data, sc.ctypes.data, rev.ctypes.data,
                                  cap, C.byref(mh), C.byref(th), C.byref(rl), C.byref(tb), C.byref(km))
        m = max(0, min(n, cap))
        return {"n": n, "loc": loc[:m].copy(), "score": sc[:m].copy(), "rev": rev[:m].copy(), "max_hit": mh.value if n >= 0 else 0.0, "thresh": th.value,
                "rlist_len": rl.value, "table_bits": tb.value, "kmer_misses": km.value}

    def close(self):
        if self.t:
            self.lib.cs_table_destroy(self.t)
            self.t = None
