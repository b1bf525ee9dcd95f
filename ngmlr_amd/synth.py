"""Seeded synthetic tile generator for the convex-gap alignment hot path.

A *tile* is exactly one ``IAlignment::SingleAlign`` call of the reference
(``/root/reference/src/IAlignment.h:227-232``): a NUL-free reference window
``ref`` (W chars), a read segment ``qry`` (H chars) and one ``CorridorLine``
(offset, length) per read row.  The corridor constructors below restate the
host-side formulas of the reference's only caller so that synthetic tiles have the
shapes the real pipeline produces (float32 arithmetic with C truncation, done with
numpy float32 so the integers come out identical):

* ``corridor_anchors``   -- getCorridorEndpointsWithAnchors, src/AlignmentBuffer.cpp:129-197
* ``corridor_endpoints`` -- getCorridorEndpoints,            src/AlignmentBuffer.cpp:107-127
* ``corridor_linear``    -- getCorridorLinear,               src/AlignmentBuffer.cpp:68-82
* ``corridor_full``      -- getCorridorFull,                 src/AlignmentBuffer.cpp:84-105
* ``estimate_corridor``  -- estimateCorridor + clamp,        src/AlignmentBuffer.cpp:1454-1467,265-266

GRCh38 and pbsim are not available offline, so reads are drawn from a seeded uniform
ACGT reference with a PacBio-like (ins:del:sub = 6:3:1) or ONT-like (4:4:2) error
model.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.zeros(256, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


@dataclass
class Tile:
    ref: bytes
    qry: bytes
    row_offset: np.ndarray  # int32[H]
    row_length: np.ndarray  # int32[H]
    ext_qstart: int = 0
    ext_qend: int = 0
    tag: str = ""
    # closed form of the corridor when the builder that made the rows is known (cvx_tile.corridor_kind):
    # (kind, k, d, right, offset, width) with kind 1 = affine, 2 = constant; None = only the arrays
    desc: Optional[tuple] = None

    @property
    def H(self) -> int:
        return len(self.qry)

    @property
    def W(self) -> int:
        return len(self.ref)

    @property
    def cells(self) -> int:
        """Sum of corridor row lengths = bytes of the reference's directionMatrix."""
        return int(self.row_length.astype(np.int64).sum())

    def algorithmic_bytes(self) -> int:
        """B ~= C + 6H + 2W (direction byte per cell + sequences,
        row offsets, backtrack reads, ops)."""
        return self.cells + 6 * self.H + 2 * self.W


# --------------------------------------------------------------------------- corridors

def _trunc_i32(a: np.ndarray) -> np.ndarray:
    return np.trunc(a).astype(np.int64).astype(np.int32)


def affine_rows(H: int, k, d, right, width: int) -> Tuple[np.ndarray, np.ndarray]:
    """offset[y] = (int) (((float) y - d) / k - right) in binary32 with C truncation: the common closed form of the
    reference's corridor builders (cvx_tile.corridor_kind = CVX_CORRIDOR_AFFINE; the device evaluates the same)."""
    i = np.arange(H, dtype=np.int64).astype(F32)
    off = _trunc_i32(F32(F32(F32(i - F32(d)) / F32(k)) - F32(right)))
    return off, np.full(H, width, dtype=np.int32)


def anchors_desc(H: int, W: int, mult: int = 1, scatter_left: float = 0.0, scatter_right: float = 0.0) -> tuple:
    """src/AlignmentBuffer.cpp:140-192 with the anchor scatter given directly
    (max positive / negative deviation of anchors from the k-line) -> closed form (kind, k, d, right, offset, width)."""
    k = F32(H) * F32(1.0) / F32(W)
    left = F32(scatter_left)
    right = F32(scatter_right)
    left = F32(left + F32(128))
    right = F32(right + F32(128))
    left = F32(left + F32(F32(left + right) * F32(0.1)))
    right = F32(right + F32(F32(left + right) * F32(0.1)))
    left = F32(left * F32(mult))
    right = F32(right * F32(mult))
    width = int(np.trunc(F32(left + right)))
    return (1, float(k), 0.0, float(right), 0, width)


def corridor_anchors(H: int, W: int, mult: int = 1, scatter_left: float = 0.0,
                     scatter_right: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    _, k, d, right, _, width = anchors_desc(H, W, mult, scatter_left, scatter_right)
    return affine_rows(H, k, d, right, width)


def endpoints_desc(H: int, W: int, corridor: int, realign: bool = False) -> tuple:
    """src/AlignmentBuffer.cpp:110-124; ``corridor`` is corridor*corridorMultiplier."""
    width = corridor // (1 if realign else 4)
    k = F32(H) * F32(1.0) / F32(W)
    d = F32(width) / F32(2.0)
    return (1, float(k), float(d), 0.0, 0, width)


def corridor_endpoints(H: int, W: int, corridor: int, realign: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    _, k, d, right, _, width = endpoints_desc(H, W, corridor, realign)
    return affine_rows(H, k, d, right, width)


def linear_desc(H: int, width: int) -> tuple:
    """src/AlignmentBuffer.cpp:77-80 (short reads): offset = i - width / 2 (integer)."""
    return (1, 1.0, float(width // 2), 0.0, 0, width)


def corridor_linear(H: int, width: int) -> Tuple[np.ndarray, np.ndarray]:
    off = (np.arange(H, dtype=np.int64) - width // 2).astype(np.int32)
    return off, np.full(H, width, dtype=np.int32)


def full_desc(H: int, W: int) -> tuple:
    """src/AlignmentBuffer.cpp:93-103: every row spans the whole window plus 20%."""
    return (2, 0.0, 0.0, 0.0, int(W * -0.2), W + int(W * 0.2))


def corridor_full(H: int, W: int) -> Tuple[np.ndarray, np.ndarray]:
    _, _, _, _, off, length = full_desc(H, W)
    return np.full(H, off, dtype=np.int32), np.full(H, length, dtype=np.int32)


def estimate_corridor(H: int, Wspan: int, refSeqLen: int) -> int:
    """src/AlignmentBuffer.cpp:1454-1467 then the 2*refSeqLen clamp of :265-266."""
    c = max(int(F32(abs(H - Wspan)) * F32(2.1)), int(F32(H) * F32(0.2)))
    c = min(8192, c)
    return min(c, refSeqLen * 2)


# --------------------------------------------------------------------------- sequences

def random_ref(rng: np.random.Generator, n: int, n_frac: float = 0.0, x_frac: float = 0.0) -> np.ndarray:
    """Uniform ACGT with optional 'N' / 'x' (the two extra symbols the reference's
    decoder emits, src/SequenceProvider.cpp:91-105, :501)."""
    s = _ACGT[rng.integers(0, 4, size=n)]
    if n_frac > 0:
        s = np.where(rng.random(n) < n_frac, np.uint8(ord("N")), s)
    if x_frac > 0:
        s = np.where(rng.random(n) < x_frac, np.uint8(ord("x")), s)
    return s.astype(np.uint8)


def mutate(rng: np.random.Generator, ref: np.ndarray, err: float,
           ratio: Sequence[float] = (6, 3, 1), n_frac: float = 0.0) -> np.ndarray:
    """Read = ref with `err` errors per base split ins:del:sub = ratio."""
    tot = float(sum(ratio))
    p_ins, p_del, p_sub = (err * r / tot for r in ratio)
    n = len(ref)
    u = rng.random(n)
    keep = u >= p_del                      # deletions drop the ref base
    sub = (u >= p_del) & (u < p_del + p_sub)
    base = ref.copy()
    shift = rng.integers(1, 4, size=n)
    is_acgt = np.isin(base, _ACGT)
    # substitution: rotate within ACGT so the base always changes
    code = _CODE[base].astype(np.int64)
    subbed = _ACGT[(code + shift) % 4]
    base = np.where(sub & is_acgt, subbed, base)
    n_ins = rng.random(n) < p_ins
    out_len = int(keep.sum() + n_ins.sum())
    out = np.empty(out_len, dtype=np.uint8)
    # positions: each kept base emits itself, each insertion flag emits one random base before it
    emit_counts = keep.astype(np.int64) + n_ins.astype(np.int64)
    starts = np.cumsum(emit_counts) - emit_counts
    ins_pos = starts[n_ins]
    out[ins_pos] = _ACGT[rng.integers(0, 4, size=len(ins_pos))]
    base_pos = (starts + n_ins.astype(np.int64))[keep]
    out[base_pos] = base[keep]
    if n_frac > 0:
        out = np.where(rng.random(out_len) < n_frac, np.uint8(ord("N")), out)
    return out


def revcomp(s: np.ndarray) -> np.ndarray:
    comp = np.arange(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    return comp[s[::-1]]


# --------------------------------------------------------------------------- tiles

def make_tile(rng: np.random.Generator, W: int, err: float = 0.15,
              ratio: Sequence[float] = (6, 3, 1), corridor: str = "anchors", mult: int = 1,
              scatter: float = 0.0, width: Optional[int] = None, n_frac: float = 0.0,
              x_frac: float = 0.0, realign: bool = False, ref_pad: int = 0,
              tag: str = "") -> Tile:
    """One tile: reference window of W bases (the read covers all but `ref_pad`
    bases on each side), read = mutated copy, corridor by name."""
    ref = random_ref(rng, W, n_frac=n_frac, x_frac=x_frac)
    qry = mutate(rng, ref[ref_pad:W - ref_pad] if ref_pad else ref, err, ratio)
    if len(qry) == 0:
        qry = ref[:1].copy()
    H = len(qry)
    if corridor == "anchors":
        sl = float(rng.random() * scatter)
        sr = float(rng.random() * scatter)
        desc = anchors_desc(H, W, mult=mult, scatter_left=sl, scatter_right=sr)
        off, ln = affine_rows(H, desc[1], desc[2], desc[3], desc[5])
    elif corridor == "endpoints":
        c = width if width is not None else estimate_corridor(H, W, W)
        desc = endpoints_desc(H, W, c * mult, realign=realign)
        off, ln = affine_rows(H, desc[1], desc[2], desc[3], desc[5])
    elif corridor == "linear":
        w = width if width is not None else 256 + 2 * int(F32(0.15) * F32(H))
        desc = linear_desc(H, w * mult)
        off, ln = corridor_linear(H, w * mult)
    elif corridor == "full":
        desc = full_desc(H, W)
        off, ln = corridor_full(H, W)
    else:
        raise ValueError(corridor)
    return Tile(ref=ref.tobytes(), qry=qry.tobytes(), row_offset=off, row_length=ln, tag=tag or corridor, desc=desc)


def workload_pacbio(n_tiles: int, seed: int = 7, read_len: int = 10000, err: float = 0.15,
                    scatter: float = 25.0) -> List[Tile]:
    """Config C2: PacBio-like 10 kb reads, 15 % error (6:3:1), anchors corridor.
    With ~25 bp of anchor scatter the width lands at the 309-369 the reference
    shows on such reads."""
    rng = np.random.default_rng(seed)
    tiles = []
    for i in range(n_tiles):
        W = int(read_len * (0.9 + 0.2 * rng.random()))
        tiles.append(make_tile(rng, W, err=err, ratio=(6, 3, 1), corridor="anchors",
                               scatter=scatter, tag="pacbio"))
    return tiles


def workload_ont(n_tiles: int, seed: int = 11, max_len: int = 20000, err: float = 0.25) -> List[Tile]:
    """Config C3: ONT-like reads (4:4:2), tile mix median ~1.3 kb up to 20 kb,
    width 309-463, 10 % of tiles at corridor multiplier 2 (retries)."""
    rng = np.random.default_rng(seed)
    tiles = []
    for i in range(n_tiles):
        W = int(min(max_len, max(200, rng.lognormal(mean=np.log(1300.0), sigma=1.0))))
        mult = 2 if rng.random() < 0.10 else 1
        tiles.append(make_tile(rng, W, err=err, ratio=(4, 4, 2), corridor="anchors",
                               scatter=60.0, mult=mult, tag="ont"))
    return tiles


def workload_ultralong_sv(n_tiles: int, seed: int = 13, read_len: int = 100000) -> List[Tile]:
    """Config C5: 100 kb tiles at widths {309, 2048, 8192} plus full-matrix
    inversion tiles (500-5000 bp)."""
    rng = np.random.default_rng(seed)
    tiles = []
    for i in range(n_tiles):
        kind = i % 4
        if kind == 3:
            W = int(rng.integers(500, 5000))
            tiles.append(make_tile(rng, W, err=0.2, ratio=(4, 4, 2), corridor="full", tag="sv-full"))
        else:
            width = (309, 2048, 8192)[kind]
            if width == 309:
                tiles.append(make_tile(rng, read_len, err=0.2, ratio=(4, 4, 2), corridor="anchors", tag="ul-309"))
            else:
                tiles.append(make_tile(rng, read_len, err=0.2, ratio=(4, 4, 2), corridor="endpoints",
                                       width=width, realign=True, tag="ul-%d" % width))
    return tiles


def workload_ultralong_mix(n_tiles: int, seed: int = 19, read_len: int = 100000, wide_frac: float = 0.05) -> List[Tile]:
    """Config C5 as a THROUGHPUT batch: thousands of 100 kb tiles, 95 % of them on the anchors corridor of the first
    alignment attempt (width 309+), `wide_frac` of them retries on widened corridors (2048 / 8192 columns, the
    endpoints corridor of attempts >= 3, src/AlignmentBuffer.cpp:291-294, 1454-1467) or full-matrix inversion checks."""
    rng = np.random.default_rng(seed)
    tiles = []
    for i in range(n_tiles):
        u = rng.random()
        if u >= wide_frac:
            tiles.append(make_tile(rng, read_len, err=0.2, ratio=(4, 4, 2), corridor="anchors", scatter=40.0, tag="ul-309"))
        elif u < wide_frac * 0.4:
            tiles.append(make_tile(rng, int(rng.integers(500, 5000)), err=0.2, ratio=(4, 4, 2), corridor="full", tag="sv-full"))
        else:
            width = 2048 if u < wide_frac * 0.8 else 8192
            tiles.append(make_tile(rng, read_len, err=0.2, ratio=(4, 4, 2), corridor="endpoints", width=width, realign=True, tag="ul-%d" % width))
    return tiles


def _workload_chunk(args):
    name, n, seed, kw = args
    return globals()["workload_" + name](n, seed=seed, **kw)


def parallel_workload(name: str, n_tiles: int, seed: int, pool=None, chunk: int = 256, **kw) -> List[Tile]:
    """workload_<name>(n_tiles) generated `chunk` tiles at a time (own seed per chunk) on `pool`'s worker processes."""
    tasks = [(name, min(chunk, n_tiles - c0), seed * 1009 + c0, kw) for c0 in range(0, n_tiles, chunk)]
    parts = list(pool.map(_workload_chunk, tasks)) if pool is not None else [_workload_chunk(t) for t in tasks]
    return [t for p in parts for t in p]


def workload_short(n_tiles: int, seed: int = 17) -> List[Tile]:
    """Short reads (<= 256 bp) through getCorridorLinear (src/AlignmentBuffer.cpp:2576-2594)."""
    rng = np.random.default_rng(seed)
    tiles = []
    for i in range(n_tiles):
        L = int(rng.integers(50, 257))
        pad = (256 + 2 * int(F32(0.15) * F32(L))) // 2
        tiles.append(make_tile(rng, L + 2 * pad, err=0.08, corridor="linear", ref_pad=pad, tag="short"))
    return tiles


# --------------------------------------------------------------------------- flat tile sets
# The bench moves tens of thousands of 10 kb tiles per step and per device.  A TileSet holds a whole
# batch in four flat arrays (chunks of it can be generated on worker processes) and hands the C ABI
# a tile table that points straight into those arrays.

TILE_DTYPE = np.dtype([("ref", np.uint64), ("qry", np.uint64), ("row_offset", np.uint64), ("row_length", np.uint64),
                       ("ref_len", np.int32), ("qry_len", np.int32), ("row_stride_bytes", np.int32), ("corridor_kind", np.int32),
                       ("corridor_k", np.float32), ("corridor_d", np.float32), ("corridor_right", np.float32),
                       ("corridor_offset", np.int32), ("corridor_width", np.int32), ("reserved", np.int32)])
assert TILE_DTYPE.itemsize == 72
DESC_DTYPE = np.dtype([("kind", np.int32), ("k", np.float32), ("d", np.float32), ("right", np.float32),
                       ("offset", np.int32), ("width", np.int32)])


class TileSet:
    """n tiles in flat arrays: ref / qry bytes back to back in tile order, one (offset, length) per read row, and
    (optionally) the closed form of every corridor.  `closed_form`: the tile table hands the C ABI the closed forms
    instead of the row arrays.  pin(lib): the sequences move into page-locked memory (cvx_host_alloc), from where the
    device pulls them without any packing on the host."""

    def __init__(self, ref, ref_off, qry, qry_off, row_offset, row_length, tag="", desc=None):
        self.ref, self.ref_off = ref, ref_off
        self.qry, self.qry_off = qry, qry_off
        self.row_offset, self.row_length = row_offset, row_length
        self.tag = tag
        self.desc = desc                      # DESC_DTYPE[n] or None
        self.closed_form = False
        self.n = len(ref_off) - 1
        self.H = np.diff(qry_off).astype(np.int64)
        self.W = np.diff(ref_off).astype(np.int64)
        self._table = None
        self._pinned = None

    def __len__(self) -> int:
        return self.n

    @property
    def read_bases(self) -> int:
        return int(self.H.sum())

    @property
    def cells(self) -> int:
        return int(self.row_length.astype(np.int64).sum())

    def widths(self) -> np.ndarray:
        return self.row_length[self.qry_off[:-1]]

    def tile(self, i: int) -> Tile:
        """Tile i as the per-tile object the oracle wrapper and the tests take (copies)."""
        r0, r1 = int(self.ref_off[i]), int(self.ref_off[i + 1])
        q0, q1 = int(self.qry_off[i]), int(self.qry_off[i + 1])
        d = self.desc[i] if self.desc is not None else None
        return Tile(ref=self.ref[r0:r1].tobytes(), qry=self.qry[q0:q1].tobytes(),
                    row_offset=self.row_offset[q0:q1], row_length=self.row_length[q0:q1], tag=self.tag,
                    desc=None if d is None else (int(d["kind"]), float(d["k"]), float(d["d"]), float(d["right"]), int(d["offset"]), int(d["width"])))

    def use_closed_form(self, on: bool = True) -> "TileSet":
        if on and self.desc is None:
            raise ValueError("this TileSet carries no corridor descriptors")
        self.closed_form, self._table = bool(on), None
        return self

    def pin(self, lib) -> bool:
        """Moves ref / qry into page-locked arenas from cvx_host_alloc (False, nothing changed, when that fails: no device)."""
        import ctypes as C
        if self._pinned is not None:
            return True
        ptrs = []
        for arr in (self.ref, self.qry):
            p = C.c_void_p()
            if lib.cvx_host_alloc(max(int(arr.nbytes), 1), C.byref(p)) != 0:
                for q in ptrs:
                    lib.cvx_host_free(q)
                return False
            ptrs.append(p)
        views = []
        for p, arr in zip(ptrs, (self.ref, self.qry)):
            v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(int(arr.nbytes), 1),))[:arr.nbytes]
            v[:] = arr
            views.append(v)
        self.ref, self.qry = views
        self._pinned = (lib, ptrs)
        self._table = None
        return True

    def unpin(self) -> None:
        if self._pinned is not None:
            lib, ptrs = self._pinned
            self.ref, self.qry = self.ref.copy(), self.qry.copy()
            for p in ptrs:
                lib.cvx_host_free(p)
            self._pinned, self._table = None, None

    def subset(self, idx) -> "TileSet":
        """The tiles idx (in that order) as a new TileSet (copies)."""
        idx = np.asarray(idx, dtype=np.int64)
        gather = lambda data, off: (np.concatenate([data[int(off[i]):int(off[i + 1])] for i in idx]) if len(idx) else data[:0].copy())  # noqa: E731
        W, H = self.W[idx], self.H[idx]
        out = TileSet(gather(self.ref, self.ref_off), np.concatenate([[0], np.cumsum(W)]).astype(np.int64),
                      gather(self.qry, self.qry_off), np.concatenate([[0], np.cumsum(H)]).astype(np.int64),
                      gather(self.row_offset, self.qry_off), gather(self.row_length, self.qry_off), tag=self.tag,
                      desc=None if self.desc is None else self.desc[idx].copy())
        out.closed_form = self.closed_form
        return out

    def table(self) -> np.ndarray:
        """cvx_tile[n] (include/cvx_align.h) as a structured array whose pointers reference this
        object's arrays -- keep the TileSet alive while the table is in use."""
        if self._table is None:
            t = np.zeros(self.n, dtype=TILE_DTYPE)
            t["ref"] = self.ref.ctypes.data + self.ref_off[:-1].astype(np.uint64)
            t["qry"] = self.qry.ctypes.data + self.qry_off[:-1].astype(np.uint64)
            t["ref_len"] = self.W
            t["qry_len"] = self.H
            if self.closed_form:
                t["corridor_kind"] = self.desc["kind"]
                t["corridor_k"], t["corridor_d"], t["corridor_right"] = self.desc["k"], self.desc["d"], self.desc["right"]
                t["corridor_offset"], t["corridor_width"] = self.desc["offset"], self.desc["width"]
            else:
                t["row_offset"] = self.row_offset.ctypes.data + 4 * self.qry_off[:-1].astype(np.uint64)
                t["row_length"] = self.row_length.ctypes.data + 4 * self.qry_off[:-1].astype(np.uint64)
                t["row_stride_bytes"] = 4
            self._table = t
        return self._table


def tileset_from_tiles(tiles: Sequence[Tile], tag: str = "") -> TileSet:
    """Per-tile objects -> flat arrays (descriptors kept when every tile has one)."""
    W = np.array([t.W for t in tiles], dtype=np.int64)
    H = np.array([t.H for t in tiles], dtype=np.int64)
    cat = lambda parts, dt: (np.concatenate(parts) if len(parts) else np.zeros(0, dt))  # noqa: E731
    desc = None
    if tiles and all(t.desc is not None for t in tiles):
        desc = np.array([t.desc for t in tiles], dtype=DESC_DTYPE)
    return TileSet(cat([np.frombuffer(t.ref, dtype=np.uint8) for t in tiles], np.uint8), np.concatenate([[0], np.cumsum(W)]).astype(np.int64),
                   cat([np.frombuffer(t.qry, dtype=np.uint8) for t in tiles], np.uint8), np.concatenate([[0], np.cumsum(H)]).astype(np.int64),
                   cat([np.asarray(t.row_offset, dtype=np.int32) for t in tiles], np.int32),
                   cat([np.asarray(t.row_length, dtype=np.int32) for t in tiles], np.int32), tag=tag, desc=desc)


def _pacbio_chunk(args):
    """`m` PacBio-like tiles (config C2, see workload_pacbio) as flat arrays; one process-pool task."""
    seed, m, read_len, err, scatter = args
    rng = np.random.default_rng(seed)
    refs, qrys, offs, lens, descs = [], [], [], [], []
    for _ in range(m):
        W = int(read_len * (0.9 + 0.2 * rng.random()))
        t = make_tile(rng, W, err=err, ratio=(6, 3, 1), corridor="anchors", scatter=scatter)
        refs.append(np.frombuffer(t.ref, dtype=np.uint8))
        qrys.append(np.frombuffer(t.qry, dtype=np.uint8))
        offs.append(t.row_offset)
        lens.append(t.row_length)
        descs.append(t.desc)
    W = np.array([len(r) for r in refs], dtype=np.int64)
    H = np.array([len(q) for q in qrys], dtype=np.int64)
    return np.concatenate(refs), np.concatenate(qrys), np.concatenate(offs), np.concatenate(lens), W, H, np.array(descs, dtype=DESC_DTYPE)


def pacbio_tileset(n_tiles: int, seed: int = 7, read_len: int = 10000, err: float = 0.15,
                   scatter: float = 25.0, chunk: int = 512, pool=None) -> TileSet:
    """Config C2 as a TileSet: the tiles of workload_pacbio, generated `chunk` at a time (every
    chunk has its own seed derived from `seed`), on the worker processes of `pool` (a
    concurrent.futures executor) when one is given."""
    tasks = [(seed * 100003 + c0, min(chunk, n_tiles - c0), read_len, err, scatter) for c0 in range(0, n_tiles, chunk)]
    parts = list(pool.map(_pacbio_chunk, tasks)) if pool is not None else [_pacbio_chunk(t) for t in tasks]
    W = np.concatenate([p[4] for p in parts]) if parts else np.zeros(0, np.int64)
    H = np.concatenate([p[5] for p in parts]) if parts else np.zeros(0, np.int64)
    cat = lambda k, dt: np.concatenate([p[k] for p in parts]) if parts else np.zeros(0, dt)  # noqa: E731
    return TileSet(cat(0, np.uint8), np.concatenate([[0], np.cumsum(W)]).astype(np.int64),
                   cat(1, np.uint8), np.concatenate([[0], np.cumsum(H)]).astype(np.int64),
                   cat(2, np.int32), cat(3, np.int32), tag="pacbio", desc=cat(6, DESC_DTYPE))


# --------------------------------------------------------------------------- genome-sized reference for the candidate search

def big_reference(total_bases: int = 512 << 20, n_contigs: int = 8, seed: int = 5, families: int = 24, microsats: int = 600) -> List[np.ndarray]:
    """A reference whose k-mer table cannot sit in any cache (SURVEY 8 f4 at the scale of a genome): `total_bases` of uniform
    ACGT in `n_contigs` sequences, with what makes a real genome hard for a k-mer vote -- `families` repeat families (a 3-12 kb
    unit copied 10-40 times over all contigs, each copy 1-5 % diverged) and `microsats` microsatellite stretches of 200-800 bp."""
    rng = np.random.default_rng(seed)
    per = total_bases // n_contigs
    contigs = [_ACGT[rng.integers(0, 4, size=per, dtype=np.uint8)].astype(np.uint8) for _ in range(n_contigs)]
    for _ in range(families):
        unit = random_ref(rng, int(rng.integers(3000, 12000)))
        for _c in range(int(rng.integers(10, 41))):
            v = mutate(rng, unit, float(rng.uniform(0.01, 0.05)), (1, 1, 8))
            c = contigs[int(rng.integers(0, n_contigs))]
            a = int(rng.integers(1000, per - len(v) - 1000))
            c[a:a + len(v)] = v
    for _ in range(microsats):
        motif = random_ref(rng, int(rng.integers(1, 6)))
        n = int(rng.integers(200, 800))
        c = contigs[int(rng.integers(0, n_contigs))]
        a = int(rng.integers(1000, per - n - 1000))
        c[a:a + n] = np.tile(motif, n // len(motif) + 1)[:n]
    return contigs


def kmer_table(lib, contigs: Sequence[np.ndarray], k: int = 13, skip: int = 2, bin_shift: int = 4, device: int = None, keep: bool = False):
    """-> (index bytes [(4^k + 2) * 5], locations uint32[], start table uint64[]) for `contigs`: cvx_genome_encode + cvx_index_build,
    i.e. what ngmlr's SequenceProvider and CompactPrefixTable build from the same sequences (host only); device = a device
    number: the table by cvx_index_build_device instead (same bytes); keep: CVX_INDEX_KEEP_RESIDENT -- a KmerIndex made from exactly
    the returned arrays on that device then takes the device's copy over."""
    import ctypes as C
    from . import capi
    n = len(contigs)
    lens = np.array([len(c) for c in contigs], dtype=np.uint64)
    lib.cvx_genome_encoded_bytes.restype = C.c_uint64
    binref = np.zeros(int(lib.cvx_genome_encoded_bytes(n, lens.ctypes.data_as(C.c_void_p))), dtype=np.uint8)
    arr = (C.c_char_p * n)()
    for i, c in enumerate(contigs):
        arr[i] = C.cast(np.ascontiguousarray(c).ctypes.data, C.c_char_p)
    nn, ns = C.c_uint64(), C.c_int32()
    starts = np.zeros(n + 1, dtype=np.uint64)
    capi.check(lib.cvx_genome_encode(n, arr, lens.ctypes.data_as(C.c_void_p), binref.ctypes.data_as(C.c_void_p), C.byref(nn),
                                     starts.ctypes.data_as(C.c_void_p), C.byref(ns)))
    kept = np.ascontiguousarray(lens[lens > 10])
    idx = np.zeros(((1 << (2 * k)) + 2) * 5, dtype=np.uint8)
    cap = int(kept.sum()) // (skip + 1) + 64
    locs = np.zeros(cap, dtype=np.uint32)
    nl = C.c_uint64()
    if device is None:
        capi.check(lib.cvx_index_build(binref.ctypes.data, nn.value, starts.ctypes.data, kept.ctypes.data, len(kept), k, skip, bin_shift,
                                       idx.ctypes.data, locs.ctypes.data, cap, C.byref(nl)))
    else:
        capi.check(lib.cvx_index_build_device(device, binref.ctypes.data, nn.value, starts.ctypes.data, kept.ctypes.data, len(kept), k, skip, bin_shift,
                                              idx.ctypes.data, locs.ctypes.data, cap, C.byref(nl), 1 if keep else 0))
    return idx, locs[:nl.value], starts[:ns.value]


def sample_subreads(contigs: Sequence[np.ndarray], n: int, length: int = 256, err: float = 0.15, seed: int = 9) -> List[bytes]:
    """n sub-reads of `length` bases (ngmlr's --subread-length) cut from PacBio-like reads of the reference (15 % error 6:3:1, half
    of them reverse-complemented), positions uniform over the contigs -- so repeat copies and microsatellites are hit in proportion."""
    rng = np.random.default_rng(seed)
    out = []
    per_read = 40                                         # sub-reads cut from one 10 kb read
    while len(out) < n:
        c = contigs[int(rng.integers(0, len(contigs)))]
        a = int(rng.integers(0, len(c) - 12000))
        q = mutate(rng, c[a:a + 10400], err, (6, 3, 1))
        if rng.random() < 0.5:
            q = revcomp(q)
        raw = q.tobytes()
        for j in range(min(per_read, len(raw) // length)):
            out.append(raw[j * length:(j + 1) * length])
    return out[:n]
