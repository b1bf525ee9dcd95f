"""Shared helpers for the tests: golden-fixture loading and a seeded tile zoo."""
import os

import numpy as np

from ngmlr_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# every SingleAlign call the reference makes on its test_3 reads (985 tiles, 1.26 G cells): written by
# tools/make_golden.sh beside the other reference-derived build artefacts (git-ignored; it travels to
# the GPU box with the snapshot).  tests/golden/ref_test_3.npz is the committed 60-tile subset.
GOLDEN_FULL = os.path.join(os.path.dirname(GOLDEN), os.pardir, "oracle", "_ref", "golden_full")


def full_golden_path(name="ref_test_3_full.npz"):
    p = os.path.normpath(os.path.join(GOLDEN_FULL, name))
    return p if os.path.exists(p) else None

FIELD_NAMES = ("position_offset", "qstart", "qend", "nm", "alignment_length", "cigar_op_count", "sv_type",
               "first_ref", "first_read", "last_ref", "last_read")


def load_golden(name):
    """-> list of (Tile, expected dict) recorded from the unmodified reference binary
    (tools/make_golden.sh) on its own test data."""
    z = np.load(name if os.path.isabs(name) else os.path.join(GOLDEN, name))
    name = os.path.basename(name)
    out = []
    for i in range(int(z["n"])):
        p = "t%d_" % i
        meta = z[p + "meta"]
        t = synth.Tile(ref=z[p + "ref"].tobytes(), qry=z[p + "qry"].tobytes(),
                       row_offset=z[p + "off"].astype(np.int32), row_length=z[p + "len"].astype(np.int32),
                       ext_qstart=int(meta[0]), ext_qend=int(meta[1]), tag="%s#%d" % (name, i))
        exp = {"ret": int(meta[2]), "score_bits": int(z[p + "bits"][0]), "identity_bits": int(z[p + "bits"][1]),
               "cigar": z[p + "cigar"].tobytes().decode(), "md": z[p + "md"].tobytes().decode(),
               "nm_per_position": z[p + "nm"].astype(np.int32)}
        for k, v in zip(FIELD_NAMES, z[p + "fields"]):
            exp[k] = int(v)
        exp["identity"] = float(np.uint32(exp["identity_bits"]).view(np.float32))
        out.append((t, exp))
    return out


def golden_diff(exp, got):
    """None if `got` (oracle or HIP dict) equals the recorded reference output."""
    if exp["ret"] < 0:
        return None if got["ret"] < 0 else "ret: reference -1, got %d" % got["ret"]
    for k in ("ret", "score_bits", "cigar", "md") + FIELD_NAMES:
        if exp[k] != got[k]:
            return "%s: %r != %r" % (k, str(exp[k])[:60], str(got[k])[:60])
    if int(np.float32(got["identity"]).view(np.uint32)) != exp["identity_bits"]:
        return "identity"
    a, b = exp["nm_per_position"], got["nm_per_position"]
    if a.shape != b.shape or not np.array_equal(a, b):
        return "nm_per_position"
    return None


def wrap16_tile(seed=5, pre=40000, ins=33000, post=20000, w=400):
    """A tile whose BEST alignment carries an insertion run past SHRT_MAX: 40 kb matched, 33 kb of
    junk inserted in the read, 20 kb matched again (score 80 k - 33 k + 40 k: bridging the gap pays),
    corridor = a 400-column band that follows that path (diagonal, vertical for the insertion,
    diagonal).  indelRun is a `short` in the reference (src/AlignmentMatrixFast.h:43): at run 32768 it
    wraps negative, the extension test `ins_run > 0` fails and the CIGAR breaks into
    ...32767I1M233I...; an aligner that keeps the run in a wider type reports one 33000I instead."""
    rng = np.random.default_rng(seed)
    ref = synth.random_ref(rng, pre + post + 200)
    junk = synth.random_ref(rng, ins)
    qry = np.concatenate([ref[:pre], junk, ref[pre:pre + post]])
    H = len(qry)
    y = np.arange(H)
    off = np.where(y < pre, y - w // 2, np.where(y < pre + ins, pre - w // 2, y - ins - w // 2)).astype(np.int32)
    return synth.Tile(ref.tobytes(), qry.tobytes(), off, np.full(H, w, np.int32), tag="wrap16")


def tile_zoo(seed=123, n=60, max_w=2500):
    """Seeded mix over every corridor constructor, error model and the odd symbols."""
    rng = np.random.default_rng(seed)
    tiles = []
    kinds = ["anchors", "endpoints", "linear", "full", "anchors", "endpoints"]
    for i in range(n):
        kind = kinds[i % len(kinds)]
        W = int(rng.integers(20, max_w))
        if kind == "full":
            W = min(W, 700)
        err = float(rng.choice([0.02, 0.1, 0.15, 0.3, 0.45]))
        ratio = [(6, 3, 1), (4, 4, 2), (1, 1, 1)][i % 3]
        t = synth.make_tile(rng, W, err=err, ratio=ratio, corridor=kind, scatter=float(rng.choice([0, 30, 120])),
                            mult=int(rng.integers(1, 4)), n_frac=0.01 if i % 4 == 0 else 0.0,
                            x_frac=0.01 if i % 7 == 0 else 0.0, realign=bool(i % 2))
        if i % 5 == 0:
            t.ext_qstart = int(rng.integers(0, 300))
            t.ext_qend = int(rng.integers(0, 300))
        tiles.append(t)
    return tiles


def edge_tiles():
    """Edge cases the reference's recurrence meets in the field."""
    rng = np.random.default_rng(99)
    out = []
    # tiny reads and windows
    for H, W in ((1, 1), (1, 40), (2, 3), (5, 300), (17, 17), (40, 2)):
        ref = synth.random_ref(rng, W)
        qry = synth.random_ref(rng, H)
        off, ln = synth.corridor_linear(H, 30)
        out.append(synth.Tile(ref.tobytes(), qry.tobytes(), off, ln, tag="tiny%dx%d" % (H, W)))
    # identical sequences (pure diagonal), all-mismatch (score 0 everywhere), poly-A
    r = synth.random_ref(rng, 400)
    off, ln = synth.corridor_anchors(400, 400)
    out.append(synth.Tile(r.tobytes(), r.tobytes(), off, ln, tag="identical"))
    out.append(synth.Tile(b"A" * 300, b"C" * 300, *synth.corridor_anchors(300, 300), tag="all-mismatch"))
    out.append(synth.Tile(b"A" * 500, b"A" * 450, *synth.corridor_anchors(450, 500), tag="poly-a"))
    # N == N is a match, 'x' never matches
    out.append(synth.Tile(b"ACGTNNNNNNNNACGTACGT" * 10, b"ACGTNNNNNNNNACGTACGT" * 10, *synth.corridor_anchors(200, 200), tag="n-match"))
    out.append(synth.Tile(b"ACGTxxxxxxxxACGTACGT" * 10, b"ACGTNNNNNNNNACGTACGT" * 10, *synth.corridor_anchors(200, 200), tag="x-never"))
    # long deletion / insertion (convex gap decay saturates at run 27)
    a = synth.random_ref(rng, 900)
    out.append(synth.Tile(a.tobytes(), np.concatenate([a[:400], a[480:]]).tobytes(), *synth.corridor_anchors(820, 900), tag="del80"))
    out.append(synth.Tile(np.concatenate([a[:400], a[480:]]).tobytes(), a.tobytes(), *synth.corridor_anchors(900, 820), tag="ins80"))
    # corridor too narrow for the true path -> validPath rejects (ret -1)
    b = synth.random_ref(rng, 1200)
    out.append(synth.Tile(b.tobytes(), np.concatenate([b[:300], b[700:]]).tobytes(), *synth.corridor_linear(800, 64), tag="narrow"))
    # corridor hanging over both ends of the window, ragged clipping
    c = synth.random_ref(rng, 300)
    H = 260
    off = (np.arange(H) * 2 - 200).astype(np.int32)
    out.append(synth.Tile(c.tobytes(), synth.mutate(rng, c, 0.1)[:H].tobytes().ljust(H, b"A"), off, np.full(H, 250, np.int32), tag="overhang"))
    # row lengths that differ row to row (ragged corridor, still monotone)
    d = synth.random_ref(rng, 600)
    q = synth.mutate(rng, d, 0.1)
    H = len(q)
    off = (np.arange(H) - 100).astype(np.int32)
    ln = (200 + (np.arange(H) % 7) * 3).astype(np.int32)
    out.append(synth.Tile(d.tobytes(), q.tobytes(), off, ln, tag="ragged"))
    return out


# --------------------------------------------------------------------------- corridor closed forms

def _f32_from_ord(o):
    """ordered int64 -> float32 (monotone bijection between float32 values and integers)"""
    o = np.asarray(o, dtype=np.int64)
    bits = np.where(o >= 0, o, (-(o + 1)) | np.int64(0x80000000))
    return bits.astype(np.uint32).view(np.float32)


def _ord_from_f32(x):
    b = np.asarray(np.float32(x)).view(np.uint32).astype(np.int64)
    return np.where(b & 0x80000000, -(b & 0x7fffffff) - 1, b)


def fit_corridor(off, ln, H, W):
    """Closed form (kind, k, d, right, offset, width) that reproduces the recorded rows of one of the reference's
    corridor builders bit for bit, or None.  Test infrastructure: the recorder sees only the CorridorLine[] a
    SingleAlign call received, not which builder made it (src/AlignmentBuffer.cpp:68-197), so the builder and --
    for the anchors corridor -- a `corridorRight` consistent with every row are recovered here.  k is always
    qryLen * 1.0f / refLen (:117, :141); the endpoints corridor has d = width / 2.0f and no shift (:118-124); the
    anchors corridor has d = 0 and offset = (int)(i / k - right) (:190): every row bounds `right` from both sides
    (the expression is monotone in it), and any float32 inside the intersection generates identical rows."""
    off = np.asarray(off, dtype=np.int32)
    ln = np.asarray(ln, dtype=np.int32)
    if H == 0:
        return (2, 0.0, 0.0, 0.0, 0, 0)
    w = int(ln[0])
    if not np.all(ln == w):
        return None
    if np.all(off == off[0]):
        return (2, 0.0, 0.0, 0.0, int(off[0]), w)
    y = np.arange(H, dtype=np.int64)
    if np.array_equal(off.astype(np.int64), y + int(off[0])) and abs(int(off[0])) < (1 << 23) and H < (1 << 23):
        return (1, 1.0, float(-int(off[0])), 0.0, 0, w)          # getCorridorLinear: i - width / 2
    F32 = np.float32
    k = F32(H) * F32(1.0) / F32(max(W, 1))
    d = F32(w) / F32(2.0)
    o2, _ = synth.affine_rows(H, k, d, 0.0, w)
    if np.array_equal(o2, off):
        return (1, float(k), float(d), 0.0, 0, w)                  # getCorridorEndpoints
    q = (y.astype(F32) - F32(0)) / k

    def g(right):
        return np.trunc(F32(q - F32(right))).astype(np.int64)
    # smallest `right` with g <= off everywhere (g falls as right grows), largest with g >= off everywhere
    lo, hi = int(_ord_from_f32(-3.0e6)), int(_ord_from_f32(3.0e6))
    a_lo, a_hi = lo, hi
    while a_lo < a_hi:
        m = (a_lo + a_hi) // 2
        if np.all(g(_f32_from_ord(m)) <= off):
            a_hi = m
        else:
            a_lo = m + 1
    right = float(_f32_from_ord(a_lo))
    o3, _ = synth.affine_rows(H, k, 0.0, right, w)
    if np.array_equal(o3, off):
        return (1, float(k), 0.0, right, 0, w)                     # getCorridorEndpointsWithAnchors
    return None


# --------------------------------------------------------------------------- N-clip flags

def nclip_tiles(seed=31):
    """Tiles whose alignment ends next to runs of 'X' in the reference window: the reference sets svType |= 0x1 when
    more than 80 % of (up to) 100 reference characters before the alignment start / after its end are 'X'
    (src/ConvexAlignFast.cpp:493-528).  -> list of (tile, expected flag).  'X' never matches a read base, so the local
    alignment clips exactly at the run."""
    rng = np.random.default_rng(seed)
    out = []

    def tile(lead, trail, core=700, tag=""):
        mid = synth.random_ref(rng, core)
        ref = np.concatenate([lead, mid, trail]).astype(np.uint8)
        qry = synth.mutate(rng, mid, 0.08)
        off, ln = synth.corridor_full(len(qry), len(ref))
        return synth.Tile(ref.tobytes(), qry.tobytes(), off, ln, tag=tag, desc=synth.full_desc(len(qry), len(ref)))
    X = lambda n: np.full(n, ord("X"), np.uint8)  # noqa: E731
    acgt = lambda n: synth.random_ref(rng, n)  # noqa: E731
    out.append((tile(X(150), acgt(0), tag="x-before"), 1))
    out.append((tile(acgt(0), X(150), tag="x-after"), 1))
    out.append((tile(X(120), X(130), tag="x-both"), 1))
    out.append((tile(X(30), acgt(0), tag="x-before-short-window"), 1))            # fewer than 100 characters to probe
    out.append((tile(np.concatenate([X(60), acgt(45)]), acgt(0), tag="x-too-far"), 0))     # 55 of 100 probes
    mixed = X(100).copy()
    mixed[::4] = ord("A")                                                          # 75 % X: below the threshold
    out.append((tile(mixed, acgt(0), tag="x-75-percent"), 0))
    mixed2 = X(100).copy()
    mixed2[::8] = ord("N")                                                         # 87 % X, 'N' does not count
    out.append((tile(mixed2, acgt(0), tag="x-87-percent"), 1))
    out.append((tile(np.full(150, ord("N"), np.uint8), acgt(0), tag="n-not-x"), 0))   # the decoder's 'N' is not what the test looks for
    return out


# --------------------------------------------------------------------------- candidate search

def synthetic_search_case(seed=41):
    """A small synthetic k-mer table + reads for the corners the recorded test_3 calls do not reach.
    -> (SearchFixture-like object for oracle and device, list of reads)"""
    from types import SimpleNamespace
    rng = np.random.default_rng(seed)
    K = 13

    def code(c):
        return (c >> 1) & 3

    def kmers(s):
        out = []
        for p in range(len(s) - K + 1):
            w = s[p:p + K]
            if ord("N") in w:
                continue
            v = 0
            for c in w:
                v = (v << 2) | code(c)
            out.append(v)
        return out

    def revcomp_code(v):
        c = (v ^ 0xAAAAAAAA) & ((1 << 26) - 1)
        r = 0
        for _ in range(K):
            r = (r << 2) | (c & 3)
            c >>= 2
        return r
    base = synth.random_ref(rng, 256).tobytes()
    heavy = synth.random_ref(rng, 256).tobytes()                         # a read whose k-mers are all over the genome
    rows = {}
    for v in kmers(base):
        rows.setdefault(v, set()).update(int(x) for x in (1000 + 16 * rng.integers(0, 4000, size=3)))
        rows.setdefault(v, set()).add(500000 + 0)                         # a shared diagonal: the true location
    # place base's true location consistently: location of k-mer at read offset p = 500000 + p
    rows = {}
    for p, v in enumerate(kmers(base)):
        rows.setdefault(v, set()).add(500000 + p)
        for x in rng.integers(10_000, 40_000_000, size=2):
            rows[v].add(int(x))
    for p, v in enumerate(kmers(heavy)):
        tgt = rows.setdefault(v, set())
        for x in rng.integers(50_000_000, 4_000_000_000, size=480):       # ~117 000 votes into ~117 000 different bins
            tgt.add(int(x))
        rv = revcomp_code(v)
        tgt = rows.setdefault(rv, set())
        for x in rng.integers(50_000_000, 4_000_000_000, size=480):
            tgt.add(int(x))
    prefixes = np.array(sorted(rows), dtype=np.uint32)
    cnt = np.array([len(rows[int(p)]) for p in prefixes], dtype=np.uint32)
    locs = np.concatenate([np.array(sorted(rows[int(p)]), dtype=np.uint32) for p in prefixes])
    fx = SimpleNamespace(k=K, unit_offset=0, prefix=prefixes, cnt=cnt, rc=np.full(len(prefixes), 50, np.int8), locs=locs)

    def index_arrays():
        n = (1 << (2 * K)) + 2
        cnt_full = np.zeros(n, dtype=np.int64)
        cnt_full[prefixes] = cnt
        tab = (1 + np.concatenate([[0], np.cumsum(cnt_full)[:-1]])).astype(np.uint32)
        rc = np.zeros(n, dtype=np.int8)
        rc[prefixes] = 50
        idx = np.zeros(n, dtype=np.dtype([("tab", "<u4"), ("rc", "i1")]))
        idx["tab"], idx["rc"] = tab, rc
        return idx, locs
    fx.index_arrays = index_arrays
    b = bytearray(base)
    with_n = bytes(b[:100] + b"NNN" + b[103:180] + b"N" + b[181:])
    tail2 = bytes(b[:241] + b"NN" + b[243:])                              # 13 characters behind a run of two N
    tail1 = bytes(b[:242] + b"N" + b[243:])                               # 13 characters behind a single N
    lead = bytes(b"NNNN" + b[4:])
    short = bytes(b[:12])
    allN = b"N" * 40
    # k-mers the table does not know (kCount, src/CS.cpp:67-69): a foreign read (all of them), a known read with a foreign
    # tail, and the heavy read -- which climbs the ladder, so whatever it misses is counted once per attempt -- with one
    foreign = synth.random_ref(rng, 256).tobytes()
    reads = [base, with_n, tail2, tail1, lead, short, allN, heavy, bytes(b[:40]), b"", bytes(b[:13]),
             foreign, bytes(b[:150]) + foreign[:106], foreign[:56] + heavy[:200]]
    return fx, reads


# --------------------------------------------------------------------------- candidate search at genome scale

_BIG = {}


def big_search_case():
    """tests/golden/cs_big.npz (tools/make_golden_cs.sh --big): 840 candidate-search calls recorded from the unmodified reference on
    ngmlr_amd.synth.big_reference(512 Mbp) -- a k-mer table that leaves every cache.  The 1 GB table is not stored: it is rebuilt here
    with cvx_index_build from the same generator and must hash to what the packer saw equal to the reference's own table.
    -> (fixture-like object with seqs / want / max_hit / thresh / rlist_len / kmer_misses / first_bits, k, index bytes, locations)"""
    if "case" in _BIG:
        return _BIG["case"]
    import hashlib
    from ngmlr_amd import capi
    z = np.load(os.path.join(GOLDEN, "cs_big.npz"))
    contigs = synth.big_reference(512 << 20, n_contigs=8)
    idx5, locs, _ = synth.kmer_table(capi.load(), contigs, k=int(z["k"]), skip=int(z["ref_skip"]))
    del contigs
    assert hashlib.sha256(idx5.tobytes()).hexdigest() == str(z["index_sha256"]), "the rebuilt index is not the recorded reference table"
    assert hashlib.sha256(locs.tobytes()).hexdigest() == str(z["locs_sha256"]) and len(locs) == int(z["n_locations"])

    class Case:
        pass
    fx = Case()
    off = np.concatenate([[0], np.cumsum(z["seq_len"].astype(np.int64))])
    raw = z["seqs"].tobytes()
    fx.seqs = [raw[int(off[i]):int(off[i + 1])] for i in range(len(z["seq_len"]))]
    so = np.concatenate([[0], np.cumsum(z["n_scores"].astype(np.int64))])
    fx.want = [(z["loc"][int(so[i]):int(so[i + 1])], z["score"][int(so[i]):int(so[i + 1])], z["rev"][int(so[i]):int(so[i + 1])]) for i in range(len(fx.seqs))]
    fx.max_hit, fx.thresh, fx.rlist_len = z["max_hit"], z["thresh"], z["rlist_len"]
    fx.kmer_misses, fx.first_bits = z["kmer_misses"], z["first_bits"]
    fx.k, fx.unit_offset = int(z["k"]), int(z["unit_offset"])
    _BIG["case"] = (fx, idx5, locs)
    return _BIG["case"]
