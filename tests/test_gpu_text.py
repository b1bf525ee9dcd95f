"""GPU (-m gpu): the device-side text stage (cvx_job_text) against the host form
(cvx_format_alignment, itself pinned to the reference's convertCigar by tests/test_host_format_cpu.py and
the golden tiles): every cvx_alignment_text field, the identity bits, and the CIGAR / MD strings byte by
byte -- over every corridor kind, external clips, long deletions and insertions, invalid tiles, tiles with
hundreds of ops per 64-op step boundary, and full-size PacBio tiles."""
import numpy as np
import pytest

from ngmlr_amd import capi
from ngmlr_amd.aligner import format_alignment
from tests import util

pytestmark = pytest.mark.gpu

FIELDS = ("ret", "score_bits", "position_offset", "qstart", "qend", "nm", "alignment_length", "cigar_op_count",
          "sv_type", "first_ref", "first_read", "last_ref", "last_read", "nm_count", "cigar_len", "md_len", "cigar", "md")


def _compare(al, tiles):
    job = al.submit(tiles)
    res, ops = job.wait()
    eqs = np.array([t.ext_qstart for t in tiles], dtype=np.int32)
    eqe = np.array([t.ext_qend for t in tiles], dtype=np.int32)
    dev = job.text(eqs, eqe)
    bad = []
    n_valid = 0
    for i, t in enumerate(tiles):
        r = capi.CvxResult.from_buffer_copy(res[i].tobytes())
        host = format_alignment(al.lib, r, ops, t)
        d = dev[i]
        n_valid += host["ret"] >= 0
        for k in FIELDS:
            if host[k] != d[k]:
                bad.append((t.tag, k, str(host[k])[:60], str(d[k])[:60]))
                break
        else:
            if np.float32(host["identity"]).view(np.uint32) != np.float32(d["identity"]).view(np.uint32):
                bad.append((t.tag, "identity", host["identity"], d["identity"]))
    job.release()
    assert not bad, bad[:5]
    return n_valid


def test_text_stage_equals_host_form_on_the_zoo(hip_aligner):
    assert _compare(hip_aligner, util.tile_zoo(seed=64, n=180, max_w=3000) + util.edge_tiles()) > 100


def test_text_stage_on_golden_tiles(hip_aligner):
    tiles = [t for name in ("ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz") for t, _ in util.load_golden(name)]
    assert _compare(hip_aligner, tiles) > 40


def test_text_stage_long_gaps_and_full_size(hip_aligner):
    from ngmlr_amd import synth
    rng = np.random.default_rng(21)
    tiles = synth.workload_pacbio(6, seed=5) + synth.workload_ont(20, seed=6, max_len=9000)
    from tests.test_gpu_parity import _sv_tile
    # deletions / insertions of 70..400 bases: single ops whose MD text is hundreds of characters
    tiles.append(_sv_tile(rng, 900, [70, 130], [65, 200], "full"))
    tiles.append(_sv_tile(rng, 1300, [400], [257], "full"))
    tiles.append(_sv_tile(rng, 700, [1, 2, 3, 33], [1, 2, 64], "endpoints"))
    tiles.append(util.wrap16_tile(pre=3000, ins=400, post=2500, w=300))
    for i, t in enumerate(tiles):
        if i % 3 == 0:
            t.ext_qstart, t.ext_qend = int(rng.integers(0, 5000)), int(rng.integers(0, 5000))
    assert _compare(hip_aligner, tiles) > 20


def _device_text_vs(al, tiles, expect):
    """device text of `tiles` against expect(i) -> dict with the reference's fields (ret, score_bits, cigar, md, ...)"""
    job = al.submit(tiles)
    job.wait()
    eqs = np.array([t.ext_qstart for t in tiles], dtype=np.int32)
    eqe = np.array([t.ext_qend for t in tiles], dtype=np.int32)
    dev = job.text(eqs, eqe)
    job.release()
    bad = []
    for i, t in enumerate(tiles):
        w, d = expect(i), dev[i]
        if w["ret"] < 0:
            if d["ret"] >= 0:
                bad.append((t.tag, "ret", w["ret"], d["ret"]))
            continue
        for k in ("ret", "score_bits", "cigar", "md", "position_offset", "qstart", "qend", "nm", "alignment_length",
                  "cigar_op_count", "sv_type", "first_ref", "first_read", "last_ref", "last_read"):
            if w[k] != d[k]:
                bad.append((t.tag, k, str(w[k])[:50], str(d[k])[:50]))
                break
        else:
            wi = w["identity_bits"] if "identity_bits" in w else int(np.float32(w["identity"]).view(np.uint32))
            if wi != int(np.float32(d["identity"]).view(np.uint32)):
                bad.append((t.tag, "identity"))
    assert not bad, bad[:5]


@pytest.mark.parametrize("name", ["ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz"])
def test_device_text_equals_recorded_reference_output(hip_aligner, name):
    """Directly against what the unmodified reference wrote for its own SingleAlign calls (not through the host form)."""
    pairs = util.load_golden(name)
    _device_text_vs(hip_aligner, [t for t, _ in pairs], lambda i: pairs[i][1])


def test_device_text_equals_reference_aligner(hip_aligner, ref_oracle):
    """... and against the reference's own ConvexAlignFast (oracle/_ref) on the zoo, the edge cases and the N-clip tiles
    (svType 0x1 fires in both directions, src/ConvexAlignFast.cpp:493-528)."""
    nclip = util.nclip_tiles()
    tiles = util.tile_zoo(seed=66, n=60, max_w=2000) + util.edge_tiles() + [t for t, _ in nclip]
    want = [ref_oracle.align(t, want_nm=False) for t in tiles]
    _device_text_vs(hip_aligner, tiles, lambda i: want[i])
    base = len(tiles) - len(nclip)
    assert sum(want[base + k]["sv_type"] for k in range(len(nclip))) == sum(f for _, f in nclip) >= 4


def test_n_clip_flags_through_the_host_form(hip_aligner, ref_oracle):
    from oracle.pyoracle import same_alignment
    nclip = util.nclip_tiles()
    got = hip_aligner.batch_align([t for t, _ in nclip])
    for (t, flag), g in zip(nclip, got):
        want = ref_oracle.align(t)
        assert same_alignment(want, g) is None, (t.tag, same_alignment(want, g))
        assert g["sv_type"] == flag == want["sv_type"]


# ------------------------------------------------------------------ nmPerPosition on the device (cvx_job_nm_profile)

def _device_profiles(al, tiles, ranges=None):
    """-> per tile the int32[entries, 3] profile computed on the device (None for invalid tiles), the text records"""
    job = al.submit(tiles)
    job.wait()
    eqs = np.array([t.ext_qstart for t in tiles], dtype=np.int32)
    eqe = np.array([t.ext_qend for t in tiles], dtype=np.int32)
    dev = job.text(eqs, eqe)
    out = [None] * len(tiles)
    for first, count in (ranges or [(0, len(tiles))]):
        off, tri, ms = job.nm_profile(first, count)
        assert ms >= 0.0 and off[0] == 0 and int(off[count]) == len(tri)
        off2, tri2, _ = job.nm_profile_resident(first, count)      # ABI 9: the triples in the job's page-locked memory
        assert np.array_equal(off, off2) and np.array_equal(tri, tri2)
        for i in range(count):
            d = dev[first + i]
            n = int(off[i + 1] - off[i])
            assert n == (d["nm_count"] if d["ret"] >= 0 else 0), (tiles[first + i].tag, n, d["nm_count"])
            out[first + i] = tri[int(off[i]):int(off[i + 1])]
    if not ranges:
        # ABI 9: both stages of the whole job in one call (what the pipeline's dispatcher runs per finished launch)
        dev3, off3, tri3 = job.text_all(eqs, eqe)
        assert dev3 == dev
        for i in range(len(tiles)):
            assert np.array_equal(tri3[int(off3[i]):int(off3[i + 1])], out[i]), tiles[i].tag
    job.release()
    return out, dev


def test_nm_profile_equals_recorded_reference_output(hip_aligner):
    """The per-position profile the unmodified reference wrote for its own SingleAlign calls (test_2/3/4)."""
    n = 0
    for name in ("ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz"):
        pairs = util.load_golden(name)
        got, dev = _device_profiles(hip_aligner, [t for t, _ in pairs])
        for (t, exp), g, d in zip(pairs, got, dev):
            if exp["ret"] < 0:
                assert len(g) == 0
                continue
            want = exp["nm_per_position"][:d["nm_count"]]
            assert g.shape == want.shape and np.array_equal(g, want), (t.tag, np.argwhere(g != want)[:3])
            n += len(g)
    assert n > 50000


def test_nm_profile_equals_reference_aligner_and_host_form(hip_aligner, ref_oracle):
    """Against the reference's own ConvexAlignFast (oracle/_ref) and the host form, on the zoo, the edge tiles, gap ops
    next to each other and next to the start, long gaps, and in tile ranges (the profile of a full batch is GBs)."""
    from tests.test_gpu_parity import _sv_tile
    rng = np.random.default_rng(77)
    tiles = util.tile_zoo(seed=67, n=70, max_w=2000) + util.edge_tiles()
    tiles.append(_sv_tile(rng, 900, [70, 130], [65, 200], "full"))
    tiles.append(_sv_tile(rng, 700, [1, 2, 3, 33], [1, 2, 64], "endpoints"))
    tiles.append(_sv_tile(rng, 600, [31, 32, 33], [31, 32, 33], "full"))
    n = len(tiles)
    got, dev = _device_profiles(hip_aligner, tiles, ranges=[(0, 7), (7, n - 20), (n - 13, 13)])
    valid = entries = 0
    for t, g, d in zip(tiles, got, dev):
        want = ref_oracle.align(t)
        if want["ret"] < 0:
            assert d["ret"] < 0 and len(g) == 0
            continue
        w = want["nm_per_position"]          # alignmentLength rows (what the consumer walks): the entries, then zeros
        assert len(g) <= len(w) and np.array_equal(g, w[:len(g)]) and not w[len(g):].any(), (t.tag, g.shape, w.shape)
        valid += 1
        entries += len(g)
    assert valid > 40 and entries > 20000
    # the same entry point refuses what it cannot do
    job = hip_aligner.submit(tiles[:4])
    job.wait()
    off = np.zeros(5, dtype=np.uint64)
    assert hip_aligner.lib.cvx_job_nm_profile(hip_aligner.h, job.j, 0, 4, off.ctypes.data, None, 0, None) == -3   # no text stage yet
    job.text()
    small = np.zeros((1, 3), dtype=np.int32)
    rc = hip_aligner.lib.cvx_job_nm_profile(hip_aligner.h, job.j, 0, 4, off.ctypes.data, small.ctypes.data, 1, None)
    assert rc == -6 and off[4] > 1
    assert hip_aligner.lib.cvx_job_nm_profile(hip_aligner.h, job.j, 2, 3, off.ctypes.data, None, 0, None) == -3   # range past the job
    job.release()


def test_nm_profile_beside_replaced_segments(hip_aligner):
    """Reads in which a stretch of the reference is replaced by novel bases of another length (the convex scoring reports
    such a stretch as mismatches plus one gap, or as two gap ops behind each other): device profile == host form."""
    from ngmlr_amd import synth
    rng = np.random.default_rng(5)
    tiles = []
    for k in range(12):
        ref = synth.random_ref(rng, 1500)
        a = 600 + 10 * k
        # read = ref[:a] + 12..30 novel bases + ref[a+d:], d = 12..30: an I run beside a D run
        ins = synth.random_ref(rng, 12 + k)
        d = 30 - k
        qry = np.concatenate([ref[100:a], ins, ref[a + d:1400]])
        H, W = len(qry), len(ref)
        off = (np.arange(H) * (W / H)).astype(np.int32) - 260
        tiles.append(synth.Tile(ref=ref.tobytes(), qry=qry.tobytes(), row_offset=off, row_length=np.full(H, 520, dtype=np.int32),
                                tag="gapchain%d" % k))
    got, dev = _device_profiles(hip_aligner, tiles)
    job = hip_aligner.submit(tiles)
    res, ops = job.wait()
    for i, t in enumerate(tiles):
        r = capi.CvxResult.from_buffer_copy(res[i].tobytes())
        host = format_alignment(hip_aligner.lib, r, ops, t)
        assert host["ret"] >= 0
        w = host["nm_per_position"][:host["nm_count"]]
        assert got[i].shape == w.shape and np.array_equal(got[i], w), t.tag
        assert len(w) > 1000
    job.release()


def test_nm_profile_of_arbitrary_op_lists(hip_aligner):
    """cvx_nm_profile_ops on op lists no alignment of the default scoring produces -- gap ops directly behind each other
    (the Yi + 1 chain), gap ops in front of everything, mismatch runs longer than the 32-column register, ops longer than a
    wave, hundreds of ops (several 64-op steps) -- against the host form walking the same lists."""
    import ctypes as C
    lib = hip_aligner.lib
    rng = np.random.default_rng(99)
    EQ, X, I, D = 7, 8, 1, 2
    lists = [
        [(5, I), (3, D), (40, EQ), (2, X), (1, I), (1, D), (1, I), (30, EQ)],
        [(20, D), (20, I), (100, EQ)],
        [(18, EQ), (50, X), (3, D), (2, I), (4, D), (70, EQ), (33, X), (1, EQ)],
        [(300, EQ), (1, D), (200, X), (7, I), (90, EQ)],
    ]
    for _ in range(40):
        n = int(rng.integers(1, 400))
        ops = []
        for _k in range(n):
            t = int(rng.choice([EQ, EQ, EQ, X, I, D]))
            ln = int(rng.integers(1, 4)) if rng.random() < 0.8 else int(rng.integers(4, 120))
            ops.append((ln, t))
        lists.append(ops)
    arena, results, qstarts = [], [], []
    for li, ops in enumerate(lists):
        r = capi.CvxResult()
        r.status = 0
        r.ref_position = 0
        r.qstart = int(rng.integers(0, 40)) if li % 2 else 0
        r.n_ops = len(ops)
        r.ops_begin = len(arena)
        arena += [(ln << 4) | t for ln, t in ops]
        results.append(r)
    # one invalid tile in the middle: no entries
    bad = capi.CvxResult()
    bad.status = 1
    results.insert(3, bad)
    lists.insert(3, [])
    arena = np.array(arena, dtype=np.uint32)
    res = (capi.CvxResult * len(results))(*results)
    off = np.zeros(len(results) + 1, dtype=np.uint64)
    assert lib.cvx_nm_profile_ops(hip_aligner.h, len(results), res, arena.ctypes.data, len(arena), off.ctypes.data, None, 0) == 0
    tri = np.zeros((int(off[-1]), 3), dtype=np.int32)
    assert lib.cvx_nm_profile_ops(hip_aligner.h, len(results), res, arena.ctypes.data, len(arena), off.ctypes.data, tri.ctypes.data, len(tri)) == 0
    total = 0
    for i, ops in enumerate(lists):
        got = tri[int(off[i]):int(off[i + 1])]
        if results[i].status != 0:
            assert len(got) == 0
            continue
        ref_len = sum(ln for ln, t in ops if t != I) + 300
        qry_len = sum(ln for ln, t in ops if t != D) + results[i].qstart
        ref = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=ref_len).astype(np.uint8))
        cap = 16 * (ref_len + qry_len) + 64
        cig, md = C.create_string_buffer(cap), C.create_string_buffer(cap)
        nm = np.zeros((ref_len + qry_len + 16, 3), dtype=np.int32)
        txt = capi.CvxAlignmentText()
        assert lib.cvx_format_alignment(C.byref(results[i]), arena.ctypes.data, ref, ref_len, qry_len, 0, 0, cig, cap, md, cap,
                                        nm.ctypes.data, len(nm), C.byref(txt)) == 0
        want = nm[:txt.nm_count]
        assert got.shape == want.shape and np.array_equal(got, want), (i, ops[:12], np.argwhere(got != want)[:3] if got.shape == want.shape else (got.shape, want.shape))
        total += len(got)
    assert total > 5000
    assert lib.cvx_nm_profile_ops(hip_aligner.h, 1, res, arena.ctypes.data, 3, off.ctypes.data, None, 0) == -3      # ops outside the arena
