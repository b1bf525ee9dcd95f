"""CPU: the product's host text stage (cvx_format_alignment: CIGAR, MD, NM, identity,
per-position mismatch profile, N-clip flags) against the oracle and the recorded
reference outputs, fed with the oracle's run-length ops."""
import ctypes as C

import numpy as np
import pytest

from oracle.pyoracle import same_alignment
from tests import util


def _format_from_oracle(port_oracle, tile):
    from ngmlr_amd import capi
    from ngmlr_amd.aligner import format_alignment
    want = port_oracle.align(tile)
    if want["ret"] < 0:
        return want, None
    f = port_oracle.last_fwd()
    ops = port_oracle.last_ops()
    r = capi.CvxResult()
    r.score = want["score"]
    r.status = 0
    r.ref_position, r.qstart, r.qend = f["ref_position"], f["qstart"], f["qend"]
    r.n_ops, r.ops_begin = len(ops), 0
    return want, format_alignment(capi.load(), r, ops if len(ops) else np.zeros(1, np.uint32), tile)


def test_format_matches_oracle_on_zoo(port_oracle):
    n = 0
    for t in util.tile_zoo(seed=21, n=120) + util.edge_tiles():
        want, got = _format_from_oracle(port_oracle, t)
        if got is None:
            continue
        assert same_alignment(want, got) is None, (t.tag, same_alignment(want, got))
        n += 1
    assert n > 60


@pytest.mark.parametrize("name", ["ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz"])
def test_format_matches_recorded_reference(port_oracle, name):
    for t, exp in util.load_golden(name):
        want, got = _format_from_oracle(port_oracle, t)
        if got is not None:
            assert util.golden_diff(exp, got) is None, t.tag


def test_invalid_tile_formats_as_minus_one(built):
    from ngmlr_amd import capi
    lib = capi.load()
    r = capi.CvxResult()
    r.status = 2
    r.score = 123.0
    txt = capi.CvxAlignmentText()
    cig = C.create_string_buffer(16)
    md = C.create_string_buffer(16)
    assert lib.cvx_format_alignment(C.byref(r), None, b"ACGT", 4, 4, 0, 0, cig, 16, md, 16, None, 0, C.byref(txt)) == 0
    assert txt.ret == -1 and txt.score == -1.0 and cig.value == b""


def test_text_truncation_reports_full_length(port_oracle):
    from ngmlr_amd import capi
    t = util.tile_zoo(seed=3, n=1, max_w=900)[0]
    want = port_oracle.align(t)
    assert want["ret"] >= 0
    f, ops = port_oracle.last_fwd(), port_oracle.last_ops()
    r = capi.CvxResult()
    r.status = 0; r.ref_position, r.qstart, r.qend = f["ref_position"], f["qstart"], f["qend"]
    r.n_ops, r.ops_begin = len(ops), 0
    txt = capi.CvxAlignmentText()
    cig = C.create_string_buffer(8)
    md = C.create_string_buffer(8)
    assert capi.load().cvx_format_alignment(C.byref(r), ops.ctypes.data, t.ref, t.W, t.H, t.ext_qstart, t.ext_qend, cig, 8, md, 8, None, 0, C.byref(txt)) == 0
    assert txt.cigar_len == len(want["cigar"]) and txt.md_len == len(want["md"])
    assert cig.value == want["cigar"][:7].encode()


def test_threaded_batch_text_stage_equals_single_calls(port_oracle):
    """cvx_format_batch (threaded host text stage) == cvx_format_alignment per tile."""
    from ngmlr_amd import capi
    lib = capi.load()
    tiles = [t for t in util.tile_zoo(seed=55, n=70, max_w=1500)]
    res = (capi.CvxResult * len(tiles))()
    arena = []
    want = []
    for i, t in enumerate(tiles):
        w = port_oracle.align(t)
        want.append(w)
        if w["ret"] < 0:
            res[i].status = 2
            continue
        f, ops = port_oracle.last_fwd(), port_oracle.last_ops()
        res[i].status = 0
        res[i].score = w["score"]
        res[i].ref_position, res[i].qstart, res[i].qend = f["ref_position"], f["qstart"], f["qend"]
        res[i].n_ops = len(ops)
        res[i].ops_begin = sum(len(a) for a in arena)
        arena.append(ops)
    ops_all = np.concatenate(arena).astype(np.uint32)
    ct = (capi.CvxTile * len(tiles))()
    bufs = (capi.CvxTextBuffers * len(tiles))()
    keep = []
    for i, t in enumerate(tiles):
        ct[i].ref, ct[i].qry, ct[i].ref_len, ct[i].qry_len = t.ref, t.qry, t.W, t.H
        cap = 4 * t.H + 64
        cig, md = C.create_string_buffer(cap), C.create_string_buffer(cap)
        nm = np.zeros((2 * (t.H + 1) + t.W + 16, 3), dtype=np.int32)
        keep.append((cig, md, nm))
        bufs[i].cigar, bufs[i].md, bufs[i].nm_triples = C.addressof(cig), C.addressof(md), nm.ctypes.data
        bufs[i].cigar_cap = bufs[i].md_cap = cap
        bufs[i].nm_cap = len(nm)
        bufs[i].ext_qstart, bufs[i].ext_qend = t.ext_qstart, t.ext_qend
    out = (capi.CvxAlignmentText * len(tiles))()
    for threads in (1, 4):
        assert lib.cvx_format_batch(len(tiles), res, ops_all.ctypes.data, ct, bufs, out, threads) == 0
        for i, t in enumerate(tiles):
            w = want[i]
            if w["ret"] < 0:
                assert out[i].ret == -1
                continue
            assert out[i].ret == w["ret"] and keep[i][0].value.decode() == w["cigar"] and keep[i][1].value.decode() == w["md"]
            assert out[i].nm == w["nm"] and out[i].position_offset == w["position_offset"]
            n = w["alignment_length"]
            assert np.array_equal(keep[i][2][:n], w["nm_per_position"])


def test_n_clip_flags_fire_like_the_reference(port_oracle, ref_oracle):
    """svType |= 0x1 when an alignment ends next to a run of 'X' in the reference window (src/ConvexAlignFast.cpp:493-528):
    positive and negative cases around the 80 % threshold -- the reference itself, the C restatement, and the product's host
    text stage (cvx_format_alignment) fed with the restatement's ops."""
    n_pos = 0
    for t, flag in util.nclip_tiles():
        want = ref_oracle.align(t)
        assert want["ret"] >= 0 and want["sv_type"] == flag, (t.tag, want["sv_type"])
        port = port_oracle.align(t)
        assert same_alignment(want, port) is None, (t.tag, same_alignment(want, port))
        _, got = _format_from_oracle(port_oracle, t)
        assert got is not None and same_alignment(want, got) is None, (t.tag, same_alignment(want, got))
        assert got["sv_type"] == flag
        n_pos += flag
    assert n_pos >= 4


def test_profile_of_an_alignment_longer_than_the_callers_buffer(built):
    """Cheap gaps (match 3, mismatch -3, gaps -1 ... -0.5) turn a short read into an alignment with more columns than the
    (read length + 1) * 2 entries the reference's caller allocates for nmPerPosition (src/AlignmentBuffer.cpp:277): the consumer
    walks alignmentLength entries of that buffer (:1320), addPosition doubles it only when the written triples do not fit
    (src/ConvexAlignFast.cpp:79-92).  The text stage and the checker must agree on exactly that view -- found by
    tools/fuzz_parity.py's scoring sweep, where the harness compared buffers of different capacity."""
    from oracle.pyoracle import Oracle, have_ref
    from ngmlr_amd import synth
    params = (3.0, -3.0, -1.0, -1.0, -0.5, 0.15)
    port = Oracle("port", params)
    ref = Oracle("reference", params) if have_ref() else None
    tiles = synth.workload_short(100, seed=9000)
    longer = 0
    for t in tiles:
        want, got = _format_from_oracle(port, t)
        if got is None:
            continue
        assert same_alignment(want, got) is None, (t.tag, t.H, same_alignment(want, got))
        if ref is not None:
            assert same_alignment(ref.align(t), got) is None, (t.tag, t.H)
        longer += got["alignment_length"] > 2 * (t.H + 1)
    assert longer >= 3
