"""GPU (-m gpu): corridors in closed form (cvx_tile.corridor_kind) -- the device evaluates the reference's corridor
builders (src/AlignmentBuffer.cpp:68-197) itself instead of receiving their rows.  The rows the DEVICE generates
must equal, bit for bit, every corridor the unmodified reference was recorded building and every corridor of the
synthetic generators; alignments from closed forms must equal alignments from the row arrays and the oracle; and
sequences that already sit in a page-locked arena (cvx_host_alloc) must give the same results without being packed."""
import numpy as np
import pytest

from ngmlr_amd import synth
from oracle.pyoracle import same_alignment
from tests import util

pytestmark = pytest.mark.gpu


def _device_rows_equal(al, t, desc):
    probe = synth.Tile(ref=b"", qry=t.qry, row_offset=t.row_offset, row_length=t.row_length, desc=desc)
    off, ln = al.corridor_rows(probe)
    return np.array_equal(off, t.row_offset) and np.array_equal(ln, t.row_length)


@pytest.mark.parametrize("name", ["ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz", "full"])
def test_device_rows_equal_every_recorded_corridor(hip_aligner, name):
    if name == "full":
        name = util.full_golden_path()
        if name is None:
            pytest.skip("oracle/_ref/golden_full not generated")
    n = 0
    for t, _ in util.load_golden(name):
        d = util.fit_corridor(t.row_offset, t.row_length, t.H, t.W)
        assert d is not None, t.tag
        assert _device_rows_equal(hip_aligner, t, d), (t.tag, d)
        n += 1
    assert n > 0


def test_device_rows_equal_the_generators(hip_aligner):
    rng = np.random.default_rng(77)
    tiles = util.tile_zoo(n=60) + synth.workload_short(30) + synth.workload_ont(40, max_len=20000) + synth.workload_ultralong_sv(8)
    tiles += [synth.make_tile(rng, 10000, corridor="anchors", scatter=25.0) for _ in range(8)]
    for t in tiles:
        assert _device_rows_equal(hip_aligner, t, t.desc), (t.tag, t.desc)


def test_alignments_from_closed_forms(hip_aligner, port_oracle):
    rng = np.random.default_rng(78)
    tiles = util.tile_zoo(n=48) + [synth.make_tile(rng, 6000, corridor="anchors", scatter=25.0) for _ in range(6)]
    tiles += synth.workload_ultralong_sv(4, read_len=12000)       # wide corridors: chained blocks plan from the host's evaluation
    got_rows = hip_aligner.batch_align(tiles)
    got_cf = hip_aligner.batch_align(tiles, closed_form=True)
    for t, a, b in zip(tiles, got_rows, got_cf):
        assert same_alignment(a, b) is None, (t.tag, same_alignment(a, b))
        assert same_alignment(port_oracle.align(t), b) is None, t.tag


def test_golden_alignments_from_fitted_closed_forms(hip_aligner):
    pairs = util.load_golden("ref_test_3.npz") + util.load_golden("ref_test_2.npz") + util.load_golden("ref_test_4.npz")
    tiles = []
    for t, _ in pairs:
        d = util.fit_corridor(t.row_offset, t.row_length, t.H, t.W)
        tiles.append(synth.Tile(ref=t.ref, qry=t.qry, row_offset=t.row_offset, row_length=t.row_length,
                                ext_qstart=t.ext_qstart, ext_qend=t.ext_qend, tag=t.tag, desc=d))
    got = hip_aligner.batch_align(tiles, closed_form=True)
    for (t, exp), g in zip(pairs, got):
        assert util.golden_diff(exp, g) is None, (t.tag, util.golden_diff(exp, g))


def test_page_locked_arena_travels_without_packing(hip_aligner):
    """A TileSet whose sequences sit back to back in cvx_host_alloc memory, corridors in closed form: nothing is
    packed on the host; results equal those of the same tiles handed over one by one with row arrays."""
    rng = np.random.default_rng(79)
    tiles = [synth.make_tile(rng, int(rng.integers(300, 4000)), corridor=c, scatter=20.0)
             for c in ("anchors", "endpoints", "linear", "full", "anchors", "anchors") for _ in range(8)]
    want = hip_aligner.batch_align(tiles, want_nm=False)
    ts = synth.tileset_from_tiles(tiles).use_closed_form()
    assert ts.pin(hip_aligner.lib), "cvx_host_alloc failed on a GPU box"
    try:
        for _ in range(2):           # second round: recycled batch arenas
            job = hip_aligner.submit(ts)
            res, ops = job.wait()
            for i, w in enumerate(want):
                r = res[i]
                assert int(r["status"]) == w["status"], tiles[i].tag
                if w["status"] == 0:
                    assert int(np.float32(r["score"]).view(np.uint32)) == w["fwd_score_bits"]
                    assert (int(r["best_ref_index"]), int(r["best_read_index"])) == (w["best_x"], w["best_y"])
                    assert int(r["ref_position"]) == w["position_offset"]
            txt = job.text()
            for i, w in enumerate(want):
                if w["ret"] >= 0:
                    assert txt[i]["cigar"] == w["cigar"] and txt[i]["md"] == w["md"], tiles[i].tag
            job.release()
    finally:
        ts.unpin()
