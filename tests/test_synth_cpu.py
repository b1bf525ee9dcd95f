"""CPU: the corridor constructors of the tile generator against the reference's formulas
(values worked out from src/AlignmentBuffer.cpp and pinned by the recorded tiles)."""
import numpy as np

from ngmlr_amd import synth
from ngmlr_amd.shard import shard_tiles
from tests import util


def test_anchor_corridor_no_scatter_is_309_wide():
    off, ln = synth.corridor_anchors(1000, 1000)
    assert int(ln[0]) == 309 and (ln == 309).all()
    # left = 153.6, right = 156.16: offset[y] = (int)(y/k - 156.16f)
    assert off[0] == -156 and off[157] == 0 and off[999] == 999 - 157 + 0


def test_recorded_long_read_tiles_use_anchor_formula():
    """Tiles the real pipeline produced (H > 256) have constant width and offsets that follow
    (int)(y/k - right) for some right: consecutive differences are 0/1/2 and monotone."""
    n = 0
    for name in ("ref_test_3.npz", "ref_test_4.npz"):
        for t, _ in util.load_golden(name):
            d = np.diff(t.row_offset)
            assert (t.row_length == t.row_length[0]).all()
            assert d.min() >= 0
            n += 1
    assert n > 20


def test_short_read_tiles_match_linear_corridor():
    for t, _ in util.load_golden("ref_test_2.npz"):
        if t.H <= 256:
            w = int(t.row_length[0])
            off, ln = synth.corridor_linear(t.H, w)
            assert np.array_equal(off, t.row_offset) and np.array_equal(ln, t.row_length)
            # shortReadCorridor (:2585) clamped to 2*refSeqLen (:265-266); refSeqLen = strlen + 1
            assert w == min(256 + 2 * int(np.float32(0.15) * np.float32(t.H)), 2 * (t.W + 1))


def test_full_corridor():
    off, ln = synth.corridor_full(10, 1000)
    assert off[0] == -200 and ln[0] == 1200


def test_generator_is_seeded():
    a = synth.workload_pacbio(3, seed=5, read_len=2000)
    b = synth.workload_pacbio(3, seed=5, read_len=2000)
    assert all(x.ref == y.ref and x.qry == y.qry and np.array_equal(x.row_offset, y.row_offset) for x, y in zip(a, b))
    w = [int(t.row_length[0]) for t in synth.workload_pacbio(50, seed=1, read_len=2000)]
    assert 309 <= min(w) and max(w) <= 375


def test_shard_balances_cells_and_keeps_groups():
    rng = np.random.default_rng(0)
    cells = rng.integers(1, 1000, size=200).tolist()
    parts = shard_tiles(cells, 8)
    assert sorted(sum(parts, [])) == list(range(200))
    loads = [sum(cells[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(cells)
    groups = [i // 4 for i in range(200)]
    parts = shard_tiles(cells, 4, group=groups)
    for p in parts:
        for i in p:
            assert all(j in p for j in range(200) if groups[j] == groups[i])
