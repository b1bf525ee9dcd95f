"""GPU (-m gpu): candidate search on the device (cvx_index_upload / cvx_search_batch, SURVEY 8 f4 search half) against
every candidate-search call recorded from the unmodified reference on its test_3 reads -- same LocationScore entries in the
same order -- and against the CPU checker on the corners the recordings do not reach (N runs, the retry ladder of vote-table
sizes, empty and too-short reads)."""
import os

import numpy as np
import pytest

from ngmlr_amd.aligner import KmerIndex
from oracle.pyoracle import SearchFixture, SearchOracle
from tests import util

pytestmark = pytest.mark.gpu


def _same(got, loc, sc, rev):
    return got is not None and len(got) == len(loc) and np.array_equal(got["location"], loc) and np.array_equal(got["score"], sc) \
        and np.array_equal(got["reverse"], rev)


@pytest.mark.parametrize("which", ["sample", "full"])
def test_device_search_equals_recorded_reference_calls(hip_aligner, which):
    path = os.path.join(util.GOLDEN, "cs_test_3.npz") if which == "sample" else util.full_golden_path("cs_test_3_full.npz")
    if path is None:
        pytest.skip("oracle/_ref/golden_full/cs_test_3_full.npz not generated")
    fx = SearchFixture(path)
    idx, locs = fx.index_arrays()
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, fx.unit_offset)
    try:
        got = ix.search(fx.seqs)
    finally:
        ix.free()
    bad = [i for i in range(len(fx.seqs)) if not _same(got[i], *fx.want[i])]
    assert not bad, (len(bad), bad[:5])
    assert sum(len(g) for g in got) == sum(len(w[0]) for w in fx.want) > 1000


def test_device_search_corners_against_the_checker(hip_aligner):
    fx, reads = util.synthetic_search_case()
    o = SearchOracle(fx)
    want = [o.search(r, cap=1 << 20) for r in reads]
    o.close()
    idx, locs = fx.index_arrays()
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, 0)
    try:
        got = ix.search(reads)
        got2 = ix.search(reads, sensitivity=0.5, min_kmer_hits=2.0, bin_shift=2)
    finally:
        ix.free()
    assert max(w["table_bits"] for w in want if w["n"] >= 0) > 16            # the retry ladder was climbed
    for i, (w, g) in enumerate(zip(want, got)):
        if w["n"] < 0:
            assert g is None, i
        else:
            assert _same(g, w["loc"], w["score"], w["rev"]), (i, len(reads[i]), w["n"], None if g is None else len(g))
    o = SearchOracle(fx)
    for i, r in enumerate(reads):
        w = o.search(r, sensitivity=0.5, min_hits=2.0, bin_shift=2, cap=1 << 20)
        assert (w["n"] < 0 and got2[i] is None) or _same(got2[i], w["loc"], w["score"], w["rev"]), i
    o.close()
