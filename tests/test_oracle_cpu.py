"""CPU: the oracle (plain-C restatement) against the reference itself.

1. golden tiles recorded from the unmodified reference binary on its own test data,
2. the reference's ConvexAlignFast compiled from /root/reference (oracle/_ref) on seeded tiles,
3. SURVEY.md Appendix D known answers,
4. scalar-spec fill == SSE-path fill under the default scoring (Appendix A)."""
import numpy as np
import pytest

from oracle.pyoracle import Oracle, same_alignment
from tests import util


@pytest.mark.parametrize("name", ["ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz"])
def test_port_matches_recorded_reference_pipeline(port_oracle, name):
    n_valid = 0
    for tile, exp in util.load_golden(name):
        got = port_oracle.align(tile)
        assert util.golden_diff(exp, got) is None, (tile.tag, util.golden_diff(exp, got))
        n_valid += exp["ret"] >= 0
    assert n_valid > 0


def test_appendix_d_known_answers(port_oracle):
    """test_2 reads "100bp".."400bp" (99/198/297/396 bases): CIGAR, AS, NM, MD
    as printed by the unmodified reference binary (tests/golden/test_2.sam)."""
    want = {99: ("8M1I4M1I25M1D8M1I51M", 172.0, 4, "37^G59"),
            198: ("8M1I4M1I25M1D8M1I150M", 370.0, 4, "37^G158"),
            297: ("8M1I4M1I25M1D8M1I249M", 568.0, 4, "37^G257"),
            396: ("8M1I4M1I25M1D8M1I253M1D58M1D10M1I5M1D16M1D3M2S", 735.0, 9, "37^G261^T58^T15^G16^G3")}
    seen = set()
    for tile, exp in util.load_golden("ref_test_2.npz"):
        if tile.H in want and exp["ret"] >= 0 and tile.H not in seen:
            cigar, score, nm, md = want[tile.H]
            got = port_oracle.align(tile)
            assert (got["cigar"], got["score"], got["nm"], got["md"]) == (cigar, score, nm, md)
            assert exp["cigar"] == cigar
            seen.add(tile.H)
    assert seen == set(want)


def test_port_equals_reference_on_seeded_tiles(port_oracle, ref_oracle):
    bad = []
    tiles = util.tile_zoo(seed=5, n=150) + util.edge_tiles()
    n_invalid = 0
    for t in tiles:
        a, b = port_oracle.align(t), ref_oracle.align(t)
        d = same_alignment(a, b)
        n_invalid += b["ret"] < 0
        if d:
            bad.append((t.tag, t.H, t.W, d))
    assert not bad, bad[:5]
    assert 0 < n_invalid < len(tiles)      # both outcomes exercised


def test_port_equals_reference_long_tiles(port_oracle, ref_oracle):
    from ngmlr_amd import synth
    for t in synth.workload_pacbio(3, seed=3, read_len=6000) + synth.workload_ont(6, seed=4, max_len=6000):
        assert same_alignment(port_oracle.align(t), ref_oracle.align(t)) is None


def test_spec_fill_equals_sse_fill_default_scoring(built):
    """Appendix A: with gap_open + gap_ext_min < mismatch the SSE path's relaxed tests are
    unreachable, so the scalar recurrence (what the HIP kernel implements) is the spec."""
    sse = Oracle("port")
    spec = Oracle("port")
    spec.set_spec_fill(True)
    for t in util.tile_zoo(seed=8, n=60) + util.edge_tiles():
        assert same_alignment(sse.align(t), spec.align(t)) is None, t.tag


def test_exotic_scoring_is_where_spec_and_sse_diverge(built):
    """--mismatch -10 violates the inequality: the two fills disagree;
    the product therefore refuses such parameters (see test_capi_cpu)."""
    p = (2.0, -10.0, -5.0, -5.0, -1.0, 0.15)
    sse, spec = Oracle("port", p), Oracle("port", p)
    spec.set_spec_fill(True)
    differ = sum(same_alignment(sse.align(t), spec.align(t)) is not None for t in util.tile_zoo(seed=9, n=40))
    assert differ > 0


def test_exotic_scoring_port_still_tracks_reference(ref_oracle, built):
    from oracle.pyoracle import have_ref
    p = (2.0, -10.0, -5.0, -5.0, -1.0, 0.15)
    a, b = Oracle("port", p), Oracle("reference", p)
    for t in util.tile_zoo(seed=10, n=40):
        assert same_alignment(a.align(t), b.align(t)) is None, t.tag
