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


@pytest.fixture(params=["wave", "wave_sorted", "wave_small", "wave_slot12", "wave_slot8", "wave_hbm", "lane"])
def search_kernel(request, monkeypatch):
    """The device kernels of the vote: one wave per read casting 64 votes at a time, with the vote map in LDS (the default; reads
    it cannot hold fall back per read) or with the real table in HBM (CVX_TUNE_SEARCH_WAVE=2 sends every read there), and one lane
    per read casting its votes one by one over a table in HBM (=0: an independent implementation of the same contract).  The LDS
    form three ways: as a small call takes it (one map size for every read), as a large call does (CVX_TUNE_SEARCH_CLASSIFY=0:
    votes counted first, reads sorted into launches by map size, 2^9 .. 2^12 slots), and with the smallest map forced on every
    read (CVX_TUNE_SEARCH_LOG2=9: most reads overflow it and are redone over the table in HBM); and with the 12-byte and the
    8-byte form of the map forced (CVX_TUNE_SEARCH_SLOT8)."""
    if request.param == "wave_sorted":
        monkeypatch.setenv("CVX_TUNE_SEARCH_CLASSIFY", "0")
    if request.param == "wave_small":
        monkeypatch.setenv("CVX_TUNE_SEARCH_LOG2", "9")
    if request.param == "wave_slot12":      # the 12-byte map everywhere (the default takes the 8-byte one for sub-reads at table sizes up to 2^16)
        monkeypatch.setenv("CVX_TUNE_SEARCH_SLOT8", "0")
        monkeypatch.setenv("CVX_TUNE_SEARCH_CLASSIFY", "0")
    if request.param == "wave_slot8":       # the 8-byte map wherever its 16-bit virtual slots allow, long reads included (counts beyond 255 fall back)
        monkeypatch.setenv("CVX_TUNE_SEARCH_SLOT8", "1")
        monkeypatch.setenv("CVX_TUNE_SEARCH_CLASSIFY", "0")
    if request.param == "wave_hbm":
        monkeypatch.setenv("CVX_TUNE_SEARCH_WAVE", "2")
    if request.param == "lane":
        monkeypatch.setenv("CVX_TUNE_SEARCH_WAVE", "0")
    return request.param


@pytest.mark.parametrize("which", ["sample", "full"])
def test_device_search_equals_recorded_reference_calls(hip_aligner, which, search_kernel):
    path = os.path.join(util.GOLDEN, "cs_test_3.npz") if which == "sample" else util.full_golden_path("cs_test_3_full.npz")
    if path is None:
        pytest.skip("oracle/_ref/golden_full/cs_test_3_full.npz not generated")
    fx = SearchFixture(path)
    idx, locs = fx.index_arrays()
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, fx.unit_offset)
    try:
        got, max_hit, misses = ix.search(fx.seqs, extras=True)
        got12 = ix.search(fx.seqs[:600], first_bits=12)        # CS::c_SrchTableBitLen adapted down (src/CS.cpp:482-489): same lists
        got8 = ix.search(fx.seqs[:600], first_bits=8)          # ... and a table so small that every read climbs the ladder
    finally:
        ix.free()
    bad = [i for i in range(len(fx.seqs)) if not _same(got[i], *fx.want[i])]
    assert not bad, (len(bad), bad[:5])
    assert sum(len(g) for g in got) == sum(len(w[0]) for w in fx.want) > 1000
    assert all(_same(got12[i], *fx.want[i]) for i in range(len(got12)))
    assert all(_same(got8[i], *fx.want[i]) for i in range(len(got8)))
    # maxHitNumber as the reference recorded it (MappedRead::s)
    assert np.array_equal(max_hit, fx.max_hit.astype(np.float32))
    assert int(misses.max()) > 0
    # kCount as the reference recorded it: summed over the attempts of the ladder, so every read is searched with the table size
    # the reference's first attempt had (adapted per batch, src/CS.cpp:482-489: 2^8 .. 2^16 in this recording)
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, fx.unit_offset)
    try:
        for b in sorted(set(int(x) for x in fx.first_bits)):
            grp = [i for i in range(len(fx.seqs)) if int(fx.first_bits[i]) == b]
            g_lists, g_max, g_miss = ix.search([fx.seqs[i] for i in grp], first_bits=b, extras=True)
            assert all(_same(g_lists[j], *fx.want[i]) for j, i in enumerate(grp)), b
            assert [int(x) for x in g_miss] == [int(fx.kmer_misses[i]) for i in grp], b
            assert np.array_equal(g_max, fx.max_hit[grp].astype(np.float32)), b
    finally:
        ix.free()


def test_device_search_on_a_repeat_rich_reference(hip_aligner, search_kernel, capfd, monkeypatch):
    """Recorded from the unmodified reference on a repeat-rich reference (tools/make_golden_cs.sh, round 5): sub-reads with thousands
    of votes, hundreds of listed bins, reads from a 400-bp unit with 700 copies -- more bins than the wave kernel's LDS vote map
    holds, so the same attempt is redone over the real table in HBM and the rest of the ladder follows there.  All three kernel
    forms, lists / maxHitNumber at 2^16 and lists + kCount at the table size the reference's first attempt had."""
    import re
    fx = SearchFixture(os.path.join(util.GOLDEN, "cs_rep.npz"))
    idx, locs = fx.index_arrays()
    monkeypatch.setenv("CVX_SEARCH_TRACE", "1")
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, fx.unit_offset)
    try:
        got, max_hit, misses = ix.search(fx.seqs, extras=True)
        bad = [i for i in range(len(fx.seqs)) if not _same(got[i], *fx.want[i])]
        assert not bad, (len(bad), bad[:5])
        assert np.array_equal(max_hit, fx.max_hit.astype(np.float32))
        for b in sorted(set(int(x) for x in fx.first_bits)):
            grp = [i for i in range(len(fx.seqs)) if int(fx.first_bits[i]) == b]
            g_lists, g_max, g_miss = ix.search([fx.seqs[i] for i in grp], first_bits=b, extras=True)
            assert all(_same(g_lists[j], *fx.want[i]) for j, i in enumerate(grp)), b
            assert [int(x) for x in g_miss] == [int(fx.kmer_misses[i]) for i in grp], b
    finally:
        ix.free()
    err = capfd.readouterr().err
    if search_kernel == "wave_sorted":
        sizes = set(int(x) for x in re.findall(r"2\^(\d+)-slot maps", err))
        assert len(sizes) >= 3, "the reads of this recording should spread over the map sizes\n" + err[-600:]
    if search_kernel == "wave_small":
        to_hbm = sum(int(x) for x in re.findall(r"wave \d+ -> (\d+) to hbm", err))
        assert to_hbm >= 10, "no sub-read overflowed the LDS vote map: the transition to the HBM-table form was not exercised\n" + err[-600:]


def test_device_search_corners_against_the_checker(hip_aligner, search_kernel):
    fx, reads = util.synthetic_search_case()
    o = SearchOracle(fx)
    want = [o.search(r, cap=1 << 20) for r in reads]
    o.close()
    idx, locs = fx.index_arrays()
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, 0)
    try:
        got, max_hit, misses = ix.search(reads, extras=True)
        got2 = ix.search(reads, sensitivity=0.5, min_kmer_hits=2.0, bin_shift=2)
        got10, max_hit10, misses10 = ix.search(reads, first_bits=10, extras=True)
        got8, max_hit8, misses8 = ix.search(reads, first_bits=8, extras=True)
    finally:
        ix.free()
    # what CS::RunRead leaves behind beside the list: maxHitNumber, and kCount summed over the attempts of the ladder
    assert [int(m) for m in misses] == [w["kmer_misses"] for w in want]
    assert [float(m) for m in max_hit] == [float(np.float32(w["max_hit"])) for w in want]
    o10 = SearchOracle(fx)
    want10 = [o10.search(r, cap=1 << 20, first_bits=10) for r in reads]
    o10.close()
    for i, (w, g) in enumerate(zip(want10, got10)):
        assert (w["n"] < 0 and g is None) or _same(g, w["loc"], w["score"], w["rev"]), i
    assert [int(m) for m in misses10] == [w["kmer_misses"] for w in want10]
    # 256 slots for a few hundred bins: probe paths run into each other all the time (two votes of one 64-vote batch opening the same
    # slot for different bins, budgets running out in mid-batch) -- the wave kernels' vote-by-vote path
    o8 = SearchOracle(fx)
    want8 = [o8.search(r, cap=1 << 20, first_bits=8) for r in reads]
    o8.close()
    for i, (w, g) in enumerate(zip(want8, got8)):
        assert (w["n"] < 0 and g is None) or _same(g, w["loc"], w["score"], w["rev"]), i
    assert [int(m) for m in misses8] == [w["kmer_misses"] for w in want8]
    assert [float(m) for m in max_hit8] == [float(np.float32(w["max_hit"])) for w in want8]
    assert max(w["kmer_misses"] for w in want) >= 200 and any(w["table_bits"] > 16 and w["kmer_misses"] > 0 for w in want)   # foreign k-mers, also on a read that climbed the ladder
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


def test_device_walk_over_reads_with_N(hip_aligner, search_kernel):
    """The wave kernels enumerate a read's k-mers from a closed form of CS::PrefixIteration, 64 window positions at a time (round 6;
    tests/test_search_cpu.py pins the form against the checker's serial walk): 1 500 variants of a read the table knows with 'N's
    sprinkled in, the tail patterns the reference treats specially forced often -- lists, maxHitNumber and kCount against the
    checker, in every kernel form."""
    from tests.test_search_cpu import n_pattern_reads
    fx, base_reads = util.synthetic_search_case()
    rng = np.random.default_rng(9)
    reads = n_pattern_reads(rng, base_reads[0], 1500)
    o = SearchOracle(fx)
    want = [o.search(r, cap=1 << 12) for r in reads]
    o.close()
    idx, locs = fx.index_arrays()
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, 0)
    try:
        got, max_hit, misses = ix.search(reads, extras=True)
    finally:
        ix.free()
    bad = [i for i, (w, g) in enumerate(zip(want, got)) if not (_same(g, w["loc"], w["score"], w["rev"]) and int(misses[i]) == w["kmer_misses"]
                                                                 and float(max_hit[i]) == float(np.float32(w["max_hit"])))]
    assert not bad, (len(bad), [(reads[i], want[i]["n"], want[i]["kmer_misses"], int(misses[i])) for i in bad[:3]])
    assert sum(1 for r in reads if b"N" in r) > 500 and sum(w["n"] for w in want) > 500


@pytest.mark.parametrize("unit_offset", [1 << 34, (1 << 35) - 4096, 1 << 37])
def test_bins_beyond_the_lds_maps_words(hip_aligner, search_kernel, unit_offset):
    """A table unit far into a large genome (TableUnit::Offset): the bins of its votes pass 31 and 32 bits -- more than the 8-byte
    and the 12-byte LDS map keep per slot, so those reads must come out of the table in HBM with the reference's lists, not
    truncated ones."""
    fx, reads = util.synthetic_search_case()
    fx.unit_offset = unit_offset
    reads = [r for r in reads if len(r) <= 300][:8] + [reads[0][:200], reads[0][30:]]
    o = SearchOracle(fx)
    want = [o.search(r, cap=1 << 16) for r in reads]
    o.close()
    idx, locs = fx.index_arrays()
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, unit_offset)
    try:
        got, max_hit, misses = ix.search(reads, extras=True)
    finally:
        ix.free()
    for i, (w, g) in enumerate(zip(want, got)):
        assert (w["n"] < 0 and g is None) or _same(g, w["loc"], w["score"], w["rev"]), (i, w["n"], None if g is None else len(g))
    assert [int(m) for m in misses] == [w["kmer_misses"] for w in want]
    assert max(int(w["loc"].max()) for w in want if w["n"] > 0) >> 4 >= (1 << 30)


@pytest.mark.parametrize("pinned", [False, True])
def test_arena_form_returns_the_recorded_lists(hip_aligner, pinned, search_kernel):
    """cvx_search_batch_arena (ABI 7): the reads back to back in ONE block -- pageable, or page-locked memory from cvx_host_alloc that
    the device pulls as it is -- against the recorded reference calls of test_3 and of the repeat-rich recording, all three kernel
    forms: lists, order, maxHitNumber, kCount as through the string form."""
    for name in ("cs_test_3.npz", "cs_rep.npz"):
        fx = SearchFixture(os.path.join(util.GOLDEN, name))
        idx, locs = fx.index_arrays()
        ix = KmerIndex(hip_aligner, fx.k, idx, locs, fx.unit_offset)
        arena, offsets, pin = KmerIndex.make_arena(fx.seqs, hip_aligner.lib if pinned else None)
        assert (pin is not None) == pinned
        try:
            want_lists, want_mh, want_ms = ix.search(fx.seqs, extras=True)
            ncand, begin, cands, mh, ms = ix.search_arena(arena, offsets)
        finally:
            ix.free()
            if pin:
                pin[0].cvx_host_free(pin[1])
        bad = [i for i in range(len(fx.seqs))
               if not _same(cands[int(begin[i]):int(begin[i]) + max(int(ncand[i]), 0)] if ncand[i] >= 0 else None, *fx.want[i])]
        assert not bad, (name, len(bad), bad[:5])
        assert np.array_equal(mh, want_mh) and np.array_equal(ms, want_ms)
        assert sum(len(g) for g in want_lists if g is not None) == int(ncand[ncand > 0].sum())
    # a block that does not end its reads with a NUL is refused
    arena, offsets, _ = KmerIndex.make_arena([b"ACGT" * 40, b"TTGA" * 50])
    arena2 = arena.copy()
    arena2[int(offsets[1]) - 1] = ord("A")
    fx = SearchFixture(os.path.join(util.GOLDEN, "cs_test_3.npz"))
    idx, locs = fx.index_arrays()
    ix = KmerIndex(hip_aligner, fx.k, idx, locs, fx.unit_offset)
    try:
        with pytest.raises(Exception):
            ix.search_arena(arena2, offsets)
        n, _, _, _, _ = ix.search_arena(arena, offsets)
        assert len(n) == 2
    finally:
        ix.free()


def test_device_search_on_recorded_calls_at_genome_scale(hip_aligner, search_kernel):
    """The device against the unmodified reference where the table is a gigabyte in HBM (tests/golden/cs_big.npz, 840 recorded RunRead
    calls on a 512 Mbp reference; the table rebuilt by cvx_index_build and checked against the recording's hashes): every list,
    its order, maxHitNumber and kCount, in all three kernel forms, through the string form and the arena form."""
    fx, idx5, locs = util.big_search_case()
    ix = KmerIndex(hip_aligner, fx.k, idx5.view(np.dtype([("tab", "<u4"), ("rc", "i1")])), locs, fx.unit_offset)
    try:
        got, max_hit, _ = ix.search(fx.seqs, extras=True)
        arena, offsets, _pin = KmerIndex.make_arena(fx.seqs)
        ncand, begin, cands, mh2, _ = ix.search_arena(arena, offsets)
        miss = np.zeros(len(fx.seqs), dtype=np.int32)
        for b in sorted(set(int(x) for x in fx.first_bits)):      # kCount is summed over the ladder: at the table size the reference's first attempt had
            sel = [i for i in range(len(fx.seqs)) if int(fx.first_bits[i]) == b]
            _, _, ms = ix.search([fx.seqs[i] for i in sel], first_bits=b, extras=True)
            miss[sel] = ms
    finally:
        ix.free()
    bad = [i for i in range(len(fx.seqs)) if not _same(got[i], *fx.want[i])]
    assert not bad, (len(bad), bad[:5])
    assert all(_same(cands[int(begin[i]):int(begin[i]) + int(ncand[i])] if ncand[i] >= 0 else None, *fx.want[i]) for i in range(len(fx.seqs)))
    assert np.array_equal(max_hit, fx.max_hit.astype(np.float32)) and np.array_equal(mh2, max_hit)
    assert np.array_equal(miss, fx.kmer_misses.astype(np.int32))
