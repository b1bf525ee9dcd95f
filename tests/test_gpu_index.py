"""GPU (-m gpu): ngmlr's k-mer table built on the device (cvx_index_build_device, cvx_index.hip) against the host builder
(cvx_index_build, pinned byte for byte against the table the unmodified reference writes: tests/test_index_cpu.py) -- index
records and locations, byte for byte, on genomes made for the rules that shape the table."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from ngmlr_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _both(lib, contigs, k=13, skip=2, bin_shift=4):
    a = synth.kmer_table(lib, contigs, k=k, skip=skip, bin_shift=bin_shift)
    b = synth.kmer_table(lib, contigs, k=k, skip=skip, bin_shift=bin_shift, device=0)
    return a, b


def _same(a, b, what):
    assert len(a[1]) == len(b[1]), (what, len(a[1]), len(b[1]))
    if not np.array_equal(a[0], b[0]):
        d = int(np.nonzero(a[0] != b[0])[0][0]) // 5
        raise AssertionError("%s: index records differ at k-mer %d: host %s device %s" % (what, d, a[0][5 * d:5 * d + 5], b[0][5 * d:5 * d + 5]))
    if not np.array_equal(a[1], b[1]):
        d = int(np.nonzero(a[1] != b[1])[0][0])
        raise AssertionError("%s: locations differ at %d: host %s device %s" % (what, d, a[1][d:d + 4], b[1][d:d + 4]))


def _rand(rng, n):
    return rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)


def _prefix_bases(v, k=13):
    """the k bases whose 2-bit codes ((c >> 1) & 3: A 0, C 1, T 2, G 3) spell v"""
    lut = {0: ord("A"), 1: ord("C"), 2: ord("T"), 3: ord("G")}
    return np.array([lut[(v >> (2 * (k - 1 - i))) & 3] for i in range(k)], dtype=np.uint8)


def test_device_builder_equals_the_host_builder_on_rule_genomes(hip_aligner):
    lib = hip_aligner.lib
    rng = np.random.default_rng(17)
    cases = {}
    # plain random sequences of odd / even lengths, several of them, one too short to be kept (<= 10) and one shorter than a chunk
    cases["plain"] = [_rand(rng, n) for n in (50001, 4096, 8, 777, 123456, 13, 14, 2049)]
    # 'N's: single ones, runs, at the start, at the end, a run with exactly k (+2 bytes of tail) behind it, chunk boundaries
    g = _rand(rng, 30000)
    for p in rng.integers(0, 30000, size=60):
        g[int(p)] = ord("N")
    for a, r in ((100, 5), (2040, 20), (4090, 12), (9000, 3000), (29990, 4)):
        g[a:a + r] = ord("N")
    h = _rand(rng, 5000); h[:7] = ord("N"); h[-3:] = ord("N")
    tails = []
    for r, behind in ((2, 11), (1, 11), (3, 11), (2, 12), (2, 10), (5, 13)):      # the walk's buffer has two more bytes than bases: 11 bases + 2 = k
        t = _rand(rng, 400 + r + behind); t[400:400 + r] = ord("N"); tails.append(t)
    lead = [np.concatenate([np.full(r, ord("N"), np.uint8), _rand(rng, b)]) for r, b in ((1, 11), (1, 12), (4, 11), (4, 30))]
    cases["N"] = [g, h] + tails + lead
    # runs of equal k-mers at sampled positions: homopolymers, period-3 and period-1.5 microsatellites, across bins and 'N's
    m = _rand(rng, 20000)
    m[1000:1400] = ord("A"); m[3000:3300] = np.frombuffer(b"ACG" * 100, dtype=np.uint8); m[5000:5600] = np.frombuffer(b"AC" * 300, dtype=np.uint8)
    m[7000:7200] = ord("T"); m[7100] = ord("N"); m[9000:9090] = ord("G")
    first = np.concatenate([_prefix_bases(111111), _prefix_bases(111111), _rand(rng, 500)])      # lastPrefix starts a sequence at 111111
    first2 = np.concatenate([np.repeat(_prefix_bases(111111)[:1], 0), _prefix_bases(111111), _rand(rng, 300)])
    cases["runs"] = [m, first, first2, np.full(3000, ord("C"), np.uint8)]
    # the frequency cutoff: one k-mer (with its reverse complement) 985 / 995 / 1005 times, 16 bases apart so that every copy is its own bin
    def many(times):
        unit = np.concatenate([_rand(rng, 13), _rand(rng, 35)])
        unit[13:] = _rand(rng, 35)
        return np.concatenate([np.concatenate([unit[:13], _rand(rng, 35)]) for _ in range(times)])
    cases["cutoff"] = [many(985), many(995), many(1005), _rand(rng, 5000)]
    for what, contigs in cases.items():
        a, b = _both(lib, contigs)
        _same(a, b, what)
        assert len(a[1]) > 100, what
    # other parameters: k, --kmer-skip, --bin-size
    contigs = cases["N"] + cases["runs"]
    for k, skip, shift in ((13, 0, 4), (13, 1, 2), (11, 2, 4), (9, 3, 6), (15, 2, 4), (13, 5, 0)):
        a, b = _both(lib, contigs, k=k, skip=skip, bin_shift=shift)
        _same(a, b, (k, skip, shift))
    # the cutoff rows did what they are for: rows with slots and weight 0
    a, _ = _both(lib, cases["cutoff"])
    rec = a[0].reshape(-1, 5)
    tab = rec[:, :4].copy().view(np.uint32).ravel().astype(np.int64)
    assert int(((np.diff(tab) > 0) & (rec[:-1, 4] == 0)).sum()) >= 1


def test_device_builder_random_sweep(hip_aligner):
    lib = hip_aligner.lib
    rep = synth.big_reference(2 << 20, n_contigs=1, seed=5)[0]      # repeat families and microsatellites to cut pieces from
    for seed in range(12):
        rng = np.random.default_rng(1000 + seed)
        contigs = []
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.choice([30, 500, 2047, 2048, 2049, 6000, 40000]))
            if rng.random() < 0.5:
                at = int(rng.integers(0, len(rep) - n))
                c = rep[at:at + n].copy()
            else:
                c = _rand(rng, n)
            pn = float(rng.choice([0.0, 0.001, 0.02]))
            c[rng.random(n) < pn] = ord("N")
            if rng.random() < 0.5:
                a = int(rng.integers(0, max(1, n - 50))); c[a:a + int(rng.integers(1, 50))] = ord("N")
            contigs.append(c)
        k, skip, shift = int(rng.choice([11, 13, 13])), int(rng.choice([0, 2, 2, 4])), int(rng.choice([2, 4, 4]))
        a, b = _both(lib, contigs, k=k, skip=skip, bin_shift=shift)
        _same(a, b, ("seed", seed, k, skip, shift, [len(c) for c in contigs]))


def test_device_builder_at_genome_scale(hip_aligner):
    """512 Mbp in 8 contigs: the device's table against the hashes recorded with tests/golden/cs_big.npz, which the packer checked
    against the table the unmodified reference built (tools/make_golden_cs.sh --big)."""
    z = np.load(os.path.join(util.GOLDEN, "cs_big.npz"))
    contigs = synth.big_reference(512 << 20, n_contigs=8)
    idx5, locs, _ = synth.kmer_table(hip_aligner.lib, contigs, k=int(z["k"]), skip=int(z["ref_skip"]), device=0)
    assert len(locs) == int(z["n_locations"])
    assert hashlib.sha256(idx5.tobytes()).hexdigest() == str(z["index_sha256"])
    assert hashlib.sha256(locs.tobytes()).hexdigest() == str(z["locs_sha256"])


def test_resident_table_is_taken_over_by_the_upload(hip_aligner):
    """CVX_INDEX_KEEP_RESIDENT: the table stays on the device in the search's own form and cvx_index_upload of the very arrays the
    build filled adopts it (no conversion of 4^13 records on the host, no copy back up) -- the search over it returns what it
    returns over a table uploaded the usual way; other arrays are uploaded the usual way."""
    import time
    from ngmlr_amd.aligner import KmerIndex
    lib = hip_aligner.lib
    contigs = synth.big_reference(8 << 20, n_contigs=2, seed=11)
    reads = synth.sample_subreads(contigs, 1500)
    idx, locs, _ = synth.kmer_table(lib, contigs, device=0, keep=True)
    dt = np.dtype([("tab", "<u4"), ("rc", "i1")])
    t0 = time.perf_counter()
    ix = KmerIndex(hip_aligner, 13, idx.view(dt), locs, 0)
    t_adopt = time.perf_counter() - t0
    try:
        got = ix.search(reads)
    finally:
        ix.free()
    idx2, locs2 = idx.copy(), locs.copy()          # other arrays: nothing to adopt
    t0 = time.perf_counter()
    ix2 = KmerIndex(hip_aligner, 13, idx2.view(dt), locs2, 0)
    t_upload = time.perf_counter() - t0
    try:
        want = ix2.search(reads)
        # the resident copy is gone after the first upload: the same arrays again take the usual path and still give the same lists
        ix3 = KmerIndex(hip_aligner, 13, idx.view(dt), locs, 0)
        try:
            again = ix3.search(reads[:200])
        finally:
            ix3.free()
    finally:
        ix2.free()
    assert all((g is None and w is None) or (g is not None and w is not None and np.array_equal(g, w)) for g, w in zip(got, want))
    assert all((g is None and w is None) or np.array_equal(g, w) for g, w in zip(again, want[:200]))
    assert sum(len(g) for g in got if g is not None) > 1000
    assert t_adopt < t_upload, (t_adopt, t_upload)


def test_device_builder_errors(hip_aligner):
    lib = hip_aligner.lib
    contigs = [_rand(np.random.default_rng(3), 5000)]
    idx, locs, starts = synth.kmer_table(lib, contigs)
    nl = C.c_uint64()
    binref = np.zeros(4096, dtype=np.uint8)
    st = np.array([0], dtype=np.uint64); ln = np.array([5002], dtype=np.uint64)
    out = np.zeros((4 ** 13 + 2) * 5, dtype=np.uint8)
    # too little room: the need comes back and the index records are complete
    rc = lib.cvx_index_build_device(0, binref.ctypes.data, 8192, st.ctypes.data, ln.ctypes.data, 1, 13, 2, 4, out.ctypes.data, None, 0, C.byref(nl), 0)
    assert rc == -6 and nl.value > 0
    assert lib.cvx_index_build_device(0, binref.ctypes.data, 8192, st.ctypes.data, ln.ctypes.data, 1, 3, 2, 4, out.ctypes.data, None, 0, C.byref(nl), 0) != 0      # k out of range
    assert lib.cvx_index_build_device(99, binref.ctypes.data, 8192, st.ctypes.data, ln.ctypes.data, 1, 13, 2, 4, out.ctypes.data, None, 0, C.byref(nl), 0) != 0     # no such device


def test_device_builder_on_the_quirk_reference(hip_aligner):
    """The reference of tests/test_index_cpu.py -- odd lengths, N runs at every place, a homopolymer, tandem repeats beyond the
    frequency cutoff, a period equal to the sampling stride, a second copy, lower case, 995 copies of a unit (slots reserved,
    weight 0) -- through both builders."""
    from tests.test_index_cpu import _reference
    seqs = _reference(np.random.default_rng(31))
    contigs = [s for _, s in seqs]
    a, b = _both(hip_aligner.lib, contigs)
    _same(a, b, "quirk reference")
    assert len(a[1]) > 50000


def test_device_builder_bound_inside_ngmlr_writes_the_reference_table(tmp_path):
    """index_build_binding.inc with a device present: the -ht-13-2.2.ngm file ngmlr_hip_all writes -- header, 4^13 + 1 index
    records, locations, unit offset -- is the unmodified binary's byte for byte, its log says the device built it, and the read
    mapped over the table the search adopted from the device gets the reference's SAM record."""
    from tests.test_index_cpu import REF_BIN, ROOT, _build_with, _reference
    hip_bin = os.path.join(ROOT, "oracle", "_ref", "ngmlr_hip_all")
    if not os.path.exists(REF_BIN) or not os.path.exists(hip_bin):
        pytest.skip("oracle/_ref/ngmlr_ref / ngmlr_hip_all not built")
    seqs = _reference(np.random.default_rng(31))
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    want_raw, want_sam, _ = _build_with(REF_BIN, tmp_path / "a", seqs)
    raw, sam, err = _build_with(hip_bin, tmp_path / "b", seqs)
    assert "cvx_index_build_device" in err, err[-1500:]
    assert raw.shape == want_raw.shape and np.array_equal(raw, want_raw)
    assert sam == want_sam and any(not l.startswith("@") for l in sam)


def test_device_builder_many_short_sequences(hip_aligner):
    """A reference of thousands of short sequences (an assembly in contigs): every sequence its own chunks, its own start of the
    drop rule's state, its own last-window test -- against the host builder."""
    rng = np.random.default_rng(77)
    contigs = []
    for _ in range(3000):
        n = int(rng.choice([11, 12, 13, 14, 15, 25, 100, 700, 2047, 2048, 2049, 3000]))
        c = _rand(rng, n)
        if rng.random() < 0.3:
            c[rng.random(n) < 0.05] = ord("N")
        if rng.random() < 0.1:
            c[:] = c[0]                       # a homopolymer: equal k-mers from the first window on
        contigs.append(c)
    a, b = _both(hip_aligner.lib, contigs)
    _same(a, b, "many short sequences")
    assert len(a[1]) > 100000
