"""cvx_index_build (host, include/cvx_align.h) against the k-mer table the unmodified reference builds and writes to
<ref>-ht-13-2.2.ngm (CompactPrefixTable::CreateTable / saveToFile, reference src/PrefixTable.cpp:265-352, :534-567): the
5-byte index records and the location table, byte for byte -- on a reference made to hit the builder's quirks (sequences
of odd length whose last characters the reference's decode drops, N runs, homopolymers and tandem repeats for the "one per
bin" rule, a repeat beyond the frequency cutoff, a sequence too short to be kept, lower case), on one thread and on several.
Needs oracle/_ref/ngmlr_ref (built from /root/reference by tools/build_ngmlr_hip.sh): skipped without it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")


def _reference(rng):
    from ngmlr_amd import synth
    seqs = []
    a = synth.random_ref(rng, 300001)                                  # odd length
    a[5000:5400] = ord("N")                                             # an N run in the middle
    a[90000:96000] = ord("A")                                           # homopolymer: equal k-mers at consecutive sampled positions
    a[120000:150000] = np.tile(np.frombuffer(b"ACGTTGCA", dtype=np.uint8), 30000 // 8)      # tandem repeat: > 1000 occurrences per k-mer
    seqs.append(("odd", a))
    b = synth.random_ref(rng, 80000)
    b[:50] = ord("N")                                                   # starts with Ns
    b[-70:] = ord("N")                                                  # ends with Ns
    b[30000:34000] = np.tile(np.frombuffer(b"CAG", dtype=np.uint8), 4000 // 3 + 1)[:4000]   # period 3 = the sampling stride
    seqs.append(("nn", b))
    seqs.append(("tiny", synth.random_ref(rng, 9)))                    # skipped by the reference (<= 10 bases)
    c = synth.random_ref(rng, 40013)
    c[100:9000] = a[1000:9900]                                          # a second copy of part of the first sequence
    seqs.append(("copy", c))
    d = np.frombuffer(synth.random_ref(rng, 20000).tobytes().lower(), dtype=np.uint8).copy()      # lower case
    seqs.append(("lower", d))
    seqs.append(("short", synth.random_ref(rng, 14)))                  # barely longer than a k-mer
    # 995 copies of a 30-base unit: its ten sampled k-mers occur 995 times each -- below the cutoff of 1000, so their slots are
    # reserved, but the weight byte (char) ((1000 - 995) * 100.0f / 1000) is 0 and the reference then treats them as unused
    seqs.append(("quirk", np.tile(synth.random_ref(rng, 30), 995)))
    return seqs


def _build_with(binary, tmp_path, seqs):
    """-> (table file bytes, SAM records) of `binary` run on `seqs` in tmp_path"""
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for name, s in seqs:
            f.write(b">" + name.encode() + b"\n")
            raw = s.tobytes()
            for i in range(0, len(raw), 70):
                f.write(raw[i:i + 70] + b"\n")
    fq = str(tmp_path / "r.fq")
    open(fq, "w").write("@r\n%s\n+\n%s\n" % (seqs[0][1][1000:1600].tobytes().decode(), "I" * 600))
    res = subprocess.run([binary, "-x", "pacbio", "-t", "1", "-r", fa, "-q", fq, "-o", str(tmp_path / "o.sam")], cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    raw = np.fromfile(fa + "-ht-13-2.2.ngm", dtype=np.uint8)
    os.remove(fa + "-ht-13-2.2.ngm")
    sam = [l for l in open(str(tmp_path / "o.sam")).read().splitlines() if not l.startswith("@PG")]
    return raw, sam, res.stderr.decode(errors="replace")


@pytest.fixture(scope="module")
def reference_run(tmp_path_factory):
    """the sequences, the table file and the SAM of the unmodified reference (one run for the module)"""
    if not os.path.exists(REF_BIN):
        pytest.skip("oracle/_ref/ngmlr_ref not built (needs /root/reference)")
    seqs = _reference(np.random.default_rng(31))
    raw, sam, _ = _build_with(REF_BIN, tmp_path_factory.mktemp("index"), seqs)
    return seqs, raw, sam


@pytest.fixture(scope="module")
def reference_table(reference_run):
    """the sequences and the table unit the unmodified reference built from them"""
    seqs, raw, _ = reference_run
    cookie, k, skip, units, isz = [int(x) for x in raw[:20].view(np.uint32)]
    assert (k, skip, units, isz) == (13, 2, 1, 4 ** 13 + 1)
    tl = int(raw[20:24].view(np.uint32)[0])
    want_idx = raw[24:24 + isz * 5].copy()
    want_locs = raw[24 + isz * 5:24 + isz * 5 + tl * 4].view(np.uint32).copy()
    assert int(raw[24 + isz * 5 + tl * 4:24 + isz * 5 + tl * 4 + 8].view(np.uint64)[0]) == 0       # TableUnit::Offset
    return seqs, isz, tl, want_idx, want_locs


@pytest.mark.parametrize("threads", ["1", "4", "16"])      # 16: more threads than sequences -- the passes over the 4^13 records still use them all
def test_index_builder_equals_the_reference_table(built, reference_table, monkeypatch, threads):
    from ngmlr_amd import capi
    lib = capi.load()
    monkeypatch.setenv("CVX_INDEX_THREADS", threads)
    seqs, isz, tl, want_idx, want_locs = reference_table

    n = len(seqs)
    lens = np.array([len(s) for _, s in seqs], dtype=np.uint64)
    lib.cvx_genome_encoded_bytes.restype = C.c_uint64
    binref = np.zeros(int(lib.cvx_genome_encoded_bytes(n, lens.ctypes.data_as(C.c_void_p))), dtype=np.uint8)
    arr = (C.c_char_p * n)(*[s.tobytes() for _, s in seqs])
    nn, ns = C.c_uint64(), C.c_int32()
    starts = np.zeros(n + 1, dtype=np.uint64)
    capi.check(lib.cvx_genome_encode(n, arr, lens.ctypes.data_as(C.c_void_p), binref.ctypes.data_as(C.c_void_p), C.byref(nn),
                                     starts.ctypes.data_as(C.c_void_p), C.byref(ns)))
    kept = np.ascontiguousarray(lens[lens > 10])
    assert ns.value == len(kept) + 1 == 7
    idx = np.zeros((4 ** 13 + 2) * 5, dtype=np.uint8)
    nl = C.c_uint64()
    # too little room first: the need comes back, then the real call
    rc = lib.cvx_index_build(binref.ctypes.data, nn.value, starts.ctypes.data, kept.ctypes.data, len(kept), 13, 2, 4, idx.ctypes.data, None, 0, C.byref(nl))
    assert rc == -6 and nl.value == tl
    locs = np.zeros(nl.value, dtype=np.uint32)
    capi.check(lib.cvx_index_build(binref.ctypes.data, nn.value, starts.ctypes.data, kept.ctypes.data, len(kept), 13, 2, 4, idx.ctypes.data,
                                   locs.ctypes.data, len(locs), C.byref(nl)))
    assert nl.value == tl
    assert np.array_equal(idx[:isz * 5], want_idx), "index records differ at entry %d" % (int(np.nonzero(idx[:isz * 5] != want_idx)[0][0]) // 5)
    assert np.array_equal(locs, want_locs), "locations differ at %d" % int(np.nonzero(locs != want_locs)[0][0])
    # the workload did hit the rules it is for: unused k-mers with reserved slots, and fewer locations than sampled positions
    tab = np.zeros(isz, dtype=np.uint32)
    tab[:] = want_idx.reshape(-1, 5)[:, :4].copy().view(np.uint32).ravel()
    used = want_idx.reshape(-1, 5)[:, 4] != 0
    assert int((np.diff(tab.astype(np.int64)) > 0).sum()) > int(used.sum()), "no k-mer beyond the frequency cutoff kept its slots"
    assert tl < int(kept.sum()) // 3


def test_index_builder_bound_inside_ngmlr_writes_the_reference_table(reference_run, tmp_path):
    """cvx_index_build where ngmlr builds its table (ngmlr_amd/csrc/index_build_binding.inc at the top of
    CompactPrefixTable::CreateTable, reference src/PrefixTable.cpp:323; tools/build_ngmlr_hip.sh, variant ngmlr_index_cpu: the
    reference's CPU code with only that change): the -ht-13-2.2.ngm file it writes -- header, 4^13 + 1 index records, locations,
    unit offset -- is the unmodified binary's byte for byte, and so is the SAM of the read mapped with it."""
    binary = os.path.join(ROOT, "oracle", "_ref", "ngmlr_index_cpu")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/ngmlr_index_cpu not built")
    seqs, want_raw, want_sam = reference_run
    raw, sam, err = _build_with(binary, tmp_path, seqs)
    assert "cvx_index_build" in err, err[-1500:]          # the bound builder ran (its log line), not the reference's
    assert raw.shape == want_raw.shape and np.array_equal(raw, want_raw)
    assert sam == want_sam and any(not l.startswith("@") for l in sam)
