"""CPU: the decode oracle (oracle/decode_oracle.c: ngmlr's 4-bit genome encoding and
DecodeRefSequenceExact) against what the unmodified reference itself produced on its own test data
-- its encoded genome (binRef, refStartPos) and every window it decoded for an alignment, recorded
by tools/make_golden.sh (tests/golden/decode_test_*.npz; all 979 windows of test_3 under
oracle/_ref/golden_full/ when /root/reference was present at build time)."""
import gzip
import os

import numpy as np
import pytest

from tests import util

E2E = os.path.join(util.GOLDEN, "e2e")
FIXTURES = [("decode_test_2.npz", "ref_chr21_20kb.fa"), ("decode_test_4.npz", "test_4_reference.fasta.gz"),
            ("decode_test_3.npz", "test_3_reference.fasta.gz")]


def read_fasta(path):
    op = gzip.open if path.endswith(".gz") else open
    seqs, cur = [], None
    with op(path, "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                cur = []
                seqs.append(cur)
            elif cur is not None:
                cur.append(line)
    return [("".join(s)).encode() for s in seqs]


@pytest.fixture(scope="module")
def decode_oracle(built):
    from oracle.pyoracle import DecodeOracle
    return DecodeOracle()


def _windows(z):
    for i in range(int(z["n"])):
        yield int(z["pos"][i]), int(z["len"][i]), z["bytes"][int(z["off"][i]):int(z["off"][i + 1])].tobytes()


@pytest.mark.parametrize("fixture,fasta", FIXTURES)
def test_encoding_equals_the_references_binref(decode_oracle, fixture, fasta):
    z = np.load(os.path.join(util.GOLDEN, fixture))
    binref, nib, starts = decode_oracle.encode(read_fasta(os.path.join(E2E, fasta)))
    assert nib == int(z["nibbles"])
    assert np.array_equal(starts, z["starts"])
    assert np.array_equal(binref[:nib // 2], z["binref"])


@pytest.mark.parametrize("fixture", [f for f, _ in FIXTURES] + ["FULL"])
def test_windows_equal_the_references_decodes(decode_oracle, fixture):
    if fixture == "FULL":
        path = util.full_golden_path("decode_test_3_full.npz")
        if path is None:
            pytest.skip("oracle/_ref/golden_full not generated (tools/make_golden.sh needs /root/reference)")
    else:
        path = os.path.join(util.GOLDEN, fixture)
    z = np.load(path)
    n_x = n_n = 0
    for pos, ln, want in _windows(z):
        got = decode_oracle.window(z["binref"], z["starts"], pos, ln)
        assert got == want, (fixture, pos, ln)
        n_x += b"x" in want[:-1]
        n_n += b"N" in want[:-1]
    if fixture in ("decode_test_3.npz", "FULL"):
        assert n_x > 0 and n_n > 0          # windows hanging over a chromosome end / into the N spacers are covered


@pytest.mark.parametrize("fixture,fasta", FIXTURES)
def test_product_encoder_equals_the_references_binref(built, fixture, fasta):
    """cvx_genome_encode (host half of the resident-genome API, no device needed) byte for byte."""
    from ngmlr_amd import capi
    from ngmlr_amd.aligner import encode_genome
    z = np.load(os.path.join(util.GOLDEN, fixture))
    binref, nib, starts = encode_genome(capi.load(), read_fasta(os.path.join(E2E, fasta)))
    assert nib == int(z["nibbles"])
    assert np.array_equal(starts, z["starts"])
    assert np.array_equal(binref[:nib // 2], z["binref"])


def test_product_encoder_skips_short_sequences_like_the_reference(built, decode_oracle):
    from ngmlr_amd import capi
    from ngmlr_amd.aligner import encode_genome
    seqs = [b"ACGTNacgtnRYK" * 7, b"ACGT", b"A" * 11, b"G" * 10, b"tTgGcCaAxX-" * 3]
    a = encode_genome(capi.load(), seqs)
    b = decode_oracle.encode(seqs)
    assert a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[0][:a[1] // 2], b[0][:b[1] // 2])
    assert len(a[2]) == 4          # three sequences kept + the upper bound
