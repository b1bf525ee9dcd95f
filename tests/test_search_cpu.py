"""CPU: the candidate-search oracle (oracle/cs_oracle.c, a restatement of CS::RunRead: src/CS.cpp:57-149, 219-268, 324-398,
src/CSstatic.cpp:23-73, src/PrefixTable.cpp:476-532) against every candidate-search call recorded from the unmodified
reference on its own test_3 reads (tools/make_golden_cs.sh): the LocationScore list in the reference's own order, maxHitNumber,
the threshold applied, the length of rList, and kCount (the k-mers found in neither orientation, summed over the attempts of the
retry ladder -- each read searched with the table size the reference's first attempt had, which the recording holds).  This is
what pins the oracle the device kernel is checked against."""
import os

import numpy as np
import pytest

from oracle.pyoracle import SearchFixture, SearchOracle
from tests import util


def _check(fx, o, idx):
    bad = []
    for i in idx:
        loc, sc, rev = fx.want[i]
        # at the thread's starting table size (what the drop-in's binding uses) and at the size the reference's first attempt had:
        # the same list either way; kCount only with the latter (it counts the failed attempts' k-mers as well)
        for bits in (16, int(fx.first_bits[i])):
            g = o.search(fx.seqs[i], first_bits=bits)
            if not (g["n"] == len(loc) and np.array_equal(g["loc"], loc) and np.array_equal(g["score"], sc) and np.array_equal(g["rev"], rev)
                    and g["max_hit"] == fx.max_hit[i] and g["thresh"] == fx.thresh[i] and g["rlist_len"] == fx.rlist_len[i]):
                bad.append(i)
        if g["kmer_misses"] != int(fx.kmer_misses[i]):
            bad.append(i)
    return bad


def test_oracle_reproduces_recorded_candidate_searches(built):
    fx = SearchFixture(os.path.join(util.GOLDEN, "cs_test_3.npz"))
    assert len(fx.seqs) > 900 and sum(len(w[0]) for w in fx.want) > 1000
    assert int(fx.first_bits.min()) == 8 and int(fx.first_bits.max()) == 16 and int(fx.kmer_misses.max()) > 256      # ladder sums are in the recording
    o = SearchOracle(fx)
    assert _check(fx, o, range(len(fx.seqs))) == []
    o.close()


def test_oracle_reproduces_every_recorded_candidate_search(built):
    path = util.full_golden_path("cs_test_3_full.npz")
    if path is None:
        pytest.skip("oracle/_ref/golden_full/cs_test_3_full.npz not generated (tools/make_golden_cs.sh needs /root/reference)")
    fx = SearchFixture(path)
    assert len(fx.seqs) == 5663
    o = SearchOracle(fx)
    assert _check(fx, o, range(len(fx.seqs))) == []
    o.close()


def test_oracle_reproduces_recorded_searches_on_a_repeat_rich_reference(built):
    """The same on a reference a k-mer vote has to work on (tools/make_golden_cs.sh, round 5: repeat families of diverged copies,
    microsatellites, one 400-bp unit in 700 copies): sub-reads with thousands of votes and hundreds of listed bins, table sizes
    the reference had adapted to (2^12 .. 2^16), kCount summed over ladders that really climbed."""
    fx = SearchFixture(os.path.join(util.GOLDEN, "cs_rep.npz"))
    assert len(fx.seqs) >= 1000 and max(len(w[0]) for w in fx.want) >= 300 and int(fx.rlist_len.max()) >= 500
    o = SearchOracle(fx)
    bad = []
    for i in range(len(fx.seqs)):
        loc, sc, rev = fx.want[i]
        for bits in (16, int(fx.first_bits[i])):
            g = o.search(fx.seqs[i], first_bits=bits, cap=1 << 16)
            if not (g["n"] == len(loc) and np.array_equal(g["loc"], loc) and np.array_equal(g["score"], sc) and np.array_equal(g["rev"], rev)
                    and g["max_hit"] == fx.max_hit[i] and g["thresh"] == fx.thresh[i] and g["rlist_len"] == fx.rlist_len[i]):
                bad.append(i)
        if g["kmer_misses"] != int(fx.kmer_misses[i]):
            bad.append(i)
    o.close()
    assert bad == []


def test_oracle_reproduces_recorded_searches_at_genome_scale(built):
    """VERDICT r5 item 5d / missing #6: RunRead calls recorded from the unmodified reference where the k-mer table leaves every cache
    (512 Mbp, 62 M used prefixes, 179 M locations; tools/make_golden_cs.sh --big).  cvx_index_build rebuilds the table the
    reference searched (SHA-256 of index and locations as the packer verified them against the reference's dump), and the CPU
    restatement reproduces all 840 recorded calls over it: lists, order, maxHitNumber, threshold, rList length, kCount."""
    fx, idx5, locs = util.big_search_case()
    assert len(fx.seqs) >= 500 and sum(len(w[0]) for w in fx.want) >= 1000
    o = SearchOracle(raw=(fx.k, fx.unit_offset, idx5, locs))
    bad = []
    for i in range(len(fx.seqs)):
        loc, sc, rev = fx.want[i]
        g = o.search(fx.seqs[i], first_bits=int(fx.first_bits[i]), cap=1 << 16)
        if not (g["n"] == len(loc) and np.array_equal(g["loc"], loc) and np.array_equal(g["score"], sc) and np.array_equal(g["rev"], rev)
                and g["max_hit"] == fx.max_hit[i] and g["thresh"] == fx.thresh[i] and g["rlist_len"] == fx.rlist_len[i]
                and g["kmer_misses"] == int(fx.kmer_misses[i])):
            bad.append(i)
    o.close()
    assert bad == []


def test_n_runs_and_the_retry_ladder(built):
    """Behaviour the recorded reads do not reach: windows holding 'N' are skipped (with the quirk that a run of N at the start of
    a restarted stretch ends the walk when 13 or fewer characters follow), and a read whose votes overflow the probe budget
    of the 2^16-entry table is searched again with 2^18 / 2^19 / 2^20 entries (src/CS.cpp:345-394)."""
    fx, reads = util.synthetic_search_case()
    o = SearchOracle(fx)
    res = [o.search(r) for r in reads]
    o.close()
    bits = [r["table_bits"] for r in res if r["n"] >= 0]
    assert 16 in bits and max(bits) > 16                                  # both the first attempt and a retry happened
    clean, with_n = res[0], res[1]
    assert clean["n"] > 0 and 0 <= with_n["n"]
    # 13 characters behind a run of two N: nothing from that tail; behind a single N: one k-mer
    a, b = res[2], res[3]
    assert a["max_hit"] < b["max_hit"] or a["n"] <= b["n"]


def n_pattern_reads(rng, base: bytes, n: int, K: int = 13):
    """Variants of `base` with 'N's sprinkled in and the tail patterns PrefixIteration treats specially (a run of 'N's with exactly
    K bases behind it) -- shared with tests/test_gpu_search.py."""
    out = []
    for _ in range(n):
        L = int(rng.integers(0, len(base) + 1)) if rng.random() < 0.5 else int(rng.integers(0, 4 * K))
        a = int(rng.integers(0, len(base) - L + 1))
        s = bytearray(base[a:a + L])
        pn = float(rng.choice([0.0, 0.01, 0.05, 0.3]))
        for p in range(L):
            if rng.random() < pn:
                s[p] = ord("N")
        if L > K + 2 and rng.random() < 0.4:
            r, q = int(rng.integers(1, 4)), L - K
            for p in range(max(0, q - r), q):
                s[p] = ord("N")
            if rng.random() < 0.7:
                for p in range(q, L):
                    s[p] = base[a + p] if base[a + p] != ord("N") else ord("A")
        out.append(bytes(s))
    return out


def walk_windows(s: bytes, K: int) -> int:
    """How many k-mers CS::PrefixIteration (reference src/CSstatic.cpp:23-73) visits, in the closed form the wave kernels of
    cvx_search.hip use: every window without an 'N', except that the read's last window is lost when it directly follows a run of
    'N's that is two or more long or starts the read (`n_skip >= length - prefixBasecount` at :33 where `>` would keep it)."""
    n_all = len(s) - K + 1
    n = n_all
    if n_all >= 2:
        q = n_all - 1
        if s[q - 1] == ord("N") and (q == 1 or s[q - 2] == ord("N")):
            n -= 1
    return sum(1 for q in range(max(n, 0)) if ord("N") not in s[q:q + K])


@pytest.mark.parametrize("K", [13, 5])
def test_closed_form_of_the_kmer_walk(built, K):
    """The wave kernels enumerate a read's k-mers 64 window positions at a time from a closed form of PrefixIteration instead of
    walking it serially (round 6).  Pinned here on the CPU: over an EMPTY table every visited k-mer counts as a miss (kCount,
    src/CS.cpp:67-69), so the checker's kCount is the number of k-mers its serial walk -- the restatement of the reference's --
    visits; 20 000 random 'N' patterns per k-mer length, the tail patterns forced often."""
    idx5 = np.zeros(((1 << (2 * K)) + 2) * 5, dtype=np.uint8)
    o = SearchOracle(raw=(K, 0, idx5, np.zeros(1, dtype=np.uint32)))
    rng = np.random.default_rng(70 + K)
    base = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300).tobytes()
    reads = n_pattern_reads(rng, base, 20000, K) + [b"", b"N", b"N" * K, b"N" + base[:K], b"NN" + base[:K], base[:3] + b"N" + base[:K],
                                                    base[:3] + b"NN" + base[:K], base[:K] + b"N", base[:K + 1] + b"NN" + base[:K - 1]]
    try:
        bad = [(r, o.search(r, cap=64)["kmer_misses"], walk_windows(r, K)) for r in reads]
        bad = [b for b in bad if b[1] != b[2]]
    finally:
        o.close()
    assert not bad, (len(bad), bad[:3])
    assert sum(walk_windows(r, K) for r in reads) > 100000
