"""Dev tool (GPU box): a randomised parity sweep of the device candidate search against oracle/cs_oracle.c.

    fuzz_search.py SECONDS [seed0]

Round r (seed = seed0 + r): a fresh reference of 1-6 Mbp in 1-5 sequences with repeat families and microsatellites
(synth.big_reference), ngmlr's k-mer table of it (cvx_genome_encode + cvx_index_build), 2 500 sub-reads of a length drawn for the
round (64 ... 1 000; 15 % error, half reverse-complemented) plus odd ones -- random junk, reads with N, reads shorter than a k-mer,
a microsatellite read, reads of one repeat unit --, CS::RunRead's parameters drawn for the round (sensitivity, minimum hits, bin
shift, first table size 2^8 ... 2^18) and one of the kernel forms (one wave per read with the vote map in LDS -- one map size, sorted
by map size, the smallest sizes forced --, the same over the table in HBM, one lane per read).  Every list is compared with the CPU restatement over the very same table on 16 host threads:
entries, order, scores, strands, maxHitNumber, kCount.  Stops at the first round with a mismatch (exit code 1)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ngmlr_amd import synth                               # noqa: E402
from ngmlr_amd.aligner import ConvexAlignHip, KmerIndex   # noqa: E402
from oracle.pyoracle import SearchOracle                  # noqa: E402

THREADS = 16
# (name, environment of the search call): the runtime reads these per call
FORMS = [("wave", {}), ("wave sorted by map size", {"CVX_TUNE_SEARCH_CLASSIFY": "0"}), ("wave sorted by map size", {"CVX_TUNE_SEARCH_CLASSIFY": "0"}),
         ("wave 2^9 maps", {"CVX_TUNE_SEARCH_LOG2": "9"}), ("wave 2^10 maps", {"CVX_TUNE_SEARCH_LOG2": "10"}), ("wave 2^12 maps", {"CVX_TUNE_SEARCH_LOG2": "12"}),
         ("wave 12-byte maps", {"CVX_TUNE_SEARCH_CLASSIFY": "0", "CVX_TUNE_SEARCH_SLOT8": "0"}), ("wave 8-byte maps", {"CVX_TUNE_SEARCH_CLASSIFY": "0", "CVX_TUNE_SEARCH_SLOT8": "1"}),
         ("wave_hbm", {"CVX_TUNE_SEARCH_WAVE": "2"}), ("lane", {"CVX_TUNE_SEARCH_WAVE": "0"})]
KNOBS = ("CVX_TUNE_SEARCH_WAVE", "CVX_TUNE_SEARCH_CLASSIFY", "CVX_TUNE_SEARCH_LOG2", "CVX_TUNE_SEARCH_SLOT8")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    t_start = time.time()
    tot = {"rounds": 0, "reads": 0, "lists": 0, "cands": 0}
    r = 0
    while time.time() - t_start < budget:
        seed = seed0 + r
        rng = np.random.default_rng(seed)
        mbp = float(rng.choice([1, 2, 4, 6]))
        n_ctg = int(rng.integers(1, 6))
        contigs = synth.big_reference(int(mbp * (1 << 20)), n_contigs=n_ctg, seed=seed, families=int(rng.integers(2, 12)), microsats=int(rng.integers(5, 60)))
        form, form_env = FORMS[int(rng.integers(0, len(FORMS)))]
        al = ConvexAlignHip(device=0)
        idx5, locs, starts = synth.kmer_table(al.lib, contigs)
        length = int(rng.choice([64, 128, 256, 256, 400, 1000]))
        reads = synth.sample_subreads(contigs, 2500, length=length, err=float(rng.choice([0.05, 0.15, 0.25])), seed=seed)
        for _ in range(40):
            reads.append(synth.random_ref(rng, int(rng.integers(13, 600))).tobytes())                 # junk
            q = bytearray(reads[int(rng.integers(0, 2500))])
            for p in rng.integers(0, len(q), size=int(rng.integers(1, 12))): q[int(p)] = ord("N")
            reads.append(bytes(q))
        reads += [b"", b"ACGT", b"ACGTACGTACGT", b"A" * 300, b"AC" * 200, b"ACGTTGCA" * 50]
        c = contigs[0]
        a = int(rng.integers(0, len(c) - 3000))
        reads.append(c[a:a + 2000].tobytes())                                                        # error-free, long
        par = {"sensitivity": float(rng.choice([0.8, 0.8, 0.5, 0.95])), "min_kmer_hits": float(rng.choice([0.0, 0.0, 2.0, 5.0])),
               "bin_shift": int(rng.choice([2, 4, 4, 6])), "first_bits": int(rng.choice([8, 10, 12, 16, 16, 18]))}
        ix = KmerIndex(al, 13, idx5.view(np.dtype([("tab", "<u4"), ("rc", "i1")])), locs, 0)
        try:
            for kn in KNOBS: os.environ.pop(kn, None)
            os.environ.update(form_env)          # (round 6: until then the variable was cleared again before the call that reads it)
            t0 = time.time()
            got, max_hit, misses = ix.search(reads, extras=True, **par)
            t1 = time.time()
        finally:
            for kn in KNOBS: os.environ.pop(kn, None)
            ix.free()
            al.close()
        orc = SearchOracle(raw=(13, 0, idx5, locs))
        want = [None] * len(reads)
        def work(k):
            for i in range(k, len(reads), THREADS):
                want[i] = orc.search(reads[i], sensitivity=par["sensitivity"], min_hits=par["min_kmer_hits"], bin_shift=par["bin_shift"],
                                     first_bits=par["first_bits"], cap=1 << 16)
        ths = [threading.Thread(target=work, args=(k,)) for k in range(THREADS)]
        for t in ths: t.start()
        for t in ths: t.join()
        orc.close()
        t2 = time.time()
        bad = []
        for i, (w, g) in enumerate(zip(want, got)):
            if w["n"] < 0:
                ok = g is None
            else:
                ok = (g is not None and w["n"] == len(g) and np.array_equal(g["location"], w["loc"]) and np.array_equal(g["score"], w["score"])
                      and np.array_equal(g["reverse"], w["rev"]) and float(max_hit[i]) == float(np.float32(w["max_hit"])))
            if ok and int(misses[i]) != int(w["kmer_misses"]):
                ok = False
            if not ok:
                bad.append((i, len(reads[i]), w["n"], None if g is None else len(g), int(misses[i]), w["kmer_misses"], w["table_bits"]))
        n_lists = sum(1 for g in got if g is not None)
        n_cand = sum(len(g) for g in got if g is not None)
        tot["rounds"] += 1; tot["reads"] += len(reads); tot["lists"] += n_lists; tot["cands"] += n_cand
        print("seed %d: %.0f Mbp in %d sequences, %d locations, %s kernel, reads of %d, %s: %d reads, %d lists, %d candidates, longest list %d, tables up to 2^%d; "
              "device %.2f s, checker %.2f s: %d mismatches" % (seed, mbp, n_ctg, len(locs), form, length, par, len(reads), n_lists, n_cand,
              max([len(g) for g in got if g is not None] or [0]), max(w["table_bits"] for w in want), t1 - t0, t2 - t1, len(bad)), flush=True)
        if bad:
            for b in bad[:20]: print("    (read, length, want n, got n, kCount got, want, table bits)", b)
            print("FAILED after %d rounds" % tot["rounds"])
            sys.exit(1)
        r += 1
    print("fuzz_search: %d rounds, %d reads, %d lists, %d candidates in %.0f s: every list identical to the CPU restatement" % (
        tot["rounds"], tot["reads"], tot["lists"], tot["cands"], time.time() - t_start))


if __name__ == "__main__":
    main()
