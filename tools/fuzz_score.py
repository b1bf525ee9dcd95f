"""Dev tool (GPU box): a randomised parity sweep of the device sub-read scoring (cvx_score_batch) against StrippedSW + ssw.c itself
(oracle/_ref) -- or its restatement when the reference build is absent.

    fuzz_score.py SECONDS [seed0]

Round r (seed = seed0 + r): 20 000 pairs -- ScoreBuffer-shaped (256 / 348-character sub-reads against windows of 308 / 400), identities
0 ... 100 %, near-identical pairs of up to 510 bases (scores far above the 8-bit kernel's 255), lengths around every switch (255, 256,
511, 512), unrelated and empty strings, N / x / lower case, windows of up to 2 047 columns, long-by-long pairs where a 255-per-base
gap can pay -- scored in one call, in calls of 1 024 (the reference's batch) and, every fourth round, by the row kernels alone
(CVX_TUNE_SCORE_NO_DIAG=1).  Integer scores, bit-exact; stops at the first mismatch (exit code 1)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ngmlr_amd import synth                         # noqa: E402
from ngmlr_amd.aligner import StrippedSWHip         # noqa: E402
from oracle.pyoracle import have_score_ref          # noqa: E402
from tests.test_gpu_score import _oracle_scores_threaded   # noqa: E402


def draw(seed, n=20000):
    rng = np.random.default_rng(seed)
    refs, qrys = [], []
    for i in range(n):
        k = int(rng.integers(0, 10))
        if k <= 2:
            sub, win = ((256, 308), (348, 400), (128, 180))[k]
            w = synth.random_ref(rng, win)
            a = int(rng.integers(0, win - sub))
            q = synth.mutate(rng, w[a:a + sub], float(rng.choice([0.0, 0.02, 0.1, 0.15, 0.25, 0.5])))[:sub]
        elif k == 3:
            L = int(rng.integers(257, 511)); w = synth.random_ref(rng, L + 30); q = synth.mutate(rng, w[15:15 + L], float(rng.choice([0.0, 0.003, 0.02])))[:510]
        elif k == 4:
            L = int(rng.choice([254, 255, 256, 257, 510, 511, 512, 513])); w = synth.random_ref(rng, L + int(rng.integers(0, 40))); q = w[:L].copy()
        elif k == 5:
            w = synth.random_ref(rng, int(rng.integers(0, 330))); q = synth.random_ref(rng, int(rng.integers(0, 260)))
        elif k == 6:
            w = synth.random_ref(rng, 300, n_frac=0.05, x_frac=0.03); q = synth.mutate(rng, w, 0.05, n_frac=0.03)[:280]
        elif k == 7:
            w = np.frombuffer(synth.random_ref(rng, 320).tobytes().lower(), dtype=np.uint8); q = np.frombuffer(w[20:280].tobytes().upper(), dtype=np.uint8)
        elif k == 8:
            w = synth.random_ref(rng, int(rng.integers(1500, 2047)), n_frac=0.02, x_frac=0.02); a = int(rng.integers(0, 1000)); q = synth.mutate(rng, w[a:a + 400], 0.1)
        else:
            if rng.random() < 0.15:      # long by long: a gap can pay (row kernels)
                w = synth.random_ref(rng, 1400); c = int(rng.integers(300, 900)); q = np.concatenate([w[100:c], w[c + int(rng.integers(1, 4)):1200]])
            else:                        # two matching blocks separated by junk in the read
                w = synth.random_ref(rng, 700); q = np.concatenate([w[50:290], synth.random_ref(rng, int(rng.integers(1, 30))), w[290:520]])
        refs.append(w.tobytes()); qrys.append(q.tobytes())
    return refs, qrys


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    kind = "reference" if have_score_ref() else "port"
    t_start = time.time()
    total = 0
    r = 0
    while time.time() - t_start < budget:
        seed = seed0 + r
        refs, qrys = draw(seed)
        rows_only = r % 4 == 3
        if rows_only: os.environ["CVX_TUNE_SCORE_NO_DIAG"] = "1"
        sw = StrippedSWHip(device=0)
        os.environ.pop("CVX_TUNE_SCORE_NO_DIAG", None)
        t0 = time.time()
        got = sw.batch_score(refs, qrys)
        got_b = np.concatenate([sw.batch_score(refs[lo:lo + 1024], qrys[lo:lo + 1024]) for lo in range(0, 4096, 1024)])
        t1 = time.time()
        sw.close()
        want = _oracle_scores_threaded(refs, qrys, kind=kind)
        t2 = time.time()
        bad = np.nonzero(got != want)[0]
        bad_b = np.nonzero(got_b != want[:4096])[0]
        total += len(refs)
        print("seed %d%s: %d pairs, scores 0 ... %d (%d above 255), device %.2f s, %s %.2f s: %d + %d mismatches" % (
            seed, " (row kernels only)" if rows_only else "", len(refs), int(want.max()), int((want > 255).sum()), t1 - t0, kind, t2 - t1, len(bad), len(bad_b)), flush=True)
        if len(bad) or len(bad_b):
            for i in list(bad[:10]) + list(bad_b[:5]):
                print("    pair %d: window %d, read %d characters: device %g, %s %g" % (i, len(refs[i]), len(qrys[i]), got[i], kind, want[i]))
            print("FAILED after %d rounds" % (r + 1))
            sys.exit(1)
        r += 1
    print("fuzz_score: %d rounds, %d pairs in %.0f s: every score identical to the %s" % (r, total, time.time() - t_start, "reference's StrippedSW" if kind == "reference" else "restatement"))


if __name__ == "__main__":
    main()
