"""GPU (-m gpu): sub-read scoring through the C ABI (cvx_score_batch) against the CPU
checker -- integer scores, bit-exact."""
import numpy as np
import pytest

from tests.test_score_cpu import score_pairs

pytestmark = pytest.mark.gpu


def test_batch_score_equals_oracle(built):
    from ngmlr_amd.aligner import StrippedSWHip
    from oracle.pyoracle import ScoreOracle
    sw = StrippedSWHip(device=0)
    refs, qrys = score_pairs(seed=17, n=360)
    got = sw.batch_score(refs, qrys)
    want = ScoreOracle("port").scores(refs, qrys)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert sw.single_score(refs[0], qrys[0]) == want[0]
    sw.close()


def test_reference_batch_shape_1024_pairs(built):
    """One ScoreBuffer flush: 1024 (sub-read, window) pairs (src/StrippedSW.h:53-55)."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import StrippedSWHip
    from oracle.pyoracle import ScoreOracle
    rng = np.random.default_rng(5)
    refs, qrys = [], []
    for _ in range(1024):
        w = synth.random_ref(rng, 308)
        a = int(rng.integers(0, 50))
        qrys.append(synth.mutate(rng, w[a:a + 256], 0.15)[:256].tobytes())
        refs.append(w.tobytes())
    sw = StrippedSWHip(device=0)
    assert np.array_equal(sw.batch_score(refs, qrys), ScoreOracle("port").scores(refs, qrys))
    assert sw.batch_score([b"A" * 100000], [b"ACGT"])[0] == -1.0
    sw.close()


def _oracle_scores_threaded(refs, qrys, kind="port", threads=16):
    import threading
    from oracle.pyoracle import ScoreOracle
    n = len(refs)
    out = np.zeros(n, dtype=np.float32)
    step = (n + threads - 1) // threads

    def work(k):
        lo, hi = k * step, min(n, (k + 1) * step)
        if lo < hi:
            out[lo:hi] = ScoreOracle(kind).scores(refs[lo:hi], qrys[lo:hi])
    ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    return out


def test_hundred_thousand_pairs_including_the_255_switch(built):
    """parity at >= 1e5 pairs.  ScoreBuffer-shaped pairs (register-resident kernel, windows of
    up to 512 columns) with identities from 0 to 100 %: scores from 0 to above 255, where ssw leaves its 8-bit
    kernel (a 255-per-base gap can pay there), plus N / x / lower case; in batches of 1024 like the reference
    (src/StrippedSW.h:53-55) and as one call."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import StrippedSWHip
    rng = np.random.default_rng(2026)
    refs, qrys = [], []
    for i in range(100000):
        k = i % 5
        if k == 0:
            w = synth.random_ref(rng, 308)
            a = int(rng.integers(0, 50))
            q = synth.mutate(rng, w[a:a + 256], float(rng.choice([0.0, 0.02, 0.1, 0.25])))[:256]
        elif k == 1:                                    # up to 500 columns, score well past 255
            L = int(rng.integers(257, 480))
            w = synth.random_ref(rng, L + 20)
            q = synth.mutate(rng, w[10:10 + L], float(rng.choice([0.0, 0.004, 0.02])))
        elif k == 2:
            w = synth.random_ref(rng, int(rng.integers(1, 330)))
            q = synth.random_ref(rng, int(rng.integers(0, 260)))
        elif k == 3:
            w = synth.random_ref(rng, 300, n_frac=0.05, x_frac=0.03)
            q = synth.mutate(rng, w, 0.05, n_frac=0.03)[:280]
        else:
            L = int(rng.integers(250, 262))             # exactly around the switch
            w = np.frombuffer(synth.random_ref(rng, L + 20).tobytes().lower(), dtype=np.uint8)
            q = np.frombuffer(w[10:10 + L].tobytes().upper(), dtype=np.uint8)
        refs.append(w.tobytes())
        qrys.append(q.tobytes())
    want = _oracle_scores_threaded(refs, qrys)
    assert want.max() > 400 and (want > 255).sum() > 10000 and (want == 0).sum() > 0
    sw = StrippedSWHip(device=0)
    got = sw.batch_score(refs, qrys)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    for lo in range(0, 8192, 1024):
        assert np.array_equal(sw.batch_score(refs[lo:lo + 1024], qrys[lo:lo + 1024]), want[lo:lo + 1024])
    sw.close()


def test_diagonal_kernel_equals_row_kernels_and_oracle(built):
    """The batched shape goes to score_diag_kernel (one lane per diagonal: with a 255-per-base gap no gapped path can win
    while the shorter sequence has at most 511 characters); the row-by-row kernels (CVX_TUNE_SCORE_NO_DIAG=1) and the CPU
    checker must agree with it pair by pair -- including scores far above 255, N / x, empty strings, a window of 2047
    columns -- and long-by-long pairs (both sides >= 512) still take the row kernels, where a gap CAN pay."""
    import os
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import StrippedSWHip
    rng = np.random.default_rng(88)
    refs, qrys = [], []
    for i in range(3000):
        k = i % 6
        if k == 0:
            w = synth.random_ref(rng, 308); a = int(rng.integers(0, 50)); q = synth.mutate(rng, w[a:a + 256], 0.15)[:256]
        elif k == 1:                                    # read of up to 510 bases, near-identical: score ~ 500
            L = int(rng.integers(400, 511)); w = synth.random_ref(rng, L + 30); q = synth.mutate(rng, w[15:15 + L], 0.003)[:510]
        elif k == 2:                                    # two long matching blocks separated by junk in the read: a gap would join them if it were cheap
            w = synth.random_ref(rng, 700); q = np.concatenate([w[50:290], synth.random_ref(rng, 12), w[290:520]])
        elif k == 3:
            w = synth.random_ref(rng, int(rng.integers(1500, 2047)), n_frac=0.02, x_frac=0.02); q = synth.mutate(rng, w[700:1100], 0.1)
        elif k == 4:
            w = synth.random_ref(rng, int(rng.integers(0, 40))); q = synth.random_ref(rng, int(rng.integers(0, 40)))
        else:                                           # long by long: the row kernels' territory (a 255-per-base gap can pay)
            w = synth.random_ref(rng, 1400); q = np.concatenate([w[100:600], w[601:1200]])
        refs.append(w.tobytes()); qrys.append(q.tobytes())
    want = _oracle_scores_threaded(refs, qrys)
    sw = StrippedSWHip(device=0)
    got = sw.batch_score(refs, qrys)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    short = [i for i in range(3000) if i % 6 != 5]
    got_s = sw.batch_score([refs[i] for i in short], [qrys[i] for i in short])      # a batch the diagonal kernel takes whole
    assert np.array_equal(got_s, want[short])
    sw.close()
    os.environ["CVX_TUNE_SCORE_NO_DIAG"] = "1"
    try:
        sw2 = StrippedSWHip(device=0)
    finally:
        del os.environ["CVX_TUNE_SCORE_NO_DIAG"]
    assert np.array_equal(sw2.batch_score([refs[i] for i in short], [qrys[i] for i in short]), want[short])
    sw2.close()
    assert want[[i for i in range(3000) if i % 6 == 5]].min() > 700          # the gap did pay there: 500 + 599 - 255
