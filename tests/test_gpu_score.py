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
