"""CPU: the scoring-path oracle.  oracle/score_oracle.c restates what
StrippedSW::BatchScore computes; it is pinned here against the reference's own StrippedSW +
ssw.c (oracle/_ref/libscore_oracle_ref.so), including pairs whose score passes 255 (the
reference then switches to its 16-bit kernel, where a 255-per-base gap can pay off)."""
import numpy as np
import pytest

from ngmlr_amd import synth


def score_pairs(seed=3, n=300):
    rng = np.random.default_rng(seed)
    refs, qrys = [], []
    for i in range(n):
        k = i % 6
        if k == 0:      # sub-read vs candidate window (ScoreBuffer: 256 bp vs ~308 bp)
            w = synth.random_ref(rng, 308)
            a = int(rng.integers(0, 40))
            q = synth.mutate(rng, w[a:a + 256], float(rng.choice([0.0, 0.05, 0.2])))
        elif k == 1:    # inversion check sized, score >= 255: gaps of cost 255 can pay
            w = synth.random_ref(rng, int(rng.integers(800, 3000)))
            q = synth.mutate(rng, w, float(rng.choice([0.0, 0.002, 0.01, 0.05])))
        elif k == 2:    # unrelated sequences
            w = synth.random_ref(rng, int(rng.integers(1, 400)))
            q = synth.random_ref(rng, int(rng.integers(1, 300)))
        elif k == 3:    # N / x symbols score 0
            w = synth.random_ref(rng, 300, n_frac=0.05, x_frac=0.03)
            q = synth.mutate(rng, w, 0.05, n_frac=0.03)
        elif k == 4:    # exactly at the 8-bit/16-bit switch
            L = int(rng.integers(250, 262))
            w = synth.random_ref(rng, L + 20)
            q = w[10:10 + L].copy()
        else:           # lower case and empty strings
            w = np.frombuffer(synth.random_ref(rng, 120).tobytes().lower(), dtype=np.uint8)
            q = synth.random_ref(rng, int(rng.integers(0, 3)))
        refs.append(w.tobytes())
        qrys.append(q.tobytes())
    return refs, qrys


def test_port_equals_reference_strippedsw(built):
    from oracle.pyoracle import ScoreOracle, have_score_ref
    if not have_score_ref():
        pytest.skip("oracle/_ref/libscore_oracle_ref.so not built")
    refs, qrys = score_pairs()
    a = ScoreOracle("port").scores(refs, qrys)
    b = ScoreOracle("reference").scores(refs, qrys)
    assert np.array_equal(a, b)
    assert (a >= 255).sum() > 20 and (a < 255).sum() > 100 and a.max() > 1000


def test_too_long_sequences_score_minus_one(built):
    from oracle.pyoracle import ScoreOracle
    s = ScoreOracle("port").scores([b"A" * 99999, b"ACGT"], [b"ACGT", b"ACGT"])
    assert s[0] == -1.0 and s[1] == 4.0
