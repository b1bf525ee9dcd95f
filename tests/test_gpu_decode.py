"""GPU (-m gpu): reference windows decoded on the device from the 4-bit genome resident in HBM
(decode half) against what the unmodified reference decoded on its own test data
(tools/make_golden.sh fixtures), against the decode oracle on windows that poke at every branch of
DecodeRefSequenceExact, and -- through cvx_submit_windows -- whole alignments whose reference never
crosses PCIe as characters."""
import os

import numpy as np
import pytest

from oracle.pyoracle import same_alignment
from tests import util

pytestmark = pytest.mark.gpu


def _load(name):
    path = os.path.join(util.GOLDEN, name) if not os.path.isabs(name) else name
    return np.load(path)


def _windows(z):
    return [(int(z["pos"][i]), int(z["len"][i]), z["bytes"][int(z["off"][i]):int(z["off"][i + 1])].tobytes()) for i in range(int(z["n"]))]


@pytest.mark.parametrize("fixture", ["decode_test_2.npz", "decode_test_4.npz", "decode_test_3.npz", "FULL"])
def test_device_decode_equals_recorded_reference_windows(hip_aligner, fixture):
    from ngmlr_amd.aligner import Genome
    if fixture == "FULL":
        path = util.full_golden_path("decode_test_3_full.npz")
        if path is None:
            pytest.skip("oracle/_ref/golden_full not generated (tools/make_golden.sh needs /root/reference)")
        z = _load(path)
    else:
        z = _load(fixture)
    g = Genome(hip_aligner, z["binref"], int(z["nibbles"]), z["starts"])
    wins = _windows(z)
    got = g.decode([w[0] for w in wins], [w[1] for w in wins])
    g.free()
    bad = [(w[0], w[1]) for w, o in zip(wins, got) if o != w[2]]
    assert not bad, bad[:5]


def test_device_decode_every_branch_vs_oracle(hip_aligner, built):
    """Windows in front of the first chromosome, inside the N spacers, across chromosome ends, of odd and
    even starts and lengths, one base long, and past the last chromosome."""
    from ngmlr_amd.aligner import Genome
    from oracle.pyoracle import DecodeOracle
    z = _load("decode_test_3.npz")
    starts = [int(x) for x in z["starts"]]
    orc = DecodeOracle()
    rng = np.random.default_rng(12)
    pos, ln = [], []
    for s0, s1 in zip(starts[:-1], starts[1:]):
        for p in (s0 - 1200, s0 - 1000, s0 - 999, s0 - 3, s0 - 1, s0, s0 + 1, s0 + 2, s1 - 1000 - 60, s1 - 1001, s1 - 1000, s1 - 999, s1 - 500):
            if p < 1 or p >= starts[-1] - 2:
                continue
            for L in (1, 2, 3, 50, 51, 2000, 2001):
                pos.append(p)
                ln.append(L)
    for _ in range(400):
        pos.append(int(rng.integers(1000, starts[-1] - 1001)))
        ln.append(int(rng.integers(1, 9000)))
    g = Genome(hip_aligner, z["binref"], int(z["nibbles"]), z["starts"])
    got = g.decode(pos, ln)
    g.free()
    bad = [(p, L) for p, L, o in zip(pos, ln, got) if o != orc.window(z["binref"], z["starts"], p, L)]
    assert not bad, bad[:8]
    assert any(b"x" in o[:-1] for o in got) and any(b"N" in o[:-1] for o in got)


def test_alignments_from_resident_genome_windows(hip_aligner, port_oracle):
    """cvx_submit_windows: tiles carry (position, length) instead of decoded reference characters; results
    must equal those of the same tiles uploaded with host-side references, and the oracle's."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import Genome
    from oracle.pyoracle import DecodeOracle
    z = _load("decode_test_3.npz")
    starts = [int(x) for x in z["starts"]]
    orc = DecodeOracle()
    rng = np.random.default_rng(7)
    tiles, positions = [], []
    for k in range(40):
        c = int(rng.integers(0, len(starts) - 1))
        W = int(rng.integers(200, 6000))
        lo = starts[c] - (300 if k % 7 == 0 else 0)                      # some windows start in the spacer in front
        hi = max(lo + 1, starts[c + 1] - 1000 - W + (300 if k % 5 == 0 else 0))   # some hang over the chromosome end
        p = int(rng.integers(lo, hi + 1))
        ref = orc.window(z["binref"], z["starts"], p, W + 1)[:W]         # the string ngmlr's caller would hold
        qry = synth.mutate(rng, np.frombuffer(ref.replace(b"x", b"A"), dtype=np.uint8), 0.12)
        off, ln = synth.corridor_anchors(len(qry), W)
        tiles.append(synth.Tile(ref, qry.tobytes(), off, ln, tag="win%d" % k))
        positions.append(p)
    g = Genome(hip_aligner, z["binref"], int(z["nibbles"]), z["starts"])
    job = g.submit(tiles, positions)
    res, ops = job.wait()
    from ngmlr_amd.aligner import format_alignment
    from ngmlr_amd import capi
    import ctypes as C
    got = []
    for i, t in enumerate(tiles):
        r = capi.CvxResult.from_buffer_copy(res[i].tobytes())
        got.append(format_alignment(hip_aligner.lib, r, ops, t))
    # ABI 9: the decoded windows came back with the results (what a host text stage reads instead of decoding itself)
    ptrs = (C.c_void_p * len(tiles))()
    capi.check(hip_aligner.lib.cvx_job_window_refs(hip_aligner.h, job.j, ptrs))
    for t, p in zip(tiles, ptrs):
        assert C.string_at(p, len(t.ref)) == t.ref, t.tag
    job.release()
    g.free()
    plain = hip_aligner.submit(tiles[:3])
    plain.wait()
    assert hip_aligner.lib.cvx_job_window_refs(hip_aligner.h, plain.j, ptrs) == -3      # not a job of cvx_submit_windows
    plain.release()
    ref_run = hip_aligner.batch_align(tiles)
    n_valid = 0
    for t, a, b in zip(tiles, got, ref_run):
        assert same_alignment(b, a) is None, (t.tag, same_alignment(b, a))
        assert same_alignment(port_oracle.align(t), a) is None, t.tag
        n_valid += a["ret"] >= 0
    assert n_valid > 20
