"""Dev tool (GPU): sub-read scoring -- cvx_score_batch against the reference's own
StrippedSW + ssw.c (oracle/_ref, SSE2) on the same pairs: pairs/s and cell updates/s, per 1024-pair call
(the reference's batch size, src/StrippedSW.h:53-55) and as large calls, host strings in -> scores out."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from ngmlr_amd import synth  # noqa: E402
from ngmlr_amd.aligner import StrippedSWHip  # noqa: E402
from oracle.pyoracle import ScoreOracle, have_score_ref  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(1)
refs, qrys = [], []
for _ in range(n):
    w = synth.random_ref(rng, 308)
    a = int(rng.integers(0, 50))
    qrys.append(synth.mutate(rng, w[a:a + 256], 0.15)[:256].tobytes())
    refs.append(w.tobytes())
cells = sum((len(r) + 1) * (len(q) + 1) for r, q in zip(refs, qrys))
sw = StrippedSWHip(device=0)
sw.batch_score(refs[:1024], qrys[:1024])
t0 = time.perf_counter()
got = sw.batch_score(refs, qrys)
dt_big = time.perf_counter() - t0
t0 = time.perf_counter()
for lo in range(0, n, 1024):
    sw.batch_score(refs[lo:lo + 1024], qrys[lo:lo + 1024])
dt_1k = time.perf_counter() - t0
print("GPU  one call of %d pairs: %.1f ms -> %.2f M pairs/s, %.1f G cell updates/s (incl. python marshalling, H2D, D2H)" % (
    n, dt_big * 1e3, n / dt_big / 1e6, cells / dt_big / 1e9))
print("GPU  %d calls of 1024 pairs: %.1f ms -> %.2f M pairs/s, %.0f us per call" % (
    (n + 1023) // 1024, dt_1k * 1e3, n / dt_1k / 1e6, dt_1k / ((n + 1023) // 1024) * 1e6))
kind = "reference" if have_score_ref() else "port"
threads = os.cpu_count() or 1
want = np.zeros(n, dtype=np.float32)
step = (n + threads - 1) // threads


def work(k):
    lo, hi = k * step, min(n, (k + 1) * step)
    if lo < hi:
        want[lo:hi] = ScoreOracle(kind).scores(refs[lo:hi], qrys[lo:hi])


ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
t0 = time.perf_counter()
[t.start() for t in ths]
[t.join() for t in ths]
dt_cpu = time.perf_counter() - t0
print("CPU  %s StrippedSW on %d threads: %.1f ms -> %.2f M pairs/s, %.1f G cell updates/s" % (kind, threads, dt_cpu * 1e3, n / dt_cpu / 1e6, cells / dt_cpu / 1e9))
print("parity: %d/%d" % (int((got == want).sum()), n))
sw.close()
