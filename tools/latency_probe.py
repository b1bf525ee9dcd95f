"""Dev tool (GPU box): how long ONE small launch takes -- the shape ngmlr's pipeline hands the BatchingAligner:
n PacBio 10 kb tiles through cvx_align_batch (host strings in -> results out), wall clock and the stage times.
    latency_probe.py [n ...]        env CVX_TUNE_LONG_STEPS / CVX_TUNE_SMALL_BATCH / CVX_TUNE_CHAIN_M select the chaining rule"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ngmlr_amd import synth
from ngmlr_amd.aligner import ConvexAlignHip

al = ConvexAlignHip()
ns = [int(x) for x in sys.argv[1:]] or [1, 4, 16, 32, 64, 256]
tiles = synth.workload_pacbio(max(ns), seed=3)
al.batch_align(tiles[:2])
for n in ns:
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        job = al.submit(tiles[:n])
        job.wait()
        walls.append(time.perf_counter() - t0)
        tm = job.timing()
        li = job.launches()
        job.release()
    print("n=%4d  submit+wait wall %7.2f ms (best %7.2f)   device: plan %.2f fill %.2f bt %.2f total %.2f ms   classes %s" % (
        n, 1e3 * float(np.median(walls)), 1e3 * min(walls), tm.plan_ms, tm.fill_ms, tm.backtrack_ms, tm.total_ms,
        ["M%d x%d %s %.2fms" % (l["slots_per_lane"], l["n_tiles"], "chain" if l.get("kind") == 2 else "gang" if l.get("kind") == 1 else "ring", l["ms"]) for l in li]), flush=True)
al.close()
