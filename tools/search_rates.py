"""Dev tool (GPU box): candidate search (cvx_search_batch) -- reads per second on the device against the CPU checker
(oracle/cs_oracle.c, one thread) on the recorded test_3 index and sub-reads, replicated to a batch that fills the device.
    search_rates.py [replicas]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ngmlr_amd.aligner import ConvexAlignHip, KmerIndex
from oracle.pyoracle import SearchFixture, SearchOracle

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fx = SearchFixture(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cs_test_3.npz"))
reads = [s for s in fx.seqs if len(s) > 0]
bases = sum(len(s) for s in reads)
print("index: k=%d, %d locations; %d recorded sub-reads, %.2f Mbp, mean %.0f bp" % (fx.k, len(fx.locs), len(reads), bases / 1e6, bases / len(reads)))
o = SearchOracle(fx)
t0 = time.perf_counter()
n_c = 0
for s in reads[:2000]:
    n_c += max(o.search(s)["n"], 0)
dt_cpu = time.perf_counter() - t0
o.close()
print("CPU checker (1 thread): %d reads in %.2f s = %.0f reads/s, %d candidates" % (min(len(reads), 2000), dt_cpu, min(len(reads), 2000) / dt_cpu, n_c))
al = ConvexAlignHip()
idx, locs = fx.index_arrays()
ix = KmerIndex(al, fx.k, idx, locs, fx.unit_offset)
for r in (1, rep):
    batch = reads * r
    ix.search(batch[:64])
    t0 = time.perf_counter()
    got = ix.search(batch)
    dt = time.perf_counter() - t0
    print("device: %6d reads (%.1f Mbp) in %7.1f ms = %9.0f reads/s = %.2f Gbp/h of sub-read bases (whole call: marshalling, tables, kernels, lists back)" % (
        len(batch), bases * r / 1e6, dt * 1e3, len(batch) / dt, bases * r / dt * 3.6e-6))
ix.free()
al.close()
