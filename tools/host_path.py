"""Dev tool: stage times of the host-buffer path (upload / run / download / free) on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngmlr_amd import synth
from ngmlr_amd.aligner import ConvexAlignHip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tiles = synth.workload_pacbio(n, seed=7)
al = ConvexAlignHip()
for rep in range(3):
    r = al.timed_host_path(tiles)
    bases = sum(t.H for t in tiles)
    print("n=%d rep %d: " % (n, rep) + " ".join("%s=%.4f" % kv for kv in r.items()) + "  -> %.0f Gbp/h" % (bases / r["total_s"] * 3.6e-6))
al.close()

# sustained rate with several handles (one per host thread, as ngmlr's workers would hold them):
# uploads of one handle overlap the kernels of another.  The tile table is packed once; only the
# C calls run inside the timed loops (ctypes releases the GIL).
import threading
import ctypes as C
from ngmlr_amd import capi
from ngmlr_amd.aligner import DeviceBatch


def c_path(a, arr, n_tiles):
    b = C.c_void_p()
    capi.check(a.lib.cvx_batch_upload(a.h, n_tiles, arr, C.byref(b)))
    capi.check(a.lib.cvx_batch_run(a.h, b))
    total = C.c_uint64()
    capi.check(a.lib.cvx_batch_ops_total(b, C.byref(total)))
    res = (capi.CvxResult * n_tiles)()
    import numpy as np
    ops = np.zeros(max(int(total.value), 1), dtype=np.uint32)
    used = C.c_uint64()
    capi.check(a.lib.cvx_batch_download(a.h, b, res, ops.ctypes.data, len(ops), C.byref(used)))
    a.lib.cvx_batch_free(a.h, b)


for nh in (1, 2, 3):
    als = [ConvexAlignHip() for _ in range(nh)]
    packed = [a._pack(tiles) for a in als]
    for a, (arr, keep) in zip(als, packed):
        c_path(a, arr, len(tiles))              # pins the staging, warms the allocator
    reps = 4

    def work(a, arr):
        for _ in range(reps):
            c_path(a, arr, len(tiles))
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(a, arr)) for a, (arr, keep) in zip(als, packed)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print("n=%d, %d handles x %d batches: %.3f s -> %.0f Gbp/h sustained, host buffers in, results out" % (
        n, nh, reps, dt, nh * reps * sum(t.H for t in tiles) / dt * 3.6e-6))
    for a in als:
        a.close()
