"""Dev tool (run on the GPU box's host): the reference's ConvexAlignFast (oracle/_ref, oracle_align_many: C++
threads inside one call) on N host threads for several N -- how does the CPU baseline of bench.py scale?"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from ngmlr_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

ts = synth.pacbio_tileset(1024, seed=7)
lib = C.CDLL(pyoracle.REF_SO)
lib.oracle_align_many.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 11
for threads in [int(x) for x in sys.argv[1:]] or [1, 16, 32, 64, 128, 256]:
    m = min(len(ts), max(4, threads * 3))
    idx = np.arange(m)
    tab = ts.table()[idx]
    caps = (4 * ts.H[idx] + 4 * ts.W[idx] + 256).astype(np.int32)
    toff = np.concatenate([[0], np.cumsum(2 * caps.astype(np.int64))]).astype(np.uint64)
    text = np.zeros(int(toff[-1]) + 16, dtype=np.uint8)
    outs = (pyoracle.OracleOut * m)()
    busy = np.zeros(threads, dtype=np.float64)
    params = (C.c_float * 6)(*pyoracle.DEFAULT_PARAMS)
    cols = [np.ascontiguousarray(tab[k]) for k in ("ref", "ref_len", "qry", "qry_len", "row_offset", "row_length")]
    t_in = np.ascontiguousarray(toff[:-1])
    t0 = time.perf_counter()
    lib.oracle_align_many(params, threads, m, cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data, cols[3].ctypes.data,
                          cols[4].ctypes.data, cols[5].ctypes.data, C.addressof(outs), text.ctypes.data, t_in.ctypes.data,
                          caps.ctypes.data, busy.ctypes.data)
    dt = time.perf_counter() - t0
    bases = int(ts.H[idx].sum())
    cells = int(sum(int(ts.row_length[ts.qry_off[i]:ts.qry_off[i + 1]].astype(np.int64).sum()) for i in idx))
    print("reference on %3d C++ threads: %4d tiles in %5.2f s -> %6.1f Gbp/h, %.2e cells/s per busy thread-second" % (
        threads, m, dt, bases / dt * 3.6e-6, cells / busy.sum()), flush=True)
