#!/usr/bin/env python3
"""One PROCESS of the CPU baseline (test infrastructure, like everything under oracle/): the reference's own
ConvexAlignFast (oracle/_ref/libcvx_oracle_ref.so, or the C restatement) on tiles p, p + P, p + 2P ... of a tile
set saved by bench.py, single-threaded.  BASELINE.md section 2/3 measured the reference as independent processes;
its threads inside one process share an allocator and the kernel's page-fault path (AlignmentMatrixFast::clean()
gives ~270 MB back per instance) and stop scaling at ~32 on the bench host.

    ref_proc_worker.py tile_dir p P go_file     -> prints "<seconds> <bases> <cells>" after the go file appears"""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    path, p, P, go = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from oracle import pyoracle
    # memory-mapped .npy files: every worker shares the page cache's single copy of the tile set
    ld = lambda k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r")  # noqa: E731
    ref, qry, roff, rlen = ld("ref"), ld("qry"), ld("row_offset"), ld("row_length")
    ref_off, qry_off = np.asarray(ld("ref_off")), np.asarray(ld("qry_off"))
    idx = np.arange(p, len(ref_off) - 1, P)
    m = len(idx)
    W = (ref_off[1:] - ref_off[:-1])[idx].astype(np.int32)
    H = (qry_off[1:] - qry_off[:-1])[idx].astype(np.int32)
    refp = (ref.ctypes.data + ref_off[:-1][idx]).astype(np.uint64)
    qryp = (qry.ctypes.data + qry_off[:-1][idx]).astype(np.uint64)
    rop = (roff.ctypes.data + 4 * qry_off[:-1][idx]).astype(np.uint64)
    rlp = (rlen.ctypes.data + 4 * qry_off[:-1][idx]).astype(np.uint64)
    caps = (4 * H.astype(np.int64) + 4 * W.astype(np.int64) + 256).astype(np.int32)
    toff = np.concatenate([[0], np.cumsum(2 * caps.astype(np.int64))]).astype(np.uint64)
    text = np.zeros(int(toff[-1]) + 16, dtype=np.uint8)
    outs = (pyoracle.OracleOut * max(m, 1))()
    busy = np.zeros(1, dtype=np.float64)
    params = (C.c_float * 6)(*pyoracle.DEFAULT_PARAMS)
    lib = C.CDLL(pyoracle.REF_SO if pyoracle.have_ref() else pyoracle.PORT_SO)
    lib.oracle_align_many.restype = C.c_int
    lib.oracle_align_many.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 11
    toff_in = np.ascontiguousarray(toff[:-1])
    print("ready", flush=True)
    while not os.path.exists(go):
        time.sleep(0.002)
    t0 = time.perf_counter()
    threw = lib.oracle_align_many(params, 1, m, refp.ctypes.data, W.ctypes.data, qryp.ctypes.data, H.ctypes.data,
                                  rop.ctypes.data, rlp.ctypes.data, C.addressof(outs), text.ctypes.data, toff_in.ctypes.data,
                                  caps.ctypes.data, busy.ctypes.data)
    dt = time.perf_counter() - t0
    cells = int(sum(int(rlen[qry_off[i]:qry_off[i + 1]].astype(np.int64).sum()) for i in idx))
    print("%.6f %d %d %d" % (dt, int(H.sum()), cells, threw), flush=True)


if __name__ == "__main__":
    main()
