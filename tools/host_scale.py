#!/usr/bin/env python3
"""tools/host_scale.py [packers] [tiles] -- the submit side of an N-device node without any device.

N host threads (one per "device", as bench.py --gpus N drives them) call cvx_pack_probe concurrently: exactly what
cvx_submit does on the host -- batch layout, packing of whatever has to be packed, on the process's shared pack
threads -- with every HIP call left out.  Reports steps per second per packer and in aggregate for the two ways a
caller can hand a batch over:

  arrays    corridor row arrays + sequences in ordinary memory (what the IAlignment shim has to do: ngmlr owns
            both): every base is copied into staging, every row read and turned into a step byte
  closed    corridors as the builders' closed forms + sequences back to back in a page-locked arena
            (cvx_host_alloc; assumed here -- there is no device to allocate it): nothing is packed, the host writes
            64 bytes per tile

A 49 152-tile PacBio step takes ~127 ms on one MI355X, so a node needs ~8 steps/s per device from its host."""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from ngmlr_amd import capi, synth  # noqa: E402


def main():
    packers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
    lib = capi.load()
    # one tile set per packer, as in the bench (distinct memory: no cache sharing between "devices")
    base = synth.pacbio_tileset(min(n_tiles, 2048), seed=7)
    reps = (n_tiles + len(base) - 1) // len(base)
    idx = np.tile(np.arange(len(base)), reps)[:n_tiles]
    sets = [base.subset(idx) for _ in range(packers)]
    print("%d packers x %d tiles (%.2f Gbp per step each), %d hardware threads, pack pool %s" % (
        packers, n_tiles, sets[0].read_bases / 1e9, os.cpu_count(), os.environ.get("CVX_PACK_THREADS", "default (min(hw, 16))")))
    for mode in ("arrays", "closed"):
        for ts in sets:
            ts.use_closed_form(mode == "closed")
        tabs = [ts.table() for ts in sets]
        iters = 3 if mode == "arrays" else 50
        out = [None] * packers

        def work(k):
            ms = C.c_double()
            touched = C.c_uint64()
            capi.check(lib.cvx_pack_probe(len(tabs[k]), tabs[k].ctypes.data, iters, 1 if mode == "closed" else 0, C.byref(ms), C.byref(touched)))
            out[k] = (ms.value, touched.value)
        for n_par in sorted({1, packers}):
            ths = [threading.Thread(target=work, args=(k,)) for k in range(n_par)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            wall = time.perf_counter() - t0
            ms = [out[k][0] for k in range(n_par)]
            print("  %-6s %d packer(s): %8.2f ms per step per packer (max %8.2f) = %7.1f steps/s each, %8.1f steps/s aggregate; host bytes touched per step %.3f GB"
                  % (mode, n_par, float(np.mean(ms)), max(ms), 1e3 / max(ms), n_par * iters / wall, out[0][1] / 1e9))


if __name__ == "__main__":
    main()
