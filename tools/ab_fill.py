#!/usr/bin/env python3
"""A/B of several builds of libcvxalign.so in ONE process on the same resident batch (GPU).

    python tools/ab_fill.py TILES [workload] name1 name2 ...   (name: default | a build under ngmlr_amd/variants/)

Tiles are generated once; every build uploads them, runs warm-up + 3 timed runs and prints the
dominant fill launch, the stage times and how many tiles needed the exact-tracking redo pass.
Results of every build are compared with those of the first one on ALL tiles (score bits, status,
best cell, every op), and the first build with the CPU oracle on a few tiles.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ngmlr_amd import synth  # noqa: E402
from ngmlr_amd.aligner import ConvexAlignHip, format_alignment  # noqa: E402


def lib_path(name):
    return None if name == "default" else os.path.join(ROOT, "ngmlr_amd", "variants", "libcvxalign_%s.so" % name)


def main():
    n = int(sys.argv[1])
    args = sys.argv[2:]
    wl = "pacbio"
    if args and args[0] in ("pacbio", "ont", "short", "ultralong"):
        wl = args.pop(0)
    gen = {"pacbio": synth.workload_pacbio, "ont": synth.workload_ont, "short": synth.workload_short,
           "ultralong": synth.workload_ultralong_sv}[wl]
    t0 = time.time()
    tiles = gen(n)
    bases = sum(t.H for t in tiles)
    print("%d %s tiles, %.1f Mbp, generated in %.1f s" % (n, wl, bases / 1e6, time.time() - t0), flush=True)
    ref = None
    for name in args:
        try:
            al = ConvexAlignHip(lib_path=lib_path(name))
            batch = al.upload(tiles)
            batch.run()
            tms = [batch.run() for _ in range(3)]
            launches = batch.launches()
            res, ops = batch.download()
            key = (np.array([(np.float32(r.score).view(np.uint32), r.status, r.best_ref_index, r.best_read_index,
                              r.ref_position, r.qstart, r.qend, r.n_ops) for r in res[:n]], dtype=np.int64), ops.copy())
            dom = max(launches, key=lambda l: l["alg_bytes"])
            msg = "%-10s dom M%d/NW%d %8.3f ms (%5.0f Gcell/s)  plan %.2f fill %.2f bt %.2f total %.2f ms  -> %6.0f Gbp/h dev" % (
                name, dom["slots_per_lane"], dom["waves"], dom["ms"], dom["cells"] / dom["ms"] / 1e6,
                np.mean([t.plan_ms for t in tms]), np.mean([t.fill_ms for t in tms]),
                np.mean([t.backtrack_ms for t in tms]), np.mean([t.total_ms for t in tms]),
                bases / np.mean([t.total_ms for t in tms]) * 3.6e-6)
            if ref is None:
                ref = key
                from oracle.pyoracle import Oracle, same_alignment
                orc = Oracle("port")
                bad = 0
                for i in range(min(6, n)):
                    want = orc.align(tiles[i], want_nm=False)
                    got = format_alignment(al.lib, res[i], ops, tiles[i], False)
                    if same_alignment(want, got, keys=("ret", "score_bits", "position_offset", "qstart", "qend", "nm", "cigar", "md")) is not None:
                        bad += 1
                msg += "  | oracle mismatches %d/6" % bad
            else:
                same = np.array_equal(key[0], ref[0]) and np.array_equal(key[1], ref[1])
                ndiff = int((key[0] != ref[0]).any(axis=1).sum()) if key[0].shape == ref[0].shape else -1
                msg += "  | vs first: %s (%d tiles differ)" % ("IDENTICAL" if same else "DIFFERENT", ndiff)
            valid = int((key[0][:, 1] == 0).sum())
            msg += "  valid %d/%d redone %d" % (valid, n, tms[-1].n_tiles_redone)
            print(msg, flush=True)
            batch.free()
            al.close()
        except Exception as e:  # keep going: a broken variant must not hide the others
            print("%-10s FAILED: %s" % (name, e), flush=True)


if __name__ == "__main__":
    main()
