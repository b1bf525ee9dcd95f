"""Dev tool (GPU box): one set of tiles, several settings of the runtime's CVX_TUNE_* knobs (read at cvx_create), device-resident
rate per fill class for each -- and a digest of every result record, so that two settings (or two builds of the library:
CVX_LIB=ngmlr_amd/variants/libcvxalign_X.so) can be compared alignment by alignment without an oracle run.

    ab_knobs.py WORKLOAD[,WORKLOAD...] [N_TILES] -- "K=V K=V" "K=V" ...

WORKLOAD: pacbio | ont | ont_wide | c5 | short.  An empty setting ("") is the default configuration.  The first setting is the yardstick:
every later one reports how many of its result records (status, score bits, best cell, path end points, op count, ops) differ."""
import hashlib
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def gen(what, n, pool):
    from ngmlr_amd import synth
    if what == "pacbio":
        return synth.parallel_workload("pacbio", n or 24576, 7, pool, chunk=64)
    if what == "ont":
        return synth.parallel_workload("ont", n or 60000, 11, pool)
    if what == "ont_wide":      # the ONT mix's retries alone (corridor multiplier 2: 257-576 live rows -- gangs of waves, or chained row blocks)
        return [t for t in synth.parallel_workload("ont", n or 60000, 11, pool) if int(t.row_length[0]) > 520]
    if what == "c5":
        return synth.parallel_workload("ultralong_mix", n or 4096, 19, pool, chunk=16)
    if what == "short":
        return synth.parallel_workload("short", n or 60000, 17, pool, chunk=2048)
    raise SystemExit("unknown workload " + what)


def main():
    args = sys.argv[1:]
    if "--" not in args:
        raise SystemExit(__doc__)
    k = args.index("--")
    head, settings = args[:k], args[k + 1:]
    workloads = head[0].split(",")
    n = int(head[1]) if len(head) > 1 else 0
    with ProcessPoolExecutor(min(os.cpu_count() or 1, 64), mp_context=mp.get_context("fork")) as pool:
        sets = {w: gen(w, n, pool) for w in workloads}
    from ngmlr_amd.aligner import ConvexAlignHip
    print("library: %s" % (os.environ.get("CVX_LIB") or "in-tree default"), flush=True)
    base = {}
    for st in settings:
        env = dict(kv.split("=", 1) for kv in st.split()) if st.strip() else {}
        for kk, vv in env.items():
            os.environ[kk] = vv
        try:
            al = ConvexAlignHip()
        finally:
            for kk in env:
                os.environ.pop(kk, None)
        for w, tiles in sets.items():
            bases = float(sum(t.H for t in tiles))
            b = al.upload(tiles, closed_form=True)
            try:
                b.run()
                best = None
                for _ in range(3):
                    tm = b.run()
                    if best is None or tm.total_ms < best.total_ms:
                        best = tm
                res, ops = b.download()
                rec = np.frombuffer(res, dtype=np.dtype([("score", "<u4"), ("status", "<i4"), ("bx", "<i4"), ("by", "<i4"), ("rp", "<i4"), ("qs", "<i4"),
                                                         ("qe", "<i4"), ("n_ops", "<i4"), ("ops_begin", "<u8"), ("cells", "<u8")]))
                opsv = np.frombuffer(ops, dtype=np.uint32) if ops is not None and len(ops) else np.zeros(0, np.uint32)
                per = []
                for i in range(len(tiles)):
                    r = rec[i]
                    hh = hashlib.blake2b(digest_size=8)
                    hh.update(r.tobytes()[:32])
                    if r["status"] == 0 and r["n_ops"] > 0:
                        hh.update(opsv[int(r["ops_begin"]):int(r["ops_begin"]) + int(r["n_ops"])].tobytes())
                    per.append(hh.digest())
                diff = ""
                if w in base:
                    nd = sum(1 for x, y in zip(per, base[w]) if x != y)
                    diff = "  results differing from the first setting: %d of %d" % (nd, len(per))
                else:
                    base[w] = per
                print("[%s] %-6s %6d tiles: plan %.2f fill %.2f bt %.2f total %.2f ms -> %.0f Gbp/h, redone %d, valid %d%s" % (
                    st or "default", w, len(tiles), best.plan_ms, best.fill_ms, best.backtrack_ms, best.total_ms, bases / best.total_ms * 3.6e-3,
                    best.n_tiles_redone, int((rec["status"] == 0).sum()), diff), flush=True)
                for li in b.launches():
                    print("        M=%d tasks/waves=%d wrap=%d: %6d tiles %8.2f ms %7.0f G cells/s" % (
                        li["slots_per_lane"], li["waves"], li["wrap16"], li["n_tiles"], li["ms"], li["cells"] / max(li["ms"], 1e-6) * 1e-6), flush=True)
                print("        digest of all records: %s" % hashlib.blake2b(b"".join(per), digest_size=8).hexdigest(), flush=True)
            finally:
                b.free()
        al.close()


if __name__ == "__main__":
    main()
