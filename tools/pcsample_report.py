#!/usr/bin/env python3
"""tools/pcsample_report.py SAMPLES [top] -- per-function CPU shares from a CVX_PC_SAMPLE=<file> run of an ngmlr_hip_* binary
(ngmlr_amd/csrc/cvx_pcsample.h: one sample per millisecond of a thread's own CPU time; carriers / contexts, CS threads and the
dispatcher are sampled).  PCs are resolved against the mapped files with `nm -C -n` (function granularity, self time)."""
import bisect
import collections
import os
import struct
import subprocess
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    raw = open(path, "rb").read()
    n = struct.unpack_from("<Q", raw, 0)[0]
    pcs = struct.unpack_from("<%dQ" % n, raw, 8)
    maps = raw[8 + 8 * n:].decode("utf-8", "replace").splitlines()
    segs = []
    base = {}                 # file -> load base: the lowest address any of its segments is mapped at (symbol values are relative to it)
    for l in maps:
        f = l.split()
        if len(f) < 6:
            continue
        lo, hi = (int(x, 16) for x in f[0].split("-"))
        base[f[5]] = min(base.get(f[5], lo), lo)
        if "x" in f[1]:
            segs.append((lo, hi, int(f[2], 16), f[5]))
    segs.sort()
    syms = {}

    def table(fn):
        if fn not in syms:
            rows = []
            try:
                out = subprocess.run(["nm", "-C", "-n", "--defined-only", fn], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True).stdout
                if not out.strip():
                    out = subprocess.run(["nm", "-C", "-n", "-D", "--defined-only", fn], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True).stdout
                for l in out.splitlines():
                    p = l.split(None, 2)
                    if len(p) == 3 and p[1] in "tTwW":
                        rows.append((int(p[0], 16), p[2]))
            except OSError:
                pass
            rows.sort()
            syms[fn] = ([r[0] for r in rows], [r[1] for r in rows])
        return syms[fn]
    names = {1: "contexts (carriers)", 2: "CS threads", 3: "dispatcher", 4: "other"}
    by_class = collections.defaultdict(collections.Counter)
    for v in pcs:
        cls, pc = v >> 56, v & ((1 << 56) - 1)
        who = "?"
        i = bisect.bisect_right(segs, (pc, 1 << 62, 0, "")) - 1
        if i >= 0 and segs[i][0] <= pc < segs[i][1]:
            lo, hi, off, fn = segs[i]
            addrs, nm = table(fn)
            rel = pc - base[fn]          # PIE / shared object: symbol values are relative to the load base
            if addrs and addrs[0] >= base[fn]:
                rel = pc                 # a non-PIE executable: absolute symbol values
            j = bisect.bisect_right(addrs, rel) - 1
            who = "%s: %s" % (os.path.basename(fn), nm[j][:110] if j >= 0 and nm else "?")
        by_class[cls][who] += 1
    print("%d samples (1 ms of thread CPU time each)" % n)
    for cls in sorted(by_class):
        c = by_class[cls]
        tot = sum(c.values())
        print("\n%s: %d ms" % (names.get(cls, "class %d" % cls), tot))
        libs = collections.Counter()
        for k, v in c.items():
            libs[k.split(":")[0]] += v
        print("  by mapped file: " + ", ".join("%s %.1f %%" % (k, 100.0 * v / tot) for k, v in libs.most_common(8)))
        for k, v in c.most_common(top):
            print("  %5.1f %%  %s" % (100.0 * v / tot, k))


if __name__ == "__main__":
    main()
