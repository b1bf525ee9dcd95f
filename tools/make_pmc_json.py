#!/usr/bin/env python3
"""tools/make_pmc_json.py TAG OUT.json -- condense the PMC passes of tools/collect_profiles.sh
(gpurun_out/TAG_{fetch,write,sq,sq2}.{txt,log}) into the small JSON bench.py reads for
`roofline.traffic` and DESIGN.md quotes.  FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 reports them;
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (64 B counted per 128 B request)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOM = "void cvx::fill_ring_kernel<3, false, 0>(cvx::FillArgs)"


def counters(path, kernel=DOM):
    out = {}
    for line in open(path):
        if line.startswith(kernel):
            f = line[len(kernel):].split()
            if len(f) == 5 and re.match(r"^[A-Z_0-9]+$", f[0]):
                out[f[0]] = float(f[4])            # largest dispatch
            elif len(f) >= 8 and "ms" not in out:
                out["ms"] = float(f[3]) / 1e3       # median duration, us -> ms
    return out


def bench_meta(path):
    txt = open(path).read()
    m = re.search(r'"launch_tiles": (\d+), "alg_bytes_per_launch": (\d+)', txt)
    g = re.search(r'"gcups": ([0-9.]+)', txt)
    return (int(m.group(1)), int(m.group(2))) if m else (0, 0)


def main():
    tag, out = sys.argv[1], sys.argv[2]
    g = lambda s: os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, s))  # noqa: E731
    fetch, write = counters(g("fetch.txt")), counters(g("write.txt"))
    tiles, alg = bench_meta(g("fetch.log"))
    hbm = int((2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024)
    sys.path.insert(0, ROOT)
    from ngmlr_amd import capi
    build_id = capi.load().cvx_build_id().decode()      # the library the passes ran on (same snapshot)
    d = {"build_id": build_id, "_comment": "rocprofv3 --pmc passes of tools/collect_profiles.sh %s (separate runs, --kernel-trace only); largest dispatch of "
                     "the dominant fill kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); hbm_bytes = 2*FETCH + WRITE. "
                     "bench.py scales hbm_bytes to its own launch by algorithmic bytes and says so." % tag,
         "fill_ring_kernel<M=3,NW=1,wrap16=0>": {"tiles": tiles, "alg_bytes": alg, "fetch_size_kib": fetch["FETCH_SIZE"],
                                                  "write_size_kib": write["WRITE_SIZE"], "fetch_correction": 2.0, "hbm_bytes": hbm,
                                                  "hbm_bytes_per_alg_byte": hbm / alg}}
    sq = counters(g("sq.txt"))
    sq2 = counters(g("sq2.txt")) if os.path.exists(g("sq2.txt")) else {}
    tiles_sq, alg_sq = bench_meta(g("sq.log"))
    if sq:
        simd_cycles = sq["ms"] * 1e-3 * 2.4e9 * 1024
        e = {"tiles": tiles_sq, "alg_bytes": alg_sq, "kernel_ms": sq["ms"],
             "valu_wave_insts": sq.get("SQ_INSTS_VALU"), "salu_wave_insts": sq.get("SQ_INSTS_SALU"),
             "wave_quad_cycles": sq.get("SQ_WAVE_CYCLES"),
             "wave_state_share": {k: sq[k] / sq["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if k in sq},
             "mean_waves_per_simd": sq["SQ_WAVE_CYCLES"] * 4 / simd_cycles if "SQ_WAVE_CYCLES" in sq else None,
             "simd_cycles_at_2p4GHz": simd_cycles}
        if "SQ_THREAD_CYCLES_VALU" in sq2:
            # thread-quad-cycles the VALU spent executing: / 64 lanes * 4 clocks = VALU-busy clocks summed over SIMDs
            busy = sq2["SQ_THREAD_CYCLES_VALU"] / 64.0 * 4.0
            e["valu_busy_clocks"] = busy
            e["valu_busy_frac"] = busy / (sq2["ms"] * 1e-3 * 2.4e9 * 1024)
            e["valu_clocks_per_inst"] = busy / sq2["SQ_INSTS_VALU"]
        d["sq_pass"] = e
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
