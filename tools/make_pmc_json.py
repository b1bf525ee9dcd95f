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
DOM = "void cvx::fill_ring_kernel<3, false, 0, true, 1>(cvx::FillArgs)"      # the two-phase float-score instantiation with the LDS penalty table


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


def all_kernels(path, prefix):
    """{kernel name: {counter: largest dispatch, "ms": median duration}} for every kernel whose name starts with prefix"""
    out = {}
    for line in open(path):
        if not line.startswith(prefix):
            continue
        # (the summary pads / cuts kernel names to 70 columns; names hold blanks and brackets of their own)
        name, f = line[:70].rstrip(), line[70:].split()
        if not f:
            continue
        e = out.setdefault(name, {})
        if len(f) == 5 and re.match(r"^[A-Z_0-9]+$", f[0]):
            e[f[0]] = float(f[4])
        elif len(f) >= 8 and "ms" not in e:
            e["ms"] = float(f[3]) / 1e3
            e["ms_largest"] = float(f[5]) / 1e3      # the longest dispatch: the one `largest` counters belong to when a kernel runs at several sizes
            e["calls"] = int(f[0])
    return out


def class_report(stats_txt, sq_txt, ab_log):
    """per fill kernel of one of the other configs: duration, instructions per slot-step, wave-state shares"""
    cells = {}
    for l in open(ab_log):
        m = re.match(r"\s+M=(\d) tasks/waves=(\d+) wrap=(\d):\s+(\d+) tiles\s+([0-9.]+) ms\s+(\d+) G cells/s", l)
        if m:
            cells[(int(m.group(1)), int(m.group(2)) > 1)] = {"tiles": int(m.group(4)), "ms_in_the_batch": float(m.group(5)), "G_cells_per_s_in_the_batch": float(m.group(6))}
    sq = all_kernels(sq_txt, "void cvx::fill_ring_kernel")
    st = all_kernels(stats_txt, "void cvx::fill_ring_kernel")
    out = {}
    for name, c in sq.items():
        m = re.search(r"<(\d), (true|false), (\d)(?:, (true|false))?(?:, (\d))?>", name)
        if not m or "SQ_WAVE_CYCLES" not in c:
            continue
        M, mode = int(m.group(1)), int(m.group(3))
        G = int(m.group(5)) if m.group(5) else 1
        key = "M=%d %s%s%s%s" % (M, {0: "two-phase", 1: "exact", 2: "chained row blocks"}[mode], " (int16 runs)" if m.group(2) == "true" else "",
                               " with the penalty table" if m.group(4) == "true" else "", (" x %d waves" % G) if G > 1 else "")
        e = {"kernel": name, "kernel_ms_alone_in_the_trace": st.get(name, {}).get("ms"),
             "valu_wave_insts": c.get("SQ_INSTS_VALU"), "salu_wave_insts": c.get("SQ_INSTS_SALU"),
             "valu_per_salu": (c["SQ_INSTS_VALU"] / c["SQ_INSTS_SALU"]) if c.get("SQ_INSTS_SALU") else None,
             "wave_state_share": {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if k in c},
             "valu_insts_per_wave_quad_cycle": c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVE_CYCLES"]}
        if mode != 1:
            e.update(cells.get((M, mode == 2), {}))
        if e.get("tiles") and "G_cells_per_s_in_the_batch" in e:
            cells_total = e["G_cells_per_s_in_the_batch"] * 1e9 * e["ms_in_the_batch"] * 1e-3
            # wave-instructions per 64 corridor cells (one per cell and lane if every slot-step were a real cell)
            e["valu_wave_insts_per_64_cells"] = c.get("SQ_INSTS_VALU", 0.0) / (cells_total / 64.0)
            e["salu_wave_insts_per_64_cells"] = c.get("SQ_INSTS_SALU", 0.0) / (cells_total / 64.0)
        out[key] = e
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
    lib_ = capi.load()
    d = {"build_id": build_id, "source_ids": {"fill": lib_.cvx_source_id(b"fill").decode(), "search": lib_.cvx_source_id(b"search").decode()}, "_comment": "rocprofv3 --pmc passes of tools/collect_profiles.sh %s (separate runs, --kernel-trace only); largest dispatch of "
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
    for w in ("ont", "c5"):
        if os.path.exists(g("%s_sq.txt" % w)) and os.path.exists(g("%s_stats.txt" % w)):
            d["%s_mix_fill_classes" % w] = class_report(g("%s_stats.txt" % w), g("%s_sq.txt" % w), g("%s_sq.log" % w))
            d["%s_mix_fill_classes" % w]["_comment"] = ("tools/ab_knobs.py %s (one batch, inputs resident, four runs) under rocprofv3: largest dispatch per fill kernel; "
                                                        "*_wave_insts_per_64_cells = wave-instructions per 64 corridor cells (idle slots and ramps included: 1 slot-step = 64 cells at best); the classes "
                                                        "of a batch run side by side on three streams" % w)
    if os.path.exists(g("search_fetch.txt")) and os.path.exists(g("search_big.json")):
        try:
            sb = json.load(open(g("search_big.json")))
            f = all_kernels(g("search_fetch.txt"), "cvx::search")
            f.update(all_kernels(g("search_fetch.txt"), "void cvx::search"))
            wr = all_kernels(g("search_write.txt"), "cvx::search")
            wr.update(all_kernels(g("search_write.txt"), "void cvx::search"))
            fetch_kib = sum(v.get("FETCH_SIZE", 0.0) for v in f.values())
            write_kib = sum(v.get("WRITE_SIZE", 0.0) for v in wr.values())
            votes = sb["votes_per_sub_read"] * sb["sub_reads"]
            hbm = (2.0 * fetch_kib + write_kib) * 1024
            d["candidate_search_big"] = {"sub_reads": sb["sub_reads"], "votes": votes, "fetch_size_kib": fetch_kib, "write_size_kib": write_kib, "fetch_correction": 2.0,
                                         "hbm_bytes": hbm, "hbm_bytes_per_vote": hbm / votes, "hbm_bytes_per_sub_read": hbm / sb["sub_reads"],
                                         "kernel_ms": sb["kernel_ms"], "hbm_GB_per_s_over_kernel_time": hbm / (sb["kernel_ms"] * 1e-3) / 1e9,
                                         "kernels": {k: {"fetch_kib": v.get("FETCH_SIZE"), "ms": v.get("ms_largest", v.get("ms"))} for k, v in f.items()},
                                         "_comment": "tools/search_rates.py --big 512 100000 under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; largest dispatch "
                                                     "of every search kernel = the 100 000-read call), FETCH doubled as for the fill"}
        except Exception as e:
            d["candidate_search_big"] = {"error": str(e)}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
