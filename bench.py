#!/usr/bin/env python3
"""bench.py -- aligned Gbp/hour of the convex-gap banded SW hot path on MI355X.

One "step" = one batch of synthetic tiles through the whole hot path, HOST BUFFERS IN ->
RESULTS OUT: cvx_submit packs the caller's sequences and corridor rows into pinned
staging and uploads them, the device plans, fills, backtracks and compacts, cvx_wait hands back
the result records and run-length ops in host memory.  Steps are pipelined three deep per device
(upload of step k+1 and download of step k-1 under the kernels of step k), exactly what a
batching driver in front of ngmlr's workers would do.  Workload = BASELINE.json configs[1]:
synthetic PacBio-like 10 kb reads (15 % error, ins:del:sub 6:3:1) against a seeded uniform-ACGT
reference (GRCh38/pbsim are not available offline), anchors corridor (width 309-369), scoring
-x pacbio defaults.

    python bench.py --gpus N --steps K --warmup W        (one host thread + one handle per device)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per device)

Reads shard across devices with no data-path collective (tiles are independent), so the scaling
is weak: every device aligns --tiles tiles per step; `value` is the whole-job aggregate.  Rank 0
prints ONE JSON line.  `roofline` is the dominant fill kernel of the timed steps (HIP events on
the stream it runs on); `device_resident` is the same step with inputs already in HBM;
`cpu_baseline` is the reference's own ConvexAlignFast on the host cores over a bounded sample of
the same tiles, every one of which is also compared with the GPU result (`parity`).
"""
from __future__ import annotations

import argparse
from ctypes import c_char_p as C_char_p
import json
import multiprocessing as mp
import os
import sys
import threading
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PARITY_KEYS = ("ret", "score_bits", "position_offset", "qstart", "qend", "nm", "alignment_length",
               "cigar_op_count", "sv_type", "first_ref", "first_read", "last_ref", "last_read", "cigar", "md")


def effective_cores():
    """What this process may really use: hardware threads it can see, its affinity mask, and the cgroup's CPU quota
    (cpu.max = quota / period; the GPU boxes behind gpurun expose 256 hardware threads and allow 16 cores' worth of time)."""
    out = {"hardware_threads": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)), "cgroup_quota_cores": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["cgroup_quota_cores"] = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            out["cgroup_quota_cores"] = None if q < 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except Exception:
            pass
    eff = min(out["hardware_threads"], out["affinity"])
    if out["cgroup_quota_cores"]:
        eff = min(eff, out["cgroup_quota_cores"])
    out["effective"] = eff
    return out


def cpu_baseline_and_parity(al, ts, results, ops, seconds_budget: float):
    """The step's tiles through the CPU checker on the host cores of this box (bounded sample), and
    every one of those alignments compared with the GPU's.  The reference's own ConvexAlignFast
    (oracle/_ref, kind "reference") runs inside ONE C call on a C++ thread per core (oracle_align_many:
    python threads around the per-tile call serialise on the interpreter lock and stop scaling at ~16
    threads); without oracle/_ref the C restatement (kind "port") runs on python threads."""
    import ctypes as C
    from ngmlr_amd.aligner import format_tileset
    from oracle import pyoracle
    from oracle.pyoracle import Oracle, have_ref
    kind = "reference" if have_ref() else "port"
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 256))
    n = len(ts)

    def run_reference(idx, threads):
        """-> (seconds wall, busy seconds summed, outs, cigar list, md list) for tiles idx (C++ threads)."""
        lib = C.CDLL(pyoracle.REF_SO)
        m = len(idx)
        caps = (4 * ts.H[idx] + 4 * ts.W[idx] + 256).astype(np.int32)
        toff = np.concatenate([[0], np.cumsum(2 * caps.astype(np.int64))]).astype(np.uint64)
        text = np.zeros(int(toff[-1]) + 16, dtype=np.uint8)
        outs = (pyoracle.OracleOut * m)()
        busy = np.zeros(threads, dtype=np.float64)
        params = (C.c_float * 6)(*pyoracle.DEFAULT_PARAMS)
        arr = lambda a: np.ascontiguousarray(a)  # noqa: E731
        # (the reference takes the corridor ROWS: pointers into the tile set's arrays, whatever form the GPU side was handed)
        refp = arr((ts.ref.ctypes.data + ts.ref_off[:-1][idx]).astype(np.uint64))
        qryp = arr((ts.qry.ctypes.data + ts.qry_off[:-1][idx]).astype(np.uint64))
        rop = arr((ts.row_offset.ctypes.data + 4 * ts.qry_off[:-1][idx]).astype(np.uint64))
        rlp = arr((ts.row_length.ctypes.data + 4 * ts.qry_off[:-1][idx]).astype(np.uint64))
        rl, ql = arr(ts.W[idx].astype(np.int32)), arr(ts.H[idx].astype(np.int32))
        lib.oracle_align_many.restype = C.c_int
        lib.oracle_align_many.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 11
        toff_in = np.ascontiguousarray(toff[:-1])          # (kept in a variable: the call must not outlive a temporary)
        c0 = time.perf_counter()
        threw = lib.oracle_align_many(params, threads, m, refp.ctypes.data, rl.ctypes.data, qryp.ctypes.data, ql.ctypes.data,
                                      rop.ctypes.data, rlp.ctypes.data, C.addressof(outs), text.ctypes.data, toff_in.ctypes.data,
                                      caps.ctypes.data, busy.ctypes.data)
        dt = time.perf_counter() - c0
        assert threw == 0, "the reference threw on %d tiles" % threw
        raw = text.tobytes()
        cig = [raw[int(toff[k]):int(toff[k]) + outs[k].cigar_len].decode() for k in range(m)]
        md = [raw[int(toff[k]) + int(caps[k]):int(toff[k]) + int(caps[k]) + outs[k].md_len].decode() for k in range(m)]
        return dt, float(busy.sum()), outs, cig, md

    scan = None
    if kind == "reference":
        # How many threads serve the reference best on this host?  (Measured on the bench box, 2 x 64 cores with
        # SMT: linear to 32 threads -- 35 Gbp/h --, HALF of that on 64 and less on 256: the job does not get all
        # the cores it can see.)  A short scan, then the bounded sample on the best count.
        scan = {}
        for tc in [c for c in (8, 16, 32, 64, 128, 256) if c <= cores] or [cores]:
            m = min(n, 2 * tc)
            sdt = run_reference(np.arange(m), tc)[0]
            scan[tc] = float(ts.H[:m].sum()) / sdt * 3.6e-6
        threads = max(scan, key=scan.get)
        per_round = float(ts.H[:min(n, threads)].sum()) / (scan[threads] / 3.6e-6)
        n_sample = int(min(n, max(threads, threads * int(seconds_budget / max(per_round, 1e-3)))))
        idx = np.arange(n_sample)
        dt, busy_sum, outs, cigs, mds = run_reference(idx, threads)
        want = []
        for k in range(n_sample):
            o = outs[k]
            d = {f: getattr(o, f) for f, _ in pyoracle.OracleOut._fields_}
            d["score_bits"] = int(np.float32(o.score).view(np.uint32))
            d["cigar"], d["md"] = cigs[k], mds[k]
            want.append(d)
    else:
        oracles = [Oracle(kind) for _ in range(threads)]
        n_cal = min(n, threads)
        cal = [threading.Thread(target=lambda k=k: oracles[k].align(ts.tile(k), want_nm=False)) for k in range(n_cal)]
        t0 = time.perf_counter()
        for th in cal:
            th.start()
        for th in cal:
            th.join()
        per_round = max(time.perf_counter() - t0, 1e-3)
        per_thread = int(max(1, min(n // threads if n >= threads else 1, seconds_budget / per_round)))
        n_sample = min(n, per_thread * threads)
        want = [None] * n_sample
        busy = [0.0] * threads

        def work(k):
            for i in range(k, n_sample, threads):
                c0 = time.perf_counter()
                want[i] = oracles[k].align(ts.tile(i), want_nm=False)
                busy[k] += time.perf_counter() - c0
        ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        busy_sum = sum(busy)
        for o in oracles:
            o.close()
    # GPU side of the comparison: text stage of the sampled tiles (all host threads, C), outside the CPU timing
    gpu_txt = format_tileset(al.lib, ts, np.arange(n_sample), results, ops)
    n_bad, detail = 0, None
    for i in range(n_sample):
        w, got = want[i], gpu_txt[i]
        diff = None
        if not (w["ret"] < 0 and got["ret"] < 0):
            for key in PARITY_KEYS:
                if w[key] != got[key]:
                    diff = key
                    break
            if diff is None and np.float32(w["identity"]).view(np.uint32) != np.float32(got["identity"]).view(np.uint32):
                diff = "identity"
        if diff is not None:
            n_bad += 1
            detail = detail or "tile %d: %s" % (i, diff)
    eff = effective_cores()
    bases = int(ts.H[:n_sample].sum())
    cells = int(sum(int(ts.row_length[ts.qry_off[i]:ts.qry_off[i + 1]].astype(np.int64).sum()) for i in range(n_sample)))
    cpu = {
        "value": bases / dt * 3600.0 / 1e9,
        "unit": "Gbp/h",
        # `cores` = what the host lets this job use (the cgroup quota when there is one), `threads` = the workers that
        # ran: the box shows 256 hardware threads, allows 16 cores' worth of CPU time, and the reference is fastest on 32 threads
        "cores": (int(round(eff["effective"])) if eff["effective"] >= 1 else 1),
        "threads": threads,
        "Gbp_per_h_per_core": bases / dt * 3600.0 / 1e9 / max(float(eff["effective"]), 1.0),
        "kind": kind,
        "sample": "%d of the step's tiles (%.2f Mbp, %.2e cells) in %.1f s wall on %d %s threads; host: %d hardware threads visible, %s" % (
            n_sample, bases / 1e6, cells, dt, threads, "C++ (best of a thread-count scan)" if kind == "reference" else "python", cores,
            ("a cgroup quota of %.1f cores' worth of CPU time -- the effective core count" % eff["cgroup_quota_cores"]) if eff["cgroup_quota_cores"] else "no CPU quota"),
        "host_cores_effective": eff,
        "cells_per_s_per_core": cells / max(busy_sum, 1e-9),
        "thread_scan_Gbp_per_h": scan,
    }
    # the same reference as independent processes, one per visible core (and at the thread scan's best count): its best foot
    try:
        if kind == "reference":
            per_tile_s = dt * threads / max(n_sample, 1)            # ~ one core's seconds per tile
            best = None
            for pc in sorted({min(cores, 128), threads}):
                m = int(min(n, max(pc, pc * int(max(1.0, seconds_budget / 2.0 / max(per_tile_s, 1e-3))))))
                r = cpu_baseline_processes(ts, m, pc, seconds_budget)
                if r is not None:
                    cpu.setdefault("as_processes", []).append(r)
                    if best is None or r["value"] > best["value"]:
                        best = r
            if best is not None and best["value"] > cpu["value"]:
                cpu["threads_in_one_process"] = {"value": cpu["value"], "threads": cpu["threads"]}
                cpu["value"], cpu["threads"] = best["value"], best["processes"]
                cpu["Gbp_per_h_per_core"] = best["value"] / max(float(eff["effective"]), 1.0)
                cpu["sample"] = ("%d of the step's tiles on %d single-threaded processes in %.1f s wall (the best of: one process with the best thread "
                                 "count -- %.1f Gbp/h on %d threads, every tile of that run compared with the GPU --, and 1 process per core); %d hardware threads visible, effective cores %s"
                                 % (best["tiles"], best["processes"], best["seconds"], cpu["threads_in_one_process"]["value"], threads, cores,
                                    ("%.1f (cgroup quota)" % eff["cgroup_quota_cores"]) if eff["cgroup_quota_cores"] else str(eff["effective"])))
    except Exception as e:        # the extra measurement must not break the contract line
        cpu["as_processes_error"] = str(e)
    return cpu, "%d/%d" % (n_sample - n_bad, n_sample), detail


def cpu_baseline_processes(ts, n_sample, procs, seconds_budget):
    """The reference as independent single-threaded PROCESSES (how BASELINE.md section 2 measured it; the reference's
    threads inside one process stop scaling at ~32 on the bench host): `procs` workers (oracle/ref_proc_worker.py),
    started together, each with every procs-th tile of the sample.  -> dict or None.  Outputs are not collected here
    (parity is checked on the threaded run)."""
    import subprocess
    import tempfile
    from oracle import pyoracle
    if not pyoracle.have_ref():
        return None
    # bounded: at most 4096 tiles (~0.5 GB of arrays, written once and memory-mapped by every worker) and 128 processes
    procs = int(max(1, min(procs, 128)))
    n_sample = int(min(len(ts), 4096, max(procs, n_sample)))
    sub = ts.subset(np.arange(n_sample))
    d = tempfile.mkdtemp(prefix="cvx_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path, go = d, os.path.join(d, "go")
    files = []
    for k_, a_ in (("ref", sub.ref), ("qry", sub.qry), ("row_offset", sub.row_offset), ("row_length", sub.row_length),
                   ("ref_off", sub.ref_off), ("qry_off", sub.qry_off)):
        np.save(os.path.join(d, k_ + ".npy"), np.ascontiguousarray(a_))
        files.append(os.path.join(d, k_ + ".npy"))
    worker = os.path.join(ROOT, "oracle", "ref_proc_worker.py")
    ps = [subprocess.Popen([sys.executable, worker, path, str(k), str(procs), go], stdout=subprocess.PIPE, text=True) for k in range(procs)]
    try:
        for q in ps:
            if q.stdout.readline().strip() != "ready":
                raise RuntimeError("worker did not start")
        t0 = time.perf_counter()
        open(go, "w").close()
        outs = [q.stdout.readline().split() for q in ps]
        dt = time.perf_counter() - t0
        for q in ps:
            q.wait(timeout=60)
        bases = sum(int(o[1]) for o in outs)
        cells = sum(int(o[2]) for o in outs)
        busy = sum(float(o[0]) for o in outs)
        return {"value": bases / dt * 3600.0 / 1e9, "unit": "Gbp/h", "processes": procs, "tiles": n_sample, "seconds": dt,
                "cells_per_s_per_process": cells / max(busy, 1e-9), "threw": sum(int(o[3]) for o in outs)}
    finally:
        for q in ps:
            if q.poll() is None:
                q.kill()
        for f in files + [go]:
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass


def other_config(al, tiles, what, parity_n, parity_max_cells=3.0e8):
    """One of the other BASELINE configs: device rate with inputs resident in HBM (best of three runs), per fill class,
    and a bounded sample of its tiles compared field by field with the reference's own ConvexAlignFast (oracle/_ref)."""
    from ngmlr_amd.aligner import format_alignment
    from oracle.pyoracle import Oracle, have_ref, same_alignment
    bases = float(sum(t.H for t in tiles))
    batch = al.upload(tiles, closed_form=True)
    try:
        batch.run()
        best = None
        for _ in range(3):
            tm = batch.run()
            if best is None or tm.total_ms < best.total_ms:
                best = tm
        classes = [{"M": li["slots_per_lane"], "tasks_or_waves": li["waves"], "kind": ("whole", "gang", "chained", "catch-all")[li.get("kind", 0) & 3], "int16_runs": li["wrap16"], "tiles": li["n_tiles"],
                    "ms": li["ms"], "G_cells_per_s": li["cells"] / max(li["ms"], 1e-6) * 1e-6} for li in batch.launches()]
        res, ops = batch.download()
        n_valid = sum(1 for i in range(len(tiles)) if res[i].status == 0)
        kind = "reference" if have_ref() else "port"
        # bounded sample: spread over the list, tiles whose CPU cost stays small (the 8192-column 100 kb tiles take the CPU ~10 s each);
        # the reference runs on a few host threads (one checker instance each; the C call releases the interpreter lock)
        cand = [i for i in range(0, len(tiles), max(1, len(tiles) // (4 * parity_n))) if tiles[i].cells <= parity_max_cells][:parity_n]
        bad, first = 0, None
        n_thr = max(1, min(16, len(cand)))
        want = [None] * len(cand)

        def check(k):
            orc = Oracle(kind)
            for q in range(k, len(cand), n_thr):
                want[q] = orc.align(tiles[cand[q]])
            orc.close()
        ths = [threading.Thread(target=check, args=(k,)) for k in range(n_thr)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        for q, i in enumerate(cand):
            d = same_alignment(want[q], format_alignment(al.lib, res[i], ops, tiles[i]))
            if d is not None:
                bad += 1
                first = first or "tile %d (%s): %s" % (i, tiles[i].tag, d)
        chain = None
        if best.n_tiles_chained:
            chain = {"tiles": int(best.n_tiles_chained), "poll_share": best.chain_poll_ticks / max(best.chain_task_ticks, 1),
                     "what": "share of the chained waves' lifetime (s_memtime ticks summed over the row-block tasks) spent polling for a boundary "
                             "record the block above had not written yet"}
        return {"what": what, "tiles": len(tiles), "read_bases": int(bases), "Gbp_per_h": bases / best.total_ms * 3.6e-3, "chained_row_blocks": chain,
                "ms": {"plan": best.plan_ms, "fill": best.fill_ms, "backtrack": best.backtrack_ms, "total": best.total_ms},
                "G_cells_per_s": best.cells / best.total_ms * 1e-6, "fill_classes": classes, "valid_alignments": "%d/%d" % (n_valid, len(tiles)),
                "parity": "%d/%d vs oracle/_ref (%s)" % (len(cand) - bad, len(cand), kind), "parity_detail": first,
                "measured": "device rate, inputs resident in HBM (cvx_batch_run, best of 3), closed-form corridors"}
    finally:
        batch.free()


def index_stage_rates(al):
    """SURVEY 8 f4 on the device, beside the line: candidate search (cvx_search_batch_ex over the recorded test_3 k-mer table,
    resident in HBM) and reference decode (cvx_genome_decode over the recorded 4-bit genome) -- rate of the whole call, and every
    result compared with what the unmodified reference produced for the same input (tests/golden/, tools/make_golden*.sh)."""
    from ngmlr_amd.aligner import Genome, KmerIndex
    from ngmlr_amd import capi as capi_mod
    from types import SimpleNamespace
    out = {}
    golden = os.path.join(ROOT, "tests", "golden")

    def recorded_search_calls(path):
        """tests/golden/cs_test_3.npz (tools/make_golden_cs.sh): the k-mer table the unmodified reference searched, every recorded
        sub-read and the LocationScore list CS::CollectResultsStd produced for it.  (A plain loader: nothing of oracle/ is used.)"""
        z = np.load(path)
        k = int(z["k"])
        off = np.concatenate([[0], np.cumsum(z["seq_len"].astype(np.int64))])
        raw = z["seqs"].tobytes()
        seqs = [raw[int(off[i]):int(off[i + 1])] for i in range(len(z["seq_len"]))]
        so = np.concatenate([[0], np.cumsum(z["n_scores"].astype(np.int64))])
        want = [(z["loc"][int(so[i]):int(so[i + 1])], z["score"][int(so[i]):int(so[i + 1])], z["rev"][int(so[i]):int(so[i + 1])]) for i in range(len(seqs))]
        # the table as ngmlr holds it: Index[4^k + 2] of 5 packed bytes (uint m_TabIndex; char m_RevCompIndex), Location[] (src/PrefixTable.h:15-31)
        n = (1 << (2 * k)) + 2
        cnt_full = np.zeros(n, dtype=np.int64)
        cnt_full[z["prefix"]] = z["cnt"]
        idx = np.zeros(n, dtype=np.dtype([("tab", "<u4"), ("rc", "i1")]))
        idx["tab"] = (1 + np.concatenate([[0], np.cumsum(cnt_full)[:-1]])).astype(np.uint32)
        idx["rc"][z["prefix"]] = z["rc"]
        return SimpleNamespace(k=k, unit_offset=int(z["unit_offset"]), seqs=seqs, want=want, max_hit=z["max_hit"], locs=np.ascontiguousarray(z["locs"], dtype=np.uint32), index=idx)
    try:
        fx = recorded_search_calls(os.path.join(golden, "cs_test_3.npz"))
        idx, locs = fx.index, fx.locs
        ix = KmerIndex(al, fx.k, idx, locs, fx.unit_offset)
        try:
            rep_n = 16
            reads = list(fx.seqs) * rep_n
            ix.search(reads[:256])
            c0 = time.perf_counter()
            got, max_hit, misses = ix.search(reads, extras=True)
            dt = time.perf_counter() - c0
            k_ms = al.stage_kernel_ms(capi_mod.STAGE_SEARCH)
        finally:
            ix.free()
        n0 = len(fx.seqs)
        ok = sum(1 for r in range(rep_n) for i in range(n0)
                 if got[r * n0 + i] is not None and len(got[r * n0 + i]) == len(fx.want[i][0]) and np.array_equal(got[r * n0 + i]["location"], fx.want[i][0])
                 and np.array_equal(got[r * n0 + i]["score"], fx.want[i][1]) and np.array_equal(got[r * n0 + i]["reverse"], fx.want[i][2])
                 and float(max_hit[r * n0 + i]) == float(fx.max_hit[i]))
        bases = sum(len(x) for x in reads)
        out["candidate_search"] = {
            "sub_reads": len(reads), "seconds": dt, "sub_reads_per_s": len(reads) / dt, "Gbp_per_h_of_sub_read_bases": bases / dt * 3.6e-6,
            "kernel_ms": k_ms, "kernel_sub_reads_per_s": len(reads) / max(k_ms * 1e-3, 1e-9),
            "parity": "%d/%d lists equal to the recorded CS::RunRead calls of the unmodified reference (entries, order, maxHitNumber)" % (ok, len(reads)),
            "bound": "latency: a sub-read's candidate list depends on the order of its votes, so a sub-read stays on one wave, which casts 64 consecutive "
                     "votes per batch with the reference's sequential semantics (search_wave_kernel: the vote table's occupied slots in LDS; the real table in "
                     "HBM for the reads LDS cannot hold); a batch is a handful of dependent LDS / memory round trips, throughput comes from sub-reads in "
                     "flight -- no HBM-roofline figure applies",
            "what": "cvx_search_batch_ex, whole call (reads in host memory -> candidate lists, maxHitNumber, kCount back), recorded test_3 index (k = %d, %d locations), "
                    "%d recorded sub-reads x %d" % (fx.k, len(fx.locs), n0, rep_n)}
    except Exception as e:
        out["candidate_search"] = {"error": str(e)}
    # the same search over a table the size of a genome's (512 Mbp synthetic reference with repeat families: 1 GB of index records and
    # locations, far beyond L2 + MALL): kernel time from HIP events, votes per sub-read, a sample of the lists against the CPU restatement
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import search_rates
        bpv, bpv_src = None, "no profiles/r*_pmc.json of this build carries a candidate_search_big entry"
        try:
            import glob
            al.lib.cvx_build_id.restype = C_char_p
            bid = al.lib.cvx_build_id().decode()
            for pm_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
                pm = json.load(open(pm_path))
                ent = pm.get("candidate_search_big")
                same_search = pm.get("source_ids", {}).get("search") == al.lib.cvx_source_id(b"search").decode()      # cvx_search.hip unchanged since that collection
                if (pm.get("build_id") == bid or same_search) and ent:
                    bpv = ent["hbm_bytes_per_vote"]
                    bpv_src = "%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/search_rates.py --big on build %s%s)" % (
                        os.path.relpath(pm_path, ROOT), pm.get("build_id"), "" if pm.get("build_id") == bid else "; this build is %s, the search kernels' sources (cvx_source_id \"search\" %s) are the same" % (bid, pm["source_ids"]["search"]))
                    break
        except Exception as e:
            bpv_src = "unavailable: %s" % e
        out["candidate_search_big"] = search_rates.big_index_rates(al, 512, 100000, 2000, pmc_bytes_per_vote=bpv)
        out["candidate_search_big"]["bytes_per_vote_source"] = bpv_src
    except Exception as e:
        out["candidate_search_big"] = {"error": str(e)}
    try:
        z = np.load(os.path.join(golden, "decode_test_3.npz"))
        wins = [(int(z["pos"][i]), int(z["len"][i]), z["bytes"][int(z["off"][i]):int(z["off"][i + 1])].tobytes()) for i in range(int(z["n"]))]
        g = Genome(al, z["binref"], int(z["nibbles"]), z["starts"])
        try:
            rep_n = max(1, 200000 // max(len(wins), 1))
            pos = [w[0] for w in wins] * rep_n
            ln = [w[1] for w in wins] * rep_n
            g.decode(pos[:len(wins)], ln[:len(wins)])
            c0 = time.perf_counter()
            got = g.decode(pos, ln)
            dt = time.perf_counter() - c0
            k_ms = al.stage_kernel_ms(capi_mod.STAGE_DECODE)
        finally:
            g.free()
        ok = sum(1 for k, o in enumerate(got) if o == wins[k % len(wins)][2])
        chars = float(sum(ln))
        out["reference_decode"] = {
            "windows": len(pos), "characters": int(chars), "seconds": dt, "G_chars_per_s": chars / dt * 1e-9,
            # the kernel alone (HIP events on its stream): 0.5 B read + 1 B written per character against the 8 TB/s peak
            "kernel_ms": k_ms, "kernel_GB_per_s": chars * 1.5 / max(k_ms * 1e-3, 1e-9) * 1e-9, "kernel_frac_of_hbm_peak": chars * 1.5 / max(k_ms * 1e-3, 1e-9) * 1e-9 / HBM_PEAK_GBS,
            "parity": "%d/%d windows byte-identical to what the unmodified reference decoded (SequenceProvider::DecodeRefSequenceExact)" % (ok, len(pos)),
            "bound": "hbm for the kernel alone (0.5 B read + 1 B written per character); the figure here is the whole call, i.e. the D2H copy of the decoded "
                     "characters into pageable memory plus python -- in the product the windows are decoded straight into the batch's sequence arena (cvx_submit_windows) and never leave HBM",
            "what": "cvx_genome_decode, whole call, recorded test_3 genome and windows x %d" % rep_n}
    except Exception as e:
        out["reference_decode"] = {"error": str(e)}
    return out


def catch_all_kernel_rate(dev, tiles, main_results):
    """fill_generic_kernel<SSE> (cvx_generic.hip): the kernel every tile takes under scoring outside the scalar-equivalent
    regime (and irregular corridors): its rate on a sample of the ONT mix with the default scoring forced through it
    (CVX_TUNE_SSE_VARIANT=1 at handle creation), results compared with the ring kernels' for the same tiles."""
    from ngmlr_amd.aligner import ConvexAlignHip
    os.environ["CVX_TUNE_SSE_VARIANT"] = "1"
    try:
        al2 = ConvexAlignHip(device=dev)
    finally:
        os.environ.pop("CVX_TUNE_SSE_VARIANT", None)
    try:
        batch = al2.upload(tiles, closed_form=True)
        try:
            batch.run()
            tm = batch.run()
            res, _ = batch.download()
            same = sum(1 for i in range(len(tiles)) if res[i].status == main_results[i].status and (res[i].status != 0 or (
                np.float32(res[i].score).view(np.uint32) == np.float32(main_results[i].score).view(np.uint32) and res[i].n_ops == main_results[i].n_ops
                and res[i].ref_position == main_results[i].ref_position and res[i].qstart == main_results[i].qstart and res[i].qend == main_results[i].qend)))
            return {"tiles": len(tiles), "fill_ms": tm.fill_ms, "G_cells_per_s": tm.cells / max(tm.fill_ms, 1e-6) * 1e-6,
                    "read_bases": int(sum(t.H for t in tiles)), "Gbp_per_h": float(sum(t.H for t in tiles)) / max(tm.total_ms, 1e-6) * 3.6e-3,
                    "equal_to_the_ring_kernels": "%d/%d (status, score bits, path end points, op count)" % (same, len(tiles)),
                    "what": "fill_generic_kernel<SSE = true>: the reference's SSE-path semantics cell by cell, slot state in a global scratch, one workgroup per tile; "
                            "ONT-mix tiles, default scoring forced through it"}
        finally:
            batch.free()
    finally:
        al2.close()


def subread_scoring_rates(lib, dev, n=32768):
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import StrippedSWHip
    from oracle.pyoracle import ScoreOracle, have_score_ref
    rng = np.random.default_rng(1)
    refs, qrys = [], []
    for _ in range(n):
        w = synth.random_ref(rng, 308)
        a = int(rng.integers(0, 50))
        qrys.append(synth.mutate(rng, w[a:a + 256], 0.15)[:256].tobytes())
        refs.append(w.tobytes())
    cells = sum((len(r) + 1) * (len(q) + 1) for r, q in zip(refs, qrys))
    sw = StrippedSWHip(device=dev)
    sw.batch_score(refs[:1024], qrys[:1024])
    import ctypes as C

    def kernel_ms():
        ms = C.c_float()
        sw.lib.cvx_score_kernel_ms(sw._al.h, C.byref(ms))
        return float(ms.value)
    c0 = time.perf_counter()
    got = sw.batch_score(refs, qrys)
    dt_gpu = time.perf_counter() - c0
    k_all = kernel_ms() * 1e-3                     # the kernel alone (HIP events on its stream), inputs resident in HBM
    c0 = time.perf_counter()
    k_1k = 0.0
    for lo in range(0, n, 1024):
        sw.batch_score(refs[lo:lo + 1024], qrys[lo:lo + 1024])
        k_1k += kernel_ms() * 1e-3
    dt_1k = time.perf_counter() - c0
    sw.close()
    kind = "reference" if have_score_ref() else "port"
    threads = os.cpu_count() or 1
    want = np.zeros(n, dtype=np.float32)
    step = (n + threads - 1) // threads

    def work(k):
        lo, hi = k * step, min(n, (k + 1) * step)
        if lo < hi:
            want[lo:hi] = ScoreOracle(kind).scores(refs[lo:hi], qrys[lo:hi])
    ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    c0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt_cpu = time.perf_counter() - c0
    return {"pairs": n, "gpu_pairs_per_s": n / dt_gpu, "gpu_cell_updates_per_s": cells / dt_gpu,
            "device_resident": {"pairs_per_s": n / max(k_all, 1e-9), "cell_updates_per_s": cells / max(k_all, 1e-9), "kernel_ms": k_all * 1e3,
                                "in_1024_pair_launches_pairs_per_s": n / max(k_1k, 1e-9),
                                "bound": "integer VALU issue: ~10 instructions per cell, one lane per diagonal, sequences as codes in LDS (score_diag_kernel); no HBM traffic to speak of",
                                "what": "the scoring kernel alone, HIP events on its stream (cvx_score_kernel_ms)"},
            "gpu_pairs_per_s_in_1024_pair_calls": n / dt_1k, "cpu_pairs_per_s": n / dt_cpu, "cpu_kind": kind, "cpu_threads": threads,
            "parity": "%d/%d" % (int((got == want).sum()), n),
            "what": "cvx_score_batch (host strings in -> scores out, incl. the python marshalling of this bench) vs StrippedSW + ssw.c"}


class Worker:
    """One device: a handle, its own tiles, a pipelined stream of steps."""

    def __init__(self, dev, ts, depth):
        from ngmlr_amd.aligner import ConvexAlignHip
        self.dev, self.ts, self.depth = dev, ts, depth
        self.al = ConvexAlignHip(device=dev)          # raises if libcvxalign.so or the GPU is missing
        self.launch_ms = {}
        self.launch_meta = {}
        self.stage = np.zeros(4)
        self.last = None                              # the last step's job (results kept for the checks)
        self.valid = 0
        self.host_s = np.zeros(2)                     # host wall time inside cvx_submit / cvx_wait (timed steps)
        self.redone = 0                               # tiles the exact-tracking pass had to redo (timed steps)
        self.wall_s = 0.0                             # this device's wall time of the last run_on_all call

    def steps(self, k, keep_last=False, record=True):
        jobs = []

        def retire(j, keep):
            c0 = time.perf_counter()
            res, _ = j.wait()
            if record:
                self.host_s[1] += time.perf_counter() - c0
                tm = j.timing()
                self.redone += int(tm.n_tiles_redone)
                self.stage += (tm.plan_ms, tm.fill_ms, tm.backtrack_ms, tm.total_ms)
                for li in j.launches():
                    key = (li["slots_per_lane"], li["waves"], li["wrap16"])
                    self.launch_ms.setdefault(key, []).append(li["ms"])
                    self.launch_meta[key] = li
            if keep:
                self.last = j
                self.valid = int((res["status"] == 0).sum())
            else:
                j.release()

        for s in range(k):
            c0 = time.perf_counter()
            jobs.append(self.al.submit(self.ts))
            if record:
                self.host_s[0] += time.perf_counter() - c0
            if len(jobs) >= self.depth:
                retire(jobs.pop(0), False)
        while jobs:
            j = jobs.pop(0)
            retire(j, keep_last and not jobs)


class BindingView:
    """The step's tiles in the form ngmlr's binding holds them: corridors as the row arrays its builders
    wrote (reference src/AlignmentBuffer.cpp:68-197), sequences in ordinary pageable memory.  fit=True: what
    Convex::ConvexAlignHip::Prepare does per SingleAlign since round 6 -- cvx_corridor_fit recovers the builder's closed form from
    the rows, every row verified, here for the whole table on the host's threads (cvx_corridor_fit_batch) EVERY step, inside the
    timed region; fit=False: round 5's binding, the rows travel (packed by the pack threads, expanded on the device)."""

    def __init__(self, ts, lib, fit, threads):
        self.ts, self.lib, self.fit, self.threads = ts, lib, fit, threads
        self.fitted = 0
        self.fit_s = 0.0
        self.read_bases = ts.read_bases

    def __len__(self):
        return len(self.ts)

    def table(self):
        tab = self.ts.table()
        if self.fit:
            import ctypes as C
            tab = tab.copy()
            n = C.c_int32(0)
            c0 = time.perf_counter()
            rc = self.lib.cvx_corridor_fit_batch(len(tab), tab.ctypes.data, self.threads, C.byref(n))
            self.fit_s += time.perf_counter() - c0
            if rc != 0:
                raise RuntimeError("cvx_corridor_fit_batch: %d" % rc)
            self.fitted = int(n.value)
        return tab


def binding_form_rates(lib, ts, dev, steps=5, warm=2, depth=2):
    """`binding_input_form` of the line: the same tiles, the same whole-path steps (host buffers in -> results out), in the two
    input forms a binding can hand over -- neither is `value`, which is measured on closed forms in a page-locked arena."""
    out = {}
    cores = min(os.cpu_count() or 1, 16)
    ts.use_closed_form(False)                       # the table now points at the row arrays; the sequences are pageable (unpinned)
    for key, fit in (("rows_fitted_to_closed_forms", True), ("row_arrays", False)):
        view = BindingView(ts, lib, fit, cores)
        w = Worker(dev, view, depth)
        try:
            w.steps(depth + 1, record=False)        # arenas of every slot
            w.steps(warm, record=False)
            lib.cvx_device_synchronize(dev)
            view.fit_s = 0.0
            c0 = time.perf_counter()
            w.steps(steps, keep_last=True)
            lib.cvx_device_synchronize(dev)
            dt = time.perf_counter() - c0
            valid = w.valid
            w.last.release()
            out[key] = {"Gbp_per_h": float(ts.read_bases) * steps / dt * 3600.0 / 1e9, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warm,
                        "batches_in_flight": depth, "valid_alignments": "%d/%d" % (valid, len(ts)),
                        "host_ms_per_step": {"cvx_submit": float(w.host_s[0]) / steps * 1e3, "cvx_wait": float(w.host_s[1]) / steps * 1e3}}
            if fit:
                out[key]["corridors_recognised"] = "%d/%d" % (view.fitted, len(ts))
                out[key]["fit_ms_per_step"] = view.fit_s / steps * 1e3
                out[key]["fit_threads"] = cores
        except Exception as e:
            out[key] = {"error": str(e)}
        finally:
            w.al.close()
    out["what"] = ("sequences in pageable memory, corridors as row arrays (int32 offset / length per read row, 8 B per row): "
                   "`rows_fitted_to_closed_forms` = cvx_corridor_fit_batch on %d host threads inside every step, then cvx_submit (what "
                   "ConvexAlignHip::Prepare + Submit do since round 6); `row_arrays` = the rows travel (round 5's binding).  Not `value`." % cores)
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=49152,
                    help="tiles per GPU per step: eight rounds of the 6144 resident fill waves, ~45 GB of direction words per batch, "
                         "~195 GB of the 288 GB of HBM with three batches in flight (batches are sized for the memory: the fixed ~5 ms "
                         "of ramp and ragged end per fill launch is 4 %% of it instead of 8 %% at 24576)")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight per device")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--resident-steps", type=int, default=3, help="extra untimed-for-`value` steps with inputs resident in HBM")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (SURVEY 8e, config C4): ONE list of --tiles tiles, partitioned over the devices by the sum of "
                         "their cells (LPT, ngmlr_amd.shard), results restored to list order; default: weak, --tiles per device")
    ap.add_argument("--row-arrays", action="store_true", help="hand over corridor row arrays instead of the builders' closed forms")
    ap.add_argument("--no-pin", action="store_true", help="keep the sequences in ordinary (pageable) memory: cvx_submit packs them into its own staging")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs (ONT mix, ultra-long + SV, short reads) reported beside the line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the e2e_pipeline extra (the reference's ngmlr with every drop-in bound against the unmodified build)")
    ap.add_argument("--e2e-reads", type=int, default=20000, help="synthetic 10 kb reads of the e2e_pipeline extra")
    ap.add_argument("--alias-device", type=int, default=-1, metavar="D",
                    help="run the --gpus N code path (N handles, N host threads, one shared pack pool; --strong too) with every handle on physical "
                         "device D: exercises the N-device path on a one-GPU box -- NOT a scaling measurement (N batches' arenas share one HBM: "
                         "use --tiles <= 16384 for N = 2)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    under_launcher = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or bool(os.environ.get("CVX_BENCH_FORCE_DIST"))
    if under_launcher and world != args.gpus:
        print("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        return 2
    n_local = 1 if under_launcher else args.gpus      # devices driven by this process

    # fail loudly, and before any work, when the box has fewer devices than asked for (the count is
    # taken in a child process: the tile generators below must be forked before HIP is initialised)
    pid = os.fork()
    if pid == 0:
        try:
            from ngmlr_amd import capi as capi_child
            os._exit(min(capi_child.load().cvx_device_count(), 200))
        except BaseException:
            os._exit(255)
    n_dev = os.WEXITSTATUS(os.waitpid(pid, 0)[1])
    need = (local_rank + 1) if under_launcher else args.gpus
    if args.alias_device >= 0:
        if under_launcher:
            print("bench.py: --alias-device is for the single-process form", file=sys.stderr)
            return 2
        need = args.alias_device + 1
    if n_dev == 255:
        print("bench.py: libcvxalign.so could not be loaded (run __graft_entry__.build())", file=sys.stderr)
        return 2
    if n_dev < need:
        print("bench.py: --gpus %d needs %d visible MI355X device(s), this box has %d" % (args.gpus, need, n_dev), file=sys.stderr)
        return 2

    # synthetic tiles first (worker processes must be forked before HIP is initialised): every
    # device owns its own reads (weak scaling, reads shard naturally)
    from ngmlr_amd import synth
    t_gen = time.perf_counter()
    procs = max(1, min((os.cpu_count() or 1) // (world if under_launcher else 1), 64))
    strong_parts = None
    with ProcessPoolExecutor(procs, mp_context=mp.get_context("fork")) as pool:
        if args.strong:
            # one list for the whole job (every rank generates the same one from the same seed), partitioned by cells
            from ngmlr_amd.shard import shard_tiles
            whole = synth.pacbio_tileset(args.tiles, seed=args.seed, read_len=args.read_len, pool=pool)
            cells = np.add.reduceat(whole.row_length.astype(np.int64), whole.qry_off[:-1]) if len(whole) else np.zeros(0, np.int64)
            strong_parts = shard_tiles(cells.tolist(), args.gpus)
            mine = [rank] if under_launcher else list(range(n_local))
            tilesets = [whole.subset(strong_parts[d]) for d in mine]
        else:
            tilesets = [synth.pacbio_tileset(args.tiles, seed=args.seed + 1000 * (rank + d), read_len=args.read_len, pool=pool)
                        for d in range(n_local)]
        extra_tiles = {}
        if rank == 0 and args.gpus == 1 and not args.no_extras and not args.no_cpu_baseline:
            # the other BASELINE configs (parity cases with a rate, outside the timed region): generated here, before HIP exists
            extra_tiles["ont"] = synth.parallel_workload("ont", 60000, 11, pool)
            extra_tiles["ultralong_sv"] = synth.parallel_workload("ultralong_mix", 4096, 19, pool, chunk=16)
            extra_tiles["short"] = synth.parallel_workload("short", 60000, 17, pool, chunk=2048)
    t_gen = time.perf_counter() - t_gen

    from ngmlr_amd import capi
    dist = torch = None
    if under_launcher:
        import torch as _torch_first   # noqa: F401  (under the launcher torch and its RCCL come first, with the runtime they were built for)
    lib = capi.load()             # launched directly: the HIP runtime this library links (/opt/rocm) is the process's runtime
    if under_launcher:
        # one rank per device: torch.distributed (RCCL) carries the barrier and the max-over-ranks
        # reduction of the timing -- plumbing only, there is no collective on the data path
        import torch as torch_mod
        import torch.distributed as dist_mod
        torch, dist = torch_mod, dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    devs = [local_rank] if under_launcher else list(range(n_local))
    if args.alias_device >= 0:
        devs = [args.alias_device] * n_local       # N handles + N host threads on ONE physical device
    workers = [Worker(d, ts, args.depth) for d, ts in zip(devs, tilesets)]
    # corridors travel as the closed forms of the reference's builders (the device generates the rows), sequences sit
    # in page-locked arenas the device pulls from directly: cvx_submit touches no base and no row
    pinned = []
    for ts in tilesets:
        if not args.row_arrays:
            ts.use_closed_form()
        pinned.append((not args.no_pin) and ts.pin(lib))

    def sync_all():
        # every device this process drives is idle: hipDeviceSynchronize behind the C ABI (what
        # torch.cuda.synchronize() is; launched directly the bench does not load torch at all, whose
        # wheel carries an older HIP runtime of its own that would replace the one the library links)
        for d in devs:
            capi.check(lib.cvx_device_synchronize(d))
        if dist is not None:
            torch.cuda.synchronize(local_rank)
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(local_rank)
            capi.check(lib.cvx_device_synchronize(local_rank))

    def run_on_all(fn):
        errs = []

        def wrap(w):
            try:
                c0 = time.perf_counter()
                fn(w)
                w.wall_s = time.perf_counter() - c0      # this device's own wall time for the call (device_skew_ms)
            except BaseException as e:  # surface worker failures in the main thread
                errs.append(e)

        ths = [threading.Thread(target=wrap, args=(w,)) for w in workers]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if errs:
            raise errs[0]

    # one-time setup, untimed: every batch slot of the pipeline allocates its device arenas (~25 GB) and
    # pinned staging (~3 GB) the first time it is used -- hundreds of ms per slot -- so cycle through
    # all of them once before the W warm-up steps
    run_on_all(lambda w: w.steps(args.depth + 2, record=False))
    run_on_all(lambda w: w.steps(args.warmup, record=False))
    sync_all()
    t0 = time.perf_counter()
    run_on_all(lambda w: w.steps(args.steps, keep_last=True))
    sync_all()
    dt = time.perf_counter() - t0
    bases = float(sum(ts.read_bases for ts in tilesets))

    def device_line(w):
        """the dominant fill launch of one device (HIP events on its stream) and that device's own timed wall"""
        if not w.launch_ms:
            return [float(w.dev), 0.0, 0.0, 0.0, float(w.wall_s)]
        dom_ = max(w.launch_ms, key=lambda k_: w.launch_meta[k_]["alg_bytes"])
        return [float(w.dev), float(np.mean(w.launch_ms[dom_])), float(w.launch_meta[dom_]["alg_bytes"]), float(w.launch_meta[dom_]["n_tiles"]), float(w.wall_s)]
    per_dev = [device_line(w) for w in workers]
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tb = torch.tensor([bases], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        bases = float(tb.item())
        mine_ = torch.tensor(per_dev[0], dtype=torch.float64, device="cuda:%d" % local_rank)
        all_ = [torch.zeros_like(mine_) for _ in range(world)]
        dist.all_gather(all_, mine_)
        per_dev = [[float(x) for x in t_.tolist()] for t_ in all_]
        for r_, row_ in enumerate(per_dev):
            row_[0] = float(r_)                                  # (rank = device of the node)

    out = None
    if rank == 0:
        w0 = workers[0]
        ts = w0.ts
        res, ops = w0.last.results, w0.last.ops
        n_ops_total = int(np.asarray(res["n_ops"], dtype=np.int64).sum())      # (res / ops are views of the job's buffers: gone after release)
        # outside the timed region: CPU baseline on a bounded sample + parity on every sampled tile
        cpu = parity = parity_detail = None
        if not args.no_cpu_baseline and args.gpus == 1:
            cpu, parity, parity_detail = cpu_baseline_and_parity(w0.al, ts, res, ops, args.cpu_seconds)
        # host text stage (CIGAR/MD/NM) of a sample on all host threads: reported beside `value`
        text_stage = None
        host_txt = []
        try:
            from ngmlr_amd.aligner import format_tileset
            k = min(len(ts), 2048)
            c0 = time.perf_counter()
            host_txt = format_tileset(w0.al.lib, ts, np.arange(k), res, ops)
            dtxt = time.perf_counter() - c0
            text_stage = {"Gbp_per_h": float(ts.H[:k].sum()) / dtxt * 3600.0 / 1e9, "seconds": dtxt, "tiles": k,
                          "threads": os.cpu_count(), "what": "cvx_format_batch (CIGAR + MD + NM) incl. the python marshalling of this bench"}
        except Exception as e:  # never let the extra measurement break the contract line
            text_stage = {"error": str(e)}
        # the same stage on the device (cvx_job_text): CIGAR + MD + fields of ALL tiles of the step, strings back on the host
        text_dev = None
        path_columns = None
        try:
            w0.last.text_raw()                                # first call allocates the job's text buffers
            c0 = time.perf_counter()
            trec, toff, tbuf = w0.last.text_raw()
            dtd = time.perf_counter() - c0
            path_columns = int(np.frombuffer(trec, dtype=np.int32).reshape(-1, 17)[:len(ts), 7].sum())      # alignment_length
            nbad = 0
            if text_stage and "error" not in text_stage:
                raw = tbuf.raw
                for i_, h_ in enumerate(host_txt):            # cross-check against the host form on its sample
                    t_ = trec[i_]
                    o_ = int(toff[i_])
                    if h_["cigar"] is None or raw[o_:o_ + t_.cigar_len].decode() != h_["cigar"] or raw[o_ + t_.cigar_len + 1:o_ + t_.cigar_len + 1 + t_.md_len].decode() != h_["md"]:
                        nbad += 1
            text_dev = {"Gbp_per_h": float(ts.read_bases) / dtd * 3600.0 / 1e9, "seconds": dtd, "tiles": len(ts), "text_bytes": len(tbuf),
                        "equal_to_host_form": "%d/%d" % (len(host_txt) - nbad, len(host_txt)),
                        "what": "cvx_job_text: text_kernel (lengths + fields), offsets scan, text_kernel (strings), D2H of the dense text"}
            # SAM record assembly (cvx_sam_batch, host pack threads) fed straight from the device text: one record per tile
            try:
                import ctypes as C
                kk = min(len(ts), 8192)
                tab_ = ts.table()
                base_ = C.addressof(tbuf)
                recs_ = (capi.CvxSamRecord * kk)()
                for i_ in range(kk):
                    t_, r_ = trec[i_], recs_[i_]
                    r_.read_name = b"synthetic_read"; r_.seq = C.cast(int(tab_[i_]["qry"]), C.c_char_p); r_.qual = None
                    r_.read_length = int(tab_[i_]["qry_len"]); r_.flags = 0; r_.primary = 1; r_.reverse = 0
                    r_.ref_name = b"chrS"; r_.ref_name_len = 4; r_.location = 1000 + t_.position_offset; r_.mq = 60
                    r_.cigar = C.cast(base_ + int(toff[i_]), C.c_char_p); r_.md = C.cast(base_ + int(toff[i_]) + t_.cigar_len + 1, C.c_char_p)
                    r_.cigar_op_count = t_.cigar_op_count; r_.mate_ref_name = b"*"; r_.mate_location = -1; r_.template_length = 0
                    r_.score = t_.score; r_.nm = t_.nm; r_.identity = t_.identity; r_.qstart = t_.qstart; r_.qend = t_.qend; r_.sv_type = t_.sv_type
                    r_.n_others = 0; r_.others = None; r_.rg_id = None; r_.hard_clip = 0; r_.bam_cigar_fix = 0; r_.skip = 0
                offs_ = np.zeros(kk + 1, dtype=np.uint64)
                w0.al.lib.cvx_sam_batch(kk, recs_, None, 0, offs_.ctypes.data)              # sizes
                sam_ = C.create_string_buffer(int(offs_[kk]))
                c0 = time.perf_counter()
                rc_ = w0.al.lib.cvx_sam_batch(kk, recs_, sam_, int(offs_[kk]), offs_.ctypes.data)
                dsam = time.perf_counter() - c0
                valid_ = sum(1 for i_ in range(kk) if trec[i_].ret >= 0)
                text_dev["sam_records"] = {"records": kk, "valid_alignments": valid_, "bytes": int(offs_[kk]), "seconds": dsam, "rc": rc_,
                                           "records_per_s": kk / dsam, "GB_per_s": int(offs_[kk]) / dsam * 1e-9,
                                           "Gbp_per_h": float(ts.H[:kk].sum()) / dsam * 3600.0 / 1e9,
                                           "what": "cvx_sam_batch: SAMWriter::DoWriteReadGeneric's record (mandatory fields + AS NM XI XS XE XR MD SV QS QE CV) on the pack threads, CIGAR / MD taken from cvx_job_text's buffer"}
                del sam_, recs_
            except Exception as e:
                text_dev["sam_records"] = {"error": str(e)}
            # nmPerPosition on the device (cvx_job_nm_profile): kernel alone for a range that fits comfortably (12 B per column),
            # and the same range brought to the host
            try:
                kk = min(len(ts), 4096)
                w0.last.nm_profile(0, kk, to_host=False)
                off_, _, kms = w0.last.nm_profile(0, kk, to_host=False)
                c0 = time.perf_counter()
                off_, tri_, _ = w0.last.nm_profile(0, kk, to_host=True)
                dnm = time.perf_counter() - c0
                ent = int(off_[kk])
                text_dev["nm_profile"] = {"tiles": kk, "entries": ent, "kernel_ms": kms, "kernel_GB_per_s": ent * 12e-6 / max(kms, 1e-9),
                                          "kernel_Gbp_per_h": float(ts.H[:kk].sum()) / max(kms, 1e-9) * 3.6e-3,
                                          "to_host_seconds": dnm, "to_host_Gbp_per_h": float(ts.H[:kk].sum()) / dnm * 3600.0 / 1e9,
                                          "what": "nm_profile_kernel (12 B per EQ/X/D column written to HBM); to_host = the same + D2H into pageable memory incl. this bench's python"}
                del tri_
            except Exception as e:
                text_dev["nm_profile"] = {"error": str(e)}
        except Exception as e:
            text_dev = {"error": str(e)}
        valid = w0.valid
        w0.last.release()
        # the same step with inputs already resident in HBM (plan -> fill -> backtrack -> compaction)
        resident = None
        try:
            if args.resident_steps > 0:
                tab = ts.table()
                import ctypes as C
                from ngmlr_amd.aligner import DeviceBatch
                b = C.c_void_p()
                capi.check(w0.al.lib.cvx_batch_upload(w0.al.h, len(tab), tab.ctypes.data_as(C.POINTER(capi.CvxTile)), C.byref(b)))
                batch = DeviceBatch(w0.al, b, ts)
                batch.run()
                capi.check(lib.cvx_device_synchronize(devs[0]))
                c0 = time.perf_counter()
                tms = [batch.run() for _ in range(args.resident_steps)]
                capi.check(lib.cvx_device_synchronize(devs[0]))
                dres = (time.perf_counter() - c0) / args.resident_steps
                resident = {"Gbp_per_h": ts.read_bases / dres * 3600.0 / 1e9, "ms_per_step": dres * 1e3,
                            "fill_ms": float(np.mean([t.fill_ms for t in tms])), "backtrack_ms": float(np.mean([t.backtrack_ms for t in tms])),
                            "plan_ms": float(np.mean([t.plan_ms for t in tms])), "steps": args.resident_steps,
                            "what": "cvx_batch_run on a batch already in HBM, one device"}
                batch.free()
        except Exception as e:
            resident = {"error": str(e)}

        # sub-read scoring, the reference's batch shape (1024 pairs of a 256-base sub-read against a
        # ~300-base window, src/ScoreBuffer.cpp:87-168): cvx_score_batch against the reference's own StrippedSW
        # on all host threads, same pairs, every score compared
        subread = None
        if not args.no_cpu_baseline and args.gpus == 1:
            try:
                subread = subread_scoring_rates(w0.al.lib, devs[0])
            except Exception as e:
                subread = {"error": str(e)}

        # the other BASELINE configs (C3 ONT mix, C5 ultra-long + SV, short reads): rate + parity, outside the timed region
        others = None
        if extra_tiles:
            others = {}
            for name_, what_, pn_ in (("ont", "configs[2]: ONT-like reads, 25 % error 4:4:2, tile mix median 1.3 kb up to 20 kb, widths 309-463, 10 % retries at 2x", 1024),
                                      ("ultralong_sv", "configs[4]: 100 kb reads, 95 % first-attempt anchors corridors (309+), 5 % widened to 2048 / 8192 columns or full-matrix inversion tiles", 256),
                                      ("short", "short reads (<= 256 bp) on the linear corridor (src/AlignmentBuffer.cpp:2576-2594)", 1024)):
                try:
                    others[name_] = other_config(w0.al, extra_tiles[name_], what_, pn_)
                except Exception as e:
                    others[name_] = {"error": str(e)}

        # SURVEY 8 f4 on the device (candidate search, reference decode) and the catch-all fill kernel: rate + parity, beside the line
        index_stage = generic_rate = None
        if not args.no_cpu_baseline and args.gpus == 1 and not args.no_extras:
            index_stage = index_stage_rates(w0.al)
            try:
                if extra_tiles.get("ont"):
                    sample = extra_tiles["ont"][:4096]
                    bm = w0.al.upload(sample, closed_form=True)
                    try:
                        bm.run()
                        main_res, _ = bm.download()          # (a ctypes array owned by python: outlives the batch)
                    finally:
                        bm.free()
                    generic_rate = catch_all_kernel_rate(devs[0], sample, main_res)
            except Exception as e:
                generic_rate = {"error": str(e)}

        value = bases * args.steps / dt * 3600.0 / 1e9
        launch_ms, launch_meta = w0.launch_ms, w0.launch_meta
        # dominant kernel = the fill launch that carries most of the work (classes run concurrently)
        dom = max(launch_ms, key=lambda k_: launch_meta[k_]["alg_bytes"])
        dms = float(np.mean(launch_ms[dom]))
        meta = launch_meta[dom]
        achieved = meta["alg_bytes"] / (dms * 1e-3) / 1e9
        wd = ts.widths()
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the
        # committed rocprofv3 passes of this round are scaled to this launch by algorithmic bytes
        traffic, traffic_src = None, None
        try:
            import glob
            lib.cvx_build_id.restype = C_char_p
            build_id = lib.cvx_build_id().decode()
            for pm_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
                pm = json.load(open(pm_path))
                lib.cvx_source_id.restype = C_char_p
                fill_id = lib.cvx_source_id(b"fill").decode()
                if pm.get("build_id") != build_id and pm.get("source_ids", {}).get("fill") != fill_id:
                    continue                                   # counters of another build of the fill kernels: not this launch's traffic
                ent = pm.get("fill_ring_kernel<M=%d,NW=%d,wrap16=%d>" % dom)
                if ent:
                    traffic = ent["hbm_bytes"] * (meta["alg_bytes"] / ent["alg_bytes"])
                    traffic_src = ("not measured in this run: %s (separate rocprofv3 --pmc FETCH_SIZE x2 / --pmc WRITE_SIZE passes of the same bench "
                                   "command on %s, %d tiles in the launch) x algorithmic-byte ratio %.3f" % (
                                       os.path.relpath(pm_path, ROOT), ("the same build %s" % build_id) if pm.get("build_id") == build_id else
                                       "build %s -- this build is %s, the fill kernels' sources (cvx_source_id \"fill\" %s) are the same" % (pm.get("build_id"), build_id, fill_id),
                                       ent.get("tiles", 0), meta["alg_bytes"] / ent["alg_bytes"]))
                    break
            if traffic is None:
                traffic_src = "no profiles/r*_pmc.json was collected on this build of the kernels (%s): null rather than a stale number" % build_id
        except Exception as e:
            traffic_src = "unavailable: %s" % e
        # the second kernel of the step: SURVEY 8(d)'s bytes for it are P (one direction per path step) + 4 n_ops
        backtrack_fig = None
        try:
            if resident and "backtrack_ms" in resident:
                path = int(path_columns) if path_columns else None
                bms = resident["backtrack_ms"]
                backtrack_fig = {"kernel": "backtrack_kernel (8 lanes per tile) + finalize + ops compaction", "ms": bms, "path_columns": path, "n_ops": n_ops_total,
                                 "alg_bytes": (path + 4 * n_ops_total) if path else None,
                                 "achieved_GB_per_s": ((path + 4 * n_ops_total) / (bms * 1e-3) / 1e9) if path else None,
                                 "path_columns_per_s": (path / (bms * 1e-3)) if path else None,
                                 "bound": "latency, not bandwidth: per tile a chain of ~P/32 dependent loads of direction words (2 bits per cell, 32 steps per word); "
                                          "the stage is 9 % of the step and its HBM-roofline fraction (bytes above / 8 TB/s) is of the order 1e-3 by construction"}
        except Exception as e:
            backtrack_fig = {"error": str(e)}
        out = {
            "metric": "aligned Gbp/hour (PacBio 10kb synthetic, convex-gap SW hot path, host buffers in -> results out, CIGAR bit-exact)",
            "value": value,
            "unit": "Gbp/h",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: synthetic PacBio-like %d bp reads (15%% err, ins:del:sub 6:3:1) vs seeded uniform ACGT reference, -x pacbio scoring, anchors corridor" % args.read_len,
                "tiles_per_gpu_per_step": args.tiles,
                "read_bases_per_gpu_per_step": ts.read_bases,
                "corridor_width_median": int(np.median(wd)),
                "corridor_width_max": int(wd.max()),
                "cells_per_gpu_per_step": ts.cells,
                "batches_in_flight_per_gpu": args.depth,
                "h2d_bytes_per_gpu_per_step": int(ts.ref.nbytes + ts.qry.nbytes + (ts.row_offset.size if args.row_arrays else 32 * len(ts))),   # sequences + one step byte per corridor row, or a 32-byte closed form per tile
                "corridors": "row arrays (one step byte per row uploaded)" if args.row_arrays else "closed forms of the reference's builders (cvx_tile.corridor_kind), rows generated on the device",
                "sequences": "page-locked arena (cvx_host_alloc), pulled by the device without host packing" if pinned and pinned[0] else "pageable memory, packed into the job's pinned staging by cvx_submit",
                "strong_scaling_partition": ({"tiles_total": args.tiles, "tiles_per_device": [len(p_) for p_ in strong_parts], "by": "sum of corridor cells (LPT)",
                                              "order_restored": sorted(i_ for p_ in strong_parts for i_ in p_) == list(range(args.tiles))} if strong_parts is not None else None),
                "launch": "torch.distributed.run, one rank per device" if under_launcher else "one process, one host thread + handle per device",
                "alias_device": (None if args.alias_device < 0 else "all %d handles on physical device %d: the N-device code path on one GPU, not a scaling measurement" % (args.gpus, args.alias_device)),
                "sharding": "reads sharded across devices, no collective on the data path",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "fill_ring_kernel<M=%d,NW=%d,wrap16=%d> (two-phase instantiation)" % dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launch_ms": dms,
                "launches_timed": len(launch_ms[dom]),
                "tiles_redone_by_the_exact_pass": "%d of %d (two-phase best-cell tracking: a tile whose best cell is not in the exactly tracked tail is filled again)" % (w0.redone, args.steps * len(ts)),
                "launch_tiles": meta["n_tiles"],
                "alg_bytes_per_launch": meta["alg_bytes"],
                "gcups": meta["cells"] / (dms * 1e-3) / 1e9,
                "all_fill_launches": {"M%d_NW%d_wrap%d" % k_: {"ms": float(np.mean(v)), "tiles": launch_meta[k_]["n_tiles"],
                                                             "kind": ("whole tiles", "gang of waves per tile", "chained row blocks", "catch-all kernel")[launch_meta[k_].get("kind", 0) & 3]}
                                      for k_, v in launch_ms.items()},
            },
            # every device of the run: its own dominant fill launch against the roofline and its own timed wall, so that a scaling run
            # explains itself (one straggling device shows as skew, a slower launch on every device as a lower `frac`)
            "per_device": [{"device": int(r_[0]), "roofline": {"launch_ms": r_[1], "launch_tiles": int(r_[3]), "achieved": (r_[2] / (r_[1] * 1e-3) / 1e9) if r_[1] > 0 else None,
                                                               "frac": (r_[2] / (r_[1] * 1e-3) / 1e9 / HBM_PEAK_GBS) if r_[1] > 0 else None, "unit": "GB/s"},
                            "timed_wall_s": r_[4]} for r_ in per_dev],
            "device_skew_ms": (max(r_[4] for r_ in per_dev) - min(r_[4] for r_ in per_dev)) * 1e3,
            "stage_ms_per_step": ({"plan": resident["plan_ms"], "fill": resident["fill_ms"], "backtrack": resident["backtrack_ms"],
                                   "what": "HIP-event stage times of the device-resident steps (in the pipelined steps the stages of neighbouring batches overlap)"}
                                  if resident and "fill_ms" in resident else None),
            "device_resident": resident,
            "backtrack": backtrack_fig,
            "host_ms_per_step": {"cvx_submit": float(w0.host_s[0]) / args.steps * 1e3, "cvx_wait": float(w0.host_s[1]) / args.steps * 1e3,
                                 "what": "wall time the device's host thread spends inside the two calls (packing + queueing / blocked on results)",
                                 "pack_threads_shared_by_the_process": int(os.environ.get("CVX_PACK_THREADS", "0")) or min(os.cpu_count() or 1, 16)},
            "valid_alignments": "%d/%d" % (valid, len(ts)),
            "parity": parity,
            "parity_detail": parity_detail,
            "cpu_baseline": cpu,
            "text_stage_host": text_stage,
            "text_stage_device": text_dev,
            "subread_scoring": subread,
            "other_configs": others,
            "index_stage_device": index_stage,
            "catch_all_fill_kernel": generic_rate,
            "tile_generation_s": t_gen,
        }
    for w in workers:
        w.al.close()
    for ts in tilesets:
        ts.unpin()
    if out is not None and args.gpus == 1 and not (args.no_extras or args.no_cpu_baseline):
        try:
            out["binding_input_form"] = binding_form_rates(lib, tilesets[0], devs[0])
        except Exception as e:        # never let the extra measurement break the contract line
            out["binding_input_form"] = {"error": str(e)}
    if out is not None and args.gpus == 1 and not (args.no_e2e or args.no_extras or args.no_cpu_baseline):
        # ngmlr's own pipeline (the reference's only published metric is end to end, README.md:25), outside the timed region and
        # after this process has given its device memory back: the reference's binary built with every drop-in (oracle/_ref/ngmlr_hip_all)
        # against the unmodified build (ngmlr_ref) on the same synthetic reads -- wall, mapping time, SAM identical, tiles per
        # launch, launch-in-flight share, CPU seconds by thread class, peak RSS of both
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import e2e_rates
            out["e2e_pipeline"] = e2e_rates.pipeline_summary(args.e2e_reads)
        except Exception as e:        # never let the extra measurement break the contract line
            out["e2e_pipeline"] = {"error": str(e)}
    if out is not None:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
