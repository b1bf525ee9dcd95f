"""Dev tool (GPU box): the reference's ngmlr end to end, unmodified against the drop-ins.

    e2e_rates.py [threads...]              test/test_3.sh's reads (142 PacBio reads, 985 convex alignments): unmodified
                                           (CPU ConvexAlignFast), a private ConvexAlignHip per worker, all workers sharing
                                           one BatchingAligner, and that plus the scoring plugin on the device;
                                           SAM compared with the recorded output of the unmodified reference.
    e2e_rates.py --synthetic-sv N [threads]  ONT-like 8-30 kb reads (20 % error) with inversions / deletions / insertions against the
                                           reference in a third of them, -x ont: ngmlr's split-read path (BASELINE.json configs[4]'s shape)
    e2e_rates.py --synthetic-rep N [threads] PacBio-like 10 kb reads on a reference with repeat families and microsatellites: several
                                           candidate regions per sub-read, close scores (what the candidate search and MAPQ see on a real genome)
    e2e_rates.py --synthetic-big N [threads] the same reads on a 512 Mbp reference with repeats (the 1 GB k-mer table leaves every cache): the
                                           reference's only published metric is a genome-sized run (README.md:25)
    e2e_rates.py --synthetic N [threads]   BASELINE.md section 2's workload: N synthetic PacBio-like 10 kb reads (15 % error,
                                           ins:del:sub 6:3:1, half of them reverse-complemented) on a 2 Mbp random reference;
                                           ngmlr_ref at -t nproc' (best of a few thread counts) against the batched drop-ins with
                                           many more workers than cores (reads in flight are what fills a launch); SAM of the
                                           drop-in compared with the SAM ngmlr_ref produced in this very run.

Environment: E2E_POOL="[all@]t:K:target:hold_us[:ENV=v...],..." (pool binaries: CS threads, contexts, batch target / hold, extra environment),
E2E_ONLY=binary (no reference run), E2E_QUICK / E2E_SKIP_OLD (fewer runs), E2E_VERBOSE (timeline and trace lines of the run's stderr),
E2E_BIG_MBP (--synthetic-big: reference size, 512), and for --synthetic: E2E_READ_LEN=lo:hi, E2E_REF_LEN=bases, E2E_CONTIGS=n (reference as n sequences, a fifth of the reads flush with a contig end).

Wall clock includes ngmlr's start-up (reference encoding + index); `map` is the wall clock minus the
index construction time ngmlr reports (thread start-up + mapping + exit).  Says how the drop-in behaves inside the real pipeline, not how fast the kernels are."""
import gzip
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E2E = os.path.join(ROOT, "tests", "golden", "e2e")
tmp = tempfile.mkdtemp()


PRESET = "pacbio"      # -x preset of the runs (the SV workload uses ont)


def write_sv_workload(path_fa, path_fq, n_reads, seed=2026, L=3_000_000):
    """BASELINE.json configs[4]'s shape at a size a CPU reference run still finishes: ONT-like reads of 8-30 kb (20 % error,
    ins:del:sub 4:4:2) on a random reference, a third of them carrying a structural variant against it -- an inverted segment of
    1-3 kb, a deletion of 0.5-2 kb, or an insertion of 0.3-1 kb of foreign sequence -- so that ngmlr's split-read path runs
    (several intervals per read, reverse-strand segments, realignment).  -> read bases"""
    from ngmlr_amd import synth
    rng = np.random.default_rng(seed)
    ref = synth.random_ref(rng, L)
    with open(path_fa, "w") as f:
        f.write(">synthSV\n")
        s_ = ref.tobytes().decode()
        for i in range(0, L, 80):
            f.write(s_[i:i + 80] + "\n")
    bases = 0
    with open(path_fq, "w") as f:
        for i in range(n_reads):
            n = int(rng.integers(8000, 30000))
            a = int(rng.integers(0, L - n - 4000))
            w = ref[a:a + n].copy()
            kind = int(rng.integers(0, 6))
            if kind == 0:                                   # inversion
                m = int(rng.integers(1000, 3000)); p0 = int(rng.integers(2000, n - m - 2000))
                w = np.concatenate([w[:p0], synth.revcomp(w[p0:p0 + m]), w[p0 + m:]])
            elif kind == 1:                                 # deletion in the read
                m = int(rng.integers(500, 2000)); p0 = int(rng.integers(2000, n - m - 2000))
                w = np.concatenate([w[:p0], w[p0 + m:]])
            elif kind == 2:                                 # insertion of foreign sequence
                m = int(rng.integers(300, 1000)); p0 = int(rng.integers(2000, n - 2000))
                w = np.concatenate([w[:p0], synth.random_ref(rng, m), w[p0:]])
            q = synth.mutate(rng, w, 0.20, (4, 4, 2))
            if rng.random() < 0.5:
                q = synth.revcomp(q)
            bases += len(q)
            f.write("@sv%d_%d_%d\n%s\n+\n%s\n" % (i, a, kind, q.tobytes().decode(), "I" * len(q)))
    return bases


def write_repeat_workload(path_fa, path_fq, n_reads, seed=2027, L=3_000_000, dense=0):
    """A reference a k-mer vote has to work on: six repeat families (a 6-15 kb unit copied 8-20 times, every copy 1-4 % diverged from
    the unit), 150 microsatellite stretches of 200-800 bp and the rest random -- sub-reads then have several candidate regions, reads
    several scored locations with close scores (MAPQ < 60, the retry ladder of the vote tables, max-cmrs).  PacBio-like 10 kb reads,
    a third of them started inside a repeat copy.  -> read bases"""
    from ngmlr_amd import synth
    rng = np.random.default_rng(seed)
    ref = synth.random_ref(rng, L)
    copies = []
    for fam in range(6):
        unit = synth.random_ref(rng, int(rng.integers(6000, 15000)))
        for c in range(int(rng.integers(8, 21))):
            v = synth.mutate(rng, unit, float(rng.uniform(0.01, 0.04)), (1, 1, 8))
            a = int(rng.integers(20000, L - 40000))
            ref[a:a + len(v)] = v
            copies.append((a, len(v)))
    if dense:
        # one short unit in `dense` nearly identical copies: every k-mer of it has hundreds of locations (below the table's cutoff of
        # 1 000), so a sub-read from a copy votes for more bins than the device kernel's LDS vote map holds (its HBM-table form runs)
        unit = synth.random_ref(rng, 400)
        for c in range(dense):
            v = synth.mutate(rng, unit, 0.01, (1, 1, 8))
            a = int(rng.integers(20000, L - 40000))
            ref[a:a + len(v)] = v
            copies.append((a, len(v)))
    for i in range(150):
        motif = synth.random_ref(rng, int(rng.integers(1, 5)))
        n = int(rng.integers(200, 800))
        a = int(rng.integers(20000, L - 40000))
        ref[a:a + n] = np.tile(motif, n // len(motif) + 1)[:n]
    with open(path_fa, "w") as f:
        f.write(">synthRep\n")
        s_ = ref.tobytes().decode()
        for i in range(0, L, 80):
            f.write(s_[i:i + 80] + "\n")
    bases = 0
    with open(path_fq, "w") as f:
        for i in range(n_reads):
            n = int(rng.integers(9000, 11000))
            if i % 3 == 0:
                ca, cl = copies[int(rng.integers(0, len(copies)))] if not (dense and i % 2 == 0) else copies[-1 - int(rng.integers(0, dense))]
                a = max(0, min(L - n - 1, ca + int(rng.integers(-3000, max(cl - 3000, 1)))))
            else:
                a = int(rng.integers(0, L - n - 1))
            q = synth.mutate(rng, ref[a:a + n], 0.15, (6, 3, 1))
            if rng.random() < 0.5:
                q = synth.revcomp(q)
            bases += len(q)
            f.write("@rep%d_%d\n%s\n+\n%s\n" % (i, a, q.tobytes().decode(), "I" * len(q)))
    return bases


def write_plain_workload(fa, fq, n_reads, rng, L):
    """BASELINE.md section 2's workload: PacBio-like reads (15 % error, ins:del:sub 6:3:1, half of them reverse-complemented) of
    E2E_READ_LEN bases (default 9-11 kb) on a random reference of L bases in E2E_CONTIGS sequences.  -> read bases"""
    from ngmlr_amd import synth
    ref = synth.random_ref(rng, L)
    n_ctg = int(os.environ.get("E2E_CONTIGS", "1"))      # E2E_CONTIGS=16: the reference as 16 sequences, a fifth of the reads flush with a contig's start or end
    clen = L // n_ctg
    with open(fa, "w") as f:
        s = ref.tobytes().decode()
        for c in range(n_ctg):
            f.write(">synth2M\n" if n_ctg == 1 else ">ctg%d\n" % c)
            for i in range(c * clen, (c + 1) * clen if n_ctg > 1 else L, 80):
                f.write(s[i:min(i + 80, (c + 1) * clen if n_ctg > 1 else L)] + "\n")
    bases = 0
    lo, hi = [int(x) for x in os.environ.get("E2E_READ_LEN", "9000:11000").split(":")]      # E2E_READ_LEN=50000:120000: ultra-long reads
    with open(fq, "w") as f:
        for i in range(n_reads):
            n = int(rng.integers(lo, hi))
            if n_ctg == 1:
                a = int(rng.integers(0, L - hi))
            else:
                c = int(rng.integers(0, n_ctg))
                edge = int(rng.integers(0, 10))
                a = c * clen + (0 if edge == 0 else clen - n if edge == 1 else int(rng.integers(0, clen - n)))
            w = ref[a:a + n]
            q = synth.mutate(rng, w, 0.15, (6, 3, 1))
            if rng.random() < 0.5:
                q = synth.revcomp(q)
            bases += len(q)
            f.write("@r%d_%d\n%s\n+\n%s\n" % (i, a, q.tobytes().decode(), "I" * len(q)))
    return bases


def _pool_numbers(line_):
    """the AlignPool statistics line of a run -> dict (None for a binary without the pool)"""
    if not line_:
        return None
    m = re.search(r"(\d+) reads on (\d+) (user-level )?contexts \(limit (\d+)\)(?: over (\d+) carrier threads)?.*?at most (\d+) reads in flight", line_)
    if not m:
        return None
    out = {"kind": "user-level contexts (fibers) on carrier threads" if m.group(3) else "pthreads", "created": int(m.group(2)), "limit": int(m.group(4)),
           "carrier_threads": int(m.group(5)) if m.group(5) else None, "max_reads_in_flight": int(m.group(6))}
    m = re.search(r"in ([0-9.]+) s:", line_) or re.search(r"over ([0-9.]+) s:", line_)
    if m:
        out["pool_lifetime_s"] = float(m.group(1))
    m = re.search(r"carriers ran read code ([0-9.]+) % of their time \(([0-9.]+) CPU-s\)", line_)
    if m:
        out["carrier_busy_share"], out["read_code_cpu_s"] = float(m.group(1)) / 100.0, float(m.group(2))
    return out


def _lock_numbers(line_):
    """ngmlr's serial input stage (parse + split under one lock, reference src/NGM.cpp:190-244) as the pool's probe saw it"""
    if not line_:
        return None
    m = re.search(r"held ([0-9.]+) s for (\d+) reads in (\d+) batches \(([0-9.]+) us per read\), CS threads waited ([0-9.]+) s", line_)
    if not m:
        return None
    return {"held_s": float(m.group(1)), "reads": int(m.group(2)), "us_per_read": float(m.group(4)), "cs_threads_waited_s": float(m.group(5)),
            "what": "_NGM::GetNextReadBatch: reads are parsed and split into sub-reads by one thread at a time; a lower bound of the mapping wall clock that is not the device path's"}


def write_big_workload(fa, fq, n_reads, seed=31):
    """a reference whose k-mer table leaves every cache -- ngmlr_amd.synth.big_reference (512 Mbp in 8 contigs,
    24 repeat families, 600 microsatellites; the reference of bench.py's candidate_search_big) -- and PacBio-like 10 kb reads drawn
    uniformly from it (15 % error 6:3:1, half of them reverse-complemented).  -> read bases"""
    from ngmlr_amd import synth
    mbp = int(os.environ.get("E2E_BIG_MBP", "512"))      # (2048: a 3.2 GB table, 5 000 votes per sub-read -- every search through the table in HBM)
    contigs = synth.big_reference(mbp << 20, n_contigs=8)
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">big%d\n" % i)
            b = c.tobytes()
            for a in range(0, len(b), 1 << 20):
                f.write(b[a:a + (1 << 20)] + b"\n")
    rng = np.random.default_rng(seed)
    bases = 0
    with open(fq, "w") as f:
        for i in range(n_reads):
            c = contigs[int(rng.integers(0, len(contigs)))]
            n = int(rng.integers(9000, 11000))
            a = int(rng.integers(0, len(c) - n - 1))
            q = synth.mutate(rng, c[a:a + n], 0.15, (6, 3, 1))
            if rng.random() < 0.5:
                q = synth.revcomp(q)
            bases += len(q)
            f.write("@b%d_%d\n%s\n+\n%s\n" % (i, a, q.tobytes().decode(), "I" * len(q)))
    return bases


def pipeline_summary(n_reads=20000, t_ref=32, spec=(20, 4096, 2048, 10000), extra_env=None):
    """What bench.py puts beside its line as `e2e_pipeline`: the reference's own ngmlr, unmodified (ngmlr_ref, CPU) against the
    build with every drop-in bound (ngmlr_hip_all: alignment, sub-read scoring, candidate search, SAM records on the device
    path, alignment contexts off the CS threads), same synthetic reads, SAM compared record by record.  -> dict (or a dict
    with `skipped` when the binaries are not there: they are built from /root/reference by tools/build_ngmlr_hip.sh)."""
    for b_ in ("ngmlr_ref", "ngmlr_hip_all"):
        if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", b_)):
            return {"skipped": "oracle/_ref/%s not built (tools/build_ngmlr_hip.sh needs /root/reference)" % b_}
    rng = np.random.default_rng(2025)
    L = 2000000
    fa, fq = os.path.join(tmp, "ps_ref.fa"), os.path.join(tmp, "ps_reads.fq")
    bases = write_plain_workload(fa, fq, n_reads, rng, L)
    t, ctx, target, hold = spec
    # CVX_CS_BATCH: reads per CS batch = per device search / scoring call (the reference's cBatchSize is 10); 20 CS threads + the pool's
    # carriers (half of the container's CPU quota) stay below a 16-core quota -- more threads get the whole process throttled
    env = {"CVX_POOL_CONTEXTS": str(ctx), "CVX_BATCH_TARGET": str(target), "CVX_BATCH_HOLD_US": str(hold), "CVX_CS_BATCH": "20"}
    env.update(extra_env or {})
    r0 = run("ngmlr_ref", t_ref, fa, fq)
    r1 = run("ngmlr_hip_all", t, fa, fq, env)

    def side(r, threads):
        infl = re.search(r"a launch was in flight ([0-9.]+) %", r["stats"] or "")
        parked = re.search(r"\(([0-9.]+) ms per alignment\)", r["stats"] or "")
        return {"threads": threads, "rc": r["rc"], "wall_s": r["wall"], "map_s": r["map_s"], "sam_records": len(r["recs"]),
                "mapped_Gbp_per_h": (bases / r["map_s"] * 3.6e-6) if r["map_s"] else None,
                "alignments": r["launch"][0] if r["launch"] else None, "device_launches": r["launch"][1] if r["launch"] else None,
                "tiles_per_launch": (r["launch"][0] / max(r["launch"][1], 1)) if r["launch"] else None,
                "launch_in_flight_share": float(infl.group(1)) / 100.0 if infl else None,
                "ms_parked_per_alignment": float(parked.group(1)) if parked else None,
                "cpu_seconds": r["cpu_total_s"], "cpu_seconds_by_thread_class": r["cpu_by_class"], "peak_rss_mb": r["peak_rss_mb"],
                "alignment_contexts": _pool_numbers(r.get("pool_stats")), "input_lock": _lock_numbers(r.get("input_lock")),
                "corridors_as_closed_forms": list(r["closed_forms"]) if r.get("closed_forms") else None}
    out = {"reads": n_reads, "read_bases": bases, "reference_bases": L, "host": effective_cores(),
           "ngmlr_ref": side(r0, t_ref), "ngmlr_hip_all": side(r1, t),
           "sam_identical": bool(r0["rc"] == 0 and r1["rc"] == 0 and r0["recs"] == r1["recs"]),
           "wall_ratio": r1["wall"] / max(r0["wall"], 1e-9),
           "map_ratio": (r1["map_s"] / r0["map_s"]) if (r0["map_s"] and r1["map_s"]) else None,
           "settings": "ngmlr_hip_all -t %d, %d alignment contexts, batch target %d tiles / %d us, %s reads per CS batch; ngmlr_ref -t %d" % (t, ctx, target, hold, env["CVX_CS_BATCH"], t_ref),
           "what": "the reference's ngmlr binary built from /root/reference with the drop-ins (tools/build_ngmlr_hip.sh) against the unmodified build, "
                   "-x pacbio, synthetic 10 kb reads on a 2 Mbp random reference; wall includes ngmlr's index construction, map = wall - index time; "
                   "CPU seconds sampled from /proc/<pid>/task by thread name; peak RSS = VmHWM"}
    return out


def run(name, t, ref, fq, extra_env=None):
    binary = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(binary):
        return None
    ref_copy = os.path.join(tmp, "%s_%d_%s" % (name, t, os.path.basename(ref)))      # own copy: every run encodes its reference afresh
    open(ref_copy, "wb").write(open(ref, "rb").read())
    env = dict(os.environ)
    env.update(extra_env or {})
    # stdout / stderr to files (a 500 MB SAM through a pipe would bill this script's reading to the run), and a sampler over
    # /proc/<pid>/task: CPU seconds by thread name (cvx-dispatch, cvx-context, cvx-pack, ngm-cs; unnamed = ngmlr's own and the
    # HIP runtime's), plus the cgroup's throttling counters -- a CPU quota stalls every thread of the process at once
    out_path, err_path = os.path.join(tmp, "run.out"), os.path.join(tmp, "run.err")
    cg0 = cgroup_cpu_stat()
    t0 = time.perf_counter()
    stamp = os.environ.get("E2E_STDERR_TIMES") is not None      # every stderr line with the second it arrived at (where a run's wall clock goes outside the pool)
    with open(out_path, "wb") as fo, open(err_path, "wb") as fe:
        proc = subprocess.Popen([binary, "--skip-write", "-x", PRESET, "-t", str(t), "-R", "0.01", "--no-progress", "-r", ref_copy, "-q", fq],
                                stdout=fo, stderr=subprocess.PIPE if stamp else fe, cwd=tmp, env=env)
        if stamp:
            import threading

            def pump():
                for raw in proc.stderr:
                    fe.write(raw)
                    sys.stderr.write("  [%7.3f s] %s\n" % (time.perf_counter() - t0, raw.decode("utf-8", "replace").rstrip()[:160]))
            pump_thread = threading.Thread(target=pump, daemon=True)
            pump_thread.start()
        ticks = {}
        peak_rss_kb = 0
        hz = os.sysconf("SC_CLK_TCK")
        while proc.poll() is None:
            try:
                for l_ in open("/proc/%d/status" % proc.pid):
                    if l_.startswith("VmHWM:"):
                        peak_rss_kb = max(peak_rss_kb, int(l_.split()[1]))
                        break
            except (OSError, ValueError):
                pass
            try:
                for tid in os.listdir("/proc/%d/task" % proc.pid):
                    try:
                        st = open("/proc/%d/task/%s/stat" % (proc.pid, tid)).read()
                    except OSError:
                        continue
                    comm = st[st.index("(") + 1:st.rindex(")")]
                    f = st[st.rindex(")") + 2:].split()
                    ticks[tid] = (comm, int(f[11]) + int(f[12]))
            except OSError:
                pass
            time.sleep(0.02)
    dt = time.perf_counter() - t0
    if stamp:
        pump_thread.join(timeout=5)
    cg1 = cgroup_cpu_stat()

    class Res:
        pass
    res = Res()
    res.returncode = proc.returncode
    res.stdout = open(out_path, "r", errors="replace").read()
    res.stderr = open(err_path, "r", errors="replace").read()
    by = {}
    for comm, tk in ticks.values():
        c, n = by.get(comm, (0.0, 0))
        by[comm] = (c + tk / hz, n + 1)
    cpu_note = "cpu %.1f s sampled (" % sum(c for c, _ in by.values()) + ", ".join("%s x%d: %.1f" % (k, n, c) for k, (c, n) in sorted(by.items(), key=lambda kv: -kv[1][0])) + ")"
    if cg0 and cg1:
        cpu_note += "; cgroup: %.1f cpu-s, throttled %d times for %.1f s" % ((cg1.get("usage_usec", 0) - cg0.get("usage_usec", 0)) * 1e-6,
                                                                           cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                                                                           (cg1.get("throttled_usec", cg1.get("throttled_time", 0) * 1e-3) - cg0.get("throttled_usec", cg0.get("throttled_time", 0) * 1e-3)) * 1e-6)
    recs = sorted(l for l in res.stdout.splitlines() if l and not l.startswith("@"))
    m = re.search(r"SharedAligner: (\d+) alignments in (\d+) device launches", res.stderr)
    mp = re.search(r"Overall time for creating RefTable: ([0-9.]+)s", res.stderr)
    st = re.search(r"SharedAligner: \d+ workers over.*", res.stderr)
    sc = re.search(r"StrippedSWHip: \d+ scoring calls.*", res.stderr)
    se = re.search(r"CandidateSearchHip: \d+ search calls.*", res.stderr)
    po = re.search(r"AlignPool: \d+ reads on.*", res.stderr)
    il = re.search(r"AlignPool: ngmlr's input lock.*", res.stderr)
    cf = re.search(r"SharedAligner: (\d+) of (\d+) corridors travelled as closed forms", res.stderr)
    tx = re.search(r"SharedAligner: text stage on the device.*", res.stderr)
    wn = re.search(r"SharedAligner: \d+ tiles in \d+ launches took their reference as windows.*", res.stderr)
    cpu_by_class = {k: round(c, 2) for k, (c, n) in by.items()}
    return {"peak_rss_mb": peak_rss_kb / 1024.0, "cpu_by_class": cpu_by_class, "cpu_total_s": sum(c for c, _ in by.values()), "wall": dt, "search_stats": se.group(0) if se else None, "cpu_note": cpu_note, "pool_stats": po.group(0) if po else None, "input_lock": il.group(0) if il else None, "text_stats": tx.group(0) if tx else None, "window_stats": wn.group(0) if wn else None,
            "closed_forms": (int(cf.group(1)), int(cf.group(2))) if cf else None, "rc": res.returncode, "recs": recs, "launch": (int(m.group(1)), int(m.group(2))) if m else None, "stats": st.group(0) if st else None, "score_stats": sc.group(0) if sc else None,
            "map_s": dt - float(mp.group(1)) if mp else None, "err": res.stderr[-400:], "full_err": res.stderr}


def line(name, t, r, same):
    l = r["launch"]
    print("%-18s -t %-4d wall %7.2f s%s  rc %d  SAM %s%s" % (
        name, t, r["wall"], ("  map %6.2f s" % r["map_s"]) if r["map_s"] is not None else "", r["rc"], same,
        ("  %d alignments in %d launches (%.1f per launch)" % (l[0], l[1], l[0] / max(l[1], 1))) if l else ""), flush=True)
    if r.get("cpu_note"):
        print("    " + r["cpu_note"], flush=True)
    if r.get("stats"):
        print("    " + r["stats"], flush=True)
    if r.get("score_stats"):
        print("    " + r["score_stats"], flush=True)
    if r.get("search_stats"):
        print("    " + r["search_stats"], flush=True)
    if r.get("pool_stats"):
        print("    " + r["pool_stats"], flush=True)
    if r.get("input_lock"):
        print("    " + r["input_lock"], flush=True)
    if r.get("closed_forms"):
        print("    corridors sent as closed forms: %d of %d" % r["closed_forms"], flush=True)
    for k in ("text_stats", "window_stats"):
        if r.get(k):
            print("    " + r[k], flush=True)
    if os.environ.get("E2E_VERBOSE"):
        for l in r["full_err"].splitlines():
            if "library loaded" in l or "time" in l.lower() or "Done" in l or l.startswith("cvx_search_batch:") or l.startswith("cvx timeline") or l.startswith("cvx launch") or l.startswith("cvx text stage") or l.startswith("cvx dispatcher") or l.startswith("cvx_submit:"):
                print("      | " + l[:460], flush=True)


def cgroup_cpu_stat():
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            return {k: int(v) for k, v in (l.split() for l in open(path))}
        except Exception:
            continue
    return None


def effective_cores():
    """what this process may actually use: affinity mask and the cgroup's CPU quota (cpu.max = quota period)"""
    out = ["affinity %d" % len(os.sched_getaffinity(0))]
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                out.append("cgroup cpu.max %s" % ("unlimited" if txt[0] == "max" else "%.1f cores" % (float(txt[0]) / float(txt[1]))))
            else:
                q = float(txt[0])
                out.append("cgroup quota %s" % ("unlimited" if q < 0 else "%.1f cores" % (q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except Exception:
            continue
    return ", ".join(out)


def synthetic(n_reads, threads, sv=False, rep=False, big=False):
    from ngmlr_amd import synth
    global PRESET
    rng = np.random.default_rng(2025)
    L = int(os.environ.get("E2E_REF_LEN", "2000000"))      # (plain synthetic workload only)
    fa = os.path.join(tmp, "synth_ref.fa")
    fq = os.path.join(tmp, "synth_reads.fq")
    if sv:
        PRESET = "ont"
        L = 3_000_000
        bases = write_sv_workload(fa, fq, n_reads, L=L)
        print("SV workload (ONT-like 8-30 kb reads, 20 % error, a third with an inversion / deletion / insertion; -x ont):")
    elif big:
        L = int(os.environ.get("E2E_BIG_MBP", "512")) << 20
        bases = write_big_workload(fa, fq, n_reads)
        print("genome-sized reference (%d Mbp in 8 contigs, repeat families, microsatellites: the k-mer table is %.1f GB and leaves every cache):" % (L >> 20, (L >> 20) / 512.0))
    elif rep:
        L = 3_000_000
        bases = write_repeat_workload(fa, fq, n_reads, L=L)
        print("repeat workload (six repeat families of 8-20 diverged copies, microsatellites; PacBio-like 10 kb reads, a third from inside a copy):")
    else:
        bases = write_plain_workload(fa, fq, n_reads, rng, L)
    print("synthetic: %d reads, %.1f Mbp, reference %d bp; host has %d hardware threads, %s" % (n_reads, bases / 1e6, L, os.cpu_count(), effective_cores()))
    cores = os.cpu_count() or 8
    base = None
    best = None
    only = os.environ.get("E2E_ONLY")          # one drop-in binary alone, no reference run (SAM then unchecked): quick looks
    for t in ([] if only else sorted({min(cores, 32)} if os.environ.get("E2E_QUICK") else {min(cores, 32), min(cores, 64)})):
        r = run("ngmlr_ref", t, fa, fq)
        if r is None:
            print("ngmlr_ref not built")
            return
        if base is None:
            base = r["recs"]
        line("ngmlr_ref", t, r, "reference" if r["recs"] == base else "DIFFERS from the first ngmlr_ref run")
        if best is None or r["wall"] < best[1]["wall"]:
            best = (t, r)
    # E2E_POOL="t:K:target:hold_us,..." : ngmlr_hip_pool (alignment contexts off the CS threads, align_pool.h) with t CS
    # threads, K contexts, the dispatcher's batch target and hold time
    runs = [(name, t, None, "") for name in ((only,) if only else () if os.environ.get("E2E_SKIP_OLD") else ("ngmlr_hip_batched", "ngmlr_hip_full")) for t in threads]
    for spec in [x for x in os.environ.get("E2E_POOL", "").split(",") if x]:
        f = spec.split(":")
        binary = "ngmlr_hip_pool"                 # "all@16:512:..." = ngmlr_hip_all (the pool + the candidate search on the device)
        if "@" in f[0]:
            binary, f[0] = "ngmlr_hip_" + f[0].split("@")[0], f[0].split("@")[1]
        env = {"CVX_POOL_CONTEXTS": f[1]}
        if len(f) > 2 and int(f[2]) > 0:
            env["CVX_BATCH_TARGET"] = f[2]
        if len(f) > 3:
            env["CVX_BATCH_HOLD_US"] = f[3]
        for extra in f[4:]:
            k, v = extra.split("=")
            env[k] = v
        runs.append((binary, int(f[0]), env, "  [" + " ".join("%s=%s" % kv for kv in sorted(env.items())) + "]"))
    for name, t, env, note in runs:
        if True:
            r = run(name, t, fa, fq, env)
            if note and r is not None:
                print(note.strip(), flush=True)
            if r is None:
                print("%-18s not built" % name)
                continue
            line(name, t, r, "unchecked" if base is None else "identical" if r["recs"] == base else "DIFFERENT (%d vs %d records)" % (len(r["recs"]), len(base)))
            if r["rc"] != 0:
                print(r["err"])
            elif best is not None:
                print("    wall / best ngmlr_ref (-t %d): %.2f   map / map: %.2f   mapped bases per hour of `map`: %.1f Gbp/h vs %.1f Gbp/h" % (
                    best[0], r["wall"] / best[1]["wall"], (r["map_s"] or r["wall"]) / (best[1]["map_s"] or best[1]["wall"]),
                    bases / (r["map_s"] or r["wall"]) * 3.6e-6, bases / (best[1]["map_s"] or best[1]["wall"]) * 3.6e-6))


def test_3(threads):
    fq = os.path.join(tmp, "test_3.fq")
    open(fq, "wb").write(gzip.open(os.path.join(E2E, "test_3_reads.fq.gz"), "rb").read())
    want = [l.rstrip("\n") for l in gzip.open(os.path.join(ROOT, "tests", "golden", "test_3.sorted.sam.gz"), "rt") if l.strip()]
    for name in ("ngmlr_ref", "ngmlr_hip", "ngmlr_hip_batched", "ngmlr_hip_full", "ngmlr_hip_pool", "ngmlr_hip_all"):
        for t in threads:
            r = run(name, t, os.path.join(E2E, "test_3_reference.fasta.gz"), fq, {"CVX_POOL_CONTEXTS": "256"} if name in ("ngmlr_hip_pool", "ngmlr_hip_all") else None)
            if r is None:
                print("%-18s not built" % name)
                break
            line(name, t, r, "identical" if r["recs"] == want else "DIFFERENT (%d records)" % len(r["recs"]))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] in ("--synthetic", "--synthetic-sv", "--synthetic-rep", "--synthetic-big"):
        n = int(args[1]) if len(args) > 1 else 2000
        synthetic(n, [int(x) for x in args[2:]] or [64, 256, 512], sv=args[0] == "--synthetic-sv", rep=args[0] == "--synthetic-rep", big=args[0] == "--synthetic-big")
    else:
        test_3([int(x) for x in args] or [1, 16, 64])
