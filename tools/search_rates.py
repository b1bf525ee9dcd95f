"""Dev tool (GPU box): candidate search (cvx_search_batch_ex).

    search_rates.py [replicas]        the recorded test_3 index (43 797 locations: lives in L2) and its sub-reads, replicated to a
                                      batch that fills the device, against the CPU checker (oracle/cs_oracle.c, one thread)
    search_rates.py --big [Mbp] [n]   a k-mer table the size of a genome's: a synthetic reference of `Mbp` (default 512) million bases
                                      with repeat families and microsatellites (ngmlr_amd.synth.big_reference), its table built by
                                      cvx_index_build (= what ngmlr builds, tests/test_index_cpu.py), `n` (default 100 000) sub-reads of
                                      256 bases; whole call, kernel time from HIP events, votes, parity of a sample against cs_oracle.c

`big_index_rates` is also what bench.py reports as index_stage_device.candidate_search_big."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def big_index_rates(al, mbp=512, n_reads=100000, parity_n=2000, n_contigs=8, pmc_bytes_per_vote=None):
    """-> dict.  al: a ConvexAlignHip (device handle).  The table is built on the device (cvx_index_build_device) and adopted
    (cvx_index_upload) and searched in one cvx_search_batch_ex call per pass; a sample of the lists is compared with the
    CPU restatement of CS::RunRead (oracle/cs_oracle.c) over the very same table."""
    from ngmlr_amd import capi, synth
    from ngmlr_amd.aligner import KmerIndex
    t0 = time.perf_counter()
    contigs = synth.big_reference(mbp << 20, n_contigs=n_contigs)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx5, locs, starts = synth.kmer_table(al.lib, contigs, device=0, keep=True)      # (round 6: built on the device and left there for the upload below)
    t_tab = time.perf_counter() - t0
    reads = synth.sample_subreads(contigs, n_reads)
    bases = sum(len(r) for r in reads)
    k = 13
    t0 = time.perf_counter()
    ix = KmerIndex(al, k, idx5.view(np.dtype([("tab", "<u4"), ("rc", "i1")])), locs, 0)
    t_up = time.perf_counter() - t0
    arena, offsets, pinned = KmerIndex.make_arena(reads, al.lib)
    try:
        ix.search(reads[:512])
        # (a) the string form, through Python's marshalling of 100 000 strings and 100 000 result arrays (round 5's figure)
        c0 = time.perf_counter()
        got, max_hit, misses = ix.search(reads, extras=True)
        dt_strings = time.perf_counter() - c0
        # (b) the arena form: reads back to back in page-locked memory, flat outputs (cvx_search_batch_arena) -- what a caller that keeps
        # its reads resident pays; the candidate buffer is sized by the first call
        ncand, begin, cands, mh2, ms2 = ix.search_arena(arena, offsets)
        cbuf = np.zeros(len(cands) + len(cands) // 8 + 64, dtype=cands.dtype)
        best = None
        for _ in range(3):
            c0 = time.perf_counter()
            ncand, begin, cands, mh2, ms2 = ix.search_arena(arena, offsets, cands=cbuf)
            dt = time.perf_counter() - c0
            kms = al.stage_kernel_ms(capi.STAGE_SEARCH)
            if best is None or dt < best[0]:
                best = (dt, kms)
        dt, kms = best
        # the two forms return the same lists
        same_forms = bool(np.array_equal(mh2, max_hit) and np.array_equal(ms2, misses) and
                          all((g is None and ncand[i] < 0) or (g is not None and ncand[i] == len(g) and np.array_equal(cands[int(begin[i]):int(begin[i]) + int(ncand[i])], g))
                              for i, g in enumerate(got)))
    finally:
        ix.free()
        if pinned:
            pinned[0].cvx_host_free(pinned[1])
    # votes of the batch: every location of every k-mer the table knows, both orientations (what CS::PrefixSearch walks)
    tab = idx5.reshape(-1, 5)[:, :4].copy().view(np.uint32).ravel()
    used = idx5.reshape(-1, 5)[:, 4] != 0
    row_len = np.zeros(len(tab), dtype=np.int64)
    row_len[:-1] = np.where(used[:-1], np.diff(tab.astype(np.int64)), 0)
    votes = 0
    sectors = 0      # 64-byte sectors the look-ups of a sub-read need at the least: one index record per k-mer and orientation, its row of locations
    n_v = min(len(reads), 4000)
    code = np.zeros(256, dtype=np.int64)
    for ch, v in ((65, 0), (67, 1), (84, 2), (71, 3)):
        code[ch] = v
    for r in reads[:n_v]:
        a = code[np.frombuffer(r, dtype=np.uint8)]
        if len(a) < k:
            continue
        w = np.lib.stride_tricks.sliding_window_view(a, k)
        p = (w * (4 ** np.arange(k - 1, -1, -1))).sum(axis=1)
        wc = (w ^ 2)[:, ::-1]
        prc = (wc * (4 ** np.arange(k - 1, -1, -1))).sum(axis=1)
        votes += int(row_len[p].sum() + row_len[prc].sum())
        sectors += 2 * len(p) + int(((row_len[p] * 4 + 63) // 64).sum() + ((row_len[prc] * 4 + 63) // 64).sum())
    votes_per_read = votes / max(n_v, 1)
    sectors_per_read = sectors / max(n_v, 1)
    # parity of a sample against the CPU restatement over the same table
    from oracle.pyoracle import SearchOracle
    orc = SearchOracle(raw=(k, 0, idx5, locs))
    step = max(1, len(reads) // max(parity_n, 1))
    sample = list(range(0, len(reads), step))[:parity_n]
    bad, first = 0, None
    c0 = time.perf_counter()
    for i in sample:
        w = orc.search(reads[i], cap=1 << 15)
        g = got[i]
        ok = (w["n"] < 0 and g is None) or (g is not None and w["n"] == len(g) and np.array_equal(g["location"], w["loc"]) and
                                            np.array_equal(g["score"], w["score"]) and np.array_equal(g["reverse"], w["rev"]) and
                                            float(max_hit[i]) == float(w["max_hit"]) and int(misses[i]) == int(w["kmer_misses"]))
        if not ok:
            bad += 1
            first = first or "sub-read %d: %s vs %s candidates" % (i, "none" if g is None else len(g), w["n"])
    dt_cpu = time.perf_counter() - c0
    orc.close()
    n_lists = sum(1 for g in got if g is not None)
    n_cand = sum(len(g) for g in got if g is not None)
    out = {
        "reference_bases": int(sum(len(c) for c in contigs)), "contigs": len(contigs), "kmer_len": k, "locations": int(len(locs)),
        "table_bytes": int(len(idx5) + 4 * len(locs)), "sub_reads": len(reads), "sub_read_bases": int(bases),
        "seconds": dt, "sub_reads_per_s": len(reads) / dt, "kernel_ms": kms, "whole_call_over_kernels": dt * 1e3 / max(kms, 1e-9),
        "string_form": {"seconds": dt_strings, "sub_reads_per_s": len(reads) / dt_strings, "equal_to_the_arena_form": same_forms,
                        "what": "cvx_search_batch_ex through this tool's Python marshalling of 100 000 strings in and 100 000 arrays out (round 5's `whole call`)"}, "kernel_sub_reads_per_s": len(reads) / max(kms * 1e-3, 1e-9),
        "votes_per_sub_read": votes_per_read, "kernel_votes_per_s": votes_per_read * len(reads) / max(kms * 1e-3, 1e-9),
        "lists": n_lists, "candidates": n_cand,
        "parity": "%d/%d lists equal to the CPU restatement of CS::RunRead over the same table (entries, order, maxHitNumber, kCount)" % (len(sample) - bad, len(sample)),
        "parity_detail": first, "cpu_checker_sub_reads_per_s_one_thread": len(sample) / max(dt_cpu, 1e-9),
        "setup_seconds": {"reference": t_ref, "cvx_genome_encode + cvx_index_build_device": t_tab, "cvx_index_upload (adopts the device's copy)": t_up},
        "bytes_per_vote": pmc_bytes_per_vote,
        "random_sector_peak": None, "sectors_per_sub_read": sectors_per_read,
        "kernel_sector_reads_per_s": sectors_per_read * len(reads) / max(kms * 1e-3, 1e-9),
        "bound": "NOT the random-access rate of HBM (tools/ubench_gather.hip puts that at ~51 G sectors/s on this device; the kernels read "
                 "`kernel_sector_reads_per_s`): the latency of a wave's dependent LDS round trips at 1.25 waves per SIMD -- a read owns a wave "
                 "whose vote map is sized from its vote count (2^9..2^12 slots of 12 bytes, at most 3/4 full; a sub-read of this genome votes for "
                 "1 200-1 450 bins: 2^11 slots, 30 KB of LDS with rList and the read), five reads fit a CU, and each of a sub-read's ~21 vote "
                 "batches is a chain of ~20 LDS round trips (row of the vote, probe, claim, duplicates, score); the loads from HBM run a batch "
                 "/ a chunk ahead.  Round 5's fixed 1 024-bin map held none of these sub-reads: each was cast twice (LDS attempt discarded, "
                 "then the 1 MB table in HBM)",
        "what": "cvx_search_batch_arena (reads back to back in page-locked memory, flat outputs) over a %d Mbp synthetic reference with repeat families and microsatellites; table by cvx_index_build_device (byte-identical "
                "to ngmlr's own: tests/test_gpu_index.py), resident in HBM; kernel_ms = every kernel of the call from HIP events (cvx_stage_kernel_ms)" % mbp}
    peak = random_sector_peak()
    out["random_sector_peak"] = peak
    if peak and "G_accesses_per_s" in peak:
        out["kernel_frac_of_random_peak"] = out["kernel_sector_reads_per_s"] / (peak["G_accesses_per_s"] * 1e9)
    return out


def random_sector_peak():
    """tools/ubench_gather.hip (built into tools/bin by __graft_entry__.build()): G random 64-byte-sector reads / s of this device,
    the ceiling "HBM random access" is measured against -> dict or None"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "ubench_gather")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120).stdout
        last = [l for l in out.splitlines() if l.startswith("RANDOM_SECTOR_PEAK")][-1].split()
        return {"G_accesses_per_s": float(last[1]), "GB_per_s_of_64B_sectors": float(last[2]),
                "what": "tools/ubench_gather.hip: best of 8 / 16 / 64 bytes per access, 1-16 loads in flight per lane, 1-8 waves per SIMD, over 4 GiB"}
    except Exception as e:
        return {"error": str(e)}


def main():
    from ngmlr_amd.aligner import ConvexAlignHip, KmerIndex
    if len(sys.argv) > 1 and sys.argv[1] == "--big":
        mbp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
        al = ConvexAlignHip()
        import json
        r = big_index_rates(al, mbp, n)
        al.close()
        print(json.dumps(r, indent=1))
        return
    from oracle.pyoracle import SearchFixture, SearchOracle
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    fx = SearchFixture(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cs_test_3.npz"))
    reads = [s for s in fx.seqs if len(s) > 0]
    bases = sum(len(s) for s in reads)
    print("index: k=%d, %d locations; %d recorded sub-reads, %.2f Mbp, mean %.0f bp" % (fx.k, len(fx.locs), len(reads), bases / 1e6, bases / len(reads)))
    o = SearchOracle(fx)
    t0 = time.perf_counter()
    n_c = 0
    for s in reads[:2000]:
        n_c += max(o.search(s)["n"], 0)
    dt_cpu = time.perf_counter() - t0
    o.close()
    print("CPU checker (1 thread): %d reads in %.2f s = %.0f reads/s, %d candidates" % (min(len(reads), 2000), dt_cpu, min(len(reads), 2000) / dt_cpu, n_c))
    al = ConvexAlignHip()
    idx, locs = fx.index_arrays()
    ix = KmerIndex(al, fx.k, idx, locs, fx.unit_offset)
    for r in (1, rep):
        batch = reads * r
        ix.search(batch[:64])
        t0 = time.perf_counter()
        got = ix.search(batch)
        dt = time.perf_counter() - t0
        print("device: %6d reads (%.1f Mbp) in %7.1f ms = %9.0f reads/s = %.2f Gbp/h of sub-read bases (whole call: marshalling, tables, kernels, lists back)" % (
            len(batch), bases * r / 1e6, dt * 1e3, len(batch) / dt, bases * r / dt * 3.6e-6))
    ix.free()
    al.close()


if __name__ == "__main__":
    main()
