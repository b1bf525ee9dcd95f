"""Dev tool (GPU box): a randomised parity sweep of the HIP path against the oracle, far beyond the seeds the test suite pins.

    fuzz_parity.py SECONDS [seed0] [scoring]

Round r (seed = seed0 + r) draws tiles from every generator the tests use -- the corridor zoo (all constructors, error models, odd
symbols), edge-hugging tiles (anchors corridors shifted by about their half-width, mult 1-3), the ONT mix with retries, short reads,
engineered long gaps, early-best tiles -- runs them through cvx_align under one set of runtime knobs drawn for the round (the default
most of the time; chained row blocks of every height, no penalty table, gangs, no chaining, the int16-run kernels, the catch-all kernel, other walk widths) and compares every
tile with oracle/convex_oracle.c on 16 host threads: status, score bits, CIGAR, MD, NM, clips, offsets, the per-position profile and
the best cell.  Stops at the first round with a mismatch (exit code 1) and prints the tiles' tags and shapes; otherwise runs until
SECONDS are over and prints the totals.  With a third argument `scoring` every round also draws its scoring parameters (the
scorings of tests/test_gpu_parity.py's EXOTIC_SCORING -- the regime where the reference's SSE path and the scalar recurrence
disagree --, decays 0 / 0.01 / 0.07 / 0.5 around the penalty table's switch, ...) and the checker is the reference itself
(oracle/_ref) when it is there.  Results of two runs: profiles/r05_fuzz_parity.txt, r05_fuzz_parity_scoring.txt."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ngmlr_amd import synth                      # noqa: E402
from ngmlr_amd.aligner import ConvexAlignHip     # noqa: E402
from oracle.pyoracle import Oracle, same_alignment  # noqa: E402
from tests import util                           # noqa: E402
from tests.test_gpu_parity import _edge_hugging_tiles, _sv_tile, _early_best_tiles, EXOTIC_SCORING  # noqa: E402

SCORINGS = EXOTIC_SCORING + [
    dict(match=2.0, mismatch=-5.0, gap_open=-5.0, gap_extend=-5.0, gap_extend_min=-1.0, gap_decay=d) for d in (0.0, 0.01, 0.07, 0.5)] + [
    dict(match=2.0, mismatch=-5.0, gap_open=-5.0, gap_extend=-2.0, gap_extend_min=-2.0, gap_decay=0.0),
    dict(match=1.0, mismatch=-1.0, gap_open=-1.0, gap_extend=-1.0, gap_extend_min=-0.5, gap_decay=0.15),      # the ont preset's commented-out values (src/ArgParser.cpp:261-265)
    dict(match=3.0, mismatch=-3.0, gap_open=-1.0, gap_extend=-1.0, gap_extend_min=-0.5, gap_decay=0.15),
    dict(match=5.0, mismatch=-4.0, gap_open=-8.0, gap_extend=-6.0, gap_extend_min=-0.25, gap_decay=0.2)]

KNOBS = [{}, {}, {}, {"CVX_TUNE_MAX_M": "1"}, {"CVX_TUNE_MAX_M": "1", "CVX_TUNE_CHAIN_M": "2"}, {"CVX_TUNE_MAX_M": "2", "CVX_TUNE_CHAIN_M": "4"},
         {"CVX_TUNE_FORCE_WRAP16": "1"}, {"CVX_TUNE_SSE_VARIANT": "1"}, {"CVX_TUNE_PEN_TABLE": "0"}, {"CVX_TUNE_GANGS": "1"}, {"CVX_TUNE_SMALL_BATCH": "1"},
         {"CVX_TUNE_BT_GROUP": "4"}, {"CVX_TUNE_BT_GROUP": "32"}, {"CVX_TUNE_BT_PER_CLASS": "0"}, {"CVX_TUNE_LATE_MIN": "1"},
         {"CVX_TUNE_BT_GROUP": "-1"}, {"CVX_TUNE_BT_GROUP": "16"}, {"CVX_TUNE_WIDE_PRIO": "0"}, {"CVX_TUNE_CHAIN_LDS_KB": "0"}, {"CVX_TUNE_CHAIN_LDS_KB": "16"}]
THREADS = 16


def draw(seed):
    rng = np.random.default_rng(seed)
    tiles = util.tile_zoo(seed=seed, n=240, max_w=int(rng.choice([900, 3000, 6000])))
    tiles += _edge_hugging_tiles(rng, 150, mult=int(rng.choice([1, 1, 2, 3])))
    tiles += synth.workload_ont(80, seed=seed, max_len=int(rng.choice([3000, 8000])))
    tiles += synth.workload_short(100, seed=seed)
    for _ in range(6):
        g = int(rng.integers(1, 420))
        tiles.append(_sv_tile(rng, int(rng.integers(200, 500)), [g], [], "full"))
        tiles.append(_sv_tile(rng, int(rng.integers(200, 500)), [], [g], "full"))
        tiles.append(_sv_tile(rng, int(rng.integers(200, 500)), [g, int(rng.integers(1, 40))], [int(rng.integers(1, 40)), g], "endpoints"))
    tiles += _early_best_tiles(rng, 8)
    return tiles


def oracle_all(oracles, tiles):
    want = [None] * len(tiles)
    def work(k):
        o = oracles[k]
        for i in range(k, len(tiles), len(oracles)):
            w = o.align(tiles[i])
            f = o.last_fwd() if (w["ret"] >= 0 and o.kind == "port") else None
            want[i] = (w, (f["best_x"], f["best_y"]) if f else None)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(oracles))]
    for t in ths: t.start()
    for t in ths: t.join()
    return want


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    vary = len(sys.argv) > 3 and sys.argv[3] == "scoring"
    from oracle.pyoracle import have_ref
    oracles = [Oracle("port") for _ in range(THREADS)]
    t_start = time.time()
    totals = {"rounds": 0, "tiles": 0, "valid": 0, "cells": 0}
    r = 0
    while time.time() - t_start < budget:
        seed = seed0 + r
        knobs = KNOBS[int(np.random.default_rng(seed ^ 0x5bd1).integers(0, len(KNOBS)))]
        tiles = draw(seed)
        sc = {}
        if vary:
            sc = SCORINGS[int(np.random.default_rng(seed ^ 0x77).integers(0, len(SCORINGS)))]
            params = (sc["match"], sc["mismatch"], sc["gap_open"], sc["gap_extend"], sc["gap_extend_min"], sc["gap_decay"])
            for o in oracles: o.close()
            oracles = [Oracle("reference" if have_ref() else "port", params) for _ in range(THREADS)]
        for k, v in knobs.items(): os.environ[k] = v
        al = ConvexAlignHip(device=0, **sc)
        for k in knobs: os.environ.pop(k)
        t0 = time.time()
        got = al.batch_align(tiles)
        t1 = time.time()
        al.close()
        want = oracle_all(oracles, tiles)
        t2 = time.time()
        bad = []
        for t, g, (w, cell) in zip(tiles, got, want):
            d = same_alignment(w, g)
            if d is None and cell is not None and cell != (g["best_x"], g["best_y"]):
                d = "argmax cell %r != %r" % ((g["best_x"], g["best_y"]), cell)
            if g["status"] == -1:
                d = "fell outside every device kernel"
            if d:
                bad.append((t.tag, t.H, t.W, int(t.row_length[0]), int(t.row_offset[0]), d))
        nv = sum(1 for w, _ in want if w["ret"] >= 0)
        cells = int(sum(int(np.asarray(t.row_length, dtype=np.int64).sum()) for t in tiles))
        totals["rounds"] += 1; totals["tiles"] += len(tiles); totals["valid"] += nv; totals["cells"] += cells
        print("seed %d knobs %s%s: %d tiles (%d with an alignment, %.2f G corridor cells), device %.2f s, oracle %.2f s: %d mismatches" % (
            seed, knobs or "default", (" scoring %s vs %s" % (tuple(sc.values()), oracles[0].kind)) if sc else "", len(tiles), nv, cells / 1e9, t1 - t0, t2 - t1, len(bad)), flush=True)
        if bad:
            for b in bad[:20]: print("    ", b)
            print("FAILED after %d rounds" % totals["rounds"])
            sys.exit(1)
        r += 1
    print("fuzz_parity: %d rounds, %d tiles (%d with an alignment), %.1f G corridor cells in %.0f s: every tile identical to the oracle" % (
        totals["rounds"], totals["tiles"], totals["valid"], totals["cells"] / 1e9, time.time() - t_start))


if __name__ == "__main__":
    main()
