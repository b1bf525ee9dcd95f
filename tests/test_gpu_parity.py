"""GPU (-m gpu): the HIP path, called through the C ABI (libcvxalign.so), against the CPU
oracle -- bit-exact on score bits, CIGAR, MD, offsets, clips, NM and the per-position
mismatch profile -- plus the golden tiles recorded from the reference binary, the edge
cases, and size-independent properties at BASELINE.json's full sizes."""
import re

import numpy as np
import pytest

from oracle.pyoracle import same_alignment
from tests import util

pytestmark = pytest.mark.gpu


def _check(hip_aligner, port_oracle, tiles, need_valid=True):
    got = hip_aligner.batch_align(tiles)
    bad = []
    n_valid = 0
    for t, g in zip(tiles, got):
        want = port_oracle.align(t)
        d = same_alignment(want, g)
        if d is None and port_oracle.kind == "port":     # (the reference does not expose its best cell)
            # the RAW fill result of EVERY tile, valid or not (a wrong fill score on tiles validPath rejects went
            # unnoticed for three rounds because only alignments were compared).  A tile without any positive score keeps the
            # reference's start value -1 and takes its first cell in the backtrack: nothing to compare there.
            f = port_oracle.last_fwd()
            fs = port_oracle.last_fill_score_bits()
            if fs != 0xBF800000 and g["status"] not in (4, 5):      # (too large / empty: never filled)
                if fs != g["fwd_score_bits"]:
                    d = "raw fill score %08x vs %08x" % (g["fwd_score_bits"], fs)
                elif (f["best_x"], f["best_y"]) != (g["best_x"], g["best_y"]):
                    d = "argmax cell"
        assert g["status"] != -1, "tile %s fell outside every device kernel" % t.tag
        n_valid += want["ret"] >= 0
        if d:
            bad.append((t.tag, t.H, t.W, int(t.row_length[0]), d))
    assert not bad, bad[:5]
    if need_valid:
        assert n_valid > 0
    return got


def test_native_library_is_the_one_running(hip_aligner):
    """The driver records loaded .so files; make the dependence explicit as well."""
    maps = open("/proc/self/maps").read()
    assert "libcvxalign.so" in maps and "libamdhip64" in maps


@pytest.mark.parametrize("name", ["ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz"])
def test_golden_tiles_from_reference_pipeline(hip_aligner, name):
    pairs = util.load_golden(name)
    got = hip_aligner.batch_align([t for t, _ in pairs])
    for (t, exp), g in zip(pairs, got):
        assert util.golden_diff(exp, g) is None, (t.tag, util.golden_diff(exp, g))


def test_every_recorded_test_3_call(hip_aligner):
    """All 985 SingleAlign calls of the reference on its test_3 reads (142 PacBio reads, the only
    multi-read long-read fixture it ships), not the committed 60-tile sample."""
    path = util.full_golden_path()
    if path is None:
        pytest.skip("oracle/_ref/golden_full not generated (tools/make_golden.sh needs /root/reference)")
    pairs = util.load_golden(path)
    assert len(pairs) == 985
    bad = []
    for lo in range(0, len(pairs), 256):
        chunk = pairs[lo:lo + 256]
        got = hip_aligner.batch_align([t for t, _ in chunk])
        for (t, exp), g in zip(chunk, got):
            d = util.golden_diff(exp, g)
            if d:
                bad.append((t.tag, d))
    assert not bad, bad[:5]


def test_zoo_all_corridor_kinds(hip_aligner, port_oracle):
    _check(hip_aligner, port_oracle, util.tile_zoo(seed=31, n=240, max_w=3000))


def test_edge_cases(hip_aligner, port_oracle):
    _check(hip_aligner, port_oracle, util.edge_tiles())


def test_ont_mix_with_retries(hip_aligner, port_oracle):
    from ngmlr_amd import synth
    _check(hip_aligner, port_oracle, synth.workload_ont(150, seed=77, max_len=8000))


def test_short_reads(hip_aligner, port_oracle):
    from ngmlr_amd import synth
    _check(hip_aligner, port_oracle, synth.workload_short(200, seed=5))


def _sv_tile(rng, flank, dels, inss, corridor):
    """Clean flanks around engineered long gaps: the path carries deletion / insertion runs far
    longer than a probe (64 cells) or a direction word (32 steps)."""
    from ngmlr_amd import synth
    ref_parts, qry_parts = [], []
    for k in range(max(len(dels), len(inss)) + 1):
        f = synth.random_ref(rng, flank)
        ref_parts.append(f)
        qry_parts.append(synth.mutate(rng, f, 0.03, (1, 1, 1)))
        if k < len(dels):
            ref_parts.append(synth.random_ref(rng, dels[k]))          # in the reference only: D run
        if k < len(inss):
            qry_parts.append(synth.random_ref(rng, inss[k]))          # in the read only: I run
    ref = np.concatenate(ref_parts)
    qry = np.concatenate(qry_parts)
    H, W = len(qry), len(ref)
    if corridor == "full":
        off, ln = synth.corridor_full(H, W)
    else:
        off, ln = synth.corridor_endpoints(H, W, synth.estimate_corridor(H, W, W) * 3, realign=True)
    return synth.Tile(ref=ref.tobytes(), qry=qry.tobytes(), row_offset=off, row_length=ln,
                      tag="sv d%s i%s %s" % (dels, inss, corridor))


def test_long_gap_runs(hip_aligner, port_oracle):
    """The backtrack resolves the gap that ends a diagonal run from the words of the same probe
    and hands over to plain gap probes where those end: runs of 1 ... 400, both kinds."""
    rng = np.random.default_rng(4242)
    tiles = []
    for g in (1, 2, 5, 17, 31, 32, 33, 63, 64, 65, 97, 130, 257, 400):
        tiles.append(_sv_tile(rng, 420, [g], [], "full"))
        tiles.append(_sv_tile(rng, 420, [], [g], "full"))
        tiles.append(_sv_tile(rng, 380, [g, 3], [2, g], "endpoints"))
    got = _check(hip_aligner, port_oracle, tiles)
    # the engineered gaps really are on the reported paths
    longest = max(int(m) for g in got if g["ret"] >= 0 for m in re.findall(r"(\d+)[ID]", g["cigar"]))
    assert longest >= 257


def _edge_hugging_tiles(rng, n, mult=1):
    """Tiles whose true alignment runs down the first (or last) column of the corridor rows: the anchors corridor shifted
    sideways by about its own half-width.  validPath rejects most of them -- and the raw fill result (best score, best cell)
    of exactly such tiles is where a wrong input at a row's first cell shows."""
    from ngmlr_amd import synth
    tiles = []
    for i in range(n):
        W = int(rng.integers(500, 2600))
        ref = synth.random_ref(rng, W)
        q = synth.mutate(rng, ref, float(rng.choice([0.02, 0.08, 0.15])), (6, 3, 1))
        off, ln = synth.corridor_anchors(len(q), W, mult=mult)      # mult 2 / 3: the retry loop's widened corridors (rings of 384 / 576 slots: gangs)
        w, right = int(ln[0]), -int(off[0])
        edge = i % 3
        shift = (right + int(rng.integers(-12, 6))) if edge == 0 else (right - w + int(rng.integers(-6, 12))) if edge == 1 else int(rng.integers(-40, 40))
        tiles.append(synth.Tile(ref=ref.tobytes(), qry=q.tobytes(), row_offset=(off + shift).astype(np.int32), row_length=ln, tag="edge%d shift %d" % (edge, shift)))
    return tiles


@pytest.mark.parametrize("env,mult", [({}, 1), ({"CVX_TUNE_MAX_M": "1"}, 1), ({"CVX_TUNE_MAX_M": "1", "CVX_TUNE_CHAIN_M": "2"}, 1), ({"CVX_TUNE_MAX_M": "2", "CVX_TUNE_CHAIN_M": "4"}, 1),
                                      ({"CVX_TUNE_GANGS": "1"}, 2), ({"CVX_TUNE_GANGS": "1"}, 3), ({"CVX_TUNE_GANGS": "1", "CVX_TUNE_GANG_PRIO": "1"}, 2), ({}, 2)])
def test_raw_fill_result_on_corridor_edge_paths(built, port_oracle, monkeypatch, env, mult):
    """cvx_result.score / best cell are the fill's curr_max and argmax whether or not validPath accepts the path.  On tiles whose
    best path hugs a corridor edge (almost all invalid, so the text-level parity checks never see their scores) the raw fill
    result must equal the oracle's forward fill bit for bit -- whole-tile rings, gangs of two and three waves on one ring
    (corridor multipliers 2 and 3) and chained row blocks of every height.
    Round 5: a chained block whose first cell fell on a multiple of 32 steps took 0 for that cell's diagonal input (6 of
    4 096 tiles of the C5 mix, 6 of 200 such tiles here)."""
    import ctypes as C
    from ngmlr_amd.aligner import ConvexAlignHip
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tiles = _edge_hugging_tiles(np.random.default_rng(99 + mult), 600 if mult == 1 else 240, mult)
    port_oracle.lib.oracle_port_last_fill_score.restype = C.c_float
    want = []
    for t in tiles:
        port_oracle.align(t, want_nm=False)
        f = port_oracle.last_fwd()
        want.append((int(np.float32(port_oracle.lib.oracle_port_last_fill_score()).view(np.uint32)), f["best_x"], f["best_y"]))
    al = ConvexAlignHip(device=0)
    b = al.upload(tiles)
    tm = b.run()
    res, _ = b.download()
    b.free()
    al.close()
    assert (tm.n_tiles_chained > 0) == (("CVX_TUNE_MAX_M" in env) or (mult > 1 and "CVX_TUNE_GANGS" not in env))
    bad = [(t.tag, t.H, got, w) for t, got, w in ((t, (int(np.float32(res[i].score).view(np.uint32)), res[i].best_ref_index, res[i].best_read_index), want[i])
                                                for i, t in enumerate(tiles)) if got != w and w[0] != 0xBF800000]
    assert not bad, bad[:5]
    assert sum(1 for i in range(len(tiles)) if res[i].status == 2) >= len(tiles) // 6      # the workload does what it is for


@pytest.mark.parametrize("table", ["1", "0"])
def test_penalty_table_and_long_gap_runs(built, port_oracle, monkeypatch, table):
    """The two-phase fill reads the convex gap penalty from an LDS table of 64 runs and clamps its run registers to 56 at every
    group end (cvx_kernels.hip, TAB instantiation): exact because the penalty is constant from run 27 on.  Whole-tile rings
    with engineered gap runs around the point where the penalty stops shrinking (27), around the clamp (56) and far beyond
    it, both kinds -- and every wide corridor carries runs of hundreds through its zero-score cells anyway; the same tiles
    with the table switched off (CVX_TUNE_PEN_TABLE=0): equal to the oracle either way, the long runs really on the paths."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import ConvexAlignHip
    monkeypatch.setenv("CVX_TUNE_PEN_TABLE", table)
    rng = np.random.default_rng(5150)
    tiles = []
    for g in (5, 26, 27, 28, 29, 40, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 64, 90, 120):
        for dels, inss in (([g], []), ([], [g]), ([g, 3], [2, g])):
            ref_parts, qry_parts = [], []
            for k in range(max(len(dels), len(inss)) + 1):
                f = synth.random_ref(rng, 500)
                ref_parts.append(f)
                qry_parts.append(synth.mutate(rng, f, 0.03, (1, 1, 1)))
                if k < len(dels):
                    ref_parts.append(synth.random_ref(rng, dels[k]))
                if k < len(inss):
                    qry_parts.append(synth.random_ref(rng, inss[k]))
            ref, qry = np.concatenate(ref_parts), np.concatenate(qry_parts)
            off, ln = synth.corridor_endpoints(len(qry), len(ref), 420, realign=True)      # 420 columns: a whole tile on a 256-slot ring
            tiles.append(synth.Tile(ref=ref.tobytes(), qry=qry.tobytes(), row_offset=off, row_length=ln, tag="pen d%s i%s" % (dels, inss)))
    tiles += util.tile_zoo(seed=33, n=24, max_w=2500)
    al = ConvexAlignHip(device=0)
    got = _check(al, port_oracle, tiles)
    al.close()
    longest = max(int(m) for g in got if g["ret"] >= 0 for m in re.findall(r"(\d+)[ID]", g["cigar"]))
    assert longest >= 120


@pytest.mark.parametrize("decay,gext,gem", [(0.0, -5.0, -1.0), (0.0, -2.0, -2.0), (0.01, -5.0, -1.0), (0.5, -3.0, -1.0), (0.07, -5.0, -1.0)])
def test_penalty_table_under_other_scorings(built, decay, gext, gem):
    """The table form is used only while the penalty is constant from the clamped run on (decided per handle from the scoring,
    cvx_runtime.cpp): no decay at all (constant), decay 0.5 (constant from run 4), 0.07 (from run 58: just beyond the clamp, so
    the arithmetic form), 0.01 (from run 400: the arithmetic form).  Each against the reference-semantics oracle with the same
    scoring; gap runs of 30-90 on the paths."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import ConvexAlignHip
    from oracle.pyoracle import Oracle
    rng = np.random.default_rng(808)
    tiles = util.tile_zoo(seed=71, n=18, max_w=1800)
    for g in (30, 57, 90):
        tiles.append(_sv_tile(rng, 300, [g], [], "endpoints"))
        tiles.append(_sv_tile(rng, 300, [], [g], "endpoints"))
    sc = dict(match=2.0, mismatch=-5.0, gap_open=-5.0, gap_extend=gext, gap_extend_min=gem, gap_decay=decay)
    orc = Oracle("port", (sc["match"], sc["mismatch"], sc["gap_open"], sc["gap_extend"], sc["gap_extend_min"], sc["gap_decay"]))
    al = ConvexAlignHip(device=0, **sc)
    _check(al, orc, tiles)
    al.close()


def test_every_ring_class(hip_aligner, port_oracle):
    """Corridor widths chosen to land in each fill kernel class (rings 64 ... 256) and, beyond
    those, in the chained row-block class (corridors with more than 256 live rows)."""
    from ngmlr_amd import synth
    rng = np.random.default_rng(1234)
    tiles = []
    for width in (40, 100, 200, 340, 369, 420, 560, 700, 900, 1500, 2048, 3500, 7000, 8192, 12000):
        W = max(1200, width + 300)
        t = synth.make_tile(rng, W, err=0.15, corridor="endpoints", width=width, realign=True, tag="w%d" % width)
        tiles.append(t)
    for W in (150, 400, 1100, 2400, 3900):
        tiles.append(synth.make_tile(rng, W, err=0.2, ratio=(4, 4, 2), corridor="full", tag="full%d" % W))
    hip = _check(hip_aligner, port_oracle, tiles, need_valid=True)
    batch = hip_aligner.upload(tiles)
    batch.run()
    rings = sorted({(li["slots_per_lane"], li["waves"]) for li in batch.launches()})
    batch.free()
    assert len(rings) >= 5 and any(nw > 3 for _, nw in rings), rings      # chained launches report their task count


def test_gangs_of_waves(built, port_oracle, monkeypatch):
    """CVX_TUNE_GANGS=1: corridors with 257-576 live rows as whole tiles on a ring of 384 / 576 slots shared by two / three
    waves of one workgroup, the lane boundary between the waves going through one LDS record per step (cvx_kernels.hip,
    GANG) instead of chained row blocks.  Selectable, not the default (measured slower on the ONT mix's short retries,
    profiles/r05_gang_ab.txt) -- and bit-exact: widths that land in both gang classes, two-phase and exact instantiations,
    with and without the penalty table, engineered gaps that cross the boundary between two waves."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import ConvexAlignHip
    rng = np.random.default_rng(4321)
    tiles = []
    for width in (530, 560, 620, 700, 760, 800, 900, 1000, 1100):
        for W in (1400, 3100):
            tiles.append(synth.make_tile(rng, W, err=0.15, corridor="endpoints", width=width, realign=True, tag="gang-w%d" % width))
    for g in (60, 191, 192, 193, 400):
        tiles.append(_sv_tile(rng, 700, [g], [g + 1], "endpoints"))
    tiles += _early_best_tiles(rng)[:6]
    for env in ({"CVX_TUNE_GANGS": "1"}, {"CVX_TUNE_GANGS": "1", "CVX_TUNE_PEN_TABLE": "0"}, {"CVX_TUNE_GANGS": "1", "CVX_TUNE_LATE_MIN": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        al = ConvexAlignHip(device=0)
        _check(al, port_oracle, tiles, need_valid=False)
        batch = al.upload(tiles)
        batch.run()
        rings = sorted({(li["slots_per_lane"], li["waves"]) for li in batch.launches()})
        batch.free()
        al.close()
        assert (3, 2) in rings and (3, 3) in rings, rings                 # rings of 384 and 576 slots
        for k in env:
            monkeypatch.delenv(k)


def test_chained_row_blocks(built, port_oracle, monkeypatch):
    """Wide corridors (more live rows than any ring) are cut into row blocks that run as a
    dependency chain through boundary streams: every block height class, block counts that do and
    do not divide H, several wide tiles in one batch (their tasks interleave), full-matrix tiles,
    engineered long gaps across block boundaries, and the int16-run instantiations."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import ConvexAlignHip
    rng = np.random.default_rng(99)
    tiles = []
    for width, W in ((1100, 1500), (1400, 1331), (2100, 2600), (3000, 2048), (4200, 4700), (4200, 2305), (8192, 5000), (9000, 4608)):
        tiles.append(synth.make_tile(rng, W, err=0.18, ratio=(4, 4, 2), corridor="endpoints", width=width, realign=True, tag="chain-w%d" % width))
    for W in (700, 1280, 2100):
        tiles.append(synth.make_tile(rng, W, err=0.2, ratio=(4, 4, 2), corridor="full", tag="chain-full%d" % W))
    tiles.append(_sv_tile(rng, 900, [64, 130], [65, 200], "full"))        # long gaps crossing 64-row block boundaries
    tiles.append(_sv_tile(rng, 1300, [300], [257], "full"))
    for wrap, chain_m in ((False, 0), (False, 2), (False, 4), (True, 1), (True, 4)):
        monkeypatch.setenv("CVX_TUNE_FORCE_WRAP16", "1" if wrap else "0")
        monkeypatch.setenv("CVX_TUNE_CHAIN_M", str(chain_m))
        al = ConvexAlignHip(device=0)
        _check(al, port_oracle, tiles, need_valid=False)
        batch = al.upload(tiles)
        tm = batch.run()
        assert tm.n_tiles_chained >= 12, tm.n_tiles_chained
        assert {li["slots_per_lane"] for li in batch.launches() if li["waves"] > 3} == {chain_m or 1}      # (a chained launch reports its row-block tasks)
        batch.free()
        al.close()


def test_int16_run_kernels(built, port_oracle, monkeypatch):
    """The WRAP instantiations (indelRun as the reference's short, taken by tiles with H or a
    row > 32767) must equal the float-run kernels and the oracle wherever no run wraps."""
    from ngmlr_amd.aligner import ConvexAlignHip
    monkeypatch.setenv("CVX_TUNE_FORCE_WRAP16", "1")
    al = ConvexAlignHip(device=0)
    tiles = util.tile_zoo(seed=41, n=90, max_w=2500) + util.edge_tiles()
    _check(al, port_oracle, tiles)
    batch = al.upload(tiles[:20])
    batch.run()
    assert all(li["wrap16"] == 1 for li in batch.launches())
    batch.free()
    al.close()


def test_genuine_int16_wrap(hip_aligner, port_oracle):
    """A gap run that really passes SHRT_MAX inside the best alignment (tests/util.py wrap16_tile;
    the CPU suite pins the port to the reference's own output on it): the int16-run kernels must
    break the insertion exactly where the reference's `short indelRun` wraps."""
    t = util.wrap16_tile()
    got = _check(hip_aligner, port_oracle, [t])
    assert "32767I1M" in got[0]["cigar"]
    batch = hip_aligner.upload([t])
    batch.run()
    assert all(li["wrap16"] == 1 for li in batch.launches())
    batch.free()


EXOTIC_SCORING = [
    dict(match=2.0, mismatch=-10.0, gap_open=-5.0, gap_extend=-5.0, gap_extend_min=-1.0, gap_decay=0.15),   # 200/200 tiles differ
    dict(match=2.0, mismatch=-6.0, gap_open=-5.0, gap_extend=-5.0, gap_extend_min=-1.0, gap_decay=0.15),    # equality: open + floor == mismatch
    dict(match=2.0, mismatch=-7.0, gap_open=-4.0, gap_extend=-3.0, gap_extend_min=-2.0, gap_decay=0.5),
    dict(match=1.0, mismatch=-4.0, gap_open=-1.0, gap_extend=-1.0, gap_extend_min=-0.5, gap_decay=0.05),
    dict(match=3.0, mismatch=-20.0, gap_open=-2.0, gap_extend=-6.0, gap_extend_min=-1.0, gap_decay=0.3),
]


@pytest.mark.parametrize("k", range(len(EXOTIC_SCORING)))
def test_sse_variant_scoring(built, k):
    """Scoring for which the reference's SSE path and the scalar recurrence disagree
    (gap_open + gap_ext_min >= mismatch): the product must reproduce the SSE path -- relaxed
    extension tests, scalar recomputation of each row's last 12 cells, both feeding the running
    maximum -- on every corridor kind, against the reference itself when oracle/_ref is there
    (the port is pinned to it for this regime by tests/test_oracle_cpu.py)."""
    from ngmlr_amd.aligner import ConvexAlignHip
    from oracle.pyoracle import Oracle, have_ref
    sc = EXOTIC_SCORING[k]
    params = (sc["match"], sc["mismatch"], sc["gap_open"], sc["gap_extend"], sc["gap_extend_min"], sc["gap_decay"])
    orc = Oracle("reference" if have_ref() else "port", params)
    al = ConvexAlignHip(device=0, **sc)
    tiles = util.tile_zoo(seed=50 + k, n=60, max_w=1800) + util.edge_tiles()
    _check(al, orc, tiles)
    if k == 0:
        # and the regime really is different: the scalar recurrence gives other answers here
        spec = Oracle("port", params)
        spec.set_spec_fill(True)
        got = al.batch_align(tiles[:40])
        assert sum(same_alignment(spec.align(t), g) is not None for t, g in zip(tiles[:40], got)) > 0
    al.close()


def test_sse_variant_kernel_equals_ring_kernels_under_default_scoring(built, port_oracle, monkeypatch):
    """Inside the default regime the SSE path IS the scalar recurrence: forcing the SSE-variant
    kernel must change nothing."""
    from ngmlr_amd.aligner import ConvexAlignHip
    monkeypatch.setenv("CVX_TUNE_SSE_VARIANT", "1")
    al = ConvexAlignHip(device=0)
    _check(al, port_oracle, util.tile_zoo(seed=77, n=60, max_w=1500) + util.edge_tiles())
    al.close()


def _early_best_tiles(rng, n=24):
    """Reads whose alignment ends long before the read does (a clean prefix, then junk): the best cell lies in
    the first 30-60 % of the anti-diagonals, outside the exactly tracked tail of the two-phase fill."""
    from ngmlr_amd import synth
    tiles = []
    for k in range(n):
        W = int(rng.integers(2500, 7000))
        ref = synth.random_ref(rng, W)
        good = int(W * float(rng.uniform(0.3, 0.6)))
        qry = np.concatenate([synth.mutate(rng, ref[:good], 0.1), synth.random_ref(rng, W - good)])
        off, ln = synth.corridor_anchors(len(qry), W)
        tiles.append(synth.Tile(ref.tobytes(), qry.tobytes(), off, ln, tag="early-best%d" % k))
    return tiles


@pytest.mark.parametrize("late_min", [None, "1", "1000000"])
def test_two_phase_tracking_redo_path(built, port_oracle, monkeypatch, late_min):
    """The two-phase fill only tracks the best cell exactly in the last groups of a tile and redoes a tile
    with the exact instantiation when an earlier score is at least as large (cvx_kernels.hip).  Tiles whose
    best cell is early MUST take that second pass and still equal the oracle (argmax cell included); a
    one-group tail (CVX_TUNE_LATE_MIN=1) sends almost everything through it, a huge one nothing."""
    from ngmlr_amd.aligner import ConvexAlignHip
    if late_min is not None:
        monkeypatch.setenv("CVX_TUNE_LATE_MIN", late_min)
    rng = np.random.default_rng(77)
    tiles = _early_best_tiles(rng) + util.tile_zoo(seed=12, n=36, max_w=3000)
    al = ConvexAlignHip(device=0)
    _check(al, port_oracle, tiles, need_valid=False)
    batch = al.upload(tiles)
    tm = batch.run()
    batch.free()
    al.close()
    if late_min == "1000000":
        assert tm.n_tiles_redone == 0
    else:
        assert tm.n_tiles_redone >= 12, tm.n_tiles_redone      # at least the engineered ones (some end in row 0: invalid, but still redone)


@pytest.mark.parametrize("env", [{"CVX_TUNE_BT_GROUP": "4"}, {"CVX_TUNE_BT_GROUP": "8"}, {"CVX_TUNE_BT_GROUP": "16"}, {"CVX_TUNE_BT_GROUP": "32"},
                                 {"CVX_TUNE_BT_GROUP": "64"}, {"CVX_TUNE_BT_GROUP": "-1"}, {"CVX_TUNE_OVERLAP_POST": "1"}, {"CVX_TUNE_BT_PER_CLASS": "0"}])
def test_runtime_knobs_do_not_change_results(built, port_oracle, monkeypatch, env):
    """Every lanes-per-tile setting of the backtrack (the default picks 8 / 16 / 32 / 64 by the number of tiles walked together; -1: round 4's rule), the
    post-fill overlap, and one walk behind all fills instead of one per fill class (the default for a batch of several
    classes, as this one is): same alignments.  A batch of > 4096 tiles so that the grouped kernels really run,
    chained and whole tiles mixed, a few reads much longer than the rest."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import ConvexAlignHip
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(404)
    tiles = util.tile_zoo(seed=91, n=48, max_w=1500) + util.edge_tiles()
    tiles.append(synth.make_tile(rng, 9000, err=0.2, ratio=(4, 4, 2), corridor="anchors", tag="long"))
    tiles.append(synth.make_tile(rng, 2500, err=0.18, ratio=(4, 4, 2), corridor="endpoints", width=1400, realign=True, tag="chained"))
    want = [port_oracle.align(t) for t in tiles]
    filler = [synth.make_tile(rng, int(rng.integers(40, 120)), err=0.1, corridor="linear", ref_pad=60, tag="filler") for _ in range(64)]
    batch = tiles + filler * 66                        # 4224 + len(tiles) tiles
    al = ConvexAlignHip(device=0)
    got = al.batch_align(batch, want_nm=False)
    al.close()
    bad = [(t.tag, same_alignment(w, g, keys=("ret", "score_bits", "position_offset", "qstart", "qend", "nm", "cigar", "md")))
           for t, w, g in zip(tiles, want, got)]
    bad = [b for b in bad if b[1] is not None and b[1] != "nm_per_position"]
    assert not bad, bad[:5]
    fill_want = [port_oracle.align(t) for t in filler[:8]]
    for t, w, g in zip(filler[:8], fill_want, got[len(tiles):len(tiles) + 8]):
        assert (w["ret"], w["cigar"]) == (g["ret"], g["cigar"]) or (w["ret"] < 0 and g["ret"] < 0), t.tag


def test_irregular_corridors_take_the_catch_all_kernel(hip_aligner, port_oracle):
    """CorridorLine[] shapes no reference caller builds (row starts that do not increase):
    computed on the device by the catch-all kernel, never on the CPU, still bit-exact."""
    from ngmlr_amd import synth
    rng = np.random.default_rng(77)
    tiles = []
    for k in range(8):
        W = int(rng.integers(150, 900))
        ref = synth.random_ref(rng, W)
        qry = synth.mutate(rng, ref, 0.12)
        H = len(qry)
        base, ln = synth.corridor_anchors(H, W)
        if k % 4 == 0:
            off = base + rng.integers(-40, 40, size=H).astype(np.int32)          # jitter
        elif k % 4 == 1:
            off = (W - np.arange(H)).astype(np.int32) - 150                       # anti-diagonal band
        elif k % 4 == 2:
            off = np.where(np.arange(H) % 50 < 25, base, base - 60).astype(np.int32)   # zigzag
        else:
            off = np.full(H, -10, dtype=np.int32)                                 # constant start (gs ties not allowed? still increasing)
            off[::7] -= 30
        ln = (ln + rng.integers(0, 30, size=H)).astype(np.int32)
        tiles.append(synth.Tile(ref.tobytes(), qry.tobytes(), off.astype(np.int32), ln, tag="irregular%d" % k))
    _check(hip_aligner, port_oracle, tiles, need_valid=False)


def test_full_size_pacbio_tiles_bit_exact(hip_aligner, port_oracle):
    """configs[1] shape: 10 kb reads, width 309-369 (a handful: the oracle needs ~50 ms each)."""
    from ngmlr_amd import synth
    _check(hip_aligner, port_oracle, synth.workload_pacbio(12, seed=2024, read_len=10000))


def _cigar_consumes(cigar):
    q = r = 0
    for n, op in re.findall(r"(\d+)([MIDS])", cigar):
        n = int(n)
        if op in "MIS":
            q += n
        if op in "MD":
            r += n
    return q, r


def _rescore(tile, g, sp=(2.0, -5.0, -5.0, -5.0, -1.0, 0.15)):
    """Score of the reported path under the convex gap model, accumulated in float32 in path
    order exactly as the DP does (so it must equal Align::Score bit for bit)."""
    mat, mis, go, ge, gem, dec = (np.float32(v) for v in sp)
    s = np.float32(0)
    x = g["position_offset"]
    y = g["qstart"] - tile.ext_qstart
    md_ops = []
    for n, op in re.findall(r"(\d+)([MIDS])", g["cigar"]):
        n = int(n)
        if op == "S":
            continue
        if op == "M":
            for _ in range(n):
                s = np.float32(s + (mat if tile.ref[x] == tile.qry[y] else mis))
                x += 1; y += 1
        else:
            for k in range(n):
                pen = go if k == 0 else np.float32(min(gem, np.float32(ge + np.float32(np.float32(k) * dec))))
                s = np.float32(s + pen)
                if op == "I":
                    y += 1
                else:
                    x += 1
    return s


def test_properties_at_full_size(hip_aligner):
    """No oracle here (too slow at this count): size-independent properties on 10 kb and
    20 kb tiles.  CIGAR consumes exactly the read; reference span stays in the window;
    re-scoring the reported path reproduces Score bit for bit; NM/MD agree with the path;
    results do not depend on batch order or batch composition (idempotence)."""
    from ngmlr_amd import synth
    tiles = synth.workload_pacbio(96, seed=11, read_len=10000) + synth.workload_ont(64, seed=12, max_len=20000)
    got = hip_aligner.batch_align(tiles, want_nm=False)
    n_valid = 0
    for t, g in zip(tiles, got):
        assert g["status"] in (0, 1, 2, 3)
        if g["ret"] < 0:
            continue
        n_valid += 1
        q, r = _cigar_consumes(g["cigar"])
        assert q == t.H == g["ret"]
        assert 0 <= g["position_offset"] and g["position_offset"] + r <= t.W
        assert g["last_ref"] == r and g["last_read"] - g["first_read"] == q - g["qstart"] - g["qend"]
        assert np.float32(_rescore(t, g)).view(np.uint32) == g["score_bits"], t.tag
        md_mism = len(re.findall(r"[A-Zx]", re.sub(r"\^[A-Zx]+", "", g["md"])))
        ins = sum(int(n) for n, op in re.findall(r"(\d+)([ID])", g["cigar"]))
        assert g["nm"] == md_mism + ins
    assert n_valid >= 0.9 * len(tiles)
    perm = np.random.default_rng(0).permutation(len(tiles))
    again = hip_aligner.batch_align([tiles[i] for i in perm], want_nm=False)
    for k, i in enumerate(perm):
        assert same_alignment(got[i], again[k], keys=("ret", "score_bits", "cigar", "md", "position_offset")) is None
    solo = hip_aligner.single_align(tiles[5], want_nm=False)
    assert same_alignment(got[5], solo, keys=("ret", "score_bits", "cigar", "md")) is None


def test_c5_full_shapes_vs_reference(hip_aligner, ref_oracle):
    """configs[4] at full size against the reference's own ConvexAlignFast (oracle/_ref): 100 kb reads
    at corridor widths 309 (anchors), 2048 and 8192 (the retry loop's cap, src/AlignmentBuffer.cpp:
    1454-1467) -- the last two run as chained row blocks -- and a 5 kb full-matrix inversion tile."""
    from ngmlr_amd import synth
    rng = np.random.default_rng(2025)
    tiles = [synth.make_tile(rng, 100000, err=0.2, ratio=(4, 4, 2), corridor="anchors", tag="ul-309"),
             synth.make_tile(rng, 100000, err=0.2, ratio=(4, 4, 2), corridor="endpoints", width=2048, realign=True, tag="ul-2048"),
             synth.make_tile(rng, 100000, err=0.2, ratio=(4, 4, 2), corridor="endpoints", width=8192, realign=True, tag="ul-8192"),
             synth.make_tile(rng, 4900, err=0.2, ratio=(4, 4, 2), corridor="full", tag="sv-full")]
    got = _check(hip_aligner, ref_oracle, tiles, need_valid=False)
    assert sum(1 for g in got if g["ret"] >= 0) >= 3


def test_ont_20kb_sample_vs_oracle(hip_aligner, port_oracle):
    """configs[2] at its full read length: a sample of 20 kb ONT-like tiles (25 % error, wider
    corridors, some at retry multiplier 2) against the oracle, not only through properties."""
    from ngmlr_amd import synth
    rng = np.random.default_rng(31)
    tiles = []
    for k in range(10):
        W = int(rng.integers(17000, 20001))
        tiles.append(synth.make_tile(rng, W, err=0.25, ratio=(4, 4, 2), corridor="anchors", scatter=60.0,
                                     mult=2 if k % 4 == 0 else 1, tag="ont20k"))
    _check(hip_aligner, port_oracle, tiles)


def test_ultralong_tile(hip_aligner, port_oracle):
    """configs[4] shape: one 100 kb tile (direction matrix > 30 MB in the reference) and a
    wide-corridor tile through the chained row-block kernel."""
    from ngmlr_amd import synth
    rng = np.random.default_rng(4)
    tiles = [synth.make_tile(rng, 100000, err=0.2, ratio=(4, 4, 2), corridor="anchors", tag="ul100k"),
             synth.make_tile(rng, 30000, err=0.2, ratio=(4, 4, 2), corridor="endpoints", width=2048, realign=True, tag="ul-w2048")]
    _check(hip_aligner, port_oracle, tiles)
