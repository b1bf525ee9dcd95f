"""scratch: chained row blocks vs whole-tile rings vs the oracle's fill score, on tiles whose alignment hugs a corridor edge"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ngmlr_amd import synth
from ngmlr_amd.aligner import ConvexAlignHip
from oracle.pyoracle import Oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
tiles = []
for i in range(600):
    W = int(rng.integers(500, 2600))
    ref = synth.random_ref(rng, W)
    q = synth.mutate(rng, ref, float(rng.choice([0.02, 0.08, 0.15])), (6, 3, 1))
    H = len(q)
    off, ln = synth.corridor_anchors(H, W)
    w = int(ln[0])
    right = -int(off[0])
    edge = i % 3
    if edge == 0:   shift = right + int(rng.integers(-12, 6))            # the main diagonal at (or just outside) the rows' first column
    elif edge == 1: shift = right - w + int(rng.integers(-6, 12))        # ... at the rows' last column
    else:           shift = int(rng.integers(-40, 40))
    off = (off + shift).astype(np.int32)
    tiles.append(synth.Tile(ref=ref.tobytes(), qry=q.tobytes(), row_offset=off, row_length=ln, tag="e%d s%d w%d" % (edge, shift, w)))
orc = Oracle("port")
orc.lib.oracle_port_last_fill_score.restype = C.c_float
want = []
for t in tiles:
    orc.align(t, want_nm=False)
    f = orc.last_fwd()
    want.append((int(np.float32(orc.lib.oracle_port_last_fill_score()).view(np.uint32)), f["best_x"], f["best_y"]))
for env in ({}, {"CVX_TUNE_MAX_M": "1"}, {"CVX_TUNE_MAX_M": "1", "CVX_TUNE_CHAIN_M": "2"}, {"CVX_TUNE_MAX_M": "1", "CVX_TUNE_CHAIN_M": "4"}, {"CVX_TUNE_PEN_TABLE": "0"}):
    for k, v in env.items(): os.environ[k] = v
    al = ConvexAlignHip()
    for k in env: os.environ.pop(k)
    b = al.upload(tiles)
    tm = b.run()
    res, ops = b.download()
    bad = []
    for i, t in enumerate(tiles):
        got = (int(np.float32(res[i].score).view(np.uint32)), res[i].best_ref_index, res[i].best_read_index)
        if got != want[i] and not (want[i][0] == 3212836864):
            bad.append((i, t.tag, t.H, len(t.ref), res[i].status, got, want[i], "row in block %d" % (want[i][2] % 64), "x - off = %d" % (want[i][1] - int(t.row_offset[want[i][2]]))))
    print(env, "chained", tm.n_tiles_chained, "status", np.bincount([max(res[i].status, 0) for i in range(len(tiles))]).tolist(), "mismatching fill results:", len(bad))
    for x in bad[:12]: print("   ", x)
    b.free(); al.close()
