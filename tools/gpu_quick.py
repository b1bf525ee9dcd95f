"""Quick GPU shake-out: HIP path vs the CPU oracle on seeded tiles (dev tool, not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ngmlr_amd import synth
from ngmlr_amd.aligner import ConvexAlignHip
from oracle.pyoracle import Oracle, same_alignment

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(3)
    tiles = []
    for i in range(n):
        kind = ['anchors', 'endpoints', 'linear', 'full', 'anchors'][i % 5]
        W = int(rng.integers(30, 4000))
        if kind == 'full': W = min(W, 900)
        tiles.append(synth.make_tile(rng, W, err=float(rng.choice([0.05, 0.15, 0.3])), corridor=kind,
                                     scatter=30.0, mult=int(rng.integers(1, 3)), n_frac=0.005))
    tiles += synth.workload_pacbio(4, seed=1)
    al = ConvexAlignHip()
    t0 = time.time()
    batch = al.upload(tiles)
    tm = batch.run()
    print("timing: plan %.3f fill %.3f bt %.3f total %.3f ms; cells %d active %d launches %d fast %d" % (
        tm.plan_ms, tm.fill_ms, tm.backtrack_ms, tm.total_ms, tm.cells, tm.active_cells, tm.n_fill_launches, tm.n_tiles_fast))
    got = batch.alignments()
    batch.free()
    print("gpu wall %.2fs" % (time.time() - t0))
    orc = Oracle('port')
    bad = 0; inv = 0; unsup = 0
    for i, t in enumerate(tiles):
        a = orc.align(t)
        g = got[i]
        f = orc.last_fwd()
        if g['status'] == -1: unsup += 1
        d = same_alignment(a, g)
        if a['ret'] < 0: inv += 1
        if d is None and a['ret'] >= 0 and (f['best_x'] != g['best_x'] or f['best_y'] != g['best_y']):
            d = 'best cell'
        if d:
            bad += 1
            if bad < 15:
                print("MISMATCH tile %d (%s H=%d W=%d w=%d): %s | status %d score %r vs %r best (%d,%d) vs (%d,%d)" % (
                    i, t.tag, t.H, t.W, int(t.row_length[0]), d, g['status'], g['score'], a['score'],
                    g['best_x'], g['best_y'], f['best_x'], f['best_y']))
    print("tiles %d mismatches %d oracle-invalid %d unsupported %d" % (len(tiles), bad, inv, unsup))
    return 1 if bad else 0

if __name__ == '__main__':
    sys.exit(main())
