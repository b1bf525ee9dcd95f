"""Dev tool: device throughput of the other BASELINE configs (ONT mix C3, ultra-long + SV C5, short
reads) with inputs resident in HBM -- per fill launch class: tiles, ms, G cells/s.  Not a bench line."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngmlr_amd import synth
from ngmlr_amd.aligner import ConvexAlignHip

al = ConvexAlignHip()
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, make in (("ont C3 (24k tiles)", lambda: synth.workload_ont(24000, seed=11)),
                   ("ont C3 (60k tiles)", lambda: synth.workload_ont(60000, seed=11)),
                   ("ultralong+SV C5 (96 tiles)", lambda: synth.workload_ultralong_sv(96, seed=13)),
                   ("ultralong mix C5 (2048 tiles)", lambda: synth.workload_ultralong_mix(2048, seed=19)),
                   ("ultralong mix C5 (6144 tiles)", lambda: synth.workload_ultralong_mix(6144, seed=19)),
                   ("short reads (100k tiles)", lambda: synth.workload_short(100000, seed=17))):
    if only not in name:
        continue
    tiles = make()
    bases = sum(t.H for t in tiles)
    b = al.upload(tiles)
    b.run()
    best = None
    for _ in range(2):
        tm = b.run()
        if best is None or tm.total_ms < best.total_ms: best = tm
    print("%s: %.1f Mbp, plan %.2f fill %.2f bt %.2f total %.2f ms -> %.0f Gbp/h, %.0f G cells/s overall" % (
        name, bases / 1e6, best.plan_ms, best.fill_ms, best.backtrack_ms, best.total_ms,
        bases / best.total_ms * 3.6e-3, best.cells / best.total_ms * 1e-6))
    for li in b.launches():
        print("    class M=%d NW=%d wrap=%d: %6d tiles %8.2f ms %7.0f G cells/s" % (
            li["slots_per_lane"], li["waves"], li["wrap16"], li["n_tiles"], li["ms"], li["cells"] / li["ms"] * 1e-6))
    b.free()
al.close()
