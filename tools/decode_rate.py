"""Dev tool (GPU box): decode_windows_kernel alone -- the recorded test_3 genome and windows repeated to ~200 000 windows
(913 M characters), cvx_genome_decode, the kernel's own time from HIP events on its stream (bench.py's `reference_decode`
without the rest of the bench).  Every window compared with what the unmodified reference decoded.

    python tools/decode_rate.py [repeats]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ngmlr_amd import capi  # noqa: E402
from ngmlr_amd.aligner import ConvexAlignHip, Genome  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    z = np.load(os.path.join(ROOT, "tests", "golden", "decode_test_3.npz"))
    wins = [(int(z["pos"][i]), int(z["len"][i]), z["bytes"][int(z["off"][i]):int(z["off"][i + 1])].tobytes()) for i in range(int(z["n"]))]
    al = ConvexAlignHip()
    g = Genome(al, z["binref"], int(z["nibbles"]), z["starts"])
    rep_n = max(1, 200000 // len(wins))
    pos = [w[0] for w in wins] * rep_n
    ln = [w[1] for w in wins] * rep_n
    chars = float(sum(ln))
    g.decode(pos[:len(wins)], ln[:len(wins)])
    out = []
    for _ in range(reps):
        c0 = time.perf_counter()
        got = g.decode(pos, ln)
        dt = time.perf_counter() - c0
        k_ms = al.stage_kernel_ms(capi.STAGE_DECODE)
        ok = sum(1 for k, o in enumerate(got) if o == wins[k % len(wins)][2])
        out.append({"windows": len(pos), "characters": int(chars), "kernel_ms": k_ms, "kernel_GB_per_s": chars * 1.5 / (k_ms * 1e-3) * 1e-9,
                    "kernel_frac_of_hbm_peak": chars * 1.5 / (k_ms * 1e-3) * 1e-9 / 8000.0, "whole_call_s": dt, "identical": "%d/%d" % (ok, len(pos))})
    g.free()
    print(json.dumps({"build": capi.load().cvx_build_id().decode(), "runs": out}))


if __name__ == "__main__":
    main()
