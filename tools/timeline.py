#!/usr/bin/env python3
"""tools/timeline.py DB [FIRST_FILL [N_FILLS]] -- kernel timeline (start / end / duration / queue) of a rocprofv3
--kernel-trace database around the FIRST_FILL-th dominant fill launch: what overlaps what on the device."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = list(c.execute("select name, start, end, queue_id, grid_x from kernels order by start"))
t0 = rows[0][1]
fills = [r for r in rows if "fill_ring_kernel<3, false, 0>" in r[0]]
print("dominant fill starts (ms):", [round((r[1] - t0) / 1e6) for r in fills])
lo, hi = fills[first][1] - 5e6, fills[min(first + nf, len(fills) - 1)][1]
for r in rows:
    if lo <= r[1] <= hi:
        print("%-44s start %9.3f end %9.3f dur %8.3f q %s grid %s" % (r[0][:44], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], r[4]))
