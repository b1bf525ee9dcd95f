"""tools/rocpd_summary.py -- summarise a rocprofv3 rocpd database (.db) as text:
per-kernel call count / total / average / min / max duration, register use, and any PMC
counters collected (summed over a kernel's dispatches, plus per-dispatch average).

    python tools/rocpd_summary.py gpurun_out/prof_stats/r01_results.db > profiles/...txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print("# rocprofv3 summary of %s" % path.split("/")[-2:])
    print("## kernel-trace stats (durations in us)")
    print("%-70s %6s %12s %12s %12s %12s %12s %5s %5s" % ("kernel", "calls", "total", "avg", "median", "min", "max", "vgpr", "sgpr"))
    q = ("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
         "max(vgpr_count), max(sgpr_count) from kernels group by name order by sum(duration) desc")
    for r in c.execute(q):
        d = sorted(x[0] / 1e3 for x in c.execute("select duration from kernels where name = ?", (r[0],)))
        med = d[len(d) // 2] if len(d) % 2 else 0.5 * (d[len(d) // 2 - 1] + d[len(d) // 2])
        print("%-70s %6d %12.1f %12.1f %12.1f %12.1f %12.1f %5d %5d" % ((r[0][:70],) + tuple(r[1:4]) + (med,) + tuple(r[4:])))
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), max(value) from counters_collection "
                              "group by kernel_name, counter_name order by sum(value) desc"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n## PMC counters (FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; `largest` = the")
        print("## biggest dispatch of that kernel, i.e. the bench batch rather than the small text-stage batch)")
        print("%-70s %-20s %6s %16s %16s %16s" % ("kernel", "counter", "disp", "sum", "avg/dispatch", "largest"))
        for r in rows:
            print("%-70s %-20s %6d %16.1f %16.1f %16.1f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main(sys.argv[1])
