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
    print("%-70s %6s %12s %12s %12s %12s %5s %5s" % ("kernel", "calls", "total", "avg", "min", "max", "vgpr", "sgpr"))
    q = ("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
         "max(vgpr_count), max(sgpr_count) from kernels group by name order by sum(duration) desc")
    for r in c.execute(q):
        print("%-70s %6d %12.1f %12.1f %12.1f %12.1f %5d %5d" % ((r[0][:70],) + tuple(r[1:])))
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                              "group by kernel_name, counter_name order by sum(value) desc"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n## PMC counters (FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them)")
        print("%-70s %-14s %6s %16s %16s" % ("kernel", "counter", "disp", "sum", "avg/dispatch"))
        for r in rows:
            print("%-70s %-14s %6d %16.1f %16.1f" % (r[0][:70], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main(sys.argv[1])
