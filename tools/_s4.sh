cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/r02d_copy -o p -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --resident-steps 0 > $R/gpurun_out/r02d_copy.log 2>&1
ls $R/gpurun_out/r02d_copy
python - <<'PY'
import sqlite3, glob, os
db = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r02d_copy/*.db')[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'cop' in t.lower() or 'mem' in t.lower()])
for t in tabs:
    if t.lower() in ('memory_copies', 'memory_copy'):
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
        print(t, cols)
        for r in c.execute("select * from %s limit 3" % t): print(r)
try:
    for r in c.execute("select name, count(*), sum(size)/1e6, sum(duration)/1e6, min(size), max(size) from memory_copies group by name"):
        print(r)
except Exception as e:
    print('err', e)
PY
cd $R
for d in 2 4; do timeout -s KILL 200 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --resident-steps 0 --depth $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth', $d, 'value', round(d['value']), 'ms', round(d['ms_per_step'],2), d['host_ms_per_step']['cvx_submit'], d['host_ms_per_step']['cvx_wait'], 'fill', round(d['roofline']['launch_ms'],2))"; done
CVX_PACK_THREADS=48 timeout -s KILL 200 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --resident-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pack48 value', round(d['value']), 'ms', round(d['ms_per_step'],2), d['host_ms_per_step']['cvx_submit'], d['host_ms_per_step']['cvx_wait'], 'fill', round(d['roofline']['launch_ms'],2))"
