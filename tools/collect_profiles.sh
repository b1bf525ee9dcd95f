#!/bin/bash
# tools/collect_profiles.sh TAG -- on the GPU box: the official bench line plus the rocprofv3 passes the
# roofline numbers come from.  Writes everything under gpurun_out/TAG_*; summarise afterwards with
#   python tools/rocpd_summary.py gpurun_out/TAG_stats/*.db > profiles/TAG_stats.txt   (same for fetch/write/sq)
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o p -- python $R/bench.py --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o p -- python $R/bench.py --tiles 4096 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o p -- python $R/bench.py --tiles 4096 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_write.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT/${TAG}_sq -o p -- python $R/bench.py --tiles 12288 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_sq.log 2>&1
ls -la $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_sq 2>&1 | tail -12
