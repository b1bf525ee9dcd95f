cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02g_tl -o p -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --resident-steps 2 > $R/gpurun_out/r02g_tl.log 2>&1
python $R/tools/timeline.py $R/gpurun_out/r02g_tl/*.db 10 3
