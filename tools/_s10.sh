cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02h_cp -o p -- $R/tools/bin/copy_probe > $R/gpurun_out/r02h_probe.txt 2>&1
tail -5 $R/gpurun_out/r02h_probe.txt
python $R/tools/rocpd_summary.py $R/gpurun_out/r02h_cp/*.db | head -8
