cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 $R/tools/bin/copy_probe > $R/gpurun_out/r02e_copy_probe.txt 2>&1
cat $R/gpurun_out/r02e_copy_probe.txt
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02e_cp -o p -- $R/tools/bin/copy_probe > /dev/null 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/r02e_cp/*.db | head -12
