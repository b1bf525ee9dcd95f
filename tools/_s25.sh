cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python tools/cpu_baseline_scan.py 16 32 64 128 256 2>&1 | tail -6
