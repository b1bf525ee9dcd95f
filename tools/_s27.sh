cd $GRAFT_REPO_ROOT
nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread|Core" | head -6
timeout -s KILL 600 python tools/cpu_baseline_scan.py 1 8 16 32 64 128 256 2>&1 | tail -8
echo "--- with MALLOC_ARENA_MAX / THP off hints"
MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=268435456 timeout -s KILL 300 python tools/cpu_baseline_scan.py 64 256 2>&1 | tail -2
