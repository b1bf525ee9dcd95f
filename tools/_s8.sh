cd $GRAFT_REPO_ROOT
run() { timeout -s KILL 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --resident-steps 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'hwq', '$GPU_MAX_HW_QUEUES', 'value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'resident', round(d['device_resident']['ms_per_step'],2), 'submit', round(d['host_ms_per_step']['cvx_submit'],1), 'wait', round(d['host_ms_per_step']['cvx_wait'],1), 'fill', round(d['roofline']['launch_ms'],2))"; }
run
run --depth 2
run --depth 4
GPU_MAX_HW_QUEUES=8 run
GPU_MAX_HW_QUEUES=2 run
