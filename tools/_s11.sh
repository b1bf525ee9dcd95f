cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02i_tl -o p -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --resident-steps 2 > $R/gpurun_out/r02i_tl.log 2>&1
tail -c 400 $R/gpurun_out/r02i_tl.log
python $R/tools/timeline.py $R/gpurun_out/r02i_tl/*.db 10 2 | grep -v "kernel<4\|<3, false, 1>" | head -40
cd $R
timeout -s KILL 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct: value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'resident', round(d['device_resident']['ms_per_step'],2), 'submit', round(d['host_ms_per_step']['cvx_submit'],1), 'wait', round(d['host_ms_per_step']['cvx_wait'],1), 'fill', round(d['roofline']['launch_ms'],2), d['roofline']['frac'])"
timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun: value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'resident', round(d['device_resident']['ms_per_step'],2), 'fill', round(d['roofline']['launch_ms'],2), d['config']['launch'])"
timeout -s KILL 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
