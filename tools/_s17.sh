cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout -s KILL 300 python bench.py --steps 20 --warmup 5 2>/dev/null > gpurun_out/r02k_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r02k_bench.json')); print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'resident', round(d['device_resident']['ms_per_step'],2), d['host_ms_per_step'], 'fill', round(d['roofline']['launch_ms'],2), d['roofline']['frac'], d['parity'], d['text_stage_device'], d['text_stage_host'], d['cpu_baseline'])"
