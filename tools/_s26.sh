cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02o_bench.json 2> gpurun_out/r02o_bench.err; tail -3 gpurun_out/r02o_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02o_bench.json')); print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'resident', round(d['device_resident']['ms_per_step'],2), 'frac', d['roofline']['frac'], d['parity'], d['parity_detail'], d['cpu_baseline'])"
