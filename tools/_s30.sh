cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout -s KILL 600 python bench.py > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err; tail -2 gpurun_out/r02p_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02p_bench.json')); print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'resident', round(d['device_resident']['ms_per_step'],2), 'frac', d['roofline']['frac'], d['parity'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['subread_scoring']['parity'], d['text_stage_device']['equal_to_host_form'])"
python -c "import __graft_entry__ as g; g.smoke()"
