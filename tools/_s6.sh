cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -q -k "sse_variant or wrap or statuses or zoo" > gpurun_out/r02f_pytest.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r02f_pytest.log
timeout -s KILL 400 python bench.py --no-cpu-baseline > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f_bench.json'))
print('value',d['value'],'ms/step',d['ms_per_step'],'resident',d['device_resident']['ms_per_step'],'host',d['host_ms_per_step']['cvx_submit'],d['host_ms_per_step']['cvx_wait'],'roof',d['roofline']['frac'],d['roofline']['launch_ms'])
PY
