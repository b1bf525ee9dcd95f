cd $GRAFT_REPO_ROOT
timeout -s KILL 1200 python -m pytest tests -m gpu -q > gpurun_out/r02c_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r02c_pytest.log
timeout -s KILL 400 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c_bench.json'))
print('value',d['value'],'ms/step',d['ms_per_step'],'resident',d['device_resident'],'host',d['host_ms_per_step'],'roof',d['roofline']['frac'],d['roofline']['launch_ms'],d['parity'])
PY
