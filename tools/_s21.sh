cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout -s KILL 400 python tools/config_rates.py 2>&1 | grep -v "^    class"
timeout -s KILL 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'resident', round(d['device_resident']['ms_per_step'],2), d['stage_ms_per_step'])"
