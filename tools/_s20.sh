cd $GRAFT_REPO_ROOT
for g in 8 16 32; do CVX_TUNE_BT_GROUP=$g timeout -s KILL 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --resident-steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bt_group', $g, 'resident', round(d['device_resident']['ms_per_step'],2), 'bt', round(d['stage_ms_per_step']['backtrack'],2), d['valid_alignments'])"; done
for g in 8 32; do echo "== group $g"; CVX_TUNE_BT_GROUP=$g timeout -s KILL 300 python tools/config_rates.py ont 2>&1 | grep -v "^    class"; done
CVX_TUNE_BT_GROUP=8 timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
