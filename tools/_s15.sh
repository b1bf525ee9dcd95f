cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_score.py -q -x 2>&1 | tail -4
timeout -s KILL 300 python tools/score_rates.py 65536 2>&1 | tail -5
timeout -s KILL 600 python tools/e2e_rates.py 1 16 64 2>&1 | tail -10
timeout -s KILL 400 python tools/config_rates.py ont 2>&1 | tail -5
