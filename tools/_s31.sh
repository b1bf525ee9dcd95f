cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -q -k "runtime_knobs" 2>&1 | tail -6
