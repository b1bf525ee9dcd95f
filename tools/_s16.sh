cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_text.py -q -x 2>&1 | tail -12
