cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python tools/ab_fill.py 24576 default med3 2>&1 | tail -3
