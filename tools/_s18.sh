cd $GRAFT_REPO_ROOT
for m in 1 2 4; do echo "== chain M=$m"; CVX_TUNE_CHAIN_M=$m timeout -s KILL 300 python tools/config_rates.py ont 2>&1 | tail -4; done
