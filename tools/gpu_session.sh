#!/bin/bash
# one gpurun call: GPU parity suite, the bench line + profiles, the other configs' rates
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02}
mkdir -p $R/gpurun_out
cd $R
timeout -s KILL 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/${TAG}_pytest.log
bash tools/collect_profiles.sh $TAG
cd $R
timeout -s KILL 400 python tools/config_rates.py > gpurun_out/${TAG}_config_rates.txt 2>&1
cat gpurun_out/${TAG}_config_rates.txt
