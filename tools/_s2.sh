cd $GRAFT_REPO_ROOT
timeout -s KILL 1200 python -m pytest tests -m gpu -q > gpurun_out/r02b_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r02b_pytest.log
timeout -s KILL 600 python tools/ab_fill.py 24576 default sched1 sched2 sched3 w7 > gpurun_out/r02b_ab.txt 2>&1
cat gpurun_out/r02b_ab.txt
