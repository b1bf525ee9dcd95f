#!/bin/bash
# round-2 GPU call 1: parity of the reworked fill (one tile per workgroup, two-phase tracking), issue-model microbenchmark, variant A/B
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
export TMPDIR=/tmp
( timeout -s KILL 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.txt 2>&1
( timeout -s KILL 120 tools/bin/ubench_mix ) > $O/ubench_mix.txt 2>&1
( bash tools/quick_variants.sh default bias pm1 biaspm1 ) > $O/quick.txt 2>&1
( CVX_TUNE_LATE_MIN=100000000 timeout -s KILL 200 python tools/ab_fill.py 12288 default ) > $O/ab_exact.txt 2>&1
( timeout -s KILL 400 python tools/ab_fill.py 12288 default w5 pm1 pm2 pm3 pm4 bias biaspm1 bias7 default ) > $O/ab.txt 2>&1
tail -3 $O/pytest.txt; cat $O/ubench_mix.txt $O/ab_exact.txt $O/ab.txt; grep -E "==|tiles|MISMATCH|exit" $O/quick.txt | head -40
