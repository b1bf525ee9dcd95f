#!/bin/bash
mkdir -p gpurun_out/r02c
O=gpurun_out/r02c
( timeout -s KILL 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.txt 2>&1
( timeout -s KILL 500 python tools/ab_fill.py 12288 default nobias w7 default ) > $O/ab.txt 2>&1
tail -3 $O/pytest.txt; cat $O/ab.txt
