#!/bin/bash
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
( timeout -s KILL 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.txt 2>&1
( timeout -s KILL 300 python bench.py --tiles 4096 --steps 4 --warmup 2 --cpu-seconds 3 ) > $O/bench_small.json 2> $O/bench_small.err
( timeout -s KILL 600 python bench.py ) > $O/bench.json 2> $O/bench.err
( timeout -s KILL 60 python bench.py --gpus 2 ) > $O/bench2.json 2> $O/bench2.err; echo "gpus2 rc=$?" >> $O/bench2.err
tail -4 $O/pytest.txt; tail -3 $O/bench_small.err; cat $O/bench_small.json; tail -3 $O/bench.err; cat $O/bench.json; cat $O/bench2.err
