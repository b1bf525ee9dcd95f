#!/bin/bash
# tools/ab.sh TILES lib1 lib2 ... : fill-kernel A/B on the GPU box (prints dominant launch ms per lib)
T=$1; shift
for L in "$@"; do
  if [ "$L" = "default" ]; then unset CVX_LIB; else export CVX_LIB=$PWD/ngmlr_amd/variants/libcvxalign_$L.so; fi
  timeout -s KILL 100 python bench.py --tiles $T --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), 'Gbp/h | fill launch ms', round(d['roofline']['launch_ms'],2), 'gcups', round(d['roofline']['gcups']), 'frac', round(d['roofline']['frac'],4), d['roofline']['all_fill_launches'], 'bt', round(d['stage_ms_per_step']['backtrack'],2), 'tot', round(d['device_resident']['ms_per_step'],2), (d['parity'] or '')[:3])"
done
