#!/bin/bash
# run tools/gpu_quick.py against several library variants with hard kill timeouts
for L in "$@"; do
  if [ "$L" = "default" ]; then unset CVX_LIB; else export CVX_LIB=$PWD/ngmlr_amd/variants/libcvxalign_$L.so; fi
  echo "== $L"; timeout -s KILL 60 python -u tools/gpu_quick.py 40 2>&1 | tail -4; echo "exit $?"
done
