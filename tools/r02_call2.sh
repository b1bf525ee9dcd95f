#!/bin/bash
mkdir -p gpurun_out/r02b
O=gpurun_out/r02b
( timeout -s KILL 500 python tools/ab_fill.py 12288 default xa8 xa16 xb4 xb8 xs8 xs16 default ) > $O/ab_probe.txt 2>&1
cat $O/ab_probe.txt
