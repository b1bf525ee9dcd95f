#!/bin/bash
# tools/e2e_kernel_stats.sh [N_READS] -- on the GPU box: rocprofv3 --kernel-trace --stats around ngmlr's own pipeline
# (oracle/_ref/ngmlr_hip_all on N synthetic 10 kb reads, tools/e2e_rates.py's plain workload): which kernels run inside ngmlr
# and for how long -- fills, walks, the window decode, the k-mer vote, sub-read scoring, the table build.  Summary (per kernel:
# calls, total, average) to gpurun_out/e2e_kernel_stats.txt; copy it to profiles/.
N=${1:-20000}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
W=$(mktemp -d /tmp/e2eks.XXXX)
cd /tmp && export TMPDIR=/tmp
python - "$R" "$W" "$N" <<'PY'
import sys, os
R, W, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import e2e_rates
b = e2e_rates.write_plain_workload(os.path.join(W, "ref.fa"), os.path.join(W, "reads.fq"), N, np.random.default_rng(2025), 2000000)
print("workload: %d reads, %.1f Mbp" % (N, b / 1e6))
PY
export CVX_POOL_CONTEXTS=4096 CVX_BATCH_TARGET=2048 CVX_BATCH_HOLD_US=10000 CVX_CS_BATCH=20
cd $W
# once without the profiler (the binary writes and caches its index files next to the reference), then under it
timeout 300 $R/oracle/_ref/ngmlr_hip_all --skip-write -x pacbio -t 20 -R 0.01 --no-progress -r ref.fa -q reads.fq > plain.sam 2> plain.err
grep -E "SharedAligner: [0-9]+ alignments|windows of" plain.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $W/prof -o p -- $R/oracle/_ref/ngmlr_hip_all --skip-write -x pacbio -t 20 -R 0.01 --no-progress -r ref.fa -q reads.fq > prof.sam 2> prof.err
echo "profiled run rc=$?"
cmp <(grep -v '^@' plain.sam | sort) <(grep -v '^@' prof.sam | sort) && echo "SAM of the profiled run = SAM of the plain run"
db=$(ls $W/prof/*.db $W/prof/*/*.db 2>/dev/null | head -1)
{ echo "# rocprofv3 --kernel-trace --stats around oracle/_ref/ngmlr_hip_all -t 20 on $N synthetic 10 kb reads (2 Mbp reference): every kernel the pipeline runs"; grep -E "SharedAligner: [0-9]+ alignments|windows of|AlignPool: [0-9]+ reads" prof.err; [ -n "$db" ] && python $R/tools/rocpd_summary.py $db; } > $OUT/e2e_kernel_stats.txt 2>&1
head -40 $OUT/e2e_kernel_stats.txt | cut -c1-200
rm -rf $W
