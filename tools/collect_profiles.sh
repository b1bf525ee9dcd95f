#!/bin/bash
# tools/collect_profiles.sh TAG -- on the GPU box: the official bench line plus the rocprofv3 passes the
# roofline numbers come from.  Writes everything under gpurun_out/TAG_*; summarise afterwards with
#   python tools/rocpd_summary.py gpurun_out/TAG_stats/*.db > profiles/TAG_stats.txt   (same for fetch/write/sq)
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 700 python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o p -- python $R/bench.py --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o p -- python $R/bench.py --steps 1 --warmup 0 --resident-steps 0 --no-cpu-baseline > $OUT/${TAG}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o p -- python $R/bench.py --steps 1 --warmup 0 --resident-steps 0 --no-cpu-baseline > $OUT/${TAG}_write.log 2>&1
# issue-side counters: instructions, wave cycles, and the three disjoint wave states (parked / issue-stalled / issuing)
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/${TAG}_sq -o p -- python $R/bench.py --tiles 12288 --steps 1 --warmup 0 --resident-steps 0 --no-cpu-baseline > $OUT/${TAG}_sq.log 2>&1
# VALU pipe occupancy proper (cycles the VALU is executing, not instruction counts), if this rocprofv3 has the counters
rocprofv3 -L > $OUT/${TAG}_counters_list.txt 2>&1
SQ2=""
for c in SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES; do
	grep -q "\b$c\b" $OUT/${TAG}_counters_list.txt && SQ2="$SQ2 $c"
done
echo "second SQ pass:$SQ2"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $SQ2 -d $OUT/${TAG}_sq2 -o p -- python $R/bench.py --tiles 12288 --steps 1 --warmup 0 --resident-steps 0 --no-cpu-baseline > $OUT/${TAG}_sq2.log 2>&1
# round 5: the other BASELINE configs' fill classes (ONT mix: chained retries / M = 4 / M = 3; C5 mix) -- kernel trace and the SQ pass
# per class (tools/ab_knobs.py: one batch, inputs resident, four runs), and the genome-scale candidate search with its HBM traffic
for W in ont c5; do
	timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${W}_stats -o p -- python $R/tools/ab_knobs.py $W -- "" > $OUT/${TAG}_${W}_stats.log 2>&1
	timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/${TAG}_${W}_sq -o p -- python $R/tools/ab_knobs.py $W -- "" > $OUT/${TAG}_${W}_sq.log 2>&1
done
timeout -s KILL 300 python $R/tools/search_rates.py --big 512 100000 > $OUT/${TAG}_search_big.json 2> $OUT/${TAG}_search_big.err
# round 6: the same over a 2 Gbp reference (3.2 GB table, 5 000 votes per sub-read: no LDS map holds one, every read goes through the table in HBM)
CVX_SEARCH_TRACE=1 timeout -s KILL 400 python $R/tools/search_rates.py --big 2048 20000 > $OUT/${TAG}_search_2gbp.json 2> $OUT/${TAG}_search_2gbp.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_search_stats -o p -- python $R/tools/search_rates.py --big 512 100000 > $OUT/${TAG}_search_stats.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_search_fetch -o p -- python $R/tools/search_rates.py --big 512 100000 > $OUT/${TAG}_search_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_search_write -o p -- python $R/tools/search_rates.py --big 512 100000 > $OUT/${TAG}_search_write.log 2>&1
for p in stats fetch write sq sq2 ont_stats ont_sq c5_stats c5_sq search_stats search_fetch search_write; do
	db=$(ls $OUT/${TAG}_$p/*.db 2>/dev/null | head -1)
	[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/${TAG}_$p.txt 2>&1
done
ls -la $OUT | tail -30
python $R/tools/make_pmc_json.py $TAG $OUT/${TAG}_pmc.json > /dev/null 2>&1 && echo "wrote $OUT/${TAG}_pmc.json (build $(python -c "import sys; sys.path.insert(0,'$R'); from ngmlr_amd import capi; print(capi.load().cvx_build_id().decode())"))"
