#!/bin/bash
# The closing call in one bench run less: suite, counters of the build in the tree, the bench line as the driver runs it WITH the reference's
# aligner on every candidate (--baseline-sample 0: the line's parity block is the whole-list one), one worker, and the A/B of one switch
# on the same box.   AB="SHASTA_MI355X_CHAIN_WAVE_STREAM=0" ROUND=r05 bash scripts/gpu_round_end_ab2.sh
ROUND=${ROUND:-r05}
AB=${AB:?the switch to set for the second measurement, NAME=value}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 400 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider --durations=5 ) > gpurun_out/${ROUND}_final_suite.log 2>&1; tail -12 gpurun_out/${ROUND}_final_suite.log
ROUND=$ROUND bash scripts/gpu_counters.sh
cp gpurun_out/${ROUND}_pmc_100k_reads.json profiles/${ROUND}_pmc_100k_reads.json
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 600 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
run bench_final --steps 20 --warmup 5 --baseline-sample 0
SHASTA_MI355X_ALIGN_WORKERS=1 run bench_final_w1 --steps 2 --warmup 1 --no-cpu-baseline
echo "== with $AB"
( export $AB; run bench_ab --steps 20 --warmup 5 --no-cpu-baseline )
( export $AB SHASTA_MI355X_ALIGN_WORKERS=1; run bench_ab_w1 --steps 2 --warmup 1 --no-cpu-baseline )
echo "== default again (the box's drift between the first and the last measurement)"
run bench_final_again --steps 20 --warmup 5 --no-cpu-baseline
python scripts/bench_summary.py gpurun_out/${ROUND}_bench_final gpurun_out/${ROUND}_bench_final_w1 gpurun_out/${ROUND}_bench_ab gpurun_out/${ROUND}_bench_ab_w1 gpurun_out/${ROUND}_bench_final_again 2>&1 | cut -c1-330
