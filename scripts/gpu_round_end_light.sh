#!/bin/bash
# The closing call without the other modes: suite, counters of the build in the tree, the bench line as the driver runs it, the whole-list parity run.
ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider --durations=5 ) > gpurun_out/${ROUND}_final_suite.log 2>&1; tail -12 gpurun_out/${ROUND}_final_suite.log
ROUND=$ROUND bash scripts/gpu_counters.sh
cp gpurun_out/${ROUND}_pmc_100k_reads.json profiles/${ROUND}_pmc_100k_reads.json
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 1500 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
run bench_final --steps 20 --warmup 5
run bench_final_whole_baseline --steps 2 --warmup 1 --baseline-sample 0 --tie-census 0
SHASTA_MI355X_ALIGN_WORKERS=1 run bench_final_w1 --steps 2 --warmup 1 --no-cpu-baseline
python scripts/bench_summary.py gpurun_out/${ROUND}_bench_final gpurun_out/${ROUND}_bench_final_whole_baseline gpurun_out/${ROUND}_bench_final_w1 2>&1 | cut -c1-400
