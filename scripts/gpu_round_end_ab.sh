#!/bin/bash
# The closing call (scripts/gpu_round_end_light.sh) followed by the same bench with ONE switch set, on the same box: the A/B of a
# default against the form it replaced.   AB="SHASTA_MI355X_CHAIN_WAVE_WIDE_D=1" ROUND=r05 bash scripts/gpu_round_end_ab.sh
ROUND=${ROUND:-r05}
AB=${AB:?the switch to set for the second measurement, NAME=value}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
ROUND=$ROUND bash scripts/gpu_round_end_light.sh
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 600 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
echo "== with $AB"
( export $AB; run bench_ab --steps 20 --warmup 5 --no-cpu-baseline )
( export $AB SHASTA_MI355X_ALIGN_WORKERS=1; run bench_ab_w1 --steps 2 --warmup 1 --no-cpu-baseline )
echo "== default again (the box's drift between the first and the last measurement)"
run bench_final_again --steps 20 --warmup 5 --no-cpu-baseline
python scripts/bench_summary.py gpurun_out/${ROUND}_bench_ab gpurun_out/${ROUND}_bench_ab_w1 gpurun_out/${ROUND}_bench_final_again 2>&1 | cut -c1-300 | grep -v "^   \(radix\|bucket\|pairWrite\|evaluate\|alignment table\|hashWindows\|DP task\|dpMetrics\|banded\)"
