#!/bin/bash
# Round 6, third call: the suite on the build with the anchor kernel's band-relative rows, the wavefront traceback, the pair-less wave kernel;
# the three workloads; the ultra-long shape with the reference aligner on EVERY candidate; A/B of the windowed class's stream and wavefronts.
ROUND=r06
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider --durations=5 ) > gpurun_out/${ROUND}_call3_suite.log 2>&1; tail -6 gpurun_out/${ROUND}_call3_suite.log
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 1500 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
run call3_headline --steps 10 --warmup 3 --no-cpu-baseline
SHASTA_MI355X_ANCHOR_BIG=0 run call3_headline_no_second_anchor_launch --steps 10 --warmup 3 --no-cpu-baseline
run call3_headline_again --steps 10 --warmup 3 --no-cpu-baseline
run call3_may2022 --workload may2022 --steps 5 --warmup 2 --no-cpu-baseline
run call3_ul --workload ul --steps 3 --warmup 1 --no-cpu-baseline
SHASTA_MI355X_CELLS_SIDE_FROM=4 run call3_ul_long_on_side --workload ul --steps 3 --warmup 1 --no-cpu-baseline
SHASTA_MI355X_LONG_WAVES=8 run call3_ul_long_8_waves --workload ul --steps 3 --warmup 1 --no-cpu-baseline
run call3_ul_whole_baseline --workload ul --steps 2 --warmup 1 --baseline-sample 0 --tie-census 0
run call3_group1 --steps 10 --warmup 3 --group --gpus 1
SHASTA_MI355X_DEBUG=1 SHASTA_MI355X_ALIGN_WORKERS=1 run call3_ul_debug --workload ul --steps 1 --warmup 0 --no-cpu-baseline
grep "cells: round\|cells: HBM" gpurun_out/${ROUND}_call3_ul_debug.err | sort | uniq -c | sort -rn | head -20
python scripts/bench_summary.py gpurun_out/${ROUND}_call3_headline gpurun_out/${ROUND}_call3_may2022 gpurun_out/${ROUND}_call3_ul gpurun_out/${ROUND}_call3_ul_whole_baseline 2>&1 | cut -c1-330
for f in call3_headline call3_headline_no_second_anchor_launch call3_headline_again call3_ul call3_ul_long_on_side call3_ul_long_8_waves call3_group1; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${ROUND}_$f.json").read().strip().splitlines()[-1]); print("$f", "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d.get("stage_seconds_per_step"), d.get("in_process_group"))
except Exception as e:
    print("$f unreadable", e)
PY
done
