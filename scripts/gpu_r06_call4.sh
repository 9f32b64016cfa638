#!/bin/bash
# Round 6, fourth call: the anchor kernel's second launch with ballot-built trace words, the windowed class's own estimate; the three lines.
ROUND=r06
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider ) > gpurun_out/${ROUND}_call4_suite.log 2>&1; tail -4 gpurun_out/${ROUND}_call4_suite.log
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 1500 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
run call4_headline --steps 10 --warmup 3 --no-cpu-baseline
run call4_headline_again --steps 10 --warmup 3 --no-cpu-baseline
run call4_may2022 --workload may2022 --steps 5 --warmup 2 --no-cpu-baseline
run call4_ul --workload ul --steps 3 --warmup 1 --baseline-sample 8000 --tie-census 0
SHASTA_MI355X_LOG_ALLOC=1 run call4_group1_alloc_log --steps 3 --warmup 3 --group --gpus 1
grep -c "device buffer" gpurun_out/${ROUND}_call4_group1_alloc_log.err; grep "device buffer\|bench: step" gpurun_out/${ROUND}_call4_group1_alloc_log.err | tail -12
python scripts/bench_summary.py gpurun_out/${ROUND}_call4_headline gpurun_out/${ROUND}_call4_may2022 gpurun_out/${ROUND}_call4_ul 2>&1 | cut -c1-300 | grep -v "^   \(radix\|bucket\|evaluate\|pairWrite\|readStat\|alignment table\|finalize\|hashWindows\|dpMetrics\|compressWrite\)"
for f in call4_headline call4_headline_again call4_group1_alloc_log; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${ROUND}_$f.json").read().strip().splitlines()[-1]); print("$f", "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d.get("stage_seconds_per_step"), d.get("in_process_group"))
except Exception as e:
    print("$f unreadable", e)
PY
done
