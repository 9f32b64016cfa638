#!/bin/bash
# Round 6, first call: the -m gpu suite on the build with the windowed cells class, then the three workloads' lines (headline: no regression;
# ul / may2022: k = 14 alphabet, suppression between the stages, parity on a sample).
ROUND=r06
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM"}'), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider --durations=5 ) > gpurun_out/${ROUND}_call2_suite.log 2>&1; tail -12 gpurun_out/${ROUND}_call2_suite.log
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 1500 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
run call2_headline --steps 10 --warmup 3 --no-cpu-baseline
run call2_ul --workload ul --steps 3 --warmup 1 --baseline-sample 8000 --tie-census 0
run call2_may2022 --workload may2022 --steps 5 --warmup 2 --baseline-sample 30000 --tie-census 0
python scripts/bench_summary.py gpurun_out/${ROUND}_call2_headline gpurun_out/${ROUND}_call2_ul gpurun_out/${ROUND}_call2_may2022 2>&1 | cut -c1-600
