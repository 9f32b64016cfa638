#!/bin/bash
# Round 6's closing measurement call on the build in the tree:
#   1 the -m gpu suite   2 kernel-trace statistics (six workers, one worker), timeline, the four PMC passes -> the PMC summary lands in
#   profiles/ ON THE BOX so that the bench lines that follow price their kernels with counters of this build   3 the bench line as the
#   driver runs it (CPU baseline + parity at bench size on one candidate in eight)   4 the whole candidate list through the reference aligner
#   5 the other workloads and modes.
ROUND=r06
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM"}'), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider --durations=5 ) > gpurun_out/${ROUND}_final_suite.log 2>&1; tail -8 gpurun_out/${ROUND}_final_suite.log
ROUND=$ROUND bash scripts/gpu_counters.sh
cp gpurun_out/${ROUND}_pmc_100k_reads.json profiles/${ROUND}_pmc_100k_reads.json
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 1500 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
run bench_final --steps 20 --warmup 5
run bench_final_default
run bench_final_whole_baseline --steps 3 --warmup 2 --baseline-sample 0 --tie-census 0
run bench_ul --workload ul --steps 5 --warmup 2 --baseline-sample 20000 --tie-census 0
run bench_may2022 --workload may2022 --steps 10 --warmup 3 --baseline-sample 100000 --tie-census 0
run bench_final_lh --steps 5 --warmup 2 --no-cpu-baseline --lowhash-only
run bench_final_m3 --steps 4 --warmup 3 --no-cpu-baseline --align-method 3
SHASTA_MI355X_ALIGN_WORKERS=1 run bench_final_w1 --steps 2 --warmup 1 --no-cpu-baseline
run bench_final_group1 --steps 10 --warmup 3 --group --gpus 1
SHASTA_BENCH_FORCE_SHARDED=1 SHASTA_BENCH_NO_GROUP_LINE=1 run bench_final_one_rank_rccl --steps 10 --warmup 3 --no-cpu-baseline
# (why is LowHash0 through the group slower than on a context?  allocations per step, and the host's clock inside the call)
SHASTA_MI355X_LOG_ALLOC=1 run group1_alloc_log --steps 3 --warmup 3 --group --gpus 1
grep -c "device buffer" gpurun_out/${ROUND}_group1_alloc_log.err; grep "device buffer" gpurun_out/${ROUND}_group1_alloc_log.err | tail -12
python scripts/bench_summary.py gpurun_out/${ROUND}_bench_final gpurun_out/${ROUND}_bench_final_default gpurun_out/${ROUND}_bench_final_whole_baseline gpurun_out/${ROUND}_bench_ul gpurun_out/${ROUND}_bench_may2022 2>&1 | cut -c1-400
for f in bench_final_lh bench_final_m3 bench_final_w1 bench_final_group1 bench_final_one_rank_rccl; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${ROUND}_$f.json").read().strip().splitlines()[-1]); print("$f", d["metric"], "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d.get("stage_seconds_per_step"), d.get("in_process_group"))
except Exception as e:
    print("$f unreadable", e)
PY
done
