#!/bin/bash
# The first GPU call of a round: the whole -m gpu suite, then the bench line with the step = computeAlignments end to end under the
# library's switches -- the default, without the anchor kernel (SHASTA_MI355X_ANCHORED_DP=0), without the sparse path
# (SHASTA_MI355X_SPARSE_DP=0: round 3's DP) -- same box, same reads; then the kernel-trace statistics and the PMC passes of the default.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'ROUND=r05 bash scripts/gpu_first_call.sh'
ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM"}'), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
if [ -z "$NO_SUITE" ]; then
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider --durations=6 ) 2>&1 | tail -24
fi
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
run() {   # name, extra environment..., then -- bench arguments
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( time env "${envs[@]}" SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 900 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real
  grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -3
}
run bench_a X=1 -- --steps 6 --warmup 2 --baseline-sample ${SAMPLE:-60000} --tie-census 0
run bench_a_no_anchors SHASTA_MI355X_ANCHORED_DP=0 -- --steps 6 --warmup 2 --no-cpu-baseline
run bench_a_dense SHASTA_MI355X_SPARSE_DP=0 -- --steps 6 --warmup 2 --no-cpu-baseline
ROUND=$ROUND python scripts/bench_summary.py gpurun_out/${ROUND}_bench_a gpurun_out/${ROUND}_bench_a_no_anchors gpurun_out/${ROUND}_bench_a_dense
if [ -n "$WITH_PROFILE" ]; then ROUND=$ROUND bash scripts/gpu_counters.sh; fi
