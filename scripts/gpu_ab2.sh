#!/bin/bash
# A/B of the library's switches on one box: suite (optional), then the bench line under each environment given as NAME:VAR=VAL,VAR=VAL ...
#   gpurun -- 'ROUND=r05 TAG=call2 SUITE=1 bash scripts/gpu_ab2.sh default: lane:SHASTA_MI355X_CHAIN_WAVE=0'
ROUND=${ROUND:-r05}; TAG=${TAG:-ab}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -n "$SUITE" ]; then ( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider -x ) 2>&1 | tail -12; fi
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
names=()
first=1
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  envs=(X=1); IFS=',' read -ra parts <<< "$vars"; for p in "${parts[@]}"; do [ -n "$p" ] && envs+=("$p"); done
  extra="--no-cpu-baseline"; if [ -n "$first" ] && [ -n "$BASELINE" ]; then extra="--baseline-sample $BASELINE --tie-census 0"; fi
  first=
  ( time env "${envs[@]}" SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${TAG}_${name}_details.json timeout 900 python bench.py --steps ${STEPS:-6} --warmup 2 $extra > gpurun_out/${ROUND}_${TAG}_${name}.json 2> gpurun_out/${ROUND}_${TAG}_${name}.err ) 2>&1 | grep real
  grep -v "^bench details: " gpurun_out/${ROUND}_${TAG}_${name}.err | tail -3
  names+=(gpurun_out/${ROUND}_${TAG}_${name})
done
python scripts/bench_summary.py "${names[@]}"
