#!/bin/bash
# usage: scripts/gpu_check.sh <reads> <tag>  -- GPU parity tests, a short bench and a kernel-trace profile
READS=${1:-20000}; TAG=${2:-r}
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 800 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value',d['value'],'ms/step', d['ms_per_step'], d['stage_seconds_per_step'], 'cand', d['config']['candidates'])"
