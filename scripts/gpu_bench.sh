#!/bin/bash
# usage: scripts/gpu_bench.sh <tag> [bench args...]   -- one bench.py run (with CPU baseline) under a timeout
TAG=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python bench.py "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"; tail -c 400 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
