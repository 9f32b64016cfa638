#!/bin/bash
# usage: scripts/gpu_pmc.sh <tag> <reads> "<counters>"   -- one rocprofv3 --pmc pass over the calibration kernels and a short bench run
TAG=$1; READS=$2; COUNTERS=$3
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc $COUNTERS --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_cal -o cal --output-format csv -- python $GRAFT_REPO_ROOT/scripts/calibrate_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_cal.log 2>&1
echo "calibration rc=$?"
timeout 900 rocprofv3 --pmc $COUNTERS --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -o $TAG --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1
echo "rocprof rc=$?"; ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
