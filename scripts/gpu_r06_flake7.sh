#!/bin/bash
# Round 6: the dense DP kernels alone again, now from SEVERAL host threads at once on the one device.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time FLAKE_THREADS=${2:-2} timeout 2400 python scripts/flake_dp_unit.py ${1:-20000} ) > gpurun_out/r06_flake7_dp_unit_threads.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake7_dp_unit_threads.log | tail -n 12
