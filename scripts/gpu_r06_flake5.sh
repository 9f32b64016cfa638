#!/bin/bash
# Round 6: the dense DP kernels alone, the same 1 600 tasks of the reproduced method-3 calls again and again (scripts/flake_dp_unit.py).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python scripts/flake_dp_unit.py ${1:-4000} ) > gpurun_out/r06_flake5_dp_unit.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake5_dp_unit.log | tail -n 12
