#!/bin/bash
# Round 6, ready for the next call with a GPU: the search in suite context (the one run that met the event) with the tools made since --
#   a  scratch and new buffers scrambled (SHASTA_MI355X_SCRAMBLE=1): a read of something the batch has not written answers differently at once
#   b  the same without the fork of the wide-band classes to the side stream (SHASTA_MI355X_DP_FORK=0)
#   c  the dense kernels through the unit seam, which now forks like a batch does
# and the event captured (flake_multi_form.py writes the differing results to gpurun_out/).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
N=${1:-1500}
( time SHASTA_MI355X_DP_FORK=1 SHASTA_MI355X_SCRAMBLE=1 timeout 1500 python scripts/flake_k16_suite_context.py $N ) > gpurun_out/r06_flake8_context_scramble.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake8_context_scramble.log | tail -n 6
( time SHASTA_MI355X_DP_FORK=0 timeout 1500 python scripts/flake_k16_suite_context.py $N ) > gpurun_out/r06_flake8_context_no_fork.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake8_context_no_fork.log | tail -n 6
( time SHASTA_MI355X_DP_FORK=1 timeout 1200 python scripts/flake_dp_unit.py 10000 ) > gpurun_out/r06_flake8_dp_unit_fork.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake8_dp_unit_fork.log | tail -n 6
