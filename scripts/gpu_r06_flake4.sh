#!/bin/bash
# Round 6: is it the trace?  The device-list form with the trace POISONED before every forward launch (a record the walk reads before the
# forward kernel's store has reached memory then shows, whatever an earlier identical call left there): scalar stores against vector stores.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
N=${1:-1500}
export SHASTA_MI355X_TRACE_POISON=1 FLAKE_DEVICES=2
( time timeout 1500 python scripts/flake_multi_form.py $N ) > gpurun_out/r06_flake4_poison_scalar_stores.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake4_poison_scalar_stores.log | tail -n 8
( time FLAKE_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_vector_stores/libshasta_mi355x.so timeout 1500 python scripts/flake_multi_form.py $N ) > gpurun_out/r06_flake4_poison_vector_stores.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake4_poison_vector_stores.log | tail -n 8
