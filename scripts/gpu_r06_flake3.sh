#!/bin/bash
# Round 6: the reproduced method-3 difference (device-list form, several contexts on one GPU), A/B of how the forward kernel stores its trace:
# the product (scalar stores, s_store_dwordx4 + s_dcache_wb) against the build with plain vector stores.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
N=${1:-2500}
( time timeout 1500 python scripts/flake_multi_form.py $N ) > gpurun_out/r06_flake3_scalar_stores.log 2>&1; tail -4 gpurun_out/r06_flake3_scalar_stores.log
( time FLAKE_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_vector_stores/libshasta_mi355x.so timeout 1500 python scripts/flake_multi_form.py $N ) > gpurun_out/r06_flake3_vector_stores.log 2>&1; tail -4 gpurun_out/r06_flake3_vector_stores.log
