#!/bin/bash
# Round 6: the device-list form with the device listed twice (the form of both events seen so far), long enough to meet the event again, with
# everything about it written down (scripts/flake_multi_form.py: both lists of pairs, whether the call made again repeats it).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time FLAKE_DEVICES=2 timeout 2700 python scripts/flake_multi_form.py ${1:-15000} ) > gpurun_out/r06_flake6_multi_form.log 2>&1; grep -v "^$\|amdgpu.ids" gpurun_out/r06_flake6_multi_form.log | tail -n 30
