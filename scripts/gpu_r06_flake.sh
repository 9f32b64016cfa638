#!/bin/bash
# Round 6: the method-3 failure of round 5 looked for in suite context (scripts/flake_k16_suite_context.py) and in the suite's own order
# (the files that precede test_gpu_config_values.py in a whole run, then that file), repeated.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python scripts/flake_k16_suite_context.py ${1:-60} ) > gpurun_out/r06_flake_context.log 2>&1; tail -5 gpurun_out/r06_flake_context.log
: > gpurun_out/r06_flake_suite_order.log
for i in $(seq 1 ${2:-12}); do
  timeout 600 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_align3_and_markers.py tests/test_gpu_align4.py tests/test_gpu_beyond_4g_markers.py tests/test_gpu_config_values.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | sed "s/^/run $i: /" >> gpurun_out/r06_flake_suite_order.log
done
grep -c passed gpurun_out/r06_flake_suite_order.log; grep -i "failed\|error" gpurun_out/r06_flake_suite_order.log | head
tail -4 gpurun_out/r06_flake_suite_order.log
