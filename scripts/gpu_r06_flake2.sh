#!/bin/bash
# Round 6, second look for round 5's method-3 failure: thousands of calls in suite context, then the WHOLE -m gpu suite repeated (where it was
# seen), then the files that precede test_gpu_config_values.py in a whole run followed by that file, repeated.  Logs -> profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python scripts/flake_k16_suite_context.py ${1:-3000} ) > gpurun_out/r06_flake2_context.log 2>&1; tail -5 gpurun_out/r06_flake2_context.log
: > gpurun_out/r06_flake2_whole_suite.log
for i in $(seq 1 ${2:-16}); do
  timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -v "^\.\|^$" | tail -4 | sed "s/^/whole suite run $i: /" >> gpurun_out/r06_flake2_whole_suite.log
done
grep -c " passed" gpurun_out/r06_flake2_whole_suite.log; grep -i "failed\|error" gpurun_out/r06_flake2_whole_suite.log | head
: > gpurun_out/r06_flake2_suite_order.log
for i in $(seq 1 ${3:-50}); do
  timeout 600 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_align3_and_markers.py tests/test_gpu_align4.py tests/test_gpu_beyond_4g_markers.py tests/test_gpu_config_values.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^\.\|^$" | tail -3 | sed "s/^/run $i: /" >> gpurun_out/r06_flake2_suite_order.log
done
grep -c " passed" gpurun_out/r06_flake2_suite_order.log; grep -i "failed\|error" gpurun_out/r06_flake2_suite_order.log | head
