#!/bin/bash
# Round 2, twenty-fourth GPU call: batch sizes that divide the candidates evenly among the workers; per-batch host phases.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_assembler_mirror.py tests/test_gpu_align4.py tests/test_gpu_kernel_sweeps.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -2
for CFG in "6 262144" "6 330700" "6 165400" "7 283500" "6 262144"; do
  set -- $CFG
  SHASTA_MI355X_ALIGN_WORKERS=$1 SHASTA_MI355X_ALIGN_BATCH=$2 timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench24_w$1_b$2.json 2> gpurun_out/bench24_w$1_b$2.err; echo "bench workers $1 batch $2 rc=$?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench24_w$1_b$2.json").read().strip().splitlines()[-1])
print("workers $1 batch $2: value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()}, "kernel s/step %.3f" % d["kernel_seconds_per_step"])
PY
done
SHASTA_MI355X_DEBUG=1 timeout 600 python bench.py --reads $READS --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench24_debug.json 2> gpurun_out/bench24_debug.err; echo "debug rc=$?"
grep "^batch " gpurun_out/bench24_debug.err | tail -24
