#!/bin/bash
# Round 2, sixteenth GPU call: host workers x batch size of the aligner call with the faster kernels.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for CFG in "6 18" "8 18" "8 17" "6 17" "8 16" "4 18" "3 18"; do
  set -- $CFG
  SHASTA_MI355X_ALIGN_WORKERS=$1 SHASTA_MI355X_ALIGN_BATCH_LOG2=$2 timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench16_w$1_b$2.json 2> gpurun_out/bench16_w$1_b$2.err; echo "bench workers $1 batch 2^$2 rc=$?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench16_w$1_b$2.json").read().strip().splitlines()[-1])
print("workers $1 batch 2^$2: value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()}, "kernel s/step %.3f" % d["kernel_seconds_per_step"])
PY
done
