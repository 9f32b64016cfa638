#!/bin/bash
# Round 2, thirty-fifth GPU call: smaller batches on the final build.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for CFG in "6 196608" "6 131072" "6 262144" "6 229376" "8 196608"; do
  set -- $CFG
  SHASTA_MI355X_ALIGN_WORKERS=$1 SHASTA_MI355X_ALIGN_BATCH=$2 timeout 300 python bench.py --reads $READS --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench35_w$1_b$2.json 2> gpurun_out/bench35_w$1_b$2.err; echo "bench workers $1 batch $2 rc=$?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench35_w$1_b$2.json").read().strip().splitlines()[-1])
print("workers $1 batch $2: value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()}, d["stage_device_ms_each_step"])
PY
done
