#!/bin/bash
# Round 2, tenth GPU call: chunks of one or two candidates worked on by the whole workgroup (COOP).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -6
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench10.json 2> gpurun_out/bench10.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench10.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench10.json").read().strip().splitlines()[-1])
print("value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["stage_seconds_per_step"], "kernel s/step %.3f" % d["kernel_seconds_per_step"])
for k, v in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
    if v["seconds_per_step"] > 0.003:
        print("   one worker: %-50s %7.2f ms/step  avg %8.3f ms" % (k, v["seconds_per_step"] * 1e3, v["avg_ms"]))
print(json.dumps(d["roofline"])[:900])
PY
