#!/bin/bash
# Round 2, thirty-third GPU call: early placement of the batches of a borrowed result (host side only).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
timeout 500 python bench.py --reads $READS --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench33.json 2> gpurun_out/bench33.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/bench33.json").read().strip().splitlines()[-1])
print("value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()}, d["stage_device_ms_each_step"])
PY
