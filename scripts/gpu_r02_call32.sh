#!/bin/bash
# Round 2, thirty-second GPU call: component ties resolved as the reference does (active cells of the tied candidates looked at again, its union-find order reproduced on the host).
# bundle); align method 3 beyond 8192 down-sampled diagonals.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench32.json 2> gpurun_out/bench32.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/bench32.json").read().strip().splitlines()[-1])
print("value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()})
for k, x in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
    if x["seconds_per_step"] > 0.004:
        print("   one worker: %-45s %7.2f ms/step  avg %8.3f ms" % (k, x["seconds_per_step"] * 1e3, x["avg_ms"]))
PY
