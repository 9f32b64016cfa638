#!/bin/bash
# Round 2, fifteenth GPU call: forward DP with scalar trace stores, gap penalties folded into the stored scores and four
# diagonals per lane in band classes 1 and 2; cells with the cell map and interleaved stream groups.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench15.json 2> gpurun_out/bench15.err; echo "bench rc=$?"
timeout 900 python bench.py --reads $READS --steps 2 --warmup 1 --baseline-sample 30000 > gpurun_out/bench15_parity.json 2> gpurun_out/bench15_parity.err; echo "bench with baseline rc=$?"
python - <<PY
import json
for f in ["bench15", "bench15_parity"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "align4_device %.1f" % (d["stage_seconds_per_step"]["align4_device"] * 1e3))
        if d.get("cpu_baseline"): print("   cpu_baseline", json.dumps(d["cpu_baseline"])[:700])
        for k, x in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            if x["seconds_per_step"] > 0.003:
                print("   one worker: %-45s %7.2f ms/step  avg %8.3f ms" % (k, x["seconds_per_step"] * 1e3, x["avg_ms"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
