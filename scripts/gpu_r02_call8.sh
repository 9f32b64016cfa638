#!/bin/bash
# Round 2, eighth GPU call: cells kernel with 32-bit table slots (19-21 tag bits) and the direct byte grid for the cell counts.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -8
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench8.json 2> gpurun_out/bench8.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench8.err
SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 python bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench8_w1.json 2> gpurun_out/bench8_w1.err; echo "bench w1 rc=$?"
SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_prof/libshasta_mi355x.so SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 python bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench8_prof.json 2> gpurun_out/bench8_prof.err; echo "phase profile rc=$?"; grep "phase cycles" gpurun_out/bench8_prof.err | tail -3
python - <<PY
import json
for f in ["bench8", "bench8_w1"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["stage_seconds_per_step"], "kernel s/step %.3f" % d["kernel_seconds_per_step"])
        for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            if v["seconds_per_step"] > 0.004:
                print("   %-55s %7.2f ms/step  %6.1f launches  avg %8.3f ms  %7.1f GB/s" % (k, v["seconds_per_step"] * 1e3, v["launches_per_step"], v["avg_ms"], v["achieved_GBps"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
