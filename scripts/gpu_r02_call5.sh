#!/bin/bash
# Round 2, fifth GPU call: LowHash0 without host round trips inside the iteration loop (device-side counts, one evaluation of
# the pair keys of all iterations); tail tracebacks on the side stream.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -8
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench5.json 2> gpurun_out/bench5.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench5.err
timeout 900 python bench.py --reads $READS --steps 5 --warmup 2 --no-cpu-baseline --lowhash-only > gpurun_out/bench5_lh.json 2> gpurun_out/bench5_lh.err; echo "bench lh rc=$?"; tail -c 300 gpurun_out/bench5_lh.err
python - <<PY
import json
for f in ["bench5", "bench5_lh"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["stage_seconds_per_step"], "kernel s/step %.3f" % d["kernel_seconds_per_step"])
        for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            if f == "bench5_lh" or v["seconds_per_step"] > 0.004:
                print("   %-55s %7.2f ms/step  %6.1f launches  avg %8.3f ms  %7.1f GB/s" % (k, v["seconds_per_step"] * 1e3, v["launches_per_step"], v["avg_ms"], v["achieved_GBps"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof5 -o lh --output-format csv -- python $R/bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline --lowhash-only > $R/gpurun_out/prof5.log 2>&1
echo "rocprof stats rc=$?"
