#!/bin/bash
# Round 2, ninth GPU call: cells kernel with the byte grid by default, overlapped LDS operations in the count step, one further
# match per lane and iteration; forward DP with the trace line in a vector register (v_writelane); class-0 table of 4096 (variant).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -8
for V in base na12; do
  if [ $V = base ]; then unset SHASTA_MI355X_LIBRARY; else export SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_$V/libshasta_mi355x.so; fi
  timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench9_$V.json 2> gpurun_out/bench9_$V.err; echo "bench $V rc=$?"
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 python bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench9_${V}_w1.json 2> gpurun_out/bench9_${V}_w1.err; echo "bench $V w1 rc=$?"
done
SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_prof/libshasta_mi355x.so SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 python bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench9_prof.json 2> gpurun_out/bench9_prof.err; echo "phase profile rc=$?"; grep "phase cycles" gpurun_out/bench9_prof.err | tail -3
unset SHASTA_MI355X_LIBRARY
python - <<PY
import json
for f in ["bench9_base", "bench9_base_w1", "bench9_na12", "bench9_na12_w1"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["stage_seconds_per_step"], "kernel s/step %.3f" % d["kernel_seconds_per_step"])
        for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            if v["seconds_per_step"] > 0.015:
                print("   %-55s %7.2f ms/step  %6.1f launches  avg %8.3f ms  %7.1f GB/s" % (k, v["seconds_per_step"] * 1e3, v["launches_per_step"], v["avg_ms"], v["achieved_GBps"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
