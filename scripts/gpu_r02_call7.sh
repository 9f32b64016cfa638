#!/bin/bash
# Round 2, seventh GPU call: suite (with the in-process multi-device tests), bench, and the phase profile of the cells kernel.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -8
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench7.json 2> gpurun_out/bench7.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench7.err
SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_prof/libshasta_mi355x.so SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 python bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench7_prof.json 2> gpurun_out/bench7_prof.err; echo "phase profile rc=$?"; grep "phase cycles" gpurun_out/bench7_prof.err | tail -3
python - <<PY
import json
for f in ["bench7"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["stage_seconds_per_step"], "kernel s/step %.3f" % d["kernel_seconds_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
