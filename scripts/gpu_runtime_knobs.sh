#!/bin/bash
# HIP runtime knobs (environment only), one bench run each: the aligner's ms per call.  Usage: gpu_runtime_knobs.sh "A=1" "B=2 C=3" ...
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/runtime_knobs.log; : > $out
run() {
  line=$(env $1 SHASTA_BENCH_NO_GROUP_LINE=1 timeout 170 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$1: $(python - "$line" <<'P'
import json,sys
try:
    d=json.loads(sys.argv[1]); s=d["stage_device_ms_each_step"]
    print("ms/step %.1f lowhash0 call %.1f align %s mean %.1f" % (d["ms_per_step"], d["stage_seconds_per_step"]["lowhash0_call"]*1e3, [round(x[1],1) for x in s], sum(x[1] for x in s)/len(s)))
except Exception as e:
    print("failed", e, sys.argv[1][:200])
P
)" >> $out
}
for k in "$@"; do run "$k"; done
cat $out
