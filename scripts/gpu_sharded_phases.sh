#!/bin/bash
# Where the N-rank branch's LowHash0 wall clock goes, with the one rank a one-GPU box allows (over RCCL).
cd "$GRAFT_REPO_ROOT"
SHASTA_BENCH_SHARDED_PHASES=${PHASES-1} SHASTA_BENCH_FORCE_SHARDED=1 SHASTA_BENCH_NO_GROUP_LINE=1 timeout 200 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline 2>gpurun_out/sharded_phases.err | tail -1 > gpurun_out/sharded_phases.json
python - <<'P'
import json
d=json.loads(open("gpurun_out/sharded_phases.json").read())
print("ms/step", round(d["ms_per_step"],1), d["stage_seconds_per_step"])
print(d["stage_device_ms_each_step"])
for k,v in d.get("sharded_lowhash0_phase_ms_per_step", {}).items(): print("  %-28s %8.3f" % (k,v))
print(d.get("sharded_lowhash0_phase_ms_each_step"))
P
