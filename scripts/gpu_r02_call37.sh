#!/bin/bash
# Forward DP with two register sets for the prefetched kmer ids (no copies at the end of a block); DPP / EXEC probe for the next step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 30 shasta_amd/_build/dpp_exec_probe > gpurun_out/dpp_exec_probe.jsonl 2> gpurun_out/dpp_exec_probe.err; echo "probe rc=$?"
STEPS=6 WARMUP=3 bash scripts/gpu_r02_final_check.sh
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_final_check.json").read().strip().splitlines()[-1])
print([x[1] for x in d["stage_device_ms_each_step"]])
for k, v in d["kernels"].items():
    if "Forward" in k: print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("avg_ms", "seconds_per_step", "launches_per_step", "gcups")})
print(d.get("one_worker_kernel_seconds_per_step") or "")
PY
