#!/bin/bash
# (the switch this script toggles existed in commit c87ea6f only)
# A/B of the aligner's batch schedule (short first batches, shrinking last ones) against batches of equal size, same box, same session.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for mode in schedule equal schedule equal; do
  if [ $mode = schedule ]; then export SHASTA_MI355X_ALIGN_GRADED_BATCHES=1; else unset SHASTA_MI355X_ALIGN_GRADED_BATCHES; fi
  timeout 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err; echo "$mode rc=$?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$mode.json").read().strip().splitlines()[-1])
print("$mode", "ms/step %.1f" % d["ms_per_step"], d["stage_device_ms_each_step"], d.get("parity_at_bench_size"))
PY
done 2>&1 | tee gpurun_out/batch_schedule.log
