#!/bin/bash
# Round 2, twenty-ninth GPU call: fewer cells workgroups per CU (LDS padding) -- do other workers' kernels then run beside them?
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for PAD in 0 12000 24000 0; do
  SHASTA_MI355X_CELLS_LDS_PAD=$PAD timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench29_$PAD.json 2> gpurun_out/bench29_$PAD.err; echo "bench pad $PAD rc=$?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench29_$PAD.json").read().strip().splitlines()[-1])
print("pad $PAD: value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()}, "cells solo %.1f" % (1e3 * sum(x["seconds_per_step"] for k, x in d["kernels_one_worker"].items() if "CellsChunk" in k)))
PY
done
