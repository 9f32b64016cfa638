#!/bin/bash
# Round 2, twenty-seventh GPU call: what each part of the cells kernel costs -- builds with one part removed (wrong results, timing only).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in base nograph nofurther nocount nostream; do
  if [ $V = base ]; then unset SHASTA_MI355X_LIBRARY; else export SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_abl_$V/libshasta_mi355x.so; fi
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 600 python bench.py --reads $READS --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench27_$V.json 2> gpurun_out/bench27_$V.err; echo "bench $V rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench27_$V.json").read().strip().splitlines()[-1])
    rows = d["kernels"]
    print("$V: ms/step %.1f" % d["ms_per_step"], " cells chunk kernel %.2f ms/step" % (1e3 * sum(x["seconds_per_step"] for k, x in rows.items() if "CellsChunk" in k)), " alignments", d["config"].get("alignments_stored"))
except Exception as e:
    print("$V unreadable", e); print(open("gpurun_out/bench27_$V.err").read()[-600:])
PY
done
