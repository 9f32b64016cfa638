#!/bin/bash
# Last call of the round: the -m gpu suite, the driver's smoke entry and a short bench line on the final build.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 500 python bench.py --steps ${STEPS:-3} --warmup ${WARMUP:-1} ${BASELINE_FLAG---no-cpu-baseline} > gpurun_out/bench_final_check.json 2> gpurun_out/bench_final_check.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_final_check.json").read().strip().splitlines()[-1])
print("value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()})
PY
