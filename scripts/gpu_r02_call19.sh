#!/bin/bash
# Round 2, nineteenth GPU call: traceback of every class through the LDS-window kernel against the register-resident one.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in regs lds; do
  if [ $V = lds ]; then export SHASTA_MI355X_TRACEBACK_LDS=1; else unset SHASTA_MI355X_TRACEBACK_LDS; fi
  timeout 300 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_assembler_mirror.py tests/test_gpu_align4.py -q -m gpu -x --timeout 200 -p no:cacheprovider 2>&1 | tail -1
  timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench19_$V.json 2> gpurun_out/bench19_$V.err; echo "bench $V rc=$?"
done
unset SHASTA_MI355X_TRACEBACK_LDS
python - <<PY
import json
for v in ["regs", "lds"]:
    d = json.loads(open("gpurun_out/bench19_%s.json" % v).read().strip().splitlines()[-1])
    print(v, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(x * 1e3, 1) for k, x in d["stage_seconds_per_step"].items()})
    for k, x in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
        if "Traceback" in k:
            print("   one worker: %-45s %7.2f ms/step  avg %8.3f ms" % (k, x["seconds_per_step"] * 1e3, x["avg_ms"]))
PY
