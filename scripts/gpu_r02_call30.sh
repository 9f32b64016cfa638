#!/bin/bash
# Round 2, thirtieth GPU call: first pass of the cells kernel on per-marker match masks; five wavefronts per SIMD forced (96 VGPRs, 16 bytes spilled).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -3
for V in base w5; do
  if [ $V = base ]; then unset SHASTA_MI355X_LIBRARY; else export SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_$V/libshasta_mi355x.so; fi
  timeout 300 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_align4.py -q -m gpu -x --timeout 200 -p no:cacheprovider 2>&1 | tail -1
  timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench30_$V.json 2> gpurun_out/bench30_$V.err; echo "bench $V rc=$?"
done
unset SHASTA_MI355X_LIBRARY
python - <<PY
import json
for v in ["base", "w5"]:
    d = json.loads(open("gpurun_out/bench30_%s.json" % v).read().strip().splitlines()[-1])
    print(v, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(x * 1e3, 1) for k, x in d["stage_seconds_per_step"].items()})
    for k, x in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
        if x["seconds_per_step"] > 0.012:
            print("   one worker: %-45s %7.2f ms/step  avg %8.3f ms" % (k, x["seconds_per_step"] * 1e3, x["avg_ms"]))
PY
