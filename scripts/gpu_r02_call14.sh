#!/bin/bash
# Round 2, fourteenth GPU call: cells variants (adjacency by cell map or by comparing all pairs, stream groups interleaved or not,
# 5 wavefronts per SIMD forced) at 3 wavefronts per workgroup; forward DP with the gap penalties folded into the stored scores.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for V in base A B C D E F; do
  if [ $V = base ]; then unset SHASTA_MI355X_LIBRARY; else export SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_$V/libshasta_mi355x.so; fi
  timeout 300 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_assembler_mirror.py -q -m gpu -x --timeout 200 -p no:cacheprovider 2>&1 | tail -1
  timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench14_$V.json 2> gpurun_out/bench14_$V.err; echo "bench $V rc=$?"
done
unset SHASTA_MI355X_LIBRARY
python - <<PY
import json
for v in "base A B C D E F".split():
    try:
        d = json.loads(open("gpurun_out/bench14_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "align4_device %.1f" % (d["stage_seconds_per_step"]["align4_device"] * 1e3))
        for k, x in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            if x["seconds_per_step"] > 0.009:
                print("   one worker: %-45s %7.2f ms/step  avg %8.3f ms" % (k, x["seconds_per_step"] * 1e3, x["avg_ms"]))
    except Exception as e:
        print(v, "unreadable", e)
PY
