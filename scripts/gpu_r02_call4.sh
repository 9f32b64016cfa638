#!/bin/bash
# Round 2, fourth GPU call: LDS geometry of the cells kernels (prebuilt variants shasta_amd/_build_{a..e}), tail tracebacks on the side stream.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in "" a b c d e; do
  if [ -z "$V" ]; then LIB=$GRAFT_REPO_ROOT/shasta_amd/_build/libshasta_mi355x.so; TAG=base; else LIB=$GRAFT_REPO_ROOT/shasta_amd/_build_$V/libshasta_mi355x.so; TAG=$V; fi
  export SHASTA_MI355X_LIBRARY=$LIB
  echo "== variant $TAG"
  timeout 600 python -m pytest tests/test_gpu_align4.py tests/test_gpu_adversarial.py -q -m gpu --timeout 300 -p no:cacheprovider -x 2>&1 | tail -2
  timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_v$TAG.json 2> gpurun_out/bench_v$TAG.err
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 python bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_v${TAG}_w1.json 2> gpurun_out/bench_v${TAG}_w1.err
done
unset SHASTA_MI355X_LIBRARY
python - <<PY
import json
for f in ["base", "a", "b", "c", "d", "e"]:
    for suffix in ("", "_w1"):
        try:
            d = json.loads(open("gpurun_out/bench_v%s%s.json" % (f, suffix)).read().strip().splitlines()[-1])
            k = d["kernels"]
            cells = {n: round(v["seconds_per_step"] * 1e3, 1) for n, v in k.items() if "Cells" in n}
            tb = {n: round(v["seconds_per_step"] * 1e3, 1) for n, v in k.items() if "Traceback" in n}
            print(f + suffix, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "align dev %.1f" % (d["stage_seconds_per_step"]["align4_device"] * 1e3), "kernel sum %.0f" % (d["kernel_seconds_per_step"] * 1e3), cells, tb)
        except Exception as e:
            print(f + suffix, "unreadable", e)
PY
