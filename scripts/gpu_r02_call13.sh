#!/bin/bash
# Round 2, thirteenth GPU call: cells graph adjacency through a cell map (eight lookups per kept cell), the stream dealt to the
# wavefronts in groups of 64 markers; scalar-store microbenchmark for the forward DP's trace records.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 shasta_amd/_build/trace_store > gpurun_out/trace_store.jsonl 2> gpurun_out/trace_store.err; echo "trace_store rc=$?"; cat gpurun_out/trace_store.jsonl; tail -3 gpurun_out/trace_store.err
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for V in base w3 w2; do
  if [ $V = base ]; then unset SHASTA_MI355X_LIBRARY; else export SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_$V/libshasta_mi355x.so; fi
  timeout 300 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_assembler_mirror.py -q -m gpu -x --timeout 200 -p no:cacheprovider 2>&1 | tail -1
  timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench13_$V.json 2> gpurun_out/bench13_$V.err; echo "bench $V rc=$?"
done
SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_prof/libshasta_mi355x.so SHASTA_MI355X_ALIGN_WORKERS=1 timeout 600 python bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench13_prof.json 2> gpurun_out/bench13_prof.err; echo "phase profile rc=$?"; grep "phase cycles" gpurun_out/bench13_prof.err | tail -1
unset SHASTA_MI355X_LIBRARY
python - <<PY
import json
for v in "base w3 w2".split():
    try:
        d = json.loads(open("gpurun_out/bench13_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "align4_device %.1f" % (d["stage_seconds_per_step"]["align4_device"] * 1e3))
        for k, x in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            if x["seconds_per_step"] > 0.004:
                print("   one worker: %-45s %7.2f ms/step  avg %8.3f ms" % (k, x["seconds_per_step"] * 1e3, x["avg_ms"]))
    except Exception as e:
        print(v, "unreadable", e)
PY
