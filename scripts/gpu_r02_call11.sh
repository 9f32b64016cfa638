#!/bin/bash
# Round 2, eleventh GPU call: cells geometry variants after the cooperative mode (which chunks run cooperatively,
# wavefronts per cooperative workgroup, partner markers per lane and round in classes 1 and 2).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in base coopall q2 coopall_q2 coopall_q2_w8 q2_w8; do
  if [ $V = base ]; then unset SHASTA_MI355X_LIBRARY; else export SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_$V/libshasta_mi355x.so; fi
  timeout 300 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_assembler_mirror.py -q -m gpu -x --timeout 200 -p no:cacheprovider 2>&1 | tail -1
  timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench11_$V.json 2> gpurun_out/bench11_$V.err; echo "bench $V rc=$?"
done
unset SHASTA_MI355X_LIBRARY
python - <<PY
import json
for v in "base coopall q2 coopall_q2 coopall_q2_w8 q2_w8".split():
    try:
        d = json.loads(open("gpurun_out/bench11_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "align4_device %.1f" % (d["stage_seconds_per_step"]["align4_device"] * 1e3))
        cells = 0.0
        for k, x in sorted(d["kernels_one_worker"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            if "Cells" in k:
                cells += x["seconds_per_step"]
                print("   one worker: %-45s %7.2f ms/step  avg %8.3f ms" % (k, x["seconds_per_step"] * 1e3, x["avg_ms"]))
        print("   one worker cells total %.1f ms/step" % (cells * 1e3))
    except Exception as e:
        print(v, "unreadable", e)
PY
