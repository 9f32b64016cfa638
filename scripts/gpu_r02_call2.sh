#!/bin/bash
# Round 2, second GPU call: suite after the kernel-table refactoring, the number of aligner workers (streams), the
# bench line with the full-size CPU baseline + parity, kernel-trace stats.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -8
for W in 1 2 3 4 6; do
  SHASTA_MI355X_ALIGN_WORKERS=$W timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_w$W.json 2> gpurun_out/bench_w$W.err
  echo "workers $W rc=$?"; tail -c 200 gpurun_out/bench_w$W.err
done
( time timeout 1500 python bench.py --reads $READS --steps 3 --warmup 1 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2>&1 | grep real
echo "bench with cpu baseline rc=$?"; tail -c 300 gpurun_out/bench_full.err
python - <<PY
import json
for f in ["bench_w1", "bench_w2", "bench_w3", "bench_w4", "bench_w6", "bench_full"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["stage_seconds_per_step"], "kernel s/step %.3f" % d["kernel_seconds_per_step"], d.get("aligner_status"))
        if f in ("bench_w2", "bench_full"):
            for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["seconds_per_step"]):
                print("   %-55s %7.2f ms/step  %6.1f launches  avg %8.3f ms  %7.1f GB/s" % (k, v["seconds_per_step"] * 1e3, v["launches_per_step"], v["avg_ms"], v["achieved_GBps"]))
            print("   roofline", json.dumps(d["roofline"])[:600])
        if f == "bench_full":
            print("   cpu_baseline", json.dumps(d["cpu_baseline"])); print("   parity", d["parity_at_bench_size"]); print("   pcie", d.get("pcie_inclusive"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2 -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof2.log 2>&1
echo "rocprof stats rc=$?"
cd $R
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
