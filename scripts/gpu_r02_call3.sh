#!/bin/bash
# Round 2, third GPU call: traceback without metrics / stores in the walk, 6 workers by default, parallel assembly;
# batch-size and worker sweeps; PMC passes for the current kernels.
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -5
run() {  # tag, env...
  TAG=$1; shift
  env "$@" timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
  echo "$TAG rc=$?"
}
run default
run w8 SHASTA_MI355X_ALIGN_WORKERS=8
run w4 SHASTA_MI355X_ALIGN_WORKERS=4
run b16 SHASTA_MI355X_ALIGN_BATCH_LOG2=16
run b15w8 SHASTA_MI355X_ALIGN_BATCH_LOG2=15 SHASTA_MI355X_ALIGN_WORKERS=8
run b18 SHASTA_MI355X_ALIGN_BATCH_LOG2=18
python - <<PY
import json
for f in ["default", "w8", "w4", "b16", "b15w8", "b18"]:
    try:
        d = json.loads(open("gpurun_out/bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["stage_seconds_per_step"], "kernel s/step %.3f" % d["kernel_seconds_per_step"])
        if f == "default":
            for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["seconds_per_step"])[:16]:
                print("   %-55s %7.2f ms/step  %6.1f launches  avg %8.3f ms  %7.1f GB/s" % (k, v["seconds_per_step"] * 1e3, v["launches_per_step"], v["avg_ms"], v["achieved_GBps"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for PASS in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sq2:SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  TAG=${PASS%%:*}; COUNTERS=${PASS#*:}
  rm -rf $R/gpurun_out/pmc_$TAG $R/gpurun_out/pmc_${TAG}_cal
  if [ "$TAG" = fetch ] || [ "$TAG" = write ]; then
    timeout 300 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_${TAG}_cal -o cal --output-format csv -- python $R/scripts/calibrate_pmc.py > $R/gpurun_out/pmc_${TAG}_cal.log 2>&1
  fi
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_$TAG -o $TAG --output-format csv -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_$TAG.log 2>&1
  echo "pmc $TAG rc=$?"
done
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3 -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof3.log 2>&1
echo "rocprof stats rc=$?"
cd $R
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
