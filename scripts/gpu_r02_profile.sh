#!/bin/bash
# The round's measurement call: suite, the bench line as the driver runs it (with the CPU baseline + parity at bench size),
# kernel-trace stats of the same command, the four PMC passes (one aligner worker: per-kernel counters without overlap),
# and the lines of the other modes (LowHash0 only = configs[1], align method 3, marker finding).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM"}')"
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -5
( time timeout 1500 python bench.py --reads $READS --steps 5 --warmup 2 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err ) 2>&1 | grep real
timeout 900 python bench.py --reads $READS --steps 5 --warmup 2 --no-cpu-baseline --lowhash-only > gpurun_out/bench_final_lh.json 2> gpurun_out/bench_final_lh.err
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline --align-method 3 > gpurun_out/bench_final_m3.json 2> gpurun_out/bench_final_m3.err
timeout 900 python bench.py --reads 20000 --steps 3 --warmup 1 --markers > gpurun_out/bench_final_markers.json 2> gpurun_out/bench_final_markers.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
echo "rocprof stats rc=$?"
for PASS in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sq2:SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  TAG=${PASS%%:*}; COUNTERS=${PASS#*:}
  rm -rf $R/gpurun_out/pmc_$TAG $R/gpurun_out/pmc_${TAG}_cal
  if [ "$TAG" = fetch ] || [ "$TAG" = write ]; then
    timeout 300 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_${TAG}_cal -o cal --output-format csv -- python $R/scripts/calibrate_pmc.py > $R/gpurun_out/pmc_${TAG}_cal.log 2>&1
  fi
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_$TAG -o $TAG --output-format csv -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_$TAG.log 2>&1
  echo "pmc $TAG rc=$?"
done
cd $R
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
python - <<PY
import json
for f in ["bench_final", "bench_final_lh", "bench_final_m3", "bench_final_markers"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["metric"], "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d.get("stage_seconds_per_step"))
        if f == "bench_final":
            print("   cpu_baseline", json.dumps(d["cpu_baseline"])[:500]); print("   parity", d["parity_at_bench_size"]); print("   roofline", json.dumps(d["roofline"])[:500])
    except Exception as e:
        print(f, "unreadable", e)
PY
