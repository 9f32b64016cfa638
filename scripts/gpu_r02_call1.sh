#!/bin/bash
# Round 2, first GPU call: the whole -m gpu suite (file by file, no -x: one failure must not hide the rest), the bench
# line, kernel-trace stats and the PMC passes of the bench command as it ships, instruction-rate microbenchmark,
# A/B of the two window-hash kernels and of the direct-grid cell counter (prebuilt in shasta_amd/_build_grid).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM, "$7" GiB available"}')"
rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk" | head -4
for f in tests/test_gpu_*.py; do
  echo "== $f"
  timeout 900 python -m pytest $f -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -6
done
shasta_amd/_build/valu_rates > gpurun_out/valu_rates.jsonl 2>&1; cat gpurun_out/valu_rates.jsonl
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_m4.json 2> gpurun_out/bench_m4.err
echo "bench rc=$?"; tail -c 300 gpurun_out/bench_m4.err
SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_grid/libshasta_mi355x.so timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_m4_grid.json 2> gpurun_out/bench_m4_grid.err
echo "bench grid rc=$?"
SHASTA_MI355X_HASH=1 timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --lowhash-only --no-cpu-baseline > gpurun_out/bench_lh_hash1.json 2> gpurun_out/bench_lh_hash1.err
timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --lowhash-only --no-cpu-baseline > gpurun_out/bench_lh.json 2> gpurun_out/bench_lh.err
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --align-method 3 --no-cpu-baseline > gpurun_out/bench_m3.json 2> gpurun_out/bench_m3.err
python - <<PY
import json
for f in ("bench_m4", "bench_m4_grid", "bench_lh_hash1", "bench_lh", "bench_m3"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "ms/step", d["ms_per_step"], d["stage_seconds_per_step"], "cand", d["config"]["candidates"], "stored", d["config"]["alignments_stored"])
        print(json.dumps(d["kernels"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m4 -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_m4.log 2>&1
echo "rocprof stats rc=$?"
export PYTHONPATH=$R
for PASS in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sq2:SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  TAG=${PASS%%:*}; COUNTERS=${PASS#*:}
  if [ "$TAG" = fetch ] || [ "$TAG" = write ]; then
    timeout 300 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_${TAG}_cal -o cal --output-format csv -- python $R/scripts/calibrate_pmc.py > $R/gpurun_out/pmc_${TAG}_cal.log 2>&1
    echo "calibration $TAG rc=$?"
  fi
  timeout 900 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_$TAG -o $TAG --output-format csv -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_$TAG.log 2>&1
  echo "pmc $TAG rc=$?"
done
cd $R
find gpurun_out -name "*.csv" -size +20M -delete      # kernel traces of long runs: keep the merge under the limit
du -sh gpurun_out | tail -1
