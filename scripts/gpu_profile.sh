#!/bin/bash
# A round's measurement call (ROUND=r03 bash scripts/gpu_profile.sh; round 2's was scripts/gpu_r02_profile.sh, in the git history).  Order matters: the PMC passes come first and their per-kernel summary is written to
# profiles/${ROUND}_pmc_100k_reads.json ON THE BOX, so that the bench line that follows takes its roofline's `traffic` and VALU
# instruction counts from counters collected on the same build in the same session.
#   1 suite   2 PMC passes (one aligner worker: per-kernel counters without overlap) + summary   3 the bench line as the driver
#   runs it (with the CPU baseline + parity at bench size)   4 kernel-trace stats + timeline of the same command
#   5 the lines of the other modes (LowHash0 only = configs[1], align method 3, marker finding)
READS=${1:-100000}
ROUND=${ROUND:-r05}          # prefix of the files under profiles/ (bench.py reads the newest rNN_pmc_100k_reads.json)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM"}')"
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
for PASS in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sq2:SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  TAG=${PASS%%:*}; COUNTERS=${PASS#*:}
  rm -rf $R/gpurun_out/pmc_$TAG $R/gpurun_out/pmc_${TAG}_cal
  if [ "$TAG" = fetch ] || [ "$TAG" = write ]; then
    timeout 300 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_${TAG}_cal -o cal --output-format csv -- python $R/scripts/calibrate_pmc.py > $R/gpurun_out/pmc_${TAG}_cal.log 2>&1
  fi
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 400 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmc_$TAG -o $TAG --output-format csv -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_$TAG.log 2>&1
  echo "pmc $TAG rc=$?"
done
cd $R
python scripts/pmc_summary.py $READS profiles/${ROUND}_pmc_100k_reads.json gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch_cal gpurun_out/pmc_write_cal > gpurun_out/pmc_summary.log 2>&1; echo "pmc summary rc=$?"
cp profiles/${ROUND}_pmc_100k_reads.json gpurun_out/${ROUND}_pmc_100k_reads.json
( time timeout 1500 python bench.py --reads $READS --steps 5 --warmup 2 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err ) 2>&1 | grep real
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
echo "rocprof stats rc=$?"
cd $R
python scripts/kernel_timeline.py $(find gpurun_out/prof_final -name "*kernel_trace.csv" | head -1) 15 100 > gpurun_out/timeline_final.txt 2>&1
timeout 900 python bench.py --reads $READS --steps 5 --warmup 2 --no-cpu-baseline --lowhash-only > gpurun_out/bench_final_lh.json 2> gpurun_out/bench_final_lh.err
timeout 900 python bench.py --reads $READS --steps 4 --warmup 3 --no-cpu-baseline --align-method 3 > gpurun_out/bench_final_m3.json 2> gpurun_out/bench_final_m3.err
timeout 900 python bench.py --reads 20000 --steps 3 --warmup 1 --markers > gpurun_out/bench_final_markers.json 2> gpurun_out/bench_final_markers.err
SHASTA_MI355X_ALIGN_WORKERS=1 timeout 900 python bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_final_w1.json 2> gpurun_out/bench_final_w1.err
# The in-process group over one device (the seam a C++ caller uses), and the reference aligner on the WHOLE candidate list.
timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --group --gpus 1 > gpurun_out/bench_final_group1.json 2> gpurun_out/bench_final_group1.err
( time timeout 1500 python bench.py --reads $READS --steps 2 --warmup 1 --baseline-sample 0 --tie-census 0 > gpurun_out/bench_final_whole_baseline.json 2> gpurun_out/bench_final_whole_baseline.err ) 2>&1 | grep real
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
python - <<PY
import json
for f in ["bench_final", "bench_final_lh", "bench_final_m3", "bench_final_markers", "bench_final_w1", "bench_final_group1", "bench_final_whole_baseline"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["metric"], "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d.get("stage_seconds_per_step"))
        if f == "bench_final":
            print("   cpu_baseline", json.dumps(d["cpu_baseline"])[:500]); print("   parity", d["parity_at_bench_size"]); print("   roofline", json.dumps(d["roofline"])[:900])
            print("   dp_tie_sensitive", json.dumps({k: v for k, v in d["dp_tie_sensitive"].items() if k != "per_policy"}))
        if f == "bench_final_whole_baseline":
            print("   cpu_baseline", json.dumps(d["cpu_baseline"])[:700]); print("   parity", d["parity_at_bench_size"])
    except Exception as e:
        print(f, "unreadable", e)
PY
