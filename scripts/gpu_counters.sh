#!/bin/bash
# Kernel-trace statistics + the four PMC passes (each its own run, one aligner worker so kernels do not overlap) of the bench command
# on the build in the tree; summaries written under gpurun_out/ for copying to profiles/${ROUND}_*.
ROUND=${ROUND:-r05}
READS=${READS:-100000}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload SHASTA_BENCH_DETAILS=/tmp/details_scratch.json
rm -rf $R/gpurun_out/prof_stats
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
echo "rocprof stats rc=$?"
cp $(find $R/gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${ROUND}_kernel_stats_100k_reads.csv
python $R/scripts/kernel_timeline.py $(find $R/gpurun_out/prof_stats -name "*kernel_trace.csv" | head -1) 15 100 > $R/gpurun_out/${ROUND}_kernel_timeline_100k_reads.txt 2>&1
rm -rf $R/gpurun_out/prof_stats_w1
SHASTA_MI355X_ALIGN_WORKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_w1 -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats_w1.log 2>&1
cp $(find $R/gpurun_out/prof_stats_w1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${ROUND}_kernel_stats_100k_reads_one_worker.csv
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
python scripts/pmc_summary.py $READS gpurun_out/${ROUND}_pmc_100k_reads.json gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch_cal gpurun_out/pmc_write_cal > gpurun_out/pmc_summary.log 2>&1; echo "pmc summary rc=$?"
cat gpurun_out/pmc_summary.log | head -40
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out -name "*counter_collection.csv" -size +20M -delete
