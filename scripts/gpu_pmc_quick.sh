#!/bin/bash
# Two SQ counter passes (one aligner worker: kernels alone on the device) of the bench command -> gpurun_out/${ROUND}_${TAG}_pmc.json
ROUND=${ROUND:-r05}; TAG=${TAG:-quick}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload SHASTA_BENCH_DETAILS=/tmp/details_scratch.json
for PASS in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sq2:SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  T=${PASS%%:*}; COUNTERS=${PASS#*:}
  rm -rf $R/gpurun_out/pmcq_$T
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 400 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmcq_$T -o $T --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcq_$T.log 2>&1
  echo "pmc $T rc=$?"
done
cd $R
python scripts/pmc_summary.py 100000 gpurun_out/${ROUND}_${TAG}_pmc.json gpurun_out/pmcq_sq gpurun_out/pmcq_sq2 | cut -c1-900 | head -12
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete; find gpurun_out -name "*counter_collection.csv" -size +20M -delete
