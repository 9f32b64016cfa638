#!/bin/bash
# SQ counters of two builds of the library over one bench step with one aligner worker (kernels do not overlap), per kernel.
#   usage: scripts/gpu_pmc_ab.sh "<tag>|<build dir under shasta_amd/>" ...        env: KERNEL (substring of the rows to print)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload PYTHONPATH=$R
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1      # fills the workload cache
cd /tmp && export TMPDIR=/tmp
for SPEC in "$@"; do
  TAG=${SPEC%%|*}; DIR=${SPEC#*|}
  for PASS in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sq2:SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD"; do
    P=${PASS%%:*}; COUNTERS=${PASS#*:}
    rm -rf $R/gpurun_out/pmcab_${TAG}_$P
    SHASTA_MI355X_LIBRARY=$R/shasta_amd/${DIR:-_build}/libshasta_mi355x.so SHASTA_MI355X_ALIGN_WORKERS=1 timeout 300 rocprofv3 --pmc $COUNTERS --kernel-trace -d $R/gpurun_out/pmcab_${TAG}_$P -o $P --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcab_${TAG}_$P.log 2>&1
    echo "pmc $TAG $P rc=$?"
  done
  python $R/scripts/pmc_summary.py 100000 $R/gpurun_out/pmcab_$TAG.json $R/gpurun_out/pmcab_${TAG}_sq $R/gpurun_out/pmcab_${TAG}_sq2 | grep "${KERNEL:-Kernel}"
  find $R/gpurun_out/pmcab_${TAG}_sq $R/gpurun_out/pmcab_${TAG}_sq2 -name "*.csv" -size +5M -delete
done
