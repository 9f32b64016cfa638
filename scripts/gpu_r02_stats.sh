#!/bin/bash
# Kernel-trace statistics and timeline of the bench command (the part of the measurement call that timed out there).
READS=${1:-100000}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/prof_final
timeout -k 10 270 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o m4 --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
echo "rocprof stats rc=$?"
cd $R
F=$(find gpurun_out/prof_final -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $F 15 100 > gpurun_out/timeline_final.txt 2>&1
python scripts/kernel_stats_summary.py $(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1) 4 | head -14
head -6 gpurun_out/timeline_final.txt
tail -2 gpurun_out/prof_final.log | cut -c1-300
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
