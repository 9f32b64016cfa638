#!/bin/bash
# Round 2, seventeenth GPU call: which kernels run beside which (kernel trace of the bench command, cut into steps).
READS=${1:-100000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
mkdir -p $R/gpurun_out
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_timeline -o tl --output-format csv -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_timeline.log 2>&1
echo "rocprof rc=$?"
cd $R
F=$(find gpurun_out/prof_timeline -name "*kernel_trace.csv" | head -1)
ls -la $F
python scripts/kernel_timeline.py $F 15 100 > gpurun_out/timeline.txt
cat gpurun_out/timeline.txt
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
