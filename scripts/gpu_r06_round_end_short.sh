#!/bin/bash
# Round 6, the short form of the closing call (about a quarter of an hour of box time) for a GPU that comes back late:
#   1 the -m gpu suite   2 kernel-trace statistics of the bench command (six workers, one worker)   3 the bench line as the driver runs it
#   4 the ultra-long shape's line.  scripts/gpu_r06_round_end.sh is the long form (PMC passes, whole-list baselines, the other modes).
ROUND=r06
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider --durations=5 ) > gpurun_out/${ROUND}_final_suite.log 2>&1; tail -8 gpurun_out/${ROUND}_final_suite.log
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
( cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R SHASTA_BENCH_DETAILS=/tmp/details_scratch.json
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o m4 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
  cp $(find $R/gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${ROUND}_kernel_stats_100k_reads.csv
  python $R/scripts/kernel_timeline.py $(find $R/gpurun_out/prof_stats -name "*kernel_trace.csv" | head -1) 15 100 > $R/gpurun_out/${ROUND}_kernel_timeline_100k_reads.txt 2>&1
  SHASTA_MI355X_ALIGN_WORKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_w1 -o m4 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats_w1.log 2>&1
  cp $(find $R/gpurun_out/prof_stats_w1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${ROUND}_kernel_stats_100k_reads_one_worker.csv
  find $R/gpurun_out -name "*kernel_trace.csv" -size +20M -delete )
run() { local name=$1; shift; ( time SHASTA_BENCH_DETAILS=$R/gpurun_out/${ROUND}_${name}_details.json timeout 900 python bench.py "$@" > gpurun_out/${ROUND}_${name}.json 2> gpurun_out/${ROUND}_${name}.err ) 2>&1 | grep real; grep -v "^bench details: " gpurun_out/${ROUND}_${name}.err | tail -2; }
run bench_final_default
run bench_ul --workload ul --steps 5 --warmup 2 --baseline-sample 20000 --tie-census 0
python scripts/bench_summary.py gpurun_out/${ROUND}_bench_final_default gpurun_out/${ROUND}_bench_ul 2>&1 | cut -c1-400
head -12 gpurun_out/${ROUND}_kernel_stats_100k_reads_one_worker.csv | cut -c1-200
