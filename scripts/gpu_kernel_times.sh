#!/bin/bash
# Average durations of the kernels whose names match KERNEL (a regex) in one bench step with one aligner worker, per build.
#   usage: KERNEL=Traceback scripts/gpu_kernel_times.sh "<tag>|<build dir under shasta_amd/>" ...
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload PYTHONPATH=$R
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for SPEC in "$@"; do
  TAG=${SPEC%%|*}; DIR=${SPEC#*|}
  rm -rf $R/gpurun_out/kt_$TAG
  SHASTA_MI355X_LIBRARY=$R/shasta_amd/${DIR:-_build}/libshasta_mi355x.so SHASTA_MI355X_ALIGN_WORKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$TAG -o kt --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/kt_$TAG.log 2>&1
  echo "== $TAG"
  python - <<PY
import csv, re
for r in csv.DictReader(open("$R/gpurun_out/kt_$TAG/kt_kernel_stats.csv")):
    if re.search(r"${KERNEL:-.}", r["Name"]):
        name = re.sub(r"\(.*", "", r["Name"]).replace("shasta_mi355x::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        print("   %-50s calls %5s  total %9.2f ms  avg %9.1f us" % (name[:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  find $R/gpurun_out/kt_$TAG -name "*kernel_trace.csv" -delete
done
