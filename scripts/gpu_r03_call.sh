cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
SHASTA_MI355X_LIBRARY=$GRAFT_REPO_ROOT/shasta_amd/_build_prof/libshasta_mi355x.so SHASTA_MI355X_ALIGN_WORKERS=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
tail -5 gpurun_out/prof_bench.err | cut -c1-2000
