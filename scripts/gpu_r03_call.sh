cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -3
STEPS=4 WARMUP=2 PATTERN="Cells" bash scripts/gpu_ab.sh "cur||" 2>&1 | grep "==\|ms/step\|kernel s/step\|Cells"
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
SHASTA_MI355X_DEBUG=1 SHASTA_MI355X_ALIGN_WORKERS=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep "cells:" | head -8
