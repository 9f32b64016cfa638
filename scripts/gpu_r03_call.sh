cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -3
STEPS=4 WARMUP=2 PATTERN="NONE" bash scripts/gpu_ab.sh "cur||" 2>&1 | grep "==\|ms/step\|kernel s/step"
