cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SHASTA_TEST_FIRST_GPU_RUN=1 timeout 900 python -m pytest tests/test_gpu_waiting_for_first_run.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -15
bash scripts/gpu_ab.sh "shipped_wl1k1||" "wl0k0|_build_wl0k0|" "wl1k0|_build_wl1k0|" "wl1k2|_build_wl1k2|" "wl0k1|_build_wl0k1|" "devprep||SHASTA_MI355X_DEVICE_BATCH_PREP=1"
