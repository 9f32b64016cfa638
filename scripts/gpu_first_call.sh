#!/bin/bash
# usage: scripts/gpu_first_call.sh [reads]   (ONE gpurun call, about 15 GPU-minutes at 100k reads)
# The first thing to run when a GPU is available again: everything that was written while the GPU was
# closed has only run on the emulated build.  In order, each step under its own
# timeout so that a hang cannot reach gpurun's limit:
#   1. the -m gpu suite file by file (validated files first, the new ones last), so that one failure
#      does not hide the rest;
#   2. bench.py for BASELINE's metric (method 4), then the comparison line for method 3;
#   3. rocprofv3 kernel-trace stats of both (copy the CSVs you keep into profiles/).
READS=${1:-100000}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM, "$7" GiB available"}')"
for f in tests/test_gpu_lowhash0.py tests/test_gpu_align4.py tests/test_gpu_host_stages.py tests/test_gpu_distributed.py \
         tests/test_gpu_zz_assembler_mirror.py tests/test_gpu_zzz_align3.py  tests/test_gpu_zzz_palindromic.py tests/test_gpu_zzzz_large_properties.py tests/test_gpu_zzzzz_kernel_versions.py; do
  echo "== $f"
  timeout 900 python -m pytest $f -q -m gpu --timeout 300 2>&1 | tail -4
done
# Kernel-level A/B of the two forward DP kernels: HIP-event times per band class on the same synthetic tasks.
for V in 1 2; do
  SHASTA_MI355X_DP_FORWARD=$V timeout 600 python scripts/dp_microbench.py --tasks 40000 --repeat 3 > gpurun_out/dp_microbench_v$V.jsonl 2> gpurun_out/dp_microbench_v$V.err
  echo "dp microbench version $V rc=$?"; tail -n 1 gpurun_out/dp_microbench_v$V.jsonl
done
# A/B of the two forward DP kernels (DESIGN.md section 4, K10b'): method 4 with the first version forced.
SHASTA_MI355X_DP_FORWARD=1 timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_m4_dp1.json 2> gpurun_out/bench_m4_dp1.err
echo "bench method 4, first forward kernel rc=$?"; tail -c 300 gpurun_out/bench_m4_dp1.err
# A/B of the two window-hash kernels (LowHash0 only): the version without shared block transforms.
SHASTA_MI355X_HASH=1 timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --lowhash-only --no-cpu-baseline > gpurun_out/bench_lh_hash1.json 2> gpurun_out/bench_lh_hash1.err
timeout 600 python bench.py --reads $READS --steps 3 --warmup 1 --lowhash-only --no-cpu-baseline > gpurun_out/bench_lh.json 2> gpurun_out/bench_lh.err
python - <<PY
import json
for f in ("gpurun_out/bench_lh_hash1.json", "gpurun_out/bench_lh.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "hash kernel", d["kernels"]["hashWindowsKernel<4>"])
    except Exception as e:
        print(f, "unreadable", e)
PY
for M in 4 3; do
  timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --align-method $M --no-cpu-baseline > gpurun_out/bench_m$M.json 2> gpurun_out/bench_m$M.err
  echo "bench method $M rc=$?"; tail -c 400 gpurun_out/bench_m$M.err
done
cd /tmp && export TMPDIR=/tmp
for M in 4 3; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_m$M -o m$M --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --reads $READS --steps 2 --warmup 1 --align-method $M --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_m$M.log 2>&1
  echo "rocprof method $M rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<PY
import json
for f in ("gpurun_out/bench_m4_dp1.json", "gpurun_out/bench_m4.json", "gpurun_out/bench_m3.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['metric'], 'value', d['value'], 'ms/step', d['ms_per_step'], d['stage_seconds_per_step'], 'cand', d['config']['candidates'], 'stored', d['config']['alignments_stored'])
        print(json.dumps(d['kernels']))
    except Exception as e:
        print(f, 'unreadable', e)
PY

# Last, because it rebuilds the library: the direct-grid cell counter (compile-time experiment, DESIGN.md section 8;
# align4_cells.hpp, SHASTA_CELLS_GRID).  Parity first, then the same bench line; the default build is restored after.
make -s -C shasta_amd/csrc clean && make -s -C shasta_amd/csrc EXTRA=-DSHASTA_CELLS_GRID=1 > gpurun_out/build_grid.log 2>&1
if [ -f shasta_amd/_build/libshasta_mi355x.so ]; then
  timeout 900 python -m pytest tests/test_gpu_align4.py -q -m gpu --timeout 300 2>&1 | tail -3
  timeout 900 python bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_m4_grid.json 2> gpurun_out/bench_m4_grid.err
  echo "bench method 4, grid cell counter rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_m4_grid.json").read().strip().splitlines()[-1])
    print("grid cell counter: value", d["value"], "ms/step", d["ms_per_step"], d["stage_seconds_per_step"])
except Exception as e:
    print("gpurun_out/bench_m4_grid.json unreadable", e)
PY
fi
make -s -C shasta_amd/csrc clean && make -s -C shasta_amd/csrc > gpurun_out/build_default.log 2>&1
