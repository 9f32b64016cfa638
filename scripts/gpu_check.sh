#!/bin/bash
# usage: scripts/gpu_check.sh <reads> <tag> [skiptests]
# GPU parity tests, a short bench and a kernel-trace profile.  Everything runs under its own
# `timeout` so that a hang cannot reach gpurun's limit.
READS=${1:-20000}; TAG=${2:-r}; SKIPTESTS=${3:-}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM, "$7" GiB available"}')"
if [ -z "$SKIPTESTS" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu --timeout 180 2>&1 | tail -6
fi
timeout 600 python bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"; tail -c 600 gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_$TAG | head
python - <<PY
import json
for f in ("gpurun_out/bench_$TAG.json", "gpurun_out/prof_$TAG.log"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'value', d['value'], 'ms/step', d['ms_per_step'], d['stage_seconds_per_step'], 'cand', d['config']['candidates'], d['roofline'])
        print(json.dumps(d['kernels']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
