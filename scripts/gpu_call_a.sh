#!/bin/bash
# Round 4, first call: the new parity tests (configs[0]/[3]/[4] values, the alignment table) on the MI355X, then the bench line with
# the step = computeAlignments end to end and the CPU leg's new parts at a reduced sample.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM"}'), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
( time timeout 1200 python -m pytest tests/test_gpu_config_values.py tests/test_gpu_base_level_reads.py tests/test_gpu_host_stages.py -q -m gpu --timeout 900 -p no:cacheprovider --durations=8 ) 2>&1 | tail -22
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
( time timeout 900 python bench.py --steps 6 --warmup 2 --baseline-sample 12000 --tie-census 0 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_a.json").read().strip().splitlines()[-1])
print("ms/step %.1f value %.0f" % (d["ms_per_step"], d["value"]), {k: (round(v * 1e3, 2) if v is not None else None) for k, v in d["stage_seconds_per_step"].items()})
print("each", d["stage_device_ms_each_step"])
print("cpu", json.dumps(d["cpu_baseline"])[:1200])
print("parity", d["parity_at_bench_size"])
solo = d.get("kernels_one_worker") or {}
for k, v in sorted(solo.items(), key=lambda kv: -kv[1]["seconds_per_step"]):
    s = d["kernels"].get(k, {})
    print("   %-52s solo %7.2f ms/step avg %7.3f ms | in step %7.2f ms/step" % (k, v["seconds_per_step"] * 1e3, v["avg_ms"], s.get("seconds_per_step", 0) * 1e3))
PY
tail -5 gpurun_out/r04_bench_a.err
