#!/bin/bash
# The first GPU call of a round (round 4 wrote it and never got a box: its three calls ended with lease faults before the command ran):
# the whole -m gpu suite -- round 4's additions have only met the emulated build: configs[0]/[3]/[4] parameter values, the alignment
# table, the sparse form of the banded alignment and its anchor kernel, the DP preparation by a counting pass -- then the bench line with
# the step = computeAlignments end to end: the default, without the anchor kernel (SHASTA_MI355X_ANCHORED_DP=0) and without the sparse path
# (SHASTA_MI355X_SPARSE_DP=0: round 3's DP), same box, same reads.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'ROUND=r05 bash scripts/gpu_first_call.sh'
# Then, for the round's profile: ROUND=r05 bash scripts/gpu_profile.sh (PMC passes, kernel stats, timeline, the other modes).
ROUND=${ROUND:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM"}'), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider --durations=6 ) 2>&1 | tail -24
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
( time timeout 900 python bench.py --steps 6 --warmup 2 --baseline-sample 12000 --tie-census 0 > gpurun_out/${ROUND}_bench_a.json 2> gpurun_out/${ROUND}_bench_a.err ) 2>&1 | grep real
( time SHASTA_MI355X_ANCHORED_DP=0 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${ROUND}_bench_a_no_anchors.json 2> gpurun_out/${ROUND}_bench_a_no_anchors.err ) 2>&1 | grep real
( time SHASTA_MI355X_SPARSE_DP=0 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${ROUND}_bench_a_dense.json 2> gpurun_out/${ROUND}_bench_a_dense.err ) 2>&1 | grep real
ROUND=$ROUND python - <<'PY'
import json
import os
for name in (os.environ.get("ROUND", "r05") + "_bench_a", os.environ.get("ROUND", "r05") + "_bench_a_no_anchors", os.environ.get("ROUND", "r05") + "_bench_a_dense"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "unreadable", e); continue
    print("==", name, "ms/step %.1f value %.0f" % (d["ms_per_step"], d["value"]), {k: (round(v * 1e3, 2) if v is not None else None) for k, v in d["stage_seconds_per_step"].items()})
    if d.get("path"):
        print("!!", d["path"], "after", d.get("earlier_attempts"))
    print("each", d["stage_device_ms_each_step"])
    if "cpu_baseline" in d:
        print("cpu", json.dumps(d["cpu_baseline"])[:1400]); print("parity", d["parity_at_bench_size"])
    print("banded_dp", d.get("banded_dp"))
    solo = d.get("kernels_one_worker") or {}
    print("kernel s/step: in step %.1f ms, solo %.1f ms" % (1e3 * d["kernel_seconds_per_step"], 1e3 * sum(v["seconds_per_step"] for v in solo.values())))
    for k, v in sorted(solo.items(), key=lambda kv: -kv[1]["seconds_per_step"]):
        s = d["kernels"].get(k, {})
        if v["seconds_per_step"] > 0.0005:
            print("   %-52s solo %7.2f ms/step avg %7.3f ms %s| in step %7.2f ms/step" % (k, v["seconds_per_step"] * 1e3, v["avg_ms"], ("%6.0f GCUPS " % v["gcups"]) if "gcups" in v else "", s.get("seconds_per_step", 0) * 1e3))
PY
tail -3 gpurun_out/${ROUND}_bench_a.err gpurun_out/${ROUND}_bench_a_no_anchors.err gpurun_out/${ROUND}_bench_a_dense.err
