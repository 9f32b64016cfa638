#!/bin/bash
# The round's last call, on the tree as committed: the -m gpu suite, smoke(), the default bench line (with the reference baseline,
# the parity block and the tie census), kernel-trace stats + timeline of the same command, the N-rank branch with one rank over
# RCCL, the in-process group over one device.  The PMC passes (one aligner worker; scripts/gpu_profile.sh) are not repeated:
# no kernel changed since.
ROUND=${ROUND:-r03}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 600 python bench.py > gpurun_out/end_bench.json 2> gpurun_out/end_bench.err ) 2>&1 | grep real
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/end_prof -o m4 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/end_prof.log 2>&1
echo "rocprof stats rc=$?"
cd $R
python scripts/kernel_timeline.py $(find gpurun_out/end_prof -name "*kernel_trace.csv" | head -1) 15 100 > gpurun_out/end_timeline.txt 2>&1
cp $(find gpurun_out/end_prof -name "*kernel_stats.csv" | head -1) gpurun_out/end_kernel_stats.csv
find gpurun_out/end_prof -name "*kernel_trace.csv" -delete
SHASTA_BENCH_FORCE_SHARDED=1 SHASTA_BENCH_NO_GROUP_LINE=1 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/end_one_rank.json 2> gpurun_out/end_one_rank.err
timeout 300 python bench.py --steps 3 --warmup 2 --group --gpus 1 > gpurun_out/end_group1.json 2> gpurun_out/end_group1.err
python - <<PY
import json
for f in ["end_bench", "end_one_rank", "end_group1"]:
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d.get("stage_seconds_per_step"), d.get("stage_device_ms_each_step"))
        if f == "end_bench":
            print("   cpu_baseline", json.dumps(d["cpu_baseline"])[:400]); print("   parity", d["parity_at_bench_size"]); print("   roofline", json.dumps(d["roofline"])[:600])
            print("   dp_tie_sensitive", json.dumps({k: v for k, v in d["dp_tie_sensitive"].items() if k != "per_policy"}))
        if f == "end_group1":
            print("  ", d["in_process_group"])
    except Exception as e:
        print(f, "unreadable", e)
PY
