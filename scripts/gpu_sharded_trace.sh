#!/bin/bash
# The N-rank branch's intermittent LowHash0 stall (open at the end of round 3): a kernel trace of the
# branch with one rank over RCCL, and for every slow lh_buckets_all call (SHASTA_MI355X_LOG_STAGES=1 names them on stderr)
# the kernels of the library's stream around it -- is the first copy after the exchange late (a gap before it) or long?
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
SHASTA_MI355X_LOG_STAGES=1 SHASTA_BENCH_FORCE_SHARDED=1 SHASTA_BENCH_NO_GROUP_LINE=1 timeout ${LIMIT:-280} rocprofv3 --kernel-trace -d $R/gpurun_out/sharded_trace -o t --output-format csv -- \
  python $R/bench.py --steps ${STEPS:-30} --warmup 2 --no-cpu-baseline > $R/gpurun_out/sharded_trace.json 2> $R/gpurun_out/sharded_trace.err
cd $R
grep "lh_buckets_all took" gpurun_out/sharded_trace.err | cut -c1-300
python - <<'P'
import csv, glob, json
d = json.loads(open("gpurun_out/sharded_trace.json").read().strip().splitlines()[-1])
print("LowHash0 wall ms per step:", [x[0] for x in d["stage_device_ms_each_step"]])
f = glob.glob("gpurun_out/sharded_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# The copies of the received records are the launches right after RCCL's kernel; print the gaps > 10 ms between consecutive
# kernels of the whole device (nothing in flight) and the 3 kernels either side of each.
end = 0
for k, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end and s - end > 10e6 or e - s > 10e6:
        print("--- gap %.1f ms before / duration %.1f ms of %s" % ((s - end) / 1e6, (e - s) / 1e6, r["Kernel_Name"][:80]))
        for q in rows[max(0, k - 3):k + 3]:
            print("      %-70s start %.3f ms  duration %.3f ms  queue %s" % (q["Kernel_Name"][:70], (int(q["Start_Timestamp"]) - s) / 1e6, (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e6, q.get("Queue_Id")))
    end = max(end, e)
P
find gpurun_out/sharded_trace -name "*kernel_trace.csv" -size +20M -delete
