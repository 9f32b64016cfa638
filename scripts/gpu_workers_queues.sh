#!/bin/bash
# workers x hardware queues (x batch size), one bench run each: the aligner's ms per call
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/workers_queues.log; : > $out
run() { # W Q BATCHLOG2
  line=$(GPU_MAX_HW_QUEUES=$2 SHASTA_MI355X_ALIGN_WORKERS=$1 SHASTA_MI355X_ALIGN_BATCH_LOG2=$3 SHASTA_BENCH_NO_GROUP_LINE=1 timeout 170 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "workers=$1 queues=$2 batch=2^$3 $(python - "$line" <<'P'
import json,sys
d=json.loads(sys.argv[1]); s=d["stage_device_ms_each_step"]
print("ms/step %.1f align %s mean %.1f" % (d["ms_per_step"], [round(x[1],1) for x in s], sum(x[1] for x in s)/len(s)))
P
)" >> $out
}
run 6 8 18
run 8 8 18
run 8 16 18
run 8 16 17
run 7 8 18
run 6 8 18
cat $out
