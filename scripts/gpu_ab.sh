#!/bin/bash
# A/B of builds and switches in ONE GPU call, same box, same read set (cached in /tmp for the call).
#   usage: scripts/gpu_ab.sh "<tag>|<build dir under shasta_amd/, empty = _build>|<ENV=V ...>" ...
#   env:   TIMEOUT (600 s per variant) STEPS (4) WARMUP (2) READS (100000) PATTERN (regex of kernel rows to print; default: every kernel above 2 ms solo)
# Per variant: the bench line (no CPU baseline) -> gpurun_out/ab_<tag>.json, one summary row per kernel (in the step and solo).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SHASTA_BENCH_WORKLOAD_CACHE=/tmp/shasta_workload
for SPEC in "$@"; do
  TAG=$(echo "$SPEC" | cut -d'|' -f1); DIR=$(echo "$SPEC" | cut -d'|' -f2); ENVS=$(echo "$SPEC" | cut -d'|' -f3)
  LIB=$GRAFT_REPO_ROOT/shasta_amd/${DIR:-_build}/libshasta_mi355x.so
  env $ENVS SHASTA_MI355X_LIBRARY=$LIB timeout ${TIMEOUT:-600} python bench.py --reads ${READS:-100000} --steps ${STEPS:-4} --warmup ${WARMUP:-2} --no-cpu-baseline > gpurun_out/ab_$TAG.json 2> gpurun_out/ab_$TAG.err
  echo "== $TAG rc=$? ($ENVS ${DIR:-_build})"
  TAG=$TAG PATTERN="$PATTERN" python - <<'PY'
import json, os, re
tag = os.environ["TAG"]; pat = os.environ.get("PATTERN") or None
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % tag).read().strip().splitlines()[-1])
except Exception as e:
    print("   unreadable:", e, open("gpurun_out/ab_%s.err" % tag).read()[-600:]); raise SystemExit
solo = d.get("kernels_one_worker") or {}
print("   ms/step %.1f  value %.0f  stages %s  each %s" % (d["ms_per_step"], d["value"], {k: round(v * 1e3, 1) for k, v in d["stage_seconds_per_step"].items()}, d["stage_device_ms_each_step"]))
print("   kernel s/step: in step %.1f ms, solo %.1f ms" % (1e3 * d["kernel_seconds_per_step"], 1e3 * sum(v["seconds_per_step"] for v in solo.values())))
fwd = sum(v["seconds_per_step"] for k, v in solo.items() if k.startswith("bandedDpForward"))
print("   forward DP solo total %.1f ms" % (fwd * 1e3))
for k, v in sorted(solo.items(), key=lambda kv: -kv[1]["seconds_per_step"]):
    if (pat and re.search(pat, k)) or (not pat and v["seconds_per_step"] > 0.002):
        s = d["kernels"].get(k, {})
        print("   %-50s solo %7.2f ms/step avg %7.3f ms %s | in step %7.2f ms/step" % (k, v["seconds_per_step"] * 1e3, v["avg_ms"], ("%6.0f GCUPS" % v["gcups"]) if "gcups" in v else "", s.get("seconds_per_step", 0) * 1e3))
PY
done
