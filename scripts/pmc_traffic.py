"""Turn the FETCH_SIZE / WRITE_SIZE passes (scripts/gpu_pmc.sh) into per-launch HBM traffic per kernel,
calibrated on the known-byte-count kernels as MI355X_MICROARCH.md (HBM section) prescribes.
usage: python scripts/pmc_traffic.py gpurun_out <reads> > profiles/rNN_traffic_<reads>.json"""
import collections
import csv
import json
import os
import re
import sys

root = sys.argv[1]
workload_reads = int(sys.argv[2]) if len(sys.argv) > 2 else None


def load(tag, counter):
    path = os.path.join(root, "pmc_%s" % tag, "%s_counter_collection.csv" % tag.split("_")[-1] if tag.endswith("_cal") else "")
    d = os.path.join(root, "pmc_%s" % tag)
    f = [x for x in os.listdir(d) if x.endswith("counter_collection.csv")][0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    order = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, f))):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("shasta_mi355x::", ""))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
        order[k].append(float(r["Counter_Value"]))
    return agg, order


out = {"workload_reads": workload_reads, "unit": "bytes per launch", "note": "FETCH_SIZE/WRITE_SIZE are reported in KiB; factors = known bytes / reported bytes "
       "on calibrateReadDwordKernel (dword loads, the hash kernel's pattern) and calibrateWriteRecordKernel "
       "(4 lanes x 8 B record stores, the DP trace's pattern)"}
_, cal_f = load("fetch_cal", "FETCH_SIZE")
_, cal_w = load("write_cal", "WRITE_SIZE")
known = [1 << 30, 3 << 30]
rf = [known[i] / (cal_f["calibrateReadDwordKernel"][i] * 1024.0) for i in range(2)]
wf = [known[i] / (cal_w["calibrateWriteRecordKernel"][i] * 1024.0) for i in range(2)]
out["calibration"] = {"read_factor_1GiB_3GiB": rf, "write_factor_1GiB_3GiB": wf,
                      "read_raw_KiB": cal_f["calibrateReadDwordKernel"], "write_raw_KiB": cal_w["calibrateWriteRecordKernel"]}
read_factor, write_factor = rf[-1], wf[-1]
fetch, _ = load("fetch", "FETCH_SIZE")
write, _ = load("write", "WRITE_SIZE")
kernels = {}
for k in set(fetch) | set(write):
    n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
    if n == 0:
        continue
    fb = fetch.get(k, [1, 0.0])[1] * 1024.0 * read_factor / max(1, fetch.get(k, [1, 0])[0])
    wb = write.get(k, [1, 0.0])[1] * 1024.0 * write_factor / max(1, write.get(k, [1, 0])[0])
    kernels[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
out["kernels"] = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
print(json.dumps(out, indent=1))
