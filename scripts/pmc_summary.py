#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes over one bench command (one counter_collection.csv per pass; the
passes must be separate runs: FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md).

    pmc_summary.py <workload reads> <out.json> <pass dir> [<pass dir> ...]

HBM bytes per launch = (FETCH_SIZE x 2 + WRITE_SIZE) KiB: gfx950's FETCH_SIZE reports half of the bytes of a
streaming read (MI355X_MICROARCH.md, HBM section); both factors are re-checked on known byte counts in the same
session (scripts/calibrate_pmc.py: `calibration` in the output).  SQ_* are device-wide sums (quad-cycles for the
cycle counters, wave-instructions for SQ_INSTS_*)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_stats_summary import short


def main():
    reads, out_path, dirs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    dispatches = collections.defaultdict(lambda: collections.defaultdict(set))
    calibration = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "*counter_collection.csv")):
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                if k.startswith("calibrate"):
                    calibration.setdefault(k + " " + r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                dispatches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    kernels = {}
    for k, c in acc.items():
        if c.get("SQ_WAVE_CYCLES", 0) < 1e8 and c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0) < 1e4:
            continue
        n = max(len(v) for v in dispatches[k].values())
        row = {"launches": n}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            row["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / n
            row["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0 / n
            row["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / n
        if "SQ_INSTS_VALU" in c:
            row["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"] / n
        for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                     "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if name in c:
                row[name + "_per_launch"] = c[name] / n
        kernels[k] = row
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import shasta_amd
    json.dump({"workload_reads": reads, "kernel_source_hash": shasta_amd.kernel_source_hash(),
               "what": "rocprofv3 --pmc passes (one run each) over `python bench.py --reads %d --steps 1 --warmup 0 --no-cpu-baseline`, per kernel and launch" % reads,
               "calibration": {k: v for k, v in calibration.items()}, "kernels": kernels}, open(out_path, "w"), indent=1)
    for k in sorted(kernels, key=lambda k: -kernels[k].get("SQ_WAVE_CYCLES_per_launch", 0) * kernels[k]["launches"])[:12]:
        print(k, {a: ("%.3g" % b) for a, b in kernels[k].items()})


if __name__ == "__main__":
    main()
