#!/usr/bin/env python3
"""Readable summary of bench runs: bench_summary.py <prefix> [<prefix> ...] reads <prefix>.json (the final line) and
<prefix>_details.json (the tables)."""
import json
import sys


def main():
    for name in sys.argv[1:]:
        try:
            text = open(name + ".json").read().strip().splitlines()[-1]
            head = json.loads(text)
            d = json.loads(open(name + "_details.json").read())
        except Exception as e:          # noqa: BLE001
            print("==", name, "unreadable:", e)
            continue
        print("==", name, "final line %d bytes; ms/step %.1f value %.0f" % (len(text), d["ms_per_step"], d["value"]),
              {k: (round(v * 1e3, 2) if v is not None else None) for k, v in d["stage_seconds_per_step"].items()})
        if d.get("path"):
            print("!!", d["path"], "after", d.get("earlier_attempts"))
        print("each", d["stage_device_ms_each_step"])
        if "cpu_baseline" in head:
            print("cpu", json.dumps(head["cpu_baseline"]))
            print("parity", head["parity_at_bench_size"])
        print("roofline", json.dumps(head.get("roofline")))
        print("banded_dp", d.get("banded_dp"))
        if d.get("give_ups"):
            print("give_ups", d["give_ups"])
        solo = d.get("kernels_one_worker") or {}
        print("kernel s/step: in step %.1f ms, solo %.1f ms" % (1e3 * d["kernel_seconds_per_step"], 1e3 * sum(v["seconds_per_step"] for v in solo.values())))
        for k, v in sorted(solo.items(), key=lambda kv: -kv[1]["seconds_per_step"]):
            s = d["kernels"].get(k, {})
            if v["seconds_per_step"] > 0.0004:
                print("   %-52s solo %7.2f ms/step avg %7.3f ms %s%s| in step %7.2f ms/step %5.1f launches" % (
                    k, v["seconds_per_step"] * 1e3, v["avg_ms"], ("%6.0f GCUPS " % v["gcups"]) if "gcups" in v else "",
                    ("valu %.2f " % v["valu_issue_frac"]) if v.get("valu_issue_frac") else "",
                    s.get("seconds_per_step", 0) * 1e3, s.get("launches_per_step", 0)))


if __name__ == "__main__":
    main()
