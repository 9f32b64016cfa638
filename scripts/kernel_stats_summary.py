#!/usr/bin/env python3
"""Short names + shares from a rocprofv3 --kernel-trace --stats kernel_stats.csv.
usage: kernel_stats_summary.py <kernel_stats.csv> [steps]"""
import csv, re, sys


def short(name):
    name = name.replace("shasta_mi355x::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    # cut the argument list: the first '(' at template depth 0
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    print("total %.2f ms over %g steps = %.2f ms/step" % (total / 1e6, steps, total / 1e6 / steps))
    for r in rows:
        t = int(r["TotalDurationNs"])
        if t < total * 0.0005:
            continue
        print("%-58s calls %6s  %9.2f ms/step  avg %9.1f us  %5.2f%%" % (short(r["Name"])[:58], r["Calls"], t / 1e6 / steps, float(r["AverageNs"]) / 1e3, 100.0 * t / total))


if __name__ == "__main__":
    main()
