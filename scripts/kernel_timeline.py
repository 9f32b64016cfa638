#!/usr/bin/env python3
"""What runs beside what: a rocprofv3 --kernel-trace kernel_trace.csv cut into segments at idle gaps (> gap_ms), and for
every segment its length, the time at least one kernel was running, the average number of kernels in flight, and per kernel
its total time and the time it was the ONLY kernel in flight (nothing overlapped it: the device was as busy as that kernel
alone keeps it).
usage: kernel_timeline.py <kernel_trace.csv> [gap_ms=15] [min_segment_ms=100]"""
import csv, re, sys
from collections import defaultdict


def short(name):
    name = name.replace("shasta_mi355x::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    path = sys.argv[1]
    gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 15e6
    min_segment = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 100e6
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    segments, cur, cur_end = [], [], None
    for s, e, n in rows:
        if cur and s - cur_end > gap:
            segments.append(cur); cur = []
        cur.append((s, e, n)); cur_end = e if len(cur) == 1 else max(cur_end, e)
    if cur: segments.append(cur)
    for seg in segments:
        t0, t1 = seg[0][0], max(e for _, e, _ in seg)
        if t1 - t0 < min_segment:
            continue
        events = []
        for s, e, n in seg:
            events.append((s, 1, n)); events.append((e, -1, n))
        events.sort(key=lambda x: (x[0], x[1]))
        running = defaultdict(int)
        count, last, busy, weighted = 0, t0, 0, 0
        total, alone = defaultdict(int), defaultdict(int)
        histogram = defaultdict(int)
        for t, d, n in events:
            dt = t - last
            if dt > 0:
                histogram[min(count, 6)] += dt
                if count > 0:
                    busy += dt; weighted += dt * count
                for name, c in running.items():
                    if c > 0:
                        total[name] += dt * c
                        if count == c:
                            alone[name] += dt
            last = t
            running[n] += d; count += d
        length = t1 - t0
        print("segment %.1f ms: %d kernels, some kernel in flight %.1f ms (%.0f %%), %.2f kernels in flight on average while busy"
              % (length / 1e6, len(seg), busy / 1e6, 100.0 * busy / length, weighted / max(busy, 1)))
        print("   kernels in flight -> ms: " + ", ".join("%s%d: %.1f" % (">=" if k == 6 else "", k, v / 1e6) for k, v in sorted(histogram.items())))
        for name, t in sorted(total.items(), key=lambda kv: -alone[kv[0]])[:14]:
            print("   %-52s total %8.2f ms   alone %8.2f ms" % (name[:52], t / 1e6, alone[name] / 1e6))


if __name__ == "__main__":
    main()
