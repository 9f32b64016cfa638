#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 device assembly file (hipcc -S
--cuda-device-only): for every innermost loop the compiler marked, the VALU / SALU / vector-memory /
LDS / branch instruction counts of one trip.  A static measure -- what one trip issues, not how long
it takes -- used to compare versions of a loop that profiling showed to be bound by instruction
issue.

    hipcc -std=c++17 -O3 --offload-arch=gfx950 --cuda-device-only -S align4.hip -o align4.s
    python scripts/isa_loop.py align4.s 'bandedDpForwardKernel2ILi32ELi2'
"""
import re
import sys


def kernel_body(path, pattern):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if name is not None:
                break
            if re.search(pattern, m.group(1)):
                name = m.group(1)
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                break
            body.append(line.rstrip("\n"))
    if name is None:
        sys.exit("no kernel matches " + pattern)
    return name, body


def classify(op):
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    name, body = kernel_body(sys.argv[1], sys.argv[2])
    print(name)
    # The compiler annotates every basic block of a loop: ".LBBa_b: ; =>This Inner Loop Header" for the
    # header, "; in Loop: Header=BBa_b" on the labels and "; %bb.N:" markers of the other blocks.
    loops, order, current = {}, [], None
    for line in body:
        block = re.match(r"^(\.LBB\d+_\d+:|;\s*%bb\.\d+:)(.*)$", line)
        if block:
            rest = block.group(2)
            m = re.search(r"in Loop: Header=(BB\d+_\d+)", rest)
            if "Inner Loop Header" in rest:
                current = block.group(1)[2:-1]
            elif m:
                current = m.group(1)
            else:
                current = None
            if current is not None and current not in loops:
                loops[current] = {}
                order.append(current)
            continue
        if current is None:
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        kind = classify(code.split()[0])
        loops[current][kind] = loops[current].get(kind, 0) + 1
    for header in order:
        print("  loop .L%s: %s" % (header, ", ".join("%s %d" % kv for kv in sorted(loops[header].items()))))


if __name__ == "__main__":
    main()
