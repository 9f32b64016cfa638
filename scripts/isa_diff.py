#!/usr/bin/env python3
"""Compares the gfx950 ISA of every kernel in two device assembly files (hipcc -S
--cuda-device-only).  Used when host code or NEW kernels are added to a .hip file whose
existing kernels were validated on the GPU: identical instruction streams mean the validated
kernels are untouched.

    hipcc -std=c++17 -O3 --offload-arch=gfx950 --cuda-device-only -S x.hip -o new.s
    python scripts/isa_diff.py old.s new.s
"""
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = body
            name = None
            continue
        code = line.split(";")[0].rstrip()
        if not code.strip():
            continue
        code = re.sub(r"\.LBB\d+_", ".LBB_", code)
        code = re.sub(r"\.Lfunc_end\d+", ".Lfunc_end", code)
        body.append(code)
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    bad = 0
    for k in sorted(a):
        if k not in b:
            print("MISSING  ", k)
            bad += 1
        elif a[k] != b[k]:
            print("DIFFERENT", k, len(a[k]), "->", len(b[k]), "instructions")
            bad += 1
    for k in sorted(set(b) - set(a)):
        print("new      ", k, len(b[k]), "instructions")
    print("%d kernels compared, %d differ" % (len(a), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
