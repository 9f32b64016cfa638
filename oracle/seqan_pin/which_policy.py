#!/usr/bin/env python3
"""SeqAn pin kit -- reads the lines of ./pin_seqan cases.txt on stdin and says which of the 12 tie policies of
oracle/banded_dp.hpp they are (expected.json, made by make_cases.py).  Needs nothing but python3.

    ./pin_seqan cases.txt | python3 which_policy.py

One policy matches every case  -> a14 of SURVEY 8 is pinned: if it is not 0, build with -DSHASTA_DP_TIE_POLICY=<n> (make -C
shasta_amd/csrc EXTRA=-DSHASTA_DP_TIE_POLICY=<n>), call oracle_set_tie_policy(<n>) / ref_set_tie_policy(<n>) in the checkers
and regenerate tests/golden/*.npz.  No policy matches -> the per-case table below shows which policies agree where; SeqAn then
does something outside this family (send the output of pin_seqan along)."""
import json
import os
import sys


def parse(lines):
    got = {}
    for line in lines:
        words = line.split()
        if len(words) < 4 or words[1] != "score" or words[3] != "pairs":
            continue
        got[words[0]] = [int(words[2]), [[int(v) for v in w.split(":")] for w in words[4:]]]
    return got


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    expected = json.load(open(os.path.join(here, "expected.json")))
    got = parse(sys.stdin.read().splitlines())
    cases = expected["cases"]
    missing = [c for c in cases if c not in got]
    if missing:
        print("no output for: %s" % " ".join(missing))
        return 2
    agree = {p: [expected["outputs"][p][k] == got[c] for k, c in enumerate(cases)] for p in expected["policies"]}
    full = [p for p, a in agree.items() if all(a)]
    for p in sorted(agree, key=int):
        print("policy %2s  %-24s agrees on %2d of %d cases%s" % (p, expected["policies"][p], sum(agree[p]), len(cases), "   <== every case" if all(agree[p]) else ""))
    if len(full) == 1:
        p = full[0]
        print("\nSeqAn's tie policy here is %s (%s)." % (p, expected["policies"][p]))
        print("This repository ships policy 0; " + ("nothing to change: a14 is pinned." if p == "0" else "set SHASTA_DP_TIE_POLICY=%s (see the docstring of this script)." % p))
        return 0
    print("\nNo single policy of the family reproduces SeqAn on every case." if not full else "\nSeveral policies match (cases.txt no longer separates them?): %s" % full)
    for k, c in enumerate(cases):
        print("  %s: agreeing policies %s; SeqAn: %s" % (c, [int(p) for p in agree if agree[p][k]], got[c]))
    return 1


if __name__ == "__main__":
    sys.exit(main())
