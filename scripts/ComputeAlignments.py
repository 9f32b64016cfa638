#!/usr/bin/python3
"""The reference's scripts/ComputeAlignments.py on the MI355X path (alignment method 4): run it in a
Shasta run directory after FindAlignmentCandidatesLowHash0.py.  Options as NAME=VALUE arguments using
the attribute names of shasta.AlignOptions, e.g.  minAlignedMarkerCount=10 minAlignedFraction=0.1 maxSkip=100."""
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd.assembler as shasta  # noqa: E402

a = shasta.Assembler()
a.accessKmers()
a.accessMarkers()
a.accessAlignmentCandidates()

alignOptions = shasta.AlignOptions()
for item in sys.argv[1:]:
    name, value = item.split("=", 1)
    if not hasattr(alignOptions, name):
        raise SystemExit("unknown AlignOptions attribute " + name)
    setattr(alignOptions, name, ast.literal_eval(value))

a.computeAlignments(alignOptions, 0)
