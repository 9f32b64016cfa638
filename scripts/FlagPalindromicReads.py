#!/usr/bin/python3
"""The reference's scripts/FlagPalindromicReads.py on the MI355X path: run it in a Shasta run directory after
FindMarkers.py.  The reference reads the [Reads] palindromicReads.* values from shasta.conf; here they are
NAME=VALUE arguments with the defaults of src/AssemblerOptions.cpp:255-288, e.g.  deltaThreshold=100 maxSkip=100."""
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd.assembler as shasta  # noqa: E402

options = dict(maxSkip=100, maxDrift=100, maxMarkerFrequency=10, alignedFractionThreshold=0.1,
               nearDiagonalFractionThreshold=0.1, deltaThreshold=100)
for item in sys.argv[1:]:
    name, value = item.split("=", 1)
    if name not in options:
        raise SystemExit("unknown option " + name)
    options[name] = ast.literal_eval(value)

a = shasta.Assembler()
a.accessKmers()
a.accessMarkers()
a.flagPalindromicReads(**options)
