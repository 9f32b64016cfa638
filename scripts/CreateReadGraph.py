#!/usr/bin/python3
"""The reference's scripts/CreateReadGraph.py (ReadGraph.creationMethod 0) on the output of ComputeAlignments.py:
run it in a Shasta run directory.  Arguments: maxAlignmentCount=6 maxTrim=30 (src/AssemblerOptions.cpp defaults)."""
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd.assembler as shasta  # noqa: E402

options = dict(maxAlignmentCount=6, maxTrim=30)
for item in sys.argv[1:]:
    name, value = item.split("=", 1)
    if name not in options:
        raise SystemExit("unknown option " + name)
    options[name] = ast.literal_eval(value)

a = shasta.Assembler()
a.accessMarkers()
a.accessAlignmentData()
a.createReadGraph(options["maxAlignmentCount"], options["maxTrim"])
