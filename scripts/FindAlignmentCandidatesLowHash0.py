#!/usr/bin/python3
"""The reference's scripts/FindAlignmentCandidatesLowHash0.py on the MI355X path: run it in a Shasta
run directory (with Data/).  MinHash parameters come from the command line instead of shasta.conf:
    FindAlignmentCandidatesLowHash0.py [m hashFraction minHashIterationCount alignmentCandidatesPerRead
                                        minBucketSize maxBucketSize minFrequency]
(defaults: MinHashOptions, src/AssemblerOptions.cpp:327-371)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd.assembler as shasta  # noqa: E402

v = sys.argv[1:]
arg = lambda k, default: v[k] if len(v) > k else default

a = shasta.Assembler()
a.accessKmers()
a.accessMarkers()
a.findAlignmentCandidatesLowHash0(
    m=int(arg(0, 4)),
    hashFraction=float(arg(1, 0.01)),
    minHashIterationCount=int(arg(2, 10)),
    alignmentCandidatesPerRead=float(arg(3, 20)),
    minBucketSize=int(arg(4, 0)),
    maxBucketSize=int(arg(5, 10)),
    minFrequency=int(arg(6, 2)))
a.computeCandidateTable()
