#!/usr/bin/python3
"""The reference's scripts/FindMarkers.py on the MI355X path: run it in a Shasta run directory (Data/ holds
Reads-Bases, Reads-BaseCount, Kmers); writes Data/Markers.{toc,data}."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd.assembler as shasta  # noqa: E402

a = shasta.Assembler()
a.accessKmers()
a.findMarkers()
