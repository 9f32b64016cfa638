"""shasta_amd: MI355X-native (gfx950) drop-in for Shasta's overlap-detection hot path.

Scope: Assembler::findAlignmentCandidatesLowHash0 (LowHash0) and
Assembler::computeAlignments with alignMethod 4 (Align4), behind a C ABI
(include/shasta_mi355x.h, shasta_amd/csrc/).  This package is the thin host-side
mirror used by tests and bench.py; the product is the shared library.
"""
from . import abi  # noqa: F401
from .lib import Library, LibraryNotBuilt, load  # noqa: F401
