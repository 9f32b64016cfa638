"""shasta_amd: MI355X-native (gfx950) drop-in for Shasta's overlap-detection hot path.

Scope: Assembler::findAlignmentCandidatesLowHash0 (LowHash0) and
Assembler::computeAlignments with alignMethod 4 (Align4), behind a C ABI
(include/shasta_mi355x.h, shasta_amd/csrc/).  This package is the thin host-side
mirror used by tests and bench.py; the product is the shared library.
"""
from . import abi  # noqa: F401
from .lib import Library, LibraryNotBuilt, load  # noqa: F401


def kernel_source_hash():
    """sha256 (16 hex digits) over shasta_amd/csrc's sources: counters and profiles say which build they were collected on
    (bench.py refuses to price a kernel with counters of another build; the GPU box has no .git to ask)."""
    import glob
    import hashlib
    import os
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
