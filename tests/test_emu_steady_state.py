"""A steady state allocates nothing on the device: the second and later steps of LowHash0 + Align4 + alignment table on one
context (and of the in-process group over one device) run in the buffers the first step left (grow-only scratch kept by
the context).  Checked on the emulated build with SHASTA_MI355X_LOG_ALLOC=1, which reports every (re)allocation of a device buffer
-- round 6 found two of LowHash0's buffers allocated and freed by every call this way (hipFree waits for the device)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
from shasta_amd import abi, lib as L
from tests import support
emu = L.Library(%(library)r)
toc, kmer, data7 = support.small_marker_set(n_reads=200, genome_markers=9000, seed=5)
p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
o = abi.default_align4_options(minAlignedMarkerCount=40)
o3 = abi.default_align3_options(minAlignedMarkerCount=40)
def mark(s):
    sys.stderr.write("### %%s\n" %% s); sys.stderr.flush()
def steps(owner, table):
    owner.set_kmer_ids(toc, kmer)
    for step in range(3):
        mark("step %%d" %% step)
        lh = owner.lowhash0(p)
        al = owner.align4(lh.candidates, o, want_ordinals=False, borrow=True)
        if table:
            owner.alignment_table(copy=False)
        a3 = owner.align3(lh.candidates[:100], o3, want_ordinals=False, borrow=True)
if sys.argv[1] == "context":
    with emu.context(0) as ctx:
        steps(ctx, True)
else:
    with emu.group([0]) as g:
        steps(g, False)
"""


def allocations_per_step(emu_lib, mode):
    env = dict(os.environ, SHASTA_MI355X_LOG_ALLOC="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "library": emu_lib.path}, mode], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    counts, step = {}, None
    for line in r.stderr.splitlines():
        if line.startswith("### step"):
            step = int(line.split()[-1]); counts[step] = 0
        elif "device buffer" in line and step is not None:
            counts[step] += 1
    return counts


def test_a_context_allocates_in_its_first_step_only(emu_lib):
    counts = allocations_per_step(emu_lib, "context")
    assert counts[0] > 20 and counts[1] == 0 and counts[2] == 0, counts


def test_a_group_over_one_device_allocates_in_its_first_step_only(emu_lib):
    counts = allocations_per_step(emu_lib, "group")
    assert counts[0] > 20 and counts[1] == 0 and counts[2] == 0, counts
