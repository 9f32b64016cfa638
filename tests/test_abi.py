"""The C ABI on a machine without a GPU: the library loads, exports exactly the entry points
include/shasta_mi355x.h declares, its PODs have the reference's sizes, and every compute entry
point fails loudly (no CPU fallback) instead of computing."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import shasta_amd
from shasta_amd import abi, lib as libmod
from tests import support

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "shasta_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(shasta_mi355x_\w+)\s*\(", text)))


def test_library_exports_every_declared_entry_point():
    library = shasta_amd.load()
    declared = declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(library.lib, name), name
    assert sorted(libmod.EXPORTS) == declared          # the Python mirror binds the whole header, nothing else
    assert "gfx950" in library.version()


def test_pod_sizes_match_the_reference_structs():
    # SURVEY Appendix B (verified there by compiling the reference headers).
    assert C.sizeof(abi.OrientedReadPair) == 12
    assert C.sizeof(abi.AlignmentInfo) == 52
    assert C.sizeof(abi.AlignmentData) == 64
    assert abi.PAIR_DTYPE.itemsize == 12 and abi.ALIGNMENT_DATA_DTYPE.itemsize == 64


def test_ctypes_mirror_matches_the_header_layout(tmp_path):
    # sizeof of every struct of the header as gcc lays it out vs the ctypes mirror.
    import subprocess
    structs = {
        "shasta_oriented_read_pair": abi.OrientedReadPair, "shasta_alignment_info": abi.AlignmentInfo,
        "shasta_alignment_data": abi.AlignmentData, "shasta_lowhash0_params": abi.LowHash0Params,
        "shasta_lowhash0_result": abi.LowHash0Result, "shasta_align4_options": abi.Align4Options,
        "shasta_align3_options": abi.Align3Options, "shasta_align4_result": abi.Align4Result,
        "shasta_mi355x_kernel_stat": abi.KernelStat, "shasta_markers_result": abi.MarkersResult,
    }
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "shasta_mi355x.h"\nint main(void) {\n' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in structs) + "return 0; }\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for name, mirror in structs.items():
        assert int(out[name]) == C.sizeof(mirror), name
    o = abi.default_align3_options()
    assert (o.matchScore, o.mismatchScore, o.gapScore, o.bandExtend, o.maxBand, o.k) == (6, -1, -1, 10, 1000, 10)


def test_missing_library_is_an_error_not_a_fallback(tmp_path):
    with pytest.raises(libmod.LibraryNotBuilt, match="no CPU fallback"):
        libmod.Library(str(tmp_path / "libshasta_mi355x.so"))


@pytest.mark.skipif(shasta_amd.load().device_count() > 0, reason="a gfx950 device is present")
def test_compute_entry_points_fail_loudly_without_a_gpu():
    library = shasta_amd.load()
    assert library.device_count() == 0
    toc, kmer, data7 = support.small_marker_set(n_reads=20, genome_markers=3000, seed=1)
    with pytest.raises(RuntimeError):
        library.lowhash0(toc, data7, None, abi.default_lowhash0_params())
    with pytest.raises(RuntimeError):
        library.align4_batch(toc, data7, abi.make_pairs([0], [1], [1]), abi.default_align4_options())
    with pytest.raises(RuntimeError):
        library.align3_batch(toc, data7, abi.make_pairs([0], [1], [1]), abi.default_align3_options())
    with pytest.raises(RuntimeError):
        library.context(0)
    with pytest.raises(RuntimeError):
        library.hash_windows(np.arange(10, dtype=np.uint32), 4, 0)


def test_the_python_wrapper_asks_for_eight_hardware_queues_and_the_library_leaves_the_environment_alone():
    # GPU_MAX_HW_QUEUES=8 (six aligner workers x (stream + side stream) on the runtime's default four queues cost 15 % of an aligner
    # call) is the CALLER's to set before the first HIP call: shasta_amd/lib.py does when it loads the library, unless the environment
    # says otherwise; the library itself does not write the host process's environment (round 4's load-time setenv is gone).
    import subprocess
    import sys
    raw = ("import ctypes, os\n"
           "ctypes.CDLL(%r)\n"
           "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
           "v = libc.getenv(b'GPU_MAX_HW_QUEUES'); print(v.decode() if v else 'unset')\n") % libmod.SO_PATH
    wrapped = ("import sys, ctypes; sys.path.insert(0, %r)\n"
               "from shasta_amd import lib as libmod\n"
               "libmod.Library(libmod.SO_PATH)\n"
               "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
               "v = libc.getenv(b'GPU_MAX_HW_QUEUES'); print(v.decode() if v else 'unset')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.check_output([sys.executable, "-c", raw], env=env).decode().strip() == "unset"
    assert subprocess.check_output([sys.executable, "-c", wrapped], env=env).decode().strip() == "8"
    assert subprocess.check_output([sys.executable, "-c", wrapped], env=dict(env, GPU_MAX_HW_QUEUES="4")).decode().strip() == "4"
