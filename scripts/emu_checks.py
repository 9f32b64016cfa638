"""TEST INFRASTRUCTURE: the emulated library's battery of checks against the oracle, on any build of tests/emu.

    python scripts/emu_checks.py [tests/emu/_build_pattern/libshasta_mi355x_emu.so] [quick]

Used with the builds and switches that make a read of something nobody wrote visible:
  * tests/emu/_build_pattern (make -C tests/emu OUT=_build_pattern SAN=-ftrivial-auto-var-init=pattern): locals nobody initialised hold 0xAA bytes;
  * HIPEMU_LDS_SCRAMBLE=<seed>: every __shared__ variable filled with pseudo-random data before every workgroup, garbage from switched-off lanes;
  * SHASTA_MI355X_SCRAMBLE=1: the library's scratch in device memory filled with pseudo-random data before every job;
  * scripts/emu_asan.sh: the same battery under AddressSanitizer.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import bindings
from shasta_amd import lib as L
from tests import adversarial, align3_checks, group_checks, long_read_checks, sparse_checks


def main():
    path = next((a for a in sys.argv[1:] if a.endswith(".so")), "tests/emu/_build/libshasta_mi355x_emu.so")
    quick = "quick" in sys.argv[1:]
    emu, orc = L.Library(path), bindings.OracleLib()
    print("library:", path, flush=True)
    for seed in (21, 22) if quick else (21, 22, 23, 24, 25, 26):
        align3_checks.against_oracle(emu, orc, seed, dict())
    print("align method 3 ok", flush=True)
    os.environ["SHASTA_MI355X_ALIGN_WORKERS"] = "1"
    os.environ["SHASTA_MI355X_MATCH_SHIFT"] = "20"
    r = long_read_checks.both_long(emu, orc, lengths=(9000, 12500, 9500, 8300, 4000), genome_markers=16000)
    print("pairs of two long reads:", {k: v for k, v in r.items() if k != "rows"}, flush=True)
    del os.environ["SHASTA_MI355X_MATCH_SHIFT"]
    print("repeat-rich pairs of long reads (full cell tables):", long_read_checks.full_tables(emu, orc, lengths=(9000, 12000, 8800), alphabet_size=150), flush=True)
    for force in ("long", "big"):
        print("every candidate forced through the windowed kernels (%s):" % force,
              long_read_checks.forced(emu, orc, None, force, n_reads=100, limit=250, adversarial_sets=not quick), flush=True)
    print("long dense paths:", sparse_checks.long_dense_paths(emu, orc), flush=True)
    print("a call without ordinals:", sparse_checks.without_ordinals(emu, orc, n_reads=100, limit=300), flush=True)
    del os.environ["SHASTA_MI355X_ALIGN_WORKERS"]
    print("aligner, share of the DP cells from the matches:", sparse_checks.aligner(emu, orc, n_reads=90, limit=160), flush=True)
    print("dp tasks:", sparse_checks.dp_tasks(emu, orc, clean=30, tie_heavy=20, alternatives=(2,), long_every=44), flush=True)
    print("locally ambiguous tasks (anchor kernel):", sparse_checks.anchored_tasks(emu, orc, seeds=(3, 4, 5), tasks=24), flush=True)
    print("tiny tasks:", sparse_checks.tiny_tasks(emu, orc, tasks=150, alternatives=(2,)), flush=True)
    print("wave kernel forms:", sparse_checks.wave_kernel_forms(emu, orc), flush=True)
    print("anchor kernel, second launch:", sparse_checks.anchor_kernel_second_launch(emu, orc, alternatives=(2,)), flush=True)
    for name in adversarial.READ_SET_NAMES[:-1]:
        print(name, adversarial.aligner_case(emu, orc, name, long_reads=False), flush=True)
    adversarial.lowhash0(emu, orc)
    print("adversarial LowHash0 ok", flush=True)
    print("group:", group_checks.lowhash0_and_aligners(emu, orc, device_lists=((0, 0),), n_reads=120, limit=200), flush=True)
    print("ALL CHECKS PASSED", flush=True)


if __name__ == "__main__":
    main()
