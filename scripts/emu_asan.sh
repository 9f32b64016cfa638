#!/bin/bash
# TEST INFRASTRUCTURE: the emulated build under AddressSanitizer.  Device memory is host heap there, so a kernel that reads or writes
# past a device buffer (what a GPU answers with a page fault or with silent corruption) stops the run with a report.
#   bash scripts/emu_asan.sh        builds tests/emu/_build_asan and runs the aligner, sparse-path, adversarial, LowHash0 and group checks on it
cd "$(dirname "$0")/.."
make -s -C tests/emu OUT=_build_asan SAN=-fsanitize=address _build_asan/libshasta_mi355x_emu.so || exit 1
ASAN=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python - <<'PY'
import sys
sys.path.insert(0, ".")
from oracle import bindings
from shasta_amd import abi, lib as L
import os
from tests import adversarial, align3_checks, group_checks, long_read_checks, sparse_checks, support
emu, orc = L.Library("tests/emu/_build_asan/libshasta_mi355x_emu.so"), bindings.OracleLib()
# (round 6's kernels first: the windowed class and its large graph, the wavefront walk of long dense paths, a call without ordinals)
os.environ["SHASTA_MI355X_ALIGN_WORKERS"] = "1"
os.environ["SHASTA_MI355X_MATCH_SHIFT"] = "20"
r = long_read_checks.both_long(emu, orc, lengths=(9000, 12500, 9500, 8300, 4000), genome_markers=16000)
print("pairs of two long reads:", {k: v for k, v in r.items() if k != "rows"}, flush=True)
del os.environ["SHASTA_MI355X_MATCH_SHIFT"]
print("repeat-rich pairs of long reads (full cell tables):", long_read_checks.full_tables(emu, orc, lengths=(9000, 12000, 8800), alphabet_size=150), flush=True)
for force in ("long", "big"):
    print("every candidate forced through the windowed kernels (%s):" % force, long_read_checks.forced(emu, orc, None, force, n_reads=100, limit=250, adversarial_sets=True), flush=True)
print("long dense paths:", sparse_checks.long_dense_paths(emu, orc), flush=True)
print("a call without ordinals:", sparse_checks.without_ordinals(emu, orc, n_reads=100, limit=300), flush=True)
del os.environ["SHASTA_MI355X_ALIGN_WORKERS"]
print("aligner, share of the DP cells from the matches:", sparse_checks.aligner(emu, orc, n_reads=90, limit=160), flush=True)
print("dp tasks:", sparse_checks.dp_tasks(emu, orc, clean=30, tie_heavy=20, alternatives=(2,), long_every=44), flush=True)
print("locally ambiguous tasks (anchor kernel):", sparse_checks.anchored_tasks(emu, orc, seeds=(3, 4, 5), tasks=24), flush=True)
print("tiny tasks:", sparse_checks.tiny_tasks(emu, orc, tasks=150, alternatives=(2,)), flush=True)
print("wave kernel forms (every capacity class, own ordering, side stream):", sparse_checks.wave_kernel_forms(emu, orc), flush=True)
print("anchor kernel, second launch (dense cells with, without):", sparse_checks.anchor_kernel_second_launch(emu, orc, alternatives=(2,)), flush=True)
for name in adversarial.READ_SET_NAMES[:-1]:
    print(name, adversarial.aligner_case(emu, orc, name, long_reads=False), flush=True)
adversarial.lowhash0(emu, orc)
print("adversarial LowHash0 ok", flush=True)
print("group:", group_checks.lowhash0_and_aligners(emu, orc, device_lists=((0, 0),), n_reads=120, limit=200), flush=True)
align3_checks.against_oracle(emu, orc, 21, dict())
print("align method 3 ok")
PY
