"""The emulated HIP runtime's deferred mode (tests/emu/hip_emu.cpp, HIPEMU_ASYNC): work waits in its stream's queue until the host's
own synchronisation delivers it, events order streams, pageable copies behave as the runtime's do.  Test of the test infrastructure:
a missing synchronisation in the host code can only show on the emulator if these hold."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def run(binary, **env):
    out = subprocess.run([binary], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    return {k: int(v) for k, v in (line.split() for line in out.stdout.strip().splitlines())}


def test_streams_and_events_of_the_deferred_mode(emu_lib, tmp_path):
    obj, binary = str(tmp_path / "async_streams.o"), str(tmp_path / "async_streams")
    subprocess.check_call([CXX, "-x", "c++", "-std=c++17", "-O0", "-I" + os.path.join(EMU, "include"), "-c", os.path.join(EMU, "selftest", "async_streams.cpp"), "-o", obj])
    subprocess.check_call([CXX, obj, os.path.join(os.path.dirname(emu_lib.path), "hip_emu.o"), "-lpthread", "-ldl", "-o", binary])
    immediate = run(binary, HIPEMU_ASYNC="0")
    assert immediate == dict(value_before_synchronisation=7, value_after_synchronisation=7, event_query_before=1, pinned_before_synchronisation=9,
                             pinned_after_synchronisation=9, event_query_after=1, pageable_round_trip=5, value_after_free=11)
    for seed in ("1", "2", "3", "4", "5"):
        deferred = run(binary, HIPEMU_ASYNC=seed)
        assert deferred == dict(value_before_synchronisation=0, value_after_synchronisation=7, event_query_before=0, pinned_before_synchronisation=-1,
                                pinned_after_synchronisation=9, event_query_after=1, pageable_round_trip=5, value_after_free=11), (seed, deferred)
