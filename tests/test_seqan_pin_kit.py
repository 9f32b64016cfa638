"""The SeqAn pin kit (oracle/seqan_pin) end to end, with the repository's SeqAn stand-in in SeqAn's place: its program compiles
against <seqan/align.h> as the reference uses it, its cases separate all 12 tie policies, and its script names the policy the
"SeqAn" at hand follows.  On a machine with the real library, `make -C oracle/seqan_pin` is the same run -- and pins a14."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "oracle", "seqan_pin")


@pytest.fixture(scope="module")
def pin_program(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pin") / "pin_seqan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "oracle", "ref_build", "shims"), os.path.join(KIT, "pin_seqan.cpp"), "-o", exe])
    return exe


def test_cases_separate_every_two_policies():
    expected = json.load(open(os.path.join(KIT, "expected.json")))
    assert len(expected["cases"]) <= 10 and len(expected["policies"]) == 12
    outputs = [json.dumps(expected["outputs"][str(p)]) for p in range(12)]
    assert len(set(outputs)) == 12
    lines = [l for l in open(os.path.join(KIT, "cases.txt")) if not l.startswith("#")]
    assert len(lines) == len(expected["cases"])


@pytest.mark.parametrize("policy", [0, 2, 3, 7, 11])
def test_script_names_the_policy_of_the_library_at_hand(pin_program, policy):
    lines = subprocess.run([pin_program, os.path.join(KIT, "cases.txt")], env=dict(os.environ, ORACLE_TIE_POLICY=str(policy)),
                           capture_output=True, text=True, check=True).stdout
    verdict = subprocess.run([sys.executable, os.path.join(KIT, "which_policy.py")], input=lines, capture_output=True, text=True)
    assert verdict.returncode == 0, verdict.stdout
    assert "SeqAn's tie policy here is %d " % policy in verdict.stdout
    assert ("a14 is pinned" in verdict.stdout) == (policy == 0)


def test_script_says_so_when_no_policy_matches(pin_program):
    lines = subprocess.run([pin_program, os.path.join(KIT, "cases.txt")], capture_output=True, text=True, check=True).stdout.splitlines()
    lines[3] = lines[3].split(" pairs")[0] + " pairs 0:0"            # an answer no policy gives
    verdict = subprocess.run([sys.executable, os.path.join(KIT, "which_policy.py")], input="\n".join(lines) + "\n", capture_output=True, text=True)
    assert verdict.returncode == 1 and "No single policy" in verdict.stdout
