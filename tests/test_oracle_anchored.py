"""The anchored form of the banded alignment (oracle/anchored_chain.hpp: dense DP only between the anchors where the optimal chains of
matches differ, corners fixed) against the dense DP of oracle/banded_dp.hpp (which restates /root/reference/src/Align4.cpp:963-1068
with SeqAn's globalAlignment) under every tie policy.  A prototype on the oracle's side: no product code calls it."""
import numpy as np
import pytest

from tests.test_gpu_align4 import noisy_copy


def _tasks(seed, count):
    rng = np.random.default_rng(seed)
    made = 0
    while made < count:
        alphabet = int(rng.choice([1 << 20, 1 << 20, 200, 50, 12, 4]))      # the small ones: repeated markers, many ties
        n = int(rng.integers(20, 700))
        genome = rng.integers(0, alphabet, size=n + 600, dtype=np.uint32)
        a = noisy_copy(rng, genome[:n], alphabet=alphabet)
        off = int(rng.integers(0, 300))
        b = noisy_copy(rng, genome[off: off + n], alphabet=alphabet)
        if len(a) == 0 or len(b) == 0:
            continue
        width = int(rng.choice([10, 20, 40, 60, 100, 300]))
        center = off + int(rng.integers(-30, 30))
        lo, hi = center - width // 2, center - width // 2 + width - 1
        if lo > len(a) or hi < -len(b):
            continue
        made += 1
        yield a, b, lo, hi


@pytest.mark.parametrize("seed", [1, 2])
def test_anchored_form_equals_the_dense_dp_under_every_tie_policy(oracle_lib, seed):
    whole = windows = 0
    try:
        for a, b, lo, hi in _tasks(seed, 120):
            for policy in range(12):
                oracle_lib.set_tie_policy(policy)
                want, score = oracle_lib.banded_dp(a, b, lo, hi)
                got, info = oracle_lib.anchored_dp(a, b, lo, hi)
                assert info["score"] == score, (policy, len(a), len(b), lo, hi)
                assert np.array_equal(want.reshape(-1, 2), got), (policy, len(a), len(b), lo, hi)
            whole += info["whole_task_dense"]
            windows += info["windows"]
    finally:
        oracle_lib.set_tie_policy(0)
    assert windows > 0 and whole < 12           # the cases reach the windows, and few tasks fall back to the whole matrix


def test_anchored_form_on_the_edges(oracle_lib):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 50, size=300, dtype=np.uint32)
    cases = [
        (a, a.copy(), -5, 5), (a, a[100:250].copy(), 80, 120), (a[100:250].copy(), a, -120, -80), (a, a[::-1].copy(), -10, 10),
        (a[:1], a[:1].copy(), 0, 0), (a, a.copy(), 250, 299), (a, a.copy(), -299, -250),
        (np.zeros(40, np.uint32), np.zeros(50, np.uint32), -20, 20),             # one marker repeated: every cell a hit
        (np.arange(30, dtype=np.uint32), np.arange(100, 130, dtype=np.uint32), -5, 5),      # no hit
    ]
    try:
        for policy in range(12):
            oracle_lib.set_tie_policy(policy)
            for k0, k1, lo, hi in cases:
                want, score = oracle_lib.banded_dp(k0, k1, lo, hi)
                got, info = oracle_lib.anchored_dp(k0, k1, lo, hi)
                assert np.array_equal(want.reshape(-1, 2), got), (policy, len(k0), len(k1), lo, hi)
                if len(want):
                    assert info["score"] == score
    finally:
        oracle_lib.set_tie_policy(0)
