"""TEST INFRASTRUCTURE ONLY -- makes the SeqAn pin kit's inputs and expected outputs (cases.txt, expected.json).

Searches seeded random tie-heavy sequence pairs (tiny alphabets, tandem repeats, bands that clip the optimum) for a small set on
which the 12 tie policies of oracle/banded_dp.hpp (6 priority orders of diagonal / vertical / horizontal x first / last maximum
among the border cells) give pairwise DIFFERENT outputs, where "output" is what src/Align4.cpp:1028-1068 produces from SeqAn's
answer: the score and the (ordinal0, ordinal1) pairs of the aligned markers.  Greedy cover of the 66 policy pairs.

    python -m oracle.seqan_pin.make_cases        # rewrites cases.txt and expected.json (deterministic)
"""
import json
import os

import numpy as np

from oracle import bindings, census

HERE = os.path.dirname(os.path.abspath(__file__))
POLICIES = range(12)


def candidates(rng):
    """One random tie-heavy case: (name, seq0, seq1, bandMin, bandMax)."""
    kind = int(rng.integers(0, 4))
    if kind == 0:                       # tiny alphabet
        n = int(rng.integers(6, 28))
        a = rng.integers(0, int(rng.integers(2, 4)), size=n + 10, dtype=np.uint32)
        s0 = a[:n]
        off = int(rng.integers(0, 6))
        s1 = a[off:off + int(rng.integers(5, n + 5))].copy()
        drop = rng.random(len(s1)) < 0.25
        s1 = s1[~drop]
    elif kind == 1:                     # tandem repeat against a copy with one unit dropped
        period = int(rng.integers(1, 4))
        unit = rng.integers(0, 5, size=period, dtype=np.uint32)
        s0 = np.tile(unit, int(rng.integers(4, 12)))
        s1 = np.tile(unit, int(rng.integers(3, 10)))
        off = 0
    elif kind == 2:                     # unique flanks around a repeat
        flank = (10 + np.arange(12)).astype(np.uint32)
        rep = np.full(int(rng.integers(2, 7)), 7, np.uint32)
        rep2 = np.full(int(rng.integers(1, 7)), 7, np.uint32)
        k = int(rng.integers(2, 6))
        s0 = np.concatenate([flank[:k], rep, flank[k:2 * k]])
        s1 = np.concatenate([flank[:k], rep2, flank[k:2 * k]])
        off = 0
    else:                               # two short random sequences over three symbols, free overlap
        s0 = rng.integers(0, 3, size=int(rng.integers(4, 16)), dtype=np.uint32)
        s1 = rng.integers(0, 3, size=int(rng.integers(4, 16)), dtype=np.uint32)
        off = int(rng.integers(-3, 4))
    if len(s0) == 0 or len(s1) == 0:
        return None
    width = int(rng.integers(3, 14))
    lo = off - width // 2 + int(rng.integers(-2, 3))
    hi = lo + width - 1
    if lo > len(s0) or hi < -len(s1):
        return None
    return s0.astype(np.uint32), s1.astype(np.uint32), lo, hi


def output_of(lib, case):
    s0, s1, lo, hi = case
    ordinals, score = lib.banded_dp(s0, s1, lo, hi)
    return [int(score), [[int(x), int(y)] for x, y in np.asarray(ordinals).reshape(-1, 2)]]


def main():
    lib = bindings.OracleLib()
    rng = np.random.default_rng(20240)
    pool = []
    try:
        while len(pool) < 4000:
            case = candidates(rng)
            if case is None:
                continue
            outs = []
            for p in POLICIES:
                lib.set_tie_policy(p)
                outs.append(output_of(lib, case))
            if any(o != outs[0] for o in outs):
                pool.append((case, outs))
    finally:
        lib.set_tie_policy(0)
    # Greedy: the case that separates the most policy pairs not separated yet; ties to the shorter case.
    pairs = {(p, q) for p in POLICIES for q in POLICIES if p < q}
    chosen = []
    while pairs:
        def gain(entry):
            case, outs = entry
            return (sum(1 for p, q in pairs if outs[p] != outs[q]), -(len(case[0]) + len(case[1])))
        best = max(pool, key=gain)
        if gain(best)[0] == 0:
            raise SystemExit("the pool does not separate %s" % sorted(pairs))
        chosen.append(best)
        pairs -= {(p, q) for p, q in pairs if best[1][p] != best[1][q]}
    # A few more on top (the next best by distinct outputs), so that a policy is recognised by more than one case.
    rest = sorted((e for e in pool if not any(e is c for c in chosen)), key=lambda e: (-len({json.dumps(o) for o in e[1]}), len(e[0][0]) + len(e[0][1])))
    chosen += rest[:max(0, 10 - len(chosen))]
    with open(os.path.join(HERE, "cases.txt"), "w") as f:
        f.write("# SeqAn pin kit inputs: one case per line -- name bandMin bandMax | marker kmer ids of sequence 0 | of sequence 1\n")
        for k, (case, outs) in enumerate(chosen):
            s0, s1, lo, hi = case
            f.write("case%02d %d %d | %s | %s\n" % (k, lo, hi, " ".join(str(int(x)) for x in s0), " ".join(str(int(x)) for x in s1)))
    expected = {"policies": {str(p): census.POLICY_NAMES[p] for p in POLICIES},
                "cases": ["case%02d" % k for k in range(len(chosen))],
                "outputs": {str(p): [outs[p] for _, outs in chosen] for p in POLICIES}}
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(expected, f, indent=0, separators=(",", ":"))
        f.write("\n")
    signatures = {p: json.dumps(expected["outputs"][str(p)]) for p in POLICIES}
    assert len(set(signatures.values())) == 12
    print("%d cases; every pair of the 12 policies differs on at least one" % len(chosen))
    for k, (case, outs) in enumerate(chosen):
        print("  case%02d: %d x %d markers, band [%d, %d], %d distinct outputs" % (k, len(case[0]), len(case[1]), case[2], case[3], len({json.dumps(o) for o in outs})))


if __name__ == "__main__":
    main()
