"""TEST INFRASTRUCTURE ONLY -- the DP tie census.

Align4's banded DP is SeqAn's (src/Align4.cpp:1028-1033); SeqAn is absent here, so which predecessor a cell keeps when two
tie, and which border cell ends the alignment when several tie on the maximum, is a READING of SeqAn 2.4.0 (oracle/banded_dp.hpp)
that the HIP kernels, the restatement, the fixtures and the CPU baseline all share.  This module sizes what depends on it: the
same candidates are aligned by the checker library (the reference's own Align4 control flow in oracle/_ref, or the restatement)
under each of the 11 other policies -- the 6 priority orders of (diagonal, vertical, horizontal) x {first, last} maximum --
and compared, candidate by candidate, with the result under the reading:

    candidates_changed      any difference in status or aligned ordinals under at least one alternative policy
    markerCount_changed     the number of aligned markers differs under at least one
    stored_set_changed      stored <-> not stored under at least one
    per_policy              the three counts for every alternative on its own

A candidate that no alternative changes is one whose result does not depend on the reading at all.
"""
import numpy as np

from shasta_amd import abi

POLICY_NAMES = ["%s, %s maximum" % (order, end) for order in ("D>=V>=H", "D>=H>=V", "V>=D>=H", "V>=H>=D", "H>=D>=V", "H>=V>=D")
                for end in ("first", "last")]


def _summary(out):
    stored = (out.status & 0x7f) == abi.SHASTA_ALIGN_STORED
    counts = np.diff(out.ordinals_toc.astype(np.int64))
    return stored, counts


def _only_the_reads_of(toc, data7, candidates):
    reads = np.unique(np.concatenate([candidates["readId0"], candidates["readId1"]])) if len(candidates) else np.zeros(0, np.uint32)
    if len(reads) == 0 or 2 * len(reads) + 1 >= len(toc):
        return toc, data7, candidates
    toc = np.asarray(toc, dtype=np.uint64)
    begins, ends = toc[2 * reads.astype(np.int64)].astype(np.int64), toc[2 * reads.astype(np.int64) + 2].astype(np.int64)
    sizes = np.stack([toc[2 * reads.astype(np.int64) + 1].astype(np.int64) - begins, ends - toc[2 * reads.astype(np.int64) + 1].astype(np.int64)], axis=1).reshape(-1)
    sub_toc = np.zeros(2 * len(reads) + 1, dtype=np.uint64)
    sub_toc[1:] = np.cumsum(sizes)
    records = np.asarray(data7, dtype=np.uint8).reshape(-1, 7)
    lengths = ends - begins
    rows = np.repeat(begins - (np.cumsum(lengths) - lengths), lengths) + np.arange(int(lengths.sum()))
    sub_data = np.ascontiguousarray(records[rows]).reshape(-1)
    renumbered = candidates.copy()
    renumbered["readId0"] = np.searchsorted(reads, candidates["readId0"])
    renumbered["readId1"] = np.searchsorted(reads, candidates["readId1"])
    return sub_toc, sub_data, renumbered


def tie_census(lib, toc, data7, candidates, options, align_method=4, threads=1, policies=range(1, 12)):
    """`lib`: oracle.bindings.RefLib or OracleLib.  Leaves the library on policy 0."""
    align = lib.align4_batch if align_method == 4 else lib.align3_batch
    n = len(candidates)
    # Twelve calls, each of which prepares the markers of every read it is given: give it the census's reads only,
    # renumbered (at the benchmark's size a third of the reads; 162 -> about 60 seconds for 20 000 candidates).
    toc, data7, candidates = _only_the_reads_of(toc, data7, candidates)
    try:
        lib.set_tie_policy(0)
        base = align(toc, data7, candidates, options, want_ordinals=True, threads=threads)
        base_stored, base_counts = _summary(base)
        base_status = base.status & 0x7f
        any_changed = np.zeros(n, dtype=bool)
        count_changed = np.zeros(n, dtype=bool)
        stored_changed = np.zeros(n, dtype=bool)
        per_policy = {}
        for index in policies:
            lib.set_tie_policy(index)
            alt = align(toc, data7, candidates, options, want_ordinals=True, threads=threads)
            stored, counts = _summary(alt)
            differs = (alt.status & 0x7f) != base_status
            differs |= counts != base_counts
            # Same count: compare the ordinals themselves (only where the ranges agree).
            same = np.flatnonzero(~differs & (counts > 0))
            if len(same):
                a0, b0 = base.ordinals_toc[same].astype(np.int64), alt.ordinals_toc[same].astype(np.int64)
                lengths = counts[same]
                # One vectorised comparison over all of them: gather row indices.
                rows = np.repeat(np.arange(len(same)), lengths)
                within = np.arange(int(lengths.sum())) - np.repeat(np.cumsum(lengths) - lengths, lengths)
                unequal = np.any(base.ordinals[a0[rows] + within] != alt.ordinals[b0[rows] + within], axis=1)
                bad = np.zeros(len(same), dtype=bool)
                np.logical_or.at(bad, rows, unequal)
                differs[same[bad]] = True
            any_changed |= differs
            count_changed |= counts != base_counts
            stored_changed |= stored != base_stored
            per_policy[POLICY_NAMES[index]] = {"candidates_changed": int(differs.sum()), "markerCount_changed": int((counts != base_counts).sum()),
                                               "stored_set_changed": int((stored != base_stored).sum())}
    finally:
        lib.set_tie_policy(0)
    return {"candidates": int(n), "stored_under_the_reading": int(base_stored.sum()),
            "candidates_changed": int(any_changed.sum()), "markerCount_changed": int(count_changed.sum()),
            "stored_set_changed": int(stored_changed.sum()), "per_policy": per_policy,
            "reading": POLICY_NAMES[0], "checker": type(lib).__name__}
