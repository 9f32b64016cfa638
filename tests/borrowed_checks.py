"""Borrowed aligner results (align4_run_borrowed) over SEVERAL batches: a batch whose predecessors are done is copied into the
context's arrays while later batches are still running; the result must equal the owned one, call after call (the second call
finds arrays sized by the first and places every batch early; a third, larger call outgrows them again).
And a call of several batches against the oracle, prepared by the host and with the batches' first chunk lists made on the
device (an experiment switch).
Run in a process of its own (the batch size is read once per process):
    SHASTA_MI355X_ALIGN_BATCH_LOG2=10 python -m tests.borrowed_checks <library.so> [oracle [both-preparations]]"""
import os
import sys

import numpy as np


def main(path, oracle=None, device_prepare=None):
    from shasta_amd import abi, lib as libmod
    from tests import support
    lib = libmod.Library(path)
    toc, kmer, data7 = support.small_marker_set(n_reads=160, genome_markers=9000, seed=77)
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=40, minFrequency=1)
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    with lib.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer)
        cand = ctx.lowhash0(p).candidates
        assert len(cand) > 1300, len(cand)                    # two batches of 1024
        for count, wants in ((1100, (False,)), (1100, (True,)), (1300, (True,)), (300, (False,))):
            part = cand[:count]
            for want in wants:
                owned = ctx.align4(part, o, want_ordinals=want)
                kept = [owned.status.copy(), owned.info_table(), owned.compressed_toc.copy(), owned.compressed_data.copy(),
                        None if not want else owned.ordinals.copy()]
                borrowed = ctx.align4(part, o, want_ordinals=want, borrow=True)
                assert np.array_equal(borrowed.status, kept[0]) and np.array_equal(borrowed.info_table(), kept[1])
                assert np.array_equal(borrowed.compressed_toc, kept[2]) and np.array_equal(borrowed.compressed_data, kept[3])
                if want:
                    assert np.array_equal(borrowed.ordinals, kept[4])
                del borrowed
        # 2060 candidates in three batches.
        assert len(cand) >= 2048, len(cand)
        equal = ctx.align4(cand, o, want_ordinals=True)
        # The three batches finish in whatever order their workers get to them: a batch whose predecessors are done goes from the
        # worker's staging buffer straight to its place, one that finishes early waits in a vector of its own.  Several calls, so
        # that both happen.
        for _ in range(6):
            again = ctx.align4(cand, o, want_ordinals=False, borrow=True)
            assert np.array_equal(again.status, equal.status) and np.array_equal(again.info_table(), equal.info_table())
            assert np.array_equal(again.compressed_toc, equal.compressed_toc) and np.array_equal(again.compressed_data, equal.compressed_data)
            del again
        if device_prepare:
            os.environ["SHASTA_MI355X_DEVICE_BATCH_PREP"] = "0"              # every batch's first chunk lists made by the host loop instead of kernels (align4_prepare.hpp)
            prepared = ctx.align4(cand, o, want_ordinals=True)
            del os.environ["SHASTA_MI355X_DEVICE_BATCH_PREP"]
            support.same_align(prepared, equal)
        if oracle:
            from oracle import bindings
            expected = bindings.OracleLib().align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
            if not (expected.status & 0x80).any():
                support.same_align(expected, equal)
            else:
                assert np.array_equal(expected.status, equal.status)
    print("borrowed results equal owned results")


if __name__ == "__main__":
    main(*sys.argv[1:4])
