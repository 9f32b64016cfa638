"""Several devices behind one call (shasta_mi355x_group, *_multi): results identical to the one-device entry points
and to the oracle for any device count.  On a box with one GPU (and on the emulated build) the device list names
device 0 several times: every exchange, split and reduction of the sharded path still runs."""
import numpy as np

from shasta_amd import abi
from tests import support


def lowhash0_and_aligners(lib, oracle_lib, device_lists=((0, 0), (0, 0, 0)), n_reads=260, limit=900):
    toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=14000, seed=61)
    flags = np.zeros(n_reads, np.uint8)
    flags[[3, n_reads - 2]] = 1
    cases = [abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=2),
             abi.default_lowhash0_params(m=5, hashFraction=0.03, minHashIterationCount=0, alignmentCandidatesPerRead=6.0, minBucketSize=2, maxBucketSize=40)]
    o4 = abi.default_align4_options(minAlignedMarkerCount=40)
    o3 = abi.default_align3_options(minAlignedMarkerCount=40)
    compared = 0
    for p in cases:
        ref = oracle_lib.lowhash0(toc, data7, flags, p)
        assert len(ref.candidates) > 100
        for devices in device_lists:
            out = lib.lowhash0_multi(toc, data7, flags, p, devices)
            support.same_lowhash(out, ref)
            compared += 1
    cand = ref.candidates[:limit]
    x4 = oracle_lib.align4_batch(toc, data7, cand, o4, want_ordinals=True, threads=0)
    x3 = oracle_lib.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=0)
    for devices in device_lists:
        y4 = lib.align4_batch_multi(toc, data7, cand, o4, devices, want_ordinals=True)
        y3 = lib.align3_batch_multi(toc, data7, cand, o3, devices, want_ordinals=True)
        if not (x4.status & 0x80).any():
            support.same_align(x4, y4)
        else:
            assert x4.per_candidate((x4.status & 0x80) == 0) == y4.per_candidate((x4.status & 0x80) == 0)
        support.same_align(x3, y3)
        assert y4.dp_cell_count == x4.dp_cell_count
    # The resident form: markers stay on every device of the group between calls.
    with lib.group(device_lists[0]) as g:
        g.set_kmer_ids(toc, kmer, flags)
        a = g.lowhash0(cases[0])
        support.same_lowhash(a, oracle_lib.lowhash0(toc, data7, flags, cases[0]))
        b = g.align4(cand, o4, want_ordinals=True)
        if not (x4.status & 0x80).any():
            support.same_align(x4, b)
        # Results owned by the group (valid until its next aligner call): the same arrays, twice over the same buffers.
        for _ in range(2):
            c = g.align4(cand, o4, want_ordinals=True, borrow=True)
            support.same_align(b, c)
            del c
        d = g.align3(cand, o3, want_ordinals=True, borrow=True)
        support.same_align(x3, d)
        del d
    # Fewer candidates than devices, and none at all.
    few = lib.align4_batch_multi(toc, data7, cand[:2], o4, (0, 0, 0), want_ordinals=True)
    assert np.array_equal(few.status, x4.status[:2])
    none = lib.align4_batch_multi(toc, data7, cand[:0], o4, (0, 0), want_ordinals=True)
    assert len(none.status) == 0 and len(none.alignment_data) == 0
    return compared


class _Sharded:
    """`lib` with lowhash0 routed through the device list: what tests/adversarial.py's LowHash0 cases call."""
    def __init__(self, lib, devices):
        self.lib, self.devices = lib, devices

    def lowhash0(self, toc, data7, flags, params):
        return self.lib.lowhash0_multi(toc, data7, flags, params, self.devices)


def adversarial_lowhash0(lib, oracle_lib, devices=(0, 0, 0)):
    """The adversarial LowHash0 parameter sets and read sets (hashFraction 0 / 1 / 1.5, 2^31 buckets, MinHash 10/50/5, a forced
    uint16 wrap, the dynamic iteration control, empty and repeated reads ...) through the sharded job: all iterations in one
    pass where their number is fixed (empty exchanges, 64-bit keys, bucket sizes beyond the histogram bins), iteration after
    iteration otherwise."""
    from tests import adversarial
    sharded = _Sharded(lib, devices)
    for name in adversarial.LOWHASH_CASE_NAMES:
        adversarial.lowhash_case(sharded, oracle_lib, name)
    for name in adversarial.LOWHASH_READ_SET_NAMES:
        adversarial.lowhash_read_set(sharded, oracle_lib, name)
    return len(adversarial.LOWHASH_CASE_NAMES) + len(adversarial.LOWHASH_READ_SET_NAMES)


def errors_do_not_hang(lib):
    import pytest
    toc, kmer, data7 = support.small_marker_set(n_reads=60, genome_markers=5000, seed=62)
    with pytest.raises(RuntimeError, match="unreasonably small"):
        lib.lowhash0_multi(toc, data7, None, abi.default_lowhash0_params(log2MinHashBucketCount=3), (0, 0))
    with pytest.raises(RuntimeError):
        lib.lowhash0_multi(toc, data7, None, abi.default_lowhash0_params(), (0, 99))      # no such device


def one_pass_that_does_not_fit(lib, oracle_lib, devices=(0, 0, 0)):
    """SHASTA_MI355X_ONE_PASS_RECORD_LIMIT lowered until the records of all iterations do not fit "one sort": the group (every
    device asks its own context, all must agree) and the one-device call fall back to iteration after iteration -- same results."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import bindings; from shasta_amd import abi, lib as libmod; from tests import support\n"
            "lib = libmod.Library(%r); orc = bindings.OracleLib()\n"
            "toc, kmer, data7 = support.small_marker_set(n_reads=150, genome_markers=9000, seed=64)\n"
            "p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=2)\n"
            "ref = orc.lowhash0(toc, data7, None, p)\n"
            "support.same_lowhash(lib.lowhash0_multi(toc, data7, None, p, %r), ref)\n"
            "support.same_lowhash(lib.lowhash0(toc, data7, None, p), ref)\n"
            "print('fell back and agreed')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), lib.path, tuple(devices))
    # (a process of its own: the limit is read once per process)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SHASTA_MI355X_ONE_PASS_RECORD_LIMIT="100000", SHASTA_MI355X_DEBUG_ONE_PASS="1"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "fell back and agreed" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    assert out.stderr.count("iteration after iteration") >= 1 + len(devices)          # the one-device call and every device of the group
    return out.stderr


class _environment:
    def __init__(self, **values):
        self.values = values

    def __enter__(self):
        import os
        self.previous = {k: os.environ.get(k) for k in self.values}
        os.environ.update(self.values)

    def __exit__(self, *exc):
        import os
        for k, v in self.previous.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def staged_job_of_one_device(lib, oracle_lib, transport="peer", n_reads=260):
    """SHASTA_MI355X_GROUP_STAGED=1: a group of ONE device runs the staged LowHash0 -- begin, hash, exchange, buckets, exchange,
    merge -- with a world of one, the device exchanging with itself.  transport "rccl": both exchanges as grouped ncclSend / ncclRecv
    on a communicator from ncclCommInitAll (multi.hip's second transport; all of it that a one-GPU box can run)."""
    toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=14000, seed=67)
    compared = 0
    for p in (abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=2),
              abi.default_lowhash0_params(m=5, hashFraction=0.03, minHashIterationCount=0, alignmentCandidatesPerRead=6.0, minBucketSize=2, maxBucketSize=40)):
        ref = oracle_lib.lowhash0(toc, data7, None, p)
        assert len(ref.candidates) > 100
        with _environment(SHASTA_MI355X_GROUP_STAGED="1", SHASTA_MI355X_GROUP_TRANSPORT=transport):
            out = lib.lowhash0_multi(toc, data7, None, p, (0,))
        support.same_lowhash(out, ref)
        compared += 1
    return compared
