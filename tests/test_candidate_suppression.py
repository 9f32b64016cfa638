"""Assembler::suppressAlignmentCandidates (the host step between the seams in the human Nanopore configurations):
the host layer against the reference's own meta data parser (oracle/_ref: ReadLoader + Reads::getMetaData + atoul,
with the decision of Assembler::suppressAlignment restated around them) and against a plain Python restatement,
on Data/ files the REFERENCE wrote from a FASTA file."""
import os

import numpy as np
import pytest

import shasta_amd.assembler as shasta
from shasta_amd import abi
from tests import host_support

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "shasta_amd", "_build", "libshasta_mi355x_host.so")


def fasta_with_meta_data(path, n_reads=60, seed=5):
    """Reads whose headers carry ONT-style meta data; some fields missing, reordered or odd on purpose."""
    rng = np.random.default_rng(seed)
    headers = []
    with open(path, "w") as f:
        for r in range(n_reads):
            ch = int(rng.integers(1, 6))
            number = int(rng.integers(0, 120)) if r % 7 else 10 ** 12 + r            # a few large read numbers
            run = "runA" if r % 11 else "runB"
            sample = "s1" if r % 13 else "s2"
            fields = ["runid=%s" % run, "sampleid=%s" % sample, "read=%d" % number, "ch=%d" % ch, "start_time=2020-01-01T00:00:%02dZ" % (r % 60)]
            if r % 9 == 0:
                fields = [x for x in fields if not x.startswith("sampleid")]         # missing field: never suppressed
            if r % 10 == 3:
                fields = fields[::-1]
            if r % 17 == 5:
                fields = []                                                          # no meta data at all
            if r % 19 == 7:
                fields = ["channel=3"] + fields + ["ch="]                            # a longer key and an empty value
            meta = " ".join(fields)
            headers.append(("read%03d" % r, meta))
            f.write(">read%03d%s\n" % (r, (" " + meta) if meta else ""))
            f.write("".join(rng.choice(list("ACGT"), size=int(rng.integers(40, 90)))) + "\n")
    return headers


def python_restatement(headers, pairs, delta):
    def value(meta, key):
        for token in meta.split():
            if len(token) > len(key) + 1 and token.startswith(key + "="):
                return token[len(key) + 1:]
        return ""
    out = []
    for r0, r1 in pairs:
        m0, m1 = headers[r0][1], headers[r1][1]
        ok = True
        for key in ("ch", "sampleid", "runid"):
            v0, v1 = value(m0, key), value(m1, key)
            if not v0 or not v1 or v0 != v1:
                ok = False
                break
        if ok:
            a, b = value(m0, "read"), value(m1, "read")
            ok = bool(a) and bool(b) and abs(int(a) - int(b)) < delta
        out.append(1 if ok else 0)
    return np.asarray(out, np.uint8)


@pytest.mark.parametrize("delta", [30, 1, 1000])
def test_suppression_equals_reference(ref_lib, tmp_path, monkeypatch, delta):
    fasta = str(tmp_path / "reads.fasta")
    headers = fasta_with_meta_data(fasta)
    n = len(headers)
    rng = np.random.default_rng(delta)
    r0 = rng.integers(0, n - 1, size=700)
    r1 = np.minimum(n - 1, r0 + 1 + rng.integers(0, 6, size=700))
    same = rng.integers(0, 2, size=700)
    d = str(tmp_path / "Data")
    os.makedirs(d)
    expected, read_count = ref_lib.suppress_alignment_flags(fasta, d, r0, r1, delta)      # also writes Data/ReadNames, Data/ReadMetaData
    assert read_count == n
    assert np.array_equal(expected, python_restatement(headers, list(zip(r0, r1)), delta))
    assert expected.sum() < len(expected) and (delta == 1 or expected.sum() > 0)      # equal read numbers (delta 1) may not occur
    candidates = abi.make_pairs(r0, r1, same)
    host_support.HostShim().store_candidates(d, candidates)
    monkeypatch.chdir(tmp_path)
    a = shasta.Assembler(hostLibrary=HOST_SO)
    assert a.suppressAlignmentCandidates(delta) == int(expected.sum())
    stored, _ = host_support.HostShim().open_vector(os.path.join(d, "AlignmentCandidates"), 12)
    kept = expected == 0
    assert np.array_equal(stored.view("<u4").reshape(-1, 3)[:, 0], r0[kept].astype(np.uint32))
    assert np.array_equal(stored.view("<u4").reshape(-1, 3)[:, 1], r1[kept].astype(np.uint32))
    assert np.array_equal(stored[:, 8], same[kept].astype(np.uint8))
    # The same decision on arrays in memory (what bench.py's configs[3] / [4] steps run between the seams).
    meta = [h[1].encode() for h in headers]
    meta_toc = np.concatenate([[0], np.cumsum([len(m) for m in meta])]).astype(np.uint64)
    in_memory = shasta.suppress_candidates_in_memory(candidates, (meta_toc, np.frombuffer(b"".join(meta) + b" ", dtype=np.uint8)), delta, hostLibrary=HOST_SO)
    assert np.array_equal(in_memory["readId0"], r0[kept].astype(np.uint32)) and np.array_equal(in_memory["readId1"], r1[kept].astype(np.uint32))
    assert np.array_equal(in_memory["isSameStrand"], same[kept].astype(np.uint8))
    # ... and from keys made once per read, on one thread and on several (slices of a list this short are a few candidates each).
    meta_data = (meta_toc, np.frombuffer(b"".join(meta) + b" ", dtype=np.uint8))
    for threads in (1, 5):
        by_keys = shasta.CandidateSuppression(meta_data, delta, hostLibrary=HOST_SO, threads=threads).apply(candidates)
        assert np.array_equal(by_keys, in_memory)
    # The side file, src/AssemblerAlign.cpp:1187-1203: one row per suppressed candidate, names and meta data verbatim.
    rows = open(tmp_path / "SuppressedAlignmentCandidates.csv").read().splitlines()
    assert rows[0] == "ReadId0,ReadId1,SameStrand,Name0,Name1,MetaData0,MetaData1" and len(rows) == 1 + int(expected.sum())
    if not expected.any():
        return
    first = int(np.nonzero(expected)[0][0])
    assert rows[1] == "%d,%d,%s,%s,%s,%s,%s" % (r0[first], r1[first], "Yes" if same[first] else "No", headers[r0[first]][0], headers[r1[first]][0],
                                                 headers[r0[first]][1], headers[r1[first]][1])
