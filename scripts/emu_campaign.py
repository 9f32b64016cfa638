#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: seeded read sets through the aligner of the EMULATED build (tests/emu) against the oracle, candidate for candidate
(AlignmentData, the compressed bytes, the ordinals where asked for).   python scripts/emu_campaign.py <first seed> <seeds> [processes]
Every seed draws its own read count, genome length, read length and MinHash parameters; odd seeds ask for the ordinals.
CAMPAIGN_LONG=1: a few dozen reads of 3 000 to 7 000 markers each.  CAMPAIGN_UL=1: a dozen reads of 9 000 to 16 000 markers (pairs of two long
reads: the windowed cells class, the sort and wave kernels' largest classes; the match-rate estimate set as a k = 14 alphabet's would be).
The switches of the build and of the emulator apply as everywhere (SHASTA_MI355X_CELLS_FORCE=long|big, SHASTA_MI355X_SCRAMBLE=1,
HIPEMU_LDS_SCRAMBLE, HIPEMU_SCHEDULE)."""
import os
import sys
from multiprocessing import Pool

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def one(seed):
    import numpy as np
    from oracle import bindings
    from shasta_amd import abi, lib as L, synthetic
    from tests import support
    rng = np.random.default_rng(10_000 + seed)
    n_reads = int(rng.integers(50, 150))
    genome = int(rng.integers(4000, 16000))
    mean = float(rng.choice([500.0, 900.0, 1600.0]))
    alphabet = None
    if os.environ.get("CAMPAIGN_LONG") == "1":            # (reads of thousands of markers: the wave kernel's larger capacity classes)
        n_reads, genome, mean = int(rng.integers(24, 48)), int(rng.integers(12000, 30000)), float(rng.choice([3000.0, 5000.0, 7000.0]))
    if os.environ.get("CAMPAIGN_UL") == "1":
        n_reads, genome, mean = int(rng.integers(8, 14)), int(rng.integers(24000, 48000)), float(rng.choice([9000.0, 12000.0, 16000.0]))
        os.environ.setdefault("SHASTA_MI355X_MATCH_SHIFT", "20")
        alphabet = synthetic.sampled_marker_alphabet(14, count=40000)        # (79 000 ids: few random matches, as at k = 14)
    toc, kmer = synthetic.marker_reads(n_reads, genome, mean_markers=mean, min_markers=200, seed=seed, alphabet=alphabet)
    data7 = synthetic.pack_markers(toc, kmer)
    orc = bindings.OracleLib()
    emu = L.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "emu", "_build", "libshasta_mi355x_emu.so"))
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=int(rng.choice([20, 30, 50])), minFrequency=int(rng.choice([1, 2])))
    cand = orc.lowhash0(toc, data7, None, p).candidates[:250]
    o = abi.default_align4_options(minAlignedMarkerCount=int(rng.choice([10, 40, 100])))
    ordinals = bool(seed & 1)
    want = orc.align4_batch(toc, data7, cand, o, want_ordinals=ordinals, threads=0)
    with emu.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer)
        got = ctx.align4(cand, o, want_ordinals=ordinals)
    ties = (want.status & 0x80) != 0
    if ties.any():
        assert want.per_candidate(~ties) == got.per_candidate(~ties), seed
    else:
        support.same_align(want, got)
    return len(cand), len(got.alignment_data)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    processes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    with Pool(processes) as pool:
        rows = pool.map(one, range(first, first + count), chunksize=1)
    print("seeds %d..%d: %d candidates, %d stored alignments, all equal to the oracle" % (first, first + count - 1, sum(r[0] for r in rows), sum(r[1] for r in rows)))


if __name__ == "__main__":
    main()
