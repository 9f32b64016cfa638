#!/usr/bin/env python3
"""Benchmark of the overlap-detection hot path (LowHash0 + Align4) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over the workload: LowHash0 (10 MinHash
iterations) on the resident markers -> candidate list -> Assembler::computeAlignments
(src/AssemblerAlign.cpp:208-304) end to end: Align4 on every candidate, AlignmentData +
CompressedAlignments on the host, and the alignment table (computeAlignmentTable, :296).  Inputs (kmer ids, toc) are resident
in HBM before the timed region.  Workload: BASELINE.json configs[2]
("Synthetic 100k reads, 1xMI355X, LowHash0 + Align4 banded marker alignment end-to-end"),
generated at marker level (shasta_amd/synthetic.py; SURVEY F5: the path never reads bases).

Prints ONE JSON line on rank 0.  `value` = candidate read pairs aligned per second, whole job.
`kernels` has one row per kernel of the path (HIP-event times on the stream each is launched
on, the library's kernel table); `roofline` describes the one with the largest total time.
`cpu_baseline` (N = 1 only): the reference's own code (oracle/_ref, compiled in place from
/root/reference) on the SAME read set on this host's cores -- LowHash0 in full, the aligner
on a sample of its candidates -- which doubles as the parity check at the benchmark's own
size (`parity_at_bench_size`): the checker is never inside the timed region.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime deals a process's streams to GPU_MAX_HW_QUEUES hardware queues (4 by default), and streams that share a
# queue run one after the other.  An aligner call keeps six workers' streams busy (plus their side streams); which of them
# share a queue depends on what else created streams before them -- with torch and RCCL initialised first (the N-rank branch)
# the same call took 185 ms instead of 146.  Eight queues: 146 and 158 (read by the runtime when it starts,
# so it has to be in the environment before the first HIP call of the process).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# Vector ALU peak in wavefront instructions per second: 256 CUs x 4 SIMDs x one wave64 instruction per 2 cycles x 2.4 GHz
# (the guide's 157.3 TFLOP/s fp32 = 2 flops x 64 lanes x this).  Dependent integer chains measured on the device reach
# 1.35-1.55 of the 2 instructions per cycle and CU (scripts/microbench/valu_rates.hip, profiles/r02_valu_rates.jsonl).
VALU_PEAK_WAVE_INSTRUCTIONS_PER_S = 256 * 4 * 0.5 * 2.4e9
def _latest_pmc_file():
    # The newest round's counter summary (profiles/rNN_pmc_100k_reads.json, written by scripts/gpu_profile.sh).
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_100k_reads.json")))
    return files[-1] if files else os.path.join(ROOT, "profiles", "r02_pmc_100k_reads.json")


PMC_FILE = _latest_pmc_file()
# Kernels whose natural bound is HBM (streaming / sorting); the others are bound by VALU issue and LDS (integer DP, hash joins).
HBM_NATURED = ("hashWindowsKernel", "radix sort", "bucket", "readStatistics", "pairWriteKernel", "evaluatePairs", "emitCandidates", "compress", "markerSweep", "packMarkers")


# The two read-set shapes of the bench (both at marker level, 45x coverage): BASELINE configs[2] -- mean 1 500 markers (~20 kb) per read --
# and the shape of configs[4]'s conf/Nanopore-UL-May2022.conf -- minReadLength 50 000 bases: no read below 3 750 markers, mean 7 500
# (~100 kb), a tail beyond 30 000.
# (round 6) ... and the shape of configs[3]'s conf/Nanopore-May2022.conf -- minReadLength 10 000 bases: no read below 750 markers, mean 1 875
# (~25 kb).  Both human configurations set Kmers.k = 14: their reads are generated over a k = 14 marker alphabet (about 609 000 run-length
# 14-mers closed under reverse complement: 4 x 3^13 run-length 14-mers at markerDensity 0.1 are 638 000 -- sixty times fewer random
# marker matches per pair of reads than the 7 900 marker k-mers of k = 10), and their steps run Assembler::suppressAlignmentCandidates
# (Align.sameChannelReadAlignment.suppressDeltaThreshold = 30) between the two stages as srcMain/main.cpp:697-702 does.
WORKLOAD_SHAPES = {"configs2": dict(mean_markers=1500.0, sigma=0.5, min_markers=790), "ul": dict(mean_markers=7500.0, sigma=0.5, min_markers=3750),
                   "may2022": dict(mean_markers=1875.0, sigma=0.5, min_markers=750)}
WORKLOAD_K = {"configs2": 10, "ul": 14, "may2022": 14}
WORKLOAD_SUPPRESS_DELTA = {"configs2": 0, "ul": 30, "may2022": 30}
WORKLOAD_SHAPE = ["configs2"]          # (set by main() from --workload)
_ALPHABET = {}


def workload_alphabet():
    """None (synthetic.marker_reads' default: every marker k-mer of k = 10) or the sampled k = 14 alphabet."""
    k = WORKLOAD_K[WORKLOAD_SHAPE[0]]
    if k == 10:
        return None
    if k not in _ALPHABET:
        from shasta_amd import synthetic
        _ALPHABET[k] = synthetic.sampled_marker_alphabet(k, count=320000)
    return _ALPHABET[k]


def make_workload(n_reads, seed, shards=0):
    """shards >= 1: the read set of an N-rank run -- n_reads / shards reads per rank from one genome, each rank's from its own
    stream, as the ranks generate them -- in one piece (the in-process group's line beside an RCCL run is about the same reads)."""
    from shasta_amd import synthetic
    # 45x coverage: n_reads * 1500 genome markers per read / genome markers.
    shape = WORKLOAD_SHAPES[WORKLOAD_SHAPE[0]]
    genome_markers = max(20000, int(round(n_reads * shape["mean_markers"] / 45.0)))
    if shards >= 1:
        parts = [synthetic.marker_reads(n_reads // shards, genome_markers,
                                        keep_probability=0.8, spurious_probability=0.25, k=10, seed=seed, shard=r, shard_count=shards, alphabet=workload_alphabet(), **shape)
                 for r in range(shards)]
        sizes = np.concatenate([np.diff(t.astype(np.int64)) for t, _ in parts])
        toc = np.zeros(len(sizes) + 1, dtype=np.uint64)
        toc[1:] = np.cumsum(sizes)
        return toc, np.concatenate([k for _, k in parts])
    # SHASTA_BENCH_WORKLOAD_CACHE=<directory>: measurement scripts that run this command several times in one GPU call keep
    # the generated read set (a minute of host time per run) in a scratch directory; same arrays either way.
    cache = os.environ.get("SHASTA_BENCH_WORKLOAD_CACHE")
    if cache:
        files = [os.path.join(cache, "workload_%s_%d_%d_%s.npy" % (WORKLOAD_SHAPE[0], n_reads, seed, x)) for x in ("toc", "kmer")]
        if all(os.path.exists(f) for f in files):
            return np.load(files[0]), np.load(files[1])
    toc, kmer = synthetic.marker_reads(n_reads, genome_markers, keep_probability=0.8, spurious_probability=0.25, k=10, seed=seed, alphabet=workload_alphabet(), **shape)
    if cache:
        os.makedirs(cache, exist_ok=True)
        np.save(files[0], toc); np.save(files[1], kmer)
    return toc, kmer


def make_meta_data(n_reads, candidates, seed=4321, share=0.001):
    """ONT-style read meta data (`runid= sampleid= read= ch= start_time=`, the keys Assembler::suppressAlignment reads,
    src/AssemblerAlign.cpp:1078-1162) for the synthetic reads, in the layout of Data/ReadMetaData (toc uint64[R + 1], bytes): every read
    its own channel-and-number, except that the second read of about one candidate in a thousand comes from the channel of the first
    with the next read number -- the two strands of one molecule through one pore, what the suppression step is there to drop."""
    rng = np.random.default_rng(seed)
    channel = rng.integers(1, 2049, size=n_reads)
    number = rng.integers(0, 200000, size=n_reads) * 100          # (far apart: only the pairs made below are within delta)
    if len(candidates):
        pick = rng.choice(len(candidates), size=max(1, int(len(candidates) * share)), replace=False)
        taken = np.zeros(n_reads, dtype=bool)
        for i in np.sort(pick):
            r0, r1 = int(candidates["readId0"][i]), int(candidates["readId1"][i])
            if taken[r0] or taken[r1]:
                continue
            taken[r0] = taken[r1] = True
            channel[r1] = channel[r0]
            number[r1] = number[r0] + 1
    rows = [("runid=0f3c9a sampleid=s1 read=%d ch=%d start_time=2022-05-01T00:00:00Z" % (number[r], channel[r])).encode() for r in range(n_reads)]
    toc = np.zeros(n_reads + 1, dtype=np.uint64)
    toc[1:] = np.cumsum([len(x) for x in rows])
    return toc, np.frombuffer(b"".join(rows) + b" ", dtype=np.uint8)


def lowhash_params():
    from shasta_amd import abi
    # m=4, f=0.01, 10 iterations, minBucketSize/maxBucketSize/minFrequency 5/30/5.
    if WORKLOAD_SHAPE[0] == "ul":          # conf/Nanopore-UL-May2022.conf: MinHash 10/50/5
        return abi.default_lowhash0_params(minBucketSize=10, maxBucketSize=50, minFrequency=5)
    return abi.default_lowhash0_params(minBucketSize=5, maxBucketSize=30, minFrequency=5)       # (conf/Nanopore-May2022.conf: the same 5/30/5)


def align3_options():
    # alignMethod 3 with the values of the shipped Nanopore configurations (downsamplingFactor 0.05) on
    # top of the defaults of src/AssemblerOptions.cpp:419-449; k = 10 as the synthetic marker alphabet.
    from shasta_amd import abi
    return abi.default_align3_options(downsamplingFactor=0.05)


def align_options():
    from shasta_amd import abi
    if WORKLOAD_SHAPE[0] in ("ul", "may2022"):          # conf/Nanopore-May2022.conf and -UL-, [Align]: maxSkip / maxDrift / maxTrim 100, minAlignedMarkerCount 10, minAlignedFraction 0.1
        return abi.default_align4_options(maxSkip=100, maxDrift=100, maxTrim=100, minAlignedMarkerCount=10, minAlignedFraction=0.1)
    return abi.default_align4_options()


def candidate_paths(table, steps, candidates):
    """Which kernels a step's candidates took (the kernel table's work counts: candidates per cells-kernel class -- one that overflows
    its class's tables is counted again in the class it climbs to -- and what the banded DP's tasks ended in)."""
    per = {}
    for name, r in table.items():
        if name.startswith("align4CellsChunkKernel") or name.startswith("align4CellsKernel") or name.startswith("align4CellsLongKernel"):
            per[name] = r["work"] / steps
    out = {"cells_kernel_candidates_per_step": per,
           "share_in_the_LDS_chunk_kernels": (sum(v for k, v in per.items() if k.startswith("align4CellsChunkKernel") or k.startswith("align4CellsLongKernel")) / candidates) if candidates else None,
           "share_in_the_windowed_LDS_kernel": (sum(v for k, v in per.items() if k.startswith("align4CellsLongKernel")) / candidates) if candidates else None,
           "share_in_the_HBM_scratch_kernel": (sum(v for k, v in per.items() if k.startswith("align4CellsKernel")) / candidates) if candidates else None}
    return out


def load_pmc(reads):
    """Per-kernel counters of the committed rocprofv3 --pmc passes over this very command (scripts/pmc_summary.py):
    HBM bytes per launch (FETCH_SIZE x 2 as the guide prescribes for gfx950, + WRITE_SIZE) and VALU wave-instructions
    per launch.  Only valid for the workload they were collected on."""
    if not os.path.exists(PMC_FILE):
        return {}
    with open(PMC_FILE) as f:
        d = json.load(f)
    # Counters describe the build they were collected on: another build's instruction counts over this build's launch times
    # are not evidence (round 4 printed VALU fractions above 1 that way).  No match, no counters.
    import shasta_amd
    if d.get("workload_reads") != reads or d.get("kernel_source_hash") != shasta_amd.kernel_source_hash():
        return {}
    return d.get("kernels", {})


def pmc_of(pmc, name):
    key = name.replace(" ", "")
    for k, v in pmc.items():
        if k.replace(" ", "") == key:
            return v
    # A row of the kernel table that stands for one launch of each of several template instances (the wave kernel's capacity
    # classes, the anchor kernel's two launches, the sort kernel's two): their per-launch counters added up.
    parts = [v for k, v in pmc.items() if k.replace(" ", "").startswith(key + "<")]
    if parts and "<" not in key:
        total = {}
        for v in parts:
            for field in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "valu_wave_instructions_per_launch"):
                if v.get(field) is not None:
                    total[field] = total.get(field, 0.0) + v[field]
        total["instances"] = len(parts)
        return total
    return None


DRY_RUN_LIBRARY = os.environ.get("SHASTA_BENCH_LIBRARY")      # see main(): pre-flight without a GPU, never a result


def available_memory_gib():
    """MemAvailable of /proc/meminfo in GiB (a large number when it cannot be read)."""
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) >> 20
    except OSError:
        pass
    return 1 << 20


def cpu_baseline(ctx, toc, kmer, p, o, align_method, gpu_lowhash, sample_size, census_size=0, all_rows=None, suppress=None):
    """The reference CPU path on the SAME read set, on this host's cores, outside the timed region; its outputs
    are compared with the device's (parity at the benchmark's own size).  LowHash0 runs in full; the aligner
    on every (candidates / sample_size)-th candidate (the whole list would take minutes)."""
    from oracle import bindings
    from shasta_amd import synthetic
    host_cores = os.cpu_count() or 1
    # Threads actually used = "cores" of the report.  Capped at 64: every reference Align4 thread zero-fills its own
    # 2 GiB arena (src/AssemblerAlign.cpp:353-355) before its first candidate, and a run with one thread per core of a
    # 256-core box (512 GiB of arenas) took a GPU box down in round 1.  Also kept under a quarter of the available
    # memory (4 GiB per thread: the arena + the thread's share of the rest).
    cores = min(host_cores, 64, max(1, available_memory_gib() // 4))
    data7 = synthetic.pack_markers(toc, kmer)
    parity = {}
    if bindings.ref_available():
        lib, kind = bindings.RefLib(), "reference"
        lh = lib.lowhash0(toc, data7, None, p, threads=cores)
        t_lh = lh.seconds
    else:
        if not bindings.oracle_available():
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        lib, kind = bindings.OracleLib(), "port"
        t0 = time.time()
        lh = lib.lowhash0(toc, data7, None, p)
        t_lh = time.time() - t0
    cand = lh.candidates
    t_suppress = 0.0
    if suppress is not None:
        # Assembler::suppressAlignmentCandidates between the stages (srcMain/main.cpp:697-702): the host layer's own function either side
        # (string meta data, one serial pass: shasta_amd/host/CandidateSuppression.cpp, checked against the reference's parser in tests/).
        t0 = time.time()
        cand = suppress(cand)
        t_suppress = time.time() - t0
    # Assembler::computeSortedMarkers (src/AssemblerAlign.cpp:236-239): once, for all oriented reads, before the alignment threads
    # start -- timed on its own; the per-candidate loop below reads it (method 4; method 3 does not use sorted markers).
    t_sorted = lib.compute_sorted_markers(toc, data7, threads=cores) if (kind == "reference" and align_method == 4) else 0.0
    parity["lowhash0_candidates"] = len(lh.candidates)
    if suppress is not None:
        parity["candidates_after_suppression"] = len(cand)
    parity["lowhash0_equal"] = bool(np.array_equal(lh.candidate_tuples(), gpu_lowhash.candidate_tuples())
                                    and np.array_equal(lh.statistics, gpu_lowhash.statistics)
                                    and np.array_equal(lh.high_frequency, gpu_lowhash.high_frequency)
                                    and np.array_equal(lh.histogram, gpu_lowhash.histogram))
    # The aligner on a sample spread over the whole candidate list.  The reference's per-thread 2 GiB arena is a fixed
    # set-up cost (seconds): two sample sizes, incremental rate.
    # --baseline-sample 0: the WHOLE candidate list through the reference aligner (a minute or two on 64 threads: not the default;
    # profiles/ holds one such run per round) -- then the rate is the run's own, set-up included, and every candidate is compared.
    whole = sample_size <= 0 or sample_size >= len(cand)
    stride = 1 if whole else max(1, len(cand) // max(1, sample_size))
    sample = np.ascontiguousarray(cand[::stride])
    small = np.ascontiguousarray(sample[:max(1, len(sample) // 10)])
    align = lib.align4_batch if align_method == 4 else lib.align3_batch
    t1 = 0.0 if whole else align(toc, data7, small, o, want_ordinals=True, threads=cores).seconds
    ref = align(toc, data7, sample, o, want_ordinals=True, threads=cores)
    t2 = ref.seconds
    per_pair = (t2 - t1) / (len(sample) - len(small)) if (not whole and len(sample) > len(small)) else 0.0
    if per_pair <= 0.0:                                 # the whole list, or too few candidates for the difference to mean anything
        per_pair = t2 / max(1, len(sample))
    if kind == "reference":
        lib.drop_sorted_markers()
    dev = (ctx.align4 if align_method == 4 else ctx.align3)(sample, o, want_ordinals=True, borrow=True)
    # Assembler::computeAlignmentTable (src/AssemblerAlign.cpp:296, 509-571; serial in the reference) on the alignments of the
    # WHOLE candidate list -- the device's rows of the bench's last step (equal to the reference's wherever both exist, below) --
    # and compared with the table the device made from the sample's.
    t_table = 0.0
    if kind == "reference":
        sample_toc, sample_values = ctx.alignment_table()
        ref_toc, ref_values, _ = lib.alignment_table(np.array(dev.alignment_data, copy=True), (len(toc) - 1) // 2)
        parity["alignment_table_equal"] = bool(np.array_equal(sample_toc, ref_toc) and np.array_equal(sample_values, ref_values))
        if all_rows is not None:
            t_table = lib.alignment_table(all_rows, (len(toc) - 1) // 2)[2]
    ties = (ref.status & 0x80) != 0
    parity["aligner_sampled_candidates"] = len(sample)
    parity["aligner_ties_in_sample"] = int(ties.sum())
    parity["aligner_tie_flags_equal"] = bool(np.array_equal(ref.status & 0x80, dev.status & 0x80))
    a, b = ref.per_candidate(~ties), dev.per_candidate(~ties)
    parity["aligner_mismatches"] = int(sum(1 for x, y in zip(a, b) if x != y)) + abs(len(a) - len(b))
    # The DP tie census (oracle/census.py): the same checker under the 11 other tie policies on a third of the sample -- how
    # many of the results depend on the reading of SeqAn that the kernels, the oracle and this baseline share.
    if census_size > 0 and len(sample):
        from oracle import census
        sub = np.ascontiguousarray(sample[::max(1, len(sample) // census_size)])
        t0 = time.time()
        # (Sixteen threads at most: every call of the checker gives each of its threads the reference's 2-GiB arena,
        # src/AssemblerAlign.cpp:353-355, and twelve calls on 64 threads spent 140 of their 150 seconds creating arenas.)
        parity["dp_tie_sensitive"] = census.tie_census(lib, toc, data7, sub, o, align_method=align_method, threads=min(cores, 16))
        parity["dp_tie_sensitive"]["seconds"] = time.time() - t0
    pairs = len(cand)
    total = t_lh + t_suppress + t_sorted + pairs * per_pair + t_table
    return {
        "value": pairs / total if total > 0 else 0.0,
        "unit": "candidate read-pairs aligned/s",
        "cores": cores if _cgroup_cpu_quota() is None else min(cores, max(1, int(_cgroup_cpu_quota()))),
        "threads": cores,
        "host_cores": host_cores,
        "cpu_quota": _cgroup_cpu_quota(),         # (the container's cgroup cpu.max: 16 CPUs on the GPU box of round 3 -- the threads share that much CPU time, whatever their number)
        "kind": kind if kind == "port" else ("reference (LowHash0 and the aligner in full on every candidate; DP = the restated SeqAn call)" if whole else
                                             "reference (LowHash0 in full; aligner: incremental per-candidate rate on a sample x all candidates; DP = the restated SeqAn call)"),
        "aligner_seconds_measured": t2,
        "sample": "the bench's own read set (%d reads, M=%d markers): LowHash0 %.2f s on %d of the host's %d cores -> %d candidates; "
                  "align method %d on every %d-th candidate (%d): %.3f ms/candidate incremental on %d threads; anonymous 4 KiB pages "
                  "(the reference warns such runs should not be used for benchmarking, srcMain/main.cpp:369-378)" % (
                      (len(toc) - 1) // 2, int(toc[-1]), t_lh, cores, host_cores, pairs, align_method, stride, len(sample), per_pair * 1e3, cores),
        "sample_short": "%d reads, LowHash0 in full, aligner on every %d-th candidate (%d) x rate" % ((len(toc) - 1) // 2, stride, len(sample)),
        "lowhash0_seconds": t_lh,
        "suppress_seconds": t_suppress,
        "sorted_markers_seconds": t_sorted,            # Assembler::computeSortedMarkers, all reads, `threads` threads
        "align_seconds_per_pair": per_pair,
        "alignment_table_seconds": t_table,            # Assembler::computeAlignmentTable on all stored alignments (serial, as in the reference)
        "what": "findAlignmentCandidatesLowHash0 + computeAlignments (computeSortedMarkers, the per-candidate loop, computeAlignmentTable)",
    }, parity


GIVE_UP_PREFIX = "dense DP because: "      # rows of the kernel table that count tasks, not launches (align4_dp.hpp, DpGiveUp)


def give_up_rows(table, steps):
    """Why DP tasks ended in the dense kernels although the sparse path is on: tasks and their DP cells per step, by reason."""
    return {name[len(GIVE_UP_PREFIX):]: {"tasks_per_step": r["launches"] / steps, "dp_cells_per_step": r["work"] / steps}
            for name, r in table.items() if name.startswith(GIVE_UP_PREFIX) and r["launches"]}


def kernel_rows(table, steps, pmc):
    """kernels{} of the report from the library's kernel table (accumulated over the timed steps)."""
    rows = {}
    for name, r in table.items():
        if r["launches"] == 0 or name.startswith(GIVE_UP_PREFIX):
            continue
        avg = r["seconds"] / r["launches"]
        per_launch = r["bytes"] / r["launches"]
        row = {"launches_per_step": r["launches"] / steps, "avg_ms": avg * 1e3, "seconds_per_step": r["seconds"] / steps,
               "algorithmic_bytes_per_launch": int(per_launch),
               "achieved_GBps": per_launch / avg / 1e9 if avg > 0 else 0.0}
        row["frac_of_hbm_peak"] = row["achieved_GBps"] / HBM_PEAK_GBS
        if name.startswith("bandedDpForwardKernel") and r["seconds"] > 0:
            row["gcups"] = r["work"] / r["seconds"] / 1e9
        c = pmc_of(pmc, name)
        if c:
            row["hbm_traffic_bytes_per_launch"] = c.get("hbm_bytes_per_launch")
            if c.get("valu_wave_instructions_per_launch") and avg > 0:
                row["valu_issue_frac"] = c["valu_wave_instructions_per_launch"] / avg / VALU_PEAK_WAVE_INSTRUCTIONS_PER_S
        rows[name] = row
    return rows


def roofline_of(name, row, pmc):
    hbm = any(name.startswith(h) or h in name for h in HBM_NATURED)
    r = {"bound": "hbm" if hbm else "valu", "achieved": row["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": row["achieved_GBps"] / HBM_PEAK_GBS, "traffic": row.get("hbm_traffic_bytes_per_launch"), "kernel": name,
         "avg_launch_ms": row["avg_ms"],
         # (where `traffic` and the VALU instruction count come from: the newest committed PMC summary -- a round that could not
         # profile keeps the previous round's counters, collected on that round's build of the kernel)
         "counters_from": os.path.relpath(PMC_FILE, ROOT) if pmc else None}
    if not hbm:
        c = pmc_of(pmc, name) or {}
        r["valu"] = {"peak_wave_instructions_per_s": VALU_PEAK_WAVE_INSTRUCTIONS_PER_S,
                     "wave_instructions_per_launch": c.get("valu_wave_instructions_per_launch"),
                     "frac": row.get("valu_issue_frac"),
                     "source": "SQ_INSTS_VALU of the committed rocprofv3 --pmc pass over this command (profiles/) / the live HIP-event launch time"}
        r["note"] = ("dominant kernel by total time; integer work bound by VALU issue and LDS, not by HBM: its algorithmic bytes "
                     "(4(nx+ny) per candidate or task) are tiny against its work, so the HBM fraction is low by construction")
        if "gcups" in row:
            r["gcups"] = row["gcups"]
    return r


def hbm_budget(markers_total, reads_total, n_gpus, hash_fraction=0.01, iterations=10, pairs_per_read=40.0, workers=6):
    """Per-GPU HBM the path needs for a job of that size on n_gpus GPUs (bytes): every GPU holds the dense
    kmer ids of ALL reads (the aligner needs both reads of any candidate), its share of the LowHash0 buffers, and the aligner
    workers' scratch (measured high-water marks at 2^18 candidates per batch).  Printed for BASELINE configs[3] and [4] so that
    the 288 GB of one MI355X are checked before such a run, not during it."""
    m, r, g = float(markers_total), float(reads_total), float(n_gpus)
    records = 2.0 * hash_fraction * m / g                                # capacity of one iteration's low-hash records on a GPU
    # All iterations in one pass (lowhash0Run and the staged job alike, lowhash0.hip); 64-bit record keys on the ranks of a
    # sharded job (owner | iteration | bucket id) and where iteration | bucket id does not fit 32 bits.
    one_pass = bool(m * hash_fraction > 0 and iterations * records < 2.0 ** 32 - 1)
    wide_keys = one_pass and (g > 1 or (5 + np.ceil(np.log2(max(2.0, hash_fraction * m)))) + np.ceil(np.log2(max(2, iterations))) > 32)
    record_rows = records * (iterations if one_pass else 1)
    pairs = pairs_per_read * r * iterations / g * 1.25                   # pair keys of all iterations owned by a GPU
    parts = {
        "kmer_ids_of_all_reads": 4.0 * m,
        "toc_flags_tile_descriptors": 16.0 * r + r + m / 16.0,
        "lowhash0_records_ping_pong": (32.0 if wide_keys else 24.0) * record_rows,
        "lowhash0_bucket_tables": 20.0 * record_rows,
        "lowhash0_pair_keys_ping_pong": 24.0 * pairs,
        "lowhash0_statistics_histograms": 24.0 * r + 16384.0 * iterations,
        "aligner_scratch_%d_workers" % workers: workers * 14.5e9,      # (round 4: + the candidates' match lists, 2.2 GB, the tasks' ordered hits, 3.4 GB, and their link words, 3.4 GB, per worker: align4_sparse.hpp, align4_anchor.hpp; estimated, not yet measured)
    }
    total = sum(parts.values())
    return {"n_gpus": int(n_gpus), "bytes_per_gpu": {k: int(v) for k, v in parts.items()}, "total_GB_per_gpu": total / 1e9,
            "fits_288_GB": bool(total < 288e9 * 0.95), "records_of_all_iterations_in_one_pass": bool(one_pass)}


def group_bench(lib, devices, toc, kmer, p, o, args, align_method):
    """The same step through the in-process group (shasta_mi355x_group: one host thread and one context per device, device-to-
    device pulls over xGMI for the two exchanges of an iteration) -- the seam a C++ caller of the two Assembler functions uses --
    timed like the step of the one-process-per-GPU driver."""
    with lib.group(devices) as g:
        g.set_kmer_ids(toc, kmer)

        stage = [0.0, 0.0]

        def step():
            t = time.perf_counter()
            lh = g.lowhash0(p)
            stage[0] += time.perf_counter() - t
            t = time.perf_counter()
            al = (g.align4 if align_method == 4 else g.align3)(lh.candidates, o, want_ordinals=False, borrow=True)
            stage[1] += time.perf_counter() - t
            return len(lh.candidates), len(al.alignment_data)

        for _ in range(args.warmup):
            step()
        stage[0] = stage[1] = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pairs, stored = step()
        elapsed = time.perf_counter() - t0
    steps = max(1, args.steps)
    return {"devices": [int(d) for d in devices], "value": pairs / (elapsed / steps), "unit": "pairs/s", "ms_per_step": elapsed / steps * 1e3,
            "lowhash0_ms_per_step": stage[0] / steps * 1e3, "align_ms_per_step": stage[1] / steps * 1e3,
            "candidates": int(pairs), "alignments_stored": int(stored),
            "what": "shasta_mi355x_group_lowhash0_run + _group_align%d_run_borrowed in one process over these devices (result assembly on the host included)" % align_method}


def markers_bench(lib, ctx, args):
    """The widening row `marker finding` (MarkerFinder, src/MarkerFinder.cpp:16-127) on its own: one step = the reads of the
    workload (2-bit planes, 0.25 B per base, uploaded inside the step as the seam does) -> dense kmer ids + toc resident in HBM."""
    from shasta_amd import synthetic
    toc, data, counts = synthetic.packed_base_reads(args.reads, seed=12345)
    is_marker = (np.random.default_rng(231).random(1 << 20) < 0.1).astype(np.uint8)
    bases = int(counts.sum())

    def step():
        return lib.find_markers(toc, data, counts, 10, is_marker, want_packed=False, context=ctx)

    for _ in range(args.warmup):
        step()
    ctx.kernel_table_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mtoc, _ = step()
    elapsed = time.perf_counter() - t0
    steps = max(1, args.steps)
    kernels = kernel_rows(ctx.kernel_table(), steps, {})
    name = max(kernels, key=lambda k: kernels[k]["seconds_per_step"])
    print(json.dumps({
        "metric": "RLE bases scanned/sec (marker finding, SURVEY 8f row 2)", "value": bases / (elapsed / steps), "unit": "bases/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64 bit planes -> u32 kmer ids", "data": "synthetic",
        "config": {"workload": "%d random run-length-encoded reads, mean 20 kb, k = 10, 10%% of the k-mers markers; reads uploaded inside the step"
                               % args.reads, "bases": bases, "markers_both_strands": int(mtoc[-1])},
        "kernels": kernels, "roofline": roofline_of(name, kernels[name], {})}))
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (default 100000; 20000 with --workload ul)")
    ap.add_argument("--workload", choices=sorted(WORKLOAD_SHAPES), default="configs2",
                    help="configs2 = BASELINE configs[2] (the headline); may2022 / ul = the read shape, k = 14 marker alphabet and parameters of configs[3]'s "
                         "conf/Nanopore-May2022.conf / configs[4]'s conf/Nanopore-UL-May2022.conf on one GPU, suppressAlignmentCandidates between the stages")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--baseline-sample", type=int, default=250000, help="candidates the reference aligner runs on: one in eight at 100 k reads (0 = all of them: a minute or two)")
    ap.add_argument("--tie-census", type=int, default=20000,
                    help="candidates (a subset of the baseline sample) the checker re-aligns under the 11 other DP tie policies; 0 = no census")
    ap.add_argument("--group", action="store_true",
                    help="ONE process drives --gpus devices through the in-process group (shasta_mi355x_group) instead of one process per "
                         "GPU with RCCL; with torch.distributed.run and N > 1 the group line is measured by rank 0 after the RCCL line anyway")
    ap.add_argument("--sharded-workload", action="store_true",
                    help="(with --group) the reads as the N ranks of `torch.distributed.run ... --gpus N` generate them: what rank 0's child process runs")
    ap.add_argument("--lowhash-only", action="store_true", help="BASELINE configs[1]")
    ap.add_argument("--markers", action="store_true",
                    help="marker finding only (SURVEY 8f row 2): random RLE reads of --reads x 20 kb, k = 10, 10 %% of the k-mers markers")
    ap.add_argument("--align-method", type=int, default=4, choices=[3, 4],
                    help="4 = Align4 (BASELINE's metric, the default); 3 = the reference's default method, for comparison")
    args = ap.parse_args()
    WORKLOAD_SHAPE[0] = args.workload
    if args.reads is None:
        args.reads = 20000 if args.workload == "ul" else 100000

    import torch
    import shasta_amd
    from shasta_amd import abi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SHASTA_BENCH_FORCE_SHARDED=1: the N-rank code path -- sharded generation, all-gather of the kmer ids, the staged LowHash0
    # with its exchanges, candidate re-split, every collective over RCCL -- with the ONE rank a one-GPU box allows (a rank
    # exchanging with itself): what the driver's per-rank work costs beside the one-GPU path, and that the path runs at all.
    sharded = world > 1 or bool(os.environ.get("SHASTA_BENCH_FORCE_SHARDED"))
    if args.gpus > 1 or sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # SHASTA_BENCH_BACKEND=gloo + SHASTA_BENCH_ONE_DEVICE=1: functional test of the multi-rank path
        # on a box with one GPU (host-staged transport, every rank on cuda:0); never used for numbers.
        if os.environ.get("SHASTA_BENCH_ONE_DEVICE") or DRY_RUN_LIBRARY:
            local_rank = 0
        if not DRY_RUN_LIBRARY:
            torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if DRY_RUN_LIBRARY else os.environ.get("SHASTA_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    else:
        dist = None
        if not DRY_RUN_LIBRARY:
            torch.cuda.set_device(0)

    if DRY_RUN_LIBRARY:
        # Pre-flight of THIS SCRIPT on a machine without a GPU: SHASTA_BENCH_LIBRARY names the emulated
        # build of the library (tests/emu: kernel sources on CPU fibers).  The line it prints is marked
        # as a dry run and is never a measurement.
        from shasta_amd import lib as libmod
        lib = libmod.Library(DRY_RUN_LIBRARY)
    else:
        lib = shasta_amd.load()
    assert lib.device_count() >= 1, "no gfx950 device: the HIP path cannot run (there is no CPU fallback)"
    p, o = lowhash_params(), (align_options() if args.align_method == 4 else align3_options())
    ctx = lib.context(local_rank)
    if args.markers:
        markers_bench(lib, ctx, args)
        return

    def align(candidates):
        if args.align_method == 4:
            return ctx.align4(candidates, o, want_ordinals=False, borrow=True)
        return ctx.align3(candidates, o, want_ordinals=False, borrow=True)

    if args.group and world == 1:
        # ONE process, --gpus devices, the in-process group (weak scaling like the other mode: --reads reads per GPU).
        n = max(1, args.gpus)
        assert lib.device_count() >= n or DRY_RUN_LIBRARY or os.environ.get("SHASTA_BENCH_ONE_DEVICE"), "--group --gpus %d needs that many devices" % n
        toc, kmer = make_workload(args.reads * n, 12345, shards=n if args.sharded_workload else 0)
        ctx.close()
        one_device = bool(os.environ.get("SHASTA_BENCH_ONE_DEVICE") or DRY_RUN_LIBRARY)        # (a box with one GPU: device 0 listed n times)
        g = group_bench(lib, [0] * n if one_device else list(range(n)), toc, kmer, p, o, args, args.align_method)
        # The same job with the group's exchanges over RCCL (multi.hip's second transport: grouped ncclSend / ncclRecv on communicators
        # from ncclCommInitAll) -- only where the devices are distinct (RCCL wants one rank per device); an error is reported, not raised.
        g_rccl = None
        if n > 1 and not one_device and not os.environ.get("SHASTA_MI355X_GROUP_TRANSPORT"):
            os.environ["SHASTA_MI355X_GROUP_TRANSPORT"] = "rccl"
            try:
                g_rccl = group_bench(lib, list(range(n)), toc, kmer, p, o, args, args.align_method)
                g_rccl["what"] += "; exchanges over RCCL (SHASTA_MI355X_GROUP_TRANSPORT=rccl)"
            except Exception as e:          # noqa: BLE001
                g_rccl = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            finally:
                del os.environ["SHASTA_MI355X_GROUP_TRANSPORT"]
        print(json.dumps({
            "metric": "candidate read-pairs aligned/sec (LowHash0+Align%d), in-process group" % (4 if args.align_method == 4 else 3),
            "value": g["value"], "unit": "pairs/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": g["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32/i32 (integer hash + integer DP)",
            "data": "synthetic" if not DRY_RUN_LIBRARY else "synthetic; DRY RUN ON THE EMULATED BUILD - NOT A MEASUREMENT",
            "config": {"workload": "BASELINE configs[2] shape, %d reads/GPU, one job over %d devices in ONE process (shasta_mi355x_group)" % (args.reads, n),
                       "reads_per_gpu": args.reads, "markers_total": int(toc[-1]), "candidates": g["candidates"], "alignments_stored": g["alignments_stored"],
                       "parallelism": "%d GPUs, in-process group: `value` with device-to-device pulls over xGMI%s" % (
                           n, "; in_process_group_over_rccl = the same job with the exchanges as grouped ncclSend / ncclRecv" if g_rccl is not None else "")},
            "in_process_group": g, "in_process_group_over_rccl": g_rccl, "hbm_budget_per_gpu": hbm_budget(int(toc[-1]), args.reads * n, n)}))
        return

    toc = kmer = None
    if not sharded:
        # Workload: BASELINE configs[2].
        toc, kmer = make_workload(args.reads, 12345)
        marker_count = int(toc[-1])
        t0 = time.perf_counter()
        ctx.set_kmer_ids(toc, kmer)                     # host -> HBM, outside the timed region
        upload_seconds = time.perf_counter() - t0

        table_seconds = [0.0]
        suppress_seconds = [0.0]
        suppressed = [0]
        suppress = None
        if WORKLOAD_SUPPRESS_DELTA[args.workload] and not args.lowhash_only:
            # Assembler::suppressAlignmentCandidates between the two stages (srcMain/main.cpp:697-702), on meta data made once, outside the
            # timed region, from the candidates of an untimed LowHash0 call.
            import shasta_amd.assembler as host_layer
            meta_data = make_meta_data(args.reads, ctx.lowhash0(p).candidates)
            host_so = os.environ.get("SHASTA_BENCH_HOST_LIBRARY")           # (the emulated twin, for the dry run)
            # (the reads' meta data parsed once, as a caller that holds it for a whole run would: keys per read, then integer comparisons per candidate)
            suppress = host_layer.CandidateSuppression(meta_data, WORKLOAD_SUPPRESS_DELTA[args.workload], hostLibrary=host_so).apply

        def step():
            lh = ctx.lowhash0(p)
            if args.lowhash_only:
                return lh, None, len(lh.candidates)
            candidates = lh.candidates
            if suppress is not None:
                t = time.perf_counter()
                candidates = suppress(candidates)
                suppress_seconds[0] += time.perf_counter() - t
                suppressed[0] = len(lh.candidates) - len(candidates)
            al = align(candidates)
            # The last step of Assembler::computeAlignments (src/AssemblerAlign.cpp:296): the alignment table of what was stored.
            t = time.perf_counter()
            al.table = ctx.alignment_table(copy=False)
            table_seconds[0] += time.perf_counter() - t
            return lh, al, len(candidates)
    else:
        # ONE job over all GPUs (weak scaling: `reads` reads per GPU of one read set at the same
        # coverage).  Every rank generates its own read range, the dense kmer ids are all-gathered
        # over xGMI so that every GPU holds every read (Align4 needs both reads of a candidate),
        # LowHash0 runs sharded with its two all-to-all exchanges, the candidate list is re-split
        # evenly for Align4.  Setup (generation, all-gather) is outside the timed region.
        from shasta_amd import distributed, synthetic
        device = torch.device("cuda", local_rank) if not DRY_RUN_LIBRARY else torch.device("cpu")
        shape = WORKLOAD_SHAPES[WORKLOAD_SHAPE[0]]
        genome_markers = max(20000, int(round(world * args.reads * shape["mean_markers"] / 45.0)))
        toc_s, kmer_s = synthetic.marker_reads(args.reads, genome_markers, keep_probability=0.8, spurious_probability=0.25, k=10, seed=12345,
                                               shard=rank, shard_count=world, alphabet=workload_alphabet(), **shape)
        sizes = [None] * world
        dist.all_gather_object(sizes, np.diff(toc_s.astype(np.int64)).astype(np.uint32).tobytes())
        per_shard = [np.frombuffer(b, dtype=np.uint32).astype(np.uint64) for b in sizes]
        toc = np.zeros(2 * world * args.reads + 1, dtype=np.uint64)
        toc[1:] = np.cumsum(np.concatenate(per_shard))
        shard_markers = [int(x.sum()) for x in per_shard]
        mine = torch.from_numpy(kmer_s.view(np.int32)).to(device)
        everything = distributed.all_gather_padded(mine, shard_markers)    # C3: every GPU gets every read's kmer ids
        del mine, kmer_s
        if not DRY_RUN_LIBRARY:
            torch.cuda.synchronize()
        ctx.set_kmer_ids_device(toc, everything.data_ptr())
        all_kmer_ids = everything          # (the context reads them in place: kept alive)
        del everything
        marker_count = int(toc[-1])
        read_count = world * args.reads
        backend = distributed.HipBackend(ctx, device)
        boundaries = np.arange(0, read_count + 1, args.reads, dtype=np.uint64)     # the generated shards
        upload_seconds = None
        # SHASTA_BENCH_SHARDED_PHASES=1: where the staged LowHash0's wall clock goes -- every stage, exchange and reduction of
        # the driver bracketed by device synchronisations (which cost a little themselves: a diagnosis, not the headline run).
        phase_seconds, phase_log = {}, []
        if os.environ.get("SHASTA_BENCH_SHARDED_PHASES"):
            def _timed(name, f):
                def g(*a, **k):
                    if not DRY_RUN_LIBRARY:
                        torch.cuda.synchronize()
                    t = time.perf_counter()
                    r = f(*a, **k)
                    if not DRY_RUN_LIBRARY:
                        torch.cuda.synchronize()
                    phase_seconds[name] = phase_seconds.get(name, 0.0) + time.perf_counter() - t
                    phase_log.append((name, round(1e3 * (time.perf_counter() - t), 2)))
                    return r
                return g
            for name in ("begin", "hash_all", "buckets_all", "merge_all", "hash", "buckets", "merge", "finish", "finish_on_device"):
                setattr(backend, name, _timed("stage " + name, getattr(backend, name)))
            for name in ("exchange", "all_reduce_sum_u64", "all_gather_padded", "_slice_by_markers_on", "candidate_share"):
                setattr(distributed, name, _timed(name, getattr(distributed, name)))
            from shasta_amd import abi as _abi                     # (inside `stage finish` and the aligner's result: the host copies)
            _abi.copy_array = _timed("abi.copy_array (inside other rows)", _abi.copy_array)

        def step():
            t_lh = time.perf_counter()
            lh = distributed.lowhash0(backend, p, read_count, boundaries, candidates_on_device=True)     # (the all-gather reads them where they are)
            share, total = distributed.candidate_share(lh.candidates, device, toc=toc)
            lh.seconds = time.perf_counter() - t_lh          # (this rank's wall clock: the staged job, both exchanges, the candidate re-split)
            if args.lowhash_only:
                return lh, None, total
            al = align(share)
            return lh, al, total

    def sync():
        if dist is not None:
            dist.barrier()
        if not DRY_RUN_LIBRARY:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if sharded:
        phase_seconds.clear()
        del phase_log[:]
    ctx.kernel_table_reset()
    if not sharded:
        table_seconds[0] = 0.0
        suppress_seconds[0] = 0.0
    t0 = time.perf_counter()
    cpu0 = _process_cpu_seconds()
    throttled0 = _cgroup_throttled_usec()
    lh_dev = al_dev = lh_wall = al_wall = 0.0
    each_step = []                                  # (LowHash0 device ms, aligner device ms) of every timed step: outliers show here
    for _ in range(args.steps):
        if os.environ.get("SHASTA_MI355X_LOG_ALLOC") == "1":      # (beside the library's allocation log, on its clock)
            sys.stderr.write("bench: step %d begins at %.1f ms\n" % (len(each_step), 1e3 * time.monotonic()))
        lh, al, pairs_total = step()
        if not sharded:
            lh_dev += lh.device_seconds
            lh_wall += lh.seconds
        else:
            lh_wall += lh.seconds
        if al is not None:
            al_dev += al.device_seconds
            al_wall += al.seconds
        each_step.append([round(1e3 * lh.device_seconds, 2) if not sharded else round(1e3 * lh.seconds, 2), round(1e3 * al.device_seconds, 2) if al is not None else None])
    sync()
    elapsed = time.perf_counter() - t0
    # What the timed region cost the host: the process's CPU seconds per second of wall clock (its threads that spin or work),
    # beside the CPU quota of the container it runs in and the time the kernel throttled it meanwhile.
    host_load = {"cpus_busy": (_process_cpu_seconds() - cpu0) / elapsed if elapsed > 0 else None, "cpu_quota": _cgroup_cpu_quota(),
                 "throttled_ms_per_step": (None if throttled0 is None or _cgroup_throttled_usec() is None
                                           else (_cgroup_throttled_usec() - throttled0) / 1e3 / max(1, args.steps))}
    table = ctx.kernel_table()
    # What the device holds after the timed steps (the library's buffers only grow: kmer ids, LowHash0's job, six workers' scratch).
    hbm_measured = None
    if not DRY_RUN_LIBRARY:
        try:
            free_bytes, total_bytes = torch.cuda.mem_get_info(local_rank)
            hbm_measured = {"used_GB": (total_bytes - free_bytes) / 1e9, "total_GB": total_bytes / 1e9,
                            "what": "hipMemGetInfo after the timed steps: everything this process holds on the device (kmer ids, LowHash0 buffers, aligner workers' scratch, torch's context)"}
        except Exception:          # noqa: BLE001
            hbm_measured = None
    # Outside the timed region: one more pass with ONE aligner worker.  With six workers the kernels of different batches share
    # the device, and the HIP-event duration of a launch includes the time it spent sharing; this pass gives every kernel's
    # duration alone on the device (what a profiler's per-kernel view and the PMC passes see).
    table_one_worker = None
    if not sharded and al is not None and not DRY_RUN_LIBRARY:
        previous = os.environ.get("SHASTA_MI355X_ALIGN_WORKERS")
        os.environ["SHASTA_MI355X_ALIGN_WORKERS"] = "1"
        ctx.kernel_table_reset()
        lh, al, pairs_total = step()
        table_one_worker = ctx.kernel_table()
        if previous is None:
            del os.environ["SHASTA_MI355X_ALIGN_WORKERS"]
        else:
            os.environ["SHASTA_MI355X_ALIGN_WORKERS"] = previous
    stored_total = 0 if al is None else len(al.alignment_data)
    status_counts = None
    if al is not None:
        st = np.asarray(al.status)
        status_counts = [int(((st & 0x7f) == abi.SHASTA_ALIGN_STORED).sum()), int(((st & 0x7f) == abi.SHASTA_ALIGN_REJECTED).sum()),
                         int(((st & 0x7f) == abi.SHASTA_ALIGN_EMPTY).sum()), int(((st & 0x7f) == abi.SHASTA_ALIGN_SKIPPED).sum()),
                         int(((st & 0x80) != 0).sum())]
    if dist is not None:
        comm = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([stored_total] + (status_counts or [0] * 5), dtype=torch.int64, device=comm)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        stored_total = int(c[0].item())
        if status_counts is not None:
            status_counts = [int(x) for x in c[1:].tolist()]

    # N > 1: a job of the same shape once more through the in-process group -- ONE process over all N devices, the seam a C++
    # caller of the two Assembler functions uses: no RCCL, device-to-device pulls -- so that one multi-GPU run compares the two
    # drivers.  It runs in a CHILD process of rank 0 (`bench.py --group --sharded-workload --gpus N`, which generates the same
    # reads again): peer access between distinct devices has never executed anywhere, and whatever it does there -- an error, a
    # crash, a hang that runs into the timeout -- must not cost this run its result.  The other ranks wait on the HOST (a gloo
    # group: an RCCL barrier would spin on their GPUs meanwhile).
    group_line = None
    if dist is not None and not args.lowhash_only and not os.environ.get("SHASTA_BENCH_NO_GROUP_LINE"):
        waiting = dist.new_group(backend="gloo")
        if rank == 0:
            try:
                import subprocess
                child_env = {k: v for k, v in os.environ.items()
                             if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                          "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                                          "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")}
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--group", "--sharded-workload", "--gpus", str(world), "--reads", str(args.reads),
                                        "--steps", str(max(1, min(args.steps, 5))), "--warmup", str(min(args.warmup, 2)),
                                        "--align-method", str(args.align_method)],
                                       env=child_env, capture_output=True, text=True, timeout=420)
                lines = [ln for ln in child.stdout.strip().splitlines() if ln.startswith("{")]
                if child.returncode == 0 and lines:
                    group_line = json.loads(lines[-1])["in_process_group"]
                    group_line["over_rccl"] = json.loads(lines[-1]).get("in_process_group_over_rccl")
                else:
                    group_line = {"error": "bench.py --group --gpus %d ended with code %d: %s" % (world, child.returncode, child.stderr[-400:])}
            except Exception as e:          # noqa: BLE001 -- reported, not raised (a timeout included)
                group_line = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
        dist.barrier(group=waiting)

    final_line = None
    retry_with = None
    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed / steps * 1e3
        value = pairs_total / (elapsed / steps)
        pmc = load_pmc(args.reads) if (not sharded and args.workload == "configs2") else {}
        kernels = kernel_rows(table, steps, pmc)
        kernel_seconds = sum(r["seconds_per_step"] for r in kernels.values())
        for r in kernels.values():
            r["share_of_kernel_time"] = r["seconds_per_step"] / kernel_seconds if kernel_seconds > 0 else 0.0
        dominant = max(kernels, key=lambda k: kernels[k]["seconds_per_step"]) if kernels else None
        roofline = roofline_of(dominant, kernels[dominant], pmc) if dominant else None
        kernels_one_worker = None
        if table_one_worker is not None:
            kernels_one_worker = kernel_rows(table_one_worker, 1, pmc)
            if dominant in kernels_one_worker:
                alone = roofline_of(dominant, kernels_one_worker[dominant], pmc)
                roofline["one_worker"] = {k: alone[k] for k in ("achieved", "frac", "avg_launch_ms")}
                if "valu" in alone:
                    roofline["one_worker"]["valu_frac"] = alone["valu"]["frac"]
                roofline["one_worker"]["note"] = ("the same kernel in a pass with one aligner worker (outside the timed region): its launches alone on the device; "
                                                  "in the timed region six workers' kernels overlap and a launch's HIP-event duration includes the time it shares")
        # The HBM-natured kernel of the path (K1), always reported beside the dominant one.
        hash_name = next((k for k in kernels if k.startswith("hashWindowsKernel")), None)
        out = {
            "metric": ("candidate read-pairs aligned/sec (LowHash0+Align4)" if args.align_method == 4
                       else "candidate read-pairs aligned/sec (LowHash0+align method 3)") if not args.lowhash_only
                      else "candidate read-pairs found/sec (LowHash0 only)",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32/i32 (integer hash + integer DP)",
            "data": "synthetic" if not DRY_RUN_LIBRARY else "synthetic; DRY RUN ON THE EMULATED BUILD - NOT A MEASUREMENT",
            "config": {
                "workload": ("BASELINE configs[2]: synthetic ONT-like reads, marker level, %d reads/GPU, "
                             "mean 1500 markers (~20 kb) per oriented read, 45x coverage; LowHash0 m=4 f=0.01 "
                             "10 iterations 5/30/5; %s" % (args.reads, "Align4 200/10/10/100, maxBand 1000, 6/-1/-1"
                                                           if args.align_method == 4 else
                                                           "align method 3: downsamplingFactor 0.05, bandExtend 10, maxBand 1000, 6/-1/-1"))
                            if args.workload == "configs2" else
                            ("the read shape and parameters of BASELINE configs[4] (conf/Nanopore-UL-May2022.conf) on one GPU: synthetic reads, marker level, k = 14 marker alphabet (609 k ids), "
                             "%d reads/GPU, none below 3750 markers (50 kb), mean 7500 (~100 kb), 45x coverage; LowHash0 m=4 f=0.01 10 iterations 10/50/5; "
                             "suppressAlignmentCandidates (delta 30) between the stages; Align4 100/100/100, minAlignedMarkerCount 10, minAlignedFraction 0.1, maxBand 1000" % args.reads)
                            if args.workload == "ul" else
                            ("the read shape and parameters of BASELINE configs[3] (conf/Nanopore-May2022.conf) on one GPU: synthetic reads, marker level, k = 14 marker alphabet (609 k ids), "
                             "%d reads/GPU, none below 750 markers (10 kb), mean 1875 (~25 kb), 45x coverage; LowHash0 m=4 f=0.01 10 iterations 5/30/5; "
                             "suppressAlignmentCandidates (delta 30) between the stages; Align4 100/100/100, minAlignedMarkerCount 10, minAlignedFraction 0.1, maxBand 1000" % args.reads),
                "step": ("findAlignmentCandidatesLowHash0 (src/AssemblerLowHash.cpp:10-55) + computeAlignments (src/AssemblerAlign.cpp:208-304) end to end: "
                         "candidates -> AlignmentData + CompressedAlignments on the host%s"
                         % (" + the alignment table (computeAlignmentTable, :296)" if not sharded else
                            "; each rank keeps the alignments of its candidate share (no alignment table: its indices are those of the whole list)"))
                        if not args.lowhash_only else "findAlignmentCandidatesLowHash0 only",
                "reads_per_gpu": args.reads, "markers_total": marker_count,
                "candidates": pairs_total, "alignments_stored": stored_total,
                "candidates_suppressed_between_the_stages": (suppressed[0] if (not sharded and suppress is not None) else None),
                "parallelism": "1 GPU" if not sharded else
                               "%d GPU%s, one job: reads sharded by id range, RCCL all-to-all of the low-hash records and of the pair "
                               "keys of all MinHash iterations (two exchanges per job), candidates re-split by sum(nx+ny) for Align4"
                               % (world, "" if world == 1 else "s"),
            },
            "stage_device_ms_each_step": each_step,
            "host_load_in_the_timed_region": host_load,
            "stage_seconds_per_step": {"lowhash0_device": lh_dev / steps, "align4_device": al_dev / steps,
                                       "lowhash0_call": lh_wall / steps, "align4_call": al_wall / steps,
                                       "alignment_table_call": (table_seconds[0] / steps) if not sharded else None,
                                       "suppress_candidates_call": (suppress_seconds[0] / steps) if (not sharded and suppress is not None) else None},
            "kernel_seconds_per_step": kernel_seconds,
            "kernels": kernels,
            "roofline": roofline,
        }
        if group_line is not None:
            out["in_process_group"] = group_line
        if sharded and phase_seconds:
            out["sharded_lowhash0_phase_ms_per_step"] = {k: round(1e3 * v / steps, 3) for k, v in sorted(phase_seconds.items(), key=lambda kv: -kv[1])}
            out["sharded_lowhash0_phase_ms_each_step"] = [[name, ms] for name, ms in phase_log if ms >= 10.0]      # (outliers show here)
        # What a GPU must hold: this run, and BASELINE configs[3] / [4] on 8 GPUs (chr1 50x M = 1.7e9, human 50x M = 2.2e10).
        out["hbm_budget_per_gpu"] = {
            "this_run_measured": hbm_measured,
            "this_run": hbm_budget(marker_count, args.reads * world, world),
            "configs[3] chr1 50x, 8 GPUs": hbm_budget(1.7e9, 6.2e5, 8, iterations=10),
            "configs[4] human 50x, 8 GPUs": hbm_budget(2.2e10, 7.7e6, 8, iterations=10),
        }
        if kernels_one_worker is not None:
            out["kernels_one_worker"] = {k: {f: v[f] for f in ("avg_ms", "seconds_per_step", "achieved_GBps", "valu_issue_frac", "gcups") if f in v}
                                         for k, v in kernels_one_worker.items()}
        if status_counts is not None:
            out["aligner_status"] = dict(zip(("stored", "rejected_by_filters", "empty", "skipped", "component_ties_flagged"), status_counts))
        if al is not None and not sharded:
            # The banded DP the reference runs for these candidates (nx x band width cells over all tasks) against what reached the dense
            # kernels here: the rest came from the matches inside the band (align4_sparse.hpp; SHASTA_MI355X_SPARSE_DP=0: all of it dense).
            dense_cells = sum(r["work"] for k, r in table.items() if k.startswith("bandedDpForwardKernel")) / steps
            chain = table.get("sparseChainWaveKernel") or table.get("sparseChainKernel")      # (the row the matches inside the bands are booked on)
            out["banded_dp"] = {"reference_cells_per_step": int(al.dp_cell_count), "cells_in_the_dense_kernels_per_step": int(dense_cells),
                                "share_from_the_matches": (1.0 - dense_cells / al.dp_cell_count) if al.dp_cell_count else None,
                                "sparse_path": chain is not None,
                                # (align4_anchor.hpp: matches of the tasks with several optimal chains that the anchor kernel walked,
                                # against all matches inside the tasks' bands)
                                "matches_in_the_bands_per_step": int(chain["work"] / steps) if chain else None,
                                "matches_walked_by_the_anchor_kernel_per_step": int(table["sparseAnchorKernel"]["work"] / steps) if "sparseAnchorKernel" in table else None,
                                "reference_cells_per_second": al.dp_cell_count / (elapsed / steps) if elapsed > 0 else None}
        if give_up_rows(table, steps):
            out["give_ups"] = give_up_rows(table, steps)
        if al is not None and not sharded:
            out["candidate_paths"] = candidate_paths(table, steps, pairs_total)
        if hash_name:
            h = kernels[hash_name]
            out["hbm_natured_kernel"] = {"kernel": hash_name, "achieved_GBps": h["achieved_GBps"], "frac_of_hbm_peak": h["frac_of_hbm_peak"],
                                         "avg_ms": h["avg_ms"], "traffic": h.get("hbm_traffic_bytes_per_launch"),
                                         "valu_issue_frac": h.get("valu_issue_frac")}
        if upload_seconds is not None:
            # What the one-shot seam pays in addition when it is handed host buffers: the upload of the markers
            # (here 4 B dense kmer ids per marker; 7 B packed through set_markers).  Never part of `value`.
            out["pcie_inclusive"] = {"upload_seconds": upload_seconds, "upload_bytes": 4 * marker_count,
                                     "value_with_upload_every_step": pairs_total / (elapsed / steps + upload_seconds)}
        if not args.no_cpu_baseline and not sharded:
            lh_check = ctx.lowhash0(p)
            all_rows = np.array(al.alignment_data, copy=True) if al is not None else None      # (the last step's, before the context's arrays are reused)
            out["cpu_baseline"], out["parity_at_bench_size"] = cpu_baseline(
                ctx, toc, kmer, p, o, args.align_method, lh_check, args.baseline_sample if not DRY_RUN_LIBRARY else 200,
                census_size=args.tie_census if not DRY_RUN_LIBRARY else 60, all_rows=all_rows, suppress=suppress)
            out["dp_tie_sensitive"] = out["parity_at_bench_size"].pop("dp_tie_sensitive", None)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None
        # The safety net (one process, one GPU, the full line with its parity check): see _fallback_environment.
        earlier = os.environ.get("SHASTA_BENCH_EARLIER_ATTEMPTS")
        if earlier:
            out["earlier_attempts"] = json.loads(earlier)
            out["default_path_parity_failed"] = True        # (the headline below is the fallback's; the default path's numbers are in earlier_attempts)
            out["path"] = "NOT the default path: " + ", ".join("%s=%s" % (k, os.environ[k]) for k in FALLBACK_SWITCHES if k in os.environ)
        retry_with = _fallback_environment(out.get("parity_at_bench_size"), None) if (world == 1 and not args.group) else None
        if retry_with:
            attempts = json.loads(earlier) if earlier else []
            attempts.append({"switches": {k: os.environ[k] for k in FALLBACK_SWITCHES if k in os.environ}, "ms_per_step": out["ms_per_step"], "value": out["value"],
                             "parity_at_bench_size": out.get("parity_at_bench_size")})
            retry_with["SHASTA_BENCH_EARLIER_ATTEMPTS"] = json.dumps(attempts)
        details_path = write_details(out)
        final_line = json.dumps(headline(out, details_path), allow_nan=False)
    ctx.close()
    if final_line is not None and retry_with:
        sys.stderr.write("bench.py: the parity check at bench size FAILED on this path; running again with %s\n" % {k: v for k, v in retry_with.items() if k in FALLBACK_SWITCHES})
        _flush_all_stdio()
        os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, **retry_with))
    if dist is not None:
        # The JSON line has to be the LAST thing on stdout: RCCL's version banner (it prints one when NCCL_DEBUG asks for it, as
        # on the GPU box) sits in the C library's stdout buffer of a process until that is flushed -- at exit, i.e. AFTER a line
        # printed from Python.  So: every rank empties both buffers, all ranks meet, the process group goes, and only then
        # rank 0 prints.
        _flush_all_stdio()
        dist.barrier()
        dist.destroy_process_group()
    _flush_all_stdio()
    if final_line is not None:
        print(final_line, flush=True)


# The final stdout line is the driver's contract: it stays under FINAL_LINE_LIMIT bytes (round 4's 20-odd KB line was cut by the
# driver's tail capture and could not be parsed).  Everything else -- the per-kernel tables, the tie census, the HBM budget -- goes
# to the details file (and to stderr as one line), which scripts/ copy under profiles/.
FINAL_LINE_LIMIT = 4096
DETAILS_FILE = os.environ.get("SHASTA_BENCH_DETAILS") or os.path.join(ROOT, "gpurun_out", "bench_details.json")


def _finite(x):
    """JSON has no NaN/Infinity: they become null (the line must pass a strict parser)."""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return _finite(float(x))
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def write_details(out):
    """The whole report -> DETAILS_FILE + one stderr line; returns the path relative to the repo (or None when it cannot be written)."""
    text = json.dumps(_finite(out), allow_nan=False)
    sys.stderr.write("bench details: " + text + "\n")
    try:
        os.makedirs(os.path.dirname(DETAILS_FILE), exist_ok=True)
        with open(DETAILS_FILE, "w") as f:
            f.write(text + "\n")
        return os.path.relpath(DETAILS_FILE, ROOT)
    except OSError:
        return None


def _rounded(x, digits=6):
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _rounded(v, digits) for k, v in x.items()}
    if isinstance(x, list):
        return [_rounded(v, digits) for v in x]
    return x


def headline(out, details_path):
    """The short final line: the contract keys, config, roofline of ONE kernel, cpu_baseline as numbers, parity at bench size."""
    out = _finite(out)
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in out}
    c = out.get("config", {})
    line["config"] = {k: c[k] for k in ("workload", "step", "reads_per_gpu", "markers_total", "candidates", "alignments_stored", "candidates_suppressed_between_the_stages", "parallelism")
                      if k in c and c[k] is not None}
    r = out.get("roofline")
    if r:
        short = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "counters_from")}
        if r.get("valu"):
            short["valu_issue_frac"] = r["valu"].get("frac")
        if "gcups" in r:
            short["gcups"] = r["gcups"]
        if r.get("one_worker"):
            short["one_worker"] = {k: v for k, v in r["one_worker"].items() if k != "note"}
        line["roofline"] = short
    b = out.get("cpu_baseline")
    if b:
        line["cpu_baseline"] = {k: b[k] for k in ("value", "unit", "cores", "kind", "sample", "threads", "host_cores", "cpu_quota", "lowhash0_seconds",
                                                  "sorted_markers_seconds", "align_seconds_per_pair", "alignment_table_seconds", "aligner_seconds_measured") if k in b}
        line["cpu_baseline"]["kind"] = str(b.get("kind", "")).split(" ")[0]
        line["cpu_baseline"]["sample"] = str(b.get("sample_short") or b.get("sample", ""))[:160]
    p = out.get("candidate_paths")
    if p and "BASELINE configs[2]" not in str(c.get("workload", "")):
        line["candidate_paths"] = {k: p.get(k) for k in ("share_in_the_LDS_chunk_kernels", "share_in_the_windowed_LDS_kernel", "share_in_the_HBM_scratch_kernel")}
    u = out.get("pcie_inclusive")
    if u:
        # What a one-shot call of the seams pays in addition (the markers cross PCIe on every call): never `value`.
        line["pcie_inclusive"] = {k: u.get(k) for k in ("value_with_upload_every_step", "upload_seconds")}
    for k in ("parity_at_bench_size", "speedup_vs_cpu_baseline", "stage_seconds_per_step", "aligner_status", "path", "default_path_parity_failed"):
        if out.get(k) is not None:
            line[k] = out[k]
    h = out.get("hbm_natured_kernel")
    if h:
        line["hbm_natured_kernel"] = {k: h.get(k) for k in ("kernel", "achieved_GBps", "frac_of_hbm_peak", "avg_ms", "traffic")}
    d = out.get("banded_dp")
    if d:
        line["banded_dp"] = {k: d.get(k) for k in ("reference_cells_per_step", "cells_in_the_dense_kernels_per_step", "sparse_path")}
    t = out.get("dp_tie_sensitive")
    if t:
        line["dp_tie_sensitive"] = {k: t.get(k) for k in ("candidates", "candidates_changed", "markerCount_changed", "stored_set_changed")}
    g = out.get("in_process_group")
    if g:
        line["in_process_group"] = {k: g[k] for k in ("value", "ms_per_step", "error") if k in g}
        if isinstance(g.get("over_rccl"), dict):
            line["in_process_group"]["over_rccl"] = {k: g["over_rccl"][k] for k in ("value", "ms_per_step", "error") if k in g["over_rccl"]}
    if out.get("earlier_attempts"):
        line["earlier_attempts"] = [{k: a.get(k) for k in ("switches", "ms_per_step", "raised") if k in a} for a in out["earlier_attempts"]]
    line["details"] = details_path
    line = _rounded(line)
    # Whatever a future key adds, the line stays short: optional parts go first, the contract keys never.
    for optional in ("dp_tie_sensitive", "banded_dp", "hbm_natured_kernel", "aligner_status", "stage_seconds_per_step", "earlier_attempts", "in_process_group",
                     "candidate_paths", "pcie_inclusive"):
        if len(json.dumps(line, allow_nan=False)) < FINAL_LINE_LIMIT:
            break
        line.pop(optional, None)
    if len(json.dumps(line, allow_nan=False)) >= FINAL_LINE_LIMIT:
        # The last resort (a long string in a key that is not optional): the contract keys and the details file, long strings cut -- a
        # line is ALWAYS printed.
        def cut(x):
            if isinstance(x, str):
                return x[:200]
            if isinstance(x, dict):
                return {k: cut(v) for k, v in x.items()}
            if isinstance(x, list):
                return [cut(v) for v in x[:8]]
            return x
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "parity_at_bench_size", "details")
        line = cut({k: line[k] for k in keep if k in line})
        line["line_shortened"] = True
        if len(json.dumps(line, allow_nan=False)) >= FINAL_LINE_LIMIT:
            line = {k: line[k] for k in keep[:12] + ("details", "line_shortened") if k in line}
    return line


def _process_cpu_seconds():
    t = os.times()
    return t.user + t.system


def _cgroup_cpu_quota():
    """CPUs the container may use (cgroup v2 cpu.max), or None."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else float(quota) / float(period)
    except Exception:          # noqa: BLE001
        return None


def _cgroup_throttled_usec():
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return float(line.split()[1])
    except Exception:          # noqa: BLE001
        pass
    return None


def _flush_all_stdio():
    sys.stdout.flush(); sys.stderr.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:          # noqa: BLE001
        pass


# The library's own switches back to earlier forms of its kernels: what the line falls back on, one step at a time,
# when the parity check at bench size fails or the library raises -- round 4's kernels had not run on a GPU when they were
# committed.  The line then says so (`path`, `earlier_attempts`): it is the same work on the same reads through kernels that
# produce the same results, never a skipped stage, and never silent.
FALLBACK_SWITCHES = ("SHASTA_MI355X_ANCHORED_DP", "SHASTA_MI355X_SPARSE_DP", "SHASTA_MI355X_STATISTICS_ATOMICS")


def _fallback_environment(parity, error):
    """-> the switches to run again with, or None: nothing failed, or nothing is left to switch."""
    forced = os.environ.get("SHASTA_BENCH_FORCE_PARITY_FAILURE") and not os.environ.get("SHASTA_BENCH_EARLIER_ATTEMPTS")     # (the test of this net)
    lowhash_bad = parity is not None and parity.get("lowhash0_equal") is False
    aligner_bad = forced or (parity is not None and (parity.get("aligner_mismatches", 0) != 0 or parity.get("alignment_table_equal") is False
                                                     or parity.get("aligner_tie_flags_equal") is False))
    if error is not None:
        lowhash_bad = aligner_bad = True
    switches = {}
    if lowhash_bad and os.environ.get("SHASTA_MI355X_STATISTICS_ATOMICS") != "1":
        switches["SHASTA_MI355X_STATISTICS_ATOMICS"] = "1"
    if aligner_bad:
        if os.environ.get("SHASTA_MI355X_ANCHORED_DP") != "0" and error is None:
            switches["SHASTA_MI355X_ANCHORED_DP"] = "0"
        elif os.environ.get("SHASTA_MI355X_SPARSE_DP") != "0":
            switches["SHASTA_MI355X_SPARSE_DP"] = "0"
    return switches or None


def _main_with_safety_net():
    try:
        main()
    except (RuntimeError, AssertionError) as e:
        single = int(os.environ.get("WORLD_SIZE", "1")) == 1 and "--group" not in sys.argv
        retry_with = _fallback_environment(None, e) if single else None
        if not retry_with:
            raise
        attempts = json.loads(os.environ.get("SHASTA_BENCH_EARLIER_ATTEMPTS", "[]"))
        attempts.append({"switches": {k: os.environ[k] for k in FALLBACK_SWITCHES if k in os.environ}, "raised": str(e)[:400]})
        retry_with["SHASTA_BENCH_EARLIER_ATTEMPTS"] = json.dumps(attempts)
        sys.stderr.write("bench.py: the library raised (%s); running again with %s\n" % (str(e)[:200], {k: v for k, v in retry_with.items() if k in FALLBACK_SWITCHES}))
        _flush_all_stdio()
        os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, **retry_with))


if __name__ == "__main__":
    _main_with_safety_net()
