#!/usr/bin/env python3
"""Benchmark of the overlap-detection hot path (LowHash0 + Align4) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over the workload: LowHash0 (10 MinHash
iterations) on the resident markers -> candidate list -> Align4 on every candidate ->
AlignmentData + CompressedAlignments on the host.  Inputs (kmer ids, toc) are resident
in HBM before the timed region.  Workload: BASELINE.json configs[2]
("Synthetic 100k reads, 1xMI355X, LowHash0 + Align4 banded marker alignment end-to-end"),
generated at marker level (shasta_amd/synthetic.py; SURVEY F5: the path never reads bases).

Prints ONE JSON line on rank 0 (see the keys below).  `value` = candidate read pairs
aligned per second, whole job.  The CPU baseline is the reference's own code
(oracle/_ref, compiled in place from /root/reference) when that library is present,
otherwise the CPU restatement (oracle/), timed on this host on a 1/10-scale sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def make_workload(n_reads, seed):
    from shasta_amd import synthetic
    # 45x coverage: n_reads * 1500 genome markers per read / genome markers.
    genome_markers = max(20000, int(round(n_reads * 1500 / 45.0)))
    return synthetic.marker_reads(n_reads, genome_markers, mean_markers=1500.0, sigma=0.5, min_markers=790,
                                  keep_probability=0.8, spurious_probability=0.25, k=10, seed=seed)


def lowhash_params():
    from shasta_amd import abi
    # m=4, f=0.01, 10 iterations, minBucketSize/maxBucketSize/minFrequency 5/30/5.
    return abi.default_lowhash0_params(minBucketSize=5, maxBucketSize=30, minFrequency=5)


def align3_options():
    # alignMethod 3 with the values of the shipped Nanopore configurations (downsamplingFactor 0.05) on
    # top of the defaults of src/AssemblerOptions.cpp:419-449; k = 10 as the synthetic marker alphabet.
    from shasta_amd import abi
    return abi.default_align3_options(downsamplingFactor=0.05)


def align_options():
    from shasta_amd import abi
    return abi.default_align4_options()


def load_traffic():
    """HBM bytes per launch per kernel from the calibrated PMC passes (scripts/gpu_pmc.sh +
    scripts/pmc_traffic.py, committed as profiles/r01_traffic_100k_reads.json)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_100k_reads.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def traffic_of(table, kernel_name):
    """Traffic only applies to the workload it was measured on (100 k reads, 1 GPU); otherwise null."""
    if table is None or TRAFFIC_WORKLOAD["reads"] != table.get("workload_reads"):
        return None
    for k, v in table.get("kernels", {}).items():
        if kernel_name.replace(" ", "") in k.replace(" ", "") or k.replace(" ", "").endswith(kernel_name.replace(" ", "")):
            return v["hbm_bytes_per_launch"]
    return None


TRAFFIC_WORKLOAD = {"reads": None}
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


def cpu_baseline(n_reads_sample, seed, align_method=4):
    """Reference CPU path on a bounded sample of the same workload (1/10 scale, same coverage)."""
    from oracle import bindings
    from shasta_amd import synthetic
    # Threads actually used = "cores" of the report.  Capped at 64: every reference Align4 thread
    # zero-fills its own 2 GiB arena (src/AssemblerAlign.cpp:353-355) before its first candidate, and a
    # run with one thread per core of a 256-core box (512 GiB of arenas) took the GPU box down.  Also kept
    # under a quarter of the available memory (4 GiB per thread: the arena + the thread's share of the rest).
    cores = min(os.cpu_count() or 1, 64, max(1, available_memory_gib() // 4))
    toc, kmer = make_workload(n_reads_sample, seed)
    data7 = synthetic.pack_markers(toc, kmer)
    p, o = lowhash_params(), (align_options() if align_method == 4 else align3_options())
    if bindings.ref_available():
        lib, kind = bindings.RefLib(), "reference"
        lh = lib.lowhash0(toc, data7, None, p, threads=cores)
        t_lh = lh.seconds
        cand = lh.candidates
        # The per-thread 2 GiB arena of the reference is a fixed setup cost (seconds): time two
        # sample sizes and use the incremental rate.
        n1, n2 = min(len(cand), 4000), min(len(cand), 24000)
        align = lib.align4_batch if align_method == 4 else lib.align3_batch
        t1 = align(toc, data7, cand[:n1], o, want_ordinals=False, threads=cores).seconds
        t2 = align(toc, data7, cand[:n2], o, want_ordinals=False, threads=cores).seconds
        per_pair = (t2 - t1) / (n2 - n1) if n2 > n1 else 0.0
        if per_pair <= 0.0:                                 # too few candidates for the difference to mean anything
            per_pair = t2 / max(1, n2)
    else:
        if not bindings.oracle_available():
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        lib, kind, used = bindings.OracleLib(), "port", cores
        t0 = time.time()
        lh = lib.lowhash0(toc, data7, None, p)
        t_lh = time.time() - t0
        cand = lh.candidates
        n2 = min(len(cand), 24000)
        t0 = time.time()
        (lib.align4_batch if align_method == 4 else lib.align3_batch)(toc, data7, cand[:n2], o, want_ordinals=False, threads=cores)
        per_pair = (time.time() - t0) / max(1, n2)
    pairs = len(cand)
    total = t_lh + pairs * per_pair
    return {
        "value": pairs / total if total > 0 else 0.0,
        "unit": "candidate read-pairs aligned/s",
        "cores": cores,
        "kind": kind,
        "sample": "%d reads (1/10-scale workload, same 45x coverage, M=%d markers): LowHash0 %.2f s on %d threads "
                  "-> %d candidates; align method %d %.3f ms/candidate incremental over %d candidates on %d threads "
                  "(restated DP, reference control flow)" % (
                      n_reads_sample, int(toc[-1]), t_lh, cores, pairs, align_method, per_pair * 1e3, n2, cores),
        "lowhash0_seconds": t_lh,
        "align_seconds_per_pair": per_pair,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lowhash-only", action="store_true", help="BASELINE configs[1]")
    ap.add_argument("--align-method", type=int, default=4, choices=[3, 4],
                    help="4 = Align4 (BASELINE's metric, the default); 3 = the reference's default method, for comparison")
    args = ap.parse_args()

    import torch
    import shasta_amd
    from shasta_amd import abi

    TRAFFIC_WORKLOAD["reads"] = args.reads if int(os.environ.get("WORLD_SIZE", "1")) == 1 else None
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
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

    def align(candidates):
        if args.align_method == 4:
            return ctx.align4(candidates, o, want_ordinals=False, borrow=True)
        return ctx.align3(candidates, o, want_ordinals=False, borrow=True)

    if world == 1:
        # Workload: BASELINE configs[2].
        toc, kmer = make_workload(args.reads, 12345)
        marker_count = int(toc[-1])
        ctx.set_kmer_ids(toc, kmer)                     # host -> HBM, outside the timed region
        del kmer

        def step():
            lh = ctx.lowhash0(p)
            if args.lowhash_only:
                return lh, None, len(lh.candidates)
            al = align(lh.candidates)
            return lh, al, len(lh.candidates)
    else:
        # ONE job over all GPUs (weak scaling: `reads` reads per GPU of one read set at the same
        # coverage).  Every rank generates its own read range, the dense kmer ids are all-gathered
        # over xGMI so that every GPU holds every read (Align4 needs both reads of a candidate),
        # LowHash0 runs sharded with its two all-to-all exchanges, the candidate list is re-split
        # evenly for Align4.  Setup (generation, all-gather) is outside the timed region.
        from shasta_amd import distributed, synthetic
        device = torch.device("cuda", local_rank) if not DRY_RUN_LIBRARY else torch.device("cpu")
        genome_markers = max(20000, int(round(world * args.reads * 1500 / 45.0)))
        toc_s, kmer_s = synthetic.marker_reads(args.reads, genome_markers, mean_markers=1500.0, sigma=0.5, min_markers=790,
                                               keep_probability=0.8, spurious_probability=0.25, k=10, seed=12345,
                                               shard=rank, shard_count=world)
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
        del everything
        marker_count = int(toc[-1])
        read_count = world * args.reads
        backend = distributed.HipBackend(ctx, device)
        boundaries = np.arange(0, read_count + 1, args.reads, dtype=np.uint64)     # the generated shards

        def step():
            lh = distributed.lowhash0(backend, p, read_count, boundaries)
            share, total = distributed.candidate_share(lh.candidates, device)
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
    t0 = time.perf_counter()
    hash_s = hash_n = hash_b = dp_s = dp_cells = dp_bytes = 0
    lh_dev = al_dev = lh_wall = al_wall = 0.0
    fw_s, fw_n, fw_cells, fw_bytes = [0.0] * 6, [0] * 6, [0] * 6, [0] * 6
    tb_s = tb_n = 0
    for _ in range(args.steps):
        lh, al, pairs_total = step()
        kt = ctx.kernel_times()
        hash_s += kt.lowhashHashSeconds; hash_n += kt.lowhashHashLaunches; hash_b += kt.lowhashHashBytes
        if world == 1:
            lh_dev += lh.device_seconds
            lh_wall += lh.seconds
        if al is not None:
            dp_s += kt.alignDpSeconds; dp_cells += kt.alignDpCells; dp_bytes += kt.alignBytes
            for c in range(6):
                fw_s[c] += kt.dpForwardSeconds[c]; fw_n[c] += kt.dpForwardLaunches[c]
                fw_cells[c] += kt.dpForwardCells[c]; fw_bytes[c] += kt.dpForwardBytes[c]
            tb_s += kt.dpTracebackSeconds; tb_n += kt.dpTracebackLaunches
            al_dev += al.device_seconds
            al_wall += al.seconds
    sync()
    elapsed = time.perf_counter() - t0
    stored_total = 0 if al is None else len(al.alignment_data)
    if dist is not None:
        comm = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([stored_total], dtype=torch.int64, device=comm)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        stored_total = int(c[0].item())

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed / steps * 1e3
        value = pairs_total / (elapsed / steps)
        hash_avg = hash_s / max(1, hash_n)
        hash_gbs = (hash_b / max(1, hash_n)) / hash_avg / 1e9 if hash_avg > 0 else 0.0
        kernels = {
            "hashWindowsKernel<4>": {
                "launches_per_step": hash_n // steps, "avg_ms": hash_avg * 1e3,
                "algorithmic_bytes_per_launch": hash_b // max(1, hash_n), "achieved_GBps": hash_gbs,
                "frac_of_hbm_peak": hash_gbs / HBM_PEAK_GBS,      # the HBM-natured kernel of the path
                "seconds_per_step": hash_s / steps,
            },
        }
        traffic_table = load_traffic()
        roofline = {"bound": "hbm", "achieved": hash_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hash_gbs / HBM_PEAK_GBS, "traffic": traffic_of(traffic_table, "hashWindowsKernel"),
                    "kernel": "hashWindowsKernel<4> (LowHash0 K1)"}
        dominant_s = hash_s
        if al is not None and dp_s > 0:
            kernels["align4_banded_dp"] = {
                "seconds_per_step": dp_s / steps, "gcups": dp_cells / dp_s / 1e9,
                "algorithmic_bytes_per_step": dp_bytes // steps, "achieved_GBps": dp_bytes / dp_s / 1e9,
                "note": "forward DP kernels + traceback, both streams; integer VALU-bound wavefront DP, not HBM-bound (SURVEY 8d)",
            }
            # Which forward kernel ran (include/shasta_mi355x.h: shasta_mi355x_dp_forward_version) and what its
            # compiled loop issues per cell and lane (scripts/isa_loop.py on the <32, 2> instantiation: first
            # version 75 VALU instructions per iteration of two cells; second version 247 per eight steady iterations).
            dp_version = lib.dp_forward_version()
            kernel_name = "bandedDpForwardKernel" if dp_version == 1 else "bandedDpForwardKernel2"
            lane_instructions_per_cell = 37.5 if dp_version == 1 else 15.4
            names = [kernel_name + suffix for suffix in ("<16, 2>", "<32, 2>", "<64, 2>", "<64, 4>", "<64, 8>", "<64, 16>")]
            for c in range(6):
                if fw_n[c] == 0:
                    continue
                avg = fw_s[c] / fw_n[c]
                per_launch = fw_bytes[c] / fw_n[c]
                kernels[names[c]] = {"launches_per_step": fw_n[c] // steps, "avg_ms": avg * 1e3,
                                     "algorithmic_bytes_per_launch": int(per_launch), "achieved_GBps": per_launch / avg / 1e9,
                                     "gcups": fw_cells[c] / fw_s[c] / 1e9, "seconds_per_step": fw_s[c] / steps}
                if fw_s[c] > dominant_s:
                    dominant_s = fw_s[c]
                    roofline = {"bound": "hbm", "achieved": per_launch / avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": per_launch / avg / 1e9 / HBM_PEAK_GBS, "traffic": traffic_of(traffic_table, names[c]),
                                "kernel": names[c], "gcups": fw_cells[c] / fw_s[c] / 1e9,
                                # What actually bounds it: VALU instructions per lane and cell in the compiled loop against
                                # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 lane-instructions/s
                                # (MI355X_MICROARCH.md: 157.3 TFLOPS fp32 = 2 flops x that).
                                "valu": {"lane_instructions_per_cell": lane_instructions_per_cell, "peak_lane_instructions_per_s": 78.6e12,
                                         "ceiling_gcups": 78.6e12 / lane_instructions_per_cell / 1e9,
                                         "frac": (fw_cells[c] / fw_s[c]) / (78.6e12 / lane_instructions_per_cell)},
                                "dp_forward_version": dp_version,
                                "note": "dominant kernel by time; integer max-plus DP bound by VALU issue: its algorithmic bytes "
                                        "(4(nx+ny) per task) are tiny against its work (nx x bandWidth cells), so the HBM fraction "
                                        "is low by construction; traffic is dominated by the 2-bit/cell trace it writes"}
            if tb_n:
                kernels["dpTracebackKernel<32>"] = {"launches_per_step": tb_n // steps, "avg_ms": tb_s / tb_n * 1e3,
                                                    "seconds_per_step": tb_s / steps}
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
                "workload": "BASELINE configs[2]: synthetic ONT-like reads, marker level, %d reads/GPU, "
                            "mean 1500 markers (~20 kb) per oriented read, 45x coverage; LowHash0 m=4 f=0.01 "
                            "10 iterations 5/30/5; %s" % (args.reads, "Align4 200/10/10/100, maxBand 1000, 6/-1/-1"
                                                          if args.align_method == 4 else
                                                          "align method 3: downsamplingFactor 0.05, bandExtend 10, maxBand 1000, 6/-1/-1"),
                "reads_per_gpu": args.reads, "markers_total": marker_count,
                "candidates": pairs_total, "alignments_stored": stored_total,
                "kernel_versions": {"dp_forward": lib.dp_forward_version() if al is not None else None,
                                    "window_hash": 1 if os.environ.get("SHASTA_MI355X_HASH") == "1" else 2},
                "parallelism": "1 GPU" if world == 1 else
                               "%d GPUs, one job: reads sharded by id range, RCCL all-to-all of low-hash records and of pair "
                               "keys per MinHash iteration, candidates re-split evenly for Align4" % world,
            },
            "stage_seconds_per_step": {"lowhash0_device": lh_dev / steps, "align4_device": al_dev / steps,
                                       "lowhash0_call": lh_wall / steps, "align4_call": al_wall / steps},
            "kernels": kernels,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(max(2000, args.reads // 10) if not DRY_RUN_LIBRARY else 300, 777, args.align_method)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
