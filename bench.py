#!/usr/bin/env python3
"""bench.py — reads/s classified on the synthetic ONT pile-up (BASELINE.json configs[1]).

A "step" is one pass of the hot path (plan -> sweeps -> scan/compact/classify) over one batch
of overlaps already resident in HBM.  One process per GPU; reads are independent, so ranks get
their own shard (weak scaling: every rank holds a configs[1]-sized batch) and there is no
data-path collective — torch.distributed is used only for the barrier and the max over ranks.

Prints ONE JSON line on rank 0 (see the contract in the task statement) carrying `roofline`
for the dominant kernel (HIP-event time measured inside the engine, on the engine's stream)
and `cpu_baseline` (the CPU oracle timed on this box's cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured-copy ceiling


def usable_cpus():
    """CPUs this process may use: os.cpu_count() capped by the cgroup quota (cpu.max)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--overlaps", type=int, default=5_000_000)
    ap.add_argument("--profile", default="ont", choices=["ont", "sequel", "skewed"])
    ap.add_argument("--coverage", type=int, default=None)
    ap.add_argument("--not-coverage", type=float, default=0.4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lds-sort", action="store_true", help="A/B: LDS-sort kernel for the small class")
    ap.add_argument("--flags", type=int, default=0, help="extra YACRD_F_* engine flags (A/B)")
    ap.add_argument("--engines", type=int, default=2,
                    help="engines (HIP streams) the batches are pipelined over on each GPU")
    ap.add_argument("--full-timing", action="store_true",
                    help="HIP events around every phase and class kernel (slower steps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import yacrd_amd
    from yacrd_amd import host
    yacrd_amd.load_library()  # binds to torch's HIP runtime before torch initialises it

    import torch
    from yacrd_amd import dist as ydist

    # YACRD_BENCH_DEVICE / YACRD_BENCH_BACKEND: plumbing test of the N>1 path on a 1-GPU box
    # (all ranks on one device, gloo); never set by the driver
    dev_index = int(os.environ.get("YACRD_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = ydist.init(backend=os.environ.get("YACRD_BENCH_BACKEND"), device=dev)  # None when WORLD_SIZE == 1

    prof = {"ont": host.SYNTH_ONT, "sequel": host.SYNTH_SEQUEL, "skewed": host.SYNTH_SKEWED}[args.profile]
    cov = args.coverage if args.coverage is not None else (3 if args.profile == "sequel" else 4)
    cfg_no = {"ont": 2, "sequel": 3, "skewed": 4}[args.profile]
    seed = 20241108 + cfg_no + 1000 * rank
    offsets, intervals, lengths = host.synth_csr(prof, args.reads, args.overlaps, seed)
    R, I = args.reads, int(offsets[-1])

    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_iv = torch.from_numpy(intervals.view(np.int32)).to(dev)
    d_len = torch.from_numpy(lengths.view(np.int32)).to(dev)
    torch.cuda.synchronize()

    flags = (yacrd_amd.F_FORCE_LDS_SORT if args.lds_sort else 0) | args.flags
    if args.full_timing:
        flags |= yacrd_amd.F_TIMING_FULL
    # Batches are pipelined over `--engines` engines on this GPU from this one host thread
    # (yacrd_engine_submit_device / yacrd_engine_wait): the plan / compaction kernels, the counter
    # copy and the launch gaps of one batch hide behind the sweep of another (the engines take
    # turns with that launch, so its start / stop events time the kernel, not the queue).
    NE = max(1, min(args.engines, args.steps))
    engs = [yacrd_amd.Engine(device_id=dev_index, flags=flags) for _ in range(NE)]
    eng = engs[0]
    ptrs = (d_off.data_ptr(), d_iv.data_ptr(), d_len.data_ptr(), R, I, cov, args.not_coverage)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k):
        """k passes over the batch, NE of them in flight; returns the last result."""
        if NE == 1:
            res = None
            for _ in range(k):
                res = eng.run_device(*ptrs)
            return res
        inflight = [False] * NE
        res = None
        for i in range(k):
            j = i % NE
            if inflight[j]:
                res = engs[j].wait()
            engs[j].submit_device(*ptrs)
            inflight[j] = True
        for j in range(NE):
            if inflight[j]:
                res = engs[j].wait()
        return res

    run_steps(max(args.warmup, 2 * NE))  # (a submit only pipelines once the engine has a prediction)
    keys = ("plan_ms", "sweep_small_ms", "sweep_medium_ms", "sweep_general_ms", "compact_ms", "total_ms")
    acc = dict.fromkeys(keys, 0.0)
    cls_ms = [0.0] * 12
    for e in engs:
        e.timing_total(reset=True)
    barrier()
    t0 = time.perf_counter()
    outs = [run_steps(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    out = outs[0]
    # HIP events recorded on the engines' streams inside every run_device of the timed region,
    # summed by the engines (one read-back each instead of one per step)
    t, n_timed = None, 0
    for e in engs:
        te, ne = e.timing_total()
        n_timed += ne
        if t is None:
            t = te
        else:
            for k2, v in te.items():
                if k2.endswith("_ms"):
                    t[k2] = [a + b for a, b in zip(t[k2], v)] if isinstance(v, list) else t[k2] + v
    assert n_timed == args.steps
    ev_overhead_ms = eng.event_overhead_ms()
    for k in keys:
        acc[k] = t[k]
    cls_ms = list(t["class_ms"])

    G = int(out.n_regions)
    elapsed = ydist.max_over_ranks(dist, elapsed, dev)

    if rank == 0:
        K = args.steps
        avg = {k: acc[k] / K for k in keys if acc[k] > 0}  # per-phase fields need --full-timing
        # algorithmic bytes per pass, SURVEY.md §8(d): 16 B per overlap + 21 B per read + 8 B per region
        # (8(R+1) + 4R read, 8(R+1) + R written; the survey's "29 B per read" shorthand over-counts)
        b_alg = 8 * I + 8 * (R + 1) + 4 * R + 8 * (R + 1) + 8 * G + R
        # dominant kernel = the size class with the largest own kernel time (HIP events around that
        # launch on the engine's stream); its algorithmic bytes are those of the reads it processed
        ci = max(range(12), key=lambda i: cls_ms[i])
        if t["fused_ms"] >= cls_ms[ci]:  # the row / half-wavefront classes run as one launch
            cname, dom = "R2..H16", "sweep_small_fused_kernel"
            dom_ms = t["fused_ms"] / K
            c_reads, c_iv = t["fused_reads"], t["fused_intervals"]
        else:
            cname = yacrd_amd.CLASS_NAMES[ci]
            dom = yacrd_amd.CLASS_KERNELS[cname]
            dom_ms = cls_ms[ci] / K
            c_reads, c_iv = t["class_reads"][ci], t["class_intervals"][ci]
        b_dom = 8 * c_iv + 21 * c_reads + 16 + 8 * (G * c_reads // max(R, 1))
        # dom_ms: HIP start / stop events attached to the launch itself (hipExtLaunchKernelGGL: the
        # dispatch's own timestamps, the figure rocprofv3 --kernel-trace reports), averaged over
        # every launch of the timed region.  An EMPTY hipEventRecord pair on the same stream measures
        # ~5 us, which is why the events are not recorded around the launch.
        achieved = b_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        avg["class_ms"] = {yacrd_amd.CLASS_NAMES[i]: cls_ms[i] / K for i in range(12) if cls_ms[i] > 0}
        if t["fused_ms"] > 0:
            avg["class_ms"]["R2..H16 (one launch)"] = t["fused_ms"] / K
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("%s_%d_%d" % (args.profile, R, args.overlaps))
            except Exception:
                traffic = None
        line = {
            "metric": "reads_per_sec_classified",
            "value": world * R * K / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic %s pile-up, %d reads / %d PAF overlaps per GPU, -c %d -n %g, inputs resident in HBM"
                                   % (args.profile.upper(), R, args.overlaps, cov, args.not_coverage),
                       "reads_per_gpu": R, "overlaps_per_gpu": args.overlaps, "intervals_per_gpu": I,
                       "regions_per_gpu": G, "parallelism": "read-partition x%d, no collective; %d batches in flight per GPU (one engine each)" % (world, NE)},
            "overlaps_per_sec": world * args.overlaps * K / elapsed,
            "kernel_ms": avg,
            "path_gbps": b_alg / (avg["total_ms"] * 1e-3) / 1e9 if avg.get("total_ms") else None,
            "roofline": {"bound": "hbm", "kernel": dom, "size_class": cname, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes": b_dom, "kernel_ms": dom_ms,
                         "empty_event_bracket_ms": ev_overhead_ms,
                         "kernel_reads": c_reads, "kernel_intervals": c_iv,
                         "whole_path_algorithmic_bytes": b_alg},
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            ncores = usable_cpus()
            l64 = lengths.astype(np.uint64)
            oracle.run(offsets[:1001], intervals[: int(offsets[1000])], l64[:1000], cov, args.not_coverage, 1)
            t1 = time.perf_counter()
            want = oracle.run(offsets, intervals, l64, cov, args.not_coverage, n_threads=ncores)
            cpu_all = time.perf_counter() - t1
            # reference default is -t 1 (src/main.rs:75-77): time a bounded single-thread sample too
            rs = min(R, 20000)
            t1 = time.perf_counter()
            oracle.run(offsets[: rs + 1], intervals[: int(offsets[rs])], l64[:rs], cov, args.not_coverage, 1)
            cpu_1 = time.perf_counter() - t1
            got = eng.fetch()
            parity = bool(np.array_equal(got.bad_offsets, want[0]) and np.array_equal(got.bad_regions, want[1])
                          and np.array_equal(got.read_type, want[2]))
            line["cpu_baseline"] = {"value": R / cpu_all, "unit": "reads/s", "cores": ncores,
                                    "hardware_threads": os.cpu_count(), "kind": "port",
                                    "sample": "the whole batch (%d reads, %d intervals) once on %d threads (= usable CPUs: "
                                              "hardware threads capped by the cgroup cpu.max quota); "
                                              "single-thread (reference default -t 1) on the first %d reads: %.0f reads/s"
                                              % (R, I, ncores, rs, rs / cpu_1),
                                    "value_1thread": rs / cpu_1}
            line["parity"] = "bit-exact vs oracle on all %d reads" % R if parity else "MISMATCH vs oracle"
            # the other half of BASELINE.json's metric: overlaps/s ingested (PAF text -> CSR on the
            # host, the stage in front of the GPU path), on a bounded sample of the same profile
            try:
                import ctypes
                import tempfile
                s_reads, s_ovl = max(R // 5, 2), max(args.overlaps // 5, 2)
                with tempfile.TemporaryDirectory() as td:
                    paf = os.path.join(td, "sample.paf")
                    host.synth_paf(prof, s_reads, s_ovl, 20241110, paf)
                    size = os.path.getsize(paf)
                    hl = host.load_library()
                    rates = {}
                    for th in (1, min(64, ncores)) if ncores > 1 else (1,):
                        best = None
                        for _ in range(2):
                            h = ctypes.c_void_p()
                            t1 = time.perf_counter()
                            rc = hl.yacrd_csr_from_file(paf.encode(), 0, th, ctypes.byref(h))
                            dt = time.perf_counter() - t1
                            if rc != 0:
                                raise RuntimeError("ingest failed")
                            hl.yacrd_csr_free(h)
                            best = dt if best is None else min(best, dt)
                        rates[th] = s_ovl / best
                line["ingest"] = {"overlaps_per_sec": max(rates.values()), "unit": "PAF overlaps/s, text -> CSR, host",
                                  "threads": max(rates, key=rates.get), "overlaps_per_sec_1thread": rates[1],
                                  "sample": "%d reads / %d overlaps, %.0f MB of PAF text" % (s_reads, s_ovl, size / 1e6)}
            except Exception as ex:  # the host library is optional for the GPU metric
                line["ingest"] = {"error": str(ex)}
        print(json.dumps(line), flush=True)
    for e in engs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
