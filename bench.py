#!/usr/bin/env python3
"""bench.py — reads/s classified on the synthetic ONT pile-up (BASELINE.json configs[1]).

A "step" is one pass of the hot path (plan -> sweeps -> scan/compact/classify) over one batch
of overlaps already resident in HBM.  One process per GPU; reads are independent, so ranks get
their own shard (weak scaling: every rank holds a configs[1]-sized batch) and there is no
data-path collective — torch.distributed is used only for the barrier and the max over ranks.

Prints ONE JSON line on rank 0 (see the contract in the task statement) carrying `roofline`
for the dominant kernel (HIP-event time measured inside the engine, on the engine's stream)
and `cpu_baseline` (the CPU oracle timed on this box's cores; rank 0, N=1 only), plus, next to
the kernel-only `value` (SURVEY.md §8d asks for all of them):
  pcie_inclusive  R / (H2D + kernels + D2H): the same batch sent from pinned host memory every step
  end_to_end      overlaps/s from PAF TEXT to read types on the full configs[1] file: parse threads
                  -> pinned buffers -> HBM during the parse -> CSR build on the GPU -> run -> D2H
  large           configs[2] (2 M reads / 200 M overlaps, 3.3 GB: outside the 256 MiB Infinity
                  Cache) read-partitioned over the ranks with yacrd_partition_reads — one GPU at
                  N=1, the north star's strong-scaling workload at N>1 — with oracle parity on a
                  sampled subset of reads.
`--strong` makes the large workload the headline (`value`, "scaling": "strong").
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
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


def alg_bytes(R, I, G):
    """16 B per overlap (8 per interval) + 21 B per read + 8 B per region
    (8(R+1) + 4R read, 8(R+1) + R written; the survey's "29 B per read" shorthand over-counts)."""
    return 8 * I + 8 * (R + 1) + 4 * R + 8 * (R + 1) + 8 * G + R


def defer_kernel_name(fused_intervals):
    """The deferring build's kernel: two groups of list entries per wavefront from 40 M intervals on."""
    return "sweep_small_fused_defer2_kernel" if fused_intervals >= 40_000_000 else "sweep_small_fused_defer_kernel"


def dominant(t, K, yacrd_amd):
    """(kernel name, class name, ms per launch, reads, intervals) of the kernel with the most time;
    K = the launches that carried the events (yacrd_timing.timed_runs)."""
    cls_ms = list(t["class_ms"])
    ci = max(range(12), key=lambda i: cls_ms[i])
    if t["fused_ms"] >= cls_ms[ci]:  # the row / half-wavefront classes run as one launch
        return "sweep_small_fused_kernel", "R2..H16", t["fused_ms"] / K, t["fused_reads"], t["fused_intervals"]
    cname = yacrd_amd.CLASS_NAMES[ci]
    return yacrd_amd.CLASS_KERNELS[cname], cname, cls_ms[ci] / K, t["class_reads"][ci], t["class_intervals"][ci]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--overlaps", type=int, default=5_000_000)
    ap.add_argument("--profile", default="ont", choices=["ont", "sequel", "skewed"])
    ap.add_argument("--jitter", type=int, default=0,
                    help="dovetail ends reflected into the read instead of clamped onto 0 / len "
                         "(YACRD_SYNTH_F_JITTER), sigma = this many positions (SURVEY.md 8d's is 30)")
    ap.add_argument("--coverage", type=int, default=None)
    ap.add_argument("--not-coverage", type=float, default=0.4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lds-sort", action="store_true", help="A/B: LDS-sort kernel for the small class")
    ap.add_argument("--flags", type=int, default=0, help="extra YACRD_F_* engine flags (A/B)")
    ap.add_argument("--engines", type=int, default=3,
                    help="engines (HIP streams) the batches are pipelined over on each GPU")
    ap.add_argument("--host-threads", action="store_true",
                    help="one host thread per engine instead of one thread pipelining all of them")
    ap.add_argument("--time-every-launch", action="store_true",
                    help="start / stop events on the dominant kernel of EVERY step (default: every 8th step of an "
                         "engine, YACRD_F_TIMING_SAMPLED: the events cost ~10 us per step against a 20 us kernel)")
    ap.add_argument("--full-timing", action="store_true",
                    help="HIP events around every phase and class kernel (slower steps)")
    ap.add_argument("--strong", action="store_true",
                    help="headline = the large fixed input partitioned over the ranks (strong scaling)")
    ap.add_argument("--no-extras", action="store_true", help="skip pcie_inclusive / end_to_end / large")
    ap.add_argument("--large-reads", type=int, default=2_000_000)
    ap.add_argument("--large-overlaps", type=int, default=200_000_000)
    ap.add_argument("--large-steps", type=int, default=5)
    args = ap.parse_args()
    if os.environ.get("YACRD_BENCH_STRONG"):
        args.strong = True

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
    sflags = (host.SYNTH_F_JITTER | host.synth_f_sigma(args.jitter)) if args.jitter else 0
    offsets, intervals, lengths = host.synth_csr(prof, args.reads, args.overlaps, seed, flags=sflags)
    R, I = args.reads, int(offsets[-1])

    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_iv = torch.from_numpy(intervals.view(np.int32)).to(dev)
    d_len = torch.from_numpy(lengths.view(np.int32)).to(dev)
    torch.cuda.synchronize()

    flags = (yacrd_amd.F_FORCE_LDS_SORT if args.lds_sort else 0) | args.flags
    if args.full_timing:
        flags |= yacrd_amd.F_TIMING_FULL
    elif not args.time_every_launch:
        flags |= yacrd_amd.F_TIMING_SAMPLED  # (the first run of every engine in the timed region is a timed one)
    # Batches are pipelined over `--engines` engines on this GPU from this one host thread
    # (yacrd_engine_submit_device / yacrd_engine_wait): the plan / compaction kernels, the counter
    # copy, the launch gaps and the host's turn of one batch hide behind the sweep of another.
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
        if args.host_threads and NE > 1:  # one host thread per engine (ctypes calls drop the GIL)
            import threading
            last = [None] * NE

            def work(j, kj):
                for _ in range(kj):
                    last[j] = engs[j].run_device(*ptrs)
            share = [k // NE + (1 if j < k % NE else 0) for j in range(NE)]
            th = [threading.Thread(target=work, args=(j, share[j])) for j in range(NE) if share[j]]
            for x in th:
                x.start()
            for x in th:
                x.join()
            return next(r for r in reversed(last) if r is not None)
        if NE == 1:
            res = None
            for _ in range(k):
                res = eng.run_device(*ptrs)
            return res
        # the submit / wait loop itself lives behind the C ABI (yacrd_engines_run_device_batches):
        # batch i on engine i mod NE, NE batches in flight, one call for the k batches
        return yacrd_amd.run_device_batches(engs, [ptrs] * k)

    run_steps(max(args.warmup, 2 * NE))  # (a submit only pipelines once the engine has a prediction)
    keys = ("plan_ms", "sweep_small_ms", "sweep_medium_ms", "sweep_general_ms", "compact_ms", "total_ms")
    for e in engs:
        e.timing_total(reset=True)
    barrier()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events recorded on the engines' streams inside every run of the timed region, summed by
    # the engines (one read-back each instead of one per step)
    t, n_timed = None, 0
    for e in engs:
        te, ne = e.timing_total()
        n_timed += ne
        if t is None:
            t = te
        else:
            for k2, v in te.items():
                if k2.endswith("_ms") or k2 == "timed_runs":
                    t[k2] = [a + b for a, b in zip(t[k2], v)] if isinstance(v, list) else t[k2] + v
    assert n_timed == args.steps
    ev_overhead_ms = eng.event_overhead_ms()
    # After the timed region: a few steps on one engine with events around every phase and class
    # kernel (YACRD_F_TIMING_FULL: +40 us per step, so never part of `value`) for the per-phase table
    # and the deferred launch's own duration.
    phases = None
    unpredicted = None
    if rank == 0 and not args.no_extras:
        # one engine, no class-count prediction (YACRD_F_NO_PREDICTION: the plan's counts come home
        # before the sweeps are launched), nothing in flight: what a caller with ONE batch per
        # process sees once the inputs are in HBM (the CLI)
        with yacrd_amd.Engine(device_id=dev_index, flags=yacrd_amd.F_NO_PREDICTION | yacrd_amd.F_NO_TIMING) as ue:
            for _ in range(5):
                ue.run_device(*ptrs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                ue.run_device(*ptrs)
            dt = (time.perf_counter() - t0) / 50
            unpredicted = {"ms_per_batch": dt * 1e3, "reads_per_sec": R / dt,
                           "what": "one engine, one batch at a time, no prediction of the class counts (a host sync "
                                   "after the plan kernel), no timing events"}
    if rank == 0 and not args.full_timing:
        with yacrd_amd.Engine(device_id=dev_index, flags=flags | yacrd_amd.F_TIMING_FULL) as fe:
            for _ in range(5):
                fe.run_device(*ptrs)
            fe.timing_total(reset=True)
            for _ in range(30):
                fe.run_device(*ptrs)
            phases, nf = fe.timing_total()
            phases = {k: (v / nf if not isinstance(v, list) else [x / nf for x in v]) for k, v in phases.items()
                      if k.endswith("_ms")}
    G = int(out.n_regions)
    elapsed = ydist.max_over_ranks(dist, elapsed, dev)

    line = None
    if rank == 0:
        K = args.steps
        avg = {k: t[k] / K for k in keys if t[k] > 0}  # per-phase fields need --full-timing
        b_alg = alg_bytes(R, I, G)
        # dominant kernel = the size class with the largest own kernel time; its algorithmic bytes
        # are those of the reads it processed.  dom_ms: HIP start / stop events attached to the
        # launch itself (hipExtLaunchKernelGGL: the dispatch's own timestamps, the figure rocprofv3
        # --kernel-trace reports), averaged over every launch of the timed region.
        n_timed_launches = int(t.get("timed_runs", 0)) or K
        dom, cname, dom_ms, c_reads, c_iv = dominant(t, n_timed_launches, yacrd_amd)
        # reads the fused kernel's filter could not thin are finished by sweep_deferred_kernel: the
        # fused kernel loads and bins them, but its bytes only count the reads it completes
        deferred = int(t.get("deferred_reads", 0)) if cname == "R2..H16" else 0
        if cname == "R2..H16" and t.get("screened"):
            dom = defer_kernel_name(c_iv)
            c_iv -= int(t.get("deferred_intervals", 0))  # exact: summed by the follow-on kernel over the reads it sorts
            c_reads -= deferred
        b_dom = 8 * c_iv + 21 * c_reads + 16 + 8 * (G * c_reads // max(R, 1))
        achieved = b_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        avg["class_ms"] = {yacrd_amd.CLASS_NAMES[i]: t["class_ms"][i] / n_timed_launches for i in range(12) if t["class_ms"][i] > 0}
        if t["fused_ms"] > 0:
            avg["class_ms"]["R2..H16 (one launch)"] = t["fused_ms"] / n_timed_launches
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
            "config": {"workload": "configs[1]: synthetic %s pile-up, %d reads / %d PAF overlaps per GPU, -c %d -n %g; "
                                   "KERNELS ONLY: inputs resident in HBM, the same batch every step, %d batches in "
                                   "flight per GPU, launch grids sized from the previous identical batch's class "
                                   "counts (validated at the final sync); see pcie_inclusive / end_to_end for the "
                                   "rates that include PCIe and the parse"
                                   % (args.profile.upper(), R, args.overlaps, cov, args.not_coverage, NE),
                       "reads_per_gpu": R, "overlaps_per_gpu": args.overlaps, "intervals_per_gpu": I,
                       "regions_per_gpu": G,
                       "parallelism": "read-partition x%d, no collective; %d batches in flight per GPU (one engine each)" % (world, NE)},
            "kernel_overlaps_per_sec": world * args.overlaps * K / elapsed,
            "kernel_ms": avg,
            "unpredicted_single_batch": unpredicted,
            "phases_full_timing_ms": ({k: phases[k] for k in keys + ("fused_ms",) if phases.get(k)}
                                      if phases else None),
            "path_gbps": b_alg / (avg["total_ms"] * 1e-3) / 1e9 if avg.get("total_ms") else None,
            "roofline": {"bound": "hbm", "kernel": dom, "size_class": cname, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes": b_dom, "kernel_ms": dom_ms,
                         "empty_event_bracket_ms": ev_overhead_ms,
                         "timed_launches": n_timed_launches, "launches": K,
                         "kernel_reads": c_reads, "kernel_intervals": c_iv,
                         "deferred_reads": deferred,
                         "finish_compact_kernel_ms": (phases or {}).get("compact_ms"),
                         "whole_path_algorithmic_bytes": b_alg,
                         "note": "batch (82 MB) fits the 256 MiB Infinity Cache: see large.roofline for the "
                                 "same kernel on a 3.3 GB input"},
        }
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline(line, eng, offsets, intervals, lengths, R, I, cov, args)
        if world == 1 and not args.no_extras:
            try:
                line["pcie_inclusive"] = pcie_inclusive(yacrd_amd, engs, offsets, intervals, lengths, cov, args, G)
            except Exception as ex:
                line["pcie_inclusive"] = {"error": repr(ex)}
            try:
                line["end_to_end"] = end_to_end(yacrd_amd, host, eng, prof, R, args.overlaps, cov, args)
            except Exception as ex:
                line["end_to_end"] = {"error": repr(ex)}
    for e in engs[1:]:
        e.close()
    del d_off, d_iv, d_len
    torch.cuda.empty_cache()

    if not args.no_extras or args.strong:
        try:
            large = large_block(yacrd_amd, host, ydist, dist, dev, torch, eng, rank, world, args)
        except Exception as ex:  # the headline must survive a failure here
            large = {"error": repr(ex)}
        if rank == 0:
            line["large"] = large
            if args.strong and "error" not in large:
                line.update({"value": large["reads_per_sec"], "scaling": "strong", "steps": large["steps"],
                             "warmup": large["warmup"], "ms_per_step": large["ms_per_step"]})
                line["config"]["workload"] = large["workload"]
                line["roofline"] = large["roofline"]
    if rank == 0:
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(line, eng, offsets, intervals, lengths, R, I, cov, args):
    """The oracle on this box's cores (kind "port"): the whole batch once on every usable CPU, and
    a bounded single-thread sample (the reference's default is -t 1, src/main.rs:75-77)."""
    import oracle
    ncores = usable_cpus()
    l64 = lengths.astype(np.uint64)
    oracle.run(offsets[:1001], intervals[: int(offsets[1000])], l64[:1000], cov, args.not_coverage, 1)
    t1 = time.perf_counter()
    want = oracle.run(offsets, intervals, l64, cov, args.not_coverage, n_threads=ncores)
    cpu_all = time.perf_counter() - t1
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


def pcie_inclusive(yacrd_amd, engs, offsets, intervals, lengths, cov, args, G):
    """R / (H2D + kernels + D2H): the batch leaves pinned host memory every step
    (yacrd_engine_submit / _collect, two engines: the H2D of one batch overlaps the kernels and the
    D2H of the other), plus one blocking yacrd_engine_run for the unpipelined latency, and the same
    from PAGEABLE memory (threaded bounce-buffer staging)."""
    R, I = len(lengths), int(offsets[-1])
    pins = [yacrd_amd.PinnedArray.copy_of(x) for x in (offsets, intervals, lengths)]
    arrs = tuple(p.array for p in pins)
    e0 = engs[0]
    h2d_bytes = offsets.nbytes + intervals.nbytes + lengths.nbytes
    d2h_bytes = 8 * (R + 1) + 8 * G + R
    # blocking, one engine
    for _ in range(2):
        e0.run(*arrs, cov, args.not_coverage)
    K1 = 10
    e0.timing_total(reset=True)
    t0 = time.perf_counter()
    for _ in range(K1):
        e0.run(*arrs, cov, args.not_coverage)
    blocking = (time.perf_counter() - t0) / K1
    tt, _ = e0.timing_total(reset=True)
    h2d_ms, d2h_ms = tt["h2d_ms"] / K1, tt["d2h_ms"] / K1
    # pipelined over the engines
    NE = len(engs)
    K2 = 40

    def steps(k):
        inflight = [False] * NE
        for i in range(k):
            j = i % NE
            if inflight[j]:
                engs[j].collect()
            engs[j].submit(*arrs, cov, args.not_coverage)
            inflight[j] = True
        for j in range(NE):
            if inflight[j]:
                engs[j].collect()
    steps(2 * NE)
    t0 = time.perf_counter()
    steps(K2)
    piped = (time.perf_counter() - t0) / K2
    # pageable source
    for _ in range(2):
        e0.run(offsets, intervals, lengths, cov, args.not_coverage)
    e0.timing_total(reset=True)
    t0 = time.perf_counter()
    for _ in range(5):
        e0.run(offsets, intervals, lengths, cov, args.not_coverage)
    pageable = (time.perf_counter() - t0) / 5
    tp, _ = e0.timing_total(reset=True)
    for p in pins:
        p.close()
    return {"reads_per_sec": R / piped, "ms_per_batch": piped * 1e3, "engines": NE, "batches": K2,
            "overlaps_per_sec": (I // 2) / piped,
            "blocking_ms_per_batch": blocking * 1e3, "blocking_reads_per_sec": R / blocking,
            "h2d_ms": h2d_ms, "h2d_GBps": h2d_bytes / (h2d_ms * 1e-3) / 1e9 if h2d_ms else None,
            "d2h_ms": d2h_ms, "h2d_bytes": h2d_bytes, "d2h_bytes": d2h_bytes,
            "pageable_ms_per_batch": pageable * 1e3,
            "pageable_h2d_GBps": h2d_bytes / (tp["h2d_ms"] / 5 * 1e-3) / 1e9 if tp["h2d_ms"] else None,
            "source": "pinned host memory (yacrd_pinned_alloc), the same batch every step; PCIe Gen5 x16 spec 63 GB/s"}


def end_to_end(yacrd_amd, host, eng, prof, R, O, cov, args):
    """BASELINE's "overlaps/sec ingested -> reads/sec classified" as ONE number: the full
    configs[1] PAF text (not a sample) to read types.  Parse threads fill pinned buffers, every
    full buffer crosses PCIe at once (yacrd_stream_*), the CSR is built on the GPU, the engine
    runs, results come home.  Also: the parse alone into a host CSR, by thread count."""
    ncores = usable_cpus()
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    paf = os.path.join(d, "yacrd_bench_%d_%d_%d.paf" % (os.getpid(), R, O))
    hl = host.load_library()
    el = yacrd_amd.load_library()
    try:
        t0 = time.perf_counter()
        host.synth_paf(prof, R, O, 20241110, paf)
        gen_s = time.perf_counter() - t0
        size = os.path.getsize(paf)
        out = {"paf_bytes": size, "overlaps": O, "reads": R, "generate_s": gen_s, "file_on": d}
        best = None
        with yacrd_amd.Stream(eng) as st:
            for th in sorted(set([ncores, min(2 * ncores, 64)])):
                for rep in range(3):
                    sink = st.sink()
                    h = ctypes.c_void_p()
                    t0 = time.perf_counter()
                    rc = hl.yacrd_ingest_stream(paf.encode(), 0, th, ctypes.addressof(sink), ctypes.byref(h))
                    t1 = time.perf_counter()
                    if rc != 0:
                        raise RuntimeError(hl.yacrd_host_last_error().decode())
                    v = host._View()
                    hl.yacrd_csr_get(h, ctypes.byref(v))
                    mp, nh = ctypes.POINTER(ctypes.c_uint32)(), ctypes.c_uint64()
                    hl.yacrd_csr_handle_map(h, ctypes.byref(mp), ctypes.byref(nh))
                    res = yacrd_amd.engine._Result()
                    rc = el.yacrd_stream_finish(st._h, mp, nh, v.lengths, v.n_reads, cov, args.not_coverage,
                                                ctypes.byref(res))
                    t2 = time.perf_counter()
                    if rc != 0:
                        raise RuntimeError(el.yacrd_last_error().decode())
                    n_reads, n_regions = int(res.n_reads), int(res.n_regions)
                    el.yacrd_result_free(ctypes.byref(res))
                    hl.yacrd_csr_free(h)
                    if best is None or t2 - t0 < best["s"]:
                        s = st.stats()
                        best = {"s": t2 - t0, "threads": th, "parse_s": t1 - t0, "finish_ms": (t2 - t1) * 1e3,
                                "h2d_GBps_while_busy": s["h2d_bytes"] / max(s["h2d_busy_ms"], 1e-6) / 1e6,
                                "h2d_bytes": s["h2d_bytes"], "csr_build_on_gpu_ms": s["build_ms"],
                                "engine_run_ms": s["run_ms"], "d2h_ms": s["d2h_ms"],
                                "reads_found": n_reads, "regions": n_regions}
        out.update({"overlaps_per_sec": O / best["s"], "reads_per_sec": best["reads_found"] / best["s"],
                    "text_GBps": size / best["s"] / 1e9, "seconds": best["s"], "stream": best,
                    "path": "PAF text -> yacrd_ingest_stream (pread blocks, shared id table) -> pinned buffers -> "
                            "hipMemcpyAsync during the parse -> CSR build on the GPU -> engine -> D2H"})
        rates = {}
        for th in sorted(set([1, ncores])):
            bt = None
            for _ in range(2 if th > 1 else 1):
                h = ctypes.c_void_p()
                t0 = time.perf_counter()
                if hl.yacrd_csr_from_file(paf.encode(), 0, th, ctypes.byref(h)) != 0:
                    raise RuntimeError(hl.yacrd_host_last_error().decode())
                dt = time.perf_counter() - t0
                hl.yacrd_csr_free(h)
                bt = dt if bt is None else min(bt, dt)
            rates[str(th)] = O / bt
        out["host_csr_ingest_overlaps_per_sec_by_threads"] = rates
        return out
    finally:
        if os.path.exists(paf):
            os.remove(paf)


def large_traffic(R, O):
    """PMC traffic of the dominant kernel on this workload, when profiles/traffic.json has it."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("sequel_%d_%d" % (R, O))
    except Exception:
        return None


def large_block(yacrd_amd, host, ydist, dist, dev, torch, eng, rank, world, args):
    """configs[2] shape, one fixed input for every N: 2 M reads / 200 M overlaps (Sequel-like,
    -c 3 -n 0.4).  yacrd_partition_reads cuts contiguous read ranges balanced by interval count,
    rank r takes range r (its own GPU, no collective), time = max over ranks."""
    R, O, cov, nc = args.large_reads, args.large_overlaps, 3, 0.4
    t0 = time.perf_counter()
    offsets, intervals, lengths = host.synth_csr(host.SYNTH_SEQUEL, R, O, 20241108 + 3)
    gen_s = time.perf_counter() - t0
    cuts = yacrd_amd.partition_reads(offsets, world)
    r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
    off, iv, ln = ydist.local_csr(offsets, intervals, lengths, r0, r1)
    Rl, Il = r1 - r0, int(off[-1])
    t0 = time.perf_counter()
    d_off = torch.from_numpy(np.ascontiguousarray(off).view(np.int64)).to(dev)
    d_iv = torch.from_numpy(np.ascontiguousarray(iv).view(np.int32)).to(dev)
    d_len = torch.from_numpy(np.ascontiguousarray(ln).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    upload_s = time.perf_counter() - t0
    eng = yacrd_amd.Engine(device_id=dev.index)  # its own engine: every launch carries the events (1 ms kernels)
    ptrs = (d_off.data_ptr(), d_iv.data_ptr(), d_len.data_ptr(), Rl, Il, cov, nc)
    W, K = 2, max(1, args.large_steps)
    for _ in range(W):
        res = eng.run_device(*ptrs)
    eng.timing_total(reset=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        res = eng.run_device(*ptrs)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    elapsed = ydist.max_over_ranks(dist, mine, dev)
    t, _ = eng.timing_total()
    G = int(res.n_regions)
    deferred_ms = 0.0
    if rank == 0:  # the follow-on kernel's duration (it sorts the deferred reads): two extra steps with events around everything
        with yacrd_amd.Engine(device_id=dev.index, flags=yacrd_amd.F_TIMING_FULL) as fe:
            fe.run_device(*ptrs)
            fe.timing_total(reset=True)
            fe.run_device(*ptrs)
            fe.run_device(*ptrs)
            tf, nf = fe.timing_total()
            deferred_ms = tf.get("compact_ms", 0.0) / max(nf, 1)
    # oracle parity on a sample of this rank's reads (every ~100th read, at most 20 000)
    import oracle
    got = eng.fetch()
    step = max(1, Rl // 20000)
    pick = np.arange(0, Rl, step)
    n = np.diff(off.astype(np.int64))[pick]
    s_off = np.zeros(len(pick) + 1, np.uint64)
    s_off[1:] = np.cumsum(n)
    idx = np.concatenate([np.arange(int(off[p]), int(off[p + 1])) for p in pick]) if len(pick) else np.zeros(0, np.int64)
    s_iv = np.asarray(iv)[idx]
    want = oracle.run(s_off, s_iv, np.asarray(ln)[pick].astype(np.uint64), cov, nc, n_threads=usable_cpus())
    ok = True
    for j, p in enumerate(pick):
        a, b = int(got.bad_offsets[p]), int(got.bad_offsets[p + 1])
        wa, wb = int(want[0][j]), int(want[0][j + 1])
        if b - a != wb - wa or not np.array_equal(got.bad_regions[a:b], want[1][wa:wb]) or got.read_type[p] != want[2][j]:
            ok = False
            break
    per_rank = {"rank": rank, "reads": Rl, "intervals": Il, "ms_per_step": mine / K * 1e3,
                "parity_sample_ok": ok, "sampled_reads": int(len(pick))}
    if dist is not None:
        allr = [None] * world
        dist.all_gather_object(allr, per_rank)
    else:
        allr = [per_rank]
    if rank != 0:
        del d_off, d_iv, d_len
        return None
    dom, cname, dom_ms, c_reads, c_iv = dominant(t, K, yacrd_amd)
    deferred = int(t.get("deferred_reads", 0)) if cname == "R2..H16" else 0
    if cname == "R2..H16" and t.get("screened"):
        dom = defer_kernel_name(c_iv)
        c_iv -= int(t.get("deferred_intervals", 0))
        c_reads -= deferred
    Gl = G
    b_dom = 8 * c_iv + 21 * c_reads + 16 + 8 * (Gl * c_reads // max(Rl, 1))
    ach = b_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    ivs = [p["intervals"] for p in allr]
    b_all = alg_bytes(R, int(offsets[-1]), G * world if world > 1 else G)
    out = {"workload": "configs[2]: synthetic SEQUEL pile-up, %d reads / %d PAF overlaps in total, -c %d -n %g, "
                       "read-partitioned over %d GPU(s) by yacrd_partition_reads (contiguous ranges balanced by "
                       "interval count, no collective); KERNELS ONLY, inputs resident in HBM" % (R, O, cov, nc, world),
           "reads_per_sec": R * K / elapsed, "kernel_overlaps_per_sec": O * K / elapsed,
           "ms_per_step": elapsed / K * 1e3, "steps": K, "warmup": W, "n_gpus": world,
           "generate_s": gen_s, "upload_s": upload_s,
           "interval_imbalance_max_over_min": max(ivs) / max(1, min(ivs)),
           "per_rank": allr,
           "whole_path_GBps": b_all / (elapsed / K) / 1e9,
           "parity": ("bit-exact vs oracle on %d sampled reads per rank" % per_rank["sampled_reads"])
                     if all(p["parity_sample_ok"] for p in allr) else "MISMATCH vs oracle",
           "roofline": {"bound": "hbm", "kernel": dom, "size_class": cname, "achieved": ach, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": large_traffic(R, O) if world == 1 else None, "algorithmic_bytes": b_dom,
                        "kernel_ms": dom_ms, "kernel_reads": c_reads, "kernel_intervals": c_iv,
                        "deferred_reads": deferred, "finish_compact_kernel_ms": deferred_ms,
                        "note": "rank 0's launch; input %.2f GB per GPU, outside the 256 MiB Infinity Cache" % (8 * Il / 1e9)}}
    del d_off, d_iv, d_len
    return out


if __name__ == "__main__":
    main()
