#!/usr/bin/env python3
"""bench.py — reads/s classified by the MI355X bad-region engine on BASELINE.json's synthetic pile-ups.

A "step" is one pass of the hot path (plan -> sweeps -> finish / scan / compact / classify) over one
batch of overlaps already resident in HBM.  One process per GPU; reads are independent, so the input is
cut into contiguous read ranges by yacrd_partition_reads (balanced by interval count), rank r sweeps
range r on its own GPU, and there is no data-path collective — torch.distributed is used only for the
barrier and the max over ranks.

HEADLINE (`value`, "scaling": "strong"): BASELINE.json configs[4] at full size — the north star's target
workload: 5 M reads / 500 M PAF overlaps = 1 G intervals, Sequel-like lengths, -c 3 -n 0.4, 8 GB of input per
pass — the same fixed input for every N (generated once per node: rank 0 writes the CSR to /dev/shm, the other
ranks map their slices).  `--config 2` / `--config 3` put another config there; `--weak` brings back rounds
1-2's headline (every rank a configs[1]-sized batch of its own, batches pipelined over three engines).

Rank 0 prints ONE JSON line carrying `roofline` for the dominant kernel (start / stop events attached to
every launch of the timed region, on the engine's stream) and `cpu_baseline` (the CPU oracle on this
box's cores on a sample of the headline workload; N = 1 only), plus, at N = 1 (SURVEY.md §8d asks for all of them):
  configs2        configs[2]: 2 M reads / 200 M overlaps (rounds 2-3's headline)
  skewed          configs[3]: 10 k ultra-long reads, 57 M intervals (workgroup / device-wide screens + fallbacks)
  small_batches   configs[1] (100 k reads / 5 M overlaps, -c 4): batches pipelined over three engines,
                  one batch at a time without class-count prediction, per-phase times
  jitter          configs[1] and configs[2] from the generator that REFLECTS the dovetail ends' offsets
                  into the read instead of clamping them onto 0 / len (no exact position holds
                  a pile), sigma = 30, 100 and 300 positions: healthy / deferred reads, kernel ms,
                  roofline fraction, oracle parity
  pcie_inclusive  R / (H2D + kernels + D2H): a configs[1] batch sent from pinned host memory every step
  end_to_end      overlaps/s from PAF TEXT to read types on the full configs[1] file
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
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; the blocks below keep three
# engines (a stream each) beside torch's, create and destroy dozens over a run, and two engines that land on ONE queue run their
# batches one after the other: configs[1]'s pipelined batches measured 60 us at sigma 300 with eight queues or a lucky four, 83 with
# two, 130-290 on boxes / builds where the mapping collided (profiles/r06/u_queues.log).  Must be set before
# the HIP runtime starts; a caller's own setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured-copy ceiling

# BASELINE.json configs (0-based): profile, reads, overlaps, -c, -n, generator seed
CONFIGS = {
    1: ("ont", 100_000, 5_000_000, 4, 0.4, 20241108 + 2),
    2: ("sequel", 2_000_000, 200_000_000, 3, 0.4, 20241108 + 3),
    3: ("skewed", 10_000, 30_000_000, 4, 0.4, 20241108 + 4),
    4: ("sequel", 5_000_000, 500_000_000, 3, 0.4, 20241108 + 5),
}


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


# which path the timed runs of a block took (yacrd_timing, ABI 7): counts over its engines — what a short batch's latency
# depends on besides its input (one box measured configs[1] at sigma 300 2.2 x slower than every other)
PATH_KEYS = ("predicted", "prediction_misses", "fused_reruns", "build_switches", "sorting_build", "screen_wide", "screened", "one_launch")


def dominant(t, K, yacrd_amd):
    """(kernel name, class name, ms per launch, reads, intervals) of the kernel with the most time;
    K = the launches that carried the events (yacrd_timing.timed_runs)."""
    cls_ms = list(t["class_ms"])
    ci = max(range(12), key=lambda i: cls_ms[i])
    if t["fused_ms"] >= cls_ms[ci]:  # the row / half-wavefront classes run as one launch
        return "sweep_small_fused_kernel", "R2..H16", t["fused_ms"] / K, t["fused_reads"], t["fused_intervals"]
    cname = yacrd_amd.CLASS_NAMES[ci]
    return yacrd_amd.CLASS_KERNELS[cname], cname, cls_ms[ci] / K, t["class_reads"][ci], t["class_intervals"][ci]


def traffic_entry(key):
    """PMC traffic of the dominant kernel on a named workload, as recorded in profiles/traffic.json
    ({"bytes": ..., "measured": date, "kernel": ..., "source": file}); None when there is no entry."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except Exception:
        return None
    return e if isinstance(e, dict) else None


def roofline_of(yacrd_amd, t, n_launches, R, G, key, note):
    """`roofline` of the dominant kernel from an engine's timing sums: algorithmic bytes of the reads it
    COMPLETES (the reads the screen defers are loaded by it but finished, and counted, elsewhere; their
    number and their intervals are exact: summed by the follow-on kernel) / its own start / stop events."""
    K = int(t.get("timed_runs", 0)) or n_launches
    dom, cname, dom_ms, c_reads, c_iv = dominant(t, K, yacrd_amd)
    deferred = 0
    if cname == "R2..H16" and t.get("screened"):
        items = int(t.get("screen_items", 0)) or (2 if c_iv >= 40_000_000 else 1)  # (what the engine's last run used)
        dom = ("sweep_small_fused_defer2_kernel" if items == 2 else
               "sweep_small_fused_defer_wide_kernel" if int(t.get("screen_wide", 0)) else "sweep_small_fused_defer_kernel")
        deferred = int(t.get("deferred_reads", 0))
        c_iv -= int(t.get("deferred_intervals", 0))
        c_reads -= deferred
    if cname in ("M1", "M2", "BIG"):
        note += ("; kernel_ms brackets the class's launches (M1 / M2: the screen, the launch over what it leaves — table again, filtered "
                 "exact sweep, whole-read sort — and the usually empty overflow kernel behind it; BIG: the device-wide screen's four, on a "
                 "side stream), `traffic` is the screen's + the second launch's: profiles/r06_kernel_stats_configs3.csv has them all")
    b_dom = 8 * c_iv + 21 * c_reads + 16 + 8 * (G * c_reads // max(R, 1))
    ach = b_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    tr = traffic_entry(key) if key else None
    return {"bound": "hbm", "kernel": dom, "size_class": cname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": tr["bytes"] if tr else None,
            "traffic_source": ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE (separate passes) of "
                               "%s, measured %s (%s); not re-measured by this run" % (tr.get("kernel"), tr.get("measured"), tr.get("source"))
                               if tr else None),
            "traffic_source_short": ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this kernel on this "
                                     "workload, %s; not re-measured by this run" % tr.get("measured") if tr else None),
            "algorithmic_bytes": b_dom, "kernel_ms": dom_ms, "timed_launches": K, "launches": n_launches,
            "kernel_reads": c_reads, "kernel_intervals": c_iv, "deferred_reads": deferred,
            "deferred_intervals": int(t.get("deferred_intervals", 0)) if deferred else 0, "note": note}


class Ctx:
    """What every block needs: the bindings, torch, this rank's device and the process group."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        import yacrd_amd
        from yacrd_amd import host
        yacrd_amd.load_library()  # binds to torch's HIP runtime before torch initialises it
        import torch
        from yacrd_amd import dist as ydist
        self.ya, self.host, self.torch, self.ydist = yacrd_amd, host, torch, ydist
        # YACRD_BENCH_DEVICE / YACRD_BENCH_BACKEND: plumbing test of the N>1 path on a 1-GPU box
        # (all ranks on one device, gloo); never set by the driver
        self.dev_index = int(os.environ.get("YACRD_BENCH_DEVICE", self.local_rank))
        if args.gpus != self.world:  # (launch_ranks has made them agree; a caller that builds Ctx by hand has not)
            raise SystemExit("bench.py: --gpus %r but %d rank(s) are running" % (args.gpus, self.world))
        if self.dev_index >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d wants GPU %d, this node shows %d (one process per GPU: --gpus N needs N GPUs)"
                             % (self.rank, self.dev_index, torch.cuda.device_count()))
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        self.dist = ydist.init(backend=os.environ.get("YACRD_BENCH_BACKEND"), device=self.dev)  # None when WORLD_SIZE == 1

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def prof(self, name):
        return {"ont": self.host.SYNTH_ONT, "sequel": self.host.SYNTH_SEQUEL, "skewed": self.host.SYNTH_SKEWED}[name]

    def sflags(self, jitter, chimeras=0):
        f = (self.host.SYNTH_F_JITTER | self.host.synth_f_sigma(jitter)) if jitter else 0
        return f | (self.host.synth_f_chimera_pct(chimeras) if chimeras else 0)


def oracle_sample_parity(got, off, iv, ln, cov, nc, max_reads):
    """The engine's result against the oracle on every k-th read of a CSR (at most ~max_reads of them)."""
    import oracle
    Rl = len(ln)
    step = max(1, Rl // max_reads)
    pick = np.arange(0, Rl, step)
    off64 = np.asarray(off).astype(np.int64)
    n = np.diff(off64)[pick]
    s_off = np.zeros(len(pick) + 1, np.uint64)
    s_off[1:] = np.cumsum(n)
    idx = np.repeat(off64[pick] - s_off[:-1].astype(np.int64), n) + np.arange(int(s_off[-1]))
    s_iv = np.ascontiguousarray(np.asarray(iv).reshape(-1, 2)[idx])
    want = oracle.run(s_off, s_iv, np.asarray(ln)[pick].astype(np.uint64), cov, nc, n_threads=usable_cpus())
    g_off = got.bad_offsets.astype(np.int64)
    g_n = np.diff(g_off)[pick]
    w_off = want[0].astype(np.int64)
    ok = bool(np.array_equal(g_n, np.diff(w_off)) and np.array_equal(got.read_type[pick], want[2]))
    if ok and int(w_off[-1]):
        gi = np.repeat(g_off[pick] - w_off[:-1], g_n) + np.arange(int(w_off[-1]))
        ok = bool(np.array_equal(got.bad_regions[gi], want[1]))
    return ok, int(len(pick))


def shared_csr(cx, profile, R, O, seed, flags):
    """The synthetic CSR of a fixed input, generated ONCE per node: with one rank, straight from the generator;
    with several, rank 0 generates and writes the three arrays to /dev/shm, the others map them read-only (the
    generator runs on every hardware thread the cgroup allows: N ranks generating the whole input side by side
    share that quota and take N times as long — 43 s x 8 for configs[4]).  Every rank must call this."""
    host = cx.host
    if cx.world == 1:
        return host.synth_csr(cx.prof(profile), R, O, seed, flags=flags)
    # where: the first of /dev/shm, the temp directory, the checkout that is writable and has ROOM (a container's /dev/shm is
    # often 64 MB: configs[4] is 8 GB) — rank 0 looks and tells the others; nowhere: every rank generates for itself
    need = 16 * O + 12 * R + (64 << 20)
    where = [None]
    if cx.rank == 0:
        dirs = os.environ.get("YACRD_BENCH_SHARE_DIRS")  # (tests: the candidates, colon-separated)
        for d in (dirs.split(":") if dirs is not None else ("/dev/shm", tempfile.gettempdir(), ROOT)):
            try:
                st = os.statvfs(d)
                if os.path.isdir(d) and os.access(d, os.W_OK) and st.f_bavail * st.f_frsize > need:
                    where[0] = d
                    break
            except OSError:
                pass
    cx.dist.broadcast_object_list(where, src=0)
    if where[0] is None:
        return host.synth_csr(cx.prof(profile), R, O, seed, flags=flags)
    tag = os.path.join(where[0], "yacrd_bench_csr_%s_%d_%d_%d_%d" % (os.environ.get("MASTER_PORT", "0"), R, O, seed, flags))
    names = [tag + sfx for sfx in (".off.npy", ".iv.npy", ".len.npy")]
    ok = [True]
    if cx.rank == 0:
        arrs = host.synth_csr(cx.prof(profile), R, O, seed, flags=flags)
        try:
            for nm, a in zip(names, arrs):
                np.save(nm, a)
        except OSError:  # (the room went away meanwhile: the others generate for themselves)
            ok[0] = False
            for nm in names:
                try:
                    os.remove(nm)
                except OSError:
                    pass
        cx.dist.broadcast_object_list(ok, src=0)
        if ok[0]:
            cx.shared_files = getattr(cx, "shared_files", []) + names
        return arrs
    cx.dist.broadcast_object_list(ok, src=0)
    if not ok[0]:
        return host.synth_csr(cx.prof(profile), R, O, seed, flags=flags)
    return tuple(np.load(nm, mmap_mode="r") for nm in names)


def drop_shared(cx):
    """Remove what shared_csr wrote (after a barrier: every rank has copied its slice)."""
    if cx.world > 1:
        cx.dist.barrier()
    for nm in getattr(cx, "shared_files", []):
        try:
            os.remove(nm)
        except OSError:
            pass
    cx.shared_files = []


def resident_block(cx, label, profile, R, O, cov, nc, seed, jitter, steps, warmup, parity_reads, traffic_key=None,
                   keep_host=False):
    """One fixed input for every N, read-partitioned over the ranks: generate, upload, W untimed + K timed
    passes bracketed by barrier + synchronize, max over ranks; dominant-kernel events on every launch.
    Every rank must call this (barriers, all_gather); rank 0 gets the block, the others None."""
    ya, host, torch, ydist, dist = cx.ya, cx.host, cx.torch, cx.ydist, cx.dist
    t0 = time.perf_counter()
    offsets, intervals, lengths = shared_csr(cx, profile, R, O, seed, cx.sflags(jitter))
    gen_s = time.perf_counter() - t0
    I_all = int(offsets[-1])
    cuts = ya.partition_reads(offsets, cx.world)
    r0, r1 = int(cuts[cx.rank]), int(cuts[cx.rank + 1])
    off, iv, ln = ydist.local_csr(offsets, intervals, lengths, r0, r1)
    if cx.world > 1:  # (slices of the shared maps: private copies, the files go away below)
        off, iv, ln = np.array(off), np.array(iv), np.array(ln)  # (copies: the maps are read-only and about to go)
        drop_shared(cx)
    Rl, Il = r1 - r0, int(off[-1])
    t0 = time.perf_counter()
    d_off = torch.from_numpy(np.ascontiguousarray(off).view(np.int64)).to(cx.dev)
    d_iv = torch.from_numpy(np.ascontiguousarray(iv).view(np.int32)).to(cx.dev)
    d_len = torch.from_numpy(np.ascontiguousarray(ln).view(np.int32)).to(cx.dev)
    torch.cuda.synchronize()
    upload_s = time.perf_counter() - t0
    # NE engines (streams) per GPU, NE passes in flight: the plan, the scan and the launch gaps of one pass hide behind the
    # screen of another (yacrd_engines_run_device_batches: what a caller with many batches does); NE = 1: one pass at a time
    NE = max(1, min(int(getattr(cx.args, "resident_engines", 1)), max(1, steps)))
    engs = [ya.Engine(device_id=cx.dev_index, flags=cx.args.flags) for _ in range(NE)]  # every launch of the dominant kernel carries its events
    eng = engs[0]
    ptrs = (d_off.data_ptr(), d_iv.data_ptr(), d_len.data_ptr(), Rl, Il, cov, nc)
    W, K = max(1, warmup), max(1, steps)

    def run_steps(k):
        if NE == 1:
            r_ = None
            for _ in range(k):
                r_ = eng.run_device(*ptrs)
            return r_
        return ya.run_device_batches(engs, [ptrs] * k)

    for e_ in engs:  # (every engine's first pass sizes its buffers and gives it a prediction)
        for _ in range(2 if NE > 1 else 0):
            e_.run_device(*ptrs)
    res = run_steps(max(W, NE))
    for e_ in engs:
        e_.timing_total(reset=True)
    cx.barrier()
    t0 = time.perf_counter()
    res = run_steps(K)
    cx.barrier()
    mine = time.perf_counter() - t0
    elapsed = ydist.max_over_ranks(dist, mine, cx.dev)
    t, nt = None, 0
    for e_ in engs:
        te, ne = e_.timing_total()
        nt += ne
        if t is None:
            t = te
        else:
            for k2, v in te.items():
                if k2.endswith("_ms") or k2 in PATH_KEYS or k2 == "timed_runs":
                    t[k2] = [a_ + b_ for a_, b_ in zip(t[k2], v)] if isinstance(v, list) else t[k2] + v
    assert nt == K
    one_engine_ms = None
    if NE > 1 and cx.rank == 0:  # the same passes one at a time on one engine, beside it (never part of `value`)
        cx.torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(max(3, K // 2)):
            eng.run_device(*ptrs)
        one_engine_ms = (time.perf_counter() - t1) / max(3, K // 2) * 1e3
    G = int(res.n_regions)
    phases = None
    if cx.rank == 0:  # per-phase times: two extra passes with events around everything (never part of `value`)
        with ya.Engine(device_id=cx.dev_index, flags=ya.F_TIMING_FULL | cx.args.flags) as fe:
            fe.run_device(*ptrs)
            fe.timing_total(reset=True)
            fe.run_device(*ptrs)
            fe.run_device(*ptrs)
            tf, nf = fe.timing_total()
            phases = {k: tf[k] / max(nf, 1) for k in ("plan_ms", "sweep_small_ms", "sweep_medium_ms", "sweep_general_ms",
                                                      "compact_ms", "total_ms") if tf.get(k)}
    got = eng.fetch()
    ok, n_checked = oracle_sample_parity(got, off, iv, ln, cov, nc, parity_reads)
    per_rank = {"rank": cx.rank, "reads": Rl, "intervals": Il, "ms_per_step": mine / K * 1e3, "regions": G,
                "parity_sample_ok": ok, "sampled_reads": n_checked}
    if dist is not None:
        allr = [None] * cx.world
        dist.all_gather_object(allr, per_rank)
    else:
        allr = [per_rank]
    keep = (offsets, intervals, lengths) if keep_host and cx.rank == 0 else None
    for e_ in engs:
        e_.close()
    del d_off, d_iv, d_len
    torch.cuda.empty_cache()
    if cx.rank != 0:
        return None, None
    G_all = sum(p["regions"] for p in allr)
    b_all = alg_bytes(R, I_all, G_all)
    ivs = [p["intervals"] for p in allr]
    jit = ", dovetail ends REFLECTED into the read (sigma %d positions) instead of clamped onto 0 / len" % jitter if jitter else ""
    screened = bool(t.get("screened"))
    out = {"workload": "%s: synthetic %s pile-up%s, %d reads / %d PAF overlaps in total, -c %d -n %g, read-partitioned "
                       "over %d GPU(s) by yacrd_partition_reads (contiguous ranges balanced by interval count, no "
                       "collective); KERNELS ONLY: inputs resident in HBM, %s, every pass the whole input"
                       % (label, profile.upper(), jit, R, O, cov, nc, cx.world,
                          "one engine per GPU, one pass at a time" if NE == 1 else
                          "%d engines (streams) per GPU, %d passes in flight (yacrd_engines_run_device_batches)" % (NE, NE)),
           "reads": R, "overlaps": O, "intervals": I_all, "regions": G_all,
           "reads_per_sec": R * K / elapsed, "kernel_overlaps_per_sec": O * K / elapsed,
           "ms_per_step": elapsed / K * 1e3, "steps": K, "warmup": W, "n_gpus": cx.world,
           "engines_per_gpu": NE, "one_engine_one_pass_at_a_time_ms": one_engine_ms,
           "generate_s": gen_s, "upload_s": upload_s,
           "interval_imbalance_max_over_min": max(ivs) / max(1, min(ivs)),
           "per_rank": allr,
           "whole_path_algorithmic_bytes": b_all, "whole_path_GBps": b_all / (elapsed / K) / 1e9,
           "whole_path_frac_of_peak": b_all / (elapsed / K) / 1e9 / HBM_PEAK_GBS / cx.world,
           "phases_full_timing_ms": phases,
           "paths": dict({k: int(t.get(k, 0)) for k in PATH_KEYS}, runs=K),
           "healthy_reads_rank0": int(t.get("fused_reads", 0)) - int(t.get("deferred_reads", 0)) if screened else None,
           "deferred_reads_rank0": int(t.get("deferred_reads", 0)) if screened else None,
           "parity": ("bit-exact vs oracle on %d sampled reads per rank" % per_rank["sampled_reads"])
                     if all(p["parity_sample_ok"] for p in allr) else "MISMATCH vs oracle",
           "roofline": roofline_of(ya, t, K, Rl, G, traffic_key if cx.world == 1 else None,
                                   "rank 0's launches; input %.2f GB per GPU%s" % (8 * Il / 1e9,
                                   ", outside the 256 MiB Infinity Cache" if 8 * Il > (256 << 20) else
                                   ": fits the 256 MiB Infinity Cache, the passes after the first re-read it from there"))}
    out["roofline"]["finish_compact_kernel_ms"] = (phases or {}).get("compact_ms")
    return out, keep


def small_batches_block(cx, jitter=0, chimeras=0):
    """configs[1]: rounds 1-2's headline.  Batches of 100 k reads pipelined over `--engines` engines on one GPU
    from one host thread (yacrd_engine_submit_device / _wait): the plan / follow-on kernels, the
    launch gaps and the host's turn of one batch hide behind the sweep of another.  Weak scaling when N > 1 (every
    rank its own batch)."""
    ya, host, torch, args = cx.ya, cx.host, cx.torch, cx.args
    profile, R, O, cov, nc, seed = CONFIGS[1]
    if args.weak:
        R, O = args.reads or R, args.overlaps or O
        if args.coverage is not None:
            cov = args.coverage
    offsets, intervals, lengths = host.synth_csr(cx.prof(profile), R, O, seed + 1000 * cx.rank, flags=cx.sflags(jitter, chimeras))
    I = int(offsets[-1])
    d_off = torch.from_numpy(offsets.view(np.int64)).to(cx.dev)
    d_iv = torch.from_numpy(intervals.view(np.int32)).to(cx.dev)
    d_len = torch.from_numpy(lengths.view(np.int32)).to(cx.dev)
    torch.cuda.synchronize()
    K, W = args.small_steps, args.small_warmup
    flags = args.flags | (0 if (args.time_every_launch or K <= 64) else ya.F_TIMING_SAMPLED)
    NE = max(1, min(args.engines, K))
    engs = [ya.Engine(device_id=cx.dev_index, flags=flags) for _ in range(NE)]
    ptrs = (d_off.data_ptr(), d_iv.data_ptr(), d_len.data_ptr(), R, I, cov, nc)

    def run_steps(k):
        if NE == 1:
            res = None
            for _ in range(k):
                res = engs[0].run_device(*ptrs)
            return res
        return ya.run_device_batches(engs, [ptrs] * k)  # batch i on engine i mod NE, NE batches in flight

    run_steps(max(W, 2 * NE))  # (a submit only pipelines once the engine has a prediction)
    # The K steps take ~12 ms and their cadence is the HOST's (a submit / wait per 25 us batch): a region that starts just after a
    # phase that used every CPU of the cgroup — the generator, the oracle's check of the block before, the parsers — can fall
    # wholly into a CFS throttling window, and did: the same block measured 25 us per batch or 57 / 126 / 290 with identical
    # paths and kernel times (profiles/r06/t_*, x_*, y_*).  As an EXTRA block the region is therefore run
    # three times after a short sleep and the fastest counts (all three are kept); as the --weak HEADLINE it is K steps once.
    reps = 1 if args.weak else 3
    tries = []
    for rep in range(reps):
        if reps > 1:
            time.sleep(0.15)
        for e in engs:
            e.timing_total(reset=True)
        cx.barrier()
        t0 = time.perf_counter()
        out = run_steps(K)
        cx.barrier()
        tries.append(time.perf_counter() - t0)
    mine = min(tries)
    elapsed = cx.ydist.max_over_ranks(cx.dist, mine, cx.dev)
    per_rank = {"rank": cx.rank, "reads": R, "intervals": I, "ms_per_step": mine / K * 1e3}
    if cx.dist is not None:
        allr = [None] * cx.world
        cx.dist.all_gather_object(allr, per_rank)
    else:
        allr = [per_rank]
    t, n_timed = None, 0
    paths = {k: 0 for k in PATH_KEYS}
    for e in engs:
        te, ne = e.timing_total()
        n_timed += ne
        for k in PATH_KEYS:
            paths[k] += int(te.get(k, 0))
        if t is None:
            t = te
        else:
            for k2, v in te.items():
                if k2.endswith("_ms") or k2 in ("timed_runs", "screened"):
                    t[k2] = [a + b for a, b in zip(t[k2], v)] if isinstance(v, list) else t[k2] + v
    assert n_timed == K
    G = int(out.n_regions)
    blk = None
    if cx.rank == 0:
        # one engine, no class-count prediction (the plan's counts come home before the sweeps are launched),
        # nothing in flight: what a caller with ONE batch per process sees once the inputs are in HBM (the CLI)
        with ya.Engine(device_id=cx.dev_index, flags=ya.F_NO_PREDICTION | ya.F_NO_TIMING) as ue:
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
        # ... and the same as ONE launch (YACRD_F_ONE_LAUNCH: csrc/one_batch.h — no plan, no class counts, one dispatch)
        with ya.Engine(device_id=cx.dev_index, flags=ya.F_ONE_LAUNCH | ya.F_NO_TIMING) as oe:
            for _ in range(5):
                oe.run_device(*ptrs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                oe.run_device(*ptrs)
            dt1 = (time.perf_counter() - t0) / 50
            ot = oe.timing()
            got1 = oe.fetch()
            one_launch = {"ms_per_batch": dt1 * 1e3, "reads_per_sec": R / dt1, "ran_as_one_launch": bool(ot.get("one_launch")),
                          "deferred_reads": int(ot.get("deferred_reads", 0)),
                          "what": "YACRD_F_ONE_LAUNCH: one engine, one batch at a time; one_batch_kernel (one wavefront per eight "
                                  "reads: screen + sorts; the last to arrive at a 128-read slab scans, compacts, classifies it)"}
        with ya.Engine(device_id=cx.dev_index, flags=ya.F_TIMING_FULL) as fe:
            for _ in range(5):
                fe.run_device(*ptrs)
            fe.timing_total(reset=True)
            for _ in range(30):
                fe.run_device(*ptrs)
            ph, nf = fe.timing_total()
            phases = {k: ph[k] / nf for k in ("plan_ms", "sweep_small_ms", "sweep_medium_ms", "sweep_general_ms",
                                              "compact_ms", "total_ms", "fused_ms") if ph.get(k)}
        b_alg = alg_bytes(R, I, G)
        jit = ", dovetail ends reflected (sigma %d)" % jitter if jitter else ""
        if chimeras:
            jit += ", %d %% of the reads chimeras (a junction no overlap crosses) instead of 2 %%" % chimeras
        screened = int(t.get("screened", 0)) >= K  # (every batch went through the screen: the last run's counts stand for all)
        blk = {"workload": "configs[1]: synthetic %s pile-up%s, %d reads / %d PAF overlaps per GPU, -c %d -n %g; KERNELS ONLY: "
                           "inputs resident in HBM, the same batch every step, %d batches in flight per GPU (one engine "
                           "each), launch grids sized from the previous identical batch's class counts (validated at the "
                           "final sync)" % (profile.upper(), jit, R, O, cov, nc, NE),
               "reads_per_sec": cx.world * R * K / elapsed, "kernel_overlaps_per_sec": cx.world * O * K / elapsed,
               "ms_per_step": elapsed / K * 1e3, "steps": K, "warmup": W, "per_rank": allr,
               "timed_regions_ms_per_step": [x / K * 1e3 for x in tries],
               "scaling": "weak", "reads_per_gpu": R, "overlaps_per_gpu": O, "intervals_per_gpu": I, "regions_per_gpu": G,
               "whole_path_algorithmic_bytes": b_alg, "whole_path_GBps": b_alg / (elapsed / K) / 1e9,
               "whole_path_frac_of_peak": b_alg / (elapsed / K) / 1e9 / HBM_PEAK_GBS,
               "unpredicted_single_batch": unpredicted, "one_launch_single_batch": one_launch, "phases_full_timing_ms": phases,
               "healthy_reads": int(t.get("fused_reads", 0)) - int(t.get("deferred_reads", 0)) if screened else None,
               "deferred_reads": int(t.get("deferred_reads", 0)) if screened else None,
               "paths": dict(paths, runs=K),
               "batches_through_the_screen": "%d of %d (the others: the sorting build, chosen from the previous batches' deferral rate)" % (int(t.get("screened", 0)), K),
               "roofline": roofline_of(ya, t, K, R, G, None if (jitter or chimeras) else "configs[1]",
                                       "batch (82 MB) fits the 256 MiB Infinity Cache; other engines' small kernels run beside "
                                       "the timed launches")}
        blk["roofline"]["finish_compact_kernel_ms"] = (phases or {}).get("compact_ms")
        alone = (phases or {}).get("fused_ms")
        if alone and blk["roofline"]["size_class"] == "R2..H16":
            # `kernel_ms` above is the launch's own events WITH the other engines' batches in flight: a bracket then holds its
            # neighbours' work too (configs[1] at sigma = 300: 62 us against 43 alone).  The same launch with nothing beside it:
            rf = blk["roofline"]
            rf["kernel_alone_ms"] = alone
            rf["frac_alone"] = rf["algorithmic_bytes"] / (alone * 1e-3) / 1e9 / HBM_PEAK_GBS
            rf["note"] += "; kernel_alone_ms / frac_alone: the same launch by its own events on one engine, one batch at a time, nothing else in flight"
        if reps > 1:
            blk["workload"] += "; the FASTEST of %d timed regions of %d steps (the host's cadence: see timed_regions_ms_per_step)" % (reps, K)
        import oracle
        want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cov, nc, n_threads=usable_cpus())
        got = engs[0].fetch()
        blk["parity"] = ("bit-exact vs oracle on all %d reads" % R
                         if (np.array_equal(got.bad_offsets, want[0]) and np.array_equal(got.bad_regions, want[1])
                             and np.array_equal(got.read_type, want[2])) else "MISMATCH vs oracle")
        blk["one_launch_single_batch"]["parity"] = (
            "bit-exact vs oracle on all %d reads" % R
            if (np.array_equal(got1.bad_offsets, want[0]) and np.array_equal(got1.bad_regions, want[1])
                and np.array_equal(got1.read_type, want[2])) else "MISMATCH vs oracle")
    keep = (offsets, intervals, lengths, engs, G, cov, nc) if cx.rank == 0 and not jitter and not chimeras else None
    if keep is None:
        for e in engs:
            e.close()
    del d_off, d_iv, d_len
    torch.cuda.empty_cache()
    return blk, keep


def cpu_baseline(offsets, intervals, lengths, cov, nc, label):
    """The oracle on this box's cores (kind "port"), as BASELINE.md §2 specifies: the WHOLE headline workload on every usable
    CPU, best of 3 after one warm-up pass; and one thread (the reference's default is -t 1, src/main.rs:75-77) on its first
    500 000 reads.  ~25 s of CPU work at configs[4]'s size."""
    import oracle
    ncores = usable_cpus()
    R = len(lengths)
    l64 = lengths.astype(np.uint64)
    off = offsets
    iv = intervals
    times = []
    for rep in range(4):  # (the first pass is the warm-up: page faults of the result arrays, the thread pool)
        t1 = time.perf_counter()
        oracle.run(off, iv, l64, cov, nc, n_threads=ncores)
        times.append(time.perf_counter() - t1)
    cpu_all = min(times[1:])
    r1 = min(R, 500_000)
    t1 = time.perf_counter()
    oracle.run(off[: r1 + 1], iv[: int(off[r1])], l64[:r1], cov, nc, 1)
    cpu_1 = time.perf_counter() - t1
    return {"value": R / cpu_all, "unit": "reads/s", "cores": ncores, "hardware_threads": os.cpu_count(), "kind": "port",
            "sample": "ALL %d reads (%d intervals) of %s on %d threads (= usable CPUs: hardware threads capped by the cgroup "
                      "cpu.max quota), best of 3 after a warm-up pass: %.2f s (the three: %s); single thread (reference "
                      "default -t 1) on the first %d reads: %.0f reads/s"
                      % (R, int(off[-1]), label, ncores, cpu_all, ", ".join("%.2f" % x for x in times[1:]), r1, r1 / cpu_1),
            "sample_short": "all %d reads of %s on %d threads, best of 3 after a warm-up: %.2f s; 1 thread on the first %d reads"
                            % (R, label, ncores, cpu_all, r1),
            "value_1thread": r1 / cpu_1, "seconds_all_threads": times}


def pcie_inclusive(ya, engs, offsets, intervals, lengths, cov, nc, G):
    """R / (H2D + kernels + D2H): the batch leaves pinned host memory every step
    (yacrd_engine_submit / _collect, the engines' batches overlap: the H2D of one, the kernels and the
    D2H of the others), plus one blocking yacrd_engine_run for the unpipelined latency, and the same
    from PAGEABLE memory (threaded bounce-buffer staging)."""
    R, I = len(lengths), int(offsets[-1])
    pins = [ya.PinnedArray.copy_of(x) for x in (offsets, intervals, lengths)]
    arrs = tuple(p.array for p in pins)
    e0 = engs[0]
    h2d_bytes = offsets.nbytes + intervals.nbytes + lengths.nbytes
    d2h_bytes = 8 * (R + 1) + 8 * G + R
    for _ in range(2):
        e0.run(*arrs, cov, nc)
    K1 = 10
    e0.timing_total(reset=True)
    t0 = time.perf_counter()
    for _ in range(K1):
        e0.run(*arrs, cov, nc)
    blocking = (time.perf_counter() - t0) / K1
    tt, _ = e0.timing_total(reset=True)
    h2d_ms, d2h_ms = tt["h2d_ms"] / K1, tt["d2h_ms"] / K1
    NE = len(engs)
    K2 = 40

    def steps(k):
        inflight = [False] * NE
        for i in range(k):
            j = i % NE
            if inflight[j]:
                engs[j].collect()
            engs[j].submit(*arrs, cov, nc)
            inflight[j] = True
        for j in range(NE):
            if inflight[j]:
                engs[j].collect()
    steps(2 * NE)
    t0 = time.perf_counter()
    steps(K2)
    piped = (time.perf_counter() - t0) / K2
    for _ in range(2):
        e0.run(offsets, intervals, lengths, cov, nc)
    e0.timing_total(reset=True)
    t0 = time.perf_counter()
    for _ in range(5):
        e0.run(offsets, intervals, lengths, cov, nc)
    pageable = (time.perf_counter() - t0) / 5
    tp, _ = e0.timing_total(reset=True)
    for p in pins:
        p.close()
    return {"workload": "configs[1]", "reads_per_sec": R / piped, "ms_per_batch": piped * 1e3, "engines": NE, "batches": K2,
            "overlaps_per_sec": (I // 2) / piped,
            "blocking_ms_per_batch": blocking * 1e3, "blocking_reads_per_sec": R / blocking,
            "h2d_ms": h2d_ms, "h2d_GBps": h2d_bytes / (h2d_ms * 1e-3) / 1e9 if h2d_ms else None,
            "d2h_ms": d2h_ms, "h2d_bytes": h2d_bytes, "d2h_bytes": d2h_bytes,
            "pageable_ms_per_batch": pageable * 1e3,
            "pageable_h2d_GBps": h2d_bytes / (tp["h2d_ms"] / 5 * 1e-3) / 1e9 if tp["h2d_ms"] else None,
            "source": "pinned host memory (yacrd_pinned_alloc), the same batch every step; PCIe Gen5 x16 spec 63 GB/s"}


def end_to_end(ya, host, eng, prof, R, O, cov, nc):
    """BASELINE's "overlaps/sec ingested -> reads/sec classified" as ONE number: the full
    configs[1] PAF text (not a sample) to read types.  Parse threads fill pinned buffers, every
    full buffer crosses PCIe at once (yacrd_stream_*), the CSR is built on the GPU, the engine
    runs, results come home.  Also: the parse alone into a host CSR, by thread count."""
    ncores = usable_cpus()
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    paf = os.path.join(d, "yacrd_bench_%d_%d_%d.paf" % (os.getpid(), R, O))
    hl = host.load_library()
    el = ya.load_library()
    try:
        t0 = time.perf_counter()
        host.synth_paf(prof, R, O, 20241110, paf)
        gen_s = time.perf_counter() - t0
        size = os.path.getsize(paf)
        out = {"workload": "configs[1] as PAF text", "paf_bytes": size, "overlaps": O, "reads": R, "generate_s": gen_s,
               "file_on": d}
        best = None
        with ya.Stream(eng) as st:
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
                    res = ya.engine._Result()
                    rc = el.yacrd_stream_finish(st._h, mp, nh, v.lengths, v.n_reads, cov, nc, ctypes.byref(res))
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
        host_path = {"overlaps_per_sec": O / best["s"], "reads_per_sec": best["reads_found"] / best["s"],
                     "text_GBps": size / best["s"] / 1e9, "seconds": best["s"], "stream": best,
                     "path": "PAF text -> yacrd_ingest_stream (host parser: pread blocks, shared id table) -> pinned buffers "
                             "-> hipMemcpyAsync during the parse -> CSR build on the GPU -> engine -> D2H"}
        # the same file with the PARSE on the GPU (yacrd_engine_ingest_paf): the host only moves the text
        dbest = None
        for th in (4, 8):
            for rep in range(3):
                res, rd, stt = ya.engine._Result(), ya.engine._Reads(), ya.engine._IngestStats()
                t0 = time.perf_counter()
                rc = el.yacrd_engine_ingest_paf(eng._h, paf.encode(), th, cov, nc, ctypes.byref(res), ctypes.byref(rd), ctypes.byref(stt))
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError("device parser: %d %s" % (rc, el.yacrd_last_error().decode()))
                nr, ng = int(rd.n_reads), int(res.n_regions)
                el.yacrd_result_free(ctypes.byref(res))
                el.yacrd_reads_free(ctypes.byref(rd))
                if dbest is None or dt < dbest["s"]:
                    dbest = {"s": dt, "threads": th, "reads_found": nr, "regions": ng,
                             **{k: getattr(stt, k) for k in ("text_ms", "parse_ms", "build_ms", "run_ms", "d2h_ms")}}
        dev_path = {"overlaps_per_sec": O / dbest["s"], "reads_per_sec": dbest["reads_found"] / dbest["s"],
                    "text_GBps": size / dbest["s"] / 1e9, "seconds": dbest["s"], "phases": dbest,
                    "same_result_as_host_parser": dbest["reads_found"] == best["reads_found"] and dbest["regions"] == best["regions"],
                    "path": "PAF text -> pread chunks -> pinned buffers -> hipMemcpyAsync -> a mirror of the file in HBM -> "
                            "scan / parse / id table / first-appearance numbering / CSR build on the GPU -> engine -> D2H "
                            "(yacrd_engine_ingest_paf; the drop-in CLI takes this path for plain PAF files)"}
        top = dev_path if dev_path["overlaps_per_sec"] >= host_path["overlaps_per_sec"] else host_path
        out.update({k: top[k] for k in ("overlaps_per_sec", "reads_per_sec", "text_GBps", "seconds", "path")})
        out["device_parser"] = dev_path
        out["host_parser"] = host_path
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
        os.remove(paf)
        out["at_scale"] = guarded(end_to_end_at_scale, ya, host, eng, d, max(1000, int(600_000 * SCALE)), max(10000, int(60_000_000 * SCALE)))
        return out
    finally:
        if os.path.exists(paf):
            os.remove(paf)


def end_to_end_at_scale(ya, host, eng, d, R=600_000, O=60_000_000):
    """The same two routes on 4.5 GB of text (configs[2]'s profile at 600 k reads / 60 M overlaps: byte offsets pass
    2^32; 367 MB is over before the copy threads are up to speed).  tools/e2e_large.py runs configs[2] (15 GB) and the
    north star's configs[4] (37 GB) the same way: profiles/r03_e2e_large.log."""
    st = os.statvfs(d)
    need = 80 * O  # ~75 bytes per line
    if st.f_bavail * st.f_frsize < 2 * need:
        return {"skipped": "less than %.0f GB free on %s" % (2 * need / 1e9, d)}
    paf = os.path.join(d, "yacrd_bench_scale_%d.paf" % os.getpid())
    el = ya.load_library()
    try:
        host.synth_paf(host.SYNTH_SEQUEL, R, O, 20250306, paf)
        size = os.path.getsize(paf)
        best = None
        for rep in range(3):
            # (the generator and the host-parser runs before this block burn the box's cgroup CPU quota — 16 CPUs of
            # 256 hardware threads — and a call right behind them finds its copy threads throttled: 0.12 s becomes
            # 0.5 s; profiles/r03_e2e_large.log)
            time.sleep(2.0 if rep == 0 else 1.0)
            res, rd, stt = ya.engine._Result(), ya.engine._Reads(), ya.engine._IngestStats()
            t0 = time.perf_counter()
            rc = el.yacrd_engine_ingest_paf(eng._h, paf.encode(), 6, 3, 0.4, ctypes.byref(res), ctypes.byref(rd), ctypes.byref(stt))
            dt = time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError("device parser: %d %s" % (rc, el.yacrd_last_error().decode()))
            sig = (int(rd.n_reads), int(res.n_regions), int(np.ctypeslib.as_array(res.read_type, shape=(int(res.n_reads),)).sum()))
            el.yacrd_result_free(ctypes.byref(res))
            el.yacrd_reads_free(ctypes.byref(rd))
            if best is None or dt < best["seconds"]:
                best = {"seconds": dt, **{k: getattr(stt, k) for k in ("text_ms", "parse_ms", "build_ms", "run_ms", "d2h_ms")}}
        # the N-GPU form of the device parser with its N engines on THIS device (no second GPU on the box): two engines
        # share the one link and the one GPU, so "no slower than one" is what there is to see; on N GPUs each has a link
        two = None
        with ya.Engine(device_id=eng.device_id) as e2:
            handles = (ctypes.c_void_p * 2)(eng._h, e2._h)
            for rep in range(3):
                time.sleep(1.0)
                res, rd, stt = ya.engine._Result(), ya.engine._Reads(), ya.engine._IngestStats()
                t0 = time.perf_counter()
                rc = el.yacrd_engines_ingest_overlaps(handles, 2, paf.encode(), 1, 6, 3, 0.4, ctypes.byref(res), ctypes.byref(rd), ctypes.byref(stt))
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError("device parser, two engines: %d %s" % (rc, el.yacrd_last_error().decode()))
                g_sig = (int(rd.n_reads), int(res.n_regions), int(np.ctypeslib.as_array(res.read_type, shape=(int(res.n_reads),)).sum()))
                el.yacrd_result_free(ctypes.byref(res))
                el.yacrd_reads_free(ctypes.byref(rd))
                if two is None or dt < two["seconds"]:
                    two = {"seconds": dt, "overlaps_per_sec": O / dt, "same_as_one_engine": g_sig == sig,
                           **{k: getattr(stt, k) for k in ("text_ms", "parse_ms", "build_ms", "run_ms", "d2h_ms")}}
            e2.trim()
        with ya.StreamGroup([eng]) as grp:
            t0 = time.perf_counter()
            c = host.ingest_stream(paf, grp.sink(), n_threads=0)
            ref = grp.finish(c.handle_map, c.lengths, 3, 0.4)
            dt_host = time.perf_counter() - t0
        same = sig == (len(c.lengths), int(ref.bad_offsets[-1]), int(ref.read_type.sum()))
        return {"workload": "SEQUEL profile, %d reads / %d overlaps as PAF text" % (R, O), "paf_bytes": size,
                "device_parser": {"overlaps_per_sec": O / best["seconds"], "reads_per_sec": R / best["seconds"],
                                  "text_GBps": size / best["seconds"] / 1e9, **best},
                "device_parser_two_engines_on_this_device": two,
                "host_parser_streamed": {"overlaps_per_sec": O / dt_host, "seconds": dt_host},
                "same_reads_regions_types": same}
    finally:
        if os.path.exists(paf):
            os.remove(paf)


def box_block(cx):
    """What kind of box this is, in three numbers a reader can normalise by (the same build measured
    configs[1]'s small batches 30-60 % slower on some boxes of the pool than on others, with the same paths taken —
    `paths` / `reruns_total` say so — while the HBM-bound headline moved by 3 %): a device-to-device copy of 64 MB (fits
    the Infinity Cache: what the small batches run from), of 2 GB (HBM), both in GB/s of bytes read + written, and the
    time per dependent launch of a one-element kernel."""
    torch = cx.torch
    out = {"device": torch.cuda.get_device_name(cx.dev_index)}
    try:
        def copy_rate(nbytes, reps):
            a = torch.empty(nbytes // 4, dtype=torch.int32, device=cx.dev)
            b = torch.empty_like(a)
            a.zero_()
            for _ in range(3):
                b.copy_(a)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                b.copy_(a)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            del a, b
            return 2 * nbytes / dt / 1e9
        out["copy_64MB_GBps"] = copy_rate(64 << 20, 200)
        out["copy_2GB_GBps"] = copy_rate(2 << 30, 10)
        x = torch.zeros(1, dtype=torch.int32, device=cx.dev)
        for _ in range(20):
            x.add_(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            x.add_(1)
        torch.cuda.synchronize()
        out["dependent_launch_us"] = (time.perf_counter() - t0) / 500 * 1e6
        torch.cuda.empty_cache()
    except Exception as ex:
        out["error"] = repr(ex)
    return out


SCALE = 1.0  # --scale (plumbing tests)


def guarded(fn, *a, **k):
    try:
        return fn(*a, **k)
    except Exception as ex:  # the headline must survive a failure in an extra block
        return {"error": repr(ex)}


def no_nan(x):
    """NaN / Infinity -> None, numpy scalars -> Python's: the line must survive a strict json.loads."""
    if isinstance(x, dict):
        return {str(k): no_nan(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [no_nan(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, np.bool_):
        return bool(x)
    return x


def _dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


COMPACT_LIMIT = 4096  # bytes: the driver parses the LAST stdout line; round 4's 30 KB line came back `parsed: null`


def compact_line(full, extras_path):
    """The line the driver parses: the contract's keys, `roofline`, `cpu_baseline` and a handful of scalars taken
    from the extra blocks — everything else is in bench_extras.json (`--print-extras` prints it on an earlier line)."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: full.get(k) for k in keys}
    out["rccl_ranks"] = full.get("rccl_ranks")
    pr = _dig(full, "headline", "per_rank")
    if isinstance(pr, list) and len(pr) > 1:  # (one number per rank: what the max over ranks was taken over)
        out["per_rank_ms"] = [_r(p.get("ms_per_step")) for p in pr]
    cfg = dict(full.get("config") or {})
    cfg["workload"] = full.get("workload_short") or cfg.get("workload", "")[:300]
    out["config"] = cfg
    out["parity"] = full.get("parity")
    rf = full.get("roofline") or {}
    out["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                  "algorithmic_bytes", "kernel_ms", "timed_launches", "kernel_reads",
                                                  "deferred_reads")}
    out["roofline"]["traffic_source"] = rf.get("traffic_source_short")
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = ({k: _r(cb.get(k), 1) for k in ("value", "unit", "cores", "kind", "value_1thread")}
                               if "error" not in cb else cb)
        if "error" not in cb:
            out["cpu_baseline"]["sample"] = cb.get("sample_short") or str(cb.get("sample"))[:160]
    sc = {"whole_path_frac_of_peak": _dig(full, "headline", "whole_path_frac_of_peak"),
          "kernel_overlaps_per_sec": full.get("kernel_overlaps_per_sec"),
          "follow_on_ms": _dig(full, "roofline", "finish_compact_kernel_ms"),
          "configs2_ms": _dig(full, "configs2", "ms_per_step"),
          "configs2_frac": _dig(full, "configs2", "roofline", "frac"),
          "skewed_ms": _dig(full, "skewed", "ms_per_step"),
          "skewed_frac": _dig(full, "skewed", "roofline", "frac"),
          "configs1_pipelined_us": None, "configs1_one_at_a_time_us": None, "one_launch_us": None,
          "e2e_overlaps_per_sec": _dig(full, "end_to_end", "overlaps_per_sec"),
          "e2e_at_scale_overlaps_per_sec": _dig(full, "end_to_end", "at_scale", "device_parser", "overlaps_per_sec"),
          "e2e_at_scale_two_engines_overlaps_per_sec": _dig(full, "end_to_end", "at_scale", "device_parser_two_engines_on_this_device", "overlaps_per_sec"),
          "pcie_inclusive_reads_per_sec": _dig(full, "pcie_inclusive", "reads_per_sec")}
    sb = full.get("small_batches") or (full.get("headline") if full.get("scaling") == "weak" else None)
    if isinstance(sb, dict):
        for k, path in (("configs1_pipelined_us", ("ms_per_step",)),
                        ("configs1_one_at_a_time_us", ("unpredicted_single_batch", "ms_per_batch")),
                        ("one_launch_us", ("one_launch_single_batch", "ms_per_batch"))):
            v = _dig(sb, *path)
            sc[k] = v * 1e3 if isinstance(v, (int, float)) else None
    out.update({k: _r(v) for k, v in sc.items()})
    # runs that were run AGAIN (a prediction that did not hold, the persistent workgroup screen giving up), over every timed block
    blocks = [full.get("headline"), full.get("configs2"), full.get("skewed"), full.get("small_batches"), full.get("configs4_sigma100")]
    blocks += [b for k, b in (full.get("jitter") or {}).items() if k != "healthy_share_of_screened_reads"]
    pp = [b["paths"] for b in blocks if isinstance(b, dict) and isinstance(b.get("paths"), dict)]
    bx = full.get("box")
    if isinstance(bx, dict) and "error" not in bx:  # [64 MB copy GB/s (Infinity Cache), 2 GB copy GB/s (HBM), us per dependent launch]
        out["box"] = [_r(bx.get("copy_64MB_GBps"), 0), _r(bx.get("copy_2GB_GBps"), 0), _r(bx.get("dependent_launch_us"), 2)]
    out["reruns_total"] = sum(p.get("prediction_misses", 0) + p.get("fused_reruns", 0) for p in pp) if pp else None
    if isinstance(full.get("value_sigma100"), dict):
        out["value_sigma100"] = {k: _r(v) for k, v in full["value_sigma100"].items()}
    jit = full.get("jitter")
    if isinstance(jit, dict):  # [ms per step, frac, share of the screened reads decided] per (config, sigma)
        share = jit.get("healthy_share_of_screened_reads") or {}
        def path3(b):  # [runs run again, runs in the build with the second looks, build switches] of the block's timed runs
            p = b.get("paths")
            return [] if not isinstance(p, dict) else [p.get("prediction_misses", 0) + p.get("fused_reruns", 0), p.get("screen_wide", 0),
                                                       p.get("build_switches", 0)]
        # (+ for the pipelined configs[1] blocks, [6]: the dominant kernel's frac with nothing else in flight)
        out["jitter"] = {k: [_r(_dig(b, "ms_per_step")), _r(_dig(b, "roofline", "frac")), _r(share.get(k))] + path3(b)
                            + ([_r(_dig(b, "roofline", "frac_alone"))] if _dig(b, "roofline", "frac_alone") is not None else [])
                         for k, b in jit.items() if isinstance(b, dict) and k != "healthy_share_of_screened_reads"}
    out["extras"] = extras_path if extras_path.startswith("not written") else os.path.basename(extras_path)
    s = json.dumps(out, allow_nan=False, separators=(",", ":"))
    # never expected (every string above is bounded), but the limit is enforced, not hoped for: optional blocks go first,
    # then the scalars, then the strings are cut — the contract's keys, `roofline` and `cpu_baseline` stay
    droppable = ["jitter", "value_sigma100", "per_rank_ms", "reruns_total", "box"] + [k for k in sc] + ["extras"]
    for drop in droppable:
        if len(s.encode()) < COMPACT_LIMIT:
            break
        out.pop(drop, None)
        s = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(s.encode()) >= COMPACT_LIMIT:
        out["config"] = {"workload": str(out["config"].get("workload", ""))[:200]}
        out["parity"] = str(out.get("parity"))[:100]
        if isinstance(out.get("cpu_baseline"), dict):
            out["cpu_baseline"] = {k: (v[:100] if isinstance(v, str) else v) for k, v in out["cpu_baseline"].items()}
        out["roofline"] = {k: (v[:100] if isinstance(v, str) else v) for k, v in out["roofline"].items()}
    return out


def launch_ranks(args):
    """`--gpus N` is the number of ranks, whoever starts them (it used to be parsed and ignored — a plain
    `python bench.py --gpus 8` ran ONE rank and printed "n_gpus": 1).  Under a launcher (WORLD_SIZE set) the two must
    agree, or the run stops before it measures anything; without one and N > 1 this process becomes
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py
    <same arguments>` — one rank per GPU (LOCAL_RANK -> device), the read partition of SURVEY.md 8(e)
    (reference: src/stack.rs:151-156 maps compute_bad_part over independent reads)."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if args.gpus is not None and args.gpus != int(env_world):
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s: the launcher's rank count and --gpus must agree\n"
                             % (args.gpus, env_world))
            sys.exit(2)
        args.gpus = int(env_world)
        return
    if args.gpus is None:
        args.gpus = 1
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if args.gpus == 1:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)  # (the ranks inherit stdout: rank 0's compact line stays the last line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (one process per GPU).  Without WORLD_SIZE in the environment and N > 1, "
                         "bench.py launches itself under torch.distributed.run with N ranks; under a launcher, N must equal "
                         "WORLD_SIZE (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[k] as the headline's fixed input (default 4: the north star's 5 M reads / "
                         "500 M overlaps)")
    ap.add_argument("--weak", action="store_true",
                    help="headline = rounds 1-2's: every rank its own configs[1]-sized batch, pipelined over --engines engines")
    ap.add_argument("--jitter", type=int, default=0,
                    help="headline input from the generator that reflects the dovetail ends' offsets into the read "
                         "(YACRD_SYNTH_F_JITTER), sigma = this many positions (SURVEY.md 8d's is 30)")
    ap.add_argument("--chimeras", type=int, default=0,
                    help="with --weak: per cent of the reads that are chimeras (YACRD_SYNTH_F_CHIMERA_PCT; SURVEY.md 8d's is 2)")
    ap.add_argument("--reads", type=int, default=0, help="override the headline's (or, with --weak, the batch's) read count")
    ap.add_argument("--overlaps", type=int, default=0)
    ap.add_argument("--coverage", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="extra YACRD_F_* engine flags (A/B)")
    ap.add_argument("--resident-engines", type=int, default=1,
                    help="engines (HIP streams) per GPU the resident blocks' passes are pipelined over (1: one pass at a time)")
    ap.add_argument("--engines", type=int, default=3, help="engines (HIP streams) the small batches are pipelined over")
    ap.add_argument("--small-steps", type=int, default=500)
    ap.add_argument("--small-warmup", type=int, default=10)
    ap.add_argument("--time-every-launch", action="store_true",
                    help="small batches: events on the dominant kernel of EVERY step (default: every 8th step of an engine "
                         "when there are more than 64 steps: the events cost ~10 us per step against a 20 us kernel)")
    ap.add_argument("--no-extras", action="store_true", help="headline only")
    ap.add_argument("--extras-file", type=str, default="", help="where the full object goes (default: bench_extras.json beside bench.py)")
    ap.add_argument("--print-extras", action="store_true", help="also print the full object, on a line BEFORE the compact one")
    ap.add_argument("--sub-steps", type=int, default=10, help="timed steps of the configs[2] / configs[3] sub-blocks")
    ap.add_argument("--sigma100-steps", type=int, default=5,
                    help="timed steps of the `value_sigma100` block (the headline workload from the jittered generator; 0 = skip)")
    ap.add_argument("--jitter-sigmas", type=str, default="30,100,300",
                    help="sigmas of the `jitter` block (comma separated; empty = no jitter block)")
    ap.add_argument("--scale", type=float, default=1.0,
                    help="scale every config's reads and overlaps (plumbing tests only: a scaled run is not a measurement, "
                         "and the workload strings say so)")
    args = ap.parse_args()
    launch_ranks(args)
    global SCALE
    SCALE = args.scale
    if args.scale != 1.0:
        for k, (pf, R, O, c, n, sd) in list(CONFIGS.items()):
            CONFIGS[k] = (pf, max(64, int(R * args.scale)), max(640, int(O * args.scale)), c, n, sd)
    cx = Ctx(args)
    ya, host = cx.ya, cx.host
    rank, world = cx.rank, cx.world
    extras = world == 1 and not args.no_extras
    backend = cx.dist.get_backend() if cx.dist is not None else None

    line, head, keep_small, keep_head = None, None, None, None
    if args.weak:
        if args.steps != ap.get_default("steps"):
            args.small_steps = args.steps
        if args.warmup != ap.get_default("warmup"):
            args.small_warmup = args.warmup
        head, keep_small = small_batches_block(cx, args.jitter, args.chimeras)
        scaling = "weak"
    elif args.config == 1:
        ap.error("configs[1] is the --weak headline (batches of 100 k reads); --config takes 2, 3 or 4")
    else:
        profile, R, O, cov, nc, seed = CONFIGS[args.config]
        R, O = args.reads or R, args.overlaps or O
        if args.coverage is not None:
            cov = args.coverage
        label = "configs[%d]" % args.config + (" (the north star's target workload)" if args.config == 4 else "")
        head, keep_head = resident_block(cx, label, profile, R, O, cov, nc, seed, args.jitter, args.steps,
                                         args.warmup, 50000 if args.config == 4 else 20000,
                                         traffic_key=None if args.jitter else "configs[%d]" % args.config,
                                         keep_host=world == 1 and not args.no_cpu_baseline)
        scaling = "strong"
    if rank == 0:
        line = {
            "metric": "reads_per_sec_classified",
            "value": head["reads_per_sec"],
            "unit": "reads/s",
            "n_gpus": world,
            "rccl_ranks": (cx.dist.get_world_size() if backend == "nccl" else 0),  # ranks in the nccl (= RCCL) group; 0: none
            "steps": head["steps"],
            "warmup": head["warmup"],
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": ("SCALED x%g (not a measurement): " % args.scale if args.scale != 1.0 else "") + head["workload"] + "; see pcie_inclusive / end_to_end for the rates that include PCIe and the parse",
                       "parallelism": "read-partition x%d, no collective" % world,
                       "torch_distributed_backend": backend if backend else "none (one process)"},
            "workload_short": ("SCALED x%g (not a measurement): " % args.scale if args.scale != 1.0 else "") + head["workload"].split(", read-partitioned")[0].split("; KERNELS ONLY")[0]
                              + "; kernels only, inputs resident in HBM",
            "kernel_overlaps_per_sec": head["kernel_overlaps_per_sec"],
            "parity": head["parity"],
            "roofline": head["roofline"],
            "headline": {k: v for k, v in head.items() if k not in ("roofline", "workload")},
        }
        if keep_head is not None:
            offsets, intervals, lengths = keep_head
            cfgk = CONFIGS[args.config]
            line["cpu_baseline"] = guarded(cpu_baseline, offsets, intervals, lengths,
                                           args.coverage if args.coverage is not None else cfgk[3], cfgk[4],
                                           "configs[%d]" % args.config)
            keep_head = None
            del offsets, intervals, lengths
    if extras:
        line["box"] = box_block(cx)
        if not args.weak:
            for key, k in (("configs2", 2), ("skewed", 3)):
                if k == args.config:
                    continue
                pk = CONFIGS[k]
                try:
                    line[key], _ = resident_block(cx, "configs[%d]" % k, pk[0], pk[1], pk[2], pk[3], pk[4], pk[5], 0,
                                                  args.sub_steps, 2, 20000, traffic_key="configs[%d]" % k)
                except Exception as ex:
                    line[key] = {"error": repr(ex)}
            if args.config == 4 and not args.jitter and args.sigma100_steps > 0:
                # the headline's input is SURVEY 8d's generator, which CLAMPS the dovetail ends onto 0 / len — the screen's best
                # case.  The same workload with the ends spread (sigma = 100 positions), beside it:
                pk = CONFIGS[4]
                try:
                    b, _ = resident_block(cx, "configs[4]", pk[0], args.reads or pk[1], args.overlaps or pk[2], pk[3], pk[4], pk[5],
                                          100, args.sigma100_steps, 2, 20000)
                    h, df = b.get("healthy_reads_rank0"), b.get("deferred_reads_rank0")
                    line["value_sigma100"] = {"value": b["reads_per_sec"], "ms_per_step": b["ms_per_step"], "steps": b["steps"],
                                              "frac": b["roofline"]["frac"], "kernel": b["roofline"]["kernel"],
                                              "whole_path_frac_of_peak": b["whole_path_frac_of_peak"],
                                              "decided_share": None if h is None or df is None else h / max(1, h + df),
                                              "parity": b["parity"]}
                    line["configs4_sigma100"] = b
                except Exception as ex:
                    line["value_sigma100"] = {"error": repr(ex)}
            line["small_batches"], keep_small = small_batches_block(cx)
        # (the blocks that re-use the small batches' engines come FIRST and give them back: every live engine is a stream, and
        #  the jitter blocks' three should not share the hardware queues with three idle ones — see GPU_MAX_HW_QUEUES above)
        if keep_small is not None:
            offsets, intervals, lengths, engs, G, cov1, nc1 = keep_small
            c1 = CONFIGS[1]
            line["pcie_inclusive"] = guarded(pcie_inclusive, ya, engs, offsets, intervals, lengths, cov1, nc1, G)
            line["end_to_end"] = guarded(end_to_end, ya, host, engs[0], cx.prof(c1[0]), len(lengths), int(offsets[-1]) // 2, cov1, nc1)
            if args.weak and not args.no_cpu_baseline:
                line["cpu_baseline"] = guarded(cpu_baseline, offsets, intervals, lengths, cov1, nc1, "configs[1]")
            for e in engs:
                e.close()
            keep_small = None
        sigmas = [int(x) for x in args.jitter_sigmas.split(",") if x.strip()]
        if sigmas:
            jit = {}
            p2 = CONFIGS[2]
            for sg in sigmas:
                key = "" if sg == 30 else "_sigma%d" % sg
                b1, _ = small_batches_block(cx, jitter=sg)
                jit["configs[1]" + key] = b1
                try:
                    b2, _ = resident_block(cx, "configs[2]", p2[0], p2[1], p2[2], p2[3], p2[4], p2[5], sg, 5, 2, 20000)
                except Exception as ex:
                    b2 = {"error": repr(ex)}
                jit["configs[2]" + key] = b2
            # what the windows of the screen are there for, in one table: the share of the screened reads it decides
            def share(b):
                if not isinstance(b, dict) or "error" in b:
                    return None
                h = b.get("healthy_reads", b.get("healthy_reads_rank0"))
                df = b.get("deferred_reads", b.get("deferred_reads_rank0"))
                return None if h is None or df is None else h / max(1, h + df)
            jit["healthy_share_of_screened_reads"] = {k: share(b) for k, b in list(jit.items())}
            line["jitter"] = jit
        # more of the reads yacrd looks for: what the screen defers grows with them, and from a quarter on the engine
        # takes the sorting build (engine.hip: nodefer_left)
        bad = {}
        for pct in (10, 40):
            b, _ = small_batches_block(cx, chimeras=pct)
            if b is not None:
                bad["%d%%" % pct] = {k: b[k] for k in ("workload", "reads_per_sec", "ms_per_step", "healthy_reads", "deferred_reads",
                                                       "batches_through_the_screen", "unpredicted_single_batch", "parity")}
        line["more_bad_reads"] = bad
    if keep_small is not None:
        for e in keep_small[3]:
            e.close()
    if rank == 0:
        full = no_nan(line)
        extras_path = args.extras_file or os.path.join(ROOT, "bench_extras.json")
        try:
            with open(extras_path, "w") as fh:
                json.dump(full, fh, allow_nan=False)
        except OSError as ex:  # (a read-only checkout: the compact line still goes out)
            extras_path = "not written: %r" % ex
        if args.print_extras:  # the whole object on an EARLIER line; the LAST line stays the compact one
            print(json.dumps(full, allow_nan=False), flush=True)
        print(json.dumps(compact_line(full, extras_path), allow_nan=False, separators=(",", ":")), flush=True)
    if cx.dist is not None:
        cx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
