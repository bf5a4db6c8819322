"""ctypes binding of include/yacrd_engine.h (libyacrd_hip.so).  No CPU fallback."""
import ctypes
import os
import subprocess
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libyacrd_hip.so")

NOT_BAD, CHIMERIC, NOT_COVERED = 0, 1, 2
TYPE_NAMES = {NOT_BAD: "NotBad", CHIMERIC: "Chimeric", NOT_COVERED: "NotCovered"}
# yacrd_engine_cfg.flags: the timing flags are part of include/yacrd_engine.h, everything else is an A/B or
# test switch from include/yacrd_engine_debug.h
F_FORCE_GENERAL = 1
F_FORCE_LDS_SORT = 2
F_XLANE_DS = 4
F_WAVE_ONLY = 8
F_NO_HALVES = 16
F_TIMING_FULL = 32
F_NO_PREDICTION = 64
F_NO_FUSED_LAUNCH = 128
F_NO_PREFILTER = 256
F_COUNT_PREFILTERED = 512
F_NO_TIMING = 1024
F_BLOCKING_WAIT = 2048
F_NO_DEFER = 4096
F_ALWAYS_DEFER = 8192
F_SWEEP_TURNS = 16384
F_TIMING_SAMPLED = 32768
F_STREAM_SCREEN = 65536
F_SCREEN_ITEMS_1 = 262144
F_SCREEN_ITEMS_2 = 524288
F_NO_FUSED_SCREEN = 1048576
F_SCREEN_WIDE = 2097152
F_ONE_LAUNCH = 4194304  # include/yacrd_engine.h: a short batch as ONE kernel launch (csrc/one_batch.h)

# every symbol include/yacrd_engine.h declares
EXPORTED_SYMBOLS = [
    "yacrd_abi_version", "yacrd_last_error", "yacrd_engine_create", "yacrd_engine_destroy",
    "yacrd_engine_run", "yacrd_result_free", "yacrd_engine_run_device", "yacrd_engine_fetch",
    "yacrd_engine_last_timing", "yacrd_partition_reads", "yacrd_engine_classify",
    "yacrd_engines_run_partitioned", "yacrd_engine_timing_total", "yacrd_engine_event_overhead", "yacrd_engine_submit_device", "yacrd_engine_wait", "yacrd_engines_run_device_batches",
    "yacrd_engine_submit", "yacrd_engine_collect", "yacrd_pinned_alloc", "yacrd_pinned_free",
    "yacrd_stream_open", "yacrd_stream_sink", "yacrd_stream_acquire", "yacrd_stream_commit",
    "yacrd_stream_finish", "yacrd_stream_last_stats", "yacrd_stream_reset", "yacrd_stream_close",
    "yacrd_engine_ingest_paf", "yacrd_engine_ingest_overlaps", "yacrd_engine_ingest_overlaps_mem", "yacrd_engines_ingest_overlaps",
    "yacrd_engines_ingest_overlaps_mem", "yacrd_reads_free", "yacrd_engine_trim",
    "yacrd_stream_device_of", "yacrd_stream_group_open", "yacrd_stream_group_sink", "yacrd_stream_group_finish",
    "yacrd_stream_group_last_stats", "yacrd_stream_group_reset", "yacrd_stream_group_close",
]


class EngineError(RuntimeError):
    pass


class NeedsHostParser(EngineError):
    """yacrd_engine_ingest_paf returned YACRD_EFALLBACK: the input is for the host parser."""


E_FALLBACK = 5


class _Reads(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_uint64), ("n_records", ctypes.c_uint64),
                ("lengths", ctypes.POINTER(ctypes.c_uint32)), ("name_off", ctypes.POINTER(ctypes.c_uint64)),
                ("names", ctypes.POINTER(ctypes.c_char))]


class _IngestStats(ctypes.Structure):
    _fields_ = [("text_bytes", ctypes.c_uint64), ("n_records", ctypes.c_uint64), ("n_reads", ctypes.c_uint64)] + \
               [(n, ctypes.c_float) for n in ("text_ms", "parse_ms", "build_ms", "run_ms", "d2h_ms")]


class _Cfg(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int32), ("flags", ctypes.c_uint32)]


class _Result(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_uint64), ("n_regions", ctypes.c_uint64),
                ("bad_offsets", ctypes.POINTER(ctypes.c_uint64)),
                ("bad_regions", ctypes.POINTER(ctypes.c_uint32)),
                ("read_type", ctypes.POINTER(ctypes.c_uint8))]


class _DevResult(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_uint64), ("n_regions", ctypes.c_uint64),
                ("d_bad_offsets", ctypes.c_void_p), ("d_bad_regions", ctypes.c_void_p),
                ("d_read_type", ctypes.c_void_p)]


class DeviceBatch(ctypes.Structure):
    """yacrd_device_batch: a CSR resident in HBM + the run's parameters."""
    _fields_ = [("d_offsets", ctypes.c_void_p), ("d_intervals", ctypes.c_void_p), ("d_lengths", ctypes.c_void_p),
                ("n_reads", ctypes.c_uint64), ("n_intervals", ctypes.c_uint64), ("coverage", ctypes.c_uint32),
                ("not_coverage", ctypes.c_double)]


BATCH_DONE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                              ctypes.POINTER(_DevResult))


class _Timing(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in
                ("h2d_ms", "plan_ms", "sweep_small_ms", "sweep_medium_ms", "sweep_general_ms",
                 "compact_ms", "d2h_ms", "total_ms")] + \
               [(n, ctypes.c_uint64) for n in
                ("n_small", "n_medium", "n_general", "iv_small", "iv_medium", "iv_general")] + \
               [("class_ms", ctypes.c_float * 12), ("class_reads", ctypes.c_uint64 * 12),
                ("class_intervals", ctypes.c_uint64 * 12), ("fused_ms", ctypes.c_float),
                ("fused_reads", ctypes.c_uint64), ("fused_intervals", ctypes.c_uint64),
                ("prefiltered_reads", ctypes.c_uint64), ("deferred_reads", ctypes.c_uint64),
                ("deferred_intervals", ctypes.c_uint64), ("screened", ctypes.c_uint32),
                ("timed_runs", ctypes.c_uint32), ("screen_items", ctypes.c_uint32), ("screen_wide", ctypes.c_uint32),
                ("one_launch", ctypes.c_uint32), ("fused_reruns", ctypes.c_uint32),
                ("predicted", ctypes.c_uint32), ("prediction_misses", ctypes.c_uint32),
                ("build_switches", ctypes.c_uint32), ("sorting_build", ctypes.c_uint32)]

CLASS_NAMES = "R2,R4,R8,R16,H16,W2,W4,W8,W16,M1,M2,BIG".split(",")
CLASS_KERNELS = {  # the HIP kernel behind each class, as rocprofv3 prints it
    "R2": "sweep_group_kernel<16, 2, 0>", "R4": "sweep_group_kernel<16, 4, 0>",
    "R8": "sweep_group_kernel<16, 8, 0>", "R16": "sweep_group_kernel<16, 16, 0>",
    "H16": "sweep_group_kernel<32, 16, 0>", "W2": "sweep_group_kernel<64, 2, 0>",
    "W4": "sweep_group_kernel<64, 4, 0>", "W8": "sweep_group_kernel<64, 8, 0>",
    "W16": "sweep_group_kernel<64, 16, 0>",
    # the workgroup classes: the screen, then the fallback of what it leaves (two launches since round 6: screen_wg.h); the bracket
    # also holds the (usually empty) sweep_lds_kernel<1024, 32768> behind it for M2
    "M1": "screen_wg_kernel + screen_wg_fused_kernel over what it leaves",
    "M2": "screen_wg_kernel + screen_wg_fused_kernel over what it leaves (+ sweep_lds_kernel<1024, 32768> over what exceeds 16 384 events)",
    "BIG": "bs_setup / bs_minmax / bs_hist / bs_verdict (screen_big.h)",
}


class _StreamStats(ctypes.Structure):
    _fields_ = [("n_records", ctypes.c_uint64), ("h2d_bytes", ctypes.c_uint64),
                ("h2d_busy_ms", ctypes.c_float), ("build_ms", ctypes.c_float),
                ("run_ms", ctypes.c_float), ("d2h_ms", ctypes.c_float)]


class OvlRec(ctypes.Structure):
    """yacrd_ovl_rec: one overlap line, both reads (handles) and their intervals."""
    _fields_ = [(n, ctypes.c_uint32) for n in ("a", "b", "sa", "ea", "sb", "eb")]


_ACQUIRE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(OvlRec)),
                            ctypes.POINTER(ctypes.c_uint64))
_COMMIT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(OvlRec), ctypes.c_uint64)


class RecSink(ctypes.Structure):
    """yacrd_rec_sink"""
    _fields_ = [("ctx", ctypes.c_void_p), ("acquire", _ACQUIRE), ("commit", _COMMIT)]


OVL_REC_DTYPE = np.dtype([(n, np.uint32) for n in ("a", "b", "sa", "ea", "sb", "eb")])

Result = namedtuple("Result", "bad_offsets bad_regions read_type")

_lib = None


def lib_path():
    return _LIB


def build(force=False):
    """Compile libyacrd_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _LIB


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so.7 (+ HSA runtime).  Two HIP runtimes in one
    process cannot both own the GPU ("No HIP GPUs are available" for whichever comes second), so
    when torch is installed we load ITS copy first (RTLD_GLOBAL); our DT_NEEDED libamdhip64.so.7
    then binds to that already-loaded SONAME.  Without torch (e.g. the C++ CLI) the library uses
    /opt/rocm's runtime through its RUNPATH.  Set YACRD_HIP_RUNTIME=system to skip this."""
    if os.environ.get("YACRD_HIP_RUNTIME", "") == "system":
        return None
    import importlib.util
    import sys
    try:
        if "torch" in sys.modules:
            tdir = os.path.dirname(sys.modules["torch"].__file__)
        else:
            spec = importlib.util.find_spec("torch")
            if spec is None or not spec.origin:
                return None
            tdir = os.path.dirname(spec.origin)
    except (ImportError, ValueError):
        return None
    cand = os.path.join(tdir, "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None
    return cand


def peer_copy_counts():
    """yacrd_debug_peer_copy_counts (tests): cross-engine copies of the N-engine device parser by route:
    [same device, hipMemcpyPeerAsync, staged through the host]."""
    lib = load_library()
    out = (ctypes.c_uint64 * 3)()
    lib.yacrd_debug_peer_copy_counts.restype = None
    lib.yacrd_debug_peer_copy_counts(out)
    return [int(x) for x in out]


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise EngineError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % _LIB)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(_LIB)
    u64p, u32p, u8p = (ctypes.POINTER(t) for t in (ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint8))
    lib.yacrd_abi_version.restype = ctypes.c_int
    lib.yacrd_last_error.restype = ctypes.c_char_p
    lib.yacrd_engine_create.argtypes = [ctypes.POINTER(_Cfg), ctypes.POINTER(ctypes.c_void_p)]
    lib.yacrd_engine_destroy.argtypes = [ctypes.c_void_p]
    lib.yacrd_engine_destroy.restype = None
    lib.yacrd_engine_run.argtypes = [ctypes.c_void_p, u64p, u32p, u32p, ctypes.c_uint64,
                                     ctypes.c_uint32, ctypes.c_double, ctypes.POINTER(_Result)]
    lib.yacrd_result_free.argtypes = [ctypes.POINTER(_Result)]
    lib.yacrd_result_free.restype = None
    lib.yacrd_engine_run_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                            ctypes.c_uint32, ctypes.c_double,
                                            ctypes.POINTER(_DevResult)]
    lib.yacrd_engine_submit_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                               ctypes.c_uint32, ctypes.c_double]
    lib.yacrd_engine_wait.argtypes = [ctypes.c_void_p, ctypes.POINTER(_DevResult)]
    lib.yacrd_engines_run_device_batches.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32,
                                                     ctypes.POINTER(DeviceBatch), ctypes.c_uint32, BATCH_DONE,
                                                     ctypes.c_void_p, ctypes.POINTER(_DevResult)]
    lib.yacrd_engine_fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Result)]
    lib.yacrd_engine_ingest_paf.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_double,
                                            ctypes.POINTER(_Result), ctypes.POINTER(_Reads), ctypes.POINTER(_IngestStats)]
    lib.yacrd_engine_ingest_overlaps.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                                 ctypes.c_double, ctypes.POINTER(_Result), ctypes.POINTER(_Reads),
                                                 ctypes.POINTER(_IngestStats)]
    lib.yacrd_engine_ingest_overlaps_mem.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_uint32, ctypes.c_double, ctypes.POINTER(_Result), ctypes.POINTER(_Reads),
                                                     ctypes.POINTER(_IngestStats)]
    lib.yacrd_engines_ingest_overlaps.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_uint32, ctypes.c_double, ctypes.POINTER(_Result),
                                                  ctypes.POINTER(_Reads), ctypes.POINTER(_IngestStats)]
    lib.yacrd_engines_ingest_overlaps_mem.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                                      ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_double,
                                                      ctypes.POINTER(_Result), ctypes.POINTER(_Reads), ctypes.POINTER(_IngestStats)]
    lib.yacrd_debug_sort_pairs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    lib.yacrd_engine_trim.argtypes = [ctypes.c_void_p]
    lib.yacrd_reads_free.argtypes = [ctypes.POINTER(_Reads)]
    lib.yacrd_reads_free.restype = None
    lib.yacrd_stream_reset.argtypes = [ctypes.c_void_p]
    lib.yacrd_engine_last_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Timing)]
    lib.yacrd_partition_reads.argtypes = [u64p, ctypes.c_uint64, ctypes.c_uint32, u64p]
    lib.yacrd_engine_classify.argtypes = [ctypes.c_void_p, u64p, u32p, u32p, ctypes.c_uint64,
                                          ctypes.c_double, u8p]
    lib.yacrd_engine_event_overhead.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.yacrd_engine_event_overhead.restype = ctypes.c_int
    lib.yacrd_engine_timing_total.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Timing),
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.yacrd_engines_run_partitioned.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32,
                                                  u64p, u32p, u32p, ctypes.c_uint64,
                                                  ctypes.c_uint32, ctypes.c_double,
                                                  ctypes.POINTER(_Result)]
    lib.yacrd_engine_submit.argtypes = [ctypes.c_void_p, u64p, u32p, u32p, ctypes.c_uint64,
                                        ctypes.c_uint32, ctypes.c_double]
    lib.yacrd_engine_collect.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Result)]
    lib.yacrd_pinned_alloc.argtypes = [ctypes.c_size_t]
    lib.yacrd_pinned_alloc.restype = ctypes.c_void_p
    lib.yacrd_pinned_free.argtypes = [ctypes.c_void_p]
    lib.yacrd_pinned_free.restype = None
    lib.yacrd_stream_open.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                      ctypes.POINTER(ctypes.c_void_p)]
    lib.yacrd_stream_sink.argtypes = [ctypes.c_void_p, ctypes.POINTER(RecSink)]
    lib.yacrd_stream_acquire.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(OvlRec)),
                                         ctypes.POINTER(ctypes.c_uint64)]
    lib.yacrd_stream_commit.argtypes = [ctypes.c_void_p, ctypes.POINTER(OvlRec), ctypes.c_uint64]
    lib.yacrd_stream_finish.argtypes = [ctypes.c_void_p, u32p, ctypes.c_uint64, u32p, ctypes.c_uint64,
                                        ctypes.c_uint32, ctypes.c_double, ctypes.POINTER(_Result)]
    lib.yacrd_stream_last_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(_StreamStats)]
    lib.yacrd_stream_close.argtypes = [ctypes.c_void_p]
    lib.yacrd_stream_close.restype = None
    lib.yacrd_stream_device_of.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.yacrd_stream_device_of.restype = ctypes.c_uint32
    lib.yacrd_stream_group_open.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_uint64,
                                            ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
    lib.yacrd_stream_group_sink.argtypes = [ctypes.c_void_p, ctypes.POINTER(RecSink)]
    lib.yacrd_stream_group_finish.argtypes = lib.yacrd_stream_finish.argtypes
    lib.yacrd_stream_group_last_stats.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(_StreamStats),
                                                  ctypes.POINTER(ctypes.c_uint64)]
    lib.yacrd_stream_group_reset.argtypes = [ctypes.c_void_p]
    lib.yacrd_stream_group_close.argtypes = [ctypes.c_void_p]
    lib.yacrd_stream_group_close.restype = None
    _lib = lib
    return lib


def _check(lib, rc):
    if rc != 0:
        raise EngineError("yacrd engine error %d: %s" % (rc, lib.yacrd_last_error().decode()))


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _take(lib, res):
    R, G = int(res.n_reads), int(res.n_regions)
    bo = np.ctypeslib.as_array(res.bad_offsets, shape=(R + 1,)).copy()
    br = (np.ctypeslib.as_array(res.bad_regions, shape=(2 * G,)).copy().reshape(-1, 2)
          if G else np.zeros((0, 2), dtype=np.uint32))
    rt = np.ctypeslib.as_array(res.read_type, shape=(R,)).copy() if R else np.zeros(0, np.uint8)
    lib.yacrd_result_free(ctypes.byref(res))
    return Result(bo, br, rt)


def partition_reads(offsets, n_parts):
    """Contiguous read ranges balanced by interval count; no GPU needed."""
    lib = load_library()
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    cuts = np.zeros(n_parts + 1, dtype=np.uint64)
    _check(lib, lib.yacrd_partition_reads(_ptr(offsets, ctypes.c_uint64), offsets.shape[0] - 1,
                                          n_parts, _ptr(cuts, ctypes.c_uint64)))
    return cuts


def run_partitioned(engines, offsets, intervals, lengths, coverage, not_coverage):
    """Read-partitioned run over several engines (one per GPU); same output as one engine."""
    lib = load_library()
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    intervals = np.ascontiguousarray(intervals, dtype=np.uint32).reshape(-1)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    if intervals.size == 0:
        intervals = np.zeros(2, dtype=np.uint32)
    handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    res = _Result()
    _check(lib, lib.yacrd_engines_run_partitioned(
        handles, len(engines), _ptr(offsets, ctypes.c_uint64), _ptr(intervals, ctypes.c_uint32),
        _ptr(lengths, ctypes.c_uint32), offsets.shape[0] - 1, min(int(coverage), 0xFFFFFFFF),
        float(not_coverage), ctypes.byref(res)))
    return _take(lib, res)


def ingest_overlaps(engines, source, coverage, not_coverage, n_threads=0, fmt=1):
    """yacrd_engines_ingest_overlaps[_mem]: overlap text -> (Result, names, lengths, stats), every engine parsing a byte
    range of the text and sweeping a range of the reads.  `source`: a path, bytes, or (address, n_bytes) of text in host
    memory.  Raises NeedsHostParser when the input is not for the device parser."""
    lib = load_library()
    handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    res, rd, st = _Result(), _Reads(), _IngestStats()
    cov = min(int(coverage), 0xFFFFFFFF)
    if isinstance(source, (str, os.PathLike)):
        rc = lib.yacrd_engines_ingest_overlaps(handles, len(engines), os.fsencode(source), int(fmt), int(n_threads), cov,
                                               float(not_coverage), ctypes.byref(res), ctypes.byref(rd), ctypes.byref(st))
    else:
        if isinstance(source, tuple):
            addr, n, keep = int(source[0]), int(source[1]), None
        else:
            keep = ctypes.create_string_buffer(bytes(source), len(source))
            addr, n = ctypes.addressof(keep), len(source)
        rc = lib.yacrd_engines_ingest_overlaps_mem(handles, len(engines), addr, n, int(fmt), int(n_threads), cov, float(not_coverage),
                                                   ctypes.byref(res), ctypes.byref(rd), ctypes.byref(st))
        del keep
    return engines[0]._ingested(rc, res, rd, st)


def run_device_batches(engines, batches, done=None):
    """yacrd_engines_run_device_batches: `batches` = sequence of (d_offsets, d_intervals, d_lengths,
    n_reads, n_intervals, coverage, not_coverage) with device pointers; batch i runs on
    engines[i % len(engines)], len(engines) of them in flight; done(batch_index, engine) is called in
    order after each batch's wait (return True to stop).  Returns the last batch's device result."""
    lib = load_library()
    arr = (DeviceBatch * len(batches))()
    for i, b in enumerate(batches):
        arr[i] = DeviceBatch(b[0], b[1], b[2], b[3], b[4], min(int(b[5]), 0xFFFFFFFF), float(b[6]))
    handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    by_handle = {e._h.value: e for e in engines}
    cb = BATCH_DONE(lambda user, i, h, res: 1 if done(i, by_handle[h]) else 0) if done else BATCH_DONE()
    out = _DevResult()
    _check(lib, lib.yacrd_engines_run_device_batches(handles, len(engines), arr, len(batches), cb, None,
                                                     ctypes.byref(out)))
    return out


class Engine:
    """One engine per GPU.  Mirrors the BadPart lifecycle: run() == compute_all_bad_part()."""

    def __init__(self, device_id=-1, flags=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.device_id = device_id
        cfg = _Cfg(device_id, flags)
        _check(self._lib, self._lib.yacrd_engine_create(ctypes.byref(cfg), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.yacrd_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run(self, offsets, intervals, lengths, coverage, not_coverage):
        """Host CSR in, host CSR out (H2D + kernels + D2H)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        intervals = np.ascontiguousarray(intervals, dtype=np.uint32).reshape(-1)
        lengths = np.asarray(lengths)
        if lengths.size and int(lengths.max()) > 0xFFFFFFFF:
            raise EngineError("read length >= 2^32 is not supported by the engine ABI")
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        n_reads = offsets.shape[0] - 1
        if lengths.shape[0] != n_reads or intervals.shape[0] != 2 * int(offsets[-1]):
            raise EngineError("malformed CSR")
        res = _Result()
        _check(self._lib, self._lib.yacrd_engine_run(
            self._h, _ptr(offsets, ctypes.c_uint64), _ptr(intervals, ctypes.c_uint32),
            _ptr(lengths, ctypes.c_uint32), n_reads, min(int(coverage), 0xFFFFFFFF),
            float(not_coverage), ctypes.byref(res)))
        return _take(self._lib, res)

    def run_device(self, d_offsets, d_intervals, d_lengths, n_reads, n_intervals, coverage,
                   not_coverage):
        """Inputs are raw device pointers (ints).  Returns (n_regions, ptrs) without D2H."""
        out = _DevResult()
        _check(self._lib, self._lib.yacrd_engine_run_device(
            self._h, d_offsets, d_intervals, d_lengths, n_reads, n_intervals,
            min(int(coverage), 0xFFFFFFFF), float(not_coverage), ctypes.byref(out)))
        return out

    def submit_device(self, d_offsets, d_intervals, d_lengths, n_reads, n_intervals, coverage,
                      not_coverage):
        """run_device without the final wait (see yacrd_engine_submit_device); pair with wait()."""
        _check(self._lib, self._lib.yacrd_engine_submit_device(
            self._h, d_offsets, d_intervals, d_lengths, n_reads, n_intervals,
            min(int(coverage), 0xFFFFFFFF), float(not_coverage)))

    def wait(self):
        out = _DevResult()
        _check(self._lib, self._lib.yacrd_engine_wait(self._h, ctypes.byref(out)))
        return out

    def ingest_paf(self, path, coverage, not_coverage, n_threads=0, fmt=1):
        """yacrd_engine_ingest_overlaps (fmt 1 = PAF, 2 = M4, 0 = by file name): overlap text -> (Result, names,
        lengths, stats) with the parse on the GPU; raises NeedsHostParser when the input is not for the device parser."""
        res, rd, st = _Result(), _Reads(), _IngestStats()
        rc = self._lib.yacrd_engine_ingest_overlaps(self._h, os.fsencode(path), int(fmt), int(n_threads),
                                                    min(int(coverage), 0xFFFFFFFF), float(not_coverage), ctypes.byref(res),
                                                    ctypes.byref(rd), ctypes.byref(st))
        return self._ingested(rc, res, rd, st)

    def ingest_text(self, text, coverage, not_coverage, n_threads=0, fmt=1):
        """yacrd_engine_ingest_overlaps_mem: the same over text in host memory — `text`: bytes, or (address, n_bytes) of
        a buffer such as host.text_from_file's (a compressed overlap file, inflated)."""
        res, rd, st = _Result(), _Reads(), _IngestStats()
        if isinstance(text, tuple):
            addr, n = int(text[0]), int(text[1])
            keep = None
        else:
            keep = ctypes.create_string_buffer(bytes(text), len(text))
            addr, n = ctypes.addressof(keep), len(text)
        rc = self._lib.yacrd_engine_ingest_overlaps_mem(self._h, addr, n, int(fmt), int(n_threads), min(int(coverage), 0xFFFFFFFF),
                                                        float(not_coverage), ctypes.byref(res), ctypes.byref(rd), ctypes.byref(st))
        del keep
        return self._ingested(rc, res, rd, st)

    def _ingested(self, rc, res, rd, st):
        if rc == E_FALLBACK:
            raise NeedsHostParser(self._lib.yacrd_last_error().decode())
        _check(self._lib, rc)
        R = int(rd.n_reads)
        lengths = np.ctypeslib.as_array(rd.lengths, shape=(R,)).copy() if R else np.zeros(0, np.uint32)
        off = np.ctypeslib.as_array(rd.name_off, shape=(R + 1,)).copy()
        blob = ctypes.string_at(rd.names, int(off[-1])) if R else b""
        names = [blob[int(off[i]):int(off[i + 1])].decode("utf-8", "surrogateescape") for i in range(R)]
        stats = {n: getattr(st, n) for n, _ in _IngestStats._fields_}
        self._lib.yacrd_reads_free(ctypes.byref(rd))
        return _take(self._lib, res), names, lengths, stats

    def debug_sort_pairs(self, keys, vals, key_bound):
        """yacrd_debug_sort_pairs (tests): (u64 key, u32 value) pairs sorted by key on the device, stable; returns copies."""
        keys = np.array(keys, dtype=np.uint64)
        vals = np.array(vals, dtype=np.uint32)
        _check(self._lib, self._lib.yacrd_debug_sort_pairs(self._h, keys.ctypes.data, vals.ctypes.data, keys.shape[0], int(key_bound)))
        return keys, vals

    def debug_counters(self):
        """yacrd_debug_last_counters (tools, tests): the device-side counter block of the last run, by name
        (csrc/device_common.h: struct Counters; the layout is mirrored here and is not part of any ABI)."""
        class _Counters(ctypes.Structure):
            _fields_ = [("n", ctypes.c_uint32 * 16), ("iv", ctypes.c_uint64 * 16), ("rej_small", ctypes.c_uint32),
                        ("rej_med", ctypes.c_uint32), ("rej_big", ctypes.c_uint32), ("region_overflow", ctypes.c_uint32),
                        ("scan_ticket", ctypes.c_uint32), ("ob_unsupported", ctypes.c_uint32), ("prefiltered", ctypes.c_uint32),
                        ("over_med", ctypes.c_uint32), ("fb_med", ctypes.c_uint32 * 2), ("fb_big", ctypes.c_uint32),
                        ("fbq_head", ctypes.c_uint32 * 2), ("fbq_done", ctypes.c_uint32 * 2), ("bs_chunks", ctypes.c_uint32),
                        ("fused_gave_up", ctypes.c_uint32), ("total_regions", ctypes.c_uint64)]
        c = _Counters()
        self._lib.yacrd_debug_last_counters.restype = ctypes.c_uint64
        self._lib.yacrd_debug_last_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        self._lib.yacrd_debug_last_counters(self._h, ctypes.byref(c), ctypes.sizeof(c))
        out = {}
        for n, _ in _Counters._fields_:
            v = getattr(c, n)
            out[n] = list(v) if hasattr(v, "__len__") else int(v)
        return out

    def trim(self):
        """yacrd_engine_trim: give the device parser's buffers back."""
        _check(self._lib, self._lib.yacrd_engine_trim(self._h))

    def fetch(self):
        res = _Result()
        _check(self._lib, self._lib.yacrd_engine_fetch(self._h, ctypes.byref(res)))
        return _take(self._lib, res)

    def timing(self):
        t = _Timing()
        _check(self._lib, self._lib.yacrd_engine_last_timing(self._h, ctypes.byref(t)))
        out = {}
        for n, _ in _Timing._fields_:
            v = getattr(t, n)
            out[n] = list(v) if n.startswith("class_") else v
        return out

    def event_overhead_ms(self):
        """Elapsed time of an empty HIP-event bracket on the engine's stream."""
        ms = ctypes.c_float()
        _check(self._lib, self._lib.yacrd_engine_event_overhead(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def timing_total(self, reset=False):
        """(sums over the runs since the last reset, number of runs)."""
        t = _Timing()
        n = ctypes.c_uint64()
        _check(self._lib, self._lib.yacrd_engine_timing_total(self._h, ctypes.byref(t),
                                                              ctypes.byref(n), 1 if reset else 0))
        out = {}
        for name, _ in _Timing._fields_:
            v = getattr(t, name)
            out[name] = list(v) if name.startswith("class_") else v
        return out, int(n.value)

    def submit(self, offsets, intervals, lengths, coverage, not_coverage):
        """run() without the final wait (yacrd_engine_submit); the arrays must stay alive and
        unchanged until collect().  Pass PinnedArray.array views for direct DMA."""
        n_reads = offsets.shape[0] - 1
        self._keep = (offsets, intervals, lengths)
        _check(self._lib, self._lib.yacrd_engine_submit(
            self._h, _ptr(offsets, ctypes.c_uint64), _ptr(intervals.reshape(-1), ctypes.c_uint32),
            _ptr(lengths, ctypes.c_uint32), n_reads, min(int(coverage), 0xFFFFFFFF),
            float(not_coverage)))

    def collect(self):
        res = _Result()
        _check(self._lib, self._lib.yacrd_engine_collect(self._h, ctypes.byref(res)))
        self._keep = None
        return _take(self._lib, res)

    def classify(self, bad_offsets, bad_regions, lengths, not_coverage):
        bad_offsets = np.ascontiguousarray(bad_offsets, dtype=np.uint64)
        bad_regions = np.ascontiguousarray(bad_regions, dtype=np.uint32).reshape(-1)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        n_reads = lengths.shape[0]
        out = np.zeros(max(n_reads, 1), dtype=np.uint8)
        if bad_regions.size == 0:
            bad_regions = np.zeros(2, dtype=np.uint32)
        _check(self._lib, self._lib.yacrd_engine_classify(
            self._h, _ptr(bad_offsets, ctypes.c_uint64), _ptr(bad_regions, ctypes.c_uint32),
            _ptr(lengths, ctypes.c_uint32), n_reads, float(not_coverage),
            _ptr(out, ctypes.c_uint8)))
        return out[:n_reads]


class PinnedArray:
    """A numpy array over page-locked host memory (yacrd_pinned_alloc): the engine moves it over
    PCIe by direct DMA.  Keep the object alive while `array` is in use."""

    def __init__(self, shape, dtype):
        self._lib = load_library()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        self._p = self._lib.yacrd_pinned_alloc(max(n, 1))
        if not self._p:
            raise EngineError("yacrd_pinned_alloc(%d) failed" % n)
        buf = (ctypes.c_char * max(n, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    @classmethod
    def copy_of(cls, a):
        a = np.ascontiguousarray(a)
        p = cls(a.shape, a.dtype)
        p.array[...] = a
        return p

    def close(self):
        if self._p:
            self.array = None
            self._lib.yacrd_pinned_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """yacrd_stream: overlap records go to HBM from pinned buffers while they are produced; finish()
    builds the CSR on the GPU and runs the engine on it."""

    def __init__(self, engine, chunk_records=0, n_buffers=0):
        self._lib = load_library()
        self._engine = engine
        self._h = ctypes.c_void_p()
        _check(self._lib, self._lib.yacrd_stream_open(engine._h, chunk_records, n_buffers,
                                                      ctypes.byref(self._h)))

    def sink(self):
        """A RecSink for host.ingest_stream (keep this Stream alive while it is in use)."""
        s = RecSink()
        _check(self._lib, self._lib.yacrd_stream_sink(self._h, ctypes.byref(s)))
        return s

    def push(self, recs):
        """Copy an OVL_REC_DTYPE array into stream buffers (tests; the parser fills them in place)."""
        recs = np.ascontiguousarray(recs, dtype=OVL_REC_DTYPE)
        at = 0
        while at < len(recs):
            buf = ctypes.POINTER(OvlRec)()
            cap = ctypes.c_uint64()
            _check(self._lib, self._lib.yacrd_stream_acquire(self._h, ctypes.byref(buf), ctypes.byref(cap)))
            n = min(int(cap.value), len(recs) - at)
            ctypes.memmove(buf, recs[at:at + n].ctypes.data, n * OVL_REC_DTYPE.itemsize)
            _check(self._lib, self._lib.yacrd_stream_commit(self._h, buf, n))
            at += n

    def finish(self, handle_map, lengths, coverage, not_coverage):
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        if handle_map is not None:
            handle_map = np.ascontiguousarray(handle_map, dtype=np.uint32)
        res = _Result()
        _check(self._lib, self._lib.yacrd_stream_finish(
            self._h, _ptr(handle_map, ctypes.c_uint32) if handle_map is not None and handle_map.size else None,
            0 if handle_map is None else handle_map.shape[0],
            _ptr(lengths, ctypes.c_uint32) if lengths.size else None, lengths.shape[0],
            min(int(coverage), 0xFFFFFFFF), float(not_coverage), ctypes.byref(res)))
        return _take(self._lib, res)

    def reset(self):
        """Discard every record committed so far (after a failed ingest)."""
        _check(self._lib, self._lib.yacrd_stream_reset(self._h))

    def stats(self):
        st = _StreamStats()
        _check(self._lib, self._lib.yacrd_stream_last_stats(self._h, ctypes.byref(st)))
        return {n: getattr(st, n) for n, _ in _StreamStats._fields_}

    def close(self):
        if self._h:
            self._lib.yacrd_stream_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


HANDLE_ELSEWHERE = 0xFFFFFFFE


def stream_device_of(handle, n_devices):
    """The device a read belongs to in a StreamGroup (yacrd_stream_device_of: handle mod N).  Loads the
    library only: no GPU needed."""
    return int(load_library().yacrd_stream_device_of(int(handle), int(n_devices)))


class StreamGroup:
    """yacrd_stream_group: one yacrd_stream per engine; the sink routes every record to the device(s) of its two
    reads while the parser runs, finish() builds one CSR per device, runs them side by side and merges the
    results into first-appearance order."""

    def __init__(self, engines, chunk_records=0, n_buffers=0):
        self._lib = load_library()
        self._engines = list(engines)
        arr = (ctypes.c_void_p * len(self._engines))(*[e._h for e in self._engines])
        self._h = ctypes.c_void_p()
        _check(self._lib, self._lib.yacrd_stream_group_open(arr, len(self._engines), chunk_records, n_buffers,
                                                            ctypes.byref(self._h)))
        self._sink = None

    def sink(self):
        s = RecSink()
        _check(self._lib, self._lib.yacrd_stream_group_sink(self._h, ctypes.byref(s)))
        return s

    def push(self, recs):
        """Copy an OVL_REC_DTYPE array through the group's sink (tests; the parser fills buffers in place)."""
        recs = np.ascontiguousarray(recs, dtype=OVL_REC_DTYPE)
        if self._sink is None:
            self._sink = self.sink()
        sink = self._sink
        at = 0
        while at < len(recs):
            buf = ctypes.POINTER(OvlRec)()
            cap = ctypes.c_uint64()
            rc = sink.acquire(sink.ctx, ctypes.byref(buf), ctypes.byref(cap))
            _check(self._lib, rc)
            n = min(int(cap.value), len(recs) - at)
            ctypes.memmove(buf, recs[at:at + n].ctypes.data, n * OVL_REC_DTYPE.itemsize)
            _check(self._lib, sink.commit(sink.ctx, buf, n))
            at += n

    def finish(self, handle_map, lengths, coverage, not_coverage):
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        if handle_map is not None:
            handle_map = np.ascontiguousarray(handle_map, dtype=np.uint32)
        res = _Result()
        _check(self._lib, self._lib.yacrd_stream_group_finish(
            self._h, _ptr(handle_map, ctypes.c_uint32) if handle_map is not None and handle_map.size else None,
            0 if handle_map is None else handle_map.shape[0],
            _ptr(lengths, ctypes.c_uint32) if lengths.size else None, lengths.shape[0],
            min(int(coverage), 0xFFFFFFFF), float(not_coverage), ctypes.byref(res)))
        return _take(self._lib, res)

    def reset(self):
        _check(self._lib, self._lib.yacrd_stream_group_reset(self._h))

    def stats(self, index):
        st = _StreamStats()
        owned = ctypes.c_uint64()
        _check(self._lib, self._lib.yacrd_stream_group_last_stats(self._h, index, ctypes.byref(st), ctypes.byref(owned)))
        d = {n: getattr(st, n) for n, _ in _StreamStats._fields_}
        d["reads_owned"] = int(owned.value)
        return d

    def close(self):
        if self._h:
            self._lib.yacrd_stream_group_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
