"""ctypes binding of include/yacrd_host.h (libyacrd_host.so): ingest, report, synthetic data."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libyacrd_host.so")

SYNTH_ONT, SYNTH_SEQUEL, SYNTH_SKEWED = 0, 1, 2
# yacrd_synth_cfg.flags (include/yacrd_host.h)
SYNTH_F_NO_INJECTION, SYNTH_F_JITTER, SYNTH_F_SIGMA_X4 = 1, 2, 4


def synth_f_chimera_pct(p):
    """YACRD_SYNTH_F_CHIMERA_PCT: per cent of the reads that are chimeras (default 2)."""
    return (int(p) & 0xFF) << 16


def synth_f_sigma(s):
    """sigma (positions) of the dovetail ends' offset (default 30): 1..255 exactly, up to 1020 in steps of 4
    (YACRD_SYNTH_F_SIGMA_X4)"""
    s = int(s)
    if not 0 <= s <= 1020:
        raise ValueError("sigma must be within 0..1020")
    return (SYNTH_F_SIGMA_X4 | ((s // 4) << 8)) if s > 255 else (s << 8)
FMT_AUTO, FMT_PAF, FMT_M4 = 0, 1, 2

EXPORTED_SYMBOLS = [
    "yacrd_host_last_error", "yacrd_csr_from_file", "yacrd_csr_from_memory", "yacrd_csr_get",
    "yacrd_csr_find", "yacrd_csr_free", "yacrd_report_write", "yacrd_synth_csr", "yacrd_synth_paf",
    "yacrd_edit_file", "yacrd_edit_file_mt", "yacrd_text_from_file", "yacrd_text_free", "yacrd_report_read", "yacrd_report_get", "yacrd_report_free",
    "yacrd_synth_fastq", "yacrd_ingest_stream", "yacrd_ingest_stream_memory", "yacrd_csr_handle_map",
]

OP_SCRUBB, OP_FILTER, OP_EXTRACT, OP_SPLIT = 0, 1, 2, 3


class HostError(RuntimeError):
    pass


class _View(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_uint64), ("n_intervals", ctypes.c_uint64),
                ("n_records", ctypes.c_uint64),
                ("offsets", ctypes.POINTER(ctypes.c_uint64)),
                ("intervals", ctypes.POINTER(ctypes.c_uint32)),
                ("lengths", ctypes.POINTER(ctypes.c_uint32)),
                ("name_off", ctypes.POINTER(ctypes.c_uint64)),
                ("names", ctypes.POINTER(ctypes.c_char))]


class _Text(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("n", ctypes.c_uint64), ("cap", ctypes.c_uint64), ("members", ctypes.c_uint64),
                ("threads", ctypes.c_uint32), ("compression", ctypes.c_int32)]


class Text:
    """yacrd_text_from_file: a compressed (gzip / bzip2 / xz) file inflated into memory; .address / .n_bytes for
    Engine.ingest_text, .bytes() for a copy; None from text_from_file when the file is not compressed."""

    def __init__(self, lib, t):
        self._lib, self._t = lib, t
        self.address, self.n_bytes = int(t.data or 0), int(t.n)
        self.members, self.threads, self.compression = int(t.members), int(t.threads), int(t.compression)

    def bytes(self):
        return ctypes.string_at(self.address, self.n_bytes) if self.n_bytes else b""

    def close(self):
        if self._t is not None:
            self._lib.yacrd_text_free(ctypes.byref(self._t))
            self._t = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def text_from_file(path, n_threads=0):
    lib = load_library()
    t = _Text()
    rc = lib.yacrd_text_from_file(os.fsencode(path), n_threads, ctypes.byref(t))
    if rc == 2:
        return None
    _check(lib, rc)
    return Text(lib, t)


class _BadParts(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_uint64),
                ("name_off", ctypes.POINTER(ctypes.c_uint64)),
                ("names", ctypes.c_char_p),
                ("lengths", ctypes.POINTER(ctypes.c_uint32)),
                ("bad_offsets", ctypes.POINTER(ctypes.c_uint64)),
                ("bad_regions", ctypes.POINTER(ctypes.c_uint32)),
                ("read_type", ctypes.POINTER(ctypes.c_uint8))]


class _SynthCfg(ctypes.Structure):
    _fields_ = [("profile", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("n_reads", ctypes.c_uint64), ("n_overlaps", ctypes.c_uint64),
                ("seed", ctypes.c_uint64)]


_lib = None


def build(force=False):
    cmd = ["make", "-s", "-C", os.path.join(_HERE, "csrc", "host")]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _LIB


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise HostError("%s is missing: run __graft_entry__.build()" % _LIB)
        lib = ctypes.CDLL(_LIB)
        lib.yacrd_host_last_error.restype = ctypes.c_char_p
        lib.yacrd_csr_from_file.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_void_p)]
        lib.yacrd_csr_from_memory.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                              ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.yacrd_ingest_stream.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_void_p)]
        lib.yacrd_ingest_stream_memory.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_void_p,
                                                   ctypes.POINTER(ctypes.c_void_p)]
        lib.yacrd_csr_handle_map.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32)),
                                             ctypes.POINTER(ctypes.c_uint64)]
        lib.yacrd_csr_get.argtypes = [ctypes.c_void_p, ctypes.POINTER(_View)]
        lib.yacrd_csr_find.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        lib.yacrd_csr_find.restype = ctypes.c_int64
        lib.yacrd_csr_free.argtypes = [ctypes.c_void_p]
        lib.yacrd_csr_free.restype = None
        lib.yacrd_report_write.argtypes = [ctypes.c_char_p, ctypes.POINTER(_View),
                                           ctypes.POINTER(ctypes.c_uint64),
                                           ctypes.POINTER(ctypes.c_uint32),
                                           ctypes.POINTER(ctypes.c_uint8)]
        lib.yacrd_synth_csr.argtypes = [ctypes.POINTER(_SynthCfg), ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.POINTER(ctypes.c_uint32),
                                        ctypes.POINTER(ctypes.c_uint32)]
        lib.yacrd_synth_paf.argtypes = [ctypes.POINTER(_SynthCfg), ctypes.c_char_p]
        lib.yacrd_synth_fastq.argtypes = [ctypes.POINTER(_SynthCfg), ctypes.c_uint64, ctypes.c_char_p]
        lib.yacrd_edit_file.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p,
                                        ctypes.POINTER(_BadParts)]
        lib.yacrd_text_from_file.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(_Text)]
        lib.yacrd_text_free.argtypes = [ctypes.POINTER(_Text)]
        lib.yacrd_text_free.restype = None
        lib.yacrd_edit_file_mt.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p,
                                           ctypes.POINTER(_BadParts), ctypes.c_int]
        lib.yacrd_report_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.yacrd_report_get.argtypes = [ctypes.c_void_p, ctypes.POINTER(_BadParts)]
        lib.yacrd_report_free.argtypes = [ctypes.c_void_p]
        lib.yacrd_report_free.restype = None
        _lib = lib
    return _lib


def _check(lib, rc):
    if rc != 0:
        raise HostError(lib.yacrd_host_last_error().decode())


class Csr:
    """Owns a yacrd_csr; numpy views are copies so they outlive it."""

    def __init__(self, handle):
        self._lib = load_library()
        self._h = handle
        v = _View()
        _check(self._lib, self._lib.yacrd_csr_get(self._h, ctypes.byref(v)))
        self._view = v
        R, I = int(v.n_reads), int(v.n_intervals)
        self.n_reads, self.n_intervals, self.n_records = R, I, int(v.n_records)
        self.streamed = not bool(v.offsets)  # yacrd_ingest_stream: the CSR lives in HBM
        if self.streamed:
            self.offsets = self.intervals = None
            mp = ctypes.POINTER(ctypes.c_uint32)()
            nh = ctypes.c_uint64()
            _check(self._lib, self._lib.yacrd_csr_handle_map(self._h, ctypes.byref(mp), ctypes.byref(nh)))
            self.handle_map = (np.ctypeslib.as_array(mp, shape=(int(nh.value),)).copy()
                               if nh.value else np.zeros(0, np.uint32))
        else:
            self.offsets = np.ctypeslib.as_array(v.offsets, shape=(R + 1,)).copy()
            self.intervals = (np.ctypeslib.as_array(v.intervals, shape=(2 * I,)).copy().reshape(-1, 2)
                              if I else np.zeros((0, 2), np.uint32))
        self.lengths = (np.ctypeslib.as_array(v.lengths, shape=(R,)).copy()
                        if R else np.zeros(0, np.uint32))
        name_off = np.ctypeslib.as_array(v.name_off, shape=(R + 1,)).copy()
        blob = ctypes.string_at(v.names, int(name_off[-1])) if R else b""
        self.names = [blob[int(name_off[i]):int(name_off[i + 1])].decode() for i in range(R)]

    def find(self, name):
        b = name.encode()
        return int(self._lib.yacrd_csr_find(self._h, b, len(b)))

    def write_report(self, path, bad_offsets, bad_regions, read_type):
        bo = np.ascontiguousarray(bad_offsets, dtype=np.uint64)
        br = np.ascontiguousarray(bad_regions, dtype=np.uint32).reshape(-1)
        if br.size == 0:
            br = np.zeros(2, np.uint32)
        rt = np.ascontiguousarray(read_type, dtype=np.uint8)
        if rt.size == 0:
            rt = np.zeros(1, np.uint8)
        _check(self._lib, self._lib.yacrd_report_write(
            path.encode(), ctypes.byref(self._view),
            bo.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
            br.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
            rt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))

    def close(self):
        if self._h:
            self._lib.yacrd_csr_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def csr_from_file(path, fmt=FMT_AUTO, n_threads=0):
    lib = load_library()
    h = ctypes.c_void_p()
    _check(lib, lib.yacrd_csr_from_file(path.encode(), fmt, n_threads, ctypes.byref(h)))
    return Csr(h)


def ingest_stream(path, sink, fmt=FMT_AUTO, n_threads=0):
    """Parse `path`, handing the overlap records to `sink` (a yacrd_rec_sink struct, e.g.
    yacrd_amd.Stream.sink()) while parsing.  Returns a Csr with names / lengths / handle_map."""
    lib = load_library()
    h = ctypes.c_void_p()
    _check(lib, lib.yacrd_ingest_stream(path.encode(), fmt, n_threads, ctypes.addressof(sink), ctypes.byref(h)))
    return Csr(h)


def ingest_stream_memory(text, sink, fmt, n_threads=0):
    lib = load_library()
    if isinstance(text, str):
        text = text.encode()
    h = ctypes.c_void_p()
    _check(lib, lib.yacrd_ingest_stream_memory(text, len(text), fmt, n_threads, ctypes.addressof(sink),
                                               ctypes.byref(h)))
    return Csr(h)


def csr_from_memory(text, fmt, n_threads=0):
    lib = load_library()
    if isinstance(text, str):
        text = text.encode()
    h = ctypes.c_void_p()
    _check(lib, lib.yacrd_csr_from_memory(text, len(text), fmt, n_threads, ctypes.byref(h)))
    return Csr(h)


def synth_csr(profile, n_reads, n_overlaps, seed, flags=0):
    """Deterministic synthetic pile-up straight to CSR.
    Returns offsets u64[R+1], intervals u32[I,2], lengths u32[R]."""
    lib = load_library()
    cfg = _SynthCfg(profile, flags, n_reads, n_overlaps, seed)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    intervals = np.zeros(4 * n_overlaps, dtype=np.uint32)
    lengths = np.zeros(n_reads, dtype=np.uint32)
    _check(lib, lib.yacrd_synth_csr(ctypes.byref(cfg),
                                    offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                    intervals.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                    lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    return offsets, intervals.reshape(-1, 2), lengths


def synth_paf(profile, n_reads, n_overlaps, seed, path, flags=0):
    lib = load_library()
    cfg = _SynthCfg(profile, flags, n_reads, n_overlaps, seed)
    _check(lib, lib.yacrd_synth_paf(ctypes.byref(cfg), path.encode()))


def synth_fastq(profile, n_reads, n_overlaps, seed, extra_reads, path, flags=0):
    lib = load_library()
    cfg = _SynthCfg(profile, flags, n_reads, n_overlaps, seed)
    _check(lib, lib.yacrd_synth_fastq(ctypes.byref(cfg), extra_reads, path.encode()))


def edit_file(op, in_path, out_path, names, lengths, bad_offsets, bad_regions, read_type, n_threads=0):
    """editor::{scrubbing,filter,extract,split} over the BadPart table (names, lengths, region CSR,
    engine read types).  n_threads: 0 = every usable CPU (plain FASTA / FASTQ are edited chunk-parallel)."""
    lib = load_library()
    blob = b"".join(n.encode() for n in names)
    name_off = np.zeros(len(names) + 1, dtype=np.uint64)
    np.cumsum([len(n.encode()) for n in names], out=name_off[1:])
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    bo = np.ascontiguousarray(bad_offsets, dtype=np.uint64)
    br = np.ascontiguousarray(bad_regions, dtype=np.uint32).reshape(-1)
    if br.size == 0:
        br = np.zeros(2, np.uint32)
    rt = np.ascontiguousarray(read_type, dtype=np.uint8)
    view = _BadParts(len(names), name_off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), blob,
                     lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                     bo.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                     br.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                     rt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    _check(lib, lib.yacrd_edit_file_mt(op, in_path.encode(), out_path.encode(), ctypes.byref(view), n_threads))


def report_read(path):
    """FromReport: .yacrd -> (names, lengths u32, bad_offsets u64, bad_regions u32[G,2])."""
    lib = load_library()
    h = ctypes.c_void_p()
    _check(lib, lib.yacrd_report_read(path.encode(), ctypes.byref(h)))
    try:
        v = _BadParts()
        _check(lib, lib.yacrd_report_get(h, ctypes.byref(v)))
        R = int(v.n_reads)
        name_off = np.ctypeslib.as_array(v.name_off, shape=(R + 1,)).copy()
        blob = ctypes.string_at(v.names, int(name_off[-1])) if R else b""
        names = [blob[int(name_off[i]):int(name_off[i + 1])].decode() for i in range(R)]
        lengths = np.ctypeslib.as_array(v.lengths, shape=(R,)).copy() if R else np.zeros(0, np.uint32)
        bo = np.ctypeslib.as_array(v.bad_offsets, shape=(R + 1,)).copy()
        G = int(bo[-1])
        br = (np.ctypeslib.as_array(v.bad_regions, shape=(2 * G,)).copy().reshape(-1, 2)
              if G else np.zeros((0, 2), np.uint32))
    finally:
        lib.yacrd_report_free(h)
    return names, lengths, bo, br
