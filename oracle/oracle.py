"""ctypes binding of oracle/yacrd_oracle.c + small pure-Python restatements of the
reference's ingest and report code.  TEST INFRASTRUCTURE ONLY (see yacrd_oracle.h).

Reference citations are paths relative to the reference root (natir/yacrd @ 2024-11-08).
"""
import ctypes
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libyacrd_oracle.so")

NOT_BAD, CHIMERIC, NOT_COVERED = 0, 1, 2
# src/editor/mod.rs:51-58 ReadType::as_str
TYPE_NAMES = {NOT_BAD: "NotBad", CHIMERIC: "Chimeric", NOT_COVERED: "NotCovered"}

_lib = None


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "yacrd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libyacrd_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        lib.yo_compute_bad_part.restype = ctypes.c_size_t
        lib.yo_compute_bad_part.argtypes = [u32p, ctypes.c_size_t, ctypes.c_uint64,
                                            ctypes.c_uint64, u32p, u32p]
        lib.yo_type_of_read.restype = ctypes.c_int
        lib.yo_type_of_read.argtypes = [ctypes.c_uint64, u32p, ctypes.c_size_t, ctypes.c_double]
        lib.yo_run.restype = ctypes.c_int
        lib.yo_run.argtypes = [u64p, u32p, u64p, ctypes.c_uint64, ctypes.c_uint64,
                               ctypes.c_double, ctypes.c_int, u64p,
                               ctypes.POINTER(u32p), ctypes.POINTER(ctypes.c_uint8)]
        libc = ctypes.CDLL(None)
        libc.free.argtypes = [ctypes.c_void_p]
        libc.free.restype = None
        lib._free = libc.free
        _lib = lib
    return _lib


def _p32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def _p64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def compute_bad_part(intervals, length, coverage):
    """src/stack.rs:61-139.  intervals: iterable of (start,end); returns list of (begin,end)."""
    lib = _load()
    iv = np.ascontiguousarray(np.array(list(intervals), dtype=np.uint32).reshape(-1, 2))
    n = iv.shape[0]
    out = np.zeros(2 * (n + 2), dtype=np.uint32)
    heap = np.zeros(max(n, 1), dtype=np.uint32)
    g = lib.yo_compute_bad_part(_p32(iv), n, int(length), int(coverage), _p32(out), _p32(heap))
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(g)]


def type_of_read(length, regions, not_covered):
    """src/editor/mod.rs:85-100."""
    lib = _load()
    reg = np.ascontiguousarray(np.array(list(regions), dtype=np.uint32).reshape(-1, 2))
    return int(lib.yo_type_of_read(int(length), _p32(reg), reg.shape[0], float(not_covered)))


def run(offsets, intervals, lengths, coverage, not_covered, n_threads=1):
    """src/stack.rs:143-162 over a CSR batch + per-read type_of_read.
    Returns (bad_offsets u64[R+1], bad_regions u32[G,2], read_type u8[R])."""
    lib = _load()
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    intervals = np.ascontiguousarray(intervals, dtype=np.uint32).reshape(-1)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
    n_reads = offsets.shape[0] - 1
    assert lengths.shape[0] == n_reads
    assert intervals.shape[0] == 2 * int(offsets[-1])
    if intervals.shape[0] == 0:
        intervals = np.zeros(2, dtype=np.uint32)
    bad_offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    read_type = np.zeros(max(n_reads, 1), dtype=np.uint8)
    reg_ptr = ctypes.POINTER(ctypes.c_uint32)()
    rc = lib.yo_run(_p64(offsets), _p32(intervals), _p64(lengths), n_reads, int(coverage),
                    float(not_covered), int(n_threads), _p64(bad_offsets),
                    ctypes.byref(reg_ptr),
                    read_type.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    if rc != 0:
        raise MemoryError("yo_run failed")
    g = int(bad_offsets[-1])
    if g:
        regions = np.ctypeslib.as_array(reg_ptr, shape=(2 * g,)).copy().reshape(-1, 2)
    else:
        regions = np.zeros((0, 2), dtype=np.uint32)
    lib._free(ctypes.cast(reg_ptr, ctypes.c_void_p))
    return bad_offsets, regions, read_type[:n_reads]


# --------------------------------------------------------------------------------------
# Ingest: src/reads2ovl/mod.rs:83-145 + src/io.rs:23-50 + src/reads2ovl/fullmemory.rs:82-90.
# Pure Python, small inputs only.  Record syntax restates the csv crate the reference reads with
# (csv 1.3.0 / csv-core 0.1.11, Cargo.lock:241; builder settings mod.rs:84-88: delimiter, no
# headers, flexible, everything else default => quote '"', double_quote on, no escape, terminator
# "CRLF" = any of \r, \n, \r\n) as its published DFA: StartRecord skips terminators; a field that
# STARTS with the quote is quoted (delimiters and terminators inside are data, "" is one quote, and
# text after the closing quote is appended as is); a quote inside an unquoted field is data.
# Integer fields: `0x` + hex via from_str_radix, else FromStr (optional leading '+').
# The crate is not vendored in /root/reference, so this part is "restated from published
# behaviour", pinned only by the reference's 2-line vectors (mod.rs:173-237) and tests/reads.paf.

def csv_records(text, delim):
    """Yield lists of fields (str) like csv::Reader::read_record with the settings above."""
    rec, field, i, n = [], [], 0, len(text)
    START_RECORD, START_FIELD, IN_FIELD, IN_QUOTED, AFTER_QUOTE = range(5)
    st = START_RECORD
    while i < n:
        c = text[i]
        i += 1
        term = c in "\r\n"
        if st == START_RECORD:
            if term:
                continue
            st = START_FIELD
        if st == START_FIELD:
            if c == '"':
                st = IN_QUOTED
            elif c == delim:
                rec.append("")
            elif term:
                rec.append("")
                yield rec
                rec, st = [], START_RECORD
            else:
                field.append(c)
                st = IN_FIELD
        elif st == IN_FIELD or st == AFTER_QUOTE:
            if st == AFTER_QUOTE and c == '"':
                field.append('"')
                st = IN_QUOTED
            elif c == delim:
                rec.append("".join(field))
                field, st = [], START_FIELD
            elif term:
                rec.append("".join(field))
                yield rec
                rec, field, st = [], [], START_RECORD
            else:
                field.append(c)
                st = IN_FIELD
        elif st == IN_QUOTED:
            if c == '"':
                st = AFTER_QUOTE
            else:
                field.append(c)
    if st in (IN_FIELD, IN_QUOTED, AFTER_QUOTE) or (st == START_FIELD and rec):
        rec.append("".join(field))
    if rec:
        yield rec


def csv_int(f, bits):
    """The csv crate's integer deserialisation: 0x-prefixed hex or decimal, optional '+'."""
    if f.startswith("0x"):
        body, base = f[2:], 16
    else:
        body, base = f, 10
    digits = body[1:] if body.startswith("+") else body
    ok = "0123456789abcdefABCDEF" if base == 16 else "0123456789"
    if not digits or any(ch not in ok for ch in digits):
        raise ValueError("not an integer: %r" % f)
    v = int(digits, base)
    if v >= 1 << bits:
        raise ValueError("integer out of range: %r" % f)
    return v


def csv_char(f):
    """serde `char` (PafRecord._strand, M4Record._strand_a/_strand_b, src/io.rs:29,41,45): exactly one scalar."""
    if len(f) != 1:
        raise ValueError("not a single character: %r" % f)


_F64 = re.compile(r"[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|inf|infinity|nan)\Z", re.IGNORECASE)


def csv_f64(f):
    """Rust's f64::from_str (M4Record._error, src/io.rs:39): sign, digits, point, exponent, or inf / infinity /
    nan in any case — no hex floats, no blanks."""
    if not _F64.match(f):
        raise ValueError("not a float: %r" % f)


def _ingest(lines, delim, cols, chars=(), f64s=(), u64s=()):
    ia, la, ba, ea, ib, lb, bb, eb = cols
    reads = {}  # id -> [list of (s,e), length]; dict keeps first-appearance order

    def add(rid, ovl, length):  # fullmemory.rs:82-90: length = first seen
        ent = reads.get(rid)
        if ent is None:
            reads[rid] = [[ovl], length]
        else:
            ent[0].append(ovl)

    text = lines if isinstance(lines, str) else "".join(
        l if l.endswith(("\n", "\r")) else l + "\n" for l in lines)
    for f in csv_records(text, delim):
        # the fields the reference deserialises and then drops fail the record all the same (mod.rs:93-97 / :125-129)
        for i in chars:
            csv_char(f[i])
        for i in f64s:
            csv_f64(f[i])
        for i in u64s:
            csv_int(f[i], 64)
        add(f[ia], (csv_int(f[ba], 32), csv_int(f[ea], 32)), csv_int(f[la], 64))  # mod.rs:108 / :140
        add(f[ib], (csv_int(f[bb], 32), csv_int(f[eb], 32)), csv_int(f[lb], 64))  # mod.rs:109 / :141
    return reads


def parse_paf(lines):
    """PafRecord columns, src/io.rs:23-34."""
    return _ingest(lines, "\t", (0, 1, 2, 3, 5, 6, 7, 8), chars=(4,))


def parse_m4(lines):
    """M4Record columns, src/io.rs:36-50: a b err shared sa ba ea la sb bb eb lb."""
    return _ingest(lines, " ", (0, 7, 5, 6, 1, 11, 9, 10), chars=(4, 8), f64s=(2,), u64s=(3,))


def to_csr(reads):
    """dict from parse_* -> (names, offsets u64, intervals u32[I,2], lengths u64)."""
    names = list(reads.keys())
    offsets = np.zeros(len(names) + 1, dtype=np.uint64)
    iv = []
    lengths = np.zeros(len(names), dtype=np.uint64)
    for i, k in enumerate(names):
        ovl, length = reads[k]
        iv.extend(ovl)
        offsets[i + 1] = len(iv)
        lengths[i] = length
    intervals = np.array(iv, dtype=np.uint32).reshape(-1, 2)
    return names, offsets, intervals, lengths


# --------------------------------------------------------------------------------------
# Report: src/editor/mod.rs:61-83 (line) and :102-107 (bad_region_format).

def report_line(name, length, regions, read_type):
    body = ";".join("%d,%d,%d" % ((int(e) - int(b)) & 0xFFFFFFFF, b, e) for b, e in regions)
    return "%s\t%s\t%d\t%s" % (TYPE_NAMES[int(read_type)], name, int(length), body)


def report_lines(reads, coverage, not_covered):
    """Whole reference pipeline on a parsed dict: compute_bad_part + report, one line/read."""
    out = []
    for name, (ovl, length) in reads.items():
        regions = compute_bad_part(ovl, length, coverage)
        out.append(report_line(name, length, regions, type_of_read(length, regions, not_covered)))
    return out


def report_from_csr(names, lengths, bad_offsets, bad_regions, read_type):
    out = []
    for i, name in enumerate(names):
        a, b = int(bad_offsets[i]), int(bad_offsets[i + 1])
        out.append(report_line(name, lengths[i], np.asarray(bad_regions)[a:b].tolist(),
                               read_type[i]))
    return out
