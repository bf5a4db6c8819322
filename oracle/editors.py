"""Pure-Python restatement of the reference's FASTQ editors — TEST INFRASTRUCTURE ONLY.
Pinned on the reference's golden files in tests/test_oracle_editors.py.

  scrubb   src/editor/scrubbing.rs:156-236      split    src/editor/split.rs:151-226
  filter   src/editor/filter.rs:101-138         extract  src/editor/extract.rs:103-142
`table` maps read id -> (regions [(b,e),...], length); unknown ids answer ([], 0) like
BadPart::get_bad_part (src/stack.rs:164-169).  type_of_read is the oracle's (C restatement).
"""
from . import oracle as _o


def _records(data):
    lines = data.split(b"\n")
    i = 0
    while i + 3 < len(lines) or (i + 3 == len(lines) - 0 and lines[i]):
        if not lines[i]:
            i += 1
            continue
        head = lines[i][1:]
        name, _, desc = head.partition(b" ")
        yield name, desc, lines[i + 1], lines[i + 3]
        i += 4


def _emit(out, name, desc, seq, qual):
    out.append(b"@" + name + (b" " + desc if desc else b"") + b"\n" + seq + b"\n+\n" + qual + b"\n")


def edit_fastq(op, data, table, not_covered):
    out = []
    for name, desc, seq, qual in _records(data):
        key = name.split()[0].decode() if name.split() else ""
        regions, length = table.get(key, ([], 0))
        rtype = _o.type_of_read(length, regions, not_covered)
        if op == "filter":
            if rtype == _o.NOT_BAD:
                _emit(out, name, desc, seq, qual)
            continue
        if op == "extract":
            if rtype != _o.NOT_BAD:
                _emit(out, name, desc, seq, qual)
            continue
        if rtype == _o.NOT_COVERED:
            continue
        if (op == "scrubb" and not regions) or (op == "split" and rtype == _o.NOT_BAD):
            _emit(out, name, desc, seq, qual)
            continue
        poss = [0]
        if op == "scrubb":
            for b, e in regions:
                poss += [b, e]
            if poss[-1] != (length & 0xFFFFFFFF):
                poss.append(length & 0xFFFFFFFF)
            if poss[0] == 0 and poss[1] == 0:
                poss = poss[2:]
            pairs = [(poss[k], poss[k + 1]) for k in range(0, len(poss) - 1, 2)]  # chunks_exact(2)
        else:
            for b, e in regions:
                if b == 0 or e == (length & 0xFFFFFFFF):
                    continue
                poss += [b, e]
            poss.append(length & 0xFFFFFFFF)
            pairs = [(poss[k], poss[k + 1]) for k in range(0, len(poss), 2)]
        for p0, p1 in pairs:
            if p0 > len(seq) or p1 > len(seq):
                break
            _emit(out, name + b"_%d_%d" % (p0, p1), desc, seq[p0:p1], qual[p0:p1])
    return b"".join(out)
