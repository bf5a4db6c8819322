"""world_size-2 run of the read-partitioned path on CPU (gloo): every rank takes its contiguous
read range, computes it independently (no collective on the data path) and the rank-ordered
concatenation equals the single-process result.  The per-rank compute stands in with the CPU
oracle here (no GPU in this container); on the GPU box the same plumbing drives the engine
(bench.py, tests/test_gpu_parity.py::test_read_partitioned_multi_engine)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle, yacrd_amd
from yacrd_amd import dist as ydist, host
offsets, intervals, lengths = host.synth_csr(host.SYNTH_ONT, 3000, 60000, 20241110)
rank, local_rank, world = ydist.env_rank()
d = ydist.init(backend="gloo")
r0, r1 = ydist.shard(offsets, rank, world, yacrd_amd.partition_reads)
off, iv, ln = ydist.local_csr(offsets, intervals, lengths, r0, r1)
part = oracle.run(off, iv, ln.astype(np.uint64), 4, 0.4)
d.barrier()
slowest = ydist.max_over_ranks(d, float(rank + 1))
full = ydist.gather_results(d, part)
if rank == 0:
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), 4, 0.4)
    ok = all(np.array_equal(a, b) for a, b in zip(full, want)) and slowest == float(world)
    print("RESULT", "OK" if ok else "MISMATCH", r0, r1, int(full[0][-1]))
d.destroy_process_group()
"""


def test_two_rank_read_partition_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RESULT OK" in outs[0][0], outs


def test_shards_cover_all_reads_once():
    sys.path.insert(0, ROOT)
    import yacrd_amd
    from yacrd_amd import dist as ydist, host
    offsets, _, _ = host.synth_csr(host.SYNTH_SEQUEL, 1000, 30000, 3)
    for world in (1, 2, 3, 8):
        spans = [ydist.shard(offsets, r, world, yacrd_amd.partition_reads) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 1000
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


GROUP_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle, yacrd_amd
from yacrd_amd import dist as ydist, host
# what yacrd_stream_group does, with the oracle standing in for the device: a record belongs to the ranks of
# its two reads (handle mod world, include/yacrd_engine.h::yacrd_stream_device_of), every rank keeps the halves
# that name its own reads, numbers them in first-appearance order and sweeps them; merged by first appearance
# the result must be the whole file's.
paf = os.path.join(sys.argv[2], "g.paf")
rank, local_rank, world = ydist.env_rank()
d = ydist.init(backend="gloo")
if rank == 0:
    host.synth_paf(host.SYNTH_ONT, 2500, 50000, 20250302, paf)
d.barrier()
from test_ingest_stream import PySink  # (the yacrd_rec_sink ABI over numpy buffers)
ref = host.csr_from_file(paf, n_threads=2)
box = [None]
if rank == 0:  # one parser (handles depend on the parser threads' timing), its records go to every rank
    sink = PySink()
    c = host.ingest_stream(paf, sink.struct, n_threads=3)
    box = [(sink.records(), np.array(c.handle_map), np.array(c.lengths))]
d.broadcast_object_list(box, src=0)
records, hmap, lengths = box[0]
mine = np.array([yacrd_amd.stream_device_of(h, world) == rank for h in range(len(hmap))])
owned = np.flatnonzero(mine & (hmap != 0xFFFFFFFF))
gids = np.sort(hmap[owned])                       # first-appearance order of this rank's reads
local_of = np.full(len(lengths), -1, np.int64); local_of[gids] = np.arange(len(gids))
per = [[] for _ in gids]
for r in records:
    for h, s, e in ((r["a"], r["sa"], r["ea"]), (r["b"], r["sb"], r["eb"])):
        if mine[h]:
            per[local_of[hmap[h]]].append((s, e))
off = np.zeros(len(gids) + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in per])
iv = np.array([p for x in per for p in x], dtype=np.uint32).reshape(-1, 2)
part = oracle.run(off, iv, lengths[gids].astype(np.uint64), 4, 0.4)
parts = [None] * world
d.all_gather_object(parts, (gids, part))
if rank == 0:
    want = oracle.run(ref.offsets, ref.intervals, ref.lengths.astype(np.uint64), 4, 0.4)
    R = len(ref.lengths)
    seen = np.zeros(R, np.int64)
    ok = True
    for g, (boff, breg, rtype) in parts:
        seen[g] += 1
        for l, r in enumerate(g):
            a = breg[int(boff[l]):int(boff[l + 1])]
            b = want[1][int(want[0][r]):int(want[0][r + 1])]
            ok = ok and np.array_equal(a, b) and rtype[l] == want[2][r]
    ok = ok and bool((seen == 1).all())
    print("RESULT", "OK" if ok else "MISMATCH")
d.destroy_process_group()
"""


def test_two_rank_handle_partition_gloo(tmp_path):
    script = tmp_path / "gworker.py"
    script.write_text(GROUP_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RESULT OK" in outs[0][0], outs


def test_handle_partition_is_a_partition():
    sys.path.insert(0, ROOT)
    import yacrd_amd
    for world in (1, 2, 3, 8):
        owners = [yacrd_amd.stream_device_of(h, world) for h in range(1000)]
        assert set(owners) == set(range(world)) and all(o == h % world for h, o in enumerate(owners))
    assert yacrd_amd.stream_device_of(0xFFFFFFFD, 8) == 0xFFFFFFFD % 8


SHARED_WORKER = r"""
import os, sys, types
import numpy as np
sys.path.insert(0, sys.argv[1])
import bench, yacrd_amd
from yacrd_amd import dist as ydist, host
# bench.py's input for N > 1: generated once (rank 0), written to /dev/shm, mapped by the other ranks; every rank
# slices its own contiguous read range out of it and the files are gone afterwards
rank, local_rank, world = ydist.env_rank()
d = ydist.init(backend="gloo")
cx = types.SimpleNamespace(world=world, rank=rank, dist=d, host=host, prof=lambda name: host.SYNTH_SEQUEL)
offsets, intervals, lengths = bench.shared_csr(cx, "sequel", 3000, 90000, 77, 0)
want = host.synth_csr(host.SYNTH_SEQUEL, 3000, 90000, 77)
same = all(np.array_equal(np.asarray(a), b) for a, b in zip((offsets, intervals, lengths), want))
r0, r1 = ydist.shard(offsets, rank, world, yacrd_amd.partition_reads)
off, iv, ln = (np.ascontiguousarray(x) for x in ydist.local_csr(offsets, intervals, lengths, r0, r1))
bench.drop_shared(cx)
left = [f for f in os.listdir("/dev/shm") if f.startswith("yacrd_bench_csr_%s_" % os.environ["MASTER_PORT"])] if os.path.isdir("/dev/shm") else []
sizes = [None] * world
d.all_gather_object(sizes, (r1 - r0, int(off[-1]), same, isinstance(offsets, np.memmap)))
if rank == 0:
    ok = (sum(s[0] for s in sizes) == 3000 and sum(s[1] for s in sizes) == 180000 and all(s[2] for s in sizes)
          and not sizes[0][3] and all(s[3] for s in sizes[1:]) and not left)
    print("RESULT", "OK" if ok else "MISMATCH", sizes, left)
d.destroy_process_group()
"""


def test_bench_input_without_room_to_share_is_generated_by_every_rank(tmp_path):
    """No directory with room for the input (a container's 64 MB /dev/shm and a full disk): rank 0 says so and every rank
    generates the input for itself — the same arrays, nothing mapped, nothing left behind."""
    script = tmp_path / "sworker.py"
    script.write_text(SHARED_WORKER.replace("not sizes[0][3] and all(s[3] for s in sizes[1:])", "not any(s[3] for s in sizes)"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   YACRD_BENCH_SHARE_DIRS="/nonexistent-yacrd-dir")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RESULT OK" in outs[0][0], outs


def test_bench_input_is_generated_once_per_node(tmp_path):
    script = tmp_path / "sworker.py"
    script.write_text(SHARED_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RESULT OK" in outs[0][0], outs


RANGE_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle, yacrd_amd
from yacrd_amd import dist as ydist, host
# what yacrd_engines_ingest_overlaps does (gpu_paf.hip), with Python standing in for the device: every rank parses a byte
# RANGE of the text, cut on chunk boundaries — a line belongs to the range it starts in —, numbers the reads it saw; the
# ranges' read lists (name, first position, first length, intervals) meet, the reads of the whole file are numbered by
# first appearance, a read's length is the one seen first; the reads are dealt out as contiguous ranges of numbers balanced
# by intervals, every rank gets the halves of ALL ranges' records that name its reads, sweeps them; end to end the results
# must be the whole file's.  (The chunk is 64 KiB here instead of 4 MiB so that a small file has several.)
CHUNK = 64 << 10
paf = os.path.join(sys.argv[2], "r.paf")
rank, local_rank, world = ydist.env_rank()
d = ydist.init(backend="gloo")
if rank == 0:
    host.synth_paf(host.SYNTH_ONT, 2500, 50000, 20250930, paf)
d.barrier()
text = open(paf, "rb").read()
n = len(text)
per = ((n + CHUNK - 1) // CHUNK + world - 1) // world
B = [min(n, r * per * CHUNK) for r in range(world + 1)]
lo, hi = B[rank], B[rank + 1]
p = lo
if lo and text[lo - 1:lo] != b"\n":          # the range begins inside a line: that line is the range's before
    p = text.index(b"\n", lo) + 1
mine = {}                                    # name -> [first position * 2 + side, first length, [(s, e) ...]]
while p < hi:
    q = text.find(b"\n", p)
    q = n if q < 0 else q
    f = text[p:q].rstrip(b"\r").split(b"\t")
    if len(f) >= 9:
        for side, (k, l, s, e) in enumerate(((f[0], f[1], f[2], f[3]), (f[5], f[6], f[7], f[8]))):
            rec = mine.setdefault(k, [2 * p + side, int(l), []])
            rec[2].append((int(s), int(e)))
    p = q + 1
lists = [None] * world
d.all_gather_object(lists, [(k, v[0], v[1], len(v[2])) for k, v in mine.items()])
merged = {}                                  # name -> [first position, length seen there, intervals in the whole file]
for lst in lists:
    for k, fp, ln, cnt in lst:
        m = merged.setdefault(k, [fp, ln, 0])
        if fp < m[0]:
            m[0], m[1] = fp, ln
        m[2] += cnt
order = sorted(merged, key=lambda k: merged[k][0])        # first-appearance numbering of the whole file
number = {k: g for g, k in enumerate(order)}
counts = np.array([merged[k][2] for k in order], np.int64)
total, cuts, run, o = int(counts.sum()), [0], 0, 1
for g in range(len(order)):
    while o < world and run >= (total * o + world - 1) // world:
        cuts.append(g); o += 1
    run += int(counts[g])
cuts += [len(order)] * (world + 1 - len(cuts))
out = [dict() for _ in range(world)]         # the exchange: the halves that name a rank's reads, to that rank
for k, v in mine.items():
    g = number[k]
    owner = max(r for r in range(world) if cuts[r] <= g)
    out[owner][g] = v[2]
boxes = [None] * world
d.all_gather_object(boxes, out)
r0, r1 = cuts[rank], cuts[rank + 1]
per_read = [[] for _ in range(r1 - r0)]
for src in range(world):                     # (rank order = file order, as in the reference)
    for g, ivs in boxes[src][rank].items():
        per_read[g - r0].extend(ivs)
off = np.zeros(r1 - r0 + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in per_read])
iv = np.array([x for r in per_read for x in r], dtype=np.uint32).reshape(-1, 2)
ln = np.array([merged[order[g]][1] for g in range(r0, r1)], np.uint64)
part = oracle.run(off, iv, ln, 4, 0.4)
full = ydist.gather_results(d, part)
if rank == 0:
    reads = oracle.parse_paf(text.decode())
    names, woff, wiv, wln = oracle.to_csr(reads)
    want = oracle.run(woff, wiv, wln, 4, 0.4)
    ok = [k.decode() for k in order] == list(names) and [merged[k][1] for k in order] == [int(x) for x in wln]
    ok = ok and all(np.array_equal(a, b) for a, b in zip(full, want)) and len(set(B)) == world + 1
    print("RESULT", "OK" if ok else "MISMATCH", B, cuts)
d.destroy_process_group()
"""


def test_two_rank_text_range_partition_gloo(tmp_path):
    script = tmp_path / "rworker.py"
    script.write_text(RANGE_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RESULT OK" in outs[0][0], outs
