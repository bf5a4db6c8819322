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
