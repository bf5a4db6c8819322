"""Process-per-GPU plumbing for the read-partitioned path: which reads a rank
owns, and the barrier / max-over-ranks used by bench.py.  There is no data-path collective:
reads are independent, ranks only synchronise for timing."""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """Initialise torch.distributed when WORLD_SIZE > 1 (nccl == RCCL on ROCm, gloo on CPU)."""
    rank, local_rank, world = env_rank()
    if world == 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if backend is None:
        backend = "nccl" if device is not None else "gloo"
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", device_id=device)
            return dist
        except Exception as ex:  # said out loud; callers report dist.get_backend() (bench.py: config.torch_distributed_backend)
            import sys
            print("yacrd_amd.dist: nccl (RCCL) process group failed (%r): falling back to gloo for the barrier / max over ranks" % (ex,),
                  file=sys.stderr, flush=True)
    dist.init_process_group("gloo")
    return dist


def shard(offsets, rank, world, partition_fn):
    """Contiguous read range [r0, r1) of `rank`, balanced by interval count."""
    cuts = partition_fn(offsets, world)
    return int(cuts[rank]), int(cuts[rank + 1])


def local_csr(offsets, intervals, lengths, r0, r1):
    """Slice a CSR to reads [r0, r1) with offsets rebased to 0."""
    base = int(offsets[r0])
    off = (np.asarray(offsets[r0:r1 + 1]) - np.uint64(base)).astype(np.uint64)
    iv = np.asarray(intervals).reshape(-1, 2)[base:int(offsets[r1])]
    return off, iv, np.asarray(lengths[r0:r1])


def max_over_ranks(dist, value, device=None):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(dist, part):
    """Concatenate per-rank (bad_offsets, bad_regions, read_type) in rank order (host side)."""
    if dist is None:
        return part
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, part)
    bo = [np.zeros(1, dtype=np.uint64)]
    base = np.uint64(0)
    for p in parts:
        bo.append(p[0][1:] + base)
        base = base + p[0][-1]
    return (np.concatenate(bo), np.concatenate([p[1] for p in parts], axis=0),
            np.concatenate([p[2] for p in parts]))
