"""Multi-GPU plumbing: one process per GPU, frames / face batches sharded with no data-path
collective, and ONE exchange step -- an all-gather(v) of the per-rank (n_faces, 512) embeddings
(north_star).  The reference has no distributed code at all, so
there is no call pattern to mirror; backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU
tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns (rank, world_size, local_rank).  A single process without those variables is
    rank 0 of 1 and does not create a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for this rank (first n%world ranks get one
    extra).  Frames of a video batch, images of a directory and faces are all independent units
    (demo_video.py:186-188 destroys the queue per batch), so no halo or exchange is needed."""
    q, r = divmod(int(n_items), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def round_robin_batches(n_batches, rank, world):
    """Frame batch b goes to rank b % world."""
    return list(range(rank, int(n_batches), int(world)))


def all_gather_embeddings(emb, group=None):
    """all-gather(v) of (n_i, D) fp32 embeddings: a tiny count all-gather, then one padded
    all-gather of (max_n, D).  Returns (list of per-rank tensors in rank order, counts).
    With world size 1 (or no process group) it is the identity."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [emb], [int(emb.shape[0])]
    world = dist.get_world_size(group)
    n = torch.tensor([emb.shape[0]], dtype=torch.int64, device=emb.device)
    counts = torch.empty(world, dtype=torch.int64, device=emb.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = [int(c) for c in counts.cpu()]
    m = max(counts)
    d = emb.shape[1]
    pad = torch.zeros((m, d), dtype=emb.dtype, device=emb.device)
    pad[: emb.shape[0]] = emb
    out = torch.empty((world * m, d), dtype=emb.dtype, device=emb.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * m: r * m + counts[r]] for r in range(world)], counts


def all_gather_fixed(out, emb, async_op=False, group=None):
    """Equal-sized all-gather used by the steady-state benchmark loop: out is (world*n, D)."""
    return dist.all_gather_into_tensor(out, emb, group=group, async_op=async_op)
