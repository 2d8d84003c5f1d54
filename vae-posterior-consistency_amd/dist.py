"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The model is 150 KB, so the per-step exchange is ONE latency-bound all-reduce of the flat bucket
[37 548 gradients | 9 loss terms] issued on the compute stream by FusedTrainer.step.  Rows are sharded
contiguously; every rank normalises by the GLOBAL batch so that SUM over ranks equals the single-process
result on the concatenated batch (src/models/VAE.py:452 divides by x.shape[0]).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank).  Single process when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()  # rehearsals with more ranks than GPUs share devices (gloo only)
    if world > 1 and not dist.is_initialized():
        if backend is None:  # VPC_DIST_BACKEND=gloo: rehearse the multi-rank path on a single GPU
            backend = os.environ.get("VPC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row shard [lo, hi) of rank; shards differ by at most one row."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_parameters(flat: torch.Tensor, src=0, group=None, model=None):
    """Make every replica start from rank `src`'s weights.  Pass `model` so that its packed weight images are
    re-packed on the next forward (the flat buffer is written behind the parameters' backs)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    if model is not None and hasattr(model, "invalidate_images"):
        model.invalidate_images()


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def allreduce_bucket(bucket: torch.Tensor, group=None):
    """The step's single collective (sum).  No-op for a single process."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    return bucket
