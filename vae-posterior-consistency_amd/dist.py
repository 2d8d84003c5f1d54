"""Data-parallel plumbing: one process per GPU, rows sharded contiguously, ONE all-reduce per step.

The model is 150 KB, so the per-step exchange is a single latency-bound all-reduce of the flat bucket
[37 548 gradients | 9 loss terms].  Every rank normalises by the GLOBAL batch so that SUM over ranks equals the
single-process result on the concatenated batch (src/models/VAE.py:452 divides by x.shape[0]); Philox counters of the
per-step draws are keyed by the global row (fused.py), so results do not depend on the world size.

Two carriers for the bucket:
  * `FlatAllReduce` (default when the process group is NCCL): ncclAllReduce bound straight from librccl through the C ABI
    (vpc_allreduce_flat) and issued on the COMPUTE stream - no hop to torch's communication stream and back, and the
    whole tail reduce_step -> all-reduce -> adam_step is capturable in one HIP graph (FusedTrainer.step_graph under data
    parallelism).  The communicator is created with ncclCommInitRank; the 128-byte unique id travels from rank 0 through
    the torch.distributed process group that torchrun set up.
  * torch.distributed.all_reduce on the process group (gloo on CPU boxes / rehearsals on one GPU, or NCCL when
    VPC_DP_COLLECTIVE=torch).
"""
from __future__ import annotations

import ctypes as C
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


class FlatAllReduce:
    """In-place sum of a flat fp32 CUDA tensor over the ranks with ncclAllReduce (RCCL) on torch's CURRENT stream."""

    def __init__(self, nranks: int, rank: int, device, group=None):
        from . import _lib as L
        self._L = L
        lib = L.lib()
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            if lib.vpc_rccl_unique_id(C.cast(buf, C.c_void_p)) == 0:  # on failure the all-zero id tells every rank
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if nranks > 1:  # ship the id through the existing process group (device tensor for NCCL, host tensor for gloo)
            dev_pg = dist.get_backend(group) == "nccl"
            t = uid.to(device) if dev_pg else uid
            dist.broadcast(t, src=0, group=group)
            uid = t.cpu()
        if not bool(uid.any()):
            raise L.VpcError("ncclGetUniqueId failed on rank 0 (librccl not loadable?)")
        raw = (C.c_char * 128).from_buffer_copy(bytes(uid.numpy().tobytes()))
        comm = C.c_void_p()
        with torch.cuda.device(device):
            L.check(lib.vpc_rccl_comm_init(C.cast(raw, C.c_void_p), nranks, rank, C.byref(comm)), "vpc_rccl_comm_init")
        self.comm, self.nranks, self.rank = comm, nranks, rank

    def __call__(self, bucket: torch.Tensor):
        L = self._L
        L.require_cuda(bucket)
        if bucket.dtype != torch.float32 or not bucket.is_contiguous():
            raise L.VpcError("FlatAllReduce takes a contiguous fp32 tensor")
        L.check(L.lib().vpc_allreduce_flat(self.comm, L.ptr(bucket), bucket.numel(), L.stream_ptr()), "vpc_allreduce_flat")
        return bucket

    def close(self):
        if self.comm:
            self._L.lib().vpc_rccl_comm_destroy(self.comm)
            self.comm = None


def make_collective(world_size: int, rank: int, device, group=None):
    """The carrier of the per-step bucket: FlatAllReduce (RCCL on the compute stream) when the process group is NCCL
    (and VPC_DP_COLLECTIVE != 'torch'), else None = torch.distributed.all_reduce on the group."""
    if world_size <= 1 or not dist.is_initialized():
        return None
    if os.environ.get("VPC_DP_COLLECTIVE", "rccl") == "torch" or dist.get_backend(group) != "nccl":
        return None
    # all ranks must take the same carrier: agree on success before using the RCCL communicator
    coll, ok = None, 1.0
    try:
        coll = FlatAllReduce(world_size, rank, device, group)
    except Exception as e:  # librccl missing / communicator creation failed on this rank
        import sys
        print(f"[vpc] RCCL communicator unavailable on rank {rank} ({e}); using torch.distributed.all_reduce", file=sys.stderr)
        ok = 0.0
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if float(flag.item()) < 0.5:
        if coll is not None:
            coll.close()
        return None
    # self-test against the process group: sum of (rank + 1) patterns through both carriers must agree
    probe = torch.arange(64, dtype=torch.float32, device=device) * (rank + 1)
    want = probe.clone()
    dist.all_reduce(want, op=dist.ReduceOp.SUM, group=group)
    coll(probe)
    torch.cuda.synchronize(device)
    good = torch.tensor([1.0 if torch.equal(probe, want) else 0.0], device=device)
    dist.all_reduce(good, op=dist.ReduceOp.MIN, group=group)
    if float(good.item()) < 0.5:
        import sys
        print(f"[vpc] RCCL self-test failed on rank {rank}; using torch.distributed.all_reduce", file=sys.stderr)
        coll.close()
        return None
    return coll


def allreduce_bucket(bucket: torch.Tensor, group=None, collective=None):
    """The step's single collective (sum): through `collective` (FlatAllReduce) when given, else torch.distributed.
    No-op for a single process."""
    if collective is not None:
        return collective(bucket)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    return bucket
