"""Data-parallel path: process groups over gloo on 127.0.0.1 (world_size 2, 4 and 8).

* CPU test: the bucket contract of dist.py - every rank contributes grads / loss terms of its row shard
  normalised by the GLOBAL batch; one all_reduce(SUM) equals the single-process result on the concatenated
  batch.  The per-rank producer here is the oracle (test infrastructure); the product wires the same bucket to
  the HIP kernels (FusedTrainer).
* GPU test: the real FusedTrainer on two ranks sharing cuda:0 (gloo moves the bucket), 3 optimiser steps,
  against a single-process FusedTrainer on the full batch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import vpc_amd as vpc
from oracle import vae_oracle as O

L = 10


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(B, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
    return x, mask, mask_p, torch.randn(B, L, generator=g), torch.randn(B, L, generator=g)


def _cpu_worker(rank, world, port, d, B, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, w, _ = vpc.dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    params = O.init_params(d, L, seed=5)
    flat = torch.cat([params[k].reshape(-1) for k in O.PARAM_KEYS])
    if rank != 0:
        flat = torch.zeros_like(flat)  # replicas start from rank 0's weights
    vpc.dp.broadcast_parameters(flat)
    off, p = 0, {}
    for k in O.PARAM_KEYS:
        n = params[k].numel()
        p[k] = flat[off:off + n].view_as(params[k]); off += n
    x, mask, mask_p, eq, ep = _inputs(B, d)
    lo, hi = vpc.dp.shard_rows(B, rank, world)
    loss_l, grads_l, _, _ = O.closed_form_reg_step(p, L, x[lo:hi], mask[lo:hi], mask_p[lo:hi], eq[lo:hi], ep[lo:hi],
                                                   alpha=0.8, beta=0.9)
    scale = (hi - lo) / B  # local mean -> contribution to the global mean (seeds carry 1/B_global)
    bucket = torch.cat([torch.from_numpy(np.concatenate([grads_l[k].reshape(-1) for k in O.PARAM_KEYS]) * scale),
                        torch.tensor([loss_l * scale], dtype=torch.float64)])
    vpc.dp.allreduce_bucket(bucket)
    if rank == 0:
        out.put(bucket.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("d,B,world", [(14, 37, 2), (128, 64, 2), (128, 203, 8)])
def test_bucket_allreduce_equals_single_process_gloo(d, B, world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, world, port, d, B, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    params = O.init_params(d, L, seed=5)
    x, mask, mask_p, eq, ep = _inputs(B, d)
    loss, grads, _, _ = O.closed_form_reg_step(params, L, x, mask, mask_p, eq, ep, alpha=0.8, beta=0.9)
    want = np.concatenate([np.concatenate([grads[k].reshape(-1) for k in O.PARAM_KEYS]), [loss]])
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.max(np.abs(want)))


def test_shard_rows_partition():
    for n in (1, 7, 128, 65536, 65537):
        for w in (1, 2, 3, 8):
            cuts = [vpc.dp.shard_rows(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


# ----------------------------------------------------------------------------------------------- GPU, 2 ranks
def _gpu_worker(rank, world, port, d, B, steps, out, device_draws=False, precision="f32"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    vpc.dp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    params = O.init_params(d, L, seed=5)
    m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": B, "patience": 1}, "dp", "kl_reg")
    sd = m.state_dict(); sd.update({k: v.clone() for k, v in params.items()}); m.load_state_dict(sd)
    m.to(dev)
    vpc.dp.broadcast_parameters(m.flatten_parameters())
    tr = vpc.FusedTrainer(m, world_size=world, rank=rank, precision=precision)
    x, mask, mask_p, eq, ep = _inputs(B, d)
    lo, hi = vpc.dp.shard_rows(B, rank, world)
    losses = []
    for i in range(steps):
        if device_draws:  # mask_p / eps from the shared-seed Philox stream, counters keyed by the GLOBAL row
            tr.step(x[lo:hi].to(dev), mask[lo:hi].to(dev), alpha=0.8, beta=0.9, epoch=i + 1, global_batch=B, row_lo=lo)
        else:
            tr.step(x[lo:hi].to(dev), mask[lo:hi].to(dev), mask_p[lo:hi].to(dev), eq[lo:hi].to(dev), ep[lo:hi].to(dev),
                    alpha=0.8, beta=0.9, epoch=i + 1, global_batch=B, row_lo=lo)
        losses.append(tr.loss_value())
    if rank == 0:
        out.put((losses, m._flat.cpu().numpy(), tr.epoch_total()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_fused_trainer_matches_single_process():
    d, B, steps = 128, 512, 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, d, B, steps, out)) for r in range(2)]
    for p in procs:
        p.start()
    losses2, flat2, total2 = out.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    tr, m = _single_trainer(d, B, dev)
    x, mask, mask_p, eq, ep = _inputs(B, d)
    for i in range(steps):
        tr.step(x.to(dev), mask.to(dev), mask_p.to(dev), eq.to(dev), ep.to(dev), alpha=0.8, beta=0.9, epoch=i + 1)
        assert abs(tr.loss_value() - losses2[i]) <= 2e-6 * abs(losses2[i])
    flat1 = m._flat.cpu().numpy()
    assert np.max(np.abs(flat1 - flat2)) <= 2e-6 * np.max(np.abs(flat1))
    # the epoch total every rank reports is the GLOBAL one (sum of the all-reduced step losses)
    assert abs(tr.epoch_total() - total2) <= 1e-5 * abs(total2)


@pytest.mark.gpu
def test_two_rank_bf16_step_kernel_matches_single_process(monkeypatch):
    """The whole-step bf16 kernel under data parallelism (throughput workgroup shape forced: VPC_TILE=128): two ranks on shards
    of 256 rows = the single process on 512 rows up to fp32 reduction order - the operand rounding is per row and identical -
    incl. the Adam updates (the compact image is re-packed lazily after the all-reduced update)."""
    monkeypatch.setenv("VPC_TILE", "128")
    d, B, steps = 128, 512, 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, d, B, steps, out, False, "bf16")) for r in range(2)]
    for p in procs:
        p.start()
    losses2, flat2, total2 = out.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    tr, m = _single_trainer(d, B, dev, precision="bf16")
    x, mask, mask_p, eq, ep = _inputs(B, d)
    for i in range(steps):
        tr.step(x.to(dev), mask.to(dev), mask_p.to(dev), eq.to(dev), ep.to(dev), alpha=0.8, beta=0.9, epoch=i + 1)
        assert tr._used_step_fused
        assert abs(tr.loss_value() - losses2[i]) <= 1e-5 * abs(losses2[i]), (i, tr.loss_value(), losses2[i])
    flat1 = m._flat.cpu().numpy()
    assert np.max(np.abs(flat1 - flat2)) <= 1e-4 * np.max(np.abs(flat1))
    assert abs(tr.epoch_total() - total2) <= 1e-5 * abs(total2)


def _single_trainer(d, B, dev, seed=0, precision="f32"):
    params = O.init_params(d, L, seed=5)
    m = vpc.Reg_VAE(d, 500, 10, L, {"batch_size": B, "patience": 1}, "dp", "kl_reg")
    sd = m.state_dict(); sd.update({k: v.clone() for k, v in params.items()}); m.load_state_dict(sd)
    m.to(dev)
    return vpc.FusedTrainer(m, seed=seed, precision=precision), m


@pytest.mark.gpu
def test_four_rank_device_draws_match_single_process():
    """World size 4 over gloo (the GPU box admits at most 6 GPU processes: 4 ranks + this one), mask_p and eps drawn ON
    THE DEVICE with one shared seed and Philox counters keyed by the global row: losses, the
    epoch total and the parameters after 3 Adam steps equal the single-process run on the concatenated batch."""
    d, B, steps, world = 128, 4 * 1024, 3, 4
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, d, B, steps, out, True)) for r in range(world)]
    for p in procs:
        p.start()
    losses4, flat4, total4 = out.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    tr, m = _single_trainer(d, B, dev)
    x, mask, _, _, _ = _inputs(B, d)
    for i in range(steps):
        tr.step(x.to(dev), mask.to(dev), alpha=0.8, beta=0.9, epoch=i + 1)
        assert abs(tr.loss_value() - losses4[i]) <= 5e-6 * abs(losses4[i]), (i, tr.loss_value(), losses4[i])
    flat1 = m._flat.cpu().numpy()
    assert np.max(np.abs(flat1 - flat4)) <= 3e-5 * np.max(np.abs(flat1))
    assert abs(tr.epoch_total() - total4) <= 1e-5 * abs(total4)


@pytest.mark.gpu
def test_eight_shards_of_the_headline_batch_match_single_process():
    """Config 4's decomposition, B = 65 536 sharded 8 x 8 192, with device-side draws.  Eight
    GPU processes do not fit the box's process limit, so the eight ranks run one after the other in this process:
    each is a FusedTrainer with its own rank / row_lo, the shared seed and global_batch = 65 536, and the all-reduce
    is a sum of the eight buckets.  Checks (i) every shard draws exactly the mask_p bytes and eps values the
    single-process step draws for those rows (bit-equal: the result does not depend on the world size) and (ii) the
    summed bucket [grads | loss terms] equals the single-process bucket."""
    d, B, world = 128, 65536, 8
    dev = torch.device("cuda:0")
    x, mask, _, _, _ = _inputs(B, d, seed=11)
    x, mask = x.to(dev), mask.to(dev)
    tr1, _ = _single_trainer(d, B, dev, seed=7)
    tr1.step(x, mask, alpha=0.8, beta=0.9, update=False)
    want = tr1.bucket.clone()
    mp1, eps1 = tr1.mask_p_buf.clone(), tr1.eps_buf[:2].clone()
    assert 0.45 < float(mp1.float().mean()) < 0.53  # 0.7 * 0.7
    total = torch.zeros_like(want, dtype=torch.float64)
    for r in range(world):
        lo, hi = vpc.dp.shard_rows(B, r, world)
        trr, _ = _single_trainer(d, B, dev, seed=7)
        trr.rank = r
        trr.step(x[lo:hi], mask[lo:hi], alpha=0.8, beta=0.9, update=False, global_batch=B, row_lo=lo)
        assert torch.equal(trr.mask_p_buf, mp1[lo:hi]), r
        assert torch.equal(trr.eps_buf[:2], eps1[:, lo:hi]), r
        total += trr.bucket.double()
    n = tr1.lay.n_params
    g1, g8 = want[:n].double(), total[:n]
    assert float((g1 - g8).abs().max()) <= 2e-5 * float(g1.abs().max())
    assert abs(float(want[n]) - float(total[n])) <= 2e-6 * abs(float(want[n]))


# ----------------------------------------------------------------------------------------------- MNAR path, 2 ranks
def _nm_inputs(B, d, K, Ld, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, d, generator=g)
    mask = (torch.rand(B, d, generator=g) < 0.7).float()
    mask_p = mask * (torch.rand(B, d, generator=g) < 0.5).float()
    eps = torch.randn(2, B, K, Ld, generator=g)
    return x, mask, mask_p, eps


def _nm_model(d, K, Ld, B):
    torch.manual_seed(17)
    return vpc.REG_notMIWAE_v2(d, 128, 10, Ld, {"batch_size": B, "patience": 1}, K, 1)


def _nm_gpu_worker(rank, world, port, d, B, K, Ld, steps, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    vpc.dp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    m = _nm_model(d, K, Ld, B).to(dev)
    vpc.dp.broadcast_parameters(m.flatten_parameters())
    tr = vpc.NMTrainer(m, world_size=world, rank=rank)
    x, mask, mask_p, eps = _nm_inputs(B, d, K, Ld)
    lo, hi = vpc.dp.shard_rows(B, rank, world)
    losses = []
    for i in range(steps):
        tr.step(x[lo:hi].to(dev), mask[lo:hi].to(dev), mask_p[lo:hi].to(dev), eps[:, lo:hi].contiguous().to(dev),
                alpha=0.5, global_batch=B, row_lo=lo)
        losses.append(tr.loss_value())
    if rank == 0:
        out.put((losses, m._flat.cpu().numpy(), tr.epoch_total()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_mnar_trainer_matches_single_process():
    """NMTrainer under data parallelism: ONE all-reduce of [grads | loss], every term normalised by the global batch,
    equals the single-process step on the concatenated batch (uneven shards: 300 rows -> 150 + 150, K = 5)."""
    d, B, K, Ld, steps = 40, 300, 5, 6, 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nm_gpu_worker, args=(r, 2, port, d, B, K, Ld, steps, out)) for r in range(2)]
    for p in procs:
        p.start()
    losses2, flat2, total2 = out.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    m = _nm_model(d, K, Ld, B).to(dev)
    tr = vpc.NMTrainer(m)
    x, mask, mask_p, eps = _nm_inputs(B, d, K, Ld)
    for i in range(steps):
        tr.step(x.to(dev), mask.to(dev), mask_p.to(dev), eps.to(dev), alpha=0.5)
        assert abs(tr.loss_value() - losses2[i]) <= 5e-6 * abs(losses2[i]), (i, tr.loss_value(), losses2[i])
    flat1 = m._flat.cpu().numpy()
    # sharding changes the fp32 summation order of the weight gradients; Adam's m / sqrt(v) turns a 1e-7 relative
    # difference of a tiny gradient into up to lr = 1e-3 per step, so parameters are compared at 3e-5 of their scale
    assert np.max(np.abs(flat1 - flat2)) <= 3e-5 * np.max(np.abs(flat1))
    assert abs(tr.epoch_total() - total2) <= 1e-5 * abs(total2)  # every rank's epoch total is the global loss


# ----------------------------------------------------------------------------------------------- EDDI family, 2 ranks
def _eddi_inputs(B, d, Ld, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, d, generator=g)
    mask = torch.rand(B, d, generator=g) < 0.7
    mask_p = mask & (torch.rand(B, d, generator=g) < 0.7)
    return x, mask, mask_p, torch.randn(B, Ld, generator=g), torch.randn(B, Ld, generator=g)


def _eddi_model(d, K, Ld, B):
    torch.manual_seed(23)
    return vpc.Reg_EDDI(d, 500, K, Ld, {"batch_size": B, "patience": 1}, "dp", "kl_reg")


def _eddi_gpu_worker(rank, world, port, d, B, K, Ld, steps, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    vpc.dp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    m = _eddi_model(d, K, Ld, B).to(dev)
    vpc.dp.broadcast_parameters(m.flatten_parameters())
    tr = vpc.EDDITrainer(m, world_size=world, rank=rank)
    x, mask, mask_p, eq, ep = _eddi_inputs(B, d, Ld)
    lo, hi = vpc.dp.shard_rows(B, rank, world)
    losses = []
    for i in range(steps):
        tr.step(x[lo:hi].to(dev), mask[lo:hi].to(dev), mask_p[lo:hi].to(dev), eq[lo:hi].to(dev), ep[lo:hi].to(dev),
                epoch=i + 1, alpha=0.5, global_batch=B, row_lo=lo)
        losses.append(tr.loss_value())
    if rank == 0:
        out.put((losses, m._flat.cpu().numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_eddi_trainer_matches_single_process():
    d, B, K, Ld, steps = 40, 257, 20, 10, 3  # uneven shards: 129 + 128 rows
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eddi_gpu_worker, args=(r, 2, port, d, B, K, Ld, steps, out)) for r in range(2)]
    for p in procs:
        p.start()
    losses2, flat2 = out.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    m = _eddi_model(d, K, Ld, B).to(dev)
    tr = vpc.EDDITrainer(m)
    x, mask, mask_p, eq, ep = _eddi_inputs(B, d, Ld)
    for i in range(steps):
        tr.step(x.to(dev), mask.to(dev), mask_p.to(dev), eq.to(dev), ep.to(dev), epoch=i + 1, alpha=0.5)
        assert abs(tr.loss_value() - losses2[i]) <= 5e-6 * abs(losses2[i]), (i, tr.loss_value(), losses2[i])
    flat1 = m._flat.cpu().numpy()
    assert np.max(np.abs(flat1 - flat2)) <= 3e-5 * np.max(np.abs(flat1))


# ----------------------------------------------------------------------------------------------- RCCL on the compute stream
@pytest.mark.gpu
def test_rccl_flat_allreduce_on_compute_stream_single_rank():
    """The C-ABI binding of ncclAllReduce (vpc_allreduce_flat, dist.FlatAllReduce) with a ONE-rank communicator - all a
    one-GPU box can host: the unique id / ncclCommInitRank / in-place all-reduce calls work, run on torch's current
    stream, and the data-parallel tail of the fused step (reduce_step -> all-reduce -> adam_step) through it equals the
    single fused launch bit for bit, eagerly and replayed from ONE captured HIP graph (the collective is a graph node).
    Multi-rank RCCL needs more than one GPU; the sharding arithmetic is covered by the gloo tests above."""
    dev = torch.device("cuda:0")
    coll = vpc.dp.FlatAllReduce(1, 0, dev)
    t = torch.arange(1000, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # issued on the CURRENT stream, whichever that is
        t.mul_(2.0)
        coll(t)
        t.add_(1.0)
    side.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32) * 2 + 1)
    d, B = 128, 1024
    x, mask, _, _, _ = _inputs(B, d, seed=21)
    x, mask = x.to(dev), mask.to(dev)
    tr0, m0 = _single_trainer(d, B, dev, seed=3)
    tr1, m1 = _single_trainer(d, B, dev, seed=3)
    tr1.collective, tr1._coll_ready, tr1.dp = coll, True, True
    tr2, m2 = _single_trainer(d, B, dev, seed=3)
    tr2.collective, tr2._coll_ready, tr2.dp = coll, True, True
    for i in range(4):
        tr0.step(x, mask, alpha=0.9, epoch=i + 1)
        tr1.step(x, mask, alpha=0.9, epoch=i + 1)
        tr2.step_graph(x, mask, alpha=0.9, epoch=1)
        assert tr0.loss_value() == tr1.loss_value()
    assert torch.equal(m0._flat, m1._flat)
    assert abs(tr0.epoch_total() - tr1.epoch_total()) < 1e-3
    # graph replay (steps 2..4 are replays; epoch fixed because beta_annealing is off): same parameters as eager
    tr3, m3 = _single_trainer(d, B, dev, seed=3)
    for i in range(4):
        tr3.step(x, mask, alpha=0.9, epoch=1)
    assert torch.equal(m2._flat, m3._flat)
    assert getattr(tr2, "_graph", None) is not None
    coll.close()
