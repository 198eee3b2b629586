"""world_size-2 gloo test of the frame-sharded data-parallel step (vhap_amd.dist): the sharded,
all-reduced gradients and the batch-global photometric normaliser reproduce the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _energy(params, sample, normaliser=None, world=1):
    """A stand-in energy with exactly the coupling structure of the tracker's: a 'photometric' sum divided
    by a batch-global count, per-frame mean terms, and shared-parameter regularisers."""
    shared, per_frame = params
    ts = sample["timestep_index"]
    pred = per_frame[ts] @ shared                                   # [b, 3]
    err = (sample["rgb"] - pred).abs().sum()
    n = (sample["rgb"] > 0.3).sum().float()
    if normaliser is not None:
        n = normaliser(n) / world
    return 30.0 * err / n + 0.03 * (per_frame[ts] ** 2).mean() + 0.3 * (shared ** 2).mean()


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vhap_amd.dist import FrameShardContext
    ctx = FrameShardContext()
    g = torch.Generator().manual_seed(0)
    shared = torch.randn(5, 3, generator=g).requires_grad_()
    per_frame = torch.randn(8, 5, generator=g).requires_grad_()
    sample = {"rgb": torch.rand(8, 3, generator=g), "timestep_index": np.arange(8), "tag": "x"}
    local = ctx.shard_sample(sample)
    assert len(local["timestep_index"]) == 4 and local["tag"] == "x"
    E = _energy((shared, per_frame), local, normaliser=ctx.all_reduce_sum, world=world)
    E.backward()
    ctx.average_gradients([shared, per_frame])
    Es = torch.tensor([float(E)])
    dist.all_reduce(Es)
    ret[rank] = (shared.grad.clone(), per_frame.grad.clone(), float(Es) / world)
    with pytest.raises(ValueError):
        ctx.shard_sample({"rgb": torch.rand(7, 3), "timestep_index": np.arange(7)})
    dist.destroy_process_group()


def test_two_rank_sharded_step_matches_single_process():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    shared = torch.randn(5, 3, generator=g).requires_grad_()
    per_frame = torch.randn(8, 5, generator=g).requires_grad_()
    sample = {"rgb": torch.rand(8, 3, generator=g), "timestep_index": np.arange(8)}
    E = _energy((shared, per_frame), sample)
    E.backward()
    for r in range(world):
        gs, gp, Er = ret[r]
        assert torch.allclose(gs, shared.grad, atol=1e-6)
        assert torch.allclose(gp, per_frame.grad, atol=1e-6)
        assert abs(Er - float(E)) < 1e-5
    assert torch.equal(ret[0][0], ret[1][0])                       # replicas see bit-identical gradients
