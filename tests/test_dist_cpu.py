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
    out = [shared.grad.clone(), per_frame.grad.clone(), float(Es) / world]
    # a ragged batch (7 frames over 2 ranks) is not split: every rank fits the whole of it, and the averaged result is the single-process one
    shared.grad = per_frame.grad = None
    rag = {"rgb": sample["rgb"][:7], "timestep_index": np.arange(7)}
    local = ctx.shard_sample(rag)
    assert len(local["timestep_index"]) == 7
    E = _energy((shared, per_frame), local, normaliser=ctx.all_reduce_sum, world=world)
    E.backward()
    ctx.average_gradients([shared, per_frame])
    ret[rank] = tuple(out) + (shared.grad.clone(), per_frame.grad.clone(), float(E))
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
    g_shared, g_frame = shared.grad.clone(), per_frame.grad.clone()
    shared.grad = per_frame.grad = None
    E7 = _energy((shared, per_frame), {"rgb": sample["rgb"][:7], "timestep_index": np.arange(7)})
    E7.backward()
    for r in range(world):
        gs, gp, Er, gs7, gp7, Er7 = ret[r]
        assert torch.allclose(gs, g_shared, atol=1e-6)
        assert torch.allclose(gp, g_frame, atol=1e-6)
        assert abs(Er - float(E)) < 1e-5
        assert torch.allclose(gs7, shared.grad, atol=1e-6) and torch.allclose(gp7, per_frame.grad, atol=1e-6) and abs(Er7 - float(E7)) < 1e-5
    assert torch.equal(ret[0][0], ret[1][0])                       # replicas see bit-identical gradients


def _lmk_tracker(n_frames=4):
    """The product's own tracker on a CPU device with landmark targets from ground-truth parameters (landmark-only pipeline)."""
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.synthetic import make_flame_model, make_scene_params, make_texture, monocular_camera
    from vhap_amd.tracker import GlobalTracker
    model, topo = make_flame_model(0)
    cfg = BaseTrackingConfig()
    cfg.device = "cpu"
    cfg.exp.photometric = False
    cfg.model.tex_resolution = 16
    H = W = 64
    gt = make_scene_params(n_frames, seed=2, image_size=(H, W))
    g = lambda k: torch.from_numpy(gt[k]).float()
    tr = GlobalTracker(cfg, model, topo, make_texture(0, 16), {"rgb": torch.zeros(n_frames, 3, H, W), "lmk2d": torch.zeros(n_frames, 70, 3)})
    with torch.no_grad():
        _, lmks = tr.flame(g("shape")[None].expand(n_frames, -1), g("expr"), g("rotation"), g("neck_pose"), g("jaw_pose"), g("eyes_pose"),
                           g("translation"))
        K, RT = monocular_camera(n_frames, (H, W), float(gt["focal_length"][0]))
        ndc = tr.render.world_to_ndc(lmks, torch.from_numpy(RT).float(), torch.from_numpy(K).float(), (H, W), flip_y=True)
        tr.dataset["lmk2d"] = torch.stack([(ndc[..., 0] * 0.5 + 0.5) * W, (ndc[..., 1] * 0.5 + 0.5) * H, torch.ones(n_frames, lmks.shape[1])], dim=-1)
    return tr


_LMK_NAMES = ("shape", "expr", "rotation", "translation", "neck_pose", "jaw_pose", "eyes_pose", "focal_length")


def _lmk_fit(tr, sample, steps=4):
    stage = "lmk_init_all"
    opt = tr.configure_optimizer(tr.get_train_parameters(stage))
    logs = [float(tr.optimize_iter(dict(sample), opt, stage)["total"]) for _ in range(steps)]
    return logs, {k: getattr(tr, k).detach().clone() for k in _LMK_NAMES}


def _worker_tracker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vhap_amd import dist as vdist
    from vhap_amd.tracker import ShuffledBatches
    tr = _lmk_tracker()
    ctx = vdist.attach(tr)
    sample = ctx.shard_sample(tr.get_sample(np.arange(4)))
    assert list(sample["timestep_index"]) == ([0, 1] if rank == 0 else [2, 3])
    logs, params = _lmk_fit(tr, sample)
    Es = torch.tensor(logs)
    dist.all_reduce(Es)
    # the shuffled global-tracking batches: every rank draws the same permutation and takes its own slice of each batch
    torch.manual_seed(100 + rank)                                  # different local RNG states on purpose
    order = [list(np.asarray(b["timestep_index"])) for b in ShuffledBatches(tr, 2)]
    ret[rank] = ((Es / world).tolist(), params, order)
    dist.destroy_process_group()


def test_two_rank_tracker_landmark_fit_matches_single_process():
    """The TRACKER's own energy under frame sharding (not a stand-in): FlameTracker.compute_energy of a landmark stage on each rank's
    slice, FrameShardContext.average_gradients, Adam -- four optimiser steps reproduce the single-process fit on the whole batch
    (energies as the mean over ranks, every trained parameter to fp32 round-off), replicas stay bit-identical, and ShuffledBatches
    yields complementary slices of the SAME permutation on both ranks."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_tracker, args=(world, port, ret), nprocs=world, join=True)
    tr = _lmk_tracker()
    logs, params = _lmk_fit(tr, tr.get_sample(np.arange(4)))
    for a, b in zip(ret[0][0], logs):
        assert abs(a - b) <= 1e-5 * abs(b), (ret[0][0], logs)
    for k in _LMK_NAMES:
        assert torch.equal(ret[0][1][k], ret[1][1][k]), f"replicas disagree on {k}"
        d = float((ret[0][1][k] - params[k]).abs().max())
        assert d <= 2e-5 * max(1.0, float(params[k].abs().max())), (k, d)
        assert float(params[k].abs().max()) > 0
    o0, o1 = ret[0][2], ret[1][2]
    assert len(o0) == len(o1) == 2 and all(len(a) == 1 and len(b) == 1 for a, b in zip(o0, o1))
    assert sorted(t for b in o0 + o1 for t in b) == [0, 1, 2, 3]


def _lmk_multiview_tracker(n_views=4):
    """BASELINE config 4 in small: calibrated views of ONE timestep (NeRSemble-style: nersemble.py:22-42), landmark-only, CPU device.  The
    landmark targets are the projections of ground-truth landmarks through each view's camera."""
    from vhap_amd.config import nersemble_config
    from vhap_amd.synthetic import arc_cameras, make_flame_model, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = make_flame_model(0)
    cfg = nersemble_config()
    cfg.device = "cpu"
    cfg.exp.photometric = False
    cfg.model.tex_resolution = 16
    H, W = 64, 48
    gt = make_scene_params(1, seed=4, image_size=(H, W))
    g = lambda k: torch.from_numpy(gt[k]).float()
    K, RT = arc_cameras(n_views, (H, W))
    K, RT = torch.from_numpy(K).float(), torch.from_numpy(RT).float()
    data = {"rgb": torch.zeros(n_views, 3, H, W), "lmk2d": torch.zeros(n_views, 70, 3), "intrinsic": K, "extrinsic": RT,
            "timestep_index": torch.zeros(n_views, dtype=torch.long), "camera_index": torch.arange(n_views)}
    tr = GlobalTracker(cfg, model, topo, make_texture(0, 16), data)
    with torch.no_grad():
        _, lmks = tr.flame(g("shape")[None], g("expr"), g("rotation") * 0, g("neck_pose"), g("jaw_pose"), g("eyes_pose"), g("translation") * 0)
        ndc = tr.render.world_to_ndc(lmks.expand(n_views, -1, -1), RT, K, (H, W), flip_y=True)
        tr.dataset["lmk2d"] = torch.stack([(ndc[..., 0] * 0.5 + 0.5) * W, (ndc[..., 1] * 0.5 + 0.5) * H, torch.ones(n_views, lmks.shape[1])], dim=-1)
    return tr


_MV_NAMES = ("shape", "expr", "rotation", "translation", "neck_pose", "jaw_pose", "eyes_pose")


def _mv_fit(tr, sample, steps=4):
    stage = "lmk_init_all"
    opt = tr.configure_optimizer(tr.get_train_parameters(stage))
    logs = [float(tr.optimize_iter(dict(sample), opt, stage)["total"]) for _ in range(steps)]
    return logs, {k: getattr(tr, k).detach().clone() for k in _MV_NAMES}


def _worker_multiview(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vhap_amd import dist as vdist
    tr = _lmk_multiview_tracker()
    ctx = vdist.attach(tr)
    sample = ctx.shard_sample(tr.get_sample(np.array([0])))
    assert len(sample["timestep_index"]) == 2 and list(np.asarray(sample["timestep_index"])) == [0, 0]      # two views of the one timestep
    assert sample["intrinsic"].shape == (2, 3, 3)
    logs, params = _mv_fit(tr, sample)
    Es = torch.tensor(logs)
    dist.all_reduce(Es)
    ret[rank] = ((Es / world).tolist(), params)
    dist.destroy_process_group()


def test_two_rank_multiview_views_of_one_timestep_split_over_ranks():
    """BASELINE config 4's sharding (B/G split): the VIEWS of one timestep go to different ranks, so both ranks produce gradients
    for the SAME per-timestep parameter row -- the averaged gradient must equal the single-process one (the row's gradient is the mean over
    all views either way), and the replicas must stay bit-identical."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_multiview, args=(world, port, ret), nprocs=world, join=True)
    tr = _lmk_multiview_tracker()
    assert tr.n_timesteps == 1 and tr.calibrated
    logs, params = _mv_fit(tr, tr.get_sample(np.array([0])))
    assert logs[-1] < logs[0]
    for a, b in zip(ret[0][0], logs):
        assert abs(a - b) <= 1e-4 * abs(b), (ret[0][0], logs)         # (fp32: the two runs sum the same terms in a different order)
    for k in _MV_NAMES:
        assert torch.equal(ret[0][1][k], ret[1][1][k]), f"replicas disagree on {k}"
        d = float((ret[0][1][k] - params[k]).abs().max())
        assert d <= 5e-5 * max(1.0, float(params[k].abs().max())), (k, d)     # measured <= 1.1e-5 (fp32 summation order + Adam)


def _worker_tex(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vhap_amd.dist import FrameShardContext
    ctx = FrameShardContext()
    T = 32
    g = torch.Generator().manual_seed(7 + rank)
    grad = torch.randn(T, T, 3, generator=g)                       # this rank's level-0 texture gradient (HWC, like the kernels')
    tex = torch.arange(3 * T * T, dtype=torch.float32).reshape(3, T, T).clone()      # the replicated parameter (CHW)
    n = T // world
    strip = torch.zeros(n, T, 3)
    ctx.reduce_scatter_mean(grad.view(-1).clone(), strip.view(-1)).wait()
    # "Adam" on this rank's rows only, then the rows travel
    tex[:, rank * n:(rank + 1) * n] -= 0.5 * strip.permute(2, 0, 1)
    for w in ctx.all_gather_rows(tex, rank * n, n, async_op=True):
        w.wait()
    ret[rank] = (grad, tex)
    dist.destroy_process_group()


def test_two_rank_sharded_texture_update_collectives():
    """The sharded texture update (reduce-scatter of the level-0 gradient -> each rank updates its rows -> all-gather of the rows) on the
    CPU over gloo equals: mean gradient, update of the whole texture, on every rank."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_tex, args=(world, port, ret), nprocs=world, join=True)
    T = 32
    mean = 0.5 * (ret[0][0] + ret[1][0])
    want = torch.arange(3 * T * T, dtype=torch.float32).reshape(3, T, T) - 0.5 * mean.permute(2, 0, 1)
    assert torch.equal(ret[0][1], ret[1][1]), "replicas disagree"
    assert torch.allclose(ret[0][1], want, rtol=0, atol=1e-6)


def _worker_probe(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("VHAP_TEX_SHARDED", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vhap_amd.dist import FrameShardContext
    ctx = FrameShardContext()
    ok = ctx.probe()
    # a library that refuses a collective (every rank raises at the same call: an unsupported op / dtype is refused before anything goes
    # on the wire) selects the replicated texture update, and nobody dies
    def boom(*a, **k):
        raise RuntimeError("collective refused")
    ctx.reduce_scatter_mean = boom
    ok2 = ctx.probe()
    ret[rank] = (ok, ok2, ctx.tex_sharded_ok, ctx.tex_sharded_usable(), os.environ.get("VHAP_TEX_SHARDED"), ctx.sharded)
    dist.destroy_process_group()


def test_collective_probe_agrees_on_a_fallback():
    """FrameShardContext.probe(): every collective of the sharded texture update once on small tensors; a refusal selects the all-reduce +
    replicated-finish form on every rank (bench.py then prints a note instead of dying: VERDICT r4 item 1d).  The verdict lives on the
    context -- GraphedStep asks tex_sharded_usable() before it captures -- and no longer in the process environment (round-5 advisor)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_probe, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        assert ret[r] == (True, False, False, False, None, True), ret[r]


def test_forced_one_rank_group(monkeypatch):
    """VHAP_FORCE_DIST=1: a single process gets a world-size-1 group and its context reports `sharded` (the four-plan step with real
    collectives on the one GPU a developer has -- the GPU half is tests/test_dist_gpu.py)."""
    from vhap_amd import dist as vdist
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    assert vdist.init_from_env("gloo") == (0, 1, 0) and not dist.is_initialized()
    monkeypatch.setenv("VHAP_FORCE_DIST", "1")
    try:
        assert vdist.init_from_env("gloo") == (0, 1, 0) and dist.is_initialized() and dist.get_world_size() == 1
        ctx = vdist.FrameShardContext()
        assert ctx.sharded and ctx.world_size == 1
        out = torch.empty(8)
        ctx.reduce_scatter_mean(torch.arange(8.0), out, async_op=True).wait()
        assert torch.equal(out, torch.arange(8.0))
        monkeypatch.setenv("VHAP_FORCE_DIST", "0")
        assert not vdist.FrameShardContext().sharded
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        os.environ.pop("MASTER_PORT", None)


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside a torchrun environment must become a torch.distributed.run launch of N ranks (VERDICT r4
    missing 1: it exited with an error); inside one (WORLD_SIZE set) it must not re-launch."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._gpus_arg(["--steps", "3", "--gpus", "8"]) == 8 and bench._gpus_arg(["--gpus=4"]) == 4 and bench._gpus_arg([]) == 1
    seen = {}
    monkeypatch.setattr(os, "execv", lambda exe, argv: seen.update(exe=exe, argv=argv))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2", "--config", "5"])
    bench._self_launch(8)
    a = seen["argv"]
    assert a[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    assert a[-8:] == ["--gpus", "8", "--steps", "5", "--warmup", "2", "--config", "5"] and a[-9].endswith("bench.py")
    assert 5 in bench.CONFIGS and bench.CONFIGS[5].get("independent")
