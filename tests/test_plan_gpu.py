"""The step plan executor's deferred join (include/vhap_hip.h VHAP_CALL_PLAN_DEFER_JOIN; vhap_amd/tracker.py::GraphedStep.replay_stream):
inside a loop of replays step k+1's geometry chain starts under step k's texture tail.  The condition is structural and checked here on the
plan of the shipped step; the result must be the joined loop's."""
import os

import numpy as np
import pytest
import torch

from tests.test_fit_parity_gpu import NAMES, _make, _record

pytestmark = pytest.mark.gpu


def _update_rel(a, b, start):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm((b - start).ravel()), 1e-12))


def test_deferred_join_loop_matches_joined_loop(flame_model):
    from vhap_amd.tracker import GraphedStep
    H = W = 128
    stage, K = "rgb_global_tracking", 12
    S = _make(flame_model, H, W, 3, 256, seed=29)
    tr = S["tr"]
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    start = {k: getattr(tr, k).detach().clone() for k in NAMES}
    sample = tr.get_sample(np.array([0, 1, 2]), device_index=True)
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=0.1)
    st = GraphedStep(tr, sample, opt, stage, warmup=0)
    assert st.ns is not None and st.single and st.gF.plan is not None
    tails, heads = st.gF.open_tails(), st.gF.free_heads()
    lines = ["open tails: " + ", ".join(t.split("(")[0] for t in tails), "free heads: " + ", ".join(t.split("(")[0] for t in heads)]
    lines += ["decision on the buffers the nodes touch (CapturedPlan.deferred_join_hazards):"] + ["  " + r for r in st.defer_report]
    assert st.defer_join, "\n".join(lines)           # the shipped step qualifies: this test must exercise the deferred path
    # the decision is made on byte ranges; the kernel names are the cross-check of what the shipped step is expected to leave open / start early
    assert tails and all(any(o in t for o in st.TEX_TAIL) for t in tails), lines
    assert all(any(o in t for o in st.GEOMETRY_HEAD) for t in heads), lines
    assert not any("UNKNOWN" in r for r in st.defer_report) and "disjoint" in st.defer_report[-1], lines
    # ... and it must say NO when the host writes, between replays, into something an open tail touches (here: the texture parameter)
    ok, why = st.gF.deferred_join_hazards(st._access, [(tr.tex_extra.data_ptr(), 64)])
    assert not ok and "overlap" in why[-1], why

    def run(deferred):
        with torch.no_grad():
            for k in NAMES:
                getattr(tr, k).copy_(start[k])
        opt.reset_state()
        E = []
        if deferred:
            with st.replay_stream():
                for _ in range(K):
                    E.append(st().clone())             # (the energy lives on the launch stream: readable without a join)
        else:
            for _ in range(K):
                E.append(st().clone())
        torch.cuda.synchronize()
        return [float(e) for e in E], {k: getattr(tr, k).detach().cpu().numpy().copy() for k in NAMES}

    E_j, P_j = run(False)
    E_d, P_d = run(True)
    E_j2, P_j2 = run(False)                           # the joined loop against itself: the noise floor (atomics -> Adam)
    s0 = {k: v.cpu().numpy() for k, v in start.items()}
    fails = []
    for i, (a, b) in enumerate(zip(E_d, E_j)):
        if abs(a - b) > 2e-4 * abs(b):
            fails.append(f"energy at step {i}: deferred {a} joined {b}")
    for k in NAMES:
        d, floor = _update_rel(P_d[k], P_j[k], s0[k]), _update_rel(P_j2[k], P_j[k], s0[k])
        lines.append(f"{k}: deferred vs joined {d:.2e}   joined vs joined {floor:.2e}")
        if d > max(10 * floor, 2e-3):
            fails.append(f"{k}: {d:.2e} (floor {floor:.2e})")
    assert E_d[-1] < E_d[0]
    _record("plan_deferred_join.txt", lines + fails)
    assert not fails, fails


def test_deferred_join_at_baseline_size_repeated():
    """The same comparison where the overlap is largest -- BASELINE config 2 (16 x 512^2, T = 2048: the texture tail is ~300 us, the next
    step's geometry head runs underneath it), disturbance ON (in-kernel random numbers: counter-based, the same in both loops), five
    deferred loops of 12 steps against the joined loop.  A race between a step's open tail and the next step's head would show as an
    occasional excursion; the bound is the joined loop's own run-to-run spread."""
    import bench
    from vhap_amd.tracker import GraphedStep
    C = bench.CONFIGS[2]
    tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
    stage, K = bench.STAGE, 12
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=0.1)
    sample = tr.get_sample(own, device_index=True)
    st = GraphedStep(tr, sample, opt, stage, warmup=0)
    assert st.single and st.defer_join
    names = [k for k in NAMES if getattr(tr, k, None) is not None]
    start = {k: getattr(tr, k).detach().clone() for k in names}
    rng0 = tr.render._rng_state.clone()                         # the in-kernel random numbers are a function of this counter: same draws in every loop

    def run(deferred):
        with torch.no_grad():
            for k in names:
                getattr(tr, k).copy_(start[k])
        opt.reset_state()
        tr.render._rng_state.copy_(rng0)
        E = []
        if deferred:
            with st.replay_stream():
                for _ in range(K):
                    E.append(st().clone())
        else:
            for _ in range(K):
                E.append(st().clone())
        torch.cuda.synchronize()
        return np.array([float(e) for e in E]), {k: getattr(tr, k).detach().cpu().numpy().copy() for k in names}

    s0 = {k: v.cpu().numpy() for k, v in start.items()}
    E_j, P_j = run(False)
    E_j2, P_j2 = run(False)
    floor = {k: _update_rel(P_j2[k], P_j[k], s0[k]) for k in names}
    e_floor = float(np.abs(E_j2 - E_j).max() / np.abs(E_j).max())
    lines = [f"config 2, {K} steps; joined vs joined: energy {e_floor:.2e}, " + " ".join(f"{k} {v:.1e}" for k, v in floor.items())]
    fails = []
    for rep in range(5):
        E_d, P_d = run(True)
        e = float(np.abs(E_d - E_j).max() / np.abs(E_j).max())
        worst = {k: _update_rel(P_d[k], P_j[k], s0[k]) for k in names}
        lines.append(f"deferred loop {rep}: energy {e:.2e}, " + " ".join(f"{k} {v:.1e}" for k, v in worst.items()))
        if e > max(10 * e_floor, 1e-4):
            fails.append(f"loop {rep}: energy {e:.2e}")
        for k, v in worst.items():
            if v > max(20 * floor[k], 5e-3):
                fails.append(f"loop {rep}: {k} {v:.2e} (floor {floor[k]:.2e})")
    _record("plan_deferred_join_cfg2.txt", lines + fails)
    assert not fails, fails


def test_stage_loop_with_a_new_batch_every_step_deferred_vs_joined(flame_model, monkeypatch):
    """What a user runs (tracker.py:1376-1416): GlobalTracker.optimize_stage over shuffled batches of a resident FrameStore -- a NEW batch
    is ingested into the captured step's static buffers BETWEEN replays, while (deferred join) the previous step's texture tail is still
    open, and the learning rate changes every epoch.  The stage with the deferred join against the same stage with VHAP_DEFER_JOIN=0
    (same shuffles, same in-kernel random numbers: the disturbance is ON): every exported array inside the joined stage's own
    run-to-run spread.  """
    from vhap_amd.ingest import FrameStore
    from vhap_amd.tracker import GlobalTracker, ShuffledBatches
    H = W = 256
    N, B, T, stage, epochs = 24, 4, 512, "rgb_global_tracking", 3
    S = _make(flame_model, H, W, N, T, seed=7)
    tr0, cfg = S["tr"], S["cfg"]
    u8 = (tr0.dataset["rgb"].permute(0, 2, 3, 1).clamp(0, 1) * 255).round().to(torch.uint8).contiguous()
    names = [k for k in NAMES if getattr(tr0, k, None) is not None]
    start = {k: getattr(tr0, k).detach().clone() for k in names}
    keep_epochs = cfg.pipeline[stage].num_epochs
    cfg.pipeline[stage].num_epochs = epochs

    def run(defer, feed=True):
        monkeypatch.setenv("VHAP_DEFER_JOIN", "1" if defer else "0")
        monkeypatch.setenv("VHAP_STEP_FEED", "1" if feed else "0")
        data = {"frames": FrameStore(u8, device="cuda"), "lmk2d": tr0.dataset["lmk2d"].clone()}
        tr = GlobalTracker(cfg, S["model"], S["topo"], tr0.flame_tex_painted()[0].cpu().numpy(), data)
        with torch.no_grad():
            for k in names:
                getattr(tr, k).copy_(start[k])
        tr.render._rng_state = torch.full((1,), 12345, dtype=torch.int32, device="cuda")   # same in-kernel draws in every run
        loader = ShuffledBatches(tr, B, device_index=True, generator=torch.Generator().manual_seed(3))
        import gc
        frozen = gc.get_freeze_count()
        tr.optimize_stage(stage, dataloader=loader, lr_scale=0.1)
        torch.cuda.synchronize()
        assert gc.get_freeze_count() <= frozen and gc.isenabled()   # (the stage loop freezes the process's objects out of the collector's reach, and thaws them)
        st = next(iter(tr._graphed.values()))
        assert st.single and st.gF.plan is not None and st.defer_join == defer, (st.defer_join, defer, st.defer_report)
        assert (st.feed is not None) == feed                         # the captured step gathers its own batches (vhap_batch_feed) / is fed by the host
        assert tr.global_step == epochs * (N // B)
        return {k: getattr(tr, k).detach().cpu().numpy().copy() for k in names}

    try:
        P_j, P_d, P_j2, P_h = run(False), run(True), run(False), run(True, feed=False)
    finally:
        cfg.pipeline[stage].num_epochs = keep_epochs
    s0 = {k: v.cpu().numpy() for k, v in start.items()}
    lines, fails = [f"optimize_stage({stage}): {N} frames {H}x{W}, batches of {B}, {epochs} epochs = {epochs * (N // B)} steps, a new batch every step"], []
    for k in names:
        if float(np.abs(P_j[k] - s0[k]).max()) == 0:
            continue
        d, floor = _update_rel(P_d[k], P_j[k], s0[k]), _update_rel(P_j2[k], P_j[k], s0[k])
        h = _update_rel(P_h[k], P_j[k], s0[k])                  # fed by the host between replays instead of by the step's own feed node
        lines.append(f"{k}: deferred vs joined {d:.2e}   joined vs joined {floor:.2e}   host-fed (deferred) vs joined {h:.2e}")
        if d > max(10 * floor, 2e-3):
            fails.append(f"{k}: {d:.2e} (floor {floor:.2e})")
        if h > max(10 * floor, 2e-3):
            fails.append(f"{k}: host-fed {h:.2e} (floor {floor:.2e})")
    _record("plan_deferred_join_stage_loop.txt", lines + fails)
    assert not fails, fails


def test_multiview_stage_loop_self_feeding_step_matches_host_fed(flame_model, monkeypatch):
    """Calibrated multi-view sequence resident as uint8 (3 timesteps x 4 views; a batch = all views of ONE timestep, nersemble.py:22-42): the
    captured step's own feed node gathers frame indices, timesteps, landmarks AND the per-view intrinsics / extrinsics from the uploaded
    table (vhap_batch_feed) -- the stage against the same stage fed by the host between replays (update_timesteps), same shuffles."""
    from vhap_amd.config import nersemble_config
    from vhap_amd.flame import FlameHead
    from vhap_amd.ingest import FrameStore
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_multiview_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker, ShuffledBatches
    model, topo = flame_model
    H, W, NV, NT, T, stage = 96, 128, 4, 3, 256, "rgb_global_tracking"
    cfg = nersemble_config()
    cfg.model.tex_resolution = T
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    parts = []
    for t in range(NT):
        gt = make_scene_params(1, seed=20 + t, image_size=(H, W))
        parts.append(make_multiview_dataset(rend, head, gt, (H, W), "cuda", n_views=NV, seed=20 + t, tex=make_texture(3, T)))
    rgb = torch.cat([p["rgb"] for p in parts])
    u8 = (rgb.permute(0, 2, 3, 1).clamp(0, 1) * 255).round().to(torch.uint8).contiguous()
    base = {"lmk2d": torch.cat([p["lmk2d"] for p in parts]).contiguous(), "intrinsic": torch.cat([p["intrinsic"] for p in parts]).float().contiguous(),
            "extrinsic": torch.cat([p["extrinsic"] for p in parts]).float().contiguous(),
            "timestep_index": torch.arange(NT).repeat_interleave(NV)}
    keep_epochs = cfg.pipeline[stage].num_epochs
    cfg.pipeline[stage].num_epochs = 3

    def run(feed):
        monkeypatch.setenv("VHAP_STEP_FEED", "1" if feed else "0")
        data = dict(base, frames=FrameStore(u8, device="cuda"))
        tr = GlobalTracker(cfg, model, topo, make_texture(0, T), data)
        assert tr.calibrated and tr.n_timesteps == NT
        g = torch.Generator().manual_seed(4)
        with torch.no_grad():
            for name, s_ in (("shape", 0.2), ("expr", 0.2), ("rotation", 0.03), ("jaw_pose", 0.05), ("tex_extra", 0.02)):
                p = getattr(tr, name)
                p.add_((torch.randn(p.shape, generator=g) * s_).cuda())
        tr.render._rng_state = torch.full((1,), 777, dtype=torch.int32, device="cuda")
        loader = ShuffledBatches(tr, 1, device_index=True, generator=torch.Generator().manual_seed(9))
        start = {k: getattr(tr, k).detach().cpu().numpy().copy() for k in NAMES if getattr(tr, k, None) is not None}
        tr.optimize_stage(stage, dataloader=loader, lr_scale=0.1)
        torch.cuda.synchronize()
        st = next(iter(tr._graphed.values()))
        assert st.single and (st.feed is not None) == feed and tr.global_step == 3 * NT
        if feed:
            assert st.ns.calibrated and st.feed["n"] == NV
            # a full SAMPLE handed to a self-feeding step (update_sample): its `timestep_index` names the timestep once per VIEW -- the table
            # must hold that timestep's views once, not once per entry (round-4 advisor finding)
            st.update_sample(tr.get_sample(np.array([2]), device_index=True))
            torch.cuda.synchronize()
            assert st.feed["cursor"].cpu().tolist() == [0, 1]
            assert st.feed["frames"][:NV].cpu().tolist() == list(tr._frames_of[2]) and st.feed["ts"][:NV].cpu().tolist() == [2] * NV
        return start, {k: getattr(tr, k).detach().cpu().numpy().copy() for k in start}

    try:
        (s0, P_f), (_, P_h), (_, P_h2) = run(True), run(False), run(False)
    finally:
        cfg.pipeline[stage].num_epochs = keep_epochs
    lines, fails = [f"multi-view optimize_stage({stage}): {NT} timesteps x {NV} views {H}x{W}, 3 epochs"], []
    for k in s0:
        if float(np.abs(P_h[k] - s0[k]).max()) == 0:
            continue
        d, floor = _update_rel(P_f[k], P_h[k], s0[k]), _update_rel(P_h2[k], P_h[k], s0[k])
        lines.append(f"{k}: self-feeding vs host-fed {d:.2e}   host-fed vs host-fed {floor:.2e}")
        # (a batch fed wrongly shows as O(0.1 .. 1) of the update; two correct runs of this tiny problem -- 12 frames of 96 x 128, 9 steps --
        # are usually 1e-4 apart and once in a while 2e-2 on the per-frame rows: Adam's g / (|g| + eps) turns a noise-level gradient entry
        # into a full step of either sign)
        if d > max(10 * floor, 5e-2):
            fails.append(f"{k}: {d:.2e} (floor {floor:.2e})")
    _record("plan_stage_loop_multiview_feed.txt", lines + fails)
    assert not fails, fails


def test_set_ints_passes_bits_under_flush_denormal():
    """The self-feeding step's int32 cursor (0, m) is written by a launch carrying the ints in its arguments.  Small ints are float32
    SUBNORMALS: any float conversion on a host thread in FTZ / DAZ mode (torch.set_flush_denormal(True)) would flush them to 0 and the
    stage would silently train on batch 0 (round-4 advisor finding).  The bits must arrive, NaN patterns included."""
    from vhap_amd.tracker import _set_ints
    vals = (0, 80, 1, 0x7fc00001, -1, 0x7f800001, 5, -2 ** 31)         # subnormals, a quiet and a signalling NaN pattern, -0.0
    dst = torch.full((len(vals),), 77, dtype=torch.int32, device="cuda")
    torch.set_flush_denormal(True)
    try:
        _set_ints(dst, vals)
        torch.cuda.synchronize()
    finally:
        torch.set_flush_denormal(False)
    assert dst.cpu().tolist() == list(vals)
