"""GPU parity of the fused per-frame / landmark / regulariser / Adam kernels (vhap_amd.native) against the host-side
torch formulation of the same reference code (FlameTracker with native=False, itself pinned to the oracle in
test_energy_gpu.py) and against torch.optim.Adam.  Both sides compute in fp32: energies must agree to 1e-4 relative,
gradients to 2e-3 of their max-norm (atomics ordering), Adam trajectories to 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H = W = 96
N = 4
T = 128


@pytest.fixture(scope="module")
def tracker(flame_model):
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    cfg = BaseTrackingConfig()
    cfg.model.tex_resolution = T
    gt = make_scene_params(N, seed=3, image_size=(H, W))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_dataset(rend, head, gt, (H, W), "cuda", seed=3, tex=make_texture(3, T))
    tr = GlobalTracker(cfg, model, topo, make_texture(0, T), data)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, s in (("shape", 0.3), ("expr", 0.3), ("rotation", 0.1), ("neck_pose", 0.05), ("jaw_pose", 0.08),
                        ("eyes_pose", 0.05), ("translation", 0.01), ("tex_extra", 0.03), ("lights", 0.05), ("static_offset", 1e-3)):
            p = getattr(tr, name)
            p.add_((torch.randn(p.shape, generator=g) * s).cuda())
        tr.translation[:, 2] += 0.45
        tr.jaw_pose[0, 0] = -0.07          # exercises relu(-jaw_x)
    return tr


NAMES = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset",
         "focal_length")


def _run(tr, stage, ts, native, dist):
    tr.native = native
    tr.get_train_parameters(stage)
    for p in tr._train_tensors:
        p.grad = None
    sample = tr.get_sample(ts)
    tr.fill_cam_params_into_sample(sample)
    E, log, verts, *_ = tr.compute_energy(sample, stage=stage, disturbance=dist)
    E.backward()
    grads = {k: (getattr(tr, k).grad.detach().clone() if getattr(tr, k).grad is not None else None) for k in NAMES}
    return float(E), {k: float(v) for k, v in log.items()}, grads, verts.detach().clone()


@pytest.mark.parametrize("stage,ts", [("rgb_global_tracking", [1, 2, 3]), ("rgb_sequential_tracking", [0]), ("lmk_init_all", [0, 2]),
                                      ("rgb_init_offset", [3, 1]), ("lmk_sequential_tracking", [2])])
def test_native_energy_matches_host_formulation(tracker, stage, ts):
    tr = tracker
    ts = np.array(ts)
    dist = None
    if stage.startswith("rgb"):
        dist = tr.render.make_disturbance((len(ts), H, W), "cuda", generator=torch.Generator("cuda").manual_seed(4))
    try:
        E0, log0, g0, v0 = _run(tr, stage, ts, False, dist)
        E1, log1, g1, v1 = _run(tr, stage, ts, True, dist)
    finally:
        tr.native = True
    assert float((v0 - v1).abs().max()) < 2e-6
    assert set(log0) == set(log1), (sorted(log0), sorted(log1))
    for k in log0:
        assert abs(log0[k] - log1[k]) <= 1e-4 * max(abs(log0[k]), 1e-4), f"{stage}: term {k}: host {log0[k]} native {log1[k]}"
    assert abs(E0 - E1) <= 1e-4 * abs(E0)
    for k in NAMES:
        a, b = g0[k], g1[k]
        if a is None or float(a.abs().max()) == 0:
            assert b is None or float(b.abs().max()) < 1e-6, f"{stage}: spurious gradient on {k}"
            continue
        assert b is not None, f"{stage}: no native gradient for {k}"
        rel = float((a - b).abs().max() / a.abs().max())
        assert rel < 2e-3, f"{stage}: grad {k}: rel {rel:.3e}"


def test_native_offset_regularisers_with_blurred_relax_weights(tracker):
    """`w.blur_iter > 0`: the relax tables become smooth per-vertex floats (host side, pinned on the reference in
    tests/test_energy_golden.py); the native offset-regulariser kernels must weight with them exactly like the host formulation."""
    tr = tracker
    stage, ts = "rgb_init_offset", np.array([3, 1])
    keep = tr.cfg.w.blur_iter
    tr.cfg.w.blur_iter = 2
    tr._nm = None                       # rebuild the native tables with the blurred weights
    try:
        assert tr._native_ok(stage)
        wl = tr._vertex_weights("lap", tr.cfg.w.reg_offset_lap_relax_coef, tr.cfg.w.reg_offset_lap_relax_for)
        assert float((wl - wl.round()).abs().max()) > 0.05              # genuinely blurred
        dist = tr.render.make_disturbance((len(ts), H, W), "cuda", generator=torch.Generator("cuda").manual_seed(4))
        E0, log0, g0, _ = _run(tr, stage, ts, False, dist)
        E1, log1, g1, _ = _run(tr, stage, ts, True, dist)
    finally:
        tr.native = True
        tr.cfg.w.blur_iter = keep
        tr._nm = None
    for k in ("reg_offset_lap", "reg_offset", "reg_offset_rigid"):
        assert abs(log0[k] - log1[k]) <= 1e-4 * max(abs(log0[k]), 1e-6), (k, log0[k], log1[k])
    a, b = g0["static_offset"], g1["static_offset"]
    assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max())


def test_hip_adam_matches_torch_adam():
    from vhap_amd.native import HipAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(300,), (7, 100), (3, 64, 64), (1, 513, 3), (1,)]
    mk = lambda: [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]
    p_ref = mk()
    p_hip = [p.detach().clone().requires_grad_() for p in p_ref]
    groups = lambda ps: [{"params": ps[:2], "lr": 3e-3}, {"params": ps[2:3], "lr": 1e-1}, {"params": ps[3:]}]
    o_ref = torch.optim.Adam(groups(p_ref), lr=5e-3)
    o_hip = HipAdam(groups(p_hip), lr=5e-3)
    for it in range(6):
        for a, b in zip(p_ref, p_hip):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (it - 3))
            a.grad, b.grad = gr.clone(), gr.clone()
        if it == 4:                      # a parameter without gradient is skipped, an lr change is picked up
            p_ref[1].grad = p_hip[1].grad = None
            o_ref.param_groups[0]["lr"] = o_hip.param_groups[0]["lr"] = 1e-3
        o_ref.step()
        o_hip.step()
        for a, b in zip(p_ref, p_hip):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), f"step {it}"
    assert int(o_hip.step_count) == 6


def test_native_step_matches_autograd_step(tracker):
    """NativeStep (hand-chained C calls, arenas, in-place gradient accumulation) == the autograd formulation of the same step:
    every energy term and every parameter gradient."""
    from vhap_amd.step import LOG_NAMES, NativeStep
    tr = tracker
    stage = "rgb_global_tracking"
    rates = (tr.render.disturb_rate_fg, tr.render.disturb_rate_bg)
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None          # in-kernel random numbers differ between two runs
    try:
        ts = np.array([1, 2, 3])
        E0, log0, g0, _ = _run(tr, stage, ts, True, None)
        assert NativeStep.supported(tr, stage)
        sample = tr.get_sample(ts, device_index=True)
        ns = NativeStep(tr, sample, stage)
        for _ in range(2):                                                 # twice: the arenas must be cleared correctly
            ns.forward()
            ns.backward(1)
        torch.cuda.synchronize()
        log1 = {k: float(v) for k, v in ns.log_dict().items()}
        for k, v in log0.items():
            assert abs(v - log1[k]) <= 1e-4 * max(abs(v), 1e-4), f"term {k}: autograd {v} native step {log1[k]}"
        for k in NAMES:
            a, b = g0[k], ns.g[k]
            assert a is not None
            rel = float((a - b.reshape(a.shape)).abs().max() / (a.abs().max() + 1e-30))
            assert rel < 2e-3, f"grad {k}: rel {rel:.3e}"
    finally:
        tr.render.disturb_rate_fg, tr.render.disturb_rate_bg = rates
        for p in tr._train_tensors:
            p.grad = None


@pytest.mark.parametrize("stage,ts", [("lmk_init_rigid", [0, 1, 2, 3]), ("lmk_init_all", [0, 2]), ("lmk_sequential_tracking", [2]),
                                      ("rgb_sequential_tracking", [1]), ("rgb_init_texture", [0, 1])])
def test_native_step_covers_every_stage_kind(tracker, stage, ts):
    """NativeStep for the landmark-only stages (no pixel chain) and the other photometric stage kinds == the autograd formulation; the
    reference runs 2 x 500 landmark steps before any photometric one (config/base.py stage table, tracker.py:1352-1357)."""
    from vhap_amd.step import NativeStep
    tr = tracker
    rates = (tr.render.disturb_rate_fg, tr.render.disturb_rate_bg)
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    try:
        ts = np.array(ts)
        E0, log0, g0, _ = _run(tr, stage, ts, True, None)
        assert NativeStep.supported(tr, stage)
        ns = NativeStep(tr, tr.get_sample(ts, device_index=True), stage)
        for _ in range(2):
            ns.forward()
            ns.backward(1)
        torch.cuda.synchronize()
        log1 = {k: float(v) for k, v in ns.log_dict().items()}
        for k, v in log0.items():
            assert abs(v - log1[k]) <= 1e-4 * max(abs(v), 1e-4), f"{stage}: term {k}: autograd {v} native step {log1[k]}"
        trained = {id(p) for v in tr.get_train_parameters(stage).values() for p in v}
        for k in NAMES:
            if id(getattr(tr, k)) not in trained:
                continue
            a, b = g0[k], ns.g[k]
            assert a is not None, k
            rel = float((a - b.reshape(a.shape)).abs().max() / (a.abs().max() + 1e-30))
            assert rel < 2e-3, f"{stage}: grad {k}: rel {rel:.3e}"
    finally:
        tr.render.disturb_rate_fg, tr.render.disturb_rate_bg = rates
        for p in tr._train_tensors:
            p.grad = None


def test_optimize_runs_every_stage_through_captured_native_steps(flame_model):
    """GlobalTracker.optimize() with the default graphed=True on the GPU (tracker.py:1343-1389): landmark stages, photometric stages,
    sequential tracking and global tracking all replay captured NativeSteps (the landmark stages used to hit a KeyError
    inside the capture); energies drop and the export is finite."""
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    Nf, Hh, Ww, Tt = 4, 96, 96, 128
    cfg = BaseTrackingConfig()
    cfg.model.tex_resolution = Tt
    cfg.batch_size = 2
    for st in cfg.pipeline.__dict__.values():
        if hasattr(st, "num_steps"):
            st.num_steps = 8
        if hasattr(st, "num_epochs"):
            st.num_epochs = 2
    gt = make_scene_params(Nf, seed=5, image_size=(Hh, Ww))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_dataset(rend, head, gt, (Hh, Ww), "cuda", seed=5, tex=make_texture(5, Tt))
    tr = GlobalTracker(cfg, model, topo, make_texture(0, Tt), data)
    with torch.no_grad():
        tr.translation[:, 2] = 0.40
    rep0 = tr.evaluate(batch_size=2)
    report = tr.optimize(batch_size=2, evaluate=True)
    rep1 = tr.evaluate(batch_size=2)
    stages = {k[0] for k in tr._graphed}
    assert {"lmk_init_rigid", "lmk_init_all", "rgb_init_texture", "rgb_init_all", "rgb_init_offset", "rgb_sequential_tracking",
            "rgb_global_tracking"} <= stages, stages
    assert all(st.ns is not None for st in tr._graphed.values()), "every stage must replay the native call sequence"
    assert report is not None and rep1["mean_lmk"] < rep0["mean_lmk"] and rep1["mean_photo"] < rep0["mean_photo"], (rep0, rep1)
    out = tr.save_result()
    assert all(np.isfinite(np.asarray(v, np.float64)).all() for v in out.values())
    assert tr.global_step == 5 * 8 + 2 * 8 + 2 * 2            # 5 init stages + 2 sequential batches + 2 epochs x 2 batches


def test_c_abi_is_reentrant_across_threads(tracker):
    """ABI 2 keeps no mutable state: a second thread calling ops (the reference's log_media thread renders while the fit runs,
    tracker.py:817-826) while this thread issues NativeStep passes -- whose calls carry VHAP_CALL_ACC_PREZEROED /
    VHAP_CALL_AA_PASSTHROUGH_DONE as per-call arguments -- must see the plain behaviour: its antialias backward still writes the
    pass-through gradient, its shading still clears its own accumulators."""
    import threading
    from vhap_amd import ops
    from vhap_amd.step import NativeStep
    tr = tracker
    stage = "rgb_global_tracking"
    tr.get_train_parameters(stage)
    rates = (tr.render.disturb_rate_fg, tr.render.disturb_rate_bg)
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    try:
        ns = NativeStep(tr, tr.get_sample(np.array([0, 1]), device_index=True), stage)
        ns.forward()
        ns.backward(1)
        torch.cuda.synchronize()
        g_ref = {k: ns.g[k].detach().clone() for k in ("expr", "lights", "static_offset")}
        e_ref = float(ns.log[15])
        rast, pos, tri = ns.rast.clone(), ns.clip.clone(), ns.tri
        color = torch.rand(2, H, W, 4, device="cuda")
        w = torch.randn(2, H, W, 4, device="cuda")
        torch.cuda.synchronize()                   # the inputs above are read on other (non-blocking) streams below

        def aa_grad(stream):
            with torch.cuda.stream(stream):
                c = color.clone().requires_grad_()
                (ops.antialias(c, rast, pos, tri, opp=ns.opp) * w).sum().backward()
                g = c.grad.clone()
                stream.synchronize()       # (behind the clone: the caller reads the result on ITS stream, which does not wait for this one --
                return g                   #  the copy used to be enqueued after the synchronisation: a full-suite run failed on it once in three)
        want = aa_grad(torch.cuda.Stream())
        assert float((want - w).abs().max()) < 10.0 and float(want.abs().max()) > 0     # pass-through part present
        want2 = aa_grad(torch.cuda.Stream())             # (the reference itself must be reproducible on THIS thread before another one is judged by it)
        assert float((want2 - want).abs().max()) <= 1e-5, ("main thread, two calls", float((want2 - want).abs().max()),
                                                            float((want - w).abs().max()), float((want2 - w).abs().max()))
        errs, stop = [], threading.Event()

        def worker():
            s = torch.cuda.Stream()
            try:
                while not stop.is_set():
                    got = aa_grad(s)
                    d = float((got - want).abs().max())      # (float atomics: the last bit depends on the order; a lost pass-through is O(1))
                    if not d <= 1e-5:
                        errs.append((d, float((got - w).abs().max()), float((want - w).abs().max()), int(((got - want).abs() > 1e-5).sum())))
            except Exception as e:          # noqa: BLE001
                errs.append(repr(e))
        th = threading.Thread(target=worker)
        th.start()
        try:
            for _ in range(30):
                ns.forward()
                ns.backward(1)
            torch.cuda.synchronize()
        finally:
            stop.set()
            th.join()
        assert not errs, errs[:3]
        assert abs(float(ns.log[15]) - e_ref) <= 1e-5 * abs(e_ref)
        for k, v in g_ref.items():
            assert float((ns.g[k] - v).abs().max()) <= 2e-3 * float(v.abs().max()), k
    finally:
        tr.render.disturb_rate_fg, tr.render.disturb_rate_bg = rates
        for p in tr._train_tensors:
            p.grad = None


def test_native_step_reads_nothing_before_writing_it(flame_model, monkeypatch):
    """Every buffer NativeStep allocates uninitialised (torch.empty) is filled with NaN / 0x7f7f7f7f instead (VHAP_POISON=1): energies and
    gradients of a forward + backward must come out as with ordinary allocations -- a read-before-write would surface as NaN (or as a
    first-run-vs-later-run difference in a long-lived process, which is how such a bug would otherwise be met)."""
    from tests.test_fit_parity_gpu import _make
    from vhap_amd.step import NativeStep
    stage = "rgb_global_tracking"
    out = []
    for poison in ("0", "1"):
        monkeypatch.setenv("VHAP_POISON", poison)
        S = _make(flame_model, 128, 128, 3, 256, seed=11)
        tr = S["tr"]
        tr.render._rng_state = torch.full((1,), 99, dtype=torch.int32, device="cuda")
        tr.get_train_parameters(stage)
        ns = NativeStep(tr, tr.get_sample(np.arange(3), device_index=True), stage)
        assert ns.disturb_on and ns.deferred
        for _ in range(2):                                       # twice: the second pass meets whatever the first one left behind
            ns.forward()
            ns.backward(1)
        torch.cuda.synchronize()
        out.append(({k: float(v) for k, v in ns.log_dict().items()}, {k: ns.g[k].detach().cpu().clone() for k in ("shape", "expr", "lights", "tex_extra", "static_offset")}))
    (la, ga), (lb, gb) = out
    for k in la:
        assert np.isfinite(lb[k]) and abs(la[k] - lb[k]) <= 1e-5 * max(abs(la[k]), 1e-3), (k, la[k], lb[k])
    for k in ga:
        assert torch.isfinite(gb[k]).all(), k
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-3 * float(ga[k].abs().max()) + 1e-12, k


def test_camera_rides_in_the_per_frame_launches_bit_exact(tracker):
    """vhap_frame_prep_fwd_camera / vhap_frame_prep_bwd_camera (the uncalibrated camera as one more workgroup of the per-frame kernels) ==
    vhap_camera_focal_fwd / vhap_camera_focal_bwd on the same buffers, bit for bit."""
    from vhap_amd import _lib
    from vhap_amd.ops import _p, _stream
    from vhap_amd.step import NativeStep, _chk
    tr = tracker
    stage = "rgb_global_tracking"
    try:
        ns = NativeStep(tr, tr.get_sample(np.array([0, 2, 3]), device_index=True), stage)
        assert not ns.calibrated
        ns.forward()
        ns.backward(1)
        torch.cuda.synchronize()
        L, B, H, W = _lib.lib(), ns.B, ns.H, ns.W
        mvp = torch.full_like(ns.mvp, float("nan"))
        _chk(L.vhap_camera_focal_fwd(_p(tr.focal_length), ns.focal_scale, 0.5 * W, 0.5 * H, _p(ns.RT), B, 0, H, W, 0.1, 10.0, _p(mvp), _stream()),
             "vhap_camera_focal_fwd")
        d_f = torch.zeros_like(ns.g["focal_length"])
        _chk(L.vhap_camera_focal_bwd(_p(ns.RT), _p(ns.d_mvp), B, 0, H, W, ns.focal_scale, _p(d_f), _stream()), "vhap_camera_focal_bwd")
        torch.cuda.synchronize()
        assert torch.equal(mvp, ns.mvp)
        assert float(ns.d_mvp.abs().max()) > 0 and torch.equal(d_f, ns.g["focal_length"])
    finally:
        for p in tr._train_tensors:
            p.grad = None
