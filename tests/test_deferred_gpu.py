"""Deferred shading (vhap_raster_shade_fwd / vhap_deferred_shade_bwd) against the separate passes it replaces
(vhap_raster_interp_fwd -> vhap_texture_fwd -> vhap_shade_fwd, and vhap_shade_bwd + the uv part of vhap_texture_bwd), on the buffers of a
real fit step.  The separate passes are the ones pinned on the oracle (test_raster_gpu / test_ops_gpu / test_fused_gpu); the captured
step itself is checked against the oracle in test_fit_parity_gpu.  The rasteriser outputs must be bit-identical; the re-computed
interpolants (uv, uv derivatives) must be bit-identical to what the G-buffer pass wrote; colours and gradients agree to fp32 round-off
(the shading arithmetic is the same code, but the compiler may contract it differently in different kernels): 2e-6 absolute on colours,
1e-5 of the max-norm on gradients."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tracker(flame_model, B, H, W, T, seed, disturb):
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    cfg = BaseTrackingConfig()
    cfg.model.tex_resolution = T
    if not disturb:
        cfg.render.disturb_rate_fg = cfg.render.disturb_rate_bg = None
    gt = make_scene_params(B, seed=seed, image_size=(H, W))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_dataset(rend, head, gt, (H, W), "cuda", seed=seed, tex=make_texture(seed, T))
    tr = GlobalTracker(cfg, model, topo, make_texture(0, T), data)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, s in (("shape", 0.2), ("expr", 0.2), ("rotation", 0.05), ("jaw_pose", 0.05), ("tex_extra", 0.03), ("lights", 0.05),
                        ("static_offset", 5e-4)):
            p = getattr(tr, name)
            p.add_((torch.randn(p.shape, generator=g) * s).cuda())
        tr.translation[:, 2] += 0.45
        tr.lights[0] += 0.6                       # pushes max(diffuse) above 1: the relu branch of the diffuse regulariser is active
    return tr


@pytest.mark.parametrize("B,H,W,T,bg", [(3, 96, 120, 128, "target"), (2, 250, 203, 512, "white"), (9, 256, 256, 2048, "target")])
def test_deferred_kernels_match_the_separate_passes(flame_model, monkeypatch, B, H, W, T, bg):
    from vhap_amd import _lib
    from vhap_amd.ops import _p, _stream
    from vhap_amd.step import NativeStep
    monkeypatch.setenv("VHAP_DEFERRED", "0")                 # the reference: the separate passes
    tr = _tracker(flame_model, B, H, W, T, seed=31, disturb=True)
    tr.cfg.render.background_train = bg
    stage = "rgb_global_tracking"
    tr.get_train_parameters(stage)
    ns = NativeStep(tr, tr.get_sample(np.arange(B), device_index=True), stage)
    assert not ns.deferred and ns.disturb_on and ns.want_reg
    disturb = ns._disturb

    def keep_composite_then_disturb(st):                       # (the disturbance is in place: keep the composited image it starts from)
        ns.rgba_composited = ns.rgba.clone()
        disturb(st)
    ns._disturb = keep_composite_then_disturb
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    L = _lib.lib()
    V, F = ns.V, ns.F
    dev = "cuda"
    E = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    rast, rgba, stats = E(B, H, W, 4), E(B, H, W, 4), torch.zeros(4, device=dev)
    cid = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
    bgc = ns.bg_col

    def shade_fwd(flags, rast, rgba, cid, stats):
        return L.vhap_raster_shade_fwd(_p(ns.clip), _p(ns.tri), _p(ns.vn), _p(ns.uv), _p(ns.tri_uv), _p(ns.albedo_tex), _p(ns.mips), T, T,
                                       _p(tr.lights), _p(ns.sh_const), _p(ns.rgb) if bgc is None else 0,
                                       ctypes.cast(bgc, ctypes.c_void_p) if bgc is not None else 0, _p(ns.fid2cid), ns.fid2cid.numel(),
                                       B, V, ns.uv.shape[0], F, H, W, _p(rast), _p(rgba), _p(cid), _p(stats), 0, _p(ns.ws), ns.ws_bytes,
                                       ns.ws_cap, flags, _stream())
    assert shade_fwd(1, rast, rgba, cid, stats) == 0
    torch.cuda.synchronize()
    assert torch.equal(rast.view(torch.int32), ns.rast.view(torch.int32)), "rasteriser output differs"
    assert torch.equal(cid, ns.cid)
    cov = rast[..., 3] > 0
    assert 0.05 < float(cov.float().mean()) < 0.95
    ref = ns.rgba_composited
    assert torch.equal(rgba[..., 3], ref[..., 3])
    assert torch.equal(rgba[~cov], ref[~cov]), "background composite differs"
    assert float((rgba - ref).abs().max()) <= 2e-6
    s_new, s_old = stats.view(torch.int32).cpu().numpy(), ns.accF[12:16].view(torch.int32).cpu().numpy()
    dec = lambda u: np.array([(u & 0x7fffffff) if (u & 0x80000000) else (~u & 0xffffffff)], np.uint32).view(np.float32)[0]
    mx_new, mx_old = dec(int(s_new[1]) & 0xffffffff), dec(int(s_old[1]) & 0xffffffff)
    assert mx_old > 1.0 and abs(mx_new - mx_old) <= 2e-6 * mx_old
    assert abs(float(stats[2]) - float(ns.accF[14])) <= 1e-5 * abs(float(ns.accF[14]))
    if mx_new == mx_old:
        assert int(s_new[0]) == int(s_old[0])
    # split call: binning only, then rasterisation from the bins -- bit-identical to the single call
    r2, c2, i2, st2 = E(B, H, W, 4), E(B, H, W, 4), torch.empty_like(cid), torch.zeros(4, device=dev)
    assert ns.bin_split
    assert shade_fwd(1 | 2, r2, c2, i2, st2) == 0
    assert shade_fwd(1 | 4, r2, c2, i2, st2) == 0
    torch.cuda.synchronize()
    assert torch.equal(r2.view(torch.int32), rast.view(torch.int32)) and torch.equal(c2.view(torch.int32), rgba.view(torch.int32))
    assert torch.equal(i2, cid) and torch.equal(st2.view(torch.int32), stats.view(torch.int32))

    # ---- backward
    texc, texd, d_alb, d_n, d_tc, d_td = E(B, H, W, 2), E(B, H, W, 4), E(B, H, W, 3), E(B, H, W, 3), E(B, H, W, 2), E(B, H, W, 4)
    for t in (texc, texd, d_n, d_tc, d_td):
        t.zero_()                                              # (background pixels of these are not written)
    d_lights = torch.zeros(9, 3, device=dev)
    work = torch.zeros(L.vhap_deferred_shade_bwd_work_floats(B, H, W), device=dev)
    # d_lights of the separate passes alone: re-run vhap_shade_bwd into a fresh accumulator
    ref_lights = torch.zeros(9, 3, device=dev)
    d_alb_ref, d_n_ref = E(B, H, W, 3), E(B, H, W, 3)
    assert L.vhap_shade_bwd(_p(ns.normal), _p(ns.albedo_px), _p(ns.rast), _p(tr.lights), _p(ns.sh_const), _p(ns.d_color), _p(ns.keep),
                            _p(ns.c_reg), _p(ns.accF[12:16]), B, H, W, _p(d_alb_ref), _p(d_n_ref), _p(ref_lights), _stream()) == 0
    assert L.vhap_deferred_shade_bwd(_p(ns.clip), _p(ns.tri), _p(ns.vn), _p(ns.uv), _p(ns.tri_uv), _p(ns.albedo_tex), _p(ns.mips), T, T,
                                     _p(tr.lights), _p(ns.sh_const), _p(ns.rast), _p(ns.d_color), 0, 0, 0, 0, _p(ns.keep), _p(ns.c_reg),
                                     _p(ns.accF[12:16]), B, V, ns.uv.shape[0], F, H, W, _p(texc), _p(texd), _p(d_alb), _p(d_n), _p(d_tc),
                                     _p(d_td), _p(d_lights), _p(work), work.numel(), 0, 0, 0, _stream()) == 0
    torch.cuda.synchronize()
    m = cov[..., None]
    assert torch.equal((texc * m).view(torch.int32), (ns.texc * m).view(torch.int32)), "re-computed uv differs from the G-buffer's"
    assert torch.equal((texd * m).view(torch.int32), (ns.texd * m).view(torch.int32)), "re-computed uv derivatives differ"
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert torch.equal(d_alb[~cov], torch.zeros_like(d_alb[~cov]))
    assert rel(d_alb, d_alb_ref) <= 1e-5 and rel(d_alb, ns.d_albedo) <= 1e-5
    assert rel(d_n * m, d_n_ref * m) <= 1e-5
    assert rel(d_tc * m, ns.d_texc * m) <= 1e-5 and rel(d_td * m, ns.d_texd * m) <= 1e-5
    assert float(ns.d_texd.abs().max()) > 0 and float(ns.d_texc.abs().max()) > 0
    assert rel(d_lights, ref_lights) <= 2e-5, (d_lights, ref_lights)
    # the pass walks a frame as 16 x 16 tiles of 8 x 8 wave blocks (round 6) or -- widths 449 .. 512 -- in row order, 256 consecutive pixels per
    # workgroup: forced either way (debug flags 16777216 row order, 268435456 tiles; 33554432: 64 x 4 tiles of 64 x 1 waves) every per-pixel output
    # is the same bits and the lights gradient the same sum in another order
    for flag in (16777216, 268435456, 33554432):
        outs = [E(B, H, W, 2), E(B, H, W, 4), E(B, H, W, 3), E(B, H, W, 3), E(B, H, W, 2), E(B, H, W, 4)]
        for t in outs:
            t.zero_()
        d_lights_t, work_t = torch.zeros(9, 3, device=dev), torch.zeros_like(work)
        _lib.debug_set_flags(flag)
        rc = L.vhap_deferred_shade_bwd(_p(ns.clip), _p(ns.tri), _p(ns.vn), _p(ns.uv), _p(ns.tri_uv), _p(ns.albedo_tex), _p(ns.mips), T, T,
                                       _p(tr.lights), _p(ns.sh_const), _p(ns.rast), _p(ns.d_color), 0, 0, 0, 0, _p(ns.keep), _p(ns.c_reg),
                                       _p(ns.accF[12:16]), B, V, ns.uv.shape[0], F, H, W, *[_p(t) for t in outs], _p(d_lights_t), _p(work_t),
                                       work_t.numel(), 0, 0, 0, _stream())
        _lib.debug_set_flags(0)
        assert rc == 0
        torch.cuda.synchronize()
        for a, b in zip(outs, (texc, texd, d_alb, d_n, d_tc, d_td)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), flag
        assert rel(d_lights_t, d_lights) <= 2e-6
    # ... and over a LIST of the covered pixels (vhap_deferred_shade_bwd_list; in the step the list is a by-product of the disturbance's counting sort):
    # the same inputs, the same bits per pixel
    covp = torch.nonzero(ns.cid.reshape(-1) != 0).reshape(-1).int()
    assert torch.equal(ns.cid != 0, cov)
    lst = torch.full((B * H * W,), -1, dtype=torch.int32, device=dev)
    lst[:covp.numel()] = covp
    n_bg_t = torch.tensor([B * H * W - covp.numel()], dtype=torch.int32, device=dev)
    outs = [E(B, H, W, 2), E(B, H, W, 4), E(B, H, W, 3), E(B, H, W, 3), E(B, H, W, 2), E(B, H, W, 4)]
    for t in outs:
        t.zero_()
    d_lights_l, work_l = torch.zeros(9, 3, device=dev), torch.zeros_like(work)
    args = (_p(ns.clip), _p(ns.tri), _p(ns.vn), _p(ns.uv), _p(ns.tri_uv), _p(ns.albedo_tex), _p(ns.mips), T, T,
            _p(tr.lights), _p(ns.sh_const), _p(ns.rast), _p(ns.d_color), 0, 0, 0, 0, _p(ns.keep), _p(ns.c_reg),
            _p(ns.accF[12:16]), B, V, ns.uv.shape[0], F, H, W, *[_p(t) for t in outs], _p(d_lights_l), _p(work_l), work_l.numel(), 0, 0)
    assert L.vhap_deferred_shade_bwd_list(*args, _p(lst), _p(n_bg_t), 0, _stream()) != 0        # (needs VHAP_CALL_SKIP_BG_GRAD: it writes nothing for the background)
    assert L.vhap_deferred_shade_bwd_list(*args, _p(lst), _p(n_bg_t), _lib.CALL_SKIP_BG_GRAD, _stream()) == 0
    torch.cuda.synchronize()
    for a, b in zip(outs, (texc, texd, d_alb, d_n, d_tc, d_td)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert rel(d_lights_l, d_lights) <= 2e-6


def test_deferred_step_matches_separate_pass_step(flame_model, monkeypatch):
    """The whole NativeStep with and without deferred shading: same energy terms, same gradients."""
    from vhap_amd.step import NativeStep
    B, H, W, T = 4, 160, 128, 256
    stage = "rgb_global_tracking"
    out = {}
    # the separate passes (G-buffer, texture, shading, out-of-place antialiasing, dense gradient images) vs the step as shipped (deferred
    # shading, in-place antialiasing, photometric gradient on the fly)
    for mode in ("0", "1"):
        monkeypatch.setenv("VHAP_DEFERRED", mode)
        tr = _tracker(flame_model, B, H, W, T, seed=7, disturb=False)
        tr.get_train_parameters(stage)
        ns = NativeStep(tr, tr.get_sample(np.arange(B), device_index=True), stage)
        assert ns.deferred == (mode == "1") and ns.aa_inplace == (mode == "1")
        for _ in range(2):
            ns.forward()
            ns.backward(1)
        torch.cuda.synchronize()
        out[mode] = ({k: float(v) for k, v in ns.log_dict().items()}, {k: v.detach().clone() for k, v in ns.g.items() if k in ns.params})
    l0, g0 = out["0"]
    for mode in ("1",):
        l1, g1 = out[mode]
        for k, v in l0.items():
            assert abs(v - l1[k]) <= 2e-6 * max(abs(v), 1e-4), (mode, k, v, l1[k])
        for k, a in g0.items():
            b = g1[k]
            assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-12, (mode, k)


@pytest.mark.parametrize("B,H,W", [(3, 96, 120), (2, 250, 203), (5, 512, 512)])
def test_shading_backward_over_the_covered_pixel_list(flame_model, monkeypatch, B, H, W):
    """vhap_disturb_inplace_list leaves the covered pixels in pixel order (+ the background count) as a by-product of its counting sort;
    vhap_deferred_shade_bwd_list walks that list instead of the frame: the step's per-pixel gradient images agree with those of the pass over
    the whole frame (VHAP_SHADE_LIST=0) to 1e-5, the energies and the parameter gradients equal up to the order of the atomic additions."""
    from vhap_amd.step import NativeStep
    T, stage = 256, "rgb_global_tracking"
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VHAP_SHADE_LIST", mode)
        tr = _tracker(flame_model, B, H, W, T, seed=11, disturb=True)
        tr.get_train_parameters(stage)
        ns = NativeStep(tr, tr.get_sample(np.arange(B), device_index=True), stage)
        ns.injected = tr.render.make_disturbance((B, H, W), torch.device("cuda"), generator=torch.Generator(device="cuda").manual_seed(3))   # the same draws in both modes
        assert ns.deferred and ns.disturb_on and (ns.cov_list is not None) == (mode == "1")
        ns.forward()
        ns.backward(1)
        torch.cuda.synchronize()
        if mode == "1":
            assert ns._cov_list_fresh
            cov = torch.nonzero(ns.cid.reshape(-1) != 0).reshape(-1)
            n_bg = int(ns.n_bg[0])
            assert n_bg == ns.cid.numel() - cov.numel() and 0.05 < cov.numel() / ns.cid.numel() < 0.95
            assert torch.equal(ns.cov_list[:cov.numel()].long(), cov), "the list is not the covered pixels in pixel order"
        m = (ns.rast[..., 3] > 0)[..., None]
        out[mode] = ({k: float(v) for k, v in ns.log_dict().items()}, {k: v.detach().clone() for k, v in ns.g.items() if k in ns.params},
                     [torch.where(m, t, torch.zeros_like(t)) for t in (ns.d_normal, ns.d_texc, ns.d_texd, ns.d_albedo)])   # (background: never written)
    (l0, g0, px0), (l1, g1, px1) = out["0"], out["1"]
    for k, v in l0.items():                                       # (the forward pass is the same; its sums are atomic additions)
        assert abs(v - l1[k]) <= 1e-6 * max(abs(v), 1e-4), (k, v, l1[k])
    for a, b in zip(px0, px1):                                    # (two forward passes: the diffuse regulariser's statistics are atomic sums -- not the same bits)
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) and float(a.abs().max()) > 0
    for k, a in g0.items():
        assert float((a - g1[k]).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-12, (k, float((a - g1[k]).abs().max()) / float(a.abs().max()))


@pytest.mark.parametrize("T", [2048, 256, 128, 96])
def test_fused_pyramid_build_is_bit_identical(T):
    """vhap_tex_prep_mip1_fwd (level 1 written while the texture is assembled) + vhap_texture_mip_build_from (four levels per launch) ==
    vhap_tex_prep_fwd + vhap_texture_mip_build, bit for bit: albedo, every pyramid level, the TV / residual terms."""
    from vhap_amd import _lib
    from vhap_amd.ops import _p, _stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(T)
    painted = torch.rand(3, T, T, generator=g).cuda()
    extra = (torch.randn(3, T, T, generator=g) * 0.05).cuda()
    mask = (torch.rand(T, T, generator=g) < 0.2).to(torch.uint8).cuda()
    n = L.vhap_texture_mip_floats(1, T, T, 3)
    a0, a1 = torch.empty(1, T, T, 3, device="cuda"), torch.empty(1, T, T, 3, device="cuda")
    m0, m1 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    t0, t1 = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    assert L.vhap_tex_prep_fwd(_p(painted), _p(extra), _p(mask), T, 1e-3, 2e-3, _p(a0), _p(t0), 0, _stream()) == 0
    assert L.vhap_texture_mip_build(_p(a0), 1, T, T, 3, _p(m0), _stream()) == 0
    assert L.vhap_tex_prep_mip1_fwd(_p(painted), _p(extra), _p(mask), T, 1e-3, 2e-3, _p(a1), _p(m1), _p(t1), 0, _stream()) == 0
    assert L.vhap_texture_mip_build_from(_p(a1), 1, T, T, 3, _p(m1), 2, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(a0, a1)
    assert torch.equal(m0.view(torch.int32), m1.view(torch.int32)), int((m0 != m1).sum())
    assert torch.allclose(t0, t1, rtol=1e-5, atol=0) and float(t0[0]) > 0 and float(t0[1]) > 0


@pytest.mark.parametrize("bg", ["target", "white"])
def test_deferred_step_with_early_stores_is_bit_identical(flame_model, monkeypatch, bg):
    """VHAP_PREFILL=1 (opt-in, csrc/raster.hip PrefillJob): the binning launch stores the blocks outside each frame's geometry box, raster
    kernel mode 2 skips them and contributes only their (constant) share of the shading statistics -- every buffer the forward leaves
    behind and the energy log must have the same bits as without."""
    from vhap_amd.step import NativeStep
    B, H, W, T = 3, 200, 168, 256
    stage = "rgb_global_tracking"
    out = {}
    for pf in ("0", "1"):
        monkeypatch.setenv("VHAP_PREFILL", pf)
        tr = _tracker(flame_model, B, H, W, T, seed=7, disturb=False)
        tr.cfg.render.background_train = bg
        tr.get_train_parameters(stage)
        ns = NativeStep(tr, tr.get_sample(np.arange(B), device_index=True), stage)
        assert ns.deferred
        for buf in (ns.rast, ns.rgba):
            buf.fill_(float("nan"))
        ns.forward()
        ns.backward(1)
        torch.cuda.synchronize()
        out[pf] = dict(rast=ns.rast.clone(), rgba=ns.rgba.clone(), tile=ns.tile_ids.clone() if ns.tb_ids else None, log=ns.log.clone(),
                       stats=ns.accF[12:16].clone(), g={k: v.clone() for k, v in ns.g.items() if torch.is_tensor(v)})
    a, b = out["0"], out["1"]
    assert not bool(torch.isnan(b["rast"]).any()) and not bool(torch.isnan(b["rgba"]).any())
    assert torch.equal(a["rast"].view(torch.int32), b["rast"].view(torch.int32))
    assert torch.equal(a["stats"].view(torch.int32), b["stats"].view(torch.int32)), "shading statistics differ"
    assert a["tile"] is None or torch.equal(a["tile"], b["tile"])
    # (rgba went through the in-place antialiasing: float atomics on shared pixels -- equal up to their order)
    assert float((a["rgba"] - b["rgba"]).abs().max()) <= 1e-6
    assert float((a["log"] - b["log"]).abs().max()) <= 1e-5 * float(a["log"].abs().max())
