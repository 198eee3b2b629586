"""Parity at BASELINE.json's FULL sizes (configs 2 and 3: 16 x 512^2 and 8 x 1024^2, T = 2048) through size-independent properties --
the oracle is too slow there:
  * the fused G-buffer kernel and the plain rasteriser (different template instantiations, different callers) agree bit for bit, and a
    random sample of covered pixels reproduces barycentric interpolation of the vertex attributes;
  * directional derivative: E(theta + eps d) - E(theta - eps d) = 2 eps <grad E, d> for random directions in the per-frame parameters;
  * the hand-chained NativeStep reproduces the autograd formulation (energy terms and gradients) and the captured step descends.
Tolerances: fp32 energies to 1e-4 relative, gradients to 2e-3 of their max-norm, finite differences to 3 % (the energy is piecewise
smooth: visibility changes are excluded by the small step)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = 2048
NAMES = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset",
         "focal_length")


def _tracker(flame_model, B, H, W, seed):
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    cfg = BaseTrackingConfig()
    cfg.model.tex_resolution = T
    cfg.render.disturb_rate_fg = cfg.render.disturb_rate_bg = None
    gt = make_scene_params(B, seed=seed, image_size=(H, W))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_dataset(rend, head, gt, (H, W), "cuda", seed=seed, tex=make_texture(seed, T))
    tr = GlobalTracker(cfg, model, topo, make_texture(0, T), data)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, s in (("shape", 0.2), ("expr", 0.2), ("rotation", 0.05), ("jaw_pose", 0.05), ("tex_extra", 0.02), ("lights", 0.03),
                        ("static_offset", 5e-4)):
            p = getattr(tr, name)
            p.add_((torch.randn(p.shape, generator=g) * s).cuda())
        tr.translation[:, 2] += 0.45
    return tr


@pytest.mark.parametrize("B,H,W", [(16, 512, 512), (8, 1024, 1024)])
def test_fullsize_properties(flame_model, B, H, W):
    from vhap_amd import ops
    from vhap_amd.step import NativeStep
    from vhap_amd.tracker import GraphedStep
    tr = _tracker(flame_model, B, H, W, seed=6)
    stage = "rgb_global_tracking"
    tr.get_train_parameters(stage)
    ts = np.arange(B)
    sample = tr.get_sample(ts, device_index=True)

    # --- 1. visibility: fused G-buffer pass == plain rasteriser, bit for bit; interpolation reproduces barycentric combination
    with torch.no_grad():
        verts, *_ = tr.forward_flame(sample["timestep_index"])
        s = dict(sample)
        tr.fill_cam_params_into_sample(s)
        clip = tr.render.world_to_clip(verts, s["extrinsic"], s["intrinsic"], (H, W)).contiguous()
        tri = tr.flame.faces.int().contiguous()
        vn = tr.render.compute_v_normals(verts, tr.flame.faces)
        tri_uv = tr.flame.textures_idx.int().contiguous()
        r0, d0 = ops.raster_fwd(tr.render.glctx, clip, tri, (H, W))
        r1, d1, nrm, texc, texd = ops.raster_interp_fwd(tr.render.glctx, clip, tri, vn, tr._verts_uv_flipped, tri_uv, (H, W))
        assert torch.equal(r0.view(torch.int32), r1.view(torch.int32)) and torch.equal(d0, d1)
        cov = r0[..., 3] > 0
        assert 0.05 < float(cov.float().mean()) < 0.9
        idx = cov.nonzero()
        pick = idx[torch.randperm(idx.shape[0], generator=torch.Generator().manual_seed(1))[:20000].cuda()]
        bb, yy, xx = pick[:, 0], pick[:, 1], pick[:, 2]
        rr = r0[bb, yy, xx]
        t = rr[:, 3].long() - 1
        u, v = rr[:, 0], rr[:, 1]
        # (b0, b1 are clamped to [0,1] one by one; their sum overshoots 1 only by the 1/16-px vertex snapping on small triangles)
        assert float(u.min()) >= 0 and float(v.min()) >= 0 and float(u.max()) <= 1 and float(v.max()) <= 1
        assert float((u + v).max()) <= 1.5 and float(((u + v) > 1.01).float().mean()) < 0.01
        f = tr.flame.faces[t]
        n_ref = u[:, None] * vn[bb, f[:, 0]] + v[:, None] * vn[bb, f[:, 1]] + (1 - u - v)[:, None] * vn[bb, f[:, 2]]
        assert float((nrm[bb, yy, xx] - n_ref).abs().max()) < 2e-6
        fu = tr.flame.textures_idx[t]
        uvv = tr._verts_uv_flipped
        t_ref = u[:, None] * uvv[fu[:, 0]] + v[:, None] * uvv[fu[:, 1]] + (1 - u - v)[:, None] * uvv[fu[:, 2]]
        assert float((texc[bb, yy, xx] - t_ref).abs().max()) < 2e-6

    # --- 2. NativeStep == autograd formulation at full size
    for p in tr._train_tensors:
        p.grad = None
    s = dict(sample)
    tr.fill_cam_params_into_sample(s)
    E, log, *_ = tr.compute_energy(s, stage=stage)
    E.backward()
    g_auto = {k: getattr(tr, k).grad.detach().clone() for k in NAMES}
    log = {k: float(v.detach()) for k, v in log.items()}
    ns = NativeStep(tr, sample, stage)
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    for k, v in log.items():
        assert abs(v - float(ns.log_dict()[k])) <= 1e-4 * max(abs(v), 1e-4), k
    for k in NAMES:
        a, b_ = g_auto[k], ns.g[k].reshape(g_auto[k].shape)
        assert float((a - b_).abs().max()) <= 2e-3 * float(a.abs().max()), k
    g = {k: ns.g[k].detach().clone() for k in NAMES}

    # --- 3. directional derivatives by central differences through the native forward.
    #  appearance (lights, tex_extra): visibility is fixed, the energy is piecewise smooth with kinks only where an L1 residual crosses
    #  zero -> 2 %.  geometry (expr / rotation / translation / jaw): the rasterised coverage itself moves; the analytic gradient sees that
    #  only through the silhouette antialiasing (as in the reference), so the two agree to ~25 %, sign included.
    def directional(names, d, eps):
        def energy_at(sign):
            with torch.no_grad():
                for k in names:
                    getattr(tr, k).add_(sign * eps * d[k])
            ns.forward()
            ns.backward(1)                   # (energy_total lives at the head of the backward)
            e = float(ns.log[15])
            with torch.no_grad():
                for k in names:
                    getattr(tr, k).sub_(sign * eps * d[k])
            return e
        fd = (energy_at(+1) - energy_at(-1)) / (2 * eps)
        an = sum(float((g[k].double() * d[k].double()).sum()) for k in names)
        return fd, an

    app = ("lights", "tex_extra")
    fd, an = directional(app, {k: g[k] / g[k].abs().max() for k in app}, 2e-3)
    assert an > 0 and abs(fd - an) <= 0.02 * an, ("appearance", fd, an)
    gen = torch.Generator().manual_seed(0)
    geo = {"expr": 1.0, "rotation": 0.2, "translation": 0.02, "jaw_pose": 0.2}
    d = {k: torch.randn(getattr(tr, k).shape, generator=gen).cuda() * sc for k, sc in geo.items()}
    fd, an = directional(tuple(geo), d, 2e-3)
    assert fd * an > 0 and abs(fd - an) <= 0.25 * abs(an), ("geometry", fd, an)
    print(f"[{B}x{H}x{W}] directional derivatives: geometry fd {fd:.4f} vs analytic {an:.4f}")

    # --- 4. the captured step descends
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=0.1)
    st = GraphedStep(tr, sample, opt, stage, warmup=0)
    e0 = float(st())
    for _ in range(8):
        e1 = float(st())
    assert np.isfinite(e1) and e1 < e0
