"""GPU parity of the differentiable raster ops (forward values and gradients) vs the torch oracle.

Tolerances (float32 kernels vs the float64 torch oracle on fixed visibility):
  - (u, v, z/w) and everything interpolated from them: <= 5e-4 abs.  The perspective barycentric
    formula cancels heavily for pixel-sized triangles; the measured fp32-vs-fp64 spread of the
    formula itself on these scenes is 1.1e-4 (the bit-exact check against the fp32 C oracle is in
    test_raster_gpu.py).
  - texture / antialias on identical inputs: <= 2e-5 abs.
  - gradients: relative to the gradient's max-norm, <= 1e-2 where they pass through the raster
    formula, <= 2e-3 otherwise (fp32 atomics in arbitrary order)."""
RAST_ATOL = 5e-4
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_ref as R
from tests.scenes import head_scene

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def scene(flame_model):
    model, topo = flame_model
    B, H, W = 2, 160, 128
    sc = head_scene(model, B, H, W, seed=21, dtype=torch.float64)
    pos = sc["clip"]
    tri = torch.from_numpy(topo.faces.astype(np.int64))
    rast_np, db_np = oracle.rasterize(pos.float().numpy(), topo.faces.astype(np.int32), (H, W))
    tid = torch.from_numpy(rast_np[..., 3].astype(np.int64) - 1)
    return dict(B=B, H=H, W=W, pos=pos, tri=tri, tid=tid, topo=topo, sc=sc, model=model)


def test_rasterize_backward(scene):
    from vhap_amd import ops
    B, H, W = scene["B"], scene["H"], scene["W"]
    g = torch.Generator().manual_seed(0)
    w_r = torch.randn(B, H, W, 2, generator=g, dtype=torch.float64)
    w_d = torch.randn(B, H, W, 4, generator=g, dtype=torch.float64) * 0.1
    pos_o = scene["pos"].clone().requires_grad_()
    rast_o, db_o = R.rast_from_ids(pos_o, scene["tri"], scene["tid"], (H, W))
    ((rast_o[..., :2] * w_r).sum() + (db_o * w_d).sum()).backward()
    pos_g = scene["pos"].float().cuda().requires_grad_()
    rast, db = ops.rasterize(ops.RasterizeHipContext(), pos_g, scene["tri"].int().cuda(), (H, W))
    assert np.array_equal(rast[..., 3].detach().cpu().numpy(), (scene["tid"] + 1).float().numpy())
    assert (rast[..., :3].detach().cpu().double() - rast_o[..., :3].detach()).abs().max() < RAST_ATOL
    assert _rel(db.detach(), db_o.detach()) < 1e-3
    ((rast[..., :2] * w_r.float().cuda()).sum() + (db * w_d.float().cuda()).sum()).backward()
    assert _rel(pos_g.grad, pos_o.grad) < 1e-2


def test_rasterize_backward_through_near_plane_clipped_triangles():
    """Gradient of (u, v, du/dX ...) w.r.t. clip-space positions for pixels won by PIECES of triangles cut by the near plane (some with a
    vertex behind the camera): the backward uses the original triangle's vertices -- against autograd through the fp64 restatement on the
    C oracle's visibility."""
    from tests.scenes import near_plane_scene
    from vhap_amd import ops
    B, H, W = 2, 256, 208
    pos_np, tri_np, _ = near_plane_scene(B)
    rast_np, _ = oracle.rasterize(pos_np, tri_np, (H, W))
    tid = torch.from_numpy(rast_np[..., 3].astype(np.int64) - 1)
    tri = torch.from_numpy(tri_np).long()
    behind = torch.from_numpy((pos_np[..., 2] + pos_np[..., 3]) < 0)                       # [B,V]
    nb = torch.stack([behind[b][tri].sum(-1) for b in range(B)])                             # [B,F]
    won_by_cut = torch.stack([(nb[b] > 0)[tid[b].clamp(min=0)] & (tid[b] >= 0) for b in range(B)])
    assert int(won_by_cut.sum()) > 20000
    g = torch.Generator().manual_seed(0)
    w_r = torch.randn(B, H, W, 2, generator=g, dtype=torch.float64) * won_by_cut[..., None]     # only the cut triangles' pixels contribute
    w_d = torch.randn(B, H, W, 4, generator=g, dtype=torch.float64) * 0.01 * won_by_cut[..., None]
    pos_o = torch.from_numpy(pos_np).double().requires_grad_()
    rast_o, db_o = R.rast_from_ids(pos_o, tri, tid, (H, W))
    ((rast_o[..., :2] * w_r).sum() + (db_o * w_d).sum()).backward()
    pos_g = torch.from_numpy(pos_np).cuda().requires_grad_()
    rast, db = ops.rasterize(ops.RasterizeHipContext(), pos_g, tri.int().cuda(), (H, W))
    assert np.array_equal(rast[..., 3].detach().cpu().numpy(), rast_np[..., 3])
    ((rast[..., :2] * w_r.float().cuda()).sum() + (db * w_d.float().cuda()).sum()).backward()
    assert float(pos_o.grad.abs().max()) > 0
    assert _rel(pos_g.grad, pos_o.grad) < 1e-2


def test_interpolate_forward_backward(scene):
    from vhap_amd import ops
    B, H, W = scene["B"], scene["H"], scene["W"]
    g = torch.Generator().manual_seed(1)
    V = scene["pos"].shape[1]
    for AB, A, use_db in ((B, 3, False), (1, 2, True), (B, 5, True)):
        attr = torch.randn(AB, V, A, generator=g, dtype=torch.float64)
        w_o = torch.randn(B, H, W, A, generator=g, dtype=torch.float64)
        w_da = torch.randn(B, H, W, 2 * A, generator=g, dtype=torch.float64)
        pos_o = scene["pos"].clone().requires_grad_()
        attr_o = attr.clone().requires_grad_()
        rast_o, db_o = R.rast_from_ids(pos_o, scene["tri"], scene["tid"], (H, W))
        out_o, da_o = R.interpolate(attr_o, rast_o, scene["tri"], db_o if use_db else None, "all" if use_db else None)
        loss = (out_o * w_o).sum() + ((da_o * w_da).sum() if use_db else 0)
        loss.backward()

        pos_g = scene["pos"].float().cuda().requires_grad_()
        attr_g = attr.float().cuda().requires_grad_()
        tri_g = scene["tri"].int().cuda()
        rast, db = ops.rasterize(ops.RasterizeHipContext(), pos_g, tri_g, (H, W))
        out, da = ops.interpolate(attr_g, rast, tri_g, rast_db=db if use_db else None, diff_attrs="all" if use_db else None)
        assert (out.detach().cpu().double() - out_o.detach()).abs().max() < RAST_ATOL * 5   # |attr| up to ~5
        loss = (out * w_o.float().cuda()).sum()
        if use_db:
            assert _rel(da.detach(), da_o.detach()) < 1e-3
            loss = loss + (da * w_da.float().cuda()).sum()
        loss.backward()
        assert _rel(attr_g.grad, attr_o.grad) < 2e-3
        assert _rel(pos_g.grad, pos_o.grad) < 1e-2


@pytest.mark.parametrize("TB,T,C", [(1, 64, 3), (2, 32, 4), (1, 256, 1)])
def test_texture_forward_backward(TB, T, C):
    from vhap_amd import ops
    g = torch.Generator().manual_seed(2)
    B, H, W = 2, 48, 40
    tex = torch.rand(TB, T, T, C, generator=g, dtype=torch.float64)
    uv = torch.rand(B, H, W, 2, generator=g, dtype=torch.float64) * 1.4 - 0.2          # exercises wrap
    scale = torch.exp(torch.rand(B, H, W, 1, generator=g, dtype=torch.float64) * 6 - 7)  # levels from 0 to coarse
    da = torch.randn(B, H, W, 4, generator=g, dtype=torch.float64) * scale
    da[0, 0, :4] = 0                                                                # background-like pixels
    w = torch.randn(B, H, W, C, generator=g, dtype=torch.float64)
    t_o, uv_o, da_o = tex.clone().requires_grad_(), uv.clone().requires_grad_(), da.clone().requires_grad_()
    out_o = R.texture(t_o, uv_o, da_o)
    (out_o * w).sum().backward()
    t_g, uv_g, da_g = (x.float().cuda().requires_grad_() for x in (tex, uv, da))
    out = ops.texture(t_g, uv_g, da_g, filter_mode="linear-mipmap-linear")
    assert (out.detach().cpu().double() - out_o.detach()).abs().max() < 2e-5
    (out * w.float().cuda()).sum().backward()
    assert _rel(t_g.grad, t_o.grad) < 1e-3
    assert _rel(uv_g.grad, uv_o.grad) < 2e-3
    assert _rel(da_g.grad, da_o.grad) < 2e-3
    # plain bilinear
    out_l = ops.texture(t_g, uv_g, None, filter_mode="linear")
    assert (out_l.detach().cpu().double() - R.texture(tex, uv, None, filter_mode="linear")).abs().max() < 2e-5


def test_texture_constant_and_box_known_answers():
    from vhap_amd import ops
    T = 64
    tex = torch.full((1, T, T, 3), 0.37, device="cuda")
    uv = torch.rand(1, 8, 8, 2, device="cuda")
    for s in (1e-4, 1e-2, 0.1, 1.0):
        da = torch.full((1, 8, 8, 4), s, device="cuda")
        assert torch.allclose(ops.texture(tex, uv, da), torch.full((1, 8, 8, 3), 0.37, device="cuda"), atol=1e-6)
    # checkerboard: level 1 is exactly the box-filtered texture (0.5 everywhere)
    yy, xx = torch.meshgrid(torch.arange(T), torch.arange(T), indexing="ij")
    cb = ((xx + yy) % 2).float()[None, :, :, None].cuda().contiguous()
    da = torch.zeros(1, 8, 8, 4, device="cuda")
    da[..., 0] = 2.0 / T                      # lambda = 4 -> level exactly 1
    da[..., 3] = 2.0 / T
    assert torch.allclose(ops.texture(cb, uv, da), torch.full((1, 8, 8, 1), 0.5, device="cuda"), atol=1e-6)


def test_antialias_forward_backward(scene):
    from vhap_amd import ops
    B, H, W = scene["B"], scene["H"], scene["W"]
    topo = scene["topo"]
    g = torch.Generator().manual_seed(3)
    color = torch.rand(B, H, W, 4, generator=g, dtype=torch.float64)
    w = torch.randn(B, H, W, 4, generator=g, dtype=torch.float64)
    opp = torch.from_numpy(topo.opp.astype(np.int64))
    pos_o = scene["pos"].clone().requires_grad_()
    col_o = color.clone().requires_grad_()
    rast_o, _ = R.rast_from_ids(pos_o.detach(), scene["tri"], scene["tid"], (H, W))
    out_o = R.antialias(col_o, rast_o, pos_o, scene["tri"], opp)
    (out_o * w).sum().backward()
    n_changed = int(((out_o.detach() - color).abs().sum(-1) > 0).sum())
    assert n_changed > 200                                               # silhouettes were found

    pos_g = scene["pos"].float().cuda().requires_grad_()
    col_g = color.float().cuda().requires_grad_()
    tri_g = scene["tri"].int().cuda()
    rast_g = rast_o.float().cuda()
    out = ops.antialias(col_g, rast_g, pos_g, tri_g, opp=opp.int().cuda())
    diff = (out.detach().cpu().double() - out_o.detach()).abs()
    # fp32-vs-fp64 decisions can differ on a handful of knife-edge pairs; require near-total agreement
    assert float((diff.amax(-1) > 1e-4).float().mean()) < 2e-4
    (out * w.float().cuda()).sum().backward()
    assert _rel(col_g.grad, col_o.grad) < 2e-2
    gp, go = pos_g.grad.double().cpu(), pos_o.grad
    assert float((gp - go).norm() / go.norm()) < 5e-2


def test_antialias_backward_with_detached_vertices(scene):
    """vhap_antialias_bwd(pos_nograd_verts) == the reference's detach_by_indices before dr.antialias (render_nvdiffrast.py:349-352,
    463-464) as restated by the oracle: the masked vertices get exactly zero silhouette gradient, the others the oracle's."""
    from vhap_amd import ops
    B, H, W = scene["B"], scene["H"], scene["W"]
    topo = scene["topo"]
    g = torch.Generator().manual_seed(13)
    color = torch.rand(B, H, W, 4, generator=g, dtype=torch.float64)
    w = torch.randn(B, H, W, 4, generator=g, dtype=torch.float64)
    opp = torch.from_numpy(topo.opp.astype(np.int64))
    vid = torch.from_numpy(topo.get_vid_by_region(["hair", "boundary", "ears"]))       # a real stage's kind of exclusion list
    V = scene["pos"].shape[1]
    assert 50 < vid.numel() < V - 50
    pos_o = scene["pos"].clone().requires_grad_()
    vc = pos_o.clone()
    vc[:, vid] = pos_o[:, vid].detach()
    rast_o, _ = R.rast_from_ids(pos_o.detach(), scene["tri"], scene["tid"], (H, W))
    (R.antialias(color, rast_o, vc, scene["tri"], opp) * w).sum().backward()
    go = pos_o.grad
    assert float(go[:, vid].abs().max()) == 0 and float(go.abs().max()) > 0
    # the mask must matter: without it those vertices do receive gradient
    pos_u = scene["pos"].clone().requires_grad_()
    (R.antialias(color, rast_o, pos_u, scene["tri"], opp) * w).sum().backward()
    assert float(pos_u.grad[:, vid].abs().max()) > 0

    mask = torch.zeros(V, dtype=torch.uint8)
    mask[vid] = 1
    pos_g = scene["pos"].float().cuda().requires_grad_()
    out = ops.antialias(color.float().cuda(), rast_o.float().cuda(), pos_g, scene["tri"].int().cuda(), opp=opp.int().cuda(),
                        pos_nograd_verts=mask.cuda())
    (out * w.float().cuda()).sum().backward()
    gp = pos_g.grad.double().cpu()
    assert float(gp[:, vid].abs().max()) == 0
    assert float((gp - go).norm() / go.norm()) < 5e-2


def test_gbuffer_backward_with_uv_detached_faces(scene):
    """vhap_gbuffer_bwd(uv_nograd_faces) == the reference's `texc = torch.where(mask[fid], texc.detach(), texc)`
    (render_nvdiffrast.py:390-396) as restated by the oracle: on masked faces d_texc is dropped while d_texd and d_normal still flow."""
    from vhap_amd import ops
    B, H, W = scene["B"], scene["H"], scene["W"]
    topo = scene["topo"]
    g = torch.Generator().manual_seed(14)
    V = scene["pos"].shape[1]
    F = scene["tri"].shape[0]
    fid = torch.from_numpy(topo.get_fid_by_region(["hair", "boundary", "neck"]))
    assert 100 < fid.numel() < F - 100
    vn = torch.nn.functional.normalize(torch.randn(B, V, 3, generator=g, dtype=torch.float64), dim=-1)
    uv = torch.from_numpy(topo.verts_uvs.astype(np.float64))
    tri_uv = torch.from_numpy(topo.faces_uv.astype(np.int64))
    wn, wc, wd = (torch.randn(B, H, W, k, generator=g, dtype=torch.float64) for k in (3, 2, 4))
    wd = wd * 0.01

    def oracle_grads(masked):
        pos_o, vn_o = scene["pos"].clone().requires_grad_(), vn.clone().requires_grad_()
        rast_o, db_o = R.rast_from_ids(pos_o, scene["tri"], scene["tid"], (H, W))
        normal, _ = R.interpolate(vn_o, rast_o, scene["tri"])
        texc, texd = R.interpolate(uv[None], rast_o, tri_uv, db_o, "all")
        if masked:
            m = torch.zeros(F + 1, dtype=torch.bool)
            m[fid + 1] = True
            texc = torch.where(m[rast_o[..., 3].detach().long()][..., None], texc.detach(), texc)
        ((normal * wn).sum() + (texc * wc).sum() + (texd * wd).sum()).backward()
        return pos_o.grad, vn_o.grad

    gp_o, gn_o = oracle_grads(True)
    gp_u, _ = oracle_grads(False)
    assert float((gp_o - gp_u).abs().max()) > 1e-3 * float(gp_u.abs().max())            # the mask changes the answer
    mask = torch.zeros(F, dtype=torch.uint8)
    mask[fid] = 1
    pos_g, vn_g = scene["pos"].float().cuda().requires_grad_(), vn.float().cuda().requires_grad_()
    rast, db, normal, texc, texd = ops.raster_interp(ops.RasterizeHipContext(), pos_g, scene["tri"].int().cuda(), vn_g, uv.float().cuda(),
                                                     tri_uv.int().cuda(), (H, W), uv_nograd_faces=mask.cuda())
    ((normal * wn.float().cuda()).sum() + (texc * wc.float().cuda()).sum() + (texd * wd.float().cuda()).sum()).backward()
    assert _rel(vn_g.grad, gn_o) < 2e-3
    assert _rel(pos_g.grad, gp_o) < 1e-2


def test_antialias_vertical_edge_known_answer():
    """A vertical silhouette at x = k + 0.25 px blends the two pixels with weights 0.25 / 0.75."""
    from vhap_amd import ops
    H = W = 16
    xe = (8.25 / W) * 2 - 1                                   # edge at pixel x = 8.25
    pos = torch.tensor([[[-1.0, -1.0, 0, 1], [xe, -1.0, 0, 1], [xe, 1.0, 0, 1], [-1.0, 1.0, 0, 1]]], device="cuda")
    tri = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32, device="cuda")
    rast, _ = ops.rasterize(ops.RasterizeHipContext(), pos, tri, (H, W))
    fg = (rast[..., 3:] > 0).float()
    color = fg.expand(-1, -1, -1, 3).contiguous()               # white mesh on black
    out = ops.antialias(color, rast, pos, tri)
    row = out[0, 8, :, 0].cpu()
    assert torch.allclose(row[:8], torch.ones(8)) and torch.allclose(row[9:], torch.zeros(7))
    assert abs(float(row[8]) - 0.25) < 1e-5                     # pixel 8 (centre 8.5) is 25 % covered


def test_fused_raster_interp_backward_matches_separate_ops(scene):
    """Triangle-parallel G-buffer backward (vhap_gbuffer_bwd) == interpolate-backward x2 + rasterize-backward."""
    from vhap_amd import ops
    B, H, W = scene["B"], scene["H"], scene["W"]
    topo = scene["topo"]
    g = torch.Generator().manual_seed(4)
    V = scene["pos"].shape[1]
    vn0 = torch.nn.functional.normalize(torch.randn(B, V, 3, generator=g), dim=-1).cuda()
    uv = torch.from_numpy(topo.verts_uvs.astype(np.float32)).cuda()
    tri = scene["tri"].int().cuda()
    tri_uv = torch.from_numpy(topo.faces_uv.astype(np.int32)).cuda()
    wn, wc, wd = (torch.randn(B, H, W, k, generator=g).cuda() for k in (3, 2, 4))
    res = []
    for fused in (False, True):
        pos = scene["pos"].float().cuda().requires_grad_()
        vn = vn0.clone().requires_grad_()
        ctx = ops.RasterizeHipContext()
        if fused:
            rast, db, normal, texc, texd = ops.raster_interp(ctx, pos, tri, vn, uv, tri_uv, (H, W))
        else:
            rast, db = ops.rasterize(ctx, pos, tri, (H, W))
            normal, _ = ops.interpolate(vn, rast, tri)
            texc, texd = ops.interpolate(uv[None], rast, tri_uv, rast_db=db, diff_attrs="all")
        ((normal * wn).sum() + (texc * wc).sum() + (texd * wd).sum() * 0.01).backward()
        res.append((normal.detach(), texc.detach(), texd.detach(), pos.grad, vn.grad))
    for a, b in zip(res[1][:3], res[0][:3]):
        assert torch.equal(a, b)
    assert _rel(res[1][3], res[0][3]) < 2e-3 and _rel(res[1][4], res[0][4]) < 2e-3


@pytest.mark.parametrize("Ht,Wt,C,B,H,W", [(64, 64, 3, 3, 48, 40), (256, 128, 1, 2, 33, 47), (32, 32, 4, 1, 16, 16), (2048, 2048, 3, 4, 96, 96)])
def test_texture_grad_binned_matches_tiled_and_oracle(Ht, Wt, C, B, H, W):
    """vhap_texture_grad_binned (uv-space binning, one shared texture) == the d_tex part of vhap_texture_bwd == the oracle, for every
    mip level, wrap-around taps, uncovered pixels (d_out == 0) and a frame batch that samples the same texels."""
    from vhap_amd import _lib
    from vhap_amd.native import texture_grad_binned
    from vhap_amd.ops import _p, _stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(Ht + W)
    uv = torch.rand(B, H, W, 2, generator=g, dtype=torch.float64) * 1.4 - 0.2
    uv[0, 0, :3] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [-1e-9, 0.999999]], dtype=torch.float64)      # seams / u - floor(u) == 1 in fp32
    scale = torch.exp(torch.rand(B, H, W, 1, generator=g, dtype=torch.float64) * 7 - 8)
    da = torch.randn(B, H, W, 4, generator=g, dtype=torch.float64) * scale
    d_out = torch.randn(B, H, W, C, generator=g, dtype=torch.float64) * 1e-3
    d_out[torch.rand(B, H, W, generator=g) < 0.4] = 0                                        # background pixels
    if B > 1:
        uv[1], da[1] = uv[0], da[0]                                                             # two frames on the same texels
    uv_g, da_g, do_g = (x.float().cuda().contiguous() for x in (uv, da, d_out))
    tex = torch.zeros(1, Ht, Wt, C, device="cuda")
    nm = L.vhap_texture_mip_floats(1, Ht, Wt, C)
    mips = torch.zeros(nm, device="cuda")

    def run(binned):
        d_tex, d_mips = torch.zeros(Ht * Wt * C, device="cuda"), torch.zeros(max(nm, 1), device="cuda")
        if binned:
            assert L.vhap_texture_grad_binned(Ht, Wt, C, _p(uv_g), _p(da_g), _p(do_g), B, H, W, _p(d_tex), _p(d_mips),
                                              _p(work), work.numel(), _stream()) == 0
        else:
            assert L.vhap_texture_bwd(_p(tex), _p(mips), 1, Ht, Wt, C, _p(uv_g), _p(da_g), _p(do_g), B, H, W, _p(d_tex), _p(d_mips), 0, 0,
                                      _stream()) == 0
        lv = [d_tex.clone(), d_mips.clone()]
        if nm:
            assert L.vhap_texture_mip_fold(_p(d_tex), _p(d_mips), 1, Ht, Wt, C, 0, _stream()) == 0
        return lv, d_tex.view(Ht, Wt, C)

    work = torch.empty(L.vhap_texture_grad_binned_work_bytes(B, H, W), dtype=torch.uint8, device="cuda")
    (l_b, f_b), (l_t, f_t) = run(True), run(False)
    for a, b in zip(l_b, l_t):                                                                   # level by level, before the fold
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12
    l_b2, f_b2 = run(True)                                                                       # workspace reuse, determinism
    assert float((f_b2 - f_b).abs().max()) <= 1e-6 * float(f_b.abs().max())
    t_o = torch.zeros(1, Ht, Wt, C, dtype=torch.float64, requires_grad=True)
    (R.texture(t_o, uv, da) * d_out).sum().backward()
    assert _rel(f_b, t_o.grad[0]) < 1e-3
    if Ht == Wt:                                                                                 # the cached-workspace wrapper
        d_tex, d_mips = torch.zeros(Ht * Wt * C, device="cuda"), torch.zeros(max(nm, 1), device="cuda")
        assert texture_grad_binned(Ht, C, uv_g, da_g, do_g, d_tex, d_mips)
        assert torch.allclose(d_tex, l_b[0], rtol=1e-5, atol=1e-12)


def test_texture_expanded_batch_is_sampled_as_one_copy():
    """dr.texture(tex.expand(B, ...)) -- how the reference passes its single texture (tracker.py:234, render_nvdiffrast.py:398) -- must not
    materialise B copies: same values and the same (summed) gradient as B explicit copies, without the B-fold allocation."""
    from vhap_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W, T = 3, 24, 20, 64
    base = torch.rand(1, T, T, 3, generator=g).cuda()
    uv = torch.rand(B, H, W, 2, generator=g).cuda()
    da = (torch.randn(B, H, W, 4, generator=g) * 0.02).cuda()
    w = torch.randn(B, H, W, 3, generator=g).cuda()
    t1 = base.clone().requires_grad_()
    torch.cuda.reset_peak_memory_stats()
    m0 = torch.cuda.memory_allocated()
    out1 = ops.texture(t1.expand(B, -1, -1, -1), uv, da)
    (out1 * w).sum().backward()
    peak = torch.cuda.max_memory_allocated() - m0
    t2 = base.clone().requires_grad_()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    m1 = torch.cuda.memory_allocated()
    out2 = ops.texture(t2.expand(B, -1, -1, -1).contiguous(), uv, da)
    (out2 * w).sum().backward()
    peak_copies = torch.cuda.max_memory_allocated() - m1
    assert torch.equal(out1, out2)
    assert float((t1.grad - t2.grad).abs().max()) <= 1e-5 * float(t2.grad.abs().max())
    assert peak <= peak_copies - 2 * (B - 1) * base.numel() * 4, (peak, peak_copies)      # no B-fold texture (+ gradient) copies

