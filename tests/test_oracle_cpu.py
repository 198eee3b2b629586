"""CPU tests: the oracle and the product's host-side mirrors against golden vectors produced by the
reference's own Python code (tools/make_golden.py), plus the known-answer tests we had to author for the
raster conventions (the reference has none -- SURVEY.md section 4)."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_ref as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz"))
T = lambda k: torch.from_numpy(GOLD[k])


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_lbs_matches_reference_lbs_py(impl):
    if impl == "oracle":
        rod, lbs_fn, lm_fn = R.batch_rodrigues, R.lbs, R.vertices2landmarks
        bs = lambda b, s: torch.einsum("bl,mkl->bmk", b, s)
    else:
        from vhap_amd import lbs as L
        rod, lbs_fn, lm_fn, bs = L.batch_rodrigues, L.lbs, L.vertices2landmarks, L.blend_shapes
    assert torch.allclose(rod(T("pose").view(-1, 3)), T("rodrigues"), atol=1e-6)
    v_shaped = T("v_template")[None] + bs(T("betas"), T("shapedirs"))
    assert torch.allclose(v_shaped, T("v_shaped"), atol=1e-6)
    verts, Jt, A1 = lbs_fn(T("pose"), v_shaped, T("posedirs"), T("J_regressor"), T("parents"), T("lbs_weights"))
    assert torch.allclose(verts, T("verts"), atol=2e-6)
    assert torch.allclose(Jt, T("J_transformed"), atol=2e-6)
    assert torch.allclose(A1, T("A1"), atol=2e-6)
    if impl == "oracle":
        lm = lm_fn(T("verts"), T("faces"), T("lmk_idx")[0], T("lmk_bary")[0])
    else:
        lm = lm_fn(T("verts"), T("faces"), T("lmk_idx"), T("lmk_bary"))
    assert torch.allclose(lm, T("lmks"), atol=1e-6)


def test_lbs_identity_known_answer():
    z = torch.zeros(2, 15)
    v = T("v_template")[None].repeat(2, 1, 1)
    verts, _, _ = R.lbs(z, v, T("posedirs"), T("J_regressor"), T("parents"), T("lbs_weights"))
    assert torch.allclose(verts, v, atol=1e-6)                 # zero pose -> template (to the 1e-8 epsilon)


def test_rodrigues_matches_scipy():
    from scipy.spatial.transform import Rotation
    r = torch.randn(20, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    assert np.allclose(R.batch_rodrigues(r).numpy(), Rotation.from_rotvec(r.numpy()).as_matrix(), atol=1e-6)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_camera_and_sh_match_reference(impl):
    if impl == "oracle":
        proj = R.projection_from_intrinsics
        sh = lambda n, l: R.get_SH_shading(n, l, T("sh_const"))
        mvp = lambda RT, K, s: R.projection_from_intrinsics(K, s) @ R._mv(RT)
    else:
        from vhap_amd.render_hip import HipDiffRenderer, get_SH_shading
        r = HipDiffRenderer(lighting_type="SH")
        proj, mvp = r.projection_from_intrinsics, r.mvp_from_camera_param
        sh = lambda n, l: get_SH_shading(n, l, r.sh_const)
        assert torch.allclose(r.sh_const, T("sh_const"), atol=1e-7)
    assert torch.allclose(proj(T("K4"), (512, 512)), T("P4"), atol=1e-6)
    assert torch.allclose(proj(T("K33"), (550, 802)), T("P33"), atol=1e-6)
    assert torch.allclose(mvp(T("RT"), T("K4"), (512, 512)), T("mvp"), atol=1e-5)
    assert torch.allclose(sh(T("normals"), T("lights")), T("sh"), atol=1e-5)
    with pytest.raises(ValueError):
        proj(torch.zeros(1, 5), (4, 4))


def test_sh_uniform_light_is_one_and_projection_values():
    lights = torch.zeros(1, 9, 3)
    lights[0, 0] = np.sqrt(4 * np.pi)
    n = torch.nn.functional.normalize(torch.randn(1, 3, 3, 3), dim=-1)
    assert torch.allclose(R.get_SH_shading(n, lights, R.sh_const()), torch.ones(1, 3, 3, 3), atol=1e-6)
    P = R.projection_from_intrinsics(torch.tensor([[768.0, 768.0, 256.0, 256.0]]), (512, 512))
    assert abs(float(P[0, 0, 0]) - 3.0) < 1e-6 and abs(float(P[0, 2, 2]) + 1.0202) < 1e-4 and abs(float(P[0, 2, 3]) + 0.20202) < 1e-4


def test_normalize_image_points_matches_reference():
    from vhap_amd.tracker import normalize_image_points
    u, v = normalize_image_points(torch.tensor([0.0, 100.0, 512.0]), torch.tensor([10.0, 256.0, 300.0]), (512, 400))
    assert torch.allclose(u, T("norm_u")) and torch.allclose(v, T("norm_v"))


# ---- raster known answers (conventions of oracle/raster_oracle.c) ----

def _tri(pos, tri, res):
    return oracle.rasterize(np.asarray(pos, np.float32)[None], np.asarray(tri, np.int32), res)


def test_raster_single_triangle_analytic():
    H = W = 32
    pos = [[-0.5, -0.5, 0.25, 1], [0.5, -0.5, 0.25, 1], [-0.5, 0.5, 0.25, 1]]
    rast, db = _tri(pos, [[0, 1, 2]], (H, W))
    ys, xs = np.nonzero(rast[0, :, :, 3])
    fx, fy = (2 * xs + 1) / W - 1, (2 * ys + 1) / H - 1
    # coverage: pixel centres with x > -0.5, y > -0.5, x + y < 0 (edges excluded/included by the fill rule)
    inside = (fx > -0.5) & (fy > -0.5) & (fx + fy < 0)
    assert inside.all()
    u, v = rast[0, ys, xs, 0], rast[0, ys, xs, 1]
    assert np.allclose(u, 1 - (fx + 0.5) - (fy + 0.5), atol=1e-6)      # weight of vertex 0
    assert np.allclose(v, fx + 0.5, atol=1e-6)                          # weight of vertex 1
    assert np.allclose(rast[0, ys, xs, 2], 0.25, atol=1e-7)
    assert np.allclose(db[0, ys, xs], [-2 / W, -2 / H, 2 / W, 0], atol=1e-7)


def test_raster_tiebreak_cull_depth_and_watertight():
    H = W = 64
    quad = [[-1, -1, 0.9, 1], [1, -1, 0.9, 1], [1, 1, 0.9, 1], [-1, 1, 0.9, 1]]
    rast, _ = _tri(quad, [[0, 1, 2], [0, 2, 3]], (H, W))
    ids = rast[0, :, :, 3]
    assert (ids > 0).all() and set(np.unique(ids)) == {1.0, 2.0}        # no holes, no double hits by construction
    rast, _ = _tri(quad, [[0, 2, 1]], (H, W))
    assert (rast[0, :, :, 3] == 0).all()                               # clockwise = back-facing = culled
    two = quad + [[-1, -1, 0.1, 1], [1, -1, 0.1, 1], [1, 1, 0.1, 1]]
    rast, _ = _tri(two, [[0, 1, 2], [4, 5, 6], [4, 5, 6]], (H, W))
    ids = rast[0, :, :, 3]
    assert (ids > 0).sum() > H * W // 3 and set(np.unique(ids[ids > 0])) == {2.0}   # nearer wins; exact tie -> lowest id
    far = [[-1, -1, 1.5, 1], [1, -1, 1.5, 1], [1, 1, 1.5, 1]]
    rast, _ = _tri(far, [[0, 1, 2]], (H, W))
    assert (rast[0, :, :, 3] == 0).all()                               # beyond the far plane


def test_raster_shared_edge_random_meshes_have_no_double_coverage():
    rng = np.random.default_rng(0)
    H, W = 48, 40
    for _ in range(5):
        n = 8
        gx, gy = np.meshgrid(np.linspace(-0.9, 0.9, n), np.linspace(-0.9, 0.9, n))
        p = np.stack([gx + rng.normal(0, 0.03, gx.shape), gy + rng.normal(0, 0.03, gy.shape)], -1).reshape(-1, 2)
        pos = np.concatenate([p, np.zeros((n * n, 1)), np.ones((n * n, 1))], 1)
        tri = []
        for j in range(n - 1):
            for i in range(n - 1):
                a = j * n + i
                tri += [[a, a + 1, a + n + 1], [a, a + n + 1, a + n]]
        rast, _ = _tri(pos, tri, (H, W))
        # every triangle alone covers a set; the sets must partition the union exactly (no overlap)
        cover = np.zeros((H, W), int)
        for t in tri:
            r1, _ = _tri(pos, [t], (H, W))
            cover += (r1[0, :, :, 3] > 0)
        assert cover.max() == 1
        assert ((cover > 0) == (rast[0, :, :, 3] > 0)).all()


def _clip_from_camera(pc, focal=1.5, near=0.1, far=10.0):
    """OpenGL projection of camera-space points (camera looks down -z; render_nvdiffrast.py:102-139 with c at the image centre)."""
    pc = np.asarray(pc, np.float64)
    A, Bz = -(far + near) / (far - near), -2 * far * near / (far - near)
    return np.stack([2 * focal * pc[:, 0], 2 * focal * pc[:, 1], A * pc[:, 2] + Bz, -pc[:, 2]], 1)


def test_raster_near_plane_clipping_floor_known_answer():
    """A floor that runs from in front of the camera to BEHIND it (vertices with w < 0): coverage, barycentrics and depth against the
    ray / plane intersection.  Before clipping existed the whole floor was dropped (any w <= 0)."""
    H = W = 96
    focal, near = 1.5, 0.1
    y0 = -0.2
    cam = np.array([[-5, y0, -5], [5, y0, -5], [5, y0, 1], [-5, y0, 1]], np.float64)     # z = +1: one metre behind the camera
    pos = _clip_from_camera(cam, focal, near)
    tri = [[0, 2, 1], [0, 3, 2]]              # winding: front-facing seen from above
    rast, db = _tri(pos, tri, (H, W))
    ids = rast[0, :, :, 3]
    ys, xs = np.mgrid[0:H, 0:W]
    fx, fy = (2 * xs + 1) / W - 1, (2 * ys + 1) / H - 1
    with np.errstate(divide="ignore", invalid="ignore"):
        s = y0 * 2 * focal / fy                # distance along -z at which the pixel's ray meets the floor
        xh = s * fx / (2 * focal)
    hit = (fy < 0) & (s >= near) & (s <= 5) & (np.abs(xh) <= 5)
    margin = (fy < 0) & (s >= near * 0.9) & (s <= 5.3) & (np.abs(xh) <= 5.3)
    assert (ids[hit & (s > near * 1.1) & (s < 4.7) & (np.abs(xh) < 4.7)] > 0).all()      # covered where the floor is visible
    assert (ids[~margin] == 0).all()                                                        # and nowhere else (nothing mirrored from w < 0)
    assert (ids > 0).sum() > 1000 and set(np.unique(ids)) == {0.0, 1.0, 2.0}
    # barycentrics are those of the ORIGINAL triangles: interpolating the camera-space vertices gives the ray / floor intersection
    b0, b1 = rast[0, :, :, 0].astype(np.float64), rast[0, :, :, 1].astype(np.float64)
    T = np.asarray(tri)[np.maximum(ids.astype(int) - 1, 0)]
    P = b0[..., None] * cam[T[..., 0]] + b1[..., None] * cam[T[..., 1]] + (1 - b0 - b1)[..., None] * cam[T[..., 2]]
    m = (ids > 0) & hit & (s > 2 * near)
    assert np.abs(P[m][:, 2] + s[m]).max() < 2e-3 * 5 and np.abs(P[m][:, 0] - xh[m]).max() < 2e-3 * 5
    A, Bz = -(10 + near) / (10 - near), -2 * 10 * near / (10 - near)
    assert np.abs(rast[0, :, :, 2][m] - (A * -s[m] + Bz) / s[m]).max() < 1e-4
    # a floor entirely behind the camera draws nothing; one entirely in front is untouched by the clipper
    behind = _clip_from_camera(cam + [0, 0, 6.5], focal, near)
    assert (_tri(behind, tri, (H, W))[0][..., 3] == 0).all()


def test_raster_near_plane_clipping_is_watertight():
    """Cut points are computed from the vertex in front towards the vertex behind, so two triangles sharing an edge share the cut point:
    a tilted random mesh through the near plane has no hole and no double hit along the shared edges or inside the two-piece triangles."""
    rng = np.random.default_rng(1)
    H, W = 64, 56
    for trial in range(4):
        n = 7
        gx, gz = np.meshgrid(np.linspace(-0.6, 0.6, n), np.linspace(-1.2, 0.3, n))       # z up to +0.3: behind the camera
        gx = gx + rng.normal(0, 0.02, gx.shape)
        gz = gz + rng.normal(0, 0.03, gz.shape)
        cam = np.stack([gx, -0.15 + 0.1 * gx + rng.normal(0, 0.003, gx.shape), gz], -1).reshape(-1, 3)
        pos = _clip_from_camera(cam)
        tri = []
        for j in range(n - 1):
            for i in range(n - 1):
                a = j * n + i
                tri += [[a, a + n + 1, a + 1], [a, a + n, a + n + 1]]
        rast, _ = _tri(pos, tri, (H, W))
        full = rast[0, :, :, 3] > 0
        assert full.sum() > 200
        cover = np.zeros((H, W), int)
        for t in tri:
            r1, _ = _tri(pos, [t], (H, W))
            cover += (r1[0, :, :, 3] > 0)
        assert cover.max() == 1
        assert ((cover > 0) == full).all()
        # no hole strictly inside the covered region: every covered row is one contiguous run (the mesh is convex in screen space per row)
        for y in range(H):
            xs = np.nonzero(full[y])[0]
            if len(xs):
                assert len(xs) == xs[-1] - xs[0] + 1


def test_texture_known_answers_oracle():
    tex = torch.full((1, 16, 16, 3), 0.37, dtype=torch.float64)
    uv = torch.rand(1, 4, 4, 2, dtype=torch.float64)
    for s in (1e-4, 1e-2, 0.1, 1.0):
        da = torch.full((1, 4, 4, 4), s, dtype=torch.float64)
        assert torch.allclose(R.texture(tex, uv, da), torch.full((1, 4, 4, 3), 0.37, dtype=torch.float64))
    yy, xx = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij")
    cb = ((xx + yy) % 2).double()[None, :, :, None]
    da = torch.zeros(1, 4, 4, 4, dtype=torch.float64)
    da[..., 0] = 2.0 / 16
    da[..., 3] = 2.0 / 16
    assert torch.allclose(R.texture(cb, uv, da), torch.full((1, 4, 4, 1), 0.5, dtype=torch.float64))


def test_oracle_backward_matches_finite_differences():
    """fp64 finite differences of the differentiable oracle ops (the backward oracle is autograd of these)."""
    g = torch.Generator().manual_seed(0)
    tex = torch.rand(1, 8, 8, 2, dtype=torch.float64, generator=g).requires_grad_()
    uv = (torch.rand(1, 3, 3, 2, dtype=torch.float64, generator=g) * 0.8 + 0.1).requires_grad_()
    da = (torch.rand(1, 3, 3, 4, dtype=torch.float64, generator=g) * 0.2 + 0.15).requires_grad_()
    assert torch.autograd.gradcheck(lambda t, u, d: R.texture(t, u, d), (tex, uv, da), eps=1e-7, atol=1e-5)
    pos = torch.tensor([[[-0.8, -0.7, 0.1, 1.0], [0.9, -0.6, 0.2, 1.2], [0.1, 0.8, 0.3, 0.9]]], dtype=torch.float64, requires_grad=True)
    tri = torch.tensor([[0, 1, 2]])
    r_np, _ = oracle.rasterize(pos.detach().float().numpy(), tri.int().numpy(), (6, 6))
    tid = torch.from_numpy(r_np[..., 3].astype(np.int64) - 1)
    f = lambda p: torch.cat([x[..., :3] if i == 0 else x for i, x in enumerate(R.rast_from_ids(p, tri, tid, (6, 6)))], -1)
    assert torch.autograd.gradcheck(f, (pos,), eps=1e-7, atol=1e-5)


def test_texture_oracle_against_independent_torch_implementations():
    """The texture restatement against implementations it shares no code with: the mip chain == repeated 2x2 average pooling, plain bilinear
    sampling == torch's grid_sample (align_corners=False: texel centres at half-integers) away from the wrap seam, and the trilinear blend ==
    the explicit lerp of two grid_sample look-ups at the level the Jacobian selects."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(4)
    T, C, B, H, W = 64, 3, 2, 9, 11
    tex = torch.rand(1, T, T, C, generator=g, dtype=torch.float64)
    mips = R.build_mips(tex)
    cur = tex.permute(0, 3, 1, 2)
    for l, m in enumerate(mips[1:], 1):
        cur = F.avg_pool2d(cur, 2)
        assert torch.allclose(m.permute(0, 3, 1, 2), cur, atol=1e-14), l
    uv = torch.rand(B, H, W, 2, generator=g, dtype=torch.float64) * 0.8 + 0.1

    def gs(level_img, uv_):                                       # [1,T,T,C] sampled at uv in [0,1]^2
        out = F.grid_sample(level_img.permute(0, 3, 1, 2).expand(B, -1, -1, -1), uv_ * 2 - 1, mode="bilinear", padding_mode="border",
                            align_corners=False)
        return out.permute(0, 2, 3, 1)
    assert torch.allclose(R.texture(tex, uv, None, filter_mode="linear"), gs(tex, uv), atol=1e-12)
    # an isotropic footprint of s texels per pixel selects level log2(s): blend levels floor and floor + 1
    for s in (1.0, 2.0, 3.0, 5.5):
        da = torch.zeros(B, H, W, 4, dtype=torch.float64)
        da[..., 0] = s / T                                       # du/dx
        da[..., 3] = s / T                                       # dv/dy
        level = np.log2(s)
        l0 = int(np.floor(level))
        f = level - l0
        want = gs(mips[l0], uv) * (1 - f) + gs(mips[l0 + 1], uv) * f if f > 0 else gs(mips[l0], uv)
        # (coarse levels: keep clear of the border where grid_sample clamps and the oracle wraps)
        inner = ((uv > 0.2) & (uv < 0.8)).all(-1)
        got = R.texture(tex, uv, da, filter_mode="linear-mipmap-linear")
        assert torch.allclose(got[inner], want[inner], atol=1e-10), s


def test_photometric_sign_from_takes_the_other_evaluations_subgradient_also_at_exact_zero():
    """oracle/torch_ref.photometric_energy(sign_from=...): the L1 term's derivative is the OTHER evaluation's sign wherever that is
    +-1 (a kink taken on its side) and ZERO where the other evaluation's residual is exactly zero -- sign(0) = 0 is what torch.abs and the
    HIP kernels take there; falling back to this evaluation's own sign put one pixel's whole contribution between the two gradients
    (round 4: d(tex_extra) 5e-3 .. 8e-3 apart at trained states, tools/trained_state_spread.py)."""
    from oracle import torch_ref as R
    gt = torch.full((1, 3, 1, 4), 0.5, dtype=torch.float64)
    pred = torch.tensor([0.5 + 3e-9, 0.5 - 2e-9, 0.7, 0.5 + 1e-9], dtype=torch.float64)
    rgba = torch.cat([pred.reshape(1, 1, 4, 1).expand(1, 1, 4, 3), torch.ones(1, 1, 4, 1, dtype=torch.float64)], -1).clone().requires_grad_()
    # the float32 evaluation of the same state: pixel 0 rounds to the other side, pixel 1 to the same, pixel 3 to EXACTLY zero
    other = torch.tensor([-1e-8, -1e-8, 0.2, 0.0]).reshape(1, 1, 4, 1).expand(1, 1, 4, 3)
    E = R.photometric_energy(gt, rgba, sign_from=other)
    E.backward()
    g = rgba.grad[0, 0, :, 0] * 12.0                      # (3 channels x 4 pixels with alpha > 0 in the denominator)
    assert torch.equal(g, torch.tensor([-1.0, -1.0, 1.0, 0.0], dtype=torch.float64))
    E0 = R.photometric_energy(gt, rgba.detach())
    assert abs(float(E) - float(E0)) < 1e-8               # the value moves by <= 2 |x| per kink element
