"""GPU parity: HIP rasterize / fused raster+interp vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_ref as R
from tests.scenes import head_scene, near_plane_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _poisoned_outputs():
    """every rasteriser output starts as NaN: since round 5 two launches share the stores (the binning launch's early stores outside the
    geometry box, the raster kernel inside it) -- a block neither writes must not inherit a correct value from recycled memory"""
    from vhap_amd import ops
    ops.POISON_OUTPUTS = True
    yield
    ops.POISON_OUTPUTS = False


def _gpu_raster(pos, tri, res):
    from vhap_amd import ops
    ctx = ops.RasterizeHipContext()
    rast, db = ops.raster_fwd(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), res)
    torch.cuda.synchronize()
    return rast.cpu().numpy(), db.cpu().numpy()


def _assert_raster_equal(got, ref, what):
    g_rast, g_db = got
    r_rast, r_db = ref
    # bit-exact: triangle ids, coverage, z/w, u, v (same IEEE op order on both sides)
    assert np.array_equal(g_rast[..., 3], r_rast[..., 3]), f"{what}: triangle ids differ"
    assert np.array_equal(g_rast.view(np.uint32), r_rast.view(np.uint32)), f"{what}: u/v/zw bits differ"
    np.testing.assert_allclose(g_db, r_db, rtol=0, atol=0, err_msg=f"{what}: rast_db")


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (2, 512, 512), (1, 550, 802), (1, 1024, 1024), (3, 100, 60)])
def test_head_matches_oracle_bit_exact(flame_model, B, H, W):
    model, topo = flame_model
    sc = head_scene(model, B, H, W, seed=B + H)
    pos = sc["clip"].numpy().astype(np.float32)
    tri = topo.faces.astype(np.int32)
    ref = oracle.rasterize(pos, tri, (H, W))
    got = _gpu_raster(pos, tri, (H, W))
    _assert_raster_equal(got, ref, f"head {B}x{H}x{W}")
    cov = (ref[0][..., 3] > 0).mean()
    assert 0.02 < cov < 0.9


def test_known_answers_quad_tiebreak_cull():
    H = W = 64
    # two coplanar overlapping triangles (tie -> lowest id), one back-facing, one behind far plane
    pos = np.array([[[-0.5, -0.5, 0.2, 1], [0.5, -0.5, 0.2, 1], [0.5, 0.5, 0.2, 1], [-0.5, 0.5, 0.2, 1],
                     [-1, -1, 0.9, 1], [1, -1, 0.9, 1], [1, 1, 0.9, 1], [-1, 1, 0.9, 1],
                     [-1, -1, 2.0, 1], [1, -1, 2.0, 1], [1, 1, 2.0, 1]]], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 1], [4, 5, 6], [4, 6, 7], [8, 9, 10]], np.int32)
    ref = oracle.rasterize(pos, tri, (H, W))
    got = _gpu_raster(pos, tri, (H, W))
    _assert_raster_equal(got, ref, "known answers")
    ids = ref[0][0, :, :, 3]
    assert set(np.unique(ids)) == {1.0, 2.0, 5.0, 6.0}          # 3 loses the tie, 4 culled, 7 beyond far
    assert (ids > 0).all()                                       # full-screen quad is watertight


def test_huge_triangles_and_guard_band():
    H, W = 128, 96
    rng = np.random.default_rng(3)
    pos = (rng.standard_normal((2, 300, 4)) * np.array([30, 30, 0.5, 0]) + np.array([0, 0, 0, 2.0])).astype(np.float32)
    pos[:, :20, 3] = -1.0                      # behind the camera -> dropped
    pos[:, 20:30, 0] = 1e6                     # outside the guard band -> dropped
    tri = rng.integers(0, 300, (500, 3)).astype(np.int32)
    ref = oracle.rasterize(pos, tri, (H, W))
    got = _gpu_raster(pos, tri, (H, W))
    _assert_raster_equal(got, ref, "huge triangles")
    assert (ref[0][..., 3] > 0).mean() > 0.3


def test_brute_force_fallback_when_bins_overflow(flame_model):
    from vhap_amd import ops
    model, topo = flame_model
    H = W = 256
    sc = head_scene(model, 1, H, W, seed=5)
    pos = sc["clip"].numpy().astype(np.float32)
    tri = topo.faces.astype(np.int32)
    ref = oracle.rasterize(pos, tri, (H, W))
    ctx = ops.RasterizeHipContext(pairs_per_triangle=0, pairs_per_block=0)        # capacity 0 -> every tile brute-forces
    rast, db = ops.raster_fwd(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), (H, W))
    _assert_raster_equal((rast.cpu().numpy(), db.cpu().numpy()), ref, "fallback")


def _crossing_stats(pos, tri, ids):
    """(# triangles crossing the near plane that WON pixels, # of those with a vertex behind the camera, # pixels they won), frame 0"""
    behind = ((pos[0, :, 2] + pos[0, :, 3]) < 0)[tri]
    nb = behind.sum(-1)
    won = np.unique(ids[0][ids[0] > 0]).astype(int) - 1
    cw = np.intersect1d(won, np.nonzero((nb > 0) & (nb < 3))[0])
    return len(cw), int((pos[0][tri[cw]][..., 3] < 0).any(-1).sum()), int(np.isin(ids[0].astype(int) - 1, cw).sum())


@pytest.mark.parametrize("path", ["fragmented", "overflow", "lists", "brute"])
def test_near_plane_clipping_matches_oracle_bit_exact(path):
    """Triangles crossing the near plane are cut into one or two pieces that carry the triangle's id (nvdiffrast clips in homogeneous clip
    space; round 1 dropped them).  Every binning path of the rasteriser -- one-launch fragmented lists, workgroups whose pairs overflow
    their region (direct scan of their record slots), the three-launch contiguous lists and the brute-force fallback -- against the C oracle
    on a scene whose cut runs across the screen (tests/scenes.py: near_plane_scene)."""
    from vhap_amd import _lib, ops
    B, H, W = 3, 256, 208
    pos, tri, _ = near_plane_scene(B)
    ref = oracle.rasterize(pos, tri, (H, W))
    n_cross, n_wneg, n_px = _crossing_stats(pos, tri, ref[0][..., 3])
    assert n_cross >= 30 and n_wneg >= 3 and n_px > 10000, (n_cross, n_wneg, n_px)
    kw = dict(fragmented={}, overflow=dict(pairs_per_triangle=1, pairs_per_block=0), lists={}, brute=dict(pairs_per_triangle=0, pairs_per_block=0))[path]
    ctx = ops.RasterizeHipContext(**kw)
    if path == "lists":
        _lib.debug_set_flags(4096)                                            # A/B switch: bin_count / bin_scan / bin_fill instead of bin_build
    try:
        rast, db = ops.raster_fwd(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), (H, W))
        torch.cuda.synchronize()
    finally:
        _lib.debug_set_flags(0)
    _assert_raster_equal((rast.cpu().numpy(), db.cpu().numpy()), ref, f"near-plane clipping ({path})")


def test_near_plane_clipping_fused_gbuffer_matches_oracle():
    """The fused G-buffer pass on the same scene: normals / uv / uv derivatives of a pixel won by a PIECE are interpolated with the
    barycentrics of the original triangle (homogeneous formulas, valid on both sides of w = 0)."""
    from vhap_amd import ops
    B, H, W = 2, 256, 208
    pos, tri, _ = near_plane_scene(B, seed=1)
    V = pos.shape[1]
    rng = np.random.default_rng(5)
    uv = rng.random((V + 7, 2)).astype(np.float32)
    tri_uv = ((tri.astype(np.int64) * 7 + 3) % (V + 7)).astype(np.int32)
    vn = rng.standard_normal((B, V, 3)).astype(np.float32)
    r_rast, r_db = oracle.rasterize(pos, tri, (H, W))
    r_n, _ = oracle.interpolate(vn, r_rast, tri)
    r_tc, r_td = oracle.interpolate(uv[None], r_rast, tri_uv, r_db)
    ctx = ops.RasterizeHipContext()
    out = ops.raster_interp_fwd(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), torch.from_numpy(vn).cuda(),
                                torch.from_numpy(uv).cuda(), torch.from_numpy(tri_uv).cuda(), (H, W))
    rast, db, normal, texc, texd = [o.cpu().numpy() for o in out]
    _assert_raster_equal((rast, db), (r_rast, r_db), "fused near-plane")
    assert np.array_equal(normal, r_n) and np.array_equal(texc, r_tc) and np.array_equal(texd, r_td)


@pytest.mark.parametrize("B,H,W", [(16, 512, 512), (8, 1024, 1024), (16, 802, 550)])
def test_baseline_sizes_match_oracle_bit_exact(flame_model, B, H, W):
    """BASELINE configs 2, 3 and 4 at their FULL batch shapes (the code that only behaves differently at size: each XCD owning whole
    frames needs B >= 8, the per-frame pair-list regions, the fragment walk over thousands of bins): plain rasteriser AND the fused
    G-buffer pass against the C oracle, every output bit for bit."""
    from vhap_amd import ops
    model, topo = flame_model
    sc = head_scene(model, B, H, W, seed=100 + B)
    pos = sc["clip"].numpy().astype(np.float32)
    tri = topo.faces.astype(np.int32)
    tri_uv = topo.faces_uv.astype(np.int32)
    uv = topo.verts_uvs.astype(np.float32).copy()
    uv[:, 1] = 1 - uv[:, 1]
    vn = R.compute_v_normals(sc["verts"], torch.from_numpy(topo.faces.astype(np.int64))).numpy().astype(np.float32)
    r_rast, r_db = oracle.rasterize(pos, tri, (H, W))
    _assert_raster_equal(_gpu_raster(pos, tri, (H, W)), (r_rast, r_db), f"plain {B}x{H}x{W}")
    r_n, _ = oracle.interpolate(vn, r_rast, tri)
    r_tc, r_td = oracle.interpolate(uv[None], r_rast, tri_uv, r_db)
    c = lambda a: torch.from_numpy(a).cuda()
    ctx = ops.RasterizeHipContext()
    for rep in range(2):                                        # second pass: workspace reuse (VHAP_RASTER_WS_CLEAN path)
        rast, db, normal, texc, texd = ops.raster_interp_fwd(ctx, c(pos), c(tri), c(vn), c(uv), c(tri_uv), (H, W))
        _assert_raster_equal((rast.cpu().numpy(), db.cpu().numpy()), (r_rast, r_db), f"fused {B}x{H}x{W} pass {rep}")
        assert np.array_equal(normal.cpu().numpy(), r_n)
        assert np.array_equal(texc.cpu().numpy(), r_tc)
        assert np.array_equal(texd.cpu().numpy(), r_td)
    cov = (r_rast[..., 3] > 0).reshape(B, -1).mean(1)
    assert (cov > 0.02).all() and (cov < 0.9).all()


@pytest.mark.parametrize("B,H,W", [(2, 256, 256), (1, 550, 802)])
def test_fused_raster_interp_matches_oracle(flame_model, B, H, W):
    from vhap_amd import ops
    model, topo = flame_model
    sc = head_scene(model, B, H, W, seed=11)
    pos = sc["clip"].numpy().astype(np.float32)
    tri = topo.faces.astype(np.int32)
    tri_uv = topo.faces_uv.astype(np.int32)
    uv = topo.verts_uvs.astype(np.float32).copy()
    uv[:, 1] = 1 - uv[:, 1]                                    # tracker.py:315-316
    vn = R.compute_v_normals(sc["verts"], torch.from_numpy(topo.faces.astype(np.int64))).numpy().astype(np.float32)
    r_rast, r_db = oracle.rasterize(pos, tri, (H, W))
    r_n, _ = oracle.interpolate(vn, r_rast, tri)
    r_tc, r_td = oracle.interpolate(uv[None], r_rast, tri_uv, r_db)
    ctx = ops.RasterizeHipContext()
    c = lambda a: torch.from_numpy(a).cuda()
    rast, db, normal, texc, texd = ops.raster_interp_fwd(ctx, c(pos), c(tri), c(vn), c(uv), c(tri_uv), (H, W))
    _assert_raster_equal((rast.cpu().numpy(), db.cpu().numpy()), (r_rast, r_db), "fused")
    assert np.array_equal(normal.cpu().numpy(), r_n)
    assert np.array_equal(texc.cpu().numpy(), r_tc)
    assert np.array_equal(texd.cpu().numpy(), r_td)


def test_bad_arguments_raise():
    from vhap_amd import ops, _lib
    ctx = ops.RasterizeHipContext()
    pos = torch.zeros(1, 4, 4, device="cuda")
    tri = torch.zeros(2, 3, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ops.raster_fwd(ctx, pos, tri, (5000, 64))              # H > limit
    with pytest.raises(TypeError):
        ops.raster_fwd(ctx, pos, tri.long(), (64, 64))
    with pytest.raises(RuntimeError):
        ops.raster_fwd(ctx, pos.cpu(), tri.cpu(), (64, 64))
    assert _lib.lib().vhap_raster_fwd(0, 0, 1, 1, 1, 8, 8, 0, 0, 0, 0, 0, 0, 0) == -1


def test_early_stores_outside_the_geometry_box(flame_model):
    """VHAP_RASTER_PREFILL (round 5, opt-in: a measured loss, csrc/raster.hip): the binning launch stores every 8x8 block outside the frame's
    vertex bounding box, the raster kernel skips those blocks.  (a) same bits with the early stores on and off, kernel modes 0 and 1
    through the ops (mode 2: tests/test_deferred_gpu.py under VHAP_PREFILL=1);
    (b) the binning launch ALONE (vhap_raster_bin_vnormal_prefill on NaN-filled outputs) writes a large part of the frame, only
    background, exactly the composite the raster kernel would have written; (c) a frame with a vertex behind the near plane is left to
    the raster kernel entirely."""
    import ctypes
    from vhap_amd import _lib, ops
    model, topo = flame_model
    B, H, W = 3, 256, 200
    sc = head_scene(model, B, H, W, seed=11)
    pos = sc["clip"].numpy().astype(np.float32).copy()
    pos[2, 17, 2] = -pos[2, 17, 3] - 0.5                          # frame 2: one vertex behind the near plane -> no statement about it
    tri = topo.faces.astype(np.int32)
    ref = oracle.rasterize(pos, tri, (H, W))
    got_off = _gpu_raster(pos, tri, (H, W))
    ctx = ops.RasterizeHipContext()
    acquire = ctx.acquire
    ctx.acquire = lambda *a: acquire(*a)[:3] + (acquire(*a)[3] | 32,)      # VHAP_RASTER_PREFILL on the one-call path
    rast, db = ops.raster_fwd(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), (H, W))
    torch.cuda.synchronize()
    _assert_raster_equal((rast.cpu().numpy(), db.cpu().numpy()), ref, "early stores on")
    _assert_raster_equal(got_off, ref, "early stores off")
    vn_ = torch.randn(B, pos.shape[1], 3, device="cuda")
    uv_ = torch.rand(pos.shape[1], 2, device="cuda")
    g_on = ops.raster_interp_fwd(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), vn_, uv_, torch.from_numpy(tri).cuda(), (H, W))
    g_off = ops.raster_interp_fwd(ops.RasterizeHipContext(), torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), vn_, uv_,
                                  torch.from_numpy(tri).cuda(), (H, W))
    for a_, b_ in zip(g_on, g_off):
        assert torch.equal(a_, b_) and not bool(torch.isnan(a_).any())
    # (b), (c): the binning launch alone
    L = _lib.lib()
    dev = "cuda"
    V, F = pos.shape[1], tri.shape[0]
    from vhap_amd.fused import MeshCSR
    csr = MeshCSR.from_faces(torch.from_numpy(tri).cuda())
    ctx = ops.RasterizeHipContext()
    ws, nbytes, cap, flags = ctx.acquire(B, F, H, W, torch.device(dev))
    t = lambda a: torch.from_numpy(a).to(dev)
    posd, trid = t(pos), t(tri)
    verts = posd[..., :3].contiguous()
    vn, inv = torch.empty(B, V, 3, device=dev), torch.empty(B, V, device=dev)
    bg = torch.rand(B, 3, H, W, device=dev)
    fid2cid = torch.arange(F + 1, dtype=torch.int32, device=dev) % 7 + 1
    nan = float("nan")
    rast, rgba = torch.full((B, H, W, 4), nan, device=dev), torch.full((B, H, W, 4), nan, device=dev)
    cid = torch.full((B, H, W), 99, dtype=torch.uint8, device=dev)
    tile_ids = torch.full((B, H, W), 1234, dtype=torch.int16, device=dev)
    p = lambda x: x.data_ptr()
    rc = L.vhap_raster_bin_vnormal_prefill(p(posd), p(trid), p(trid), B, V, F, H, W, p(ws), nbytes, cap, flags, p(verts), p(csr.ptr), p(csr.idx),
                                           p(vn), p(inv), p(bg), 0, p(fid2cid), F + 1, p(rast), p(rgba), p(cid), p(tile_ids),
                                           torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "vhap_raster_bin_vnormal_prefill")
    torch.cuda.synchronize()
    written = ~torch.isnan(rast[..., 0])
    frac = written.float().mean(dim=(1, 2)).cpu().numpy()
    assert 0.25 < frac[0] < 0.95 and 0.25 < frac[1] < 0.95, frac
    assert frac[2] == 0.0, frac                                   # (c)
    ids = torch.from_numpy(ref[0][..., 3]).to(dev)
    assert bool((ids[written] == 0).all()), "an early store landed on a covered pixel"
    assert bool((rast[written] == 0).all())
    want = torch.cat([bg.permute(0, 2, 3, 1).flip(1), torch.zeros(B, H, W, 1, device=dev)], dim=-1)
    assert torch.equal(rgba[written], want[written])
    assert bool((cid[written] == int(fid2cid[0])).all()) and bool((cid[~written] == 99).all())
    assert bool((tile_ids[written] == -1).all()) and bool((tile_ids[~written] == 1234).all())
    assert bool(torch.isnan(rgba[~written]).all())
    # whole 8x8 blocks only
    blk = written[:, :H - H % 8, :W - W % 8].reshape(B, H // 8, 8, W // 8, 8).float().mean(dim=(2, 4))
    assert bool(((blk == 0) | (blk == 1)).all())
