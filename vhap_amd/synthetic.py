"""Seeded synthetic FLAME-shaped model and scenes.

The licensed FLAME weights (flame2023.pkl, FLAME_masks.pkl; vhap/model/flame.py:38-40) are not
redistributable, so benchmarks and tests run a *synthetic* linear model on the true FLAME
topology / UV layout / landmark embedding: random shape/expression/pose bases of the right shape,
smooth skinning weights, 5 joints with the FLAME kinematic tree.  All arrays are float32 numpy.
"""
import numpy as np

from .topology import FlameTopology

N_SHAPE, N_EXPR = 300, 100
PARENTS = np.array([-1, 0, 1, 1, 1], np.int64)          # root, neck, jaw, eye_L, eye_R


def make_flame_model(seed=0, add_teeth=True, n_shape=N_SHAPE, n_expr=N_EXPR, topo=None):
    """Returns (model dict of numpy arrays, FlameTopology)."""
    rng = np.random.default_rng(seed)
    topo = topo or FlameTopology(add_teeth=add_teeth)
    v = topo.v_template_obj.copy()
    v -= v.mean(0, keepdims=True)                                     # centre the head
    nv0 = v.shape[0]
    NB = n_shape + n_expr
    # smooth random bases: low-frequency Fourier features of the vertex position, so that shape /
    # expression / pose-corrective displacements deform the head smoothly (mesh stays mesh-like)
    def smooth_basis(n, amp, fmin, fmax):
        Wf = rng.standard_normal((n, 3, 3)).astype(np.float32)
        Wf *= (2 * np.pi / 0.25) * rng.uniform(fmin, fmax, (n, 1, 1)).astype(np.float32) / np.sqrt(3)
        ph = rng.uniform(0, 2 * np.pi, (n, 3)).astype(np.float32)
        a = (rng.standard_normal((n, 3)) * amp).astype(np.float32)
        return a[None] * np.sin(np.einsum("vk,nck->vnc", v, Wf) + ph[None])         # [V,n,3]
    shapedirs = np.concatenate([smooth_basis(n_shape, 2e-3, 0.3, 2.0), smooth_basis(n_expr, 2e-3, 0.5, 3.0)], 1)
    shapedirs = np.ascontiguousarray(shapedirs.transpose(0, 2, 1)).astype(np.float32)      # [V,3,NB]
    posedirs = np.ascontiguousarray(smooth_basis(36, 3e-3, 0.3, 2.0).transpose(1, 0, 2)).reshape(36, nv0 * 3).astype(np.float32)

    # joints: anchor points in the centred template frame (x right, y up, z front)
    ymin, ymax = v[:, 1].min(), v[:, 1].max()
    h = ymax - ymin
    eye_l = v[topo.v_regions["left_eyeball"]].mean(0) if "left_eyeball" in topo.v_regions else np.array([0.03, 0.03, 0.05])
    eye_r = v[topo.v_regions["right_eyeball"]].mean(0) if "right_eyeball" in topo.v_regions else np.array([-0.03, 0.03, 0.05])
    anchors = np.array([
        [0.0, ymin + 0.10 * h, -0.02],     # root (neck base)
        [0.0, ymin + 0.30 * h, -0.01],     # neck
        [0.0, ymin + 0.48 * h, 0.00],      # jaw hinge
        eye_l, eye_r,
    ], np.float32)
    J_regressor = np.zeros((5, nv0), np.float32)
    for j in range(5):
        d2 = ((v - anchors[j]) ** 2).sum(1)
        nn = np.argsort(d2)[:32]
        w = rng.random(32).astype(np.float32) + 0.1
        J_regressor[j, nn] = w / w.sum()
    joints = J_regressor @ v
    # skinning: eyeballs rigid to their joint, the rest a smooth blend of root / neck / jaw
    d2 = ((v[:, None, :] - joints[None, :3, :]) ** 2).sum(-1)
    logits = -d2 / (2 * (0.06 ** 2))
    logits[:, 1] += 1.0                                               # most of the head follows the neck
    e = np.exp(logits - logits.max(1, keepdims=True))
    w3 = e / e.sum(1, keepdims=True)
    lbs_weights = np.zeros((nv0, 5), np.float32)
    lbs_weights[:, :3] = w3
    for j, name in ((3, "left_eyeball"), (4, "right_eyeball")):
        if name in topo.v_regions:
            idx = topo.v_regions[name]
            lbs_weights[idx] = 0
            lbs_weights[idx, j] = 1

    arrays = dict(v_template=v.astype(np.float32), shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor, lbs_weights=lbs_weights)
    if topo.has_teeth:                                                # flame.py:206-325
        from .flame_assets import append_teeth
        arrays = append_teeth(arrays, topo, n_shape)
    v, shapedirs, posedirs = arrays["v_template"], arrays["shapedirs"], arrays["posedirs"]
    J_regressor, lbs_weights = arrays["J_regressor"], arrays["lbs_weights"]

    model = dict(
        v_template=v.astype(np.float32), shapedirs=shapedirs, posedirs=posedirs,
        J_regressor=J_regressor, parents=PARENTS.copy(), lbs_weights=lbs_weights,
        faces=topo.faces.astype(np.int64), faces_uv=topo.faces_uv.astype(np.int64),
        verts_uvs=topo.verts_uvs.copy(),
        lmk_faces_idx=topo.lmk_faces_idx.copy(), lmk_bary_coords=topo.lmk_bary_coords.copy(),
    )
    return model, topo


def smooth_noise(rng, shape, octaves=4):
    """Cheap smooth value noise in [0,1] of shape [..., H, W] via bilinear upsampling of coarse grids."""
    *lead, H, W = shape
    out = np.zeros(shape, np.float32)
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        n = 4 * (2 ** o)
        g = rng.random((*lead, n + 1, n + 1)).astype(np.float32)
        ys = np.linspace(0, n, H, endpoint=False)
        xs = np.linspace(0, n, W, endpoint=False)
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        Y0, X0 = y0[:, None], x0[None, :]
        a, b = g[..., Y0, X0], g[..., Y0, X0 + 1]
        c, d = g[..., Y0 + 1, X0], g[..., Y0 + 1, X0 + 1]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        tot += amp
        amp *= 0.5
    return out / tot


def make_texture(seed=0, size=2048):
    """Procedural stand-in for asset/flame/tex_mean_painted.png: [3,size,size] in [0,1]."""
    rng = np.random.default_rng(seed + 1000)
    base = np.array([0.78, 0.60, 0.52], np.float32)[:, None, None]
    n = smooth_noise(rng, (3, size, size), octaves=6)
    return np.clip(base + 0.35 * (n - 0.5), 0, 1).astype(np.float32)


def make_tex_space(seed=0, S=512, n=200):
    """Procedural stand-in for FLAME_texture.npz (flame.py:665-680): `mean` [S,S,3] and `tex_dir` [S,S,3,n] in 0..255 units, channels
    B G R -- a skin-coloured mean and smooth basis images of decaying amplitude; the mean is pushed outside 0..255 in two corners so that
    the model's clamp is exercised."""
    rng = np.random.default_rng(seed + 3000)
    mean = (np.array([132.0, 153.0, 199.0], np.float32)[None, None, :] +
            60.0 * (smooth_noise(rng, (3, S, S), octaves=4).transpose(1, 2, 0) - 0.5)).astype(np.float32)
    mean[: S // 16, : S // 16] += 150.0
    mean[-S // 16:, -S // 16:] -= 220.0
    tex_dir = np.empty((S, S, 3, n), np.float32)
    for k in range(n):
        lo = smooth_noise(rng, (3, S // 8, S // 8), octaves=3).transpose(1, 2, 0) - 0.5
        tex_dir[..., k] = np.kron(lo, np.ones((8, 8, 1), np.float32))[:S, :S] * (40.0 / (1.0 + 0.05 * k))
    return {"mean": mean, "tex_dir": tex_dir}


def make_scene_params(n_frames, seed=0, image_size=(512, 512), translation_z=0.45, n_shape=N_SHAPE, n_expr=N_EXPR):
    """Ground-truth per-frame FLAME parameters of a synthetic monocular video."""
    rng = np.random.default_rng(seed + 2000)
    t = np.linspace(0, 1, n_frames, dtype=np.float32)[:, None]
    ph = rng.random((1, 3)).astype(np.float32) * 6.28
    rotation = 0.3 * np.sin(2 * np.pi * t * np.array([[1.0, 0.7, 0.5]], np.float32) + ph)
    jaw = np.zeros((n_frames, 3), np.float32)
    jaw[:, 0] = 0.1 + 0.1 * np.sin(2 * np.pi * 2 * t[:, 0] + 1.0)
    return dict(
        shape=(rng.standard_normal(n_shape) * 0.5).astype(np.float32),
        expr=(rng.standard_normal((n_frames, n_expr)) * 0.5).astype(np.float32),
        rotation=rotation.astype(np.float32),
        neck_pose=(rng.standard_normal((n_frames, 3)) * 0.03).astype(np.float32),
        jaw_pose=jaw,
        eyes_pose=(rng.standard_normal((n_frames, 6)) * 0.05).astype(np.float32),
        translation=(rng.standard_normal((n_frames, 3)) * 0.01 + np.array([0, 0, translation_z])).astype(np.float32),
        lights=_lights(rng),
        focal_length=np.array([1.5], np.float32),
        image_size=np.array(image_size, np.int64),
    )


def _lights(rng):
    l = np.zeros((9, 3), np.float32)
    l[0] = np.sqrt(4 * np.pi)
    return (l + rng.standard_normal((9, 3)).astype(np.float32) * 0.1).astype(np.float32)


def monocular_camera(n, image_size, focal_length=1.5):
    """Uncalibrated camera of tracker.py:141-157,1335-1337: K=[f,f,cx,cy], RT=[I|(0,0,-1)]."""
    H, W = image_size
    f = focal_length * max(H, W)
    K = np.tile(np.array([[f, f, 0.5 * W, 0.5 * H]], np.float32), (n, 1))
    RT = np.zeros((n, 3, 4), np.float32)
    RT[:, :3, :3] = np.eye(3)
    RT[:, 2, 3] = -1
    return K, RT


def make_dataset(tracker_like_render, flame_head, gt, image_size, device, seed=0, lmk_noise_px=1.0, tex=None):
    """Render a synthetic monocular video from ground-truth parameters with the product renderer.
    Returns the in-memory dataset dict GlobalTracker expects: rgb [N,3,H,W], lmk2d [N,70,3]."""
    import torch
    H, W = image_size
    N = gt["expr"].shape[0]
    rng = np.random.default_rng(seed + 3000)
    g = lambda k: torch.from_numpy(gt[k]).to(device)
    with torch.no_grad():
        verts, lmks = flame_head(g("shape")[None].expand(N, -1), g("expr"), g("rotation"), g("neck_pose"), g("jaw_pose"),
                                 g("eyes_pose"), g("translation"))
        K, RT = monocular_camera(N, image_size, float(gt["focal_length"][0]))
        K, RT = torch.from_numpy(K).to(device), torch.from_numpy(RT).to(device)
        r = tracker_like_render
        bg = torch.from_numpy(smooth_noise(rng, (N, 3, H, W), octaves=4)).to(device).permute(0, 2, 3, 1).contiguous()
        rast = r.rasterize(verts, flame_head.faces, RT, K, image_size)
        uv = flame_head.verts_uvs.clone()
        uv[:, 1] = 1 - uv[:, 1]
        tex_t = torch.from_numpy(tex if tex is not None else make_texture(seed, 512)).to(device)[None]
        out = r.render_rgba(rast, verts, flame_head.faces, uv, flame_head.textures_idx, tex_t, g("lights")[None], bg)
        rgb = out["rgba"][..., :3].permute(0, 3, 1, 2).clamp(0, 1).contiguous()
        ndc = r.world_to_ndc(lmks, RT, K, image_size, flip_y=True)
        u = (ndc[..., 0] * 0.5 + 0.5) * W + torch.from_numpy(rng.standard_normal((N, lmks.shape[1])).astype(np.float32)).to(device) * lmk_noise_px
        v = (ndc[..., 1] * 0.5 + 0.5) * H + torch.from_numpy(rng.standard_normal((N, lmks.shape[1])).astype(np.float32)).to(device) * lmk_noise_px
        lmk2d = torch.stack([u, v, torch.ones_like(u)], dim=-1)
    return {"rgb": rgb, "lmk2d": lmk2d}


def arc_cameras(n_views, image_size, arc_deg=60.0, radius=1.0, focal_px=None):
    """Calibrated cameras of a NeRSemble-like rig: `n_views` cameras on a +-arc_deg arc of `radius` metres around
    the origin, looking at it.  Returns K [n,3,3] (pixels) and RT [n,3,4] (world -> camera, OpenGL: camera looks down -z)."""
    H, W = image_size
    # default focal length: the 0.31 m tall template head fills 60 % of the long image side at `radius`
    f = float(focal_px) if focal_px is not None else 0.6 * max(H, W) / 0.31 * radius
    Ks, RTs = [], []
    for a in np.linspace(-np.deg2rad(arc_deg), np.deg2rad(arc_deg), n_views):
        c, s = np.cos(a), np.sin(a)
        R = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], np.float32)               # rotation about the vertical axis
        RTs.append(np.concatenate([R, np.array([[0.0], [0.0], [-radius]], np.float32)], axis=1))
        Ks.append(np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]], np.float32))
    return np.stack(Ks), np.stack(RTs)


def make_multiview_dataset(render, flame_head, gt, image_size, device, n_views=16, seed=0, lmk_noise_px=1.0, tex=None, timestep=0):
    """All `n_views` calibrated views of ONE timestep of `gt`, rendered with the product renderer (BASELINE config 4).  Returns the dataset
    dict GlobalTracker expects for a multi-view capture: rgb [n,3,H,W], lmk2d [n,70,3], intrinsic [n,3,3], extrinsic [n,3,4],
    timestep_index [n] (all 0), camera_index [n]."""
    import torch
    H, W = image_size
    rng = np.random.default_rng(seed + 4000)
    g = lambda k: torch.from_numpy(gt[k]).to(device)
    t = slice(timestep, timestep + 1)
    K, RT = arc_cameras(n_views, image_size)
    with torch.no_grad():
        verts, lmks = flame_head(g("shape")[None], g("expr")[t], g("rotation")[t] * 0, g("neck_pose")[t], g("jaw_pose")[t], g("eyes_pose")[t],
                                 g("translation")[t] * 0)
        verts, lmks = verts.expand(n_views, -1, -1).contiguous(), lmks.expand(n_views, -1, -1).contiguous()
        Kt, RTt = torch.from_numpy(K).to(device), torch.from_numpy(RT).to(device)
        bg = torch.from_numpy(smooth_noise(rng, (n_views, 3, H, W), octaves=4)).to(device).permute(0, 2, 3, 1).contiguous()
        rast = render.rasterize(verts, flame_head.faces, RTt, Kt, image_size)
        uv = flame_head.verts_uvs.clone()
        uv[:, 1] = 1 - uv[:, 1]
        tex_t = torch.from_numpy(tex if tex is not None else make_texture(seed, 512)).to(device)[None]
        out = render.render_rgba(rast, verts, flame_head.faces, uv, flame_head.textures_idx, tex_t, g("lights")[None], bg)
        rgb = out["rgba"][..., :3].permute(0, 3, 1, 2).clamp(0, 1).contiguous()
        ndc = render.world_to_ndc(lmks, RTt, Kt, image_size, flip_y=True)
        noise = lambda: torch.from_numpy(rng.standard_normal((n_views, lmks.shape[1])).astype(np.float32)).to(device) * lmk_noise_px
        lmk2d = torch.stack([(ndc[..., 0] * 0.5 + 0.5) * W + noise(), (ndc[..., 1] * 0.5 + 0.5) * H + noise(), torch.ones_like(ndc[..., 0])], dim=-1)
    return {"rgb": rgb, "lmk2d": lmk2d, "intrinsic": Kt, "extrinsic": RTt,
            "timestep_index": torch.zeros(n_views, dtype=torch.long), "camera_index": torch.arange(n_views)}
