"""Seeded test scenes shared by CPU and GPU tests (oracle side: numpy / torch-CPU only)."""
import numpy as np
import torch

from oracle import torch_ref as R
from vhap_amd.synthetic import make_scene_params, monocular_camera


def head_scene(model, B, H, W, seed=0, dtype=torch.float32, translation_z=0.45):
    """GT FLAME params -> world verts, clip-space verts, landmarks (all torch CPU)."""
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dtype)
    sp = make_scene_params(B, seed, (H, W), translation_z=translation_z)
    g = lambda k: torch.from_numpy(sp[k]).to(dtype)
    verts, v_shaped, lmks = R.flame_forward(tm, g("shape")[None].expand(B, -1), g("expr"), g("rotation"), g("neck_pose"),
                                            g("jaw_pose"), g("eyes_pose"), g("translation"))
    K, RT = monocular_camera(B, (H, W))
    K, RT = torch.from_numpy(K).to(dtype), torch.from_numpy(RT).to(dtype)
    clip = R.camera_to_clip(R.world_to_camera(verts, RT), K, (H, W))
    return dict(tm=tm, sp=sp, verts=verts, v_shaped=v_shaped, lmks=lmks, K=K, RT=RT, clip=clip)


def near_plane_scene(B, seed=0, focal=1.5, near=0.1, far=10.0):
    """Clip-space geometry that CROSSES the near plane close to the view axis (so the cut is on screen): perturbed grid surfaces of three
    resolutions arranged around the axis like the walls of a corridor the camera stands in, each running from 1 m in front of the camera to
    0.3 m BEHIND it (vertices with w < 0), drawn with both windings (one of the two survives the back-face cull).
    Returns pos [B,V,4] float32, tri [F,3] int32, cam [B,V,3] float64 (camera space)."""
    rng = np.random.default_rng(seed)
    cams, tris, voff = [], [], 0
    for k, n in enumerate((5, 24, 50, 24, 5, 50)):      # 24 808 triangles (<= 32 768: the one-launch binning path)
        theta = 2 * np.pi * k / 6 + 0.2
        u, z = np.meshgrid(np.linspace(-0.25, 0.25, n + 1), np.linspace(-1.0, 0.3, n + 1))
        cell = 0.5 / n
        u = u + rng.normal(0, 0.15 * cell, u.shape)
        z = z + rng.normal(0, 0.15 * cell, z.shape)
        h = -(0.012 + 0.006 * k) + 0.05 * (z + near) + rng.normal(0, 0.05 * cell, u.shape)
        cams.append(np.stack([u * np.cos(theta) - h * np.sin(theta), u * np.sin(theta) + h * np.cos(theta), z], -1).reshape(-1, 3))
        idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1) + voff
        a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
        t = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)
        tris += [t, t[:, [0, 2, 1]]]
        voff += (n + 1) * (n + 1)
    cam = np.concatenate(cams, 0)
    cam = np.stack([cam + np.array([0.004 * b, -0.003 * b, 0.013 * b]) for b in range(B)], 0)      # every frame cuts elsewhere
    A, Bz = -(far + near) / (far - near), -2 * far * near / (far - near)
    pos = np.stack([2 * focal * cam[..., 0], 2 * focal * cam[..., 1], A * cam[..., 2] + Bz, -cam[..., 2]], -1).astype(np.float32)
    tri = np.concatenate(tris, 0).astype(np.int32)
    tri = tri[rng.permutation(len(tri))]              # no spatial coherence in triangle order either (stresses the pair-list regions)
    return pos, tri, cam
