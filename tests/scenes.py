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
