#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own Python code (runs only in the build container).

Imports, unmodified, from /root/reference:
  vhap/model/lbs.py                    -> lbs(), batch_rodrigues(), blend_shapes(), vertices2landmarks()
  vhap/util/render_nvdiffrast.py       -> get_SH_shading(), NVDiffRenderer.projection_from_intrinsics /
                                           mvp_from_camera_param (with `nvdiffrast.torch` stubbed: the
                                           module only needs the import to succeed for these functions)
  vhap/util/mesh.py                    -> normalize_image_points()
on small seeded inputs and stores inputs + outputs in tests/golden/reference_golden.npz.
These pin the oracle (oracle/torch_ref.py) and the product's host-side mirrors.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_golden.npz")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    torch.manual_seed(0)
    lbs = load(f"{REF}/vhap/model/lbs.py", "ref_lbs")
    # stub nvdiffrast + vhap.util.vector_ops import chain for render_nvdiffrast
    nv = types.ModuleType("nvdiffrast"); nvt = types.ModuleType("nvdiffrast.torch")
    sys.modules["nvdiffrast"], sys.modules["nvdiffrast.torch"] = nv, nvt
    sys.path.insert(0, REF)
    rn = load(f"{REF}/vhap/util/render_nvdiffrast.py", "ref_render")
    mesh = load(f"{REF}/vhap/util/mesh.py", "ref_mesh")

    B, V, NB, J = 3, 37, 11, 5
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g)
    v_template = r(V, 3) * 0.1
    shapedirs = r(V, 3, NB) * 0.01
    posedirs = r((J - 1) * 9, V * 3) * 0.01
    Jreg = torch.softmax(r(J, V), dim=1)
    parents = torch.tensor([-1, 0, 1, 1, 1])
    W = torch.softmax(r(V, J), dim=1)
    betas = r(B, NB)
    pose = r(B, J * 3) * 0.3
    pose[0] = 0                                                     # exercises the 1e-8 epsilon of batch_rodrigues
    v_shaped = v_template[None] + lbs.blend_shapes(betas, shapedirs)
    verts, Jt, A1 = lbs.lbs(pose, v_shaped.clone(), posedirs, Jreg, parents, W)
    faces = torch.randint(0, V, (20, 3), generator=g)
    lmk_idx = torch.randint(0, 20, (B, 6), generator=g)
    lmk_idx[:] = lmk_idx[0]
    bary = torch.softmax(r(1, 6, 3), dim=-1).repeat(B, 1, 1)
    lmks = lbs.vertices2landmarks(verts, faces, lmk_idx, bary)
    rod = lbs.batch_rodrigues(pose.view(-1, 3))

    # camera / SH
    rend = rn.NVDiffRenderer.__new__(rn.NVDiffRenderer)              # no __init__: it would create a CUDA context
    K4 = torch.tensor([[700.0, 710.0, 250.0, 260.0]])           # the reference only supports [1,4] here (tracker.py:155)
    K33 = torch.tensor([[[900.0, 0, 400.0], [0, 905.0, 270.0], [0, 0, 1]]])
    P4 = rend.projection_from_intrinsics(K4, (512, 512))
    P33 = rend.projection_from_intrinsics(K33, (550, 802))
    RT = torch.eye(4)[:3][None].repeat(2, 1, 1)
    RT[:, 2, 3] = -1
    RT[1, :3, :3] = lbs.batch_rodrigues(torch.tensor([[0.1, -0.2, 0.3]]))[0]
    mvp = rend.mvp_from_camera_param(RT.clone(), K4, (512, 512))
    pi = np.pi
    sh_const = torch.tensor([1 / np.sqrt(4 * pi)] + [((2 * pi) / 3) * (np.sqrt(3 / (4 * pi)))] * 3 +
                            [(pi / 4) * 3 * (np.sqrt(5 / (12 * pi)))] * 3 +
                            [(pi / 4) * (3 / 2) * (np.sqrt(5 / (12 * pi))), (pi / 4) * (1 / 2) * (np.sqrt(5 / (4 * pi)))],
                            dtype=torch.float32)
    normals = torch.nn.functional.normalize(r(2, 4, 5, 3), dim=-1)
    lights = r(1, 9, 3) * 0.2
    lights[0, 0] += np.sqrt(4 * pi)
    sh = rn.get_SH_shading(normals, lights, sh_const)
    u, v = mesh.normalize_image_points(torch.tensor([0.0, 100.0, 512.0]), torch.tensor([10.0, 256.0, 300.0]), (512, 400))

    out = dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=Jreg, parents=parents,
               lbs_weights=W, betas=betas, pose=pose, v_shaped=v_shaped, verts=verts, J_transformed=Jt, A1=A1,
               faces=faces, lmk_idx=lmk_idx, lmk_bary=bary, lmks=lmks, rodrigues=rod,
               K4=K4, K33=K33, P4=P4, P33=P33, RT=RT, mvp=mvp, sh_const=sh_const, normals=normals, lights=lights, sh=sh,
               norm_u=u, norm_v=v)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **{k: t.numpy() for k, t in out.items()})
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
