#!/usr/bin/env python3
"""Derive the compact FLAME *topology* asset used by the synthetic model.

Runs ONLY in the build container (needs /root/reference).  It reads DATA from the
reference checkout -- the OBJ topology/UVs, the landmark embedding, the UV-space region
masks, and the integer index literals inside vhap/model/flame.py (teeth faces, lip rings,
custom vertex masks; flame.py:206-504, 771-911) -- and writes one small npz:

    vhap_amd/assets/flame_topology.npz

Nothing here is executed at run time on the GPU box; the npz is the only thing that travels.
The licensed FLAME weights (flame2023.pkl, FLAME_masks.pkl) are absent from the reference
checkout, so vertex *regions* are approximated by sampling uv_masks.npz at the vertex UVs
(SURVEY.md section 7 step 0).
"""
import ast
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(__file__), "..", "vhap_amd", "assets", "flame_topology.npz")


def load_obj(path):
    v, vt, f, ft = [], [], [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("vt "):
                vt.append([float(x) for x in line.split()[1:3]])
            elif line.startswith("f "):
                a, b = [], []
                for tok in line.split()[1:4]:
                    p = tok.split("/")
                    a.append(int(p[0]) - 1)
                    b.append(int(p[1]) - 1)
                f.append(a)
                ft.append(b)
    return (np.asarray(v, np.float32), np.asarray(vt, np.float32),
            np.asarray(f, np.int32), np.asarray(ft, np.int32))


def _literal_list(node):
    """Return a python list if `node` is torch.tensor(<list literal>) else None."""
    if isinstance(node, ast.Call) and getattr(node.func, "attr", None) == "tensor" and node.args:
        try:
            val = ast.literal_eval(node.args[0])
        except Exception:
            return None
        if isinstance(val, list):
            return val
    return None


def extract_index_literals(flame_py):
    """Walk the AST of the reference flame.py and collect integer index tables."""
    tree = ast.parse(open(flame_py).read())
    named, masks = {}, {}
    for node in ast.walk(tree):
        # f_teeth_upper = torch.tensor([...])
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            val = _literal_list(node.value)
            if val is not None and node.targets[0].id.startswith("f_"):
                named[node.targets[0].id] = np.asarray(val, np.int32)
        # self.v.register_buffer("name", torch.tensor([...]))
        if isinstance(node, ast.Call) and getattr(node.func, "attr", None) == "register_buffer" and len(node.args) >= 2:
            if isinstance(node.args[0], ast.Constant) and isinstance(node.args[0].value, str):
                val = _literal_list(node.args[1])
                if val is not None:
                    masks[node.args[0].value] = np.asarray(val, np.int32)
    return named, masks


def main():
    v, vt, f, ft = load_obj(f"{REF}/asset/flame/head_template_mesh.obj")
    assert v.shape == (5023, 3) and vt.shape == (5118, 2) and f.shape == (9976, 3)

    emb = np.load(f"{REF}/asset/flame/landmark_embedding_with_eyes.npy", allow_pickle=True, encoding="latin1")[()]
    lmk_faces_idx = np.asarray(emb["full_lmk_faces_idx"], np.int64).reshape(-1)          # [70]
    lmk_bary = np.asarray(emb["full_lmk_bary_coords"], np.float64).reshape(-1, 3)        # [70,3]

    named, vmasks_lit = extract_index_literals(f"{REF}/vhap/model/flame.py")
    assert named["f_teeth_upper"].shape == (84, 3) and named["f_teeth_lower"].shape == (84, 3)

    # ---- vertex regions from the UV-space masks, sampled at every (vertex, vt) pair ----
    uvm = np.load(f"{REF}/asset/flame/uv_masks.npz")
    region_names = sorted(uvm.keys())
    T = uvm[region_names[0]].shape[0]
    col = np.clip((vt[:, 0] * T).astype(np.int64), 0, T - 1)
    row = np.clip(((1.0 - vt[:, 1]) * T).astype(np.int64), 0, T - 1)
    vt_region = np.stack([uvm[k][row, col] for k in region_names], axis=1)               # [VT,R] bool
    v_region = np.zeros((v.shape[0], len(region_names)), bool)
    # a vertex is in a region if ANY of its uv copies is
    np.logical_or.at(v_region, f.reshape(-1), vt_region[ft.reshape(-1)])

    out = dict(
        v_template=v, verts_uvs=vt, faces=f, faces_uv=ft,
        lmk_faces_idx=lmk_faces_idx, lmk_bary_coords=lmk_bary,
        f_teeth_upper=named["f_teeth_upper"], f_teeth_lower=named["f_teeth_lower"],
        region_names=np.asarray(region_names),
        v_region=np.packbits(v_region, axis=1),
        n_regions=np.int32(len(region_names)),
    )
    # exact (literal) custom vertex masks of the reference override the sampled ones
    for k, idx in vmasks_lit.items():
        out[f"vmask_{k}"] = idx
    # UV-space masks the regularisers need at texture resolution (packed bits)
    for k in ("sclerae", "teeth"):
        out[f"uvmask_{k}"] = np.packbits(uvm[k])
    out["uvmask_size"] = np.int32(T)

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes;",
          len(vmasks_lit), "literal vertex masks;", len(region_names), "uv regions")


if __name__ == "__main__":
    sys.exit(main())
