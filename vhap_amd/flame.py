"""FLAME head model -- host-side mirror of vhap/model/flame.py (FlameHead :64, FlameTexPainted
:649, FlameUvMask :1057, FlameMask :719) for the MI355X path.

`FlameHead.forward` keeps the reference signature and return convention (flame.py:571-646).  The
model tensors come from a dict (synthetic model of vhap_amd.synthetic, or a user-supplied FLAME
pickle via `FlameHead.from_flame_pickle`) instead of the licensed pkl hard-wired at flame.py:38.
Everything topological lives in vhap_amd.topology.FlameTopology.
"""
import numpy as np
import torch
import torch.nn as nn

from . import lbs as L
from . import fused as FU
from .topology import FlameTopology


class FlameMask:
    """Region -> index lookups (flame.py:999-1054), backed by FlameTopology, cached on device."""

    def __init__(self, topo, device="cpu"):
        self.topo = topo
        self.device = device
        self._cache = {}
        self.fid2cid = torch.from_numpy(topo.fid2cid[1:].astype(np.int64))   # unpadded, like FlameMask.fid2cid

    def _get(self, kind, regions):
        if isinstance(regions, str):
            regions = [regions]
        key = (kind, tuple(regions), str(self.device))
        if key not in self._cache:
            fn = self.topo.get_vid_by_region if kind == "v" else self.topo.get_fid_by_region
            self._cache[key] = torch.from_numpy(fn(list(regions))).long().to(self.device)
        return self._cache[key]

    def get_vid_by_region(self, regions, keep_order=False):
        return self._get("v", regions)

    def get_fid_by_region(self, regions):
        return self._get("f", regions)

    def to(self, device):
        self.device = device
        return self


class FlameHead(nn.Module):
    def __init__(self, model, topo: FlameTopology, shape_params=300, expr_params=100):
        super().__init__()
        self.n_shape_params, self.n_expr_params = shape_params, expr_params
        self.topo = topo
        t = lambda k, dt=torch.float32: torch.as_tensor(np.asarray(model[k])).to(dt)
        self.register_buffer("v_template", t("v_template"))
        self.register_buffer("shapedirs", t("shapedirs"))                       # [V,3,NB]
        self.register_buffer("posedirs", t("posedirs"))                         # [36,3V]
        self.register_buffer("J_regressor", t("J_regressor"))
        self.register_buffer("parents", t("parents", torch.long))
        self._parents = [int(x) for x in np.asarray(model["parents"])]           # host copy: indexing with it never syncs
        self.register_buffer("lbs_weights", t("lbs_weights"))
        self.register_buffer("faces", t("faces", torch.long), persistent=False)
        self.register_buffer("textures_idx", t("faces_uv", torch.long), persistent=False)
        self.register_buffer("verts_uvs", t("verts_uvs"), persistent=False)
        self.register_buffer("full_lmk_faces_idx", t("lmk_faces_idx", torch.long)[None])
        self.register_buffer("full_lmk_bary_coords", t("lmk_bary_coords")[None])
        # sparse uniform Laplacian (the reference keeps it dense: flame.py:196-201)
        self.register_buffer("lap_ptr", torch.from_numpy(topo.lap_ptr.astype(np.int64)), persistent=False)
        self.register_buffer("lap_col", torch.from_numpy(topo.lap_col.astype(np.int64)), persistent=False)
        self.register_buffer("lap_val", torch.from_numpy(topo.lap_val), persistent=False)
        rows = np.repeat(np.arange(topo.num_verts), np.diff(topo.lap_ptr))
        self.register_buffer("lap_row", torch.from_numpy(rows.astype(np.int64)), persistent=False)
        self.mask = FlameMask(topo)
        self.fused = True          # MFMA blend+skin kernels on a HIP device (False: host-side torch ops only)
        self._fb = None

    @classmethod
    def from_flame_pickle(cls, flame_model_path, flame_masks_path=None, shape_params=300, expr_params=100, add_teeth=True, **kw):
        """The head from the licensed FLAME files, like the reference's FlameHead(shape_params, expr_params, flame_model_path, ...,
        add_teeth=True) (flame.py:70-204; vhap_amd.flame_assets).  `.model` / `.topo` are what GlobalTracker takes."""
        from .flame_assets import load_flame_model
        model, topo = load_flame_model(flame_model_path, flame_masks_path, shape_params, expr_params, add_teeth, **kw)
        head = cls(model, topo, shape_params, expr_params)
        head.model = model
        return head

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.mask.to(self.v_template.device)
        return out

    @property
    def laplacian_matrix(self):
        """Sparse COO form of flame.py:196 `laplacian_matrix` ([V,V])."""
        V = self.v_template.shape[0]
        return torch.sparse_coo_tensor(torch.stack([self.lap_row, self.lap_col]), self.lap_val, (V, V))

    def laplacian_apply(self, x):
        """L @ x for x [B,V,3] without the dense [V,V] matrix (tracker.py:682-690 uses 2 dense bmm)."""
        contrib = self.lap_val[None, :, None] * x[:, self.lap_col]
        return torch.zeros_like(x).index_add_(1, self.lap_row, contrib)

    def forward(self, shape, expr, rotation, neck, jaw, eyes, translation, zero_centered_at_root_node=False,
                return_landmarks=True, return_verts_cano=False, static_offset=None, dynamic_offset=None):
        betas = torch.cat([shape, expr], dim=1)
        full_pose = torch.cat([rotation, neck, jaw, eyes], dim=1)
        translated = False
        if self.fused and betas.is_cuda and dynamic_offset is None:
            translated = not zero_centered_at_root_node          # the kernel adds the translation itself
            vertices, v_shaped, J = self._forward_fused(betas, full_pose, static_offset, translation if translated else None)
        else:
            v_shaped = self.v_template[None] + L.blend_shapes(betas, self.shapedirs)
            if static_offset is not None:
                v_shaped = v_shaped + static_offset
            if dynamic_offset is not None:
                v_shaped = v_shaped + dynamic_offset
            vertices, J, _ = L.lbs(full_pose, v_shaped, self.posedirs, self.J_regressor, self._parents, self.lbs_weights)
        if zero_centered_at_root_node:
            vertices = vertices - J[:, [0]]
        if not translated:
            vertices = vertices + translation[:, None, :]
        ret = [vertices]
        if return_verts_cano:
            ret.append(v_shaped)
        if return_landmarks:
            ret.append(L.vertices2landmarks(vertices, self.faces, self.full_lmk_faces_idx, self.full_lmk_bary_coords))
        return ret if len(ret) > 1 else ret[0]


    def _forward_fused(self, betas, full_pose, static_offset, translation=None):
        """Blendshapes, pose correctives and skinning on the matrix cores (vhap_amd/csrc/flame.hip); the per-frame
        5-joint algebra (Rodrigues, joint regression through the pre-contracted J_regressor @ shapedirs, chain)
        stays here as a few tiny host-side ops."""
        if self._fb is None or self._fb.basis.device != betas.device:
            self._fb = FU.FlameBasis(self.shapedirs, self.posedirs, self.J_regressor, self.v_template, self.lbs_weights)
        fb = self._fb
        B = betas.shape[0]
        R = L.batch_rodrigues(full_pose.reshape(-1, 3)).view(B, -1, 3, 3)
        eye = torch.eye(3, dtype=betas.dtype, device=betas.device)
        pose_feature = (R[:, 1:] - eye).reshape(B, -1)
        J = fb.JT[None] + (betas @ fb.JS.t()).view(B, -1, 3)
        if static_offset is not None:
            J = J + (self.J_regressor[:, :, None] * static_offset.reshape(1, -1, 3)).sum(dim=1)[None]
        J_posed, A = L.batch_rigid_transform(R, J, self._parents)
        coef = torch.cat([betas, pose_feature], dim=1)
        Bp = (B + 15) // 16 * 16
        coef = torch.nn.functional.pad(coef, [0, fb.Kp - coef.shape[1], 0, Bp - B])
        if translation is None:
            translation = torch.zeros(B, 3, dtype=betas.dtype, device=betas.device)
        vertices, v_shaped = FU.flame_skin(fb, coef, A[:, :, :3, :].reshape(B, -1, 12), translation, static_offset)
        return vertices, v_shaped, J_posed


class FlameTexPainted(nn.Module):
    """flame.py:649-662 -- a fixed base texture [1,3,T,T]; here supplied as an array (the painted PNG of
    the reference, or the procedural stand-in of vhap_amd.synthetic.make_texture)."""

    def __init__(self, tex_chw):
        super().__init__()
        self.register_buffer("tex_painted", torch.as_tensor(np.asarray(tex_chw), dtype=torch.float32)[None])
        self.tex_size = self.tex_painted.shape[-1]

    def forward(self):
        return self.tex_painted


class FlameTexPCA(nn.Module):
    """flame.py:665-688 -- the FLAME PCA texture space (`tex_painted = False`): texture = mean + basis . code on a 512 x 512 x 3 grid in
    0..255, B G R; forward(texcode [B, n]) -> [B, 3, T, T]: nearest resize to the working resolution, channels to R G B, / 255, clamp.
    `tex_space`: a path to FLAME_texture.npz (`mean`, `tex_dir`) or a mapping with those two arrays (tests: a synthetic space of the same
    layout -- the licensed file is not in this container)."""

    def __init__(self, tex_params, tex_size=512, tex_space=None):
        super().__init__()
        self.tex_size = int(tex_size)
        space = np.load(tex_space) if isinstance(tex_space, (str, bytes)) or hasattr(tex_space, "__fspath__") else tex_space
        mean = np.asarray(space["mean"], np.float32).reshape(1, -1)
        n_all = int(np.asarray(space["tex_dir"]).shape[-1])
        basis = np.asarray(space["tex_dir"], np.float32).reshape(-1, n_all)[:, :tex_params]
        self.src_size = int(round((mean.shape[1] // 3) ** 0.5))
        assert self.src_size * self.src_size * 3 == mean.shape[1] == basis.shape[0]
        self.register_buffer("texture_mean", torch.from_numpy(mean)[None, ...])                           # [1, 1, S*S*3]
        self.register_buffer("texture_basis", torch.from_numpy(np.ascontiguousarray(basis))[None, ...])   # [1, S*S*3, n]

    def forward(self, texcode):
        S = self.src_size
        texture = self.texture_mean + (self.texture_basis * texcode[:, None, :]).sum(-1)
        texture = texture.reshape(texcode.shape[0], S, S, 3).permute(0, 3, 1, 2)
        texture = torch.nn.functional.interpolate(texture, [self.tex_size, self.tex_size])
        texture = texture[:, [2, 1, 0], :, :]
        texture = texture / 255.0
        return texture.clamp(0, 1)


class FlameUvMask(nn.Module):
    """flame.py:1057-1070."""

    def __init__(self, topo):
        super().__init__()
        for k, m in topo.uvmasks.items():
            self.register_buffer(k, torch.from_numpy(m))

    def get_uvmask_by_region(self, regions):
        if isinstance(regions, str):
            regions = [regions]
        m = getattr(self, regions[0])
        for r in regions[1:]:
            m = m | getattr(self, r)
        return m
