"""Real FLAME assets -> the model dict / FlameTopology pair the rest of vhap_amd consumes (host side, numpy only).

What FlameHead.__init__ does with the licensed files (vhap/model/flame.py:70-204) -- `flame2023.pkl` / `generic_model.pkl`
(v_template, shapedirs, posedirs, J_regressor, kintree_table, weights) and `FLAME_masks.pkl` (vertex part masks, flame.py:754-767) --
plus FlameHead.add_teeth (flame.py:206-325: 120 vertices derived from the lip rings, shape directions copied from the lips, no pose
correctives, rigid to neck / jaw).  Topology, UVs and the landmark embedding come from the un-licensed assets already baked into
vhap_amd/assets/flame_topology.npz (head_template_mesh.obj, landmark_embedding_with_eyes.npy).

    model, topo = load_flame_model("flame2023.pkl", "FLAME_masks.pkl")
    head = FlameHead(model, topo)            # or FlameHead.from_flame_pickle(...)
    tracker = GlobalTracker(cfg, model, topo, base_texture, dataset)

The synthetic model (vhap_amd.synthetic.make_flame_model) goes through the same append_teeth().
"""
import io
import pickle

import numpy as np

from .topology import TEX_CLUSTERS, FlameTopology, load_flame_masks


class _ChumpyStub:
    """Stands in for chumpy.ch.Ch while unpickling generic_model.pkl (its arrays are chumpy objects; chumpy is not a dependency here):
    keeps the instance state; the ndarray sits under 'x'."""

    def __init__(self, *a, **k):
        self.__dict__["_state"] = {}

    def __setstate__(self, state):
        self.__dict__["_state"] = state if isinstance(state, dict) else {"x": state}

    def __array__(self, dtype=None, copy=None):
        x = np.asarray(self.__dict__["_state"]["x"])
        return x.astype(dtype) if dtype is not None else x


class _FlameUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return _ChumpyStub
        return super().find_class(module, name)


def load_flame_pickle(path):
    """-> dict of plain numpy arrays (float32 / int64) under the pickle's own keys.  Python-2 pickle (latin1); scipy.sparse
    J_regressor densified like the reference's to_np (flame.py:52-55); chumpy arrays read without chumpy."""
    with open(path, "rb") as f:
        raw = _FlameUnpickler(io.BytesIO(f.read()), encoding="latin1").load()
    out = {}
    for k, v in raw.items():
        if hasattr(v, "todense"):
            v = np.asarray(v.todense())
        if isinstance(v, (str, bytes)):
            continue
        try:
            a = np.asarray(v)
        except Exception:
            continue
        if a.dtype == object:
            continue
        out[str(k)] = a
    return out


def teeth_vertices(v, topo):
    """The geometric construction of FlameHead.add_teeth (flame.py:208-260): 8 rows of 15 vertices hung below / above the midline of
    the two outer lip rings."""
    up = v[topo.v_regions["lip_outside_ring_upper"]]
    lo = v[topo.v_regions["lip_outside_ring_lower"]]
    mean_dist = np.linalg.norm(up - lo, axis=-1).mean()
    mid = (up + lo) / 2
    mid[:, 1] = mid[:, 1].mean()
    mid[:, 2] -= mean_dist * 1.5
    ey = np.array([[0, mean_dist, 0]], np.float32)
    ez = np.array([[0, 0, mean_dist]], np.float32)
    upper_edge = mid + ey * 0.1
    upper_root = upper_edge + ey * 2
    lower_edge = mid - ey * 0.1 - ez * 0.4
    lower_root = lower_edge - ey * 2
    th = np.array([[0, 0, mean_dist * 1.0]], np.float32)
    return np.concatenate([upper_root, lower_root, upper_edge, lower_edge,
                           upper_root - th, upper_edge - th, lower_root - th, lower_edge - th], 0).astype(np.float32)


def append_teeth(arrays, topo, n_shape):
    """flame.py:262-325 on numpy arrays: v_template / shapedirs / posedirs / J_regressor / lbs_weights grown by the 120 teeth vertices
    (`topo` must have been built with add_teeth=True).  shapedirs [V,3,NB]: the shape part (first n_shape columns) is the mean of the
    two lip rings' directions, the expression part zero; posedirs [36,3V] and J_regressor zero; weights one-hot: upper teeth -> joint 1
    (neck), lower -> joint 2 (jaw)."""
    v, shapedirs, posedirs = arrays["v_template"], arrays["shapedirs"], arrays["posedirs"]
    nv0, NB = v.shape[0], shapedirs.shape[2]
    assert topo.has_teeth and nv0 == topo.num_verts_orig
    vt = teeth_vertices(v.copy(), topo)
    up, lo = topo.v_regions["lip_outside_ring_upper"], topo.v_regions["lip_outside_ring_lower"]
    sd_mean = (shapedirs[up, :, :n_shape] + shapedirs[lo, :, :n_shape]) / 2        # [15,3,n_shape]
    sd_t = np.zeros((120, 3, NB), np.float32)
    for r in range(8):
        sd_t[15 * r:15 * (r + 1), :, :n_shape] = sd_mean
    nj = posedirs.shape[0] // 9
    pd = posedirs.reshape(nj, 9, nv0, 3)
    pd = np.concatenate([pd, np.zeros((nj, 9, 120, 3), np.float32)], 2)
    lw_t = np.zeros((120, arrays["lbs_weights"].shape[1]), np.float32)
    lw_t[topo.v_regions["teeth_upper"] - nv0, 1] = 1
    lw_t[topo.v_regions["teeth_lower"] - nv0, 2] = 1
    out = dict(arrays)
    out.update(v_template=np.concatenate([v, vt], 0).astype(np.float32),
               shapedirs=np.concatenate([shapedirs, sd_t], 0),
               posedirs=pd.reshape(nj * 9, (nv0 + 120) * 3),
               J_regressor=np.concatenate([arrays["J_regressor"], np.zeros((arrays["J_regressor"].shape[0], 120), np.float32)], 1),
               lbs_weights=np.concatenate([arrays["lbs_weights"], lw_t], 0))
    return out


def model_from_flame_dict(fl, topo, shape_params=300, expr_params=100):
    """flame.py:97-126: the pickle's arrays -> the buffers of FlameHead (before add_teeth)."""
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))
    sd = f32(fl["shapedirs"])
    if sd.shape[2] < 300 + expr_params or shape_params > 300:
        raise ValueError(f"shapedirs {sd.shape}: expected the 300 shape + >= {expr_params} expression directions of FLAME")
    shapedirs = np.concatenate([sd[:, :, :shape_params], sd[:, :, 300:300 + expr_params]], 2)
    pd = f32(fl["posedirs"])
    posedirs = np.ascontiguousarray(pd.reshape(-1, pd.shape[-1]).T)                 # [V,3,P] -> [P, 3V]
    parents = np.asarray(fl["kintree_table"])[0].astype(np.int64)
    parents[0] = -1
    if f32(fl["v_template"]).shape[0] != topo.num_verts_orig:
        raise ValueError("the FLAME pickle and the template mesh disagree on the vertex count")
    return dict(v_template=f32(fl["v_template"]), shapedirs=shapedirs, posedirs=posedirs, J_regressor=f32(fl["J_regressor"]),
                parents=parents, lbs_weights=f32(fl["weights"]))


def load_flame_model(flame_model_path, flame_masks_path=None, shape_params=300, expr_params=100, add_teeth=True,
                     tex_clusters=TEX_CLUSTERS):
    """-> (model dict, FlameTopology), the pair vhap_amd.synthetic.make_flame_model returns for the synthetic model.
    Without `flame_masks_path` the vertex regions are the uv-sampled approximation of the topology asset."""
    part_masks = load_flame_masks(flame_masks_path) if flame_masks_path is not None else None
    topo = FlameTopology(add_teeth=add_teeth, tex_clusters=tex_clusters, part_masks=part_masks)
    model = model_from_flame_dict(load_flame_pickle(flame_model_path), topo, shape_params, expr_params)
    if topo.has_teeth:
        model = append_teeth(model, topo, shape_params)
    model.update(faces=topo.faces.astype(np.int64), faces_uv=topo.faces_uv.astype(np.int64), verts_uvs=topo.verts_uvs.copy(),
                 lmk_faces_idx=topo.lmk_faces_idx.copy(), lmk_bary_coords=topo.lmk_bary_coords.copy())
    return model, topo
