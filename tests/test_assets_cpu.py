"""Real-asset constructors: FLAME_masks.pkl -> FlameTopology and flame2023.pkl -> FlameHead, without the
licensed files: a synthetic masks pickle is pinned against the reference's UNMODIFIED FlameMask (tests/golden/flame_masks_golden.npz,
tools/make_golden_masks.py), a synthetic model pickle in FLAME's layout (chumpy-typed arrays, scipy.sparse J_regressor, uint32
kintree_table, Python-2 protocol) round-trips to the arrays of vhap_amd.synthetic.make_flame_model."""
import os
import pickle
import sys
import types

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flame_masks_golden.npz")


def _synthetic_part_masks():
    from vhap_amd.topology import FLAME_PART_NAMES, FlameTopology
    t = FlameTopology(add_teeth=False)
    return {k: np.asarray(t.v_regions[k], np.int64) for k in FLAME_PART_NAMES}


def _write_masks(path):
    with open(path, "wb") as f:
        pickle.dump(_synthetic_part_masks(), f, protocol=2)


def test_topology_from_flame_masks_matches_reference_flame_mask(tmp_path):
    from vhap_amd.topology import FlameTopology
    p = str(tmp_path / "FLAME_masks.pkl")
    _write_masks(p)
    topo = FlameTopology.from_flame_masks(p)
    g = np.load(GOLD)
    for k in g["v_names"]:
        k = str(k)
        assert np.array_equal(np.asarray(topo.v_regions[k]), g[f"v/{k}"]), f"vertex region {k}"
    assert sorted(topo.f_regions) == [str(x) for x in g["f_names"]]
    for k in g["f_names"]:
        assert np.array_equal(topo.f_regions[str(k)], g[f"f/{k}"]), f"face region {k}"
    F = topo.num_faces      # FlameMask.fid2cid: [F + 1] indexed by face id (flame.py:965-984); the renderer pads a leading 0 for "background"
    assert g["fid2cid"].shape[0] == F + 1 and topo.fid2cid[0] == 0 and np.array_equal(topo.fid2cid[1:], g["fid2cid"][:F])    # (render_nvdiffrast.py:78)
    for key in g.files:
        kind, _, regions = key.partition("/")
        if kind == "vid":
            assert np.array_equal(topo.get_vid_by_region(regions.split("+")), g[key]), key
        elif kind == "fid":
            assert np.array_equal(topo.get_fid_by_region(regions.split("+")), g[key]), key
    # ... and without teeth: the mask as FlameHead.__init__ first builds it (flame.py:168-175)
    t0 = FlameTopology(add_teeth=False, part_masks=_synthetic_part_masks())
    assert sorted(t0.f_regions) == [str(x) for x in g["f_names_noteeth"]]
    for k in g["f_names_noteeth"]:
        assert np.array_equal(t0.f_regions[str(k)], g[f"f0/{k}"]), f"face region {k} (no teeth)"
    assert np.array_equal(t0.fid2cid[1:], g["fid2cid_noteeth"][:t0.num_faces])


def _write_flame_pickle(path, model, nv0):
    """The synthetic model's head part in the layout of generic_model.pkl / flame2023.pkl."""
    import scipy.sparse as sp
    mod = types.ModuleType("chumpy")
    sub = types.ModuleType("chumpy.ch")

    class Ch:                                                     # pickled as chumpy.ch.Ch with its state dict, like the real thing
        def __init__(self, x):
            self.x = x

        def __getstate__(self):
            return {"x": self.x, "_dirty_vars": set(), "_itr": None}
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    sub.Ch = Ch
    mod.ch = sub
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
    try:
        V = nv0
        d = {"v_template": Ch(model["v_template"][:V].astype(np.float64)),
             "shapedirs": Ch(model["shapedirs"][:V].astype(np.float64)),
             "posedirs": model["posedirs"].reshape(36, -1, 3)[:, :V].reshape(36, V * 3).T.reshape(V, 3, 36).astype(np.float64),
             "J_regressor": sp.csc_matrix(model["J_regressor"][:, :V].astype(np.float64)),
             "kintree_table": np.array([[4294967295, 0, 1, 1, 1], [0, 1, 2, 3, 4]], np.uint32),
             "weights": model["lbs_weights"][:V].astype(np.float64),
             "f": model["faces"][:9976].astype(np.uint32), "bs_style": "lbs", "bs_type": "lrotmin"}
        with open(path, "wb") as f:
            pickle.dump(d, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]


def test_flame_pickle_round_trip(tmp_path, flame_model):
    from vhap_amd.flame import FlameHead
    from vhap_amd.flame_assets import load_flame_model, load_flame_pickle
    model, topo = flame_model
    pm, pk = str(tmp_path / "flame2023.pkl"), str(tmp_path / "FLAME_masks.pkl")
    _write_flame_pickle(pm, model, topo.num_verts_orig)
    _write_masks(pk)
    assert "chumpy" not in sys.modules
    raw = load_flame_pickle(pm)
    assert raw["shapedirs"].shape == (5023, 3, 400) and raw["J_regressor"].shape == (5, 5023) and "bs_style" not in raw
    got, topo2 = load_flame_model(pm, pk)
    assert topo2.has_teeth and got["v_template"].shape == (5143, 3)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "parents", "faces", "faces_uv", "verts_uvs", "lmk_faces_idx",
              "lmk_bary_coords"):
        assert np.array_equal(np.asarray(got[k]), np.asarray(model[k])), k
    # fewer directions than the file holds (the reference's shape_params / expr_params arguments, flame.py:104-109)
    small, _ = load_flame_model(pm, pk, shape_params=100, expr_params=50)
    assert small["shapedirs"].shape == (5143, 3, 150)
    assert np.array_equal(small["shapedirs"][:5023, :, 100:], model["shapedirs"][:5023, :, 300:350])
    head = FlameHead.from_flame_pickle(pm, pk)
    ref = FlameHead(model, topo)
    g = torch.Generator().manual_seed(0)
    args = [torch.randn(2, n, generator=g) * s for n, s in ((300, 0.3), (100, 0.3), (3, 0.1), (3, 0.05), (3, 0.05), (6, 0.05), (3, 0.01))]
    v1, l1 = head(*args)
    v2, l2 = ref(*args)
    assert torch.equal(v1, v2) and torch.equal(l1, l2)
    # the tracker takes (head.model, head.topo)
    assert head.model is not None and head.topo.fid2cid.shape == (10144 + 1,)
