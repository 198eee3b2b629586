#!/usr/bin/env python3
"""Golden vectors for `w.blur_iter > 0` (vhap/config/base.py:181), produced by the REFERENCE's own code (build container only):
FlameTracker.scale_vertex_weights_by_region (vhap/model/tracker.py:607-614) and, through it, the relaxed Laplacian / L1 offset
regularisers of compute_regularization_energy (:552-590), called on a FlameTracker created without __init__ exactly as
tools/make_golden_energy.py does (same stubs, same synthetic FLAME-topology model).  The uniform Laplacian comes from the oracle
restatement (pytorch3d is absent) and is stored alongside.  Output: tests/golden/blur_golden.npz."""
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "blur_golden.npz")


def main():
    import make_golden_energy as G
    base, rn, T, FL = G.load_reference()
    from oracle import energy_ref, torch_ref as R
    from vhap_amd.synthetic import make_flame_model
    import dataclasses
    import typing
    from pathlib import Path
    dt = torch.float64

    def default_instance(cls):
        hints = typing.get_type_hints(cls)
        kw = {}
        for f in dataclasses.fields(cls):
            if f.default is not dataclasses.MISSING or f.default_factory is not dataclasses.MISSING:
                continue
            t = hints[f.name]
            kw[f.name] = default_instance(t) if dataclasses.is_dataclass(t) else (Path(".") if t is Path else "x")
        return cls(**kw)
    rcfg = default_instance(base.BaseTrackingConfig)
    model, topo = make_flame_model(0)
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    V = tm["v_template"].shape[0]
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g, dtype=dt) * sc
    N = 3
    P = dict(shape=rnd(300, sc=0.3), expr=rnd(N, 100, sc=0.3), rotation=rnd(N, 3, sc=0.1), neck_pose=rnd(N, 3, sc=0.05),
             jaw_pose=rnd(N, 3, sc=0.1), eyes_pose=rnd(N, 6, sc=0.1), translation=rnd(N, 3, sc=0.02), static_offset=rnd(1, V, 3, sc=1e-3))
    ts = np.array([2])          # B = 1: the reference blurs with M[None].bmm(weights), which only accepts a batch of one (tracker.py:611-613)
    B = len(ts)
    verts, v_cano, lmks = R.flame_forward(tm, P["shape"][None].expand(B, -1), P["expr"][ts], P["rotation"][ts], P["neck_pose"][ts],
                                          P["jaw_pose"][ts], P["eyes_pose"][ts], P["translation"][ts], static_offset=P["static_offset"])
    Lap = energy_ref._laplacian(V, topo).to(dt)
    tr = object.__new__(T.FlameTracker)
    tr.cfg, tr.device = rcfg, "cpu"
    tr.opt_dict = defaultdict(bool, {"static_offset": True})
    tr.n_timesteps = N
    tr.static_offset, tr.dynamic_offset = P["static_offset"], None
    vid = lambda regions: torch.from_numpy(topo.get_vid_by_region(list(regions))).long()
    tr.flame = types.SimpleNamespace(mask=types.SimpleNamespace(get_vid_by_region=vid), laplacian_matrix=Lap,
                                     laplacian_matrix_negate_diag=Lap - 2 * torch.diag(torch.diag(Lap)))
    save = {"ts": ts, "v_cano": v_cano.numpy(), "static_offset": P["static_offset"].numpy()}
    w = rcfg.w
    for it in (1, 3):
        w.blur_iter = it
        save[f"w_lap_{it}"] = tr.scale_vertex_weights_by_region(torch.ones(1, V, 1, dtype=dt), w.reg_offset_lap_relax_coef,
                                                               w.reg_offset_lap_relax_for).numpy()
        save[f"w_off_{it}"] = tr.scale_vertex_weights_by_region(torch.ones(1, V, 1, dtype=dt), w.reg_offset_relax_coef,
                                                               w.reg_offset_relax_for).numpy()
        # the offset part of compute_regularization_energy (tracker.py:552-590), the reference's lines themselves
        offset = tr.static_offset
        v0 = (v_cano - offset).detach()
        lap = tr.compute_laplacian_smoothing_loss(v0, v0 + offset)
        lap = lap * tr.scale_vertex_weights_by_region(torch.ones_like(verts[:, :, :1]), w.reg_offset_lap_relax_coef, w.reg_offset_lap_relax_for)
        save[f"reg_offset_lap_{it}"] = (w.reg_offset_lap * lap.mean()).numpy()
        ro = tr.scale_vertex_weights_by_region(torch.ones_like(verts[:, :, :1]), w.reg_offset_relax_coef, w.reg_offset_relax_for) * offset.abs()
        save[f"reg_offset_{it}"] = (w.reg_offset * ro.mean()).numpy()
    np.savez_compressed(OUT, **save)
    print("wrote", OUT, {k: np.asarray(v).shape for k, v in save.items()})


if __name__ == "__main__":
    main()
