#!/usr/bin/env python3
"""Golden vectors for the `use_dynamic_offset` branch (vhap/config/base.py:69): the reference's UNMODIFIED
FlameTracker.compute_regularization_energy (tracker.py:480-605) with a per-timestep dynamic offset -- offset = static_offset +
dynamic_offset[timesteps] in the Laplacian / L1 / rigidity regularisers (:552-592) and the temporal smoothness term
reg_offset_dynamic (:594-600) -- and its backward, on a tracker object created without __init__ and filled from the synthetic model.
(FlameHead.forward with a dynamic offset is pinned by tools/make_golden_energy.py section 6.)

Two cases, because the reference itself cannot do more: with BOTH offsets it builds `offset = 0; offset += static_offset [1,V,3];
offset += dynamic_offset[timesteps] [B,V,3]` (tracker.py:555-559) -- an in-place add that raises for B > 1 -- so static + dynamic is
recorded for a one-frame batch and dynamic-only (use_static_offset = False) for a three-frame batch.

Runs only in the build container (needs /root/reference).  Output: tests/golden/dynoffset_golden.npz.
"""
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "dynoffset_golden.npz")


def main():
    import dataclasses
    import typing
    from pathlib import Path
    from make_golden_energy import load_reference
    base, rn, T, FL = load_reference()
    from oracle import energy_ref, torch_ref as R
    from vhap_amd.synthetic import make_flame_model
    dt = torch.float64
    torch.Tensor.cuda = lambda self, *a, **k: self

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
    rcfg.model.use_dynamic_offset = True
    model, topo = make_flame_model(0)
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    N = 4
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g, dtype=dt) * sc
    V = tm["v_template"].shape[0]
    P = dict(shape=rnd(300, sc=0.3), expr=rnd(N, 100, sc=0.3), rotation=rnd(N, 3, sc=0.1), neck_pose=rnd(N, 3, sc=0.05),
             jaw_pose=rnd(N, 3, sc=0.1), eyes_pose=rnd(N, 6, sc=0.1), translation=rnd(N, 3, sc=0.02),
             static_offset=rnd(1, V, 3, sc=1e-3), dynamic_offset=rnd(N, V, 3, sc=5e-4))
    # (the test re-draws the inputs from the same seeded generator; their sums are stored as a check)
    pick = np.sort(np.random.default_rng(0).choice(V, 500, replace=False))
    out = {"pick": pick, **{f"in_sum/{k}": np.array(float(v.sum())) for k, v in P.items()}}
    Lap = energy_ref._laplacian(V, topo).to(dt)
    vid = lambda regions: torch.from_numpy(topo.get_vid_by_region(list(regions))).long()
    for case, ts, with_static in (("both_B1", np.array([2]), True), ("dynamic_only_B3", np.array([1, 3, 0]), False)):
        B = len(ts)
        out[f"{case}/timesteps"] = ts
        for stage in ("rgb_sequential_tracking", "rgb_global_tracking"):
            Pl = {k: v.clone().requires_grad_() for k, v in P.items()}
            if not with_static:
                Pl["static_offset"] = None
            verts, v_cano, lmks = R.flame_forward(tm, Pl["shape"][None].expand(B, -1), Pl["expr"][ts], Pl["rotation"][ts], Pl["neck_pose"][ts],
                                                  Pl["jaw_pose"][ts], Pl["eyes_pose"][ts], Pl["translation"][ts], static_offset=Pl["static_offset"],
                                                  dynamic_offset=Pl["dynamic_offset"][ts])
            tr = object.__new__(T.FlameTracker)
            tr.cfg, tr.device = rcfg, "cpu"
            st = rcfg.pipeline[stage]
            tr.opt_dict = defaultdict(bool, {k: True for k in st.optimizable_params})
            tr.n_timesteps = N
            for k in P:
                setattr(tr, k, Pl[k])
            tr.lights_uniform = torch.zeros(9, 3, dtype=dt)
            tr.flame = types.SimpleNamespace(mask=types.SimpleNamespace(get_vid_by_region=vid), laplacian_matrix=Lap,
                                             laplacian_matrix_negate_diag=Lap - 2 * torch.diag(torch.diag(Lap)))
            for k in ("texture", "lights"):                      # (need the albedo / a render: pinned elsewhere)
                tr.opt_dict[k] = False
            log = tr.compute_regularization_energy({}, verts, v_cano, lmks, None, ts, stage)
            E = torch.stack([v for v in log.values()]).sum()
            E.backward()
            for k, v in log.items():
                out[f"{case}/{stage}/log/{k}"] = v.detach().numpy()
            for k in ("static_offset", "dynamic_offset", "expr", "shape"):
                if Pl[k] is not None and Pl[k].grad is not None:
                    gk = Pl[k].grad.numpy()
                    out[f"{case}/{stage}/grad/{k}"] = gk[:, pick] if k.endswith("offset") else gk
            print(case, stage, {k: float(v) for k, v in log.items()})
    # ... and the failure itself: both offsets, B = 3
    Pl = {k: v.clone() for k, v in P.items()}
    ts = np.array([1, 3, 0])
    verts, v_cano, lmks = R.flame_forward(tm, Pl["shape"][None].expand(3, -1), Pl["expr"][ts], Pl["rotation"][ts], Pl["neck_pose"][ts],
                                          Pl["jaw_pose"][ts], Pl["eyes_pose"][ts], Pl["translation"][ts], static_offset=Pl["static_offset"],
                                          dynamic_offset=Pl["dynamic_offset"][ts])
    tr = object.__new__(T.FlameTracker)
    tr.cfg, tr.device, tr.n_timesteps = rcfg, "cpu", N
    tr.opt_dict = defaultdict(bool, {"dynamic_offset": True})
    for k in P:
        setattr(tr, k, Pl[k])
    try:
        tr.compute_regularization_energy({}, verts, v_cano, lmks, None, ts, "rgb_sequential_tracking")
        out["both_B3_raises"] = np.array(0)
    except RuntimeError as e:
        out["both_B3_raises"] = np.array(1)
        print("both offsets, B = 3: the reference raises:", str(e)[:100])
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
