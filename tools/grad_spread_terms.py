#!/usr/bin/env python3
"""Where does the HIP step's gradient stand FURTHER from the float64 oracle than the float32 oracle does?  (round 5: at BASELINE config 4 the
pose-like parameters are 2 - 4x the fp32 oracle's distance, at config 2 they coincide.)  Per-parameter distances for variants of the
2-view 802 x 550 scene: all terms / landmarks off / the separate passes instead of the deferred kernels / a square power-of-two frame.

    python tools/grad_spread_terms.py > gpurun_out/grad_spread_terms.txt          (GPU + the CPU oracle; ~3 min)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(tag, NV=2, H=802, W=550, mutate=None, deferred=True):
    from oracle import energy_ref
    from vhap_amd.config import nersemble_config
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.step import NativeStep
    from vhap_amd.synthetic import make_flame_model, make_multiview_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    os.environ["VHAP_DEFERRED"] = "1" if deferred else "0"
    T = 2048
    model, topo = make_flame_model(seed=0)
    cfg = nersemble_config()
    cfg.model.tex_resolution = T
    if mutate:
        mutate(cfg)
    gt = make_scene_params(1, seed=3, image_size=(H, W))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_multiview_dataset(rend, head, gt, (H, W), "cuda", n_views=NV, seed=3, tex=make_texture(3, T))
    base_tex = make_texture(0, T)
    tr = GlobalTracker(cfg, model, topo, base_tex, data)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for name, s_ in (("shape", 0.3), ("expr", 0.3), ("rotation", 0.05), ("neck_pose", 0.03), ("jaw_pose", 0.05), ("eyes_pose", 0.05),
                         ("translation", 0.005), ("tex_extra", 0.03), ("lights", 0.05), ("static_offset", 1e-3)):
            p = getattr(tr, name)
            p.add_((torch.randn(p.shape, generator=g) * s_).cuda())
        tr.jaw_pose[:, 0] += 0.1
    stage = "rgb_global_tracking"
    names = ["shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset"]
    sample = tr.get_sample(np.array([0]), device_index=True)
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": sample["timestep_index"].cpu().numpy(),
                "intrinsic": sample["intrinsic"].cpu(), "extrinsic": sample["extrinsic"].cpu()}
    tr.get_train_parameters(stage)
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None           # (disturbance off: one source of difference less)
    ns = NativeStep(tr, sample, stage)
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    tid = (ns.rast[..., 3].long() - 1).cpu()
    res_hip = (ns.rgba_aa[..., :3].detach().flip(1) - sample["rgb"].permute(0, 2, 3, 1)).cpu()
    g_n = {k: ns.g[k].detach().cpu().double().reshape(-1) for k in names if k in ns.g}
    uvm = tr._uvmask_res().cpu()
    out = {}
    for dt in (torch.float64, torch.float32):
        tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
            tm[k] = tm[k].to(dt)
        P = {k: getattr(tr, k).detach().cpu().to(dt).requires_grad_() for k in names}
        E, _, _ = energy_ref.total_energy(P, tm, topo, cfg, o_sample, stage, torch.from_numpy(base_tex)[None].to(dt), uvm.to(dt), (H, W), dtype=dt,
                                          tid=tid, photo_sign_from=res_hip)
        E.backward()
        out[dt] = {k: P[k].grad.double().reshape(-1) for k in names if P[k].grad is not None}
    print(f"---- {tag}: {NV} views {H}x{W}, deferred kernels {bool(ns.deferred)}, coverage {float((tid >= 0).float().mean()):.3f}")
    for k in names:
        g64 = out[torch.float64].get(k)
        if g64 is None or float(g64.abs().max()) == 0 or k not in g_n:
            continue
        nrm = float(g64.abs().max())
        e_h = float((g_n[k] - g64).abs().max()) / nrm
        e_32 = float((out[torch.float32][k] - g64).abs().max()) / nrm
        e_h32 = float((g_n[k] - out[torch.float32][k]).abs().max()) / nrm
        print(f"  {k:14s} HIP-f64 {e_h:.2e}   orc32-f64 {e_32:.2e}   HIP-orc32 {e_h32:.2e}   ratio {e_h / max(e_32, 1e-30):5.2f}   max-norm {nrm:.3e}", flush=True)
    del ns, tr


if __name__ == "__main__":
    def no_lmk(cfg):
        cfg.w.landmark = 1e-9
    run("all terms")
    run("landmark term off", mutate=no_lmk)
    run("separate passes (VHAP_DEFERRED=0)", deferred=False)
    run("square 512 x 512 frame", H=512, W=512)
    run("four views", NV=4)
