"""How far apart are two CORRECT evaluations of the step gradient -- the oracle in float64 and the same oracle in float32 (same triangle
ids) -- at the BASELINE sizes?  The yardstick for the gradient tolerances of tests/test_parity_sizes_gpu.py.  CPU only.

    python tools/grad_fp32_spread.py cfg2|cfg3|cfg4      -> prints per-parameter max-norm relative distance and cosine
"""
import sys
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    import oracle
    from oracle import energy_ref
    from oracle import torch_ref as R
    from vhap_amd.config import BaseTrackingConfig, nersemble_config
    from vhap_amd.synthetic import arc_cameras, make_flame_model, make_texture, smooth_noise
    T = 2048
    model, topo = make_flame_model(seed=0)
    BS = int(sys.argv[2]) if len(sys.argv) > 2 else None
    H, W, B, stage, calibrated = {"cfg2": (512, 512, BS or 2, "rgb_global_tracking", False), "cfg3": (1024, 1024, BS or 1, "rgb_init_offset", False),
                                  "cfg4": (802, 550, BS or 2, "rgb_global_tracking", True)}[which]
    cfg = nersemble_config() if calibrated else BaseTrackingConfig()
    cfg.model.tex_resolution = T
    g = torch.Generator().manual_seed(29)
    N = 1 if calibrated else B
    P0 = {"shape": torch.randn(300, generator=g) * 0.3, "expr": torch.randn(N, 100, generator=g) * 0.3, "rotation": torch.randn(N, 3, generator=g) * 0.1,
          "neck_pose": torch.randn(N, 3, generator=g) * 0.03, "jaw_pose": torch.randn(N, 3, generator=g) * 0.05,
          "eyes_pose": torch.randn(N, 6, generator=g) * 0.05, "translation": torch.randn(N, 3, generator=g) * 0.01,
          "tex_extra": torch.randn(3, T, T, generator=g) * 0.03, "lights": torch.randn(9, 3, generator=g) * 0.05,
          "static_offset": torch.randn(1, 5143, 3, generator=g) * 1e-3, "focal_length": torch.tensor([1.5])}
    P0["lights"][0] += float(np.sqrt(4 * np.pi))
    P0["jaw_pose"][:, 0] += 0.1
    if not calibrated:
        P0["translation"][:, 2] += 0.45
    rng = np.random.default_rng(0)
    ts = np.zeros(B, np.int64) if calibrated else np.arange(B)
    sample = {"rgb": torch.from_numpy(smooth_noise(rng, (B, 3, H, W))), "lmk2d": torch.cat([torch.rand(B, 70, 2) * W, torch.ones(B, 70, 1)], -1),
              "timestep_index": ts}
    if calibrated:
        K, RT = arc_cameras(B, (H, W))
        sample["intrinsic"], sample["extrinsic"] = torch.from_numpy(K), torch.from_numpy(RT)
    base = torch.from_numpy(make_texture(0, T))[None]
    uvm = torch.from_numpy(topo.get_uvmask_by_region(["sclerae", "teeth"]).astype(np.float32))[None]
    if uvm.shape[-1] != T:
        uvm = torch.nn.functional.interpolate(uvm[None], (T, T))[0]
    out = {}
    tid = None
    for dt in (torch.float64, torch.float32):
        tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
            tm[k] = tm[k].to(dt)
        P = {k: v.to(dt).clone().requires_grad_() for k, v in P0.items() if not (calibrated and k == "focal_length")}
        E, log, ex = energy_ref.total_energy(P, tm, topo, cfg, sample, stage, base.to(dt), uvm.to(dt), (H, W), dtype=dt, tid=tid)
        tid = ex["tid"]
        E.backward()
        out[dt] = ({k: p.grad.double().reshape(-1) for k, p in P.items() if p.grad is not None}, float(E))
    g64, g32 = out[torch.float64][0], out[torch.float32][0]
    print(f"{which}: {B} x {H}x{W}, T = {T}, stage {stage}: oracle float32 vs oracle float64 (same triangle ids); E rel "
          f"{abs(out[torch.float32][1] - out[torch.float64][1]) / abs(out[torch.float64][1]):.2e}")
    for k in g64:
        a, b = g32[k], g64[k]
        if float(b.abs().max()) == 0:
            continue
        print(f"  grad {k}: max-norm rel {float((a - b).abs().max() / b.abs().max()):.2e}  cos {float((a @ b) / (a.norm() * b.norm())):.7f}")


if __name__ == "__main__":
    main()
