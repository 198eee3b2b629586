"""How the covered pixels of the bench scene sit in the linear pixel order the backward pixel kernels walk (one pixel per thread, 64 per wave, 256 per
workgroup): all-background workgroups / waves, and the lane utilisation of the waves that do work.  (What a launch over covered pixels only could save.)
usage: python tools/coverage_stats.py [--config 2]"""
import argparse
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_for_cov", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
C = bench.CONFIGS[a.config]
tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
sample = tr.get_sample(own, device_index=True)
H, W = C["H"], C["W"]
with torch.no_grad():
    s = dict(sample)
    tr.fill_cam_params_into_sample(s)
    verts, *_ = tr.forward_flame(s["timestep_index"])
    rd = tr.render.rasterize(verts, tr.flame.faces, s["extrinsic"], s["intrinsic"], (H, W), defer=False)
rast = rd["rast_out"]
cov = (rast[..., 3] > 0).reshape(-1)
n = cov.numel()
pad = (-n) % 256
c = torch.cat([cov, torch.zeros(pad, dtype=torch.bool, device=cov.device)])
w64, b256 = c.view(-1, 64).sum(1), c.view(-1, 256).sum(1)
act = w64 > 0
print(f"config {a.config}: {n} pixels, covered {cov.float().mean():.3f}")
print(f"256-pixel workgroups: {b256.numel()}, all background {(b256 == 0).float().mean():.3f}, full {(b256 == 256).float().mean():.3f}")
print(f"64-pixel waves: {w64.numel()}, all background {(~act).float().mean():.3f}, full {(w64 == 64).float().mean():.3f}, partial {((w64 > 0) & (w64 < 64)).float().mean():.3f}")
print(f"waves that do work: {int(act.sum())}; their lane utilisation {w64[act].float().mean() / 64:.3f}; waves of mixed workgroups that are all background: "
      f"{((~act) & (b256.repeat_interleave(4)[:w64.numel()] > 0)).float().mean():.3f} of all waves")
print(f"a launch over a pixel-compacted list: {int((cov.sum() + 63) // 64)} waves = {float((cov.sum() + 63) // 64) / int(act.sum()):.3f} of the waves that do work today")
