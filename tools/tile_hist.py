"""Distribution of the texture-gradient work over the uv tiles at a bench configuration (how unbalanced is texgrad_tile_kernel's grid?).
usage: python tools/tile_hist.py [--config 2]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
args = ap.parse_args()
C = bench.CONFIGS[args.config]
tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
from vhap_amd.step import NativeStep  # noqa: E402
opt = tr.configure_optimizer(tr.get_train_parameters(bench.STAGE), lr_scale=0.1)
ns = NativeStep(tr, tr.get_sample(own, device_index=True), bench.STAGE)
ns.forward()
ns.backward(1)
torch.cuda.synchronize()
ids = ns.tile_ids.reshape(-1).to(torch.int64) & 0xFFFF
keep = ns.keep.reshape(-1) if ns.disturb_on else torch.ones_like(ids, dtype=torch.float32)
live = (ids != 0xFFFF) & (keep != 0)
cnt = torch.bincount(ids[live], minlength=4096).cpu().numpy()
nz = cnt[cnt > 0]
print(f"config {args.config}: {int(live.sum())} gradient pixels over {len(nz)} of {len(cnt)} tiles")
print("  per-tile pixels: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %d" % (nz.mean(), np.median(nz), np.percentile(nz, 90), np.percentile(nz, 99), nz.max()))
order = np.sort(nz)[::-1]
print("  largest 16:", order[:16].tolist())
for thr in (256, 512, 1024, 2048, 4096, 8192):
    print(f"  tiles with > {thr} pixels: {(nz > thr).sum()}  (hold {nz[nz > thr].sum() / nz.sum():.1%} of the pixels)")
