"""first-run effect hunt: the multi-view stage loop three times in one process, pairwise differences of the fitted parameters"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vhap_amd.config import nersemble_config
from vhap_amd.flame import FlameHead
from vhap_amd.ingest import FrameStore
from vhap_amd.render_hip import HipDiffRenderer
from vhap_amd.synthetic import make_flame_model, make_multiview_dataset, make_scene_params, make_texture
from vhap_amd.tracker import GlobalTracker, ShuffledBatches
model, topo = make_flame_model(0)
H, W, NV, NT, T, stage = 96, 128, 4, 3, 256, "rgb_global_tracking"
cfg = nersemble_config(); cfg.model.tex_resolution = T
if os.environ.get("DIST", "1") != "1":
    cfg.render.disturb_rate_fg = cfg.render.disturb_rate_bg = None
head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
parts = [make_multiview_dataset(rend, head, make_scene_params(1, seed=20 + t, image_size=(H, W)), (H, W), "cuda", n_views=NV, seed=20 + t, tex=make_texture(3, T)) for t in range(NT)]
rgb = torch.cat([p["rgb"] for p in parts])
u8 = (rgb.permute(0, 2, 3, 1).clamp(0, 1) * 255).round().to(torch.uint8).contiguous()
base = {"lmk2d": torch.cat([p["lmk2d"] for p in parts]).contiguous(), "intrinsic": torch.cat([p["intrinsic"] for p in parts]).float().contiguous(),
        "extrinsic": torch.cat([p["extrinsic"] for p in parts]).float().contiguous(), "timestep_index": torch.arange(NT).repeat_interleave(NV)}
cfg.pipeline[stage].num_epochs = int(os.environ.get("EPOCHS", "3"))
NAMES = ("shape", "expr", "rotation", "jaw_pose", "translation", "tex_extra", "lights")
res = []
for r in range(3):
    os.environ["VHAP_STEP_FEED"] = os.environ.get("ORDER", "111")[r]
    tr = GlobalTracker(cfg, model, topo, make_texture(0, T), dict(base, frames=FrameStore(u8, device="cuda")))
    if os.environ.get("DIST", "1") != "1":
        tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for name, s_ in (("shape", 0.2), ("expr", 0.2), ("rotation", 0.03), ("jaw_pose", 0.05), ("tex_extra", 0.02)):
            p = getattr(tr, name); p.add_((torch.randn(p.shape, generator=g) * s_).cuda())
    tr.render._rng_state = torch.full((1,), 777, dtype=torch.int32, device="cuda")
    start = {k: getattr(tr, k).detach().cpu().numpy().copy() for k in NAMES}
    tr.optimize_stage(stage, dataloader=ShuffledBatches(tr, 1, device_index=True, generator=torch.Generator().manual_seed(9)), lr_scale=0.1)
    torch.cuda.synchronize()
    st = next(iter(tr._graphed.values()))
    print("run", r, "feed", st.feed is not None, "defer", st.defer_join, "rng", int(tr.render._rng_state), "steps", tr.global_step)
    res.append((start, {k: getattr(tr, k).detach().cpu().numpy().copy() for k in NAMES}))
upd = lambda a, b, s: float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm((b - s).ravel()), 1e-12))
for a, b in ((0, 1), (1, 2)):
    print("run", a, "vs", b, {k: f"{upd(res[a][1][k], res[b][1][k], res[b][0][k]):.1e}" for k in NAMES})
