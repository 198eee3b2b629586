import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vhap_amd.config import nersemble_config
from vhap_amd.flame import FlameHead
from vhap_amd.ingest import FrameStore
from vhap_amd.render_hip import HipDiffRenderer
from vhap_amd.synthetic import make_flame_model, make_multiview_dataset, make_scene_params, make_texture
from vhap_amd.tracker import GlobalTracker, GraphedStep
model, topo = make_flame_model(0)
H, W, NV, NT, T, stage = 96, 128, 4, 3, 256, "rgb_global_tracking"
cfg = nersemble_config(); cfg.model.tex_resolution = T
DIST = os.environ.get("DIST", "1") == "1"
LR = float(os.environ.get("LR", "0.1"))
if not DIST:
    cfg.render.disturb_rate_fg = cfg.render.disturb_rate_bg = None
head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
parts = [make_multiview_dataset(rend, head, make_scene_params(1, seed=20 + t, image_size=(H, W)), (H, W), "cuda", n_views=NV, seed=20 + t, tex=make_texture(3, T)) for t in range(NT)]
rgb = torch.cat([p["rgb"] for p in parts])
u8 = (rgb.permute(0, 2, 3, 1).clamp(0, 1) * 255).round().to(torch.uint8).contiguous()
base = {"lmk2d": torch.cat([p["lmk2d"] for p in parts]).contiguous(), "intrinsic": torch.cat([p["intrinsic"] for p in parts]).float().contiguous(),
        "extrinsic": torch.cat([p["extrinsic"] for p in parts]).float().contiguous(), "timestep_index": torch.arange(NT).repeat_interleave(NV)}
print({k: (tuple(v.shape), v.dtype, v.is_contiguous()) for k, v in base.items()})
out = {}
runs = []
for feed in [bool(int(c)) for c in os.environ.get("ORDER", "10")]:
    tr = GlobalTracker(cfg, model, topo, make_texture(0, T), dict(base, frames=FrameStore(u8, device="cuda")))
    if not DIST:
        tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    tr.render._rng_state = torch.full((1,), 777, dtype=torch.int32, device="cuda")
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=LR)
    s0 = tr.get_sample(np.array([0]), device_index=True)
    st = GraphedStep(tr, s0, opt, stage, warmup=0, feed=feed)
    print("feed", feed, st.feed is not None, "defer", st.defer_join)
    Es = []
    order = [2, 1, 0, 1]
    fidx = np.concatenate([tr._frames_of[t] for t in order]); tsf = tr.frame_timestep[fidx]
    fd, td = torch.as_tensor(fidx, device="cuda"), torch.as_tensor(tsf, device="cuda")
    with st.replay_stream():
        if feed:
            st.feed_upload(fd, td)
        for i, t in enumerate(order):
            if not feed:
                st.update_timesteps(np.array([t]), fd[i * NV:(i + 1) * NV], td[i * NV:(i + 1) * NV])
            Es.append(float(st()))
            torch.cuda.synchronize()
            out[(feed, i)] = {k: v.detach().cpu().clone() for k, v in st.sample.items()}
            out[(feed, i)].update({"P_" + k: getattr(tr, k).detach().cpu().clone() for k in ("expr", "jaw_pose", "shape", "tex_extra")})
            out[(feed, i)]["rng"] = tr.render._rng_state.detach().cpu().clone() if tr.render._rng_state is not None else torch.zeros(1)
    print("E", Es)
    runs.append({i: {k: v for k, v in out[(feed, i)].items() if k.startswith("P_")} for i in range(4)})
if os.environ.get("ORDER", "10") == "10":
    for i in range(4):
        for k in out[(True, i)]:
            a, b = out[(True, i)][k].double(), out[(False, i)][k].double()
            print(i, k, float((a - b).abs().max()))

if len(runs) > 2:
    for a_, b_ in ((0, 1), (1, 2)):
        for i in range(4):
            print("run", a_, "vs run", b_, "step", i, {k: float((runs[a_][i][k].double() - runs[b_][i][k].double()).abs().max()) for k in runs[a_][i]})
