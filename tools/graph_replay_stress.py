import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vhap_amd.config import BaseTrackingConfig
from vhap_amd.flame import FlameHead
from vhap_amd.render_hip import HipDiffRenderer
from vhap_amd.synthetic import make_dataset, make_flame_model, make_scene_params, make_texture
from vhap_amd.tracker import GlobalTracker, GraphedStep
H = W = 128; N = 3; T = 256
model, topo = make_flame_model(0)
gt = make_scene_params(N, seed=9, image_size=(H, W))
head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
data = make_dataset(rend, head, gt, (H, W), "cuda", seed=9, tex=make_texture(9, T))
stage = "rgb_global_tracking"
cfg = BaseTrackingConfig(); cfg.model.tex_resolution = T
cfg.render.disturb_rate_fg = cfg.render.disturb_rate_bg = None
tr = GlobalTracker(cfg, model, topo, make_texture(0, T), data)
with torch.no_grad():
    tr.translation[:, 2] = 0.45; tr.expr.add_(0.05)
    tr.static_offset.add_(torch.randn(tr.static_offset.shape, generator=torch.Generator().manual_seed(2)).cuda() * 1e-4)
opt = tr.configure_optimizer(tr.get_train_parameters(stage))
sample = tr.get_sample(np.array([0, 1]), device_index=True)
st = GraphedStep(tr, sample, opt, stage, warmup=0)

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
stream = torch.cuda.Stream() if mode == "stream" else torch.cuda.current_stream()
bad = 0
best = 0.0
with torch.cuda.stream(stream):
    for i in range(80):
        if mode == "call":
            st()
        else:
          st.gF.replay()
        if mode == "call": pass
        elif mode == "sync": torch.cuda.synchronize()
        if mode != "call":
          st.inv_n.copy_(1 / (3.0 * st.N))
          if mode == "sync": torch.cuda.synchronize()
          st.gB.replay()
          if mode == "sync": torch.cuda.synchronize()
          st.gA.replay()
        if mode == "sync": torch.cuda.synchronize()
        if mode == "eagerops":
            for _ in range(5):
                junk = [torch.randn(1000, device="cuda") * 3 for _ in range(10)]
        terms = {k: float(v.detach()) for k, v in st.log_dict.items()}
        E = float(st.E)
        if st.ns is not None:      # NativeStep: total = sum of the logged terms, every term finite and in range
            tot, ph = terms["total"], terms["photo"]
            parts = sum(v for k, v in terms.items() if k != "total")
            ok = abs(E - tot) < 1e-6 * abs(E) and abs(parts - tot) < 1e-3 * abs(tot) and all(0 <= v < 1e3 for v in terms.values())
            # and a second opinion from the eager autograd formulation at the parameters the step just used is not possible after the
            # update; check monotone sanity instead: the energy must stay within a factor 3 of its running minimum
            best = min(best, E) if i else E
            ok = ok and E < 3.0 * best + 1.0
        else:
            tot = float(st.log_dict["total"].detach()); ph = 30 * float(st.S) * float(st.inv_n)
            ok = abs(E - tot - ph) < 1e-3 * abs(E) and all(0 <= v < 1e3 for v in terms.values()) and abs(sum(v for k, v in terms.items() if k != "total") - tot) < 1e-3
        if not ok:
            bad += 1
            if bad <= 3: print("   BAD step", i, "E=%.4g tot=%.4g photo=%.4g" % (E, tot, ph), {k: "%.3g" % v for k, v in terms.items() if not (0 <= v < 1e3)})
print(mode, os.environ.get("VHAP_GRAPH_POOLS"), os.environ.get("AMD_SERIALIZE_KERNEL"), "bad steps:", bad, "final E=%.4f" % E)
