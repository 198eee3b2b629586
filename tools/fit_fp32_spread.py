"""How far apart do two CORRECT implementations of the same K-step fit land when one computes in fp32 and the other in fp64?
Runs oracle/fit_ref.py (energy_ref.total_energy + torch.optim.Adam) twice from the same start -- float64 and float32 -- on the
K-step parity test's configuration (tests/test_fit_parity_gpu.py) and prints the same per-array measures the test uses.  CPU only.
The target frames are rendered by the oracle itself here (the GPU test renders them with the product); only the spread matters."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import energy_ref, fit_ref
from vhap_amd.config import BaseTrackingConfig
from vhap_amd.synthetic import make_flame_model, make_texture, smooth_noise

NAMES = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset", "focal_length")
stage, lr_scale, K = sys.argv[1] if len(sys.argv) > 1 else "rgb_init_offset", float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, 10
H = W = 128
T, N = 256, 3
model, topo = make_flame_model(0)
cfg = BaseTrackingConfig()
cfg.model.tex_resolution = T
g = torch.Generator().manual_seed(23)
start = {"shape": torch.zeros(300), "expr": torch.zeros(N, 100), "rotation": torch.zeros(N, 3), "neck_pose": torch.zeros(N, 3), "jaw_pose": torch.zeros(N, 3),
         "eyes_pose": torch.zeros(N, 6), "translation": torch.zeros(N, 3), "tex_extra": torch.zeros(3, T, T), "lights": torch.zeros(9, 3),
         "static_offset": torch.zeros(1, topo.num_verts, 3), "focal_length": torch.tensor([1.5])}
start["lights"][0] = float(np.sqrt(4 * np.pi))
for name, s in (("shape", 0.3), ("expr", 0.3), ("rotation", 0.1), ("neck_pose", 0.03), ("jaw_pose", 0.05), ("eyes_pose", 0.05), ("translation", 0.01),
                ("tex_extra", 0.03), ("lights", 0.05), ("static_offset", 1e-3)):
    start[name] = start[name] + torch.randn(start[name].shape, generator=g) * s
start["translation"][:, 2] += 0.45
start["jaw_pose"][:, 0] += 0.1
ts = np.array([1, 2]) if stage != "rgb_global_tracking" else np.array([0, 1, 2])
rng = np.random.default_rng(0)
sample = {"rgb": torch.from_numpy(smooth_noise(rng, (len(ts), 3, H, W))), "lmk2d": torch.cat([torch.rand(len(ts), 70, 2, generator=g) * W, torch.ones(len(ts), 70, 1)], -1),
          "timestep_index": ts}
uvm = torch.from_numpy(topo.get_uvmask_by_region(list(cfg.w.reg_tex_res_for)))[None].float()
uvm = torch.nn.functional.interpolate(uvm[None], (T, T), mode="nearest")[0]
base = torch.from_numpy(make_texture(0, T))[None]


def run(dt):
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    P = {k: v.clone().to(dt).requires_grad_() for k, v in start.items()}
    opt = fit_ref.configure_optimizer(P, cfg, stage, lr_scale=lr_scale)
    E = []
    for _ in range(K):
        Et, log, _ = energy_ref.total_energy(P, tm, topo, cfg, sample, stage, base.to(dt), uvm.to(dt), (H, W), dtype=dt)
        opt.zero_grad()
        Et.backward()
        opt.step()
        E.append(float(Et))
    return {k: v.detach().double().numpy() for k, v in P.items()}, E


a, Ea = run(torch.float32)
b, Eb = run(torch.float64)
print(f"fp32 oracle vs fp64 oracle, stage {stage} lr_scale {lr_scale} K {K}")
print("energy rel:", " ".join(f"{abs(x - y) / abs(y):.1e}" for x, y in zip(Ea, Eb)))
for k in NAMES:
    s0 = start[k].double().numpy()
    mx = np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-12)
    l2 = np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(b[k]), 1e-300)
    dl2 = np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(b[k] - s0), 1e-300)
    print(f"{k}: max-norm rel {mx:.2e}  L2 rel {l2:.2e}  update L2 rel {dl2:.2e}")
