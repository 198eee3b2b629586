"""Is the 1e-3 .. 1e-2 distance of ONE gradient per evaluation that bench.py's parity leg shows at TRAINED states (DESIGN section 7,
profiles/r04_call31_32_bench_parity_states.txt) a property of comparing a float32 with a float64 evaluation, whoever computes them?  CPU
only, no HIP: the oracle fits config 2 (B frames of 512^2, T = 2048, lr_scale 0.1 like bench.py, disturbance off) in float64, and at the
states after the given numbers of steps its gradient is evaluated in float64 and in float32 -- same triangle ids, the float32 side of every
L1 kink (photo_sign_from), exactly what the HIP-vs-oracle comparison does -- and compared per parameter as a fraction of its max-norm, with
the distribution of d(tex_extra)'s distance over the texel channels.

    python tools/trained_state_spread.py [B=2] [steps ...=0 40 100]
    PLANT=K python tools/trained_state_spread.py 2 0     K covered pixel channels of the TARGET are set to the float32 prediction bit for bit first (exactly-zero
                                                         float32 residuals, which a fit reaches by itself about once per 10^7 channels); OLD_ZERO=1: the oracle's
                                                         handling of that case before the fix (fall back to the float64 sign)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import energy_ref, fit_ref
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.synthetic import make_flame_model, make_texture, smooth_noise
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    marks = [int(a) for a in sys.argv[2:]] or [0, 40, 100]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    T, H, W, stage = 2048, 512, 512, "rgb_global_tracking"
    model, topo = make_flame_model(seed=0)
    cfg = BaseTrackingConfig()
    cfg.model.tex_resolution = T
    g = torch.Generator().manual_seed(29)
    P = {"shape": torch.randn(300, generator=g) * 0.3, "expr": torch.randn(B, 100, generator=g) * 0.3, "rotation": torch.randn(B, 3, generator=g) * 0.1,
         "neck_pose": torch.randn(B, 3, generator=g) * 0.03, "jaw_pose": torch.randn(B, 3, generator=g) * 0.05,
         "eyes_pose": torch.randn(B, 6, generator=g) * 0.05, "translation": torch.randn(B, 3, generator=g) * 0.01,
         "tex_extra": torch.randn(3, T, T, generator=g) * 0.03, "lights": torch.randn(9, 3, generator=g) * 0.05,
         "static_offset": torch.randn(1, 5143, 3, generator=g) * 1e-3, "focal_length": torch.tensor([1.5])}
    P["lights"][0] += float(np.sqrt(4 * np.pi))
    P["jaw_pose"][:, 0] += 0.1
    P["translation"][:, 2] += 0.45
    P = {k: v.double().requires_grad_() for k, v in P.items()}
    rng = np.random.default_rng(0)
    sample = {"rgb": torch.from_numpy(smooth_noise(rng, (B, 3, H, W))), "lmk2d": torch.cat([torch.rand(B, 70, 2) * W, torch.ones(B, 70, 1)], -1),
              "timestep_index": np.arange(B)}
    base = torch.from_numpy(make_texture(0, T))[None]
    uvm = torch.from_numpy(topo.get_uvmask_by_region(["sclerae", "teeth"]).astype(np.float32))[None]
    if uvm.shape[-1] != T:
        uvm = torch.nn.functional.interpolate(uvm[None], (T, T))[0]

    def tm_of(dt):
        tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
            tm[k] = tm[k].to(dt)
        return tm
    tm64 = tm_of(torch.float64)
    opt = fit_ref.configure_optimizer(P, cfg, stage, lr_scale=0.1)

    def compare(step):
        # float32 first: its triangle ids and its side of the L1 kinks are handed to the float64 evaluation (the roles HIP plays in bench.py)
        P32 = {k: v.detach().float().clone().requires_grad_() for k, v in P.items()}
        E32, _, ex32 = energy_ref.total_energy(P32, tm_of(torch.float32), topo, cfg, sample, stage, base.float(), uvm.float(), (H, W), dtype=torch.float32)
        E32.backward()
        res32 = (ex32["rgba"][..., :3].detach() - sample["rgb"].permute(0, 2, 3, 1)).float()
        P64 = {k: v.detach().clone().requires_grad_() for k, v in P.items()}
        E64, _, ex64 = energy_ref.total_energy(P64, tm64, topo, cfg, sample, stage, base.double(), uvm.double(), (H, W), tid=ex32["tid"], photo_sign_from=res32)
        E64.backward()
        kink = int((torch.sign(ex64["rgba"][..., :3].detach() - sample["rgb"].permute(0, 2, 3, 1).double()) != torch.sign(res32.double())).sum())
        print(f"after {step} steps: E {float(E64):.6f}, float32 vs float64 energy rel {abs(float(E32) - float(E64)) / abs(float(E64)):.1e}, L1 kink pixels {kink}", flush=True)
        for k in P64:
            b = P64[k].grad
            if b is None or float(b.abs().max()) == 0:
                continue
            a, b = P32[k].grad.double().reshape(-1), b.reshape(-1)
            line = f"   grad {k}: {float((a - b).abs().max() / b.abs().max()):.2e} of its max-norm {float(b.abs().max()):.2e}"
            if k == "tex_extra":
                d = (a - b).abs() / b.abs().max()
                line += f"; texel channels over 1e-4: {int((d > 1e-4).sum())}, over 1e-3: {int((d > 1e-3).sum())}, without the worst 32: {float(d.topk(33).values[-1]):.2e}"
            print(line, flush=True)

    if os.environ.get("OLD_ZERO") == "1":                 # (the sign_from handling of an exactly-zero residual before round 4's fix)
        from oracle import torch_ref as R

        def old_photo(gt_rgb_nchw, rgba_nhwc_flipped, sign_from=None):
            pred = rgba_nhwc_flipped.permute(0, 3, 1, 2)
            mask = (pred[:, 3:4].detach() > 0).expand(-1, 3, -1, -1)
            x = pred[:, :3] - gt_rgb_nchw
            if sign_from is None:
                return x.abs().sum() / mask.sum()
            sg = torch.sign(sign_from.to(x.dtype).permute(0, 3, 1, 2))
            sg = torch.where(sg == 0, torch.sign(x.detach()), sg)
            return (sg * x).sum() / mask.sum()
        R.photometric_energy = old_photo
    K = int(os.environ.get("PLANT", "0"))
    if K:
        P32 = {k: v.detach().float() for k, v in P.items()}
        with torch.no_grad():
            _, _, ex = energy_ref.total_energy(P32, tm_of(torch.float32), topo, cfg, sample, stage, base.float(), uvm.float(), (H, W), dtype=torch.float32)
        rgba = ex["rgba"].detach()
        cov = (rgba[..., 3] > 0).nonzero()
        pick = cov[torch.randperm(cov.shape[0], generator=torch.Generator().manual_seed(5))[:K]]
        tgt = sample["rgb"].permute(0, 2, 3, 1).clone()
        for i, (b, y, x) in enumerate(pick.tolist()):
            tgt[b, y, x, i % 3] = rgba[b, y, x, i % 3]
        sample["rgb"] = tgt.permute(0, 3, 1, 2).contiguous()
        print(f"planted: {K} covered pixel channels of the target := the float32 prediction" + ("  [OLD_ZERO=1: the oracle's zero handling before the fix]" if os.environ.get("OLD_ZERO") == "1" else ""), flush=True)
    t0, step = time.time(), 0
    for m in sorted(marks):
        while step < m:
            fit_ref.optimize_iter(P, opt, tm64, topo, cfg, sample, stage, base.double(), uvm.double(), (H, W))
            step += 1
        compare(step)
    print(f"({time.time() - t0:.0f} s on {torch.get_num_threads()} threads)")


if __name__ == "__main__":
    main()
