"""Where does the HIP-vs-oracle distance of d(tex_extra) at FULL batch come from?  (tests/test_parity_fullbatch_gpu.py: 4.8e-4 of the
max-norm at 16 x 512^2 and 4.2e-3 at 8 x 1024^2, against 1e-5 at B <= 2.)  Runs the shipped NativeStep with injected disturbance and the
float64 oracle on the same state, then: the texels whose gradient differs by more than 1e-4 of the max-norm, the pixels whose L1 residual
changes sign between the two evaluations (|x| has a kink at 0: both signs are valid subgradients there), and how much of the distance is
left when the oracle is told to take the HIP path's side of every kink (`sign_from`).

    python tools/diag_texgrad.py [cfg2|cfg3] [B]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import energy_ref
    from tests.test_fit_parity_gpu import NAMES, _make
    from vhap_amd.step import NativeStep
    from vhap_amd.synthetic import make_flame_model
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    H, W, B, stage, seed, dseed = {"cfg2": (512, 512, 16, "rgb_global_tracking", 17, 12), "cfg3": (1024, 1024, 8, "rgb_init_offset", 29, 5)}[which]
    if len(sys.argv) > 2:
        B = int(sys.argv[2])
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    T = 2048
    S = _make(make_flame_model(seed=0), H, W, B, T, seed=seed)
    tr, cfg, topo, tm = S["tr"], S["cfg"], S["topo"], S["tm"]
    ts = np.arange(B)
    sample = tr.get_sample(ts, device_index=True)
    tr.get_train_parameters(stage)
    dist = tr.render.make_disturbance((B, H, W), "cuda", generator=torch.Generator("cuda").manual_seed(dseed))
    ns = NativeStep(tr, sample, stage)
    ns.injected = dist
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    g_hip = ns.g["tex_extra"].detach().cpu().double().reshape(3, T, T)
    pred_hip = ns.rgba_aa.detach().cpu().double().flip(1)         # [B,H,W,4]: renderer space (row 0 = bottom) -> image space
    keep = ns.keep.detach().cpu().flip(1)
    tid = (ns.rast[..., 3].long() - 1).cpu()
    ncl = int(topo.fid2cid.max()) + 1
    o_dist = dict(w_fg=dist["w_fg"].cpu(), w_bg=dist["w_bg"].cpu(), idx=[dist["idx"].cpu()] * ncl, fid2cid=torch.from_numpy(topo.fid2cid.astype(np.int64)))
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts}
    gt = sample["rgb"].cpu().double().permute(0, 2, 3, 1)         # -> [B,H,W,3] image space
    res_hip = pred_hip[..., :3] - gt

    def oracle(sign_from=None):
        P = {k: getattr(tr, k).detach().cpu().double().requires_grad_() for k in NAMES}
        Eo, logo, ex = energy_ref.total_energy(P, tm, topo, cfg, o_sample, stage, S["base_tex"], tr._uvmask_res().cpu().double(), (H, W),
                                               disturb=o_dist, tid=tid, photo_sign_from=sign_from)
        Eo.backward()
        return P["tex_extra"].grad.reshape(3, T, T), ex["rgba"].detach(), float(Eo.detach())

    g_ora, pred_ora, E0 = oracle()
    res_ora = pred_ora[..., :3] - gt
    mx = float(g_ora.abs().max())
    err = (g_hip - g_ora).abs()
    print(f"{which}: {B} x {H}x{W}: d(tex_extra) max-norm rel {float(err.max()) / mx:.2e}, L2 rel {float((g_hip - g_ora).norm() / g_ora.norm()):.2e}, "
          f"texels off by > 1e-4 max: {int((err > 1e-4 * mx).sum())}, > 1e-5 max: {int((err > 1e-5 * mx).sum())} of {int((g_ora != 0).sum())} non-zero")
    flips = (torch.sign(res_hip) != torch.sign(res_ora)) & (pred_hip[..., 3:4] >= 0)
    print(f"pixels x channels whose residual changes sign between HIP (fp32) and oracle (fp64): {int(flips.sum())}; max |residual| among them "
          f"{float(res_ora.abs()[flips].max()) if int(flips.sum()) else 0:.2e}; kept (not disturbed) among them: {int((flips & (keep[..., None] > 0)).sum())}")
    print(f"max |pred_hip - pred_oracle| = {float((pred_hip - pred_ora).abs().max()):.2e}")
    g_ora2, _, E1 = oracle(sign_from=res_hip)
    err2 = (g_hip - g_ora2).abs()
    print(f"oracle on the HIP side of every kink: d(tex_extra) max-norm rel {float(err2.max()) / mx:.2e}, L2 rel "
          f"{float((g_hip - g_ora2).norm() / g_ora2.norm()):.2e}; energy moved by {abs(E1 - E0) / abs(E0):.1e}")
    k = torch.topk(err2.reshape(-1), 5)
    for v, i in zip(k.values.tolist(), k.indices.tolist()):
        c, y, x = i // (T * T), (i // T) % T, i % T
        print(f"  texel c{c} ({y},{x}): hip {float(g_hip[c, y, x]):+.4e} oracle {float(g_ora2[c, y, x]):+.4e}  err/max {v / mx:.2e}")


if __name__ == "__main__":
    main()
