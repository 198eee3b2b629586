"""Is the K-step fit trajectory of the HIP path RACE-free, and where does its run-to-run spread come from?

Round 2 saw ONE run (of five) of tests/test_fit_parity_gpu.py's same-visibility rgb_init_offset trajectory end with `lights` 1e-3 (max-norm) /
3.4e-2 (update L2) away from the oracle instead of the usual 1.5e-6 / 4e-5.  Explanation offered then: the gradient sums are float
atomics in arbitrary order, and Adam's g / (|g| + eps) turns a component whose gradient sits inside that noise into a full +-lr step of
either sign.  A missing stream / graph edge would look the same.  This tool separates the candidates:

  1. R trajectories of K steps from the SAME start, in two executors -- eager NativeStep (side streams + events: what the failing test
     ran) and the captured step replayed by the plan executor (what ships) -- recording after EVERY step the gradient of every small
     parameter and, for run 0 and for outliers, every parameter;
  2. per step k and parameter: the spread of the step-k GRADIENT over runs that still share (bit for bit) the parameters of step k
     -- pure summation-order noise; a race (a kernel reading a buffer another branch has not finished) would show as a spread orders of
     magnitude above the fp32 round-off of the sums;
  3. for run 0 and every run whose exported `lights` leave the pack: per step the gradient the run computed AND an eager re-evaluation
     at the same parameters, each against the oracle at those parameters (same visibility); the oracle's max(diffuse) and the lead of its
     first colour channel over the second (reg_diffuse's max term); the components of `lights` a step moved differently from run 0,
     with their gradients.

What it found (profiles/r03_fit_flake_hunt_*.txt): before the fix, at about one parameter state in five the `lights` gradient of BOTH
executors was 90-97 % (max-norm) off the oracle's -- bit-identically in the two executors, so no race and no noise: the max term of
reg_diffuse was missing, because the backward's re-computed diffuse values did not bit-match the maximum recorded by the forward (fma
contraction differed between the two kernels; csrc/shade_common.h now pins the arithmetic).

    python tools/fit_flake_hunt.py [runs=60] [K=10]        -> gpurun_out/fit_flake_hunt.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SMALL = ("lights", "shape", "focal_length", "rotation", "translation", "neck_pose", "jaw_pose", "eyes_pose", "expr")


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    # 1: near-white lights -- the fit runs onto the ridge of reg_diffuse's max over colour channels; "1.3,1.15,1.0": channels apart, off it
    lights_scale = tuple(float(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1.0,)
    tag = "x".join(f"{v:g}" for v in lights_scale)
    from oracle import energy_ref
    from tests.test_fit_parity_gpu import NAMES, _make
    from vhap_amd.step import NativeStep
    from vhap_amd.synthetic import make_flame_model
    from vhap_amd.tracker import GraphedStep
    H = W = 128
    S = _make(make_flame_model(seed=0), H, W, 3, 256, seed=23, lights_scale=lights_scale)
    tr, cfg, topo, tm = S["tr"], S["cfg"], S["topo"], S["tm"]
    stage, lr_scale = "rgb_init_offset", 1.0
    ts = np.array([1, 2])
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    start = {k: getattr(tr, k).detach().clone() for k in NAMES}
    sample = tr.get_sample(ts, device_index=True)
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts}
    uvmask = tr._uvmask_res().cpu().double()
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=lr_scale)
    lines = [f"{runs} runs x {K} steps, stage {stage}, lr_scale {lr_scale}, {len(ts)} x {H}x{W}, T = 256, disturbance off, lights x {lights_scale}"]

    def reset():
        with torch.no_grad():
            for k in NAMES:
                getattr(tr, k).copy_(start[k])
        opt.reset_state()

    def snapshot(ns):
        return {k: ns.g[k].detach().clone() for k in NAMES if k in ns.g}

    def oracle_grad(params, tid):
        P = {k: params[k].cpu().double().requires_grad_() for k in NAMES}
        E, _, ex = energy_ref.total_energy(P, tm, topo, cfg, o_sample, stage, S["base_tex"], uvmask, (H, W), tid=tid)
        E.backward()
        out = {k: (P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])) for k in NAMES}
        d = ex["diffuse_detach_normal"].detach()                                 # reg_diffuse: max over pixels AND channels, gradient to the arg-max only
        pc = torch.sort(d.reshape(-1, d.shape[-1]).max(dim=0).values, descending=True).values
        out["_dmax"], out["_gap"] = float(pc[0]), float(pc[0] - pc[1])
        return out

    for mode in ("eager", "captured"):
        reset()
        ns_e = st = None
        if mode == "eager":
            ns_e = NativeStep(tr, sample, stage)
        else:
            st = GraphedStep(tr, sample, opt, stage, warmup=0)
            assert st.ns is not None and st.single
            reset()
        ns = ns_e if ns_e is not None else st.ns
        G = []      # [run][step] -> {name: grad}     (small parameters only, on the CPU)
        PR = []     # [run][step] -> {name: param BEFORE the step}  (small parameters)
        full = {}   # run -> [step] -> (params before the step (all), tid)
        final = []
        for r in range(runs):
            reset()
            g_run, p_run, f_run = [], [], []
            for k in range(K):
                before = {n: getattr(tr, n).detach().clone() for n in NAMES}
                if mode == "eager":
                    ns.forward()
                    ns.backward(1)
                    torch.cuda.synchronize()
                    g = snapshot(ns)
                    opt.step()
                else:
                    st()
                    torch.cuda.synchronize()
                    g = snapshot(ns)          # (the arena is cleared by the NEXT step's forward branch: still intact here)
                tid = (ns.rast[..., 3].long() - 1).cpu()
                g_run.append({n: g[n].cpu().double().reshape(-1) for n in SMALL if n in g})
                p_run.append({n: before[n].cpu().double().reshape(-1) for n in SMALL})
                f_run.append(({n: v.cpu() for n, v in before.items()}, tid, {n: v.cpu().double() for n, v in g.items()}))
            torch.cuda.synchronize()
            G.append(g_run)
            PR.append(p_run)
            final.append({n: getattr(tr, n).detach().cpu().double().reshape(-1) for n in NAMES})
            full[r] = f_run
            # keep full snapshots only for run 0 and for runs whose lights left run 0's
            if r > 0:
                d = float((final[r]["lights"] - final[0]["lights"]).abs().max())
                if d < 0.2 * cfg.lr.light * lr_scale:
                    del full[r]
        # ---- 2. gradient spread at steps where the runs still share their parameters bit for bit ----
        lines.append(f"== {mode}: spread of the step-k gradient over runs with bit-identical step-k parameters (max over runs of |g - g_run0|_inf / |g_run0|_inf)")
        for k in range(K):
            same = [r for r in range(runs) if all(torch.equal(PR[r][k][n], PR[0][k][n]) for n in SMALL)]
            if len(same) < 2:
                lines.append(f"step {k}: {len(same)} run(s) still bit-identical to run 0 -- trajectories have split (summation-order noise through Adam)")
                continue
            row = []
            for n in SMALL:
                if n not in G[0][k]:
                    continue
                ref = G[0][k][n]
                mx = max(float((G[r][k][n] - ref).abs().max()) for r in same[1:])
                row.append(f"{n} {mx / max(float(ref.abs().max()), 1e-300):.1e}")
            lines.append(f"step {k}: {len(same)} runs identical so far | " + "  ".join(row))
        # ---- 3. spread of the exported arrays, outliers ----
        lines.append(f"== {mode}: exported arrays after {K} steps, distance to run 0 (update L2 rel): median / max over runs")
        upd0 = {n: final[0][n] - start[n].cpu().double().reshape(-1) for n in NAMES}
        for n in NAMES:
            if float(upd0[n].norm()) == 0:
                continue
            d = [float(((final[r][n] - final[0][n])).norm() / upd0[n].norm()) for r in range(1, runs)]
            lines.append(f"{n}: median {np.median(d):.2e} max {np.max(d):.2e} (run {1 + int(np.argmax(d))})")
        outliers = [r for r in full if r > 0]
        lines.append(f"runs whose `lights` ended >= 0.2 lr away from run 0 in some component: {outliers if outliers else 'none'}")
        # an eager NativeStep to re-evaluate the gradient at recorded parameters (same kernels, issued launch by launch on streams + events)
        ns_chk = NativeStep(tr, sample, stage)
        lr_l = cfg.lr.light * lr_scale
        for r in ([0] + outliers)[:4]:
            lines.append(f"-- run {r}: per step, the gradient the run computed (cap) and an eager re-evaluation at the SAME parameters (eag), each vs the "
                         f"oracle at those parameters and visibility, max-norm relative; cap-vs-eag; pixels whose triangle id differs between the two; "
                         f"distance of the run's lights to run 0's before the step, in units of lr")
            for k in range(K):
                params, tid, g = full[r][k]
                go = oracle_grad(params, tid)
                with torch.no_grad():
                    for n in NAMES:
                        getattr(tr, n).copy_(params[n].cuda())
                ns_chk.forward()
                ns_chk.backward(1)
                torch.cuda.synchronize()
                ge = {n: ns_chk.g[n].detach().cpu().double() for n in NAMES if n in ns_chk.g}
                tid_e = (ns_chk.rast[..., 3].long() - 1).cpu()
                ndiff = int((tid_e != tid).sum())
                rel = lambda a, b: float((a.reshape(-1) - b.reshape(-1)).abs().max() / max(float(b.abs().max()), 1e-300))
                row = []
                for n in ("lights", "shape", "focal_length", "expr", "static_offset", "tex_extra"):
                    if n in g:
                        row.append(f"{n} cap {rel(g[n], go[n]):.1e} eag {rel(ge[n], go[n]):.1e} c-e {rel(g[n], ge[n]):.1e}")
                dl = float((PR[r][k]["lights"] - PR[0][k]["lights"]).abs().max()) / lr_l
                lines.append(f"   step {k}: " + " | ".join(row) + f" | tid diff {ndiff} px | lights vs run 0: {dl:.2f} lr | oracle max(diffuse) "
                             f"{go['_dmax']:.4f}, leading channel ahead of the second by {go['_gap']:.2e}")
                if r > 0 and k > 0:
                    # components of lights that this step moved the other way than run 0 did: their gradient in both runs and in the oracle
                    mv_r = PR[r][k]["lights"] - PR[r][k - 1]["lights"]
                    mv_0 = PR[0][k]["lights"] - PR[0][k - 1]["lights"]
                    for c in ((mv_r - mv_0).abs() > 0.5 * lr_l).nonzero().reshape(-1).tolist():
                        a, a0 = G[r][k - 1]["lights"], G[0][k - 1]["lights"]
                        gprev = oracle_grad(full[r][k - 1][0], full[r][k - 1][1])["lights"].reshape(-1)
                        lines.append(f"      step {k - 1} moved lights[{c}] by {float(mv_r[c]) / lr_l:+.2f} lr (run 0: {float(mv_0[c]) / lr_l:+.2f} lr): "
                                     f"g(run {r}) {float(a[c]):+.3e}  g(run 0) {float(a0[c]):+.3e}  g(oracle at run {r}) {float(gprev[c]):+.3e}  "
                                     f"|g|max {float(gprev.abs().max()):.3e}")
        del ns_chk
        del ns, ns_e, st
    reset()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"fit_flake_hunt_lights_{tag}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
