"""Parity of the FIT, not just of one energy evaluation (BASELINE north_star: "photometric loss and exported FLAME
parameters within a stated fp32 tolerance"):

  * energy terms and gradients against the oracle at the BASELINE texture size (512 x 512 frames, T = 2048: the uv-binned texture gradient
    with its 64 x 64 tiles, the 11-level pyramid fold and tex_prep_bwd's gathered levels are exercised for real), for BOTH product
    formulations -- the autograd one and the hand-chained NativeStep that the captured step replays;
  * K = 10 optimiser steps (tracker.py:1418-1462: energy, backward, Adam) against the oracle's energy + torch.optim.Adam in float64
    (oracle/fit_ref.py), every array of the exported npz (tracker.py:1152-1218) compared.

Stated tolerances (fp32 product vs fp64 oracle):
  same visibility (the oracle is handed the triangle ids the HIP rasteriser produced, so that the comparison is arithmetic only):
      energy terms 5e-5 relative, gradients 5e-4 of their max-norm (measured on MI355X: 1.7e-6 and 1.8e-4); after 10 steps every exported
      array to 1e-3 in relative L2 AND in max-norm -- except static_offset's max-norm, 5e-3: its entries are ~1e-3 in size
      and move by lr = 5e-5 per step, and Adam's g / (|g| + eps) makes the step of an entry whose gradient sits inside the fp32-atomics
      noise a full +-lr step of either sign (measured 1.8e-3 max-norm, 1.3e-4 L2) -- and the parameter UPDATE (export - start) to 2e-2 in
      L2 (measured <= 2e-4); the energies along the trajectory to 5e-4 (measured <= 1e-6).  At the FULL learning rates (rgb_init_offset,
      lr_scale 1: static_offset moves by more than its own size in 10 steps) the same effect is 10x larger on the two element-wise
      arrays -- static_offset: L2 1e-2 / max-norm 1e-1 (measured 4.6e-3 / 5.2e-2), tex_extra: max-norm 2e-2 (measured 6.9e-3); everything
      else stays below 1e-4.  For scale: the ORACLE ITSELF evaluated in fp32 instead of fp64 lands 7e-2 (L2) away from its fp64 run on
      static_offset and 1-2e-2 on expr / eyes_pose after the same 10 steps (tools/fit_fp32_spread.py, profiles/r02_fit_fp32_spread.txt)
      -- the HIP path is an order of magnitude closer to the fp64 oracle than fp32 arithmetic needs to be;
  independent visibility (the oracle rasterises its own fp64 vertices; a handful of border pixels resolve differently -- each moves the
  gradient of its triangle's vertices by that pixel's whole contribution, and Adam's g / (|g| + eps) turns a gradient whose sign is
  inside that noise into a full-size step of either sign, so element-wise agreement of noise-level entries is not a meaningful target):
  every exported array to 5e-3 in relative L2 (measured <= 1.0e-3), static_offset at the full learning rates -- it moves by its own
  size in these 10 steps -- to 5e-2 (measured 2.1e-2); energies along the trajectory to 5e-3 (measured <= 2.8e-4, 2e-7 while the two
  visibilities still agree).
  history: round 2 saw one run in five of the full-learning-rate trajectory end with `lights` 3.4e-2 (update L2) from the oracle instead
  of 4e-5, blamed summation-order noise amplified by Adam, and widened every bound to twice the oracle's own fp32-vs-fp64 spread.
  Round 3 hunted it down (tools/fit_flake_hunt.py, profiles/r03_fit_flake_hunt_*.txt: 40-60 runs x 2 executors, the gradient of every
  step against the oracle at each run's OWN parameters, eager re-evaluation of every suspicious step): no race -- the captured and the
  eagerly issued step give the same gradient at the same parameters, bit for bit on `lights` -- and no noise either, but a BUG: the
  backward re-computes every pixel's diffuse value and routes reg_diffuse's max-gradient (tracker.py:547-550: relu(diffuse.max() - 1))
  to the pixels whose value EQUALS the maximum the forward recorded; the forward kernel and the backward kernel contracted
  different products of that arithmetic into fmas, so at about one parameter state in five the re-computed maximum missed the recorded
  one by an ulp and the whole max term (up to 97 % of the gradient's max-norm on the affected rows of `lights`) silently vanished for
  that step.  The shading arithmetic is now explicit round-to-nearest intrinsics in every kernel (csrc/shade_common.h); the escape
  hatch is gone: every array is held to the tight bounds above.
The measured values are written to gpurun_out/fit_parity_*.txt for the record."""
import os

import numpy as np
import pytest
import torch

from oracle import energy_ref, fit_ref

pytestmark = pytest.mark.gpu

NAMES = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset",
         "focal_length")

LIGHTS_SCALE = 1.0       # (a scalar, or one factor per colour channel: tools/fit_flake_hunt.py varies it)


def _record(name, lines):
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", name), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


def _make(flame_model, H, W, N, T, seed, lights_scale=1.0, dynamic_offset=False):
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    cfg = BaseTrackingConfig()
    cfg.model.tex_resolution = T
    cfg.model.use_dynamic_offset = bool(dynamic_offset)
    gt = make_scene_params(N, seed=seed, image_size=(H, W))
    head = FlameHead(model, topo).cuda()
    rend = HipDiffRenderer(lighting_type="SH").cuda()
    data = make_dataset(rend, head, gt, (H, W), "cuda", seed=seed, tex=make_texture(seed, T))
    base_tex = make_texture(0, T)
    tr = GlobalTracker(cfg, model, topo, base_tex, data)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():      # a perturbed state: every term has a non-trivial gradient, no parameter sits at exactly 0
        for name, s in (("shape", 0.3), ("expr", 0.3), ("rotation", 0.1), ("neck_pose", 0.03), ("jaw_pose", 0.05), ("eyes_pose", 0.05),
                        ("translation", 0.01), ("tex_extra", 0.03), ("lights", 0.05), ("static_offset", 1e-3)):
            p = getattr(tr, name)
            p.add_((torch.randn(p.shape, generator=g) * s).cuda())
        tr.translation[:, 2] += 0.45
        tr.jaw_pose[:, 0] += 0.1
        tr.lights.mul_(torch.as_tensor(lights_scale, dtype=torch.float32, device=tr.lights.device))     # scalar, or one factor per colour channel
        if dynamic_offset:
            tr.dynamic_offset.add_((torch.randn(tr.dynamic_offset.shape, generator=g) * 5e-4).cuda())
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].double()
    return dict(tr=tr, cfg=cfg, model=model, topo=topo, tm=tm, base_tex=torch.from_numpy(base_tex)[None].double())


def _oracle_params(tr):
    return {k: getattr(tr, k).detach().cpu().double().requires_grad_() for k in NAMES}


def _compare_grads(P, grads, lines, tag, rel_bound, cos_bound, fails=None):
    worst = 0.0
    for k, po in P.items():
        gp = grads.get(k)
        if po.grad is None or float(po.grad.abs().max()) == 0 or gp is None:
            continue
        a, b = gp.detach().cpu().double().reshape(-1), po.grad.reshape(-1)
        rel = float((a - b).abs().max() / b.abs().max())
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        worst = max(worst, rel)
        lines.append(f"{tag} grad {k}: rel {rel:.2e} cos {cos:.7f}")
        if fails is not None and cos < cos_bound:
            fails.append(f"{tag} grad {k}: cosine {cos:.7f}")
    return worst


def test_energy_and_gradients_match_oracle_at_baseline_texture_size(flame_model):
    from vhap_amd.step import NativeStep
    H = W = 512
    T = 2048
    S = _make(flame_model, H, W, 2, T, seed=17)
    tr, cfg, topo, tm = S["tr"], S["cfg"], S["topo"], S["tm"]
    stage = "rgb_global_tracking"
    tr.get_train_parameters(stage)
    ts = np.array([0, 1])
    uvmask = tr._uvmask_res().cpu().double()
    lines = [f"{H}x{W}, T = {T}, B = 2, stage {stage}, same visibility"]
    fails = []

    def oracle(tid, dist):
        P = _oracle_params(tr)
        o_dist = None
        if dist is not None:
            ncl = int(topo.fid2cid.max()) + 1
            o_dist = dict(w_fg=dist["w_fg"].cpu(), w_bg=dist["w_bg"].cpu(), idx=[dist["idx"].cpu()] * ncl,
                          fid2cid=torch.from_numpy(topo.fid2cid.astype(np.int64)))
        o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts}
        Eo, logo, _ = energy_ref.total_energy(P, tm, topo, cfg, o_sample, stage, S["base_tex"], uvmask, (H, W), disturb=o_dist, tid=tid)
        Eo.backward()
        return P, {k: float(v.detach()) for k, v in logo.items()}

    # ---- 1. the hand-chained NativeStep (what the captured step replays), disturbance off (its random numbers are drawn in-kernel)
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    sample = tr.get_sample(ts, device_index=True)
    ns = NativeStep(tr, sample, stage)
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    tid = (ns.rast[..., 3].long() - 1).cpu()
    cov = float((tid >= 0).float().mean())
    assert 0.05 < cov < 0.95
    log_n = {k: float(v) for k, v in ns.log_dict().items()}
    g_n = {k: ns.g[k].detach().clone().reshape(getattr(tr, k).shape) for k in NAMES}
    P, logo = oracle(tid, None)
    for k, b in logo.items():
        e = abs(log_n[k] - b) / max(abs(b), 1e-3)
        lines.append(f"native term {k}: {e:.2e}")
        if e > 5e-5:
            fails.append(f"native term {k}: {log_n[k]} vs {b}")
    worst = _compare_grads(P, g_n, lines, "native", 5e-4, 0.999999)
    if worst > 5e-4:
        fails.append(f"native gradients: worst rel {worst:.2e}")
    # the texture gradient must have been exercised at every level of the pyramid
    assert float(P["tex_extra"].grad.abs().max()) > 0

    # ---- 2. the autograd formulation, with the colour disturbance on (injected randomness, replayed by the oracle)
    cfg_r = cfg.render
    tr.render.disturb_rate_fg, tr.render.disturb_rate_bg = cfg_r.disturb_rate_fg, cfg_r.disturb_rate_bg
    dist = tr.render.make_disturbance((2, H, W), "cuda", generator=torch.Generator("cuda").manual_seed(12))
    s = dict(sample)
    tr.fill_cam_params_into_sample(s)
    for k in NAMES:
        getattr(tr, k).grad = None
    E, log, *_ = tr.compute_energy(s, stage=stage, disturbance=dist)
    E.backward()
    P, logo = oracle(tid, dist)
    for k, b in logo.items():
        e = abs(float(log[k].detach()) - b) / max(abs(b), 1e-3)
        lines.append(f"autograd term {k}: {e:.2e}")
        if e > 5e-5:
            fails.append(f"autograd term {k}: {float(log[k])} vs {b}")
    worst = _compare_grads(P, {k: getattr(tr, k).grad for k in NAMES}, lines, "autograd", 5e-4, 0.999999)
    if worst > 5e-4:
        fails.append(f"autograd gradients: worst rel {worst:.2e}")
    _record("fit_parity_fullsize_energy.txt", lines + fails)
    assert not fails, fails


def _trajectory(S, stage, lr_scale, K, H, W, ts, same_visibility):
    """Run K steps on the GPU and K oracle steps from the same start; returns (start, export_hip, export_oracle, energies)."""
    from vhap_amd.step import NativeStep
    from vhap_amd.tracker import GraphedStep
    tr, cfg, topo, tm = S["tr"], S["cfg"], S["topo"], S["tm"]
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    start = {k: getattr(tr, k).detach().clone() for k in NAMES}
    P = {k: start[k].cpu().double().requires_grad_() for k in NAMES}
    sample = tr.get_sample(ts, device_index=True)
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts}
    uvmask = tr._uvmask_res().cpu().double()
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=lr_scale)
    opt_o = fit_ref.configure_optimizer(P, cfg, stage, lr_scale=lr_scale)
    assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in opt_o.param_groups]
    E_hip, E_ora, dmax = [], [], []          # dmax: the oracle's (max(diffuse), gap between the two leading channels' maxima) per step
    if same_visibility:
        # the same call sequence as the captured step, issued eagerly so that each step's triangle ids can be handed to the oracle
        ns = NativeStep(tr, sample, stage)
        for _ in range(K):
            ns.forward()
            ns.backward(1)
            tid = (ns.rast[..., 3].long() - 1).cpu()
            E_hip.append(float(ns.log[15]))
            opt.step()
            o = fit_ref.optimize_iter(P, opt_o, tm, topo, cfg, o_sample, stage, S["base_tex"], uvmask, (H, W), tid=tid)
            E_ora.append(o["total"])
            dmax.append((o.get("diffuse_max"), o.get("diffuse_channel_gap")))
    else:
        st = GraphedStep(tr, sample, opt, stage, warmup=0)
        assert st.ns is not None, "the captured step must be the native call sequence"
        for _ in range(K):
            E_hip.append(float(st()))
            o = fit_ref.optimize_iter(P, opt_o, tm, topo, cfg, o_sample, stage, S["base_tex"], uvmask, (H, W))
            E_ora.append(o["total"])
            dmax.append((o.get("diffuse_max"), o.get("diffuse_channel_gap")))
    torch.cuda.synchronize()
    exp_hip = tr.save_result()
    exp_ora = fit_ref.export(P, (H, W))
    with torch.no_grad():                                     # restore: the fixture is shared
        for k in NAMES:
            getattr(tr, k).copy_(start[k])
    return {k: v.cpu().numpy() for k, v in start.items()}, exp_hip, exp_ora, (E_hip, E_ora, dmax)


@pytest.fixture(scope="module")
def small(flame_model):
    return _make(flame_model, 128, 128, 3, 256, seed=23, lights_scale=LIGHTS_SCALE)


@pytest.fixture(scope="module")
def small_tinted(flame_model):
    """Lights with distinct colour channels (x 1.3 / 1.15 / 1.0): the white-light scene reaches max(diffuse) in all three channels at once;
    here the maximum is one element, and a backward that re-computes it one ulp off loses the whole term (csrc/shade_common.h)."""
    return _make(flame_model, 128, 128, 3, 256, seed=23, lights_scale=(1.3, 1.15, 1.0))


def test_ten_steps_with_tinted_lights_match_oracle_fit(small_tinted):
    K, H, W, stage = 10, 128, 128, "rgb_init_offset"
    start, hip, ora, (E_hip, E_ora, dmax) = _trajectory(small_tinted, stage, 1.0, K, H, W, np.array([1, 2]), True)
    lines = ["tinted lights, same visibility; oracle max(diffuse) / gap per step: " + " ".join(f"{d:.3f}/{g:.3f}" for d, g in dmax)]
    assert min(d for d, _ in dmax) > 1.0 and min(g for _, g in dmax) > 1e-3
    fails = []
    for i, (a, b) in enumerate(zip(E_hip, E_ora)):
        e = abs(a - b) / abs(b)
        lines.append(f"step {i}: E hip {a:.6f} oracle {b:.6f} rel {e:.2e}")
        if e > 5e-4:
            fails.append(f"energy at step {i}: {a} vs {b}")
    for k in ("lights", "shape", "expr", "static_offset", "tex_extra", "focal_length"):
        a, b, s0 = np.asarray(hip[k], np.float64), np.asarray(ora[k], np.float64), np.asarray(start[k], np.float64)
        e = np.linalg.norm((a - b).ravel()) / max(np.linalg.norm((b - s0).ravel()), 1e-12)
        lines.append(f"{k}: update L2 rel {e:.2e}")
        if e > (2e-2 if k == "static_offset" else 5e-3):
            fails.append(f"{k}: {e:.2e}")
    _record("fit_parity_tinted_lights.txt", lines + fails)
    assert not fails, fails


@pytest.mark.parametrize("stage,lr_scale", [("rgb_global_tracking", 0.1), ("rgb_init_offset", 1.0)])
@pytest.mark.parametrize("same_visibility", [True, False])
def test_ten_steps_export_matches_oracle_fit(small, stage, lr_scale, same_visibility):
    K, H, W = 10, 128, 128
    ts = np.array([0, 1, 2]) if stage == "rgb_global_tracking" else np.array([1, 2])
    start, hip, ora, (E_hip, E_ora, dmax) = _trajectory(small, stage, lr_scale, K, H, W, ts, same_visibility)
    assert set(hip) == set(ora), (sorted(hip), sorted(ora))              # same npz schema (tracker.py:1158-1218)
    lines = [f"stage {stage} lr_scale {lr_scale} K {K} same_visibility {same_visibility}; oracle max(diffuse) / gap to the second channel per step: "
             + " ".join(f"{d:.3f}/{g:.3f}" for d, g in dmax)]
    assert max(d for d, _ in dmax) > 1.0, "reg_diffuse's max term must be active (its gradient path is what this guards)"
    fails = []
    e_bound = 5e-4 if same_visibility else 5e-3          # (measured <= 1e-6 / 2.2e-5)
    for i, (a, b) in enumerate(zip(E_hip, E_ora)):
        e = abs(a - b) / abs(b)
        lines.append(f"step {i}: E hip {a:.6f} oracle {b:.6f} rel {e:.2e}")
        if e > e_bound:
            fails.append(f"energy at step {i}: {a} vs {b}")
    assert E_hip[-1] < E_hip[0] and E_ora[-1] < E_ora[0]
    for k in sorted(hip):
        a, b = np.asarray(hip[k], np.float64), np.asarray(ora[k], np.float64)
        assert a.shape == b.shape, k
        if k in ("timestep_id", "n_processed_frames", "image_size"):
            assert np.array_equal(a, b), k
            continue
        moved = float(np.abs(b - start[k]).max())
        mx = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
        l2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        dl2 = float(np.linalg.norm((a - start[k]) - (b - start[k])) / max(np.linalg.norm(b - start[k]), 1e-300)) if moved > 0 else 0.0
        lines.append(f"{k}: max-norm rel {mx:.2e}  L2 rel {l2:.2e}  update L2 rel {dl2:.2e}  (oracle moved by {moved:.2e})")
        if moved == 0:
            assert np.array_equal(a, start[k].astype(np.float64)), f"{k} must not move in {stage}"
            continue
        assert float(np.abs(a - start[k]).max()) > 0, f"{k} did not move"
        if same_visibility:
            full_lr = lr_scale >= 1.0
            mx_b = {"static_offset": 1e-1 if full_lr else 5e-3, "tex_extra": 2e-2 if full_lr else 1e-3}.get(k, 1e-3)
            l2_b = 1e-2 if (full_lr and k == "static_offset") else 1e-3
            if mx > mx_b or l2 > l2_b or dl2 > 2e-2:
                fails.append(f"{k}: max-norm rel {mx:.2e}, L2 rel {l2:.2e}, update L2 rel {dl2:.2e}")
        elif l2 > (5e-2 if (lr_scale >= 1.0 and k == "static_offset") else 5e-3):
            fails.append(f"{k}: L2 rel {l2:.2e}")
    _record(f"fit_parity_{stage}_{'same' if same_visibility else 'indep'}_visibility.txt", lines + fails)
    assert not fails, fails
