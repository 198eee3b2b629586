"""The step AS SHIPPED against the oracle at the BASELINE sizes (SURVEY 8(c) known-answer item 12 "oracle loss
vs HIP loss on cfg 1-4, disturbance off and with injected randomness"):

the hand-chained NativeStep with deferred shading, in-place antialiasing, the uv-binned texture gradient and the colour disturbance ON --
its random numbers INJECTED (NativeStep.injected: the same in-place pool kernels the captured step runs, fed the oracle's draws instead
of the in-kernel generator) -- every energy term and the gradient w.r.t. every parameter:

    config 2   2 x 512 x 512 monocular, T = 2048, rgb_global_tracking
    config 3   1 x 1024 x 1024 monocular, T = 2048, static_offset TRAINED (rgb_init_offset)
    config 4   2 calibrated views of one timestep, 802 x 550, T = 2048, the NeRSemble configuration

and K = 10 optimiser steps at 2 x 512 x 512, T = 2048 against the oracle's fit loop (fp64 + torch.optim.Adam).

Same visibility throughout (the oracle is handed the triangle ids the HIP rasteriser produced: the rasteriser itself is compared bit for
bit at these sizes in tests/test_raster_gpu.py).  Stated tolerances, fp32 product vs fp64 oracle: energy terms 5e-6 relative (measured
<= 1.3e-6); gradients: PER PARAMETER against the float32 oracle's own distance from the float64 oracle on the same batch (see
_native_vs_oracle), and as absolute ceilings, as a fraction of their max-norm and with cosine >= 0.999999: 5e-4 at 2 x 512^2 (measured <= 1.8e-4), 3e-3 at
1024^2 and at 802 x 550 (measured <= 7.2e-4, one array -- static_offset at 1024^2 -- 2.7e-3).  The yardstick is what fp32 arithmetic
itself costs on this energy: the ORACLE evaluated in float32 instead of float64 (same triangle ids, torch-CPU) lands 1e-4 .. 9e-4 from
its own float64 gradient on the geometry parameters at all three sizes, 2.3e-3 on `lights` at 802 x 550
(tools/grad_fp32_spread.py -> profiles/r03_oracle_fp32_vs_fp64_gradient_cfg{2,3,4}.txt) -- the pixel-sized triangles of a 5 000-vertex
head cancel heavily in the barycentric derivatives, and every pixel-level kink of the energy (L1 sign, antialiasing pair membership)
that float32 and float64 resolve differently moves one vertex's gradient by that pixel's whole contribution."""
import numpy as np
import pytest
import torch

from oracle import energy_ref
from tests.test_fit_parity_gpu import NAMES, _compare_grads, _make, _record

pytestmark = pytest.mark.gpu

T = 2048


TERM_BOUND = 5e-6        # energy terms, relative (measured <= 1.3e-6 at every size)
# The gradient gate, MEASURED on the batch under test.  SURVEY 8(c) hoped for 1e-4 "to be confirmed by the fp32-vs-fp64
# oracle spread": the oracle evaluated in float32 (same triangle ids, same disturbance draws, same side of the L1 kinks) is itself
# 1e-4 .. 2e-3 from its float64 self on the geometry parameters at these sizes, so 1e-4 is not a property any fp32 implementation has.
# What IS asserted: no parameter of the HIP step stands further from the float64 oracle than SPREAD_FACTOR times the WORST float32-oracle
# distance among the parameters of its kind on this batch, and the step is not systematically worse than float32 arithmetic (median over
# the parameters of HIP's distance / the float32 oracle's <= SPREAD_MEDIAN).  Why pooled per kind and not parameter by parameter: each
# distance is one draw of rounding noise -- the ratio of two such draws scatters by 4x either way (round 5, call 5: at 2 x 802 x 550
# HIP / fp32-oracle = 0.34 .. 4.5 across the parameters, median 0.69; with the disturbance off HIP is CLOSER than the fp32 oracle on every
# parameter: profiles/r05_call5_grad_spread_terms.txt).  Measured medians: 1.0 (config 2, 16 frames), 0.7 / 1.65 (config 4, 2 / 16 views).
GEOMETRY = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "static_offset", "dynamic_offset", "focal_length")
SPREAD_FACTOR = 3.0
SPREAD_MEDIAN = 3.0
SPREAD_FLOOR = 2e-5      # where float32 torch-CPU happens to land closer than that (tex_extra, lights: 1e-6 .. 2e-5)
# ... and, per parameter, its OWN float32-oracle distance (round-5 review, weak 2: pooled alone, `shape` could get 40 x worse at config 3
# unnoticed).  The ratio of two draws of rounding noise has been seen at 4.5 (above), so the ceiling sits at 8 x, not at the 5 x the review
# suggested: a gate that trips on noise once in a dozen runs teaches people to re-run it.
OWN_FACTOR = 8.0
OWN_FLOOR = 2e-5
KINK_RESIDUAL = 2e-5     # an L1 residual that changes sign between the two evaluations must be zero to fp32-vs-fp64 rounding of the colour


def _native_vs_oracle(tr, cfg, topo, tm, base_tex, sample, o_sample, stage, image_size, names, lines, tag, seed, grad_bound=5e-4, spread=True):
    """-> list of failures.  NativeStep (as the captured step runs it) with injected disturbance vs energy_ref.total_energy.
    `spread`: the gradient gate measured against the float32 oracle on this very batch (see SPREAD_FACTOR above); `grad_bound` stays as
    an absolute ceiling on top of it."""
    from vhap_amd.step import NativeStep
    H, W = image_size
    B = sample["rgb"].shape[0]
    tr.get_train_parameters(stage)
    uvmask = tr._uvmask_res().cpu().double()
    cr = cfg.render
    tr.render.disturb_rate_fg, tr.render.disturb_rate_bg = cr.disturb_rate_fg, cr.disturb_rate_bg
    dist = tr.render.make_disturbance((B, H, W), "cuda", generator=torch.Generator("cuda").manual_seed(seed))
    ns = NativeStep(tr, sample, stage)
    assert ns.deferred and ns.aa_inplace and ns.disturb_on and ns.photometric
    ns.injected = dist
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    keep = ns.keep.clone()
    assert 0.2 < float(1 - keep.mean()) < 0.8, "the disturbance must have replaced about half of the pixels"
    tid = (ns.rast[..., 3].long() - 1).cpu()
    cov = float((tid >= 0).float().mean())
    assert 0.03 < cov < 0.97, cov
    # the L1 term's kinks: the residual pred - gt of the HIP evaluation (image space).  A handful of the 4-8 M pixel channels of a full
    # batch have |residual| ~ 1e-7 and round to opposite signs in float32 and float64; each flips its pixel's WHOLE contribution to the few
    # texels it samples (d tex_extra: 4.8e-4 of the max-norm at 16 x 512^2 from TWO such pixels, 4.5e-6 once the oracle takes the same
    # side; tools/diag_texgrad.py -> profiles/r04_texgrad_kink_diagnosis_cfg2.txt).  Both signs are subgradients of |x| at 0: the oracle is
    # handed the HIP residual's signs -- the same decomposition as the triangle ids (energy unchanged to 1e-13)
    res_hip = (ns.rgba_aa[..., :3].detach().flip(1) - sample["rgb"].permute(0, 2, 3, 1)).cpu()
    log_n = {k: float(v) for k, v in ns.log_dict().items()}
    g_n = {k: ns.g[k].detach().clone().reshape(getattr(tr, k).shape) for k in names if k in ns.g}
    P = {k: getattr(tr, k).detach().cpu().double().requires_grad_() for k in names}
    ncl = int(topo.fid2cid.max()) + 1
    o_dist = dict(w_fg=dist["w_fg"].cpu(), w_bg=dist["w_bg"].cpu(), idx=[dist["idx"].cpu()] * ncl,
                  fid2cid=torch.from_numpy(topo.fid2cid.astype(np.int64)))
    Eo, logo, ex = energy_ref.total_energy(P, tm, topo, cfg, o_sample, stage, base_tex, uvmask, (H, W), disturb=o_dist, tid=tid,
                                           photo_sign_from=res_hip)
    Eo.backward()
    res_ora = ex["rgba"][..., :3].detach() - sample["rgb"].permute(0, 2, 3, 1).cpu().double()
    n_kink = int((torch.sign(res_ora) != torch.sign(res_hip.double())).sum())
    fails = []
    lines.append(f"{tag}: {B} x {H}x{W}, T = {T}, stage {stage}, coverage {cov:.3f}, disturbed {float(1 - keep.mean()):.3f}, "
                 f"L1 residuals on opposite sides of zero in the two evaluations: {n_kink} of {res_hip.numel()}")
    # observed: 0 - 5 per batch (round-4 advisor: 1e-5 * numel would have hidden a small systematic sign defect); every one of them must be a
    # residual that IS zero to rounding in the float64 evaluation
    flip = torch.sign(res_ora) != torch.sign(res_hip.double())
    assert n_kink <= 16, n_kink
    assert n_kink == 0 or float(res_ora[flip].abs().max()) < KINK_RESIDUAL, float(res_ora[flip].abs().max())
    for k, b in logo.items():
        b = float(b.detach())
        e = abs(log_n[k] - b) / max(abs(b), 1e-3)
        lines.append(f"{tag} term {k}: {e:.2e}")
        if e > TERM_BOUND:
            fails.append(f"{tag} term {k}: {log_n[k]} vs {b}")
    e = abs(log_n["total"] - float(Eo.detach())) / abs(float(Eo.detach()))
    lines.append(f"{tag} total: {e:.2e}")
    if e > TERM_BOUND:
        fails.append(f"{tag} total energy: {log_n['total']} vs {float(Eo.detach())}")
    worst = _compare_grads(P, g_n, lines, tag, grad_bound, 0.999999, fails)
    if worst > grad_bound:
        fails.append(f"{tag} gradients: worst rel {worst:.2e} (bound {grad_bound:.0e})")
    if spread:
        P32 = {k: getattr(tr, k).detach().cpu().float().requires_grad_() for k in names}
        tm32 = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in tm.items()}
        E32, _, _ = energy_ref.total_energy(P32, tm32, topo, cfg, o_sample, stage, base_tex.float(), uvmask.float(), (H, W), dtype=torch.float32,
                                            disturb=o_dist, tid=tid, photo_sign_from=res_hip)
        E32.backward()
        e_hip, e_32 = {}, {}
        for k in names:
            g64 = P[k].grad
            if g64 is None or float(g64.abs().max()) == 0 or k not in g_n or P32[k].grad is None:
                continue
            nrm = float(g64.abs().max())
            e_hip[k] = float((g_n[k].detach().cpu().double().reshape(-1) - g64.reshape(-1)).abs().max()) / nrm
            e_32[k] = float((P32[k].grad.double().reshape(-1) - g64.reshape(-1)).abs().max()) / nrm
        pooled = {True: max([e_32[k] for k in e_32 if k in GEOMETRY] or [0.0]), False: max([e_32[k] for k in e_32 if k not in GEOMETRY] or [0.0])}
        ratios = []
        for k in e_hip:
            allowed = max(SPREAD_FACTOR * pooled[k in GEOMETRY], SPREAD_FLOOR)
            ratios.append(e_hip[k] / max(e_32[k], 1e-30))
            lines.append(f"{tag} spread {k}: HIP {e_hip[k]:.2e}  oracle-fp32 {e_32[k]:.2e}  ratio {ratios[-1]:.2f}  allowed {allowed:.2e}  "
                         f"{'ok' if e_hip[k] <= allowed else 'OVER'}")
            if e_hip[k] > allowed:
                fails.append(f"{tag} grad {k}: HIP is {e_hip[k]:.2e} from the float64 oracle; the float32 oracle's worst parameter of this kind "
                             f"{pooled[k in GEOMETRY]:.2e} (x{SPREAD_FACTOR} = {allowed:.2e})")
            # ... and against the float32 oracle's distance on THIS parameter (the pooled bound alone would let one parameter get 40 x
            # worse unnoticed where another of its kind is noisy: round-5 review, weak 2)
            own = max(OWN_FACTOR * e_32[k], OWN_FLOOR)
            if e_hip[k] > own:
                fails.append(f"{tag} grad {k}: HIP is {e_hip[k]:.2e} from the float64 oracle, the float32 oracle {e_32[k]:.2e} on the same "
                             f"parameter (x{OWN_FACTOR}, floor {OWN_FLOOR:g} = {own:.2e})")
        med = float(np.median(ratios)) if ratios else 0.0
        lines.append(f"{tag} spread: median over the parameters of HIP / oracle-fp32 = {med:.2f} (bound {SPREAD_MEDIAN})")
        if med > SPREAD_MEDIAN:
            fails.append(f"{tag}: HIP is systematically further from the float64 oracle than float32 arithmetic: median ratio {med:.2f}")
    assert float(P["tex_extra"].grad.abs().max()) > 0
    tr.render.disturb_rate_fg, tr.render.disturb_rate_bg = cr.disturb_rate_fg, cr.disturb_rate_bg
    return fails


def run_config2(flame_model, B, record, spread=True):
    """BASELINE config 2 at batch B (tests/test_parity_fullbatch_gpu.py runs B = 16, the batch BASELINE names)."""
    H = W = 512
    S = _make(flame_model, H, W, B, T, seed=17)
    tr = S["tr"]
    ts = np.arange(B)
    sample = tr.get_sample(ts, device_index=True)
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts}
    lines = []
    fails = _native_vs_oracle(tr, S["cfg"], S["topo"], S["tm"], S["base_tex"], sample, o_sample, "rgb_global_tracking", (H, W), NAMES, lines,
                              "cfg2", seed=12, spread=spread)
    _record(record, lines + fails)
    assert not fails, fails


def test_shipped_native_step_with_injected_disturbance_config2_size(flame_model):
    run_config2(flame_model, 2, "parity_native_injected_cfg2.txt")


def run_config3(flame_model, B, record, spread=True):
    """BASELINE config 3: 1024 x 1024 with static_offset trained (stage rgb_init_offset: the offset regularisers, the full learning-rate
    stage) at batch B (the oracle needs ~20 s per megapixel)."""
    H = W = 1024
    S = _make(flame_model, H, W, B, T, seed=29)
    tr = S["tr"]
    ts = np.arange(B)
    sample = tr.get_sample(ts, device_index=True)
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts}
    lines = []
    fails = _native_vs_oracle(tr, S["cfg"], S["topo"], S["tm"], S["base_tex"], sample, o_sample, "rgb_init_offset", (H, W), NAMES, lines,
                              "cfg3", seed=5, grad_bound=3e-3, spread=spread)
    _record(record, lines + fails)
    assert not fails, fails


def test_shipped_native_step_config3_size_static_offset_trained(flame_model):
    run_config3(flame_model, 1, "parity_native_injected_cfg3.txt")


def run_config4(flame_model, NV, record, spread=True):
    """BASELINE config 4: calibrated views (K [B,3,3], RT [B,3,4]) of ONE timestep at 802 x 550 under the NeRSemble configuration
    (w.landmark 3, reg_tex_tv 1e5, jawline landmarks off): NV views on the arc of vhap_amd.synthetic.arc_cameras."""
    from vhap_amd.config import nersemble_config
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_multiview_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    H, W = 802, 550
    cfg = nersemble_config()
    cfg.model.tex_resolution = T
    gt = make_scene_params(1, seed=3, image_size=(H, W))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_multiview_dataset(rend, head, gt, (H, W), "cuda", n_views=NV, seed=3, tex=make_texture(3, T))
    base_tex = make_texture(0, T)
    tr = GlobalTracker(cfg, model, topo, base_tex, data)
    assert tr.calibrated
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for name, s_ in (("shape", 0.3), ("expr", 0.3), ("rotation", 0.05), ("neck_pose", 0.03), ("jaw_pose", 0.05), ("eyes_pose", 0.05),
                         ("translation", 0.005), ("tex_extra", 0.03), ("lights", 0.05), ("static_offset", 1e-3)):
            p = getattr(tr, name)
            p.add_((torch.randn(p.shape, generator=g) * s_).cuda())
        tr.jaw_pose[:, 0] += 0.1
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].double()
    sample = tr.get_sample(np.array([0]), device_index=True)
    assert sample["rgb"].shape[0] == NV
    ts = sample["timestep_index"].cpu().numpy()
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts, "intrinsic": sample["intrinsic"].cpu(),
                "extrinsic": sample["extrinsic"].cpu()}
    names = [n for n in NAMES if n != "focal_length"]
    lines = []
    fails = _native_vs_oracle(tr, cfg, topo, tm, torch.from_numpy(base_tex)[None].double(), sample, o_sample, "rgb_global_tracking", (H, W),
                              names, lines, "cfg4", seed=21, grad_bound=3e-3, spread=spread)
    _record(record, lines + fails)
    assert not fails, fails


def test_shipped_native_step_config4_size_calibrated_views(flame_model):
    run_config4(flame_model, 2, "parity_native_injected_cfg4.txt")


def test_ten_steps_at_baseline_size_match_oracle_fit(flame_model):
    """K = 10 optimiser steps (tracker.py:1418-1462) at 2 x 512 x 512, T = 2048, the step as captured (disturbance off: its in-kernel
    draws cannot be replayed; the injected form is covered above), against the oracle's fit loop; every exported array
    (tracker.py:1152-1218) to SURVEY 8(c)'s 1e-3 in relative L2, energies along the trajectory to 5e-6."""
    from tests.test_fit_parity_gpu import _trajectory
    H = W = 512
    S = _make(flame_model, H, W, 2, T, seed=17)
    stage, lr_scale, K = "rgb_global_tracking", 0.1, 10
    start, hip, ora, (E_hip, E_ora, dmax) = _trajectory(S, stage, lr_scale, K, H, W, np.array([0, 1]), same_visibility=True)
    lines = [f"stage {stage} lr_scale {lr_scale} K {K} 2 x {H}x{W} T {T} same visibility; oracle max(diffuse) / gap to the second channel per step: "
             + " ".join(f"{d:.3f}/{g:.3f}" for d, g in dmax)]
    fails = []
    for i, (a, b) in enumerate(zip(E_hip, E_ora)):
        e = abs(a - b) / abs(b)
        lines.append(f"step {i}: E hip {a:.6f} oracle {b:.6f} rel {e:.2e}")
        if e > TERM_BOUND:
            fails.append(f"energy at step {i}: {a} vs {b}")
    assert E_hip[-1] < E_hip[0] and E_ora[-1] < E_ora[0]
    for k in sorted(hip):
        a, b = np.asarray(hip[k], np.float64), np.asarray(ora[k], np.float64)
        if k in ("timestep_id", "n_processed_frames", "image_size"):
            assert np.array_equal(a, b), k
            continue
        moved = float(np.abs(b - start[k]).max())
        if moved == 0:
            continue
        l2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        mx = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
        dl2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b - start[k]), 1e-300))
        lines.append(f"{k}: L2 rel {l2:.2e}  max-norm rel {mx:.2e}  update L2 rel {dl2:.2e}")
        if l2 > 1e-3:
            fails.append(f"{k}: L2 rel {l2:.2e}")
    _record("fit_parity_10_steps_512_T2048.txt", lines + fails)
    assert not fails, fails


@pytest.mark.parametrize("stage", ["rgb_sequential_tracking", "rgb_global_tracking"])
def test_native_step_with_dynamic_offset_matches_oracle(flame_model, stage):
    """`use_dynamic_offset` (base.py:69; tracker.py:213-235, 552-600) on the NativeStep: one vertex-offset row per frame
    (static_offset + dynamic_offset[timesteps]) through skinning and joint regression, the offset regularisers per frame, the temporal term
    reg_offset_dynamic -- energy terms and gradients (incl. d dynamic_offset) against the oracle, whose dynamic-offset terms are pinned on
    the reference (tests/test_energy_golden.py::test_dynamic_offset_regularisers_match_reference).  Three frames and BOTH offsets: more than
    the reference can run (its in-place offset sum raises for B > 1), the same mathematics."""
    from vhap_amd.step import NativeStep
    H = W = 128
    Tt = 256
    S = _make(flame_model, H, W, 4, Tt, seed=41, dynamic_offset=True)
    tr, cfg, topo, tm = S["tr"], S["cfg"], S["topo"], S["tm"]
    assert tr.dynamic_offset is not None and NativeStep.supported(tr, stage)
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    tr.get_train_parameters(stage)
    ts = np.array([2, 0, 3])
    sample = tr.get_sample(ts, device_index=True)
    ns = NativeStep(tr, sample, stage)
    assert ns.dyn and ns.deferred
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    names = NAMES + ("dynamic_offset",)
    tid = (ns.rast[..., 3].long() - 1).cpu()
    P = {k: getattr(tr, k).detach().cpu().double().requires_grad_() for k in names}
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": ts}
    Eo, logo, _ = energy_ref.total_energy(P, tm, topo, cfg, o_sample, stage, S["base_tex"], tr._uvmask_res().cpu().double(), (H, W), tid=tid)
    Eo.backward()
    log_n = {k: float(v) for k, v in ns.log_dict().items()}
    lines, fails = [f"dynamic offset, {len(ts)} x {H}x{W}, T = {Tt}, stage {stage}"], []
    assert "reg_offset_dynamic" in logo and float(logo["reg_offset_dynamic"]) > 0
    for k, b in logo.items():
        b = float(b.detach())
        e = abs(log_n[k] - b) / max(abs(b), 1e-3)
        lines.append(f"term {k}: {e:.2e}")
        if e > TERM_BOUND:
            fails.append(f"term {k}: {log_n[k]} vs {b}")
    g_n = {k: ns.g[k].detach().clone().reshape(getattr(tr, k).shape) for k in names if k in ns.g}
    worst = _compare_grads(P, g_n, lines, "dyn", 5e-4, 0.999999, fails)
    if worst > 5e-4:
        fails.append(f"gradients: worst rel {worst:.2e}")
    # (timestep 1 is not in the batch, but it is the predecessor of timestep 2: the temporal term reaches its row, nothing else does)
    assert float(P["dynamic_offset"].grad.abs().max()) > 0 and float(g_n["dynamic_offset"][1].abs().max()) > 0
    # ... and the same step captured and replayed: the plan executor must cope with its per-frame launches
    from vhap_amd.tracker import GraphedStep
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=0.1)
    before = tr.dynamic_offset.detach().clone()
    st = GraphedStep(tr, sample, opt, stage, warmup=0)
    assert st.ns is not None and st.ns.dyn and st.gF.plan is not None
    E0 = float(st())
    for _ in range(4):
        E1 = float(st())
    torch.cuda.synchronize()
    assert abs(E0 - float(Eo.detach())) <= 1e-4 * abs(float(Eo.detach())) and E1 < E0
    if "dynamic_offset" in cfg.pipeline[stage].optimizable_params:
        moved = (tr.dynamic_offset.detach() - before).abs().amax(dim=(1, 2)).cpu().numpy()
        assert moved[2] > 0 and moved[0] > 0 and moved[3] > 0
    _record(f"parity_native_dynamic_offset_{stage}.txt", lines + fails)
    assert not fails, fails
