"""The FLAME PCA texture model (`tex_painted = False`; vhap/model/flame.py:665-688, tracker.py:57-60, 241-244, 519-521, 1312-1313, 1485-1487):
  * vhap_amd.flame.FlameTexPCA and the oracle's restatement against vectors the REFERENCE's own class produced on a synthetic texture space
    of FLAME_texture.npz's layout (tools/make_golden_texpca.py -> tests/golden/texpca_golden.npz): nearest resize up AND down, channel order,
    the clamp;
  * the tracker's host formulation with the PCA base texture against the oracle's total energy incl. d(tex_pca)  (CPU);
  * the HIP kernels (vhap_tex_pca_fwd / _bwd) and the NativeStep with the PCA model against the oracle  (GPU)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "texpca_golden.npz")
CASES = ((100, 512), (100, 1024), (50, 256), (100, 384))


def _space():
    from vhap_amd.synthetic import make_tex_space
    return make_tex_space(0, 512, 200)


@pytest.mark.parametrize("n_tex,T", CASES)
def test_tex_pca_model_and_oracle_match_reference_golden(n_tex, T):
    from oracle import torch_ref as R
    from vhap_amd.flame import FlameTexPCA
    G = np.load(GOLD)
    sp = _space()
    code = torch.from_numpy(G[f"code_{n_tex}_{T}"])
    idx = torch.from_numpy(G[f"idx_{n_tex}_{T}"])
    tex = FlameTexPCA(n_tex, tex_size=T, tex_space=sp)(code)
    assert tex.shape == (2, 3, T, T)
    assert np.allclose(tex[:, :, idx[:, 0], idx[:, 1]].numpy(), G[f"val_{n_tex}_{T}"], rtol=0, atol=2e-6)
    assert np.allclose(tex.double().sum(dim=(2, 3)).numpy(), G[f"sum_{n_tex}_{T}"], rtol=1e-6)
    assert abs(float((tex == 0).float().mean()) - G[f"clamped_{n_tex}_{T}"][0]) < 1e-5 and float((tex == 1).float().mean()) > 0.01
    mean = torch.from_numpy(sp["mean"]).reshape(-1).double()
    basis = torch.from_numpy(sp["tex_dir"]).reshape(-1, 200)[:, :n_tex].double()
    for b in range(2):
        o = R.tex_pca_texture(mean, basis, code[b].double(), T)
        assert np.allclose(o[0][:, idx[:, 0], idx[:, 1]].numpy(), G[f"val_{n_tex}_{T}"][b], rtol=0, atol=2e-6)


def _pca_tracker(flame_model, device, T=512, H=64, W=64, N=2, seed=3):
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_scene_params, make_texture, smooth_noise
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    cfg = BaseTrackingConfig()
    cfg.device = device
    cfg.model.tex_resolution = T
    cfg.model.tex_painted = False
    rng = np.random.default_rng(seed)
    data = {"rgb": torch.from_numpy(smooth_noise(rng, (N, 3, H, W))).to(device),
            "lmk2d": torch.cat([torch.rand(N, 70, 2, generator=torch.Generator().manual_seed(2)) * W, torch.ones(N, 70, 1)], -1).to(device)}
    sp = _space()
    tr = GlobalTracker(cfg, model, topo, sp, data)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, s in (("shape", 0.3), ("expr", 0.3), ("rotation", 0.05), ("jaw_pose", 0.05), ("tex_extra", 0.03), ("lights", 0.05), ("tex_pca", 1.5)):
            p = getattr(tr, name)
            p.add_((torch.randn(p.shape, generator=g) * s).to(device))
        tr.translation[:, 2] += 0.45
    return tr, cfg, sp


def _oracle(tr, cfg, flame_model, sp, sample, stage, image_size, tid=None, names=None):
    from oracle import energy_ref
    model, topo = flame_model
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].double()
    names = names or ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset",
                      "focal_length", "tex_pca")
    P = {k: getattr(tr, k).detach().cpu().double().requires_grad_() for k in names}
    n = int(tr.tex_pca.shape[0])
    space = {"mean": torch.from_numpy(sp["mean"]).reshape(-1), "basis": torch.from_numpy(sp["tex_dir"]).reshape(-1, 200)[:, :n]}
    o_sample = {"rgb": sample["rgb"].cpu(), "lmk2d": sample["lmk2d"].cpu(), "timestep_index": np.asarray(sample["timestep_index"].cpu() if torch.is_tensor(sample["timestep_index"]) else sample["timestep_index"])}
    E, log, ex = energy_ref.total_energy(P, tm, topo, cfg, o_sample, stage, None, tr._uvmask_res().cpu().double(), image_size, tid=tid, tex_pca_space=space)
    E.backward()
    return E, log, P, ex


def test_tracker_host_formulation_with_pca_texture_matches_oracle(flame_model):
    """CPU: the tracker's host formulation of the texture terms with the PCA base texture (TV of base + residual, reg_tex_pca) against the
    oracle, the tex_pca gradient included; the parameter surface: 'tex' optimiser group, `tex` in the exported npz and back.  (The
    photometric term needs the HIP ops: GPU test below.)"""
    from oracle import energy_ref
    from oracle import torch_ref as R
    H = W = 64
    tr, cfg, sp = _pca_tracker(flame_model, "cpu", T=512, H=H, W=W)
    stage = "rgb_init_texture"
    params = tr.get_train_parameters(stage)
    assert "tex" in params and params["tex"][0] is tr.tex_pca and tr.tex_pca.requires_grad
    ts = np.arange(2)
    verts, v_cano, lmks, albedos = tr.forward_flame(ts)
    n = int(tr.tex_pca.shape[0])
    mean, basis = torch.from_numpy(sp["mean"]).reshape(-1).double(), torch.from_numpy(sp["tex_dir"]).reshape(-1, 200)[:, :n].double()
    code = tr.tex_pca.detach().double().requires_grad_()
    base_o = R.tex_pca_texture(mean, basis, code, 512)
    assert float((albedos[0].detach().double() - (base_o[0] + tr.tex_extra.detach().double())).abs().max()) < 2e-6      # get_albedo (tracker.py:247-258)
    log = tr.compute_regularization_energy({"diffuse_detach_normal": torch.ones(2, 3, H, W)}, verts, v_cano, lmks, None, ts, stage)
    assert "reg_tex_pca" in log and "reg_tex_tv" in log
    (log["reg_tex_pca"] + log["reg_tex_tv"]).backward()
    w = cfg.w
    want_pca = w.reg_tex_pca * (code ** 2).mean()
    want_tv = tr._w_tv() * R.tex_tv_energy((base_o + tr.tex_extra.detach().double()[None])[0])
    (want_pca + want_tv).backward()
    assert abs(float(log["reg_tex_pca"]) - float(want_pca)) <= 1e-6 * float(want_pca)
    assert abs(float(log["reg_tex_tv"]) - float(want_tv)) <= 1e-4 * float(want_tv)
    a, b = tr.tex_pca.grad.double(), code.grad
    assert float(b.abs().max()) > 0 and float((a - b).abs().max() / b.abs().max()) < 1e-3
    out = {k: np.array(v, copy=True) for k, v in tr.save_result().items()}      # (on a CPU device save_result's arrays alias the parameters)
    assert "tex" in out and out["tex"].shape == (cfg.model.n_tex,)
    keep = out["tex"].copy()
    with torch.no_grad():
        tr.tex_pca.zero_()
    tr.load_from_tracked_flame_params(out)
    assert np.array_equal(tr.tex_pca.detach().numpy(), keep)


@pytest.mark.gpu
def test_tex_pca_kernels_match_the_model():
    from vhap_amd import _lib
    from vhap_amd.flame import FlameTexPCA
    sp = _space()
    L = _lib.lib()
    p = lambda t: t.data_ptr()
    for n_tex, T in ((100, 1024), (50, 256), (100, 384)):
        m = FlameTexPCA(n_tex, tex_size=T, tex_space=sp).cuda()
        code = (torch.randn(n_tex, generator=torch.Generator().manual_seed(n_tex + T)) * 2).cuda().requires_grad_()
        ref = m(code[None])[0]
        S = m.src_size
        mean, basis = m.texture_mean.reshape(-1).contiguous(), m.texture_basis[0].contiguous()
        src, base, term = torch.empty(S * S * 3, device="cuda"), torch.empty(3, T, T, device="cuda"), torch.zeros(1, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(L.vhap_tex_pca_fwd(p(mean), p(basis), p(code.detach()), n_tex, S, T, 0.25, p(src), p(base), p(term), st), "fwd")
        assert float((base - ref).abs().max()) < 2e-6
        assert abs(float(term) - 0.25 * float((code.detach() ** 2).sum())) < 1e-3 * float(term)
        d_base = torch.randn(3, T, T, device="cuda")
        (ref * d_base).sum().backward()
        want = code.grad + 2 * 0.25 * code.detach()
        work, d_code = torch.empty(S * S * 3, device="cuda"), torch.zeros(n_tex, device="cuda")
        ones = torch.ones(1, device="cuda")
        _lib.check(L.vhap_tex_pca_bwd(p(basis), p(src), p(d_base), p(code.detach()), n_tex, S, T, 0.25, p(ones), p(work), p(d_code), st), "bwd")
        assert float((d_code - want).abs().max() / want.abs().max()) < 1e-4, float((d_code - want).abs().max() / want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["rgb_init_texture", "rgb_global_tracking"])
def test_native_step_with_pca_texture_matches_oracle(flame_model, stage):
    """The captured step's call sequence with the PCA texture model: energy terms (reg_tex_pca among them) and the gradients of tex_pca,
    tex_extra and the rest against the oracle (same visibility), then replays of the captured step move the code."""
    from vhap_amd.step import NativeStep
    from vhap_amd.tracker import GraphedStep
    H = W = 128
    tr, cfg, sp = _pca_tracker(flame_model, "cuda", T=512, H=H, W=W, N=2)
    tr.render.disturb_rate_fg = tr.render.disturb_rate_bg = None
    assert NativeStep.supported(tr, stage)
    tr.get_train_parameters(stage)
    sample = tr.get_sample(np.arange(2), device_index=True)
    ns = NativeStep(tr, sample, stage)
    assert ns.pca is not None
    ns.forward()
    ns.backward(1)
    torch.cuda.synchronize()
    tid = (ns.rast[..., 3].long() - 1).cpu()
    Eo, logo, P, _ = _oracle(tr, cfg, flame_model, sp, sample, stage, (H, W), tid=tid)
    log_n = {k: float(v) for k, v in ns.log_dict().items()}
    assert "reg_tex_pca" in logo
    for k, b in logo.items():
        b = float(b.detach())
        assert abs(log_n[k] - b) <= 5e-5 * max(abs(b), 1e-3), (k, log_n[k], b)
    assert abs(log_n["total"] - float(Eo)) <= 5e-5 * abs(float(Eo))
    for k in ("tex_pca", "tex_extra", "shape", "expr", "lights"):
        if P[k].grad is None or float(P[k].grad.abs().max()) == 0 or k not in ns.g:
            continue
        a, b = ns.g[k].detach().cpu().double().reshape(-1), P[k].grad.reshape(-1)
        rel = float((a - b).abs().max() / b.abs().max())
        assert rel < 1e-3, (k, rel)
    assert float(P["tex_pca"].grad.abs().max()) > 0
    opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=0.1)
    before = tr.tex_pca.detach().clone()
    st = GraphedStep(tr, sample, opt, stage, warmup=0)
    assert st.ns is not None and st.ns.pca is not None and st.gF.plan is not None
    E0 = float(st())
    for _ in range(4):
        E1 = float(st())
    torch.cuda.synchronize()
    assert abs(E0 - float(Eo)) <= 1e-4 * abs(float(Eo)) and E1 < E0
    assert float((tr.tex_pca.detach() - before).abs().max()) > 0
