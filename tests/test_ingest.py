"""Frame ingest.  CPU: the oracle restatement against golden vectors produced by the reference's own
VideoDataset.apply_background_color / apply_to_tensor (tools/make_golden_ingest.py).  GPU: vhap_frame_ingest through the C ABI against
the oracle and the same golden vectors, bit for bit (uint8 compositing is integer-valued; the fp32 /255 is one correctly rounded divide)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ingest_golden.npz")
MODES = (None, "white", "black")


def test_oracle_matches_reference_golden():
    from oracle import ingest_ref as R
    g = np.load(GOLD)
    a_all, f_all = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for mode in MODES:
        rgb, alpha = R.frame_ingest(g["rgb"], g["alpha"], None, mode)
        assert rgb.dtype == np.float32 and np.array_equal(rgb, g[f"rgb_{mode}"])
        assert np.array_equal(alpha, g[f"alpha_{mode}"])
        ex = R.apply_background_color(np.repeat(f_all[:, :, None], 3, axis=2), a_all, mode)[..., 0]
        assert np.array_equal(ex, g[f"exhaustive_{mode}"])
    with pytest.raises(NotImplementedError):
        R.apply_background_color(g["rgb"], g["alpha"], "green")


def test_framestore_host_errors():
    from vhap_amd.ingest import FrameStore
    rgb = np.zeros((2, 4, 4, 3), np.uint8)
    with pytest.raises(NotImplementedError):
        FrameStore(rgb, np.zeros((2, 4, 4), np.uint8), "green", device="cpu")
    with pytest.raises(AssertionError):
        FrameStore(rgb, None, "white", device="cpu")
    with pytest.raises(ValueError):
        FrameStore(rgb, np.zeros((2, 4, 5), np.uint8), None, device="cpu")
    with pytest.raises(RuntimeError):
        FrameStore(rgb, None, None, device="cpu").batch()          # no CPU path


@pytest.mark.gpu
def test_hip_ingest_golden_and_exhaustive():
    import torch
    from vhap_amd.ingest import FrameStore
    g = np.load(GOLD)
    a_all, f_all = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for mode in MODES:
        st = FrameStore(g["rgb"], g["alpha"], mode)                 # 9 x 14: H*W % 4 != 0 -> the scalar kernel
        rgb, alpha = st.batch()
        assert np.array_equal(rgb.cpu().numpy(), g[f"rgb_{mode}"]) and np.array_equal(alpha.cpu().numpy(), g[f"alpha_{mode}"])
        ex = FrameStore(np.repeat(f_all[None, :, :, None], 3, axis=3), a_all[None], mode)       # 256 x 256 -> the 4-pixel kernel
        out, _ = ex.batch()
        want = g[f"exhaustive_{mode}"].astype(np.float32) / np.float32(255)
        assert np.array_equal(out.cpu().numpy()[0, 0], want) and torch.equal(out[0, 0], out[0, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(5, 33, 47), (6, 64, 96), (20, 512, 512)])
def test_hip_ingest_vs_oracle(N, H, W):
    import torch
    from oracle import ingest_ref as R
    from vhap_amd.ingest import FrameStore
    rng = np.random.default_rng(N * 1000 + H)
    rgb = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    alpha = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
    alpha[rng.random((N, H, W)) < 0.3] = 255
    alpha[rng.random((N, H, W)) < 0.3] = 0
    idx = np.array([N - 1, 0, 2, 2, -1, 1] + list(range(N)))[: max(3, min(16, N + 6))]
    for mode in MODES:
        st = FrameStore(rgb, alpha, mode)
        out, a = st.batch(idx, check=True)
        want, want_a = R.frame_ingest(rgb, alpha, idx, mode)
        assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(a.cpu().numpy(), want_a)
    st = FrameStore(rgb, None, None)                                # no alpha at all
    out, a = st.batch(idx)
    assert a is None and np.array_equal(out.cpu().numpy(), R.frame_ingest(rgb, None, idx, None)[0])
    with pytest.raises(IndexError):
        st.batch([0, N], check=True)
    # static-buffer form used by the captured step
    buf = torch.full((2, 3, H, W), -1.0, device="cuda")
    st.batch(torch.tensor([1, 0], device="cuda"), out=buf)
    assert np.array_equal(buf.cpu().numpy(), R.frame_ingest(rgb, None, [1, 0], None)[0])


@pytest.mark.gpu
def test_tracker_on_uint8_store_matches_fp32_dataset(flame_model):
    """The fit fed from the resident uint8 store takes the same steps as the fit fed the host-converted fp32 frames (the frames themselves
    are bit-identical, see above)."""
    import torch
    from oracle import ingest_ref as R
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    from vhap_amd.ingest import FrameStore
    from vhap_amd.render_hip import HipDiffRenderer
    from vhap_amd.synthetic import make_dataset, make_scene_params, make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    N, H, W, T = 6, 128, 128, 256
    gt = make_scene_params(N, seed=3, image_size=(H, W))
    head, rend = FlameHead(model, topo).cuda(), HipDiffRenderer(lighting_type="SH").cuda()
    data = make_dataset(rend, head, gt, (H, W), "cuda", seed=3, tex=make_texture(3, T))
    rgb_u8 = (data["rgb"].permute(0, 2, 3, 1) * 255).round().to(torch.uint8).cpu().numpy()
    yy, xx = np.mgrid[0:H, 0:W]
    alpha_u8 = np.clip(300 - 4 * np.hypot(yy - H / 2, xx - W / 2), 0, 255).astype(np.uint8)[None].repeat(N, 0)
    fp32 = torch.from_numpy(R.frame_ingest(rgb_u8, alpha_u8, None, "white")[0]).cuda()

    def run(dataset):
        cfg = BaseTrackingConfig()
        cfg.model.tex_resolution = T
        cfg.render.disturb_rate_fg = cfg.render.disturb_rate_bg = None
        tr = GlobalTracker(cfg, model, topo, make_texture(0, T), dataset)
        with torch.no_grad():
            tr.translation[:, 2] += 0.45
        stage = "rgb_global_tracking"
        opt = tr.configure_optimizer(tr.get_train_parameters(stage))
        from vhap_amd.tracker import GraphedStep
        st = GraphedStep(tr, tr.get_sample([0, 1, 2], device_index=True), opt, stage, warmup=0)
        es = [float(st()) for _ in range(3)]
        st.update_timesteps([3, 4, 5])
        es += [float(st()) for _ in range(3)]
        return es, tr.expr.detach().clone()

    e_a, x_a = run({"rgb": fp32, "lmk2d": data["lmk2d"]})
    e_b, x_b = run({"frames": FrameStore(rgb_u8, alpha_u8, "white"), "lmk2d": data["lmk2d"]})
    # (the two fits are separate runs of kernels that accumulate with float atomics: equal up to summation order, amplified over 6 steps)
    np.testing.assert_allclose(e_a, e_b, rtol=3e-4)
    assert float((x_a - x_b).abs().max()) <= 2e-3 * float(x_a.abs().max()) + 1e-6
