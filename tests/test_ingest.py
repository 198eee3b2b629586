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


# ---- round 5: the transforms ahead of the compositing -- per-camera colour correction, scale-factor resize ----
GOLD_CC = os.path.join(os.path.dirname(__file__), "golden", "ingest_cc_golden.npz")
SF_CASES = 5


def test_oracle_color_correction_and_resize_match_reference_golden():
    """oracle/ingest_oracle.c against vectors the reference's OWN methods produced (tools/make_golden_ingest.py:
    NeRSembleDataset.apply_color_correction, VideoDataset.apply_scale_factor with PIL), and -- where PIL is importable -- against PIL live."""
    from oracle import ingest_ref as R
    g = np.load(GOLD_CC)
    A = g["ccm"]
    for cam in range(A.shape[0]):
        assert np.array_equal(R.apply_color_correction(g["cc_rgb"][cam], A[cam]), g["cc_out"][cam])
        assert np.array_equal(R.apply_color_correction(g["cc_sweep_in"], A[cam]), g["cc_sweep_out"][cam])
        assert np.array_equal(R.apply_color_correction(g["cc_triples_in"], A[cam]), g["cc_triples_out"][cam])
    assert np.array_equal(g["cc_sweep_out"][0], g["cc_sweep_in"])              # the identity transform returns every 8-bit value
    assert (g["cc_triples_out"][3] == 0).any() and (g["cc_triples_out"][3] == 255).any()      # the clip is exercised at both ends
    for i in range(SF_CASES):
        sf, nds = g[f"sf{i}_cfg"]
        out = R.apply_scale_factor(g[f"sf{i}_in_rgb"], g[f"sf{i}_in_alpha_map"], float(sf), int(nds) or None, g[f"sf{i}_in_lmk2d"],
                                   g[f"sf{i}_in_intrinsic"])
        for k in ("rgb", "alpha_map", "lmk2d", "intrinsic"):
            assert np.array_equal(out[k], g[f"sf{i}_out_{k}"]), (i, k)
        assert out["scale_factor"] == float(g[f"sf{i}_out_scale"])
    try:
        from PIL import Image
    except ImportError:
        return
    rng = np.random.default_rng(4)
    for H, W, h, w in [(61, 47, 30, 23), (48, 64, 48, 20), (33, 90, 11, 90), (200, 150, 66, 50)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(R.pil_resize(img, h, w), np.array(Image.fromarray(img).resize((w, h), resample=Image.BILINEAR)))


def test_host_resampling_tables_match_the_oracle():
    """vhap_amd.ingest.pil_bilinear_coeffs (the product's host-side restatement of Pillow's precompute_coeffs + normalize_coeffs_8bpc,
    what vhap_frame_resize_u8 is fed) against the C oracle's independent one."""
    import ctypes
    import oracle
    from vhap_amd.ingest import pil_bilinear_coeffs, scale_item_properties
    ip = ctypes.POINTER(ctypes.c_int32)
    for n_in, n_out in [(550, 275), (802, 401), (3208, 802), (100, 33), (64, 63), (17, 5), (2200, 550), (9, 1)]:
        b, k, ks = pil_bilinear_coeffs(n_in, n_out)
        bb, kk = np.zeros(2 * n_out, np.int32), np.zeros(n_out * ks, np.int32)
        assert oracle.lib().oracle_pil_coeffs(n_in, n_out, bb.ctypes.data_as(ip), kk.ctypes.data_as(ip), n_out * ks) == ks
        assert np.array_equal(b, bb) and np.array_equal(k, kk)
    g = np.load(GOLD_CC)
    for i in range(SF_CASES):
        sf, nds = g[f"sf{i}_cfg"]
        h, w = g[f"sf{i}_out_rgb"].shape[:2]
        got = scale_item_properties({"scale_factor": float(sf) / (nds or 1), "image_size": (h, w)}, g[f"sf{i}_in_lmk2d"], g[f"sf{i}_in_intrinsic"])
        assert np.array_equal(got["lmk2d"], g[f"sf{i}_out_lmk2d"]) and np.array_equal(got["intrinsic"], g[f"sf{i}_out_intrinsic"])


@pytest.mark.gpu
def test_hip_color_correction_and_resize_match_reference_golden():
    """FrameStore.from_decoded (vhap_frame_color_correct, vhap_frame_resize_u8) against the reference-made golden, bit for bit: every
    camera's transform on the sample image, the exhaustive per-channel sweeps, 16 384 random triples; every scale-factor case incl. the
    alpha map stored at n_downsample_rgb times the rgb's size; then the composited fp32 batch of the whole chain."""
    import torch
    from oracle import ingest_ref as R
    from vhap_amd.ingest import FrameStore
    g = np.load(GOLD_CC)
    A = g["ccm"]
    n_cam = A.shape[0]
    st = FrameStore.from_decoded(g["cc_rgb"], camera_index=np.arange(n_cam), color_correction=A)
    assert np.array_equal(st.rgb.cpu().numpy(), g["cc_out"])
    for cam in range(n_cam):
        for key in ("cc_sweep", "cc_triples"):
            src = g[key + "_in"]
            st = FrameStore.from_decoded(src[None], camera_index=[cam], color_correction=A[:, :3])       # ([n_cam,3,4] form)
            assert np.array_equal(st.rgb.cpu().numpy()[0], g[key + "_out"][cam]), (cam, key)
    for i in range(SF_CASES):
        sf, nds = g[f"sf{i}_cfg"]
        st = FrameStore.from_decoded(g[f"sf{i}_in_rgb"][None], g[f"sf{i}_in_alpha_map"][None], "white", scale_factor=float(sf),
                                     n_downsample_rgb=int(nds) or None)
        assert np.array_equal(st.rgb.cpu().numpy()[0], g[f"sf{i}_out_rgb"]), i
        assert np.array_equal(st.alpha.cpu().numpy()[0], g[f"sf{i}_out_alpha_map"]), i
        assert st.prepared["scale_factor"] == float(g[f"sf{i}_out_scale"]) and st.prepared["image_size"] == g[f"sf{i}_out_rgb"].shape[:2]
        out, a = st.batch()
        want, want_a = R.frame_ingest(g[f"sf{i}_out_rgb"][None], g[f"sf{i}_out_alpha_map"][None], None, "white")
        assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(a.cpu().numpy(), want_a)
    with pytest.raises(AssertionError):
        FrameStore.from_decoded(g["cc_rgb"], scale_factor=1.5)
    with pytest.raises(ValueError):
        FrameStore.from_decoded(g["cc_rgb"], color_correction=A)                # four transforms, no camera index


@pytest.mark.gpu
def test_hip_frame_preparation_vs_oracle_at_nersemble_size():
    """BASELINE config 4's ingest: 16 views 802 x 550 (3208 x 2200 / n_downsample_rgb 4), one colour transform per camera, alpha maps at
    full size, 'white' background -- and the same with scale_factor 0.5 on top -- against the C oracle, bit for bit."""
    import torch
    from oracle import ingest_ref as R
    from vhap_amd.ingest import FrameStore
    rng = np.random.default_rng(16)
    N, H, W = 16, 802, 550
    rgb = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    alpha = (rng.random((N, 2 * H, 2 * W)) * 1.4 * 255).clip(0, 255).astype(np.uint8)          # (stored at twice the rgb's size here)
    A = np.tile(np.eye(4), (N, 1, 1))
    A[:, :3, :3] += rng.standard_normal((N, 3, 3)) * 0.06
    A[:, :3, 3] = rng.standard_normal((N, 3)) * 0.02
    cams = rng.permutation(N)
    for sf in (1.0, 0.5):
        st = FrameStore.from_decoded(rgb, alpha, "white", camera_index=cams, color_correction=A, scale_factor=sf, n_downsample_rgb=2)
        h, w = int(H * sf), int(W * sf)
        want_rgb = np.stack([R.pil_resize(R.apply_color_correction(rgb[i], A[cams[i]]), h, w) for i in range(N)])
        want_a = np.stack([R.pil_resize(alpha[i], h, w) for i in range(N)])
        assert np.array_equal(st.rgb.cpu().numpy(), want_rgb) and np.array_equal(st.alpha.cpu().numpy(), want_a)
        out, _ = st.batch(torch.arange(N, device="cuda"))
        assert np.array_equal(out.cpu().numpy(), R.frame_ingest(want_rgb, want_a, None, "white")[0])


class _DuckDataset:
    """The slice of the reference's dataset interface vhap_amd.reference_adapter reads (video_dataset.py:209-259, nersemble_dataset.py:
    160-171): cfg, items, __getitem__ -> apply_transforms, color_correction -- with the ORACLE's restatements as the host transforms (the
    reference's own classes play this part in tests/test_reference_adapter.py, where the checkout exists)."""
    batchify_all_views = False
    img_to_tensor = False

    def __init__(self, cfg, rgb, alpha, lmk, cams, A, K, RT):
        self.cfg, self.rgb, self.alpha, self.lmk, self.cams, self.K, self.RT = cfg, rgb, alpha, lmk, cams, K, RT
        self.color_correction = {c: A[i] for i, c in enumerate(cams)}
        nt = len(rgb) // len(cams)
        self.items = [{"timestep_index": t, "camera_id": c} for t in range(nt) for c in cams]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        it = dict(self.items[i], rgb=self.rgb[i].copy(), alpha_map=self.alpha[i].copy(), lmk2d=self.lmk[i].copy(),
                  intrinsic=self.K.copy(), extrinsic=self.RT[i % len(self.cams)].copy())
        return self.apply_transforms(it)

    def apply_transforms(self, it):
        from oracle import ingest_ref as R
        c = self.cfg
        if c.use_color_correction:
            it["rgb"] = R.apply_color_correction(it["rgb"], self.color_correction[it["camera_id"]])
        sc = R.apply_scale_factor(it["rgb"], it["alpha_map"], c.scale_factor, c.n_downsample_rgb, it["lmk2d"], it["intrinsic"])
        it.update(rgb=sc["rgb"], alpha_map=sc["alpha_map"], lmk2d=sc["lmk2d"], intrinsic=sc["intrinsic"])
        it["rgb"] = R.apply_background_color(it["rgb"], it["alpha_map"], c.background_color)
        return it


@pytest.mark.gpu
@pytest.mark.parametrize("bg,sf,nds", [("white", 0.5, 2), (None, 1.0, 2), ("black", 0.6, None)])
def test_adapter_device_preparation_equals_host_transforms(bg, sf, nds):
    """reference_adapter.frames_from_reference_dataset: device_prepare=True (decoder output uploaded; colour correction, scale factor and
    compositing on the device) gives the fit the SAME batch, bit for bit, as device_prepare=False (the dataset's own host transforms),
    and the same landmarks / intrinsics -- multi-view layout, alpha maps at n_downsample_rgb times the rgb's size."""
    import types
    import torch
    from vhap_amd.reference_adapter import frames_from_reference_dataset
    rng = np.random.default_rng(9)
    cams = ["a", "b", "c"]
    nt, H, W = 2, 44, 60
    N = nt * len(cams)
    up = nds or 1
    rgb = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    alpha = rng.integers(0, 256, (N, H * up, W * up), dtype=np.uint8)
    lmk = np.concatenate([rng.random((N, 68, 2)).astype(np.float32), np.ones((N, 68, 1), np.float32)], -1)
    A = np.tile(np.eye(4), (3, 1, 1))
    A[:, :3, :3] += rng.standard_normal((3, 3, 3)) * 0.05
    K = np.array([[900.0, 0, 400.5], [0, 910.0, 300.25], [0, 0, 1]], np.float32)
    RT = rng.standard_normal((3, 3, 4)).astype(np.float32)
    cfg = types.SimpleNamespace(use_color_correction=True, scale_factor=sf, n_downsample_rgb=nds, background_color=bg, calibrated=True,
                                use_alpha_map=True)
    host = frames_from_reference_dataset(_DuckDataset(cfg, rgb, alpha, lmk, cams, A, K, RT), "cuda", device_prepare=False)
    dev = frames_from_reference_dataset(_DuckDataset(cfg, rgb, alpha, lmk, cams, A, K, RT), "cuda", device_prepare=True)
    a, _ = host["frames"].batch()
    b, _ = dev["frames"].batch()
    assert a.shape == (N, 3, int(H * sf), int(W * sf)) and torch.equal(a, b)
    for k in ("lmk2d", "intrinsic", "extrinsic"):
        assert torch.equal(host[k], dev[k]), k
    assert list(dev["timestep_index"]) == [0, 0, 0, 1, 1, 1] and list(dev["camera_index"]) == [0, 1, 2] * 2
