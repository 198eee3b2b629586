"""The face detector of the landmark stage (vhap/util/landmark_detector_fa.py:32,45,51 -> the `face_alignment` package's `sfd` = S3FD):
vhap_amd/face_detector.py against the restatement in oracle/sfd_ref.py, seeded random weights (the package and its weights are absent: parity
unpinned against the package itself -- see the oracle's header)."""
import numpy as np
import pytest
import torch

from oracle import sfd_ref


def _frame(seed=4, h=192, w=256):
    return np.random.default_rng(seed).integers(0, 255, (h, w, 3), dtype=np.uint8)


# ---------------------------------------------------------------- CPU: the restatement and the host-side post-processing ----------------------------------------------------------------
def test_restated_detector_has_the_published_shape():
    """VGG-16 trunk + fc6 / fc7 + two extra stages + three L2Norm layers + six heads: 22.5 M parameters, the package's parameter names; six (cls, loc)
    pairs at strides 4 ... 128, two labels each (the first head's three background scores maxed out)"""
    net = sfd_ref.random_s3fd(seed=1)
    n = sum(p.numel() for p in net.parameters())
    assert 22.3e6 < n < 22.6e6, n
    names = set(net.state_dict())
    for k in ("conv1_1.weight", "conv5_3.bias", "fc6.weight", "fc7.bias", "conv6_2.weight", "conv7_1.bias", "conv3_3_norm.weight", "conv5_3_norm.weight",
              "conv3_3_norm_mbox_conf.weight", "conv3_3_norm_mbox_loc.bias", "fc7_mbox_conf.weight", "conv7_2_mbox_loc.weight"):
        assert k in names, k
    assert net.state_dict()["conv3_3_norm_mbox_conf.weight"].shape[0] == 4 and net.state_dict()["conv4_3_norm_mbox_conf.weight"].shape[0] == 2
    with torch.no_grad():
        out = net(sfd_ref.preprocess(_frame(h=128, w=160)))
    assert len(out) == 12
    want = [(32, 40), (16, 20), (8, 10), (8, 9), (4, 5), (2, 3)]          # (fc6: a 3 x 3 convolution with padding 3 -- the map grows by 4)
    for i, (hh, ww) in enumerate(want):
        assert out[2 * i].shape == (1, 2, hh, ww) and out[2 * i + 1].shape == (1, 4, hh, ww), (i, out[2 * i].shape)


def test_host_postprocessing_matches_the_restatement():
    """box decoding and the greedy NMS (the package's +1-pixel areas) of the product against the restatement's, on random boxes"""
    from vhap_amd import face_detector as FD
    rng = np.random.default_rng(0)
    loc = rng.normal(size=(50, 4))
    priors = np.concatenate([rng.uniform(0, 300, (50, 2)), rng.uniform(16, 256, (50, 2))], 1)
    assert np.array_equal(FD.decode(loc, priors), sfd_ref.decode(loc, priors))
    c = rng.uniform(0, 200, (300, 2))
    s = rng.uniform(10, 80, (300, 2))
    dets = np.concatenate([c - s / 2, c + s / 2, rng.uniform(0, 1, (300, 1))], 1)
    assert FD.nms(dets, 0.3) == sfd_ref.nms(dets, 0.3) and 5 < len(FD.nms(dets, 0.3)) < 300
    assert FD.nms(np.zeros((0, 5))) == []
    # the face probability: max-out over the first head's background scores, softmax over two labels
    cls = torch.randn(1, 5, 7, 4)
    c4 = cls.permute(0, 3, 1, 2)
    two = torch.cat([torch.max(torch.max(c4[:, 0:1], c4[:, 1:2]), c4[:, 2:3]), c4[:, 3:4]], 1)
    assert torch.allclose(FD.face_probability(cls), torch.softmax(two, 1)[:, 1], atol=1e-7)


def test_no_cpu_path():
    from vhap_amd import face_detector as FD
    sd = sfd_ref.random_s3fd(seed=0).state_dict()
    with pytest.raises(RuntimeError):
        FD.S3FD(sd, device="cpu")


# ---------------------------------------------------------------- GPU ----------------------------------------------------------------
def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.gpu
def test_maxpool_and_l2norm_match_torch():
    from vhap_amd import _lib
    from vhap_amd.ops import _p, _stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 9, 14, 40, generator=g).cuda()                    # odd height: the trailing row is dropped (floor mode)
    out = torch.empty(2, 4, 7, 40).cuda()
    assert L.vhap_nhwc_maxpool2(_p(x), 2, 9, 14, 40, _p(out), _stream()) == 0
    assert torch.equal(out.cpu(), torch.nn.functional.max_pool2d(x.cpu().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1))
    for C in (256, 512, 70):
        x = torch.randn(3, 5, 6, C, generator=g).cuda()
        w = (torch.rand(C, generator=g) * 10).cuda()
        o = torch.empty_like(x)
        assert L.vhap_nhwc_l2norm(_p(x), 3 * 5 * 6, C, _p(w), 1e-10, _p(o), _stream()) == 0
        m = sfd_ref.L2Norm(C)
        with torch.no_grad():
            m.weight.copy_(w.cpu())
            ref = m(x.cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        assert _rel(o.cpu(), ref) <= 1e-6


@pytest.mark.gpu
def test_network_matches_the_torch_restatement():
    """every head's scores and box regressions against torch fp32 on the CPU, seeded random weights: <= 2e-4 of their max-norm"""
    from vhap_amd import face_detector as FD
    net = sfd_ref.random_s3fd(seed=2, face_bias=-2.0)
    img = _frame()
    with torch.no_grad():
        ref = net(sfd_ref.preprocess(img))
    det = FD.S3FD(net.state_dict())
    x = sfd_ref.preprocess(img).permute(0, 2, 3, 1).contiguous().cuda()
    out = det(x)
    torch.cuda.synchronize()
    assert len(out) == 6
    for i, (cls, loc) in enumerate(out):
        rc, rl = ref[2 * i], ref[2 * i + 1]
        if cls.shape[-1] == 4:                                              # (the product keeps the four scores; the maximum is taken with the softmax)
            cls = torch.cat([cls[..., :3].max(-1, keepdim=True).values, cls[..., 3:]], -1)
        assert cls.shape[1:3] == rc.shape[2:] and loc.shape[1:3] == rl.shape[2:]
        assert _rel(cls.cpu().permute(0, 3, 1, 2), rc) <= 2e-4, (i, _rel(cls.cpu().permute(0, 3, 1, 2), rc))
        assert _rel(loc.cpu().permute(0, 3, 1, 2), rl) <= 2e-4, (i, _rel(loc.cpu().permute(0, 3, 1, 2), rl))
    with pytest.raises(RuntimeError):
        det(x.cpu())


@pytest.mark.gpu
def test_detector_end_to_end_and_in_front_of_the_landmark_network():
    """SFDDetector = the package's detect_from_image: candidates above the pre-threshold, decoding, NMS, score filter -- against the restatement;
    then as `face_detector=` of LandmarkDetectorFA (the reference's detect_single_image: the best box goes to the landmark network)"""
    from oracle import fan_ref
    from vhap_amd import face_detector as FD
    from vhap_amd import landmarks as LM
    net = sfd_ref.random_s3fd(seed=2, face_bias=-2.0)
    img = _frame()
    want = sfd_ref.detect(net, img)
    det = FD.SFDDetector(net.state_dict())
    got = det(img)
    with torch.no_grad():
        cand_ref = sfd_ref.candidates([o.numpy() for o in net(sfd_ref.preprocess(img))])
    cand = det.candidates(img)
    assert abs(len(cand) - len(cand_ref)) <= max(2, len(cand_ref) // 500)           # (a location may sit on the 0.05 pre-threshold)
    assert len(want) >= 1 and len(got) == len(want), (len(got), len(want))
    for a, b in zip(got, want):
        assert np.abs(a[:4] - b[:4]).max() <= 2e-2 and abs(a[4] - b[4]) <= 1e-4, (a, b)
    fan = fan_ref.random_fan(seed=5, num_modules=1)
    lm = LM.LandmarkDetectorFA(fan.state_dict(), face_detector=det, num_modules=1)
    bbox, lmks = lm.detect_single_image(img)
    best = max(want, key=lambda d: d[-1])
    assert len(bbox) == 1 and np.allclose(bbox[0][:4] * np.array([256, 192, 256, 192]), best[:4], atol=2e-2) and lmks.shape == (68, 3)
