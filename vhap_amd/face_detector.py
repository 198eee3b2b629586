"""The face detector of the reference's landmark stage on ROCm (second half): S3FD on the MI355X's matrix cores.

Reference: `LandmarkDetectorFA.detect_single_image` (vhap/util/landmark_detector_fa.py:50-58) asks the third-party `face_alignment` package's
`sfd` detector for boxes first (`self.fa.face_detector.detect_from_image(img)`, `face_detector='sfd'` at :32,45), keeps the box with the highest
score and hands it to the landmark network.  The package is absent from the reference checkout and from this image; its detector is the published
S3FD (Zhang et al., ICCV 2017: VGG-16 trunk with fc6 / fc7 as convolutions and two extra stages, L2-normalised conv3_3 / conv4_3 / conv5_3, six heads
at strides 4 ... 128, one square anchor of 4 x stride per location, max-out background on the first head), restated for the tests in oracle/sfd_ref.py.

Built here:
  * `S3FD`: the network's forward as 21 launches of `vhap_conv2d_nhwc_ws` (exact-fp32 MFMA, ReLU on the way out) + 5 `vhap_nhwc_maxpool2` +
    3 `vhap_nhwc_l2norm` + 6 head launches (a head's `conf` and `loc` convolutions share their input: ONE convolution with 8 output channels).  Takes
    the package's state dict (same parameter names).  No CPU path: CPU tensors raise.
  * `SFDDetector(weights)(img)` = the package's `detect_from_image`: RGB uint8 frame -> BGR minus the channel means on the device -> network -> softmax,
    0.05 pre-threshold (device) -> box decoding (variances 0.1 / 0.2), greedy NMS at 0.3, 0.5 score filter (host, float64 numpy: a handful of boxes) ->
    list of [x1, y1, x2, y2, score]; plugs into `LandmarkDetectorFA(face_detector=...)`.
Pinned by tests/test_face_detector.py on the restatement with seeded random weights (the package and its weights are absent: parity unpinned against it)."""
import os

import numpy as np
import torch

from . import _lib
from .landmarks import _chk, _Conv
from .ops import _p, _stream

TRUNK = (("conv1_1", "conv1_2"), ("conv2_1", "conv2_2"), ("conv3_1", "conv3_2", "conv3_3"), ("conv4_1", "conv4_2", "conv4_3"),
         ("conv5_1", "conv5_2", "conv5_3"))
HEADS = (("conv3_3_norm", 4), ("conv4_3_norm", 2), ("conv5_3_norm", 2), ("fc7", 2), ("conv6_2", 2), ("conv7_2", 2))
BGR_MEAN = (104.0, 117.0, 123.0)


def _maxpool2(x):
    N, H, W, C = x.shape
    out = torch.empty(N, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    _chk(_lib.lib().vhap_nhwc_maxpool2(_p(x), N, H, W, C, _p(out), _stream()), "vhap_nhwc_maxpool2")
    return out


class S3FD:
    """The network of oracle/sfd_ref.py::S3FD from the package's state dict.  forward(x [N,H,W,3] float32 on the GPU: BGR minus the means, channel
    last) -> list of six (cls [N,h,w,ncls], loc [N,h,w,4]) pairs, channel-last views of the heads' 8-channel outputs."""

    def __init__(self, state_dict, device="cuda"):
        if not str(device).startswith("cuda"):
            raise RuntimeError("S3FD runs on the HIP device only (there is no CPU path)")
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        self.device = device
        c = lambda name, **kw: _Conv(sd[name + ".weight"], sd[name + ".bias"], out_relu=True, device=device, **kw)
        self.trunk = [[c(n, pad=1) for n in stage] for stage in TRUNK]
        self.fc6, self.fc7 = c("fc6", pad=3), c("fc7")
        self.conv6_1, self.conv6_2 = c("conv6_1"), c("conv6_2", stride=2, pad=1)
        self.conv7_1, self.conv7_2 = c("conv7_1"), c("conv7_2", stride=2, pad=1)
        self.norm = [sd[f"{n}.weight"].detach().to(torch.float32).contiguous().to(device) for n in ("conv3_3_norm", "conv4_3_norm", "conv5_3_norm")]
        self.heads = []
        for name, ncls in HEADS:                           # conf [ncls] | loc [4] | zeros: one convolution, 8 output channels (16-byte weight rows)
            wc, bc = sd[f"{name}_mbox_conf.weight"].detach().float(), sd[f"{name}_mbox_conf.bias"].detach().float()
            wl, bl = sd[f"{name}_mbox_loc.weight"].detach().float(), sd[f"{name}_mbox_loc.bias"].detach().float()
            pad = 8 - ncls - 4
            w = torch.cat([wc, wl, torch.zeros(pad, *wc.shape[1:])], 0)
            b = torch.cat([bc, bl, torch.zeros(pad)], 0)
            self.heads.append((_Conv(w, b, pad=1, device=device), ncls))

    @staticmethod
    def _run(conv, x):
        N, H, W, _ = x.shape
        Ho, Wo = (H + 2 * conv.pad - conv.KH) // conv.stride + 1, (W + 2 * conv.pad - conv.KW) // conv.stride + 1
        return conv(x, 0, conv.cin, torch.empty(N, Ho, Wo, conv.cout, dtype=torch.float32, device=x.device), 0)

    def __call__(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("S3FD.forward needs a tensor on the HIP device (there is no CPU path)")
        assert x.dim() == 4 and x.shape[3] == 3 and x.dtype == torch.float32 and min(x.shape[1:3]) >= 128, "[N,H,W,3] float32, at least 128 pixels a side"
        h, taps = x.contiguous(), []
        for i, stage in enumerate(self.trunk):
            for conv in stage:
                h = self._run(conv, h)
            if i >= 2:
                taps.append(h)
            h = _maxpool2(h)
        f7 = self._run(self.fc7, self._run(self.fc6, h))
        f6_2 = self._run(self.conv6_2, self._run(self.conv6_1, f7))
        f7_2 = self._run(self.conv7_2, self._run(self.conv7_1, f6_2))
        feats = []
        for t, w in zip(taps, self.norm):
            o = torch.empty_like(t)
            _chk(_lib.lib().vhap_nhwc_l2norm(_p(t), t.numel() // t.shape[3], t.shape[3], _p(w), 1e-10, _p(o), _stream()), "vhap_nhwc_l2norm")
            feats.append(o)
        feats += [f7, f6_2, f7_2]
        out = []
        for (conv, ncls), f in zip(self.heads, feats):
            o = self._run(conv, f)
            out.append((o[..., :ncls], o[..., ncls:ncls + 4]))
        return out


def face_probability(cls):
    """[N,h,w,ncls] head scores -> [N,h,w] face probability: max-out over the first head's three background scores, softmax over (background, face)"""
    bg = cls[..., :-1].max(dim=-1).values
    two = torch.stack([bg, cls[..., -1]], -1)
    e = torch.exp(two - two.max(dim=-1, keepdim=True).values)
    return e[..., 1] / e.sum(-1)


def decode(loc, priors, variances=(0.1, 0.2)):
    boxes = np.concatenate((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:], priors[:, 2:] * np.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def nms(dets, thresh=0.3):
    """greedy non-maximum suppression with the package's +1 pixel areas; -> indices kept, best first"""
    if len(dets) == 0:
        return []
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        ovr = w * h / (areas[i] + areas[rest] - w * h)
        order = rest[ovr <= thresh]
    return keep


class SFDDetector:
    """`face_alignment`'s SFDDetector.detect_from_image over S3FD: callable(img [H,W,3] uint8 RGB) -> list of [x1, y1, x2, y2, score] arrays.
    `weights`: the package's state dict, or a path torch.load can read."""

    def __init__(self, weights, device="cuda", filter_threshold=0.5, pre_threshold=0.05, nms_threshold=0.3):
        if isinstance(weights, (str, os.PathLike)):
            weights = torch.load(weights, map_location="cpu")
            weights = weights.get("state_dict", weights) if isinstance(weights, dict) else weights.state_dict()
        self.net = S3FD(weights, device=device)
        self.device, self.filter_threshold, self.pre_threshold, self.nms_threshold = device, filter_threshold, pre_threshold, nms_threshold

    def candidates(self, img):
        """every location above the pre-threshold, decoded: [n,5] float64 (x1, y1, x2, y2, score), head by head in row-major order"""
        x = torch.as_tensor(np.ascontiguousarray(np.asarray(img)[..., ::-1]), device=self.device).to(torch.float32)       # RGB -> BGR
        x = (x - torch.tensor(BGR_MEAN, device=self.device))[None]
        rows = []
        for i, (cls, loc) in enumerate(self.net(x)):
            prob = face_probability(cls)[0]
            idx = torch.nonzero(prob > self.pre_threshold)
            if idx.numel() == 0:
                continue
            hy, wx = idx[:, 0], idx[:, 1]
            got = torch.cat([loc[0][hy, wx], prob[hy, wx][:, None], idx.to(torch.float32)], 1).cpu().numpy().astype(np.float64)
            stride = 2 ** (i + 2)
            priors = np.stack([stride / 2 + got[:, 6] * stride, stride / 2 + got[:, 5] * stride, np.full(len(got), stride * 4.0),
                               np.full(len(got), stride * 4.0)], 1)
            rows.append(np.concatenate([decode(got[:, :4], priors), got[:, 4:5]], 1))
        return np.concatenate(rows, 0) if rows else np.zeros((0, 5))

    def __call__(self, img):
        dets = self.candidates(img)
        if len(dets) == 0:
            return []
        dets = dets[nms(dets, self.nms_threshold)]
        return [d for d in dets if d[-1] > self.filter_threshold]

    detect_from_image = __call__
