"""Landmark detection on ROCm: the 2-D landmark network of the reference's detector on the MI355X's matrix cores.

Reference: `GlobalTracker.detect_landmarks` (vhap/model/tracker.py:1263-1277) -> `vhap/util/landmark_detector_fa.py`:
`LandmarkDetectorFA` (:25-78) wraps the third-party `face_alignment.FaceAlignment(LandmarksType.TWO_HALF_D, face_detector='sfd',
flip_input=True)`, `annotate_landmarks` (:143-190) runs it over a dataset and writes `landmark2d/face-alignment/<camera>.npz` with
`face_landmark_2d [T,68,3]` (x / w, y / h, confidence) and `bounding_box [T,5]`.  The package is absent from the reference checkout and from
this image; its network is the published FAN (Bulat & Tzimiropoulos, ICCV 2017), restated for the tests in oracle/fan_ref.py.

Built here (round 6 -- what is and is not there):
  * `FAN2D`: the network's forward as ~200 launches of `vhap_conv2d_nhwc_ws` (exact-fp32 MFMA implicit GEMM, K split over workgroups where
    the output is small; 4.0 ms for a crop and its mirror image, profiles/r06_fan_bench.txt; the blocks' pre-activation
    BatchNorm + ReLU fused into the staging of the input tile, the three convolutions of a block writing the channel slices of its concatenated
    output in place, the projected skip accumulated by the last launch) and the hourglass's elementwise glue (`vhap_nhwc_avgpool2`,
    `vhap_nhwc_upsample2_add`, `vhap_nhwc_add`).  Takes the package's state dict (same parameter names), so its weights load unchanged.
    There is NO eager / CPU fallback: CPU tensors raise.
  * `LandmarkDetectorFA.detect_single_image(img)` with the reference's return convention (`detect_images(imgs)`: several frames and their
    mirror images in ONE pass of the network -- a frame's result does not depend on its batch), `annotate_landmarks(dataset, detector)` with the
    reference's npz layout; crop / flip-averaging / heat-map decoding restated from the package's published behaviour.
  * NOT built: the face detector (`sfd`, a second third-party network): boxes are an input (`face_detector=` any callable, or the whole frame);
    the STAR detector (vhap/util/landmark_detector_star.py); the package's weights (a third-party download).
Pinned by tests/test_landmarks.py: every layer kind and the whole 4-stack network against the torch restatement with seeded random weights and
non-trivial BatchNorm statistics (fp32, heat maps to 2e-4 of their max-norm), the decoding against the restatement, the npz layout against the
reference's writer."""
import os

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

# mirror pairs of the 68-point markup (the package's flip(..., is_label=True) swaps them when it averages the flipped pass)
_PAIRS = [(0, 16), (1, 15), (2, 14), (3, 13), (4, 12), (5, 11), (6, 10), (7, 9), (17, 26), (18, 25), (19, 24), (20, 23), (21, 22), (36, 45), (37, 44),
          (38, 43), (39, 42), (41, 46), (40, 47), (31, 35), (32, 34), (50, 52), (49, 53), (48, 54), (61, 63), (60, 64), (67, 65), (59, 55), (58, 56)]
MIRROR_68 = np.arange(68)
for _a, _b in _PAIRS:
    MIRROR_68[_a], MIRROR_68[_b] = _b, _a


def _chk(rc, what):
    _lib.check(rc, what)


_WS = {}


def _workspace(device):
    """the split-K workspace of vhap_conv2d_nhwc_ws, one per device and stream (launches of one stream use it one after the other): 4 M floats hold the
    partial sums of every layer the library splits at the detector's batch sizes (the largest: 4 slices of a 32 x 32 x 128 output, batch 4)"""
    key = str(device)
    if key in _WS:                                         # (tools/fan_bench.py --no-split plants an empty one under the device's name)
        return _WS[key]
    key = (key, torch.cuda.current_stream(device).cuda_stream)   # one per stream: two streams running networks side by side must not share partial sums
    if key not in _WS:
        _WS[key] = torch.empty(1 << 22, dtype=torch.float32, device=device)
    return _WS[key]


class _Conv:
    """one convolution's device-side constants: weight [KH,KW,Cin,Cout], optional bias, optional input BatchNorm as (scale, shift)"""

    def __init__(self, w, bias=None, bn_in=None, bn_out=None, stride=1, pad=0, in_relu=False, out_relu=False, device="cuda"):
        w = w.detach().to(torch.float32)
        cout = w.shape[0]
        b = None if bias is None else bias.detach().to(torch.float32).clone()
        if bn_out is not None:                            # conv -> BatchNorm (-> ReLU): folded into the weights and the bias
            s, t = bn_out
            w = w * s.view(-1, 1, 1, 1)
            b = t.clone() if b is None else b * s + t
        self.w = w.permute(2, 3, 1, 0).contiguous().to(device)
        self.bias = None if b is None else b.contiguous().to(device)
        self.scale, self.shift = (None, None) if bn_in is None else (bn_in[0].contiguous().to(device), bn_in[1].contiguous().to(device))
        self.KH, self.KW, self.cin, self.cout = int(w.shape[2]), int(w.shape[3]), int(w.shape[1]), int(cout)
        self.stride, self.pad = int(stride), int(pad)
        self.flags = (_lib.CONV_IN_RELU if in_relu else 0) | (_lib.CONV_OUT_RELU if out_relu else 0)

    def __call__(self, x, x_off, x_cin, out, out_off, accumulate=False):
        """x, out: NHWC buffers; the convolution reads channels [x_off, x_off + cin) of x and writes [out_off, out_off + cout) of out"""
        N, H, W, Cs = x.shape
        assert x_cin == self.cin and x.is_cuda and out.is_cuda and x.dtype == out.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
        ws = _workspace(x.device)
        rc = _lib.lib().vhap_conv2d_nhwc_ws(x.data_ptr() + 4 * x_off, Cs, N, H, W, self.cin, _p(self.w), _p(self.bias), _p(self.scale), _p(self.shift),
                                            self.KH, self.KW, self.stride, self.pad, out.data_ptr() + 4 * out_off, out.shape[3], self.cout,
                                            _p(ws), ws.numel(), self.flags | (_lib.CONV_ACCUMULATE if accumulate else 0), _stream())
        _chk(rc, "vhap_conv2d_nhwc_ws")
        return out


def _bn(sd, prefix, eps=1e-5):
    g, b, m, v = (sd[prefix + k].detach().to(torch.float32) for k in (".weight", ".bias", ".running_mean", ".running_var"))
    s = g / torch.sqrt(v + eps)
    return s, b - m * s


class _Block:
    """pre-activation residual block (oracle/fan_ref.py::ConvBlock): three launches into the slices of the concatenated output, the skip added by
    a fourth (the projection, accumulating) or by vhap_nhwc_add (identity)"""

    def __init__(self, sd, prefix, cin, cout, device):
        c = lambda name, bn: _Conv(sd[f"{prefix}.{name}.weight"], bn_in=_bn(sd, f"{prefix}.{bn}"), pad=1, in_relu=True, device=device)
        self.cin, self.cout = cin, cout
        self.c1, self.c2, self.c3 = c("conv1", "bn1"), c("conv2", "bn2"), c("conv3", "bn3")
        self.ds = None
        if cin != cout:
            self.ds = _Conv(sd[f"{prefix}.downsample.2.weight"], bn_in=_bn(sd, f"{prefix}.downsample.0"), in_relu=True, device=device)

    def __call__(self, x):
        N, H, W, _ = x.shape
        out = torch.empty(N, H, W, self.cout, dtype=torch.float32, device=x.device)
        h, q = self.cout // 2, self.cout // 4
        self.c1(x, 0, self.cin, out, 0)
        self.c2(out, 0, h, out, h)
        self.c3(out, h, q, out, h + q)
        if self.ds is not None:
            self.ds(x, 0, self.cin, out, 0, accumulate=True)
        else:
            _chk(_lib.lib().vhap_nhwc_add(_p(out), _p(x), 0, out.numel(), _p(out), _stream()), "vhap_nhwc_add")
        return out


def _avgpool2(x):
    N, H, W, C = x.shape
    out = torch.empty(N, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    _chk(_lib.lib().vhap_nhwc_avgpool2(_p(x), N, H, W, C, _p(out), _stream()), "vhap_nhwc_avgpool2")
    return out


class FAN2D:
    """The network of oracle/fan_ref.py::FAN from the package's state dict.  forward(x [N,3,256,256] float32 on the GPU, RGB in [0, 1]) ->
    list of `num_modules` heat-map tensors [N,68,64,64] (channel-first VIEWS of the channel-last buffers the kernels write)."""

    def __init__(self, state_dict, num_modules=4, n_landmarks=68, device="cuda"):
        if not str(device).startswith("cuda"):
            raise RuntimeError("FAN2D runs on the HIP device only (there is no CPU path)")
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        self.device, self.num_modules, self.L = device, int(num_modules), int(n_landmarks)
        self.stem = _Conv(sd["conv1.weight"], sd["conv1.bias"], bn_out=_bn(sd, "bn1"), stride=2, pad=3, out_relu=True, device=device)
        self.b2, self.b3, self.b4 = _Block(sd, "conv2", 64, 128, device), _Block(sd, "conv3", 128, 128, device), _Block(sd, "conv4", 128, 256, device)
        self.hg, self.top, self.last, self.l, self.bl, self.al = [], [], [], [], [], []
        for i in range(self.num_modules):
            blocks = {}
            for level in range(4, 0, -1):
                for name in ("b1", "b2", "b3"):
                    blocks[f"{name}_{level}"] = _Block(sd, f"m{i}.{name}_{level}", 256, 256, device)
            blocks["b2_plus_1"] = _Block(sd, f"m{i}.b2_plus_1", 256, 256, device)
            self.hg.append(blocks)
            self.top.append(_Block(sd, f"top_m_{i}", 256, 256, device))
            self.last.append(_Conv(sd[f"conv_last{i}.weight"], sd[f"conv_last{i}.bias"], bn_out=_bn(sd, f"bn_end{i}"), out_relu=True, device=device))
            self.l.append(_Conv(sd[f"l{i}.weight"], sd[f"l{i}.bias"], device=device))
            if i < self.num_modules - 1:
                self.bl.append(_Conv(sd[f"bl{i}.weight"], sd[f"bl{i}.bias"], device=device))
                self.al.append(_Conv(sd[f"al{i}.weight"], sd[f"al{i}.bias"], device=device))

    def _hourglass(self, B, level, x):
        up1 = B[f"b1_{level}"](x)
        low1 = B[f"b2_{level}"](_avgpool2(x))
        low2 = self._hourglass(B, level - 1, low1) if level > 1 else B["b2_plus_1"](low1)
        low3 = B[f"b3_{level}"](low2)
        N, H, W, C = up1.shape
        out = torch.empty_like(up1)
        _chk(_lib.lib().vhap_nhwc_upsample2_add(_p(up1), _p(low3), N, H, W, C, _p(out), _stream()), "vhap_nhwc_upsample2_add")
        return out

    def _conv_new(self, conv, x):
        N, H, W, C = x.shape
        Ho, Wo = (H + 2 * conv.pad - conv.KH) // conv.stride + 1, (W + 2 * conv.pad - conv.KW) // conv.stride + 1
        return conv(x, 0, C, torch.empty(N, Ho, Wo, conv.cout, dtype=torch.float32, device=x.device), 0)

    @torch.no_grad()
    def forward(self, x):
        if not (torch.is_tensor(x) and x.is_cuda):
            raise RuntimeError("FAN2D.forward needs a tensor on the HIP device (there is no CPU path)")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 64 or x.shape[3] % 64:
            raise ValueError("FAN2D.forward takes [N,3,H,W] with H, W multiples of 64 (the network halves the resolution six times)")
        x = x.to(torch.float32).permute(0, 2, 3, 1).contiguous()                 # channel-last
        x = self._conv_new(self.stem, x)
        x = _avgpool2(self.b2(x))
        x = self.b4(self.b3(x))
        previous, outputs = x, []
        for i in range(self.num_modules):
            ll = self.top[i](self._hourglass(self.hg[i], 4, previous))
            ll = self._conv_new(self.last[i], ll)
            hm = self._conv_new(self.l[i], ll)
            outputs.append(hm.permute(0, 3, 1, 2))
            if i < self.num_modules - 1:
                a, b = self._conv_new(self.bl[i], ll), self._conv_new(self.al[i], hm)
                nxt = torch.empty_like(previous)
                _chk(_lib.lib().vhap_nhwc_add(_p(previous), _p(a), _p(b), previous.numel(), _p(nxt), _stream()), "vhap_nhwc_add")
                previous = nxt
        return outputs

    __call__ = forward


# ---- pre- / post-processing of the package, restated from its published behaviour (host side: a few hundred numbers per face) ----
def _crop_matrix(center, scale, resolution):
    h = 200.0 * scale
    t = np.eye(3)
    t[0, 0] = t[1, 1] = resolution / h
    t[0, 2] = resolution * (-center[0] / h + 0.5)
    t[1, 2] = resolution * (-center[1] / h + 0.5)
    return t


def box_center_scale(box, reference_scale=195.0):
    center = np.array([box[2] - (box[2] - box[0]) / 2.0, box[3] - (box[3] - box[1]) / 2.0], np.float64)
    center[1] -= (box[3] - box[1]) * 0.12
    return center, (box[2] - box[0] + box[3] - box[1]) / reference_scale


def crop_face(img, center, scale, resolution=256, device="cuda"):
    """img [H,W,3] uint8 -> [3,resolution,resolution] float32 in [0, 1] on the device: the window of 200 * scale pixels around `center`, zero
    outside the image, resized with half-pixel bilinear sampling (what the package's cv2.resize(INTER_LINEAR) computes, up to its 11-bit
    fixed-point weights)."""
    inv = np.linalg.inv(_crop_matrix(center, scale, resolution))
    ul = (inv @ np.array([1.0, 1.0, 1.0]))[:2].astype(np.int64)
    br = (inv @ np.array([float(resolution), float(resolution), 1.0]))[:2].astype(np.int64)
    ht, wd = img.shape[:2]
    win = np.zeros((int(br[1] - ul[1]), int(br[0] - ul[0]), 3), np.uint8)
    nx = (max(1, -ul[0] + 1), min(br[0], wd) - ul[0])
    ny = (max(1, -ul[1] + 1), min(br[1], ht) - ul[1])
    ox = (max(1, ul[0] + 1), min(br[0], wd))
    oy = (max(1, ul[1] + 1), min(br[1], ht))
    if nx[1] > nx[0] - 1 and ny[1] > ny[0] - 1:
        win[ny[0] - 1:ny[1], nx[0] - 1:nx[1]] = img[oy[0] - 1:oy[1], ox[0] - 1:ox[1], :3]
    t = torch.from_numpy(win).to(device).permute(2, 0, 1)[None].float()
    t = torch.nn.functional.interpolate(t, (resolution, resolution), mode="bilinear", align_corners=False)
    return (t[0] / 255.0).contiguous()


def heatmaps_to_points(hm, center=None, scale=None):
    """hm [N,L,R,R] (tensor or array) -> (heat-map pixels [N,L,2], image pixels [N,L,2] or None, peak values [N,L]): arg-max, a quarter pixel
    towards the higher neighbour in x and y, minus one half (1-based), then the inverse crop transform"""
    hm = hm.detach().cpu().numpy() if torch.is_tensor(hm) else np.asarray(hm)
    N, L, R, _ = hm.shape
    flat = hm.reshape(N, L, -1)
    idx = flat.argmax(-1)
    peak = np.take_along_axis(flat, idx[..., None], -1)[..., 0]
    px, py = idx % R, idx // R
    pts = np.stack([px, py], -1).astype(np.float64) + 1.0
    inner = (px > 0) & (px < R - 1) & (py > 0) & (py < R - 1)
    n, l = np.nonzero(inner)
    dx = hm[n, l, py[n, l], px[n, l] + 1] - hm[n, l, py[n, l], px[n, l] - 1]
    dy = hm[n, l, py[n, l] + 1, px[n, l]] - hm[n, l, py[n, l] - 1, px[n, l]]
    pts[n, l, 0] += np.sign(dx) * 0.25
    pts[n, l, 1] += np.sign(dy) * 0.25
    pts -= 0.5
    if center is None:
        return pts, None, peak
    inv = np.linalg.inv(_crop_matrix(center, scale, R))
    img = pts @ inv[:2, :2].T + inv[:2, 2]
    return pts, img, peak


class LandmarkDetectorFA:
    """vhap/util/landmark_detector_fa.py:25-78 over FAN2D.  `weights`: the package's state dict, or a path torch.load can read.
    `face_detector`: callable(img [H,W,3] uint8) -> list of (x1, y1, x2, y2, score); None: the whole frame is the box (the package's `sfd`
    detector is a second third-party network, not built)."""

    def __init__(self, weights, face_detector=None, flip_input=True, device="cuda", num_modules=4):
        if isinstance(weights, (str, os.PathLike)):
            weights = torch.load(weights, map_location="cpu")
            weights = weights.get("state_dict", weights) if isinstance(weights, dict) else weights.state_dict()
        self.net = FAN2D(weights, num_modules=num_modules, device=device)
        self.face_detector, self.flip_input, self.device = face_detector, bool(flip_input), device

    def landmarks_from_boxes(self, imgs, boxes):
        """68 image-space landmarks per (frame, box): the crops of all frames AND their mirror images (flip_input) go through the network as ONE
        batch -- the package runs two passes per frame; a sample's result does not depend on its batch."""
        n = len(imgs)
        cs = [box_center_scale(b) for b in boxes]
        x = torch.stack([crop_face(img, c, s, 256, self.device) for img, (c, s) in zip(imgs, cs)])
        if self.flip_input:
            x = torch.cat([x, torch.flip(x, dims=[3])])
        hm = self.net(x)[-1]
        if self.flip_input:                                # the package averages the heat maps of the mirrored pass (left / right swapped back)
            hm = hm[:n] + torch.flip(hm[n:], dims=[3])[:, torch.as_tensor(MIRROR_68, device=hm.device)]
        return [heatmaps_to_points(hm[i:i + 1], c, s)[1][0] for i, (c, s) in enumerate(cs)]

    def landmarks_from_box(self, img, box):
        return self.landmarks_from_boxes([img], [box])[0]

    def _boxes(self, img):
        h, w = img.shape[:2]
        boxes = [np.array([0.0, 0.0, float(w), float(h), 1.0])] if self.face_detector is None else [np.asarray(b, np.float64) for b in self.face_detector(img)]
        if len(boxes) > 1:
            boxes = [boxes[int(np.argmax(np.array(boxes)[:, -1]))]]
        return boxes

    @staticmethod
    def _normalised(img, box, lmks):
        h, w = img.shape[:2]
        lmks = np.concatenate([lmks, np.ones_like(lmks[:, :1])], axis=1)
        lmks[:, 2:] = 0.0 if (lmks[:, :2] == -1).sum() > 0 else 1.0
        lmks[:, 0] /= w
        lmks[:, 1] /= h
        box = box.copy()
        box[[0, 2]] /= w
        box[[1, 3]] /= h
        return [box], lmks

    def detect_images(self, imgs):
        """detect_single_image over several frames with ONE pass of the network: -> list of (bbox list, lmks [68,3])"""
        imgs = [np.asarray(im) for im in imgs]
        boxes = [self._boxes(im) for im in imgs]
        have = [i for i, b in enumerate(boxes) if len(b)]
        pts = self.landmarks_from_boxes([imgs[i] for i in have], [boxes[i][0] for i in have]) if have else []
        out = [([], np.zeros([68, 3]) - 1) for _ in imgs]           # no face found: all -1, like the reference
        for i, p in zip(have, pts):
            out[i] = self._normalised(imgs[i], boxes[i][0], p)
        return out

    def detect_single_image(self, img):
        """-> (bbox list, lmks [68,3]): x / w, y / h, confidence -- and all -1 when no face was found, like the reference"""
        return self.detect_images([img])[0]


def annotate_landmarks(dataset, detector, property_name="landmark2d/face-alignment", batch_frames=8):
    """vhap/util/landmark_detector_fa.py:143-190 with `detector` (a LandmarkDetectorFA of this module): every item of a reference dataset
    (items with 'rgb' [H,W,3] uint8, 'timestep_id', 'camera_id'; `get_property_path(name, camera_id=)`) -> one npz per camera with
    `face_landmark_2d [T,68,3]` and `bounding_box [T,5]`, timesteps in sorted order.  -> {camera_id: path}"""
    landmarks, bboxes = {}, {}
    pending = []

    def flush():
        imgs = [np.asarray(it["rgb"]) for it in pending]
        many = getattr(detector, "detect_images", None)
        for item, (bbox, lmks) in zip(pending, many(imgs) if many is not None else [detector.detect_single_image(im) for im in imgs]):
            cam, ts = item["camera_id"], item["timestep_id"]
            landmarks.setdefault(cam, {})[ts] = lmks
            bboxes.setdefault(cam, {})[ts] = bbox[0] if len(bbox) > 0 else np.zeros(5) - 1
        pending.clear()
    for i in range(len(dataset)):
        item = dataset[i]
        if item is None:
            continue
        pending.append(item)
        if len(pending) >= batch_frames:                   # (the reference goes frame by frame; the network fills the chip from ~8 frames = 16 crops on)
            flush()
    if pending:
        flush()
    paths = {}
    for cam, per_ts in landmarks.items():
        order = sorted(per_ts.keys())
        out = {"bounding_box": np.concatenate([np.asarray(bboxes[cam][t])[None] for t in order], 0),
               "face_landmark_2d": np.concatenate([np.asarray(per_ts[t])[None] for t in order], 0)}
        path = dataset.get_property_path(property_name, camera_id=cam)
        os.makedirs(os.path.dirname(str(path)), exist_ok=True)
        np.savez(path, **out)
        paths[cam] = path
    return paths
