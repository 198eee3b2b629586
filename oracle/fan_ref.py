"""TEST INFRASTRUCTURE (never imported by the product path): torch restatement of the 2-D landmark network the reference's detector runs.

The reference (vhap/util/landmark_detector_fa.py:41-46, called from vhap/model/tracker.py:1263-1277) instantiates
`face_alignment.FaceAlignment(LandmarksType.TWO_HALF_D, face_detector='sfd', flip_input=True)` -- the third-party package `face_alignment`
(pyproject.toml of the reference: `face-alignment`, no version pin; absent from /root/reference and from this image), whose network is the
published FAN: A. Bulat, G. Tzimiropoulos, "How far are we from solving the 2D & 3D face alignment problem?", ICCV 2017 -- a 7x7 / stride-2
stem, three pre-activation residual blocks with a hierarchical (1/2, 1/4, 1/4) channel split, and four stacked depth-4 hourglasses with
intermediate supervision, 68 heatmaps of 64 x 64 per stack.  Parameter names follow the package's state dict, so that its weights -- a
third-party download this build never had -- load into both this restatement and the product (vhap_amd/landmarks.py).

PARITY UNPINNED against the package itself (absent): the product is pinned on THIS restatement with seeded random weights
(tests/test_landmarks.py), the restatement on the published description and on the call shapes of the reference's own call site."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def conv3x3(cin, cout):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=False)


class ConvBlock(nn.Module):
    """pre-activation residual block: three 3x3 convolutions to C/2, C/4, C/4 channels, concatenated, plus the (projected) input"""

    def __init__(self, cin, cout):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv1 = conv3x3(cin, cout // 2)
        self.bn2 = nn.BatchNorm2d(cout // 2)
        self.conv2 = conv3x3(cout // 2, cout // 4)
        self.bn3 = nn.BatchNorm2d(cout // 4)
        self.conv3 = conv3x3(cout // 4, cout // 4)
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(nn.BatchNorm2d(cin), nn.ReLU(True), nn.Conv2d(cin, cout, kernel_size=1, stride=1, bias=False))

    def forward(self, x):
        out1 = self.conv1(F.relu(self.bn1(x)))
        out2 = self.conv2(F.relu(self.bn2(out1)))
        out3 = self.conv3(F.relu(self.bn3(out2)))
        res = x if self.downsample is None else self.downsample(x)
        return torch.cat((out1, out2, out3), 1) + res


class HourGlass(nn.Module):
    def __init__(self, depth=4, features=256):
        super().__init__()
        self.depth, self.features = depth, features
        self._make(depth)

    def _make(self, level):
        f = self.features
        self.add_module(f"b1_{level}", ConvBlock(f, f))
        self.add_module(f"b2_{level}", ConvBlock(f, f))
        if level > 1:
            self._make(level - 1)
        else:
            self.add_module(f"b2_plus_{level}", ConvBlock(f, f))
        self.add_module(f"b3_{level}", ConvBlock(f, f))

    def _fwd(self, level, x):
        up1 = self._modules[f"b1_{level}"](x)
        low1 = self._modules[f"b2_{level}"](F.avg_pool2d(x, 2, stride=2))
        low2 = self._fwd(level - 1, low1) if level > 1 else self._modules[f"b2_plus_{level}"](low1)
        low3 = self._modules[f"b3_{level}"](low2)
        return up1 + F.interpolate(low3, scale_factor=2, mode="nearest")

    def forward(self, x):
        return self._fwd(self.depth, x)


class FAN(nn.Module):
    def __init__(self, num_modules=4, n_landmarks=68):
        super().__init__()
        self.num_modules = num_modules
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = ConvBlock(64, 128)
        self.conv3 = ConvBlock(128, 128)
        self.conv4 = ConvBlock(128, 256)
        for i in range(num_modules):
            self.add_module(f"m{i}", HourGlass(4, 256))
            self.add_module(f"top_m_{i}", ConvBlock(256, 256))
            self.add_module(f"conv_last{i}", nn.Conv2d(256, 256, kernel_size=1, stride=1, padding=0))
            self.add_module(f"bn_end{i}", nn.BatchNorm2d(256))
            self.add_module(f"l{i}", nn.Conv2d(256, n_landmarks, kernel_size=1, stride=1, padding=0))
            if i < num_modules - 1:
                self.add_module(f"bl{i}", nn.Conv2d(256, 256, kernel_size=1, stride=1, padding=0))
                self.add_module(f"al{i}", nn.Conv2d(n_landmarks, 256, kernel_size=1, stride=1, padding=0))

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.avg_pool2d(self.conv2(x), 2, stride=2)
        x = self.conv4(self.conv3(x))
        previous, outputs = x, []
        for i in range(self.num_modules):
            ll = self._modules[f"top_m_{i}"](self._modules[f"m{i}"](previous))
            ll = F.relu(self._modules[f"bn_end{i}"](self._modules[f"conv_last{i}"](ll)))
            tmp = self._modules[f"l{i}"](ll)
            outputs.append(tmp)
            if i < self.num_modules - 1:
                previous = previous + self._modules[f"bl{i}"](ll) + self._modules[f"al{i}"](tmp)
        return outputs


def random_fan(seed=0, num_modules=4, dtype=torch.float32):
    """A FAN with seeded random weights AND non-trivial BatchNorm statistics (running mean / variance, affine), in eval mode."""
    g = torch.Generator().manual_seed(seed)
    net = FAN(num_modules).to(dtype)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.0 / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.3 * torch.rand(m.running_var.shape, generator=g))
    return net.eval()


# ---- the package's pre- / post-processing, restated from its published behaviour ----
def transform_point(point, center, scale, resolution, invert=False):
    """pixel (1-based in the crop) <-> image coordinates of a crop of `resolution` pixels around `center`, 200 * scale image pixels wide"""
    h = 200.0 * scale
    t = np.eye(3)
    t[0, 0] = t[1, 1] = resolution / h
    t[0, 2] = resolution * (-center[0] / h + 0.5)
    t[1, 2] = resolution * (-center[1] / h + 0.5)
    if invert:
        t = np.linalg.inv(t)
    return (t @ np.array([point[0], point[1], 1.0]))[:2]


def box_center_scale(box, reference_scale=195.0):
    """(x1, y1, x2, y2[, score]) of the face detector -> crop centre (shifted up by 12 % of the box height) and scale"""
    center = np.array([box[2] - (box[2] - box[0]) / 2.0, box[3] - (box[3] - box[1]) / 2.0])
    center[1] = center[1] - (box[3] - box[1]) * 0.12
    scale = (box[2] - box[0] + box[3] - box[1]) / reference_scale
    return center, scale


def heatmaps_to_points(hm, center=None, scale=None):
    """hm [N, L, R, R] (numpy) -> (points in heatmap pixels [N, L, 2], points in image pixels or None): per map the arg-max, moved a quarter of a
    pixel towards the higher neighbour in x and in y, 1-based, minus one half; then the inverse crop transform."""
    N, L, R, _ = hm.shape
    flat = hm.reshape(N, L, -1)
    idx = flat.argmax(-1)
    pts = np.stack([idx % R, idx // R], -1).astype(np.float64) + 1.0
    for n in range(N):
        for l in range(L):
            px, py = int(pts[n, l, 0]) - 1, int(pts[n, l, 1]) - 1
            if 0 < px < R - 1 and 0 < py < R - 1:
                d = np.array([hm[n, l, py, px + 1] - hm[n, l, py, px - 1], hm[n, l, py + 1, px] - hm[n, l, py - 1, px]])
                pts[n, l] += np.sign(d) * 0.25
    pts -= 0.5
    if center is None:
        return pts, None
    img = np.zeros_like(pts)
    for n in range(N):
        for l in range(L):
            img[n, l] = transform_point(pts[n, l], center, scale, R, invert=True)
    return pts, img
