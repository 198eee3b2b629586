"""TEST INFRASTRUCTURE (never imported by the product path): torch / numpy restatement of the face detector the reference's landmark detector runs first.

The reference (vhap/util/landmark_detector_fa.py:32,41-51) builds `face_alignment.FaceAlignment(..., face_detector='sfd', ...)` and calls
`self.fa.face_detector.detect_from_image(img)` -- the third-party package `face_alignment` (absent from /root/reference and from this image, like its
weights), whose `sfd` detector is S3FD: S. Zhang et al., "S3FD: Single Shot Scale-invariant Face Detector", ICCV 2017 -- a VGG-16 trunk (fc6 / fc7 as
convolutions, two extra stages), L2-normalised conv3_3 / conv4_3 / conv5_3, six detection heads at strides 4 ... 128 with ONE square anchor of
4 x stride per location, max-out background label on the first head.  Parameter names follow the package's state dict, so that its weights -- a
third-party download this build never had -- load into both this restatement and the product (vhap_amd/face_detector.py).  The post-processing (softmax,
0.05 pre-threshold, SSD box decoding with variances 0.1 / 0.2, greedy NMS at 0.3 with the +1 pixel convention, 0.5 score filter) is restated from the
package's published behaviour.

PARITY UNPINNED against the package itself (absent): the product is pinned on THIS restatement with seeded random weights (tests/test_face_detector.py)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

TRUNK = (("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool", ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool",
         ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "tap", "pool",
         ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "tap", "pool",
         ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), "tap", "pool")
HEADS = (("conv3_3_norm", 256, 4), ("conv4_3_norm", 512, 2), ("conv5_3_norm", 512, 2), ("fc7", 1024, 2), ("conv6_2", 512, 2), ("conv7_2", 256, 2))
BGR_MEAN = (104.0, 117.0, 123.0)


class L2Norm(nn.Module):
    def __init__(self, n_channels, scale=1.0):
        super().__init__()
        self.eps = 1e-10
        self.weight = nn.Parameter(torch.full((n_channels,), float(scale)))

    def forward(self, x):
        norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + self.eps
        return x / norm * self.weight.view(1, -1, 1, 1)


class S3FD(nn.Module):
    def __init__(self):
        super().__init__()
        for item in TRUNK:
            if not isinstance(item, str):
                self.add_module(item[0], nn.Conv2d(item[1], item[2], kernel_size=3, stride=1, padding=1))
        self.fc6 = nn.Conv2d(512, 1024, kernel_size=3, stride=1, padding=3)
        self.fc7 = nn.Conv2d(1024, 1024, kernel_size=1, stride=1, padding=0)
        self.conv6_1 = nn.Conv2d(1024, 256, kernel_size=1, stride=1, padding=0)
        self.conv6_2 = nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1)
        self.conv7_1 = nn.Conv2d(512, 128, kernel_size=1, stride=1, padding=0)
        self.conv7_2 = nn.Conv2d(128, 256, kernel_size=3, stride=2, padding=1)
        self.conv3_3_norm, self.conv4_3_norm, self.conv5_3_norm = L2Norm(256, 10), L2Norm(512, 8), L2Norm(512, 5)
        for name, cin, ncls in HEADS:
            self.add_module(f"{name}_mbox_conf", nn.Conv2d(cin, ncls, kernel_size=3, stride=1, padding=1))
            self.add_module(f"{name}_mbox_loc", nn.Conv2d(cin, 4, kernel_size=3, stride=1, padding=1))

    def forward(self, x):
        h, taps = x, []
        for item in TRUNK:
            if item == "pool":
                h = F.max_pool2d(h, 2, 2)
            elif item == "tap":
                taps.append(h)
            else:
                h = F.relu(self._modules[item[0]](h))
        h = F.relu(self.fc7(F.relu(self.fc6(h))))
        f7 = h
        h = F.relu(self.conv6_2(F.relu(self.conv6_1(h))))
        f6_2 = h
        f7_2 = F.relu(self.conv7_2(F.relu(self.conv7_1(h))))
        feats = [self.conv3_3_norm(taps[0]), self.conv4_3_norm(taps[1]), self.conv5_3_norm(taps[2]), f7, f6_2, f7_2]
        out = []
        for (name, _, ncls), f in zip(HEADS, feats):
            cls, reg = self._modules[f"{name}_mbox_conf"](f), self._modules[f"{name}_mbox_loc"](f)
            if ncls == 4:                                  # max-out background label: the largest of three background scores against the face score
                c = torch.chunk(cls, 4, 1)
                cls = torch.cat([torch.max(torch.max(c[0], c[1]), c[2]), c[3]], dim=1)
            out += [cls, reg]
        return out


def random_s3fd(seed=0, face_bias=0.0):
    """seeded random weights; `face_bias` is added to the face logit of every head (random heads score ~0.5 everywhere: a bias thins the candidates
    out to a testable number)"""
    g = torch.Generator().manual_seed(seed)
    net = S3FD()
    with torch.no_grad():
        for name, m in net.named_modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
                if name == "conv1_1":
                    m.weight.mul_(1.0 / 100.0)                    # (pixels minus the means are ~ +-100: keeps the deep heads' scores O(1))
                if name.endswith("_mbox_conf"):
                    m.bias[-1] += face_bias
            elif isinstance(m, L2Norm):
                m.weight.copy_(m.weight * (1.0 + 0.1 * torch.randn(m.weight.shape, generator=g)))
    return net.eval()


def preprocess(img):
    """[H,W,3] uint8 RGB -> [1,3,H,W] float32 BGR minus the channel means"""
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(img)[..., ::-1])).to(torch.float32) - torch.tensor(BGR_MEAN)
    return x.permute(2, 0, 1)[None].contiguous()


def decode(loc, priors, variances=(0.1, 0.2)):
    boxes = np.concatenate((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:], priors[:, 2:] * np.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def candidates(olist, pre_threshold=0.05):
    """the network's raw outputs (numpy, batch 1) -> [n,5] boxes (x1, y1, x2, y2, score): softmax over the two labels, every location whose face
    probability exceeds `pre_threshold`, ONE anchor of 4 x stride centred on the location"""
    out = []
    for i in range(len(olist) // 2):
        cls, reg = olist[2 * i], olist[2 * i + 1]
        e = np.exp(cls - cls.max(axis=1, keepdims=True))
        prob = (e / e.sum(axis=1, keepdims=True))[0, 1]
        stride = 2 ** (i + 2)
        for hy, wx in zip(*np.where(prob > pre_threshold)):
            priors = np.array([[stride / 2 + wx * stride, stride / 2 + hy * stride, stride * 4.0, stride * 4.0]])
            box = decode(reg[0, :, hy, wx].reshape(1, 4).astype(np.float64), priors)
            out.append(np.concatenate([box[0], [prob[hy, wx]]]))
    return np.array(out).reshape(-1, 5)


def nms(dets, thresh=0.3):
    if len(dets) == 0:
        return []
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1, yy1 = np.maximum(x1[i], x1[order[1:]]), np.maximum(y1[i], y1[order[1:]])
        xx2, yy2 = np.minimum(x2[i], x2[order[1:]]), np.minimum(y2[i], y2[order[1:]])
        w, h = np.maximum(0.0, xx2 - xx1 + 1), np.maximum(0.0, yy2 - yy1 + 1)
        ovr = w * h / (areas[i] + areas[order[1:]] - w * h)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


def detect(net, img, filter_threshold=0.5):
    """the package's detect_from_image: -> list of [x1, y1, x2, y2, score]"""
    with torch.no_grad():
        olist = [o.numpy() for o in net(preprocess(img))]
    dets = candidates(olist)
    if len(dets) == 0:
        return []
    dets = dets[nms(dets, 0.3)]
    return [d for d in dets if d[-1] > filter_threshold]
