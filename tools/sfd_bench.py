"""Timing of the face detector's network (vhap_amd.face_detector.S3FD: 27 vhap_conv2d_nhwc_ws launches on the matrix cores, 5 max-pools, 3 L2Norms) on whole frames
of the BASELINE configurations' sizes, seeded random weights; FLOPs counted from the convolution shapes.
usage: python tools/sfd_bench.py [--reps 10] [--torch]   (--torch: the same graph through torch's own convolution library on this GPU, for scale)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sfd_ref                       # noqa: E402  (random weights + the torch graph: test infrastructure, used here as the workload generator)
from vhap_amd import face_detector as FD         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--torch", action="store_true")
a = ap.parse_args()
if os.environ.get("VHAP_DEBUG"):                       # the library's A/B switches (8388608: 64-pixel workgroups everywhere)
    from vhap_amd import _lib
    _lib.debug_set_flags(int(os.environ["VHAP_DEBUG"]))
net = sfd_ref.random_s3fd(seed=0, face_bias=-9.0)        # (random heads score ~0.5 everywhere: the bias leaves a few dozen candidates, like a trained detector)
det = FD.SFDDetector(net.state_dict())
for (H, W) in ((512, 512), (550, 802), (1024, 1024)):
    img = np.random.default_rng(1).integers(0, 255, (H, W, 3), dtype=np.uint8)
    flops = [0]

    def hook(m, inp, out):
        if isinstance(m, torch.nn.Conv2d):
            flops[0] += 2 * out.numel() * m.in_channels * m.kernel_size[0] * m.kernel_size[1]
    x_cpu = sfd_ref.preprocess(img)
    if H <= 550:
        hs = [m.register_forward_hook(hook) for m in net.modules()]
        with torch.no_grad():
            ref = net(x_cpu)
        for h in hs:
            h.remove()
    else:
        ref = None
    x = x_cpu.permute(0, 2, 3, 1).contiguous().cuda()
    out = det.net(x)
    torch.cuda.synchronize()
    err = "-" if ref is None else f"{float((out[3][0].cpu().permute(0, 3, 1, 2) - ref[6]).abs().max() / ref[6].abs().max()):.1e}"
    ts, tw = [], []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        det.net(x)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        boxes = det(img)
        tw.append(time.perf_counter() - t0)
    t, t_all = sorted(ts)[len(ts) // 2], sorted(tw)[len(tw) // 2]
    gf = flops[0] / 1e9
    print(f"S3FD forward, frame {W} x {H}: " + (f"{gf:.1f} GFLOP, " if gf else "") + f"{t * 1e3:.2f} ms (median of {a.reps})" +
          (f" = {flops[0] / t / 1e12:.1f} TFLOP/s fp32" if gf else "") + f"; detect_from_image end to end (upload, network, threshold, decode, NMS): {t_all * 1e3:.2f} ms; "
          f"{len(det.candidates(img))} candidates -> {len(boxes)} boxes; fc7 head scores vs torch-CPU fp32: {err} of the max-norm")
    if a.torch:
        g = net.cuda()
        xg = x_cpu.cuda()
        with torch.no_grad():
            g(xg)
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                g(xg)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        print(f"    torch (its own convolution library) on the same GPU: {sorted(ts)[len(ts) // 2] * 1e3:.2f} ms")
        net.cpu()
