"""Timing of the landmark network's forward (vhap_amd.landmarks.FAN2D: ~200 vhap_conv2d_nhwc launches on the matrix cores) at the detector's shape --
2 x 3 x 256 x 256 (a face and its mirror image: flip_input) -- with seeded random weights; FLOPs counted from the convolution shapes.
usage: python tools/fan_bench.py [--batch 2] [--reps 10] [--torch]  (--torch: the same graph through torch's own convolution on this GPU, for scale)"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fan_ref                      # noqa: E402  (random weights + the torch graph: test infrastructure, used here as the workload generator)
from vhap_amd import landmarks as LM            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--torch", action="store_true")
ap.add_argument("--no-split", action="store_true", help="no split-K workspace: every convolution as one launch")
a = ap.parse_args()
if os.environ.get("VHAP_DEBUG"):                       # the library's A/B switches (8388608: 64-pixel workgroups everywhere)
    from vhap_amd import _lib
    _lib.debug_set_flags(int(os.environ["VHAP_DEBUG"]))
net = fan_ref.random_fan(seed=0, num_modules=4)
flops = [0]


def hook(m, inp, out):
    if isinstance(m, torch.nn.Conv2d):
        flops[0] += 2 * out.numel() * m.in_channels * m.kernel_size[0] * m.kernel_size[1]


hs = [m.register_forward_hook(hook) for m in net.modules()]
x = torch.rand(a.batch, 3, 256, 256)
with torch.no_grad():
    ref = net(x)
for h in hs:
    h.remove()
fan = LM.FAN2D(net.state_dict())
xd = x.cuda()
if a.no_split:
    LM._WS[str(xd.device)] = torch.empty(0, device=xd.device)
out = fan(xd)
torch.cuda.synchronize()
err = float((out[-1].cpu() - ref[-1]).abs().max() / ref[-1].abs().max())
ts = []
for _ in range(a.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fan(xd)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
t = sorted(ts)[len(ts) // 2]
print(f"FAN2D forward{' (no split-K)' if a.no_split else ''}, batch {a.batch} x 256^2, 4 stacks: {flops[0] / 1e9:.1f} GFLOP, {t * 1e3:.2f} ms (median of {a.reps}) = {flops[0] / t / 1e12:.1f} TFLOP/s fp32 "
      f"({a.batch / t:.0f} crops/s); last stack vs torch-CPU fp32: {err:.1e} of the max-norm")
if a.torch:
    g = net.cuda()
    with torch.no_grad():
        g(xd)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            g(xd)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    t2 = sorted(ts)[len(ts) // 2]
    print(f"torch (its own convolution library) on the same GPU: {t2 * 1e3:.2f} ms")
