"""Ad-hoc timing of the RI-fwd pass (used during development; bench.py is the contract).  The input scene comes from the test-suite's
scene builder (tests/scenes.py); everything timed is the HIP path."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vhap_amd import ops
from vhap_amd.synthetic import make_flame_model
from tests.scenes import head_scene
from vhap_amd.render_hip import HipDiffRenderer

B, H, W = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 512, 512)))
model, topo = make_flame_model(0)
sc = head_scene(model, B, H, W, seed=0)
c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pos = sc["clip"].float().cuda()
tri, tri_uv = c(topo.faces.astype(np.int32)), c(topo.faces_uv.astype(np.int32))
uv = c(topo.verts_uvs.astype(np.float32))
vn = HipDiffRenderer(lighting_type="SH").cuda().compute_v_normals(sc["verts"].float().cuda(), torch.from_numpy(topo.faces.astype(np.int64)).cuda()).contiguous()
ctx = ops.RasterizeHipContext()
from vhap_amd import _lib
_lib.debug_set_flags(int(os.environ.get('VHAP_DEBUG', '0')))
for fused in (False, True):
    f = (lambda: ops.raster_interp_fwd(ctx, pos, tri, vn, uv, tri_uv, (H, W))) if fused else (lambda: ops.raster_fwd(ctx, pos, tri, (H, W)))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): out = f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    alg = B * (429364 + 68 * H * W) if fused else B * (82288 + 121728 + 32 * H * W)
    print(f"fused={fused} B={B} {H}x{W}: {ms*1e3:.1f} us/batch  alg {alg/1e6:.1f} MB -> {alg/ms/1e9:.3f} TB/s ({alg/ms/1e9/8*100:.1f}% of 8 TB/s)  cov={(out[0][...,3]>0).float().mean().item():.3f}")
