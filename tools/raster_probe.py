"""Binning and raster kernel of the RI-fwd pass timed separately (HIP events around BIN_ONLY / PREBINNED calls) for a list of debug flags.
    python tools/raster_probe.py [flags ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vhap_amd import ops, _lib
from vhap_amd.synthetic import make_flame_model
from tests.scenes import head_scene
from vhap_amd.render_hip import HipDiffRenderer

B, H, W = 16, 512, 512
model, topo = make_flame_model(0)
sc = head_scene(model, B, H, W, seed=0)
c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pos = sc["clip"].float().cuda()
tri, tri_uv = c(topo.faces.astype(np.int32)), c(topo.faces_uv.astype(np.int32))
uv = c(topo.verts_uvs.astype(np.float32))
vn = HipDiffRenderer(lighting_type="SH").cuda().compute_v_normals(sc["verts"].float().cuda(), torch.from_numpy(topo.faces.astype(np.int64)).cuda()).contiguous()
L = _lib.lib()
ctx = ops.RasterizeHipContext()
ws, nbytes, cap, wflags = ctx.acquire(B, tri.shape[0], H, W, pos.device)
E_ = lambda *s: torch.empty(*s, device="cuda")
rast, db, nrm, texc, texd = E_(B, H, W, 4), E_(B, H, W, 4), E_(B, H, W, 3), E_(B, H, W, 2), E_(B, H, W, 4)
st = _lib.private_stream("side")
p = lambda t: t.data_ptr()

def call(flags):
    _lib.check(L.vhap_raster_interp_fwd(p(pos), p(tri), p(vn), p(uv), p(tri_uv), B, pos.shape[1], uv.shape[0], tri.shape[0], H, W, p(rast), p(db), p(nrm),
                                        p(texc), p(texd), p(ws), nbytes, cap, flags, st.cuda_stream), "ri")

def run(dbg, n=30):
    _lib.debug_set_flags(dbg)
    tb = tr = 0.0
    with torch.cuda.stream(st):
        for i in range(n + 5):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(); call(wflags | 2); e1.record(); call(wflags | 4); e2.record()
            torch.cuda.synchronize()
            if i >= 5:
                tb += e0.elapsed_time(e1); tr += e1.elapsed_time(e2)
    _lib.debug_set_flags(0)
    return tb / n * 1e3, tr / n * 1e3

for d in [int(x) for x in sys.argv[1:]] or [0]:
    b_, r_ = run(d)
    print(f"debug={d}: binning {b_:.1f} us  raster<1> {r_:.1f} us  (each incl. ~3 us of event overhead)")
