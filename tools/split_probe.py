"""Experiment: the RI-fwd raster kernel as TWO concurrent launches -- blocks with candidates (compute) and blocks without (pure stores) --
on two streams, against the one-launch form.  (debug flags 32 / 64 of raster_kernel)"""
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
ws, nbytes, cap, _ = ctx.acquire(B, tri.shape[0], H, W, pos.device)
E_ = lambda *s: torch.empty(*s, device="cuda")
rast, db, nrm, texc, texd = E_(B, H, W, 4), E_(B, H, W, 4), E_(B, H, W, 3), E_(B, H, W, 2), E_(B, H, W, 4)
s1, s2 = _lib.private_stream("side"), _lib.private_stream("side2")
p = lambda t: t.data_ptr()

def call(flags, stream, dbg):
    _lib.debug_set_flags(dbg)
    _lib.check(L.vhap_raster_interp_fwd(p(pos), p(tri), p(vn), p(uv), p(tri_uv), B, pos.shape[1], uv.shape[0], tri.shape[0], H, W, p(rast), p(db), p(nrm),
                                        p(texc), p(texd), p(ws), nbytes, cap, flags, stream.cuda_stream), "ri")

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
        e0.record()
        for _ in range(n): fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

BIN_ONLY, PREBINNED, CLEAN = 2, 4, 1
def one():
    call(CLEAN | BIN_ONLY, s1, 0)
    call(CLEAN | PREBINNED, s1, 0)
def split():
    call(CLEAN | BIN_ONLY, s1, 0)
    ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
    call(CLEAN | PREBINNED, s1, 32)        # compute blocks
    call(CLEAN | PREBINNED, s2, 64)        # background blocks
    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
def only(dbg):
    return lambda: (call(CLEAN | BIN_ONLY, s1, 0), call(CLEAN | PREBINNED, s1, dbg))
print(f"one launch: {timed(one):.1f} us   split (compute || background): {timed(split):.1f} us   compute blocks only: {timed(only(32)):.1f} us   "
      f"background blocks only: {timed(only(64)):.1f} us")
_lib.debug_set_flags(0)
