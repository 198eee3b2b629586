"""Single-GPU timings of the pieces the sharded texture update adds / removes (DESIGN section 6's estimate is built from these):
the pyramid fold into level 0, the finish + Adam pass on the whole texture and on a 1/N row strip, the strip's staging copy."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vhap_amd import _lib

T, N = 2048, 8
L = _lib.lib()
dev = "cuda"
f = lambda *s: torch.randn(*s, device=dev) * 0.01
albedo, extra, mask = f(T, T, 3), f(3, T, T), (torch.rand(T, T, device=dev) > 0.9).to(torch.uint8)
nm = L.vhap_texture_mip_floats(1, T, T, 3)
d_tex, d_mips, d_extra, m, v = f(T, T, 3), f(nm), f(3, T, T), torch.zeros(3, T, T, device=dev), torch.zeros(3, T, T, device=dev)
ones, lr, step = torch.ones(8, device=dev), torch.full((1,), 1e-3, device=dev), torch.ones(1, dtype=torch.int32, device=dev)
p = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
rows = T // N

def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

full = lambda: _lib.check(L.vhap_tex_prep_bwd_adam(p(albedo), p(extra), p(mask), p(d_tex), p(d_mips), 11, p(ones), T, 1e-3, 1e-3, p(d_extra), p(m), p(v),
                                                  p(lr), p(step), 0.9, 0.999, 1e-8, 0, st), "full")
strip = lambda: _lib.check(L.vhap_tex_prep_bwd_adam_rows(p(albedo), p(extra), p(mask), p(d_tex[:rows]), p(ones), T, 3 * rows, rows, 1e-3, 1e-3, p(d_extra),
                                                        p(m), p(v), p(lr), p(step), 0.9, 0.999, 1e-8, 0, st), "strip")
fold = lambda: _lib.check(L.vhap_texture_mip_fold(p(d_tex), p(d_mips), 1, T, T, 3, 0, st), "fold")
foldg = lambda: _lib.check(L.vhap_texture_mip_fold_gather(p(d_tex), p(d_mips), T, T, 3, st), "fold_gather")
# (same numbers: the cascade on one copy, the gather on another)
a_, b_, m1, m2 = d_tex.clone(), d_tex.clone(), d_mips.clone(), d_mips.clone()
_lib.check(L.vhap_texture_mip_fold(p(a_), p(m1), 1, T, T, 3, 0, st), "fold"); _lib.check(L.vhap_texture_mip_fold_gather(p(b_), p(m2), T, T, 3, st), "fold_gather")
print("gather fold vs cascade fold: max rel diff", float((a_ - b_).abs().max() / a_.abs().max()))
stage = lambda: [extra[c, 3 * rows:4 * rows].clone() for c in range(3)]
print(f"T = {T}, N = {N}: finish + Adam whole texture {timed(full):.1f} us; rows {rows} (1/{N}) {timed(strip):.1f} us; pyramid fold into level 0: cascade {timed(fold):.1f} us, one gathering pass {timed(foldg):.1f} us; "
      f"staging copy of the strip's three planes {timed(stage):.1f} us")
