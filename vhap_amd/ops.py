"""nvdiffrast-shaped operator surface on top of the C ABI (boundary b2 of SURVEY.md section 8(b)).

Same names, argument meaning and error behaviour as the four ops the reference binds from
`nvdiffrast.torch` (vhap/util/render_nvdiffrast.py:12,74,254,384,389,399,465):

    ctx = RasterizeHipContext()                      # dr.RasterizeCudaContext()
    rast, rast_db = rasterize(ctx, pos, tri, (H, W))
    out, out_da   = interpolate(attr, rast, tri, rast_db=None, diff_attrs=None)
    out           = texture(tex, uv, uv_da, filter_mode='linear-mipmap-linear')
    out           = antialias(color, rast, pos, tri)

Every op is a torch.autograd.Function whose forward/backward enqueue hand-written gfx950 kernels
through ctypes on torch's CURRENT stream.  torch only owns memory, streams and the autograd tape.
There is no eager/CPU fallback: CPU tensors raise, a missing library raises.
"""
import os

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional profiling hook: bench.py installs a callable (name, phase) -> None that records HIP events on
# torch's current stream right before ('begin') and after ('end') a C-ABI launch sequence.
PROFILE_HOOK = None


def _hook(name, phase):
    if PROFILE_HOOK is not None:
        PROFILE_HOOK(name, phase)


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vhap_amd ops require tensors on a HIP device (no CPU fallback)")


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32 tensor, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _i32c(t):
    if t.dtype != torch.int32:
        raise TypeError(f"expected int32 tensor, got {t.dtype} (the reference passes tri.int())")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    if t is None:
        return 0
    if _lib.ACCESS is not None:          # (a step capture is recording which buffers its calls touch: _lib.AccessLog)
        _lib.ACCESS.touch(t)
    return t.data_ptr()


class RasterizeHipContext:
    """Stands in for dr.RasterizeCudaContext().  Holds only sizing policy (no device state, so it is
    safe to share between threads -- the reference's log_media thread may render concurrently)."""

    def __init__(self, pairs_per_triangle=8, pairs_per_block=8, persistent=True):
        self.pairs_per_triangle = pairs_per_triangle
        self.pairs_per_block = pairs_per_block
        # persistent=True: keep ONE zero-initialised workspace per (device, stream, shape).  The kernels leave it clean, so
        # the per-call memset is skipped (VHAP_RASTER_WS_CLEAN) and the address is stable under graph capture.  A context is
        # then tied to in-order use on one stream at a time; use persistent=False (or one context per thread) otherwise.
        self.persistent = persistent
        self._ws = {}

    def workspace(self, B, F, H, W, device):
        ws, nbytes, cap = self._workspace(B, F, H, W, device)
        return ws, nbytes, cap

    def acquire(self, B, F, H, W, device):
        """-> (workspace tensor, nbytes, capacity, flags)."""
        if not self.persistent:
            ws, nbytes, cap = self._workspace(B, F, H, W, device)
            return ws, nbytes, cap, 0
        key = (str(device), torch.cuda.current_stream(device).cuda_stream if torch.cuda.is_available() else 0, B, F, H, W)
        hit = self._ws.get(key)
        if hit is None:
            ws, nbytes, cap = self._workspace(B, F, H, W, device, zero=True)
            if len(self._ws) > 8:
                self._ws.clear()
            hit = self._ws[key] = (ws, nbytes, cap)
        return hit[0], hit[1], hit[2], 1

    def _workspace(self, B, F, H, W, device, zero=False):
        # capacity of the (triangle, 8x8-block) lists; exceeding it is still correct (brute-force path)
        nblk = ((int(H) + 7) // 8) * ((int(W) + 7) // 8)
        cap = int(B) * (int(F) * self.pairs_per_triangle + nblk * self.pairs_per_block)
        nbytes = _lib.lib().vhap_raster_workspace_bytes(B, F, H, W, cap)
        if nbytes == 0:
            raise ValueError(f"rasterize: dimensions out of range (B={B}, F={F}, H={H}, W={W})")
        alloc = torch.zeros if zero else torch.empty
        return alloc(nbytes, dtype=torch.uint8, device=device), nbytes, cap



# nvdiffrast's constructor names, so that `sys.modules['nvdiffrast.torch'] = vhap_amd.ops` serves the reference's
# render_nvdiffrast.py unmodified (render_nvdiffrast.py:74: dr.RasterizeGLContext() if use_opengl else dr.RasterizeCudaContext())
RasterizeCudaContext = RasterizeHipContext
RasterizeGLContext = RasterizeHipContext

def _check_raster_args(pos, tri, resolution):
    if pos.dim() != 3 or pos.shape[-1] != 4:
        raise ValueError("pos must have shape [B, V, 4] (instanced mode)")
    if tri.dim() != 2 or tri.shape[-1] != 3:
        raise ValueError("tri must have shape [F, 3]")
    H, W = int(resolution[0]), int(resolution[1])
    return H, W


POISON_OUTPUTS = False        # tests: raster outputs start as NaN, so that a pixel no workgroup stores cannot pass for a correct one


def _out(*shape, device=None):
    if POISON_OUTPUTS:
        return torch.full(shape, float("nan"), dtype=torch.float32, device=device)
    return torch.empty(*shape, dtype=torch.float32, device=device)


def raster_fwd(ctx, pos, tri, resolution, with_db=True):
    """Forward only (no autograd).  Returns rast [B,H,W,4], rast_db [B,H,W,4] or None."""
    _chk_cuda(pos, tri)
    H, W = _check_raster_args(pos, tri, resolution)
    pos, tri = _f32c(pos), _i32c(tri)
    B, V, _ = pos.shape
    F = tri.shape[0]
    ws, nbytes, cap, flags = ctx.acquire(B, F, H, W, pos.device)
    rast = _out(B, H, W, 4, device=pos.device)
    db = _out(B, H, W, 4, device=pos.device) if with_db else None
    rc = _lib.lib().vhap_raster_fwd(_p(pos), _p(tri), B, V, F, H, W, _p(rast), _p(db), _p(ws), nbytes, cap, flags, _stream())
    _lib.check(rc, "vhap_raster_fwd")
    return rast, db


def raster_interp_fwd(ctx, pos, tri, vnormal, uv, tri_uv, resolution):
    """Fused G-buffer pass (forward only): rast, rast_db, normal [B,H,W,3], texc [B,H,W,2], texd [B,H,W,4]."""
    _chk_cuda(pos, tri, vnormal, uv, tri_uv)
    H, W = _check_raster_args(pos, tri, resolution)
    pos, tri, vnormal, uv, tri_uv = _f32c(pos), _i32c(tri), _f32c(vnormal), _f32c(uv), _i32c(tri_uv)
    B, V, _ = pos.shape
    F = tri.shape[0]
    if vnormal.shape != (B, V, 3) or uv.dim() != 2 or uv.shape[1] != 2 or tri_uv.shape != tri.shape:
        raise ValueError("raster_interp_fwd: vnormal [B,V,3], uv [VT,2], tri_uv [F,3] expected")
    ws, nbytes, cap, flags = ctx.acquire(B, F, H, W, pos.device)
    dev = pos.device
    rast, db = _out(B, H, W, 4, device=dev), _out(B, H, W, 4, device=dev)
    normal, texc, texd = _out(B, H, W, 3, device=dev), _out(B, H, W, 2, device=dev), _out(B, H, W, 4, device=dev)
    _hook("raster_interp_fwd", "begin")
    rc = _lib.lib().vhap_raster_interp_fwd(_p(pos), _p(tri), _p(vnormal), _p(uv), _p(tri_uv), B, V, uv.shape[0], F, H, W,
                                           _p(rast), _p(db), _p(normal), _p(texc), _p(texd), _p(ws), nbytes, cap, flags, _stream())
    _hook("raster_interp_fwd", "end")
    _lib.check(rc, "vhap_raster_interp_fwd")
    return rast, db, normal, texc, texd


# ------------------------------------------------------------------------------------------------
# autograd Functions (b2 shim): rasterize / interpolate / texture / antialias
# ------------------------------------------------------------------------------------------------


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, glctx, pos, tri, resolution):
        rast, db = raster_fwd(glctx, pos, tri, resolution, with_db=True)
        ctx.save_for_backward(pos, tri, rast)
        ctx.res = (int(resolution[0]), int(resolution[1]))
        return rast, db

    @staticmethod
    def backward(ctx, d_rast, d_db):
        pos, tri, rast = ctx.saved_tensors
        H, W = ctx.res
        B, V, _ = pos.shape
        d_pos = torch.zeros_like(pos)
        d_rast = _f32c(d_rast)
        d_db = _f32c(d_db) if d_db is not None else None
        rc = _lib.lib().vhap_raster_bwd(_p(pos), _p(tri), _p(rast), _p(d_rast), _p(d_db), B, V, tri.shape[0], H, W,
                                        _p(d_pos), _stream())
        _lib.check(rc, "vhap_raster_bwd")
        return None, d_pos, None, None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """dr.rasterize(glctx, pos, tri, resolution) -> (rast [B,H,W,4], rast_db [B,H,W,4]).
    Differentiable w.r.t. pos through (u, v) and the pixel differentials, like nvdiffrast."""
    if ranges is not None:
        raise NotImplementedError("range mode is not used by the reference and not implemented")
    _chk_cuda(pos, tri)
    _check_raster_args(pos, tri, resolution)
    pos, tri = _f32c(pos), _i32c(tri)
    if pos.requires_grad and torch.is_grad_enabled():
        return _Rasterize.apply(glctx, pos, tri, tuple(resolution))
    return raster_fwd(glctx, pos, tri, resolution, with_db=True)


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, rast_db):
        AB, V, A = attr.shape
        B, H, W, _ = rast.shape
        out = torch.empty(B, H, W, A, dtype=torch.float32, device=rast.device)
        out_da = torch.empty(B, H, W, 2 * A, dtype=torch.float32, device=rast.device) if rast_db is not None else None
        rc = _lib.lib().vhap_interp_fwd(_p(attr), AB, _p(rast), _p(tri), _p(rast_db), B, H, W, V, tri.shape[0], A,
                                        _p(out), _p(out_da), _stream())
        _lib.check(rc, "vhap_interp_fwd")
        ctx.save_for_backward(attr, rast, tri, rast_db)
        ctx.has_da = rast_db is not None
        if out_da is None:
            out_da = torch.empty(0, device=rast.device)
            ctx.mark_non_differentiable(out_da)
        return out, out_da

    @staticmethod
    def backward(ctx, d_out, d_da):
        attr, rast, tri, rast_db = ctx.saved_tensors
        AB, V, A = attr.shape
        B, H, W, _ = rast.shape
        need_attr, need_rast, _, need_db = ctx.needs_input_grad
        d_attr = torch.zeros_like(attr) if need_attr else None
        d_rast = torch.empty_like(rast) if need_rast else None
        d_db = torch.empty_like(rast_db) if (ctx.has_da and need_db) else None
        d_out = _f32c(d_out)
        d_da = _f32c(d_da) if (ctx.has_da and d_da is not None) else None
        rc = _lib.lib().vhap_interp_bwd(_p(attr), AB, _p(rast), _p(tri), _p(rast_db), _p(d_out), _p(d_da), B, H, W, V,
                                        tri.shape[0], A, _p(d_attr), _p(d_rast), _p(d_db), _stream())
        _lib.check(rc, "vhap_interp_bwd")
        return d_attr, d_rast, None, d_db


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """dr.interpolate(attr, rast, tri, rast_db=None, diff_attrs=None) -> (out, out_da).
    attr [1|B,V,A]; diff_attrs must be None or 'all' (the only forms the reference uses)."""
    _chk_cuda(attr, rast, tri, rast_db)
    if attr.dim() != 3:
        raise ValueError("attr must have shape [1|B, V, A]")
    if diff_attrs not in (None, "all"):
        raise NotImplementedError("diff_attrs must be None or 'all'")
    attr, rast, tri = _f32c(attr), _f32c(rast), _i32c(tri)
    if attr.shape[0] not in (1, rast.shape[0]):
        raise ValueError("attr batch dimension must be 1 or match rast")
    use_db = rast_db is not None and diff_attrs == "all"
    db = _f32c(rast_db) if use_db else None
    out, out_da = _Interpolate.apply(attr, rast, tri, db)
    return out, (out_da if use_db else None)


def build_mips(tex):
    """Build the mip pyramid buffer (levels 1..L) of tex [TB,Ht,Wt,C]; no autograd."""
    TB, Ht, Wt, C = tex.shape
    n = _lib.lib().vhap_texture_mip_floats(TB, Ht, Wt, C)
    mips = torch.empty(max(n, 1), dtype=torch.float32, device=tex.device)
    rc = _lib.lib().vhap_texture_mip_build(_p(tex), TB, Ht, Wt, C, _p(mips), _stream())
    _lib.check(rc, "vhap_texture_mip_build")
    return mips


class _Texture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, uv_da):
        TB, Ht, Wt, C = tex.shape
        B, H, W, _ = uv.shape
        mips = build_mips(tex) if uv_da is not None else None
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=uv.device)
        rc = _lib.lib().vhap_texture_fwd(_p(tex), _p(mips), TB, Ht, Wt, C, _p(uv), _p(uv_da), B, H, W, _p(out), _stream())
        _lib.check(rc, "vhap_texture_fwd")
        ctx.save_for_backward(tex, uv, uv_da, mips)
        return out

    @staticmethod
    def backward(ctx, d_out):
        tex, uv, uv_da, mips = ctx.saved_tensors
        TB, Ht, Wt, C = tex.shape
        B, H, W, _ = uv.shape
        need_tex, need_uv, need_da = ctx.needs_input_grad
        d_tex = torch.zeros_like(tex) if need_tex else None
        d_mips = torch.zeros_like(mips) if (need_tex and mips is not None) else None
        d_uv = torch.empty_like(uv) if need_uv else None
        d_da = torch.empty_like(uv_da) if (need_da and uv_da is not None) else None
        d_out = _f32c(d_out)
        rc = _lib.lib().vhap_texture_bwd(_p(tex), _p(mips), TB, Ht, Wt, C, _p(uv), _p(uv_da), _p(d_out), B, H, W,
                                         _p(d_tex), _p(d_mips), _p(d_uv), _p(d_da), _stream())
        _lib.check(rc, "vhap_texture_bwd")
        if d_mips is not None:
            rc = _lib.lib().vhap_texture_mip_fold(_p(d_tex), _p(d_mips), TB, Ht, Wt, C, 0, _stream())
            _lib.check(rc, "vhap_texture_mip_fold")
        return d_tex, d_uv, d_da


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
    """dr.texture(tex [1|B,Ht,Wt,C], uv [B,H,W,2], uv_da [B,H,W,4], filter_mode=...) -> [B,H,W,C].
    Supported (= what the reference uses): boundary 'wrap'; 'linear-mipmap-linear' with uv_da, or 'linear'."""
    _chk_cuda(tex, uv, uv_da)
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if uv_da is not None else "linear"
    if filter_mode not in ("linear", "linear-mipmap-linear"):
        raise NotImplementedError(f"filter_mode {filter_mode!r} is not implemented")
    if boundary_mode != "wrap" or mip_level_bias is not None or mip is not None or max_mip_level is not None:
        raise NotImplementedError("only boundary_mode='wrap' with the full default mip chain is implemented")
    if tex.dim() != 4 or tex.shape[-1] > 4:
        raise ValueError("tex must have shape [1|B, Ht, Wt, C<=4]")
    if tex.shape[0] not in (1, uv.shape[0]):
        raise ValueError("tex batch dimension must be 1 or match uv")
    if tex.shape[0] > 1 and tex.stride(0) == 0:
        # the reference hands over ONE texture expanded to the batch (tracker.py:234 `.expand(B, ...)`): sample the single copy (TB = 1)
        # instead of materialising B of them (805 MB at B = 16, T = 2048); autograd sums the gradient over the expansion as before
        tex = tex[:1]
    tex, uv = _f32c(tex), _f32c(uv)
    da = None
    if filter_mode == "linear-mipmap-linear":
        if uv_da is None:
            raise ValueError("linear-mipmap-linear needs uv_da")
        da = _f32c(uv_da)
    return _Texture.apply(tex, uv, da)


_OPP_CACHE = {}


def opposite_table(tri):
    """Static edge -> opposite-vertex table for `tri` (cached per tensor storage / content hash)."""
    from .topology import build_opposite_table
    key = (tri.data_ptr(), tuple(tri.shape), str(tri.device))
    hit = _OPP_CACHE.get(key)
    if hit is None:
        opp = torch.from_numpy(build_opposite_table(tri.detach().cpu().numpy())).to(tri.device)
        _OPP_CACHE.clear() if len(_OPP_CACHE) > 8 else None
        _OPP_CACHE[key] = hit = (tri, opp)      # keep tri alive so data_ptr stays unique
    return hit[1]


class _Antialias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, opp, pos_nograd):
        B, H, W, C = color.shape
        V, F = pos.shape[1], tri.shape[0]
        out = torch.empty_like(color)
        work = torch.empty(_lib.lib().vhap_antialias_work_ints(B, H, W, F), dtype=torch.int32, device=color.device)
        rc = _lib.lib().vhap_antialias_fwd(_p(color), _p(rast), _p(pos), _p(tri), _p(opp), B, H, W, C, V, F, _p(out),
                                           _p(work), _stream())
        _lib.check(rc, "vhap_antialias_fwd")
        ctx.save_for_backward(color, rast, pos, tri, opp, work, pos_nograd)
        return out

    @staticmethod
    def backward(ctx, d_out):
        color, rast, pos, tri, opp, work, pos_nograd = ctx.saved_tensors
        B, H, W, C = color.shape
        V, F = pos.shape[1], tri.shape[0]
        need_color, _, need_pos = ctx.needs_input_grad[:3]
        d_out = _f32c(d_out)
        d_color = torch.empty_like(color) if need_color else None
        d_pos = torch.zeros_like(pos) if need_pos else None
        rc = _lib.lib().vhap_antialias_bwd(_p(color), _p(rast), _p(pos), _p(tri), _p(opp), _p(d_out), _p(work), _p(pos_nograd), B, H, W,
                                           C, V, F, _p(d_color), _p(d_pos), 0, _stream())
        _lib.check(rc, "vhap_antialias_bwd")
        return d_color, None, d_pos, None, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0, opp=None, pos_nograd_verts=None):
    """dr.antialias(color [B,H,W,C], rast, pos [B,V,4], tri) -> [B,H,W,C].  `pos_nograd_verts` (MI355X extension): uint8 [V],
    vertices that receive no silhouette gradient (the reference detaches them in `pos` beforehand)."""
    _chk_cuda(color, rast, pos, tri)
    if pos_gradient_boost != 1.0:
        raise NotImplementedError("pos_gradient_boost != 1 is not used by the reference")
    if color.shape[-1] not in (1, 3, 4):
        raise ValueError("antialias supports 1, 3 or 4 channels")
    color, rast, pos, tri = _f32c(color), _f32c(rast), _f32c(pos), _i32c(tri)
    if opp is None:
        opp = opposite_table(tri)
    return _Antialias.apply(color, rast, pos, tri, _i32c(opp), pos_nograd_verts)


class _RasterInterp(torch.autograd.Function):
    """Fused G-buffer pass: dr.rasterize + dr.interpolate(normals) + dr.interpolate(uv, 'all') in one launch;
    the backward chains the two interpolate backwards into the rasterize backward."""

    @staticmethod
    def forward(ctx, glctx, pos, tri, vnormal, uv, tri_uv, resolution, uv_nograd):
        rast, db, normal, texc, texd = raster_interp_fwd(glctx, pos, tri, vnormal, uv, tri_uv, resolution)
        ctx.set_materialize_grads(False)           # unused outputs (rast, rast_db) arrive as None, not as zero images
        ctx.save_for_backward(pos, tri, vnormal, uv, tri_uv, rast, db, uv_nograd)
        ctx.res = (int(resolution[0]), int(resolution[1]))
        return rast, db, normal, texc, texd

    @staticmethod
    def backward(ctx, d_rast, d_db, d_normal, d_texc, d_texd):
        pos, tri, vnormal, uv, tri_uv, rast, db, uv_nograd = ctx.saved_tensors
        H, W = ctx.res
        B, V, _ = pos.shape
        F = tri.shape[0]
        need_pos, need_n = ctx.needs_input_grad[1], ctx.needs_input_grad[3]
        c = lambda t: _f32c(t) if t is not None else None
        d_pos = torch.zeros_like(pos) if need_pos else None
        d_vn = torch.zeros_like(vnormal) if (need_n and d_normal is not None) else None
        _lib.check(_lib.lib().vhap_gbuffer_bwd(_p(pos), _p(tri), _p(vnormal), _p(uv), _p(tri_uv), _p(rast), _p(c(d_normal)),
                                               _p(c(d_texc)), _p(c(d_texd)), _p(c(d_rast)), _p(c(d_db)), _p(uv_nograd), B, V, F, H, W,
                                               _p(d_pos), _p(d_vn), _stream()), "vhap_gbuffer_bwd")
        return None, d_pos, None, d_vn, None, None, None, None


def raster_interp(glctx, pos, tri, vnormal, uv, tri_uv, resolution, uv_nograd_faces=None):
    """Differentiable fused pass -> (rast, rast_db, normal [B,H,W,3], texc [B,H,W,2], texd [B,H,W,4]).
    `uv_nograd_faces`: uint8 [F], faces whose texture coordinates carry no gradient."""
    _chk_cuda(pos, tri, vnormal, uv, tri_uv)
    _check_raster_args(pos, tri, resolution)
    return _RasterInterp.apply(glctx, _f32c(pos), _i32c(tri), _f32c(vnormal), _f32c(uv), _i32c(tri_uv), tuple(resolution), uv_nograd_faces)
