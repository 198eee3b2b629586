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
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


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
    return 0 if t is None else t.data_ptr()


class RasterizeHipContext:
    """Stands in for dr.RasterizeCudaContext().  Holds only sizing policy (no device state, so it is
    safe to share between threads -- the reference's log_media thread may render concurrently)."""

    def __init__(self, pairs_per_triangle=8, pairs_per_block=8):
        self.pairs_per_triangle = pairs_per_triangle
        self.pairs_per_block = pairs_per_block

    def workspace(self, B, F, H, W, device):
        # capacity of the (triangle, 8x8-block) lists; exceeding it is still correct (brute-force path)
        nblk = ((int(H) + 7) // 8) * ((int(W) + 7) // 8)
        cap = int(B) * (int(F) * self.pairs_per_triangle + nblk * self.pairs_per_block)
        nbytes = _lib.lib().vhap_raster_workspace_bytes(B, F, H, W, cap)
        if nbytes == 0:
            raise ValueError(f"rasterize: dimensions out of range (B={B}, F={F}, H={H}, W={W})")
        return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes, cap


def _check_raster_args(pos, tri, resolution):
    if pos.dim() != 3 or pos.shape[-1] != 4:
        raise ValueError("pos must have shape [B, V, 4] (instanced mode)")
    if tri.dim() != 2 or tri.shape[-1] != 3:
        raise ValueError("tri must have shape [F, 3]")
    H, W = int(resolution[0]), int(resolution[1])
    return H, W


def raster_fwd(ctx, pos, tri, resolution, with_db=True):
    """Forward only (no autograd).  Returns rast [B,H,W,4], rast_db [B,H,W,4] or None."""
    _chk_cuda(pos, tri)
    H, W = _check_raster_args(pos, tri, resolution)
    pos, tri = _f32c(pos), _i32c(tri)
    B, V, _ = pos.shape
    F = tri.shape[0]
    ws, nbytes, cap = ctx.workspace(B, F, H, W, pos.device)
    rast = torch.empty(B, H, W, 4, dtype=torch.float32, device=pos.device)
    db = torch.empty_like(rast) if with_db else None
    rc = _lib.lib().vhap_raster_fwd(_p(pos), _p(tri), B, V, F, H, W, _p(rast), _p(db), _p(ws), nbytes, cap, _stream())
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
    ws, nbytes, cap = ctx.workspace(B, F, H, W, pos.device)
    dev = pos.device
    rast = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
    db = torch.empty_like(rast)
    normal = torch.empty(B, H, W, 3, dtype=torch.float32, device=dev)
    texc = torch.empty(B, H, W, 2, dtype=torch.float32, device=dev)
    texd = torch.empty_like(rast)
    rc = _lib.lib().vhap_raster_interp_fwd(_p(pos), _p(tri), _p(vnormal), _p(uv), _p(tri_uv), B, V, uv.shape[0], F, H, W,
                                           _p(rast), _p(db), _p(normal), _p(texc), _p(texd), _p(ws), nbytes, cap, _stream())
    _lib.check(rc, "vhap_raster_interp_fwd")
    return rast, db, normal, texc, texd
