"""Fused stages of the fit step as autograd Functions over the C ABI (the MI355X-first replacements
for the eager-torch arithmetic the reference wraps around nvdiffrast):

    flame_skin      blendshapes + pose correctives + skinning (MFMA)       flame.py:595-634, lbs.py
    transform       world -> clip                                          render_nvdiffrast.py:162-206
    vertex_normals  area-weighted vertex normals (CSR gather)              render_nvdiffrast.py:297-316
    shade           normalise + SH shading + composite + reg_diffuse stats render_nvdiffrast.py:386-421, tracker.py:547-550
    photo_sum       sum |gt - pred| and #(alpha > 0)                       tracker.py:430-439
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .ops import _f32c, _p, _stream


def _chk(rc, what):
    _lib.check(rc, what)


# ------------------------------------------------------------------------------------------------
class FlameBasis:
    """Device-resident bases for the fused FLAME kernels, built once from FlameHead buffers."""

    def __init__(self, shapedirs, posedirs, J_regressor, v_template, lbs_weights):
        V, _, NB = shapedirs.shape
        P = posedirs.shape[0]
        dev = shapedirs.device
        self.V, self.Kb, self.K = V, NB, NB + P
        assert self.Kb % 16 == 0 and self.K % 4 == 0, "shape+expr count must be a multiple of 16, pose rows of 4"
        self.Vp = (V + 63) // 64 * 64
        self.Kp = (self.K + 15) // 16 * 16
        full = torch.zeros(3, self.K, self.Vp, dtype=torch.float32, device=dev)
        full[:, :NB, :V] = shapedirs.permute(1, 2, 0)                       # [3,NB,V]
        full[:, NB:, :V] = posedirs.view(P, V, 3).permute(2, 0, 1)          # [3,P,V]
        self.basis = full.contiguous()
        bt = torch.zeros(3, self.Vp, self.Kp, dtype=torch.float32, device=dev)
        bt[:, :, :self.K] = full.permute(0, 2, 1)
        self.basisT = bt.contiguous()
        self.JS = torch.einsum("jv,vcl->jcl", J_regressor, shapedirs).reshape(-1, NB).contiguous()   # [15,NB]
        self.JT = (J_regressor @ v_template).contiguous()                                           # [5,3]
        self.w = lbs_weights.contiguous()
        self.templ = v_template.contiguous()
        self.n_partial = _lib.lib().vhap_flame_bwd_partial_floats


class _FlameSkin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fb, coef, A, transl, offset):
        B = A.shape[0]
        V = fb.V
        dev = coef.device
        verts = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
        v_shaped = torch.empty_like(verts)
        v_posed = torch.empty_like(verts)
        _chk(_lib.lib().vhap_flame_skin_fwd(_p(coef), _p(fb.basis), _p(A), _p(fb.w), _p(fb.templ), _p(offset), _p(transl), B, V,
                                            fb.Vp, fb.K, fb.Kb, fb.Kp, _p(verts), _p(v_shaped), _p(v_posed), 0, _stream()),
             "vhap_flame_skin_fwd")
        ctx.fb = fb
        ctx.has_offset = offset is not None
        ctx.save_for_backward(A, v_posed)
        ctx.coef_shape = coef.shape
        return verts, v_shaped

    @staticmethod
    def backward(ctx, d_verts, d_vshaped):
        fb = ctx.fb
        A, v_posed = ctx.saved_tensors
        B, V = A.shape[0], fb.V
        dev = A.device
        d_verts = _f32c(d_verts)
        d_vs = _f32c(d_vshaped) if d_vshaped is not None else None
        g_posed = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
        g_shaped = torch.empty_like(g_posed)
        partial = torch.empty(fb.n_partial(B, fb.Vp, fb.Kp), dtype=torch.float32, device=dev)
        d_coef = torch.empty(ctx.coef_shape, dtype=torch.float32, device=dev)
        d_A = torch.zeros_like(A)
        d_t = torch.zeros(B, 3, dtype=torch.float32, device=dev)
        _chk(_lib.lib().vhap_flame_skin_bwd(_p(d_verts), _p(d_vs), _p(v_posed), _p(A), _p(fb.w), _p(fb.basisT), B, V, fb.Vp, fb.Kb,
                                            fb.Kp, _p(g_posed), _p(g_shaped), _p(partial), _p(d_coef), _p(d_A), _p(d_t), 0, _stream()),
             "vhap_flame_skin_bwd")
        d_off = g_shaped.sum(dim=0, keepdim=True) if ctx.has_offset else None
        return None, d_coef, d_A, d_t, d_off


def flame_skin(fb, coef, A, transl, offset=None):
    """coef [Bp,Kp] (padded), A [B,5,12], transl [B,3], offset [1,V,3] or None -> verts, v_shaped [B,V,3]."""
    off = _f32c(offset.reshape(-1, 3)) if offset is not None else None
    return _FlameSkin.apply(fb, _f32c(coef), _f32c(A), _f32c(transl), off)


# ------------------------------------------------------------------------------------------------
class _Transform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, M):
        B, V, _ = verts.shape
        clip = torch.empty(B, V, 4, dtype=torch.float32, device=verts.device)
        _chk(_lib.lib().vhap_transform_fwd(_p(verts), _p(M), B, V, _p(clip), _stream()), "vhap_transform_fwd")
        ctx.save_for_backward(verts, M)
        return clip

    @staticmethod
    def backward(ctx, d_clip):
        verts, M = ctx.saved_tensors
        B, V, _ = verts.shape
        d_verts = torch.empty_like(verts)
        d_M = torch.zeros_like(M) if ctx.needs_input_grad[1] else None
        _chk(_lib.lib().vhap_transform_bwd(_p(verts), _p(M), _p(_f32c(d_clip)), B, V, 0, _p(d_verts), _p(d_M), _stream()),
             "vhap_transform_bwd")
        return d_verts, d_M


def transform(verts, M):
    """clip [B,V,4] = [verts;1] @ M^T with M [B,4,4] (or [1,4,4], expanded)."""
    if M.shape[0] != verts.shape[0]:
        M = M.expand(verts.shape[0], -1, -1)
    return _Transform.apply(_f32c(verts), _f32c(M))


# ------------------------------------------------------------------------------------------------
class MeshCSR:
    """Static vertex -> incident-corner CSR on the device (vhap_amd.topology.build_vertex_corner_csr)."""

    def __init__(self, faces_i32, vc_ptr, vc_idx):
        self.tri, self.ptr, self.idx = faces_i32.contiguous(), vc_ptr.contiguous(), vc_idx.contiguous()

    @staticmethod
    def from_faces(faces):
        from .topology import build_vertex_corner_csr
        f = faces.detach().cpu().numpy()
        ptr, idx = build_vertex_corner_csr(f, int(f.max()) + 1)
        dev = faces.device
        return MeshCSR(faces.int(), torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev))


class _VertexNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, csr):
        B, V, _ = verts.shape
        vn = torch.empty_like(verts)
        _chk(_lib.lib().vhap_vnormal_fwd(_p(verts), _p(csr.tri), _p(csr.ptr), _p(csr.idx), B, V, _p(vn), _stream()), "vhap_vnormal_fwd")
        ctx.csr = csr
        ctx.save_for_backward(verts)
        return vn

    @staticmethod
    def backward(ctx, d_vn):
        (verts,) = ctx.saved_tensors
        csr = ctx.csr
        B, V, _ = verts.shape
        d_verts = torch.empty_like(verts)
        scratch = torch.empty_like(verts)
        _chk(_lib.lib().vhap_vnormal_bwd(_p(verts), _p(csr.tri), _p(csr.ptr), _p(csr.idx), _p(_f32c(d_vn)), B, V, 0, _p(scratch),
                                         _p(d_verts), _stream()), "vhap_vnormal_bwd")
        return d_verts, None


def vertex_normals(verts, csr):
    return _VertexNormals.apply(_f32c(verts), csr)


# ------------------------------------------------------------------------------------------------
def _decode_ordered_max(stats):
    u = stats[1:2].view(torch.int32)
    return torch.where(u < 0, (u & 0x7FFFFFFF).view(torch.float32), (~u).view(torch.float32))[0]


class _Shade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, normal_raw, albedo, lights, rast, bg_image, bg_color, sh_const, want_reg):
        B, H, W, _ = rast.shape
        rgba = torch.empty(B, H, W, 4, dtype=torch.float32, device=rast.device)
        stats = torch.empty(4, dtype=torch.float32, device=rast.device) if want_reg else None
        col = (ctypes.c_float * 3)(*bg_color) if bg_color is not None else None
        _chk(_lib.lib().vhap_shade_fwd(_p(normal_raw), _p(albedo), _p(rast), _p(bg_image), ctypes.cast(col, ctypes.c_void_p) if col else 0,
                                       _p(lights), _p(sh_const), 0, 0, B, H, W, _p(rgba), _p(stats), 0, 0, _stream()), "vhap_shade_fwd")
        ctx.save_for_backward(normal_raw, albedo, lights, rast, sh_const, stats)
        if want_reg:
            mx = _decode_ordered_max(stats)
            reg = torch.relu(mx - 1.0) + stats[2] / float(B * H * W)
        else:
            reg = torch.zeros((), dtype=torch.float32, device=rast.device)
        return rgba, reg

    @staticmethod
    def backward(ctx, d_rgba, d_reg):
        normal_raw, albedo, lights, rast, sh_const, stats = ctx.saved_tensors
        B, H, W, _ = rast.shape
        need_n, need_a, need_l = ctx.needs_input_grad[:3]
        d_n = torch.empty_like(normal_raw) if need_n else None
        d_a = torch.empty_like(albedo) if need_a else None
        d_l = torch.zeros_like(lights) if need_l else None
        use_reg = stats is not None and d_reg is not None
        d_reg_c = _f32c(d_reg.reshape(1)) if use_reg else None
        _chk(_lib.lib().vhap_shade_bwd(_p(normal_raw), _p(albedo), _p(rast), _p(lights), _p(sh_const), _p(_f32c(d_rgba)), 0, _p(d_reg_c),
                                       _p(stats if use_reg else None), B, H, W, _p(d_a), _p(d_n), _p(d_l), _stream()), "vhap_shade_bwd")
        return d_n, d_a, d_l, None, None, None, None, None


def shade(normal_raw, albedo, lights, rast, background, sh_const, want_reg=False):
    """-> (rgba [B,H,W,4] renderer space, reg_diffuse scalar).  background: list of 3 floats, or an image-space
    tensor [B,H,W,3] (a permuted view of a contiguous [B,3,H,W] tensor is used in place)."""
    bg_img, bg_col = None, None
    if isinstance(background, (list, tuple)):
        bg_col = [float(x) for x in background]
    else:
        nchw = background.permute(0, 3, 1, 2)
        bg_img = _f32c(nchw.detach())
    return _Shade.apply(_f32c(normal_raw), _f32c(albedo), _f32c(lights.reshape(9, 3)), _f32c(rast), bg_img, bg_col, _f32c(sh_const), want_reg)


# ------------------------------------------------------------------------------------------------
class _PhotoSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        B, H, W, _ = pred.shape
        out = torch.empty(2, dtype=torch.float32, device=pred.device)
        _chk(_lib.lib().vhap_photo_fwd(_p(pred), _p(gt), B, H, W, _p(out), 0, _stream()), "vhap_photo_fwd")
        ctx.save_for_backward(pred, gt)
        n = out[1].detach()
        ctx.mark_non_differentiable(n)
        return out[0], n

    @staticmethod
    def backward(ctx, d_sum, _d_n):
        pred, gt = ctx.saved_tensors
        B, H, W, _ = pred.shape
        d_pred = torch.empty_like(pred)
        _chk(_lib.lib().vhap_photo_bwd(_p(pred), _p(gt), _p(_f32c(d_sum.reshape(1))), B, H, W, _p(d_pred), 0, _stream()), "vhap_photo_bwd")
        return d_pred, None


def photo_sum(pred_rgba_renderer_space, gt_nchw):
    """-> (sum |gt - pred_rgb|, #(alpha > 0)) ; pred [B,H,W,4] row 0 = bottom, gt [B,3,H,W] image space."""
    return _PhotoSum.apply(_f32c(pred_rgba_renderer_space), _f32c(gt_nchw))


# ------------------------------------------------------------------------------------------------
class _Disturb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgba, rast, fid2cid, ncl, w_fg, w_bg, idx, rng=None):
        B, H, W, _ = rgba.shape
        L = _lib.lib()
        ws = torch.empty(L.vhap_disturb_workspace_ints(B, H, W), dtype=torch.int32, device=rgba.device)
        out = torch.empty_like(rgba)
        keep = torch.empty(B, H, W, dtype=torch.float32, device=rgba.device)
        if rng is not None:
            state, rate_fg, rate_bg = rng
            _chk(L.vhap_disturb_fwd_rng(_p(rgba), _p(rast), _p(fid2cid), fid2cid.numel(), ncl, rate_fg, rate_bg, _p(state), B, H, W,
                                        _p(ws), _p(out), _p(keep), _stream()), "vhap_disturb_fwd_rng")
        else:
            _chk(L.vhap_disturb_fwd(_p(rgba), _p(rast), _p(fid2cid), fid2cid.numel(), ncl, _p(w_fg), _p(w_bg), _p(idx), B, H, W, _p(ws),
                                    _p(out), _p(keep), _stream()), "vhap_disturb_fwd")
        ctx.save_for_backward(keep)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (keep,) = ctx.saved_tensors
        B, H, W = keep.shape
        d = torch.empty(B, H, W, 4, dtype=torch.float32, device=keep.device)
        _chk(_lib.lib().vhap_disturb_bwd(_p(_f32c(d_out)), _p(keep), B, H, W, _p(d), _stream()), "vhap_disturb_bwd")
        return d, None, None, None, None, None, None, None


def disturb(rgba, rast, fid2cid_i32, ncl, w_fg, w_bg, idx):
    """Cluster-wise colour disturbance of the composited image; w_fg / w_bg int32 [B,H,W(,1)], idx int64 [B*H*W]."""
    return _Disturb.apply(_f32c(rgba), _f32c(rast), fid2cid_i32, int(ncl), w_fg.contiguous(), w_bg.contiguous(), idx.contiguous())


def disturb_rng(rgba, rast, fid2cid_i32, ncl, rng_state, rate_fg, rate_bg):
    """Same with in-kernel random numbers; `rng_state` uint32 [1] device counter (advanced by every call)."""
    return _Disturb.apply(_f32c(rgba), _f32c(rast), fid2cid_i32, int(ncl), None, None, None,
                          (rng_state, float(rate_fg or 0.0), float(rate_bg or 0.0)))
