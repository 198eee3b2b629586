"""Host-side wrappers (autograd Functions over the C ABI) for the per-frame parameter stage, the camera / landmark
energy, the offset and texture regularisers, and the fused Adam update (vhap_amd/csrc/frame.hip, reg.hip).

Together they replace the several hundred tiny eager launches per step that the reference spends outside nvdiffrast:
    frame_prep       FlameHead.forward up to the skinning (flame.py:571-634, lbs.py:25-57, 254-301) + the parameter
                     energies of tracker.py:486-500, 616-680
    camera           projection_from_intrinsics / mvp_from_camera_param (render_nvdiffrast.py:102-160)
    landmark_energy  vertices2landmarks (lbs.py:60-98) + compute_lmk_energy (tracker.py:347-389)
    offset_reg       reg_offset_lap / reg_offset / reg_offset_rigid (tracker.py:552-600)
    tex_prep         get_albedo (tracker.py:247-258) + reg_tex_tv / reg_tex_res_clusters (tracker.py:518-541)
    HipAdam          torch.optim.Adam (tracker.py:159-211)
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .ops import _f32c, _p, _stream

FW = {"smooth_trans": 0, "smooth_rot": 1, "smooth_neck": 2, "smooth_jaw": 3, "smooth_eyes": 4, "smooth_expr": 5,
      "reg_neck": 6, "reg_jaw": 7, "reg_eyes": 8, "reg_expr": 9, "reg_shape": 10}
FRAME_TERMS = ("smooth_pose", "reg_joint", "smooth_joint", "reg_expr", "smooth_expr", "reg_shape")


def _chk(rc, what):
    _lib.check(rc, what)


def frame_weights(**kw):
    """-> ctypes float[12] for vhap_frame_prep_*; unspecified terms are disabled (0)."""
    w = (ctypes.c_float * 12)()
    for k, v in kw.items():
        w[FW[k]] = float(v)
    return w


class FrameModel:
    """Static tensors of the per-frame stage: JT = J_regressor v_template, JS = J_regressor shapedirs, J_regressor, parents."""

    def __init__(self, fb, J_regressor, parents):
        self.fb = fb
        self.JT, self.JS = fb.JT.contiguous(), fb.JS.contiguous()
        nzv = (J_regressor != 0).any(dim=0).nonzero().reshape(-1)             # J_regressor is sparse: compact column list
        self.jreg_idx = nzv.int().contiguous()
        self.jreg_w = J_regressor[:, nzv].t().contiguous()                     # [M,J]
        self.jreg_n = int(nzv.numel())
        self.J = int(J_regressor.shape[0])
        self.parents = (ctypes.c_int32 * self.J)(*[int(p) for p in parents])
        self.V = int(J_regressor.shape[1])


class _FramePrep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fm, weights, ts, shape, expr, rotation, translation, neck, jaw, eyes, offset):
        fb = fm.fb
        B, N = ts.shape[0], expr.shape[0]
        NS, NE = shape.shape[0], expr.shape[1]
        Bp = (B + 15) // 16 * 16
        dev = expr.device
        coef = torch.empty(Bp, fb.Kp, dtype=torch.float32, device=dev)
        A = torch.empty(B, fm.J, 12, dtype=torch.float32, device=dev)
        small = torch.empty(B * 3 + B * fm.J * 3 + 6, dtype=torch.float32, device=dev)
        transl, Jrest, terms = small[:B * 3].view(B, 3), small[B * 3:B * 3 + B * fm.J * 3], small[-6:]
        _chk(_lib.lib().vhap_frame_prep_fwd(_p(ts), _p(shape), _p(expr), _p(rotation), _p(translation), _p(neck), _p(jaw), _p(eyes),
                                            _p(fm.JT), _p(fm.JS), _p(fm.jreg_idx), _p(fm.jreg_w), fm.jreg_n, _p(offset), fm.parents, weights, B, Bp, N, NS, NE,
                                            fm.J, fb.Kp, fm.V, _p(coef), _p(A), _p(transl), _p(Jrest), _p(terms), 0, _stream()),
             "vhap_frame_prep_fwd")
        ctx.fm, ctx.weights, ctx.dims = fm, weights, (B, Bp, N, NS, NE)
        ctx.save_for_backward(ts, shape, expr, rotation, translation, neck, jaw, eyes, offset, Jrest)
        ctx.set_materialize_grads(False)
        return coef, A, transl, terms

    @staticmethod
    def backward(ctx, d_coef, d_A, d_transl, d_terms):
        ts, shape, expr, rotation, translation, neck, jaw, eyes, offset, Jrest = ctx.saved_tensors
        fm, fb = ctx.fm, ctx.fm.fb
        B, Bp, N, NS, NE = ctx.dims
        need = ctx.needs_input_grad[3:11]
        tens = (shape, expr, rotation, translation, neck, jaw, eyes, offset)
        sizes = [t.numel() if (t is not None and n) else 0 for t, n in zip(tens, need)]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=expr.device)        # one zero-fill for all gradients
        grads, o = [], 0
        for t, s in zip(tens, sizes):
            grads.append(flat[o:o + s].view(t.shape) if s else None)
            o += s
        c = lambda t: _f32c(t) if t is not None else None
        _chk(_lib.lib().vhap_frame_prep_bwd(_p(ts), _p(shape), _p(expr), _p(rotation), _p(translation), _p(neck), _p(jaw), _p(eyes),
                                            _p(fm.JS), _p(fm.jreg_idx), _p(fm.jreg_w), fm.jreg_n, _p(offset), fm.parents, ctx.weights, _p(Jrest), _p(c(d_coef)),
                                            _p(c(d_A)), _p(c(d_transl)), _p(c(d_terms)), B, Bp, N, NS, NE, fm.J, fb.Kp, fm.V,
                                            *[_p(g) for g in grads], 0, _stream()), "vhap_frame_prep_bwd")
        return (None, None, None, *grads)


def frame_prep(fm, weights, ts, shape, expr, rotation, translation, neck, jaw, eyes, offset=None):
    """-> coef [Bp,Kp], A [B,J,12], transl [B,3], terms [6] (FRAME_TERMS, weighted)."""
    off = _f32c(offset.reshape(-1, 3)) if offset is not None else None
    f = _f32c
    return _FramePrep.apply(fm, weights, ts.contiguous(), f(shape), f(expr), f(rotation), f(translation), f(neck), f(jaw), f(eyes), off)


# ------------------------------------------------------------------------------------------------
class _Camera(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K, RT, B, H, W, near, far):
        mvp = torch.empty(B, 4, 4, dtype=torch.float32, device=K.device)
        kb, rb = int(K.shape[0] > 1), int(RT.shape[0] > 1)
        _chk(_lib.lib().vhap_camera_fwd(_p(K), _p(RT), B, kb, rb, H, W, near, far, _p(mvp), _stream()), "vhap_camera_fwd")
        ctx.save_for_backward(RT)
        ctx.dims = (B, rb, H, W, K.shape[0])
        return mvp

    @staticmethod
    def backward(ctx, d_mvp):
        (RT,) = ctx.saved_tensors
        B, rb, H, W, nk = ctx.dims
        d_K = torch.empty(B, 4, dtype=torch.float32, device=RT.device)
        _chk(_lib.lib().vhap_camera_bwd(_p(RT), _p(_f32c(d_mvp)), B, rb, H, W, _p(d_K), _stream()), "vhap_camera_bwd")
        if nk == 1:
            d_K = d_K.sum(dim=0, keepdim=True)
        return d_K, None, None, None, None, None, None


def camera(K, RT, B, image_size, near=0.1, far=10.0):
    """K [B|1,4] (fx, fy, cx, cy), RT [B|1,3,4] -> mvp [B,4,4]."""
    if K.dim() != 2 or K.shape[-1] != 4:
        raise ValueError(f"Expected K to be (N, 4) but got: {tuple(K.shape)}")
    RT = _f32c(RT[..., :3, :])
    return _Camera.apply(_f32c(K), RT, int(B), int(image_size[0]), int(image_size[1]), float(near), float(far))


class LandmarkModel:
    def __init__(self, faces, lmk_faces_idx, lmk_bary):
        idx = lmk_faces_idx.reshape(-1)
        self.vidx = faces[idx].int().contiguous()                  # [L,3]
        self.bary = lmk_bary.reshape(-1, 3).float().contiguous()
        self.L = int(self.vidx.shape[0])


class _Landmark(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lm, verts, mvp, lmk2d, cfg, want_lmk3d):
        B, V, _ = verts.shape
        l0, l1, b0, b1, boost, H, W = cfg
        e = torch.empty((), dtype=torch.float32, device=verts.device)
        lmk3d = torch.empty(B, lm.L, 3, dtype=torch.float32, device=verts.device) if want_lmk3d else None
        _chk(_lib.lib().vhap_landmark_fwd(_p(verts), _p(lm.vidx), _p(lm.bary), _p(mvp), _p(lmk2d), B, V, lm.L, lmk2d.shape[1], l0, l1,
                                          b0, b1, boost, H, W, _p(lmk3d), _p(e), 0, _stream()), "vhap_landmark_fwd")
        ctx.lm, ctx.cfg = lm, cfg
        ctx.save_for_backward(verts, mvp, lmk2d)
        if lmk3d is None:
            lmk3d = torch.empty(0, device=verts.device)
        ctx.mark_non_differentiable(lmk3d)
        return e, lmk3d

    @staticmethod
    def backward(ctx, d_e, _d_l):
        verts, mvp, lmk2d = ctx.saved_tensors
        lm = ctx.lm
        B, V, _ = verts.shape
        l0, l1, b0, b1, boost, H, W = ctx.cfg
        d_verts = torch.zeros_like(verts)
        d_mvp = torch.empty_like(mvp) if ctx.needs_input_grad[2] else None
        _chk(_lib.lib().vhap_landmark_bwd(_p(verts), _p(lm.vidx), _p(lm.bary), _p(mvp), _p(lmk2d), _p(_f32c(d_e.reshape(1))), B, V, lm.L,
                                          lmk2d.shape[1], l0, l1, b0, b1, boost, H, W, _p(d_verts), _p(d_mvp), _stream()),
             "vhap_landmark_bwd")
        return None, d_verts, d_mvp, None, None, None


def landmark_energy(lm, verts, mvp, lmk2d, image_size, disable_jawline=False, want_lmk3d=False):
    """-> (mean confidence-weighted L1 landmark error, landmarks [B,L,3] or an empty tensor)."""
    cfg = (17, 68, 0, 0, 1.0, int(image_size[0]), int(image_size[1])) if disable_jawline else \
        (0, 68, 27, 36, 10.0, int(image_size[0]), int(image_size[1]))
    return _Landmark.apply(lm, _f32c(verts), _f32c(mvp), _f32c(lmk2d), cfg, want_lmk3d)


# ------------------------------------------------------------------------------------------------
class OffsetRegModel:
    """Static tables of the offset regularisers: Laplacian CSR, per-vertex relax weights, rigid-region CSR."""

    def __init__(self, lap_ptr, lap_col, lap_val, w_lap, w_abs, regions, device):
        i32 = lambda t: torch.as_tensor(np.asarray(t.cpu() if torch.is_tensor(t) else t)).to(torch.int32).to(device).contiguous()
        self.ptr, self.col = i32(lap_ptr), i32(lap_col)
        self.val = lap_val.float().to(device).contiguous()
        self.w_lap = w_lap.reshape(-1).float().contiguous() if w_lap is not None else None
        self.w_abs = w_abs.reshape(-1).float().contiguous() if w_abs is not None else None
        self.V = int(self.ptr.numel() - 1)
        ptr = np.zeros(len(regions) + 1, dtype=np.int32)
        for i, r in enumerate(regions):
            ptr[i + 1] = ptr[i] + int(r.numel())
        self.nreg = len(regions)
        self.rptr = torch.from_numpy(ptr).to(device)
        self.ridx = torch.cat([r.reshape(-1) for r in regions]).int().to(device).contiguous() if regions else None


class _OffsetReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, om, off, scales):
        terms = torch.empty(3, dtype=torch.float32, device=off.device)
        _chk(_lib.lib().vhap_offset_reg_fwd(_p(off), _p(om.ptr), _p(om.col), _p(om.val), _p(om.w_lap), _p(om.w_abs), _p(om.rptr), _p(om.ridx),
                                            om.V, om.nreg, *scales, _p(terms), 0, _stream()), "vhap_offset_reg_fwd")
        ctx.om, ctx.scales = om, scales
        ctx.save_for_backward(off)
        return terms

    @staticmethod
    def backward(ctx, d_terms):
        (off,) = ctx.saved_tensors
        om = ctx.om
        d_off = torch.zeros_like(off)
        _chk(_lib.lib().vhap_offset_reg_bwd(_p(off), _p(om.ptr), _p(om.col), _p(om.val), _p(om.w_lap), _p(om.w_abs), _p(om.rptr), _p(om.ridx),
                                            om.V, om.nreg, *ctx.scales, _p(_f32c(d_terms)), _p(d_off), _stream()), "vhap_offset_reg_bwd")
        return None, d_off, None


def offset_reg(om, offset, w_lap, w_abs, w_rigid):
    """offset [1,V,3] -> terms [3] = reg_offset_lap, reg_offset, reg_offset_rigid (weighted; a None weight gives 0)."""
    V = om.V
    scales = (float(w_lap or 0.0) / V, float(w_abs or 0.0) / (3 * V), float(w_rigid or 0.0) / 3.0)
    return _OffsetReg.apply(om, _f32c(offset.reshape(-1, 3)), scales)


# ------------------------------------------------------------------------------------------------
def _n_gather(T):
    """Mip levels whose fold is fused into vhap_tex_prep_bwd (gathered per texel) instead of separate read-modify-write passes."""
    n = 0
    while n < 12 and T % (1 << (n + 1)) == 0 and (T >> (n + 1)) >= 1:      # (12 = TEXB_MAXG of csrc/reg.hip: every level up to T = 4096)
        n += 1
    return n


_TEXBIN_WORK = {}


def texture_grad_binned(T, C, texc, texd, d_out, d_tex, d_mips, work=None):
    """d_tex / d_mips += texture gradient of a SHARED T x T x C texture sampled at (texc, texd) by all frames, through uv-space binning
    (vhap_texture_grad_binned).  Returns False when the library declines (texture too large): the caller then uses vhap_texture_bwd."""
    L = _lib.lib()
    B, H, W, _ = texc.shape
    if work is None:
        key = (B, H, W, texc.device)
        work = _TEXBIN_WORK.get(key)
        if work is None:
            work = _TEXBIN_WORK[key] = torch.empty(L.vhap_texture_grad_binned_work_bytes(B, H, W), dtype=torch.uint8, device=texc.device)
    rc = L.vhap_texture_grad_binned(T, T, C, _p(texc), _p(texd), _p(d_out), B, H, W, _p(d_tex), _p(d_mips) if d_mips is not None and d_mips.numel() else 0,
                                    _p(work), work.numel(), _stream())
    if rc == -5:        # VHAP_E_UNSUPPORTED
        return False
    _chk(rc, "vhap_texture_grad_binned")
    return True


def use_binned_texgrad():
    return True      # (the screen-tiled kernel -- 433 us at 16 x 512^2 against 110 -- is only the fallback for textures the binning cannot tile)


class _TexSample(torch.autograd.Function):
    """albedo = painted + residual (channel-last) -> mip pyramid -> trilinear sample at (texc, texd), with the TV / residual
    energies of the texture computed in the same pass over it.  One autograd node, so the backward can keep the texture
    gradient in its pyramid form until the last kernel: texture_bwd (atomics into the levels) -> fold down to level 1 ->
    tex_prep_bwd adds level 0 + 0.25 * level 1 + the regulariser gradients and writes d_extra in one pass."""

    @staticmethod
    def forward(ctx, painted, extra, mask, scales, texc, texd):
        L = _lib.lib()
        src = extra if extra is not None else painted
        T, dev = src.shape[-1], src.device
        B, H, W, _ = texc.shape
        albedo = torch.empty(1, T, T, 3, dtype=torch.float32, device=dev)
        terms = torch.empty(2, dtype=torch.float32, device=dev)
        _chk(L.vhap_tex_prep_fwd(_p(painted), _p(extra), _p(mask), T, *scales, _p(albedo), _p(terms), 0, _stream()), "vhap_tex_prep_fwd")
        mips = torch.empty(L.vhap_texture_mip_floats(1, T, T, 3), dtype=torch.float32, device=dev)
        _chk(L.vhap_texture_mip_build(_p(albedo), 1, T, T, 3, _p(mips), _stream()), "vhap_texture_mip_build")
        out = torch.empty(B, H, W, 3, dtype=torch.float32, device=dev)
        _chk(L.vhap_texture_fwd(_p(albedo), _p(mips), 1, T, T, 3, _p(texc), _p(texd), B, H, W, _p(out), _stream()), "vhap_texture_fwd")
        ctx.scales, ctx.T = scales, T
        ctx.save_for_backward(albedo, mips, extra, mask, texc, texd)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(albedo)
        return out, terms, albedo

    @staticmethod
    def backward(ctx, d_out, d_terms, _d_albedo):
        albedo, mips, extra, mask, texc, texd = ctx.saved_tensors
        L = _lib.lib()
        T = ctx.T
        B, H, W, _ = texc.shape
        need_tex = extra is not None and ctx.needs_input_grad[1]
        need_uv, need_da = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        dev = albedo.device
        d_tex = d_mips = d_extra = d_uv = d_da = None
        if d_out is not None:
            if need_tex:
                # level 0 and levels 1.. in ONE zero-filled buffer
                buf = torch.zeros(albedo.numel() + mips.numel(), dtype=torch.float32, device=dev)
                d_tex, d_mips = buf[:albedo.numel()], buf[albedo.numel():]
            d_uv = torch.empty_like(texc) if need_uv else None
            d_da = torch.empty_like(texd) if need_da else None
            d_out = _f32c(d_out)
            binned = need_tex and use_binned_texgrad() and texture_grad_binned(T, 3, texc, texd, d_out, d_tex, d_mips)
            if not binned or need_uv or need_da:
                _chk(L.vhap_texture_bwd(_p(albedo), _p(mips), 1, T, T, 3, _p(texc), _p(texd), _p(d_out), B, H, W,
                                        _p(None if binned else d_tex), _p(None if binned else d_mips), _p(d_uv), _p(d_da), _stream()),
                     "vhap_texture_bwd")
            if need_tex and mips.numel() > 0:
                _chk(L.vhap_texture_mip_fold(_p(d_tex), _p(d_mips), 1, T, T, 3, _n_gather(T), _stream()), "vhap_texture_mip_fold")
        if need_tex:
            if d_terms is None:
                d_terms = torch.zeros(2, dtype=torch.float32, device=dev)
            d_extra = torch.empty_like(extra)
            has = d_mips is not None and d_mips.numel() > 0
            _chk(L.vhap_tex_prep_bwd(_p(albedo), _p(extra), _p(mask), _p(d_tex), _p(d_mips) if has else 0, _n_gather(T) if has else 0,
                                     _p(_f32c(d_terms)), T, *ctx.scales,
                                     _p(d_extra), _stream()), "vhap_tex_prep_bwd")
        return None, d_extra, None, None, d_uv, d_da


class TexSampler:
    """Callable handed to HipDiffRenderer.render_rgba(tex_sampler=...): samples painted + residual at the interpolated
    texture coordinates; afterwards `.terms` = (reg_tex_tv, reg_tex_res_clusters) weighted, `.albedo_cl` = [1,T,T,3]."""

    def __init__(self, painted, extra, res_mask_u8, w_tv, w_res):
        T = (extra if extra is not None else painted).shape[-1]
        self.scales = (float(w_tv or 0.0) / (3.0 * T * (T - 1)), float(w_res or 0.0) / (3.0 * T * T))
        f = lambda t: _f32c(t.reshape(3, T, T)) if t is not None else None
        self.painted, self.extra, self.mask = f(painted), f(extra), res_mask_u8
        self.terms = self.albedo_cl = None

    def __call__(self, texc, texd):
        out, self.terms, self.albedo_cl = _TexSample.apply(self.painted, self.extra, self.mask, self.scales, _f32c(texc), _f32c(texd))
        return out

    def prep_only(self):
        """No photometric term in this stage: only the channel-last albedo and the regulariser terms."""
        self.albedo_cl, self.terms = tex_prep(self.painted, self.extra, self.mask, None, None, scales=self.scales)
        return self.albedo_cl


class _TexPrep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, painted, extra, mask, scales):
        T = (extra if extra is not None else painted).shape[-1]
        dev = (extra if extra is not None else painted).device
        albedo = torch.empty(1, T, T, 3, dtype=torch.float32, device=dev)
        terms = torch.empty(2, dtype=torch.float32, device=dev)
        _chk(_lib.lib().vhap_tex_prep_fwd(_p(painted), _p(extra), _p(mask), T, *scales, _p(albedo), _p(terms), 0, _stream()), "vhap_tex_prep_fwd")
        ctx.scales, ctx.T = scales, T
        ctx.save_for_backward(albedo, extra, mask)
        ctx.set_materialize_grads(False)
        return albedo, terms

    @staticmethod
    def backward(ctx, d_albedo, d_terms):
        albedo, extra, mask = ctx.saved_tensors
        if extra is None or not ctx.needs_input_grad[1]:
            return None, None, None, None
        if d_terms is None:
            d_terms = torch.zeros(2, dtype=torch.float32, device=albedo.device)
        d_extra = torch.empty_like(extra)
        _chk(_lib.lib().vhap_tex_prep_bwd(_p(albedo), _p(extra), _p(mask), _p(_f32c(d_albedo) if d_albedo is not None else None), 0, 0,
                                          _p(_f32c(d_terms)), ctx.T, *ctx.scales, _p(d_extra), _stream()), "vhap_tex_prep_bwd")
        return None, d_extra, None, None


def tex_prep(painted, extra, res_mask_u8, w_tv, w_res, scales=None):
    """painted [3,T,T] or None, extra [3,T,T] or None -> albedo [1,T,T,3] (channel-last), terms [2] = reg_tex_tv,
    reg_tex_res_clusters (weighted)."""
    T = (extra if extra is not None else painted).shape[-1]
    if scales is None:
        scales = (float(w_tv or 0.0) / (3.0 * T * (T - 1)), float(w_res or 0.0) / (3.0 * T * T))
    f = lambda t: _f32c(t.reshape(3, T, T)) if t is not None else None
    return _TexPrep.apply(f(painted), f(extra), res_mask_u8, scales)


# ------------------------------------------------------------------------------------------------
class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) with the update of ALL parameter tensors in one launch
    (vhap_adam_step).  Same param_groups / lr-scheduler interface; the learning rates and the step counter live on
    the device so a captured step sees later changes (call sync_lr() after changing group['lr'] outside a capture)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tab = None
        self._lr_host = None

    def _build(self):
        ps = [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"]]
        if len(ps) > 16:
            raise ValueError("HipAdam handles at most 16 parameter tensors")
        dev = ps[0][1].device
        for _, p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("HipAdam needs contiguous float32 parameters on a HIP device")
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
        self._step = torch.zeros(1, dtype=torch.int32, device=dev)
        self._lr_dev = torch.zeros(len(self.param_groups), dtype=torch.float32, device=dev)
        n = len(ps)
        P = ctypes.c_void_p * n
        self._tab = {
            "ps": [p for _, p in ps], "n": n,
            "p": P(*[p.data_ptr() for _, p in ps]),
            "m": P(*[self.state[p]["exp_avg"].data_ptr() for _, p in ps]),
            "v": P(*[self.state[p]["exp_avg_sq"].data_ptr() for _, p in ps]),
            "numel": (ctypes.c_int64 * n)(*[p.numel() for _, p in ps]),
            "lr_index": (ctypes.c_int32 * n)(*[gi for gi, _ in ps]),
        }
        self._lr_host = None
        self.sync_lr()

    def lr_changed(self):
        return [float(g["lr"]) for g in self.param_groups] != self._lr_host

    def sync_lr(self):
        lrs = [float(g["lr"]) for g in self.param_groups]
        if lrs != self._lr_host:
            if len(lrs) <= 16:
                # one tiny launch on the current stream, the values in its kernel arguments: a pageable host-to-device copy would block
                # the host until every replay queued on this stream has run (a stage loop changes the rates once per epoch)
                _chk(_lib.lib().vhap_set_floats(_p(self._lr_dev), (ctypes.c_float * len(lrs))(*lrs), len(lrs), _stream()), "vhap_set_floats")
            else:
                self._lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=False)
            self._lr_host = lrs

    def reset_state(self, lr_scale_base=None):
        """Back to a freshly constructed optimiser: zero moments, step 0 (the reference builds a new Adam per optimize_stage call)."""
        if self._tab is None:
            return
        for p in self._tab["ps"]:
            st = self.state[p]
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
        self._step.zero_()

    def fused_update_args(self, p):
        """(exp_avg, exp_avg_sq, lr pointer, step pointer, beta1, beta2, eps) of parameter `p` for a kernel that applies the Adam update itself
        (vhap_tex_prep_bwd_adam); the caller must still run step(skip=(p,)) to advance the counter.  None if `p` is not ours."""
        if self._tab is None:
            self._build()
        for gi, g in enumerate(self.param_groups):
            if any(q is p for q in g["params"]):
                st = self.state[p]
                return (st["exp_avg"], st["exp_avg_sq"], self._lr_dev[gi:gi + 1], self._step, float(g["betas"][0]), float(g["betas"][1]),
                        float(g["eps"]))
        return None

    @property
    def step_count(self):
        return self._step

    @torch.no_grad()
    def advance(self):
        """Advance the step counter NOW (one tiny launch, e.g. at the head of a captured step, off the critical path); the pieces of this
        step are then issued as step(..., advanced=True) in any order, on any streams, with no counter launch behind them."""
        if self._tab is None:
            self._build()
        _chk(_lib.lib().vhap_adam_advance(_p(self._step), _stream()), "vhap_adam_advance")

    @torch.no_grad()
    def step(self, closure=None, only=None, skip=None, advance=True, advanced=False):
        """`only` / `skip` (tensors): update a subset -- a step may be issued in pieces, e.g. the texture as soon as its gradient is
        complete, on a side stream; every piece but the last passes advance=False so that all of them see the same step count
        (or all pass advanced=True after advance())."""
        if self._tab is None:
            self._build()
        t = self._tab
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        pick_p = lambda p: (only is None or any(p is q for q in only)) and (skip is None or not any(p is q for q in skip))
        sel = [i for i, p in enumerate(t["ps"]) if p.grad is not None and pick_p(p)]      # like torch: parameters without .grad are skipped
        if not sel:
            return
        for i in sel:
            p = t["ps"][i]
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        n = len(sel)
        P = ctypes.c_void_p * n
        if _lib.ACCESS is not None:       # (the tensors travel as pointer tables: tell a recording capture what this call touches)
            for i in sel:
                q = t["ps"][i]
                for x in (q, q.grad, self.state[q]["exp_avg"], self.state[q]["exp_avg_sq"]):
                    _lib.ACCESS.touch(x)
        pick = lambda arr, ty: (ty * n)(*[arr[i] for i in sel])
        G = P(*[t["ps"][i].grad.data_ptr() for i in sel])
        g0 = self.param_groups[0]
        L = _lib.lib()
        _chk(L.vhap_adam_step(n, pick(t["p"], ctypes.c_void_p), G, pick(t["m"], ctypes.c_void_p), pick(t["v"], ctypes.c_void_p),
                              pick(t["numel"], ctypes.c_int64), pick(t["lr_index"], ctypes.c_int32), _p(self._lr_dev),
                              _p(self._step), float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]),
                              _lib.CALL_ADAM_STEP_ADVANCED if advanced else (0 if advance else _lib.CALL_ADAM_KEEP_STEP), _stream()),
             "vhap_adam_step")
