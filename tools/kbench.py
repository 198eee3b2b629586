"""Un-contended timing of the individual C-ABI calls of the native step at a bench configuration: builds the NativeStep, runs one
forward + backward (so that every buffer holds real data), then times each call alone (HIP events, N repetitions back to back).
usage: python tools/kbench.py [--config 2] [--reps 20] [--only name,name]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--shape", default="", help="B,H,W in place of the configuration's")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--debug-flags", type=int, default=0, help="vhap_debug_set_flags for the timed calls (A/B switches inside kernels)")
ap.add_argument("--cold", action="store_true", help="every timed call starts with cold caches (1 GiB written in between: L2 and the 256 MB "
                "Infinity Cache hold nothing of the call's inputs) -- what a kernel sees inside the step, where ~1 GB of other traffic "
                "flows between its producer and it; the median of --reps single calls")
args = ap.parse_args()
C = dict(bench.CONFIGS[args.config])
if args.shape:                                             # (a probe size, not a BASELINE configuration: B,H,W)
    C["B"], C["H"], C["W"] = (int(v) for v in args.shape.split(","))
tr, own, n_local, model, topo, gt = bench.build_tracker(C, 0, 1, "cuda:0", "weak")
from vhap_amd import _lib  # noqa: E402
from vhap_amd.ops import _p, _stream  # noqa: E402
from vhap_amd.step import NativeStep, PRE  # noqa: E402
stage = bench.STAGE
opt = tr.configure_optimizer(tr.get_train_parameters(stage), lr_scale=0.1)
ns = NativeStep(tr, tr.get_sample(own, device_index=True), stage)
ns.overlap = False
if not hasattr(ns, "d_normal"):
    E = lambda *s: torch.empty(*s, dtype=torch.float32, device="cuda")
    ns.d_normal, ns.d_texc, ns.d_texd = E(ns.B, ns.H, ns.W, 3), E(ns.B, ns.H, ns.W, 2), E(ns.B, ns.H, ns.W, 4)
for _ in range(2):
    ns.forward()
    ns.backward(1)
torch.cuda.synchronize()
L = ns.L
B, H, W, V, F, T = ns.B, ns.H, ns.W, ns.V, ns.F, ns.T
acc, g = ns.accF, ns.g
st = lambda: _stream()
n0 = ns.albedo_tex.numel()
d_tex, d_mips = g["d_tex"][:n0], g["d_tex"][n0:]


def raster(flags):
    return L.vhap_raster_shade_fwd(_p(ns.clip), _p(ns.tri), _p(ns.vn), _p(ns.uv), _p(ns.tri_uv), _p(ns.albedo_tex), _p(ns.mips), T, T,
                                   _p(tr.lights), _p(ns.sh_const), _p(ns.rgb) if ns.bg_col is None else 0,
                                   ctypes.cast(ns.bg_col, ctypes.c_void_p) if ns.bg_col is not None else 0,
                                   _p(ns.fid2cid) if ns.disturb_on else 0, ns.fid2cid.numel() if ns.disturb_on else 0, B, V, ns.uv.shape[0], F, H, W,
                                   _p(ns.rast), _p(ns.rgba), _p(ns.cid) if ns.disturb_on else 0, _p(acc[12:16]), 0, _p(ns.ws),
                                   ns.ws_bytes, ns.ws_cap, flags, st())


def deferred_bwd(tb):
    return L.vhap_deferred_shade_bwd(_p(ns.clip), _p(ns.tri), _p(ns.vn), _p(ns.uv), _p(ns.tri_uv), _p(ns.albedo_tex), _p(ns.mips), T, T,
                                     _p(tr.lights), _p(ns.sh_const), _p(ns.rast), *ns._upstream(), _p(ns.keep) if ns.disturb_on else 0,
                                     _p(ns.c_reg) if ns.want_reg else 0, _p(acc[12:16]) if ns.want_reg else 0, B, V, ns.uv.shape[0], F, H, W,
                                     _p(ns.texc), _p(ns.texd), _p(ns.d_albedo), _p(ns.d_normal), _p(ns.d_texc), _p(ns.d_texd), 0,
                                     _p(ns.def_work), ns.def_work.numel(), _p(ns.texbin_work) if tb else 0, 0, ns.delta_flag, st())


calls = {
    "tex_forward (tex_prep + mips + offset_reg)": lambda: ns._tex_forward(),
    "frame_prep+skin (geometry head)": None,
    "vnormal_fwd": lambda: L.vhap_vnormal_fwd_saved(_p(ns.verts), _p(ns.csr.tri), _p(ns.csr.ptr), _p(ns.csr.idx), B, V, _p(ns.vn), _p(ns.vn_inv), st()),
    "raster_shade bin_only": lambda: raster(1 | 2),
    "raster_shade prebinned (+stats reduce)": lambda: raster(1 | 4),
    "raster_shade whole": lambda: raster(1),
    "disturb": (lambda: L.vhap_disturb_inplace(_p(ns.rgba), _p(ns.cid), ns.ncl, 0, 0, 0, float(ns.rate_fg or 0.0), float(ns.rate_bg or 0.0), _p(ns.rng),
                                               B, H, W, _p(ns.dist_ws), _p(ns.keep), st())) if ns.disturb_on else None,
    "antialias_inplace_fwd": lambda: L.vhap_antialias_inplace_fwd(_p(ns.aa_in), _p(ns.rast), _p(ns.clip), _p(ns.tri), _p(ns.opp), B, H, W, V, F,
                                                                  _p(ns.aa_work), st()),
    "photo_fwd": lambda: L.vhap_photo_fwd(_p(ns.rgba_aa), _p(ns.rgb), B, H, W, _p(acc[16:18]), PRE, st()),
    "photo_fwd_total (sum + energy assembly + antialias colour job: the shipped launch)": (lambda: L.vhap_photo_fwd_total(
        _p(ns.rgba_aa), _p(ns.rgb), B, H, W, _p(acc[16:19]), _p(acc[0:6]), _p(acc[6:7]) if ns.w_lmk else 0, _p(acc[7:10]), _p(acc[20:24]), _p(acc[12:16]),
        ns.w_lmk, ns.w_reg, ns.w_photo, _p(ns.log), _p(ns.d_sum), _p(ns.gmax_bound), _p(ns.photo_work), _p(ns.aa_work) if ns.aa_early_bwd else 0,
        _p(ns.d_delta) if ns.aa_early_bwd else 0, 0, st())) if ns.deferred else None,
    "antialias_photo_bwd + clear": lambda: (L.vhap_antialias_photo_bwd(_p(ns.rgba_aa), _p(ns.rgb), _p(ns.d_sum), _p(ns.rast), _p(ns.clip), _p(ns.tri),
                                                                     _p(ns.opp), _p(ns.aa_work), _p(ns.vert_mask), B, H, W, V, F, _p(ns.d_delta),
                                                                     _p(g["d_clip"]), st()), ns._clear_delta()),
    "deferred_shade_bwd": lambda: deferred_bwd(False),
    "gbuffer_bwd": lambda: L.vhap_gbuffer_bwd(_p(ns.clip), _p(ns.tri), _p(ns.vn), _p(ns.uv), _p(ns.tri_uv), _p(ns.rast), _p(ns.d_normal), _p(ns.d_texc),
                                              _p(ns.d_texd), 0, 0, _p(ns.face_mask), B, V, F, H, W, _p(g["d_clip"]), _p(g["d_vn"]), st()),
    "texture_grad_binned (count+scan+scatter+tile)": lambda: L.vhap_texture_grad_binned(T, T, 3, _p(ns.texc), _p(ns.texd), _p(ns.d_albedo), B, H, W, _p(d_tex),
                                                                                      _p(d_mips), _p(ns.texbin_work), ns.texbin_work.numel(), st()),
    "tex_finish (pyramid gather + TV / residual gradients)": lambda: ns.tex_finish(),
    "tex_finish + Adam (the fused pass of the captured step)": lambda: ns.tex_finish(opt),
    "texbin_sort_ids (count + scan + scatter)": (lambda: L.vhap_texbin_sort_ids(_p(ns.tile_ids), _p(ns.keep) if ns.disturb_on else 0, T, T, B, H, W,
                                                                                _p(ns.texbin_work), ns.texbin_work.numel(), st())) if ns.tb_ids else None,
    "texture_grad_binned_sorted (texgrad_tile)": (lambda: L.vhap_texture_grad_binned_sorted(T, T, 3, _p(ns.texc), _p(ns.texd), _p(ns.d_albedo), B, H, W,
                                                                                           _p(d_tex), _p(d_mips), _p(ns.texbin_work),
                                                                                           ns.texbin_work.numel(), _p(ns.gmax_bound), st())) if ns.tb_ids else None,
    "verts_bwd_fused .. frame_prep_bwd (geometry tail as shipped)": lambda: ns._bwd_vertex_stage(),
    "adam (texture only)": lambda: opt.step(only=(tr.tex_extra,), advance=False),
    "adam (all but texture)": lambda: opt.step(skip=(tr.tex_extra,), advance=False),
    "geometry tail (vnormal_bwd .. frame_prep_bwd)": None,
}


def geometry_tail():
    L.vhap_vnormal_bwd_saved(_p(ns.verts), _p(ns.csr.tri), _p(ns.csr.ptr), _p(ns.csr.idx), _p(ns.vn), _p(ns.vn_inv), _p(g["d_vn"]), B, V, 1, _p(ns.vn_scratch), _p(g["d_verts"]), st())
    L.vhap_transform_bwd(_p(ns.verts), _p(ns.mvp), _p(g["d_clip"]), B, V, 1, _p(g["d_verts"]), _p(ns.d_mvp), st())
    ns._bwd_params()


def geometry_head():
    fb, fm, so = ns.fb, ns.fm, tr.static_offset
    ns._camera_forward()
    L.vhap_frame_prep_fwd(_p(ns.ts), _p(tr.shape), _p(tr.expr), _p(tr.rotation), _p(tr.translation), _p(tr.neck_pose), _p(tr.jaw_pose), _p(tr.eyes_pose),
                          _p(fm.JT), _p(fm.JS), _p(fm.jreg_idx), _p(fm.jreg_w), fm.jreg_n, _p(so), fm.parents, ns.weights, B, ns.Bp, ns.N, ns.NS, ns.NE,
                          ns.J, fb.Kp, V, _p(ns.coef), _p(ns.A), _p(ns.transl), _p(ns.Jrest), _p(acc), PRE, st())
    L.vhap_flame_skin_clip_fwd(_p(ns.coef), _p(fb.basis), _p(ns.A), _p(fb.w), _p(fb.templ), _p(so), _p(ns.transl), _p(ns.mvp), B, V, fb.Vp, fb.K, fb.Kb,
                               fb.Kp, _p(ns.verts), _p(ns.v_shaped), _p(ns.v_posed), _p(ns.clip), 0, st())


calls["geometry tail (vnormal_bwd .. frame_prep_bwd)"] = geometry_tail
calls["frame_prep+skin (geometry head)"] = geometry_head
only = [s for s in args.only.split(",") if s]
if args.debug_flags:
    _lib.debug_set_flags(args.debug_flags)
print(f"config {args.config}: B={B} {H}x{W} T={T}; un-contended, {args.reps} repetitions back to back")
for name, fn in calls.items():
    if fn is None or (only and not any(o in name for o in only)):
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if args.cold:
        if "scratch" not in globals():
            scratch = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        ts = []
        for _ in range(args.reps):
            scratch.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        print(f"{float(np.median(ts)):9.1f} us  {name}   (cold caches, median of {args.reps})")
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    e1.synchronize()
    print(f"{e0.elapsed_time(e1) / args.reps * 1e3:9.1f} us  {name}")
