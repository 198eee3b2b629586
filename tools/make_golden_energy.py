#!/usr/bin/env python3
"""Golden vectors for the ENERGY terms of the hot path, produced by the REFERENCE's own code (runs only in the build container).

Imported unmodified from /root/reference (third-party imports that are absent here -- tyro, tensorboard, torchvision, matplotlib,
nvdiffrast -- are stubbed: none of them is touched by the functions called):
  vhap/config/base.py   -> BaseTrackingConfig(): every default weight / stage list (pins vhap_amd.config field by field)
  vhap/model/tracker.py -> FlameTracker.compute_lmk_energy (:347-389), compute_regularization_energy (:480-605) with its helpers
                           compute_pose_smooth_energy / compute_joint_smooth_energy / compute_expr_smooth_energy /
                           compute_joint_L2_energy / compute_laplacian_smoothing_loss / scale_vertex_weights_by_region (:607-690);
                           compute_photometric_energy (:391-478) and the WHOLE GlobalTracker.compute_energy (:692-750) with its backward
                           (sections 5 and 9: the four nvdiffrast ops inside are the oracle's restatements, the colour-disturbance
                           draws are replayed); get_train_parameters / configure_optimizer / initialize_next_timtestep (section 8)
  vhap/util/render_nvdiffrast.py -> NVDiffRenderer.world_to_clip / world_to_ndc / compute_v_normals / compute_face_normals; rasterize /
                           render_rgba with the four nvdiffrast ops replaced by the oracle's (section 5 below)
  vhap/model/flame.py   -> FlameHead.forward (:571-646) and FlameMask.construct_vid_table / process_face_mask / process_face_clusters /
                           get_vid_by_region / get_fid_by_region (:940-1033), on objects created without __init__
The tracker methods are called on a FlameTracker created WITHOUT __init__ (it would need FLAME assets and a CUDA context) whose
attributes are filled from the synthetic FLAME-topology model of this repo; what they compute from those attributes is the
reference's arithmetic.  Inputs that come from parts of the reference which cannot run here are taken from the oracle restatement and
stored alongside: canonical vertices (FLAME assets), the shaded `diffuse_detach_normal` image (nvdiffrast) and the uniform Laplacian
(pytorch3d) -- so these vectors pin the energy formulas, not those three inputs.
Output: tests/golden/energy_golden.npz (inputs + reference outputs, float64).
"""
import dataclasses
import importlib.util
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "tests", "golden", "energy_golden.npz")
sys.path.insert(0, ROOT)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    class _Any:
        def __init__(self, *a, **k): pass
        def __getattr__(self, k): return _Any()
        def __call__(self, *a, **k): return _Any()
    stub("tyro", cli=lambda *a, **k: None, to_yaml=lambda *a, **k: "", conf=_Any(), extras=_Any())
    stub("nvdiffrast"); stub("nvdiffrast.torch")
    stub("torchvision"); stub("torchvision.transforms"); stub("torchvision.transforms.functional")
    stub("matplotlib", cm=_Any()); stub("matplotlib.cm")
    stub("torch.utils.tensorboard", SummaryWriter=_Any)
    for pkg, sub in (("vhap", ""), ("vhap.util", "/util"), ("vhap.config", "/config"), ("vhap.model", "/model")):
        stub(pkg).__path__ = [f"{REF}/vhap{sub}"]            # real sub-modules load on demand; the stubs below take precedence
    stub("vhap.util.log", get_logger=lambda name: _Any())
    stub("vhap.util.visualization", plot_landmarks_2d=None)
    stub("pytorch3d"); stub("pytorch3d.io", load_obj=None); stub("pytorch3d.structures"); stub("pytorch3d.structures.meshes", Meshes=_Any)
    base = load(f"{REF}/vhap/config/base.py", "vhap.config.base")
    load(f"{REF}/vhap/model/lbs.py", "vhap.model.lbs")
    load(f"{REF}/vhap/util/mesh.py", "vhap.util.mesh")
    flame = load(f"{REF}/vhap/model/flame.py", "vhap.model.flame")
    sys.path.insert(0, REF)
    rn = load(f"{REF}/vhap/util/render_nvdiffrast.py", "vhap.util.render_nvdiffrast")
    tracker = load(f"{REF}/vhap/model/tracker.py", "ref_tracker")
    return base, rn, tracker, flame


def config_to_dict(cfg, prefix=""):
    out = {}
    for f in dataclasses.fields(cfg):
        v = getattr(cfg, f.name)
        if dataclasses.is_dataclass(v):
            out.update(config_to_dict(v, prefix + f.name + "."))
        else:
            out[prefix + f.name] = v
    return out


class _IndexDataset(torch.utils.data.Dataset):
    """stands in for the reference's VideoDataset in the stage-scheduler trace: item i = {"timestep_index": i}"""
    batchify_all_views = False

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"timestep_index": i}


def main():
    base, rn, T, FL = load_reference()
    from oracle import energy_ref, torch_ref as R
    from vhap_amd.synthetic import make_flame_model, make_scene_params, make_texture, monocular_camera
    dt = torch.float64
    torch.manual_seed(0)
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference moves helper tensors with .cuda(); there is no GPU here

    # ---- 1. configuration defaults ------------------------------------------------------------------------------------------
    import typing
    from pathlib import Path

    def default_instance(cls):           # the reference builds its config through tyro; here: defaults, sub-configs recursively
        hints = typing.get_type_hints(cls)
        kw = {}
        for f in dataclasses.fields(cls):
            if f.default is not dataclasses.MISSING or f.default_factory is not dataclasses.MISSING:
                continue
            t = hints[f.name]
            kw[f.name] = default_instance(t) if dataclasses.is_dataclass(t) else (Path(".") if t is Path else "x")
        return cls(**kw)
    rcfg = default_instance(base.BaseTrackingConfig)
    cfg_items = {k: v for k, v in config_to_dict(rcfg).items()
                 if isinstance(v, (int, float, bool, str, type(None), list, tuple)) and not k.startswith(("data.root", "exp.", "data.sequence"))}

    ners = load(f"{REF}/vhap/config/nersemble.py", "vhap.config.nersemble")
    ncfg = default_instance(ners.NersembleTrackingConfig)
    ners_items = {k: v for k, v in config_to_dict(ncfg).items()
                  if isinstance(v, (int, float, bool, str, type(None), list, tuple)) and not k.startswith(("data.root", "exp.", "data.sequence"))}

    # ---- 2. a small state on the synthetic FLAME-topology model ----------------------------------------------------------------
    model, topo = make_flame_model(0)
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    N, H, W, Tt = 4, 96, 80, 32
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g, dtype=dt) * sc
    V = tm["v_template"].shape[0]
    P = dict(shape=rnd(300, sc=0.3), expr=rnd(N, 100, sc=0.3), rotation=rnd(N, 3, sc=0.1), neck_pose=rnd(N, 3, sc=0.05),
             jaw_pose=rnd(N, 3, sc=0.1), eyes_pose=rnd(N, 6, sc=0.1), translation=rnd(N, 3, sc=0.02), tex_extra=rnd(3, Tt, Tt, sc=0.05),
             lights=rnd(9, 3, sc=0.1), static_offset=rnd(1, V, 3, sc=1e-3), focal_length=torch.tensor([2.3], dtype=dt))
    P["translation"][:, 2] += 0.4
    ts = np.array([1, 3, 0])                                   # includes timestep 0 (its 'previous' frame clamps to itself)
    B = len(ts)
    tex_painted = torch.from_numpy(make_texture(0, Tt)).to(dt)[None]
    uvmask = (torch.rand(Tt, Tt, generator=g) < 0.3).to(dt)
    lmk2d = torch.cat([torch.rand(B, 70, 2, generator=g, dtype=dt) * torch.tensor([W, H], dtype=dt), (torch.rand(B, 70, 1, generator=g) < 0.9).to(dt)], -1)
    diffuse = (torch.rand(B, 3, H, W, generator=g, dtype=dt) * 1.3).float().double()      # stands in for result_dict['diffuse_detach_normal'] (nvdiffrast); stored as fp32
    verts, v_cano, lmks = R.flame_forward(tm, P["shape"][None].expand(B, -1), P["expr"][ts], P["rotation"][ts], P["neck_pose"][ts],
                                          P["jaw_pose"][ts], P["eyes_pose"][ts], P["translation"][ts], static_offset=P["static_offset"])
    f = P["focal_length"] * max(H, W)
    K = torch.stack([f, f, torch.full_like(f, 0.5 * W), torch.full_like(f, 0.5 * H)], dim=1)
    RT = torch.eye(3, 4, dtype=dt); RT[2, 3] = -1
    RT = RT[None].expand(B, -1, -1).contiguous()
    Lap = energy_ref._laplacian(V, topo).to(dt)

    # ---- 3. the reference's renderer helpers -------------------------------------------------------------------------------------
    rend = rn.NVDiffRenderer.__new__(rn.NVDiffRenderer)
    # (the reference builds its projection matrices in fp32: these helpers and the landmark energy run in fp32)
    clip_ref = rend.world_to_clip(verts.float(), RT.float(), K.float(), (H, W)).double()
    ndc_ref = rend.world_to_ndc(lmks.float(), RT.float(), K.float(), (H, W), flip_y=True).double()
    real_tensor = torch.tensor
    rn.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    rn.torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})     # the reference hard-codes device='cuda'
    faces = tm["faces"].long()
    # collapse a vertex and its two-ring to one point: the vertex and its one-ring then have only degenerate faces -> the (0,0,1) fallback
    fnp = tm["faces"].numpy()
    ring = {100}
    for _ in range(2):
        ring |= set(fnp[np.isin(fnp, list(ring)).any(1)].ravel().tolist())
    collapse = np.array(sorted(ring))
    vflat = verts.clone(); vflat[0, collapse] = 0
    vn_ref = rend.compute_v_normals(vflat.float(), faces).double()
    fn_ref = rend.compute_face_normals(verts.float(), faces).double()
    rn.torch = torch

    # ---- 4. the reference's tracker energies on a tracker object without __init__ ----------------------------------------------------
    tr = object.__new__(T.FlameTracker)
    tr.cfg, tr.device = rcfg, "cpu"
    tr.render = rend
    tr.opt_dict = defaultdict(bool, {k: True for k in ("pose", "joints", "expr", "shape", "texture", "lights", "static_offset")})
    tr.n_timesteps = N
    for k in ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset"):
        setattr(tr, k, P[k])
    tr.dynamic_offset = None
    tr.lights_uniform = torch.zeros(9, 3, dtype=dt)
    tr.get_albedo = lambda: tex_painted + P["tex_extra"][None]
    vid = lambda regions: torch.from_numpy(topo.get_vid_by_region(list(regions))).long()
    tr.flame = types.SimpleNamespace(mask=types.SimpleNamespace(get_vid_by_region=vid), laplacian_matrix=Lap,
                                     laplacian_matrix_negate_diag=Lap - 2 * torch.diag(torch.diag(Lap)))
    tr.flame_uvmask = types.SimpleNamespace(get_uvmask_by_region=lambda regions: uvmask)
    sample = {"rgb": torch.zeros(B, 3, H, W, dtype=dt), "lmk2d": lmk2d, "intrinsic": K, "extrinsic": RT}
    out = {}
    for tag, dis in (("jaw", False), ("nojaw", True)):
        for always in (True, False):
            rcfg.w.always_enable_jawline_landmarks = always
            e, rd = tr.compute_lmk_energy({k: v.float() for k, v in sample.items()}, lmks.float(), disable_jawline_landmarks=dis)
            out[f"lmk_{tag}_always{int(always)}"] = e
    rcfg.w.always_enable_jawline_landmarks = True
    for stage in ("rgb_init_offset", "rgb_global_tracking"):
        log = tr.compute_regularization_energy({"diffuse_detach_normal": diffuse}, verts, v_cano, lmks, None, ts, stage)
        for k, v in log.items():
            out[f"reg/{stage}/{k}"] = v
    out["joint_l2"] = tr.compute_joint_L2_energy(ts)

    # ---- 5. the photometric energy: compute_photometric_energy (:391-478) -> FlameTracker.render_rgba (:305-335) -> NVDiffRenderer.rasterize /
    #         render_rgba (render_nvdiffrast.py:216-245, 354-484), with ONLY the four nvdiffrast ops replaced by the oracle's restatements.
    #         Pins everything around the ops: camera chain, normals, region detach, SH shading, compositing / flips, the colour disturbance
    #         (random draws replayed below), boundary detach, the loss normalisation -- values and gradients.
    import oracle
    dr = sys.modules["nvdiffrast.torch"]
    B2, H2, W2, T2 = 2, 64, 56, 32
    faces_l, faces_uv_l = tm["faces"].long(), tm["faces_uv"].long()
    opp = torch.from_numpy(topo.opp.astype(np.int64))
    stash = {}

    def dr_rasterize(glctx, clip, tri, size):
        rast_np, _ = oracle.rasterize(clip.detach().float().numpy(), tri.numpy().astype(np.int32), tuple(size))
        stash["tid"] = torch.from_numpy(rast_np[..., 3].astype(np.int64) - 1)
        return R.rast_from_ids(clip, tri.long(), stash["tid"], tuple(size))
    dr_calls = set()                                                  # how the reference calls into nvdiffrast: (name, #positional, keyword names)

    def traced(name, fn):
        def w(*a, **k):
            dr_calls.add((name, len(a), tuple(sorted(k))))
            return fn(*a, **k)
        return w
    dr.rasterize = traced("rasterize", dr_rasterize)
    dr.interpolate = traced("interpolate", lambda attr, rast, tri, rast_db=None, diff_attrs=None: R.interpolate(attr, rast, tri.long(), rast_db, diff_attrs))
    dr.texture = traced("texture", lambda tex, uv, uv_da=None, filter_mode="linear", max_mip_level=None: R.texture(tex, uv, uv_da, filter_mode))
    dr.antialias = traced("antialias", lambda color, rast, pos, tri: R.antialias(color, rast, pos, tri.long(), opp))
    dr.RasterizeCudaContext = traced("RasterizeCudaContext", lambda: None)
    rn.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    rn.torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    rend2 = rn.NVDiffRenderer(use_opengl=False, lighting_type="SH", lighting_space="world", disturb_rate_fg=0.5, disturb_rate_bg=0.5,
                              fid2cid=torch.from_numpy(topo.fid2cid[1:].astype(np.int64)))
    f32 = torch.float32
    sp = make_scene_params(B2, 5, (H2, W2))
    q = lambda k: torch.from_numpy(sp[k]).to(dt)
    v2, _, _ = R.flame_forward(tm, q("shape")[None].expand(B2, -1), q("expr"), q("rotation"), q("neck_pose"), q("jaw_pose"), q("eyes_pose"),
                               q("translation"))
    K2n, RT2n = monocular_camera(B2, (H2, W2))
    photo_in = dict(verts=v2.to(f32).numpy(), K=K2n[:1].astype(np.float32),      # [1,4] like fill_cam_params_into_sample (tracker.py:150-156)
                     RT=RT2n.astype(np.float32),
                    tex_painted=make_texture(1, T2).astype(np.float32), tex_extra=(rnd(3, T2, T2, sc=0.05)).to(f32).numpy(),
                    lights=(rnd(9, 3, sc=0.1) + torch.tensor([[3.5, 3.5, 3.5]] + [[0.0] * 3] * 8, dtype=dt)).to(f32).numpy(),
                    rgb=torch.rand(B2, 3, H2, W2, generator=g).to(f32).numpy())
    tr2 = object.__new__(T.FlameTracker)
    tr2.cfg, tr2.device, tr2.render, tr2.image_size = rcfg, "cpu", rend2, (H2, W2)
    fid = lambda regions: torch.from_numpy(topo.get_fid_by_region(list(regions))).long()
    tr2.flame = types.SimpleNamespace(mask=types.SimpleNamespace(get_vid_by_region=vid, get_fid_by_region=fid), textures_idx=faces_uv_l,
                                      verts_uvs=tm["verts_uvs"].to(f32))
    photo_out = {}
    for stage, seed in ((None, None), ("rgb_global_tracking", 1234)):
        verts_l = torch.from_numpy(photo_in["verts"]).requires_grad_()
        tex_l = torch.from_numpy(photo_in["tex_extra"]).requires_grad_()
        lights_l = torch.from_numpy(photo_in["lights"]).requires_grad_()
        tr2.lights = lights_l
        albedos = (torch.from_numpy(photo_in["tex_painted"])[None] + tex_l[None]).expand(B2, -1, -1, -1)
        sample2 = {"rgb": torch.from_numpy(photo_in["rgb"]), "intrinsic": torch.from_numpy(photo_in["K"]), "extrinsic": torch.from_numpy(photo_in["RT"])}
        rend2.clear_cache()
        rast_dict = tr2.rasterize_flame(sample2, verts_l, faces_l, train_mode=True)
        if seed is not None:
            torch.manual_seed(seed)
        E, rd = tr2.compute_photometric_energy(sample2, verts_l, faces_l, albedos, rast_dict, None, stage)
        E.backward()
        tag = "eval" if stage is None else stage
        photo_out[f"{tag}/E"] = E.detach().double().numpy()
        photo_out[f"{tag}/rgba"] = rd["rgba"].detach().numpy()
        photo_out[f"{tag}/d_verts"] = verts_l.grad.numpy()
        photo_out[f"{tag}/d_tex_extra"] = tex_l.grad.numpy()
        photo_out[f"{tag}/d_lights"] = lights_l.grad.numpy()
        photo_out[f"{tag}/diffuse_detach_normal"] = rd["diffuse_detach_normal"].detach().numpy()
        if seed is not None:
            # replay the reference's draws (render_nvdiffrast.py:428-457) for the oracle's injected-randomness interface
            torch.manual_seed(seed)
            like = torch.zeros(B2, H2, W2, 1, dtype=f32)
            w_fg = (torch.rand_like(like) < 0.5).int()
            w_bg = (torch.rand_like(like) < 0.5).int()
            cid = rend2.fid2cid[(stash["tid"] + 1)]
            ncl = int(rend2.fid2cid.max()) + 1
            idx = np.zeros((ncl, B2 * H2 * W2), np.uint16)
            for i in range(ncl):
                n_i = int((cid == i).sum())
                if i != 1 and n_i > 0:
                    idx[i] = torch.randint(0, n_i, (B2 * H2 * W2,)).numpy()
            photo_out["disturb/w_fg"], photo_out["disturb/w_bg"], photo_out["disturb/idx"] = w_fg.numpy(), w_bg.numpy(), idx
        photo_out["tid"] = stash["tid"].numpy().astype(np.int32)
    rn.torch = torch
    for k, v in photo_out.items():
        if k.endswith("/E"):
            print(f"  photo {k:40s} {float(v):.10g}")

    # ---- 6. FlameHead.forward (flame.py:571-646) on an object without __init__ carrying the synthetic model's buffers ---------------------
    fh = object.__new__(FL.FlameHead)
    torch.nn.Module.__init__(fh)
    fh.dtype = dt
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        fh.register_buffer(k, tm[k])
    fh.register_buffer("parents", tm["parents"].long())
    fh.register_buffer("faces", tm["faces"].long())
    fh.register_buffer("full_lmk_faces_idx", tm["lmk_faces_idx"].long()[None])
    fh.register_buffer("full_lmk_bary_coords", tm["lmk_bary_coords"][None])
    dyn = rnd(B, V, 3, sc=5e-4).float().double()               # (stored as fp32)
    fv, fcano, flm = FL.FlameHead.forward(fh, P["shape"][None].expand(B, -1), P["expr"][ts], P["rotation"][ts], P["neck_pose"][ts], P["jaw_pose"][ts],
                                           P["eyes_pose"][ts], P["translation"][ts], return_verts_cano=True,
                                           static_offset=P["static_offset"], dynamic_offset=dyn)
    pick = np.sort(np.random.default_rng(0).choice(V, 600, replace=False))     # a vertex subset keeps the fixture small
    flame_out = dict(dynamic_offset=dyn.float().numpy(), pick=pick, verts=fv[:, pick].numpy(), verts_cano=fcano[:, pick].numpy(), lmks=flm.numpy())

    # ---- 7. FlameMask (flame.py:940-1040): vertex regions -> face regions, fid2cid, region look-ups, on the synthetic topology's regions -----
    fm = object.__new__(FL.FlameMask)
    torch.nn.Module.__init__(fm)
    fm.num_verts, fm.num_faces = topo.num_verts, topo.num_faces
    fm.v = FL.BufferContainer()
    for name, vids in topo.v_regions.items():
        fm.v.register_buffer(name, torch.from_numpy(np.asarray(vids)).long())
    fm.construct_vid_table()
    fm.process_face_mask(torch.from_numpy(topo.faces.astype(np.int64)))
    fm.process_face_clusters(list(topo.tex_clusters))
    mask_out = {"fid2cid": fm.fid2cid.numpy()}
    names = sorted(k for k, _ in fm.f)
    mask_out["f_names"] = np.array(names)
    for k in names:
        mask_out[f"f/{k}"] = fm.f.get_buffer(k).numpy()
    for stage_name in ("rgb_init_texture", "rgb_init_all", "rgb_global_tracking"):
        st = rcfg.pipeline[stage_name]
        mask_out[f"fid/{stage_name}"] = fm.get_fid_by_region(list(st.align_texture_except)).numpy()
        mask_out[f"vid/{stage_name}"] = fm.get_vid_by_region(list(st.align_boundary_except)).numpy()

    # ---- 8. host logic of GlobalTracker: which tensors a stage trains (get_train_parameters :1465-1513), the Adam parameter groups and
    #         learning rates (configure_optimizer :159-211), seeding the next timesteps (initialize_next_timtestep :1515-1529) -------------
    import json
    tr3 = object.__new__(T.GlobalTracker)
    tr3.cfg, tr3.calibrated, tr3.n_timesteps = rcfg, False, 9
    pnames = ("focal_length", "shape", "tex_pca", "tex_extra", "static_offset", "lights", "translation", "rotation", "eyes_pose", "neck_pose",
              "jaw_pose", "expr", "dynamic_offset")
    for i, k in enumerate(pnames):
        setattr(tr3, k, torch.full((9, 3), float(i), requires_grad=True))
    name_of = lambda t: next(k for k in pnames if getattr(tr3, k) is t)
    host = {}
    for stage_name in rcfg.pipeline.__dict__:
        params = tr3.get_train_parameters(stage_name)
        opt = tr3.configure_optimizer(params, lr_scale=0.5)
        host[stage_name] = {"params": {k: [name_of(t) for t in v] for k, v in params.items() if len(v)},
                            "groups": [[sorted(name_of(t) for t in gr["params"]), gr["lr"]] for gr in opt.param_groups],
                            "opt_dict": sorted(k for k, v in tr3.opt_dict.items() if v)}
    g3 = torch.Generator().manual_seed(9)
    for k in ("translation", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "expr", "dynamic_offset"):
        setattr(tr3, k, torch.randn(9, 4, generator=g3, dtype=dt))
    before = {k: getattr(tr3, k).clone().numpy() for k in ("translation", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "expr")}
    tr3.initialize_next_timtestep(torch.tensor([2, 3, 4]))
    after = {k: getattr(tr3, k).numpy() for k in before}

    # ---- 9. the WHOLE energy: GlobalTracker.compute_energy (:692-750) = forward_flame (:213-235) + fill_cam_params_into_sample (:141-157) +
    #         landmark + photometric + regularisation terms, and its backward, for four stage kinds; fp32 like the reference; the nvdiffrast
    #         ops inside are the oracle's, the colour-disturbance draws are replayed as in section 5 -------------------------------------------
    rn.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    rn.torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    H9, W9 = H, W
    fh32 = object.__new__(FL.FlameHead)
    torch.nn.Module.__init__(fh32)
    fh32.dtype = f32
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        fh32.register_buffer(k, tm[k].to(f32))
    fh32.register_buffer("parents", tm["parents"].long())
    fh32.register_buffer("faces", faces_l)
    fh32.register_buffer("full_lmk_faces_idx", tm["lmk_faces_idx"].long()[None])
    fh32.register_buffer("full_lmk_bary_coords", tm["lmk_bary_coords"].to(f32)[None])
    fh32.mask = types.SimpleNamespace(get_vid_by_region=vid, get_fid_by_region=fid)
    fh32.textures_idx, fh32.verts_uvs = faces_uv_l, tm["verts_uvs"].to(f32)
    fh32.laplacian_matrix = Lap.to(f32)
    fh32.laplacian_matrix_negate_diag = (Lap - 2 * torch.diag(torch.diag(Lap))).to(f32)
    rend_calls = set()                                                # how the reference TRACKER calls its renderer (boundary b1)
    for mname in ("rasterize", "render_rgba", "world_to_ndc", "clear_cache"):
        def wrap(fn, mname=mname):
            def w(*a, **k):
                rend_calls.add((mname, len(a), tuple(sorted(k))))
                return fn(*a, **k)
            return w
        setattr(rend2, mname, wrap(getattr(rend2, mname)))
    rgb9 = torch.rand(B, 3, H9, W9, generator=g).to(f32)
    full_out = {"rgb": rgb9.numpy()}
    leafs = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "tex_extra", "lights", "static_offset", "focal_length")
    for stage, seed in ((None, None), ("lmk_init_all", None), ("rgb_init_offset", 77), ("rgb_global_tracking", 78)):
        tr9 = object.__new__(T.GlobalTracker)
        tr9.cfg, tr9.device, tr9.render, tr9.image_size, tr9.calibrated, tr9.n_timesteps = rcfg, "cpu", rend2, (H9, W9), False, N
        tr9.flame = fh32
        for k in leafs:
            setattr(tr9, k, P[k].to(f32).clone().requires_grad_())
        tr9.dynamic_offset, tr9.tex_pca = None, None
        tr9.RT = torch.eye(3, 4); tr9.RT[2, 3] = -1
        tr9.lights_uniform = torch.zeros(9, 3)
        tr9.flame_tex_painted = lambda: tex_painted.to(f32)
        tr9.flame_uvmask = types.SimpleNamespace(get_uvmask_by_region=lambda regions: uvmask.to(f32))
        tr9.opt_dict = defaultdict(bool)
        if stage is not None:
            tr9.get_train_parameters(stage)
        sample9 = {"rgb": rgb9, "lmk2d": lmk2d.to(f32), "timestep_index": ts}
        rend2.clear_cache()
        tr9.fill_cam_params_into_sample(sample9)
        if seed is not None:
            torch.manual_seed(seed)
        E9, log9, *_ = tr9.compute_energy(sample9, stage=stage)
        E9.backward()
        tag = "eval" if stage is None else stage
        for k, v in log9.items():
            full_out[f"{tag}/log/{k}"] = np.asarray(float(v))
        for k in leafs:
            gk = getattr(tr9, k).grad
            full_out[f"{tag}/grad/{k}"] = (torch.zeros_like(getattr(tr9, k)) if gk is None else gk).numpy()
        if "tid" in stash and (stage is None or stage.startswith("rgb")):
            full_out[f"{tag}/tid"] = stash["tid"].numpy().astype(np.int16)
            full_out[f"{tag}/coverage"] = np.asarray(float((stash["tid"] >= 0).float().mean()))
        if seed is not None:
            torch.manual_seed(seed)
            like = torch.zeros(B, H9, W9, 1, dtype=f32)
            w_fg = (torch.rand_like(like) < 0.5).int()
            w_bg = (torch.rand_like(like) < 0.5).int()
            cid = rend2.fid2cid[(stash["tid"] + 1)]
            ncl = int(rend2.fid2cid.max()) + 1
            idx = np.zeros((ncl, B * H9 * W9), np.uint16)
            for i in range(ncl):
                n_i = int((cid == i).sum())
                if i != 1 and n_i > 0:
                    idx[i] = torch.randint(0, n_i, (B * H9 * W9,)).numpy()
            full_out[f"{tag}/w_fg"], full_out[f"{tag}/w_bg"], full_out[f"{tag}/idx"] = w_fg.numpy().astype(np.uint8), w_bg.numpy().astype(np.uint8), idx
        print(f"  full {tag:22s} total {float(E9):.8g}  terms {sorted(log9)}")
    rn.torch = torch

    # ---- 10. the on-disk contract: save_result (:1152-1218) -> tracked_flame_params.npz keys / shapes / dtypes ---------------------------
    import tempfile
    from pathlib import Path as _Path
    tr10 = object.__new__(T.GlobalTracker)
    tr10.cfg, tr10.calibrated, tr10.image_size = rcfg, False, (H, W)
    for k in leafs:
        setattr(tr10, k, P[k].to(f32))
    tr10.dynamic_offset, tr10.tex_pca = None, torch.zeros(5)
    tr10.dataset = types.SimpleNamespace(timestep_ids=[f"{i:05d}" for i in range(N)])
    tr10.timestep = N
    with tempfile.TemporaryDirectory() as d_:
        tr10.out_dir = _Path(d_)
        tr10.save_result(epoch=3)
        rep = np.load(_Path(d_) / "tracked_flame_params_3.npz")
        schema = {k: [list(rep[k].shape), str(rep[k].dtype)] for k in rep.files}

    # ---- 11. the stage scheduler: GlobalTracker.optimize (:1343-1389) call trace, and optimize_stage's (:1391-1416) step count / learning-rate
    #          schedule, with the per-step work stubbed out --------------------------------------------------------------------------------------
    trace = []
    tr11 = object.__new__(T.GlobalTracker)
    tr11.cfg, tr11.n_timesteps, tr11.logger = rcfg, 20, types.SimpleNamespace(info=lambda *a, **k: None)
    tr11.dataset = _IndexDataset(20)

    def rec_stage(stage, sample=None, dataloader=None, lr_scale=1.0):
        if sample is not None:
            trace.append(["stage", stage, [int(t) for t in sample["timestep_index"]], float(lr_scale)])
        else:
            trace.append(["stage", stage, {"shuffle": isinstance(dataloader.sampler, torch.utils.data.RandomSampler),
                                           "batch_size": dataloader.batch_size, "batches": len(dataloader)}, float(lr_scale)])
    tr11.optimize_stage = rec_stage
    tr11.initialize_next_timtestep = lambda ts: trace.append(["init_next", [int(t) for t in ts]])
    tr11.evaluate = lambda make_visualization=True, epoch=0: trace.append(["evaluate", int(epoch)])
    tr11.optimize()
    sched = {}
    for stage_name, kw in (("lmk_init_rigid", dict(sample={"timestep_index": torch.arange(3)})),
                           ("rgb_global_tracking", dict(dataloader=[{"timestep_index": torch.arange(2)}, {"timestep_index": torch.arange(2, 3)}],
                                                        lr_scale=0.1))):
        tr12 = object.__new__(T.GlobalTracker)
        tr12.cfg, tr12.calibrated, tr12.n_timesteps, tr12.logger = rcfg, False, 9, types.SimpleNamespace(info=lambda *a, **k: None)
        for i, k in enumerate(pnames):
            setattr(tr12, k, torch.full((9, 3), float(i), requires_grad=True))
        calls = []
        tr12.optimize_iter = lambda sample, optimizer, stage, calls=calls: calls.append([gr["lr"] for gr in optimizer.param_groups])
        tr12.evaluate = lambda make_visualization=True, epoch=0, calls=calls: calls.append(["evaluate", int(epoch)])
        tr12.optimize_stage(stage_name, **kw)
        sched[stage_name] = calls

    save = {f"P/{k}": v.numpy() for k, v in P.items()}
    save["sched/json"] = np.array(json.dumps({"trace": trace, "stage_calls": sched}))
    save["rend_calls/json"] = np.array(json.dumps(sorted([n, k, list(kw)] for n, k, kw in rend_calls)))
    save["dr_calls/json"] = np.array(json.dumps(sorted([n, k, list(kw)] for n, k, kw in dr_calls)))
    save["schema/json"] = np.array(json.dumps(schema))
    save.update({f"full/{k}": v for k, v in full_out.items()})
    save["host/json"] = np.array(json.dumps(host))
    save.update({f"host/before/{k}": v for k, v in before.items()})
    save.update({f"host/after/{k}": v for k, v in after.items()})
    save.update({f"flame/{k}": v for k, v in flame_out.items()})
    save.update({f"mask/{k}": v for k, v in mask_out.items()})
    save.update({f"photo_in/{k}": v for k, v in photo_in.items()})
    save.update({f"photo_out/{k}": v for k, v in photo_out.items()})
    save.update(ts=ts, tex_painted=tex_painted.numpy(), uvmask=uvmask.numpy().astype(np.uint8), lmk2d=lmk2d.numpy(), diffuse=diffuse.float().numpy(),
                image_size=np.array([H, W]), collapse=collapse,
                clip_ref=clip_ref.float().numpy(), ndc_ref=ndc_ref.numpy(), vn_ref=vn_ref.float().numpy(),
                fn_ref=fn_ref[:, :500].float().numpy())          # (fp32 results of the reference; face normals: the first 500 faces)
    save.update({f"out/{k}": np.asarray(float(v)) for k, v in out.items()})
    save["ners_keys"] = np.array(sorted(ners_items))
    save["ners_vals"] = np.array([repr(ners_items[k]) for k in sorted(ners_items)])
    save["cfg_keys"] = np.array(sorted(cfg_items))
    save["cfg_vals"] = np.array([repr(cfg_items[k]) for k in sorted(cfg_items)])
    np.savez_compressed(OUT, **save)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes;", len(out), "energy values;", len(cfg_items), "config defaults")
    for k, v in out.items():
        print(f"  {k:45s} {float(v):.10g}")


if __name__ == "__main__":
    main()
