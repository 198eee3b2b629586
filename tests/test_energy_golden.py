"""Energy terms pinned on the REFERENCE's own code (CPU).  tests/golden/energy_golden.npz holds what
FlameTracker.compute_lmk_energy / compute_regularization_energy (+ helpers), NVDiffRenderer.world_to_clip / world_to_ndc /
compute_v_normals / compute_face_normals and BaseTrackingConfig() of /root/reference return on a small seeded state
(tools/make_golden_energy.py).  Checked here: (1) the oracle restatement (fp64: 1e-9; fp32 reference helpers: 1e-5), (2) the product's
host formulation -- vhap_amd.tracker.FlameTracker on a CPU device, fp32: 2e-4 -- which is what the HIP kernels are compared with on the
GPU (tests/test_native_gpu.py), (3) every default of vhap_amd.config that the reference also has; further down: the photometric chain
and the whole GlobalTracker.compute_energy + backward of the reference (the four nvdiffrast ops inside replaced by the oracle's), the FLAME
forward, the region tables and the tracker's host logic."""
import os

import numpy as np
import pytest
import torch

from oracle import energy_ref
from oracle import torch_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "energy_golden.npz"))
OPT = {"pose", "joints", "expr", "shape", "texture", "lights", "static_offset"}
STAGES = ("rgb_init_offset", "rgb_global_tracking")


def _ref(prefix):
    return {k[len(prefix):]: float(G[k]) for k in G.files if k.startswith(prefix)}


def _state(dtype):
    P = {k[2:]: torch.from_numpy(G[k]).to(dtype) for k in G.files if k.startswith("P/")}
    H, W = (int(x) for x in G["image_size"])
    return P, np.asarray(G["ts"]), (H, W)


def _camera(P, B, H, W, dtype):
    f = P["focal_length"] * max(H, W)
    K = torch.stack([f, f, torch.full_like(f, 0.5 * W), torch.full_like(f, 0.5 * H)], dim=1)
    RT = torch.eye(3, 4, dtype=dtype)
    RT[2, 3] = -1
    return K, RT[None].expand(B, -1, -1).contiguous()


def test_oracle_energies_match_reference(flame_model):
    from vhap_amd.config import BaseTrackingConfig
    model, topo = flame_model
    dt = torch.float64
    P, ts, (H, W) = _state(dt)
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    B = len(ts)
    verts, v_cano, lmks = R.flame_forward(tm, P["shape"][None].expand(B, -1), P["expr"][ts], P["rotation"][ts], P["neck_pose"][ts],
                                          P["jaw_pose"][ts], P["eyes_pose"][ts], P["translation"][ts], static_offset=P["static_offset"])
    K, RT = _camera(P, B, H, W, dt)
    # camera chain and normals (the reference computes these in fp32)
    assert float((R.world_to_clip(verts, RT, K, (H, W)) - torch.from_numpy(G["clip_ref"]).double()).abs().max()) < 2e-5
    assert float((R.world_to_ndc(lmks, RT, K, (H, W), flip_y=True) - torch.from_numpy(G["ndc_ref"])).abs().max()) < 2e-5
    faces = tm["faces"].long()
    vflat = verts.clone()
    vflat[0, torch.from_numpy(G["collapse"])] = 0            # a collapsed two-ring: zero normals -> the (0, 0, 1) fallback of the reference
    vn = R.compute_v_normals(vflat, faces)
    ref_vn = torch.from_numpy(G["vn_ref"]).double()
    fallback = (ref_vn == torch.tensor([0.0, 0.0, 1.0], dtype=dt)).all(-1)
    assert int(fallback.sum()) >= 5 and torch.equal(vn[fallback], ref_vn[fallback])
    # (vertices touching the collapsed patch sum near-cancelling sliver normals: ill-conditioned in fp32, excluded)
    fnp = faces.numpy()
    near = np.unique(fnp[np.isin(fnp, G["collapse"]).any(1)])
    keep = torch.ones(vn.shape[1], dtype=torch.bool)
    keep[torch.from_numpy(near)] = False
    assert float((vn - ref_vn)[:, keep].abs().max()) < 5e-4
    fn = R.safe_normalize(torch.cross(verts[:, faces[:, 1]] - verts[:, faces[:, 0]], verts[:, faces[:, 2]] - verts[:, faces[:, 0]], dim=-1))
    assert float((fn[:, :500] - torch.from_numpy(G["fn_ref"]).double()).abs().max()) < 5e-4       # compute_face_normals (:318-330)
    # landmark energy: the four (disable_jawline, always_enable) combinations of tracker.py:371-381
    lmk2d = torch.from_numpy(G["lmk2d"])
    out = _ref("out/")
    for dis in (False, True):
        for always in (True, False):
            use_jaw = not (not always and dis)
            e = float(R.landmark_energy(lmks, lmk2d, RT, K, (H, W), use_jawline=use_jaw))
            want = out[f"lmk_{'nojaw' if dis else 'jaw'}_always{int(always)}"]
            assert abs(e - want) <= 2e-5 * abs(want), (dis, always, e, want)
    # every regulariser / smoothness term, both stage kinds
    w = BaseTrackingConfig().w
    for stage in STAGES:
        log = energy_ref.regularization_energy(P, ts, w, stage, OPT, torch.from_numpy(G["tex_painted"]), torch.from_numpy(G["uvmask"]).to(dt), v_cano,
                                               torch.from_numpy(G["diffuse"]).to(dt), topo, dt)
        ref = _ref(f"out/reg/{stage}/")
        assert set(log) == set(ref), (stage, sorted(log), sorted(ref))
        for k, want in ref.items():
            assert abs(float(log[k].detach()) - want) <= 1e-9 * abs(want) + 1e-15, (stage, k, float(log[k].detach()), want)
    assert abs(float(energy_ref.joint_l2(P["neck_pose"][ts], P["jaw_pose"][ts], P["eyes_pose"][ts], w)) - out["joint_l2"]) < 1e-12


def test_product_host_energies_match_reference(flame_model):
    """The product's torch formulation of the same terms (FlameTracker on a CPU device, fp32) against the reference's numbers."""
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.synthetic import make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    P, ts, (H, W) = _state(torch.float32)
    N, Tt = P["expr"].shape[0], P["tex_extra"].shape[-1]
    cfg = BaseTrackingConfig()
    cfg.device = "cpu"
    cfg.model.tex_resolution = Tt
    tr = GlobalTracker(cfg, model, topo, make_texture(0, Tt), {"rgb": torch.zeros(N, 3, H, W), "lmk2d": torch.zeros(N, 70, 3)})
    with torch.no_grad():
        for k, v in P.items():
            getattr(tr, k).copy_(v.reshape(getattr(tr, k).shape))
    tr.opt_dict.update({k: True for k in OPT})
    mask = torch.from_numpy(G["uvmask"]).float()
    tr._uvmask_res = lambda: mask
    assert torch.allclose(tr.flame_tex_painted(), torch.from_numpy(G["tex_painted"]).float(), atol=1e-6)
    verts, v_cano, lmks, _ = tr.forward_flame(ts)
    for stage in STAGES:
        log = tr.compute_regularization_energy({"diffuse_detach_normal": torch.from_numpy(G["diffuse"]).float()}, verts, v_cano, lmks, None, ts,
                                               stage)
        ref = _ref(f"out/reg/{stage}/")
        assert set(log) == set(ref), (stage, sorted(log), sorted(ref))
        for k, want in ref.items():
            assert abs(float(log[k].detach()) - want) <= 2e-4 * abs(want) + 1e-9, (stage, k, float(log[k].detach()), want)
    sample = {"rgb": torch.zeros(len(ts), 3, H, W), "lmk2d": torch.from_numpy(G["lmk2d"]).float()}
    tr.fill_cam_params_into_sample(sample)
    # the renderer's camera helpers (host-side torch code of HipDiffRenderer) against the reference's world_to_clip / world_to_ndc
    with torch.no_grad():
        clip = tr.render.world_to_clip(verts, sample["extrinsic"], sample["intrinsic"], (H, W))
        ndc = tr.render.world_to_ndc(lmks, sample["extrinsic"], sample["intrinsic"], (H, W), flip_y=True)
    assert float((clip.double() - torch.from_numpy(G["clip_ref"]).double()).abs().max()) < 2e-5
    assert float((ndc.double() - torch.from_numpy(G["ndc_ref"])).abs().max()) < 2e-5
    out = _ref("out/")
    for dis in (False, True):
        for always in (True, False):
            cfg.w.always_enable_jawline_landmarks = always
            e = float(tr.compute_lmk_energy(sample, lmks, dis)[0].detach())
            want = out[f"lmk_{'nojaw' if dis else 'jaw'}_always{int(always)}"]
            assert abs(e - want) <= 2e-4 * abs(want), (dis, always, e, want)


def test_nersemble_config_matches_reference():
    """vhap/config/nersemble.py (the calibrated multi-view configuration of BASELINE config 4): every default the product's
    nersemble_config() shares with NersembleTrackingConfig() has the same value."""
    from vhap_amd.config import nersemble_config
    cfg = nersemble_config()
    seen = 0
    for key, val in zip((str(x) for x in G["ners_keys"]), (str(x) for x in G["ners_vals"])):
        obj, ok = cfg, True
        for part in key.split("."):
            if not hasattr(obj, part):
                ok = False
                break
            obj = getattr(obj, part)
        if not ok:
            assert not key.startswith(("w.", "lr.", "pipeline.", "render.")), key
            continue
        if key == "render.backend":
            continue
        seen += 1
        norm = lambda v: repr(tuple(v)) if isinstance(v, (list, tuple)) else repr(v)
        assert norm(obj) == norm(eval(val, {"__builtins__": {}}, {})), (key, obj, val)
    assert seen > 80


def test_config_defaults_match_reference():
    """Every default the product's config shares with the reference's BaseTrackingConfig() has the same value; the loss weights, the
    learning rates and the stage lists must all be present."""
    from vhap_amd.config import BaseTrackingConfig
    cfg = BaseTrackingConfig()
    missing, seen = [], 0
    for key, val in zip((str(x) for x in G["cfg_keys"]), (str(x) for x in G["cfg_vals"])):
        obj, ok = cfg, True
        for part in key.split("."):
            if not hasattr(obj, part):
                ok = False
                break
            obj = getattr(obj, part)
        if not ok:
            missing.append(key)
            continue
        seen += 1
        norm = lambda v: repr(tuple(v)) if isinstance(v, (list, tuple)) else repr(v)
        if key == "render.backend":                         # the one intended difference: the drop-in switch itself
            assert (obj, eval(val)) == ("hip", "nvdiffrast")
            continue
        assert norm(obj) == norm(eval(val, {"__builtins__": {}}, {})), (key, obj, val)
    must = [k for k in missing if k.startswith(("w.", "lr.", "pipeline.", "render.")) or (k.startswith("model.") and k != "model.flame_params_path")]
    assert not must, must
    assert seen > 80, (seen, missing)


def test_oracle_photometric_energy_matches_reference_around_the_raster_ops(flame_model):
    """The reference's compute_photometric_energy -> render_rgba chain, run in the build container with only the four nvdiffrast ops
    replaced by the oracle's restatements, against the oracle's own render_rgba + photometric_energy on the same visibility: pins the
    camera chain, normals, region detach, SH shading, compositing and flips, the colour disturbance (the reference's random draws are
    replayed through the oracle's injected-randomness interface), boundary detach and the loss normalisation -- values and gradients.
    The reference runs this in fp32, the oracle here in fp64: 2e-5 on values, 2e-3 of the max-norm on gradients."""
    from vhap_amd.config import BaseTrackingConfig
    model, topo = flame_model
    dt = torch.float64
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    faces, faces_uv = tm["faces"].long(), tm["faces_uv"].long()
    pin = {k[len("photo_in/"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("photo_in/")}
    pout = {k[len("photo_out/"):]: G[k] for k in G.files if k.startswith("photo_out/")}
    B, _, H, W = pin["rgb"].shape
    tid = torch.from_numpy(pout["tid"].astype(np.int64))
    uv = tm["verts_uvs"].clone()
    uv[:, 1] = 1 - uv[:, 1]
    opp = torch.from_numpy(topo.opp.astype(np.int64))
    cfg = BaseTrackingConfig()
    fid2cid = torch.from_numpy(topo.fid2cid.astype(np.int64))
    for tag, stage in (("eval", None), ("rgb_global_tracking", "rgb_global_tracking")):
        verts = pin["verts"].to(dt).requires_grad_()
        tex_extra = pin["tex_extra"].to(dt).requires_grad_()
        lights = pin["lights"].to(dt).requires_grad_()
        K, RT = pin["K"].to(dt), pin["RT"].to(dt)
        clip = R.camera_to_clip(R.world_to_camera(verts, RT), K, (H, W))
        rast, db = R.rast_from_ids(clip, faces, tid, (H, W))
        tex = (pin["tex_painted"].to(dt)[None] + tex_extra[None]).permute(0, 2, 3, 1)
        tmask = amask = disturb = None
        if stage is not None:
            st = cfg.pipeline[stage]
            tmask = torch.zeros(faces.shape[0] + 1, dtype=torch.bool)
            tmask[torch.from_numpy(topo.get_fid_by_region(list(st.align_texture_except))) + 1] = True
            amask = torch.from_numpy(topo.get_vid_by_region(list(st.align_boundary_except)))
            disturb = dict(w_fg=torch.from_numpy(pout["disturb/w_fg"]), w_bg=torch.from_numpy(pout["disturb/w_bg"]), fid2cid=fid2cid,
                           idx=[torch.from_numpy(r.astype(np.int64)) for r in pout["disturb/idx"]])
        gt = pin["rgb"].to(dt)
        out = R.render_rgba(rast, db, verts, clip, faces, uv, faces_uv, tex, lights[None], gt.permute(0, 2, 3, 1), opp, R.sh_const(dt),
                            tex_detach_mask=tmask, aa_detach_vid=amask, disturb=disturb)
        E = R.photometric_energy(gt, out["rgba"])
        E.backward()
        want = float(pout[f"{tag}/E"])
        assert abs(float(E.detach()) - want) <= 2e-5 * abs(want), (tag, float(E.detach()), want)
        assert float((out["rgba"].permute(0, 3, 1, 2).detach() - torch.from_numpy(pout[f"{tag}/rgba"])).abs().max()) < 2e-5
        ddn = out["diffuse_detach_normal"].permute(0, 3, 1, 2).detach()
        assert float((ddn - torch.from_numpy(pout[f"{tag}/diffuse_detach_normal"])).abs().max()) < 2e-5
        for name, g in (("d_verts", verts.grad), ("d_tex_extra", tex_extra.grad), ("d_lights", lights.grad)):
            ref = torch.from_numpy(pout[f"{tag}/{name}"]).to(dt)
            err = float((g - ref).abs().max() / ref.abs().max())
            assert err < 2e-3, (tag, name, err)


def test_flame_forward_and_region_tables_match_reference(flame_model):
    """FlameHead.forward (flame.py:571-646, with static AND dynamic offsets) and the FlameMask derivations (flame.py:940-1033: vertex regions
    -> face regions, fid2cid with later clusters overwriting earlier ones, region look-ups of the stage configs), computed by the
    reference's own methods on the synthetic model / regions, against the oracle (fp64) and the product (FlameHead fp32, Topology)."""
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.flame import FlameHead
    model, topo = flame_model
    dt = torch.float64
    P, ts, _ = _state(dt)
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    B = len(ts)
    dyn = torch.from_numpy(G["flame/dynamic_offset"]).to(dt)
    pick = torch.from_numpy(G["flame/pick"])
    args = (P["shape"][None].expand(B, -1), P["expr"][ts], P["rotation"][ts], P["neck_pose"][ts], P["jaw_pose"][ts], P["eyes_pose"][ts],
            P["translation"][ts])
    verts, cano, lmks = R.flame_forward(tm, *args, static_offset=P["static_offset"], dynamic_offset=dyn)
    for got, key in ((verts[:, pick], "verts"), (cano[:, pick], "verts_cano"), (lmks, "lmks")):
        assert float((got - torch.from_numpy(G[f"flame/{key}"])).abs().max()) < 1e-12, key
    head = FlameHead(model, topo)                                # the product's module, fp32, CPU
    v32, c32, l32 = head(*[a.float() for a in args], return_verts_cano=True, static_offset=P["static_offset"].float(), dynamic_offset=dyn.float())
    for got, key in ((v32[:, pick], "verts"), (c32[:, pick], "verts_cano"), (l32, "lmks")):
        assert float((got.double() - torch.from_numpy(G[f"flame/{key}"])).abs().max()) < 5e-6, key
    # region tables
    names = [str(x) for x in G["mask/f_names"]]
    assert sorted(topo.f_regions) == sorted(names)
    for k in names:
        assert np.array_equal(np.sort(topo.f_regions[k]), np.sort(G[f"mask/f/{k}"])), k
    # the reference's table has F + 1 entries indexed by the UNSHIFTED face id (the last one is never addressed); NVDiffRenderer.__init__ pads a
    # leading 0 for "no face" (render_nvdiffrast.py:78) -- the product stores that padded form
    F_ = topo.num_faces
    assert G["mask/fid2cid"].shape[0] == F_ + 1 and np.array_equal(topo.fid2cid[1:], G["mask/fid2cid"][:F_]) and topo.fid2cid[0] == 0
    cfg = BaseTrackingConfig()
    for stage in ("rgb_init_texture", "rgb_init_all", "rgb_global_tracking"):
        st = cfg.pipeline[stage]
        assert np.array_equal(topo.get_fid_by_region(list(st.align_texture_except)), G[f"mask/fid/{stage}"]), stage
        assert np.array_equal(topo.get_vid_by_region(list(st.align_boundary_except)), G[f"mask/vid/{stage}"]), stage


def test_tracker_host_logic_matches_reference(flame_model):
    """GlobalTracker.get_train_parameters (:1465-1513), configure_optimizer (:159-211: parameter groups and learning rates) and
    initialize_next_timtestep (:1515-1529) of the reference, run on a tracker object without __init__, against the product's."""
    import json
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.synthetic import make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    ref = json.loads(str(G["host/json"]))
    cfg = BaseTrackingConfig()
    cfg.device = "cpu"
    cfg.model.tex_resolution = 16
    N = 9
    tr = GlobalTracker(cfg, model, topo, make_texture(0, 16), {"rgb": torch.zeros(N, 3, 8, 8), "lmk2d": torch.zeros(N, 70, 3)})
    names = ("focal_length", "shape", "tex_extra", "static_offset", "lights", "translation", "rotation", "eyes_pose", "neck_pose", "jaw_pose",
             "expr")
    name_of = lambda t: next(k for k in names if getattr(tr, k) is t)
    assert list(cfg.pipeline.__dict__) == list(ref)                         # same stages, same order
    for stage, want in ref.items():
        params = tr.get_train_parameters(stage)
        assert sorted(k for k, v in tr.opt_dict.items() if v) == want["opt_dict"], stage
        assert {k: [name_of(t) for t in v] for k, v in params.items() if len(v)} == want["params"], stage
        opt = tr.configure_optimizer(params, lr_scale=0.5)
        got = [[sorted(name_of(t) for t in g["params"]), g["lr"]] for g in opt.param_groups]
        assert len(got) == len(want["groups"]), stage
        for (gn, glr), (wn, wlr) in zip(got, want["groups"]):
            assert gn == wn and abs(glr - wlr) <= 1e-12, (stage, gn, glr, wn, wlr)
    keys = ("translation", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "expr")
    with torch.no_grad():
        for k in keys:
            p = getattr(tr, k)
            p.zero_()
            p[:, :4] = torch.from_numpy(G[f"host/before/{k}"]).float()[:, :min(4, p.shape[1])] if p.shape[1] >= 4 else \
                torch.from_numpy(G[f"host/before/{k}"]).float()[:, :p.shape[1]]
    tr.initialize_next_timtestep(np.array([2, 3, 4]))
    for k in keys:
        p = getattr(tr, k)
        w = min(4, p.shape[1])
        assert torch.equal(p[:, :w], torch.from_numpy(G[f"host/after/{k}"]).float()[:, :w]), k


def test_oracle_fit_groups_match_reference_golden(flame_model):
    """oracle/fit_ref.py (the oracle-side optimiser set-up used by the K-step parity test): the tensors a stage trains and the Adam groups /
    learning rates, against what the reference's own get_train_parameters / configure_optimizer produced (host/json of the golden)."""
    import json
    from oracle import fit_ref
    from vhap_amd.config import BaseTrackingConfig
    ref = json.loads(str(G["host/json"]))
    cfg = BaseTrackingConfig()
    names = ("focal_length", "shape", "tex_extra", "static_offset", "lights", "translation", "rotation", "eyes_pose", "neck_pose", "jaw_pose",
             "expr")
    P = {k: torch.zeros(3, dtype=torch.float64, requires_grad=True) for k in names}
    name_of = lambda t: next(k for k in names if P[k] is t)
    for stage, want in ref.items():
        params = fit_ref.train_parameters(P, cfg, stage)
        assert {k: [name_of(t) for t in v] for k, v in params.items()} == want["params"], stage
        groups = fit_ref.optimizer_groups(params, cfg, lr_scale=0.5)
        got = [[sorted(name_of(t) for t in g["params"]), g["lr"]] for g in groups]
        assert len(got) == len(want["groups"]), stage
        for (gn, glr), (wn, wlr) in zip(got, want["groups"]):
            assert gn == wn and abs(glr - wlr) <= 1e-12, (stage, gn, glr, wn, wlr)
        opt = fit_ref.configure_optimizer(P, cfg, stage, lr_scale=0.5)
        assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in groups]


@pytest.mark.parametrize("stage", [None, "lmk_init_all", "rgb_init_offset", "rgb_global_tracking"])
def test_oracle_total_energy_and_gradients_match_reference_compute_energy(flame_model, stage):
    """The reference's GlobalTracker.compute_energy (:692-750: forward_flame, camera, landmark + photometric + regularisation terms, their
    assembly per stage kind) and its backward -- run in the build container on a tracker object without __init__, fp32, with only the four
    nvdiffrast ops replaced by the oracle's and the colour-disturbance draws replayed -- against the oracle's total_energy (fp64) on the
    same visibility: every term of the log to 5e-6 relative, the gradient w.r.t. EVERY parameter to 2e-4 of its max-norm (measured spread of
    the fp32 reference against the fp64 oracle: 1.4e-7 and 1.3e-5)."""
    from vhap_amd.config import BaseTrackingConfig
    model, topo = flame_model
    dt = torch.float64
    tag = "eval" if stage is None else stage
    P, ts, (H, W) = _state(torch.float32)
    P = {k: v.to(dt).requires_grad_() for k, v in P.items()}
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].float().to(dt)                                       # (the reference run used the fp32 model)
    full = {k[len(f"full/{tag}/"):]: G[k] for k in G.files if k.startswith(f"full/{tag}/")}
    sample = {"rgb": torch.from_numpy(G["full/rgb"]).to(dt), "lmk2d": torch.from_numpy(G["lmk2d"]).float().to(dt), "timestep_index": ts}
    cfg = BaseTrackingConfig()
    disturb = tid = None
    if "tid" in full:
        tid = torch.from_numpy(full["tid"].astype(np.int64))
        assert 0.05 < float(full["coverage"]) < 0.9
    if "idx" in full:
        disturb = dict(w_fg=torch.from_numpy(full["w_fg"].astype(np.int32)), w_bg=torch.from_numpy(full["w_bg"].astype(np.int32)),
                       fid2cid=torch.from_numpy(topo.fid2cid.astype(np.int64)), idx=[torch.from_numpy(r.astype(np.int64)) for r in full["idx"]])
    E, log, _ = energy_ref.total_energy(P, tm, topo, cfg, sample, stage, torch.from_numpy(G["tex_painted"]).float().to(dt),
                                        torch.from_numpy(G["uvmask"]).float().to(dt), (H, W), dt, disturb=disturb, tid=tid)
    E.backward()
    ref_log = {k[len("log/"):]: float(v) for k, v in full.items() if k.startswith("log/")}
    assert set(log) | {"total"} == set(ref_log), (sorted(log), sorted(ref_log))
    for k, want in ref_log.items():
        got = float(E.detach()) if k == "total" else float(log[k].detach())
        assert abs(got - want) <= 5e-6 * abs(want) + 1e-9, (tag, k, got, want)
    for k, p in P.items():
        ref = torch.from_numpy(full[f"grad/{k}"]).to(dt).reshape(p.shape)
        g = torch.zeros_like(p) if p.grad is None else p.grad
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(g.abs().max()) == 0.0, (tag, k)                   # not part of this stage's energy
            continue
        err = float((g - ref).abs().max()) / scale
        assert err < 2e-4, (tag, k, err)


def test_saved_npz_schema_matches_reference_save_result(flame_model, tmp_path):
    """The on-disk contract with the downstream consumers (export_as_nerf_dataset.py / GaussianAvatars): keys, shapes and dtypes of the
    reference's save_result (:1152-1218) output, produced by the reference's own method, against the product's save_result."""
    import json
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.synthetic import make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    schema = json.loads(str(G["schema/json"]))
    N, (T_,) = schema["expr"][0][0], set(schema["tex_extra"][0][1:])
    H, W = (int(x) for x in G["image_size"])
    cfg = BaseTrackingConfig()
    cfg.device = "cpu"
    cfg.model.tex_resolution = T_
    tr = GlobalTracker(cfg, model, topo, make_texture(0, T_), {"rgb": torch.zeros(N, 3, H, W), "lmk2d": torch.zeros(N, 70, 3)})
    tr.save_result(tmp_path / "tracked_flame_params.npz")
    rep = np.load(tmp_path / "tracked_flame_params.npz")
    assert set(rep.files) == set(schema), (sorted(rep.files), sorted(schema))
    for k, (shape, dtype) in schema.items():
        assert list(rep[k].shape) == shape, (k, rep[k].shape, shape)
        if k == "timestep_id":                                   # the reference stores the dataset's frame names; the product has no dataset IO: indices
            continue
        assert str(rep[k].dtype) == dtype, (k, rep[k].dtype, dtype)


def test_stage_scheduler_matches_reference(flame_model):
    """GlobalTracker.optimize (:1343-1389) and optimize_stage (:1391-1416) of the reference with the per-step work stubbed out: the sequence of
    stages, the timesteps of every batch, the seeding of the next timesteps, the evaluation points, the learning-rate scale of the global
    stage, the number of steps per stage and the per-step learning rates of every Adam group (ExponentialLR 0.9 per epoch) -- against the
    product's scheduler driven through the same stubs."""
    import json
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.synthetic import make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    ref = json.loads(str(G["sched/json"]))
    cfg = BaseTrackingConfig()
    cfg.device = "cpu"
    cfg.model.tex_resolution = 16

    def tracker(n):
        return GlobalTracker(cfg, model, topo, make_texture(0, 16), {"rgb": torch.zeros(n, 3, 8, 8), "lmk2d": torch.zeros(n, 70, 3)})

    tr = tracker(20)
    trace = []

    def rec_stage(stage, sample=None, dataloader=None, lr_scale=1.0, **kw):
        if sample is not None:
            trace.append(["stage", stage, [int(t) for t in sample["timestep_index"]], float(lr_scale)])
        else:
            trace.append(["stage", stage, {"shuffle": bool(dataloader.shuffle), "batch_size": dataloader.bs, "batches": len(dataloader)},
                          float(lr_scale)])
            seen = sorted(int(t) for b in dataloader for t in b["timestep_index"])
            assert seen == list(range(20))                                   # one pass = every frame once
    tr.optimize_stage = rec_stage
    tr.initialize_next_timtestep = lambda ts: trace.append(["init_next", [int(t) for t in ts]])
    tr.evaluate = lambda **kw: trace.append(["evaluate", 0])
    tr.optimize()
    assert trace == ref["trace"], (trace, ref["trace"])

    for stage, kw in (("lmk_init_rigid", dict(sample={"timestep_index": np.arange(3)})),
                      ("rgb_global_tracking", dict(dataloader=[{"timestep_index": np.arange(2)}, {"timestep_index": np.arange(2, 3)}],
                                                   lr_scale=0.1, evaluate_every=10))):
        tr = tracker(9)
        calls = []
        def stub_iter(sample, optimizer, stage, calls=calls):
            calls.append([g["lr"] for g in optimizer.param_groups])
            optimizer._opt_called = True                     # (the stub stands for optimizer.step(): what torch's schedulers look for before they warn)
        tr.optimize_iter = stub_iter
        tr.evaluate = lambda calls=calls, **kw: calls.append(["evaluate", None])
        tr.optimize_stage(stage, graphed=False, **kw)
        want = ref["stage_calls"][stage]
        assert len(calls) == len(want), (stage, len(calls), len(want))
        for got, w in zip(calls, want):
            if w[0] == "evaluate":
                assert got[0] == "evaluate"
            else:
                assert len(got) == len(w) and all(abs(a - b) <= 1e-12 * abs(b) for a, b in zip(got, w)), (stage, got, w)


def test_op_shim_accepts_the_reference_call_sites():
    """How the reference's renderer actually calls nvdiffrast (names, positional counts, keyword names -- recorded while its render_rgba /
    rasterize ran in the build container) binds to vhap_amd.ops, so that `sys.modules['nvdiffrast.torch'] = vhap_amd.ops` serves
    render_nvdiffrast.py unmodified."""
    import inspect
    import json
    from vhap_amd import ops
    calls = json.loads(str(G["dr_calls/json"]))
    assert {c[0] for c in calls} == {"RasterizeCudaContext", "rasterize", "interpolate", "texture", "antialias"}
    for name, npos, kws in calls:
        fn = getattr(ops, name)
        inspect.signature(fn).bind(*([None] * npos), **{k: None for k in kws})           # raises TypeError if the call would not bind


def test_renderer_surface_accepts_the_reference_tracker_calls():
    """Boundary b1: the positional / keyword shape of every call the reference's FlameTracker makes into its renderer (recorded while its
    compute_energy ran in the build container) binds to HipDiffRenderer's methods."""
    import inspect
    import json
    from vhap_amd.render_hip import HipDiffRenderer
    calls = json.loads(str(G["rend_calls/json"]))
    assert {c[0] for c in calls} >= {"rasterize", "render_rgba", "world_to_ndc", "clear_cache"}
    for name, npos, kws in calls:
        fn = getattr(HipDiffRenderer, name)
        inspect.signature(fn).bind(*([None] * (npos + 1)), **{k: None for k in kws})     # (+1: self)


def test_blur_iter_vertex_weights_match_reference(flame_model):
    """`w.blur_iter > 0` (base.py:181): the reference's own scale_vertex_weights_by_region (tracker.py:607-614; golden from
    tools/make_golden_blur.py) against the oracle's vertex_weights and the product's host-side table -- which is all the native kernels
    ever see of it -- and the two relaxed offset regularisers computed with them.  (The reference's blur only accepts a batch of one.)"""
    from oracle import energy_ref
    from vhap_amd.config import BaseTrackingConfig
    from vhap_amd.synthetic import make_texture
    from vhap_amd.tracker import GlobalTracker
    model, topo = flame_model
    GB = np.load(os.path.join(os.path.dirname(__file__), "golden", "blur_golden.npz"))
    V = GB["v_cano"].shape[1]
    cfg = BaseTrackingConfig()
    cfg.device = "cpu"
    cfg.model.tex_resolution = 32
    tr = GlobalTracker(cfg, model, topo, make_texture(0, 32), {"rgb": torch.zeros(3, 3, 16, 16), "lmk2d": torch.zeros(3, 70, 3)})
    L = energy_ref._laplacian(V, topo)
    w = cfg.w
    off = torch.from_numpy(GB["static_offset"])
    v_cano = torch.from_numpy(GB["v_cano"])
    for it in (1, 3):
        w.blur_iter = it
        for tag, coef, region in (("lap", w.reg_offset_lap_relax_coef, w.reg_offset_lap_relax_for), ("off", w.reg_offset_relax_coef, w.reg_offset_relax_for)):
            want = torch.from_numpy(GB[f"w_{tag}_{it}"])
            got_o = energy_ref.vertex_weights(V, topo, coef, region, it, L, torch.float64)
            got_p = tr._vertex_weights(tag, coef, region)
            assert float((got_o - want).abs().max()) < 1e-12, (tag, it)
            assert float((got_p.double() - want).abs().max()) < 1e-6, (tag, it)
            assert float((want - 1).abs().max()) > 0.05 and float(want.min()) < 1 - 1e-3          # blurred, not the 0/1 table
        wl = energy_ref.vertex_weights(V, topo, w.reg_offset_lap_relax_coef, w.reg_offset_lap_relax_for, it, L, torch.float64)
        wo = energy_ref.vertex_weights(V, topo, w.reg_offset_relax_coef, w.reg_offset_relax_for, it, L, torch.float64)
        v0 = v_cano - off
        e_lap = w.reg_offset_lap * R.laplacian_energy(L.double(), v0, v0 + off, wl)
        e_off = w.reg_offset * (off.abs() * wo).mean()
        assert abs(float(e_lap) - float(GB[f"reg_offset_lap_{it}"])) <= 1e-9 * abs(float(GB[f"reg_offset_lap_{it}"]))
        assert abs(float(e_off) - float(GB[f"reg_offset_{it}"])) <= 1e-9 * abs(float(GB[f"reg_offset_{it}"]))


def test_dynamic_offset_regularisers_match_reference(flame_model):
    """`use_dynamic_offset` (base.py:69): the oracle's regularisers with offset = static_offset + dynamic_offset[timesteps] and the temporal
    term reg_offset_dynamic (tracker.py:552-600), values and gradients, against the reference's own compute_regularization_energy
    (tools/make_golden_dynoffset.py -> tests/golden/dynoffset_golden.npz) -- for the two cases the reference can run at all: both
    offsets on a one-frame batch (its in-place `offset += dynamic_offset[timesteps]` raises for B > 1) and dynamic-only on three frames."""
    from vhap_amd.config import BaseTrackingConfig
    from oracle import energy_ref, torch_ref as R
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dynoffset_golden.npz"))
    assert int(G["both_B3_raises"]) == 1
    model, topo = flame_model
    dt = torch.float64
    tm = {k: torch.from_numpy(np.asarray(v)) for k, v in model.items()}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "lmk_bary_coords", "verts_uvs"):
        tm[k] = tm[k].to(dt)
    N, V = 4, tm["v_template"].shape[0]
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g, dtype=dt) * sc
    P0 = dict(shape=rnd(300, sc=0.3), expr=rnd(N, 100, sc=0.3), rotation=rnd(N, 3, sc=0.1), neck_pose=rnd(N, 3, sc=0.05),
              jaw_pose=rnd(N, 3, sc=0.1), eyes_pose=rnd(N, 6, sc=0.1), translation=rnd(N, 3, sc=0.02),
              static_offset=rnd(1, V, 3, sc=1e-3), dynamic_offset=rnd(N, V, 3, sc=5e-4))
    for k, v in P0.items():
        assert abs(float(v.sum()) - float(G[f"in_sum/{k}"])) <= 1e-9 * max(1.0, abs(float(G[f"in_sum/{k}"]))), k
    cfg = BaseTrackingConfig()
    cfg.model.use_dynamic_offset = True
    pick = G["pick"]
    for case, with_static in (("both_B1", True), ("dynamic_only_B3", False)):
        ts = G[f"{case}/timesteps"]
        B = len(ts)
        for stage in ("rgb_sequential_tracking", "rgb_global_tracking"):
            P = {k: v.clone().requires_grad_() for k, v in P0.items()}
            if not with_static:
                P["static_offset"] = None
            verts, v_cano, lmks = R.flame_forward(tm, P["shape"][None].expand(B, -1), P["expr"][ts], P["rotation"][ts], P["neck_pose"][ts],
                                                  P["jaw_pose"][ts], P["eyes_pose"][ts], P["translation"][ts],
                                                  static_offset=P["static_offset"], dynamic_offset=P["dynamic_offset"][ts])
            opt = set(cfg.pipeline[stage].optimizable_params) - {"texture", "lights"}
            log = energy_ref.regularization_energy(P, ts, cfg.w, stage, opt, None, None, v_cano, None, topo, dt)
            keys = sorted(k.split("/")[-1] for k in G.files if k.startswith(f"{case}/{stage}/log/"))
            assert sorted(log) == keys, (sorted(log), keys)
            for k in keys:
                a, b = float(log[k]), float(G[f"{case}/{stage}/log/{k}"])
                assert abs(a - b) <= 1e-10 * max(abs(b), 1e-6), (case, stage, k, a, b)
            torch.stack(list(log.values())).sum().backward()
            for k in ("static_offset", "dynamic_offset", "expr", "shape"):
                key = f"{case}/{stage}/grad/{k}"
                if key not in G.files:
                    assert P.get(k) is None or P[k].grad is None or float(P[k].grad.abs().max()) == 0, key
                    continue
                a = P[k].grad.numpy()
                a = a[:, pick] if k.endswith("offset") else a
                b = G[key]
                assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-12), (key, np.abs(a - b).max())
