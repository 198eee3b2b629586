"""Photometric FLAME tracker -- MI355X-side mirror of vhap/model/tracker.py.

Keeps the reference's entry points, attribute names and dictionaries for the hot path:
FlameTracker.{forward_flame :213, get_albedo :247, rasterize_flame :260, render_rgba :305,
compute_lmk_energy :347, compute_photometric_energy :391, compute_regularization_energy :480,
compute_energy :692, configure_optimizer :159} and GlobalTracker.{init_params :1279,
optimize_stage :1391, optimize_iter :1418, get_train_parameters :1465, initialize_next_timtestep
:1515, save_result :1152}.  What changes is underneath: the renderer is HipDiffRenderer, region
lookups are cached per stage, the dense [V,V] Laplacian bmm (tracker.py:682-690, 1.7 GB at B=16)
is a sparse apply, the B-fold texture copy is gone, and frame batches can be sharded across GPUs
with one all-reduce per Adam step (vhap_amd.dist).  Logging / TensorBoard / landmark detection /
dataset IO of the reference are out of scope.
"""
import ctypes
import os
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F

from .config import BaseTrackingConfig, PhotometricStageConfig
from . import _lib
from . import fused as FU
from . import native as NV
from .flame import FlameHead, FlameTexPainted, FlameTexPCA, FlameUvMask
from .lbs import batch_rodrigues
from .render_hip import HipDiffRenderer


def normalize_image_points(u, v, resolution):
    """util/mesh.py:41-51: pixel coordinates -> [-1, 1]."""
    return 2 * (u - resolution[1] / 2.0) / resolution[1], 2 * (v - resolution[0] / 2.0) / resolution[0]


class FlameTracker:
    def __init__(self, cfg: BaseTrackingConfig, flame_model, topo, base_texture):
        self.cfg = cfg
        self.device = cfg.device
        self.flame = FlameHead(flame_model, topo, cfg.model.n_shape, cfg.model.n_expr).to(self.device)
        # tracker.py:57-60: the painted base texture, or (tex_painted = False) the FLAME PCA texture space -- `base_texture` is then the
        # texture space: a path to FLAME_texture.npz or a mapping with `mean` / `tex_dir`
        self.flame_tex_pca = None
        if cfg.model.tex_painted:
            self.flame_tex_painted = FlameTexPainted(base_texture).to(self.device)
        else:
            self.flame_tex_pca = FlameTexPCA(cfg.model.n_tex, tex_size=cfg.model.tex_resolution, tex_space=base_texture).to(self.device)
            self.flame_tex_painted = lambda: self.flame_tex_pca(self.tex_pca[None, :]).detach()     # (read-only views of the current base texture)
        self.flame_uvmask = FlameUvMask(topo).to(self.device)
        if cfg.render.backend != "hip":
            raise NotImplementedError(f"Unknown renderer backend: {cfg.render.backend}")
        self.render = HipDiffRenderer(
            use_opengl=cfg.render.use_opengl, lighting_type=cfg.render.lighting_type,
            lighting_space=cfg.render.lighting_space, disturb_rate_fg=cfg.render.disturb_rate_fg,
            disturb_rate_bg=cfg.render.disturb_rate_bg, fid2cid=self.flame.mask.fid2cid,
        ).to(self.device)
        self.render._ncl = int(self.flame.mask.fid2cid.max()) + 1
        uv = self.flame.verts_uvs.clone()
        uv[:, 1] = 1 - uv[:, 1]                                  # tracker.py:315-316 (constant: hoisted)
        self._verts_uv_flipped = uv.contiguous()
        self._faces_i32 = self.flame.faces.int().contiguous()
        self._region_cache = {}
        self.dist = None                                          # set by vhap_amd.dist.attach()
        self.fused_gbuffer = True                                 # one launch for rasterize + both interpolates
        self.fused = True                                         # fused shading / loss kernels (False: reference-shaped torch ops)
        self.native = True                                        # per-frame stage / landmarks / regularisers as fused HIP kernels too
        self._nm = None
        self.opt_dict = defaultdict(bool)
        self._split = None

    # ---- helpers ----
    def clear_cache(self):
        self.render.clear_cache()

    def _regions(self, stage):
        """Stage -> (align_texture_except_fid, align_boundary_except_vid); the reference rebuilds these
        with cat+unique on every step (tracker.py:417-422)."""
        if stage not in self._region_cache:
            st = self.cfg.pipeline[stage]
            self._region_cache[stage] = (self.flame.mask.get_fid_by_region(st.align_texture_except),
                                         self.flame.mask.get_vid_by_region(st.align_boundary_except))
        return self._region_cache[stage]

    def fill_cam_params_into_sample(self, sample):
        """tracker.py:141-157."""
        if self.calibrated:
            assert "intrinsic" in sample and "extrinsic" in sample
        else:
            b, _, h, w = sample["rgb"].shape
            f = self.focal_length * max(h, w)
            cx = torch.full_like(f, 0.5 * w)
            cy = torch.full_like(f, 0.5 * h)
            sample["intrinsic"] = torch.stack([f, f, cx, cy], dim=1)
            sample["extrinsic"] = self.RT[None, ...].expand(b, -1, -1)

    def configure_optimizer(self, params, lr_scale=1.0):
        """tracker.py:159-211: Adam, per-group learning rates."""
        params = dict(params)
        lr = self.cfg.lr
        group_lr = {"translation": lr.translation, "expr": lr.expr, "lights": lr.light}
        if not self.calibrated:
            group_lr["cam"] = lr.camera
        if self.cfg.model.use_static_offset:
            group_lr["static_offset"] = lr.static_offset
        if self.cfg.model.use_dynamic_offset:
            group_lr["dynamic_offset"] = lr.dynamic_offset
        groups = []
        for key, g_lr in group_lr.items():
            sel = params.pop(key, [])
            if len(sel) > 0:
                groups.append({"params": sel, "lr": g_lr * lr_scale})
        rest = [p for v in params.values() for p in v]
        groups.append({"params": rest})
        on_gpu = str(self.device).startswith("cuda")
        if on_gpu and self.fused and self.native:
            return NV.HipAdam(groups, lr=lr.base * lr_scale)        # one launch for all parameter tensors
        return torch.optim.Adam(groups, lr=lr.base * lr_scale, capturable=on_gpu)   # device step counter: graph-capturable

    # ---- model ----
    def forward_flame(self, timesteps):
        dyn = self.dynamic_offset[timesteps] if self.cfg.model.use_dynamic_offset else None
        verts, verts_cano, lmks = self.flame(
            self.shape[None, ...].expand(len(timesteps), -1), self.expr[timesteps], self.rotation[timesteps],
            self.neck_pose[timesteps], self.jaw_pose[timesteps], self.eyes_pose[timesteps],
            self.translation[timesteps], return_verts_cano=True, static_offset=self.static_offset,
            dynamic_offset=dyn)
        albedos = self.get_albedo().expand(len(timesteps), -1, -1, -1)
        return verts, verts_cano, lmks, albedos

    def get_base_texture(self):
        if self.cfg.model.tex_extra and not self.cfg.model.residual_tex:
            return self.tex_extra[None, ...]
        if self.flame_tex_pca is not None:                        # tracker.py:243-244
            return self.flame_tex_pca(self.tex_pca[None, :])
        return self.flame_tex_painted()

    def get_albedo(self):
        base = self.get_base_texture()
        if self.cfg.model.tex_extra and self.cfg.model.residual_tex:
            res = self.tex_extra[None, :]
            if base.shape[-2:] != res.shape[-2:]:
                base = F.interpolate(base, res.shape[-2:], mode="bilinear")
            return base + res
        return base

    def rasterize_flame(self, sample, verts, faces, camera_index=None, train_mode=False):
        K = sample["intrinsic"].to(self.device)
        RT = sample["extrinsic"].to(self.device)
        if camera_index is not None:
            K, RT = K[[camera_index]], RT[[camera_index]]
        return self.render.rasterize(verts, faces, RT, K, tuple(self.image_size), False, train_mode, defer=self.fused_gbuffer)

    @torch.no_grad()
    def get_background_color(self, gt_rgb, gt_alpha, stage):
        background = self.cfg.render.background_eval if stage is None else self.cfg.render.background_train
        if background == "target":
            return gt_rgb.permute(0, 2, 3, 1)
        if background == "white":
            return [1, 1, 1]
        if background == "black":
            return [0, 0, 0]
        raise NotImplementedError(f"Unknown background mode: {background}")

    def render_rgba(self, rast_dict, verts, faces, albedos, lights, background_color=[1, 1, 1],
                    align_texture_except_fid=None, align_boundary_except_vid=None, enable_disturbance=False,
                    disturbance=None):
        out = self.render.render_rgba(rast_dict, verts, faces, self._verts_uv_flipped, self.flame.textures_idx, albedos,
                                      lights, background_color, align_texture_except_fid, align_boundary_except_vid,
                                      enable_disturbance, disturbance=disturbance)
        return {k: v.permute(0, 3, 1, 2) for k, v in out.items()}

    # ---- energies ----
    def compute_lmk_energy(self, sample, pred_lmks, disable_jawline_landmarks=False):
        """tracker.py:347-389."""
        img_size = sample["rgb"].shape[-2:]
        lmk2d = sample["lmk2d"].to(pred_lmks)
        conf = lmk2d[:, :, 2]
        gu, gv = normalize_image_points(lmk2d[:, :, 0], lmk2d[:, :, 1], img_size)
        gt = torch.stack([gu, gv], dim=-1)
        pred = self.render.world_to_ndc(pred_lmks, sample["extrinsic"].to(self.device), sample["intrinsic"].to(self.device),
                                        img_size, flip_y=True)[:, :, :2]
        if not self.cfg.w.always_enable_jawline_landmarks and disable_jawline_landmarks:
            diff, conf = gt[:, 17:68] - pred[:, 17:68], conf[:, 17:68]
        else:
            diff = gt[:, :68] - pred[:, :68]
            boost = torch.ones(68, dtype=conf.dtype, device=conf.device)
            boost[27:36] = 10                                    # nose landmarks are robust
            conf = conf[:, :68] * boost
        loss = diff.abs().sum(dim=2) * conf
        return loss.mean(), {"gt_lmk2d": gt, "pred_lmk2d": pred}

    def compute_photometric_energy(self, sample, verts, faces, albedos, rast_dict, step_i=None, stage=None,
                                   disturbance=None):
        """tracker.py:391-478: sum|gt - pred| / (3 * #{alpha > 0})."""
        gt_rgb = sample["rgb"].to(verts)
        lights = self.lights[None] if self.lights is not None else None
        bg_color = self.get_background_color(gt_rgb, None, stage)
        fid, vid = self._regions(stage) if stage is not None else (None, None)
        if self.fused and stage is not None and self.render.lighting_type == "SH":
            want_reg = bool(self.opt_dict["lights"]) and self.cfg.w.reg_diffuse is not None
            out = self.render.render_rgba(rast_dict, verts, faces, self._verts_uv_flipped, self.flame.textures_idx, albedos, lights,
                                          bg_color, fid, vid, True, disturbance=disturbance, outputs="loss",
                                          want_reg_diffuse=want_reg)
            abs_sum, n_alpha = FU.photo_sum(out["rgba_rs"], gt_rgb)
            if self._split is not None:                            # graphed step: the normaliser is applied later (GraphedStep)
                self._split["S"], self._split["N"] = abs_sum, n_alpha
                return abs_sum * 0.0, {"gt_rgb": gt_rgb, "rgba_rs": out["rgba_rs"], "reg_diffuse_value": out["reg_diffuse"]}
            n_mask = n_alpha * 3
            if self.dist is not None:
                n_mask = self.dist.all_reduce_sum(n_mask) / self.dist.world_size
            return abs_sum / n_mask, {"gt_rgb": gt_rgb, "rgba_rs": out["rgba_rs"], "reg_diffuse_value": out["reg_diffuse"]}
        render_out = self.render_rgba(rast_dict, verts, faces, albedos, lights, bg_color, fid, vid,
                                      enable_disturbance=stage is not None, disturbance=disturbance)
        pred_rgb = render_out["rgba"][:, :3]
        pred_alpha = render_out["rgba"][:, 3:]
        n_mask = (pred_alpha.detach() > 0).sum() * 3
        error_rgb = gt_rgb - pred_rgb
        abs_sum = error_rgb.abs().sum()
        if self.dist is not None:                                  # batch-global normaliser
            n_mask = self.dist.all_reduce_sum(n_mask.to(abs_sum.dtype)) / self.dist.world_size
        render_out.update({"gt_rgb": gt_rgb, "pred_rgb": pred_rgb, "error_rgb": error_rgb, "pred_alpha": pred_alpha})
        return abs_sum / n_mask, render_out

    def compute_regularization_energy(self, result_dict, verts, verts_cano, lmks, albedos, timesteps, stage):
        """tracker.py:480-605."""
        w = self.cfg.w
        log = {}
        tracking = "tracking" in stage
        if self.opt_dict["pose"] and tracking:
            log["smooth_pose"] = self.compute_pose_smooth_energy(timesteps)
        if self.opt_dict["joints"]:
            log["reg_joint"] = self.compute_joint_L2_energy(timesteps)
            if tracking:
                log["smooth_joint"] = self.compute_joint_smooth_energy(timesteps)
        if self.opt_dict["expr"]:
            log["reg_expr"] = w.reg_expr * (self.expr[timesteps] ** 2).mean()
            if tracking:
                log["smooth_expr"] = self.compute_expr_smooth_energy(timesteps)
        if self.opt_dict["shape"]:
            log["reg_shape"] = w.reg_shape * (self.shape ** 2).mean()
        if self.opt_dict["texture"] and self.flame_tex_pca is not None:        # tracker.py:519-521 (std_tex = 1)
            log["reg_tex_pca"] = w.reg_tex_pca * (self.tex_pca ** 2).mean()
        if self.opt_dict["texture"] and self.cfg.model.tex_extra and self.cfg.model.residual_tex:
            if w.reg_tex_tv is not None:
                tex = self.get_albedo()[0]
                tv_y = (tex[..., :-1, :] - tex[..., 1:, :]) ** 2
                tv_x = (tex[..., :, :-1] - tex[..., :, 1:]) ** 2
                tv = tv_y.reshape(tv_y.shape[0], -1) + tv_x.reshape(tv_x.shape[0], -1)
                w_tv = w.reg_tex_tv * self.cfg.data.scale_factor ** 2
                if self.cfg.data.n_downsample_rgb is not None:
                    w_tv /= self.cfg.data.n_downsample_rgb ** 2
                log["reg_tex_tv"] = w_tv * tv.mean()
            if w.reg_tex_res_clusters is not None:
                m = self._uvmask_res()
                log["reg_tex_res_clusters"] = w.reg_tex_res_clusters * (self.tex_extra ** 2 * m).mean()
        if self.opt_dict["lights"] and self.lights is not None:
            if w.reg_light is not None:
                log["reg_light"] = w.reg_light * ((self.lights - self.lights_uniform) ** 2).mean()
            if w.reg_diffuse is not None:
                if "reg_diffuse_value" in result_dict:            # computed inside the fused shading kernel
                    log["reg_diffuse"] = w.reg_diffuse * result_dict["reg_diffuse_value"]
                else:
                    diffuse = result_dict["diffuse_detach_normal"]
                    log["reg_diffuse"] = w.reg_diffuse * (F.relu(diffuse.max() - 1) + diffuse.var(dim=1).mean())
        if (self.opt_dict["static_offset"] or self.opt_dict["dynamic_offset"]) and \
                (self.static_offset is not None or self.dynamic_offset is not None):
            offset = 0
            if self.static_offset is not None:
                offset = offset + self.static_offset
            if self.dynamic_offset is not None:
                offset = offset + self.dynamic_offset[timesteps]
            if w.reg_offset_lap is not None:
                v0 = (verts_cano - offset).detach()
                lap = self.compute_laplacian_smoothing_loss(v0, v0 + offset)
                if len(w.reg_offset_lap_relax_for) > 0:
                    lap = lap * self._vertex_weights("lap", w.reg_offset_lap_relax_coef, w.reg_offset_lap_relax_for)
                log["reg_offset_lap"] = w.reg_offset_lap * lap.mean()
            if w.reg_offset is not None:
                ro = offset.abs()
                if len(w.reg_offset_relax_for) > 0:
                    ro = ro * self._vertex_weights("off", w.reg_offset_relax_coef, w.reg_offset_relax_for)
                log["reg_offset"] = w.reg_offset * ro.mean()
            if w.reg_offset_rigid is not None:
                rigid = 0
                for region in w.reg_offset_rigid_for:
                    vids = self.flame.mask.get_vid_by_region([region])
                    rigid = rigid + offset[:, vids, :].var(dim=-2).mean()
                log["reg_offset_rigid"] = w.reg_offset_rigid * rigid
            if w.reg_offset_dynamic is not None and self.dynamic_offset is not None and self.opt_dict["dynamic_offset"]:
                prev = self._prev(timesteps)
                log["reg_offset_dynamic"] = w.reg_offset_dynamic * \
                    ((self.dynamic_offset[timesteps] - self.dynamic_offset[prev]) ** 2).mean()
        return log

    def _uvmask_res(self):
        if not hasattr(self, "_uvmask_res_cache"):
            m = self.flame_uvmask.get_uvmask_by_region(self.cfg.w.reg_tex_res_for)[None, :, :].float()
            T = self.tex_extra.shape[-1]
            if m.shape[-1] != T:
                m = F.interpolate(m[None], (T, T), mode="nearest")[0]
            self._uvmask_res_cache = m
        return self._uvmask_res_cache

    def _vertex_weights(self, key, scale_factor, region):
        """scale_vertex_weights_by_region (tracker.py:607-614): per-vertex weights [1,V,1] -- ones, the region scaled, then `blur_iter` rounds
        of w <- M w / 2 with M = L - 2 diag(L) (flame.py:199-201) applied through the CSR Laplacian (M w = L w - 2 d * w).  A static table:
        built once per (stage table, blur_iter) on the host side and cached; the native kernels only ever read it."""
        ck = ("vw", key, int(self.cfg.w.blur_iter))
        if ck not in self._region_cache:
            wv = torch.ones(1, self.flame.v_template.shape[0], 1, device=self.device)
            wv[:, self.flame.mask.get_vid_by_region(list(region))] *= scale_factor
            if self.cfg.w.blur_iter:
                fl = self.flame
                diag = torch.zeros(wv.shape[1], device=self.device, dtype=wv.dtype)
                on_diag = fl.lap_row == fl.lap_col
                diag.index_add_(0, fl.lap_row[on_diag], fl.lap_val[on_diag].to(wv.dtype))
                for _ in range(int(self.cfg.w.blur_iter)):
                    wv = (fl.laplacian_apply(wv) - 2 * diag[None, :, None] * wv) / 2
            self._region_cache[ck] = wv
        return self._region_cache[ck]

    def _prev(self, idx):
        if torch.is_tensor(idx):
            return torch.clamp(idx - 1, 0, self.n_timesteps - 1)
        return np.clip(idx - 1, 0, self.n_timesteps - 1)

    def compute_pose_smooth_energy(self, timesteps):
        p = self._prev(timesteps)
        return ((self.translation[timesteps] - self.translation[p].detach()) ** 2).mean() * self.cfg.w.smooth_trans + \
            ((self.rotation[timesteps] - self.rotation[p].detach()) ** 2).mean() * self.cfg.w.smooth_rot

    def compute_joint_smooth_energy(self, timesteps):
        p = self._prev(timesteps)
        w = self.cfg.w
        return ((self.neck_pose[timesteps] - self.neck_pose[p].detach()) ** 2).mean() * w.smooth_neck + \
            ((self.jaw_pose[timesteps] - self.jaw_pose[p].detach()) ** 2).mean() * w.smooth_jaw + \
            ((self.eyes_pose[timesteps] - self.eyes_pose[p].detach()) ** 2).mean() * w.smooth_eyes

    def compute_expr_smooth_energy(self, timesteps):
        p = self._prev(timesteps)
        return ((self.expr[timesteps] - self.expr[p].detach()) ** 2).mean() * self.cfg.w.smooth_expr

    def compute_joint_L2_energy(self, timesteps):
        """tracker.py:650-680: mean((R(0) - R(pose))^2) per joint + plausibility terms."""
        neck, jaw = self.neck_pose[timesteps], self.jaw_pose[timesteps]
        eye_l, eye_r = self.eyes_pose[timesteps, :3], self.eyes_pose[timesteps, 3:]
        B = neck.shape[0]
        R = batch_rodrigues(torch.cat([neck, jaw, eye_l, eye_r, torch.zeros_like(neck[:1])], dim=0))
        R0 = R[-1:]
        # reference quirk kept for parity (tracker.py:665-666): it stacks B identity rotations in front of
        # the B pose rotations and averages over rows 1..2B-1, i.e. over (2B-1)*9 entries, B-1 of them zero
        d = lambda i: ((R0 - R[i * B:(i + 1) * B]) ** 2).sum() / (9 * (2 * B - 1))
        w = self.cfg.w
        e = d(0) * w.reg_neck
        e = e + (d(1) + F.relu(-jaw[:, 0]).mean() * 10 + (jaw[:, 1:] ** 2).mean() * 3) * w.reg_jaw
        eye_diff = ((eye_l - eye_r) ** 2).mean()
        e = e + (d(2) + eye_diff) * w.reg_eyes + (d(3) + eye_diff) * w.reg_eyes
        return e

    def compute_laplacian_smoothing_loss(self, verts, offset_verts):
        """tracker.py:682-690 with a sparse Laplacian: sum_k (L(v+o) - L v)_k^2 per vertex."""
        basis = self.flame.laplacian_apply(verts).detach()
        off = self.flame.laplacian_apply(offset_verts)
        return ((off - basis) ** 2).sum(dim=-1, keepdim=True)

    def compute_energy(self, sample, step_i=None, stage=None, disturbance=None):
        """tracker.py:692-750."""
        if self._native_ok(stage):
            return self._compute_energy_native(sample, stage, disturbance)
        log_dict = {}
        result_dict = {"gt_rgb": sample["rgb"]}
        timesteps = sample["timestep_index"]
        verts, verts_cano, lmks, albedos = self.forward_flame(timesteps)
        faces = self.flame.faces
        if self.cfg.w.landmark is not None:
            disable_jaw = (not self.cfg.w.always_enable_jawline_landmarks and stage is not None and
                           self.cfg.pipeline[stage]["disable_jawline_landmarks"])
            E_lmk, rd = self.compute_lmk_energy(sample, lmks, disable_jaw)
            log_dict["lmk"] = self.cfg.w.landmark * E_lmk
            result_dict.update(rd)
        if stage is None or isinstance(self.cfg.pipeline[stage], PhotometricStageConfig):
            if self.cfg.w.photo is not None:
                rast_dict = self.rasterize_flame(sample, verts, faces, train_mode=True)
                E_photo, rd = self.compute_photometric_energy(sample, verts, faces, albedos, rast_dict, step_i, stage,
                                                              disturbance=disturbance)
                result_dict.update(rd)
                log_dict["photo"] = self.cfg.w.photo * E_photo
        if stage is not None:
            log_dict.update(self.compute_regularization_energy(result_dict, verts, verts_cano, lmks, albedos, timesteps, stage))
        E_total = torch.stack([v for v in log_dict.values()]).sum()
        log_dict["total"] = E_total
        return E_total, log_dict, verts, faces, lmks, albedos, result_dict

    # ---- native step: the whole energy through fused HIP stages (vhap_amd.native / fused / ops) ----
    def _native_ok(self, stage, dynamic_offset_ok=False, tex_pca_ok=False):
        """`dynamic_offset_ok` / `tex_pca_ok`: the caller handles per-frame vertex offsets / the PCA texture model (vhap_amd/step.py::NativeStep
        does; the autograd formulation over the fused stages, _compute_energy_native, does not -- those configurations then take the host
        formulation)"""
        return (self.fused and self.native and stage is not None and str(self.device).startswith("cuda") and
                (tex_pca_ok or self.flame_tex_pca is None) and
                (dynamic_offset_ok or not self.cfg.model.use_dynamic_offset) and self.render.lighting_type == "SH" and
                self.render.lighting_space == "world" and len(self.flame._parents) == 5 and
                self.cfg.model.tex_extra and self.cfg.model.residual_tex)

    def _native_models(self):
        if self._nm is None:
            fl = self.flame
            if fl._fb is None:
                fl._fb = FU.FlameBasis(fl.shapedirs, fl.posedirs, fl.J_regressor, fl.v_template, fl.lbs_weights)
            w = self.cfg.w
            wl = self._vertex_weights("lap", w.reg_offset_lap_relax_coef, w.reg_offset_lap_relax_for) if len(w.reg_offset_lap_relax_for) else None
            wa = self._vertex_weights("off", w.reg_offset_relax_coef, w.reg_offset_relax_for) if len(w.reg_offset_relax_for) else None
            regions = [fl.mask.get_vid_by_region([r]) for r in w.reg_offset_rigid_for] if w.reg_offset_rigid is not None else []
            self._nm = {
                "frame": NV.FrameModel(fl._fb, fl.J_regressor, fl._parents),
                "lmk": NV.LandmarkModel(fl.faces, fl.full_lmk_faces_idx, fl.full_lmk_bary_coords),
                "off": NV.OffsetRegModel(fl.lap_ptr, fl.lap_col, fl.lap_val, wl, wa, regions, self.device),
                "res_mask": (self._uvmask_res()[0] > 0).to(torch.uint8).contiguous(),
                "weights": {},
            }
            h, wd = self.image_size
            m = float(max(h, wd))
            self._nm["K1"] = torch.tensor([[m, m, 0.0, 0.0]], device=self.device)
            self._nm["K0"] = torch.tensor([[0.0, 0.0, 0.5 * wd, 0.5 * h]], device=self.device)
        return self._nm

    def _w_tv(self):
        w_tv = self.cfg.w.reg_tex_tv
        if w_tv is None:
            return None
        w_tv = w_tv * self.cfg.data.scale_factor ** 2
        if self.cfg.data.n_downsample_rgb is not None:
            w_tv /= self.cfg.data.n_downsample_rgb ** 2
        return w_tv

    def _frame_weights(self, stage):
        nm = self._native_models()
        if stage not in nm["weights"]:
            w, o, tracking = self.cfg.w, self.opt_dict, "tracking" in stage
            kw = {}
            if o["pose"] and tracking:
                kw.update(smooth_trans=w.smooth_trans, smooth_rot=w.smooth_rot)
            if o["joints"]:
                kw.update(reg_neck=w.reg_neck, reg_jaw=w.reg_jaw, reg_eyes=w.reg_eyes)
                if tracking:
                    kw.update(smooth_neck=w.smooth_neck, smooth_jaw=w.smooth_jaw, smooth_eyes=w.smooth_eyes)
            if o["expr"]:
                kw.update(reg_expr=w.reg_expr)
                if tracking:
                    kw.update(smooth_expr=w.smooth_expr)
            if o["shape"]:
                kw.update(reg_shape=w.reg_shape)
            nm["weights"][stage] = (NV.frame_weights(**kw), dict(o))
        wts, o_then = nm["weights"][stage]
        if o_then != dict(self.opt_dict):                          # the stage's optimisable set changed: rebuild
            del nm["weights"][stage]
            return self._frame_weights(stage)
        return wts

    def _compute_energy_native(self, sample, stage, disturbance=None):
        """Same energies as compute_energy (tracker.py:692-750), every stage a fused HIP kernel."""
        nm = self._native_models()
        w, o, cfg = self.cfg.w, self.opt_dict, self.cfg
        tracking = "tracking" in stage
        ts = sample["timestep_index"]
        if not torch.is_tensor(ts):
            ts = torch.as_tensor(np.asarray(ts), device=self.device)
        ts = ts.long()
        B = ts.shape[0]
        coef, A, transl, pterms = NV.frame_prep(nm["frame"], self._frame_weights(stage), ts, self.shape, self.expr, self.rotation,
                                                self.translation, self.neck_pose, self.jaw_pose, self.eyes_pose, self.static_offset)
        verts, verts_cano = FU.flame_skin(self.flame._fb, coef, A, transl, self.static_offset)
        if self.calibrated:
            K = sample["intrinsic"].to(self.device)
            if K.shape[-2:] == (3, 3):
                K = torch.stack([K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]], dim=-1)
            RT = sample["extrinsic"].to(self.device)
        else:
            K = self.focal_length * nm["K1"] + nm["K0"]            # [1,4]: f, f, cx, cy (tracker.py:141-157)
            RT = self.RT[None]
        mvp = NV.camera(K, RT, B, self.image_size)
        faces = self.flame.faces
        log_dict, result_dict = {}, {"gt_rgb": sample["rgb"]}
        parts = []
        lmks = None
        if w.landmark is not None:
            disable_jaw = not w.always_enable_jawline_landmarks and cfg.pipeline[stage]["disable_jawline_landmarks"]
            E_lmk, lmks = NV.landmark_energy(nm["lmk"], verts, mvp, sample["lmk2d"].to(self.device), self.image_size, disable_jaw,
                                             want_lmk3d=not torch.is_grad_enabled())
            log_dict["lmk"] = w.landmark * E_lmk
        tex_on = o["texture"]
        painted = self.flame_tex_painted()[0]
        if painted.shape[-1] != self.tex_extra.shape[-1]:
            painted = F.interpolate(painted[None], self.tex_extra.shape[-2:], mode="bilinear")[0]
        sampler = NV.TexSampler(painted, self.tex_extra, nm["res_mask"], self._w_tv() if tex_on else None,
                                w.reg_tex_res_clusters if tex_on else None)
        photometric = isinstance(cfg.pipeline[stage], PhotometricStageConfig) and w.photo is not None
        if not photometric:
            sampler.prep_only()
        if photometric:
            verts_clip = FU.transform(verts, mvp)
            rast_dict = {"rast_out": None, "rast_out_db": None, "verts": verts, "verts_camera": None, "verts_clip": verts_clip,
                         "image_size": tuple(self.image_size), "require_grad": True}
            gt_rgb = sample["rgb"].to(verts)
            bg_color = self.get_background_color(gt_rgb, None, stage)
            fid, vid = self._regions(stage)
            want_reg = bool(o["lights"]) and w.reg_diffuse is not None
            out = self.render.render_rgba(rast_dict, verts, faces, self._verts_uv_flipped, self.flame.textures_idx, None,
                                          self.lights[None], bg_color, fid, vid, True, disturbance=disturbance, outputs="loss",
                                          want_reg_diffuse=want_reg, tex_sampler=sampler)
            abs_sum, n_alpha = FU.photo_sum(out["rgba_rs"], gt_rgb)
            result_dict.update({"rgba_rs": out["rgba_rs"], "reg_diffuse_value": out["reg_diffuse"]})
            if self._split is not None:                            # graphed step: the normaliser is applied later (GraphedStep)
                self._split["S"], self._split["N"] = abs_sum, n_alpha
                log_dict["photo"] = abs_sum * 0.0
            else:
                n_mask = n_alpha * 3
                if self.dist is not None:
                    n_mask = self.dist.all_reduce_sum(n_mask) / self.dist.world_size
                log_dict["photo"] = w.photo * (abs_sum / n_mask)
        tterms = sampler.terms
        albedos = sampler.albedo_cl.permute(0, 3, 1, 2).expand(B, -1, -1, -1)
        # regularisers, in the reference's order (tracker.py:480-605)
        pt = dict(zip(NV.FRAME_TERMS, pterms.unbind(0)))
        if o["pose"] and tracking:
            log_dict["smooth_pose"] = pt["smooth_pose"]
        if o["joints"]:
            log_dict["reg_joint"] = pt["reg_joint"]
            if tracking:
                log_dict["smooth_joint"] = pt["smooth_joint"]
        if o["expr"]:
            log_dict["reg_expr"] = pt["reg_expr"]
            if tracking:
                log_dict["smooth_expr"] = pt["smooth_expr"]
        if o["shape"]:
            log_dict["reg_shape"] = pt["reg_shape"]
        if tex_on:
            if w.reg_tex_tv is not None:
                log_dict["reg_tex_tv"] = tterms[0]
            if w.reg_tex_res_clusters is not None:
                log_dict["reg_tex_res_clusters"] = tterms[1]
        if o["lights"] and self.lights is not None:
            if w.reg_light is not None:
                log_dict["reg_light"] = w.reg_light * ((self.lights - self.lights_uniform) ** 2).mean()
            if w.reg_diffuse is not None and "reg_diffuse_value" in result_dict:
                log_dict["reg_diffuse"] = w.reg_diffuse * result_dict["reg_diffuse_value"]
        if (o["static_offset"] or o["dynamic_offset"]) and self.static_offset is not None:
            ot = NV.offset_reg(nm["off"], self.static_offset, w.reg_offset_lap, w.reg_offset, w.reg_offset_rigid)
            for i, k in enumerate(("reg_offset_lap", "reg_offset", "reg_offset_rigid")):
                if getattr(w, k) is not None:
                    log_dict[k] = ot[i]
        E_total = torch.stack(list(log_dict.values())).sum()
        log_dict["total"] = E_total
        return E_total, log_dict, verts, faces, lmks, albedos, result_dict


class GlobalTracker(FlameTracker):
    """tracker.py:1221-1529 on an in-memory dataset (list/tensor of frames).  `dataset` must provide
    `rgb` [N,3,H,W] float in [0,1], `lmk2d` [N,68+,3] (pixel u, v, confidence) and, if calibrated,
    `intrinsic` / `extrinsic`.

    Multi-view captures (NeRSemble, nersemble_dataset.py with batchify_all_views: ONE dataset item = all views of a timestep, so the
    reference's n_timesteps = len(dataset) counts timesteps, not images): the dataset additionally carries `timestep_index` [N] -- the
    timestep each of the N frames (images) belongs to, 0..n_timesteps-1 -- and optionally `camera_index` [N].  All views of a timestep
    share that timestep's row of the per-frame parameters; samples, shuffled batches and evaluate() are formed by TIMESTEP."""

    def __init__(self, cfg, flame_model, topo, base_texture, dataset):
        super().__init__(cfg, flame_model, topo, base_texture)
        self.calibrated = cfg.data.calibrated
        self.dataset = dataset
        self.frames = dataset.get("frames")                      # optional ingest.FrameStore: the sequence resident as uint8 (8(f) rank 1)
        if self.frames is not None:
            self.image_size, self.n_frames = self.frames.image_size, len(self.frames)
        else:
            self.image_size = tuple(dataset["rgb"].shape[-2:])
            self.n_frames = dataset["rgb"].shape[0]
        if "timestep_index" in dataset:                               # multi-view: frames (images) grouped by timestep
            ft = np.asarray(dataset["timestep_index"].cpu() if torch.is_tensor(dataset["timestep_index"]) else dataset["timestep_index"]).astype(np.int64)
            if ft.shape != (self.n_frames,) or ft.min() < 0:
                raise ValueError("dataset['timestep_index'] must hold one non-negative timestep per frame")
            self.frame_timestep = ft
            self.n_timesteps = int(ft.max()) + 1
            order = np.argsort(ft, kind="stable")
            starts = np.searchsorted(ft[order], np.arange(self.n_timesteps + 1))
            self._frames_of = [order[starts[t]:starts[t + 1]] for t in range(self.n_timesteps)]
            if any(len(f) == 0 for f in self._frames_of):
                raise ValueError("dataset['timestep_index']: every timestep 0..max needs at least one frame")
        else:
            self.frame_timestep = np.arange(self.n_frames)
            self.n_timesteps = self.n_frames
            self._frames_of = None
        self.global_step = 0
        self._graphed = {}
        self.init_params()

    @classmethod
    def from_reference_config(cls, ref_cfg, **kw):
        """`GlobalTracker(cfg)` of the reference (vhap/model/tracker.py:1221-1261, vhap/track.py:16-21) from the REFERENCE's config object:
        the dataset opened through the reference's own class (`cfg.data._target`, imported from the user's checkout), FLAME from the
        licensed pickles, the frames resident as uint8.  Keyword arguments: vhap_amd.reference_adapter.tracker_from_reference_config."""
        from .reference_adapter import tracker_from_reference_config
        return tracker_from_reference_config(cls, ref_cfg, **kw)

    def init_params(self):
        """tracker.py:1279-1341."""
        dev, m = self.device, self.cfg.model
        N = self.n_timesteps
        z = lambda *s: torch.zeros(*s, device=dev)
        self.shape = z(m.n_shape)
        self.expr = z(N, m.n_expr)
        self.neck_pose, self.jaw_pose, self.eyes_pose = z(N, 3), z(N, 3), z(N, 6)
        self.translation, self.rotation = z(N, 3), z(N, 3)
        self.tex_pca = z(m.n_tex)
        train = [self.shape, self.translation, self.rotation, self.neck_pose, self.jaw_pose, self.eyes_pose, self.expr]
        if not m.tex_painted:                                     # tracker.py:1312-1313
            train.append(self.tex_pca)
        self.tex_extra = None
        if m.tex_extra:
            self.tex_extra = z(3, m.tex_resolution, m.tex_resolution)
            train.append(self.tex_extra)
        if self.cfg.render.lighting_type == "SH":
            self.lights_uniform = z(9, 3)
            self.lights_uniform[0] = float(np.sqrt(4 * np.pi))
            self.lights = self.lights_uniform.clone()
            train.append(self.lights)
        else:
            self.lights = None
        V = self.flame.v_template.shape[0]
        self.static_offset = z(1, V, 3) if m.use_static_offset else None
        if self.static_offset is not None:
            train.append(self.static_offset)
        self.dynamic_offset = z(N, V, 3) if m.use_dynamic_offset else None
        if self.dynamic_offset is not None:
            train.append(self.dynamic_offset)
        if not self.calibrated:
            self.focal_length = torch.tensor([1.5], device=dev)
            self.RT = torch.eye(3, 4, device=dev)
            self.RT[2, 3] = -1
            train.append(self.focal_length)
        for t in train:
            t.requires_grad = True
        self._train_tensors = train

    def get_train_parameters(self, stage):
        """tracker.py:1465-1513."""
        self.opt_dict = defaultdict(bool)
        for p in self.cfg.pipeline[stage].optimizable_params:
            self.opt_dict[p] = True
        o, m = self.opt_dict, self.cfg.model
        params = defaultdict(list)
        if o["cam"] and not self.calibrated:
            params["cam"] = [self.focal_length]
        if o["shape"]:
            params["shape"] = [self.shape]
        if o["texture"] and not m.tex_painted:                    # tracker.py:1485-1487
            params["tex"] = [self.tex_pca]
        if o["texture"] and m.tex_extra:
            params["tex_extra"] = [self.tex_extra]
        if o["static_offset"] and m.use_static_offset:
            params["static_offset"] = [self.static_offset]
        if o["lights"] and self.lights is not None:
            params["lights"] = [self.lights]
        if o["pose"]:
            params["translation"].append(self.translation)
            params["rotation"].append(self.rotation)
        if o["joints"]:
            params["eyes"].append(self.eyes_pose)
            params["neck"].append(self.neck_pose)
            params["jaw"].append(self.jaw_pose)
        if o["expr"]:
            params["expr"].append(self.expr)
        if o["dynamic_offset"] and m.use_dynamic_offset:
            params["dynamic_offset"].append(self.dynamic_offset)
        return params

    def get_sample(self, timesteps, device_index=False):
        """`device_index=True` keeps `timestep_index` as a device LongTensor (needed for graph capture: a numpy
        index would be re-uploaded on every step)."""
        ts = np.asarray(timesteps)
        if self._frames_of is not None:                           # every view of the requested timesteps; each frame carries ITS timestep
            fidx = np.concatenate([self._frames_of[int(t)] for t in ts.reshape(-1)])
            ts = self.frame_timestep[fidx]
        else:
            fidx = ts
        idx = torch.as_tensor(fidx, device=self.dataset["lmk2d"].device)
        ts_dev = torch.as_tensor(ts, device=idx.device)
        if self.frames is not None:                               # gather + composite + convert in one launch (vhap_frame_ingest)
            rgb, alpha = self.frames.batch(idx)
        else:
            rgb, alpha = self.dataset["rgb"][idx], None
        s = {"rgb": rgb, "lmk2d": self.dataset["lmk2d"][idx], "timestep_index": ts_dev if device_index else ts}
        if alpha is not None:
            s["alpha_map"] = alpha
        for k in ("intrinsic", "extrinsic"):
            if k in self.dataset:
                s[k] = self.dataset[k][idx]
        return s

    def optimize(self, batch_size=None, evaluate=True):
        """tracker.py:1343-1389, the stage scheduler: sequential tracking over the frames in order -- the first batch also runs the
        initialisation stages -- each batch seeding the next timesteps, an evaluation pass, then global tracking over shuffled
        batches at a tenth of the learning rates.  The frames come from the in-memory dataset / the resident uint8 store instead of a
        DataLoader; logging and media output are out of scope.  Returns the evaluation report (or None)."""
        self.global_step = 0
        # multi-view: one timestep (all its views) per batch, like the reference's DataLoader(batch_size=None) over a batchify_all_views dataset
        bs = int(batch_size or (1 if self._frames_of is not None else self.cfg.batch_size) or self.n_timesteps)
        on_gpu = str(self.device).startswith("cuda")
        for t0 in range(0, self.n_timesteps, bs):
            ts = np.arange(t0, min(t0 + bs, self.n_timesteps))
            sample = self.get_sample(ts, device_index=on_gpu)
            if self.dist is not None:                                 # frame sharding: this rank's contiguous slice of the batch
                sample = self.dist.shard_sample(sample)
            if t0 == 0:
                self.optimize_stage("lmk_init_rigid", sample)
                self.optimize_stage("lmk_init_all", sample)
                if self.cfg.exp.photometric:
                    self.optimize_stage("rgb_init_texture", sample)
                    self.optimize_stage("rgb_init_all", sample)
                    if self.cfg.model.use_static_offset:
                        self.optimize_stage("rgb_init_offset", sample)
            self.optimize_stage("rgb_sequential_tracking" if self.cfg.exp.photometric else "lmk_sequential_tracking", sample)
            self.initialize_next_timtestep(ts)
        report = self.evaluate(batch_size=bs) if evaluate else None
        loader = ShuffledBatches(self, bs, device_index=on_gpu)
        self.optimize_stage("rgb_global_tracking" if self.cfg.exp.photometric else "lmk_global_tracking", dataloader=loader, lr_scale=0.1,
                            evaluate_every=10 if evaluate else None)
        return report

    def optimize_stage(self, stage, sample=None, dataloader=None, lr_scale=1.0, num_steps=None, graphed=None, evaluate_every=None):
        """tracker.py:1391-1416.  `graphed` (default: on for the fused GPU path): run the stage's identical steps as replays of a
        captured GraphedStep.  The capture is kept per (stage, batch shape, lr scale) and fed new batches by
        copying into its static sample buffers -- sequential tracking re-enters here once per timestep with a same-shaped sample --
        and, like the reference, every call starts from a fresh Adam state.  `evaluate_every`: run evaluate() after every that many
        epochs of a dataloader stage (the reference does so every 10, tracker.py:1414-1415; off by default)."""
        if graphed is None:
            graphed = self.fused and self.native and str(self.device).startswith("cuda")
        if not graphed:
            optimizer = self.configure_optimizer(self.get_train_parameters(stage), lr_scale=lr_scale)
            if sample is not None:
                n = self.cfg.pipeline[stage].num_steps if num_steps is None else num_steps
                for _ in range(n):
                    self.optimize_iter(sample, optimizer, stage)
            else:
                assert dataloader is not None
                sched = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.9)
                for epoch_i in range(self.cfg.pipeline[stage].num_epochs):
                    for s in dataloader:
                        self.optimize_iter(s, optimizer, stage)
                    sched.step()
                    if evaluate_every and (epoch_i + 1) % evaluate_every == 0:
                        self.evaluate()
            return optimizer

        def step_for(smp, opt=None, feed=False):
            """the captured step for this batch shape; `opt`: share this optimiser (a ragged last batch of the same stage call); `feed`: a
            step that gathers its own batches from an uploaded table (dataloader stages over a resident FrameStore)"""
            if not torch.is_tensor(smp["timestep_index"]):
                smp = dict(smp, timestep_index=torch.as_tensor(np.asarray(smp["timestep_index"]), device=self.device))
            key = (stage, tuple(smp["rgb"].shape), float(lr_scale))
            st = self._graphed.get(key)
            if st is not None and opt is not None and st.opt is not opt:
                st = None                                             # captured against another optimiser: capture again
            if st is None:
                params = self.get_train_parameters(stage)
                if opt is None:
                    opt = self.configure_optimizer(params, lr_scale=lr_scale)
                st = self._graphed[key] = GraphedStep(self, smp, opt, stage, warmup=0, feed=feed)
                st.fresh = True
            else:
                self.get_train_parameters(stage)
                st.update_sample(smp)
            return st

        if sample is not None:
            st = step_for(sample)
            if not getattr(st, "fresh", False):
                _reset_optimizer(st.opt)
            st.fresh = False
            n = self.cfg.pipeline[stage].num_steps if num_steps is None else num_steps
            with st.replay_stream():                                  # back-to-back replays on the step's own stream: no stream hop per step
                for _ in range(n):
                    st()
            return st.opt
        assert dataloader is not None
        opt, sched = None, None
        # frames resident as uint8 (FrameStore): the captured step is fed by timestep -- vhap_frame_ingest writes straight into its static
        # buffers -- instead of through a materialised fp32 batch and a 50 MB copy per step
        direct = isinstance(dataloader, ShuffledBatches) and self.frames is not None and self.dist is None
        H, W = self.image_size
        n_epochs = self.cfg.pipeline[stage].num_epochs
        if direct:
            def draw_epoch():
                """the batches of one pass: (timesteps, frame indices, timestep of every frame)"""
                out = []
                for ts in dataloader.index_batches():
                    ts = np.asarray(ts).reshape(-1)
                    if self._frames_of is not None:
                        fidx = np.concatenate([self._frames_of[int(t)] for t in ts])
                        ts_f = self.frame_timestep[fidx]
                    else:
                        fidx = ts_f = ts
                    out.append((ts, fidx, ts_f))
                return out

            def upload(batches):
                return (torch.as_tensor(np.concatenate([b[1] for b in batches]), device=self.device),
                        torch.as_tensor(np.concatenate([b[2] for b in batches]), device=self.device))
            # Without evaluation passes in between, the shuffles of ALL epochs are drawn ahead (the same draws in the same order) and their
            # frame indices / timesteps go up in ONE upload, and the step's own stream stays current across the epochs: an upload is a
            # blocking copy -- at an epoch boundary it waited for every queued replay and the GPU then idled until the host had caught up
            # (profiles/r04_stage_timeline_*.txt) -- per step only device-side slices of the table are touched
            ahead = not evaluate_every
            plan = [draw_epoch() for _ in range(n_epochs)] if ahead else None
            if ahead and plan:
                fidx_dev, tsf_dev = upload([b for ep in plan for b in ep])
            ctx, cur_st, o = None, None, 0
            fed = {}                      # self-feeding steps (GraphedStep.feed) -> batches left in the table they were last given

            def leave():
                nonlocal ctx, cur_st
                if ctx is not None:
                    ctx.__exit__(None, None, None)
                ctx, cur_st = None, None

            def table_for(st, batches_left, o0):
                """the batches of shape st.feed['n'] among `batches_left` (in order) as one table: device slices of the upload"""
                n, parts_f, parts_t, oo = st.feed["n"], [], [], o0
                for _, fidx, _ in batches_left:
                    if len(fidx) == n:
                        parts_f.append(fidx_dev[oo:oo + n])
                        parts_t.append(tsf_dev[oo:oo + n])
                    oo += len(fidx)
                    if len(parts_f) == st.feed["capacity"]:
                        break
                st.feed_upload(torch.cat(parts_f), torch.cat(parts_t))
                return len(parts_f)
            # The cyclic collector and a latency-bound host loop: ONE full (generation 2) collection walks every tracked object of the process
            # -- 267 k of them here, 72-76 ms, as long as 80 steps of this stage -- and whether it falls into a stage is a matter of
            # allocation counts (profiles/r04_call38_stage_gc.txt: the same stage at 17.7 k or 8.8 k frames/s from process to process).
            # What exists now is FROZEN for the length of the stage (gc.freeze: moved out of the collector's reach without being walked -- a
            # gc.collect() here would itself be those 75 ms -- so a full collection inside the loop only walks what the loop created);
            # thawed in the finally below.
            import gc
            gc.freeze()
            try:
                for epoch_i in range(n_epochs):
                    if ahead:
                        batches = plan[epoch_i]
                        rest = [b for ep in plan[epoch_i:] for b in ep]
                    else:
                        batches = draw_epoch()
                        leave()
                        fidx_dev, tsf_dev = upload(batches)
                        o = 0
                        rest = list(batches)
                        fed = {}
                    for bi, (ts, fidx, _) in enumerate(batches):
                        n = len(fidx)
                        st = self._graphed.get((stage, (n, 3, H, W), float(lr_scale)))
                        fresh = st is None or (opt is not None and st.opt is not opt)
                        if fresh:
                            leave()
                            st = step_for(self.get_sample(ts, device_index=True), opt, feed=True)
                        if opt is None:
                            opt = st.opt
                            if not getattr(st, "fresh", False):
                                _reset_optimizer(opt)
                            for grp in opt.param_groups:          # a scheduler of a previous call may have decayed them
                                grp["lr"] = grp["initial_lr"] if "initial_lr" in grp else grp["lr"]
                            sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
                        st.fresh = False
                        if st is not cur_st:                      # the step's own stream stays current across the steps of a pass: no stream hop per step
                            leave()
                            ctx = st.replay_stream()
                            ctx.__enter__()
                            cur_st = st
                        self.get_train_parameters(stage)
                        if st.feed is not None:
                            if not fed.get(st):                    # ONE table per (upload, step): every later batch of this shape, in order
                                fed[st] = table_for(st, rest[bi:], o)       # (at most GraphedStep.FEED_CAPACITY batches: then the next table)
                            fed[st] -= 1
                        elif not fresh:
                            st.update_timesteps(ts, fidx_dev[o:o + n], tsf_dev[o:o + n])
                        o += n
                        st()
                    sched.step()
                    if evaluate_every and (epoch_i + 1) % evaluate_every == 0:
                        leave()
                        self.evaluate()
            finally:
                leave()
                gc.unfreeze()
            return opt
        for epoch_i in range(n_epochs):
            for s in dataloader:
                st = step_for(s, opt)                          # one optimiser (one Adam state) for every batch shape of this call
                if opt is None:
                    opt = st.opt
                    if not getattr(st, "fresh", False):
                        _reset_optimizer(opt)
                    for grp in opt.param_groups:              # a scheduler of a previous call may have decayed them
                        grp["lr"] = grp["initial_lr"] if "initial_lr" in grp else grp["lr"]
                    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
                st.fresh = False
                with st.replay_stream():
                    st()
            sched.step()
            if evaluate_every and (epoch_i + 1) % evaluate_every == 0:
                self.evaluate()
        return opt

    def optimize_iter(self, sample, optimizer, stage, disturbance=None):
        """tracker.py:1418-1462 without the logging branches."""
        self.clear_cache()
        self.fill_cam_params_into_sample(sample)
        E_total, log_dict, *_ = self.compute_energy(sample, stage=stage, disturbance=disturbance)
        optimizer.zero_grad()
        E_total.backward()
        if self.dist is not None:
            self.dist.average_gradients([p for g in optimizer.param_groups for p in g["params"]])
        optimizer.step()
        self.global_step += 1
        return log_dict

    def initialize_next_timtestep(self, timesteps):
        """tracker.py:1515-1529 (including its skip of the last frame)."""
        stride = int(timesteps[-1]) - int(timesteps[0]) + 1
        t_src = int(timesteps[-1])
        with torch.no_grad():
            for s in range(stride):
                t_tgt = t_src + s + 1
                if t_tgt < self.n_timesteps - 1:
                    for p in (self.translation, self.rotation, self.neck_pose, self.jaw_pose, self.eyes_pose, self.expr):
                        p[t_tgt].copy_(p[t_src])
                    if self.cfg.model.use_dynamic_offset:
                        self.dynamic_offset[t_tgt].copy_(self.dynamic_offset[t_src])

    def save_result(self, path=None):
        """tracker.py:1152-1218: the npz schema consumed by export_as_nerf_dataset.py / GaussianAvatars."""
        c = lambda t: t.detach().cpu().numpy()
        out = {
            "rotation": c(self.rotation), "translation": c(self.translation), "neck_pose": c(self.neck_pose),
            "jaw_pose": c(self.jaw_pose), "eyes_pose": c(self.eyes_pose), "shape": c(self.shape), "expr": c(self.expr),
            "timestep_id": np.arange(self.n_timesteps), "n_processed_frames": np.array(self.n_timesteps),
            "image_size": np.array(self.image_size),
        }
        if not self.calibrated:
            out["focal_length"] = c(self.focal_length)
        if not self.cfg.model.tex_painted:                        # tracker.py:1184-1186
            out["tex"] = c(self.tex_pca)
        if self.cfg.model.tex_extra:
            out["tex_extra"] = c(self.tex_extra)
        if self.lights is not None:
            out["lights"] = c(self.lights)
        if self.static_offset is not None:
            out["static_offset"] = c(self.static_offset)
        if self.dynamic_offset is not None:
            out["dynamic_offset"] = c(self.dynamic_offset)
        if path is not None:
            np.savez(path, **out)
        return out

    def load_from_tracked_flame_params(self, fp):
        """tracker.py:79-130: counterpart of save_result (path, NpzFile or dict).  Per-frame arrays load their first
        min(len) rows, like the reference's load_param_list; optional entries that are missing are left untouched."""
        report = np.load(fp) if isinstance(fp, (str, os.PathLike)) else fp
        dev = self.device

        @torch.no_grad()
        def load_param(param, arr):
            param.copy_(torch.as_tensor(np.asarray(arr)).to(dev).reshape(param.shape))

        @torch.no_grad()
        def load_param_list(param, arr):
            n = min(len(param), len(arr))
            param[:n].copy_(torch.as_tensor(np.asarray(arr[:n])).to(dev))

        for k in ("rotation", "translation", "neck_pose", "jaw_pose", "eyes_pose", "expr"):
            load_param_list(getattr(self, k), report[k])
        load_param(self.shape, report["shape"])
        if self.lights is not None:
            load_param(self.lights, report["lights"])
        if not self.calibrated:
            load_param(self.focal_length, report["focal_length"])
        if not self.cfg.model.tex_painted and "tex" in report:    # tracker.py:107-109
            load_param(self.tex_pca, report["tex"])
        if self.cfg.model.tex_extra and "tex_extra" in report:
            load_param(self.tex_extra, report["tex_extra"])
        if self.cfg.model.use_static_offset and "static_offset" in report:
            load_param(self.static_offset, report["static_offset"])
        if self.cfg.model.use_dynamic_offset and "dynamic_offset" in report:
            load_param_list(self.dynamic_offset, report["dynamic_offset"])
        self.clear_cache()

    @torch.no_grad()
    def evaluate(self, batch_size=16, path=None):
        """tracker.py:1079-1117: save the parameters, then render every timestep in evaluation mode
        (stage None: `background_eval`, no disturbance, no region alignment, no regularisers) and report the photometric and
        landmark energies per timestep and their means.  The reference evaluates one timestep per call; here `batch_size`
        timesteps go through the kernels at once and the per-timestep normalisation sum|err| / (3 #(alpha > 0)) is done on
        per-frame partial sums, which is the same number.  Media / TensorBoard logging is out of scope."""
        if path is not None:
            self.save_result(path)
        w = self.cfg.w
        # per-timestep sums over all views of the timestep (the reference evaluates one dataset item = one timestep with all its views)
        photo_s, photo_n = torch.zeros(self.n_timesteps, device=self.device), torch.zeros(self.n_timesteps, device=self.device)
        lmk_s, lmk_n = torch.zeros(self.n_timesteps, device=self.device), torch.zeros(self.n_timesteps, device=self.device)
        faces = self.flame.faces
        for t0 in range(0, self.n_timesteps, batch_size):
            ts = np.arange(t0, min(t0 + batch_size, self.n_timesteps))
            sample = self.get_sample(ts, device_index=True)
            self.clear_cache()
            self.fill_cam_params_into_sample(sample)
            verts, _, lmks, albedos = self.forward_flame(sample["timestep_index"])
            if w.landmark is not None:
                rd = self.compute_lmk_energy(sample, lmks, False)[1]
                conf = sample["lmk2d"][:, :68, 2].to(verts).clone()
                conf[:, 27:36] *= 10
                tsi = sample["timestep_index"].long()
                lmk_s.index_add_(0, tsi, w.landmark * ((rd["gt_lmk2d"][:, :68] - rd["pred_lmk2d"][:, :68]).abs().sum(2) * conf).mean(1))
                lmk_n.index_add_(0, tsi, torch.ones_like(tsi, dtype=lmk_n.dtype))
            if w.photo is not None:
                gt_rgb = sample["rgb"].to(verts)
                rast_dict = self.rasterize_flame(sample, verts, faces, train_mode=True)
                lights = self.lights[None] if self.lights is not None else None
                out = self.render_rgba(rast_dict, verts, faces, albedos, lights, self.get_background_color(gt_rgb, None, None))
                rgba = out["rgba"]
                abs_sum = (gt_rgb - rgba[:, :3]).abs().sum(dim=(1, 2, 3))
                n_mask = (rgba[:, 3:] > 0).sum(dim=(1, 2, 3)) * 3
                tsi = sample["timestep_index"].long()
                photo_s.index_add_(0, tsi, w.photo * abs_sum)
                photo_n.index_add_(0, tsi, n_mask.to(photo_n.dtype))
        photo = (photo_s / photo_n.clamp_min(1)).cpu().numpy() if w.photo is not None else photo_s.cpu().numpy()
        lmk = (lmk_s / lmk_n.clamp_min(1)).cpu().numpy()
        return {"photo": photo, "lmk": lmk, "mean_photo": float(photo.mean()), "mean_lmk": float(lmk.mean())}


def _set_ints(dst, values):
    """dst (int32 device tensor) = values, as one tiny launch carrying them in its arguments (no blocking copy: vhap_set_floats on the bits)"""
    # the BITS travel, never a float value: small ints are float32 subnormals (flushed to zero by a host thread in FTZ / DAZ mode --
    # torch.set_flush_denormal(True) -- on any int -> float -> double -> float conversion) and some patterns are NaNs
    raw = np.ascontiguousarray(np.asarray(values, np.int32)).tobytes()
    n = len(raw) // 4
    _lib.check(_lib.lib().vhap_set_floats(dst.data_ptr(), (ctypes.c_float * n).from_buffer_copy(raw), n,
                                          torch.cuda.current_stream().cuda_stream), "vhap_set_floats")


def _reset_optimizer(opt):
    """Back to a freshly constructed Adam (the reference builds a new optimiser per optimize_stage call, tracker.py:1399)."""
    if isinstance(opt, NV.HipAdam):
        opt.reset_state()
        return
    for st in opt.state.values():                                  # torch.optim.Adam(capturable=True): state tensors live on the device
        for k in ("exp_avg", "exp_avg_sq", "step"):
            if k in st and torch.is_tensor(st[k]):
                st[k].zero_()


class ShuffledBatches:
    """What DataLoader(dataset, batch_size, shuffle=True) is to the reference's global tracking (tracker.py:1376-1385): every pass over
    it yields the frames once, in a fresh random order, in batches (the last one may be smaller)."""

    shuffle = True

    def __init__(self, tracker, batch_size, device_index=False, generator=None):
        self.tr, self.bs, self.device_index, self.gen = tracker, int(batch_size), device_index, generator

    def __len__(self):
        return (self.tr.n_timesteps + self.bs - 1) // self.bs

    def index_batches(self):
        """the timesteps of every batch of one pass (the sampling half of __iter__: a captured step whose frames live in a uint8
        FrameStore is fed by GraphedStep.update_timesteps, without materialising the fp32 batch in between)"""
        tr = self.tr
        if tr.dist is not None and self.gen is None:
            # frame sharding: every rank must draw the SAME permutation (each then takes its slice of every batch); a generator seeded
            # with a value broadcast from rank 0, advanced in lock step afterwards
            self.gen = torch.Generator().manual_seed(tr.dist.broadcast_int(int(torch.randint(0, 2 ** 31 - 1, (1,)))))
        perm = torch.randperm(tr.n_timesteps, generator=self.gen).numpy()
        for i in range(0, len(perm), self.bs):
            yield perm[i:i + self.bs]

    def __iter__(self):
        tr = self.tr
        for ts in self.index_batches():
            s = tr.get_sample(ts, device_index=self.device_index)
            yield tr.dist.shard_sample(s) if tr.dist is not None else s


class CapturedPlan:
    """One captured piece of a step: recorded under HIP stream capture into a hipGraph that is KEPT (never instantiated) and replayed by
    the library's own executor (include/vhap_hip.h "Step plans", csrc/plan.hip): the nodes laid out over streams of the plan's own, plain
    kernel launches, cross-stream edges as event pairs -- one C call per replay.  hipGraphLaunch is not used: ROCm 7's re-partitions the
    branches on its internal streams and dereferences garbage when the launch stream shares a hardware queue with two of them
    (profiles/r02_graph_launch_crash.txt).  A graph holding a node the executor does not know (anything but kernel / memset / empty nodes)
    falls back to torch's replay of the instantiated graph, with a warning."""

    MAX_STREAMS = 4

    def __init__(self, side_base=0):
        self.g = torch.cuda.CUDAGraph(keep_graph=True)
        self.plan = None
        self.side_base = int(side_base)      # the plan's side stream k = the thread's pool stream side_base + k (vhap_plan_set_side_base)
        self.fallback = os.environ.get("VHAP_EXECUTOR", "plan") == "graph"     # (A/B and debugging: the runtime's own graph launch)

    def capture(self, **kw):
        outer = self

        class _Ctx:
            # The cyclic collector stays off for the length of the capture: a dead torch.cuda.CUDAGraph (or anything else whose destructor
            # calls the runtime) collected in the middle of it is an "operation not permitted when stream is capturing" that ends the
            # capture, and the error thrown from the next graph destructor ends the PROCESS (seen once in ~60 bench runs of round 4:
            # profiles/r04_call33_capture_abort.txt).  torch.cuda.graph collects once on entry, which leaves the window open.
            def __enter__(self):
                import gc
                self._gc = gc.isenabled()
                self.ctx = torch.cuda.graph(outer.g, **kw)
                r = self.ctx.__enter__()
                gc.disable()
                return r

            def __exit__(self, *a):
                import gc
                try:
                    r = self.ctx.__exit__(*a)
                finally:
                    if self._gc:
                        gc.enable()
                if a[0] is None:
                    outer._finish()
                return r
        return _Ctx()

    def _finish(self):
        if self.fallback:
            self.g.instantiate()
            return
        L = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(L.vhap_plan_set_side_base(self.side_base), "vhap_plan_set_side_base")
        try:
            rc = L.vhap_plan_from_graph(self.g.raw_cuda_graph(), self.MAX_STREAMS, ctypes.byref(h))
        finally:
            L.vhap_plan_set_side_base(0)
        if rc == -5:                                               # VHAP_E_UNSUPPORTED
            import warnings
            warnings.warn("CapturedPlan: the captured step holds a node type the plan executor does not replay; using hipGraphLaunch")
            self.fallback = True
            self.g.instantiate()
            return
        _lib.check(rc, "vhap_plan_from_graph")
        self.plan = h

    def pool(self):
        return self.g.pool()

    def replay(self, defer_join=False):
        """`defer_join`: the replay leaves its open tails (open_tails()) running -- the current stream does not wait for them; join() does."""
        if self.plan is None:
            return self.g.replay()
        _lib.check(_lib.lib().vhap_plan_launch(self.plan, torch.cuda.current_stream().cuda_stream,
                                               _lib.CALL_PLAN_DEFER_JOIN if defer_join else 0), "vhap_plan_launch")

    def join(self):
        if self.plan is not None:
            _lib.check(_lib.lib().vhap_plan_join(self.plan, torch.cuda.current_stream().cuda_stream), "vhap_plan_join")

    def _indices(self, fn):
        if self.plan is None:
            return []
        n = fn(self.plan, None, 0)
        idx = (ctypes.c_int * max(n, 1))()
        fn(self.plan, idx, n)
        return [int(idx[k]) for k in range(n)]

    def node_name(self, k):
        buf = ctypes.create_string_buffer(512)
        _lib.lib().vhap_plan_node_name(self.plan, k, buf, 512)
        return buf.value.decode()

    def _names(self, fn):
        return [self.node_name(k) for k in self._indices(fn)]

    def deferred_join_hazards(self, log, extra_head_ranges=()):
        """May the next replay's head run under this replay's open tails?  Decided on the BUFFERS (`log`: the _lib.AccessLog recorded while
        this plan's graph was captured): every node a DEFER_JOIN replay leaves open, and every node of the next replay that is not ordered
        behind all of them (plus `extra_head_ranges`: what the host writes between replays), maps back to the C-ABI call that created it
        and to the byte ranges that call was given; the two sets must not overlap.  -> (ok, report lines)."""
        L = _lib.lib()
        lines, ok = [], True
        sets = {}
        for kind, fn in (("open tail", L.vhap_plan_open_tails), ("free head", L.vhap_plan_free_heads)):
            rs = []
            for k in self._indices(fn):
                hit = log.ranges_of_node(L.vhap_plan_node_handle(self.plan, k))
                name = self.node_name(k).split("(")[0]
                if hit is None:
                    ok = False
                    lines.append(f"{kind} {k} {name}: created by no recorded call (UNKNOWN buffers): no deferred join")
                else:
                    lines.append(f"{kind} {k} {name} <- {hit[0]}: {len(hit[1])} buffers, {sum(n for _, n in hit[1])} bytes")
                    rs += hit[1]
            sets[kind] = rs
        if not sets["open tail"]:
            return False, lines + ["no open tails: nothing to defer"]
        clash = _lib.ranges_overlap(sets["open tail"], sets["free head"] + list(extra_head_ranges))
        if clash is not None:
            ok = False
            lines.append(f"open tails and free heads / host writes overlap: {clash[0][1]} bytes at {clash[0][0]:#x} vs {clash[1][1]} bytes at {clash[1][0]:#x}")
        else:
            lines.append(f"open tails ({len(sets['open tail'])} buffers) and free heads + host writes "
                         f"({len(sets['free head']) + len(extra_head_ranges)} buffers) are disjoint")
        return ok, lines

    def nodes_touching_not_behind(self, log, ranges, stream_index=1):
        """Nodes of this plan that touch any of `ranges` [(address, bytes)] -- by the byte ranges of the C-ABI call that created them (`log`, the
        _lib.AccessLog of the capture); a node no recorded call accounts for counts as touching -- and are NOT ordered behind the plan's
        stream `stream_index` (1 = its first side stream): neither on it, nor behind one of its nodes through dependencies or stream order.
        What a caller needs to know before it lets ONLY that stream wait for a producer of those buffers (GraphedStep: the all-gathered
        texture).  -> [(node index, kernel name, call name | 'UNKNOWN')]"""
        L = _lib.lib()
        rows = []
        for line in self.describe().splitlines():
            head, _, rest = line.partition("<-")
            parts = head.split()
            if len(parts) < 2 or not parts[0].isdigit():
                continue
            deps = [int(t) for t in rest.split("|")[0].split() if t.isdigit()]
            rows.append((int(parts[0]), int(parts[1][1:]), deps))
        behind, last_on, bad = {}, {}, []
        for k, st, deps in rows:
            b = st == stream_index or any(behind.get(d, False) for d in deps) or behind.get(last_on.get(st), False)
            behind[k] = b
            last_on[st] = k
            hit = log.ranges_of_node(L.vhap_plan_node_handle(self.plan, k))
            touches = hit is None or _lib.ranges_overlap(hit[1], ranges) is not None
            if touches and not b:
                bad.append((k, self.node_name(k).split("(")[0], "UNKNOWN" if hit is None else hit[0]))
        return bad

    def open_tails(self):
        """Kernel names of the nodes a defer_join replay leaves un-joined."""
        return self._names(_lib.lib().vhap_plan_open_tails)

    def free_heads(self):
        """Kernel names of the nodes of the next replay that are NOT ordered behind those open tails."""
        return self._names(_lib.lib().vhap_plan_free_heads)

    def describe(self):
        if self.plan is None:
            return "(hipGraph replay)"
        L = _lib.lib()
        n = L.vhap_plan_describe(self.plan, None, 0)
        buf = ctypes.create_string_buffer(n)
        L.vhap_plan_describe(self.plan, buf, n)
        return buf.value.decode()

    def timed(self):
        """One replay with per-node timing events -> [(kernel name, stream-relative start us, duration us)]; blocks."""
        if self.plan is None:
            raise RuntimeError("CapturedPlan.timed() needs the plan executor")
        L = _lib.lib()
        n = ctypes.c_int()
        L.vhap_plan_info(self.plan, ctypes.byref(n), None, None)
        st, du = (ctypes.c_float * n.value)(), (ctypes.c_float * n.value)()
        _lib.check(L.vhap_plan_launch_timed(self.plan, torch.cuda.current_stream().cuda_stream, st, du, n.value), "vhap_plan_launch_timed")
        out = []
        buf = ctypes.create_string_buffer(512)
        for k in range(n.value):
            L.vhap_plan_node_name(self.plan, k, buf, 512)
            out.append((buf.value.decode(), float(st[k]), float(du[k])))
        return out

    def __del__(self):
        try:
            if self.plan is not None:
                _lib.lib().vhap_plan_destroy(self.plan)         # before the graph it borrows its kernel arguments from
                self.plan = None
        except Exception:
            pass


class GraphedStep:
    """One optimiser step recorded once under HIP stream capture and replayed by the library's plan executor (CapturedPlan; SURVEY section
    8(f) rank 3: the 50-500 identical steps of a stage are launch-bound in eager mode -- ~1300 launches per step).

    One GPU: the whole step (vhap_amd/step.py::NativeStep: forward, backward, Adam -- ~45 C-ABI calls on up to three chains) is ONE
    plan.  Frame sharding: the collectives stay ordinary eager calls between the plans --

        F  : forward -> every energy term, photometric numerator S and alpha count N
             [all-reduce N over ranks]
        B  : backward: pixel chain + the complete texture gradient      [asynchronous all-reduce of the texture gradient (50 MB)]
        B2 : backward: geometry chain, underneath that collective       [all-reduce (average) of the flat bucket of every other gradient]
        A  : Adam

    (the autograd formulation of a stage the NativeStep does not cover is captured as F / B / A the same way).
    Everything the plans touch is static: the sample tensors, the parameters, their .grad and the Adam state.
    A new batch is fed by copying into `self.sample` (same shapes) -- exactly the sequential-tracking pattern.
    Replays always go to the step's own launch stream."""

    # What the shipped photometric step is EXPECTED to leave open / start early (documentation and the tests' cross-check; the decision
    # itself is made on the buffers: CapturedPlan.deferred_join_hazards): the texture gradient's sort -- workspace clear, count, scan,
    # scatter -- its tile pass, the texture finish + Adam; and the geometry head
    TEX_TAIL = ("vhap_zero_words_kernel", "texbin_pass_kernel", "texbin_scan_kernel", "texgrad_tile_kernel", "tex_prep_bwd_kernel",
                "tex_carry_border_kernel")
    GEOMETRY_HEAD = ("camera_fwd_kernel", "frame_prep_fwd_kernel", "flame_skin_fwd_kernel", "flame_skin_clip_fwd_kernel", "bin_build_kernel")

    def __init__(self, tracker, sample, optimizer, stage, warmup=2, unroll=1, feed=False):
        """`feed`: (dataloader stages over a resident FrameStore) let the captured step gather its own batch from an uploaded table --
        feed_upload() -- instead of being fed by the host between replays; see _enable_feed."""
        assert tracker.fused, "graph capture needs the fused (sync-free) path"
        self.feed, self._want_feed = None, bool(feed)
        self.defer_join, self._in_loop = False, False
        self.tr, self.opt, self.stage = tracker, optimizer, stage
        dev = tracker.device
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        ts = sample["timestep_index"]
        ts = ts if torch.is_tensor(ts) else torch.as_tensor(np.asarray(ts), device=dev)
        self.sample = {"rgb": sample["rgb"].clone(), "lmk2d": sample["lmk2d"].clone(), "timestep_index": ts.clone()}
        for k in ("intrinsic", "extrinsic"):
            if k in sample and tracker.calibrated:
                self.sample[k] = sample[k].clone()
        from .step import NativeStep
        self.ns = None
        self.single = False
        self.unroll = 1
        use_native_step = os.environ.get("VHAP_NATIVE_STEP", "1") != "0" and NativeStep.supported(tracker, stage) and \
            isinstance(optimizer, NV.HipAdam)
        side = _lib.private_stream("warm", dev)                   # (library-owned streams, never torch's pool: _lib.private_stream)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                                # optional real steps before the capture
                tracker.optimize_iter(dict(self.sample), optimizer, stage)
            # dry pass: populate every lazy cache (region tables, mesh tables, FLAME bases) outside the capture ...
            s = dict(self.sample)
            tracker.fill_cam_params_into_sample(s)
            E_dry, *_ = tracker.compute_energy(s, stage=stage)
            torch.autograd.grad(E_dry, self.params, allow_unused=True)   # backward too: BLAS kernels for every GEMM shape get loaded now
            del E_dry
            # ... and create the Adam state with a zero-gradient step (a no-op on the parameters), then rewind its counter
            for p in self.params:
                p.grad = torch.zeros_like(p)
            if isinstance(optimizer, NV.HipAdam):
                if optimizer._tab is None:
                    optimizer._build()
                optimizer.sync_lr()
            elif len(optimizer.state) == 0:
                optimizer.step()
                for st in optimizer.state.values():
                    st["step"].zero_()
            if use_native_step:
                # the whole step as a hand-chained call sequence (vhap_amd/step.py): no autograd glue launches
                self.ns = NativeStep(tracker, self.sample, stage)
                self.ns.forward()
                self.ns.backward(1)
        torch.cuda.current_stream().wait_stream(side)
        self.inv_n = torch.zeros((), device=dev)
        # Replays go to a stream of the library's own (never torch's pool, never the null stream): the plan executor forks the captured
        # step's side chains from it and joins them back into it
        self.stream = _lib.private_stream("launch", dev)
        # with a process group alive, its helper threads (RCCL watchdog, heartbeat) issue runtime calls of their own: keep those from
        # invalidating a capture in progress on this thread
        cap = dict(capture_error_mode="thread_local") if tracker.dist is not None else {}
        self.gF, self.gB, self.gA = CapturedPlan(), CapturedPlan(), CapturedPlan()
        if self.ns is not None:
            ns = self.ns
            world = tracker.dist.world_size if tracker.dist is not None else 1
            self.single = tracker.dist is None or not tracker.dist.sharded      # (a one-rank group takes the sharded form under VHAP_FORCE_DIST)
            if not self.single:
                ns.n_global = ns.accF[17:18]                        # the alpha count: all-reduced IN PLACE before every backward replay
            if self.single:
                # nothing happens between the passes on one GPU: ONE plan per step.  `unroll` > 1: that many consecutive steps per replay
                self.unroll = max(1, int(unroll))
                # the texture's Adam update (7 x 50 MB of traffic at T = 2048) is fused into the last kernel of the backward's texture chain;
                # the step counter is advanced at the head of the step (side branch), so that no piece of the update has to be last
                tex = tracker.tex_extra
                split = tex is not None and any(p is tex for p in self.params) and len(self.params) > 1 and ns.tex_bwd_on and \
                    ns.photometric and ns.overlap and hasattr(optimizer, "advance")
                ns.step_optimizer = optimizer if split else None
                # the carried texture (step.py: enable_carry): the finish + Adam pass hands the next replay its albedo + pyramid level 1;
                # replays prime it when they cannot know it is current (_carry_prime_if_needed)
                self._carry_epoch, self._carry_loop = None, 0
                if split and os.environ.get("VHAP_TEX_CARRY", "1") != "0":
                    ns.enable_carry(keep_grad=os.environ.get("VHAP_TEX_KEEP_GRAD", "0") == "1")
                if ns.photometric:
                    ns.one_graph = True
                    ns.accF.zero_()
                    ns._acc_clean = True
                self._enable_feed()
                self._access = _lib.AccessLog()
                with self._access, self.gF.capture(**cap):
                    for _ in range(self.unroll):
                        ns.forward()
                        if split:
                            ns.backward(1, optimizer=optimizer)
                        else:
                            ns.backward(1)
                            optimizer.step()
            elif not ns.photometric:
                # Frame sharding of a landmark-only stage: forward, backward, ONE all-reduce of the (small) gradient arena, Adam
                self.lmk_only = True
                with self.gF.capture(**cap):
                    ns.forward()
                with self.gB.capture(pool=self.gF.pool(), **cap):
                    ns.backward(world)
                with self.gA.capture(pool=self.gF.pool(), **cap):
                    optimizer.step()
            else:
                # Frame sharding.  The big collective is the texture gradient (50 MB at T = 2048).  The backward is captured in two plans:
                # 'pixel_tex' makes that gradient final first, its asynchronous all-reduce is launched, and 'geometry' (G-buffer backward,
                # normals, skinning, per-frame parameters: ~0.3 ms) runs underneath it; the small gradients follow in a second collective.
                # SHARDED TEXTURE UPDATE (round 4): the texture gradient is not all-reduced and the 50 MB texture not updated N times over.
                # 'pixel_tex' stops at the gradient pyramid and folds it into level 0; a reduce-scatter hands rank r the mean of rows
                # [r T / N, (r + 1) T / N); the Adam plan finishes and updates THOSE rows (vhap_tex_prep_bwd_adam_rows: 1 / N of the 124 us /
                # 497 MB pass) and an asynchronous all-gather of the updated rows runs until the next step's forward needs the texture.
                # Same bytes on the wire as the all-reduce it replaces (a ring all-reduce IS this reduce-scatter + all-gather).
                T = int(tracker.tex_extra.shape[-1]) if tracker.tex_extra is not None else 0
                tex = tracker.tex_extra
                self.tex_sharded = bool(ns.tex_bwd_on and ns.pca is None and tex is not None and any(p is tex for p in self.params) and T % (16 * world) == 0 and
                                        isinstance(optimizer, NV.HipAdam) and os.environ.get("VHAP_TEX_SHARDED", "1") != "0")
                if self.tex_sharded and not tracker.dist.tex_sharded_usable():   # (every collective of the path once, on small tensors, all ranks together)
                    self.tex_sharded = False
                if self.tex_sharded:
                    ns.split_tex = True
                    self.tex_rows = T // world
                    self.tex_row0 = tracker.dist.rank * self.tex_rows
                    self.tex_strip = torch.zeros(self.tex_rows, T, 3, device=dev)        # reduce-scatter output: this rank's rows of the level-0 gradient
                    self._tex_gather = None
                    self.comm = _lib.private_stream("comm", dev)                         # carries the texture collective (see _replay)
                if self.tex_sharded and ns.overlap:
                    # the pixel plan leaves its texture chain running under the geometry plan: it takes the pool's SECOND side stream (the
                    # geometry plan's one side chain takes the first); and the four streams the sharded loop keeps busy -- launch, two
                    # side streams, communication -- are touched one after the other, which puts them on four different hardware
                    # queues (HIP binds a stream to a queue at its first command, round-robin over GPU_MAX_HW_QUEUES = 4)
                    self.gB = CapturedPlan(side_base=1)
                    scratch = torch.zeros(4, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize(dev)
                    with torch.cuda.stream(self.stream):
                        scratch[:1].zero_()
                    _lib.check(_lib.lib().vhap_plan_touch_side_streams(2, scratch[1:].data_ptr()), "vhap_plan_touch_side_streams")
                    with torch.cuda.stream(self.comm):
                        scratch[3:].zero_()
                    torch.cuda.synchronize(dev)
                # the texture's own path -- tile accumulation, fold, reduce-scatter, row finish + Adam, all-gather -- runs on the side / the
                # communication stream from end to end; the step counter is advanced at the head of the step so that the two pieces of
                # the Adam update (this rank's texture rows there, everything else on the launch stream) need no order between them
                self.tex_path = bool(self.tex_sharded and ns.overlap and hasattr(optimizer, "advance"))
                # texture first (8 GPUs: the exchange is the critical path -- reduce-scatter and all-gather are ~0.15 ms each on xGMI): the
                # geometry plan starts when the gradient is folded and runs under the reduce-scatter, instead of sharing the chip with
                # the tile accumulation (69 us alone, 174 us next to the G-buffer backward).  Default from the world size.
                tf = os.environ.get("VHAP_SHARD_TEX_FIRST")
                self.tex_first = self.tex_path and (world >= 4 if tf is None else tf == "1")
                # texture first, round 6: the geometry plan waits for the TILE accumulation only (two LDS-bound kernels beside each other:
                # 147 + 114 us, alone 70 + 97) -- the fold of the gradient pyramid (47 us of streaming) is issued on the communication
                # stream in front of the reduce-scatter and runs beside the G-buffer backward
                # (measured at world size 1, profiles/r06_call12_sharded_ab.txt: the fold then takes 103 us beside the G-buffer backward
                # instead of 47 alone and the backward 134 instead of 114 -- no gain; off unless VHAP_SHARD_FOLD_OUTSIDE=1)
                ns.fold_outside = bool(self.tex_first) and os.environ.get("VHAP_SHARD_FOLD_OUTSIDE", "0") == "1"
                if self.tex_path:
                    ns.step_optimizer = optimizer                   # (the forward's side branch issues optimizer.advance())
                    ns.sort_in_backward = True                      # (the uv-tile sort beside the shading backward, like the one-plan step)
                self._accessF = _lib.AccessLog()
                with self._accessF, self.gF.capture(**cap):
                    ns.forward()
                pool = self.gF.pool()
                self.gB2 = CapturedPlan()
                with self.gB.capture(pool=pool, **cap):
                    ns.backward(world, part="pixel_tex")
                with self.gB2.capture(pool=pool, **cap):
                    ns.backward(world, part="geometry")
                if self.tex_path:
                    self.gAt = CapturedPlan()
                    with self.gAt.capture(pool=pool, **cap):
                        ns.tex_finish_rows(optimizer, self.tex_strip, self.tex_row0, self.tex_rows, advanced=True)
                    with self.gA.capture(pool=pool, **cap):
                        optimizer.step(skip=(tex,), advanced=True)
                else:
                    with self.gA.capture(pool=pool, **cap):
                        if self.tex_sharded:
                            ns.tex_finish_rows(optimizer, self.tex_strip, self.tex_row0, self.tex_rows)      # (reads the step counter + 1 ...)
                            optimizer.step(skip=(tex,))                                                      # (... which this call then advances)
                        else:
                            optimizer.step()
            # Step k+1 under step k's texture tail: inside replay_stream() the single-GPU plan is replayed WITHOUT joining its side
            # streams at the end -- what it leaves open (the texture gradient's sort, tile pass, finish + Adam on the side stream) keeps
            # running while the next replay's launch-stream kernels up to the rasteriser (camera, per-frame parameters, skinning, binning)
            # start; the next replay's texture chain follows the tail on its own stream, and the rasteriser waits for that chain.
            # Whether that is safe is decided on the BUFFERS the nodes touch, recorded call by call during the capture (_lib.AccessLog):
            # the open tails' byte ranges must be disjoint from those of every node of the next replay that is not ordered behind them
            # (vhap_plan_free_heads: per tail stream) and from what the host writes between two replays (a new batch into the sample
            # buffers); a node no recorded call accounts for -> joined replays.  A changed learning rate (the tail reads the table) joins
            # first (__call__).
            # THE PRECISE WAIT of the sharded texture path (_replay): only the forward plan's first side stream waits for the communication
            # stream (reduce-scatter, row finish + Adam, all-gather).  That is safe iff every node of the forward plan that touches what
            # the communication stream reads or writes -- the texture parameter and its Adam state, the assembled albedo (the row finish
            # reads it), the gradient pyramid (the arena clear zeroes the reduce-scatter's input), the strip, the step counter
            # (advanced by the forward plan, read by the row finish) -- sits on that stream or behind it.  Checked HERE, on the buffers
            # the captured calls were given, not assumed from the capture order (round-5 advisor); otherwise the launch stream waits.
            self._precise_ok, self._precise_report = False, ["not a sharded texture path"]
            if getattr(self, "tex_path", False) and self.gF.plan is not None:
                tex = tracker.tex_extra
                st_t = optimizer.state[tex]
                touched = [tex, ns.albedo_tex, ns.g["d_tex"], self.tex_strip, st_t["exp_avg"], st_t["exp_avg_sq"], ns.g["tex_extra"],
                           optimizer.step_count]
                ranges = [(t.data_ptr(), t.numel() * t.element_size()) for t in touched if t is not None and t.numel()]
                bad = self.gF.nodes_touching_not_behind(self._accessF, ranges, stream_index=1)
                self._precise_ok = not bad
                self._precise_report = [f"node {k} {nm} <- {call}: touches a buffer of the texture path but is not ordered behind side stream 0"
                                        for k, nm, call in bad] or ["every forward-plan node that touches the texture path's buffers is on / behind side stream 0"]
            self.defer_join, self.defer_report = False, ["deferred join not considered (sharded step, hipGraph fallback or VHAP_DEFER_JOIN=0)"]
            if self.single and self.gF.plan is not None and os.environ.get("VHAP_DEFER_JOIN", "1") != "0":
                host_writes = [(t.data_ptr(), t.numel() * t.element_size()) for t in self.sample.values()]
                if getattr(self, "feed", None) is not None:          # (a table upload between two replays)
                    host_writes += [(self.feed[k].data_ptr(), self.feed[k].numel() * self.feed[k].element_size()) for k in ("frames", "ts", "cursor")]
                self.defer_join, self.defer_report = self.gF.deferred_join_hazards(self._access, host_writes)
            self.E = ns.log[15]
            self.log_dict = ns.log_dict()
            # (one GPU: the forward accumulators are cleared at the END of the captured step -- the photometric sum / count of the last step
            # are in the log vector, not here)
            self.S, self.N = (None, None) if self.single else (ns.accF[16], ns.accF[17])
            return
        tracker._split = {}
        try:
            with self.gF.capture(**cap):
                s = dict(self.sample)
                tracker.clear_cache()
                tracker.fill_cam_params_into_sample(s)
                E_rest, self.log_dict, *_ = tracker.compute_energy(s, stage=stage)
                self.S, self.N = tracker._split.get("S"), tracker._split.get("N")     # None: the stage has no photometric term
            pool = self.gF.pool()
            for p in self.params:
                p.grad = None
            with self.gB.capture(pool=pool, **cap):
                E = E_rest + tracker.cfg.w.photo * self.S * self.inv_n if self.S is not None else E_rest
                # .grad is None: autograd hands its gradient tensors over to the parameters (no copy); they live in the graph's
                # pool at fixed addresses and are rewritten by every replay
                torch.autograd.backward(E, inputs=self.params)
                for p in self.params:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self.E = E.detach()
            with self.gA.capture(pool=pool, **cap):
                optimizer.step()
        finally:
            tracker._split = None

    FEED_CAPACITY = 4096          # batches per uploaded table

    def _enable_feed(self):
        """Frames resident as uint8 (FrameStore), one GPU, photometric stage with the early texture branch: the captured step feeds itself
        (NativeStep._feed_batch) from a table of batches uploaded once per pass (feed_upload) -- nothing but the replay is enqueued per step.
        Off: VHAP_STEP_FEED=0 (the host enqueues ingest + index copies between replays: update_timesteps)."""
        tr, ns = self.tr, self.ns
        self.feed = None
        if not self._want_feed or tr.frames is None or not self.single or ns is None or not (ns.photometric and ns.deferred and ns.overlap) or ns.dyn or \
                tr.frames.alpha is not None or os.environ.get("VHAP_STEP_FEED", "1") == "0":
            return
        dev, n = self.sample["timestep_index"].device, int(self.sample["timestep_index"].shape[0])
        if not (ns.ts.data_ptr() == self.sample["timestep_index"].data_ptr() and ns.lmk2d.data_ptr() == self.sample["lmk2d"].data_ptr() and
                ns.rgb.data_ptr() == self.sample["rgb"].data_ptr() and tr.dataset["lmk2d"].is_contiguous() and tr.dataset["lmk2d"].dtype == torch.float32):
            return                                                  # (the step reads copies, not the static sample buffers)
        if ns.calibrated and not (ns.K_in.is_contiguous() and ns.RT_in.is_contiguous() and tr.dataset["intrinsic"].is_contiguous() and
                                  tr.dataset["extrinsic"].is_contiguous() and ns.K_in.dtype == ns.RT_in.dtype == torch.float32 and
                                  tr.dataset["intrinsic"].dtype == tr.dataset["extrinsic"].dtype == torch.float32):
            return
        cap = self.FEED_CAPACITY
        self.feed = {"frames": torch.zeros(cap * n, dtype=torch.int64, device=dev), "ts": torch.zeros(cap * n, dtype=torch.int64, device=dev),
                     "cursor": torch.tensor([0, 1], dtype=torch.int32, device=dev), "frame_index": torch.zeros(n, dtype=torch.int64, device=dev),
                     "capacity": cap, "n": n}
        # until a table is uploaded: a table of ONE batch, the one the step was built on (taken again by every replay)
        self.feed["ts"][:n] = self.sample["timestep_index"]
        self.feed["frames"][:n] = self._frame_index_of(self.sample["timestep_index"])
        ns.feed = self.feed

    def _frame_index_of(self, timestep_dev):
        tr = self.tr
        if tr._frames_of is None:
            return timestep_dev
        ts = timestep_dev.cpu().numpy()
        # multi-view: the frames of a batch are all views of its timesteps in order -- recover them from the distinct timesteps
        uniq = list(dict.fromkeys(int(t) for t in ts))
        return torch.as_tensor(np.concatenate([tr._frames_of[t] for t in uniq]), device=timestep_dev.device)

    def feed_upload(self, frame_index_dev, timestep_dev):
        """The batches this step will take, in order: [m * n] frame indices / timesteps (device tensors).  Resets the cursor."""
        f = self.feed
        m = frame_index_dev.numel() // f["n"]
        if m > f["capacity"] or frame_index_dev.numel() != m * f["n"]:
            raise ValueError("GraphedStep.feed_upload: table too large or not a whole number of batches")
        f["frames"][:m * f["n"]].copy_(frame_index_dev)
        f["ts"][:m * f["n"]].copy_(timestep_dev)
        _set_ints(f["cursor"], (0, m))

    def update_timesteps(self, timesteps, frame_index_dev=None, timestep_dev=None):
        """Feed the batch of these timesteps from the tracker's dataset; with a uint8 FrameStore the frames are converted straight
        into the static rgb buffer (no intermediate fp32 batch).  Multi-view datasets: every view of the timesteps, each frame carrying
        ITS timestep -- the same frame-index expansion as GlobalTracker.get_sample.  `frame_index_dev` / `timestep_dev`: the frame
        indices and their timesteps already on the device (optimize_stage uploads a whole pass of shuffled batches at once: a per-step
        host-to-device copy of the indices would stall the launch queue every step)."""
        tr = self.tr
        if tr.frames is None:
            return self.update_sample(tr.get_sample(timesteps, device_index=True))
        dev = self.sample["timestep_index"].device
        if self.feed is not None:                                  # a self-feeding step: a table of this one batch
            if frame_index_dev is None:
                timestep_dev = torch.as_tensor(np.asarray(timesteps).reshape(-1), device=dev) if not torch.is_tensor(timesteps) else timesteps
                if tr._frames_of is not None:
                    # (multi-view: every view of the DISTINCT timesteps, in order -- a per-frame timestep list, e.g. a sample's own
                    # `timestep_index`, names each timestep once per view)
                    uniq = list(dict.fromkeys(int(t) for t in np.asarray(timestep_dev.cpu()).reshape(-1)))
                    fidx = np.concatenate([tr._frames_of[t] for t in uniq])
                    frame_index_dev = torch.as_tensor(fidx, device=dev)
                    timestep_dev = torch.as_tensor(tr.frame_timestep[fidx], device=dev)
                else:
                    frame_index_dev = timestep_dev
            return self.feed_upload(frame_index_dev, timestep_dev)
        if frame_index_dev is None:
            ts = np.asarray(timesteps.cpu() if torch.is_tensor(timesteps) else timesteps).reshape(-1)
            if tr._frames_of is not None:
                # (multi-view: every view of the DISTINCT timesteps, in order -- the same normalisation as the self-feeding branch above: a
                # per-frame timestep list, e.g. a sample's own `timestep_index`, names each timestep once per view)
                uniq = list(dict.fromkeys(int(t) for t in ts))
                fidx = np.concatenate([tr._frames_of[t] for t in uniq])
                ts = tr.frame_timestep[fidx]
            else:
                fidx = ts
            frame_index_dev = torch.as_tensor(fidx, device=dev)
            timestep_dev = torch.as_tensor(ts, device=dev)
        idx = frame_index_dev
        if idx.shape != self.sample["timestep_index"].shape:
            raise ValueError("GraphedStep.update_timesteps: batch size differs from the captured one")
        tr.frames.batch(idx, out=self.sample["rgb"])
        torch.index_select(tr.dataset["lmk2d"], 0, idx, out=self.sample["lmk2d"])
        self.sample["timestep_index"].copy_(timestep_dev)
        for k in ("intrinsic", "extrinsic"):
            if k in self.sample:
                torch.index_select(tr.dataset[k], 0, idx, out=self.sample[k])

    def update_sample(self, sample):
        """Feed a new batch of the SAME shapes: copied into the static buffers the graphs read."""
        if self.feed is not None:      # (a self-feeding step re-gathers its batch from the frame store: hand it the timesteps)
            return self.update_timesteps(sample["timestep_index"])
        for k, dst in self.sample.items():
            src = sample[k]
            if not torch.is_tensor(src):
                src = torch.as_tensor(np.asarray(src), device=dst.device)
            if src.shape != dst.shape:
                raise ValueError(f"GraphedStep.update_sample: {k} has shape {tuple(src.shape)}, captured {tuple(dst.shape)}")
            dst.copy_(src.to(dst.dtype))

    def __call__(self):
        if isinstance(self.opt, NV.HipAdam):
            if self.opt.lr_changed() and (self.defer_join or getattr(self, "_tex_gather", None) is not None):
                # the open texture tail of the last replay reads the lr table -- on a side stream (one GPU), or on the communication
                # stream behind the reduce-scatter (sharded texture path: the row finish + Adam plan): the write below must come behind it
                self.join()
                if torch.cuda.current_stream().cuda_stream != self.stream.cuda_stream:
                    torch.cuda.current_stream().wait_stream(self.stream)
            self.opt.sync_lr()                                     # lr schedulers act on the host copy
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != self.stream.cuda_stream:              # (inside replay_stream() the step's stream IS current)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self._replay()
                self.wait_texture()                                # (a lone step: the caller may read the texture next)
            cur.wait_stream(self.stream)
        else:
            self._replay()
        self.tr.global_step += self.unroll
        self.opt._opt_called = True        # (the replay WAS optimizer.step(): torch's lr schedulers look for this flag and warn otherwise)
        return self.E

    def replay_stream(self):
        """Context for a loop of replays: makes the step's private stream current (after it has caught up with the caller's stream), so that
        consecutive replays are enqueued back to back instead of hopping null stream -> private stream -> null stream every step (two
        cross-stream event waits, ~10-15 us of GPU idle per step); on exit the caller's stream waits for the step's."""
        step = self

        class _Ctx:
            def __enter__(self):
                self.cur = torch.cuda.current_stream()
                step.stream.wait_stream(self.cur)
                self.ctx = torch.cuda.stream(step.stream)
                self.ctx.__enter__()
                step._in_loop = True
                step._carry_loop = getattr(step, "_carry_loop", 0) + 1
                return step

            def __exit__(self, *a):
                step._in_loop = False
                step.join()
                self.ctx.__exit__(*a)
                self.cur.wait_stream(step.stream)
                return False
        return _Ctx()

    def join(self):
        """The step's stream waits for what the last replay left running (replays inside replay_stream() defer their join)."""
        with torch.cuda.stream(self.stream):
            self.wait_texture()
        if self.defer_join:
            with torch.cuda.stream(self.stream):
                self.gF.join()

    def wait_texture(self):
        """The all-gather of the texture rows the last sharded step updated: the current stream waits for it (no-op otherwise).  Called at the
        head of the next replay and by join(): anything that reads tex_extra after a sharded step must come behind it."""
        works, self._tex_gather = getattr(self, "_tex_gather", None), None
        if works == "comm":
            torch.cuda.current_stream().wait_stream(self.comm)
            return
        for w in works or ():
            w.wait()

    def _carry_prime_if_needed(self):
        """The carried texture is current iff the LAST thing that wrote tex_extra was this step's own finish pass.  That is taken for granted
        only between consecutive replays of ONE replay_stream() loop, with the tensor's version counter unchanged and no other step having
        replayed in between; a lone replay, the first replay of a loop and anything that looks different re-assemble it (vhap_tex_carry_prime:
        what tex_prep_fwd did every step, ~60 us)."""
        tr, ns = self.tr, self.ns
        epoch = (self._carry_loop if self._in_loop else None, tr.tex_extra._version, id(self))
        if self._in_loop and self._carry_epoch == epoch and getattr(tr, "_tex_carrier", None) is self:
            return
        if self.defer_join:
            self.gF.join()                                         # (an open tail of the last replay is still writing the texture)
        ns.tex_prime()
        self._carry_epoch = epoch
        tr._tex_carrier = self

    def _replay(self):
        tr = self.tr
        if not (self.ns is not None and self.single and self.ns.carry):
            tr._tex_carrier = None                                 # (this replay's Adam update writes tex_extra behind a carrying step's back)
        if getattr(self, "_tex_gather", None) == "comm" and self._in_loop and self.gF.plan is not None and self.gF.side_base == 0 and \
                getattr(self, "_precise_ok", False) and os.environ.get("VHAP_SHARD_PRECISE_WAIT", "1") != "0":
            # only the forward plan's TEXTURE CHAIN (a root of the plan on its first side stream: assembly + pyramid, then the arena clear and
            # the step counter behind it) needs the all-gathered texture: that stream waits for the communication stream, the launch stream's
            # geometry head (per-frame stage, skinning, binning) starts under the transfer
            _lib.check(_lib.lib().vhap_plan_side_stream_wait(0, self.comm.cuda_stream), "vhap_plan_side_stream_wait")
            self._tex_gather = None
        self.wait_texture()
        if self.ns is not None and self.single:
            if self.ns.carry:
                self._carry_prime_if_needed()
            self.gF.replay(defer_join=self.defer_join and self._in_loop)
            return
        self.gF.replay()
        if self.ns is not None:
            if getattr(self, "lmk_only", False):
                self.gB.replay()
                tr.dist.all_reduce_mean_(self.ns.param_grad_flat)
                if self.ns.tex_bwd_on:
                    tr.dist.all_reduce_mean_(self.ns.g["tex_extra"])
                self.gA.replay()
                return
            if tr.dist is not None:
                # the alpha count summed over the ranks IN PLACE on the forward accumulator (ns.n_global is that word: no copy launch on the
                # chain between the photometric sum and the shading backward); nothing else reads the local count afterwards
                tr.dist.all_reduce_sum_(self.ns.n_global)
            # the gradients sit in contiguous buffers: collectives straight on them (ReduceOp.AVG), no staging copies
            if getattr(self, "tex_sharded", False):
                n0 = self.ns.albedo_tex.numel()
                if getattr(self, "tex_path", False) and self.gB.plan is not None:
                    # The pixel chain on the launch stream; the texture gradient's tile accumulation + fold are the pixel plan's open tail
                    # on a side stream.  The COMMUNICATION stream (one of ours, on a hardware queue of its own) waits for that tail and then
                    # carries the texture's whole way home: reduce-scatter -> finish + Adam of this rank's rows -> all-gather.  The
                    # collectives are SYNCHRONOUS calls with that stream current: c10d issues them on it (builds that keep a stream of
                    # their own hand over from / to it).  The launch stream runs the geometry plan, the small all-reduce and the Adam
                    # update of everything else meanwhile, and meets the texture again at the head of the next step (wait_texture).
                    self.gB.replay(defer_join=True)
                    cur = torch.cuda.current_stream()
                    folded = torch.cuda.Event()
                    with torch.cuda.stream(self.comm):
                        self.gB.join()                                     # the gradient pyramid (tex_first: not yet folded); and the plan's other side chain (lights gradient, delta clear)
                        folded.record()                                    # (tex_first: the tile accumulation is done -- what the geometry plan waits for)
                        if self.ns.fold_outside:
                            self.ns.tex_fold()
                        tr.dist.reduce_scatter_mean(self.ns.g["d_tex"][:n0], self.tex_strip.view(-1), async_op=False)
                        self.gAt.replay()
                        tr.dist.all_gather_rows(tr.tex_extra.detach(), self.tex_row0, self.tex_rows, async_op=False)
                    if self.tex_first:
                        cur.wait_event(folded)                             # geometry under the reduce-scatter, not beside the tile accumulation
                    self.gB2.replay()
                    tr.dist.all_reduce_mean_(self.ns.param_grad_flat)
                    if not self.tex_first:
                        cur.wait_event(folded)
                    self.gA.replay()
                    self._tex_gather = "comm"                              # (the next replay's launch stream waits for the all-gather: wait_texture)
                    return
                else:
                    self.gB.replay()                                       # pixel chain + the complete texture gradient
                    if self.ns.fold_outside:
                        self.ns.tex_fold()
                    work = tr.dist.reduce_scatter_mean(self.ns.g["d_tex"][:n0], self.tex_strip.view(-1), async_op=True)   # level 0, pyramid folded in
                    self.gB2.replay()                                      # geometry chain: runs under the texture collective
                tr.dist.all_reduce_mean_(self.ns.param_grad_flat)
                work.wait()
                if getattr(self, "tex_path", False):                       # (captured as a plan of its own, replayed here when the executor is not in use)
                    self.gAt.replay()
                self.gA.replay()                                           # this rank's rows of the texture + every other parameter
                # the updated rows travel while the host comes round to the next step (whose forward waits: wait_texture)
                self._tex_gather = tr.dist.all_gather_rows(tr.tex_extra.detach(), self.tex_row0, self.tex_rows, async_op=True)
                return
            self.gB.replay()                                               # pixel chain + the complete texture gradient
            work = tr.dist.all_reduce_mean_(self.ns.g["tex_extra"], async_op=True) if self.ns.tex_bwd_on else None
            self.gB2.replay()                                              # geometry chain: runs under the texture collective
            tr.dist.all_reduce_mean_(self.ns.param_grad_flat)
            if work is not None:
                work.wait()
            self.gA.replay()
            return
        if self.N is not None:
            n = self.N
            world = 1
            if tr.dist is not None:
                world = tr.dist.world_size
                n = tr.dist.all_reduce_sum(n)
            self.inv_n.copy_(world / (3.0 * n))
        self.gB.replay()
        if tr.dist is not None:
            tr.dist.average_gradients(self.params)
        self.gA.replay()
