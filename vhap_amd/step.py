"""NativeStep: one photometric fit step (vhap/model/tracker.py:1418-1462 -> compute_energy :692-750 -> backward) as a
fixed sequence of ~60 calls into libvhap_hip.so, chained by hand instead of by torch autograd.

The autograd formulation (vhap_amd.tracker.FlameTracker._compute_energy_native) spends a quarter of the captured step in
glue: ~70 launches of a few microseconds each that zero-fill gradient buffers, add gradient contributions, multiply by scalar
weights, concatenate and sum the energy terms and copy gradients.  Here every buffer is allocated once, all accumulators live in
two arenas cleared by ONE launch each, the kernels accumulate straight into the parameters' .grad storage, and the energy is
assembled in the epilogue of the photometric sum (one GPU; under frame sharding by two one-thread kernels, vhap_energy_finalize /
vhap_energy_total, around the all-reduce of the pixel count).  Same kernels, same arithmetic, same results
(tests/test_native_gpu.py::test_native_step_matches_autograd_step); every stage kind and model option of the pipeline, dynamic
offsets included.

The call sequence is laid out over up to three streams (main chain = the dependency chain of the algorithm; the texture chain and a
chain of small latency-bound launches beside it: _side / _flush) and is what vhap_amd.tracker.GraphedStep records once and the
library's plan executor replays (csrc/plan.hip); torch only provides memory, streams and the collectives.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import native as NV
from .config import PhotometricStageConfig
from .native import _n_gather
from .ops import _hook, _p, _stream

PRE = _lib.CALL_ACC_PREZEROED        # per-call flag (ABI 2): this call's small accumulators were cleared by the arena clear

FUSE_TEX_ADAM = True     # the texture's Adam update inside the gradient-finishing pass (tools flip it to time the two-pass form)

LOG_NAMES = ("lmk", "photo", "smooth_pose", "reg_joint", "smooth_joint", "reg_expr", "smooth_expr", "reg_shape", "reg_tex_tv",
             "reg_tex_res_clusters", "reg_diffuse", "reg_offset_lap", "reg_offset", "reg_offset_rigid", "rest", "total", "reg_offset_dynamic",
             "reg_tex_pca")


def _chk(rc, what):
    _lib.check(rc, what)


def _zero(t):
    """t.zero_() as a RECORDED call: the host framework's fill becomes a node of the captured step like any other, and the decisions taken
    on the buffers a plan's nodes touch (deferred join, the sharded step's precise wait) know what it writes instead of treating it as
    UNKNOWN."""
    if _lib.ACCESS is not None:
        _lib.ACCESS.touch(t)
    t.zero_()
    if _lib.ACCESS is not None:
        _lib.ACCESS.close("torch.Tensor.zero_")


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class NativeStep:
    @staticmethod
    def supported(tracker, stage):
        """Every stage kind of the pipeline: photometric stages (monocular: focal length trained; calibrated multi-view: per-view K / RT from
        the sample, tracker.py:141-157) and the landmark-only stages (lmk_init_*, lmk_*_tracking: no pixel chain at all)."""
        cfg = tracker.cfg
        photometric = isinstance(cfg.pipeline[stage], PhotometricStageConfig) and cfg.w.photo is not None
        return tracker._native_ok(stage, dynamic_offset_ok=True, tex_pca_ok=True) and \
            (not photometric or cfg.render.background_train in ("target", "white", "black"))

    def __init__(self, tracker, sample, stage):
        tr = self.tr = tracker
        self.stage = stage
        cfg, w = tr.cfg, tr.cfg.w
        dev = tr.device
        L = self.L = _lib.lib()
        nm = self.nm = tr._native_models()
        fl = tr.flame
        fb = self.fb = fl._fb
        self.fm, self.lm, self.om = nm["frame"], nm["lmk"], nm["off"]
        self.weights = tr._frame_weights(stage)
        ts = sample["timestep_index"]
        self.ts = (ts if torch.is_tensor(ts) else torch.as_tensor(np.asarray(ts), device=dev)).long().contiguous()
        self.rgb = sample["rgb"].contiguous()                        # [B,3,H,W] image space (static: new batches are copied in)
        self.photometric = isinstance(cfg.pipeline[stage], PhotometricStageConfig) and w.photo is not None
        self.calibrated = bool(tr.calibrated)
        # monocular: the camera's forward / backward ride in the per-frame launches (VHAP_CAMERA_FUSED=0: as launches of their own, for A/B)
        self.cam_fused = (not self.calibrated) and os.environ.get("VHAP_CAMERA_FUSED", "1") != "0"
        self.has_offset = tr.static_offset is not None
        # `use_dynamic_offset` (base.py:69; tracker.py:213-235, 552-600): the kernels read ONE offset row per frame of the batch,
        # off_b = static_offset + dynamic_offset[timesteps] (VHAP_CALL_OFFSET_PER_FRAME), and the offset regularisers run per frame
        self.dyn = getattr(tr, "dynamic_offset", None) is not None
        self.off_flag = _lib.CALL_OFFSET_PER_FRAME if self.dyn else 0
        if self.calibrated:                                          # static sample tensors: read by every forward (new batches are copied in)
            self.K_in, self.RT_in = sample["intrinsic"], sample["extrinsic"]
        self.lmk2d = sample["lmk2d"].float().contiguous()
        B = self.B = int(self.ts.shape[0])
        H, W = self.H, self.W = tr.image_size
        V = self.V = fb.V
        J = self.J = self.fm.J
        N = self.N = int(tr.expr.shape[0])
        NS, NE = int(tr.shape.shape[0]), int(tr.expr.shape[1])
        self.NS, self.NE = NS, NE
        T = self.T = int(tr.tex_extra.shape[-1])
        Bp = self.Bp = (B + 15) // 16 * 16
        o = tr.opt_dict
        f32 = dict(dtype=torch.float32, device=dev)
        # VHAP_POISON=1 (debugging): every buffer that is allocated uninitialised starts as NaN / 0x7f7f7f7f instead of whatever the allocator
        # hands out -- a read-before-write shows up at once instead of as a first-run-vs-later-run difference
        poison = os.environ.get("VHAP_POISON", "0") == "1"
        E = (lambda *s: torch.full(tuple(int(x) for x in (s[0] if len(s) == 1 and isinstance(s[0], (tuple, list)) else s)), float("nan"), **f32)) if poison \
            else (lambda *s: torch.empty(*s, **f32))
        Ei = (lambda n, dt: torch.full((int(n),) if not isinstance(n, tuple) else n, 0x7f, dtype=torch.uint8, device=dev).view(dt) if dt != torch.uint8
              else torch.full(n if isinstance(n, tuple) else (int(n),), 0x7f, dtype=dt, device=dev)) if poison else \
            (lambda n, dt: torch.empty(n, dtype=dt, device=dev))
        # ---- static tables ----
        mesh = tr.render._mesh(fl.faces)
        self.tri, self.opp, self.csr = mesh["tri"], mesh["opp"].int().contiguous(), mesh["csr"]
        self.F = int(self.tri.shape[0])
        self.tri_uv = tr.render._tri32(fl.textures_idx)
        self.uv = tr._verts_uv_flipped
        fid, vid = tr._regions(stage) if self.photometric else (None, None)
        self.face_mask = tr.render._u8_mask(fid, self.F) if fid is not None else None
        self.vert_mask = tr.render._u8_mask(vid, V) if vid is not None else None
        self.painted = tr.flame_tex_painted()[0].contiguous()
        if self.painted.shape[-1] != T:
            self.painted = torch.nn.functional.interpolate(self.painted[None], (T, T), mode="bilinear")[0].contiguous()
        tex_on = bool(o["texture"])
        # the FLAME PCA texture model (tex_painted = False; flame.py:665-688): `painted` is re-assembled from tex_pca at the head of every
        # step (vhap_tex_pca_fwd) and the code gets its gradient from d(base texture) (vhap_tex_pca_bwd)
        self.pca = tr.flame_tex_pca
        if self.pca is not None:
            if int(self.pca.tex_size) != T:
                raise NotImplementedError("the PCA texture model at a resolution other than tex_resolution")
            S, n = self.pca.src_size, int(tr.tex_pca.shape[0])
            self.pca_mean = self.pca.texture_mean.reshape(-1).contiguous()
            self.pca_basis = self.pca.texture_basis[0].contiguous()              # [S*S*3, n]
            self.pca_src, self.pca_work = E(S * S * 3), E(S * S * 3)
            self.pca_train = tex_on
            self.pca_scale = float((w.reg_tex_pca if tex_on else None) or 0.0) / n
        self.tex_scales = (float((tr._w_tv() if tex_on else None) or 0.0) / (3.0 * T * (T - 1)),
                           float((w.reg_tex_res_clusters if tex_on else None) or 0.0) / (3.0 * T * T))
        off_on = bool(o["static_offset"] or o["dynamic_offset"])
        nb = B if self.dyn else 1                                  # (per-frame offsets: the regularisers' means run over the frames too)
        self.off_scales = (float((w.reg_offset_lap if off_on else None) or 0.0) / (V * nb),
                           float((w.reg_offset if off_on else None) or 0.0) / (3 * V * nb),
                           float((w.reg_offset_rigid if off_on else None) or 0.0) / (3.0 * nb))
        self.dyn_scale = float((w.reg_offset_dynamic if (self.dyn and o["dynamic_offset"]) else None) or 0.0) / (3.0 * V * B)
        self.w_lmk = float(w.landmark or 0.0)
        self.want_reg = bool(o["lights"]) and w.reg_diffuse is not None
        self.w_reg = float(w.reg_diffuse or 0.0) if self.want_reg else 0.0
        self.w_photo = float(w.photo or 0.0)
        disable_jaw = not w.always_enable_jawline_landmarks and cfg.pipeline[stage]["disable_jawline_landmarks"]
        self.lmk_cfg = (17, 68, 0, 0, 1.0) if disable_jaw else (0, 68, 27, 36, 10.0)
        bg = cfg.render.background_train
        self.bg_col = None if bg == "target" else (ctypes.c_float * 3)(*([1.0, 1.0, 1.0] if bg == "white" else [0.0, 0.0, 0.0]))
        self.rate_fg, self.rate_bg = tr.render.disturb_rate_fg, tr.render.disturb_rate_bg
        self.disturb_on = bool(self.rate_fg or self.rate_bg) and self.photometric
        if self.disturb_on:
            r = tr.render
            if r._rng_state is None or r._rng_state.device != self.rgb.device:
                r._rng_state = torch.randint(0, 2 ** 31 - 1, (1,), device=dev).to(torch.int32)
            if not hasattr(r, "_fid2cid_i32") or r._fid2cid_i32.device != self.rgb.device:
                r._fid2cid_i32 = r.fid2cid.int().contiguous()
            self.fid2cid, self.ncl, self.rng = r._fid2cid_i32, r._ncl, r._rng_state
            if not hasattr(r, "_bg_cluster_is_background_only"):        # cluster 0 = the background and nothing else (fid 0 is "no triangle"): the
                # disturbance's list of covered pixels (cluster != 0), which the shading backward walks, then misses no triangle.  One host read per renderer.
                r._bg_cluster_is_background_only = bool(int(r._fid2cid_i32[0]) == 0 and (r._fid2cid_i32.numel() < 2 or int(r._fid2cid_i32[1:].min()) >= 1))
        self.K1, self.K0 = nm["K1"], nm["K0"]
        self.focal_scale = float(max(H, W))
        self.RT = E(B, 3, 4) if self.calibrated else tr.RT[None, :3, :].contiguous()
        self.sh_const = tr.render.sh_const.contiguous()
        # ---- forward buffers ----
        self.coef, self.A, self.transl, self.Jrest = E(Bp, fb.Kp), E(B, J, 12), E(B, 3), E(B, J * 3)
        self.verts, self.v_shaped, self.v_posed = E(B, V, 3), E(B, V, 3), E(B, V, 3)
        self.K, self.mvp = E(B if self.calibrated else 1, 4), E(B, 4, 4)
        self.tex_fwd_on = self.photometric or tex_on                  # the texture is assembled only when something reads it
        if self.tex_fwd_on:
            self.albedo_tex = E(1, T, T, 3)
            self.mips = E(L.vhap_texture_mip_floats(1, T, T, 3))
        else:
            self.albedo_tex, self.mips = E(0), E(0)
        # deferred shading (default): the rasteriser samples the texture and shades in registers; normal / rast_db / albedo images do not exist.
        # VHAP_DEFERRED=0: the separate passes (G-buffer, texture, shading, out-of-place antialiasing, dense gradient images) -- the
        # formulation the deferred kernels are tested against (tests/test_deferred_gpu.py) and whose raster kernel IS the RI-fwd op that
        # bench.py's roofline line is defined on
        self.deferred = self.photometric and os.environ.get("VHAP_DEFERRED", "1") != "0"
        self.tb_ids = False
        # one GPU: energy assembly + upstream gradient in the epilogue of the photometric sum (under sharding the pixel count is all-reduced
        # between the passes, so the two glue launches stay)
        self.energy_fused = self.deferred and (tracker.dist is None or not tracker.dist.sharded)
        # one GPU, in-place antialiasing: the colour part of the antialias backward is computed UNSCALED by extra workgroups of the photometric
        # sum's launch (it needs the final image, not the sum), the shading backward multiplies it by the upstream gradient, and the position
        # part runs on the side chain -- no antialias kernel between the photometric sum and the shading backward (20 us + a hand-over)
        self.aa_early_bwd = False
        self.delta_flag = 0
        # antialiasing in place + photometric gradient on the fly: no copy of the image, no dense gradient images (d_rgba_aa / d_color)
        self.aa_inplace = self.deferred
        # the photometric sum's launch assembles the energy and does the antialias colour job whenever the step is deferred; under frame
        # sharding its total / upstream gradient are provisional (local pixel count) and vhap_energy_total_bound redoes them behind the
        # count's all-reduce -- what matters there is that no energy_finalize launch and no antialias backward (22 us) sit on the chain
        # between the sum and the shading backward (round 6: 83 us of glue in the sharded step, 6 us in the one-plan step)
        self.photo_total = self.deferred
        self.aa_early_bwd = self.photo_total and self.aa_inplace
        self.delta_flag = _lib.CALL_DELTA_UNSCALED if self.aa_early_bwd else 0
        self._delta_dirty = False
        if self.photometric:
            self.clip, self.vn, self.vn_inv = E(B, V, 4), E(B, V, 3), E(B, V)
            self.rast, self.texc, self.texd = E(B, H, W, 4), E(B, H, W, 2), E(B, H, W, 4)
            if not self.deferred:
                self.db, self.normal, self.albedo_px = E(B, H, W, 4), E(B, H, W, 3), E(B, H, W, 3)
            self.rgba = E(B, H, W, 4)
            if not self.aa_inplace:
                self.rgba_aa = E(B, H, W, 4)
            self.cov_list, self.n_bg, self._cov_list_fresh = None, None, False
            if self.disturb_on:
                self.keep = E(B, H, W)                               # (the disturbance itself is in place: pools of copies, csrc/disturb.hip)
                self.dist_ws = Ei(int(L.vhap_disturb_workspace_ints(B, H, W)) * 4, torch.int32) if poison else \
                    torch.empty(L.vhap_disturb_workspace_ints(B, H, W), dtype=torch.int32, device=dev)
                self.cid = Ei((B, H, W), torch.uint8)
                if self.deferred and tr.render._bg_cluster_is_background_only and os.environ.get("VHAP_SHADE_LIST", "1") != "0":
                    # the list of covered pixels the disturbance's counting sort leaves (+ the number of background pixels), for the shading backward
                    self.cov_list, self.n_bg = torch.empty(B * H * W, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
            n_aa = int((L.vhap_antialias_inplace_work_ints if self.aa_inplace else L.vhap_antialias_work_ints)(B, H, W, self.F))
            self.aa_work = Ei(n_aa * 4, torch.int32) if poison else torch.empty(n_aa, dtype=torch.int32, device=dev)
            self.ws, self.ws_bytes, self.ws_cap, _ = tr.render.glctx.acquire(B, self.F, H, W, self.rgb.device)
            # one-launch binning available (raster.hip: LDS_BIN_LIMIT bins, MAX_FRAG x 1024 triangles): binning and rasterisation can be split
            nfrag = (self.F + 1023) // 1024
            self.bin_split = ((W + 7) // 8) * ((H + 7) // 8) <= 16384 and nfrag <= 32 and self.ws_cap // (B * nfrag) >= 1
        # forward accumulators: frame terms [0:6], landmark [6], texture terms [7:9], shade stats [12:16], photo [16:19], offset terms [20:24]
        self.accF = torch.zeros(32, **f32)
        self.log = torch.zeros(len(LOG_NAMES), **f32)
        if self.dyn:
            self.off_b = E(B, V, 3)
        self.n_global = self.accF[17:18]                      # replaced by the all-reduced count under frame sharding
        # ---- backward: one arena for everything that is accumulated into ----
        params = {"shape": tr.shape, "expr": tr.expr, "rotation": tr.rotation, "translation": tr.translation, "neck_pose": tr.neck_pose,
                  "jaw_pose": tr.jaw_pose, "eyes_pose": tr.eyes_pose, "lights": tr.lights}
        if self.pca is not None:
            params["tex_pca"] = tr.tex_pca
        if self.has_offset:
            params["static_offset"] = tr.static_offset
        if self.dyn:
            params["dynamic_offset"] = tr.dynamic_offset
        if not self.calibrated:
            params["focal_length"] = tr.focal_length
        sizes = {k: p.numel() for k, p in params.items()}
        extra = {"d_verts": B * V * 3, "d_A": B * J * 12, "d_t": B * 3, "d_coef": Bp * fb.Kp}
        if self.dyn:
            extra["d_off_b"] = B * V * 3                           # per-frame offset gradient besides the skinning part: joint regression + regularisers
        if self.photometric:
            extra.update({"d_clip": B * V * 4, "d_vn": B * V * 3, "d_tex": self.albedo_tex.numel() + self.mips.numel()})
            if self.deferred:
                extra["def_work"] = int(L.vhap_deferred_shade_bwd_work_floats(B, H, W))    # zero on entry: lives in the arena
        al = lambda n: (n + 63) // 64 * 64
        total = sum(al(n) for n in sizes.values()) + sum(al(n) for n in extra.values())
        self.arena = torch.zeros(total, **f32)
        off = 0
        self.g = {}
        for k, p in params.items():
            self.g[k] = self.arena[off:off + sizes[k]].view(p.shape)
            off += al(sizes[k])
        self.param_grad_flat = self.arena[:off]                      # every parameter gradient but the texture's: one all-reduce
        for k, n in extra.items():
            self.g[k] = self.arena[off:off + n]
            off += al(n)
        self.params = dict(params)
        self.tex_bwd_on = tex_on and (self.photometric or any(self.tex_scales))
        if self.tex_bwd_on:
            self.g["tex_extra"] = torch.zeros_like(tr.tex_extra)      # overwritten by tex_prep_bwd, never accumulated
            self.params["tex_extra"] = tr.tex_extra
            if self.pca is not None:
                self.d_base = torch.zeros_like(tr.tex_extra)          # d(base texture): overwritten by the same pass
        for k, p in self.params.items():                              # the optimiser and the gradient all-reduce see these
            p.grad = self.g[k]
        # scratch that is overwritten
        if self.photometric:
            if self.aa_inplace:
                self.d_delta = torch.zeros(B, H, W, 4, **f32)       # zero except, transiently, at the pixels of the antialias pair list
            else:
                self.d_rgba_aa, self.d_color = E(B, H, W, 4), E(B, H, W, 4)
            self.d_albedo = E(B, H, W, 3)
            self.d_normal, self.d_texc, self.d_texd = E(B, H, W, 3), E(B, H, W, 2), E(B, H, W, 4)
            self.texbin_work = torch.zeros(self.L.vhap_texture_grad_binned_work_bytes(B, H, W), dtype=torch.uint8, device=dev)
            # the uv tile of every pixel is handed to the sorting passes of the texture gradient as 2 B/px (written by the rasteriser)
            # instead of uv + d_albedo (20 B/px, twice)
            self.tb_ids = self.deferred and self.tex_bwd_on and T <= 2048
            if self.tb_ids:
                self.tile_ids = Ei(B * H * W * 2, torch.int16).view(B, H, W) if poison else torch.empty(B, H, W, dtype=torch.int16, device=dev)
            self.vn_scratch = E(B, V, 3)
            if self.deferred:
                self.def_work = self.g["def_work"]
        self.g_posed, self.g_shaped = E(B, V, 3), E(B, V, 3)
        self.d_mvp, self.d_K, self.d_sum, self.gmax_bound = E(B, 16), E(B, 4), E(1), E(1)
        self.ones = torch.ones(8, **f32)
        self.photo_work = torch.zeros(1024, **f32)               # VHAP_PHOTO_WORK_FLOATS
        # two independent chains per pass run on two streams (two branches of the captured graph): the bandwidth / atomics bound texture
        # work next to the latency-bound geometry chain of small launches
        self.overlap = os.environ.get("VHAP_STEP_OVERLAP", "1") != "0"      # (0: one chain -- what tools/step_pmc.py attributes counters on)
        self.one_graph = False        # GraphedStep, one GPU: forward + backward + Adam are ONE captured graph -- the forward accumulators are then cleared
                                      # at the END of the step (off the critical path; the step's first kernels become roots of the graph) and the
                                      # texture-gradient sort is not joined before the backward needs it
        self._acc_clean = False
        self.injected = None          # dict(w_fg, w_bg, idx): random numbers of the colour disturbance handed in (parity tests) instead of drawn in-kernel
        self.step_optimizer = None    # a HipAdam whose WHOLE update is issued inside forward()/backward() (GraphedStep, one GPU): counter advanced
                                      # at the head of the step, texture update behind its gradient, the rest before the final join
        self.sort_in_backward = False  # (GraphedStep's sharded texture path: the uv-tile sort is forked by the pixel plan, not by the forward plan)
        self.fold_outside = False     # (with split_tex, GraphedStep's texture path: tex_fold() is issued by the caller, not inside the pixel plan)
        self.split_tex = False        # True: stop at the gradient pyramid, the caller runs tex_finish() later (pyramid-level exchange under sharding)
        self.feed = None              # dict set by GraphedStep.enable_feed(): the step begins by gathering its batch from an uploaded table (vhap_batch_feed)
        self.carry = False            # enable_carry(): the texture is CARRIED from step to step (the finish + Adam pass writes the next step's albedo)
        self.tex_halo = None
        # streams of the library's own (never torch's pool: see _lib.private_stream), shared by every step of this thread
        self.side = _lib.private_stream("side", dev)
        self.side2 = _lib.private_stream("side2", dev)
        self._pending = []
        self.c_lmk = torch.full((1,), self.w_lmk, **f32)
        self.c_reg = torch.full((1,), self.w_reg, **f32)

    # ------------------------------------------------------------------------------------------------
    def _fork(self):
        if self.overlap:
            self.side.wait_stream(torch.cuda.current_stream())

    def _join(self):
        self._flush()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.side)

    def _branch(self):
        return torch.cuda.stream(self.side) if self.overlap else _Null()

    def _side(self, fn, stream=None):
        """Run fn() on the side branch, forked at the CURRENT point of this stream.  The fork is only marked (an event); fn is issued by the
        next _flush(), i.e. AFTER the main chain's next kernel was captured: the plan executor (csrc/plan.hip) keeps a node's
        first-captured successor on the node's own stream and forks the others onto side streams (a hand-over costs ~10 us)."""
        stream = self.side if stream is None else stream
        if not self.overlap:
            return fn()
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((ev, stream, fn))

    def _flush(self):
        for ev, stream, fn in self._pending:
            stream.wait_event(ev)
            with torch.cuda.stream(stream):
                fn()
        self._pending = []

    def enable_carry(self, keep_grad=False):
        """THE CARRIED TEXTURE (GraphedStep, one GPU, the texture's Adam update fused into the gradient-finishing pass): that pass also
        rewrites `albedo_tex` in place with painted + the UPDATED tex_extra, writes level 1 of the pyramid and sums this step's TV /
        residual energies into the log (vhap_tex_finish_carry) -- the forward no longer assembles the texture (tracker.py:237-258 does it at
        the head of every step: 49 us and 198 MB beside the binning launch and the rasteriser), it only rebuilds pyramid levels >= 2.
        The caller must run tex_prime() before the first step of a loop and whenever anything else has written tex_extra since
        (GraphedStep._replay does).  -> True when the step qualifies."""
        L, T = self.L, self.T
        ok = self.photometric and self.overlap and self.tex_bwd_on and self.pca is None and T % 64 == 0 and \
            self.mips.numel() > 0 and _n_gather(T) >= L.vhap_texture_num_levels(T, T) and self.step_optimizer is not None and FUSE_TEX_ADAM and \
            hasattr(self.step_optimizer, "fused_update_args") and self.step_optimizer.fused_update_args(self.tr.tex_extra) is not None
        if ok:
            self.tex_halo = torch.empty(int(L.vhap_tex_carry_halo_floats(T)), dtype=torch.float32, device=self.tr.device)
            # TV / residual / (unused: PCA) energies of the carried texture: accumulated by the finish pass (or tex_prime) for the NEXT
            # step, read and cleared by that step's energy assembly -- outside the forward accumulators, which the step clears at its end
            self.carry_terms = torch.zeros(4, dtype=torch.float32, device=self.tr.device)
            self.carry_keep_grad = bool(keep_grad)
            if not keep_grad:
                # the gradient is consumed inside the pass and not written (50 MB per step that nothing reads): rather no .grad than a stale
                # one (VHAP_TEX_KEEP_GRAD=1 keeps it)
                self.tr.tex_extra.grad = None
        self.carry = bool(ok)
        return self.carry

    def tex_prime(self):
        """albedo, pyramid level 1 and the halo copies of the carried texture from painted + tex_extra, from scratch (current stream)."""
        _chk(self.L.vhap_tex_carry_prime(_p(self.painted), _p(self.tr.tex_extra), _p(self.nm["res_mask"]), self.T, *self.tex_scales,
                                         _p(self.albedo_tex), _p(self.mips), _p(self.tex_halo), _p(self.carry_terms), _stream()),
             "vhap_tex_carry_prime")
        # (the captured step builds the pyramid from level 3 up -- the finish pass hands it levels 1 and 2 --: level 2 of a primed texture here)
        _chk(self.L.vhap_texture_mip_build_from(_p(self.albedo_tex), 1, self.T, self.T, 3, _p(self.mips), 2, _stream()), "vhap_texture_mip_build_from")

    def _feed_batch(self):
        """The step's own batch hand-over (vhap_batch_feed): the next batch of the uploaded table -> timesteps, frame indices, landmarks
        (+ per-view cameras) in the static buffers the kernels read; the frame ingest follows on the texture branch (its output, the target
        image, is first read by the rasteriser, which waits for that branch)."""
        f, L, tr = self.feed, self.L, self.tr
        rows = [(tr.dataset["lmk2d"], self.lmk2d)]
        if self.calibrated:
            rows += [(tr.dataset["intrinsic"], self.K_in), (tr.dataset["extrinsic"], self.RT_in)]
        args = []
        for src, dst in rows:
            assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.float32
            args += [_p(src), _p(dst), int(src[0].numel())]
        args += [0, 0, 0] * (3 - len(rows))
        _chk(L.vhap_batch_feed(_p(f["frames"]), _p(f["ts"]), _p(f["cursor"]), self.B, f["capacity"], int(tr.dataset["lmk2d"].shape[0]), *args,
                               _p(f["frame_index"]), _p(self.ts), _stream()), "vhap_batch_feed")

        def ingest():
            tr.frames.batch(f["frame_index"], out=self.rgb)
        self._side(ingest)

    def _tex_forward(self, ready=None):
        """texture assembly + pyramid + the offset regularisers: independent of the geometry chain until the texture is sampled.
        `ready()` is called behind the pyramid -- what the rasteriser waits for -- ahead of the offset regularisers."""
        L, tr, T, acc = self.L, self.tr, self.T, self.accF
        st = _stream()
        if self.pca is not None and self.tex_fwd_on:
            _chk(L.vhap_tex_pca_fwd(_p(self.pca_mean), _p(self.pca_basis), _p(tr.tex_pca), int(tr.tex_pca.shape[0]), self.pca.src_size, T,
                                    self.pca_scale, _p(self.pca_src), _p(self.painted), _p(acc[9:10]), st), "vhap_tex_pca_fwd")
        if self.carry:
            # the carried texture: albedo, pyramid level 1 and the TV / residual energies (carry_terms) were written by the previous step's
            # finish + Adam pass (or tex_prime())
            _chk(L.vhap_texture_mip_build_from(_p(self.albedo_tex), 1, T, T, 3, _p(self.mips), 3, st), "vhap_texture_mip_build_from")   # (levels 1, 2: the finish pass)
            if ready is not None:
                ready()
                ready = None
            # ... but for the TV pairs across the finish pass's ownership tiles: from its halo copies, BEHIND the pyramid (the rasteriser
            # waits for that; only the energy assembly reads the terms) and ahead of this step's advance of the counter (side_work, issued
            # behind this branch on the same stream), whose parity says which halo copy is the new one (after tex_prime() the two are the same)
            _chk(L.vhap_tex_carry_border(T, self.tex_scales[0], _p(self.step_optimizer.step_count), _p(self.tex_halo), _p(self.carry_terms),
                                         _lib.CALL_ADAM_STEP_ADVANCED, st), "vhap_tex_carry_border")
        elif self.tex_fwd_on and self.photometric and T % 2 == 0 and self.mips.numel() > 0:
            # texture assembly + TV / residual energies + level 1 of the pyramid in one pass; the rest of the pyramid four levels per launch
            _chk(L.vhap_tex_prep_mip1_fwd(_p(self.painted), _p(tr.tex_extra), _p(self.nm["res_mask"]), T, *self.tex_scales, _p(self.albedo_tex),
                                          _p(self.mips), _p(acc[7:10]), PRE, st), "vhap_tex_prep_mip1_fwd")
            _chk(L.vhap_texture_mip_build_from(_p(self.albedo_tex), 1, T, T, 3, _p(self.mips), 2, st), "vhap_texture_mip_build_from")
        elif self.tex_fwd_on:
            _chk(L.vhap_tex_prep_fwd(_p(self.painted), _p(tr.tex_extra), _p(self.nm["res_mask"]), T, *self.tex_scales, _p(self.albedo_tex),
                                     _p(acc[7:10]), PRE, st), "vhap_tex_prep_fwd")
            if self.photometric:
                _chk(L.vhap_texture_mip_build(_p(self.albedo_tex), 1, T, T, 3, _p(self.mips), st), "vhap_texture_mip_build")
        if ready is not None:
            ready()
        if (self.has_offset or self.dyn) and any(self.off_scales):
            om = self.om
            # the regularised offset (tracker.py:552-559): static_offset, or -- dynamic offsets -- one combined row per frame (means over the frames
            # too: the scales carry 1 / B)
            if self.dyn:                                           # one launch for the B per-frame rows
                _chk(L.vhap_offset_reg_fwd_batch(_p(self.off_b), _p(om.ptr), _p(om.col), _p(om.val), _p(om.w_lap), _p(om.w_abs), _p(om.rptr),
                                                 _p(om.ridx), self.B, om.V, om.nreg, *self.off_scales, _p(acc[20:24]), PRE, st),
                     "vhap_offset_reg_fwd_batch")
            else:
                _chk(L.vhap_offset_reg_fwd(_p(tr.static_offset), _p(om.ptr), _p(om.col), _p(om.val), _p(om.w_lap), _p(om.w_abs), _p(om.rptr),
                                           _p(om.ridx), om.V, om.nreg, *self.off_scales, _p(acc[20:24]), PRE, st), "vhap_offset_reg_fwd")
        if self.dyn and self.dyn_scale:
            _chk(L.vhap_offset_dynamic_reg(_p(tr.dynamic_offset), _p(self.ts), self.B, self.N, self.V, self.dyn_scale, 0, _p(acc[23:24]), 0, st),
                 "vhap_offset_dynamic_reg")

    def _camera_forward(self):
        L, tr, B, H, W = self.L, self.tr, self.B, self.H, self.W
        if self.calibrated:
            # per-view intrinsics / extrinsics of the sample (tracker.py:141-147): K [B,3,3] or [B,4] -> (fx, fy, cx, cy); RT [B,3|4,4].
            # The sample tensors are static (new batches are copied INTO them): whatever already has the kernel's layout is read in place
            # -- a same-layout copy_ would be a memcpy node in the captured step, which the plan executor does not replay
            K, RT = self.K_in, self.RT_in
            if K.shape[-2:] == (3, 3):
                torch.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], dim=-1, out=self.K)
                K = self.K
            elif not (K.is_contiguous() and K.dtype == torch.float32):
                K = self.K.copy_(K)
            if RT.shape[-2] == 3 and RT.is_contiguous() and RT.dtype == torch.float32:
                self.RT = RT
            else:
                self.RT.copy_(RT[:, :3, :])                       # (a strided copy: a kernel)
            _chk(L.vhap_camera_fwd(_p(K), _p(self.RT), B, 1, 1, H, W, 0.1, 10.0, _p(self.mvp), _stream()), "vhap_camera_fwd")
            return
        # monocular: K = (f, f, cx, cy), f = focal * max(h, w): built inside the kernel
        _chk(L.vhap_camera_focal_fwd(_p(tr.focal_length), self.focal_scale, 0.5 * W, 0.5 * H, _p(self.RT), B, 0, H, W, 0.1, 10.0, _p(self.mvp),
                                     _stream()), "vhap_camera_focal_fwd")

    def _landmark_forward(self):
        L, B, H, W, V = self.L, self.B, self.H, self.W, self.V
        l0, l1, b0, b1, boost = self.lmk_cfg
        _chk(L.vhap_landmark_fwd(_p(self.verts), _p(self.lm.vidx), _p(self.lm.bary), _p(self.mvp), _p(self.lmk2d), B, V, self.lm.L,
                                 self.lmk2d.shape[1], l0, l1, b0, b1, boost, H, W, 0, _p(self.accF[6:7]), PRE, _stream()), "vhap_landmark_fwd")

    def forward(self):
        """Every accumulator below comes from the arena cleared by the first launch: the calls get VHAP_CALL_ACC_PREZEROED."""
        L, tr, fb, fm = self.L, self.tr, self.fb, self.fm
        B, H, W, V, F, T, J = self.B, self.H, self.W, self.V, self.F, self.T, self.J
        st = _stream()
        acc = self.accF
        self._cov_list_fresh = False
        if not self._acc_clean:
            _zero(acc)                                                # ONE launch clears every forward accumulator
        self._acc_clean = False
        if self._delta_dirty:                                         # (eager use only: a forward whose backward never came left its antialias
            self._clear_delta()                                       # colour gradients in d_delta; its pair list is still intact here)
        if self.feed is not None:
            self._feed_batch()
        so = tr.static_offset
        if self.dyn:                                              # one offset row per frame: static_offset + dynamic_offset[timesteps]
            _chk(L.vhap_offset_combine(_p(tr.static_offset) if self.has_offset else 0, _p(tr.dynamic_offset), _p(self.ts), B, self.N, V,
                                       _p(self.off_b), st), "vhap_offset_combine")
            so = self.off_b
        self._tex_ready = None
        early_tex = self.photometric and self.deferred and self.overlap
        # the camera first, alone (one tiny workgroup per frame, ~5 us): beside the bandwidth-bound texture assembly it took 50 us, and the
        # skinning kernel behind the per-frame stage waited for it.  (The texture branch below is FORKED ahead of it -- a root of the captured
        # step: no event record between the camera and the per-frame stage, ~6 us on the main chain -- but issued behind the per-frame
        # stage's launch, see _side.)
        # (monocular: the camera rides in the per-frame launch as one more workgroup, vhap_frame_prep_fwd_camera -- no launch of its own)
        cam_fused = self.cam_fused
        if not early_tex and not cam_fused:
            self._camera_forward()
        if early_tex:
            # deferred shading: the rasteriser itself samples the texture, so the texture assembly + pyramid (~75 us, bandwidth-bound) heads the
            # critical path together with the geometry chain: start it at once on the side branch.  (Measured alternatives: forked after the
            # per-frame stage, the skinning kernel -- 27 MB of basis -- runs 70 us instead of 26 next to the texture assembly and the
            # rasteriser starts 25 us later; forked after the skinning, the rasteriser waits for the pyramid.)
            def tex_ready():
                self._tex_ready = torch.cuda.Event()
                self._tex_ready.record()
            self._side(lambda: self._tex_forward(ready=tex_ready))
            if not cam_fused:
                self._camera_forward()
        fp_args = (_p(self.ts), _p(tr.shape), _p(tr.expr), _p(tr.rotation), _p(tr.translation), _p(tr.neck_pose),
                   _p(tr.jaw_pose), _p(tr.eyes_pose), _p(fm.JT), _p(fm.JS), _p(fm.jreg_idx), _p(fm.jreg_w), fm.jreg_n,
                   _p(so), fm.parents, self.weights, B, self.Bp, self.N, self.NS, self.NE, J, fb.Kp, V,
                   _p(self.coef), _p(self.A), _p(self.transl), _p(self.Jrest), _p(acc), PRE | self.off_flag)
        if cam_fused:
            H, W = self.H, self.W
            _chk(L.vhap_frame_prep_fwd_camera(*fp_args, _p(tr.focal_length), self.focal_scale, 0.5 * W, 0.5 * H, _p(self.RT), 0, H, W, 0.1, 10.0,
                                              _p(self.mvp), st), "vhap_frame_prep_fwd_camera")
        else:
            _chk(L.vhap_frame_prep_fwd(*fp_args, st), "vhap_frame_prep_fwd")
        self._flush()
        if self.photometric:                                      # skinning fused with the world -> clip transform (one launch, same bits)
            _chk(L.vhap_flame_skin_clip_fwd(_p(self.coef), _p(fb.basis), _p(self.A), _p(fb.w), _p(fb.templ), _p(so), _p(self.transl), _p(self.mvp),
                                            B, V, fb.Vp, fb.K, fb.Kb, fb.Kp, _p(self.verts), _p(self.v_shaped), _p(self.v_posed), _p(self.clip),
                                            self.off_flag, st), "vhap_flame_skin_clip_fwd")
        else:
            _chk(L.vhap_flame_skin_fwd(_p(self.coef), _p(fb.basis), _p(self.A), _p(fb.w), _p(fb.templ), _p(so), _p(self.transl), B, V, fb.Vp,
                                       fb.K, fb.Kb, fb.Kp, _p(self.verts), _p(self.v_shaped), _p(self.v_posed), self.off_flag, st),
                 "vhap_flame_skin_fwd")
        if not self.photometric:
            # landmark-only stage (lmk_init_*, lmk_*_tracking): no pixel chain, a handful of latency-bound launches
            self._tex_forward()
            if self.w_lmk:
                self._landmark_forward()
            _zero(self.arena)
            self._arena_clean = True
            _chk(L.vhap_energy_finalize(_p(acc[0:6]), _p(acc[6:7]) if self.w_lmk else 0, _p(acc[7:10]), _p(acc[20:24]), 0, self.w_lmk, 0.0,
                                        B, H, W, _p(self.log), st), "vhap_energy_finalize")
            return
        # fork here, not at the top: next to the bandwidth-bound texture assembly the two latency-bound kernels above take 3x as long,
        # and they head the critical path of the forward pass; the texture branch still finishes long before the rasteriser does
        def side_work():
            if not early_tex:
                self._tex_forward()
            if self.w_lmk:                                        # needs only verts + mvp: off the rasteriser's critical path
                self._landmark_forward()
            if self.aa_inplace and self.overlap:
                # the antialiasing's silhouette flags are a property of the clip-space geometry alone: beside the rasteriser, so that only
                # the pixel-pair discovery is left for the gap between the rasteriser and the blend
                _chk(L.vhap_antialias_inplace_silhouette(_p(self.clip), _p(self.tri), _p(self.opp), B, H, W, V, F, _p(self.aa_work), _stream()),
                     "vhap_antialias_inplace_silhouette")
            _zero(self.arena)                                     # ONE launch clears every gradient accumulator of the backward
            self._arena_clean = True
            if self.step_optimizer is not None:
                self.step_optimizer.advance()
        self._side(side_work)
        if self.deferred:
            return self._forward_deferred()
        _chk(L.vhap_vnormal_fwd_saved(_p(self.verts), _p(self.csr.tri), _p(self.csr.ptr), _p(self.csr.idx), B, V, _p(self.vn), _p(self.vn_inv), st), "vhap_vnormal_fwd")
        self._flush()
        _hook("raster_interp_fwd", "begin")                       # (bench.py: HIP events around the RI-fwd pass of eagerly issued steps)
        _chk(L.vhap_raster_interp_fwd(_p(self.clip), _p(self.tri), _p(self.vn), _p(self.uv), _p(self.tri_uv), B, V, self.uv.shape[0], F, H, W,
                                      _p(self.rast), _p(self.db), _p(self.normal), _p(self.texc), _p(self.texd), _p(self.ws), self.ws_bytes,
                                      self.ws_cap, 1, st), "vhap_raster_interp_fwd")
        _hook("raster_interp_fwd", "end")
        self._join()
        _chk(L.vhap_texture_fwd(_p(self.albedo_tex), _p(self.mips), 1, T, T, 3, _p(self.texc), _p(self.texd), B, H, W, _p(self.albedo_px), st),
             "vhap_texture_fwd")
        _chk(L.vhap_shade_fwd(_p(self.normal), _p(self.albedo_px), _p(self.rast), _p(self.rgb) if self.bg_col is None else 0,
                              ctypes.cast(self.bg_col, ctypes.c_void_p) if self.bg_col is not None else 0, _p(tr.lights), _p(self.sh_const),
                              _p(self.fid2cid) if self.disturb_on else 0, self.fid2cid.numel() if self.disturb_on else 0,
                              B, H, W, _p(self.rgba), _p(acc[12:16]), _p(self.cid) if self.disturb_on else 0, PRE, st),
             "vhap_shade_fwd")
        color = self.rgba
        if self.disturb_on:
            self._disturb(st)
        self.aa_in = color
        _chk(L.vhap_antialias_fwd(_p(color), _p(self.rast), _p(self.clip), _p(self.tri), _p(self.opp), B, H, W, 4, V, F, _p(self.rgba_aa),
                                  _p(self.aa_work), st), "vhap_antialias_fwd")
        _chk(L.vhap_photo_fwd(_p(self.rgba_aa), _p(self.rgb), B, H, W, _p(acc[16:18]), PRE, st), "vhap_photo_fwd")
        _chk(L.vhap_energy_finalize(_p(acc[0:6]), _p(acc[6:7]) if self.w_lmk else 0, _p(self.carry_terms) if self.carry else _p(acc[7:10]),
                                    _p(acc[20:24]), _p(acc[12:16]) if self.want_reg else 0, self.w_lmk, self.w_reg, B, H, W, _p(self.log), st),
             "vhap_energy_finalize")
        if self.carry:
            _zero(self.carry_terms)             # consumed (the deferred step's energy assembly does it itself: VHAP_CALL_TEX_TERMS_CONSUME)

    def _forward_deferred(self):
        """binning || vertex normals -> rasterise + interpolate + texture + shade + composite in ONE kernel -> disturbance -> antialias ->
        photometric sum"""
        L, tr = self.L, self.tr
        B, H, W, V, F, T = self.B, self.H, self.W, self.V, self.F, self.T
        st = _stream()
        acc = self.accF
        cur = torch.cuda.current_stream()

        def raster(flags):
            return L.vhap_raster_shade_fwd(_p(self.clip), _p(self.tri), _p(self.vn), _p(self.uv), _p(self.tri_uv), _p(self.albedo_tex),
                                           _p(self.mips), T, T, _p(tr.lights), _p(self.sh_const), _p(self.rgb) if self.bg_col is None else 0,
                                           ctypes.cast(self.bg_col, ctypes.c_void_p) if self.bg_col is not None else 0,
                                           _p(self.fid2cid) if self.disturb_on else 0, self.fid2cid.numel() if self.disturb_on else 0,
                                           B, V, self.uv.shape[0], F, H, W, _p(self.rast), _p(self.rgba), _p(self.cid) if self.disturb_on else 0,
                                           _p(acc[12:16]), _p(self.tile_ids) if self.tb_ids else 0, _p(self.ws),
                                           self.ws_bytes, self.ws_cap, flags, st)
        _hook("raster_interp_fwd", "begin")
        # the shading statistics (diffuse maximum / variance) are ALWAYS collected: reg_diffuse reads them when the lights are trained, and the
        # fixed-point scale of the texture-gradient accumulation is derived from the measured diffuse maximum whatever the stage trains.
        # Their reduction (only the energy assembly reads it) runs beside the pixel chain instead of inside it
        stats_later = 16 if self.overlap else 0
        # EARLY STORES (VHAP_RASTER_PREFILL): the binning launch also stores every 8x8 block outside the frame's geometry box (rast zeros, the
        # background composite, cluster byte, tile id); the raster kernel's waves of those blocks leave at once.  Not with a self-feeding
        # step: its target image is written on the texture branch, which the binning launch does not wait for.
        # OFF by default: measured, the binning launch grows by more than the raster kernel shrinks (csrc/raster.hip, PrefillJob).
        prefill = 32 if (self.bin_split and self.feed is None and os.environ.get("VHAP_PREFILL", "0") == "1") else 0
        if self.bin_split and prefill:
            _chk(L.vhap_raster_bin_vnormal_prefill(_p(self.clip), _p(self.tri), _p(self.tri_uv), B, V, F, H, W, _p(self.ws), self.ws_bytes, self.ws_cap,
                                                   1, _p(self.verts), _p(self.csr.ptr), _p(self.csr.idx), _p(self.vn), _p(self.vn_inv),
                                                   _p(self.rgb) if self.bg_col is None else 0,
                                                   ctypes.cast(self.bg_col, ctypes.c_void_p) if self.bg_col is not None else 0,
                                                   _p(self.fid2cid) if self.disturb_on else 0, self.fid2cid.numel() if self.disturb_on else 0,
                                                   _p(self.rast), _p(self.rgba), _p(self.cid) if self.disturb_on else 0,
                                                   _p(self.tile_ids) if self.tb_ids else 0, st), "vhap_raster_bin_vnormal_prefill")
        elif self.bin_split:
            # binning + vertex normals in ONE launch (independent work, both inputs of the raster kernel): no fork / join -- a hand-over
            # between streams costs ~10 us each way on this critical path
            _chk(L.vhap_raster_bin_vnormal(_p(self.clip), _p(self.tri), _p(self.tri_uv), B, V, F, H, W, _p(self.ws), self.ws_bytes, self.ws_cap,
                                           1, _p(self.verts), _p(self.csr.ptr), _p(self.csr.idx), _p(self.vn), _p(self.vn_inv), st),
                 "vhap_raster_bin_vnormal")
        else:                                                     # (meshes / frames too large for the one-launch binning: binning inside the raster call)
            _chk(L.vhap_vnormal_fwd_saved(_p(self.verts), _p(self.csr.tri), _p(self.csr.ptr), _p(self.csr.idx), B, V, _p(self.vn), _p(self.vn_inv), st), "vhap_vnormal_fwd")
        self._flush()
        if self._tex_ready is not None:
            cur.wait_event(self._tex_ready)
        else:
            self._join()
        _chk(raster(((1 | 4) if self.bin_split else 1) | stats_later | prefill), "vhap_raster_shade_fwd")   # VHAP_RASTER_WS_CLEAN (| VHAP_RASTER_PREBINNED | _PREFILL)
        _hook("raster_interp_fwd", "end")
        # the antialiasing's pair discovery needs the rasteriser's output only, not the colours: beside the colour disturbance instead of
        # behind it (and ahead of the statistics reduction on the side stream: the blend waits for it, the energy assembly for the other)
        self._aa_det = None
        stats = lambda: _chk(L.vhap_raster_shade_stats(B, F, H, W, _p(self.ws), self.ws_bytes, self.ws_cap, 1, _p(acc[12:16]), _stream()),
                             "vhap_raster_shade_stats")
        if self.aa_inplace and self.overlap:
            def detect_branch():
                _chk(L.vhap_antialias_inplace_pairs(_p(self.rast), B, H, W, F, _p(self.aa_work), _stream()), "vhap_antialias_inplace_pairs")
                if stats_later:
                    # ONE hand-over back to the main chain for both: the blend waits for the pair list AND the statistics (17 us more, well
                    # inside its slack), and the photometric sum behind it needs no wait of its own (~8 us on the main chain)
                    stats()
                self._aa_det = torch.cuda.Event()
                self._aa_det.record()
            self._side(detect_branch)                             # (on the texture branch's stream: idle here)
        elif stats_later:
            self._side(stats)
        color = self.rgba
        if self.disturb_on:
            self._disturb(st)
        self._flush()
        sort_branch = None
        if self.tb_ids:
            # the counting sort of the pixels by uv tile (for the texture gradient) needs only the tile ids the rasteriser wrote and the
            # disturbance's keep mask (replaced pixels pass no gradient): beside the pixel chain instead of on the backward's critical path
            def sort_branch():
                _chk(L.vhap_texbin_sort_ids(_p(self.tile_ids), _p(self.keep) if self.disturb_on else 0, T, T, B, H, W,
                                            _p(self.texbin_work), self.texbin_work.numel(), _stream()), "vhap_texbin_sort_ids")
                if prefill and self.overlap:
                    # with early stores the NEXT step's binning launch writes tile_ids: the sort (the only reader off the main chain) must be
                    # ordered ahead of this step's last main-chain kernel, or it counts as an open tail of a deferred join (tracker.GraphedStep)
                    self._sort_done = torch.cuda.Event()
                    self._sort_done.record()
            self._sort_fn = None
            if self.sort_in_backward:
                # sharded texture path: the forward is a plan of its own that ENDS with the photometric sum -- forked here, the sort has
                # nothing of the main chain left to run beside and the executor put its four launches (45 us) ON the launch stream ahead of
                # the sum (profiles/r06_call12_step_timeline_sharded_tf1.txt); the pixel plan forks it beside the shading backward instead
                self._sort_fn = sort_branch
            elif not self.one_graph:
                self._side(sort_branch)                           # (eager: next to the rest of the forward pass)
        self.aa_in = color
        if self.aa_inplace:
            if self._aa_det is not None or self._pending:
                self._flush()                                      # (the detect branch, had no _flush() come since)
            if self._aa_det is not None:
                torch.cuda.current_stream().wait_event(self._aa_det)
                _chk(L.vhap_antialias_inplace_blend(_p(color), _p(self.rast), _p(self.clip), _p(self.tri), _p(self.opp), B, H, W, V, F,
                                                    _p(self.aa_work), st), "vhap_antialias_inplace_blend")
            else:
                _chk(L.vhap_antialias_inplace_fwd(_p(color), _p(self.rast), _p(self.clip), _p(self.tri), _p(self.opp), B, H, W, V, F,
                                                  _p(self.aa_work), st), "vhap_antialias_inplace_fwd")
            self.rgba_aa = color                                   # the prediction: the same buffer, antialiased
        else:
            _chk(L.vhap_antialias_fwd(_p(color), _p(self.rast), _p(self.clip), _p(self.tri), _p(self.opp), B, H, W, 4, V, F, _p(self.rgba_aa),
                                      _p(self.aa_work), st), "vhap_antialias_fwd")
        self._flush()
        if self.photo_total:
            # the photometric sum's last workgroup assembles the energy and the upstream gradient (no single-thread launches -- and
            # no cross-queue hand-overs -- between the forward and the backward pass; sharded: redone behind the count's all-reduce)
            self._join()          # the side branch: texture assembly, landmarks, statistics, arena clear, antialias pair discovery
            _chk(L.vhap_photo_fwd_total(_p(self.rgba_aa), _p(self.rgb), B, H, W, _p(acc[16:19]), _p(acc[0:6]), _p(acc[6:7]) if self.w_lmk else 0,
                                        _p(self.carry_terms) if self.carry else _p(acc[7:10]), _p(acc[20:24]), _p(acc[12:16]), self.w_lmk,
                                        self.w_reg, self.w_photo, _p(self.log), _p(self.d_sum), _p(self.gmax_bound), _p(self.photo_work),
                                        _p(self.aa_work) if self.aa_early_bwd else 0, _p(self.d_delta) if self.aa_early_bwd else 0,
                                        PRE | (_lib.CALL_TEX_TERMS_CONSUME if self.carry else 0), st),
                 "vhap_photo_fwd_total")
            self._delta_dirty = self.aa_early_bwd
            if sort_branch is not None and self.one_graph:
                # captured step: the sort is needed only by the backward's texture chain, ~250 us from here.  Forked BEHIND the photometric
                # sum -- next to it, it cost that bandwidth-bound reduction 20 us on the critical path (47 vs 26 us) -- it runs beside the
                # shading backward (issued by the backward's first _flush)
                self._side(sort_branch)
            return
        _chk(L.vhap_photo_fwd(_p(self.rgba_aa), _p(self.rgb), B, H, W, _p(acc[16:18]), PRE, st), "vhap_photo_fwd")
        self._join()
        _chk(L.vhap_energy_finalize(_p(acc[0:6]), _p(acc[6:7]) if self.w_lmk else 0, _p(self.carry_terms) if self.carry else _p(acc[7:10]),
                                    _p(acc[20:24]), _p(acc[12:16]) if self.want_reg else 0, self.w_lmk, self.w_reg, B, H, W, _p(self.log), st),
             "vhap_energy_finalize")
        if self.carry:
            _zero(self.carry_terms)

    def _disturb(self, st):
        """colour disturbance, in place on rgba (render_nvdiffrast.py:424-460); random numbers drawn in-kernel, or -- `self.injected`, a
        dict(w_fg, w_bg, idx) as HipDiffRenderer.make_disturbance draws it -- handed in, so that the oracle can replay them"""
        inj = self.injected
        B, H, W = self.B, self.H, self.W
        if inj is not None:
            w_fg, w_bg, idx = inj["w_fg"].int().contiguous(), inj["w_bg"].int().contiguous(), inj["idx"].long().contiguous()
            assert w_fg.numel() == w_bg.numel() == idx.numel() == B * H * W
            self._inj_keep = (w_fg, w_bg, idx)                    # (alive until the launch has run)
        if self.cov_list is not None:
            # the counting sort also leaves the list of covered pixels, in pixel order: the shading backward walks it (vhap_deferred_shade_bwd_list)
            _chk(self.L.vhap_disturb_inplace_list(_p(self.rgba), _p(self.cid), self.ncl, _p(w_fg) if inj is not None else 0,
                                                  _p(w_bg) if inj is not None else 0, _p(idx) if inj is not None else 0,
                                                  float(self.rate_fg or 0.0), float(self.rate_bg or 0.0), 0 if inj is not None else _p(self.rng),
                                                  B, H, W, _p(self.dist_ws), _p(self.keep), _p(self.cov_list), _p(self.n_bg), st),
                 "vhap_disturb_inplace_list")
            self._cov_list_fresh = True
            return
        _chk(self.L.vhap_disturb_inplace(_p(self.rgba), _p(self.cid), self.ncl, _p(w_fg) if inj is not None else 0,
                                         _p(w_bg) if inj is not None else 0, _p(idx) if inj is not None else 0,
                                         float(self.rate_fg or 0.0), float(self.rate_bg or 0.0), 0 if inj is not None else _p(self.rng),
                                         B, H, W, _p(self.dist_ws), _p(self.keep), st), "vhap_disturb_inplace")

    def _tex_backward(self, optimizer=None):
        """texel part of the texture-sampling backward (-> the gradient pyramid d_tex / d_mips), then -- unless `split_tex` -- tex_finish().
        -> True when the texture's Adam update was applied inside (fused into the last kernel of the chain)"""
        L, T, g = self.L, self.T, self.g
        B, H, W = self.B, self.H, self.W
        st = _stream()
        if not self.tex_bwd_on:
            return False
        n0 = self.albedo_tex.numel()
        d_tex, d_mips = g["d_tex"][:n0], g["d_tex"][n0:]
        if self.tb_ids:                                            # (sorted during the forward pass: only the accumulation is left)
            _chk(L.vhap_texture_grad_binned_sorted(T, T, 3, _p(self.texc), _p(self.texd), _p(self.d_albedo), B, H, W, _p(d_tex), _p(d_mips),
                                                   _p(self.texbin_work), self.texbin_work.numel(), _p(self.gmax_bound), st),
                 "vhap_texture_grad_binned_sorted")
        elif not NV.texture_grad_binned(T, 3, self.texc, self.texd, self.d_albedo, d_tex, d_mips, self.texbin_work):
            _chk(L.vhap_texture_bwd(_p(self.albedo_tex), _p(self.mips), 1, T, T, 3, _p(self.texc), _p(self.texd), _p(self.d_albedo), B, H, W,
                                    _p(d_tex), _p(d_mips), 0, 0, st), "vhap_texture_bwd")
        if not self.split_tex:
            return self.tex_finish(optimizer)
        return False

    def tex_finish(self, optimizer=None):
        """Gradient pyramid -> d(tex_extra) in ONE pass over the texture: every mip level gathered per texel (no fold cascade), TV / residual
        gradients, layout change and -- with `optimizer`, a HipAdam holding tex_extra -- the Adam update itself (the gradient is still
        written: it is the parameter's .grad; the update reads the step counter advanced at the head of the step).  Under frame sharding
        this runs AFTER the pyramid was averaged over the ranks (the regulariser part is identical on every rank, so it is added once,
        afterwards).  -> True when the update was applied."""
        L, tr, T, g = self.L, self.tr, self.T, self.g
        st = _stream()
        d_base = _p(self.d_base) if self.pca is not None else 0       # (PCA texture model: d(base texture) comes out of the same pass)

        def pca_bwd():
            if self.pca is not None:
                _chk(L.vhap_tex_pca_bwd(_p(self.pca_basis), _p(self.pca_src), _p(self.d_base), _p(tr.tex_pca), int(tr.tex_pca.shape[0]),
                                        self.pca.src_size, T, self.pca_scale, _p(self.ones), _p(self.pca_work), _p(g["tex_pca"]), _stream()),
                     "vhap_tex_pca_bwd")
        if not self.photometric:                                      # only the TV / residual gradients (a landmark stage that trains the texture)
            _chk(L.vhap_tex_prep_bwd_base(_p(self.albedo_tex), _p(tr.tex_extra), _p(self.nm["res_mask"]), 0, 0, 0, _p(self.ones), T, *self.tex_scales,
                                          _p(g["tex_extra"]), d_base, st), "vhap_tex_prep_bwd")
            pca_bwd()
            return False
        n0 = self.albedo_tex.numel()
        d_tex, d_mips = g["d_tex"][:n0], g["d_tex"][n0:]
        has_mips = self.mips.numel() > 0
        ng = _n_gather(T) if has_mips else 0
        if has_mips and ng < L.vhap_texture_num_levels(T, T):      # (texture sizes whose coarse levels cannot be gathered: fold them first)
            _chk(L.vhap_texture_mip_fold(_p(d_tex), _p(d_mips), 1, T, T, 3, ng, st), "vhap_texture_mip_fold")
        fu = optimizer.fused_update_args(tr.tex_extra) if (FUSE_TEX_ADAM and optimizer is not None and hasattr(optimizer, "fused_update_args")) else None
        if fu is not None and self.carry:
            m, v, lr, step, b1, b2, eps = fu
            assert has_mips and ng >= L.vhap_texture_num_levels(T, T)
            _chk(L.vhap_tex_finish_carry(_p(self.albedo_tex), _p(tr.tex_extra), _p(self.nm["res_mask"]), _p(self.painted), _p(d_tex), _p(d_mips), ng,
                                         _p(self.ones), T, *self.tex_scales, _p(g["tex_extra"]) if self.carry_keep_grad else 0, _p(m), _p(v),
                                         _p(lr), _p(step), b1, b2, eps, _p(self.mips), _p(self.tex_halo), _p(self.carry_terms),
                                         _lib.CALL_ADAM_STEP_ADVANCED, st), "vhap_tex_finish_carry")
            return True
        if fu is not None:
            m, v, lr, step, b1, b2, eps = fu
            flags = _lib.CALL_ADAM_STEP_ADVANCED if self.step_optimizer is not None else 0
            _chk(L.vhap_tex_prep_bwd_adam_base(_p(self.albedo_tex), _p(tr.tex_extra), _p(self.nm["res_mask"]), _p(d_tex),
                                               _p(d_mips) if has_mips else 0, ng, _p(self.ones), T, *self.tex_scales, _p(g["tex_extra"]), _p(m), _p(v),
                                               _p(lr), _p(step), b1, b2, eps, d_base, flags, st), "vhap_tex_prep_bwd_adam")
            pca_bwd()
            return True
        _chk(L.vhap_tex_prep_bwd_base(_p(self.albedo_tex), _p(tr.tex_extra), _p(self.nm["res_mask"]), _p(d_tex),
                                      _p(d_mips) if has_mips else 0, ng, _p(self.ones), T, *self.tex_scales, _p(g["tex_extra"]), d_base, st),
             "vhap_tex_prep_bwd")
        pca_bwd()
        return False

    def tex_fold(self):
        """the whole gradient pyramid folded into its level 0 (frame sharding: only level 0 -- 50 MB of the pyramid's 67 -- is exchanged)"""
        if not (self.tex_bwd_on and self.photometric and self.mips.numel() > 0):
            return
        n0 = self.albedo_tex.numel()
        _chk(self.L.vhap_texture_mip_fold_gather(_p(self.g["d_tex"][:n0]), _p(self.g["d_tex"][n0:]), self.T, self.T, 3, _stream()),
             "vhap_texture_mip_fold_gather")

    def tex_finish_rows(self, optimizer, d_strip, row0, nrows, advanced=False):
        """tex_finish() + Adam on the row strip [row0, row0 + nrows) of the texture from `d_strip` [nrows, T, 3], this rank's slice of the
        rank-averaged, folded level-0 gradient (frame sharding: vhap_tex_prep_bwd_adam_rows).  advanced: the step counter was advanced at
        the head of the step (HipAdam.advance) -- this piece of the update may then run on any stream, before or after the others."""
        tr, T, g = self.tr, self.T, self.g
        m, v, lr, step, b1, b2, eps = optimizer.fused_update_args(tr.tex_extra)
        _chk(self.L.vhap_tex_prep_bwd_adam_rows(_p(self.albedo_tex), _p(tr.tex_extra), _p(self.nm["res_mask"]), _p(d_strip), _p(self.ones), T,
                                                int(row0), int(nrows), *self.tex_scales, _p(g["tex_extra"]), _p(m), _p(v), _p(lr), _p(step),
                                                b1, b2, eps, _lib.CALL_ADAM_STEP_ADVANCED if advanced else 0, _stream()),
             "vhap_tex_prep_bwd_adam_rows")

    def _bwd_early(self):
        """landmark and offset-regulariser gradients: they depend on nothing the pixel chain produces (pure launch latency)"""
        L, tr, g, om = self.L, self.tr, self.g, self.om
        B, H, W, V = self.B, self.H, self.W, self.V
        st = _stream()
        if self.photometric and self.aa_early_bwd:
            # the POSITION part of the antialias backward (silhouette edges -> clip-space vertices); the colour part was computed beside the
            # photometric sum.  Needs the upstream gradient d_sum, nothing else of the backward: here, off the pixel chain
            _chk(L.vhap_antialias_photo_bwd(_p(self.rgba_aa), _p(self.rgb), _p(self.d_sum), _p(self.rast), _p(self.clip), _p(self.tri), _p(self.opp),
                                            _p(self.aa_work), _p(self.vert_mask), B, H, W, V, self.F, 0, _p(g["d_clip"]), st),
                 "vhap_antialias_photo_bwd")
        if self.w_lmk:
            l0, l1, b0, b1, boost = self.lmk_cfg
            _chk(L.vhap_landmark_bwd(_p(self.verts), _p(self.lm.vidx), _p(self.lm.bary), _p(self.mvp), _p(self.lmk2d), _p(self.c_lmk), B, V,
                                     self.lm.L, self.lmk2d.shape[1], l0, l1, b0, b1, boost, H, W, _p(g["d_verts"]), _p(self.d_mvp), st),
                 "vhap_landmark_bwd")
        else:
            _zero(self.d_mvp)
        if (self.has_offset or self.dyn) and any(self.off_scales):
            if self.dyn:
                _chk(L.vhap_offset_reg_bwd_batch(_p(self.off_b), _p(om.ptr), _p(om.col), _p(om.val), _p(om.w_lap), _p(om.w_abs), _p(om.rptr),
                                                 _p(om.ridx), B, om.V, om.nreg, *self.off_scales, _p(self.ones), _p(g["d_off_b"]), st),
                     "vhap_offset_reg_bwd_batch")
            else:
                _chk(L.vhap_offset_reg_bwd(_p(tr.static_offset), _p(om.ptr), _p(om.col), _p(om.val), _p(om.w_lap), _p(om.w_abs), _p(om.rptr),
                                           _p(om.ridx), om.V, om.nreg, *self.off_scales, _p(self.ones), _p(g["static_offset"]), st),
                     "vhap_offset_reg_bwd")
        if self.dyn and self.dyn_scale:
            _chk(L.vhap_offset_dynamic_reg(_p(tr.dynamic_offset), _p(self.ts), B, self.N, V, self.dyn_scale, _p(self.ones), 0,
                                           _p(g["dynamic_offset"]), st), "vhap_offset_dynamic_reg")

    def _bwd_pixel(self, world_size, after_first=None):
        """energy total -> photometric -> antialias -> shading backward (-> d_albedo, d_normal per pixel)"""
        L, tr, g, acc = self.L, self.tr, self.g, self.accF
        B, H, W, V, F, T = self.B, self.H, self.W, self.V, self.F, self.T
        st = _stream()
        if self.energy_fused:
            assert int(world_size) == 1                             # (done by the forward's photometric sum)
        else:
            _chk(L.vhap_energy_total_bound(_p(self.log), _p(acc[16:18]), _p(self.n_global), self.w_photo, int(world_size), _p(self.d_sum),
                                           _p(acc[12:16]), _p(self.gmax_bound), st), "vhap_energy_total_bound")
        early_aa = self.aa_inplace and self.aa_early_bwd       # (nothing to do here: see aa_early_bwd)
        if early_aa:
            pass
        elif self.aa_inplace:
            # no dense gradient images: the loss gradient is evaluated on the fly (here at the pixels of the antialias pair list, in the
            # shading backward everywhere); the sparse colour part of the antialias backward travels in d_delta
            _chk(L.vhap_antialias_photo_bwd(_p(self.rgba_aa), _p(self.rgb), _p(self.d_sum), _p(self.rast), _p(self.clip), _p(self.tri), _p(self.opp),
                                            _p(self.aa_work), _p(self.vert_mask), B, H, W, V, F, _p(self.d_delta), _p(g["d_clip"]), st),
                 "vhap_antialias_photo_bwd")
        else:
            _chk(L.vhap_photo_bwd(_p(self.rgba_aa), _p(self.rgb), _p(self.d_sum), B, H, W, _p(self.d_rgba_aa), _p(self.d_color), st), "vhap_photo_bwd")
            _chk(L.vhap_antialias_bwd(_p(self.aa_in), _p(self.rast), _p(self.clip), _p(self.tri), _p(self.opp), _p(self.d_rgba_aa), _p(self.aa_work),
                                      _p(self.vert_mask), B, H, W, 4, V, F, _p(self.d_color), _p(g["d_clip"]),
                                      _lib.CALL_AA_PASSTHROUGH_DONE, st), "vhap_antialias_bwd")   # d_color already holds the pass-through copy of d_rgba_aa
        if after_first is not None and not early_aa:
            after_first()
        # (the backward of the disturbance -- d_rgba = d_color * keep -- is folded into the shading backward)
        if self.deferred:
            # shading + texture-coordinate backward in one pass, from re-computed attributes (nothing of the forward's G-buffer is re-read)
            # (the texture gradient walks the sorted list of covered pixels -- or does not run at all: background d_albedo is never read)
            skip_bg = bool(self.tb_ids or not self.tex_bwd_on)
            args = (_p(self.clip), _p(self.tri), _p(self.vn), _p(self.uv), _p(self.tri_uv), _p(self.albedo_tex), _p(self.mips),
                    T, T, _p(tr.lights), _p(self.sh_const), _p(self.rast), *self._upstream(),
                    _p(self.keep) if self.disturb_on else 0, _p(self.c_reg) if self.want_reg else 0,
                    _p(acc[12:16]) if self.want_reg else 0, B, V, self.uv.shape[0], F, H, W, _p(self.texc), _p(self.texd),
                    _p(self.d_albedo), _p(self.d_normal), _p(self.d_texc), _p(self.d_texd), 0, _p(self.def_work), self.def_work.numel(), 0, 0)
            flags = self.delta_flag | (_lib.CALL_SKIP_BG_GRAD if skip_bg else 0)
            if self.cov_list is not None and self._cov_list_fresh and skip_bg:
                # over the covered pixels only, in the order the disturbance's counting sort listed them during this forward pass
                _chk(L.vhap_deferred_shade_bwd_list(*args, _p(self.cov_list), _p(self.n_bg), flags, st), "vhap_deferred_shade_bwd_list")
            else:
                _chk(L.vhap_deferred_shade_bwd(*args, flags, st), "vhap_deferred_shade_bwd")
            if after_first is not None and early_aa:          # (the main chain's kernel first, then the side branches forked behind the sum)
                after_first()
            return
        _chk(L.vhap_shade_bwd(_p(self.normal), _p(self.albedo_px), _p(self.rast), _p(tr.lights), _p(self.sh_const), _p(self.d_color),
                              _p(self.keep) if self.disturb_on else 0, _p(self.c_reg) if self.want_reg else 0,
                              _p(acc[12:16]) if self.want_reg else 0, B, H, W, _p(self.d_albedo), _p(self.d_normal), _p(g["lights"]), st),
             "vhap_shade_bwd")

    def _upstream(self):
        """(d_rgba | pred, gt, d_sum, d_delta) arguments of the deferred backward: a dense gradient image, or the on-the-fly form"""
        if self.aa_inplace:
            return 0, _p(self.rgba_aa), _p(self.rgb), _p(self.d_sum), _p(self.d_delta)
        return _p(self.d_color), 0, 0, 0, 0

    def _bwd_pixel_finish(self):
        """the tail of the deferred shading backward that nothing downstream waits for: lights-gradient partials -> d_lights, and d_delta back
        to all-zero (issued AFTER the texture branch was forked, so that neither delays it)"""
        if not self.deferred:
            return
        L, tr, acc = self.L, self.tr, self.accF
        _chk(L.vhap_deferred_lights_reduce(_p(self.def_work), _p(tr.lights), _p(self.sh_const), _p(self.c_reg) if self.want_reg else 0,
                                           _p(acc[12:16]) if self.want_reg else 0, self.B, self.H, self.W, _p(self.g["lights"]), _stream()),
             "vhap_deferred_lights_reduce")
        self._clear_delta()
        if self.one_graph:            # the next step's forward accumulators: their last reader of THIS step is the reduction above
            _zero(self.accF)
            self._acc_clean = True

    def _clear_delta(self):
        if self.aa_inplace:
            _chk(self.L.vhap_antialias_clear_delta(_p(self.aa_work), self.B, self.H, self.W, _p(self.d_delta), _stream()), "vhap_antialias_clear_delta")
            self._delta_dirty = False

    def _bwd_uv(self):
        """gradient w.r.t. the texture coordinates and their screen-space derivatives (input of the G-buffer backward)"""
        L, B, H, W, T = self.L, self.B, self.H, self.W, self.T
        if self.deferred:
            return                                                # produced by vhap_deferred_shade_bwd already
        _chk(L.vhap_texture_bwd(_p(self.albedo_tex), _p(self.mips), 1, T, T, 3, _p(self.texc), _p(self.texd), _p(self.d_albedo), B, H, W,
                                0, 0, _p(self.d_texc), _p(self.d_texd), _stream()), "vhap_texture_bwd")

    def _frame_prep_bwd(self, st, camera=False):
        """per-frame parameters (+ the joint-regression part of the offset gradient); with dynamic offsets: then the offset gradient of every
        frame -- skinning part g_shaped + d_off_b (joint regression, regularisers) -- to static_offset (summed) and dynamic_offset[timesteps].
        camera: the uncalibrated camera's backward (d_mvp -> d focal_length) rides in the same launch as one more workgroup"""
        L, tr, fb, fm, g = self.L, self.tr, self.fb, self.fm, self.g
        B, V, J = self.B, self.V, self.J
        off_in = self.off_b if self.dyn else tr.static_offset
        g_off = g["d_off_b"] if self.dyn else (g["static_offset"] if self.has_offset else None)
        fp_args = (_p(self.ts), _p(tr.shape), _p(tr.expr), _p(tr.rotation), _p(tr.translation), _p(tr.neck_pose),
                   _p(tr.jaw_pose), _p(tr.eyes_pose), _p(fm.JS), _p(fm.jreg_idx), _p(fm.jreg_w), fm.jreg_n,
                   _p(off_in), fm.parents, self.weights, _p(self.Jrest), _p(g["d_coef"]), _p(g["d_A"]), _p(g["d_t"]),
                   _p(self.ones), B, self.Bp, self.N, self.NS, self.NE, J, fb.Kp, V, _p(g["shape"]), _p(g["expr"]),
                   _p(g["rotation"]), _p(g["translation"]), _p(g["neck_pose"]), _p(g["jaw_pose"]), _p(g["eyes_pose"]),
                   _p(g_off), self.off_flag)
        if camera:
            _chk(L.vhap_frame_prep_bwd_camera(*fp_args, _p(self.RT), _p(self.d_mvp), 0, self.H, self.W, self.focal_scale, _p(g["focal_length"]), st),
                 "vhap_frame_prep_bwd_camera")
        else:
            _chk(L.vhap_frame_prep_bwd(*fp_args, st), "vhap_frame_prep_bwd")
        if self.dyn:
            _chk(L.vhap_offset_grad_finish(_p(self.g_shaped), _p(g["d_off_b"]), _p(self.ts), B, self.N, V,
                                           _p(g["static_offset"]) if self.has_offset else 0, _p(g["dynamic_offset"]), st),
                 "vhap_offset_grad_finish")

    def _bwd_params(self):
        """d_verts, d_mvp -> camera -> skinning -> per-frame parameters (the tail shared by photometric and landmark-only stages)"""
        L, tr, fb, fm, g = self.L, self.tr, self.fb, self.fm, self.g
        B, H, W, V, J = self.B, self.H, self.W, self.V, self.J
        st = _stream()
        _chk(L.vhap_flame_skin_bwd(_p(g["d_verts"]), 0, _p(self.v_posed), _p(self.A), _p(fb.w), _p(fb.basisT), B, V, fb.Vp, fb.Kb, fb.Kp,
                                   _p(self.g_posed), _p(self.g_shaped), 0, _p(g["d_coef"]), _p(g["d_A"]), _p(g["d_t"]), PRE, st),
             "vhap_flame_skin_bwd")
        if self.has_offset and not self.dyn:
            _chk(L.vhap_sum_frames(_p(self.g_shaped), B, V * 3, _p(g["static_offset"]), st), "vhap_sum_frames")
        if not self.calibrated and not self.cam_fused:                # (the focal length is a parameter only without calibration, tracker.py:148-157)
            _chk(L.vhap_camera_focal_bwd(_p(self.RT), _p(self.d_mvp), B, 0, H, W, self.focal_scale, _p(g["focal_length"]), st),
                 "vhap_camera_focal_bwd")
        self._frame_prep_bwd(st, camera=self.cam_fused)

    def _bwd_geometry(self, early=None, after_first=None):
        """G-buffer backward -> vertex normals -> clip transform -> camera -> skinning -> per-frame parameters"""
        L, g = self.L, self.g
        B, H, W, V, F = self.B, self.H, self.W, self.V, self.F
        st = _stream()
        _chk(L.vhap_gbuffer_bwd(_p(self.clip), _p(self.tri), _p(self.vn), _p(self.uv), _p(self.tri_uv), _p(self.rast), _p(self.d_normal),
                                _p(self.d_texc), _p(self.d_texd), 0, 0, _p(self.face_mask), B, V, F, H, W, _p(g["d_clip"]), _p(g["d_vn"]), st),
             "vhap_gbuffer_bwd")
        if after_first is not None:
            after_first()
        if early == "side2":
            # backward(part="all"): the vertex stage waits for EVERYTHING the second side stream holds by now -- the early branch (landmarks,
            # offset regularisers, the antialiasing's position part) and, behind it, the tail of the shading backward (lights gradient, delta
            # clear, accumulator clear), all long finished -- so that the Adam update at the end of the chain needs no wait of its own: one
            # hand-over on the main chain instead of two (~6 us each)
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.side2)
            early = None
        if early is True:                                             # (the event of the geometry part's early branch, issued by now)
            early = self._early_ev
        if early is not None:
            torch.cuda.current_stream().wait_event(early)
        self._bwd_vertex_stage()

    def _bwd_vertex_stage(self):
        """vertex-normal / clip-transform / skinning backward + offset sum in one kernel (+ the split-K coefficient GEMM), then the tiny
        camera chain and the per-frame parameters"""
        L, g, tr, fb, fm = self.L, self.g, self.tr, self.fb, self.fm
        B, H, W, V, J = self.B, self.H, self.W, self.V, self.J
        st = _stream()
        _chk(L.vhap_verts_bwd_fused(_p(self.verts), _p(self.csr.tri), _p(self.csr.ptr), _p(self.csr.idx), _p(self.vn), _p(self.vn_inv), _p(g["d_vn"]),
                                    _p(self.mvp), _p(g["d_clip"]), _p(g["d_verts"]), _p(self.v_posed), _p(self.A), _p(fb.w), _p(fb.basisT), B, V,
                                    fb.Vp, fb.Kb, fb.Kp, _p(self.vn_scratch), _p(self.g_posed), _p(self.g_shaped), _p(g["d_coef"]), _p(g["d_A"]),
                                    _p(g["d_t"]), _p(self.d_mvp), _p(g["static_offset"]) if (self.has_offset and not self.dyn) else 0, PRE, st),
             "vhap_verts_bwd_fused")
        # the camera backward (d_mvp -> d focal_length, a single wave) feeds nothing but Adam: one more workgroup of the per-frame backward (as a
        # launch of its own on a side branch its event record delayed the per-frame backward and Adam waited for a second event)
        if not self.calibrated and not self.cam_fused:
            self._side(lambda: _chk(L.vhap_camera_focal_bwd(_p(self.RT), _p(self.d_mvp), B, 0, H, W, self.focal_scale, _p(g["focal_length"]),
                                                            _stream()), "vhap_camera_focal_bwd"), self.side2)
        self._frame_prep_bwd(st, camera=self.cam_fused)
        self._flush()

    def backward(self, world_size=1, part="all", optimizer=None):
        """part = 'all': the whole backward as one DAG of up to three branches (one GPU).  `optimizer`: a HipAdam whose texture update is
        fused into the last kernel of the texture chain; with `step_optimizer` set as well (GraphedStep) the rest of its update is issued
        here too, next to the tail of the texture chain.
        Under frame sharding the backward is captured in two pieces so that the big collective can start early: 'pixel_tex' (energy,
        pixel chain, the complete texture gradient -> g['tex_extra']) -- the caller launches the asynchronous all-reduce of that
        gradient -- then 'geometry' (everything else), which runs underneath it."""
        if part in ("all", "pixel_tex"):
            if not getattr(self, "_arena_clean", False):              # (normally done on the forward's side branch already)
                _zero(self.arena)
            self._arena_clean = False
        if not self.photometric:
            # landmark-only stage: E = landmark + regularisers; one short serial chain ('geometry' of a sharded split is empty)
            if part in ("all", "pixel_tex"):
                _chk(self.L.vhap_energy_total(_p(self.log), 0, 0, 0.0, int(world_size), 0, _stream()), "vhap_energy_total")
                self._bwd_early()
                if self.tex_bwd_on:
                    self.tex_finish()
                    if optimizer is not None:
                        optimizer.step(only=(self.tr.tex_extra,), advance=False)
                self._bwd_params()
            return
        if part == "all":
            self._early_ev = None

            def early_branch():
                self._bwd_early()
                if self.overlap:
                    self._early_ev = torch.cuda.Event()
                    self._early_ev.record()
            self._side(early_branch, self.side2)                     # (the other side stream: the texture-gradient sort may still be on the first)
            self._bwd_pixel(world_size, after_first=self._flush)
            # fork as soon as d_albedo exists: the texture gradient (uv-binned accumulation, then ONE pass that gathers the pyramid, adds the
            # regulariser gradients and applies the Adam update) on the side branch, the geometry chain on this one
            def tex_chain():
                done = self._tex_backward(optimizer)
                if optimizer is not None and self.tex_bwd_on and not done:
                    optimizer.step(only=(self.tr.tex_extra,), advance=False, advanced=self.step_optimizer is not None)
                if self.step_optimizer is not None and self.pca is not None and self.tex_bwd_on:
                    # the PCA code's gradient is the LAST thing the texture chain produces (vhap_tex_pca_bwd): its update belongs here,
                    # not into the launch stream's Adam call, which runs long before
                    self.step_optimizer.step(only=(self.tr.tex_pca,), advanced=True)
            self._side(tex_chain)
            self._side(self._bwd_pixel_finish, self.side2)            # (nothing downstream reads these two: beside the geometry chain, not ahead of it)
            self._bwd_uv()
            self._bwd_geometry("side2", after_first=self._flush)
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.side2)  # (the camera's backward when it is a launch of its own; otherwise already waited for: dropped by the executor)
            if getattr(self, "_sort_done", None) is not None:          # (early stores: see _forward_deferred; long complete by now)
                torch.cuda.current_stream().wait_event(self._sort_done)
                self._sort_done = None
            if self.step_optimizer is not None:                       # every other parameter: next to the tail of the texture branch
                late = (self.tr.tex_extra, self.tr.tex_pca) if (self.pca is not None and self.tex_bwd_on) else (self.tr.tex_extra,)
                self.step_optimizer.step(skip=late, advanced=True)
            self._join()
        elif part == "pixel_tex":
            if getattr(self, "_sort_fn", None) is not None and self.sort_in_backward:
                self._side(self._sort_fn)                              # (forked at the head of the plan: beside the shading backward)
            self._bwd_pixel(world_size)
            self._side(self._bwd_pixel_finish, self.side2)
            if self.split_tex and self.overlap:
                # sharded texture update: the texture gradient (tile accumulation, then the pyramid folded into level 0 -- what is exchanged)
                # is a SIDE chain of this plan and its open tail: the caller replays the plan with a deferred join, lets its communication
                # stream wait for the tail and launches the reduce-scatter there, while the launch stream goes straight on to the geometry
                # plan -- tile accumulation, fold and collective all run UNDER the G-buffer backward, as the tile accumulation does on one
                # GPU (round 5: issued on the launch stream they sat in series with it, profiles/r05_call7_sharded_step_timeline.txt)
                def tex_chain():
                    self._tex_backward()
                    if not self.fold_outside:                          # (fold_outside: the caller folds on its communication stream, so that a
                        self.tex_fold()                                # geometry plan that waits for the TILE pass need not wait for the fold)
                self._side(tex_chain)
                self._flush()
                torch.cuda.current_stream().wait_stream(self.side2)
                self._join()                                           # (ends the capture's fork; a deferred-join replay leaves it out)
                return
            self._tex_backward()
            if self.split_tex:                                        # (sharded texture update: the exchange is on level 0 of the pyramid)
                self.tex_fold()
            self._flush()
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.side2)
        elif part == "geometry":
            # like part 'all': the early branch (landmarks, offset regularisers, the antialiasing's position part) is marked here but issued
            # behind the G-buffer backward's launch, so that the plan executor keeps the G-buffer backward -- the chain the step waits for --
            # on the LAUNCH stream and gives the side stream to the early branch (captured first, the early branch took the launch stream
            # and the G-buffer backward a side stream shared with the pixel plan's texture chain: round 5)
            self._early_ev = None

            def early_branch():
                self._bwd_early()
                if self.overlap:
                    self._early_ev = torch.cuda.Event()
                    self._early_ev.record()
            self._side(early_branch)
            self._bwd_uv()
            self._bwd_geometry(True, after_first=self._flush)
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.side2)  # (the camera backward)
            self._join()
        else:
            raise ValueError(part)

    def log_dict(self):
        """Views into the device log vector, keyed like FlameTracker.compute_energy's log_dict (terms the stage does not have read 0)."""
        return {k: self.log[i] for i, k in enumerate(LOG_NAMES) if k != "rest"}
