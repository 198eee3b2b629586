"""Oracle-side fit loop -- TEST INFRASTRUCTURE ONLY (never imported by vhap_amd).

Restates, on torch-CPU float64 leaves and torch.optim.Adam, what the reference does around compute_energy:
    train_parameters     GlobalTracker.get_train_parameters   (vhap/model/tracker.py:1465-1513)
    configure_optimizer  FlameTracker.configure_optimizer     (vhap/model/tracker.py:159-211)
    optimize_iter        GlobalTracker.optimize_iter          (vhap/model/tracker.py:1418-1462, logging branches dropped)
    export               GlobalTracker.save_result            (vhap/model/tracker.py:1152-1218; the arrays only)
so that "K optimiser steps, then export" can be compared array by array with the HIP path (tests/test_fit_parity_gpu.py).

Pinned: the group / learning-rate table against the reference's own configure_optimizer (tests/golden/energy_golden.npz ->
tests/test_energy_golden.py::test_oracle_fit_groups_match_reference_golden); the energy it differentiates is energy_ref.total_energy
(pinned as described there; the raster-op internals remain "parity unpinned")."""
import numpy as np
import torch

from . import energy_ref

PER_FRAME = ("rotation", "translation", "neck_pose", "jaw_pose", "eyes_pose", "expr")


def train_parameters(P, cfg, stage, calibrated=False):
    """tracker.py:1465-1513: stage -> {group key: [leaf tensors]} (insertion order = the reference's)."""
    o = set(cfg.pipeline[stage].optimizable_params)
    m = cfg.model
    params = {}
    if "cam" in o and not calibrated:
        params["cam"] = [P["focal_length"]]
    if "shape" in o:
        params["shape"] = [P["shape"]]
    if "texture" in o and m.tex_extra:
        params["tex_extra"] = [P["tex_extra"]]
    if "static_offset" in o and m.use_static_offset:
        params["static_offset"] = [P["static_offset"]]
    if "lights" in o and P.get("lights") is not None:
        params["lights"] = [P["lights"]]
    if "pose" in o:
        params["translation"] = [P["translation"]]
        params["rotation"] = [P["rotation"]]
    if "joints" in o:
        params["eyes"] = [P["eyes_pose"]]
        params["neck"] = [P["neck_pose"]]
        params["jaw"] = [P["jaw_pose"]]
    if "expr" in o:
        params["expr"] = [P["expr"]]
    if "dynamic_offset" in o and m.use_dynamic_offset:
        params["dynamic_offset"] = [P["dynamic_offset"]]
    return params


def optimizer_groups(params, cfg, lr_scale=1.0, calibrated=False):
    """tracker.py:159-211 -> list of (group keys, lr): named groups with their own learning rate first, then the default group."""
    params = dict(params)
    lr = cfg.lr
    group_def = {"translation": ["translation"], "expr": ["expr"], "light": ["lights"]}
    group_lr = {"translation": lr.translation, "expr": lr.expr, "light": lr.light}
    if not calibrated:
        group_def["cam"], group_lr["cam"] = ["cam"], lr.camera
    if cfg.model.use_static_offset:
        group_def["static_offset"], group_lr["static_offset"] = ["static_offset"], lr.static_offset
    if cfg.model.use_dynamic_offset:
        group_def["dynamic_offset"], group_lr["dynamic_offset"] = ["dynamic_offset"], lr.dynamic_offset
    groups = []
    for name, keys in group_def.items():
        sel, used = [], []
        for k in keys:
            if k in params:
                sel += params.pop(k)
                used.append(k)
        if sel:
            groups.append({"params": sel, "lr": group_lr[name] * lr_scale, "keys": used})
    rest = [p for v in params.values() for p in v]
    groups.append({"params": rest, "lr": lr.base * lr_scale, "keys": list(params.keys())})
    return groups


def configure_optimizer(P, cfg, stage, lr_scale=1.0, calibrated=False):
    groups = optimizer_groups(train_parameters(P, cfg, stage, calibrated), cfg, lr_scale, calibrated)
    return torch.optim.Adam([{"params": g["params"], "lr": g["lr"]} for g in groups], lr=cfg.lr.base * lr_scale)


def optimize_iter(P, optimizer, model, topo, cfg, sample, stage, tex_painted, uvmask_res, image_size, disturb=None, tid=None, dtype=torch.float64):
    """One step of tracker.py:1418-1462: energy, zero_grad, backward, Adam.  Returns the log dict (floats)."""
    E, log, extras = energy_ref.total_energy(P, model, topo, cfg, sample, stage, tex_painted, uvmask_res, image_size, dtype=dtype,
                                             disturb=disturb, tid=tid)
    optimizer.zero_grad()
    E.backward()
    optimizer.step()
    out = {k: float(v.detach()) for k, v in log.items()}
    out["total"] = float(E.detach())
    if "diffuse_detach_normal" in extras:
        # reg_diffuse = w (relu(max(diffuse) - 1) + ...) (tracker.py:547-550): the max runs over pixels AND colour channels, and its gradient
        # goes to the arg-max element only.  Reported (maximum, lead of the first channel over the second) so that trajectory records show
        # whether the max term was active and how far the fit is from a tie between channels.
        d = extras["diffuse_detach_normal"].detach()
        per_channel = torch.sort(d.reshape(-1, d.shape[-1]).max(dim=0).values, descending=True).values
        out["diffuse_max"] = float(per_channel[0])
        out["diffuse_channel_gap"] = float(per_channel[0] - per_channel[1])
    return out


def export(P, image_size, calibrated=False):
    """The arrays of tracker.py:1158-1218 (save_result) that depend on the fit."""
    c = lambda t: t.detach().cpu().numpy().astype(np.float32)
    out = {k: c(P[k]) for k in PER_FRAME + ("shape",)}
    n = P["expr"].shape[0]
    out["timestep_id"], out["n_processed_frames"], out["image_size"] = np.arange(n), np.array(n), np.array(image_size)
    if not calibrated:
        out["focal_length"] = c(P["focal_length"])
    for k in ("tex_extra", "lights", "static_offset", "dynamic_offset"):
        if P.get(k) is not None:
            out[k] = c(P[k])
    return out
