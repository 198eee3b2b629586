"""The reference's entry point over this library: `GlobalTracker(cfg)` with the REFERENCE's config object (vhap/track.py:16-21,
vhap/model/tracker.py:1221-1277).

The reference's constructor does five things: builds FLAME + renderer from `cfg.model` / `cfg.render` (FlameTracker.__init__,
tracker.py:39-77), runs the landmark detector if needed (:1263-1277 -- out of scope here: landmarks must exist on disk, as with
`cfg.exp.reuse_landmarks`), opens the dataset class `cfg.data._target` names (:1250-1253), reads image size and number of timesteps
from it (:1257-1258) and allocates the parameters (:1261).  `tracker_from_reference_config` does the same with

  * the config converted field by field into vhap_amd.config (same names, same defaults; `render.backend` becomes 'hip');
  * the dataset opened through the reference's OWN class, imported from the user's checkout (datasets / decoding are out of scope and
    are not re-implemented): every image goes through the reference's `__getitem__` once -- decode, colour correction, scale factor,
    compositing exactly as the reference does them on the host -- and the resulting uint8 frames stay resident in HBM
    (ingest.FrameStore; the reference re-runs those transforms on DataLoader workers every epoch).  `device_prepare=True` instead
    uploads the DECODER's output and applies colour correction / scale factor / compositing on the device, bit for bit
    (ingest.FrameStore.from_decoded; tests/test_ingest.py);
  * FLAME from the licensed pickles through FlameHead.from_flame_pickle (asset paths of vhap/model/flame.py:38-46).

Nothing here is on the timed path; it is host-side plumbing so that `python -m vhap.track`-style drivers can switch libraries.
"""
import dataclasses
import importlib
import os
import sys

import numpy as np
import torch

from . import config as C

# vhap/model/flame.py:38-46, relative to the working directory like the reference
FLAME_MODEL_PATH = "asset/flame/flame2023.pkl"
FLAME_PARTS_PATH = "asset/flame/FLAME_masks.pkl"
FLAME_PAINTED_TEX_PATH = "asset/flame/tex_mean_painted.png"
FLAME_TEX_PATH = "asset/flame/FLAME_texture.npz"


def convert_config(ref_cfg, target=None):
    """The reference's BaseTrackingConfig (or NersembleTrackingConfig, or anything with the same attribute tree) -> vhap_amd.config.
    Every field the product's config has is read from the reference object BY NAME (lists become tuples); fields the hot path does not
    consume (logging, experiment folders, data paths) are simply not present on this side.  The reference's __post_init__ has already
    run on `ref_cfg` (occluded regions appended to the stages' align_* lists): values are copied as they stand."""
    out = C.BaseTrackingConfig() if target is None else target

    def copy(dst, src):
        for f in dataclasses.fields(dst):
            if not hasattr(src, f.name):
                continue
            v = getattr(src, f.name)
            cur = getattr(dst, f.name)
            if dataclasses.is_dataclass(cur):
                copy(cur, v)
            else:
                setattr(dst, f.name, tuple(v) if isinstance(v, list) else v)
    copy(out, ref_cfg)
    if getattr(out.render, "backend", "hip") != "hip":
        out.render.backend = "hip"                                  # ('nvdiffrast' in the reference: the renderer this library replaces)
    out.device = str(getattr(ref_cfg, "device", out.device))
    return out


def open_reference_dataset(data_cfg, checkout=None, **kw):
    """The reference's dataset object: class `data_cfg._target` (vhap.data.video_dataset.VideoDataset, ...nersemble_dataset.NeRSembleDataset)
    imported from the user's checkout (`checkout`: its root directory, put on sys.path when `vhap` is not importable yet)."""
    if checkout is not None and checkout not in sys.path:
        sys.path.insert(0, checkout)
    module_name, class_name = data_cfg._target.rsplit(".", 1)       # base.py:19-22 import_module
    cls = getattr(importlib.import_module(module_name), class_name)
    return cls(cfg=data_cfg, **kw)


def frames_from_reference_dataset(ds, device="cuda", device_prepare=False):
    """Reads every image of a reference dataset opened with img_to_tensor=False, batchify_all_views=False ONCE and returns the dataset
    dict GlobalTracker takes: `frames` (ingest.FrameStore, uint8 resident), `lmk2d` [N,L,3], `timestep_index` [N], `camera_index` [N] and,
    for calibrated data, `intrinsic` [N,3,3] / `extrinsic` [N,3|4,4].

    device_prepare=False: items come out of the reference's own `__getitem__` -> apply_transforms (host; colour correction, scale factor,
        compositing done by the reference's code); the store composites nothing more.
    device_prepare=True: the transforms are held back on the host (`apply_transforms` of this ONE dataset object is bypassed) and
        applied on the device from the dataset's own settings: cfg.scale_factor / n_downsample_rgb / background_color and, when the
        dataset carries them, its per-camera colour transforms (NeRSembleDataset.color_correction)."""
    from .ingest import FrameStore, scale_item_properties
    if getattr(ds, "batchify_all_views", False) or getattr(ds, "img_to_tensor", False):
        raise ValueError("open the reference dataset with img_to_tensor=False, batchify_all_views=False")
    cfg = ds.cfg
    keep = None
    if device_prepare:
        keep = ds.apply_transforms
        ds.apply_transforms = lambda item: item                     # (instance attribute: the class stays untouched)
    try:
        items = [ds[i] for i in range(len(ds))]
    finally:
        if keep is not None:
            del ds.apply_transforms
    n = len(items)
    if n == 0:
        raise ValueError("the reference dataset is empty")
    rgb = np.stack([np.asarray(it["rgb"], np.uint8) for it in items])
    has_alpha = all("alpha_map" in it for it in items)
    alpha = np.stack([np.asarray(it["alpha_map"], np.uint8) for it in items]) if has_alpha else None
    cam_ids = [it.get("camera_id", "0") for it in items]
    cams = sorted(set(cam_ids), key=cam_ids.index)
    cam_index = np.array([cams.index(c) for c in cam_ids], np.int32)
    out = {"camera_index": cam_index}
    if len(cams) > 1:                                               # multi-view: the frames are grouped by timestep (GlobalTracker docstring)
        out["timestep_index"] = np.array([int(it["timestep_index"]) for it in items], np.int64)
    lmk = np.stack([np.asarray(it["lmk2d"], np.float32) for it in items]) if all("lmk2d" in it for it in items) else None
    K = np.stack([np.asarray(it["intrinsic"], np.float32) for it in items]) if all("intrinsic" in it for it in items) else None
    RT = np.stack([np.asarray(it["extrinsic"], np.float32) for it in items]) if all("extrinsic" in it for it in items) else None
    if device_prepare:
        cc = None
        if getattr(cfg, "use_color_correction", False) and getattr(ds, "color_correction", None):
            cc = np.stack([np.asarray(ds.color_correction[c], np.float64) for c in cams])
        bg = getattr(cfg, "background_color", None)
        store = FrameStore.from_decoded(rgb, alpha if (bg is not None or getattr(cfg, "use_alpha_map", False)) else None, bg, device=device,
                                        camera_index=cam_index if cc is not None else None, color_correction=cc,
                                        scale_factor=float(cfg.scale_factor), n_downsample_rgb=getattr(cfg, "n_downsample_rgb", None))
        scaled = scale_item_properties(store.prepared, lmk, K)
        lmk, K = scaled.get("lmk2d", lmk), scaled.get("intrinsic", K)
    else:
        store = FrameStore(rgb, None, None, device=device)           # (already composited by the reference's apply_background_color)
    out["frames"] = store
    if lmk is not None:
        out["lmk2d"] = torch.from_numpy(lmk).to(device)
    if getattr(cfg, "calibrated", False):
        if K is None or RT is None:
            raise ValueError("calibrated data without intrinsic / extrinsic in the dataset items")
        out["intrinsic"], out["extrinsic"] = torch.from_numpy(K).to(device), torch.from_numpy(RT).to(device)
    return out


def load_base_texture(cfg_model, painted_tex_path=FLAME_PAINTED_TEX_PATH, tex_space_path=FLAME_TEX_PATH):
    """FlameTexPainted (flame.py:649-662): the painted PNG / 255, first three channels, nearest-resized to tex_resolution when its size
    differs (F.interpolate's default mode) -- or, tex_painted = False, the path of the FLAME texture space (FlameTexPCA reads it)."""
    if not cfg_model.tex_painted:
        return tex_space_path
    from PIL import Image
    tex = torch.tensor(np.array(Image.open(painted_tex_path))[:, :, :3]) / 255
    tex = tex[None].permute(0, 3, 1, 2)
    T = int(cfg_model.tex_resolution)
    if tex.shape[-1] != T or tex.shape[-2] != T:
        tex = torch.nn.functional.interpolate(tex, [T, T])
    return tex[0].numpy()


def tracker_from_reference_config(cls, ref_cfg, dataset=None, flame=None, base_texture=None, checkout=None, device_prepare=False,
                                  flame_model_path=FLAME_MODEL_PATH, flame_masks_path=FLAME_PARTS_PATH):
    """GlobalTracker(cfg) of the reference (tracker.py:1221-1261): see the module docstring.
        dataset       None: opened from ref_cfg.data through the reference's class; or an already opened reference dataset object; or the
                      dict GlobalTracker takes
        flame         None: (model, topo) loaded from the licensed pickles at the reference's asset paths; or that pair
        base_texture  None: the painted texture asset (or the texture space's path for tex_painted = False)"""
    cfg = convert_config(ref_cfg)
    dev = cfg.device
    if flame is None:
        for p in (flame_model_path, flame_masks_path):
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p}: the licensed FLAME assets are needed (README of the reference, 'FLAME'); "
                                        "pass flame=(model, topo) to use another model")
        from .flame_assets import load_flame_model
        flame = load_flame_model(flame_model_path, flame_masks_path, cfg.model.n_shape, cfg.model.n_expr, cfg.model.add_teeth,
                                 tex_clusters=tuple(cfg.model.tex_clusters))
    model, topo = flame
    if base_texture is None:
        base_texture = load_base_texture(cfg.model)
    if dataset is None:
        dataset = open_reference_dataset(ref_cfg.data, checkout, img_to_tensor=False, batchify_all_views=False)
    if not isinstance(dataset, dict):
        dataset = frames_from_reference_dataset(dataset, dev, device_prepare)
    tr = cls(cfg, model, topo, base_texture, dataset)
    tr.reference_cfg = ref_cfg
    path = getattr(getattr(ref_cfg, "model", None), "flame_params_path", None)
    if path is not None:                                            # tracker.py:1260-1261
        tr.load_from_tracked_flame_params(path)
    return tr


def detect_landmarks(ref_cfg, weights, checkout=None, face_detector=None, device="cuda", face_detector_weights=None):
    """`GlobalTracker.detect_landmarks(cfg)` of the reference (vhap/model/tracker.py:1263-1277) over vhap_amd.landmarks: the dataset opened through
    the reference's own class with `use_landmark = False`, and -- for `landmark_source == 'face-alignment'`, unless `cfg.exp.reuse_landmarks` finds
    the file -- every frame annotated by the FAN network on the matrix cores, the npz written where the reference's dataset will look for it.
    `weights`: the `face_alignment` package's state dict (or a path torch.load reads): a third-party download, like the reference's.
    `face_detector`: callable(img) -> boxes, or `face_detector_weights`: the state dict of the package's `sfd` network (vhap_amd.face_detector.SFDDetector:
    S3FD on the matrix cores); neither: the whole frame is the box.  -> {camera_id: path} or None."""
    import copy
    from .landmarks import LandmarkDetectorFA, annotate_landmarks
    if face_detector is None and face_detector_weights is not None:
        from .face_detector import SFDDetector
        face_detector = SFDDetector(face_detector_weights, device=device)
    cfg_data = copy.deepcopy(ref_cfg.data)
    cfg_data.use_landmark = False
    dataset = open_reference_dataset(cfg_data, checkout, batchify_all_views=False)
    source = ref_cfg.data.landmark_source
    if source == "face-alignment":
        if ref_cfg.exp.reuse_landmarks and dataset.get_property_path("landmark2d/face-alignment", -1).exists():
            return None
        return annotate_landmarks(dataset, LandmarkDetectorFA(weights, face_detector=face_detector, device=device))
    if source == "star":
        raise NotImplementedError("the STAR detector (vhap/util/landmark_detector_star.py) is not built: provide landmark2d/STAR.npz")
    raise ValueError(f"Unknown landmark source: {source}")
