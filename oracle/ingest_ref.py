"""CPU restatement of the reference's per-image ingest transforms -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

Follows vhap/data/video_dataset.py:
  :302-323 apply_background_color   w = alpha[..., None] / 255 (fp64); img = (w * fg + (1 - w) * bg).astype(uint8), bg = 255 | 0
  :261-268 apply_to_tensor          torchvision to_tensor: HWC uint8 -> CHW float32 / 255 (2-D alpha gets a leading channel)
Pinned: tests/golden/ingest_golden.npz was produced by the reference's own methods (tools/make_golden_ingest.py), including the
exhaustive 256 x 256 (alpha, fg) table for both colours; tests/test_ingest.py checks this file against it.
"""
import numpy as np


def apply_background_color(rgb_u8, alpha_u8, background_color):
    """rgb_u8 [...,H,W,3] uint8, alpha_u8 [...,H,W] uint8 -> uint8 [...,H,W,3]"""
    if background_color is None:
        return rgb_u8
    if background_color == "white":
        bg = np.full(rgb_u8.shape, 255, dtype=np.uint8)
    elif background_color == "black":
        bg = np.zeros_like(rgb_u8)
    else:
        raise NotImplementedError(f"Unknown background color: {background_color}.")
    w = alpha_u8[..., None] / 255
    return (w * rgb_u8 + (1 - w) * bg).astype(np.uint8)


def to_tensor(img_u8):
    """uint8 [N,H,W,C] or [N,H,W] -> float32 [N,C,H,W]"""
    if img_u8.ndim == 3:
        img_u8 = img_u8[..., None]
    return np.ascontiguousarray(img_u8.transpose(0, 3, 1, 2)).astype(np.float32) / np.float32(255)


def frame_ingest(rgb_u8, alpha_u8, index, background_color):
    """The batch the fit consumes: (rgb [B,3,H,W] fp32, alpha [B,1,H,W] fp32 or None)."""
    idx = np.arange(len(rgb_u8)) if index is None else np.asarray(index)
    rgb = rgb_u8[idx]
    alpha = None if alpha_u8 is None else alpha_u8[idx]
    rgb = apply_background_color(rgb, alpha, background_color)
    return to_tensor(rgb), (None if alpha is None else to_tensor(alpha))


# ---- colour correction + scale factor (C restatements: oracle/ingest_oracle.c) ------------------------------------------------------
def apply_color_correction(rgb_u8, A):
    """NeRSembleDataset.apply_color_correction (nersemble_dataset.py:165-171) of ONE camera's image(s): rgb_u8 [...,3] uint8, A the camera's
    affine colour transform ([4,4] as stored, or its top [3,4]) -> uint8, same shape."""
    import ctypes
    import oracle
    rgb = np.ascontiguousarray(rgb_u8, np.uint8)
    A = np.ascontiguousarray(A, np.float64)
    out = np.empty_like(rgb)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    oracle.lib().oracle_color_correct_u8(rgb.ctypes.data_as(u8), rgb.size // 3, A.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), A.shape[1],
                                         out.ctypes.data_as(u8))
    return out


def pil_resize(img_u8, h, w):
    """Image.fromarray(img).resize((w, h), resample=Image.BILINEAR) for an 8-bit [H,W,C] or [H,W] image (video_dataset.py:271-274, 293-296)."""
    import ctypes
    import oracle
    img = np.ascontiguousarray(img_u8, np.uint8)
    H, W = img.shape[:2]
    C = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((h, w) + img.shape[2:], np.uint8)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    rc = oracle.lib().oracle_pil_resize_u8(img.ctypes.data_as(u8), H, W, C, out.ctypes.data_as(u8), h, w)
    assert rc == 0
    return out


def apply_scale_factor(rgb_u8, alpha_u8, scale_factor, n_downsample_rgb=None, lmk2d=None, intrinsic=None):
    """VideoDataset.apply_scale_factor (video_dataset.py:266-300) of one image: -> dict(rgb, alpha_map, lmk2d, intrinsic, scale_factor).
    Landmarks come normalised and are multiplied by the NEW size; the intrinsics' first two rows and the alpha map follow the effective
    factor scale_factor / n_downsample_rgb only when that is < 1."""
    assert scale_factor <= 1.0
    H, W, _ = rgb_u8.shape
    h, w = int(H * scale_factor), int(W * scale_factor)
    out = {"rgb": pil_resize(rgb_u8, h, w)}
    if lmk2d is not None:
        lm = np.array(lmk2d, copy=True)
        lm[..., 0] *= w
        lm[..., 1] *= h
        out["lmk2d"] = lm
    eff = scale_factor / (n_downsample_rgb if n_downsample_rgb else 1)
    out["scale_factor"] = eff
    K, a = intrinsic, alpha_u8
    if eff < 1.0:
        if K is not None:
            K = np.array(K, copy=True)
            K[:2] *= eff
        if a is not None:
            a = pil_resize(a, h, w)
    out["intrinsic"], out["alpha_map"] = K, a
    return out
