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
