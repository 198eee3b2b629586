"""Device-resident uint8 frame store.

The reference decodes each image on DataLoader workers, composites it over the background colour and converts it to an fp32 CHW
tensor on the host (vhap/data/video_dataset.py:209-323), then uploads fp32 batches (tracker.py:1352-1357).  Here a whole sequence
stays in HBM as the decoder's uint8 HWC arrays -- a 512 x 512 frame with alpha is 1 MiB instead of 4 MiB, so 288 GB holds ~270k
frames -- and `batch()` produces the fp32 batch with ONE launch (`vhap_frame_ingest`: gather by timestep, composite, convert),
bit-identical to the host transforms.  `batch(out=...)` writes straight into a captured step's static sample buffers.
"""
import numpy as np
import torch

from . import _lib
from .ops import _stream

BG_MODES = {None: 0, "white": 1, "black": 2}


class FrameStore:
    def __init__(self, rgb_u8, alpha_u8=None, background_color=None, device="cuda"):
        if background_color not in BG_MODES:
            raise NotImplementedError(f"Unknown background color: {background_color}.")      # video_dataset.py:315-318
        if background_color is not None and alpha_u8 is None:
            raise AssertionError("'alpha_map' is required to apply background color.")         # video_dataset.py:304-306
        self.rgb = torch.as_tensor(rgb_u8).to(device=device, dtype=torch.uint8).contiguous()
        if self.rgb.dim() != 4 or self.rgb.shape[-1] != 3:
            raise ValueError("rgb_u8 must be [N,H,W,3] uint8")
        self.alpha = None if alpha_u8 is None else torch.as_tensor(alpha_u8).to(device=device, dtype=torch.uint8).contiguous()
        if self.alpha is not None and tuple(self.alpha.shape) != tuple(self.rgb.shape[:3]):
            raise ValueError("alpha_u8 must be [N,H,W] uint8")
        self.background_color = background_color
        self._bad = torch.zeros(1, dtype=torch.int32, device=device)

    def __len__(self):
        return self.rgb.shape[0]

    @property
    def image_size(self):
        return tuple(self.rgb.shape[1:3])

    def batch(self, index=None, out=None, alpha_out=None, check=False):
        """index: None (all frames), or int64 timesteps (numpy / list / device tensor).  Returns (rgb [B,3,H,W], alpha [B,1,H,W] | None)."""
        if not self.rgb.is_cuda:
            raise RuntimeError("FrameStore.batch needs the HIP library and a GPU (there is no CPU path)")
        N, H, W, _ = self.rgb.shape
        idx = None
        if index is not None:
            idx = index if torch.is_tensor(index) else torch.as_tensor(np.asarray(index))
            idx = idx.to(device=self.rgb.device, dtype=torch.int64).contiguous()
        B = N if idx is None else idx.numel()
        if out is None:
            out = torch.empty(B, 3, H, W, dtype=torch.float32, device=self.rgb.device)
        if self.alpha is not None and alpha_out is None:
            alpha_out = torch.empty(B, 1, H, W, dtype=torch.float32, device=self.rgb.device)
        assert out.is_contiguous() and tuple(out.shape) == (B, 3, H, W) and out.dtype == torch.float32
        assert alpha_out is None or (alpha_out.is_contiguous() and tuple(alpha_out.shape) == (B, 1, H, W))
        _lib.check(_lib.lib().vhap_frame_ingest(
            self.rgb.data_ptr(), 0 if self.alpha is None else self.alpha.data_ptr(), 0 if idx is None else idx.data_ptr(), N, B, H, W,
            BG_MODES[self.background_color], out.data_ptr(), 0 if alpha_out is None else alpha_out.data_ptr(), self._bad.data_ptr(),
            _stream()), "vhap_frame_ingest")
        if check and int(self._bad.item()):
            self._bad.zero_()
            raise IndexError("timestep index out of range")
        return out, alpha_out
