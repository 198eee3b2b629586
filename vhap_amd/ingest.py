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
from .ops import _p, _stream

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
        # (pointers through ops._p: a step capture records which buffers each of its calls touches -- _lib.AccessLog -- and a call that
        # hands over raw addresses leaves its node UNKNOWN, which keeps a plan from deferring its join)
        _lib.check(_lib.lib().vhap_frame_ingest(
            _p(self.rgb), _p(self.alpha), _p(idx), N, B, H, W, BG_MODES[self.background_color], _p(out), _p(alpha_out), _p(self._bad),
            _stream()), "vhap_frame_ingest")
        if check and int(self._bad.item()):
            self._bad.zero_()
            raise IndexError("timestep index out of range")
        return out, alpha_out

    # ---- preparation of decoded frames: colour correction + scale factor, once per sequence (round 5) ----
    @classmethod
    def from_decoded(cls, rgb_u8, alpha_u8=None, background_color=None, device="cuda", camera_index=None, color_correction=None,
                     scale_factor=1.0, n_downsample_rgb=None):
        """The store the reference's dataset pipeline would fill: the decoder's frames after NeRSembleDataset.apply_color_correction
        (nersemble_dataset.py:160-171) and VideoDataset.apply_scale_factor (video_dataset.py:266-300), in that order, applied ON THE
        DEVICE ONCE -- the reference repeats them on a DataLoader worker every time an image is fetched.  Compositing / conversion stay
        per batch (`batch`).
            camera_index [N] int: the camera of each frame, an index into color_correction [n_cam,4,4] (or [n_cam,3,4]); both None = off
            scale_factor <= 1: rgb -> (int(H * s), int(W * s)) by Pillow's bilinear resampling; the alpha map follows (resized to the
                rgb's new size) iff scale_factor / n_downsample_rgb < 1, exactly as the reference (the alpha maps of a pre-downsampled
                NeRSemble sequence are stored at full size)
        Returns the store; `store.prepared` = dict(scale_factor=<effective factor>, image_size=(h, w)) for `scale_item_properties`."""
        if scale_factor > 1.0:
            raise AssertionError("scale_factor <= 1.0")                          # video_dataset.py:267
        rgb = torch.as_tensor(rgb_u8).to(device=device, dtype=torch.uint8).contiguous()
        if rgb.dim() != 4 or rgb.shape[-1] != 3:
            raise ValueError("rgb_u8 must be [N,H,W,3] uint8")
        if not rgb.is_cuda:
            raise RuntimeError("FrameStore.from_decoded needs the HIP library and a GPU (there is no CPU path)")
        alpha = None if alpha_u8 is None else torch.as_tensor(alpha_u8).to(device=device, dtype=torch.uint8).contiguous()
        N, H, W, _ = rgb.shape
        L = _lib.lib()
        if color_correction is not None:
            A = np.asarray(color_correction, np.float64)
            if A.ndim == 2:
                A = A[None]
            if A.ndim != 3 or A.shape[1] < 3 or A.shape[2] != 4:
                raise ValueError("color_correction must be [n_cam, 4, 4] or [n_cam, 3, 4]")
            ccm = torch.from_numpy(np.ascontiguousarray(A[:, :3, :]).reshape(-1, 12)).to(device)
            cam = None
            if camera_index is not None:
                cam = torch.as_tensor(np.asarray(camera_index)).to(device=device, dtype=torch.int32).contiguous()
                if cam.numel() != N or int(cam.min()) < 0 or int(cam.max()) >= A.shape[0]:
                    raise ValueError("camera_index must hold one valid camera per frame")
            elif A.shape[0] != 1:
                raise ValueError("camera_index is required with more than one colour transform")
            corrected = torch.empty_like(rgb)                              # (never in place: `rgb` may be the caller's own device tensor)
            _lib.check(L.vhap_frame_color_correct(rgb.data_ptr(), 0 if cam is None else cam.data_ptr(), ccm.data_ptr(), A.shape[0], N, H, W,
                                                  corrected.data_ptr(), _stream()), "vhap_frame_color_correct")
            rgb = corrected
        h, w = int(H * scale_factor), int(W * scale_factor)
        if (h, w) != (H, W):
            rgb = cls._resize(rgb, h, w)
        eff = scale_factor / (n_downsample_rgb if n_downsample_rgb else 1)
        if alpha is not None and eff < 1.0 and tuple(alpha.shape[1:3]) != (h, w):
            alpha = cls._resize(alpha[..., None], h, w)[..., 0].contiguous()
        st = cls(rgb, alpha, background_color, device=device)
        st.prepared = {"scale_factor": eff, "image_size": (h, w)}
        return st

    @staticmethod
    def _resize(img, h, w):
        """[N,H,W,C] uint8 on the device -> [N,h,w,C]: PIL's Image.resize((w, h), BILINEAR), bit for bit (vhap_frame_resize_u8)."""
        N, H, W, C = img.shape
        dev = img.device
        tabs = []
        for n_in, n_out in ((W, w), (H, h)):
            if n_in == n_out:
                tabs += [None, None, 0]
            else:
                b, k, ks = pil_bilinear_coeffs(n_in, n_out)
                tabs += [torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks]
        out = torch.empty(N, h, w, C, dtype=torch.uint8, device=dev)
        p = lambda t: 0 if t is None else t.data_ptr()
        _lib.check(_lib.lib().vhap_frame_resize_u8(img.data_ptr(), N, H, W, C, out.data_ptr(), h, w, p(tabs[0]), p(tabs[1]), tabs[2],
                                                   p(tabs[3]), p(tabs[4]), tabs[5], _stream()), "vhap_frame_resize_u8")
        return out


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's resampling coefficients for Image.resize(..., BILINEAR) of an 8-bit image along one axis over the whole axis (box 0 ..
    in_size): src/libImaging/Resample.c precompute_coeffs() + normalize_coeffs_8bpc(), statement by statement in float64 -- the filter's
    support grows with the downscale factor (antialiasing), weights are normalised, then rounded half away from zero to 22-bit fixed
    point.  -> (bounds [2 * out] int32: first tap, tap count; coef [out * ksize] int32; ksize)."""
    import math
    scale = filterscale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros(2 * out_size, np.int32)
    coef = np.zeros(out_size * ksize, np.int32)
    ss = 1.0 / filterscale
    one = float(1 << 22)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            wgt = 1.0 - v if v < 1.0 else 0.0
            k.append(wgt)
            ww += wgt
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            coef[xx * ksize + x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[2 * xx], bounds[2 * xx + 1] = xmin, xmax
    return bounds, coef, ksize


def scale_item_properties(prepared, lmk2d=None, intrinsic=None):
    """The size-dependent item properties of VideoDataset.apply_scale_factor (video_dataset.py:276-292): normalised landmarks times the
    NEW image size (always), the first two rows of the intrinsics times the effective factor when it is < 1.  numpy in, numpy out."""
    h, w = prepared["image_size"]
    out = {}
    if lmk2d is not None:
        lm = np.array(lmk2d, copy=True)
        lm[..., 0] *= w
        lm[..., 1] *= h
        out["lmk2d"] = lm
    if intrinsic is not None:
        K = np.array(intrinsic, copy=True)
        if prepared["scale_factor"] < 1.0:
            K[..., :2, :] *= prepared["scale_factor"]
        out["intrinsic"] = K
    return out
