"""Image preparation on the device: decoded uint8 image -> normalised bf16 `pixel_values`, the work of the reference's `load_image`
(Vlaser_VLM/internvl_chat/eval/eval_example.py:38-82 = `dynamic_preprocess` dataset.py:830-866 + `build_transform` dataset.py:276-310) without the host round trip
through Pillow: the aspect-ratio grid is chosen on the host (integer logic on two numbers, `prep.dynamic_grid`), the bicubic resize to the grid, the thumbnail, the
crop into 448-px tiles and ToTensor + Normalize run in `csrc/image.hip` -- bit-exact with Pillow's 8-bit resampler (tests/test_image_gpu.py).

What stays on the host: decoding the file (JPEG / PNG -> uint8), as in the reference.  What is NOT covered: the VLA environment adapter's `cv2.resize(LANCZOS4)`
(env_adapter/simpler.py:82-92; OpenCV is not in this image, so that resampler cannot be pinned) -- PiZero.infer_action takes the already-resized uint8 observation."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L, ops, prep

BF16 = torch.bfloat16


class _Axis:
    """Device copies of Pillow's weight tables for one (in_size -> out_size) axis."""

    def __init__(self, in_size, out_size, device):
        lib = L.lib()
        self.ksize = lib.vlaser_resample_ksize(in_size, out_size)
        if self.ksize < 1:
            raise ValueError(f'resample: bad sizes {in_size} -> {out_size}')
        bounds = np.empty(2 * out_size, np.int32)
        kk = np.empty(self.ksize * out_size, np.int32)
        L.check(0 if lib.vlaser_resample_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data) == self.ksize else -1, 'vlaser_resample_coeffs')
        self.bounds = torch.from_numpy(bounds).to(device)
        self.kk = torch.from_numpy(kk).to(device)


class ImagePrep:
    """Caches the per-axis weight tables (a few KB per distinct size pair).  One instance per device / stream of use."""

    def __init__(self, device='cuda', input_size=448, mean=prep.IMAGENET_MEAN, std=prep.IMAGENET_STD):
        if not torch.cuda.is_available():
            raise L.VlaserHipError('vlaser_amd.image needs an MI355X (gfx950) GPU: there is no CPU fallback (prep.load_image is the host path through Pillow)')
        self.device = torch.device(device)
        self.input_size = input_size
        self.mean, self.std = tuple(mean), tuple(std)
        self._axes = {}

    def _axis(self, a, b):
        if a == b:
            return None
        t = self._axes.get((a, b))
        if t is None:
            t = self._axes[(a, b)] = _Axis(a, b, self.device)
        return t

    def resize(self, img, out_w, out_h, out=None):
        """`PIL.Image.resize((out_w, out_h))` (BICUBIC) of a uint8 [H, W, 3] device tensor (any row stride, unit pixel stride) -> uint8 [out_h, out_w, 3]."""
        assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.stride(2) == 1 and img.stride(1) == 3 and img.is_cuda
        H, W = img.shape[:2]
        if out is None:
            out = torch.empty(out_h, out_w, 3, dtype=torch.uint8, device=img.device)
        assert out.shape == (out_h, out_w, 3) and out.stride(2) == 1 and out.stride(1) == 3
        ax, ay = self._axis(W, out_w), self._axis(H, out_h)
        tmp, ld_tmp = None, 0
        if ax is not None and ay is not None:
            ld_tmp = (out_w * 3 + 3) & ~3
            tmp = torch.empty(H * ld_tmp, dtype=torch.uint8, device=img.device)
        p = lambda t: t.data_ptr() if t is not None else None
        L.check(L.lib().vlaser_resize_u8(img.data_ptr(), H, W, img.stride(0), p(tmp), ld_tmp, out.data_ptr(), out_h, out_w, out.stride(0),
                                         p(ax.bounds) if ax else None, p(ax.kk) if ax else None, ax.ksize if ax else 0,
                                         p(ay.bounds) if ay else None, p(ay.kk) if ay else None, ay.ksize if ay else 0, ops._stream()), 'vlaser_resize_u8')
        return out

    def tiles_normalize(self, big, cols, rows, out, mode='totensor'):
        """[rows * S, cols * S, 3] uint8 -> out[:cols * rows] bf16 [n, 3, S, S] (crop loop + ToTensor + Normalize)."""
        S = self.input_size
        assert big.dtype == torch.uint8 and big.shape == (rows * S, cols * S, 3) and big.stride(1) == 3 and out.dtype == BF16 and out.is_contiguous()
        m3 = (C.c_float * 3)(*self.mean)
        s3 = (C.c_float * 3)(*self.std)
        L.check(L.lib().vlaser_tiles_normalize_u8(big.data_ptr(), big.stride(0), cols, rows, S, out.data_ptr(), 0 if mode == 'vla' else 1, m3, s3, ops._stream()),
                'vlaser_tiles_normalize_u8')
        return out

    def load_image(self, image_u8, max_num=12, min_num=1, use_thumbnail=True):
        """eval_example.py:76-82 `load_image(image_file, input_size, max_num)` from the decoded image on: uint8 [H, W, 3] (host or device tensor / numpy array)
        -> bf16 [n_tiles, 3, S, S] on the device, n_tiles = cols * rows (+ 1 thumbnail iff more than one tile)."""
        if isinstance(image_u8, np.ndarray):
            image_u8 = torch.from_numpy(np.ascontiguousarray(image_u8))
        img = image_u8.to(self.device, non_blocking=True).contiguous()
        H, W = img.shape[:2]
        S = self.input_size
        cols, rows = prep.dynamic_grid(W, H, min_num, max_num, S)
        n = cols * rows
        thumb = use_thumbnail and n != 1
        out = torch.empty(n + (1 if thumb else 0), 3, S, S, dtype=BF16, device=self.device)
        big = self.resize(img, S * cols, S * rows)
        self.tiles_normalize(big, cols, rows, out)
        if thumb:
            self.tiles_normalize(self.resize(img, S, S), 1, 1, out[n:])
        return out


_PREPS = {}


def load_image(image_file, input_size=448, max_num=12, device='cuda'):
    """Drop-in for the reference's `load_image(image_file, input_size=448, max_num=12)` (eval_example.py:76-82): a path / file object / PIL image -> bf16 pixel_values
    [n_tiles, 3, input_size, input_size] ON THE DEVICE (the reference returns fp32 on the host and the caller does `.to(torch.bfloat16).cuda()`, eval_example.py:96).  Only the
    file decoding runs on the host."""
    from PIL import Image
    im = image_file if isinstance(image_file, Image.Image) else Image.open(image_file)
    arr = np.asarray(im.convert('RGB'), dtype=np.uint8)
    key = (str(device), input_size)
    ip = _PREPS.get(key)
    if ip is None:
        ip = _PREPS[key] = ImagePrep(device, input_size)
    return ip.load_image(arr, max_num=max_num)
