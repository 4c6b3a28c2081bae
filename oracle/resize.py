"""Bicubic resize of 8-bit RGB images as the reference's image preparation performs it -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Where the reference does it: `dynamic_preprocess` (Vlaser_VLM/internvl_chat/internvl/train/dataset.py:830-866, `image.resize((target_width, target_height))` :849 and
the thumbnail :864; eval_example.py:38-74) and `build_transform` (dataset.py:276-310, `T.Resize((448, 448), interpolation=BICUBIC)` :294 -- torchvision hands a PIL
image to `Image.resize`).  The arithmetic therefore lives in a third-party dependency that is not under /root/reference: **Pillow** (`pillow==11.2.1`,
Vlaser_VLA/Simpler/requirements.txt:165), `Image.resize(size)` -> `ImagingResample` (src/libImaging/Resample.c), default filter BICUBIC.  Restated here from its
published algorithm:

  * per axis, `precompute_coeffs`: scale = in / out, filterscale = max(scale, 1), support = 2 * filterscale (bicubic, a = -0.5), for every output index the window
    [xmin, xmin + n) = round(center -+ support) clipped to the image, weights w(( x + xmin - center + 0.5) / filterscale) normalised by their sum -- all in doubles;
  * `normalize_coeffs_8bpc`: weights to fixed point, 22 fractional bits, rounded half away from zero;
  * two passes, horizontal then vertical, each `clip8((2^21 + sum pixel * k) >> 22)` in int32 -- the intermediate image is ROUNDED TO 8 BITS between the passes;
  * a pass whose size does not change is skipped; same size both ways = a copy.

Pinning: Pillow itself is installed in this image (here and on the GPU box: 12.2.0; the resampler has not changed since the 8-bit fixed-point path of 3.x), so
tests/test_resize.py compares this restatement with `PIL.Image.resize` directly on seeded images over up- and down-scales, and with the committed fixture
tests/golden/g12_resize.npz written by tools/gen_golden_resize.py from Pillow (bit-exact: byte work).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support_1=2.0, filt=bicubic_filter):
    """Resample.c precompute_coeffs over the whole axis (box = (0, in_size)): (ksize, bounds [out, 2] = (xmin, n), kk [out, ksize] doubles)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = support_1 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def normalize_coeffs_8bpc(kk):
    """Doubles -> int32 fixed point, (int)(+-0.5 + k * 2^22) with C truncation toward zero."""
    v = kk * float(1 << PRECISION_BITS)
    return np.where(kk < 0, np.trunc(-0.5 + v), np.trunc(0.5 + v)).astype(np.int32)


def coeffs_8bpc(in_size, out_size):
    ksize, bounds, kk = precompute_coeffs(in_size, out_size)
    return ksize, bounds, normalize_coeffs_8bpc(kk)


def _pass(img, out_size, axis):
    """One resampling pass along `axis` (0 rows / 1 columns) of an [H, W, C] uint8 image."""
    in_size = img.shape[axis]
    _, bounds, kk = coeffs_8bpc(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)                # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        # int32 arithmetic in C: |sum| <= 255 * sum|k| stays far inside int32 for a bicubic kernel; >> is arithmetic
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, out_w, out_h):
    """`PIL.Image.resize((out_w, out_h))` (BICUBIC, whole-image box, no reducing gap) of an [H, W, C] uint8 array."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W = img.shape[:2]
    if (out_w, out_h) == (W, H):
        return img.copy()
    cur = img
    if out_w != W:
        cur = _pass(cur, out_w, 1)             # horizontal first (ImagingResample: need_horizontal, then need_vertical)
    if out_h != H:
        cur = _pass(cur, out_h, 0)
    return np.ascontiguousarray(cur)


def load_image_u8(img, input_size=448, max_num=12, grid=None):
    """eval_example.py:76-82 `load_image` up to (not including) ToTensor / Normalize: the uint8 tiles [n, input_size, input_size, 3] of dynamic_preprocess
    (use_thumbnail=True) after build_transform's resize (an identity on tiles that already have the target size).  `grid` = (cols, rows) from
    dynamic_preprocess's aspect-ratio search (host integer logic, vlaser_amd.prep.dynamic_grid restates it and tests/test_prep.py pins it)."""
    cols, rows = grid
    big = resize_bicubic_u8(img, input_size * cols, input_size * rows)
    tiles = [big[(i // cols) * input_size:(i // cols + 1) * input_size, (i % cols) * input_size:(i % cols + 1) * input_size] for i in range(cols * rows)]
    if len(tiles) != 1:
        tiles.append(resize_bicubic_u8(img, input_size, input_size))
    return np.stack(tiles)
