"""Oracle: the image input pipeline of ``lib/data/transforms.py:4-43``.  Test infrastructure only.

The reference composes torchvision transforms on PIL images: Resize((H, W)) -> RandomHorizontalFlip(0.5) ->
[Pad(PADDING) -> RandomCrop((H, W))] -> ToTensor -> Normalize(mean, std) -> [RandomErasing(scale=(0.02, 0.4),
value=PIXEL_MEAN)].  torchvision (pinned 0.11.1, ``requirements.txt:18``) is NOT in the image, so the chain is
restated here with its published semantics; the one non-trivial arithmetic step - ``Resize`` on a PIL image =
``Image.resize(size, BILINEAR)``, Pillow's antialiased two-pass fixed-point resampling (libImaging/Resample.c) - is
restated in ``resample_coeffs`` / ``resize_bilinear_u8`` and PINNED bit-exactly against Pillow itself (present in
the image) by tests/test_host_cpu.py.  The random draws are inputs (``params``): the reference's RNG stream is
torchvision's and cannot be reproduced without it.
"""

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c


def resample_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs for the BILINEAR (triangle, support 1) filter, 8 bits per channel:
    (bounds [out, 2] = (first source index, count), kk [out, ksize] int32 fixed-point weights)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        ww = w[:xmax].sum()
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):  # normalize_coeffs_8bpc
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One resampling pass over ``axis`` of a uint8 [h, w, c] image (ImagingResampleHorizontal/Vertical_8bpc)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for i, (lo, n) in enumerate(bounds):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for k in range(n):
            acc += src[lo + k] * int(kk[i, k])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, height, width):
    """T.Resize((height, width)) on a PIL image: horizontal pass, then vertical (each rounds to uint8)."""
    h, w = img.shape[:2]
    out = img
    if w != width:
        out = _pass(out, *resample_coeffs(w, width), axis=1)
    if h != height:
        out = _pass(out, *resample_coeffs(h, height), axis=0)
    return out


def pipeline(img_u8, height, width, mean, std, flip=False, padding=0, crop=(0, 0), erase=None, erase_value=None):
    """uint8 [h, w, 3] -> float32 [3, height, width] (transforms.py:15-27 order).  crop = (top, left) in the padded
    frame, erase = (i, j, eh, ew) or None; erase_value = per-channel values written AFTER normalisation
    (the reference passes PIXEL_MEAN, transforms.py:24)."""
    x = resize_bilinear_u8(img_u8, height, width)
    if flip:
        x = x[:, ::-1]
    if padding:
        x = np.pad(x, ((padding, padding), (padding, padding), (0, 0)))  # T.Pad: constant fill 0
        x = x[crop[0] : crop[0] + height, crop[1] : crop[1] + width]
    t = x.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)  # ToTensor
    t = (t - np.asarray(mean, dtype=np.float32)[:, None, None]) / np.asarray(std, dtype=np.float32)[:, None, None]
    if erase is not None:
        i, j, eh, ew = erase
        t[:, i : i + eh, j : j + ew] = np.asarray(erase_value, dtype=np.float32)[:, None, None]
    return np.ascontiguousarray(t, dtype=np.float32)
