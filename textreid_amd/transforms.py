"""Device-side image input pipeline behind the reference's ``build_transforms(cfg, is_train)`` surface
(``lib/data/transforms.py:4-43``).

The reference runs Resize -> RandomHorizontalFlip -> [Pad -> RandomCrop] -> ToTensor -> Normalize ->
[RandomErasing] per sample on PIL images inside DataLoader worker processes.  Here a ``BatchTransform`` takes the
whole batch of raw uint8 HWC images (numpy arrays or PIL images), uploads the bytes once and runs two HIP kernels
(``csrc/transforms.hip``); the output is the fp32 NCHW batch the image encoder reads.  The resize is Pillow's
antialiased BILINEAR bit for bit (the weight tables below are Resample.c's ``precompute_coeffs`` +
``normalize_coeffs_8bpc``, vectorised).  The random draws (flip, crop offset, erase rectangle) follow
torchvision's published sampling rules on a numpy generator - the reference's torch RNG stream is not reproducible
without torchvision - and can be passed in explicitly (``params=``), which is what the parity tests do.
"""

import ctypes
import math

import numpy as np
import torch

from . import ops

_BITS = 22


def resample_tables(in_size, out_size):
    """(bounds [out, 2] int32, weights [out, ksize] int32) of Pillow's BILINEAR filter for one axis."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = fscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast truncates; values >= 0 here
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    k = np.arange(ksize, dtype=np.float64)[None, :]
    t = np.abs((k + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale))
    w = np.where((t < 1.0) & (k < xmax[:, None]), 1.0 - t, 0.0)
    ww = w.sum(axis=1, keepdims=True)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    kk = np.where(w < 0, -0.5 + w * (1 << _BITS), 0.5 + w * (1 << _BITS)).astype(np.int64).astype(np.int32)  # C (int) truncation
    return np.stack([xmin, xmax], axis=1).astype(np.int32), kk


def sample_params(n, height, width, padding, use_aug, rng, erase_p=0.5, scale=(0.02, 0.4), ratio=(0.3, 3.3)):
    """[n, 8] int32 {flip, crop_top, crop_left, erase_i, erase_j, erase_h, erase_w, 0} drawn with torchvision's rules:
    RandomHorizontalFlip(0.5); RandomCrop offsets uniform in [0, 2*padding]; RandomErasing: with probability 0.5 up to
    10 attempts of (area * U(scale), exp(U(log ratio))) until the rectangle fits."""
    p = np.zeros((n, 8), dtype=np.int32)
    p[:, 0] = rng.random(n) < 0.5
    if use_aug:
        p[:, 1] = rng.integers(0, 2 * padding + 1, n)
        p[:, 2] = rng.integers(0, 2 * padding + 1, n)
        area = height * width
        for b in range(n):
            if rng.random() >= erase_p:
                continue
            for _ in range(10):
                ea = area * rng.uniform(scale[0], scale[1])
                ar = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
                eh, ew = int(round(math.sqrt(ea * ar))), int(round(math.sqrt(ea / ar)))
                if eh < height and ew < width:
                    p[b, 3] = rng.integers(0, height - eh + 1)
                    p[b, 4] = rng.integers(0, width - ew + 1)
                    p[b, 5], p[b, 6] = eh, ew
                    break
    return p


class BatchTransform:
    """``transform(images) -> float32 CUDA tensor [B, 3, H, W]``; ``images``: sequence of uint8 [h, w, 3] arrays / PIL images."""

    def __init__(self, height, width, mean, std, is_train=True, use_aug=False, padding=10, seed=0, device=None):
        self.height, self.width = int(height), int(width)
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.is_train, self.use_aug, self.padding = is_train, bool(use_aug and is_train), int(padding)
        self.rng = np.random.default_rng(seed)
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self._tables = {}

    def _axis(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            self._tables[key] = resample_tables(n_in, n_out)
        return self._tables[key]

    def __call__(self, images, params=None):
        if self.device.type != "cuda":
            raise RuntimeError("textreid_amd.transforms runs on the HIP kernel library only (CUDA device); no CPU fallback")
        arrs = [np.ascontiguousarray(np.asarray(im, dtype=np.uint8)) for im in images]
        B, H, W = len(arrs), self.height, self.width
        for a in arrs:
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("images must be uint8 [h, w, 3]")
        if params is None:
            if self.is_train:
                params = sample_params(B, H, W, self.padding, self.use_aug, self.rng)
            else:
                params = np.zeros((B, 8), dtype=np.int32)
        pad = self.padding if self.use_aug else 0
        xt = [self._axis(a.shape[1], W) for a in arrs]
        yt = [self._axis(a.shape[0], H) for a in arrs]
        KX, KY = max(t[1].shape[1] for t in xt), max(t[1].shape[1] for t in yt)
        xb, yb = np.stack([t[0] for t in xt]), np.stack([t[0] for t in yt])
        xk = np.zeros((B, W, KX), dtype=np.int32)
        yk = np.zeros((B, H, KY), dtype=np.int32)
        for b in range(B):
            xk[b, :, : xt[b][1].shape[1]] = xt[b][1]
            yk[b, :, : yt[b][1].shape[1]] = yt[b][1]
        sizes = np.array([a.size for a in arrs], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        hw = np.array([a.shape[:2] for a in arrs], dtype=np.int32)
        dev = self.device
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev, non_blocking=True)
        src = up(np.concatenate([a.reshape(-1) for a in arrs]))
        t_off, t_hw, t_xb, t_xk, t_yb, t_yk, t_p = (up(x) for x in (offs, hw, xb, xk, yb, yk, np.asarray(params, dtype=np.int32)))
        maxh = int(hw[:, 0].max())
        ws = torch.empty(ops.L.load().trid_image_pipeline_ws_bytes(B, maxh, W), dtype=torch.uint8, device=dev)
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        consts = (ctypes.c_float * 9)(*(self.mean + self.std + self.mean))  # erase value = PIXEL_MEAN (transforms.py:24)
        ops.call("trid_image_pipeline_u8", ops._p(src), ops._p(t_off), ops._p(t_hw), ops._p(t_xb), ops._p(t_xk), ops._p(t_yb),
                 ops._p(t_yk), ops._p(t_p), B, H, W, KX, KY, maxh, pad, ctypes.addressof(consts), ops._p(ws), ops._p(out),
                 ops.stream())
        return out


def build_transforms(cfg, is_train=True, device=None, seed=0):
    """Reference signature ``build_transforms(cfg, is_train)`` (transforms.py:4); returns a BATCH transform."""
    return BatchTransform(cfg.INPUT.HEIGHT, cfg.INPUT.WIDTH, cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD, is_train=is_train,
                          use_aug=cfg.INPUT.USE_AUG, padding=cfg.INPUT.PADDING, seed=seed, device=device)
