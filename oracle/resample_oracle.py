"""TEST INFRASTRUCTURE — CPU restatement of the image resize that sits in front of the encoder.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference resizes with Pillow in two places:
  * ``transforms.Resize((224, 224))`` on a PIL image (reference src/core/feature_extractor.py:54-61, :105-116)
    = ``Image.resize((224, 224), BILINEAR)`` — a stretch to the square, antialiased when shrinking;
  * the live path's ``CLIPProcessor`` (reference video_search_overhaul.py:129-135, :218-221, :283-289) =
    transformers' CLIP image processor: shortest edge → 224 with BICUBIC, centre crop 224x224
    (transformers image_processing_backends.py PilBackend.resize/center_crop, image_transforms.py:246-299).
The arithmetic is Pillow's, an unvendored third-party dependency (requirements.txt lists ``Pillow`` unpinned;
12.2.0 is installed in the build container): src/libImaging/Resample.c — ``precompute_coeffs``,
``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc`` / ``Vertical_8bpc`` and ``ImagingResampleInner``
(horizontal pass first, into a uint8 intermediate, then vertical).  This file restates that published
algorithm in numpy; tests/test_oracle_golden.py pins it bit-for-bit against Pillow itself (run in the build
container) and against the committed outputs in tests/golden/resample_pil.npz.

Bit-exact integer work: every function returns uint8 arrays equal to Pillow's.
"""
import math

import numpy as np

BILINEAR, BICUBIC = 2, 3            # PIL.Image.Resampling values
PRECISION_BITS = 32 - 8 - 2         # Resample.c: coefficients are 22-bit fixed point


def _bilinear(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0)}


def precompute_coeffs(in_size, in0, in1, out_size, filt):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: → (bounds int32[out,2] = (first tap, tap count),
    kk int32[out, ksize] fixed-point weights).  All intermediate arithmetic in IEEE double, as in C."""
    fn, fsupport = _FILTERS[filt]
    scale = float(in1 - in0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One separable pass over `axis` (0 = vertical, 1 = horizontal) of a uint8 [h, w, c] image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)                   # [len, other, c]
    n_in = src.shape[0]
    taps = np.minimum(bounds[:, :1] + np.arange(kk.shape[1])[None, :], n_in - 1)   # weights past the count are 0
    acc = np.full((bounds.shape[0],) + src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
    for t in range(kk.shape[1]):
        acc += src[taps[:, t]] * kk[:, t].astype(np.int64)[:, None, None]
    out = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)      # clip8()
    return np.moveaxis(out, 0, axis)


def resize_u8(img, out_w, out_h, filt):
    """``Image.fromarray(img).resize((out_w, out_h), filt)`` for a uint8 [h, w, c] array
    (ImagingResampleInner with the full-image box)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    need_h, need_v = out_w != w, out_h != h
    bh, kh = precompute_coeffs(w, 0.0, float(w), out_w, filt)
    bv, kv = precompute_coeffs(h, 0.0, float(h), out_h, filt)
    if need_h:
        first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])        # source rows the vertical pass touches
        img = _pass(img[first:last], bh, kh, 1)
        bv = bv.copy()
        bv[:, 0] -= first
    if need_v:
        img = _pass(img, bv, kv, 0)
    return img


def stretch_to_square(img, size=224):
    """E1's ``transforms.Resize((S, S))`` on a PIL image (reference feature_extractor.py:55)."""
    return resize_u8(img, size, size, BILINEAR)


def clip_processor_geometry(h, w, size=224, crop=224):
    """(resized_h, resized_w, crop_top, crop_left) of the CLIP image processor: shortest edge → `size`
    (``int(size * long / short)`` for the other edge), centre crop."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_long = int(size * long_ / short)
    rh, rw = (new_long, size) if w <= h else (size, new_long)
    return rh, rw, (rh - crop) // 2, (rw - crop) // 2


def clip_processor_u8(img, size=224, crop=224):
    """uint8 [crop, crop, 3] the CLIP image processor feeds to rescale/normalise."""
    rh, rw, top, left = clip_processor_geometry(img.shape[0], img.shape[1], size, crop)
    out = resize_u8(img, rw, rh, BICUBIC)
    return np.ascontiguousarray(out[top:top + crop, left:left + crop])
