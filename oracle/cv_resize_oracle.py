"""TEST INFRASTRUCTURE — CPU restatement of ``cv2.resize(frame, (w, h))`` for uint8 BGR frames, the resize
``OptimizedFrameExtractor.extract_frames`` applies (reference src/core/frame_extractor.py:283-284; default
interpolation INTER_LINEAR).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED: OpenCV (requirements.txt ``opencv-python``, unpinned) is not installed in the build container
and the reference's tests hold no vector for it.  This restates the published algorithm of OpenCV 4.x
modules/imgproc/src/resize.cpp for 8-bit input:
  * coordinates: ``fx = (float)((dx + 0.5) * scale_x - 0.5)``, ``sx = floor(fx)``, ``fx -= sx`` with
    ``scale_x = src_w / dst_w`` in double; left/right borders snap to the edge pixel with weight 1;
  * weights: 11-bit fixed point, ``saturate_cast<short>(w * 2048)`` (round half to even);
  * horizontal pass into 32-bit integers ``S[sx]*a0 + S[sx+1]*a1``; vertical pass
    ``(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2`` (VResizeLinear for uchar), rows clipped to
    the image;
  * an exact 2x2 down-scale is routed to INTER_AREA's fast path ``(a + b + c + d + 2) >> 2``; equal sizes copy.
No antialiasing: a large down-scale reads four source pixels per output pixel.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _round_short(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v), -32768, 32767).astype(np.int64)     # cvRound = round half to even, then saturate


def linear_coeffs(src: int, dst: int, snap_borders: bool):
    """→ (ofs int64[dst], w0, w1 int64[dst]).  `snap_borders`: the horizontal rule (fx = 0 at the image edges);
    rows are clipped later instead (resizeGeneric_Invoker)."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if snap_borders:
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= src - 1
        f[hi], s[hi] = 0.0, src - 1
    w0 = _round_short((np.float32(1.0) - f) * np.float32(COEF_SCALE))
    w1 = _round_short(f * np.float32(COEF_SCALE))
    return s, w0, w1


def resize_linear_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``cv2.resize(img, (out_w, out_h))`` for a uint8 [h, w, c] array."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    if (out_w, out_h) == (w, h):
        return img.copy()
    if w == 2 * out_w and h == 2 * out_h:                        # INTER_LINEAR -> INTER_AREA fast path
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = linear_coeffs(w, out_w, True)
    sy, b0, b1 = linear_coeffs(h, out_h, False)
    src = img.astype(np.int64)
    x1 = np.minimum(sx + 1, w - 1)
    rows = src[:, sx] * a0[None, :, None] + src[:, x1] * a1[None, :, None]      # [h, out_w, c] 32-bit sums
    y0 = np.clip(sy, 0, h - 1)
    y1 = np.clip(sy + 1, 0, h - 1)
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)
