"""TEST INFRASTRUCTURE — CPU restatement of ``OptimizedFrameExtractor._is_low_quality``
(reference src/core/frame_extractor.py:301-316).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

PARITY UNPINNED: the reference computes the grey image and the Laplacian with OpenCV (``cv2.cvtColor(frame,
COLOR_BGR2GRAY)``, ``cv2.Laplacian(gray, CV_64F)``), an unvendored dependency (requirements.txt
``opencv-python``, unpinned) that is not installed in the build container, and the reference's tests hold no
vector for it.  This restates OpenCV 4.x's published algorithms:
  * BGR2GRAY on uint8 (modules/imgproc/src/color_rgb.simd.hpp, RGB2Gray<uchar>): 15-bit fixed point,
    ``(B*3735 + G*19235 + R*9798 + (1 << 14)) >> 15``;
  * Laplacian with the default aperture 1 (modules/imgproc/src/deriv.cpp): the kernel [[0,1,0],[1,-4,1],[0,1,0]],
    border BORDER_REFLECT_101, in float64.
``np.mean(frame)`` and ``.var()`` are numpy and are used as such.
"""
import numpy as np


def bgr_to_gray(frame: np.ndarray) -> np.ndarray:
    f = frame.astype(np.int64)
    return ((f[..., 0] * 3735 + f[..., 1] * 19235 + f[..., 2] * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def laplacian_f64(gray: np.ndarray) -> np.ndarray:
    g = gray.astype(np.float64)
    h, w = g.shape
    ys, xs = np.arange(h), np.arange(w)
    up, down = g[_reflect101(ys - 1, h)], g[_reflect101(ys + 1, h)]
    left, right = g[:, _reflect101(xs - 1, w)], g[:, _reflect101(xs + 1, w)]
    return up + down + left + right - 4.0 * g


def quality(frame: np.ndarray):
    """→ (mean_brightness, laplacian_var) as the reference computes them (:305, :311)."""
    return float(np.mean(frame)), float(laplacian_f64(bgr_to_gray(frame)).var())


def is_low_quality(frame: np.ndarray) -> bool:
    mean_brightness, laplacian_var = quality(frame)
    if mean_brightness < 20 or mean_brightness > 235:
        return True
    return laplacian_var < 100
