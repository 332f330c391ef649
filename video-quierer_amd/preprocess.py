"""GPU frame preprocessing in front of the encoder: Pillow-exact resize and the
frame-quality statistics of the reference's frame extractor, over ``vq_resampler_*`` / ``vq_frame_quality_u8``.

  * :meth:`FramePreprocessor.stretch` — ``transforms.Resize((S, S))`` of reference
    src/core/feature_extractor.py:54-61 (PIL bilinear, antialiased);
  * :meth:`FramePreprocessor.clip_processor` — the CLIP image processor of the live path
    (reference video_search_overhaul.py:129-135, :218-221): short edge → 224 bicubic, centre crop 224;
  * :meth:`FramePreprocessor.cv_resize` — ``cv2.resize(frame, frame_size)`` of reference
    src/core/frame_extractor.py:283-284 (OpenCV INTER_LINEAR; parity unpinned);
  * :meth:`FramePreprocessor.quality` / :meth:`is_low_quality` — reference
    src/core/frame_extractor.py:301-316.
"""
import ctypes
from ctypes import c_double, c_int, c_int64, c_void_p
from typing import Optional, Tuple

import numpy as np

from . import _lib

BILINEAR, BICUBIC = 2, 3            # PIL.Image.Resampling values (VQ_RESAMPLE_*)
CV_LINEAR = 100                     # cv2.resize's default INTER_LINEAR (VQ_RESAMPLE_CV_LINEAR)


def clip_processor_geometry(h: int, w: int, size: int = 224, crop: int = 224) -> Tuple[int, int, int, int]:
    """(resized_h, resized_w, crop_top, crop_left) of the CLIP image processor for an h x w frame."""
    rh, rw, top, left = c_int(), c_int(), c_int(), c_int()
    _lib.check(_lib.load().vq_clip_processor_geometry(int(h), int(w), int(size), int(crop), ctypes.byref(rh),
                                                      ctypes.byref(rw), ctypes.byref(top), ctypes.byref(left)))
    return rh.value, rw.value, top.value, left.value


def _frames(frames) -> np.ndarray:
    a = np.asarray(frames)
    if a.ndim == 3:
        a = a[None]
    if a.ndim != 4 or a.shape[3] != 3:
        raise ValueError(f"expected uint8 frames [n, h, w, 3], got shape {a.shape}")
    if a.dtype != np.uint8:
        raise TypeError(f"expected uint8 pixels, got {a.dtype}")
    return np.ascontiguousarray(a)


class FramePreprocessor:
    def __init__(self, device: Optional[int] = None):
        self.device = _lib.init(device)
        h = c_void_p()
        _lib.check(_lib.load().vq_resampler_create(ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().vq_resampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass

    def set_stream(self, stream_ptr: Optional[int]):
        _lib.check(_lib.load().vq_resampler_set_stream(self._h, c_void_p(stream_ptr or 0)))

    def synchronize(self):
        _lib.check(_lib.load().vq_resampler_synchronize(self._h))

    # -- resize ---------------------------------------------------------------
    def resize(self, frames, out_h: int, out_w: int, filter: int = BILINEAR,
               crop: Optional[Tuple[int, int, int, int]] = None, keep_on_device: bool = False):
        """``Image.resize((out_w, out_h), filter)`` of every frame, then the window
        ``crop = (top, left, h, w)`` (default: the whole resized frame).  → uint8 [n, crop_h, crop_w, 3], or with
        ``keep_on_device`` the device address of that array (valid until the next call on this object)."""
        a = _frames(frames)
        n, h, w = a.shape[:3]
        top, left, ch, cw = crop if crop is not None else (0, 0, out_h, out_w)
        out = None if keep_on_device else np.empty((n, ch, cw, 3), dtype=np.uint8)
        _lib.check(_lib.load().vq_resampler_run_u8(self._h, a.ctypes.data_as(c_void_p), n, h, w, int(filter), int(out_h),
                                                   int(out_w), int(top), int(left), int(ch), int(cw),
                                                   out.ctypes.data_as(c_void_p) if out is not None else None))
        if out is not None:
            return out
        ptr, nbytes = c_void_p(), c_int64()
        _lib.check(_lib.load().vq_resampler_device_output(self._h, ctypes.byref(ptr), ctypes.byref(nbytes)))
        return ptr.value

    def resize_list(self, frames, out_h: int, out_w: int, filter: int = BILINEAR,
                    crop: Optional[Tuple[int, int, int, int]] = None) -> np.ndarray:
        """:meth:`resize` for a list of separately allocated uint8 [h, w, 3] frames of one size (no stacking copy)."""
        arrs = [np.ascontiguousarray(f) for f in frames]
        if not arrs:
            return np.empty((0, out_h, out_w, 3), np.uint8)
        h, w = arrs[0].shape[:2]
        for a in arrs:
            if a.dtype != np.uint8 or a.shape != (h, w, 3):
                raise ValueError(f"expected uint8 frames of one shape ({h}, {w}, 3), got {a.dtype} {a.shape}")
        top, left, ch, cw = crop if crop is not None else (0, 0, out_h, out_w)
        out = np.empty((len(arrs), ch, cw, 3), dtype=np.uint8)
        ptrs = (c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        _lib.check(_lib.load().vq_resampler_run_u8_list(self._h, ptrs, len(arrs), h, w, int(filter), int(out_h), int(out_w),
                                                        int(top), int(left), int(ch), int(cw), out.ctypes.data_as(c_void_p)))
        return out

    def resize_device(self, src_ptr: int, n: int, h: int, w: int, out_h: int, out_w: int, filter: int = BILINEAR,
                      crop: Optional[Tuple[int, int, int, int]] = None, dst_ptr: Optional[int] = None) -> int:
        """Device-resident frames in, device result out; asynchronous on the handle's stream.  → device address."""
        top, left, ch, cw = crop if crop is not None else (0, 0, out_h, out_w)
        _lib.check(_lib.load().vq_resampler_run_u8_device(self._h, c_void_p(src_ptr), int(n), int(h), int(w), int(filter),
                                                          int(out_h), int(out_w), int(top), int(left), int(ch), int(cw),
                                                          c_void_p(dst_ptr or 0)))
        if dst_ptr:
            return dst_ptr
        ptr = c_void_p()
        _lib.check(_lib.load().vq_resampler_device_output(self._h, ctypes.byref(ptr), None))
        return ptr.value

    def cv_resize(self, frames, dsize: Tuple[int, int], **kw):
        """``cv2.resize(frame, dsize)`` (dsize = (width, height), INTER_LINEAR) of every frame — the resize of
        ``OptimizedFrameExtractor.extract_frames`` (reference frame_extractor.py:283-284).  Restated from OpenCV's
        published algorithm; parity unpinned (OpenCV is not installed in the build container)."""
        return self.resize(frames, int(dsize[1]), int(dsize[0]), CV_LINEAR, **kw)

    def stretch(self, frames, size: int = 224, **kw):
        """E1's ``transforms.Resize((S, S))`` (reference feature_extractor.py:55)."""
        return self.resize(frames, size, size, BILINEAR, **kw)

    def clip_processor(self, frames, size: int = 224, crop: int = 224, **kw):
        """The CLIP image processor's resize + centre crop (before its rescale/normalise, which the encoder's
        patchify kernel applies)."""
        a = _frames(frames)
        rh, rw, top, left = clip_processor_geometry(a.shape[1], a.shape[2], size, crop)
        return self.resize(a, rh, rw, BICUBIC, crop=(top, left, crop, crop), **kw)

    # -- quality filter -------------------------------------------------------
    def quality(self, frames) -> Tuple[np.ndarray, np.ndarray]:
        """→ (mean_brightness[n], laplacian_var[n]) float64 of BGR uint8 frames (reference frame_extractor.py:305-313)."""
        a = _frames(frames)
        n, h, w = a.shape[:3]
        mean, var = np.empty(n, np.float64), np.empty(n, np.float64)
        _lib.check(_lib.load().vq_frame_quality_u8(self._h, a.ctypes.data_as(c_void_p), n, h, w, 0,
                                                   mean.ctypes.data_as(ctypes.POINTER(c_double)),
                                                   var.ctypes.data_as(ctypes.POINTER(c_double))))
        return mean, var

    def is_low_quality(self, frames) -> np.ndarray:
        """``OptimizedFrameExtractor._is_low_quality`` per frame: very dark / very bright (mean < 20 or > 235) or
        blurry (Laplacian variance < 100)."""
        mean, var = self.quality(frames)
        return (mean < 20) | (mean > 235) | (var < 100)
