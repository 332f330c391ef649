"""ctypes binding of libvq_amd.so (include/vq_amd.h).

There is no CPU fallback: if the shared library is missing, or no gfx950
device is visible, the calls below raise.  ``load()`` only needs the file (so
the symbol-export test can run on a box without a GPU); ``init()`` needs the
device.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# $VQ_AMD_LIB: another build of the same library (A/B timing of kernels on one GPU box)
LIB_PATH = os.environ.get("VQ_AMD_LIB") or os.path.join(_HERE, "lib", "libvq_amd.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

# Encoder handles on separate streams overlap (one batch's epilogues and tail workgroups under the other's MFMAs) only when the
# streams sit on different HARDWARE queues.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues, shared
# with the caller's own streams and the copy streams; measured on one MI355X: two handles 82k frames/s with 4 queues, 100k with
# 8 (three handles: 99k / 100k).  The runtime reads it when it initialises, so this only takes effect when the package is imported
# before the process's first HIP call; a value the caller exported wins.
_HWQ_PRESET = os.environ.get("GPU_MAX_HW_QUEUES")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _warn_if_hip_started_first() -> None:
    """The variable above is read when the HIP runtime starts.  If this process had already initialised HIP (torch.cuda in use
    before this package was imported) and had not exported the variable itself, the default of 4 hardware queues is in force
    and encoder handles on separate streams overlap poorly (two handles: ~82k instead of ~100k frames/s): say so once."""
    import sys
    if _HWQ_PRESET is not None:
        return
    torch = sys.modules.get("torch")
    try:
        started = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:                        # noqa: BLE001 - advisory only
        started = False
    if started:
        import warnings
        warnings.warn("video_quierer_amd was imported after this process had initialised HIP: GPU_MAX_HW_QUEUES=8 could not take "
                      "effect (the runtime keeps its 4 hardware queues) and batches in flight on separate streams will overlap "
                      "poorly.  Export GPU_MAX_HW_QUEUES=8 before the process's first HIP call, or import this package first.",
                      RuntimeWarning, stacklevel=3)


_warn_if_hip_started_first()

ENC_NCLASS = 11
IDX_NCLASS = 6


class VqError(RuntimeError):
    """A libvq_amd call failed (message from vq_last_error)."""


class TextConfigC(ctypes.Structure):
    _fields_ = [("vocab", c_int32), ("max_positions", c_int32), ("hidden", c_int32), ("mlp", c_int32),
                ("layers", c_int32), ("heads", c_int32), ("proj_dim", c_int32), ("eos_token_id", c_int32),
                ("ln_eps", c_float)]


class VitConfigC(ctypes.Structure):
    _fields_ = [("image_size", c_int32), ("patch_size", c_int32), ("hidden", c_int32), ("mlp", c_int32),
                ("layers", c_int32), ("heads", c_int32), ("proj_dim", c_int32), ("ln_eps", c_float)]


# name -> (restype, argtypes); every symbol include/vq_amd.h declares
SIGNATURES = {
    "vq_init": (c_int, [c_int]),
    "vq_device_count": (c_int, [POINTER(c_int)]),
    "vq_last_error": (c_char_p, []),
    "vq_version": (c_char_p, []),
    "vq_encoder_create": (c_int, [POINTER(VitConfigC), POINTER(POINTER(c_float)), c_int, c_int, POINTER(c_void_p)]),
    "vq_encoder_create_ex": (c_int, [POINTER(VitConfigC), POINTER(POINTER(c_float)), c_int, c_int, c_int, POINTER(c_void_p)]),
    "vq_encoder_create_shared": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "vq_encoder_destroy": (c_int, [c_void_p]),
    "vq_encoder_encode_u8": (c_int, [c_void_p, POINTER(c_uint8), c_int, c_int, POINTER(c_float)]),
    "vq_encoder_encode_u8_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vq_encoder_staging": (c_int, [c_void_p, c_int, POINTER(POINTER(c_uint8)), POINTER(ctypes.c_size_t)]),
    "vq_encoder_encode_staged": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_float)]),
    "vq_encoder_synchronize": (c_int, [c_void_p]),
    "vq_encoder_set_stream": (c_int, [c_void_p, c_void_p]),
    "vq_encoder_output_dim": (c_int, [c_void_p, POINTER(c_int)]),
    "vq_encoder_profile_begin": (c_int, [c_void_p]),
    "vq_encoder_profile_end": (c_int, [c_void_p, POINTER(c_float), POINTER(c_int)]),
    "vq_encoder_profile_class_name": (c_char_p, [c_int]),
    "vq_encoder_profile_bracket_overhead": (c_int, [c_void_p, POINTER(c_float)]),
    "vq_encoder_debug_set_layers": (c_int, [c_void_p, c_int]),
    "vq_encoder_debug_read": (c_int, [c_void_p, c_char_p, c_int, POINTER(c_float)]),
    "vq_text_encoder_create": (c_int, [POINTER(TextConfigC), POINTER(POINTER(c_float)), c_int, c_int, c_int, POINTER(c_void_p)]),
    "vq_text_encoder_encode_ids": (c_int, [c_void_p, POINTER(c_int32), c_int, c_int, POINTER(c_float)]),
    "vq_text_encoder_destroy": (c_int, [c_void_p]),
    "vq_debug_gemm": (c_int, [POINTER(c_float), POINTER(c_float), c_int, c_int, c_int, c_int, POINTER(c_float)]),
    "vq_index_create": (c_int, [c_int, POINTER(c_void_p)]),
    "vq_index_destroy": (c_int, [c_void_p]),
    "vq_index_add": (c_int, [c_void_p, POINTER(c_float), c_int64, c_int]),
    "vq_index_add_device": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "vq_index_update_rows": (c_int, [c_void_p, POINTER(c_float), POINTER(c_int64), c_int64, c_int]),
    "vq_index_size": (c_int, [c_void_p, POINTER(c_int64)]),
    "vq_index_clear": (c_int, [c_void_p]),
    "vq_index_set_id_ranks": (c_int, [c_void_p, POINTER(c_int32), c_int64]),
    "vq_index_search": (c_int, [c_void_p, POINTER(c_float), c_int, c_int, c_int, POINTER(c_int32), POINTER(c_float)]),
    "vq_index_search_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vq_index_synchronize": (c_int, [c_void_p]),
    "vq_index_set_stream": (c_int, [c_void_p, c_void_p]),
    "vq_index_export": (c_int, [c_void_p, POINTER(c_float)]),
    "vq_index_read_rows": (c_int, [c_void_p, POINTER(c_int64), c_int64, POINTER(c_float)]),
    "vq_index_profile_begin": (c_int, [c_void_p]),
    "vq_index_profile_end": (c_int, [c_void_p, POINTER(c_float), POINTER(c_int)]),
    "vq_index_profile_class_name": (c_char_p, [c_int]),
    "vq_index_last_search_stats": (c_int, [c_void_p, POINTER(c_int64)]),
    "vq_encoder_stage_frames": (c_int, [c_void_p, c_int, POINTER(c_void_p), c_int, c_int]),
    "vq_encoder_submit_staged": (c_int, [c_void_p, c_int, c_int, c_int]),
    "vq_encoder_wait_staged": (c_int, [c_void_p, c_int, POINTER(c_float)]),
    "vq_resampler_create": (c_int, [POINTER(c_void_p)]),
    "vq_resampler_destroy": (c_int, [c_void_p]),
    "vq_resampler_set_stream": (c_int, [c_void_p, c_void_p]),
    "vq_resampler_synchronize": (c_int, [c_void_p]),
    "vq_resampler_run_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vq_resampler_run_u8_list": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vq_resampler_run_u8_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vq_resampler_device_output": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "vq_clip_processor_geometry": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "vq_comm_unique_id": (c_int, [c_void_p, c_int]),
    "vq_comm_init": (c_int, [c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "vq_comm_destroy": (c_int, [c_void_p]),
    "vq_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "vq_allgather_rows": (c_int, [c_void_p, c_void_p, POINTER(c_int64), c_int, c_void_p, c_void_p]),
    "vq_compact_gathered_rows": (c_int, [c_void_p, POINTER(c_int64), c_int, c_int64, c_int, c_void_p, c_void_p]),
    "vq_comm_check": (c_int, [c_void_p]),
    "vq_index_search_sharded": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "vq_merge_topk_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vq_frame_quality_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_double), POINTER(c_double)]),
}

# include/vq_amd_diag.h: present only in a `make DIAG=1` build (scripts/ point $VQ_AMD_LIB at one); bound when found
DIAG_SIGNATURES = {
    "vq_debug_gemm_stamps": (c_int, [c_int, c_int, c_int, c_int, POINTER(ctypes.c_uint64)]),
    "vq_debug_gemm_ablate": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float)]),
    "vq_debug_gemm_clock": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float)]),
    "vq_debug_gemm_narrow": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_float)]),
    "vq_debug_gemm_stamps_deep": (c_int, [c_int, c_int, c_int, c_int, c_void_p]),
    "vq_debug_gemm_bench": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), c_void_p]),
}

_lock = threading.Lock()
_lib = None
_device = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into lib/libvq_amd.so (hipcc cross-compiles without a GPU).  Always runs make — it is
    incremental, so an edited kernel never leaves a stale library behind.  Serialised by a file lock: under
    torch.distributed.run every rank may find the library missing at the same moment; the Makefile links to a temporary
    name and renames, so a concurrent dlopen never sees a half-written file.  Builds the DEFAULT library only: a path given
    in $VQ_AMD_LIB is somebody else's build and is never (re)made here."""
    import fcntl
    with open(os.path.join(CSRC_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-C", CSRC_DIR, "-j4"] + (["-B"] if force else []))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return os.path.join(_HERE, "lib", "libvq_amd.so")


def load() -> ctypes.CDLL:
    """dlopen the library and declare every prototype.  Does not touch the GPU."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH) and os.environ.get("VQ_AMD_LIB"):
                raise VqError(f"$VQ_AMD_LIB points at {LIB_PATH}, which does not exist (it is never built automatically)")
            if not os.path.exists(LIB_PATH):
                # a fresh checkout has sources only: build in place if the toolchain is here (hipcc
                # cross-compiles gfx950 without a GPU); never substitute anything else for the library
                try:
                    build()
                except Exception as e:
                    raise VqError(f"{LIB_PATH} is missing and building it failed ({e}); "
                                  "there is no CPU fallback for this path") from e
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)      # AttributeError here = header/library mismatch
                fn.restype, fn.argtypes = res, args
            for name, (res, args) in DIAG_SIGNATURES.items():
                fn = getattr(lib, name, None)
                if fn is not None:
                    fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().vq_last_error()
        text = msg.decode("utf-8", "replace") if msg else "unknown error"
        if rc == -1:
            raise ValueError(f"libvq_amd: {text}")
        raise VqError(f"libvq_amd (code {rc}): {text}")


def init(device: int | None = None) -> int:
    """Bind this process to a GPU (default: $LOCAL_RANK or 0).  Raises without a gfx950 device."""
    global _device
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _device is None else _device
    if _device is not None and _device != device:
        raise VqError(f"this process is bound to GPU {_device}; one process drives one GPU "
                      f"(asked for {device}): launch one rank per GPU")
    if _device is None:
        check(lib.vq_init(int(device)))
        _device = device
    return _device


def device_count() -> int:
    n = c_int(0)
    check(load().vq_device_count(ctypes.byref(n)))
    return n.value


def fptr(a):
    return a.ctypes.data_as(POINTER(c_float))
