"""ctypes binding of libobb_hip.so (C ABI in include/obb_hip.h).

PyTorch is used here only as plumbing: device memory (caching allocator),
the current HIP stream, and tensors as typed views of device pointers.
The library is loaded eagerly on first use and its absence is a hard error:
there is no CPU / eager fallback anywhere in this package.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OBB_HIP_LIB: another build of the same library (A/B measurements of compile-time variants); default: the in-tree build
LIB_PATH = os.environ.get("OBB_HIP_LIB") or os.path.join(_HERE, "libobb_hip.so")

OBB_OK = 0
OBB_NMS_DROP_SMALL = 1
_ERR = {-1: "bad argument", -2: "workspace missing or too small", -3: "kernel launch failed",
        -4: "internal error", -5: "no usable HIP device"}

_lib = None
_lock = threading.RLock()

_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/obb_hip.h declares
SIGNATURES = {
    "obb_version": (C.c_char_p, []),
    "obb_device_info": (_i32, [C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, _i32]),
    "obb_nms_set_max_grid": (_i32, [_i32]),
    "obb_profile_enable": (_i32, [_i32]),
    "obb_profile_collect": (_i32, [_vp, _vp, _i32]),
    "obb_nms_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "obb_nms_rotated_f32": (_i32, [_vp, _vp, _i64, _f32, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "obb_nms_rotated_f64": (_i32, [_vp, _vp, _i64, _f32, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "obb_nms_poly_f32": (_i32, [_vp, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "obb_merge_nms_poly_f64": (_i32, [_vp, _i64, _vp, _vp, _i64, C.c_double, _vp, _vp, _vp, _sz, _vp]),
    "obb_merge_nms_poly_all_f64": (_i32, [_vp, _i64, _vp, _vp, _i64, C.c_double, _vp, _vp, _vp, _sz, _vp]),
    "obb_merge_nms_hbb_f64": (_i32, [_vp, _i64, _i64, _vp, _vp, _i64, C.c_double, _vp, _vp, _vp, _sz, _vp]),
    "obb_task1_parse_tiles": (_i64, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "obb_task1_format_rows": (_i64, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64]),
    "obb_task1_parse_dets": (_i64, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "obb_task1_parse_gt": (_i64, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "obb_eval_best_gt_f64": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "obb_nms_obb_workspace_bytes": (_sz, [_i64, _i64, _i64, _i32]),
    "obb_non_max_suppression_obb": (_i32, [_vp, _i32, _i64, _i64, _i64, _f32, _f32, _vp, _i32, _i32, _i32, _i64, _i64, _f32,
                                           _vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "obb_non_max_suppression_obb_col": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _f32, _f32, _vp, _i32, _i32, _i32, _i64, _i64, _f32,
                                               _vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "obb_nms_obb_state_bytes": (_sz, [_i64]),
    "obb_non_max_suppression_obb_st": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _f32, _f32, _vp, _i32, _i32, _i32, _i64, _i64, _f32,
                                              _vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "obb_loss_workspace_bytes": (_sz, [_vp, _i64]),
    "obb_loss_build_targets": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "obb_loss_export_targets": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "obb_loss_forward": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "obb_loss_backward": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "obb_detect_decode": (_i32, [_vp, _i32, _i64, _i64, _i64, _i64, _i64, _vp, _f32, _vp, _vp, _i64, _i64, _vp]),
    "obb_detect_decode_col": (_i32, [_vp, _i32, _i64, _i64, _i64, _i64, _i64, _vp, _f32, _vp, _vp, _i64, _i64, _vp, _vp]),
    "obb_detect_decode_levels": (_i32, [_i32, _vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "obb_csl_encode_f32": (_i32, [_vp, _i64, _i32, C.c_double, C.c_double, _vp, _vp]),
    "obb_rbox2poly_f32": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "obb_val_postprocess_f32": (_i32, [_vp, _i64, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "obb_val_tail_batch_workspace_bytes": (_sz, [_i64, _i64]),
    "obb_val_tail_batch_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "obb_val_tail_batch_polled_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "obb_val_tail_batch_rows_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "obb_process_batch_workspace_bytes": (_sz, [_i64, _i64]),
    "obb_process_batch_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _sz, _vp]),
    "obb_rotated_iou_pairs_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "obb_rotated_iou_matrix_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "obb_quad_iou_matrix_f32": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "obb_rbox_overlaps_f32": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "_poly_nms": (None, [_vp, _vp, _vp, _i32, _i32, _f32, _i32]),
    "_overlaps": (None, [_vp, _vp, _vp, _i32, _i32, _i32]),
}


def lib():
    """Load libobb_hip.so (once).  Raises ImportError with build instructions when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C yolov5_obb_amd/csrc`). yolov5_obb_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == library/header mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


# ---- the compiled binding (csrc/torch_ext/nms_rotated_ext.cpp -> nms_rotated_ext_c.so, built by __graft_entry__.build())
# OBB_BINDING=ctypes keeps every call on the ctypes binding of this file (the fallback binding); default: the compiled
# module when it has been built, which binds the very library `lib()` loaded (dlopen of the same path: one copy of its state).
EXT_PATH = os.path.join(_HERE, "nms_rotated_ext_c.so")
_ext = None
_ext_tried = False


def compiled():
    """The compiled torch binding (module nms_rotated_ext_c) or None."""
    global _ext, _ext_tried
    if _ext_tried:
        return _ext
    with _lock:
        if _ext_tried:
            return _ext
        mod = None
        if os.environ.get("OBB_BINDING", "auto").lower() != "ctypes":
            if os.path.exists(EXT_PATH):
                import importlib.machinery
                import importlib.util
                loader = importlib.machinery.ExtensionFileLoader("nms_rotated_ext_c", EXT_PATH)
                spec = importlib.util.spec_from_file_location("nms_rotated_ext_c", EXT_PATH, loader=loader)
                try:
                    mod = importlib.util.module_from_spec(spec)
                    loader.exec_module(mod)
                    lib()                          # the library must load (ImportError with build instructions otherwise)
                    mod.init(LIB_PATH)
                except (ImportError, RuntimeError, OSError) as e:
                    # a stale artefact (built against another torch / ROCm) or a missing symbol: OBB_BINDING=compiled insists,
                    # the default falls back to the ctypes binding of the same library -- once, with a warning
                    if os.environ.get("OBB_BINDING", "auto").lower() == "compiled":
                        raise
                    import warnings
                    warnings.warn(f"{EXT_PATH} did not load ({type(e).__name__}: {e}): yolov5_obb_amd uses its ctypes binding of "
                                  "libobb_hip.so (same kernels, slower host side); rebuild with __graft_entry__.build()")
                    mod = None
                    lib()                          # (the library itself must load either way)
                _ext, _ext_tried = mod, True
                return _ext
            elif os.environ.get("OBB_BINDING", "auto").lower() == "compiled":
                raise ImportError(f"{EXT_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
            else:
                import warnings
                warnings.warn(f"{EXT_PATH} is not built: yolov5_obb_amd falls back to its ctypes binding of libobb_hip.so "
                              "(same kernels, slower host side); run __graft_entry__.build()")
        if mod is not None:
            lib()                              # the library must load (ImportError with build instructions otherwise)
            mod.init(LIB_PATH)
        _ext, _ext_tried = mod, True
    return _ext


def check(rc, what):
    if rc != OBB_OK:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, 'error')} (code {rc})")


class NmsAborted(RuntimeError):
    pass


def checked_count(n, what):
    """A negative kept-count is the device-side abort signal of the persistent NMS kernel (a team barrier timed out)."""
    if n < 0:
        raise NmsAborted(f"{what}: the NMS kernel aborted (a workgroup barrier timed out); results are invalid")
    return n


_abort_retries = [0]


def abort_retries():
    """Calls of this process that a barrier time-out of the persistent NMS kernel sent to the 8-workgroup grid (both bindings)."""
    ext = compiled()
    return _abort_retries[0] + (int(ext.abort_retries()) if ext is not None and hasattr(ext, "abort_retries") else 0)


def retry_on_abort(run):
    """The persistent NMS kernel needs all its workgroups resident (include/obb_hip.h: obb_nms_set_max_grid).  When a call
    aborts on a barrier, run it once more with a grid of 8 workgroups -- resident under any CU mask -- before giving up."""
    try:
        return run()
    except NmsAborted:
        L = lib()
        _abort_retries[0] += 1
        L.obb_nms_set_max_grid(8)
        try:
            return run()
        finally:
            L.obb_nms_set_max_grid(0)


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def stream_sync(device, handle=None):
    """Block until the current stream of `device` has finished (hipStreamSynchronize through torch)."""
    torch.cuda.current_stream(device).synchronize()


def stream_handle(device):
    """Raw hipStream_t of the current stream (an int): look it up once per call, it costs a few microseconds."""
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def guard(device):
    """`with torch.cuda.device(device)` only when the device is not already current (the context manager itself costs
    several microseconds per call, comparable to a kernel launch)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_cuda(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        # the reference dispatches CPU tensors to nms_rotated_cpu (nms_rotated_ext.cpp:38); this build is GPU-only
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor: yolov5_obb_amd is compiled for MI355X only "
                           f"(no CPU path, by design)")


# ---- kept-count read-back through polled pinned memory (single-list NMS entry points)
# The last kernel of a call writes the count with one aligned 8-byte store; the device reaches pinned host memory through the
# same pointer, so the calling thread can poll it: no copy kernel behind the NMS, no wake-up of a blocked stream wait
# (utils/general.py does the same for the fused driver's bs + 2 counters).
_PENDING = -(1 << 62)
_count_memo = {}


def pinned_count(device):
    """(tensor, numpy view) of one pinned int64 per (device, thread), armed with the PENDING marker."""
    import threading
    key = (device.index, threading.get_ident())
    ent = _count_memo.get(key)
    if ent is None:
        t = torch.empty(1, dtype=torch.int64).pin_memory()
        ent = _count_memo[key] = (t, t.numpy())
    ent[1][0] = _PENDING
    return ent


def wait_count(view, device, give_up=1.0):
    """Poll the armed counter; after `give_up` seconds fall back to a stream synchronise (a very long call)."""
    import time
    t0 = time.perf_counter()
    while view[0] == _PENDING:
        waited = time.perf_counter() - t0
        if waited > give_up:
            stream_sync(device)
            break
        if waited > 2e-3:
            time.sleep(0)                                    # let other Python threads run between polls
    return int(view[0])


_ws_cache = {}


def workspace(nbytes, device, stream=None):
    """A reusable scratch buffer per (device, stream); grown geometrically, never shrunk.  `stream`: the raw handle when the
    caller has looked it up already."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           stream if stream is not None else torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        size = max(int(nbytes), 1 << 20)
        if buf is not None:
            size = max(size, 2 * buf.numel())
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
