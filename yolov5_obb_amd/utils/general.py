"""Mirror of the OBB part of the reference's ``utils/general.py``: ``non_max_suppression_obb``
(utils/general.py:772-862), backed by ONE call into libobb_hip.so for the whole batch
(``obb_non_max_suppression_obb``, include/obb_hip.h) instead of a per-image loop of small ATen
kernels, host syncs and a device->host mask copy.
"""
import ctypes as C
import os
import threading
import time

import torch

from .. import _lib

pi = 3.141592  # utils/general.py:34 (truncated on purpose: it is the constant the labels were encoded with)

_MAX_WH = 4096      # utils/general.py:793
_POLL_COUNTS = os.environ.get("OBB_NMS_POLL_COUNTS", "1") != "0"      # counts through polled pinned memory (see below)
_PENDING = -(1 << 62)
_POLL_SECONDS = 2e-3  # busy-poll this long (a bs16 step is ~0.2 ms), then yield the GIL between polls, then give up polling
_POLL_GIVE_UP = 1.0   # ... after this many seconds: fall back to a stream synchronise
_MAX_NMS = 30000    # utils/general.py:794
_CSL = 180          # utils/general.py:784
_cap_memo = {}      # (device, thread, A, nc, multi_label, conf_thres) -> candidate slots per image that sufficed last time
_ws_memo = {}       # (bs, cap, nc, agnostic, grid capped) -> workspace bytes (a ctypes call saved per batch)
_meta_memo = {}     # (device index, bs, thread) -> the int64 buffer the counters are read back from
_cand_memo = {}     # same key -> largest candidate count of an image in the previous call (sort-algorithm hint; 0 forces the generic sort)
_SORT_LDS_HINT = 6144   # include/obb_hip.h OBB_NMS_SORT_LDS_HINT: hints up to this select the one-workgroup-per-image sort ...
_SORT_LDS_MAX = 8192    # ... OBB_NMS_SORT_LDS_MAX: which takes at most this many candidates of an image
_hold = {}              # (key, "seg" | "cand") -> calls the larger regime is still held after a repeated call (hysteresis)
_HOLD = 8
_seg_memo = {}          # same key -> largest NMS segment (an image's class) of the previous call: chooses the NMS kernel (the one-
_SEG_SMALL = 384        # workgroup-per-segment kernel of csrc/nms_small.h takes segments up to this size; 0 = unknown: the persistent one)


_small_memo = {}        # same key -> the previous call met boxes with a sub-pixel side (status[1] bit 62): the next call runs k_tiny_cross


_MEMO_LIMIT = 512       # keys (device, thread, shape, threshold): thread churn or a swept threshold must not grow the memos without bound


def _key(dev_index, A, nc, multi, conf_thres):
    # per device, calling thread and confidence threshold: other callers' batches say nothing about this one's
    if len(_cap_memo) > _MEMO_LIMIT:          # hints only: forgetting them costs the next call of every shape one generic pass
        for d in (_cap_memo, _cand_memo, _seg_memo, _hold, _small_memo):
            d.clear()
    return (dev_index, threading.get_ident(), int(A), int(nc), bool(multi), float(conf_thres))


def hints_clear():
    """Forget what earlier calls learned about their shapes (tests, tools)."""
    for d in (_cap_memo, _cand_memo, _seg_memo, _hold, _small_memo):
        d.clear()
    ext = _lib.compiled()
    if ext is not None:
        ext.hints_clear()


def hint_get(device, A, nc, multi_label, conf_thres):
    """The hint state of a shape on the active binding: dict(cand=, seg=, cap=, small_boxes=) or None."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    ext = _lib.compiled()
    if ext is not None:
        return ext.hint_get(idx, int(A), int(nc), bool(multi_label), float(conf_thres))
    key = _key(idx, A, nc, multi_label, conf_thres)
    if key not in _cand_memo and key not in _seg_memo:
        return None
    return {"cand": _cand_memo.get(key, _SORT_LDS_HINT), "seg": _seg_memo.get(key, 1), "cap": _cap_memo.get(key, 0),
            "small_boxes": bool(_small_memo.get(key, False)), "small_resolved": bool(_small_memo.get((key, "resolved"), False))}


def hint_set(device, A, nc, multi_label, conf_thres, cand=None, seg=None):
    """Force the hints of the next call of a shape (cand = 0: the generic sort; seg = 1: the small-segment kernel)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    ext = _lib.compiled()
    if ext is not None:
        ext.hint_set(idx, int(A), int(nc), bool(multi_label), float(conf_thres), -1 if cand is None else int(cand), -1 if seg is None else int(seg))
        return
    key = _key(idx, A, nc, multi_label, conf_thres)
    if cand is not None:
        _cand_memo[key] = int(cand); _hold[key, "cand"] = 0
    if seg is not None:
        _seg_memo[key] = int(seg); _hold[key, "seg"] = 0


def _label_rows(labels, bs, nc, device):
    """Apriori labels for autolabelling (utils/general.py:807-813) as candidate rows [img, x, y, l, s, theta, conf, cls]:
    obj = 1, one-hot class = 1 -> conf = 1; the CSL part of such a row is all zeros -> arg-max bin 0 -> theta = -90/180*pi."""
    rows = []
    theta0 = torch.tensor((0 - 90) / 180, dtype=torch.float32).mul(pi).item()
    for b in range(bs):
        if b < len(labels) and len(labels[b]):
            lb = labels[b].to(device=device, dtype=torch.float32)
            r = torch.zeros((lb.shape[0], 8), device=device, dtype=torch.float32)
            r[:, 0] = b
            r[:, 1:5] = lb[:, 1:5]
            r[:, 5] = theta0
            r[:, 6] = 1.0
            r[:, 7] = lb[:, 0].long().float()
            rows.append(r)
    return torch.cat(rows, 0).contiguous() if rows else None


def _objectness_column(prediction, pred):
    """The dense copy of prediction[..., 4] that this package's Detect hangs on its output (models/yolo.py), or None.
    Only trusted for the very tensor object Detect returned, unchanged since: same object (the attribute lives on it), same
    version counter (any in-place write through the tensor or a view of it bumps it), same geometry."""
    tag = getattr(prediction, "_obb_objcol", None)
    if tag is None or pred is not prediction:
        return None
    col, version = tag
    try:
        current = prediction._version      # inference tensors (torch.inference_mode) track no version: nothing to trust
    except RuntimeError:
        return None
    if (version != current or not isinstance(col, torch.Tensor) or col.shape != prediction.shape[:2]
            or col.dtype != prediction.dtype or col.device != prediction.device or not col.is_contiguous()):
        return None
    return col


def non_max_suppression_obb(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                            labels=(), max_det=1500):
    """Runs Non-Maximum Suppression (NMS) on inference results_obb (utils/general.py:772-862).

    Args:
        prediction (tensor): (b, n_all_anchors, [cx cy l s obj num_cls theta_cls]), fp32 or fp16, on the GPU
        agnostic (bool): True = NMS will be applied between elements of different categories
        labels : () or per-image apriori labels (n, [cls x y l s]) for autolabelling
    Returns:
        list of detections, len=batch_size, on (n,7) tensor per image [xylsθ, conf, cls] θ ∈ [-pi/2, pi/2)
    """
    _lib.require_cuda(prediction, "prediction")
    if prediction.dim() != 3:
        raise RuntimeError(f"prediction must be (bs, anchors, no), got {tuple(prediction.shape)}")
    nc = prediction.shape[2] - 5 - _CSL  # number of classes
    # Checks (same asserts as the reference, :789-790)
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    if nc < 1 or nc > 256:
        raise RuntimeError(f"non_max_suppression_obb: 1 <= nc <= 256 supported, got nc = {nc}")
    if prediction.dtype == torch.float32:
        dtype = 0
    elif prediction.dtype == torch.float16:
        dtype = 1
    else:
        raise RuntimeError(f"non_max_suppression_obb: float32 or float16 expected, got {prediction.dtype}")
    pred = prediction.contiguous()
    bs, A, no = pred.shape
    dev = pred.device
    col = _objectness_column(prediction, pred)
    multi = bool(multi_label) and nc > 1
    if bs == 0:
        return []
    if A == 0:
        return [torch.zeros((0, 7), device=dev)] * bs

    extra = _label_rows(labels, bs, nc, dev) if labels else None
    ext = _lib.compiled()
    if ext is not None:
        # the compiled binding (csrc/torch_ext/nms_rotated_ext.cpp): the same call sequence, retries and hint memo as below, in C++
        cl = None
        if classes is not None:
            cl = [int(c) for c in (classes if isinstance(classes, (list, tuple)) else list(classes))]
        return ext.non_max_suppression_obb(pred, float(conf_thres), float(iou_thres), cl, bool(agnostic), multi, extra, int(max_det), col)
    n_extra = 0 if extra is None else extra.shape[0]
    cls_arr = None
    if classes is not None:
        cl = [int(c) for c in (classes if isinstance(classes, (list, tuple)) else list(classes))]
        cls_arr = (C.c_int32 * max(1, len(cl)))(*cl)
        n_cls = len(cl)
        if n_cls == 0:
            return [torch.zeros((0, 7), device=dev)] * bs
    else:
        n_cls = 0

    worst = A * (nc if multi else 1) + n_extra
    key = _key(dev.index, A, nc, multi, conf_thres)
    cap = min(worst, max(_cap_memo.get(key, 0), 65536))
    L = _lib.lib()
    max_det = int(max_det)
    out = torch.empty((bs * max_det, 7), dtype=torch.float32, device=dev)   # packed: image b's rows follow image b-1's
    mkey = (dev.index, bs, threading.get_ident())
    meta = _meta_memo.get(mkey)                                       # counts[bs] + status[2]: read back before returning,
    if meta is None:                                                  # never handed out -> one buffer per (device, bs, thread)
        # Pinned host memory the last kernel writes straight into (the device reaches it through the same pointer), polled
        # by this thread: no copy kernel, no wake-up of a blocked stream wait.  (Device memory + one blocking copy: 0.254 ms
        # per bs16 step; pinned memory + stream synchronise: 0.282 ms.)
        meta = torch.empty(bs + 2, dtype=torch.int64).pin_memory() if _POLL_COUNTS else torch.empty(bs + 2, dtype=torch.int64, device=dev)
        _meta_memo[mkey] = meta = (meta, meta.numpy() if _POLL_COUNTS else None)
    meta, meta_np = meta
    agn = int(bool(agnostic))
    capped = False                # obb_nms_set_max_grid is per calling thread (thread_local in the library): no other thread sees it
    try:
        while True:
            # no history for this shape: assume the regime of the reference's default thresholds (at most a few thousand
            # candidates per image: the one-workgroup-per-image sort); a batch that turns out larger reports it (status[1]) and is
            # run once more on the multi-workgroup sort, which the memo then selects directly -- the first call of a shape is the fast
            # path, not the slow one (round 3 started from hint 0 = the generic sort)
            hint = int(_cand_memo.get(key, _SORT_LDS_HINT))
            # the same for the NMS kernel: assume the small segments of that regime (a few hundred boxes per image and class);
            # a call that meets a larger one reports it (status[0] = -1) and is repeated on the persistent kernel
            seg_hint = int(_seg_memo.get(key, 1))
            if meta_np is not None:
                meta_np.fill(_PENDING)
            with _lib.guard(dev):
                st = _lib.stream_handle(dev)
                wkey = (bs, cap, nc, agn, capped)          # (the library sizes the workspace from the calling thread's grid cap)
                nbytes = _ws_memo.get(wkey)
                if nbytes is None:
                    nbytes = _ws_memo[wkey] = L.obb_nms_obb_workspace_bytes(bs, cap, nc, agn)
                ws = _lib.workspace(nbytes, dev, st)
                rc = L.obb_non_max_suppression_obb_col(
                    _lib.ptr(pred), _lib.ptr(col), dtype, bs, A, no, float(conf_thres), float(iou_thres),
                    C.cast(cls_arr, C.c_void_p) if cls_arr is not None else C.c_void_p(0), n_cls, agn, int(multi),
                    max_det, _MAX_NMS, float(_MAX_WH), _lib.ptr(extra), n_extra, cap,
                    (hint & 0xffffffff) | ((seg_hint & 0x1fffffff) << 32) | ((1 << 62) if _small_memo.get(key) else 0), _lib.ptr(out), 1, _lib.ptr(meta),
                    C.c_void_p(meta.data_ptr() + 8 * bs), _lib.ptr(ws), ws.numel(), C.c_void_p(st))
            _lib.check(rc, "obb_non_max_suppression_obb")
            if meta_np is not None:                                   # every entry is one aligned 8-byte store of the last kernel
                t_poll = time.perf_counter()
                while meta_np.min() == _PENDING:
                    waited = time.perf_counter() - t_poll
                    if waited > _POLL_GIVE_UP:                        # something is badly wrong, or a very long call
                        _lib.stream_sync(dev)
                        break
                    if waited > _POLL_SECONDS:                        # the stream still holds earlier work (the model's forward):
                        time.sleep(0)                                 # let other Python threads (DataLoader, pin-memory) run
                m = meta_np.tolist()
            else:
                m = meta.tolist()                                     # the single device->host sync of the call
            _small_memo[key] = bool((m[bs + 1] >> 62) & 1)                    # status[1]: bit 62 = boxes with a sub-pixel side were met,
            _small_memo[key, "resolved"] = bool((m[bs + 1] >> 61) & 1)        # bit 61 = ... and such an image kept its class segments
            seg_max, m[bs + 1] = (m[bs + 1] >> 32) & 0x1fffffff, m[bs + 1] & 0xffffffff       # largest segment | largest candidate count
            if m[bs] == -1:                                           # a segment above the small kernel's limit: nothing is valid
                _seg_memo[key] = max(int(seg_max), _SEG_SMALL + 1)
                _hold[key, "seg"] = _HOLD
                continue
            if min(m[:bs]) < 0:                                       # a team barrier of the NMS kernel timed out
                if capped:
                    _lib.checked_count(min(m[:bs]), "obb_non_max_suppression_obb")
                capped = True
                L.obb_nms_set_max_grid(8)                             # once more with a grid that is resident under any CU mask
                continue
            if m[bs] > cap:                                           # an image produced more candidates than slots
                cap = min(worst, max(int(m[bs]), 2 * cap))
                continue
            if 0 < hint <= _SORT_LDS_HINT and m[bs + 1] > _SORT_LDS_MAX:   # the hint undersold this batch: such images were left out
                _cand_memo[key] = int(m[bs + 1])
                _hold[key, "cand"] = _HOLD
                continue
            break
    finally:
        if capped:
            L.obb_nms_set_max_grid(0)
    _cap_memo[key] = cap
    # the hints of the next call.  After a repeated call the larger regime is held for _HOLD calls unless the batch falls clearly
    # (25 %) below the limit: a stream whose batches hover around a limit does not pay the repeat on every other batch
    cand, seg = int(m[bs + 1]), int(seg_max)                          # (seg 0: the sort path of this call does not report it)
    if _hold.get((key, "cand"), 0) > 0 and cand > _SORT_LDS_HINT * 3 // 4:
        _hold[key, "cand"] -= 1
        cand = max(cand, _SORT_LDS_HINT + 1)
    else:
        _hold[key, "cand"] = 0
    if _hold.get((key, "seg"), 0) > 0 and (seg == 0 or seg > _SEG_SMALL * 3 // 4):
        _hold[key, "seg"] -= 1
        seg = max(seg, _SEG_SMALL + 1)
    else:
        _hold[key, "seg"] = 0
    _cand_memo[key] = cand
    _seg_memo[key] = seg
    counts = m[:bs]
    return list(out.narrow(0, 0, sum(counts)).split_with_sizes(counts))     # one call instead of bs slicing ops
