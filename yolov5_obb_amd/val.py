"""Mirror of the per-image tail of the reference's ``val.py``: what runs for every image right
after ``non_max_suppression_obb`` -- ``rbox2poly`` -> ``poly2hbb`` -> ``xywh2xyxy`` -> ``scale_polys`` (val.py:226-236) and
``process_batch`` (val.py:69-90) -- as two fused launches of libobb_hip.so instead of ~15 small ATen kernels and a
``.cpu().numpy()`` round trip per image."""
import torch

from . import _lib


def val_postprocess(pred, ratio_pad=None, img1_shape=None, img0_shape=None):
    """pred (n,7) [x y l s theta conf cls] (CUDA) -> pred_poly (n,10), pred_hbb (n,6), pred_polyn (n,10), pred_hbbn (n,6).

    ratio_pad = ((h_ratio, w_ratio), (pad_w, pad_h)) as val.py passes it (shapes[si][1]); when None it is derived from the
    two shapes exactly like utils/general.py:639-641."""
    _lib.require_cuda(pred, "pred")
    if pred.dim() != 2 or pred.shape[1] != 7:
        raise RuntimeError(f"val_postprocess: pred must be (n,7), got {tuple(pred.shape)}")
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    p = pred.to(torch.float32).contiguous()
    n, dev = p.shape[0], p.device
    poly = torch.empty((n, 10), dtype=torch.float32, device=dev)
    hbb = torch.empty((n, 6), dtype=torch.float32, device=dev)
    polyn = torch.empty((n, 10), dtype=torch.float32, device=dev)
    hbbn = torch.empty((n, 6), dtype=torch.float32, device=dev)
    with _lib.guard(dev):
        rc = _lib.lib().obb_val_postprocess_f32(_lib.ptr(p), n, float(pad[0]), float(pad[1]), float(gain), _lib.ptr(poly), _lib.ptr(hbb),
                                                _lib.ptr(polyn), _lib.ptr(hbbn), _lib.stream_ptr(dev))
    _lib.check(rc, "obb_val_postprocess_f32")
    return poly, hbb, polyn, hbbn


def process_batch(detections, labels, iouv):
    """Return correct predictions matrix (val.py:69-90).  Both sets of boxes are in (x1, y1, x2, y2) format.
    Arguments:
        detections (Array[N, 6]), x1, y1, x2, y2, conf, class
        labels (Array[M, 5]), class, x1, y1, x2, y2
    Returns:
        correct (Array[N, 10]), for 10 IoU levels
    """
    _lib.require_cuda(detections, "detections")
    dev = detections.device
    det = detections.to(torch.float32).contiguous()
    lab = labels.to(device=dev, dtype=torch.float32).contiguous()
    iv = iouv.to(device=dev, dtype=torch.float32).contiguous()
    n, m, niou = det.shape[0], lab.shape[0], iv.shape[0]
    correct = torch.zeros((n, niou), dtype=torch.bool, device=dev)
    if n == 0:
        return correct
    L = _lib.lib()
    with _lib.guard(dev):
        ws = torch.empty(L.obb_process_batch_workspace_bytes(n, m), dtype=torch.uint8, device=dev)
        out = correct.view(torch.uint8)
        rc = L.obb_process_batch_f32(_lib.ptr(det), n, _lib.ptr(lab) if m else 0, m, _lib.ptr(iv), niou, _lib.ptr(out), _lib.ptr(ws),
                                     ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "obb_process_batch_f32")
    return correct


_TAIL_MAX_BS = 64      # csrc/head.hip kValTailMaxBs: images per obb_val_tail_batch_f32 call
_pin_cache = {}
_arr_cache = {}


def _pinned_rows(n, cols):
    """A pinned (n, cols) float32 view of a cached host buffer (per thread; grown geometrically)."""
    import threading
    key = (threading.get_ident(), cols)
    buf = _pin_cache.get(key)
    if buf is None or buf.shape[0] < n:
        buf = torch.empty((max(1024, 2 * n), cols), dtype=torch.float32).pin_memory()
        _pin_cache[key] = buf
    return buf[:n]


def _as_f32_cuda(t, dev):
    """t as a contiguous float32 tensor on dev -- without a torch call when it already is one (the common case)."""
    if t.dtype == torch.float32 and t.device == dev and t.is_contiguous():
        return t
    return t.to(device=dev, dtype=torch.float32).contiguous()


def val_tail_batch(preds, targets, shapes, iouv, want_boxes=False):
    """The tail of val.py:209-250 for ALL images of a batch: two launches, and the statistics land in pinned host memory
    that this thread polls (no copy kernel, no blocked stream wait; the host side of this function is most of its time, so it
    avoids every torch call it can).

    preds    list of (n_i, 7) CUDA tensors [x y l s theta conf cls], the output of non_max_suppression_obb (consecutive views
             of its packed buffer are used in place; anything else is concatenated)
    targets  (nt, >= 7) labels of the batch [img cls cx cy l s theta ...] in pixels of the letterboxed frame
    shapes   per image (shape (h, w), ratio_pad ((gain, gain), (pad_x, pad_y))) as LoadImagesAndLabels yields them
    Returns  stats: per image (correct bool (n_i, niou), conf (n_i), cls (n_i)) on the HOST -- what val.py:250 appends -- and,
             with want_boxes, the packed device arrays (pred_poly, pred_hbb, pred_polyn, pred_hbbn) + the offsets."""
    import ctypes as C
    import threading
    bs = len(preds)
    if bs == 0:
        return ([], None) if want_boxes else []
    ext = _lib.compiled()
    if ext is not None:          # the compiled binding builds the per-image tuples in C++ (csrc/torch_ext/nms_rotated_ext.cpp)
        return ext.val_tail_batch(list(preds), targets, shapes, iouv, bool(want_boxes))
    dev = preds[0].device
    if dev.type != "cuda":
        _lib.require_cuda(preds[0], "pred")
    counts = [p.shape[0] for p in preds]
    offs = [0] * (bs + 1)
    for b in range(bs):
        offs[b + 1] = offs[b] + counts[b]
    n = offs[bs]
    niou = iouv.shape[0]
    # the detections as ONE (n, 7) array: the split views of non_max_suppression_obb's packed buffer in order, else a copy
    packed, first = None, None
    ok = True
    for p in preds:
        if p.shape[0] == 0:
            continue
        if p.device != dev or p.dtype != torch.float32 or p.dim() != 2 or p.shape[1] != 7 or not p.is_contiguous():
            if p.device.type != "cuda":
                _lib.require_cuda(p, "pred")
            ok = False
            break
        if first is None:
            first, nxt = p, p.data_ptr()
        if p.data_ptr() != nxt:
            ok = False
            break
        nxt += p.shape[0] * 28
    if n:
        if ok and first is not None:
            packed = first if first.shape[0] == n else torch.as_strided(first, (n, 7), (7, 1))
        else:
            packed = torch.cat([p.to(torch.float32) for p in preds], 0).contiguous()
    tg = _as_f32_cuda(targets, dev) if targets.dim() == 2 and targets.shape[0] else None
    nt, tcols = (tg.shape[0], tg.shape[1]) if tg is not None else (0, 0)
    iv = _as_f32_cuda(iouv, dev)
    host = _pinned_rows(n, niou + 2)                             # the kernels write the rows straight into it
    boxes = None
    if want_boxes:
        boxes = (torch.empty((n, 10), dtype=torch.float32, device=dev), torch.empty((n, 6), dtype=torch.float32, device=dev),
                 torch.empty((n, 10), dtype=torch.float32, device=dev), torch.empty((n, 6), dtype=torch.float32, device=dev))
    if n:
        L = _lib.lib()
        null = C.c_void_p(0)
        with _lib.guard(dev):
            st = _lib.stream_handle(dev)
            ws = _lib.workspace(L.obb_val_tail_batch_workspace_bytes(n, nt), dev, st)
            flag, flag_np = _lib.pinned_count(dev)
            for b0 in range(0, bs, _TAIL_MAX_BS):                # (one call for any batch size val.py uses)
                b1 = min(bs, b0 + _TAIL_MAX_BS)
                k = b1 - b0
                akey = (threading.get_ident(), k)           # ctypes releases the GIL during the call that reads them: per thread
                arrs = _arr_cache.get(akey)
                if arrs is None:
                    arrs = _arr_cache[akey] = ((C.c_int64 * (k + 1))(), (C.c_float * (5 * k))())
                doff, img5 = arrs
                base = offs[b0]
                doff[:] = [o - base for o in offs[b0:b1 + 1]] if base else offs[b0:b1 + 1]
                flat = []
                for j in range(b0, b1):
                    shape, ratio_pad = shapes[j][0], shapes[j][1]
                    flat += (ratio_pad[1][0], ratio_pad[1][1], ratio_pad[0][0], shape[1], shape[0])
                img5[:] = flat
                tgk, ntk = tg, nt
                if nt and (b0 or b1 < bs):                       # a chunk of a very large batch: its labels, re-based
                    sel = (tg[:, 0] >= b0) & (tg[:, 0] < b1)
                    tgk = tg[sel].clone()
                    tgk[:, 0] -= b0
                    ntk = int(tgk.shape[0])
                lo, hi = offs[b0], offs[b1]
                if hi == lo:
                    continue

                def part(t, cols):
                    return C.c_void_p(t.data_ptr() + lo * cols * 4)
                if b1 < bs or b0:                                # not the last chunk of several: the rows are waited for at the end
                    flag_np[0] = _lib._PENDING
                rc = L.obb_val_tail_batch_polled_f32(
                    part(packed, 7), C.cast(doff, C.c_void_p), k, C.c_void_p(tgk.data_ptr()) if ntk else null, ntk, tcols,
                    C.cast(img5, C.c_void_p), C.c_void_p(iv.data_ptr()), niou,
                    part(boxes[0], 10) if boxes else null, part(boxes[1], 6) if boxes else null,
                    part(boxes[2], 10) if boxes else null, part(boxes[3], 6) if boxes else null,
                    part(host, niou + 2), C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st), C.c_void_p(flag.data_ptr()))
                _lib.check(rc, "obb_val_tail_batch_f32")
                _lib.wait_count(flag_np, dev)                    # every row of this chunk is in `host`
    # (the pinned buffer is reused by the next batch.  Copy and compare through numpy, single-threaded: a torch CPU op on these
    #  ~44k elements fans out to the whole intra-op pool -- 128 OpenMP workers on the GPU boxes -- whose spinning uses up the
    #  container's CPU quota: the kernel then parks the process for the rest of the 100 ms period, profiles/r5_host_stall.md)
    arr = host.numpy().copy()
    host = torch.from_numpy(arr)
    correct = torch.from_numpy(arr[:, :niou] > 0.5).split(counts)
    conf = host[:, niou].split(counts)
    pcls = host[:, niou + 1].split(counts)
    out = list(zip(correct, conf, pcls))
    return (out, (boxes, offs)) if want_boxes else out
