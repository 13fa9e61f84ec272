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


def _pinned_rows(n, cols):
    """A pinned (n, cols) float32 view of a cached host buffer (per thread; grown geometrically)."""
    import threading
    key = (threading.get_ident(), cols)
    buf = _pin_cache.get(key)
    if buf is None or buf.shape[0] < n:
        buf = torch.empty((max(1024, 2 * n), cols), dtype=torch.float32).pin_memory()
        _pin_cache[key] = buf
    return buf[:n]


def val_tail_batch(preds, targets, shapes, iouv, want_boxes=False):
    """The tail of val.py:209-250 for ALL images of a batch: three launches and ONE device -> host copy.

    preds    list of (n_i, 7) CUDA tensors [x y l s theta conf cls], the output of non_max_suppression_obb (consecutive views
             of its packed buffer are used in place; anything else is concatenated)
    targets  (nt, >= 7) labels of the batch [img cls cx cy l s theta ...] in pixels of the letterboxed frame
    shapes   per image (shape (h, w), ratio_pad ((gain, gain), (pad_x, pad_y))) as LoadImagesAndLabels yields them
    Returns  stats: per image (correct bool (n_i, niou), conf (n_i), cls (n_i)) on the HOST -- what val.py:250 appends -- and,
             with want_boxes, the packed device arrays (pred_poly, pred_hbb, pred_polyn, pred_hbbn) + the offsets."""
    import ctypes as C
    bs = len(preds)
    if bs == 0:
        return ([], None) if want_boxes else []
    for p in preds:
        _lib.require_cuda(p, "pred")
    dev = preds[0].device
    counts = [int(p.shape[0]) for p in preds]
    n = sum(counts)
    niou = int(iouv.shape[0])
    packed = None
    nz = [p for p in preds if p.shape[0]]
    if nz and all(p.dtype == torch.float32 and p.is_contiguous() and p.shape[1] == 7 for p in nz):
        base = nz[0].data_ptr()
        ok = True
        for p in nz:
            ok = ok and p.data_ptr() == base
            base += p.numel() * 4
        if ok:                                                   # the split views of one packed buffer, in order
            packed = torch.as_strided(nz[0], (n, 7), (7, 1))
    if packed is None:
        packed = torch.cat([p.to(torch.float32) for p in preds], 0).contiguous() if n else torch.zeros((0, 7), device=dev)
    tg = targets.to(device=dev, dtype=torch.float32).contiguous()
    nt, tcols = (int(tg.shape[0]), int(tg.shape[1])) if tg.dim() == 2 else (0, 0)
    iv = iouv.to(device=dev, dtype=torch.float32).contiguous()
    stats = torch.empty((n, niou + 2), dtype=torch.float32, device=dev)
    boxes = None
    if want_boxes:
        boxes = (torch.empty((n, 10), dtype=torch.float32, device=dev), torch.empty((n, 6), dtype=torch.float32, device=dev),
                 torch.empty((n, 10), dtype=torch.float32, device=dev), torch.empty((n, 6), dtype=torch.float32, device=dev))
    L = _lib.lib()
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    if n:
        with _lib.guard(dev):
            ws = _lib.workspace(L.obb_val_tail_batch_workspace_bytes(n, nt), dev)
            for b0 in range(0, bs, _TAIL_MAX_BS):                # (one call for any batch size val.py uses)
                b1 = min(bs, b0 + _TAIL_MAX_BS)
                k = b1 - b0
                doff = (C.c_int64 * (k + 1))(*[o - offs[b0] for o in offs[b0:b1 + 1]])
                img5 = (C.c_float * (5 * k))()
                for j in range(k):
                    shape, ratio_pad = shapes[b0 + j][0], shapes[b0 + j][1]
                    img5[5 * j + 0], img5[5 * j + 1] = float(ratio_pad[1][0]), float(ratio_pad[1][1])
                    img5[5 * j + 2] = float(ratio_pad[0][0])
                    img5[5 * j + 3], img5[5 * j + 4] = float(shape[1]), float(shape[0])
                tgk = tg
                if b0 or b1 < bs:                                # a chunk of a very large batch: its labels, re-based
                    sel = (tg[:, 0] >= b0) & (tg[:, 0] < b1)
                    tgk = tg[sel].clone()
                    tgk[:, 0] -= b0
                ntk = int(tgk.shape[0]) if tgk.dim() == 2 else 0
                sl = slice(offs[b0], offs[b1])

                def part(t):
                    return _lib.ptr(t[sl]) if t is not None and offs[b1] > offs[b0] else C.c_void_p(0)
                rc = L.obb_val_tail_batch_f32(part(packed), C.cast(doff, C.c_void_p), k, _lib.ptr(tgk) if ntk else C.c_void_p(0), ntk,
                                              tcols, C.cast(img5, C.c_void_p), _lib.ptr(iv), niou,
                                              part(boxes[0]) if boxes else C.c_void_p(0), part(boxes[1]) if boxes else C.c_void_p(0),
                                              part(boxes[2]) if boxes else C.c_void_p(0), part(boxes[3]) if boxes else C.c_void_p(0),
                                              part(stats), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
                _lib.check(rc, "obb_val_tail_batch_f32")
    # the one copy (and sync) of the batch: into a cached pinned buffer, then three splits (a per-image slicing loop of small
    # host tensor ops cost more than the three launches)
    host = _pinned_rows(n, niou + 2)
    if n:
        host.copy_(stats, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
    host = host.clone()                                          # (the pinned buffer is reused by the next batch)
    correct = (host[:, :niou] > 0.5).split(counts)
    conf = host[:, niou].contiguous().split(counts)
    pcls = host[:, niou + 1].contiguous().split(counts)
    out = list(zip(correct, conf, pcls))
    return (out, (boxes, offs)) if want_boxes else out
