"""DOTA Task-1 (oriented box) evaluation with the det x GT polygon IoUs on the GPU: the interface of the reference's
DOTA_devkit/dota_evaluation_task1.py (`parse_gt`, `voc_ap`, `voc_eval`, `evaluate` = its main loop), same file formats.

The reference walks the confidence-sorted detections of a class one by one and, for each, calls SWIG
`polyiou.iou_poly` per ground-truth quad of the image (:168-223).  Those (ovmax, jmax) pairs do not depend on each
other, so they are one device call per class (`obb_eval_best_gt_f64`); the only sequential part -- a ground truth can be
claimed once, by the best-scored detection that reaches it (:225-233) -- is a first-occurrence pass over (image, jmax)
done with numpy.  Text parsing, `np.argsort(-confidence)` (the reference's tie order) and the AP integration stay on the
host, in double, as in the reference.  GPU only.
"""
import os

import numpy as np
import torch

from .. import _lib


def parse_gt(filename):
    """:21-53.  `x1 y1 x2 y2 x3 y3 x4 y4 name [difficult]` per line; shorter lines are skipped."""
    objects = []
    with open(filename, 'r') as f:
        for line in f:
            tok = line.strip().split(' ')
            if len(tok) < 9:
                continue
            obj = {'name': tok[8]}
            if len(tok) == 9:
                obj['difficult'] = 0
            elif len(tok) == 10:
                obj['difficult'] = int(tok[9])
            obj['bbox'] = [float(v) for v in tok[:8]]
            objects.append(obj)
    return objects


def voc_ap(rec, prec, use_07_metric=False):
    """:54-86."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _names(text, off, length):
    """str of text[off[i] : off[i] + length[i]] for every i (None: not UTF-8, the caller takes the path that raises)."""
    try:
        if text.isascii():
            t = text.decode('ascii')
            return [t[o:o + l] for o, l in zip(off.tolist(), length.tolist())]
        return [text[o:o + l].decode() for o, l in zip(off.tolist(), length.tolist())]
    except UnicodeDecodeError:
        return None


def _read_detections_native(detfile):
    """obb_task1_parse_dets (csrc/textio.hip): one pass over the file, exact decimal -> double; None when declined."""
    with open(detfile, 'rb') as f:
        text = f.read()
    cap = text.count(b'\n') + 1
    conf = np.empty(cap, dtype=np.float64)
    bb = np.empty((cap, 8), dtype=np.float64)
    off = np.empty(cap, dtype=np.int32)
    ln = np.empty(cap, dtype=np.int32)
    n = _lib.lib().obb_task1_parse_dets(text, len(text), cap, conf.ctypes.data, bb.ctypes.data, off.ctypes.data, ln.ctypes.data)
    if n <= 0:
        return None
    ids = _names(text, off[:n], ln[:n])
    if ids is None:
        return None
    return ids, conf[:n].copy(), bb[:n].copy()


def load_gt(annopath, imagenames, classname):
    """The ground truth of one class over the listed images, as arrays: (gts (k, 8) float64, difficult (k,) bool, gt_off
    (len(imagenames) + 1,) int64 -- the records of image i are gts[gt_off[i]:gt_off[i + 1]]), what voc_eval builds from
    parse_gt's records (:131-149).  One native pass per file (obb_task1_parse_gt); None when any file is not the plain
    layout (voc_eval then goes through parse_gt, which behaves like the reference on such input)."""
    texts, cap = [], 0
    for imagename in imagenames:
        with open(annopath.format(imagename), 'rb') as f:
            t = f.read()
        texts.append(t)
        cap += t.count(b'\n') + 1
    bbox = np.empty((max(cap, 1), 8), dtype=np.float64)
    off = np.empty(max(cap, 1), dtype=np.int32)
    ln = np.empty(max(cap, 1), dtype=np.int32)
    diff = np.empty(max(cap, 1), dtype=np.int32)
    L = _lib.lib()
    n, base, rec0 = 0, 0, [0]
    for t in texts:
        k = L.obb_task1_parse_gt(t, len(t), cap - n, bbox.ctypes.data + 64 * n, off.ctypes.data + 4 * n, ln.ctypes.data + 4 * n,
                                 diff.ctypes.data + 4 * n)
        if k < 0:
            return None
        off[n:n + k] += base
        n += k
        base += len(t)
        rec0.append(n)
    big = b''.join(texts)
    try:
        big.decode()                                     # the reference reads text: undecodable bytes make it raise
        cls = classname.encode()
    except UnicodeError:
        return None
    arr = np.frombuffer(big, dtype=np.uint8)
    cand = np.nonzero(ln[:n] == len(cls))[0]
    if len(cls) and len(cand):
        same = (arr[off[cand, None].astype(np.int64) + np.arange(len(cls))] == np.frombuffer(cls, dtype=np.uint8)).all(1)
        cand = cand[same]
    gt_off = np.searchsorted(cand, np.asarray(rec0, dtype=np.int64)).astype(np.int64)
    return bbox[cand], diff[cand] != 0, gt_off


def read_detections(detfile):
    """`image score x1 y1 .. x4 y4` per line (:152-160) -> (image ids, confidence (n,), BB (n, 8)).  The native reader when the
    file is the plain 10-column layout (then pandas' C reader with the round-trip float parser), line by line otherwise."""
    got = _read_detections_native(detfile)
    if got is not None:
        return got
    try:
        import pandas as pd
        df = pd.read_csv(detfile, sep=' ', header=None, float_precision='round_trip', dtype={0: str}, skip_blank_lines=False)
        if df.shape[1] == 10 and not df.isna().any().any():
            return df[0].tolist(), df[1].to_numpy(dtype=np.float64), df[[2, 3, 4, 5, 6, 7, 8, 9]].to_numpy(dtype=np.float64)
    except Exception:
        pass
    with open(detfile, 'r') as f:
        splitlines = [x.strip().split(' ') for x in f.readlines()]
    image_ids = [x[0] for x in splitlines]
    confidence = np.array([float(x[1]) for x in splitlines])
    BB = np.array([[float(z) for z in x[2:]] for x in splitlines])
    return image_ids, confidence, BB


def best_gt(dets8, det_img, gts8, gt_off):
    """(ovmax, jmax) per detection over the ground truth of its image (include/obb_hip.h: obb_eval_best_gt_f64)."""
    if not torch.cuda.is_available():
        raise RuntimeError("yolov5_obb_amd Task-1 evaluation needs a HIP device (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    nd = len(dets8)
    d_det = torch.from_numpy(np.ascontiguousarray(dets8, dtype=np.float64).reshape(-1, 8)).to(dev)
    d_img = torch.from_numpy(np.ascontiguousarray(det_img, dtype=np.int32)).to(dev)
    d_gt = torch.from_numpy(np.ascontiguousarray(gts8, dtype=np.float64).reshape(-1, 8)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(gt_off, dtype=np.int32)).to(dev)
    ov = torch.empty(nd, dtype=torch.float64, device=dev)
    jm = torch.empty(nd, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().obb_eval_best_gt_f64(_lib.ptr(d_det), _lib.ptr(d_img), nd, _lib.ptr(d_gt), _lib.ptr(d_off),
                                               len(gt_off) - 1, _lib.ptr(ov), _lib.ptr(jm), _lib.stream_ptr(dev)),
               "obb_eval_best_gt_f64")
    return ov.cpu().numpy(), jm.cpu().numpy()


def voc_eval(detpath, annopath, imagesetfile, classname, ovthresh=0.5, use_07_metric=False):
    """:88-249.  rec, prec, ap of one class."""
    with open(imagesetfile, 'r') as f:
        imagenames = [x.strip() for x in f.readlines()]
    index = {imagename: i for i, imagename in enumerate(imagenames)}   # a name listed twice: the last record wins, as in the reference's dict
    fast = load_gt(annopath, imagenames, classname)
    if fast is not None:
        gts, difficult, gt_off = fast
    else:
        gts, gt_off, difficult = [], [0], []
        for imagename in imagenames:
            objs = [o for o in parse_gt(annopath.format(imagename)) if o['name'] == classname]
            gts.extend(o['bbox'] for o in objs)
            difficult.extend(bool(o['difficult']) for o in objs)
            gt_off.append(len(gts))
        difficult = np.array(difficult, dtype=np.bool_)
        gt_off = np.array(gt_off, dtype=np.int64)
    # npos counts every listed record (:140-149 adds per imagename, duplicates included)
    npos = int((~difficult).sum())

    image_ids, confidence, BB = read_detections(detpath.format(classname))
    sorted_ind = np.argsort(-confidence)                 # :163
    BB = BB[sorted_ind, :]
    det_img = np.array([index[image_ids[x]] for x in sorted_ind], dtype=np.int32)   # KeyError for an unlisted image, as :169
    nd = len(det_img)

    ovmax, jmax = best_gt(BB, det_img, np.array(gts, dtype=np.float64).reshape(-1, 8), gt_off)
    tp = np.zeros(nd)
    fp = np.zeros(nd)
    hit = ovmax > ovthresh                               # NaN and -inf: False (:225)
    fp[~hit] = 1.
    h = np.nonzero(hit)[0]
    gidx = gt_off[det_img[h]] + jmax[h]                  # the ground truth a passing detection points at
    easy = ~difficult[gidx]                              # difficult ones are neither TP nor FP (:226)
    h, gidx = h[easy], gidx[easy]
    first = np.zeros(len(h), dtype=np.bool_)
    first[np.unique(gidx, return_index=True)[1]] = True   # h ascends = confidence order: first claim wins (:227-231)
    tp[h[first]] = 1.
    fp[h[~first]] = 1.

    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    ap = voc_ap(rec, prec, use_07_metric)
    return rec, prec, ap


def evaluate(detpath, annopath, imagesetfile, classnames, ovthresh=0.5, use_07_metric=True):
    """The loop of main() (:300-340): per-class AP (classes without a result file are skipped) and their mean."""
    classaps, skipped = [], 0
    for classname in classnames:
        if not os.path.exists(detpath.format(classname)):
            skipped += 1
            continue
        _, _, ap = voc_eval(detpath, annopath, imagesetfile, classname, ovthresh=ovthresh, use_07_metric=use_07_metric)
        classaps.append(ap)
    mean_ap = sum(classaps) / (len(classnames) - skipped)
    return mean_ap, 100 * np.array(classaps)
