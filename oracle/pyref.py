"""TEST INFRASTRUCTURE -- numpy / torch-CPU restatement of the reference's *Python* hot path.

Nothing here is imported by the product.  Each function cites the reference file:line it
follows and is pinned against the reference itself (imported from /root/reference in the
build container by tests/golden/gen_golden.py; frozen outputs live in tests/golden/).

Native pieces (rotated IoU / NMS) are delegated to liboracle.so through ``oracle``.
"""
import math

import numpy as np
import torch

import oracle

PI = 3.141592            # the reference's truncated constant (utils/rboxs_utils.py:5, utils/general.py:34)
CSL_BINS = 180           # hard-coded width of the angle head (utils/general.py:784, models/yolo.py:40)
MAX_WH = 4096            # class offset in pixels (utils/general.py:793)
MAX_NMS = 30000          # candidate cap per image (utils/general.py:794)


# ----------------------------------------------------------------------------- CSL / box utils
def gaussian_label(label, num_class=180, u=0.0, sig=4.0):
    """utils/rboxs_utils.py:9-26.  Gaussian window over x = -n/2 .. n/2-1 rolled so that the peak lands
    on bin ceil(label) mod n  (index = int(n/2 - label) relies on python slice semantics for negatives)."""
    x = np.arange(-num_class / 2, num_class / 2)
    win = np.exp(-(x - u) ** 2 / (2 * sig ** 2))
    k = int(num_class / 2 - label)
    return np.concatenate([win[k:], win[:k]], axis=0)


def regular_theta(theta, mode='180', start=-PI / 2):
    """utils/rboxs_utils.py:28-37."""
    cycle = 2 * PI if mode == '360' else PI
    return (theta - start) % cycle + start


def rbox2poly(obb):
    """utils/rboxs_utils.py:106-145 (tensor or ndarray, (...,5) -> (...,8))."""
    if isinstance(obb, torch.Tensor):
        ctr, w, h, th = obb[:, :2], obb[:, 2:3], obb[:, 3:4], obb[:, 4:5]
        c, s = torch.cos(th), torch.sin(th)
        v1 = torch.cat((w / 2 * c, -w / 2 * s), -1)
        v2 = torch.cat((-h / 2 * s, -h / 2 * c), -1)
        return torch.cat((ctr + v1 + v2, ctr + v1 - v2, ctr - v1 - v2, ctr - v1 + v2), -1).reshape(*obb.shape[:-1], 8)
    ctr, w, h, th = np.split(obb, (2, 3, 4), axis=-1)
    c, s = np.cos(th), np.sin(th)
    v1 = np.concatenate([w / 2 * c, -w / 2 * s], -1)
    v2 = np.concatenate([-h / 2 * s, -h / 2 * c], -1)
    return np.concatenate([ctr + v1 + v2, ctr + v1 - v2, ctr - v1 - v2, ctr - v1 + v2], -1).reshape(*obb.shape[:-1], 8)


def poly2hbb(polys):
    """utils/rboxs_utils.py:147-181: axis-aligned hull [xc, yc, w, h] of (n,8) polygons."""
    lib = torch if isinstance(polys, torch.Tensor) else np
    x, y = polys[:, 0::2], polys[:, 1::2]
    if lib is torch:
        x1, x0, y1, y0 = x.amax(1), x.amin(1), y.amax(1), y.amin(1)
        return torch.stack(((x1 + x0) / 2.0, (y1 + y0) / 2.0, x1 - x0, y1 - y0), 1)
    x1, x0, y1, y0 = x.max(1), x.min(1), y.max(1), y.min(1)
    return np.stack(((x1 + x0) / 2.0, (y1 + y0) / 2.0, x1 - x0, y1 - y0), 1)


# ----------------------------------------------------------------------------- obb_nms wrapper
def obb_nms(dets, scores, iou_thr, ge=False):
    """utils/nms_rotated/nms_rotated_wrapper.py:6-46 on CPU tensors: drop boxes with min(w,h) < 0.001,
    greedy rotated NMS, indices into the unfiltered input.  ge=False: CUDA compare (>), True: CPU compare (>=)."""
    d = dets.detach().cpu().float().numpy()
    s = scores.detach().cpu().float().numpy()
    if d.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64)
    ok = ~(d[:, 2:4].min(1) < 0.001)
    if not ok.any():
        return torch.zeros(0, dtype=torch.int64)
    idx = np.nonzero(ok)[0]
    keep = oracle.nms_rotated(d[ok], s[ok], iou_thr, ge=ge)
    return torch.from_numpy(idx[keep].astype(np.int64))


# ----------------------------------------------------------------------------- NMS driver
def non_max_suppression_obb(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                            multi_label=False, labels=(), max_det=1500, ge=False):
    """utils/general.py:772-862 restated for CPU tensors (fp32 or fp16 input).

    Differences that are pinned rather than inherited: argmax / argsort ties resolve to the first index /
    stable order (the reference leaves them to torch); no wall-clock time limit (:858-860)."""
    nc = prediction.shape[2] - 5 - CSL_BINS
    ci = nc + 5
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    multi_label = bool(multi_label) and nc > 1
    out = [torch.zeros((0, 7))] * prediction.shape[0]
    for b in range(prediction.shape[0]):
        x = prediction[b]
        x = x[x[:, 4] > conf_thres].clone()                                # :785,804
        if labels and len(labels[b]):                                       # :807-813
            lb = labels[b]
            v = torch.zeros((len(lb), x.shape[1]), dtype=x.dtype)
            v[:, :4] = lb[:, 1:5].to(x.dtype)
            v[:, 4] = 1.0
            v[torch.arange(len(lb)), lb[:, 0].long() + 5] = 1.0
            x = torch.cat((x, v), 0)
        if not x.shape[0]:
            continue
        x[:, 5:ci] *= x[:, 4:5]                                             # :820 (in the input dtype)
        tidx = torch.argmax(x[:, ci:].float(), 1, keepdim=True)             # :822
        theta = (tidx - 90) / 180 * PI                                      # :823 -> float32
        if multi_label:                                                     # :826-828
            i, j = (x[:, 5:ci] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((x[i, :4].float(), theta[i], x[i, j + 5, None].float(), j[:, None].float()), 1)
        else:                                                               # :830-831
            conf, j = x[:, 5:ci].max(1, keepdim=True)
            x = torch.cat((x[:, :4].float(), theta, conf.float(), j.float()), 1)[conf.view(-1) > conf_thres]
        if classes is not None:                                             # :834-835
            x = x[(x[:, 6:7] == torch.tensor(classes, dtype=x.dtype)).any(1)]
        n = x.shape[0]
        if not n:
            continue
        if n > MAX_NMS:                                                     # :845-846
            x = x[torch.argsort(x[:, 5], descending=True, stable=True)[:MAX_NMS]]
        c = x[:, 6:7] * (0 if agnostic else MAX_WH)                         # :849
        rb = x[:, :5].clone()
        rb[:, :2] = rb[:, :2] + c                                           # :851
        keep = obb_nms(rb, x[:, 5], iou_thres, ge=ge)[:max_det]             # :853-855
        out[b] = x[keep]
    return out


# ----------------------------------------------------------------------------- Detect decode
def detect_decode(xs, anchors, strides):
    """models/yolo.py:67-81 inference branch for already-convolved, already-permuted head outputs.
    xs[i]: (bs, na, ny, nx, no) raw logits; anchors (nl, na, 2) in grid units; strides (nl,)."""
    z = []
    for i, x in enumerate(xs):
        bs, na, ny, nx, no = x.shape
        yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing='ij')
        grid = torch.stack((xv, yv), 2).expand(1, na, ny, nx, 2).to(x.dtype)
        ag = (anchors[i].clone() * strides[i]).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2).to(x.dtype)
        y = x.sigmoid()
        xy = (y[..., 0:2] * 2 - 0.5 + grid) * strides[i]
        wh = (y[..., 2:4] * 2) ** 2 * ag
        z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(bs, -1, no))
    return torch.cat(z, 1)


# ----------------------------------------------------------------------------- loss
def bbox_ciou(box1, box2, eps=1e-7):
    """utils/metrics.py:201-243 with x1y1x2y2=False, CIoU=True. box1 (4,n) (transposed), box2 (n,4); xywh."""
    box2 = box2.T
    b1x1, b1x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
    b1y1, b1y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
    b2x1, b2x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
    b2y1, b2y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(b1x2, b2x2) - torch.max(b1x1, b2x1)).clamp(0) * \
            (torch.min(b1y2, b2y2) - torch.max(b1y1, b2y1)).clamp(0)
    w1, h1 = b1x2 - b1x1, b1y2 - b1y1 + eps
    w2, h2 = b2x2 - b2x1, b2y2 - b2y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1x2, b2x2) - torch.min(b1x1, b2x1)
    ch = torch.max(b1y2, b2y2) - torch.min(b1y1, b2y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2x1 + b2x2 - b1x1 - b1x2) ** 2 + (b2y1 + b2y2 - b1y1 - b1y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


class LossSpec:
    """The attributes ComputeLoss reads from the model (utils/loss.py:93-120)."""

    def __init__(self, hyp, anchors, strides, nc, balance=None):
        self.hyp = dict(hyp)
        self.anchors = anchors.float()          # (nl, na, 2) grid units
        self.stride = strides.float()
        self.nl, self.na = self.anchors.shape[0], self.anchors.shape[1]
        self.nc = nc
        self.balance = list(balance) if balance is not None else \
            ({3: [4.0, 1.0, 0.4]}.get(self.nl, [4.0, 1.0, 0.25, 0.06, 0.02]))
        eps = self.hyp.get('label_smoothing', 0.0)
        self.cp, self.cn = 1.0 - 0.5 * eps, 0.5 * eps       # smooth_BCE, :13-15
        self.gr = 1.0


def build_targets(spec, p, targets):
    """utils/loss.py:194-275.  Returns per level: b, a, gj, gi (clamped), tbox (n,4), anchors (n,2), tcls (n), csl (n,180).
    Row order = the reference's: offset-major (5 neighbour candidates), then anchor-major, then target order."""
    na, nt = spec.na, targets.shape[0]
    g = 0.5
    off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=torch.float32) * g
    ai = torch.arange(na, dtype=torch.float32).view(na, 1).repeat(1, nt)
    tt = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)           # (na, nt, 7+180+1)
    out = []
    for i in range(spec.nl):
        anchors = spec.anchors[i]
        ny, nx = p[i].shape[2], p[i].shape[3]
        t = tt.clone()
        t[:, :, 2:6] /= spec.stride[i]
        if nt:
            r = t[:, :, 4:6] / anchors[:, None]
            t = t[torch.max(r, 1 / r).max(2)[0] < spec.hyp['anchor_t']]        # :237-240
            gxy = t[:, 2:4]
            gxi = torch.tensor([nx, ny], dtype=torch.float32) - gxy
            j, k = ((gxy % 1 < g) & (gxy > 1)).T
            l, m = ((gxi % 1 < g) & (gxi > 1)).T
            sel = torch.stack((torch.ones_like(j), j, k, l, m))
            t = t.repeat((5, 1, 1))[sel]
            offsets = (torch.zeros_like(gxy)[None] + off[:, None])[sel]
        else:
            t = tt[0]
            offsets = 0
        b, c = t[:, :2].long().T
        gxy, gwh = t[:, 2:4], t[:, 4:6]
        gij = (gxy - offsets).long()
        gi, gj = gij[:, 0].clamp(0, nx - 1), gij[:, 1].clamp(0, ny - 1)        # :267 (clamp_ mutates gij in place)
        gij_c = torch.stack((gi, gj), 1)
        a = t[:, -1].long()
        out.append(dict(b=b, a=a, gj=gj, gi=gi, tbox=torch.cat((gxy - gij_c, gwh), 1), anch=anchors[a], tcls=c,
                        csl=t[:, 7:-1]))
    return out


def focal_bce(pred, true, pos_weight, gamma, alpha=0.25):
    """utils/loss.py:35-62 FocalLoss(nn.BCEWithLogitsLoss(pos_weight), gamma) with the wrapped loss's 'mean' reduction."""
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, true, pos_weight=pos_weight, reduction='none')
    pred_prob = torch.sigmoid(pred)
    p_t = true * pred_prob + (1 - true) * (1 - pred_prob)
    alpha_factor = true * alpha + (1 - true) * (1 - alpha)
    modulating_factor = (1.0 - p_t) ** gamma
    return (loss * alpha_factor * modulating_factor).mean()


def compute_loss(spec, p, targets, sort_obj_iou=False, autobalance=False):
    """utils/loss.py:122-192 (sort_obj_iou :156-158 with a stable sort; hyp['fl_gamma'] > 0 wraps the three BCE terms in
    FocalLoss, :107-110; autobalance :180-184: spec.balance is updated IN PLACE from the per-level objectness losses, after each
    level has used its old weight, and renormalised to the stride-16 level, :115).  p[i]: (bs, na, ny, nx, no) logits
    (requires_grad ok)."""
    h = spec.hyp
    fg = float(h.get('fl_gamma', 0.0))
    if fg > 0:
        def bce(pred, true, pos_weight):
            return focal_bce(pred, true, pos_weight, fg)
    else:
        bce = torch.nn.functional.binary_cross_entropy_with_logits
    pw_cls, pw_th, pw_obj = (torch.tensor([h[k]], dtype=torch.float32) for k in ('cls_pw', 'theta_pw', 'obj_pw'))
    lbox = torch.zeros(1); lobj = torch.zeros(1); lcls = torch.zeros(1); lth = torch.zeros(1)
    tg = build_targets(spec, p, targets)
    ci = 5 + spec.nc
    for i, pi in enumerate(p):
        t = tg[i]
        tobj = torch.zeros_like(pi[..., 0])
        n = t['b'].shape[0]
        if n:
            ps = pi[t['b'], t['a'], t['gj'], t['gi']]
            pxy = ps[:, :2].sigmoid() * 2 - 0.5
            pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * t['anch']
            iou = bbox_ciou(torch.cat((pxy, pwh), 1).T, t['tbox'])
            lbox = lbox + (1.0 - iou).mean()
            score = iou.detach().clamp(0).type(tobj.dtype)
            ib, ia, igj, igi = t['b'], t['a'], t['gj'], t['gi']
            if sort_obj_iou:                                                            # :156-158
                sid = torch.argsort(score, stable=True)
                ib, ia, igj, igi, score = ib[sid], ia[sid], igj[sid], igi[sid], score[sid]
            # :159  tobj[b, a, gj, gi] = ...  Duplicate cells: torch's index_put_ is sequential (last writer wins) only
            # for small inputs; it runs multi-threaded (CPU) / unordered (CUDA) for large ones.  Pinned rule: the LAST
            # row in row order wins, made explicit here so that the oracle does not depend on torch's threading.
            lin = ((ib * pi.shape[1] + ia) * pi.shape[2] + igj) * pi.shape[3] + igi
            last = torch.full((tobj.numel(),), -1, dtype=torch.long)
            last.scatter_reduce_(0, lin, torch.arange(n), reduce='amax', include_self=True)
            win = last[lin] == torch.arange(n)
            tobj.view(-1)[lin[win]] = ((1.0 - spec.gr) + spec.gr * score)[win]
            if spec.nc > 1:
                tc = torch.full_like(ps[:, 5:ci], spec.cn)
                tc[torch.arange(n), t['tcls']] = spec.cp
                lcls = lcls + bce(ps[:, 5:ci], tc, pos_weight=pw_cls)
            lth = lth + bce(ps[:, ci:], t['csl'].type(ps.dtype), pos_weight=pw_th)
        obji = bce(pi[..., 4], tobj, pos_weight=pw_obj)                                  # :178
        lobj = lobj + obji * spec.balance[i]                                            # :179
        if autobalance:
            spec.balance[i] = spec.balance[i] * 0.9999 + 0.0001 / obji.detach().item()  # :180-181
    if autobalance:
        ssi = [float(x) for x in spec.stride].index(16.0)                               # :115
        spec.balance = [x / spec.balance[ssi] for x in spec.balance]                    # :183-184
    lbox = lbox * h['box']; lobj = lobj * h['obj']; lcls = lcls * h['cls']; lth = lth * h['theta']
    bs = p[0].shape[0]
    return (lbox + lobj + lcls + lth) * bs, torch.cat((lbox, lobj, lcls, lth)).detach()


# ----------------------------------------------------------------------------- post-NMS tail of val.py
def xywh2xyxy(x):
    """utils/general.py:590-597."""
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def val_postprocess(pred, gain, pad):
    """val.py:226-236 with scale_polys (utils/general.py:636-650)."""
    poly = rbox2poly(pred[:, :5])
    pred_poly = torch.cat((poly, pred[:, -2:]), 1)
    pred_hbb = torch.cat((xywh2xyxy(poly2hbb(pred_poly[:, :8])), pred_poly[:, -2:]), 1)
    pred_polyn = pred_poly.clone()
    pred_polyn[:, [0, 2, 4, 6]] -= pad[0]
    pred_polyn[:, [1, 3, 5, 7]] -= pad[1]
    pred_polyn[:, :8] /= gain
    pred_hbbn = torch.cat((xywh2xyxy(poly2hbb(pred_polyn[:, :8])), pred_polyn[:, -2:]), 1)
    return pred_poly, pred_hbb, pred_polyn, pred_hbbn


def box_iou(box1, box2):
    """utils/metrics.py:246-268."""
    area1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    area2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (area1[:, None] + area2 - inter)


def process_batch(detections, labels, iouv):
    """val.py:69-90, numpy steps included (argsort()[::-1], two np.unique(return_index=True))."""
    correct = torch.zeros(detections.shape[0], iouv.shape[0], dtype=torch.bool)
    iou = box_iou(labels[:, 1:], detections[:, :4])
    x = torch.where((iou >= iouv[0]) & (labels[:, 0:1] == detections[:, 5]))
    if x[0].shape[0]:
        matches = torch.cat((torch.stack(x, 1), iou[x[0], x[1]][:, None]), 1).numpy()
        if x[0].shape[0] > 1:
            matches = matches[matches[:, 2].argsort()[::-1]]
            matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
            matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
        matches = torch.Tensor(matches)
        correct[matches[:, 1].long()] = matches[:, 2:3] >= iouv
    return correct


# ----------------------------------------------------------------------------- tile -> full-image merge (DOTA_devkit)
def merge_nms_poly_fast(dets, thresh):
    """DOTA_devkit/ResultMerge_multi_process.py:62-123 (py_cpu_nms_poly_fast): horizontal-box gate, then
    polyiou.cpp's iou_poly(kept, candidate) in double (oracle.piou_matrix float64 flavour, pinned to polyiou.cpp)."""
    import oracle
    dets = np.asarray(dets, dtype=np.float64).reshape(-1, 9)
    obbs = dets[:, 0:-1]
    x1 = np.min(obbs[:, 0::2], axis=1); y1 = np.min(obbs[:, 1::2], axis=1)
    x2 = np.max(obbs[:, 0::2], axis=1); y2 = np.max(obbs[:, 1::2], axis=1)
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = dets[:, 8].argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        idx = np.where(ovr > 0)[0]
        if idx.size:
            ovr[idx] = oracle.piou_matrix(dets[i:i + 1, :8].copy(), dets[rest[idx], :8].copy())[0]
        order = rest[np.where(ovr <= thresh)[0]]
    return keep


def merge_nms_poly_all(dets, thresh):
    """DOTA_devkit/ResultMerge_multi_process.py:24-60 (py_cpu_nms_poly): like the fast variant WITHOUT the horizontal-box gate --
    iou_poly(kept, candidate) for every remaining candidate."""
    import oracle
    dets = np.asarray(dets, dtype=np.float64).reshape(-1, 9)
    order = dets[:, 8].argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        if rest.size == 0:
            break
        ovr = oracle.piou_matrix(dets[i:i + 1, :8].copy(), dets[rest, :8].copy())[0]
        order = rest[np.where(ovr <= thresh)[0]]
    return keep


def merge_nms_hbb(dets, thresh):
    """DOTA_devkit/ResultMerge_multi_process.py:125-157 (py_cpu_nms, what mergebyrec hands to mergebase): horizontal boxes in
    columns 0..3, the score in column 4, the "+ 1" pixel convention in areas AND intersections, numpy double arithmetic."""
    dets = np.asarray(dets, dtype=np.float64)
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = dets[:, 4].argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        with np.errstate(invalid='ignore', divide='ignore'):
            ovr = inter / (areas[i] + areas[rest] - inter)
        order = rest[np.where(ovr <= thresh)[0]]
    return keep


def merge_result_lines(lines, thresh=0.2, nms=None):
    """mergesingle of ResultMerge_multi_process.py:183-234 on the lines of one Task1_<class>.txt: returns the output lines."""
    import re
    boxes = {}
    for line in lines:
        tok = line.strip().split(' ')
        sub = tok[0]
        oriname = sub.split('__')[0]
        xy = re.findall(r'\d+', re.findall(r'__\d+___\d+', sub)[0])
        x, y = int(xy[0]), int(xy[1])
        rate = re.findall(r'__([\d+\.]+)__\d+___', sub)[0]
        poly = list(map(float, tok[2:]))
        det = []
        for i in range(len(poly) // 2):
            det.append(float(poly[i * 2] + x) / float(rate))
            det.append(float(poly[i * 2 + 1] + y) / float(rate))
        det.append(float(tok[1]))
        boxes.setdefault(oriname, []).append(det)
    out = []
    for name, dets in boxes.items():
        for k in (nms or merge_nms_poly_fast)(np.array(dets), thresh):
            det = dets[k]
            out.append(name + ' ' + str(round(det[-1], 2)) + ' ' + ' '.join(str(round(v, 1)) for v in det[:8]))
    return out


# ----------------------------------------------------------------------------- DOTA Task-1 evaluation (DOTA_devkit)
def task1_best_gt(bb, BBGT):
    """dota_evaluation_task1.py:172-218 for one detection: (ovmax, jmax); jmax is None when no ground truth passes the gate."""
    import oracle
    ovmax, jmax = -np.inf, None
    if BBGT.size > 0:
        gx1 = np.min(BBGT[:, 0::2], axis=1); gy1 = np.min(BBGT[:, 1::2], axis=1)
        gx2 = np.max(BBGT[:, 0::2], axis=1); gy2 = np.max(BBGT[:, 1::2], axis=1)
        bx1, by1, bx2, by2 = np.min(bb[0::2]), np.min(bb[1::2]), np.max(bb[0::2]), np.max(bb[1::2])
        iw = np.maximum(np.minimum(gx2, bx2) - np.maximum(gx1, bx1) + 1., 0.)
        ih = np.maximum(np.minimum(gy2, by2) - np.maximum(gy1, by1) + 1., 0.)
        inters = iw * ih
        uni = ((bx2 - bx1 + 1.) * (by2 - by1 + 1.) + (gx2 - gx1 + 1.) * (gy2 - gy1 + 1.) - inters)
        keep = np.where(inters / uni > 0)[0]
        if len(keep) > 0:
            ov = oracle.piou_matrix(np.ascontiguousarray(BBGT[keep]), bb[None, :].copy())[:, 0]     # iou_poly(GT, det)
            ovmax = np.max(ov)
            jmax = int(keep[np.argmax(ov)])
    return ovmax, jmax


def task1_voc_eval(gt_by_image, imagenames, det_lines, classname, ovthresh=0.5, use_07_metric=False):
    """voc_eval of dota_evaluation_task1.py:88-249 on parsed inputs: gt_by_image name -> list of
    {'name','difficult','bbox'} (parse_gt's records), det_lines = the lines of Task1_<class>.txt."""
    class_recs, npos = {}, 0
    for name in imagenames:
        R = [o for o in gt_by_image[name] if o['name'] == classname]
        diff = np.array([o['difficult'] for o in R]).astype(np.bool_)
        npos += int(sum(~diff))
        class_recs[name] = {'bbox': np.array([o['bbox'] for o in R]), 'difficult': diff, 'det': [False] * len(R)}
    split = [x.strip().split(' ') for x in det_lines]
    image_ids = [x[0] for x in split]
    confidence = np.array([float(x[1]) for x in split])
    BB = np.array([[float(z) for z in x[2:]] for x in split])
    sorted_ind = np.argsort(-confidence)
    BB = BB[sorted_ind, :]
    image_ids = [image_ids[x] for x in sorted_ind]
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = class_recs[image_ids[d]]
        ovmax, jmax = task1_best_gt(BB[d, :].astype(float), R['bbox'].astype(float))
        if ovmax > ovthresh:
            if not R['difficult'][jmax]:
                if not R['det'][jmax]:
                    tp[d] = 1.
                    R['det'][jmax] = 1
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
    else:
        mrec = np.concatenate(([0.], rec, [1.])); mpre = np.concatenate(([0.], prec, [0.]))
        for i in range(mpre.size - 1, 0, -1):
            mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
        i = np.where(mrec[1:] != mrec[:-1])[0]
        ap = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return rec, prec, ap
