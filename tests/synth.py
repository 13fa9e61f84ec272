"""Seeded synthetic inputs.  No dataset or checkpoint is available offline."""
import math

import numpy as np
import torch


def s_uniform(n, seed=0, extent=1024.0):
    """S-uniform(N, seed): few suppressions -- worst case for a greedy CPU NMS."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * extent
    wh = torch.rand(n, 2, generator=g) * 60 + 4
    th = (torch.rand(n, 1, generator=g) - 0.5) * math.pi
    scores = torch.rand(n, generator=g)
    return torch.cat([xy, wh, th], 1).float(), scores.float()


def s_clustered(n, k=300, seed=0, extent=1024.0):
    """S-clustered(N, K, seed): K objects, N/K near-duplicate candidates each -- resembles detector output."""
    g = torch.Generator().manual_seed(seed)
    cxy = torch.rand(k, 2, generator=g) * extent
    cwh = torch.rand(k, 2, generator=g) * 60 + 8
    cth = (torch.rand(k, 1, generator=g) - 0.5) * math.pi
    idx = torch.randint(0, k, (n,), generator=g)
    xy = cxy[idx] + torch.randn(n, 2, generator=g) * 2
    wh = cwh[idx] * (1 + 0.1 * torch.randn(n, 2, generator=g)).clamp(0.5, 1.5)
    th = cth[idx] + 0.05 * torch.randn(n, 1, generator=g)
    scores = torch.rand(n, generator=g)
    return torch.cat([xy, wh, th], 1).float(), scores.float()


def with_classes(dets, nc, seed=0, max_wh=4096.0):
    """S-class: class offset trick of utils/general.py:849-851 (xy += cls * 4096)."""
    g = torch.Generator().manual_seed(seed + 7)
    cls = torch.randint(0, nc, (dets.shape[0],), generator=g)
    d = dets.clone()
    d[:, :2] = d[:, :2] + cls[:, None].float() * max_wh
    return d, cls


def tie_free(scores):
    """Re-draw duplicate scores so that the kept set does not depend on the sort's tie rule."""
    s = scores.clone()
    g = torch.Generator().manual_seed(99)
    for _ in range(20):
        u, inv, cnt = torch.unique(s, return_inverse=True, return_counts=True)
        dup = cnt[inv] > 1
        if not dup.any():
            break
        s[dup] = torch.rand(int(dup.sum()), generator=g)
    return s


def regime_100k(name, n=100000):
    """The N = 100k regimes bench.py reports (SURVEY section 8d distributions; `raw` = the exact tensor bench.py times,
    scores not made tie-free: ties follow the documented rule, ascending original index)."""
    if name == "clustered_k300_raw":
        return s_clustered(n, 300, seed=0)
    if name == "clustered_k300":
        d, s = s_clustered(n, 300, seed=0)
    elif name == "clustered_k300_18cls":                    # the natural shape of BASELINE configs[3] (DOTAv2.0, nc 18)
        d, s = s_clustered(n, 300, seed=0)
        d, _ = with_classes(d, 18, 0)
    elif name == "clustered_k3000":
        d, s = s_clustered(n, 3000, seed=0)
    elif name == "clustered_k3000_18cls":                   # thousands of objects x 18 classes (the realistic configs[3] shape)
        d, s = s_clustered(n, 3000, seed=0)
        d, _ = with_classes(d, 18, 0)
    elif name == "uniform":
        d, s = s_uniform(n, 0)
    elif name == "uniform_18cls":
        d, s = s_uniform(n, 0)
        d, _ = with_classes(d, 18, 0)
    else:
        raise KeyError(name)
    return d, tie_free(s)


def rbox_to_quad(dets):
    """(n,5) -> (n,8) corners (float64 math, cast to float32) for poly fixtures."""
    d = dets.double()
    c, s = torch.cos(d[:, 4]), torch.sin(d[:, 4])
    w, h = d[:, 2] / 2, d[:, 3] / 2
    pts = []
    for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
        pts += [d[:, 0] + sx * w * c - sy * h * s, d[:, 1] + sx * w * s + sy * h * c]
    return torch.stack(pts, 1).float()


def s_pred(bs, A, nc, seed=0, n_obj=40, fg_frac=0.03, extent=1024.0, device="cpu", dtype=torch.float32):
    """S-pred: a synthetic, already decoded Detect output (bs, A, 5+nc+180), post-sigmoid like models/yolo.py:71-79.

    A fraction `fg_frac` of the anchors sits on one of `n_obj` planted objects per image (jittered copy of the
    object's box, high objectness, the object's class and a CSL bump at its angle bin); the rest is background
    (random box, objectness logit ~ N(-6, 1.5), cf. the bias init of models/yolo.py:230)."""
    g = torch.Generator(device=device).manual_seed(seed)
    R = lambda *s: torch.rand(*s, generator=g, device=device)
    N = lambda *s: torch.randn(*s, generator=g, device=device)
    no = 5 + nc + 180
    ocx = R(bs, n_obj, 2) * extent
    owh = torch.stack((R(bs, n_obj) * 150 + 12, R(bs, n_obj) * 40 + 6), -1)      # long edge, short edge
    obin = torch.randint(0, 180, (bs, n_obj), generator=g, device=device)
    ocls = torch.randint(0, nc, (bs, n_obj), generator=g, device=device)
    fg = R(bs, A) < fg_frac
    k = torch.randint(0, n_obj, (bs, A), generator=g, device=device)
    bi = torch.arange(bs, device=device)[:, None].expand(bs, A)
    out = torch.empty(bs, A, no, device=device, dtype=torch.float32)
    # boxes
    bg_xy = R(bs, A, 2) * extent
    bg_wh = torch.stack((R(bs, A) * 120 + 4, R(bs, A) * 40 + 2), -1)
    fg_xy = ocx[bi, k] + N(bs, A, 2) * 3
    fg_wh = owh[bi, k] * (1 + 0.1 * N(bs, A, 2)).clamp(0.5, 1.5)
    out[..., 0:2] = torch.where(fg[..., None], fg_xy, bg_xy)
    out[..., 2:4] = torch.where(fg[..., None], fg_wh, bg_wh)
    # objectness
    out[..., 4] = torch.sigmoid(torch.where(fg, N(bs, A) * 1.5 + 1.0, N(bs, A) * 1.5 - 6.0))
    # classes
    cl = N(bs, A, nc) - 4.0
    hot = torch.nn.functional.one_hot(ocls[bi, k], nc).bool() & fg[..., None]
    cl = torch.where(hot, N(bs, A, nc) + 2.0, cl)
    out[..., 5:5 + nc] = torch.sigmoid(cl)
    # CSL
    bins = torch.arange(180, device=device)
    d = (bins[None, None, :] - obin[bi, k][..., None]).abs()
    d = torch.minimum(d, 180 - d).float()
    bump = 5.0 * torch.exp(-d * d / 8.0) - 4.0
    csl = torch.where(fg[..., None], bump, torch.full_like(bump, -4.0)) + 0.5 * N(bs, A, 180)
    out[..., 5 + nc:] = torch.sigmoid(csl)
    return out.to(dtype)


DEFAULT_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]  # models/yolov5s.yaml:8-10
DEFAULT_STRIDES = [8.0, 16.0, 32.0]


def grid_anchors():
    """(nl, na, 2) anchors in grid units, as Detect stores them after Model.__init__ (models/yolo.py:119-124)."""
    a = torch.tensor(DEFAULT_ANCHORS).float().view(3, 3, 2)
    return a / torch.tensor(DEFAULT_STRIDES).view(3, 1, 1)


def s_head(bs, nc, sizes=(128, 64, 32), seed=0, n_obj=120, device="cpu", dtype=torch.float32):
    """S-head: synthetic conv outputs of the three Detect levels, [(bs, na*no, n, n)] -- what `Detect.m[i]` would produce
    (models/yolo.py:62) -- with `n_obj` planted objects per image that are SPATIALLY consistent: an object fires on the 3x3
    cells around its centre at the level whose anchors fit its size, on two of the three anchors, with xy / wh logits that
    decode (models/yolo.py:71-74) to the object's box plus jitter, its class and a CSL bump at its angle.  Everything else is
    background: objectness logit ~ N(-6, 1.5) (the bias init of models/yolo.py:230), class / angle logits ~ N(-4, .)."""
    g = torch.Generator(device=device).manual_seed(seed)
    R = lambda *s: torch.rand(*s, generator=g, device=device)
    N = lambda *s: torch.randn(*s, generator=g, device=device)
    na, no = 3, 5 + nc + 180
    lv = torch.randint(0, 3, (bs, n_obj), generator=g, device=device)
    ocls = torch.randint(0, nc, (bs, n_obj), generator=g, device=device)
    obin = torch.randint(0, 180, (bs, n_obj), generator=g, device=device)
    ouv = R(bs, n_obj, 2)                                                                      # centre in [0,1]^2
    oscale = R(bs, n_obj, 2) * 2.0 + 0.5                                                       # size / anchor
    bins = torch.arange(180, device=device)
    out = []
    for i, n in enumerate(sizes):
        x = torch.empty(bs, na, n, n, no, device=device)
        x[..., 0:4] = N(bs, na, n, n, 4) * 0.5
        x[..., 4] = N(bs, na, n, n) * 1.5 - 6.0
        x[..., 5:5 + nc] = N(bs, na, n, n, nc) - 4.0
        x[..., 5 + nc:] = 0.5 * N(bs, na, n, n, 180) - 4.0
        for b in range(bs):
            sel = (lv[b] == i).nonzero().flatten()
            if sel.numel() == 0:
                continue
            c = ouv[b, sel] * n                                                                # centre in cells
            cell = c.floor().long()
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    gx, gy = (cell[:, 0] + dx).clamp(0, n - 1), (cell[:, 1] + dy).clamp(0, n - 1)
                    for a in (0, 1, 2):
                        if (a + dy + dx) % 3 == 0:
                            continue                                                           # two of the three anchors
                        m = sel.numel()
                        # two objects on one cell: the later one wins, decided here (an index assignment with duplicate
                        # indices has no defined order)
                        lin, ar = gy * n + gx, torch.arange(m, device=device)
                        win = torch.full((n * n,), -1, dtype=torch.long, device=device).scatter_reduce(0, lin, ar, "amax")
                        own = win[lin] == ar
                        gyo, gxo = gy[own], gx[own]
                        t = torch.stack((c[:, 0] - gx, c[:, 1] - gy), -1) + 0.02 * N(m, 2)     # (sig*2 - 0.5) must give t
                        sxy = ((t + 0.5) / 2).clamp(0.02, 0.98)
                        swh = (oscale[b, sel] * (1 + 0.05 * N(m, 2))).clamp(0.05, 3.9).sqrt() / 2
                        x[b, a, gyo, gxo, 0:2] = torch.logit(sxy)[own]
                        x[b, a, gyo, gxo, 2:4] = torch.logit(swh.clamp(0.02, 0.98))[own]
                        x[b, a, gyo, gxo, 4] = (N(m) * 1.5 + 1.0)[own]
                        cl = N(m, nc) - 4.0
                        cl[torch.arange(m, device=device), ocls[b, sel]] = N(m) + 2.0
                        x[b, a, gyo, gxo, 5:5 + nc] = cl[own]
                        d = (bins[None, :] - obin[b, sel][:, None]).abs()
                        d = torch.minimum(d, 180 - d).float()
                        x[b, a, gyo, gxo, 5 + nc:] = (5.0 * torch.exp(-d * d / 8.0) - 4.0 + 0.5 * N(m, 180))[own]
        out.append(x.permute(0, 1, 4, 2, 3).contiguous().view(bs, na * no, n, n).to(dtype))
    return out


HYP_DOTA = dict(box=0.05, cls=0.5, cls_pw=1.0, theta=0.5, theta_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0,
                label_smoothing=0.0, cls_theta=180, csl_radius=2.0)       # data/hyps/obb/hyp.finetune_dota.yaml


def scaled_hyp(nc, imgsz=1024, nl=3):
    """train.py:249-252 scaling of the loss gains."""
    h = dict(HYP_DOTA)
    h['box'] *= 3 / nl
    h['cls'] *= nc / 80 * 3 / nl
    h['obj'] *= (imgsz / 640) ** 2 * 3 / nl
    h['theta'] *= 3 / nl
    return h


def s_loss(bs, nc, nt, seed=0, imgsz=1024, sizes=None):
    """S-loss: raw head outputs p[i] ~ N(0,1) at (bs,3,ny,nx,5+nc+180) and nt targets
    [img, cls, cx, cy, l, s, theta, csl x 180] in pixels (utils/datasets.py:637-659 layout)."""
    from oracle import pyref
    g = torch.Generator().manual_seed(seed)
    no = 5 + nc + 180
    sizes = sizes or [int(imgsz / s) for s in DEFAULT_STRIDES]
    p = [torch.randn(bs, 3, n, n, no, generator=g) for n in sizes]
    t = torch.zeros(nt, 7 + 180)
    if nt:
        t[:, 0] = torch.randint(0, bs, (nt,), generator=g).float()
        t[:, 1] = torch.randint(0, nc, (nt,), generator=g).float()
        t[:, 2:4] = torch.rand(nt, 2, generator=g) * imgsz
        t[:, 4] = torch.rand(nt, generator=g) * 100 + 20
        t[:, 5] = torch.rand(nt, generator=g) * 20 + 8
        th = (torch.rand(nt, generator=g) - 0.5) * pyref.PI
        t[:, 6] = th
        ang = th.double().numpy() * 180 / pyref.PI + 90
        csl = np.stack([pyref.gaussian_label(a, 180, 0, 2.0) for a in ang]) if nt else np.zeros((0, 180))
        t[:, 7:] = torch.from_numpy(csl).float()
    return p, t


def canon_rows(rows):
    """Rows of an (n,7) NMS output ordered by (conf desc, then the remaining columns): removes the dependence on how
    the sort broke score ties (torch's unstable sort in the reference vs this project's ascending-index rule)."""
    r = rows.detach().cpu().double().numpy() if isinstance(rows, torch.Tensor) else np.asarray(rows, dtype=np.float64)
    if len(r) == 0:
        return r
    key = np.lexsort((r[:, 6], r[:, 4], r[:, 3], r[:, 2], r[:, 1], r[:, 0], -r[:, 5]))
    return r[key]


class FakeDetect(torch.nn.Module):
    """The attributes ComputeLoss reads from Detect (models/yolo.py:33-47)."""

    def __init__(self, nc, anchors, strides):
        super().__init__()
        self.nc, self.nl, self.na = nc, anchors.shape[0], anchors.shape[1]
        self.no = nc + 5 + 180
        self.register_buffer('anchors', anchors.clone().float())
        self.stride = strides.clone().float()


class FakeModel(torch.nn.Module):
    """What ComputeLoss.__init__ needs from the model (utils/loss.py:93-120): parameters(), .hyp, .model[-1]."""

    def __init__(self, nc, hyp, device, anchors=None, strides=None):
        super().__init__()
        anchors = grid_anchors() if anchors is None else anchors
        strides = torch.tensor(DEFAULT_STRIDES) if strides is None else strides
        self.model = torch.nn.ModuleList([torch.nn.Identity(), FakeDetect(nc, anchors, strides)])
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.hyp = dict(hyp)
        self.to(device)
