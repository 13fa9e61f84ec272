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


def rbox_to_quad(dets):
    """(n,5) -> (n,8) corners (float64 math, cast to float32) for poly fixtures."""
    d = dets.double()
    c, s = torch.cos(d[:, 4]), torch.sin(d[:, 4])
    w, h = d[:, 2] / 2, d[:, 3] / 2
    pts = []
    for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
        pts += [d[:, 0] + sx * w * c - sy * h * s, d[:, 1] + sx * w * s + sy * h * c]
    return torch.stack(pts, 1).float()
