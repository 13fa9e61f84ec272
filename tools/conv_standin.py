"""MEASUREMENT TOOL (not product, not oracle): a yolov5s-shaped convolutional stand-in for the part of val.py this
repository does NOT replace, so that bench.py can report the reference's three time buckets (val.py:183-207,286-291:
pre-process, inference, NMS) and `seen / sum(dt)` on the GPU box, where neither the reference tree nor a checkpoint
exists.

The backbone / neck follow the layer table of the reference's models/yolov5s.yaml (depth_multiple 0.33, width_multiple
0.50: Conv-BN-SiLU, C3 with 1/2/3/1 bottlenecks, SPPF, PANet neck with two upsample and two stride-2 paths) as plain
torch.nn modules on PyTorch-ROCm (MIOpen), random-init weights -- the reference runs exactly this part on PyTorch too.
The head is the PRODUCT's `Detect` (yolov5_obb_amd/models/yolo.py, one HIP pass per level) and the loop is the product's
`val_sharded.run`.  Timing only: random weights detect nothing meaningful, so `calibrate()` shifts the head's objectness
and class biases until about `fg_frac` of the anchors pass the confidence filter, like tests/synth.py:s_pred does for the
synthetic head output.
"""
import math

import torch
import torch.nn as nn

ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]   # models/yolov5s.yaml:8-10
STRIDES = [8.0, 16.0, 32.0]


class CBS(nn.Module):                       # models/common.py:37-49  Conv = conv + bn + SiLU
    def __init__(self, cin, cout, k=1, s=1, p=None):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, k // 2 if p is None else p, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Residual(nn.Module):                  # models/common.py:94-104  Bottleneck
    def __init__(self, c, shortcut):
        super().__init__()
        self.a, self.b, self.shortcut = CBS(c, c, 1), CBS(c, c, 3), shortcut

    def forward(self, x):
        y = self.b(self.a(x))
        return x + y if self.shortcut else y


class CSP3(nn.Module):                      # models/common.py:126-138  C3
    def __init__(self, cin, cout, n, shortcut=True):
        super().__init__()
        h = cout // 2
        self.left, self.right, self.out = CBS(cin, h, 1), CBS(cin, h, 1), CBS(2 * h, cout, 1)
        self.body = nn.Sequential(*[Residual(h, shortcut) for _ in range(n)])

    def forward(self, x):
        return self.out(torch.cat((self.body(self.left(x)), self.right(x)), 1))


class PoolPyramid(nn.Module):               # models/common.py:181-196  SPPF
    def __init__(self, c, k=5):
        super().__init__()
        self.inp, self.out = CBS(c, c // 2, 1), CBS(2 * c, c, 1)
        self.pool = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        x = self.inp(x)
        a = self.pool(x)
        b = self.pool(a)
        return self.out(torch.cat((x, a, b, self.pool(b)), 1))


class StandinV5s(nn.Module):
    """yolov5s-shaped network with the product's OBB Detect head; forward(im) -> (z (b, A, no), levels) in eval mode."""

    def __init__(self, nc=16, width=0.5, depths=(1, 2, 3, 1)):
        super().__init__()
        from yolov5_obb_amd.models.yolo import Detect
        c = [int(round(v * width)) for v in (64, 128, 256, 512, 1024)]
        d = depths
        self.s0 = CBS(3, c[0], 6, 2, 2)
        self.s1 = nn.Sequential(CBS(c[0], c[1], 3, 2), CSP3(c[1], c[1], d[0]))
        self.s2 = nn.Sequential(CBS(c[1], c[2], 3, 2), CSP3(c[2], c[2], d[1]))          # P3 / 8
        self.s3 = nn.Sequential(CBS(c[2], c[3], 3, 2), CSP3(c[3], c[3], d[2]))          # P4 / 16
        self.s4 = nn.Sequential(CBS(c[3], c[4], 3, 2), CSP3(c[4], c[4], d[3]), PoolPyramid(c[4]))   # P5 / 32
        self.lat5, self.td4 = CBS(c[4], c[3], 1), CSP3(2 * c[3], c[3], 1, False)
        self.lat4, self.td3 = CBS(c[3], c[2], 1), CSP3(2 * c[2], c[2], 1, False)
        self.down3, self.bu4 = CBS(c[2], c[2], 3, 2), CSP3(2 * c[2], c[3], 1, False)
        self.down4, self.bu5 = CBS(c[3], c[3], 3, 2), CSP3(2 * c[3], c[4], 1, False)
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.detect = Detect(nc=nc, anchors=ANCHORS, ch=(c[2], c[3], c[4]))
        self.detect.stride = torch.tensor(STRIDES)
        self.detect.anchors /= self.detect.stride.view(-1, 1, 1)
        self.nc = nc
        for i, s in enumerate(STRIDES):                                                 # models/yolo.py:226-233 _initialize_biases
            b = self.detect.m[i].bias.view(self.detect.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:5 + nc] += math.log(0.6 / (nc - 0.99))

    def features(self, im):
        x1 = self.s1(self.s0(im))
        p3 = self.s2(x1)
        p4 = self.s3(p3)
        p5 = self.s4(p4)
        l5 = self.lat5(p5)
        t4 = self.td4(torch.cat((self.up(l5), p4), 1))
        l4 = self.lat4(t4)
        o3 = self.td3(torch.cat((self.up(l4), p3), 1))
        o4 = self.bu4(torch.cat((self.down3(o3), l4), 1))
        o5 = self.bu5(torch.cat((self.down4(o4), l5), 1))
        return [o3, o4, o5]

    def forward(self, im):
        return self.detect(self.features(im))

    @torch.no_grad()
    def calibrate(self, im, conf_thres=0.25, fg_frac=0.03):
        """Shift the head's objectness bias so that about fg_frac of the anchors have obj > conf_thres on `im`, and the class
        biases so that a passing anchor carries one or two confident classes (random weights; timing only)."""
        # random weights + BatchNorm in eval mode (running mean 0 / var 1) let the signal fade layer by layer until every anchor
        # carries the same logits: first give the BN layers the statistics of real passes (cumulative average, two batches)
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        for m in bns:
            m.reset_running_stats()
            m.momentum = None
        self.train()
        for k in range(2):
            self.features(im.roll(k, 0).flip(3) if k else im)
        self.eval()
        feats = self.features(im)
        na, no, nc = self.detect.na, self.detect.no, self.nc
        lg = torch.logit(torch.tensor(conf_thres)).item()
        for i, f in enumerate(feats):
            conv = self.detect.m[i]
            w = conv.weight.view(na, no, -1)
            b = conv.bias.view(na, no)
            raw = conv(f).float().view(f.shape[0], na, no, -1)
            # spread of the logits like a trained head's (S-pred draws them from N(0, 1.5)): scale the objectness / class rows
            # ... and the four box rows: a trained head regresses sizes within a factor of ~3 of its anchors (t_wh within +-1: std 0.2 here, random features have heavy tails); the
            # raw spread of random weights puts many sides below 1 px, boxes the reference's fp32 clip treats erratically from far
            # away (csrc/riou_device.h: rbox_pair_well_conditioned) -- each of them forces its image onto the single-list path with an
            # exact clip against every box of every class (measured: 1.6 ms of NMS kernel per batch instead of 0.07)
            for sl, target in ((slice(0, 4), 0.2), (slice(4, 5), 1.5), (slice(5, 5 + nc), 1.0)):
                sd = raw[:, :, sl].std().item()
                g = target / max(sd, 1e-6)
                mu = raw[:, :, sl].mean().item()
                w.data[:, sl] *= g
                b.data[:, sl] = (b.data[:, sl].float() - mu) * g      # centred; the quantile shifts below place the thresholds
            raw = conv(f).float().view(f.shape[0], na, no, -1)
            b.data[:, 0:4] -= raw[:, :, 0:4].mean((0, 3)).to(b.dtype)   # the box rows centred per anchor and channel
            raw = conv(f).float().view(f.shape[0], na, no, -1)
            obj = raw[:, :, 4].flatten()
            q = torch.quantile(obj[: 1 << 20], 1.0 - fg_frac).item()
            b.data[:, 4] += lg - q                                     # fg_frac of the anchors pass obj > conf_thres
            cls = raw[:, :, 5:5 + nc].flatten()
            qc = torch.quantile(cls[: 1 << 20], 1.0 - 1.2 / nc).item()
            b.data[:, 5:5 + nc] += 0.0 - qc                            # ~1.2 classes per anchor above sigmoid 0.5: one or two labels per passing anchor


class SyntheticVal:
    """A loader with the item layout of LoadImagesAndLabels.collate_fn (utils/datasets.py:676-681): (im uint8 (b,3,h,w),
    targets (n, 7) [img cls cx cy l s theta], paths, shapes); the images are drawn once on the host and reused."""

    def __init__(self, n_images, batch, imgsz=1024, nc=16, labels_per_image=23, seed=0, distinct=4):
        g = torch.Generator().manual_seed(seed)
        self.batch, self.n, self.imgsz = batch, n_images, imgsz
        self.pool = [torch.randint(0, 256, (batch, 3, imgsz, imgsz), dtype=torch.uint8, generator=g).pin_memory()
                     if torch.cuda.is_available() else torch.randint(0, 256, (batch, 3, imgsz, imgsz), dtype=torch.uint8, generator=g)
                     for _ in range(distinct)]
        self.targets = []
        for _ in range(distinct):
            t = []
            for b in range(batch):
                k = labels_per_image
                cxy = torch.rand(k, 2, generator=g) * imgsz
                l = torch.rand(k, 1, generator=g) * 100 + 20
                s = torch.rand(k, 1, generator=g) * 20 + 8
                th = (torch.rand(k, 1, generator=g) - 0.5) * 3.141592
                cls = torch.randint(0, nc, (k, 1), generator=g).float()
                t.append(torch.cat((torch.full((k, 1), float(b)), cls, cxy, l, s, th), 1))
            self.targets.append(torch.cat(t, 0))
        self.dataset = range(n_images)

    def __iter__(self):
        done = 0
        i = 0
        while done < self.n:
            b = min(self.batch, self.n - done)
            im, tg = self.pool[i % len(self.pool)][:b], self.targets[i % len(self.pool)]
            tg = tg[tg[:, 0] < b]
            shapes = [((self.imgsz, self.imgsz), ((1.0, 1.0), (0.0, 0.0)))] * b
            yield im, tg, [f"synthetic_{done + k}.png" for k in range(b)], shapes
            done += b
            i += 1


def val_buckets(device, n_images=160, batch=16, nc=16, conf_thres=0.25, iou_thres=0.45, half=True, seed=0):
    """Run the product's val_sharded.run over this rank's synthetic shard; returns the reference's buckets in ms/img."""
    from yolov5_obb_amd import val_sharded
    # (MIOpen's exhaustive search -- torch.backends.cudnn.benchmark -- and a Focus-form stem were measured: the inference bucket stays
    #  at 0.47-0.50 ms per image, ~88 TFLOP/s of fp16 convolutions; the search only adds ~14 s to the run)
    torch.manual_seed(seed)
    model = StandinV5s(nc).to(device).eval()
    if half:
        model = model.half()
    loader = SyntheticVal(n_images, batch, nc=nc, seed=seed)
    im0 = (next(iter(loader))[0].to(device).half() if half else next(iter(loader))[0].to(device).float()) / 255
    model.calibrate(im0, conf_thres)
    with torch.no_grad():
        for _ in range(3):                                        # MIOpen picks its kernels on the first passes
            model(im0)
    torch.cuda.synchronize(device)
    # one untimed pass over two batches: the NMS driver sizes its candidate slots / workspace on its first calls
    val_sharded.run(model, SyntheticVal(2 * batch, batch, nc=nc, seed=seed), conf_thres=conf_thres, iou_thres=iou_thres, half=half, device=device)
    # collect=False: no collective in here -- a rank that fails returns an error while the others would wait in the gather;
    # bench.py reduces the buckets itself, outside any try block
    # the library's stage events around the timed loop: what the NMS bucket of this (random-init) workload is made of
    import ctypes as C
    from yolov5_obb_amd import _lib
    L = _lib.lib()
    L.obb_profile_enable(1)
    res = val_sharded.run(model, loader, conf_thres=conf_thres, iou_thres=iou_thres, half=half, device=device, collect=False)
    torch.cuda.synchronize(device)
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 8)
    L.obb_profile_enable(0)
    names = ("decode", "sort", "prep", "nms_kernel", "gather")
    stages = {names[i]: round(ms[i] / max(1, cnt[i]), 4) for i in range(5)}
    stages["calls"] = int(cnt[0])
    with torch.no_grad():
        z = model(im0)[0]
        n_pass = int((z[..., 4] > conf_thres).sum()) // max(1, im0.shape[0])
    return {"images_per_rank": int(res["seen"]), "batch": batch, "anchors_passing_obj_per_image": n_pass,
            "model": "yolov5s-shaped conv stand-in (tools/conv_standin.py, random init, fp16) + product Detect",
            "dt_seconds": [float(x) for x in res["dt"]], "nms_stages_ms_per_batch": stages}
