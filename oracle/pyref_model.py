"""TEST INFRASTRUCTURE -- restatement of the reference's yolov5n OBB network for the `detect.py --device cpu` baseline.

bench.py times the reference's CPU detect path on the GPU box's host cores; /root/reference does not
exist there, so the network of models/yolov5n.yaml (depth_multiple 0.33, width_multiple 0.25; modules of
models/common.py:37-49 Conv, :94-104 Bottleneck, :126-138 C3, :181-196 SPPF; head models/yolo.py:33-92) is restated here
in plain torch with random-init weights -- timing only, no checkpoint exists offline.  tests/test_oracle_vs_ref.py checks,
in the build container, that it has the reference model's parameter count and output shapes.  Never imported by the product."""
import torch
import torch.nn as nn

from . import pyref


class Conv(nn.Module):                      # models/common.py:37-49
    def __init__(self, c1, c2, k=1, s=1, p=None):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2 if p is None else p, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):                # :94-104
    def __init__(self, c1, c2, shortcut=True):
        super().__init__()
        self.cv1, self.cv2 = Conv(c1, c2, 1, 1), Conv(c2, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C3(nn.Module):                        # :126-138
    def __init__(self, c1, c2, n=1, shortcut=True):
        super().__init__()
        c_ = c2 // 2
        self.cv1, self.cv2, self.cv3 = Conv(c1, c_, 1, 1), Conv(c1, c_, 1, 1), Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class SPPF(nn.Module):                      # :181-196
    def __init__(self, c1, c2, k=5):
        super().__init__()
        self.cv1, self.cv2 = Conv(c1, c1 // 2, 1, 1), Conv(c1 * 2, c2, 1, 1)
        self.m = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.m(x)
        y2 = self.m(y1)
        return self.cv2(torch.cat((x, y1, y2, self.m(y2)), 1))


ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]   # models/yolov5n.yaml:7-10
STRIDES = [8.0, 16.0, 32.0]


class YoloV5nObb(nn.Module):
    """models/yolov5n.yaml with the OBB head (no = nc + 185): widths 16..256, C3 depths 1/2/3/1, PANet neck."""

    def __init__(self, nc=16, ch=3):
        super().__init__()
        w = [16, 32, 64, 128, 256]
        self.b0, self.b1, self.b2 = Conv(ch, w[0], 6, 2, 2), Conv(w[0], w[1], 3, 2), C3(w[1], w[1], 1)
        self.b3, self.b4 = Conv(w[1], w[2], 3, 2), C3(w[2], w[2], 2)
        self.b5, self.b6 = Conv(w[2], w[3], 3, 2), C3(w[3], w[3], 3)
        self.b7, self.b8, self.b9 = Conv(w[3], w[4], 3, 2), C3(w[4], w[4], 1), SPPF(w[4], w[4], 5)
        self.h10, self.h13 = Conv(w[4], w[3], 1, 1), C3(w[4], w[3], 1, False)
        self.h14, self.h17 = Conv(w[3], w[2], 1, 1), C3(w[3], w[2], 1, False)
        self.h18, self.h20 = Conv(w[2], w[2], 3, 2), C3(w[3], w[3], 1, False)
        self.h21, self.h23 = Conv(w[3], w[3], 3, 2), C3(w[4], w[4], 1, False)
        self.up = nn.Upsample(None, 2, 'nearest')
        self.nc, self.no, self.na = nc, nc + 185, 3
        self.m = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in (w[2], w[3], w[4]))   # Detect.m (models/yolo.py:46)
        self.anchors = torch.tensor(ANCHORS).float().view(3, 3, 2) / torch.tensor(STRIDES).view(3, 1, 1)
        self.strides = torch.tensor(STRIDES)
        import math
        for mi, st in zip(self.m, STRIDES):                  # Model._initialize_biases (models/yolo.py:224-231)
            b = mi.bias.view(self.na, -1)
            b.data[:, 4] += math.log(8 / (640 / st) ** 2)    # obj (8 objects per 640 image)
            b.data[:, 5:] += math.log(0.6 / (nc - 0.999999))     # (the reference adds the class prior to the CSL channels as well)
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def forward(self, x):
        x = self.b2(self.b1(self.b0(x)))
        p3 = self.b4(self.b3(x))
        p4 = self.b6(self.b5(p3))
        p5 = self.b9(self.b8(self.b7(p4)))
        h10 = self.h10(p5)
        h13 = self.h13(torch.cat((self.up(h10), p4), 1))
        h14 = self.h14(h13)
        h17 = self.h17(torch.cat((self.up(h14), p3), 1))
        h20 = self.h20(torch.cat((self.h18(h17), h14), 1))
        h23 = self.h23(torch.cat((self.h21(h20), h10), 1))
        xs = []
        for i, f in enumerate((h17, h20, h23)):             # Detect.forward, inference branch (models/yolo.py:61-81)
            y = self.m[i](f)
            bs, _, ny, nx = y.shape
            xs.append(y.view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
        return pyref.detect_decode(xs, self.anchors, self.strides)
