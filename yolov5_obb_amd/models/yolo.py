"""Mirror of the OBB ``Detect`` head of the reference's ``models/yolo.py`` (:33-92).

Only the head is mirrored: ``Model`` / ``parse_model`` / the CSP backbone stay the reference's PyTorch code (they run on
PyTorch-ROCm unchanged).  This class keeps the reference's constructor, attributes (``nc no nl na anchors m grid
anchor_grid stride inplace onnx_dynamic``), parameter names and return values, so checkpoints and ``parse_model`` can use it
in place of the reference's ``Detect``.  In inference mode on the GPU the per-level chain
``view -> permute -> contiguous -> sigmoid -> 2 slice updates -> cat`` is ONE pass of ``obb_detect_decode_col``
(libobb_hip.so, csrc/head.hip) per level.

Coupling with the NMS (the next stage of val.py / detect.py): the same pass stores the objectness column ``z[..., 4]`` once
more as a dense (bs, A) tensor and hangs it on the returned ``z`` (``z._obb_objcol = (column, z._version)``).
``utils.general.non_max_suppression_obb`` reads its confidence filter from that column -- 2 bytes per anchor instead of one
128-byte line of every 400-byte row -- when it is handed this very tensor object, unmodified (same ``_version``); any other
tensor (a clone, a cast, the TTA concatenation, an in-place edit) takes the plain path.  The results are identical.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


class Detect(nn.Module):
    stride = None  # strides computed during build
    onnx_dynamic = False  # ONNX export parameter
    # host copy of anchors * stride as plain Python lists (picklable; absent in checkpoints written by the reference: the
    # class default makes such modules work without a cast first)
    _anchor_px = None
    # inference on CPU tensors: this package has no CPU path.  yolov5_obb_amd.dropin.install() points this at the
    # reference's own Detect.forward (models/yolo.py:50-81), so that `detect.py --device cpu` keeps running its code.
    _cpu_forward = None
    # store z[..., 4] densely next to z for the confidence filter of non_max_suppression_obb (module docstring)
    couple_nms = True
    fused_levels = True      # all levels decoded by one launch (obb_detect_decode_levels); False: one launch per level

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):  # detection layer
        super().__init__()
        self.nc = nc  # number of classes
        self.no = nc + 5 + 180  # number of outputs per anchor (models/yolo.py:40)
        self.nl = len(anchors)  # number of detection layers
        self.na = len(anchors[0]) // 2  # number of anchors
        self.grid = [torch.zeros(1)] * self.nl  # init grid
        self.anchor_grid = [torch.zeros(1)] * self.nl  # init anchor grid
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))  # shape(nl,na,2)
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)  # output conv
        self.inplace = inplace  # use in-place ops (e.g. slice assignment)

    @staticmethod
    def _tensor_key(t):
        """(version, storage address) of a tensor; inference tensors track no version counter: None = "cannot be cached"."""
        try:
            return (t._version, t.data_ptr())
        except RuntimeError:
            return None

    def _host_tables(self):
        # The reference reads anchors / stride on every call (models/yolo.py:90-91), so an in-place update after the first
        # inference (autoanchor's `m.anchors[:] = ...`, check_anchor_order) is picked up there; the host copy here is keyed
        # on the tensors' version counters and storage, like ComputeLoss._refresh_host_tables.
        stride_t = torch.as_tensor(self.stride)
        key = (self._tensor_key(self.anchors), self._tensor_key(stride_t))
        # (a tensor without a version counter -- created under torch.inference_mode() -- can change unnoticed: the small host table
        #  is then rebuilt on every call, as the reference re-reads the anchors on every call)
        if self._anchor_px is None or None in key or self._anchor_px[2] != key:
            st = [float(s) for s in stride_t.float().cpu().tolist()]
            an = self.anchors.detach().float().cpu()
            px = [(an[i] * st[i]).reshape(-1).tolist() for i in range(self.nl)]      # anchor_grid values (:90-91)
            self._anchor_px = (px, st, key)
        px, st = self._anchor_px[0], self._anchor_px[1]
        return [(C.c_float * len(p))(*p) for p in px], st      # ctypes arrays are built per call: they do not pickle

    def _apply(self, fn):  # anchors / stride may change (Model._apply, autoanchor): drop the host cache
        self._anchor_px = None
        return super()._apply(fn)

    def forward(self, x):
        """
        Args:
            x (list[P3_in,...]): torch.Size(b, c_i, h_i, w_i)
        Return：
            if train:
                x (list[P3_out,...]): torch.Size(b, self.na, h_i, w_i, self.no)
            else:
                inference (tensor): (b, n_all_anchors, self.no)
                x (list[P3_out,...]): torch.Size(b, self.na, h_i, w_i, self.no)
        """
        if self.training:
            for i in range(self.nl):
                x[i] = self.m[i](x[i])  # conv
                bs, _, ny, nx = x[i].shape
                x[i] = x[i].view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            return x

        if not x[0].is_cuda:
            if type(self)._cpu_forward is None:
                raise RuntimeError("Detect (inference): yolov5_obb_amd is compiled for MI355X only (no CPU path, by design); "
                                   "under yolov5_obb_amd.dropin.install() CPU tensors run the reference's own Detect.forward")
            return type(self)._cpu_forward(self, x)
        convs = [self.m[i](x[i]).contiguous() for i in range(self.nl)]
        c0 = convs[0]
        if c0.dtype == torch.float32:
            code = 0
        elif c0.dtype == torch.float16:
            code = 1
        else:
            raise RuntimeError(f"Detect (inference): float32 or float16 expected, got {c0.dtype}")
        anchor_px, strides = self._host_tables()
        bs = c0.shape[0]
        shapes = [(c.shape[2], c.shape[3]) for c in convs]
        a_total = sum(self.na * ny * nx for ny, nx in shapes)
        z = torch.empty((bs, a_total, self.no), dtype=c0.dtype, device=c0.device)
        col = torch.empty((bs, a_total), dtype=c0.dtype, device=c0.device) if self.couple_nms else None
        L = _lib.lib()
        for i, (ny, nx) in enumerate(shapes):
            if self.onnx_dynamic or self.grid[i].shape[2:4] != (ny, nx):
                self.grid[i], self.anchor_grid[i] = self._make_grid(nx, ny, i)      # kept for attribute compatibility
        xps = [torch.empty((bs, self.na, ny, nx, self.no), dtype=c0.dtype, device=c0.device) for ny, nx in shapes]
        with torch.cuda.device(c0.device):
            st = _lib.stream_ptr(c0.device)
            if self.nl <= 4 and self.fused_levels:
                # the loop over the levels and the torch.cat of models/yolo.py:61-79 as ONE launch
                nl = self.nl
                rc = L.obb_detect_decode_levels(nl, (C.c_void_p * nl)(*[c.data_ptr() for c in convs]), code, bs, self.na, self.no,
                                                (C.c_int64 * nl)(*[s[0] for s in shapes]), (C.c_int64 * nl)(*[s[1] for s in shapes]),
                                                (C.c_float * (nl * self.na * 2))(*[v for a in anchor_px for v in a]),
                                                (C.c_float * nl)(*strides[:nl]), (C.c_void_p * nl)(*[t.data_ptr() for t in xps]),
                                                _lib.ptr(z), a_total, _lib.ptr(col), st)
                _lib.check(rc, "obb_detect_decode_levels")
            else:
                off = 0
                for i in range(self.nl):
                    ny, nx = shapes[i]
                    rc = L.obb_detect_decode_col(_lib.ptr(convs[i]), code, bs, self.na, self.no, ny, nx, C.cast(anchor_px[i], C.c_void_p),
                                                 strides[i], _lib.ptr(xps[i]), _lib.ptr(z), a_total, off, _lib.ptr(col), st)
                    _lib.check(rc, "obb_detect_decode_col")
                    off += self.na * ny * nx
        for i in range(self.nl):
            x[i] = xps[i]
        if col is not None and not torch.is_inference(z):
            # read by utils.general.non_max_suppression_obb (module docstring).  Under torch.inference_mode() tensors carry no
            # version counter, so "unchanged since Detect wrote it" cannot be checked: the column is not attached and the NMS
            # scans z[..., 4] itself (same results).
            z._obb_objcol = (col, z._version)
        return z, x

    def _make_grid(self, nx=20, ny=20, i=0):  # models/yolo.py:83-92
        d = self.anchors[i].device
        yv, xv = torch.meshgrid([torch.arange(ny, device=d), torch.arange(nx, device=d)], indexing='ij')
        grid = torch.stack((xv, yv), 2).expand((1, self.na, ny, nx, 2)).float()
        anchor_grid = (self.anchors[i].clone() * self.stride[i]).view((1, self.na, 1, 1, 2)).expand((1, self.na, ny, nx, 2)).float()
        return grid, anchor_grid
