"""Mirror of the reference's ``utils/loss.py`` for the OBB head: ``ComputeLoss`` (utils/loss.py:90-275).

Same constructor, attributes (``hyp, gr, balance, autobalance, sort_obj_iou, ssi, na, nc, nl, anchors, stride, cp, cn``),
call signature and return values as the reference, so ``train.py:269,326`` and ``val.py:199`` run unchanged.  The work is
done by the loss entry points of libobb_hip.so (``obb_loss_forward`` / ``obb_loss_backward`` / ``obb_loss_build_targets``,
include/obb_hip.h) behind one ``torch.autograd.Function``: no per-level Python loop, no boolean-mask gathers, no host
synchronisation, one write of the gradient tensors.
"""
import ctypes as C

import torch

from .. import _lib

_MAX_LV, _MAX_NA, _CSL = 8, 8, 180


class _LossConfig(C.Structure):
    """obb_loss_config of include/obb_hip.h."""
    _fields_ = [("nl", C.c_int32), ("na", C.c_int32), ("nc", C.c_int32), ("no", C.c_int32), ("bs", C.c_int32),
                ("ny", C.c_int32 * _MAX_LV), ("nx", C.c_int32 * _MAX_LV),
                ("anchors", ((C.c_float * 2) * _MAX_NA) * _MAX_LV),
                ("stride", C.c_float * _MAX_LV), ("balance", C.c_float * _MAX_LV),
                ("anchor_t", C.c_float), ("cp", C.c_float), ("cn", C.c_float),
                ("cls_pw", C.c_float), ("theta_pw", C.c_float), ("obj_pw", C.c_float),
                ("gain_box", C.c_float), ("gain_obj", C.c_float), ("gain_cls", C.c_float), ("gain_theta", C.c_float),
                ("gr", C.c_float), ("sort_obj_iou", C.c_int32), ("csl_radius", C.c_float), ("fl_gamma", C.c_float)]


def smooth_BCE(eps=0.1):  # utils/loss.py:13-15
    # return positive, negative label smoothing BCE targets
    return 1.0 - 0.5 * eps, 0.5 * eps


def _is_parallel(model):  # utils/torch_utils.py:215-217
    return type(model) in (torch.nn.parallel.DataParallel, torch.nn.parallel.DistributedDataParallel)


def _dtype_code(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.float16:
        return 1
    raise RuntimeError(f"ComputeLoss: head outputs must be float32 or float16, got {t.dtype}")


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _ObbLossFn(torch.autograd.Function):
    """loss_out = obb_loss_forward(p, targets); d loss_out[0] / d p[i] = obb_loss_backward(...)."""

    @staticmethod
    def forward(ctx, owner, targets, *p):
        dev = p[0].device
        code = _dtype_code(p[0])
        for i, pi in enumerate(p):       # the kernels read every level with p[0]'s element size and on p[0]'s device
            if pi.dtype != p[0].dtype or pi.device != dev:
                raise RuntimeError(f"ComputeLoss: p[{i}] is {pi.dtype} on {pi.device}, p[0] is {p[0].dtype} on {dev}: "
                                   "all head outputs must share dtype and device")
        ps = [pi.contiguous() for pi in p]
        tg = targets.to(device=dev, dtype=torch.float32).contiguous()
        cfg = owner._config(ps)
        nt, tcols = int(tg.shape[0]), int(tg.shape[1]) if tg.dim() == 2 else 0
        L = _lib.lib()
        out = torch.empty(5 + _MAX_LV, dtype=torch.float32, device=dev)
        with _lib.guard(dev):
            nbytes = L.obb_loss_workspace_bytes(C.byref(cfg), nt)
            if nbytes == 0:
                raise RuntimeError("ComputeLoss: unsupported head configuration (see obb_loss_config in include/obb_hip.h)")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # owned by this call: backward reads it
            rc = L.obb_loss_forward(C.byref(cfg), _ptr_array(ps), code, _lib.ptr(tg), nt, tcols, _lib.ptr(out), _lib.ptr(ws),
                                    ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "obb_loss_forward")
        ctx.save_for_backward(tg, *ps)
        ctx.cfg, ctx.ws, ctx.code = cfg, ws, code
        ctx.mark_non_differentiable(out)
        return out[0:1].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gout):
        tg, *ps = ctx.saved_tensors
        dev = ps[0].device
        grads = [torch.empty_like(pi) for pi in ps]
        gscale = gloss.reshape(-1)[:1].to(torch.float32).contiguous()
        L = _lib.lib()
        with _lib.guard(dev):
            rc = L.obb_loss_backward(C.byref(ctx.cfg), _ptr_array(ps), ctx.code, _lib.ptr(tg), int(tg.shape[0]),
                                     int(tg.shape[1]) if tg.dim() == 2 else 0, _lib.ptr(gscale), _ptr_array(grads),
                                     _lib.ptr(ctx.ws), ctx.ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "obb_loss_backward")
        return (None, None, *grads)


class ComputeLoss:
    # Compute losses (utils/loss.py:90-192)
    def __init__(self, model, autobalance=False):
        self.sort_obj_iou = False
        device = next(model.parameters()).device  # get model device
        h = model.hyp  # hyperparameters
        if not device.type == 'cuda':
            raise RuntimeError("ComputeLoss: yolov5_obb_amd is compiled for MI355X only (no CPU path, by design)")
        self.device = device

        # Class label smoothing https://arxiv.org/pdf/1902.04103.pdf eqn 3
        self.cp, self.cn = smooth_BCE(eps=h.get('label_smoothing', 0.0))  # positive, negative BCE targets

        # Focal loss (utils/loss.py:107-110): FocalLoss(BCE, g) around the class, angle and objectness terms, inside the kernels
        self.fl_gamma = float(h.get('fl_gamma', 0.0))

        det = model.module.model[-1] if _is_parallel(model) else model.model[-1]  # Detect() module
        self.stride = det.stride  # tensor([8., 16., 32., ...])
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, 0.02])  # P3-P7
        self.ssi = list(self.stride).index(16) if autobalance else 0  # stride 16 index
        self.gr, self.hyp, self.autobalance = 1.0, h, autobalance
        for k in 'na', 'nc', 'nl', 'anchors':
            setattr(self, k, getattr(det, k))
        if self.nl > _MAX_LV or self.na > _MAX_NA:
            raise RuntimeError(f"ComputeLoss: at most {_MAX_LV} levels x {_MAX_NA} anchors supported")
        self._det = det
        self._host_key = None
        self._refresh_host_tables()

    def _refresh_host_tables(self):
        """Host copies of det.anchors / det.stride for the kernel configuration.  The reference reads the tensors on every
        call, so an in-place update after construction (autoanchor's `m.anchors[:] = ...`, check_anchor_order) is picked
        up there; here the copy is rebuilt whenever the tensors' version counters or storage change."""
        a, st = self._det.anchors, self._det.stride
        key = (a._version, a.data_ptr(), st._version, st.data_ptr())
        if key != self._host_key:
            self.anchors, self.stride = a, st
            self._anchors_host = a.detach().float().cpu().reshape(self.nl, self.na, 2).tolist()
            self._stride_host = [float(s) for s in st.detach().float().cpu().tolist()]
            self._host_key = key

    # ---- plumbing
    def _config(self, p):
        h = self.hyp
        self._refresh_host_tables()
        cfg = _LossConfig()
        cfg.nl, cfg.na, cfg.nc = int(self.nl), int(self.na), int(self.nc)
        cfg.no = int(p[0].shape[-1])
        cfg.bs = int(p[0].shape[0])
        if len(p) != self.nl:
            raise RuntimeError(f"ComputeLoss: expected {self.nl} head outputs, got {len(p)}")
        for i, pi in enumerate(p):
            if pi.dim() != 5 or pi.shape[1] != self.na or pi.shape[-1] != 5 + self.nc + _CSL or pi.shape[0] != cfg.bs:
                raise RuntimeError(f"ComputeLoss: p[{i}] must be (bs, {self.na}, ny, nx, {5 + self.nc + _CSL}), got {tuple(pi.shape)}")
            cfg.ny[i], cfg.nx[i] = int(pi.shape[2]), int(pi.shape[3])
            cfg.stride[i] = self._stride_host[i]
            cfg.balance[i] = float(self.balance[i])
            for a in range(self.na):
                cfg.anchors[i][a][0], cfg.anchors[i][a][1] = self._anchors_host[i][a]
        cfg.anchor_t = float(h['anchor_t'])
        cfg.cp, cfg.cn = float(self.cp), float(self.cn)
        cfg.cls_pw, cfg.theta_pw, cfg.obj_pw = float(h['cls_pw']), float(h['theta_pw']), float(h['obj_pw'])
        cfg.gain_box, cfg.gain_obj, cfg.gain_cls, cfg.gain_theta = float(h['box']), float(h['obj']), float(h['cls']), float(h['theta'])
        cfg.gr = float(self.gr)
        cfg.sort_obj_iou = int(bool(self.sort_obj_iou))
        cfg.csl_radius = float(h.get('csl_radius', 2.0))      # only used for (nt,7) targets: labels regenerated on the device
        cfg.fl_gamma = self.fl_gamma if self.fl_gamma > 0 else 0.0
        return cfg

    def __call__(self, p, targets):  # predictions, targets, model
        """
        Args:
            p (list[P3_out,...]): torch.Size(b, self.na, h_i, w_i, self.no), self.na means the number of anchors scales
            targets (tensor): (n_gt_all_batch, [img_index clsid cx cy l s theta gaussian_θ_labels])
        Return：
            total_loss * bs (tensor): [1]
            torch.cat((lbox, lobj, lcls, ltheta)).detach(): [4]
        """
        for pi in p:
            _lib.require_cuda(pi, "p[i]")
        loss, out = _ObbLossFn.apply(self, targets, *p)
        if self.autobalance:  # utils/loss.py:180-184 (host read of the per-level objectness loss, as in the reference)
            obji = out[5:5 + self.nl].tolist()
            self.balance = [self.balance[i] * 0.9999 + 0.0001 / obji[i] for i in range(self.nl)]
            self.balance = [x / self.balance[self.ssi] for x in self.balance]
        return loss, out[1:5].detach()

    def build_targets(self, p, targets):
        """utils/loss.py:194-275.  Returns tcls, tbox, indices, anch, tgaussian_theta (lists over levels) with the
        reference's row order (offset-major, anchor-major, target order)."""
        for pi in p:
            _lib.require_cuda(pi, "p[i]")
        dev = p[0].device
        cfg = self._config(p)
        tg = targets.to(device=dev, dtype=torch.float32).contiguous()
        nt, tcols = int(tg.shape[0]), int(tg.shape[1]) if tg.dim() == 2 else 0
        L = _lib.lib()
        counts = torch.zeros(_MAX_LV + 1, dtype=torch.int32, device=dev)
        tcls, tbox, indices, anch, tgt = [], [], [], [], []
        with _lib.guard(dev):
            ws = torch.empty(L.obb_loss_workspace_bytes(C.byref(cfg), nt), dtype=torch.uint8, device=dev)
            if nt:
                _lib.check(L.obb_loss_build_targets(C.byref(cfg), _lib.ptr(tg), nt, tcols, _lib.ptr(counts), _lib.ptr(ws),
                                                    ws.numel(), _lib.stream_ptr(dev)), "obb_loss_build_targets")
            cnt = counts.tolist()
            if cnt[_MAX_LV]:
                raise IndexError("build_targets: a target row names an image or class outside the batch")
            for i in range(self.nl):
                n = cnt[i]
                idx = torch.empty((n, 4), dtype=torch.int64, device=dev)
                tb = torch.empty((n, 4), dtype=torch.float32, device=dev)
                an = torch.empty((n, 2), dtype=torch.float32, device=dev)
                tc = torch.empty((n,), dtype=torch.int64, device=dev)
                cs = torch.empty((n, _CSL), dtype=torch.float32, device=dev)
                if n:
                    _lib.check(L.obb_loss_export_targets(C.byref(cfg), nt, i, n, _lib.ptr(idx), _lib.ptr(tb), _lib.ptr(an),
                                                         _lib.ptr(tc), _lib.ptr(cs), _lib.ptr(ws), ws.numel(),
                                                         _lib.stream_ptr(dev)), "obb_loss_export_targets")
                b, a, gj, gi = idx.unbind(1)
                indices.append((b, a, gj, gi))
                tbox.append(tb); anch.append(an); tcls.append(tc); tgt.append(cs)
        return tcls, tbox, indices, anch, tgt
