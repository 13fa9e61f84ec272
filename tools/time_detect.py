"""Detect inference decode timing (development aid): bs 16, 1024^2, nc 15, conv outputs resident."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from tests import synth
from yolov5_obb_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
for dtype, code in ((torch.float16, 1), (torch.float32, 0)):
    bs, na, nc = 16, 3, 15
    no = 5 + nc + 180
    sizes = (128, 64, 32)
    convs = [torch.randn(bs, na * no, n, n, device=dev, dtype=dtype) for n in sizes]
    a_total = sum(na * n * n for n in sizes)
    z = torch.empty(bs, a_total, no, device=dev, dtype=dtype)
    xs = [torch.empty(bs, na, n, n, no, device=dev, dtype=dtype) for n in sizes]
    px = [(synth.grid_anchors()[i] * synth.DEFAULT_STRIDES[i]).reshape(-1).tolist() for i in range(3)]
    arrs = [(C.c_float * 6)(*p) for p in px]
    def run():
        off = 0
        for i, n in enumerate(sizes):
            rc = L.obb_detect_decode(_lib.ptr(convs[i]), code, bs, na, no, n, n, C.cast(arrs[i], C.c_void_p), synth.DEFAULT_STRIDES[i],
                                     _lib.ptr(xs[i]), _lib.ptr(z), a_total, off, _lib.stream_ptr(dev))
            assert rc == 0
            off += na * n * n
    nl = len(sizes)
    conv_arr = (C.c_void_p * nl)(*[c.data_ptr() for c in convs])
    xs_arr = (C.c_void_p * nl)(*[t.data_ptr() for t in xs])
    ny_arr = (C.c_int64 * nl)(*sizes)
    px_arr = (C.c_float * (nl * 6))(*[v for p in px for v in p])
    st_arr = (C.c_float * nl)(*synth.DEFAULT_STRIDES[:nl])
    def run_levels():
        rc = L.obb_detect_decode_levels(nl, conv_arr, code, bs, na, no, ny_arr, ny_arr, px_arr, st_arr, xs_arr, _lib.ptr(z), a_total, None,
                                        _lib.stream_ptr(dev))
        assert rc == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): run_levels()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): run_levels()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{str(dtype):14s} detect decode (all levels, one launch): {ms:.3f} ms  -> {3 * z.numel() * z.element_size() / ms / 1e6:.0f} GB/s")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = 3 * z.numel() * z.element_size()
    print(f"{str(dtype):14s} detect decode (3 levels): {ms:.3f} ms  -> {nbytes / ms / 1e6:.0f} GB/s of 1 read + 2 writes ({nbytes/1e6:.0f} MB)")
    # the reference's op chain on the same tensors (torch ops, PyTorch-ROCm)
    anchors = synth.grid_anchors().to(dev)
    def ref():
        zs = []
        for i, c in enumerate(convs):
            n = sizes[i]
            x = c.view(bs, na, no, n, n).permute(0, 1, 3, 4, 2).contiguous()
            yv, xv = torch.meshgrid(torch.arange(n, device=dev), torch.arange(n, device=dev), indexing='ij')
            grid = torch.stack((xv, yv), 2).expand(1, na, n, n, 2).float()
            ag = (anchors[i] * synth.DEFAULT_STRIDES[i]).view(1, na, 1, 1, 2).expand(1, na, n, n, 2).float()
            y = x.sigmoid()
            y[..., 0:2] = (y[..., 0:2] * 2 - 0.5 + grid) * synth.DEFAULT_STRIDES[i]
            y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * ag
            zs.append(y.view(bs, -1, no))
        return torch.cat(zs, 1)
    for _ in range(2): ref()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5): zr = ref()
    e1.record(); torch.cuda.synchronize()
    print(f"{str(dtype):14s} same chain as torch ops (models/yolo.py:61-79): {e0.elapsed_time(e1) / 5:.3f} ms")
