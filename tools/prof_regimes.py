"""Four 100k regimes: wall time (events) and, with OBB_NMS_PHASE_PROF=1, the in-kernel phase split.
Development aid; bench.py is the contract."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from tests import synth
from yolov5_obb_amd import _lib, nms_rotated_ext
L = _lib.lib()


def stages():
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 8)
    return [ms[i] / max(1, cnt[i]) for i in range(8)]

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000


def regimes(n):
    d, s = synth.s_clustered(n, 300, 0); yield "clustered_k300", d, s
    d2, _ = synth.with_classes(d, 18, 0); yield "clustered_k300_18cls", d2, s
    d, s = synth.s_clustered(n, 3000, 0); yield "clustered_k3000", d, s
    d, s = synth.s_uniform(n, 0); yield "uniform", d, s


for name, d, s in regimes(N):
    d, s = d.to(dev), s.to(dev)
    k = nms_rotated_ext.nms_rotated(d, s, 0.4)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); k = nms_rotated_ext.nms_rotated(d, s, 0.4); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    L.obb_profile_enable(1)
    for _ in range(5):
        k = nms_rotated_ext.nms_rotated(d, s, 0.4)
    torch.cuda.synchronize()
    st = stages()
    L.obb_profile_enable(0)
    print(f"{name:24s} n={len(d):7d} kept={len(k):6d}  min {min(ts):8.3f} ms  med {sorted(ts)[len(ts)//2]:8.3f} ms | sort {st[5]*1e3:6.1f} prep {st[6]*1e3:6.1f} steps {st[7]*1e3:7.1f} us", flush=True)
