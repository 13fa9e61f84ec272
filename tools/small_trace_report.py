"""Per-workgroup timeline of k_nms_small on one bench tensor, from the stamps a -DOBB_SMALL_TRACE library keeps (tools/small_trace.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import synth
from yolov5_obb_amd import _lib
from yolov5_obb_amd.utils.general import non_max_suppression_obb
dev = torch.device("cuda:0")
pred = synth.s_pred(16, 64512, 16, seed=int(os.environ.get("SEED", "1000")), n_obj=int(os.environ.get("N_OBJ", "120")), fg_frac=float(os.environ.get("FG_FRAC", "0.03")), device=dev, dtype=torch.float16)
kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
L = C.CDLL(_lib.LIB_PATH)
seg = np.zeros(2048 * 16, dtype=np.uint64); tail = np.zeros(2048 * 4, dtype=np.uint64)
for i in range(5):
    out = non_max_suppression_obb(pred, **kw)
    torch.cuda.synchronize()
    tr2 = np.zeros(2048 * 8, dtype=np.uint64)
    L.obb_debug_small_trace2(tr2.ctypes.data_as(C.c_void_p))
    rc = L.obb_debug_small_trace(seg.ctypes.data_as(C.c_void_p), tail.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
seg = seg.reshape(2048, 16).astype(np.int64); tail = tail.reshape(2048, 4).astype(np.int64)
idle = [(b, int(r[1])) for b, r in enumerate(seg) if r[0] == 1]
recs = []
for b, r in enumerate(seg):
    if r[0] not in (2, 3): continue
    d = dict(blk=b, seg=int(r[1] & 0xffffff), part=int((r[1] >> 24) & 15), np=int(r[1] >> 28), n=int(r[2]), kept=int(r[3]), notlast=r[0] == 2,
             t=[int(x) for x in r[4:11]], nit=int(r[11]), d0=int(r[12]), d1=int(r[13]), d2=int(r[14]))
    d["x"] = [int(v) for v in tr2.reshape(2048, 8)[b]]
    d["start"] = d["t"][0]; d["end"] = d["t"][4] if d["notlast"] else d["t"][6]
    recs.append(d)
tails = [(int(r[0]) - 1, int(r[1]), int(r[2]), int(r[3])) for r in tail if r[0] > 0]
t0 = min([d["start"] for d in recs] + [t for _, t in idle])
us = lambda x: x / 100.0
print(f"HELPERS={os.environ.get('OBB_NMS_SMALL_HELPERS', '-')}: {len(recs)} workgroups with work, {len(idle)} idle helpers"
      + (f" (exits {us(min(t for _, t in idle) - t0):.1f} .. {us(max(t for _, t in idle) - t0):.1f} us)" if idle else ""))
st = sorted(us(d["start"] - t0) for d in recs)
print("starts (us): min %.1f median %.1f p90 %.1f max %.1f" % (st[0], st[len(st) // 2], st[int(len(st) * .9)], st[-1]))
print("the 24 workgroups that end last (us):")
for d in sorted(recs, key=lambda d: -d["end"])[:24]:
    t = d["t"]
    tailtxt = ("publish+exit %5.1f" % us(t[4] - t[3])) if d["notlast"] else ("merge %5.1f scan %4.1f out %4.1f" % (us(t[4] - t[3]), us(t[5] - t[4]), us(t[6] - t[5])))
    print("  blk %4d seg %4d part %d/%d n %3d start %5.1f load %4.1f items %5.1f (%2d) leftovers %5.1f drains %d %d %d %s end %6.1f" % (
        d["blk"], d["seg"], d["part"], d["np"], d["n"], us(d["start"] - t0), us(t[1] - t[0]), us(t[2] - t[1]), d["nit"], us(t[3] - t[2]), d["d0"], d["d1"], d["d2"], tailtxt, us(d["end"] - t0)))
    x = d["x"]
    print("        wave 0: in-item quick drains %.1f us, draws %.1f us | leftovers: quick %.1f interval %.1f clip %.1f us" % (us(x[0]), us(x[3]), us(x[4]), us(x[5]), us(x[6]))
          + ("; waiting at the pooled interval pass: %d for the interval, %d for the clip" % ((x[7] - 1) & 0xfffff, (x[7] - 1) >> 20) if x[7] else ""))
print("image tails (end us, image, rows, tail us):", [(round(us(e - t0), 1), g, n, round(us(e - s), 1)) for g, n, s, e in sorted(tails, key=lambda x: x[3])][-8:])
big = sorted(set(d["n"] for d in recs if d["n"] > 128))
print("segment sizes > 128:", big, " helpers wanted:", sum(d["np"] - 1 for d in recs if d["part"] == 0 and d["np"] > 1) if any(d["np"] > 1 for d in recs) else "-")
ns = sorted(d["n"] for d in recs if d["part"] == 0)
print("sizes: median %d p90 %d max %d; segments %d" % (ns[len(ns) // 2], ns[int(len(ns) * .9)], ns[-1], len(ns)))

w1 = sorted(((d["x"][7] - 1) & 0xfffff) for d in recs if d["x"][7]); w2 = sorted(((d["x"][7] - 1) >> 20) for d in recs if d["x"][7])
if w1: print("pairs waiting at the pooled interval pass, over %d segments: interval median %d p90 %d max %d; clip median %d p90 %d max %d" % (len(w1), w1[len(w1) // 2], w1[int(len(w1) * .9)], w1[-1], w2[len(w2) // 2], w2[int(len(w2) * .9)], w2[-1]))
