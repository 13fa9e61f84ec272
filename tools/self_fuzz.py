"""Differential check of the self-sorting segments against the sort-kernel path over random shapes (development aid):
every case runs three times per mode (un-hinted, hinted, hinted) and the rows must be identical; every fourth case is also
compared with the oracle restatement."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from oracle import pyref
from yolov5_obb_amd.utils import general
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(N):
    bs = rng.choice([1, 2, 3, 5, 16]); nc = rng.choice([2, 3, 5, 15, 16, 18, 40, 100, 170, 200])
    A = rng.choice([700, 3000, 9000, 20000]); conf = rng.choice([0.05, 0.1, 0.25, 0.5]); half = rng.random() < 0.4
    multi = rng.random() < 0.7; n_obj = rng.choice([5, 40, 200, 800]); fg = rng.choice([0.01, 0.05, 0.2])
    md = rng.choice([7, 300, 1500]); iou = rng.choice([0.0, 0.2, 0.45, 0.9])
    pred = synth.s_pred(bs, A, nc, seed=1000 + it, n_obj=n_obj, fg_frac=fg, dtype=torch.float16 if half else torch.float32)
    if rng.random() < 0.3:                                   # a few sub-pixel / oversized boxes
        b = rng.randrange(bs); r = rng.randrange(A)
        pred[b, r, 3] = 0.4 if rng.random() < 0.5 else pred[b, r, 3]
        pred[b, r, 2] = 5000.0 if rng.random() < 0.5 else pred[b, r, 2]
        pred[b, r, 4] = 0.95
    kw = dict(conf_thres=conf, iou_thres=iou, multi_label=multi, max_det=md)
    if rng.random() < 0.25:
        kw["classes"] = sorted(rng.sample(range(nc), max(1, nc // 3)))
    if rng.random() < 0.15:
        kw["agnostic"] = True
    if rng.random() < 0.2:
        kw["labels"] = [torch.tensor([[rng.randrange(nc), 100., 120., 60., 20.], [rng.randrange(nc), 101., 121., 58., 21.]]) if (b % 2 == 0) else torch.zeros((0, 5)) for b in range(bs)]
    p = pred.to(dev)
    outs = {}
    MODES = os.environ.get("FUZZ_MODES", "0 2").split()
    for mode in MODES:
        os.environ["OBB_NMS_SELF_SORT"] = mode
        general.hints_clear()
        rs = [general.non_max_suppression_obb(p, **kw) for _ in range(3)]
        for r in rs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(rs[0], r)), ("call-to-call difference", it, mode)
        outs[mode] = [o.cpu() for o in rs[0]]
    same = all(torch.equal(a, b) for a, b in zip(outs[MODES[0]], outs[MODES[-1]]))
    ref_ok = True
    if it % int(os.environ.get("FUZZ_ORACLE_EVERY", "4")) == 0:
        ref = pyref.non_max_suppression_obb(pred.clone(), **kw)
        if half:     # equal-confidence groups: the reference's sort is unstable, rows are compared as sets inside them (tests/_cmp ties=True)
            ref_ok = all(a.shape == torch.as_tensor(b).shape and torch.equal(a[:, 5], torch.as_tensor(b)[:, 5]) and
                         __import__("numpy").array_equal(synth.canon_rows(a), synth.canon_rows(torch.as_tensor(b))) for a, b in zip(outs[MODES[-1]], ref))
        else:
            ref_ok = all(torch.equal(a, torch.as_tensor(b)) for a, b in zip(outs[MODES[-1]], ref))
    if not (same and ref_ok):
        bad += 1
        os.makedirs("gpurun_out/fuzz", exist_ok=True)
        torch.save({"pred": pred, "kw": kw, "gpu": outs[MODES[-1]], "gpu0": outs[MODES[0]]}, f"gpurun_out/fuzz/case_{sys.argv[1] if len(sys.argv) > 1 else 0}_{it}.pt")
        if not ref_ok:
            for b, (a_, r_) in enumerate(zip(outs[MODES[-1]], ref)):
                r_ = torch.as_tensor(r_)
                if a_.shape != r_.shape or not torch.equal(a_, r_):
                    print("   image", b, "gpu rows", tuple(a_.shape), "ref rows", tuple(r_.shape), "labels" in kw, flush=True)
                    k = min(a_.shape[0], r_.shape[0])
                    d = (a_[:k] != r_[:k]).any(1).nonzero()
                    if len(d): print("   first differing row", int(d[0]), a_[int(d[0])].tolist(), r_[int(d[0])].tolist(), flush=True)
        print("MISMATCH", it, dict(bs=bs, nc=nc, A=A, conf=conf, half=half, multi=multi, n_obj=n_obj, fg=fg, md=md, iou=iou, kw={k: v for k, v in kw.items() if k in ("classes",)}), same, ref_ok, flush=True)
print(f"self_fuzz: {N} cases, {bad} mismatches", flush=True)
