"""Random differential check of the pairwise IoU entry points against the C oracle (development aid): rotated IoU matrices
(bit-exact up to the documented last-bit sin / cos difference) and quad IoU matrices (bit-exact, the proved and the searched skip rules
included) on random sizes, extents and box shapes."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from tests import synth
from yolov5_obb_amd import ops
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
bad = 0
for it in range(N):
    n = rng.choice([0, 1, 5, 64, 257, 1000, 3000]); m = rng.choice([1, 3, 64, 100, 1000])
    ext = rng.choice([8.0, 256.0, 1024.0, 4096.0, 30000.0, 60000.0])
    k = rng.choice([2, 30, 300])
    mk = (lambda cnt, sd: synth.s_clustered(cnt, k, seed=sd, extent=ext)[0]) if rng.random() < 0.6 else (lambda cnt, sd: synth.s_uniform(cnt, seed=sd, extent=ext)[0])
    a = mk(max(n, 1), seed * 100000 + it)[:n]; b = mk(m, seed * 100000 + it + 50000)
    if rng.random() < 0.3 and n > 2:
        b[: min(m, n)] = a[: min(m, n)]                                   # identical boxes: union == intersection
    if rng.random() < 0.3:
        b[:, 2:4] *= rng.choice([0.01, 0.2, 5.0, 40.0])
    if rng.random() < 0.2 and m > 2:
        b[0, 2] = 0.0; b[1, 3] = 0.0                                       # zero-area boxes
    kind = rng.choice(["quad", "quad", "rot"])
    if kind == "quad":
        qa, qb = synth.rbox_to_quad(a).contiguous(), synth.rbox_to_quad(b).contiguous()
        if rng.random() < 0.5 and n > 0:                                 # reversed rings
            qa[::7] = qa[::7].reshape(-1, 4, 2).flip(1).reshape(-1, 8); qb[::5] = qb[::5].reshape(-1, 4, 2).flip(1).reshape(-1, 8)
        got = ops.quad_iou_matrix(qa.to(dev), qb.to(dev)).cpu().numpy() if n else np.zeros((0, m), np.float32)
        ref = oracle.piou_matrix(qa.numpy(), qb.numpy()) if n else np.zeros((0, m), np.float32)
        ok = got.shape == ref.shape and bool(((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))).all())
        worst = float(np.nanmax(np.abs(got - ref))) if got.size else 0.0
    else:
        got = ops.rotated_iou_matrix(a.to(dev), b.to(dev)).cpu().numpy() if n else np.zeros((0, m), np.float32)
        ref = oracle.riou_matrix(a.numpy(), b.numpy()) if n else np.zeros((0, m), np.float32)
        # (ocml against glibc sin / cos: a last-bit difference of a hoisted feature can move an IoU by a few ulp -- tests/test_iou_gpu.py's bar)
        worst = float(np.abs(got - ref).max()) if got.size else 0.0
        ok = got.shape == ref.shape and worst <= 2e-6 and (got.size == 0 or (got.view(np.uint32) == ref.view(np.uint32)).mean() > 0.999)
    if not ok:
        bad += 1
        print("MISMATCH", it, dict(kind=kind, n=n, m=m, ext=ext, k=k), "max |diff|", worst, "differing", int((got.view(np.uint32) != ref.view(np.uint32)).sum()) if got.shape == ref.shape else "shape", flush=True)
print(f"iou_fuzz seed {seed}: {N} cases, {bad} mismatches", flush=True)
