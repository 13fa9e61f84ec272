import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
def run(name, d, s, thr, reps=4):
    s = synth.tie_free(s)
    ref = oracle.nms_rotated(d.numpy(), s.numpy(), thr)
    order = np.argsort(-s.numpy(), kind="stable")
    rank = np.empty(len(s), dtype=np.int64); rank[order] = np.arange(len(s))
    for r in range(reps):
        k = nms_rotated_ext.nms_rotated(d.to(dev), s.to(dev), thr).cpu().numpy()
        if np.array_equal(k, ref):
            print(name, "rep", r, "OK kept", len(ref)); continue
        sk, sr = set(k.tolist()), set(ref.tolist())
        extra = sorted(rank[list(sk - sr)].tolist()); missing = sorted(rank[list(sr - sk)].tolist())
        print(name, "rep", r, f"MISMATCH kept {len(k)} ref {len(ref)} extra {len(extra)} missing {len(missing)}",
              "first extra ranks", extra[:8], "first missing ranks", missing[:8], flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
run("clustered", *synth.s_clustered(n, 300, 1), 0.2)
run("uniform", *synth.s_uniform(n, 3), 0.4)
