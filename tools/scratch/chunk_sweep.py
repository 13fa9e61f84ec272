import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import synth
from yolov5_obb_amd import nms_rotated_ext
dev = torch.device("cuda:0")
def run(name, dets, scores, thr, iters=10):
    d, s = dets.to(dev), scores.to(dev)
    for _ in range(3): k = nms_rotated_ext.nms_rotated(d, s, thr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); k = nms_rotated_ext.nms_rotated(d, s, thr); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"chunk {os.environ.get('OBB_NMS_CHUNK')} max {os.environ.get('OBB_NMS_CHUNK_MAX')} {name:22s} n={len(d):7d} kept={len(k):6d} min {min(ts):8.3f} ms", flush=True)
run("clustered", *synth.s_clustered(100000, 300, 0), 0.4)
run("clustered K=3000", *synth.s_clustered(100000, 3000, 1), 0.4)
run("clustered 30k", *synth.s_clustered(30000, 300, 0), 0.4)
run("uniform 30k", *synth.s_uniform(30000, 0), 0.4)
