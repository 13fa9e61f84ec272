"""Whole-file times of the two text-heavy "next" rows on the GPU box (same workloads as bench.py's next_rows)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden import gen_golden as gg
from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV


def wall(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


lines = gg.merge_input_lines(300, 40, 7, False)
with tempfile.TemporaryDirectory() as td:
    src = os.path.join(td, "Task1_plane.txt")
    open(src, "w").write("\n".join(lines) + "\n")
    dst = os.path.join(td, "merged"); os.makedirs(dst)
    print(f"merge of {len(lines)} tile detections, whole file incl. text: {wall(lambda: RM.mergesingle(dst, RM.py_cpu_nms_poly_fast, src) and None, 5):.2f} ms", flush=True)
gt, det = gg.eval_inputs(200, 60, 5)
with tempfile.TemporaryDirectory() as td:
    detpath, annopath, imagesetfile = gg.eval_write(td, gt, det)
    print(f"voc_eval of {len(det['plane'])} detections, 200 images, whole incl. text: {wall(lambda: EV.voc_eval(detpath, annopath, imagesetfile, 'plane', 0.5, True) and None, 5):.2f} ms", flush=True)
