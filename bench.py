#!/usr/bin/env python3
"""bench.py -- the hot-path benchmark of the MI355X oriented-box path (contract: see the task's bench section).

One "step" = one pass of the hot path over one batch of synthetic input, with the input already resident in HBM:
the shape BASELINE.json's `metric` is quoted on -- DOTAv1.5 (nc = 16, no = 201) 1024^2 tiles, batch 16, val.py --task
speed -- i.e. the `non_max_suppression_obb` call of val.py:206 on a (16, 64512, 201) fp16 Detect output with the
speed-task thresholds (conf 0.25, iou 0.45, multi_label=True, max_det 1500; val.py:378-383,206), including the one
device->host read of the per-image counts the call ends with.  val.py hands the NMS a FRESH tensor every batch
(val.py:197-206), so the timed loop rotates through ROTATE = 4 distinct prediction tensors (1.66 GB working set, far
beyond the 256 MB Infinity Cache): `ms_per_step` is the cold-cache number; `warm` reports the same loop on one tensor.
Data is synthetic (tests/synth.py: S-pred), weights do not exist on this path.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE set, the way the driver starts it) the process IS a
rank; started plainly with --gpus N > 1 it re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1` (the reference's multi-GPU entry is a launcher too: sh/ddp_train.sh:1,
train.py:526) and fails loudly when fewer than N devices are visible.  Images shard across ranks (pure data parallel, no
collective on the data path; RCCL = backend "nccl" is only used for the timing barrier / max-reduce); value = whole-job
images/s; `n_gpus` = dist.get_world_size(), `rccl_ranks` = the ranks that met at the RCCL barrier.
--dry-run: the launcher / rank plumbing alone on the CPU (gloo), no GPU work -- what the CPU test exercises.

The timed K steps run with HIP events around the step's dominant kernel (the persistent NMS kernel; `obb_profile_enable(2)`: two
event records per step on the kernels' stream); `ms_per_step_with_all_stage_events` is the same K steps with events around all
five stages (ten records, ~30 us per step: what `stages_ms` / `kernels` list for the other stages), and
`ms_per_step_without_stage_events` the same K steps without any.

The JSON line also carries
  roofline      BASELINE.json's target figure: rotated NMS at N = 100k candidates, SURVEY.md section 8d's algorithmic bytes
                bytes_nms(N) = 24N + 8N + 8N*ceil(N/64) over the time of the WHOLE call (sort + prep + the persistent
                kernel; HIP events on the stream the kernels run on).  Reported for the WORST of the two regimes that
                resemble detector output (S-clustered K = 300, and the same with 18 class offsets = the natural shape of
                configs[3]); `traffic` = measured HBM bytes per launch from the rocprofv3 --pmc passes (profiles/).  The
                kernel never builds the mask, so this is a time target in bytes' clothing: `pair_tests_per_s` (N(N-1)/2
                pair decisions over the call time) is the honest companion figure.
  nms_100k      all four regimes (clustered K300, +18 classes, K3000, uniform): ms per call, kept, stage times, fraction
  kernels       roofline figures of the other kernels of the step (k_decode: HBM-bound streaming filter; fractions over
                both the algorithmic bytes and the measured traffic)
  hbm_copy      the measured copy ceiling of this device next to the 8 TB/s spec peak
  nmsobb_nc16, nmsobb_tta   the fused driver on the DOTAv1.5 batch (nc = 16, what `metric` names) and on the TTA stress
                tensor (1, 114627, 203) of models/yolo.py:149-161 with conf 0.01 / iou 0.4 (configs[3])
  loss, detect  secondary timings of the other rows of the hot path, not part of `value`
  detect_nms_chain   Detect decode -> non_max_suppression_obb per batch, with and without the objectness column Detect hands to
                the filter; same detections either way
  cpu_baseline  the CPU oracle (port of the reference's CPU path) on the host cores, bounded samples: the NMS bucket of the
                step, the single-thread rotated NMS at N = 1k..30k on both distributions, and the
                `detect.py --device cpu`-equivalent buckets with a random-init yolov5n (8d(ii))
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def bytes_nms(n):
    return 24 * n + 8 * n + 8 * n * ((n + 63) // 64)


def collect_profile(L, nst=8):
    ms = (C.c_double * nst)()
    cnt = (C.c_int64 * nst)()
    rc = L.obb_profile_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), nst)
    assert rc == 0
    return list(ms), list(cnt)


def bench_next_rows(dev, dets_per_image):
    """Timings of the rows SURVEY 8(f) ranks behind the hot path, on synthetic inputs of realistic size; the CPU side is the
    oracle restatement of the reference (numpy + the C port of polyiou.cpp) on a bounded sample."""
    import tempfile
    import numpy as np
    import oracle
    from oracle import pyref
    from tests.golden import gen_golden as gg
    from yolov5_obb_amd import val as V
    from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
    from yolov5_obb_amd.DOTA_devkit import dota_evaluation_task1 as EV
    oracle.build(with_ref=False)
    res = {}

    def wall(fn, reps):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    # ---- val.py:226-250 per image: rbox2poly/poly2hbb/xywh2xyxy/scale_polys + process_batch (the bench step's own detections)
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)
    g = torch.Generator().manual_seed(0)
    labels = []
    for d in dets_per_image:
        k = max(1, d.shape[0] // 2)
        src = d[torch.randperm(d.shape[0], generator=g)[:k].to(dev)]
        hbb = V.val_postprocess(src, ratio_pad=((1.0, 1.0), (0.0, 0.0)))[1]
        labels.append(torch.cat((hbb[:, 5:6], hbb[:, :4] + 2.0), 1).contiguous())

    def tail():
        for d, lb in zip(dets_per_image, labels):
            poly, hbb, polyn, hbbn = V.val_postprocess(d, ratio_pad=((0.7314, 0.7314), (12.0, 3.5)))
            V.process_batch(hbbn, lb, iouv)
    ms = wall(tail, 20)
    # ... and the whole batch in one call (val.val_tail_batch: two launches, rows into polled pinned memory; what val_sharded.run uses)
    tg = torch.cat([torch.cat((torch.full((lb.shape[0], 1), float(i), device=dev), lb[:, :1], torch.zeros((lb.shape[0], 5), device=dev)), 1)
                    for i, lb in enumerate(labels)], 0)
    for i, (d, lb) in enumerate(zip(dets_per_image, labels)):          # label rows [img cls cx cy l s theta]: boxes around the detections' own
        sel = tg[:, 0] == i
        k = int(sel.sum())
        tg[sel, 2:7] = d[:k, :5]
    shapes_b = [((1400, 1400), ((0.7314, 0.7314), (12.0, 3.5)))] * len(dets_per_image)
    ms_batch = wall(lambda: V.val_tail_batch(dets_per_image, tg, shapes_b, iouv), 20)
    # (the labels above are half as many as the detections -- ~115 per image; a DOTA tile holds a few dozen: the same call with 23 per image)
    tg23 = torch.cat([tg[tg[:, 0] == i][:23] for i in range(len(dets_per_image))], 0).contiguous()
    ms_batch23 = wall(lambda: V.val_tail_batch(dets_per_image, tg23, shapes_b, iouv), 20)
    d0, l0 = dets_per_image[0].cpu(), labels[0].cpu()
    t0 = time.perf_counter()
    for _ in range(5):
        pp = pyref.val_postprocess(d0.clone(), 0.7314, (12.0, 3.5))
        pyref.process_batch(pp[3], l0, iouv.cpu())
    cms = (time.perf_counter() - t0) / 5 * 1e3
    res["val_tail"] = {"workload": f"{len(dets_per_image)} images x ~{int(dets_per_image[0].shape[0])} detections: val_postprocess + process_batch",
                       "ms_per_batch": round(ms_batch, 3), "ms_per_image": round(ms_batch / len(dets_per_image), 4),
                       "labels_per_image": int(tg.shape[0]) // len(dets_per_image), "ms_per_batch_23_labels_per_image": round(ms_batch23, 3),
                       "ms_per_batch_per_image_calls": round(ms, 3),
                       "note": "ms_per_batch: val.val_tail_batch through the active binding (two launches since round 5, the statistics written straight into "
                               "polled pinned host memory), with as many labels as half the detections; ms_per_batch_23_labels_per_image: the same with a DOTA "
                               "tile's label count; ms_per_batch_per_image_calls: round 3's loop of val_postprocess + process_batch per image",
                       "cpu_port_ms_per_image": round(cms, 3)}
    # ---- ResultMerge: one class file of 300 source images (tiles 1024/824, two rates)
    lines = gg.merge_input_lines(300, 40, 7, False)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "Task1_plane.txt")
        with open(src, "w") as f:
            f.write("\n".join(lines) + "\n")
        boxes = RM.parse_result_file(src)
        names = list(boxes)
        arrs = [np.asarray(boxes[k], dtype=np.float64) for k in names]
        base = np.cumsum([0] + [len(a) for a in arrs])
        orders = [a[:, 8].argsort()[::-1] + base[i] for i, a in enumerate(arrs)]
        allb = np.concatenate(arrs)
        ms_dev = wall(lambda: RM.merge_nms_segments(allb, orders, 0.2), 10)
        dst = os.path.join(td, "merged")
        os.makedirs(dst)
        ms_file = wall(lambda: RM.mergesingle(dst, RM.py_cpu_nms_poly_fast, src) and None, 3)
    sub = [ln for ln in lines if int(ln[1:5]) < 12]
    t0 = time.perf_counter()
    pyref.merge_result_lines(sub)
    cms = (time.perf_counter() - t0) * 1e3
    res["result_merge"] = {"workload": f"Task1_<class>.txt with {len(lines)} tile detections of 300 source images -> merged file (poly NMS 0.2, double)",
                           "ms_device_call_incl_copies": round(ms_dev, 3), "ms_whole_file_incl_text": round(ms_file, 2),
                           "cpu_port_ms": round(cms * len(lines) / max(1, len(sub)), 1),
                           "cpu_sample": f"{len(sub)} lines (12 images) through oracle.pyref.merge_result_lines, scaled by line count"}
    # ---- Task-1 evaluation: one class, 200 images
    gt, det = gg.eval_inputs(200, 60, 5)
    with tempfile.TemporaryDirectory() as td:
        detpath, annopath, imagesetfile = gg.eval_write(td, gt, det)
        ms_eval = wall(lambda: EV.voc_eval(detpath, annopath, imagesetfile, "plane", 0.5, True) and None, 3)
        parsed = {k: EV.parse_gt(annopath.format(k)) for k in list(gt)[:10]}
    names10 = list(parsed)
    det10 = [ln for ln in det["plane"] if ln.split(" ")[0] in parsed]
    t0 = time.perf_counter()
    pyref.task1_voc_eval(parsed, names10, det10, "plane", 0.5, True)
    cms = (time.perf_counter() - t0) * 1e3
    res["task1_eval"] = {"workload": f"voc_eval of one class: {len(det['plane'])} detections, 200 images x ~60 ground-truth quads",
                         "ms_whole_incl_text": round(ms_eval, 2),
                         "cpu_port_ms": round(cms * len(det["plane"]) / max(1, len(det10)), 1),
                         "cpu_sample": f"{len(det10)} detections (10 images) through oracle.pyref.task1_voc_eval, scaled by detection count"}
    return res


def provenance(_lib, L):
    """Which binaries were timed: library version string, size / mtime / sha256 of the shared objects, the binding in use."""
    import hashlib

    def ident(path):
        try:
            h = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
            st = os.stat(path)
            return {"path": os.path.relpath(path, ROOT), "bytes": st.st_size, "mtime": int(st.st_mtime), "sha256_16": h}
        except OSError as e:
            return {"path": path, "error": str(e)}
    ext = _lib.compiled()
    return {"obb_version": L.obb_version().decode(), "library": ident(_lib.LIB_PATH),
            "binding": "compiled (nms_rotated_ext_c, pybind11)" if ext is not None else "ctypes (fallback)",
            "binding_module": ident(_lib.EXT_PATH) if ext is not None else None,
            "torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}


def host_cpu_budget():
    """CPUs this process may really use: min(visible cores, the cgroup's CPU quota) -- cpu.max (cgroup v2) or cfs_quota_us / cfs_period_us."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` started plainly (no RANK / WORLD_SIZE): become the launcher of N ranks, one per GPU, the way
    the reference's multi-GPU entry does (sh/ddp_train.sh:1: python -m torch.distributed.launch --nproc_per_node N).  Returns
    the exit code of the job."""
    import subprocess
    if not args.dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible on this node -- refusing to run a "
                             f"{args.gpus}-rank job on fewer devices\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cpu_budget() // max(1, args.gpus))))      # (the container's CPU quota, shared by the ranks)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """The rank plumbing of the bench without GPU work (CPU, gloo): rendezvous, barrier + max-over-ranks timing, the line."""
    import torch.distributed as dist
    from yolov5_obb_amd.utils import shard
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    ranks_met = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(ranks_met)
    bs = 16
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))                            # ranks of different speed: the job is as slow as the slowest
    if world > 1:
        dist.barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "img/s", "n_gpus": dist.get_world_size() if world > 1 else 1,
                          "rccl_ranks": int(ranks_met.item()), "backend": "gloo (dry run, no GPU work)", "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dry_run": True,
                          "config": {"workload": "dry run of the launcher / rank plumbing", "global_batch": bs * world,
                                     "parallelism": f"dp{world} (images sharded, no data-path collective)"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


METRIC = "val.py img/s (hot path: non_max_suppression_obb, bs16) + NMS ms/img @100k cand, DOTAv1.5 1024^2, 1/2/4/8 GPU"
WINDOWS = 5         # the timed window of K steps is repeated this many times; the median window is the reported one
ROTATE = 4          # distinct prediction tensors rotated through the timed loop (4 x 415 MB: nothing stays Infinity-Cache-warm)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline step + the 100k NMS regimes")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rank plumbing only, on the CPU with gloo")
    ap.add_argument("--nms-n", type=int, default=100000)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        met = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(met)                                        # every rank is a real device behind RCCL
        rccl_ranks = int(met.item())
        assert rccl_ranks == dist.get_world_size() == args.gpus

    from tests import synth
    from yolov5_obb_amd import _lib, nms_rotated_ext
    from yolov5_obb_amd.utils import shard
    from yolov5_obb_amd.utils.general import non_max_suppression_obb
    L = _lib.lib()

    # ---------------- workload: the shape `metric` names (DOTAv1.5: nc = 16, no = 201), ROTATE fresh tensors
    bs, A, nc = 16, 64512, 16
    no = 5 + nc + 180
    kw = dict(conf_thres=0.25, iou_thres=0.45, multi_label=True, max_det=1500)
    preds = [synth.s_pred(bs, A, nc, seed=1000 + 16 * rank + r, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
             for r in range(ROTATE)]
    pred = preds[0]
    torch.cuda.synchronize()
    # Host threads: this container may use 16 CPUs' worth of time per 100 ms (cgroup cpu.max = 1600000 100000 on the GPU boxes) while
    # torch sizes its intra-op pool from the 256 cores it sees (128 threads).  Every parallel CPU op (a host-side torch.rand of 10^5
    # rows, a clone of a 44k-element tensor) leaves 128 OpenMP workers spinning for a few ms, the cgroup's quota of the period is gone,
    # and the kernel's CFS bandwidth control parks EVERY thread of the process until the period ends: the "~70-88 ms stall of any call"
    # of rounds 2-4 (profiles/r5_host_stall.md: cpu.stat nr_throttled grows with the stalls, OMP_NUM_THREADS=1 removes both).  The
    # bench keeps torch inside the budget it really has.
    torch.set_num_threads(max(1, host_cpu_budget() // max(1, world)))      # (the ranks of one node share the container's quota)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    # device spin-up, outside the W warmup steps: ~0.4 s of the same calls before anything is counted.  (Round 4 saw the FIRST
    # timed loop run 9-30 us per step slower than the later ones and first blamed the clocks; the cause was the library creating
    # its HIP events lazily inside that loop -- obb_profile_enable creates them up front now.  The spin-up stays: it is cheap.)
    t_spin = time.perf_counter()
    i = 0
    while time.perf_counter() - t_spin < 0.4:
        out = non_max_suppression_obb(preds[i % ROTATE], **kw)
        i += 1
    for i in range(max(args.warmup, ROTATE)):
        out = non_max_suppression_obb(preds[i % ROTATE], **kw)
    barrier()
    # THE timed region: K steps with HIP events around the step's dominant kernel only (the persistent NMS kernel: two records per
    # step on the kernels' own stream; recording all five stages costs ten records = ~30 us of a 0.2 ms step)
    # (one window of K x 0.14 ms is a 3 ms measurement -- one scheduler hiccup moves `value` by more than the
    #  box-to-box spread.  The window of EXACTLY K steps, barrier + synchronize on both sides, is repeated WINDOWS times back to back;
    #  `ms_per_step` / `value` are the MEDIAN window's, the others are reported next to it.)
    L.obb_profile_enable(2)
    win_dt = []
    for _w in range(WINDOWS):
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = non_max_suppression_obb(preds[i % ROTATE], **kw)
        barrier()
        win_dt.append(time.perf_counter() - t0)
    ms_sum_t, cnts_t = collect_profile(L)
    L.obb_profile_enable(0)
    # the same K steps with events around every stage: the per-stage figures of `stages_ms` / `kernels` (the NMS kernel's own figure
    # stays the one measured inside the timed region above)
    barrier()
    L.obb_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = non_max_suppression_obb(preds[i % ROTATE], **kw)
    barrier()
    ms_all_events = shard.max_over_ranks(time.perf_counter() - t0, device=dev) / args.steps * 1e3
    ms_sum, cnts = collect_profile(L)
    L.obb_profile_enable(0)
    ms_sum[3], cnts[3] = ms_sum_t[3], cnts_t[3]
    # ... and once more without any event: informational
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = non_max_suppression_obb(preds[i % ROTATE], **kw)
    barrier()
    ms_plain = shard.max_over_ranks(time.perf_counter() - t0, device=dev) / args.steps * 1e3
    # ... and on ONE tensor (round-2's loop): the objectness lines of a 415 MB tensor stay in the Infinity Cache
    for _ in range(3):
        out = non_max_suppression_obb(pred, **kw)
    barrier()
    L.obb_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = non_max_suppression_obb(pred, **kw)
    barrier()
    ms_warm = shard.max_over_ranks(time.perf_counter() - t0, device=dev) / args.steps * 1e3
    ms_sum_w, cnts_w = collect_profile(L)
    L.obb_profile_enable(0)
    win_dt = [shard.max_over_ranks(w, device=dev) for w in win_dt]      # a window is as slow as its slowest rank
    dt = sorted(win_dt)[len(win_dt) // 2]
    ms_per_step = dt / args.steps * 1e3
    windows_obj = {"n": WINDOWS, "steps_each": args.steps, "ms_per_step": [round(w / args.steps * 1e3, 4) for w in win_dt],
                   "median": round(ms_per_step, 4), "min": round(min(win_dt) / args.steps * 1e3, 4), "max": round(max(win_dt) / args.steps * 1e3, 4),
                   "note": "every window: exactly `steps` steps between barrier + synchronize; value / ms_per_step = the median window"}
    value = world * bs * args.steps / dt

    # the step's kernels, measured with HIP events inside the timed region (cold: rotating tensors; warm: one tensor)
    dec_ms = ms_sum[0] / max(1, cnts[0])
    dec_ms_warm = ms_sum_w[0] / max(1, cnts_w[0])
    nms_ms_step = ms_sum[3] / max(1, cnts[3])
    n_det = sum(int(o.shape[0]) for o in out)
    alg_bytes = bs * A * no * 2 + 28 * n_det            # bytes_dec = bs*A*no*sizeof(elem) + 28*n_out
    # candidates per image (what the NMS kernel sees), same arithmetic as the kernel: conf = obj*cls rounded to fp16
    with torch.no_grad():
        objm = pred[..., 4:5] > kw["conf_thres"]
        cand = (((pred[..., 5:5 + nc] * pred[..., 4:5]) > kw["conf_thres"]) & objm).sum((1, 2)).clamp(max=30000).tolist()
        n_pass = int(objm.sum())
    # what k_decode has to move at least: the 128-byte line holding obj of every row + the whole row of every anchor that passes
    line_bytes = bs * A * 128 + n_pass * no * 2 + 28 * n_det
    nms_alg = int(sum(bytes_nms(int(c)) for c in cand))
    nms_ach = nms_alg / (nms_ms_step * 1e-3) / 1e9
    pmc = {}
    for name in ("r6_pmc.json", "r5_pmc.json", "r4_pmc.json", "r3_pmc.json", "r2_pmc.json", "r1_pmc.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            pmc["_file"] = "profiles/" + name
            break
        except Exception:
            continue
    sq = {}
    try:
        for name in ("r6_sq.json", "r5_sq.json"):
            if os.path.exists(os.path.join(ROOT, "profiles", name)):
                sq = json.load(open(os.path.join(ROOT, "profiles", name)))
                sq["_file"] = "profiles/" + name
                break
    except Exception:
        pass
    # (through the compiled binding the step is three launches: "gather" -- k_gather_out -- reads 0, the output rows are written by the
    #  NMS kernel and counted in "nms_steps"; there is no reset launch either: obb_non_max_suppression_obb_st)
    stage_names = ["decode", "segsort", "prep", "nms_steps", "gather"]
    stages = {stage_names[i]: round(ms_sum[i] / max(1, cnts[i]), 4) for i in range(5)}
    stages_warm = {stage_names[i]: round(ms_sum_w[i] / max(1, cnts_w[i]), 4) for i in range(5)}

    # ---------------- NMS @ 100k candidates (configs[3] stress): the four regimes of SURVEY 8d / VERDICT r1
    n100 = args.nms_n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    regimes = {}
    reps = 20
    for rname, label in (("clustered_k300_raw", "S-clustered(K=300)"), ("clustered_k300_18cls", "S-clustered(K=300) + 18 class offsets"),
                         ("clustered_k3000", "S-clustered(K=3000)"), ("clustered_k3000_18cls", "S-clustered(K=3000) + 18 class offsets"),
                         ("uniform", "S-uniform")):
        d100, s100 = synth.regime_100k(rname, n100)
        d100, s100 = d100.to(dev), s100.to(dev)
        for _ in range(3):
            k100 = nms_rotated_ext.nms_rotated(d100, s100, 0.4)
        torch.cuda.synchronize()
        time.sleep(0.05)                         # (rounds 2-4 waited 0.3 s here for "a ~86 ms stall that follows host-side data preparation": CFS throttling of the container after 128 OpenMP workers had spun, see torch.set_num_threads above)
        k100 = nms_rotated_ext.nms_rotated(d100, s100, 0.4)
        torch.cuda.synchronize()
        per_call = []                            # the call as a user sees it (no stage events inside the timed region)
        for _ in range(reps):
            e0.record()
            k100 = nms_rotated_ext.nms_rotated(d100, s100, 0.4)
            e1.record()
            torch.cuda.synchronize()
            per_call.append(e0.elapsed_time(e1))
        per_call.sort()
        nms_ms = per_call[len(per_call) // 2]     # median; mean and max are reported next to it
        L.obb_profile_enable(1)                  # second pass: per-stage HIP events recorded by the library on the same stream
        for _ in range(reps):
            k100 = nms_rotated_ext.nms_rotated(d100, s100, 0.4)
        torch.cuda.synchronize()
        pms, pc = collect_profile(L)
        L.obb_profile_enable(0)
        ach = bytes_nms(n100) / (nms_ms * 1e-3) / 1e9
        regimes[rname] = {
            "n": n100, "distribution": label, "iou_thres": 0.4, "kept": int(k100.numel()), "ms_per_call": round(nms_ms, 4),
            "ms_mean": round(sum(per_call) / len(per_call), 4), "ms_min": round(per_call[0], 4), "ms_max": round(per_call[-1], 4),
            "stages_ms": {"sort": round(pms[5] / max(1, pc[5]), 4), "prep": round(pms[6] / max(1, pc[6]), 4),
                          "steps": round(pms[7] / max(1, pc[7]), 4)},
            "achieved_GBs": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 4),
            "pair_tests_per_s": round(n100 * (n100 - 1) / 2 / (nms_ms * 1e-3), 1)}
        del d100, s100
    # (the roofline object is the WORST of the regimes, not the friendliest; the real HBM rate sits next to the
    #  notional one: real_hbm_frac = measured HBM bytes of the whole call / avg_call_ms / peak)
    worst = min(regimes, key=lambda r: regimes[r]["frac"])
    wr = regimes[worst]
    traffic = pmc.get("nms_100k_call_" + worst.replace("_raw", ""))
    for r in regimes:
        t_r = pmc.get("nms_100k_call_" + r.replace("_raw", ""))
        regimes[r]["traffic"] = t_r
        regimes[r]["real_hbm_frac"] = None if not t_r else round(t_r / (regimes[r]["ms_per_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    nms_obj = {"regimes": regimes, "roofline_regime": worst,
               "note": "fraction = SURVEY 8d bytes_nms(N) over the whole call; the roofline object reports the WORST of the five regimes"}
    roofline = {"bound": "hbm", "achieved": wr["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": wr["frac"],
                "traffic": traffic, "real_hbm_frac": wr["real_hbm_frac"], "algorithmic_bytes": bytes_nms(n100),
                "kernel": "single-list rotated NMS @ N = 100k, the whole call (k_ps_* sort, k_prep_rot, then the phase kernels k_mk_* of csrc/nms_mk.h or "
                          "k_slab_split + k_nms_persist<obb::RotGeom, true>, by the library's own choice per regime)", "regime": wr["distribution"],
                "avg_call_ms": wr["ms_per_call"], "avg_kernel_ms": wr["stages_ms"]["steps"], "pair_tests_per_s": wr["pair_tests_per_s"],
                "frac_by_regime": {r: regimes[r]["frac"] for r in regimes},
                "real_hbm_frac_by_regime": {r: regimes[r]["real_hbm_frac"] for r in regimes},
                # the roofline that binds this kernel is not HBM (traffic << algorithmic bytes): SQ counters of k_nms_persist per regime --
                # valu_frac = issued VALU cycles / (256 CUs x 4 SIMDs x kernel cycles), wait_frac = share of the waves' resident time spent
                # parked (s_waitcnt, barrier spins), from profiles/r5_sq.md (rocprofv3 --pmc, separate passes; tools/rocpd_sq.py)
                "sq": {r: sq.get("nms_100k_" + r.replace("_raw", ""), sq.get("k_nms_persist_100k_" + r.replace("_raw", ""))) for r in regimes}, "sq_source": sq.get("_file"),
                "valu_frac": (sq.get("nms_100k_" + worst.replace("_raw", ""), sq.get("k_nms_persist_100k_" + worst.replace("_raw", ""))) or {}).get("valu_frac"),
                "note": "frac = SURVEY 8d bytes over avg_call_ms, the whole NMS call, for the WORST regime: three sort launches (k_ps_*), the "
                        "record kernel, and either the phase kernels of csrc/nms_mk.h (select / probe / decide+resolve / cross / decide+select per "
                        "step) or k_slab_split + k_nms_persist<RotGeom, true>; avg_kernel_ms = the HIP-event time of everything behind the "
                        "records (the stage 'steps'); compare profiles/r6_nms100k_kernel_stats.md.  No path builds the mask: traffic << "
                        "algorithmic bytes, `frac` is a time target in bytes' clothing -- real_hbm_frac is the HBM rate the call really "
                        "reaches; what binds it is latency (profiles/r6_sq.md)"}

    # ---------------- measured copy ceiling next to the spec peak (256 MiB device-to-device, read + write)
    cbuf = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    cdst = torch.empty_like(cbuf)
    for _ in range(3):
        cdst.copy_(cbuf)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        cdst.copy_(cbuf)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 2 * cbuf.numel() * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    hbm_copy = {"measured_copy_GBs": round(copy_gbs, 1), "spec_peak_GBs": HBM_PEAK_GBS, "bytes": "256 MiB read + 256 MiB written per copy"}
    del cbuf, cdst

    # ---------------- the fused driver on the other shapes BASELINE names (rank 0 reports; not part of `value`)
    def time_nmsobb(pr, kwargs, reps_=10):
        for _ in range(3):
            o = non_max_suppression_obb(pr, **kwargs)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps_):
            o = non_max_suppression_obb(pr, **kwargs)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps_, sum(int(x.shape[0]) for x in o)
    del preds[1:]
    torch.cuda.empty_cache()
    extras = rank == 0 and not args.no_extras
    nc15_obj = nc2_obj = tta_obj = dense_obj = None
    if extras:
        try:
            p15 = synth.s_pred(bs, A, 15, seed=2000, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
            ms15, nd15 = time_nmsobb(p15, kw)
            nc15_obj = {"workload": "BASELINE configs[1]: DOTAv1.0 batch (16, 64512, 200) fp16, speed-task thresholds (one tensor, warm)",
                        "ms_per_batch": round(ms15, 4), "img_per_s": round(bs / (ms15 * 1e-3), 1), "detections": nd15}
            del p15
            p2 = synth.s_pred(bs, A, 2, seed=2002, n_obj=120, fg_frac=0.03, device=dev, dtype=torch.float16)
            ms2, nd2 = time_nmsobb(p2, kw)
            nc2_obj = {"workload": "BASELINE configs[4] shape: DroneVehicle batch (16, 64512, 187) fp16, nc = 2, speed-task thresholds (one tensor, warm)",
                       "ms_per_batch": round(ms2, 4), "img_per_s": round(bs / (ms2 * 1e-3), 1), "detections": nd2}
            del p2
            # the regime of the conv stand-in's val loop: thousands of candidates per image of which most are
            # KEPT, every image at max_det -- 4000 planted objects per image, ~1 passing anchor each (tests/test_nmsobb_gpu.py::
            # test_dense_mostly_kept_regime_matches_the_oracle is the parity test of this generator)
            pdk = synth.s_pred(bs, A, nc, seed=2003, n_obj=4000, fg_frac=0.08, device=dev, dtype=torch.float16)
            for _ in range(3):
                non_max_suppression_obb(pdk, **kw)
            L.obb_profile_enable(1)
            msdk, nddk = time_nmsobb(pdk, kw)
            pmd, pcd = collect_profile(L)
            L.obb_profile_enable(0)
            with torch.no_grad():
                cdk = int((((pdk[..., 5:5 + nc] * pdk[..., 4:5]) > kw["conf_thres"]) & (pdk[..., 4:5] > kw["conf_thres"])).sum()) // bs
            dense_obj = {"workload": "(16, 64512, 201) fp16 with 4000 planted objects per image: thousands of candidates per image, most kept, every image at max_det "
                                     "(the regime of val_buckets' random-init heads), speed-task thresholds (one tensor, warm)",
                         "ms_per_batch": round(msdk, 4), "img_per_s": round(bs / (msdk * 1e-3), 1), "candidates_per_image": cdk, "detections": nddk,
                         "stages_ms": {n_: round(pmd[i_] / max(1, pcd[i_]), 4) for i_, n_ in enumerate(("decode", "sort", "prep", "nms_kernel", "gather"))}}
            del pdk
            ptta = synth.s_pred(1, 114627, 18, seed=2001, n_obj=300, fg_frac=0.05, device=dev, dtype=torch.float16)
            kw_tta = dict(conf_thres=0.01, iou_thres=0.4, multi_label=True, max_det=1500)
            mstta, ndtta = time_nmsobb(ptta, kw_tta)
            with torch.no_grad():
                ctta = int((((ptta[..., 5:23] * ptta[..., 4:5]) > 0.01) & (ptta[..., 4:5] > 0.01)).sum())
            tta_obj = {"workload": "TTA stress tensor (1, 114627, 203) fp16 (models/yolo.py:149-161), conf 0.01, iou 0.4, multi_label (configs[3])",
                       "ms_per_image": round(mstta, 4), "candidates": ctta, "detections": ndtta}
            del ptta
        except Exception as e:
            nc15_obj = nc15_obj or {"error": str(e)}

    # ---------------- the polygon paths (rank 0; not part of `value`): nms_poly (utils/nms_rotated/src/poly_nms_cuda.cu:197-261),
    # the devkit's host-pointer _poly_nms / _overlaps incl. their copies (poly_nms_kernel.cu:277-329, poly_overlaps_kernel.cu:368-427)
    poly_obj = None
    if extras:
        try:
            import numpy as np
            from yolov5_obb_amd import ops
            from yolov5_obb_amd.DOTA_devkit.poly_nms_gpu.poly_nms import poly_gpu_nms
            from yolov5_obb_amd.DOTA_devkit.poly_nms_gpu.poly_overlaps import poly_overlaps
            poly_obj = {}

            def ev_time(fn, reps_):
                # median of per-call event times: a call right after the devkit's host-pointer path (hipMalloc / hipFree inside,
                # poly_nms_kernel.cu:277-329 convention) or after a large torch allocation was seen to take 50 ms once in ~10
                # (host / driver side: the kernels of that call are as fast as ever); the mean of five would report that
                fn(); torch.cuda.synchronize()
                ts, r = [], None
                for _ in range(reps_):
                    e0.record()
                    r = fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ts.sort()
                return ts[len(ts) // 2], r
            for npoly in (30000, 100000):
                dq, sq = synth.s_clustered(npoly, 300, seed=0)
                q9 = torch.cat((synth.rbox_to_quad(dq), sq[:, None]), 1).contiguous().to(dev)
                msq, kq = ev_time(lambda: nms_rotated_ext.nms_poly(q9, 0.4), 9)
                bq = 40 * npoly + 8 * npoly + 8 * npoly * ((npoly + 63) // 64)          # poly_nms = NMS with 40 B rows
                poly_obj[f"nms_poly_{npoly}"] = {"distribution": "S-clustered(K=300) quads", "iou_thres": 0.4, "kept": int(kq.numel()),
                                                 "ms_per_call": round(msq, 4), "algorithmic_bytes": bq,
                                                 "frac": round(bq / (msq * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                if npoly == 30000:
                    h9 = q9.cpu().numpy()
                    t0p = time.perf_counter()
                    for _ in range(3):
                        kh = poly_gpu_nms(h9, 0.4)
                    poly_obj["devkit_poly_gpu_nms_30000"] = {"ms_per_call_incl_host_sort_malloc_copies": round((time.perf_counter() - t0p) / 3 * 1e3, 3),
                                                              "kept": len(kh)}
                del q9
            # what rule B (the searched bounding-box skip of csrc/piou_device.h) buys: the same two calls with OBB_NMS_POLY_STRICT=1 (the
            # proved cone rule only).  The library reads the switch once per process: a child process times them.
            try:
                import subprocess
                code = ("import sys, json, torch; sys.path.insert(0, %r); from tests import synth; from yolov5_obb_amd import nms_rotated_ext\n"
                        "dev = torch.device('cuda:%d'); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); out = {}\n"
                        "for n in (30000, 100000):\n"
                        "    d, s = synth.s_clustered(n, 300, seed=0)\n"
                        "    q9 = torch.cat((synth.rbox_to_quad(d), s[:, None]), 1).contiguous().to(dev)\n"
                        "    k = nms_rotated_ext.nms_poly(q9, 0.4); torch.cuda.synchronize(); ts = []\n"
                        "    for _ in range(9):\n"
                        "        e0.record(); k = nms_rotated_ext.nms_poly(q9, 0.4); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))\n"
                        "    ts.sort(); out['nms_poly_%%d' %% n] = {'ms_per_call': round(ts[4], 4), 'kept': int(k.numel())}\n"
                        "print('STRICT ' + json.dumps(out))\n") % (ROOT, local_rank)
                r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OBB_NMS_POLY_STRICT="1"), capture_output=True, text=True, timeout=300)
                got = [ln for ln in r.stdout.splitlines() if ln.startswith("STRICT ")]
                strict = json.loads(got[-1][7:]) if got else {"error": (r.stderr or r.stdout)[-300:]}
                for k_, v_ in strict.items():
                    if k_ in poly_obj and isinstance(v_, dict):
                        poly_obj[k_]["strict_ms_per_call"] = v_["ms_per_call"]
                        poly_obj[k_]["strict_kept"] = v_["kept"]
                        poly_obj[k_]["strict_over_default"] = round(v_["ms_per_call"] / max(poly_obj[k_]["ms_per_call"], 1e-9), 3)
                    elif k_ == "error":
                        poly_obj["strict_error"] = v_
            except Exception as e:
                poly_obj["strict_error"] = str(e)
            bo, _ = synth.s_uniform(10000, 3)
            qo, _ = synth.s_uniform(1000, 4)
            bod, qod = bo.to(dev), qo.to(dev)
            mso, _ = ev_time(lambda: ops.rbox_overlaps(bod, qod), 10)
            t0p = time.perf_counter()
            for _ in range(3):
                poly_overlaps(bo.numpy(), qo.numpy())
            poly_obj["poly_overlaps_10000x1000"] = {"ms_device_call": round(mso, 4), "algorithmic_bytes": 20 * 11000 + 4 * 10000 * 1000,
                                                    "ms_devkit_call_incl_copies": round((time.perf_counter() - t0p) / 3 * 1e3, 3),
                                                    "pairs_per_s": round(1e7 / (mso * 1e-3), 1)}
        except Exception as e:
            poly_obj = poly_obj or {}
            poly_obj["error"] = str(e)

    # ---------------- M1 with the reference's three buckets (val.py:183-207,286-291): pre-process, inference, NMS per image and
    # seen / sum(dt), on a yolov5s-shaped conv stand-in (tools/conv_standin.py, PyTorch-ROCm convolutions like the reference's
    # backbone) + the product's Detect + val_sharded.run.  EVERY rank runs its shard (the final gather is a collective).
    val_obj = None
    if not args.no_extras:
        local = None
        try:
            from tools import conv_standin
            local = conv_standin.val_buckets(dev, n_images=160, batch=16, nc=nc, conf_thres=0.25, iou_thres=0.45, half=True, seed=rank)
        except Exception as e:
            val_obj = {"error": f"rank {rank}: {e}"}
        # the reduction is outside the try block: every rank takes part, whatever happened to its own run
        ldt = local["dt_seconds"] if local else [0.0, 0.0, 0.0]
        slow = [shard.max_over_ranks(x, device=dev) for x in ldt]                 # the job is as slow as its slowest rank
        ok_all = shard.max_over_ranks(0.0 if local else 1.0, device=dev) == 0.0
        if local and ok_all:
            per_rank = local["images_per_rank"]
            val_obj = {"images": per_rank * world, "images_per_rank": per_rank, "batch": local["batch"], "n_gpus": world,
                       "anchors_passing_obj_per_image": local["anchors_passing_obj_per_image"], "model": local["model"],
                       "ms_per_img": {"pre": round(slow[0] / per_rank * 1e3, 4), "inference": round(slow[1] / per_rank * 1e3, 4),
                                      "nms": round(slow[2] / per_rank * 1e3, 4)},
                       "img_per_s_seen_over_sum_dt": round(per_rank * world / max(sum(slow), 1e-12), 1),
                       "nms_share_of_step": round(slow[2] / max(sum(slow), 1e-12), 4),
                       "nms_stages_ms_per_batch": local.get("nms_stages_ms_per_batch"),
                       "note": "dt buckets are the slowest rank's, img/s = images of ALL ranks / sum(dt) (val.py:286-291); secondary to `value`: "
                               "the convolutions are PyTorch-ROCm's (MIOpen), not this repository's; random-init heads place their candidates "
                               "at random, so few suppress each other -- the NMS bucket's worst case"}
        elif val_obj is None:
            val_obj = {"error": "another rank failed"}
        torch.cuda.empty_cache()

    # ---------------- secondary rows of the hot path (rank 0 reports; not part of `value`)
    loss_obj = detect_obj = coupled_obj = None
    if extras:
        try:
            import ctypes as C2
            from yolov5_obb_amd.utils.loss import ComputeLoss
            lnc = 16
            p_l, t_l = synth.s_loss(16, lnc, 1500, 3, imgsz=1024, sizes=[128, 64, 32])
            cl = ComputeLoss(synth.FakeModel(lnc, synth.scaled_hyp(lnc, 1024), dev))
            pg = [x.to(dev).requires_grad_(True) for x in p_l]
            tg = t_l.to(dev)

            def loss_step():
                for x in pg:
                    x.grad = None
                ls, _ = cl(pg, tg)
                ls.backward()
            for _ in range(3):
                loss_step()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                loss_step()
            e1.record()
            torch.cuda.synchronize()
            lms = e0.elapsed_time(e1) / 10
            gbytes = sum(x.numel() * 4 for x in pg)
            loss_obj = {"workload": "ComputeLoss fwd+bwd, p = (16,3,{128,64,32}^2,201) fp32, nt = 1500 (configs[2] per GPU)",
                        "ms_fwd_bwd": round(lms, 4), "grad_bytes": gbytes,
                        "note": "gradient tensors written exactly once (k_loss_bwd_dense, HBM-write bound)"}
            del pg
            # the dtype train.py runs it in: fp16 head outputs (amp.autocast, train.py:324-326) and a GradScaler's scaled incoming
            # gradient (:332) -- same shapes
            ph = [x.to(dev).half().requires_grad_(True) for x in p_l]

            def loss_step_h():
                for x in ph:
                    x.grad = None
                ls, _ = cl(ph, tg)
                (ls * 1024.0).backward()
            for _ in range(3):
                loss_step_h()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                loss_step_h()
            e1.record()
            torch.cuda.synchronize()
            loss_obj["loss_fp16"] = {"workload": "the same shapes with fp16 head outputs and a x1024 incoming gradient (amp.autocast + GradScaler, train.py:324-332)",
                                     "ms_fwd_bwd": round(e0.elapsed_time(e1) / 10, 4), "grad_bytes": sum(x.numel() * 2 for x in ph)}
            del ph
            # one training step of a small model around the HIP loss the way train.py drives it (:245,320-345): DDP on RCCL when the job
            # is a process group, autocast, GradScaler, SGD -- a plumbing figure (three conv stems + this package's Detect at 1024^2,
            # bs 16: the convolutions are PyTorch-ROCm's), next to the loss's own share of it
            try:
                from tests.test_train_leg_gpu import TinyObb
                hyp_t = synth.scaled_hyp(lnc, 1024)
                torch.manual_seed(0)
                net = TinyObb(lnc, hyp_t).to(dev).train()
                step_model = net                                  # (rank 0 alone runs the extras: no DDP wrapper here -- its hooks and buckets with
                cl_t = ComputeLoss(step_model)                    #  this loss are exercised by tests/test_train_leg_gpu.py on an RCCL group)
                opt = torch.optim.SGD(step_model.parameters(), lr=0.01, momentum=0.9)
                scaler = torch.amp.GradScaler("cuda")
                im_t = torch.rand(16, 3, 1024, 1024, device=dev)

                def train_step():
                    opt.zero_grad(set_to_none=True)
                    with torch.autocast("cuda", dtype=torch.float16):
                        ls, _ = cl_t(step_model(im_t), tg)
                    scaler.scale(ls).backward()
                    scaler.step(opt)
                    scaler.update()
                for _ in range(3):
                    train_step()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    train_step()
                e1.record()
                torch.cuda.synchronize()
                loss_obj["train_step"] = {"workload": "TinyObb (3 conv stems + Detect, tests/test_train_leg_gpu.py) at (16,3,1024,1024): forward under autocast(fp16), "
                                                      "HIP ComputeLoss (nt = 1500), GradScaler backward, SGD step; one process, no DDP wrapper (that leg: tests/test_train_leg_gpu.py)",
                                          "ms_per_step": round(e0.elapsed_time(e1) / 10, 4), "scale": float(scaler.get_scale())}
                del net, step_model, im_t
            except Exception as e:
                loss_obj["train_step"] = {"error": str(e)}
            # Detect decode of the batch: 3 conv outputs (16, 3*no, n, n) fp16 -> z (16,64512,no) + permuted heads
            na_d, sizes_d = 3, (128, 64, 32)
            convs = [torch.randn(bs, na_d * no, n, n, device=dev, dtype=torch.float16) for n in sizes_d]
            z = torch.empty(bs, A, no, device=dev, dtype=torch.float16)
            xs = [torch.empty(bs, na_d, n, n, no, device=dev, dtype=torch.float16) for n in sizes_d]
            arrs = [(C2.c_float * 6)(*(synth.grid_anchors()[i] * synth.DEFAULT_STRIDES[i]).reshape(-1).tolist()) for i in range(3)]

            nl_d = len(sizes_d)
            conv_arr = (C2.c_void_p * nl_d)(*[c.data_ptr() for c in convs])
            xs_arr = (C2.c_void_p * nl_d)(*[t.data_ptr() for t in xs])
            ny_arr = (C2.c_int64 * nl_d)(*sizes_d)
            px_arr = (C2.c_float * (nl_d * 6))(*[v for i in range(nl_d) for v in (synth.grid_anchors()[i] * synth.DEFAULT_STRIDES[i]).reshape(-1).tolist()])
            st_arr = (C2.c_float * nl_d)(*synth.DEFAULT_STRIDES[:nl_d])

            def det_step():        # what Detect.forward issues: all levels in one launch (obb_detect_decode_levels)
                rc = L.obb_detect_decode_levels(nl_d, conv_arr, 1, bs, na_d, no, ny_arr, ny_arr, px_arr, st_arr, xs_arr, _lib.ptr(z), A, None,
                                                _lib.stream_ptr(dev))
                assert rc == 0
            for _ in range(3):
                det_step()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                det_step()
            e1.record()
            torch.cuda.synchronize()
            dms = e0.elapsed_time(e1) / 20
            dbytes = 3 * z.numel() * 2
            detect_obj = {"workload": f"Detect inference decode, 3 levels in one launch, (16,64512,{no}) fp16", "ms": round(dms, 4),
                          "algorithmic_bytes": dbytes, "achieved_GBs": round(dbytes / (dms * 1e-3) / 1e9, 1),
                          "frac_of_peak": round(dbytes / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del convs, z, xs
        except Exception as e:                                  # secondary figures never fail the bench
            loss_obj = loss_obj or {"error": str(e)}
        # Detect -> NMS as val.py chains them (val.py:197-206): Detect's decode pass also stores z[..., 4] densely and the
        # filter of non_max_suppression_obb reads that column instead of one line of every row (models/yolo.py mirror)
        try:
            from yolov5_obb_amd.models.yolo import Detect
            det = Detect(nc=nc, anchors=synth.DEFAULT_ANCHORS, ch=(8, 8, 8))
            det.stride = torch.tensor(synth.DEFAULT_STRIDES)
            det.anchors /= det.stride.view(-1, 1, 1)
            det = det.to(dev).half().eval()
            det.m = torch.nn.ModuleList([torch.nn.Identity() for _ in range(3)])      # the conv outputs are the input here
            heads = [h.to(dev) for h in synth.s_head(bs, nc, (128, 64, 32), seed=2000 + rank, n_obj=120, dtype=torch.float16)]
            chain = {}
            for mode in ("plain", "coupled"):
                det.couple_nms = mode == "coupled"

                def chain_step():
                    with torch.no_grad():
                        zc, _ = det(list(heads))
                    return non_max_suppression_obb(zc, **kw)
                for _ in range(10):
                    oc = chain_step()
                torch.cuda.synchronize()
                L.obb_profile_enable(1)
                t0c = time.perf_counter()
                for _ in range(50):
                    oc = chain_step()
                torch.cuda.synchronize()
                chain[mode] = (time.perf_counter() - t0c) / 50 * 1e3
                pm, pcn = collect_profile(L)
                L.obb_profile_enable(0)
                chain[mode + "_decode_ms"] = pm[0] / max(1, pcn[0])
                chain[mode + "_det"] = sum(int(o.shape[0]) for o in oc)
            coupled_obj = {"workload": "Detect decode of synthetic conv outputs (16, 3*200, {128,64,32}^2) fp16 with 120 planted objects per "
                                       "image (tests/synth.py s_head) -> non_max_suppression_obb, per batch of 16 (val.py:197-206)",
                           "ms_plain": round(chain["plain"], 4), "ms_coupled": round(chain["coupled"], 4),
                           "k_decode_ms_plain": round(chain["plain_decode_ms"], 4), "k_decode_ms_coupled": round(chain["coupled_decode_ms"], 4),
                           "detections": chain["coupled_det"], "same_detections": chain["coupled_det"] == chain["plain_det"],
                           "note": "plain = the NMS scans z[..., 4] (one 128-byte line per 400-byte row, freshly written by Detect); "
                                   "coupled = it reads the dense (16, 64512) column Detect stored in the same pass"}
            del heads
        except Exception as e:
            coupled_obj = {"error": str(e)}

    # ---------------- SURVEY 8(f) rows behind the NMS (rank 0, N=1 only; not part of `value`): val.py tail, tile->image merge,
    # Task-1 evaluation -- GPU time of the mirrored call, the oracle port of the reference on a bounded sample beside it
    next_rows = None
    if extras and world == 1 and not args.no_cpu_baseline:
        try:
            next_rows = bench_next_rows(dev, out)
        except Exception as e:
            next_rows = {"error": str(e)}

    # ---------------- CPU baseline (rank 0, N=1 only): the oracle port of the reference CPU path, bounded samples
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            import oracle
            from oracle import pyref, pyref_model
            oracle.build(with_ref=False)
            ncores = host_cpu_budget()                         # (cores the cgroup lets this process use: 16 of the 256 it sees on the GPU boxes)
            sample = pred[:1].float().cpu()                     # 1 image of the same batch, fp32 like --device cpu
            torch.set_num_threads(ncores)
            t0 = time.perf_counter()
            nimg = 0
            while True:
                pyref.non_max_suppression_obb(sample.clone(), **kw)
                nimg += 1
                if time.perf_counter() - t0 > 6.0 or nimg >= 64:
                    break
            cdt = time.perf_counter() - t0
            cpu = {"value": round(nimg / cdt, 3), "unit": "img/s", "cores": int(torch.get_num_threads()), "kind": "port",
                   "sample": f"{nimg} x 1 image (64512 anchors, fp32) of the same synthetic batch through oracle.pyref."
                             f"non_max_suppression_obb (torch CPU filter/decode + single-thread C greedy rotated NMS)"}
            # (i) the reference's rotated NMS is single-threaded (nms_rotated_cpu.cpp): 1 thread, both distributions, time-boxed
            sweep, budget = {}, 35.0
            t_sw = time.perf_counter()
            for nn_ in (1000, 4000, 10000, 30000, 100000):
                for dname, gen in (("clustered", lambda m: synth.s_clustered(m, 300, seed=0)), ("uniform", lambda m: synth.s_uniform(m, 0))):
                    if nn_ == 100000 and dname == "uniform":
                        continue                                              # ~80 min extrapolated: the 100k uniform run is a parity test, not a bench leg
                    key_ = f"{dname}_{nn_}"
                    est = {"clustered": 2.5e-5, "uniform": 2.2e-5 * nn_ / 1000}[dname] * nn_      # seconds, from the survey's probes
                    if time.perf_counter() - t_sw + est > budget:
                        sweep[key_] = None                                    # would not fit the bench's time box
                        continue
                    dsm, ssm = gen(nn_)
                    t1 = time.perf_counter()
                    kk = oracle.nms_rotated(dsm.numpy(), ssm.numpy(), 0.4, ge=True, threads=1)
                    sweep[key_] = {"ms": round((time.perf_counter() - t1) * 1e3, 2), "kept": int(len(kk))}
            cpu["nms_1thread_ms"] = sweep
            cpu["nms_1thread_note"] = "oracle port of nms_rotated_cpu (>=), iou 0.4, one thread; null = skipped by the time box"
            # (ii) detect.py --device cpu equivalent: model forward + NMS + rbox2poly + scale_polys per image, all host cores
            net = pyref_model.YoloV5nObb(16).eval()
            torch.set_num_threads(ncores)                        # (= the cgroup's CPU budget, see host_cpu_budget)
            img = torch.rand(1, 3, 1024, 1024)
            buckets = [0.0, 0.0, 0.0]
            nimg2 = 0
            t_all = time.perf_counter()
            with torch.no_grad():
                while True:
                    t1 = time.perf_counter()
                    zz = net(img)
                    t2 = time.perf_counter()
                    dd = pyref.non_max_suppression_obb(zz, conf_thres=0.25, iou_thres=0.2, multi_label=True, max_det=1000)   # detect.py:215-218 defaults
                    t3 = time.perf_counter()
                    for det in dd:
                        if len(det):
                            pyref.val_postprocess(det, 0.75, (0.0, 0.0))
                    t4 = time.perf_counter()
                    buckets[0] += t2 - t1; buckets[1] += t3 - t2; buckets[2] += t4 - t3
                    nimg2 += 1
                    if time.perf_counter() - t_all > 6.0 or nimg2 >= 32:
                        break
            cpu["detect_cpu_equiv"] = {"model": "yolov5n OBB (nc 16), random init, oracle restatement of models/yolov5n.yaml",
                                       "images": nimg2, "input": "1x3x1024x1024 synthetic",
                                       "ms_per_image": {"inference": round(buckets[0] / nimg2 * 1e3, 2), "nms": round(buckets[1] / nimg2 * 1e3, 3),
                                                        "rbox2poly_scale": round(buckets[2] / nimg2 * 1e3, 3)},
                                       "img_per_s": round(nimg2 / sum(buckets), 3), "threads": int(torch.get_num_threads()), "nproc": int(os.cpu_count() or 1), "cpu_budget": int(ncores),
                                       "threads_note": "SURVEY 8d(ii) asks for all host cores: the leg runs as many threads as the container's CPU quota allows "
                                                       "(cpu_budget; more threads than that only trigger CFS throttling -- what rounds 2-4 saw as 'convolutions get slower "
                                                       "beyond a few dozen threads')",
                                       "note": "random-init logits: the objectness prior passes few anchors, so the NMS bucket is near "
                                               "its floor; the NMS-heavy case is the `value` / `sample` pair above"}
        except Exception as e:                                  # the baseline is informative; never fail the bench on it
            cpu = cpu or {"value": None, "unit": "img/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
            cpu["error"] = str(e)

    if rank == 0:
        line = {
            "metric": METRIC,
            "value": round(value, 2), "unit": "img/s", "n_gpus": dist.get_world_size() if dist is not None else 1, "rccl_ranks": rccl_ranks,
            "steps": args.steps, "warmup": args.warmup, "build": provenance(_lib, L),
            "ms_per_step": round(ms_per_step, 4), "timed_windows": windows_obj, "ms_per_step_with_all_stage_events": round(ms_all_events, 4), "ms_per_step_without_stage_events": round(ms_plain, 4),
            "ms_per_step_one_tensor_warm": round(ms_warm, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"DOTAv1.5 1024^2 bs16 (BASELINE metric; configs[1] thresholds): yolov5 OBB head output (16,64512,201) fp16, "
                                   f"{ROTATE} distinct tensors rotated (cold: {ROTATE * bs * A * no * 2 / 1e9:.2f} GB working set) -> non_max_suppression_obb "
                                   "(conf .25, iou .45, multi_label, max_det 1500), val.py --task speed hot path = dt[2] of val.py",
                       "global_batch": bs * world, "anchors_per_image": A, "nc": nc, "detections_per_batch": n_det,
                       "parallelism": f"dp{world} (images sharded, no data-path collective)"},
            "stages_ms": stages, "stages_ms_one_tensor_warm": stages_warm,
            # the roofline target of BASELINE.json's north_star: rotated NMS at 100k candidates (configs[3] stress),
            # SURVEY 8d formula over the whole call, the WORSE of clustered-K300 and clustered-K300 + 18 class offsets
            "roofline": roofline,
            "hbm_copy": hbm_copy,
            "val_buckets": val_obj,
            "kernels": {
                "obb::k_nms_small<obb::RotGeom, obb::SmallGather, obb::SmallSelfSort> (bs16 step)": {
                    "bound": "hbm", "achieved": round(nms_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(nms_ach / HBM_PEAK_GBS, 5), "traffic": pmc.get("k_nms_persist_bs16"), "algorithmic_bytes": nms_alg,
                    "avg_kernel_ms": round(nms_ms_step, 5), "candidates_per_image": [int(c) for c in cand],
                    "note": "largest share of the step; one workgroup per (image, class) segment of ~100 boxes held in LDS (csrc/nms_small.h; round 3: the persistent kernel, 0.073 ms): VALU bound by the decision stages, not HBM bound.  Since round 5 the time includes the output rows: the workgroup that finishes an image merges its kept lists and writes them (csrc/nmsobb_impl.h SmallGather; the stage 'gather' is 0).  Since round 6 it also includes the per-class ordering: every workgroup picks its class out of the image's candidate keys and ranks it in LDS (SmallSelfSort; the stage 'segsort' is 0) -- the call is two launches (filter, this kernel)"},
                "obb::k_decode<__half>": {
                    "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "avg_kernel_ms": round(dec_ms, 5), "avg_kernel_ms_one_tensor_warm": round(dec_ms_warm, 5),
                    "line_granular_bytes": line_bytes, "traffic": pmc.get("k_decode"),
                    "achieved": round(line_bytes / (dec_ms * 1e-3) / 1e9, 2), "frac": round(line_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_warm": round(line_bytes / (dec_ms_warm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "survey_8d_bytes_not_moved": alg_bytes,
                    "note": "achieved / frac are over the bytes the filter must move (one 128-byte line per row for obj + the rows that pass + "
                            "output), cold; SURVEY 8d's bs*A*no*2 figure is listed but not used: the kernel never reads most of the tensor"},
                "obb::k_detect_decode_levels<__half> (3 levels, one launch)": None if not detect_obj or "ms" not in detect_obj else {
                    "bound": "hbm", "achieved": detect_obj["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": detect_obj["frac_of_peak"], "frac_of_measured_copy": round(detect_obj["achieved_GBs"] / max(copy_gbs, 1e-9), 4),
                    "traffic": pmc.get("k_detect_decode"),
                    "algorithmic_bytes": detect_obj["algorithmic_bytes"], "avg_kernel_ms": detect_obj["ms"]},
                "obb::k_loss_bwd_dense<float>": {
                    "bound": "hbm", "traffic": pmc.get("k_loss_bwd_dense"), "algorithmic_bytes": 829882368,
                    "avg_kernel_ms": pmc.get("k_loss_bwd_dense_ms"),
                    "achieved": None if not pmc.get("k_loss_bwd_dense_ms") else round(829882368 / (pmc["k_loss_bwd_dense_ms"] * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if not pmc.get("k_loss_bwd_dense_ms") else round(829882368 / (pmc["k_loss_bwd_dense_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "kernel time from the rocprofv3 kernel trace in profiles/ (the bench times ComputeLoss fwd+bwd as a whole)"},
            },
            "pmc_source": pmc.get("_file"),
            "nms_100k": nms_obj, "nmsobb_nc15": nc15_obj, "nmsobb_nc2": nc2_obj, "nmsobb_dense_kept": dense_obj, "nmsobb_tta": tta_obj, "polygon_paths": poly_obj,
            "loss": loss_obj, "detect": detect_obj, "detect_nms_chain": coupled_obj, "next_rows": next_rows,
            "cpu_baseline": cpu,
            "parity_unpinned": ["poly2rbox against cv2.minAreaRect (utils/rboxs_utils.py:39-81: OpenCV is not in this image; without it the minimum-area rectangle is computed natively and the function is property-tested, tests/test_poly2rbox_props.py)",
                                "OBB mAP@0.5 within 0.1 of the reference (DOTA_devkit/dota_evaluation_task1.py:320: no weights / dataset offline)"],
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
