"""Sharded validation: one process per GPU, the image list split by rank, NO collective on the data
path; the per-image statistics meet once, after the loop, where the reference computes its metrics (val.py:269-274).

The reference's ``val.run`` is single-process (``select_device`` returns cuda:0, utils/torch_utils.py:83) and computes the
metrics inside the same function as the loop, so N copies of it cannot simply be pointed at N shards: a rank whose shard
has no true positive skips ``ap_per_class`` (val.py:270-271).  ``run`` below is the per-batch loop of val.py:180-250 on this
package's hot path -- ``non_max_suppression_obb`` (one fused call per batch), ``val_postprocess`` and ``process_batch``
(two launches per image) -- over the rank's shard, followed by ONE ``gather_object`` of the (correct, conf, pcls, tcls)
tuples in the original image order.  The metric itself stays the reference's: pass its ``utils.metrics.ap_per_class``.

    torchrun --nnodes 1 --nproc-per-node 8 --master-addr 127.0.0.1 my_val.py
        dist.init_process_group("nccl")                      # RCCL over xGMI; only the final gather uses it
        loader = val_sharded.shard_loader(full_loader)       # rank r keeps images r, r + world, ...
        res = val_sharded.run(model, loader, n_total=len(full_loader.dataset), ap_per_class=ap_per_class, names=names)
        if res["rank"] == 0: print(res["metrics"], res["img_per_s"])
"""
import time

import numpy as np
import torch

from .utils import shard


def shard_loader(loader, rank=None, world_size=None):
    """A DataLoader over this rank's share of ``loader.dataset`` (strided split, shard.shard_indices) with the same batch
    size, collate function, workers and pinning; ``.global_indices`` holds the dataset indices in iteration order."""
    idx = shard.shard_indices(len(loader.dataset), rank, world_size)
    sub = torch.utils.data.Subset(loader.dataset, idx)
    out = torch.utils.data.DataLoader(sub, batch_size=loader.batch_size, shuffle=False, num_workers=loader.num_workers,
                                      collate_fn=loader.collate_fn, pin_memory=loader.pin_memory, drop_last=False)
    out.global_indices = idx
    return out


def _sync(device):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


def run(model, loader, n_total=None, conf_thres=0.001, iou_thres=0.4, half=True, single_cls=False, augment=False, device=None,
        ap_per_class=None, names=None, nms=None, postprocess=None, match=None, niou=10, collect=True):
    """The loop of val.py:180-250 over ``loader`` (this rank's shard, see shard_loader), then the gather.

    model(im) -> (out (b, A, no), train_out), like the reference's Model in eval mode.  ``loader`` yields
    (im uint8 (b,3,h,w), targets (n, >= 7) [img_in_batch cls cx cy l s theta ...] in pixels, paths, shapes) like
    LoadImagesAndLabels.collate_fn.  nms / postprocess / match default to this package's HIP path
    (utils.general.non_max_suppression_obb, val.val_postprocess, val.process_batch); the CPU tests inject stand-ins.
    Returns a dict: rank, world, seen (all ranks), dt (pre-process, inference, NMS seconds of the slowest rank), img_per_s
    (whole job), stats (rank 0: the four concatenated arrays in original image order), metrics (rank 0: what
    ap_per_class returned, or None).  collect=False: no exchange at all -- the rank's own seen / dt (a caller that only wants
    the time buckets and does its own reduction, bench.py)."""
    # the HIP path's tail runs once per BATCH (val.val_tail_batch: three launches, one copy); injected stand-ins keep the
    # reference's per-image loop
    batch_tail = None
    if postprocess is None and match is None:
        from . import val as V
        batch_tail = V.val_tail_batch
    if nms is None or postprocess is None or match is None:
        from . import val as V
        from .utils.general import non_max_suppression_obb
        nms = nms or non_max_suppression_obb
        postprocess = postprocess or V.val_postprocess
        match = match or V.process_batch
    rank, world = shard.world()
    if device is None:
        device = next(model.parameters()).device if hasattr(model, "parameters") else torch.device("cpu")
    device = torch.device(device)
    half = bool(half) and device.type != "cpu"                      # val.py:128
    iouv = torch.linspace(0.5, 0.95, niou, device=device)           # val.py:172
    gidx = getattr(loader, "global_indices", None)
    per_image, dt, seen = [], [0.0, 0.0, 0.0], 0
    dt_batches = []                                                  # (pre-process, inference, NMS) seconds of every batch
    with torch.no_grad():
        for im, targets, paths, shapes in loader:
            t1 = _sync(device)
            im = im.to(device, non_blocking=True)
            targets = targets.to(device)
            im = (im.half() if half else im.float()) / 255           # val.py:187-188
            nb, _, height, width = im.shape
            t2 = _sync(device)
            dt[0] += t2 - t1
            res = model(im, augment=augment) if augment else model(im)
            out = res[0] if isinstance(res, (tuple, list)) else res
            t3 = _sync(device)
            dt[1] += t3 - t2
            out = nms(out, conf_thres, iou_thres, multi_label=True, agnostic=single_cls)      # val.py:206
            t4 = _sync(device)
            dt[2] += t4 - t3
            dt_batches.append((t2 - t1, t3 - t2, t4 - t3))
            if batch_tail is not None and device.type == "cuda":
                if single_cls:
                    for pred in out:
                        pred[:, 6] = 0
                tail = batch_tail(out, targets, shapes, iouv)                    # val.py:209-250 for the whole batch
                tc = targets[:, :2].cpu() if len(targets) else torch.zeros((0, 2))     # (image, class) of every label: one copy
                for si, pred in enumerate(out):
                    tcls = tc[tc[:, 0] == si, 1].tolist()
                    seen += 1
                    if len(pred) == 0:
                        per_image.append((torch.zeros(0, niou, dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls) if len(tcls) else None)
                        continue
                    correct, conf, pcls = tail[si]
                    per_image.append((correct, conf, pcls, tcls))                # val.py:250
                continue
            for si, pred in enumerate(out):
                labels = targets[targets[:, 0] == si, 1:7]           # (n_gt, [cls cx cy l s theta])
                nl = len(labels)
                tcls = labels[:, 0].tolist() if nl else []
                shape, ratio_pad = shapes[si][0], shapes[si][1]
                seen += 1
                if len(pred) == 0:
                    per_image.append((torch.zeros(0, niou, dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls) if nl else None)
                    continue
                if single_cls:
                    pred[:, 6] = 0
                poly, hbb, polyn, hbbn = postprocess(pred, ratio_pad=ratio_pad)               # val.py:226-236
                if nl:
                    # val.py:238-241 in the reference's operation order: rbox2poly -> poly2hbb -> xywh2xyxy on the UNSCALED
                    # labels (output [1] of the tail kernel: boxes in the letterboxed frame), THEN scale_coords (pad, gain,
                    # clip; utils/general.py:621-633) -- scaling the polygon first rounds differently in fp32 and could flip
                    # a borderline IoU match
                    lab7 = torch.cat((labels[:, 1:6], torch.zeros_like(labels[:, :1]), labels[:, :1]), 1)   # [x y l s theta 0 cls]
                    tb = postprocess(lab7.float(), ratio_pad=ratio_pad)[1][:, :4].clone()
                    gain, pad = ratio_pad[0][0], ratio_pad[1]
                    tb[:, [0, 2]] -= pad[0]
                    tb[:, [1, 3]] -= pad[1]
                    tb[:, :4] /= gain
                    tb[:, [0, 2]] = tb[:, [0, 2]].clamp(0, float(shape[1]))
                    tb[:, [1, 3]] = tb[:, [1, 3]].clamp(0, float(shape[0]))
                    correct = match(hbbn, torch.cat((labels[:, 0:1].float(), tb), 1), iouv)   # val.py:244
                else:
                    correct = torch.zeros(pred.shape[0], niou, dtype=torch.bool)
                per_image.append((correct.cpu(), poly[:, 8].cpu(), poly[:, 9].cpu(), tcls))     # val.py:250
    if not collect:
        return {"rank": rank, "world": world, "seen": seen, "dt": list(dt), "img_per_s": seen / max(sum(dt), 1e-12), "stats": None,
                "metrics": None, "dt_batches": dt_batches}
    # ---- the one exchange: per-image tuples to rank 0, in the original order of the image list
    if gidx is None:
        gidx = list(range(rank, rank + world * len(per_image), world)) if world > 1 else list(range(len(per_image)))
    n_total = int(n_total) if n_total is not None else (len(per_image) if world == 1 else None)
    if n_total is None:
        t = torch.tensor([len(per_image)], dtype=torch.int64)
        if world > 1:
            import torch.distributed as dist
            t = t.to(device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t)
        n_total = int(t.item())
    full = shard.gather_results(gidx[:len(per_image)], per_image, n_total, dst=0)
    seen_all = n_total
    slow = [shard.max_over_ranks(x, device=device if device.type == "cuda" else None) for x in dt]
    res = {"rank": rank, "world": world, "seen": seen_all, "dt": slow, "img_per_s": seen_all / max(sum(slow), 1e-12),
           "stats": None, "metrics": None, "dt_batches": dt_batches}
    if rank == 0:
        st = [s for s in full if s is not None]
        if st:
            cols = [np.concatenate([np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x) for x in col], 0) for col in zip(*st)]
        else:
            cols = []
        res["stats"] = cols
        if ap_per_class is not None and len(cols) and cols[0].any():                          # val.py:269-271
            res["metrics"] = ap_per_class(*cols, plot=False, save_dir=".", names=names or {})
    return res
