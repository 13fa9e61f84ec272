"""CPU, world_size 2, gloo: the N>1 path of the inference sharding (yolov5_obb_amd/utils/shard.py) -- every image is
processed by exactly one rank, results come back on rank 0 in the original order, timing is the MAX over ranks.
(The data path itself has no collective; the HIP kernels are covered by the -m gpu tests.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_detections(i):
    """Deterministic stand-in for one image's (n,7) NMS output."""
    g = np.random.default_rng(1000 + i)
    return g.random((i % 5, 7)).astype(np.float32)


def _worker(rank, world_size, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from yolov5_obb_amd.utils import shard
        assert shard.world() == (rank, world_size)
        idx = shard.shard_indices(n_items)
        res = [torch.from_numpy(_fake_detections(i)) for i in idx]
        full = shard.gather_results(idx, res, n_items, dst=0)
        t = shard.max_over_ranks(1.0 + rank)
        if rank == 0:
            ok = all(np.array_equal(full[i].numpy(), _fake_detections(i)) for i in range(n_items))
            q.put(("ok" if ok and len(full) == n_items and t == float(world_size) else "bad", len(idx)))
        else:
            assert full is None and t == float(world_size)
            q.put(("peer", len(idx)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_items = 37
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert got[0][0] == "ok" and got[1][0] == "peer"
    assert got[0][1] + got[1][1] == n_items and abs(got[0][1] - got[1][1]) <= 1


def test_single_process_paths():
    from yolov5_obb_amd.utils import shard
    assert shard.world() == (0, 1)
    assert shard.shard_indices(5) == [0, 1, 2, 3, 4]
    assert shard.shard_indices(7, 1, 3) == [1, 4]
    assert shard.gather_results([0, 1], ["a", "b"], 2) == ["a", "b"]
    assert shard.max_over_ranks(0.25) == 0.25


def _merge_worker(rank, world_size, port, src, dst, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from oracle import pyref
        from yolov5_obb_amd.DOTA_devkit import ResultMerge_multi_process as RM
        # the class files are split over the ranks; the NMS is passed in like in the reference (here: the CPU oracle, so that
        # the sharding of the merge runs without a GPU -- the device NMS itself is covered by tests/test_merge_gpu.py)
        RM.mergebase_parallel(src, dst, pyref.merge_nms_poly_fast)
        dist.barrier()
        q.put((rank, sorted(os.listdir(dst))))
    finally:
        dist.destroy_process_group()


def test_two_rank_merge_splits_the_class_files(tmp_path):
    """mergebase_parallel under a world of 2: every Task1_<class>.txt is merged by exactly one rank, and the result equals
    what the oracle restatement of the reference's mergesingle writes."""
    from oracle import pyref
    from tests.golden.gen_golden import merge_input_lines
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    classes = ["plane", "ship", "harbor", "bridge", "helicopter"]
    want = {}
    for k, c in enumerate(classes):
        lines = merge_input_lines(3, 12, 40 + k, False)
        (src / f"Task1_{c}.txt").write_text("\n".join(lines) + "\n")
        want[f"Task1_{c}.txt"] = "\n".join(pyref.merge_result_lines(lines)) + "\n"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, str(src), str(dst), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    seen = [q.get(timeout=10) for _ in range(2)]
    assert all(files == sorted(want) for _, files in seen)
    for name, text in want.items():
        assert (dst / name).read_text() == text


# ---------------------------------------------------------------------------- the sharded validation runner
class _FakeSet(torch.utils.data.Dataset):
    """23 'images': content and labels are functions of the index (stand-in for LoadImagesAndLabels)."""

    def __len__(self):
        return 23

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(500 + i)
        im = torch.randint(0, 256, (3, 32, 32), dtype=torch.uint8, generator=g)
        nl = i % 4
        lab = torch.zeros(nl, 7)
        if nl:
            lab[:, 1] = torch.randint(0, 3, (nl,), generator=g).float()
            lab[:, 2:4] = torch.rand(nl, 2, generator=g) * 32
            lab[:, 4:6] = torch.rand(nl, 2, generator=g) * 8 + 2
        return im, lab, f"img{i}.png", ((40, 40), ((0.8, 0.8), (0.0, 0.0)))

    @staticmethod
    def collate_fn(batch):
        im, lab, path, shapes = zip(*batch)
        for k, l in enumerate(lab):
            l[:, 0] = k
        return torch.stack(im, 0), torch.cat(lab, 0), path, shapes


def _fake_model(im):
    s = im.float().mean((1, 2, 3))                                   # one number per image
    return (s[:, None, None].expand(-1, 5, 7) * torch.arange(1, 36).view(1, 5, 7).float(),)


def _fake_nms(out, conf, iou, multi_label=True, agnostic=False):
    res = []
    for o in out:
        k = int(o[0, 0] * 1000) % 4                                  # 0..3 "detections"
        d = o[:k].clone()
        d[:, 6] = (d[:, 6] * 10).floor() % 3
        res.append(d)
    return res


def _fake_post(pred, ratio_pad=None):
    n = pred.shape[0]
    poly = torch.cat((pred[:, :4].repeat(1, 2), pred[:, 5:7]), 1)
    hbb = torch.cat((pred[:, :2], pred[:, :2] + pred[:, 2:4].abs(), pred[:, 5:7]), 1)
    return poly, hbb, poly / ratio_pad[0][0], hbb / ratio_pad[0][0]


def _fake_match(det, lab, iouv):
    return (det[:, 5:6] == lab[:, 0].view(1, -1)).any(1, keepdim=True) & (iouv.view(1, -1) < 0.8)


def _run_fake(loader, n_total):
    from yolov5_obb_amd import val_sharded
    seen_ap = {}

    def fake_ap(tp, conf, pcls, tcls, plot=False, save_dir=".", names=None):
        seen_ap["n"] = (len(tp), len(tcls))
        return float(tp.sum())
    r = val_sharded.run(_fake_model, loader, n_total=n_total, device="cpu", half=False, ap_per_class=fake_ap,
                        nms=_fake_nms, postprocess=_fake_post, match=_fake_match)
    return r, seen_ap


def _val_worker(rank, world_size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from yolov5_obb_amd import val_sharded
        ds = _FakeSet()
        full = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=_FakeSet.collate_fn)
        r, ap = _run_fake(val_sharded.shard_loader(full), len(ds))
        if rank == 0:
            q.put(("ok", [c.tolist() for c in r["stats"]], r["seen"], r["metrics"]))
        else:
            assert r["stats"] is None and r["metrics"] is None and r["seen"] == len(ds)
            q.put(("peer",))
    finally:
        dist.destroy_process_group()


def test_sharded_validation_equals_single_process():
    """val_sharded.run over two ranks (gloo) returns, on rank 0, exactly the statistics of a single-process run over the whole
    list, in the original image order, and the metric callable sees the same arrays (the reference's ap_per_class in a real
    run).  The hot-path callables are stand-ins here; the HIP ones are covered by the -m gpu tests."""
    ds = _FakeSet()
    full = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=_FakeSet.collate_fn)
    single, ap1 = _run_fake(full, len(ds))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_val_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted((q.get(timeout=10) for _ in range(2)), key=lambda t: t[0])
    assert got[0][0] == "ok" and got[1][0] == "peer"
    _, stats2, seen2, metric2 = got[0]
    assert seen2 == len(ds) == single["seen"]
    assert [c.tolist() for c in single["stats"]] == stats2 and len(stats2) == 4 and len(stats2[0]) > 5
    assert metric2 == single["metrics"]
