"""GPU parity of the post-NMS tail (yolov5_obb_amd/val.py -> obb_val_postprocess_f32 / obb_process_batch_f32) against the
restated reference (oracle/pyref.py: val.py:69-90,226-236; the process_batch restatement keeps the reference's numpy steps)."""
import numpy as np
import pytest
import torch

from oracle import pyref
from tests import synth

pytestmark = pytest.mark.gpu


def test_val_postprocess_matches_reference_chain(dev):
    from yolov5_obb_amd.val import val_postprocess
    d, s = synth.s_uniform(700, 4)
    pred = torch.cat((d, s[:, None], torch.randint(0, 15, (700, 1)).float()), 1)
    gain, pad = 0.7314, (12.0, 3.5)
    got = val_postprocess(pred.to(dev), ratio_pad=((gain, gain), pad))
    ref = pyref.val_postprocess(pred.clone(), gain, pad)
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        assert torch.allclose(g.cpu(), r, rtol=1e-6, atol=2e-4), (g.cpu() - r).abs().max()
    # ratio_pad derived from the shapes (utils/general.py:639-641)
    got2 = val_postprocess(pred.to(dev), img1_shape=(1024, 1024), img0_shape=(1400, 1300))
    g2 = min(1024 / 1400, 1024 / 1300)
    ref2 = pyref.val_postprocess(pred.clone(), g2, ((1024 - 1300 * g2) / 2, (1024 - 1400 * g2) / 2))
    assert torch.allclose(got2[2].cpu(), ref2[2], rtol=1e-6, atol=2e-4)
    assert val_postprocess(torch.zeros(0, 7, device=dev), ratio_pad=((1.0, 1.0), (0.0, 0.0)))[0].shape == (0, 10)


@pytest.mark.parametrize("n,m,seed", [(300, 40, 0), (1500, 200, 1), (50, 0, 2), (7, 90, 3)])
def test_process_batch_matches_reference(dev, n, m, seed):
    from yolov5_obb_amd.val import process_batch
    g = torch.Generator().manual_seed(seed)
    nc = 5
    lab_xy = torch.rand(m, 2, generator=g) * 900
    lab_wh = torch.rand(m, 2, generator=g) * 80 + 10
    labels = torch.cat((torch.randint(0, nc, (m, 1), generator=g).float(), lab_xy, lab_xy + lab_wh), 1)
    # detections: jittered copies of labels (several per label) + noise boxes
    if m:
        src = torch.randint(0, m, (n,), generator=g)
        box = labels[src, 1:] + torch.randn(n, 4, generator=g) * 6
        cls = torch.where(torch.rand(n, generator=g) < 0.85, labels[src, 0], torch.randint(0, nc, (n,), generator=g).float())
    else:
        xy = torch.rand(n, 2, generator=g) * 900
        box = torch.cat((xy, xy + 30), 1)
        cls = torch.randint(0, nc, (n,), generator=g).float()
    det = torch.cat((box, torch.rand(n, 1, generator=g), cls[:, None]), 1)
    iouv = torch.linspace(0.5, 0.95, 10)
    ref = pyref.process_batch(det.clone(), labels.clone(), iouv)
    got = process_batch(det.to(dev), labels.to(dev), iouv.to(dev))
    assert got.dtype == torch.bool and got.shape == ref.shape
    assert torch.equal(got.cpu(), ref)


def test_against_outputs_frozen_from_the_reference_val_py(dev):
    import os
    from tests.golden.gen_golden import VALPOST_CASES, valpost_inputs, valpost_dets
    from yolov5_obb_amd.val import process_batch, val_postprocess
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs.npz"))
    for name, (n, m, seed) in VALPOST_CASES.items():
        det, labels, iouv = valpost_inputs(n, m, seed)
        got = process_batch(det.to(dev), labels.to(dev), iouv.to(dev))
        assert np.array_equal(got.cpu().numpy(), G[f"pb_{name}"])
    d7, gain, pad = valpost_dets(600, 5)
    got = val_postprocess(d7.to(dev), ratio_pad=((gain, gain), pad))
    for g_, key in zip(got, ("vp_poly", "vp_hbb", "vp_polyn", "vp_hbbn")):
        assert np.allclose(g_.cpu().numpy(), G[key], rtol=1e-6, atol=2e-4)


def test_sharded_val_loop_hip_vs_oracle(dev):
    """yolov5_obb_amd.val_sharded.run (the per-batch loop of val.py:180-250, single process here) on the HIP hot path equals
    the same loop with the oracle's restatements as the three callables -- same (correct, conf, pcls, tcls) statistics."""
    from yolov5_obb_amd import val_sharded

    class Set(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(70 + i)
            im = torch.randint(0, 256, (3, 64, 64), dtype=torch.uint8, generator=g)
            nl = 12
            lab = torch.zeros(nl, 7)
            lab[:, 1] = torch.randint(0, 15, (nl,), generator=g).float()
            lab[:, 2:4] = torch.rand(nl, 2, generator=g) * 1000
            lab[:, 4] = torch.rand(nl, generator=g) * 120 + 20
            lab[:, 5] = torch.rand(nl, generator=g) * 30 + 8
            lab[:, 6] = (torch.rand(nl, generator=g) - 0.5) * 3.0
            return im, lab, f"i{i}", ((1300, 1400), ((0.7314, 0.7314), (12.0, 3.5)))

        @staticmethod
        def collate_fn(batch):
            im, lab, path, shapes = zip(*batch)
            for k, l in enumerate(lab):
                l[:, 0] = k
            return torch.stack(im, 0), torch.cat(lab, 0), path, shapes

    preds = {}

    def model_for(device):
        def model(im):
            key = int(im.float().sum().item()) % 100000
            if key not in preds:
                preds[key] = synth.s_pred(im.shape[0], 3000, 15, seed=key % 1000, n_obj=30)
            return (preds[key].to(device),)
        return model
    loader = torch.utils.data.DataLoader(Set(), batch_size=3, collate_fn=Set.collate_fn)
    hip = val_sharded.run(model_for(dev), loader, conf_thres=0.25, iou_thres=0.45, half=False, device=dev)
    ora = val_sharded.run(model_for("cpu"), loader, conf_thres=0.25, iou_thres=0.45, half=False, device="cpu",
                          nms=lambda o, c, i, multi_label=True, agnostic=False: pyref.non_max_suppression_obb(o, c, i, multi_label=multi_label, agnostic=agnostic),
                          postprocess=lambda p, ratio_pad=None: pyref.val_postprocess(p.clone(), ratio_pad[0][0], ratio_pad[1]),
                          match=pyref.process_batch)
    assert hip["seen"] == ora["seen"] == 6 and len(hip["stats"]) == 4 and len(hip["stats"][1]) > 20
    assert np.array_equal(hip["stats"][0], ora["stats"][0])                       # correct (n, 10) bool
    assert np.allclose(hip["stats"][1], ora["stats"][1], rtol=0, atol=0)          # conf: copied through
    assert np.array_equal(hip["stats"][2], ora["stats"][2]) and np.array_equal(hip["stats"][3], ora["stats"][3])


@pytest.mark.parametrize("bs", [1, 5, 70])
def test_batched_tail_equals_the_per_image_tail(dev, bs):
    """val.val_tail_batch (three launches + one copy per batch) against the per-image val_postprocess / process_batch pair it
    replaces in the loop: the same four box arrays bit for bit, the same correct rows, conf, cls -- with images without
    detections, images without labels, and more images than one call takes (64)."""
    from yolov5_obb_amd.val import process_batch, val_postprocess, val_tail_batch
    g = torch.Generator().manual_seed(100 + bs)
    iouv = torch.linspace(0.5, 0.95, 10)
    preds, tgs, shapes = [], [], []
    for b in range(bs):
        n = 0 if b % 7 == 3 else int(torch.randint(1, 60, (1,), generator=g))
        d = torch.zeros(n, 7)
        d[:, :2] = torch.rand(n, 2, generator=g) * 900 + 50
        d[:, 2] = torch.rand(n, generator=g) * 100 + 10
        d[:, 3] = torch.rand(n, generator=g) * 30 + 5
        d[:, 4] = (torch.rand(n, generator=g) - 0.5) * 3.1
        d[:, 5] = torch.sort(torch.rand(n, generator=g), descending=True)[0]
        d[:, 6] = torch.randint(0, 4, (n,), generator=g).float()
        preds.append(d)
        nl = 0 if b % 5 == 1 else int(torch.randint(1, 30, (1,), generator=g))
        t = torch.zeros(nl, 9)                                   # two extra columns like a collate row with more fields
        t[:, 0] = b
        src = torch.randint(0, max(n, 1), (nl,), generator=g)
        if n:
            t[:, 1] = d[src, 6]
            t[:, 2:7] = d[src, :5] + torch.randn(nl, 5, generator=g) * torch.tensor([3., 3., 4., 2., 0.02])
        else:
            t[:, 2:4] = torch.rand(nl, 2, generator=g) * 900
            t[:, 4:6] = 20.0
        tgs.append(t)
        gain = 0.5 + 0.5 * float(torch.rand(1, generator=g))
        shapes.append(((1100 + b, 1300 - b), ((gain, gain), (4.0 + b % 3, 9.5))))
    targets = torch.cat(tgs, 0)
    packed = torch.cat(preds, 0).to(dev)
    views = list(packed.split([p.shape[0] for p in preds]))     # consecutive views of one buffer, as the NMS returns them
    stats, (boxes, offs) = val_tail_batch(views, targets.to(dev), shapes, iouv.to(dev), want_boxes=True)
    loose = val_tail_batch([p.to(dev).clone() for p in preds], targets.to(dev), shapes, iouv.to(dev))     # not consecutive: concatenated
    for b in range(bs):
        p = preds[b].to(dev)
        sl = slice(offs[b], offs[b + 1])
        correct, conf, cls = stats[b]
        assert correct.shape == (p.shape[0], 10) and torch.equal(conf, preds[b][:, 5]) and torch.equal(cls, preds[b][:, 6])
        assert all(torch.equal(x, y) for x, y in zip(stats[b], loose[b]))
        if p.shape[0] == 0:
            continue
        ratio_pad = shapes[b][1]
        ref = val_postprocess(p, ratio_pad=ratio_pad)
        for got, want in zip(boxes, ref):
            assert torch.equal(got[sl], want)
        lab = tgs[b]
        if len(lab):
            lab7 = torch.cat((lab[:, 2:7], torch.zeros_like(lab[:, :1]), lab[:, 1:2]), 1).to(dev)
            tb = val_postprocess(lab7, ratio_pad=ratio_pad)[1][:, :4].clone()
            tb[:, [0, 2]] -= ratio_pad[1][0]; tb[:, [1, 3]] -= ratio_pad[1][1]
            tb[:, :4] /= ratio_pad[0][0]
            tb[:, [0, 2]] = tb[:, [0, 2]].clamp(0, float(shapes[b][0][1])); tb[:, [1, 3]] = tb[:, [1, 3]].clamp(0, float(shapes[b][0][0]))
            want = process_batch(ref[3], torch.cat((lab[:, 1:2].to(dev), tb), 1), iouv.to(dev)).cpu()
        else:
            want = torch.zeros(p.shape[0], 10, dtype=torch.bool)
        assert torch.equal(correct, want), b
    assert sum(int(s[0].any()) for s in stats) >= max(1, bs // 3)
