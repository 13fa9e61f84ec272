"""GPU parity of the HIP ComputeLoss (yolov5_obb_amd/utils/loss.py -> obb_loss_* in libobb_hip.so) against the CPU
oracle (oracle/pyref.py: build_targets / compute_loss, pinned to the reference's utils/loss.py by tests/golden).

Tolerances (BASELINE.json north_star: "within 1e-5 on IoU/loss scalars"): loss scalars rtol 1e-5; gradients rtol 1e-4
of the tensor's max magnitude (hand-written derivatives vs torch autograd, both fp32); build_targets indices bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import pyref
from tests import synth

pytestmark = pytest.mark.gpu

G = None


def golden():
    global G
    if G is None:
        import os
        G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs.npz"))
    return G


def make(dev, bs=2, nc=16, nt=50, seed=31, sizes=(32, 16, 8), imgsz=256, hyp_over=None, dtype=torch.float32):
    from yolov5_obb_amd.utils.loss import ComputeLoss
    hyp = synth.scaled_hyp(nc, imgsz)
    if hyp_over:
        hyp.update(hyp_over)
    p, t = synth.s_loss(bs, nc, nt, seed, imgsz=imgsz, sizes=list(sizes))
    spec = pyref.LossSpec(hyp, synth.grid_anchors(), torch.tensor(synth.DEFAULT_STRIDES), nc)
    model = synth.FakeModel(nc, hyp, dev)
    cl = ComputeLoss(model)
    return cl, spec, p, t


def run_both(cl, spec, p, t, dev, dtype=torch.float32):
    pc = [x.clone().to(dtype).float().requires_grad_(True) for x in p]       # the oracle sees the dtype-rounded logits
    lo, io = pyref.compute_loss(spec, pc, t.clone())
    lo.backward()
    pg = [x.clone().to(device=dev, dtype=dtype).requires_grad_(True) for x in p]
    lg, ig = cl(pg, t.to(dev))
    lg.backward()
    return (lo, io, pc), (lg, ig, pg)


ACHIEVED = {"scalar_rel": 0.0, "grad_rel": 0.0, "tensors": 0}      # reported by test_zz_report_achieved_errors


def check(o, g, rtol=1e-5, grtol=1e-5):
    lo, io, pc = o
    lg, ig, pg = g
    assert lg.shape == (1,) and ig.shape == (4,)
    assert np.allclose(lg.detach().cpu().numpy(), lo.detach().numpy(), rtol=rtol, atol=1e-6), (lg, lo)
    assert np.allclose(ig.cpu().numpy(), io.numpy(), rtol=rtol, atol=1e-6), (ig, io)
    ref = np.concatenate([lo.detach().numpy().ravel(), io.numpy().ravel()])
    got = np.concatenate([lg.detach().cpu().numpy().ravel(), ig.cpu().numpy().ravel()])
    ok = np.abs(ref) > 1e-3
    if ok.any():
        ACHIEVED["scalar_rel"] = max(ACHIEVED["scalar_rel"], float(np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok]))))
    for a, b in zip(pg, pc):
        ga, gb = a.grad.float().cpu(), b.grad
        scale = gb.abs().max().item() + 1e-12
        err = (ga - gb).abs().max().item()
        assert err <= grtol * scale, (err, scale)
        if b.dtype == torch.float32 and a.dtype == torch.float32:
            ACHIEVED["grad_rel"] = max(ACHIEVED["grad_rel"], err / scale)
            ACHIEVED["tensors"] += 1


@pytest.mark.parametrize("nt", [0, 1, 50, 400])
def test_loss_and_gradients_match_oracle(dev, nt):
    cl, spec, p, t = make(dev, nt=nt, seed=31 + nt)
    check(*run_both(cl, spec, p, t, dev))


GOLDEN_CASES = {'nt50': dict(bs=2, nc=16, nt=50, seed=31, sizes=(32, 16, 8), imgsz=256),
                'nt0': dict(bs=2, nc=16, nt=0, seed=32, sizes=(16, 8, 4), imgsz=128),
                'nt400': dict(bs=4, nc=15, nt=400, seed=33, sizes=(32, 16, 8), imgsz=256),
                'smooth': dict(bs=2, nc=16, nt=80, seed=34, sizes=(32, 16, 8), imgsz=256, label_smoothing=0.1)}


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_golden_reference_case(dev, name):
    """The cases frozen from the reference's own utils/loss.py (tests/golden/gen_golden.py, section F)."""
    from yolov5_obb_amd.utils.loss import ComputeLoss
    g = golden()
    cfg = GOLDEN_CASES[name]
    hyp = synth.scaled_hyp(cfg['nc'], 1024)
    hyp['label_smoothing'] = cfg.get('label_smoothing', 0.0)
    p, t = synth.s_loss(cfg['bs'], cfg['nc'], cfg['nt'], cfg['seed'], imgsz=cfg['imgsz'], sizes=list(cfg['sizes']))
    cl = ComputeLoss(synth.FakeModel(cfg['nc'], hyp, dev))
    pg = [x.clone().to(dev).requires_grad_(True) for x in p]
    loss, items = cl(pg, t.to(dev))
    loss.backward()
    assert np.allclose(loss.detach().cpu().numpy(), g[f"loss_{name}_loss"], rtol=1e-5, atol=1e-6)
    assert np.allclose(items.cpu().numpy(), g[f"loss_{name}_items"], rtol=1e-5, atol=1e-6)
    tcls, tbox, indices, anch, tcsl = cl.build_targets(pg, t.to(dev))
    for i in range(3):
        idx = torch.stack(indices[i], 1).cpu().numpy()
        assert np.array_equal(idx, g[f"loss_{name}_idx{i}"])
        assert np.array_equal(tbox[i].cpu().numpy(), g[f"loss_{name}_tbox{i}"])
        gs = np.array([pg[i].grad.double().sum().item(), pg[i].grad.double().abs().sum().item()])
        assert np.allclose(gs, g[f"loss_{name}_gradsum{i}"], rtol=1e-4, atol=1e-7), (gs, g[f"loss_{name}_gradsum{i}"])


def test_build_targets_rows_bit_exact(dev):
    cl, spec, p, t = make(dev, nt=300, seed=5)
    ref = pyref.build_targets(spec, p, t)
    tcls, tbox, indices, anch, tcsl = cl.build_targets([x.to(dev) for x in p], t.to(dev))
    for i in range(3):
        r = ref[i]
        assert np.array_equal(torch.stack(indices[i], 1).cpu().numpy(), torch.stack((r['b'], r['a'], r['gj'], r['gi']), 1).numpy())
        assert np.array_equal(tbox[i].cpu().numpy(), r['tbox'].numpy())
        assert np.array_equal(anch[i].cpu().numpy(), r['anch'].numpy())
        assert np.array_equal(tcls[i].cpu().numpy(), r['tcls'].numpy())
        assert np.array_equal(tcsl[i].cpu().numpy(), r['csl'].numpy())


def test_colliding_targets_last_writer_and_summed_gradients(dev):
    """Several targets in one cell: tobj takes the LAST row's iou (utils/loss.py:159), gradients of all rows add up."""
    cl, spec, p, t = make(dev, nt=60, seed=9)
    t = t.clone()
    t[:40, 0] = 0
    t[:40, 2:4] = torch.tensor([100.3, 60.7]) + 0.2 * torch.rand(40, 2, generator=torch.Generator().manual_seed(1))
    t[:40, 4] = 40 + torch.arange(40) * 0.5
    t[:40, 5] = 12
    check(*run_both(cl, spec, p, t, dev))


def test_hyper_parameters_label_smoothing_pos_weights(dev):
    cl, spec, p, t = make(dev, nt=80, seed=12, hyp_over=dict(label_smoothing=0.1, cls_pw=1.7, theta_pw=0.6, obj_pw=2.2,
                                                               anchor_t=2.5, box=0.07, theta=0.9))
    check(*run_both(cl, spec, p, t, dev))


def test_sort_obj_iou(dev):
    cl, spec, p, t = make(dev, nt=120, seed=14)
    t = t.clone()
    t[:30, 0] = 1
    t[:30, 2:4] = torch.tensor([40.2, 200.1])
    cl.sort_obj_iou = True
    pg = [x.clone().to(dev).requires_grad_(True) for x in p]
    loss, items = cl(pg, t.to(dev))
    # oracle with the winner = largest iou: emulate by sorting rows inside pyref through its own flag-free path
    pc = [x.clone().requires_grad_(True) for x in p]
    lo, io = pyref.compute_loss(spec, pc, t.clone(), sort_obj_iou=True)
    assert np.allclose(loss.detach().cpu().numpy(), lo.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert np.allclose(items.cpu().numpy(), io.numpy(), rtol=1e-5, atol=1e-6)


def test_fp16_heads(dev):
    """AMP: p is fp16 (train.py:324-326); the kernels compute in fp32 on the fp16 logits and write fp16 gradients."""
    cl, spec, p, t = make(dev, nt=100, seed=21)
    o, g = run_both(cl, spec, p, t, dev, dtype=torch.float16)
    lo, io, pc = o
    lg, ig, pg = g
    assert np.allclose(lg.detach().cpu().numpy(), lo.detach().numpy(), rtol=2e-3)       # tobj is rounded to fp16 (:155)
    assert np.allclose(ig.cpu().numpy(), io.numpy(), rtol=2e-3, atol=1e-5)
    for a, b in zip(pg, pc):
        assert a.grad.dtype == torch.float16
        scale = b.grad.abs().max().item()
        assert (a.grad.float().cpu() - b.grad).abs().max().item() <= 2e-3 * scale + 1e-7


def test_grad_scale_and_full_size(dev):
    """BASELINE configs[2] per-GPU shape (16,3,{128,64,32}^2,201), nt=1500, incoming gradient != 1 (GradScaler)."""
    cl, spec, p, t = make(dev, bs=16, nc=16, nt=1500, seed=3, sizes=(128, 64, 32), imgsz=1024)
    pg = [x.clone().to(dev).requires_grad_(True) for x in p]
    loss, items = cl(pg, t.to(dev))
    (loss * 512.0).backward()
    pc = [x.clone().requires_grad_(True) for x in p]
    lo, io = pyref.compute_loss(spec, pc, t.clone())
    (lo * 512.0).backward()
    assert np.allclose(loss.detach().cpu().numpy(), lo.detach().numpy(), rtol=1e-5)
    assert np.allclose(items.cpu().numpy(), io.numpy(), rtol=1e-5, atol=1e-7)
    for a, b in zip(pg, pc):
        scale = b.grad.abs().max().item()
        assert (a.grad.cpu() - b.grad).abs().max().item() <= 1e-4 * scale
        # every element written exactly once: untouched rows are exactly zero outside channel 4
        z = a.grad[..., 5:].abs().sum(-1) == 0
        assert z.float().mean().item() > 0.9


def test_out_of_range_target_is_loud(dev):
    cl, spec, p, t = make(dev, nt=20, seed=2)
    t = t.clone()
    t[3, 0] = 7          # image index outside the batch of 2: the reference raises IndexError
    pg = [x.clone().to(dev).requires_grad_(True) for x in p]
    loss, _ = cl(pg, t.to(dev))
    assert torch.isnan(loss).all()
    loss.backward()                                 # NaN gradients too: a GradScaler skips the step
    assert all(torch.isnan(x.grad).any() for x in pg)          # (every level: GradScaler.unscale_ finds them)
    with pytest.raises(IndexError):
        cl.build_targets([x.to(dev) for x in p], t.to(dev))


def test_rejects_cpu_and_mixed_dtypes(dev):
    cl, spec, p, t = make(dev, nt=5)
    with pytest.raises(RuntimeError):
        cl(p, t)                                    # CPU tensors
    pg = [x.to(dev) for x in p]
    pg[1] = pg[1].half()
    with pytest.raises(RuntimeError, match="share dtype"):
        cl(pg, t.to(dev))                           # the kernels read every level with p[0]'s element size


@pytest.mark.parametrize("gamma", [1.5, 2.0])
def test_focal_loss(dev, gamma):
    """hyp['fl_gamma'] > 0: FocalLoss(BCE, gamma) around the class, angle and objectness terms (utils/loss.py:35-62,107-110),
    forward and gradients vs the oracle's restatement."""
    cl, spec, p, t = make(dev, nt=120, seed=51, hyp_over=dict(fl_gamma=gamma, cls_pw=1.3, obj_pw=0.8))
    assert cl.fl_gamma == gamma
    check(*run_both(cl, spec, p, t, dev), rtol=2e-5, grtol=2e-5)


def test_anchor_update_after_construction_is_seen(dev):
    """autoanchor writes m.anchors[:] = ... in place after the loss object may exist: the reference reads the tensor on every
    call (utils/loss.py:120), so must this class."""
    cl, spec, p, t = make(dev, nt=80, seed=3)
    l0, _ = cl([x.to(dev) for x in p], t.to(dev))
    with torch.no_grad():
        cl._det.anchors[:] = cl._det.anchors * 1.7
    spec.anchors = spec.anchors * 1.7
    (lo, io, pc), (lg, ig, pg) = run_both(cl, spec, p, t, dev)
    assert not torch.allclose(lg.detach().cpu(), l0.detach().cpu())
    check((lo, io, pc), (lg, ig, pg))


def test_targets_without_csl_columns_are_encoded_on_the_device(dev):
    """(nt,7) targets -- the 180-bin CSL rows are regenerated inside the loss kernels from theta with
    hyp['csl_radius'], exactly as gaussian_label_cpu rolls its window; same loss / gradients as the (nt,187) wire format."""
    cl, spec, p, t = make(dev, nt=200, seed=44)
    # anchored on the ORACLE: its (nt,187) rows carry gaussian_label_cpu's labels (tests/synth.py), the HIP path only gets (nt,7)
    pc = [x.clone().requires_grad_(True) for x in p]
    lo, io = pyref.compute_loss(spec, pc, t.clone())
    lo.backward()
    pg0 = [x.clone().to(dev).requires_grad_(True) for x in p]
    lg0, ig0 = cl(pg0, t[:, :7].contiguous().to(dev))
    lg0.backward()
    check((lo, io, pc), (lg0, ig0, pg0))
    pg1 = [x.clone().to(dev).requires_grad_(True) for x in p]
    l1, i1 = cl(pg1, t.to(dev))
    l1.backward()
    pg2 = [x.clone().to(dev).requires_grad_(True) for x in p]
    l2, i2 = cl(pg2, t[:, :7].contiguous().to(dev))
    l2.backward()
    assert torch.allclose(l1, l2, rtol=1e-6, atol=1e-7) and torch.allclose(i1, i2, rtol=1e-6, atol=1e-7)
    for a, b in zip(pg1, pg2):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-9)
    tc1 = cl.build_targets(pg1, t.to(dev))[4]
    tc2 = cl.build_targets(pg2, t[:, :7].contiguous().to(dev))[4]
    for a, b in zip(tc1, tc2):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-30)


def test_autobalance_three_steps(dev, oracle_lib):
    """ComputeLoss(model, autobalance=True) (utils/loss.py:98,115,180-184): the per-level objectness weights move with every call --
    each level's loss uses the weight it had BEFORE the call, then balance[i] <- 0.9999 balance[i] + 0.0001 / obj_i and everything
    is divided by the stride-16 level's weight.  Three calls on fresh logits against the restated reference: the scalars of every
    call (they depend on the weights the previous calls left), the gradients of the last one, and the weights themselves."""
    from yolov5_obb_amd.utils.loss import ComputeLoss
    nc = 16
    hyp = synth.scaled_hyp(nc, 256)
    spec = pyref.LossSpec(hyp, synth.grid_anchors(), torch.tensor(synth.DEFAULT_STRIDES), nc)
    cl = ComputeLoss(synth.FakeModel(nc, hyp, dev), autobalance=True)
    assert cl.autobalance and cl.ssi == 1 and list(cl.balance) == list(spec.balance) == [4.0, 1.0, 0.4]
    for step in range(3):
        p, t = synth.s_loss(2, nc, 40, 500 + step, imgsz=256, sizes=[32, 16, 8])
        pc = [x.clone().requires_grad_(True) for x in p]
        lo, io = pyref.compute_loss(spec, pc, t.clone(), autobalance=True)
        lo.backward()
        pg = [x.clone().to(dev).requires_grad_(True) for x in p]
        lg, ig = cl(pg, t.to(dev))
        lg.backward()
        check((lo, io, pc), (lg, ig, pg))
        assert np.allclose(np.asarray(cl.balance, dtype=np.float64), np.asarray(spec.balance, dtype=np.float64), rtol=1e-6), (step, cl.balance, spec.balance)
        assert cl.balance[1] == 1.0 and cl.balance[0] != 4.0


def test_zz_report_achieved_errors(dev):
    """Not a check of its own: prints what the fp32 comparisons above achieved (pytest shows it in the warnings summary).  The
    asserted tolerances are 1e-5 relative on the loss scalars (north_star) and 1e-5 of the tensor's largest |gradient| (1e-4 until the achieved figure was printed: 3.3e-7)."""
    import warnings
    warnings.warn(UserWarning(f"ComputeLoss vs oracle (fp32 cases of this run): loss scalars max rel. error {ACHIEVED['scalar_rel']:.2e}; "
                              f"gradients max |err| / max |grad| = {ACHIEVED['grad_rel']:.2e} over {ACHIEVED['tensors']} tensors"))
