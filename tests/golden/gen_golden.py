"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

    YOLOV5_CONFIG_DIR=/tmp/refcfg python tests/golden/gen_golden.py

What is frozen (the reference has no tests / golden vectors of its own):
  * rotated IoU of seeded pairs from the reference header compiled in place (oracle/_ref, device + host branch),
    plus the survey's known answers;
  * polygon IoU (double) from the reference's DOTA_devkit/polyiou.cpp;
  * kept indices of the reference's CPU torch extension (nms_rotated_cpu.cpp) on seeded box sets;
  * outputs of the reference's Python hot path imported from /root/reference with stub modules for the
    uninstalled cv2 / torchvision / seaborn: non_max_suppression_obb, gaussian_label_cpu, rbox2poly, poly2hbb,
    regular_theta, Detect (inference decode), ComputeLoss (+ build_targets).
Inputs are regenerated from seeds by tests/synth.py, so only outputs (and small inputs) are stored.
While generating, the oracle restatement (oracle/) is checked against every reference output.
"""
import ctypes as C
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import oracle                      # noqa: E402
from oracle import pyref           # noqa: E402
from tests import synth            # noqa: E402


VALPOST_CASES = {'a': (300, 40, 0), 'b': (1500, 200, 1), 'c': (7, 90, 3)}


def valpost_inputs(n, m, seed):
    """Detections (n,6) [xyxy conf cls] around labels (m,5) [cls xyxy] -- shared by the generator and the tests."""
    g = torch.Generator().manual_seed(seed)
    nc = 5
    lab_xy = torch.rand(m, 2, generator=g) * 900
    lab_wh = torch.rand(m, 2, generator=g) * 80 + 10
    labels = torch.cat((torch.randint(0, nc, (m, 1), generator=g).float(), lab_xy, lab_xy + lab_wh), 1)
    src = torch.randint(0, m, (n,), generator=g)
    box = labels[src, 1:] + torch.randn(n, 4, generator=g) * 6
    cls = torch.where(torch.rand(n, generator=g) < 0.85, labels[src, 0], torch.randint(0, nc, (n,), generator=g).float())
    det = torch.cat((box, torch.rand(n, 1, generator=g), cls[:, None]), 1)
    return det, labels, torch.linspace(0.5, 0.95, 10)


def valpost_dets(n, seed):
    d, s = synth.s_uniform(n, seed)
    g = torch.Generator().manual_seed(seed + 1)
    return torch.cat((d, s[:, None], torch.randint(0, 15, (n, 1), generator=g).float()), 1), 0.7314, (12.0, 3.5)


MERGE_CASES = {'a': (6, 40, 0, False), 'b': (3, 400, 1, True), 'c': (40, 12, 2, False)}


def merge_input_lines(n_img, n_obj, seed, dense):
    """Lines of a synthetic Task1_<class>.txt before the merge: `<orig>__<rate>__<x>___<y> score x1 y1 .. x4 y4`
    (tools/TestJson2VocClassTxt.py:39-47).  Objects of n_img source images are seen from every 1024-px tile (stride 824,
    rates 1 and 0.5) that contains their centre, each sighting with its own jitter; scores carry 5 decimals
    (val.py:61-66) and are sometimes duplicated (ties), some boxes are degenerate."""
    rng = np.random.RandomState(seed)
    lines = []
    for im in range(n_img):
        name = f"P{im:04d}"
        size = 4000 if not dense else 1500
        for _ in range(n_obj):
            cx, cy = rng.rand(2) * size
            w, h = rng.rand(2) * (120 if not dense else 300) + 6
            th = (rng.rand() - 0.5) * np.pi
            for rate in ("1", "0.5"):
                r = float(rate)
                for tx in range(0, int(size * r), 824):
                    for ty in range(0, int(size * r), 824):
                        px, py = cx * r - tx, cy * r - ty
                        if not (0 <= px < 1024 and 0 <= py < 1024) or rng.rand() < 0.25:
                            continue
                        jx, jy = rng.randn(2) * 2.0
                        ww, hh = w * r * (1 + rng.randn() * 0.05), h * r * (1 + rng.randn() * 0.05)
                        t = th + rng.randn() * 0.03
                        c, s_ = np.cos(t), np.sin(t)
                        pts = []
                        for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
                            pts += [px + jx + sx * ww / 2 * c - sy * hh / 2 * s_, py + jy + sx * ww / 2 * s_ + sy * hh / 2 * c]
                        kind = rng.rand()
                        if kind < 0.02:
                            pts = [pts[0], pts[1]] * 4                     # a point
                        elif kind < 0.04:
                            pts = pts[:4] + pts[:4]                        # a segment walked twice
                        score = round(float(rng.rand()), 5 if rng.rand() < 0.8 else 1)
                        lines.append(f"{name}__{rate}__{tx}___{ty} {score} " + ' '.join(f"{v:.2f}" for v in pts))
    return lines


EVAL_CASES = {'a': (5, 30, 0), 'b': (2, 700, 1), 'c': (25, 6, 2)}
EVAL_CLASSES = ('plane', 'ship')


def eval_inputs(n_img, n_gt, seed):
    """A synthetic Task-1 evaluation set: {image name -> labelTxt lines}, {class -> Task1_<class>.txt lines}.
    Ground truth `x1 y1 .. x4 y4 name difficult` (dota_evaluation_task1.py:21-53), detections
    `image score x1 y1 .. x4 y4`: jittered copies of the ground truth (several per object, some of the wrong class),
    clutter, degenerate quads, repeated scores."""
    rng = np.random.RandomState(seed)

    def quad(cx, cy, w, h, t):
        c, s_ = np.cos(t), np.sin(t)
        pts = []
        for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
            pts += [cx + sx * w / 2 * c - sy * h / 2 * s_, cy + sx * w / 2 * s_ + sy * h / 2 * c]
        return pts
    gt, det = {}, {c: [] for c in EVAL_CLASSES}
    extent = 60.0 * np.sqrt(n_gt) + 100
    for im in range(n_img):
        name = f"P{im:04d}"
        lines = []
        for _ in range(int(n_gt * (0.5 + rng.rand())) if im % 7 != 6 else 0):
            cls = EVAL_CLASSES[int(rng.rand() < 0.3)]
            cx, cy = rng.rand(2) * extent
            w, h, t = rng.rand() * 70 + 8, rng.rand() * 25 + 5, (rng.rand() - 0.5) * np.pi
            q = quad(cx, cy, w, h, t)
            tail = f" {cls}" + ("" if rng.rand() < 0.1 else f" {int(rng.rand() < 0.15)}")
            lines.append(' '.join(f"{v:.1f}" for v in q) + tail)
            for _ in range(rng.randint(0, 4)):
                dc = cls if rng.rand() < 0.9 else EVAL_CLASSES[1 - EVAL_CLASSES.index(cls)]
                j = rng.randn(2) * (1 + 6 * rng.rand())
                qq = quad(cx + j[0], cy + j[1], w * (1 + rng.randn() * 0.08), h * (1 + rng.randn() * 0.08), t + rng.randn() * 0.05)
                k = rng.rand()
                if k < 0.02:
                    qq = [qq[0], qq[1]] * 4
                score = round(float(rng.rand()), 2 if rng.rand() < 0.7 else 1)
                det[dc].append(f"{name} {score} " + ' '.join(f"{v:.1f}" for v in qq))
        lines.append("imagesource:GoogleEarth")          # short lines are skipped by parse_gt
        for _ in range(rng.randint(0, 6)):                # clutter
            q = quad(rng.rand() * extent, rng.rand() * extent, rng.rand() * 70 + 8, rng.rand() * 25 + 5, (rng.rand() - 0.5) * np.pi)
            det[EVAL_CLASSES[int(rng.rand() < 0.5)]].append(f"{name} {round(float(rng.rand()), 2)} " + ' '.join(f"{v:.1f}" for v in q))
        gt[name] = lines
    for c in det:
        rng.shuffle(det[c])
    return gt, det


def eval_write(root, gt, det):
    """Lay the set out on disk the way the devkit expects it; returns (detpath, annopath, imagesetfile)."""
    os.makedirs(os.path.join(root, 'labelTxt')); os.makedirs(os.path.join(root, 'res'))
    for name, lines in gt.items():
        with open(os.path.join(root, 'labelTxt', name + '.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    with open(os.path.join(root, 'imgnamefile.txt'), 'w') as f:
        f.write('\n'.join(gt) + '\n')
    for c, lines in det.items():
        with open(os.path.join(root, 'res', f'Task1_{c}.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    return os.path.join(root, 'res', 'Task1_{:s}.txt'), os.path.join(root, 'labelTxt', '{:s}.txt'), os.path.join(root, 'imgnamefile.txt')


def load_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub('cv2', setNumThreads=lambda n: None)          # utils/general.py:45 only needs this at import
    stub('torchvision'); stub('torchvision.ops'); stub('seaborn')
    os.environ.setdefault('YOLOV5_CONFIG_DIR', '/tmp/refcfg')
    os.makedirs(os.environ['YOLOV5_CONFIG_DIR'], exist_ok=True)
    sys.path.insert(0, REF)
    os.chdir(REF)                                       # Arial.ttf is looked up in the cwd (utils/plots.py:56)
    spec = importlib.util.spec_from_file_location('utils.nms_rotated.nms_rotated_ext',
                                                  os.path.join(ROOT, 'oracle/_ref/nms_rotated_ext.so'))
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    sys.modules['utils.nms_rotated.nms_rotated_ext'] = ext
    import utils.general as G
    import utils.rboxs_utils as R
    import utils.loss as L
    import utils.metrics as M
    import models.yolo as Y
    return ext, G, R, L, M, Y


def main():
    oracle.build(with_ref=True)
    ext, G, R, L, M, Y = load_reference()
    torch.set_num_threads(1)
    out = {}
    f32p = np.ctypeslib.ndpointer(np.float32, flags='C')
    f64p = np.ctypeslib.ndpointer(np.float64, flags='C')

    # ------------------------------------------------------------------ A. IoU pairs
    dev = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libref_riou_dev.so'))
    host = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libref_riou_host.so'))
    pol = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libref_polyiou.so'))
    dev.ref_dev_riou_pairs_f32.argtypes = [f32p, f32p, C.c_int64, f32p]
    host.ref_host_riou_pairs_f32.argtypes = [f32p, f32p, C.c_int64, f32p]
    dev.ref_dev_riou_pairs_f64.argtypes = [f64p, f64p, C.c_int64, f64p]
    pol.ref_iou_poly_pairs.argtypes = [f64p, f64p, C.c_int64, f64p]
    n = 4000
    a, _ = synth.s_uniform(n, 101, extent=120.0)
    b, _ = synth.s_uniform(n, 102, extent=120.0)
    a[:200] = b[:200]                                           # identical
    a[200:400, 4] = 0; b[200:400, 4] = 0                        # axis aligned
    a[400:600, :4] = a[400:600, :4].round(); b[400:600, :4] = b[400:600, :4].round(); a[400:600, 4] = 0; b[400:600, 4] = 0
    a[600:650, 3] = 1e-9; b[650:700, 3] = 1e-9                  # degenerate thin boxes (non-local in the reference)
    a[600:700, :2] *= 8; b[600:700, :2] *= 8                    # ... far apart
    a[700:800, :2] += 4096 * 5; b[700:800, :2] += 4096 * 5      # class-offset magnitudes
    deg = np.pi / 180
    kat = np.array([[0, 0, 2, 2, 0, 1, 1, 2, 2, 0], [0, 0, 2, 2, 0, 0, 0, 2, 2, np.pi / 4],
                    [136.6, 111.6, 200, 100, -60 * deg, 136.6, 111.6, 100, 200, -30 * deg],
                    [136.6, 111.6, 200, 100, -60 * deg, 100, 100, 141.4, 141.4, -45 * deg],
                    [136.6, 111.6, 100, 200, -30 * deg, 100, 100, 141.4, 141.4, -45 * deg],
                    [0, 0, 4, 2, 0, 0, 0, 4, 2, np.pi / 2]], np.float32)
    an = np.ascontiguousarray(np.concatenate([a.numpy(), kat[:, :5]]))
    bn = np.ascontiguousarray(np.concatenate([b.numpy(), kat[:, 5:]]))
    m = len(an)
    rd = np.empty(m, np.float32); rh = np.empty(m, np.float32)
    dev.ref_dev_riou_pairs_f32(an.reshape(-1), bn.reshape(-1), m, rd)
    host.ref_host_riou_pairs_f32(an.reshape(-1), bn.reshape(-1), m, rh)
    rd64 = np.empty(m, np.float64)
    dev.ref_dev_riou_pairs_f64(an.astype(np.float64).reshape(-1), bn.astype(np.float64).reshape(-1), m, rd64)
    assert np.array_equal(oracle.riou_pairs(an, bn).view(np.uint32), rd.view(np.uint32)), "oracle != reference (device branch)"
    assert np.array_equal(oracle.riou_pairs(an.astype(np.float64), bn.astype(np.float64)).view(np.uint64), rd64.view(np.uint64))
    kat_bits = np.array([0x3e124925, 0x3f3504f3, 0x3ecfc89d, 0x3ef26639, 0x3ed6fded, 0x3eaaaaab], np.uint32)
    assert np.array_equal(rd[-6:].view(np.uint32), kat_bits), "survey KATs"
    out.update(riou_a=an, riou_b=bn, riou_ref_dev=rd, riou_ref_host=rh, riou_ref_dev_f64=rd64)
    print(f"riou: {m} pairs, nonzero {np.mean(rd > 0):.2f}, host!=dev on {np.count_nonzero(rd.view(np.uint32) != rh.view(np.uint32))}")

    qa = synth.rbox_to_quad(synth.s_uniform(1500, 103, extent=90.0)[0]).double().numpy()
    qb = synth.rbox_to_quad(synth.s_uniform(1500, 104, extent=90.0)[0]).double().numpy()
    qb[:100] = qa[:100]
    qb[100:300] = qb[100:300].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)     # clockwise rings
    qa, qb = np.ascontiguousarray(qa), np.ascontiguousarray(qb)
    pr = np.empty(len(qa), np.float64)
    pol.ref_iou_poly_pairs(qa.reshape(-1), qb.reshape(-1), len(qa), pr)
    mine = np.array([oracle.lib().oracle_piou_f64(qa[i], qb[i]) for i in range(len(qa))])
    assert np.array_equal(mine.view(np.uint64), pr.view(np.uint64)), "oracle quad IoU (double) != polyiou.cpp"
    assert abs(oracle.lib().oracle_piou_f64(np.array([0, 0, 1, 0, 1, 1, 0, 1.]), np.array([.5, .5, 1.5, .5, 1.5, 1.5, .5, 1.5])) - 1 / 7) < 1e-15
    out.update(piou_a=qa, piou_b=qb, piou_ref_f64=pr)
    print(f"piou: {len(qa)} pairs, nonzero {np.mean(pr > 0):.2f}")

    # ------------------------------------------------------------------ B. NMS kept sets (reference CPU extension)
    cases = {
        'uniform_1500_t0.4': (*synth.s_uniform(1500, 0), 0.4),
        'uniform_700_t0.1': (*synth.s_uniform(700, 1), 0.1),
        'clustered_4000_t0.45': (*synth.s_clustered(4000, 120, 2), 0.45),
        'clustered_2500_t0.2': (*synth.s_clustered(2500, 60, 3), 0.2),
        'classes_3000_t0.4': (synth.with_classes(synth.s_clustered(3000, 50, 4)[0], 16, 4)[0], synth.s_clustered(3000, 50, 4)[1], 0.4),
    }
    for name, (d, s, thr) in cases.items():
        s = synth.tie_free(s)
        ref = ext.nms_rotated(d, s, thr).numpy()                               # nms_rotated_cpu: >= compare, host hull
        o_ge = oracle.nms_rotated(d.numpy(), s.numpy(), thr, ge=True)
        o_gt = oracle.nms_rotated(d.numpy(), s.numpy(), thr, ge=False)
        assert np.array_equal(ref, o_ge), name
        assert np.array_equal(o_ge, o_gt), name + ": >= and > differ on this fixture, pick another seed"
        out['nms_' + name] = ref
        print(f"nms {name}: kept {len(ref)} of {len(d)}")

    # ------------------------------------------------------------------ C. non_max_suppression_obb (reference python, CPU)
    nms_obb_cases = {
        'ml_conf0.25': dict(seed=11, bs=2, A=2500, nc=15, kw=dict(conf_thres=0.25, iou_thres=0.45, multi_label=True)),
        'ml_conf0.01': dict(seed=12, bs=2, A=2500, nc=15, kw=dict(conf_thres=0.01, iou_thres=0.4, multi_label=True)),
        'best_conf0.1': dict(seed=13, bs=2, A=2500, nc=16, kw=dict(conf_thres=0.1, iou_thres=0.45, multi_label=False)),
        'agnostic': dict(seed=14, bs=1, A=3000, nc=3, kw=dict(conf_thres=0.05, iou_thres=0.2, multi_label=True, agnostic=True)),
        'classes': dict(seed=15, bs=2, A=2000, nc=15, kw=dict(conf_thres=0.05, iou_thres=0.45, multi_label=True, classes=[0, 3, 7])),
        'maxdet': dict(seed=16, bs=1, A=4000, nc=15, kw=dict(conf_thres=0.002, iou_thres=0.45, multi_label=True, max_det=50)),
        'nc1': dict(seed=17, bs=1, A=1500, nc=1, kw=dict(conf_thres=0.05, iou_thres=0.45, multi_label=True)),
        'fp16_ml': dict(seed=18, bs=2, A=2500, nc=15, half=True, kw=dict(conf_thres=0.05, iou_thres=0.45, multi_label=True)),
    }
    for name, cfg in nms_obb_cases.items():
        pred = synth.s_pred(cfg['bs'], cfg['A'], cfg['nc'], cfg['seed'], dtype=torch.float16 if cfg.get('half') else torch.float32)
        ref = G.non_max_suppression_obb(pred.clone(), **cfg['kw'])
        mine = pyref.non_max_suppression_obb(pred.clone(), ge=True, **cfg['kw'])
        mine_gt = pyref.non_max_suppression_obb(pred.clone(), ge=False, **cfg['kw'])
        for bi, (r, m_, g_) in enumerate(zip(ref, mine, mine_gt)):
            if cfg.get('half'):
                # fp16 confidences tie; torch's CPU sort is unstable for n >= ~100, this project's rule is
                # ascending original index: compare up to the order inside equal-score groups
                assert r.shape == m_.shape and np.array_equal(synth.canon_rows(r), synth.canon_rows(m_)), (name, bi)
            else:
                assert r.shape == m_.shape and torch.equal(r, m_), (name, bi, r.shape, m_.shape)
            assert torch.equal(m_, g_), name + ": >= and > differ on this fixture"
            out[f'nmsobb_{name}_{bi}'] = r.numpy()
        print(f"nmsobb {name}: {[tuple(r.shape) for r in ref]}")

    # ------------------------------------------------------------------ D. CSL / box utils
    angles = np.concatenate([np.linspace(0, 179.999, 37), np.array([0.0, 0.4, 89.5, 90.0, 179.2, 45.3])])
    for sig in (2.0, 4.0, 6.0):
        ref = np.stack([R.gaussian_label_cpu(a, 180, 0, sig) for a in angles])
        mine = np.stack([pyref.gaussian_label(a, 180, 0, sig) for a in angles])
        assert np.array_equal(ref, mine)
        out[f'csl_sig{sig}'] = ref
    out['csl_angles'] = angles
    rb = synth.s_uniform(300, 21)[0]
    rp = R.rbox2poly(rb)
    assert torch.equal(rp, pyref.rbox2poly(rb)) and np.array_equal(R.rbox2poly(rb.numpy()), pyref.rbox2poly(rb.numpy()))
    hb = R.poly2hbb(rp)
    assert torch.equal(hb, pyref.poly2hbb(rp)) and np.array_equal(R.poly2hbb(rp.numpy()), pyref.poly2hbb(rp.numpy()))
    th = np.linspace(-7, 7, 50)
    assert np.array_equal(R.regular_theta(th), pyref.regular_theta(th))
    out.update(rbox2poly_ref=rp.numpy(), poly2hbb_ref=hb.numpy(), regular_theta_in=th, regular_theta_ref=R.regular_theta(th))
    print("csl/box utils ok")

    # ------------------------------------------------------------------ E. Detect decode (reference module)
    torch.manual_seed(5)
    nc = 3
    det = Y.Detect(nc=nc, anchors=synth.DEFAULT_ANCHORS, ch=(8, 16, 32))
    det.stride = torch.tensor(synth.DEFAULT_STRIDES)
    det.anchors /= det.stride.view(-1, 1, 1)                       # models/yolo.py:122
    det.eval()
    feats = [torch.randn(2, c, s, s) for c, s in zip((8, 16, 32), (16, 8, 4))]
    with torch.no_grad():
        z, raw = det([f.clone() for f in feats])
    mine = pyref.detect_decode(raw, det.anchors, det.stride)
    assert torch.allclose(z, mine, rtol=0, atol=0) or torch.equal(z, mine), (z - mine).abs().max()
    out.update(detect_raw0=raw[0].numpy(), detect_raw1=raw[1].numpy(), detect_raw2=raw[2].numpy(), detect_z=z.numpy())
    print("detect decode ok", tuple(z.shape))

    # ------------------------------------------------------------------ F. ComputeLoss
    # torch >= 1.11 rejects float-tensor bounds in clamp_ on int64 (utils/loss.py:267): same patch as SURVEY 8c(3),
    # applied from the outside so that no reference code is copied.
    _orig_clamp_ = torch.Tensor.clamp_
    def _clamp_(self, min=None, max=None):
        cv = lambda v: int(v.item()) if isinstance(v, torch.Tensor) and not self.is_floating_point() else v
        return _orig_clamp_(self, cv(min), cv(max))
    torch.Tensor.clamp_ = _clamp_

    class _M(torch.nn.Module):                                       # what ComputeLoss.__init__ reads (utils/loss.py:93-120)
        def __init__(self, nc, hyp):
            super().__init__()
            d = Y.Detect(nc=nc, anchors=synth.DEFAULT_ANCHORS, ch=(4, 4, 4))
            d.stride = torch.tensor(synth.DEFAULT_STRIDES)
            d.anchors /= d.stride.view(-1, 1, 1)
            self.model = torch.nn.ModuleList([d])
            self.hyp = hyp

    loss_cases = {'nt50': dict(bs=2, nc=16, nt=50, seed=31, sizes=[32, 16, 8], imgsz=256),
                  'nt0': dict(bs=2, nc=16, nt=0, seed=32, sizes=[16, 8, 4], imgsz=128),
                  'nt400': dict(bs=4, nc=15, nt=400, seed=33, sizes=[32, 16, 8], imgsz=256),
                  'smooth': dict(bs=2, nc=16, nt=80, seed=34, sizes=[32, 16, 8], imgsz=256, label_smoothing=0.1)}
    for name, cfg in loss_cases.items():
        hyp = synth.scaled_hyp(cfg['nc'], 1024)
        hyp['label_smoothing'] = cfg.get('label_smoothing', 0.0)
        p, t = synth.s_loss(cfg['bs'], cfg['nc'], cfg['nt'], cfg['seed'], imgsz=cfg['imgsz'], sizes=cfg['sizes'])
        if cfg['nt']:
            t[:, 2:6] *= 1.0      # pixels already; keep
        model = _M(cfg['nc'], hyp)
        cl = L.ComputeLoss(model)
        pr = [q.clone().requires_grad_(True) for q in p]
        loss, items = cl(pr, t.clone())
        loss.backward()
        spec = pyref.LossSpec(hyp, model.model[0].anchors, model.model[0].stride, cfg['nc'])
        pm = [q.clone().requires_grad_(True) for q in p]
        loss2, items2 = pyref.compute_loss(spec, pm, t.clone())
        loss2.backward()
        assert torch.allclose(loss, loss2, rtol=1e-6, atol=1e-7), (name, loss, loss2)
        assert torch.allclose(items, items2, rtol=1e-6, atol=1e-7), (name, items, items2)
        for q1, q2 in zip(pr, pm):
            assert torch.allclose(q1.grad, q2.grad, rtol=1e-5, atol=1e-8), name
        tg = cl.build_targets(pr, t.clone())
        tg2 = pyref.build_targets(spec, pm, t.clone())
        for i in range(3):
            b, a_, gj, gi = tg[2][i]
            assert torch.equal(b, tg2[i]['b']) and torch.equal(a_, tg2[i]['a']) and torch.equal(gj, tg2[i]['gj']) and torch.equal(gi, tg2[i]['gi'])
            assert torch.equal(tg[1][i], tg2[i]['tbox']) and torch.equal(tg[0][i], tg2[i]['tcls']) and torch.equal(tg[4][i], tg2[i]['csl'])
            out[f'loss_{name}_idx{i}'] = torch.stack((b, a_, gj, gi), 1).numpy()
            out[f'loss_{name}_tbox{i}'] = tg[1][i].numpy()
            out[f'loss_{name}_gradsum{i}'] = np.array([pr[i].grad.double().sum().item(), pr[i].grad.double().abs().sum().item()])
        out[f'loss_{name}_loss'] = loss.detach().numpy()
        out[f'loss_{name}_items'] = items.numpy()
        print(f"loss {name}: loss {loss.item():.6f} items {items.tolist()} n_pos {[len(tg[2][i][0]) for i in range(3)]}")
    torch.Tensor.clamp_ = _orig_clamp_

    # ------------------------------------------------------------------ G. post-NMS tail of val.py (process_batch, scale_polys chain)
    import val as V                                                   # the reference's val.py (imports with the stubs above)
    for name, (n, m, seed) in VALPOST_CASES.items():
        det, labels, iouv = valpost_inputs(n, m, seed)
        ref = V.process_batch(det.clone(), labels.clone(), iouv)
        mine = pyref.process_batch(det.clone(), labels.clone(), iouv)
        assert torch.equal(ref, mine), name
        out[f'pb_{name}'] = ref.numpy()
    d7, gain, pad = valpost_dets(600, 5)
    poly = R.rbox2poly(d7[:, :5])
    pred_poly = torch.cat((poly, d7[:, -2:]), 1)
    pred_hbb = torch.cat((G.xywh2xyxy(R.poly2hbb(pred_poly[:, :8])), pred_poly[:, -2:]), 1)
    pred_polyn = pred_poly.clone()
    G.scale_polys((1024, 1024), pred_polyn[:, :8], (1, 1), ((gain, gain), pad))
    pred_hbbn = torch.cat((G.xywh2xyxy(R.poly2hbb(pred_polyn[:, :8])), pred_polyn[:, -2:]), 1)
    mine = pyref.val_postprocess(d7.clone(), gain, pad)
    for a_, b_ in zip((pred_poly, pred_hbb, pred_polyn, pred_hbbn), mine):
        assert torch.equal(a_, b_)
    out.update(vp_poly=pred_poly.numpy(), vp_hbb=pred_hbb.numpy(), vp_polyn=pred_polyn.numpy(), vp_hbbn=pred_hbbn.numpy())
    print("val.py tail ok")

    # ------------------------------------------------------------------ H. ResultMerge (tile -> full image, poly NMS 0.2)
    import tempfile
    pol.ref_iou_poly.restype = C.c_double
    pol.ref_iou_poly.argtypes = [f64p, f64p]
    sys.modules.setdefault('shapely', types.ModuleType('shapely'))
    sys.modules['shapely.geometry'] = types.ModuleType('shapely.geometry')
    stubp = types.ModuleType('DOTA_devkit.polyiou')                    # the SWIG module, backed by polyiou.cpp compiled in place
    stubp.VectorDouble = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    stubp.iou_poly = lambda p, q: pol.ref_iou_poly(p, q)
    sys.modules['DOTA_devkit.polyiou'] = stubp
    import DOTA_devkit
    DOTA_devkit.polyiou = stubp
    import DOTA_devkit.ResultMerge_multi_process as RM
    for name, cfg in MERGE_CASES.items():
        lines = merge_input_lines(*cfg)
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, 'src')); os.makedirs(os.path.join(td, 'dst'))
            with open(os.path.join(td, 'src', 'Task1_plane.txt'), 'w') as f:
                f.write('\n'.join(lines) + '\n')
            RM.mergebase(os.path.join(td, 'src'), os.path.join(td, 'dst'), RM.py_cpu_nms_poly_fast)
            ref_text = open(os.path.join(td, 'dst', 'Task1_plane.txt')).read()
        mine = '\n'.join(pyref.merge_result_lines(lines)) + '\n'
        assert mine == ref_text, name
        out[f'merge_{name}'] = np.array(ref_text)
        print(f"merge {name}: {len(lines)} lines in, {ref_text.count(chr(10))} out")

    # ------------------------------------------------------------------ I. DOTA Task-1 evaluation (voc_eval with polygon IoU)
    import contextlib, io
    sys.modules['polyiou'] = stubp                                     # dota_evaluation_task1.py:18 imports the SWIG module by its bare name
    sys.modules.setdefault('matplotlib', types.ModuleType('matplotlib'))
    sys.modules.setdefault('matplotlib.pyplot', types.ModuleType('matplotlib.pyplot'))
    import DOTA_devkit.dota_evaluation_task1 as EV
    for name, cfg in EVAL_CASES.items():
        gt, det = eval_inputs(*cfg)
        with tempfile.TemporaryDirectory() as td:
            detpath, annopath, imagesetfile = eval_write(td, gt, det)
            parsed = {k: EV.parse_gt(annopath.format(k)) for k in gt}
            for ci, cls in enumerate(EVAL_CLASSES):
                for m07 in (True, False):
                    with contextlib.redirect_stdout(io.StringIO()):
                        rec, prec, ap = EV.voc_eval(detpath, annopath, imagesetfile, cls, ovthresh=0.5, use_07_metric=m07)
                    rec2, prec2, ap2 = pyref.task1_voc_eval(parsed, list(gt), det[cls], cls, 0.5, m07)
                    assert np.array_equal(rec, rec2) and np.array_equal(prec, prec2) and ap == ap2, (name, cls)
                    out[f'eval_{name}_{cls}_ap{int(m07)}'] = np.array(ap)
                out[f'eval_{name}_{cls}_rec'] = rec
                out[f'eval_{name}_{cls}_prec'] = prec
                print(f"eval {name} {cls}: nd {len(rec)} ap {ap:.4f}")

    np.savez_compressed(os.path.join(HERE, 'reference_outputs.npz'), **out)
    sz = os.path.getsize(os.path.join(HERE, 'reference_outputs.npz'))
    print(f"wrote tests/golden/reference_outputs.npz ({sz / 1024:.0f} KiB, {len(out)} arrays)")


if __name__ == '__main__':
    main()
