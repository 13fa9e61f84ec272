"""CPU, build container only (needs /root/reference): yolov5_obb_amd.dropin.install() makes the reference's own scripts bind
this package's hot path without any edit of the reference tree.  Runs in a subprocess: it rewires sys.modules."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "val.py")), reason="needs the reference checkout")

SNIPPET = r'''
import os, sys, types
def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
stub("cv2", setNumThreads=lambda n: None); stub("torchvision"); stub("torchvision.ops"); stub("seaborn")
os.environ.setdefault("YOLOV5_CONFIG_DIR", "/tmp/refcfg"); os.makedirs(os.environ["YOLOV5_CONFIG_DIR"], exist_ok=True)
sys.path.insert(0, REF); sys.path.insert(0, ROOT); os.chdir(REF)
import yaml
def _hyp(model):
    with open(os.path.join(REF, "data/hyps/obb/hyp.finetune_dota.yaml")) as f:
        model.hyp = yaml.safe_load(f)
    return model
import yolov5_obb_amd.dropin as dropin
from yolov5_obb_amd.utils import general as G, loss as L, nms_rotated as N
from yolov5_obb_amd.models import yolo as Y
from yolov5_obb_amd import nms_rotated_ext as E
changed = dropin.install(reference_cpu_ext=REF_EXT)
assert dropin.install() == changed                       # idempotent
import val, utils.general, utils.loss, models.yolo, utils.nms_rotated
from utils.nms_rotated import nms_rotated_ext
# what the reference's scripts bind: device dispatchers whose GPU side is this package's HIP path
assert val.non_max_suppression_obb is utils.general.non_max_suppression_obb
assert utils.general.non_max_suppression_obb.hip is G.non_max_suppression_obb and utils.general.obb_nms.hip is N.obb_nms
assert utils.loss.ComputeLoss.hip is L.ComputeLoss and models.yolo.Detect is Y.Detect
assert utils.nms_rotated.obb_nms is N.obb_nms and nms_rotated_ext is E
assert "utils.nms_rotated" in sys.modules and not getattr(sys.modules["utils.nms_rotated"], "__file__", "").startswith(REF)
# the reference's parse_model finds our Detect by name
assert eval("Detect", vars(models.yolo)) is Y.Detect

# ---- BASELINE configs[0] (--device cpu): CPU tensors run the REFERENCE'S OWN code (its non_max_suppression_obb, its
# obb_nms wrapper, its compiled nms_rotated_cpu), exactly the rows the golden fixtures froze from the untouched reference
import numpy as np, torch
from tests import synth
from tests.test_oracle_golden import G as GOLD, NMSOBB_CASES, nmsobb_input
for name in ("ml_conf0.25", "best_conf0.1", "agnostic", "maxdet"):
    cfg = NMSOBB_CASES[name]
    out = utils.general.non_max_suppression_obb(nmsobb_input(cfg), **cfg["kw"])
    assert len(out) == cfg["bs"]
    for bi, o in enumerate(out):
        assert o.device.type == "cpu" and np.array_equal(o.numpy(), GOLD[f"nmsobb_{name}_{bi}"]), (name, bi)
d, sc = synth.s_clustered(800, 40, 3)
_, keep = utils.general.obb_nms(d, synth.tie_free(sc), 0.3)                    # CPU tensors -> nms_rotated_cpu (>=)
import oracle
assert np.array_equal(keep.numpy(), oracle.nms_rotated(d.numpy(), synth.tie_free(sc).numpy(), 0.3, ge=True))
# a GPU-only call on CPU tensors still fails loudly in the package itself (no fallback below the dispatch level)
try:
    G.non_max_suppression_obb(nmsobb_input(NMSOBB_CASES["nc1"]))
    raise SystemExit("the package ran a CPU tensor")
except RuntimeError:
    pass
# ---- the reference's model on the CPU with our Detect class in it: eval forward = the reference's own Detect.forward
from models.yolo import Model
m = Model(os.path.join(REF, "models/yolov5n.yaml"), ch=3, nc=16).eval()
det = m.model[-1]
assert type(det) is Y.Detect
with torch.no_grad():
    z, xs = m(torch.zeros(1, 3, 64, 64))
assert z.shape == (1, 3 * (8 * 8 + 4 * 4 + 2 * 2), 16 + 185) and len(xs) == 3
# the oracle's restatement of this network (bench.py's `detect.py --device cpu` baseline on the GPU box, where the reference
# tree does not exist) has the reference model's parameter count and output shape
from oracle import pyref_model
pm = pyref_model.YoloV5nObb(16).eval()
assert sum(q.numel() for q in pm.parameters()) == sum(q.numel() for q in m.parameters())
with torch.no_grad():
    assert pm(torch.zeros(1, 3, 64, 64)).shape == z.shape
# ADVICE r1: a module that has run inference pickles, and the pickle names models.yolo.Detect (loadable by a plain checkout)
import io, pickle, pickletools
det._host_tables()
buf = io.BytesIO(); torch.save(m, buf)
assert b"models.yolo" in buf.getvalue() and b"yolov5_obb_amd.models.yolo" not in buf.getvalue()
buf.seek(0); m2 = torch.load(buf, weights_only=False)
assert type(m2.model[-1]) is Y.Detect and m2.model[-1]._host_tables()[1] == det._host_tables()[1]
# a Detect unpickled from a reference-made checkpoint has no _anchor_px in its __dict__: the class default covers it
del det.__dict__["_anchor_px"]
assert det._host_tables()[1] == [8.0, 16.0, 32.0]
# the loss of a CPU model is the reference's class, of a GPU model ours (not constructible here: no GPU)
assert type(utils.loss.ComputeLoss(_hyp(m))).__module__ == "utils.loss"
dropin.uninstall()
assert utils.general.non_max_suppression_obb is not G.non_max_suppression_obb and models.yolo.Detect is not Y.Detect
assert Y.Detect.__module__ == "yolov5_obb_amd.models.yolo" and Y.Detect._cpu_forward is None
assert sys.modules.get("utils.nms_rotated") is not N
print("dropin ok", len(changed))
'''


def test_install_rebinds_the_reference_hot_path(oracle_lib):
    import glob
    import oracle
    oracle.build(with_ref=True)                         # oracle/_ref/nms_rotated_ext.so = the reference's own CPU extension, compiled in place
    ext = glob.glob(os.path.join(ROOT, "oracle", "_ref", "nms_rotated_ext*.so"))
    assert ext, "oracle/_ref/nms_rotated_ext.so missing"
    code = f"REF = {REF!r}\nROOT = {ROOT!r}\nREF_EXT = {ext[0]!r}\n" + SNIPPET
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "dropin ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
