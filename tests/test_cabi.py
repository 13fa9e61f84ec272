"""CPU: the C-ABI library loads and exports exactly what include/obb_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "obb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(obb_[a-z0-9_]+|_poly_nms|_overlaps)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from yolov5_obb_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/obb_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "yolov5_obb_amd/_lib.py binding table out of sync with the header"
    assert _lib.lib().obb_version().startswith(b"obb_hip")


def test_no_cpu_fallback_in_product():
    """The product must not import the oracle, and must refuse CPU tensors loudly."""
    import torch
    from yolov5_obb_amd import nms_rotated_ext
    with pytest.raises(RuntimeError):
        nms_rotated_ext.nms_rotated(torch.zeros(4, 5), torch.zeros(4), 0.5)
    pkg = os.path.join(ROOT, "yolov5_obb_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
