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


def test_argument_checks_answer_before_any_device_call():
    """Entries validate their arguments first (OBB_ERR_BAD_ARG = -1, nothing is launched): callable without a GPU."""
    import ctypes as C
    from yolov5_obb_amd import _lib
    L = _lib.lib()
    null = C.c_void_p(0)
    one = (C.c_int64 * 1)(8)
    ptrs = (C.c_void_p * 1)(16)
    f2 = (C.c_float * 6)(*([1.0] * 6))
    # obb_detect_decode_levels: level count out of range, missing tables, too many anchors, unknown dtype
    for nl, conv, dtype, na in ((0, ptrs, 1, 3), (5, ptrs, 1, 3), (1, null, 1, 3), (1, ptrs, 7, 3), (1, ptrs, 1, 0), (1, ptrs, 1, 1000)):
        rc = L.obb_detect_decode_levels(nl, conv, dtype, 2, na, 21, one, one, C.cast(f2, C.c_void_p), C.cast(f2, C.c_void_p), null, null, 0, null, null)
        assert rc == -1, (nl, dtype, na, rc)
    # obb_detect_decode: no conv output / no anchors
    assert L.obb_detect_decode(null, 1, 2, 3, 21, 8, 8, C.cast(f2, C.c_void_p), 8.0, null, null, 0, 0, null) == -1
    # quad NMS: rows shorter than [8 coordinates, score]
    assert L.obb_nms_poly_f32(null, 8, 0, 0.1, 0, null, null, null, 0, null) == -1


def test_compiled_binding_loads_and_binds_the_same_library():
    """nms_rotated_ext_c (csrc/torch_ext/nms_rotated_ext.cpp; the reference's nms_rotated_ext is a pybind11 torch extension,
    utils/nms_rotated/src/nms_rotated_ext.cpp:57-60) loads, binds the library _lib.py loaded, and checks its arguments before any
    device call -- with the reference's error types."""
    import torch
    from yolov5_obb_amd import _lib
    if not os.path.exists(_lib.EXT_PATH):
        import __graft_entry__
        __graft_entry__.build_torch_ext()
        _lib._ext_tried = False
    ext = _lib.compiled()
    assert ext is not None and ext.library() == _lib.LIB_PATH
    for name in ("nms_rotated", "nms_poly", "non_max_suppression_obb", "val_tail_batch"):
        assert hasattr(ext, name)
    with pytest.raises(RuntimeError):                      # CPU tensors: no CPU path in this build
        ext.nms_rotated(torch.zeros(4, 5), torch.zeros(4), 0.5)
    with pytest.raises(RuntimeError, match="not implemented on CPU"):      # AT_ERROR of nms_rotated_ext.cpp:54
        ext.nms_poly(torch.zeros(4, 9), 0.5)
    with pytest.raises(TypeError):
        ext.nms_rotated([1.0], torch.zeros(1), 0.5)
    with pytest.raises(RuntimeError):
        ext.non_max_suppression_obb(torch.zeros(1, 8, 201))
    # the source of the binding contains no device code and reaches the library through the header only
    src = open(os.path.join(ROOT, "yolov5_obb_amd", "csrc", "torch_ext", "nms_rotated_ext.cpp")).read()
    assert '#include "obb_hip.h"' in src and "__global__" not in src and "hipLaunch" not in src


def test_ctypes_binding_can_be_forced(monkeypatch):
    """OBB_BINDING=ctypes keeps the fallback binding (the same C ABI through ctypes) reachable."""
    import subprocess
    import sys
    code = ("import os; os.environ['OBB_BINDING'] = 'ctypes'; from yolov5_obb_amd import _lib; "
            "assert _lib.compiled() is None; print(_lib.lib().obb_version().decode())")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "obb_hip" in out.stdout, out.stderr[-400:]


def test_compiled_binding_builds_its_output_views_like_as_strided():
    """The binding makes the per-image views it returns directly (csrc/torch_ext/nms_rotated_ext.cpp: view_of) instead of one
    dispatcher call per view: same storage, offsets, strides and values as torch.as_strided, writes go through to the base."""
    import torch
    from yolov5_obb_amd import _lib
    ext = _lib.compiled()
    assert ext is not None
    t = torch.arange(240 * 7, dtype=torch.float32).view(240, 7)
    for row0, rows in ((0, 0), (0, 5), (30, 17), (239, 1)):
        v = ext._view_of(t, row0 * 7, [rows, 7], [7, 1])
        w = torch.as_strided(t, (rows, 7), (7, 1), row0 * 7)
        assert v.shape == w.shape and v.stride() == w.stride() and v.storage_offset() == w.storage_offset() and torch.equal(v, w)
        assert v.dtype == t.dtype and v.device == t.device and not v.requires_grad and v.is_contiguous()
    b = t[10:]                                               # a base with a storage offset of its own
    assert torch.equal(ext._view_of(b, 7 * 5, [3, 7], [7, 1]), b[5:8])
    c = torch.stack((torch.arange(50.), torch.arange(50.) + 100), 1)
    assert torch.equal(ext._view_of(c, 2 * 10 + 1, [7], [2]), c[10:17, 1])
    m = torch.zeros(20, 10, dtype=torch.bool)
    v = ext._view_of(m, 3 * 10, [4, 10], [10, 1])
    v[1, 2] = True
    assert m[4, 2] and int(m.sum()) == 1 and v.dtype == torch.bool
    assert (v + 0).sum() == 1 and torch.cat([v, v]).shape == (8, 10)          # ordinary ops accept the views
