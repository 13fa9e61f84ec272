"""CPU, build container only: the oracle against the reference's own sources compiled in place (oracle/_ref).
Skipped where /root/reference (hence oracle/_ref) does not exist, e.g. on the GPU box."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest
import torch

import oracle
from tests import synth

REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "libref_riou_dev.so")),
                                reason="oracle/_ref not built (needs /root/reference)")
f32p = np.ctypeslib.ndpointer(np.float32, flags='C')


def test_riou_100k_pairs_bit_exact(oracle_lib):
    dev = C.CDLL(os.path.join(REFDIR, "libref_riou_dev.so"))
    dev.ref_dev_riou_pairs_f32.argtypes = [f32p, f32p, C.c_int64, f32p]
    a, _ = synth.s_uniform(100000, 7, extent=100.0)
    b, _ = synth.s_uniform(100000, 8, extent=100.0)
    a, b = np.ascontiguousarray(a.numpy()), np.ascontiguousarray(b.numpy())
    ref = np.empty(len(a), np.float32)
    dev.ref_dev_riou_pairs_f32(a.reshape(-1), b.reshape(-1), len(a), ref)
    assert np.array_equal(oracle.riou_pairs(a, b).view(np.uint32), ref.view(np.uint32))


def test_nms_vs_reference_cpu_extension_fresh_seed(oracle_lib):
    spec = importlib.util.spec_from_file_location("nms_rotated_ext", os.path.join(REFDIR, "nms_rotated_ext.so"))
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    d, s = synth.s_clustered(3000, 90, 77)
    s = synth.tie_free(s)
    ref = ext.nms_rotated(d, s, 0.35).numpy()
    assert np.array_equal(oracle.nms_rotated(d.numpy(), s.numpy(), 0.35, ge=True), ref)
