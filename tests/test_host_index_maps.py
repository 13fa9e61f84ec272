"""CPU: the pure index arithmetic of two kernels, replayed in numpy with the device's types (float32 sqrt, int32).

  * nms_pairs (csrc/nms_core.h): item t of the upper triangle of nb x nb tiles -> (row tile, column tile) through a float
    square root estimate that is fixed up with the exact row offsets;
  * k_decode (csrc/nmsobb_impl.h): 16-row chunks dealt round-robin to the workgroups of an image -- every row of the image
    must be read by exactly one (workgroup, thread, q), for every G the host picks.
A mistake in either would not crash: it would silently skip or repeat work."""
import numpy as np
import pytest


def _row_off(rb, nb):
    return rb * nb - ((rb * (rb - 1)) >> 1)


@pytest.mark.parametrize("nb", [1, 2, 3, 7, 32, 64, 127, 128, 129, 255, 256, 469])
def test_upper_triangle_tile_enumeration_is_a_bijection(nb):
    tri = nb * (nb + 1) // 2
    item = np.arange(tri, dtype=np.int64)
    a = np.float32(2 * nb + 1)
    disc = (np.int64(2 * nb + 1) * np.int64(2 * nb + 1) - 8 * item).astype(np.float32)      # the device casts the int to float
    rb = ((a - np.sqrt(disc, dtype=np.float32)) * np.float32(0.5)).astype(np.int32).astype(np.int64)
    rb = np.clip(rb, 0, nb - 1)
    for _ in range(4):                                                       # the two fix-up loops (they move by one or two)
        up = (rb + 1 < nb) & (_row_off(np.minimum(rb + 1, nb - 1), nb) <= item)
        rb = np.where(up, rb + 1, rb)
    for _ in range(4):
        down = _row_off(rb, nb) > item
        rb = np.where(down, rb - 1, rb)
    assert not ((rb + 1 < nb) & (_row_off(np.minimum(rb + 1, nb - 1), nb) <= item)).any()    # the loops had converged
    assert not (_row_off(rb, nb) > item).any()
    cb = rb + (item - _row_off(rb, nb))
    assert (rb >= 0).all() and (cb >= rb).all() and (cb < nb).all()
    assert len({(int(r), int(c)) for r, c in zip(rb, cb)}) == tri             # every tile of the triangle exactly once


def _rows_per_thread(bs, A, threads=512):
    return 4 if bs * A >= 4 * threads * 480 else (2 if bs * A >= 2 * threads * 480 else 1)


@pytest.mark.parametrize("bs,A", [(16, 64512), (1, 114627), (1, 1), (1, 15), (2, 511), (1, 512), (3, 8193), (8, 64512), (4, 21504), (1, 1500)])
def test_decode_chunk_dealing_covers_every_row_once(bs, A):
    threads, G = 512, _rows_per_thread(bs, A)
    nwg = (A + threads * G - 1) // (threads * G)
    tid = np.arange(threads)
    seen = np.zeros(A, dtype=np.int32)
    for wg in range(nwg):
        for q in range(G):
            row = ((q * (threads // 16) + (tid >> 4)) * nwg + wg) * 16 + (tid & 15)
            row = row[row < A]
            np.add.at(seen, row, 1)
    assert (seen == 1).all()
    assert threads * G <= 512 * 4                                             # the workgroup's LDS row list has 2048 entries
