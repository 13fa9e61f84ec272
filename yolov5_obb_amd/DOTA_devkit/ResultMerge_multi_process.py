"""Tile -> full-image merge of DOTA Task-1 result files on the GPU: the interface of the reference's
DOTA_devkit/ResultMerge_multi_process.py (`py_cpu_nms_poly_fast`, `py_cpu_nms_poly`, `py_cpu_nms`, `nmsbynamedict`,
`poly2origpoly`, `mergesingle`, `mergebase`, `mergebase_parallel`, `mergebypoly`, `mergebyrec`, `nms_thresh`), same on-disk
formats in and out.

The reference runs one Python process per class file (`Pool(16)`, :238-244) and, inside each, a Python double loop over
SWIG `polyiou.iou_poly` calls per source image (:62-123).  Here a class file is ONE device call
(`obb_merge_nms_poly_f64`, include/obb_hip.h): every source image is a segment of the persistent NMS kernel, the
polygon IoU is polyiou.cpp's algorithm in IEEE double, and the host keeps only what is text: parsing, numpy's
`argsort()[::-1]` per image (so that score ties are ordered exactly like the reference's) and the rounded output lines.
With several ranks (torch.distributed initialised) the class files are split by rank; there is no collective.

GPU only: no CPU fallback (the reference's own CPU code is the fallback a user already has).
"""
import os
import re
import shutil

import numpy as np
import torch

from .. import _lib
from ..utils import shard

nms_thresh = 0.2                                   # ResultMerge_multi_process.py:22

_TILE_XY = re.compile(r'__\d+___\d+')              # :196
_TILE_RATE = re.compile(r'__([\d+\.]+)__\d+___')   # :201
_INT = re.compile(r'\d+')


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("yolov5_obb_amd ResultMerge needs a HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def merge_nms_segments(dets, orders, thresh, variant="poly_fast"):
    """dets (n,9) float64 host array ((n, >= 4) for variant "hbb"); orders: list of int arrays (row indices in processing order, one
    per segment).  Returns a list of kept row-index arrays (processing order).  One device call for all segments.
    variant: "poly_fast" (py_cpu_nms_poly_fast), "poly_all" (py_cpu_nms_poly: no horizontal-box gate), "hbb" (py_cpu_nms)."""
    dev = _device()
    dets = np.ascontiguousarray(dets, dtype=np.float64)
    dets = dets.reshape(-1, 9) if variant != "hbb" else dets.reshape(len(dets), -1)
    nseg = len(orders)
    if nseg == 0:
        return []
    lens = np.array([len(o) for o in orders], dtype=np.int64)
    off = np.zeros(nseg + 1, dtype=np.int32)
    off[1:] = np.cumsum(lens)
    n = int(off[-1])
    if n == 0:
        return [np.zeros(0, dtype=np.int64) for _ in orders]
    order = np.concatenate([np.asarray(o, dtype=np.int32) for o in orders]).astype(np.int32, copy=False)
    L = _lib.lib()
    d_dets = torch.from_numpy(dets).to(dev)
    d_order = torch.from_numpy(order).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    cnt = torch.empty(nseg, dtype=torch.int64, device=dev)
    if variant == "hbb":
        ws = _lib.workspace(L.obb_nms_workspace_bytes(n, nseg, 5), dev)
        rc = L.obb_merge_nms_hbb_f64(_lib.ptr(d_dets), dets.shape[1], dets.shape[0], _lib.ptr(d_order), _lib.ptr(d_off), nseg, float(thresh),
                                     _lib.ptr(keep), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    else:
        fn, kind = (L.obb_merge_nms_poly_all_f64, 4) if variant == "poly_all" else (L.obb_merge_nms_poly_f64, 2)
        ws = _lib.workspace(L.obb_nms_workspace_bytes(n, nseg, kind), dev)
        rc = fn(_lib.ptr(d_dets), dets.shape[0], _lib.ptr(d_order), _lib.ptr(d_off), nseg, float(thresh),
                _lib.ptr(keep), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "obb_merge_nms_*_f64")
    cnt_h = cnt.cpu().numpy()
    if (cnt_h < 0).any():
        raise RuntimeError("obb_merge_nms_poly_f64: device-side abort")
    keep_h = keep.cpu().numpy()
    return [keep_h[off[g]:off[g] + cnt_h[g]].copy() for g in range(nseg)]


def py_cpu_nms_poly_fast(dets, thresh):
    """Signature of ResultMerge_multi_process.py:62 -- (n,9) [x1 y1 .. x4 y4 score] -> list of kept row indices.
    (The name is the reference's; the work happens on the GPU.)"""
    dets = np.asarray(dets, dtype=np.float64).reshape(-1, 9)
    if len(dets) == 0:
        return []
    order = dets[:, 8].argsort()[::-1]             # :79
    return list(merge_nms_segments(dets, [order], thresh)[0])


def py_cpu_nms_poly(dets, thresh):
    """Signature of ResultMerge_multi_process.py:24 -- the variant without the horizontal-box gate: iou_poly of the kept box
    and every remaining candidate.  On the GPU like the fast one."""
    dets = np.asarray(dets, dtype=np.float64).reshape(-1, 9)
    if len(dets) == 0:
        return []
    order = dets[:, 8].argsort()[::-1]             # :35
    return list(merge_nms_segments(dets, [order], thresh, variant="poly_all")[0])


def py_cpu_nms(dets, thresh):
    """Signature of ResultMerge_multi_process.py:125 ("Pure Python NMS baseline"): horizontal boxes in columns 0..3, the score
    in column 4 of `dets`, whatever else the rows hold -- mergebyrec passes the nine-column rows of a class file."""
    dets = np.asarray(dets, dtype=np.float64)
    if dets.ndim != 2 or dets.shape[1] < 5:
        raise IndexError("py_cpu_nms: rows [x1 y1 x2 y2 score ...] expected")      # (the reference fails on dets[:, 4])
    if len(dets) == 0:
        return []
    order = dets[:, 4].argsort()[::-1]             # :136
    return list(merge_nms_segments(dets, [order], thresh, variant="hbb")[0])


def nmsbynamedict(nameboxdict, nms, thresh):
    """:159-174.  With one of this module's three NMS functions all images go to the device in ONE call (every image a segment of
    the persistent kernel, its own `argsort()[::-1]` as the processing order); any other callable is applied image by image like
    the reference does."""
    variant = {py_cpu_nms_poly_fast: ("poly_fast", 8), py_cpu_nms_poly: ("poly_all", 8), py_cpu_nms: ("hbb", 4)}.get(nms)
    if variant is not None:
        names = list(nameboxdict)
        arrs = [np.asarray(nameboxdict[k], dtype=np.float64) for k in names]
        # (rows of nine numbers, [8 coordinates, score]: what parse_result_file produces; anything else goes image by image, where
        #  the per-image functions raise what the reference raises)
        if all(a.ndim == 2 and a.shape[1] == 9 for a in arrs):
            base = np.cumsum([0] + [len(a) for a in arrs])
            orders = [a[:, variant[1]].argsort()[::-1] + base[i] for i, a in enumerate(arrs)]          # :79 / :35 / :136
            keeps = merge_nms_segments(np.concatenate(arrs) if arrs else np.zeros((0, 9)), orders, thresh, variant=variant[0])
            return {k: [nameboxdict[k][int(j - base[i])] for j in keeps[i]] for i, k in enumerate(names)}
    return {k: [nameboxdict[k][int(j)] for j in nms(np.array(nameboxdict[k]), thresh)] for k in nameboxdict}


def poly2origpoly(poly, x, y, rate):
    """:175-182: tile coordinates -> source-image coordinates."""
    r = float(rate)
    out = []
    for i in range(len(poly) // 2):
        out.append(float(poly[2 * i] + x) / r)
        out.append(float(poly[2 * i + 1] + y) / r)
    return out


def parse_result_file(fullname):
    """A Task1_<class>.txt file -> dict source image -> list of [8 source-image coordinates, confidence] (:186-213).
    Lines are `<orig>__<rate>__<x>___<y> score x1 y1 .. x4 y4`."""
    boxes = {}
    with open(fullname, 'r') as f:
        for line in f:
            tok = line.strip().split(' ')
            sub = tok[0]
            oriname = sub.split('__')[0]
            xy = _INT.findall(_TILE_XY.findall(sub)[0])
            x, y = int(xy[0]), int(xy[1])
            rate = _TILE_RATE.findall(sub)[0]
            det = poly2origpoly(list(map(float, tok[2:])), x, y, rate)
            det.append(float(tok[1]))
            boxes.setdefault(oriname, []).append(det)
    return boxes


def format_result_line(imgname, det):
    """:218-233: confidence to 2 decimals, coordinates to 1, Python's round() and str()."""
    return imgname + ' ' + str(round(det[-1], 2)) + ' ' + ' '.join(str(round(v, 1)) for v in det[:8])


class ResultTable:
    """A parsed Task1_<class>.txt file (native reader, include/obb_hip.h: obb_task1_parse_tiles): `names` of the source
    images in first-appearance order, `codes` (n) = source image of a line, `dets` (n, 9) float64 [8 source-image
    coordinates, confidence]; keeps the text so that the writer can copy the names."""

    def __init__(self, text, names, codes, dets, name_off, name_len):
        self.text, self.names, self.codes, self.dets, self.name_off, self.name_len = text, names, codes, dets, name_off, name_len

    def format_rows(self, rows):
        """The output lines (one bytes object) of the given input lines, format_result_line's text (obb_task1_format_rows)."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cap = 200 * len(rows) + int(self.name_len[rows].sum()) + 64 if len(rows) else 64
        out = np.empty(cap, dtype=np.uint8)
        w = _lib.lib().obb_task1_format_rows(self.text, self.name_off.ctypes.data, self.name_len.ctypes.data, self.dets.ctypes.data,
                                             rows.ctypes.data, len(rows), out.ctypes.data, cap)
        if w < 0:
            raise RuntimeError(f"obb_task1_format_rows failed ({w})")
        return out[:w].tobytes()


def parse_result_table(fullname):
    """The same parse as parse_result_file, one native pass over the file.  None when the file is not the plain layout
    (then the caller parses line by line, which behaves like the reference on such input)."""
    with open(fullname, 'rb') as f:
        text = f.read()
    max_lines = text.count(b'\n') + 1
    dets = np.empty((max_lines, 9), dtype=np.float64)
    name_off = np.empty(max_lines, dtype=np.int32)
    name_len = np.empty(max_lines, dtype=np.int32)
    group = np.empty(max_lines, dtype=np.int32)
    first = np.empty(max_lines, dtype=np.int32)
    import ctypes
    ng = ctypes.c_int64(0)
    n = _lib.lib().obb_task1_parse_tiles(text, len(text), max_lines, dets.ctypes.data, name_off.ctypes.data, name_len.ctypes.data,
                                         group.ctypes.data, first.ctypes.data, ctypes.addressof(ng))
    if n <= 0:
        return None
    try:
        names = [text[name_off[i]:name_off[i] + name_len[i]].decode() for i in first[:ng.value]]
    except UnicodeDecodeError:
        return None
    return ResultTable(text, names, group[:n].copy(), dets[:n].copy(), name_off[:n].copy(), name_len[:n].copy())


def format_result_rows(names, dets):
    """format_result_line for many rows at once (native): names per row, dets (n, 9)."""
    blob = ' '.join(names).encode()
    lens = np.array([len(x.encode()) for x in names], dtype=np.int32)
    offs = np.zeros(len(names), dtype=np.int32)
    if len(names) > 1:
        offs[1:] = np.cumsum(lens[:-1] + 1)
    t = ResultTable(blob, None, None, np.ascontiguousarray(dets, dtype=np.float64), offs, lens)
    return t.format_rows(np.arange(len(names))).decode().splitlines()


def mergesingle(dstpath, nms, fullname):
    """:183-234: one class file in, one merged class file out (same base name)."""
    name = os.path.basename(os.path.splitext(fullname)[0])
    dstname = os.path.join(dstpath, name + '.txt')
    table = parse_result_table(fullname) if nms is py_cpu_nms_poly_fast else None
    # (str(round(v, d)) is the d-decimal string of the native writer while that string has <= 15 significant digits)
    if table is not None and np.isfinite(table.dets).all() and np.abs(table.dets[:, :8]).max(initial=0.0) < 1e14 \
            and np.abs(table.dets[:, 8]).max(initial=0.0) < 1e13:
        codes, dets = table.codes, table.dets
        rows = np.argsort(codes, kind='stable')          # lines of one source image together, file order inside
        bounds = np.searchsorted(codes[rows], np.arange(len(table.names) + 1))
        orders = []
        for g in range(len(table.names)):
            idx = rows[bounds[g]:bounds[g + 1]]
            orders.append(idx[dets[idx, 8].argsort()[::-1]])          # :79 on this image's rows, numpy's own tie order
        keeps = merge_nms_segments(dets, orders, nms_thresh)
        with open(dstname, 'wb') as f:
            f.write(table.format_rows(np.concatenate(keeps) if keeps else np.zeros(0, dtype=np.int64)))
        return dstname
    merged = nmsbynamedict(parse_result_file(fullname), nms, nms_thresh)
    with open(dstname, 'w') as f:
        for imgname, dets in merged.items():
            for det in dets:
                f.write(format_result_line(imgname, det) + '\n')
    return dstname


def _files(srcpath):
    out = []
    for root, _, files in os.walk(srcpath):
        out.extend(os.path.join(root, f) for f in files)
    return out


def mergebase(srcpath, dstpath, nms):
    """:246-249."""
    for f in _files(srcpath):
        mergesingle(dstpath, nms, f)


def mergebase_parallel(srcpath, dstpath, nms):
    """:236-244.  The reference's Pool(16) becomes: class files split over the ranks of the default process group (one GPU
    each); a single process handles all of them."""
    files = sorted(_files(srcpath))
    for i in shard.shard_indices(len(files)):
        mergesingle(dstpath, nms, files[i])


def _fresh_dir_on_rank0(dstpath):
    """The reference deletes and recreates dstpath (:253-256, :267-270); under a multi-rank launch only rank 0 may, and the others
    wait until it is done (every rank writes its own class files into it afterwards)."""
    rank, world = shard.world()
    if rank == 0:
        if os.path.exists(dstpath):
            shutil.rmtree(dstpath)
        os.makedirs(dstpath)
    if world > 1:
        torch.distributed.barrier()


def mergebyrec(srcpath, dstpath):
    """:251-263: mergebase with py_cpu_nms.  Class files are split over the ranks like mergebypoly's (one process: all of them)."""
    _fresh_dir_on_rank0(dstpath)
    mergebase_parallel(srcpath, dstpath, py_cpu_nms)


def mergebypoly(srcpath, dstpath):
    """:265-281."""
    _fresh_dir_on_rank0(dstpath)
    mergebase_parallel(srcpath, dstpath, py_cpu_nms_poly_fast)
