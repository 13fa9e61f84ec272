// nms_rotated_ext -- the compiled torch binding of libobb_hip.so.
//
// The reference ships its rotated / polygon NMS as a pybind11 torch extension of this name
// (utils/nms_rotated/src/nms_rotated_ext.cpp:57-60: m.def("nms_rotated", ...), m.def("nms_poly", ...)).  This file is
// that module for the MI355X path: the same two functions with the same argument checks and return conventions, plus
// the two batch-level calls that sit on the same boundary in this repository -- non_max_suppression_obb
// (utils/general.py:772-862) and the post-NMS tail of val.py:209-250.  It contains no device code: every function
// prepares plain pointers and calls the C ABI declared in include/obb_hip.h, which it binds with dlopen / dlsym at
// init() time (the SAME shared object the ctypes binding of yolov5_obb_amd/_lib.py uses: one copy of the library's
// state per process, and OBB_HIP_LIB keeps selecting the build under test).
//
// What C++ buys over the ctypes binding (yolov5_obb_amd/nms_rotated_ext.py, utils/general.py, val.py keep that path as
// the fallback): no ctypes argument marshalling (19-26 arguments per call), the workspace straight from the caching
// allocator on the current stream, the count read-back polled without the interpreter, the per-image views and tuples
// built without a Python-level tensor call each.  PyTorch is plumbing here: device memory, the current stream.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "obb_hip.h"

namespace py = pybind11;

namespace {

// ---------------------------------------------------------------- the C ABI, bound at run time
#define OBB_API(X)                                                                                                         \
  X(obb_version) X(obb_nms_set_max_grid) X(obb_nms_workspace_bytes) X(obb_nms_rotated_f32) X(obb_nms_rotated_f64)       \
  X(obb_nms_poly_f32) X(obb_nms_obb_workspace_bytes) X(obb_nms_obb_state_bytes) X(obb_non_max_suppression_obb_st)                                   \
  X(obb_val_tail_batch_workspace_bytes) X(obb_val_tail_batch_rows_f32)

struct Api {
#define X(name) decltype(&::name) name = nullptr;
  OBB_API(X)
#undef X
  void* handle = nullptr;
  std::string path;
} api;

void need_api() {
  if (!api.handle) throw std::runtime_error("nms_rotated_ext: init(path of libobb_hip.so) has not been called");
}

const char* err_text(int rc) {
  switch (rc) {
    case OBB_ERR_BAD_ARG: return "bad argument";
    case OBB_ERR_WORKSPACE: return "workspace missing or too small";
    case OBB_ERR_LAUNCH: return "kernel launch failed";
    case OBB_ERR_INTERNAL: return "internal error";
    case OBB_ERR_NO_DEVICE: return "no usable HIP device";
    default: return "error";
  }
}
void check(int rc, const char* what) {          // same text as yolov5_obb_amd/_lib.py:check
  if (rc != OBB_OK) throw std::runtime_error(std::string(what) + " failed: " + err_text(rc) + " (code " + std::to_string(rc) + ")");
}

constexpr int64_t kPending = -(int64_t(1) << 62);

// ---------------------------------------------------------------- pinned words the last kernel of a call writes
struct Pinned {
  at::Tensor t;
  int64_t* p = nullptr;
  int64_t n = 0;
};
// one buffer per (thread, device, purpose, size): never handed out, re-armed before every call
// (The buffers are never freed: a pinned tensor released from a thread_local destructor at interpreter shutdown would call into a
//  host allocator that may be gone already; a few dozen bytes per (thread, device, batch size).)
Pinned& pinned_words(int dev, int purpose, int64_t n) {
  static thread_local std::map<std::tuple<int, int, int64_t>, Pinned*> memo;
  Pinned*& e = memo[std::make_tuple(dev, purpose, n)];
  if (!e) {
    e = new Pinned();
    e->t = at::empty({n}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    e->p = e->t.data_ptr<int64_t>();
    e->n = n;
  }
  return *e;
}
void arm(Pinned& w) {
  for (int64_t i = 0; i < w.n; i++) __atomic_store_n(w.p + i, kPending, __ATOMIC_RELAXED);
  std::atomic_thread_fence(std::memory_order_seq_cst);
}
// Poll until no word is PENDING (the GIL is released by the caller).  Busy for 2 ms (a bs16 step is ~0.15 ms), then yield
// between polls; after one second fall back to a stream synchronise (a very long call, or something badly wrong).
void wait_words(const Pinned& w, const c10::hip::HIPStream& stream) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool pending = false;
    for (int64_t i = 0; i < w.n && !pending; i++) pending = __atomic_load_n(w.p + i, __ATOMIC_RELAXED) == kPending;
    if (!pending) break;
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > 1.0) { stream.synchronize(); break; }
    if (waited > 2e-3) std::this_thread::yield();
    else __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
}

at::Tensor workspace(size_t bytes, const at::Device& dev) {
  // the caching allocator hands the same block back call after call; stream-ordered with the kernels that use it
  return at::empty({(int64_t)(bytes ? bytes : 1)}, at::TensorOptions().dtype(at::kByte).device(dev));
}

// The candidate counters of obb_non_max_suppression_obb_st: a small device buffer per (thread, device, stream, batch size) that
// this module keeps -- zeroed when it is made, left zeroed by every call that completes (include/obb_hip.h), zeroed again
// after one that did not.  Never freed (see pinned_words).
struct StateBuf { at::Tensor t; bool clean = false; };
StateBuf& state_buf(const at::Device& dev, const c10::hip::HIPStream& stream, int64_t bs) {
  static thread_local std::map<std::tuple<int, int64_t, int64_t>, StateBuf*> memo;
  StateBuf*& e = memo[std::make_tuple((int)dev.index(), (int64_t)stream.id(), bs)];
  if (!e) {
    e = new StateBuf();
    e->t = at::zeros({(int64_t)api.obb_nms_obb_state_bytes(bs)}, at::TensorOptions().dtype(at::kByte).device(dev));
    e->clean = true;
  }
  return *e;
}

// A view of `base`'s storage made directly (what at::as_strided does underneath, ATen/native/TensorShape.cpp: as_strided_tensorimpl)
// without a trip through the dispatcher and the autograd keys per view: the fused driver hands back one view per image, the val tail
// three -- 16 to 48 `narrow` calls were 10-30 us of host time behind kernels that had already finished.  `base` never requires grad here.
at::Tensor view_of(const at::Tensor& base, int64_t offset_elems, c10::IntArrayRef sizes, c10::IntArrayRef strides) {
  auto impl = c10::make_intrusive<c10::TensorImpl>(c10::TensorImpl::VIEW, c10::Storage(base.storage()), base.key_set(), base.dtype());
  impl->set_storage_offset(base.storage_offset() + offset_elems);
  impl->set_sizes_and_strides(sizes, strides);
  impl->set_version_counter(base.unsafeGetTensorImpl()->version_counter());
  return at::Tensor(std::move(impl));
}

void require_cuda(const at::Tensor& t, const char* name) {      // _lib.require_cuda
  if (!t.is_cuda())
    throw std::runtime_error(std::string(name) + " must be a CUDA/HIP tensor: yolov5_obb_amd is compiled for MI355X only (no CPU path, by design)");
}

std::atomic<long long> g_abort_retries{0};   // calls of this process that were repeated on the 8-workgroup grid (tests report it: abort_retries())
struct AbortRetry {            // obb_nms_set_max_grid(8) for one retry of the calling thread, restored on every path
  bool capped = false;
  void cap() { api.obb_nms_set_max_grid(8); capped = true; g_abort_retries.fetch_add(1, std::memory_order_relaxed); }
  ~AbortRetry() { if (capped) api.obb_nms_set_max_grid(0); }
};

// ---------------------------------------------------------------- nms_rotated / nms_poly (nms_rotated_ext.cpp:25-55)
at::Tensor run_single_list(int kind, const at::Tensor& dets, const at::Tensor& scores, double thr, int64_t flags, int64_t max_keep) {
  need_api();
  const int64_t n = dets.size(0);
  const at::Device dev = dets.device();
  c10::DeviceGuard guard(dev);
  const auto stream = c10::hip::getCurrentHIPStream(dev.index());
  at::Tensor keep = at::empty({n}, at::TensorOptions().dtype(at::kLong).device(dev));
  Pinned& cnt = pinned_words(dev.index(), 0, 1);
  AbortRetry retry;
  const char* what = kind == 0 ? "obb_nms_rotated_f32" : (kind == 3 ? "obb_nms_rotated_f64" : "obb_nms_poly_f32");
  for (;;) {
    const size_t bytes = api.obb_nms_workspace_bytes(n, 1, kind);      // (depends on the calling thread's grid cap)
    at::Tensor ws = workspace(bytes, dev);
    arm(cnt);
    int rc;
    if (kind == 0)
      rc = api.obb_nms_rotated_f32(dets.data_ptr<float>(), scores.data_ptr<float>(), n, (float)thr, (int)flags, max_keep, keep.data_ptr<int64_t>(),
                                   cnt.p, ws.data_ptr(), (size_t)ws.numel(), stream.stream());
    else if (kind == 3)
      rc = api.obb_nms_rotated_f64(dets.data_ptr<double>(), scores.data_ptr<double>(), n, (float)thr, (int)flags, max_keep, keep.data_ptr<int64_t>(),
                                   cnt.p, ws.data_ptr(), (size_t)ws.numel(), stream.stream());
    else
      rc = api.obb_nms_poly_f32(dets.data_ptr<float>(), dets.size(1), n, (float)thr, max_keep, keep.data_ptr<int64_t>(), cnt.p, ws.data_ptr(),
                                (size_t)ws.numel(), stream.stream());
    check(rc, what);
    {
      py::gil_scoped_release nogil;
      wait_words(cnt, stream);
    }
    const int64_t k = cnt.p[0];
    if (k < 0) {                                                       // a team barrier of the persistent kernel timed out
      if (retry.capped)
        throw std::runtime_error(std::string(what) + ": the NMS kernel aborted (a workgroup barrier timed out); results are invalid");
      retry.cap();
      continue;
    }
    return keep.narrow(0, 0, k);
  }
}

at::Tensor nms_rotated_opts(const at::Tensor& dets, const at::Tensor& scores, double iou_threshold, int64_t flags, int64_t max_keep) {
  require_cuda(dets, "dets");
  require_cuda(scores, "scores");
  if (dets.device() != scores.device()) throw std::runtime_error("dets and scores must be on the same device");      // :29
  if (dets.scalar_type() != scores.scalar_type()) throw std::runtime_error("dets should have the same type as scores");   // nms_rotated_cpu.cpp:19-21
  if (dets.scalar_type() != at::kFloat && dets.scalar_type() != at::kDouble)
    throw std::runtime_error(std::string("nms_rotated: float32 or float64 expected, got ") + c10::toString(dets.scalar_type()));   // nms_rotated_cuda.cu:96
  if (dets.dim() != 2 || dets.size(1) != 5 || scores.dim() != 1 || scores.size(0) != dets.size(0))
    throw std::runtime_error("nms_rotated: expected dets (N,5) and scores (N)");
  if (dets.numel() == 0) return at::empty({0}, at::TensorOptions().dtype(at::kLong).device(dets.device()));
  return run_single_list(dets.scalar_type() == at::kDouble ? 3 : 0, dets.contiguous(), scores.contiguous(), iou_threshold, flags, max_keep);
}

at::Tensor nms_rotated(const at::Tensor& dets, const at::Tensor& scores, double iou_threshold) {
  return nms_rotated_opts(dets, scores, iou_threshold, 0, 0);
}

at::Tensor nms_poly(const at::Tensor& dets, double iou_threshold) {
  if (!dets.is_cuda()) throw std::runtime_error("POLY_NMS is not implemented on CPU");                 // AT_ERROR, :54
  if (dets.numel() == 0) return at::empty({0}, at::TensorOptions().dtype(at::kLong).device(at::kCPU));  // :47-48
  if (dets.scalar_type() != at::kFloat)
    throw std::runtime_error(std::string("nms_poly: float32 expected (the reference wrapper casts with .float()), got ") + c10::toString(dets.scalar_type()));
  if (dets.dim() != 2 || dets.size(1) < 9) throw std::runtime_error("nms_poly: expected dets (N,9)");
  return run_single_list(1, dets.contiguous(), dets, iou_threshold, 0, 0);
}

// ---------------------------------------------------------------- non_max_suppression_obb (utils/general.py:772-862)
constexpr int64_t kMaxWh = 4096, kMaxNms = 30000, kCsl = 180;          // utils/general.py:793-794, 784
constexpr int64_t kHold = 8;                                           // calls a regime is held after a repeated call (hysteresis)

struct ShapeKey {
  int dev; int64_t A, nc; bool multi; uint32_t conf_bits;
  bool operator<(const ShapeKey& o) const {
    return std::tie(dev, A, nc, multi, conf_bits) < std::tie(o.dev, o.A, o.nc, o.multi, o.conf_bits);
  }
};
struct ShapeMemo {
  int64_t cap = 0;                             // candidate slots per image that sufficed last time
  int64_t cand = OBB_NMS_SORT_LDS_HINT;        // largest candidate count of an image in the previous call: the sort hint.  No history:
                                               // the regime of the reference's default thresholds (a few thousand candidates per image)
  int64_t seg = 1;                             // largest NMS segment of the previous call: chooses the NMS kernel (no history: small)
  bool small_resolved = false;                 // ... and (informational) such an image kept its class segments in that call
  bool small_boxes = false;                    // the previous call met boxes with a sub-pixel side: the next one runs the cross-class check
  int hold_seg = 0, hold_cand = 0;             // > 0: a call was repeated because its hint undersold it -- keep the larger regime this
                                               // many calls unless the batch falls clearly (25 %) below the limit: a stream whose
                                               // batches hover around a limit does not pay the repeat on every other batch
};

std::map<ShapeKey, ShapeMemo>& shape_memos() {            // per calling thread: other threads' batches say nothing about this one's
  static thread_local std::map<ShapeKey, ShapeMemo> memos;
  return memos;
}
uint32_t conf_key(double conf_thres) {
  const float f = (float)conf_thres;
  uint32_t b;
  std::memcpy(&b, &f, 4);
  return b;
}
// Introspection of the hint memo (tests, tools): the state of a shape, forcing a hint, forgetting everything.
void hints_clear() { shape_memos().clear(); }
py::object hint_get(int dev, int64_t A, int64_t nc, bool multi, double conf_thres) {
  auto& m = shape_memos();
  auto it = m.find(ShapeKey{dev, A, nc, multi, conf_key(conf_thres)});
  if (it == m.end()) return py::none();
  py::dict d;
  d["cap"] = it->second.cap; d["cand"] = it->second.cand; d["seg"] = it->second.seg;
  d["hold_cand"] = it->second.hold_cand; d["hold_seg"] = it->second.hold_seg; d["small_boxes"] = it->second.small_boxes; d["small_resolved"] = it->second.small_resolved;
  return std::move(d);
}
void hint_set(int dev, int64_t A, int64_t nc, bool multi, double conf_thres, int64_t cand, int64_t seg) {
  ShapeMemo& e = shape_memos()[ShapeKey{dev, A, nc, multi, conf_key(conf_thres)}];
  if (cand >= 0) { e.cand = cand; e.hold_cand = 0; }
  if (seg >= 0) { e.seg = seg; e.hold_seg = 0; }
}

std::vector<at::Tensor> non_max_suppression_obb(const at::Tensor& prediction, double conf_thres, double iou_thres,
                                                const c10::optional<std::vector<int64_t>>& classes, bool agnostic, bool multi_label,
                                                const c10::optional<at::Tensor>& extra, int64_t max_det,
                                                const c10::optional<at::Tensor>& objcol) {
  need_api();
  require_cuda(prediction, "prediction");
  if (prediction.dim() != 3) throw std::runtime_error("prediction must be (bs, anchors, no)");
  const int64_t nc = prediction.size(2) - 5 - kCsl;
  if (nc < 1 || nc > 256) throw std::runtime_error("non_max_suppression_obb: 1 <= nc <= 256 supported, got nc = " + std::to_string(nc));
  int dtype;
  if (prediction.scalar_type() == at::kFloat) dtype = 0;
  else if (prediction.scalar_type() == at::kHalf) dtype = 1;
  else throw std::runtime_error(std::string("non_max_suppression_obb: float32 or float16 expected, got ") + c10::toString(prediction.scalar_type()));
  const at::Tensor pred = prediction.contiguous();
  const int64_t bs = pred.size(0), A = pred.size(1), no = pred.size(2);
  const at::Device dev = pred.device();
  const bool multi = multi_label && nc > 1;
  std::vector<at::Tensor> result;
  if (bs == 0) return result;
  if (A == 0) {
    at::Tensor z = at::zeros({0, 7}, at::TensorOptions().dtype(at::kFloat).device(dev));
    result.assign((size_t)bs, z);
    return result;
  }
  std::vector<int32_t> cls;
  if (classes.has_value()) {
    if (classes->empty()) {
      at::Tensor z = at::zeros({0, 7}, at::TensorOptions().dtype(at::kFloat).device(dev));
      result.assign((size_t)bs, z);
      return result;
    }
    for (int64_t c : *classes) cls.push_back((int32_t)c);
  }
  const float* extra_p = nullptr;
  int64_t n_extra = 0;
  at::Tensor extra_c;
  if (extra.has_value() && extra->defined() && extra->numel() > 0) {
    extra_c = extra->to(dev, at::kFloat).contiguous();
    if (extra_c.dim() != 2 || extra_c.size(1) != 8) throw std::runtime_error("non_max_suppression_obb: label rows must be (n, 8)");
    extra_p = extra_c.data_ptr<float>();
    n_extra = extra_c.size(0);
  }
  const void* col_p = nullptr;
  if (objcol.has_value() && objcol->defined()) {
    const at::Tensor& c = *objcol;
    if (c.dim() == 2 && c.size(0) == bs && c.size(1) == A && c.scalar_type() == pred.scalar_type() && c.device() == dev && c.is_contiguous() &&
        pred.data_ptr() == prediction.data_ptr())
      col_p = c.data_ptr();
  }

  static thread_local std::map<std::tuple<int64_t, int64_t, int64_t, int, bool>, size_t> ws_memo;
  const float conf_f = (float)conf_thres;
  ShapeMemo& memo = shape_memos()[ShapeKey{(int)dev.index(), A, nc, multi, conf_key(conf_thres)}];
  const int64_t worst = A * (multi ? nc : 1) + n_extra;
  int64_t cap = std::min(worst, std::max<int64_t>(memo.cap, 65536));

  c10::DeviceGuard guard(dev);
  const auto stream = c10::hip::getCurrentHIPStream(dev.index());
  at::Tensor out = at::empty({bs * max_det, 7}, at::TensorOptions().dtype(at::kFloat).device(dev));   // image b's rows start at b * max_det (out_packed = 0:
                                                                                                       // the NMS kernel writes them itself, one launch less)
  Pinned& meta = pinned_words(dev.index(), 1, bs + 2);                                                 // counts[bs] + status[2]
  AbortRetry retry;
  int64_t seg_max = 0, cand_max = 0;
  for (;;) {
    const int64_t hint = memo.cand & 0xffffffffll, seg_hint = memo.seg & 0x1fffffffll;
    const auto wkey = std::make_tuple(bs, cap, nc, (int)agnostic, retry.capped);
    auto it = ws_memo.find(wkey);
    if (it == ws_memo.end()) it = ws_memo.emplace(wkey, api.obb_nms_obb_workspace_bytes(bs, cap, nc, agnostic ? 1 : 0)).first;
    at::Tensor ws = workspace(it->second, dev);
    StateBuf& sb = state_buf(dev, stream, bs);
    if (!sb.clean) sb.t.zero_();                                       // the previous call on it did not complete
    sb.clean = false;
    arm(meta);
    const int rc = api.obb_non_max_suppression_obb_st(pred.data_ptr(), col_p, dtype, bs, A, no, conf_f, (float)iou_thres, cls.empty() ? nullptr : cls.data(),
                                                       (int)cls.size(), agnostic ? 1 : 0, multi ? 1 : 0, max_det, kMaxNms, (float)kMaxWh, extra_p, n_extra, cap,
                                                       hint | (seg_hint << 32) | (memo.small_boxes ? (int64_t(1) << 62) : 0), out.data_ptr<float>(), 0, meta.p, meta.p + bs,
                                                       ws.data_ptr(), (size_t)ws.numel(), sb.t.data_ptr(), (size_t)sb.t.numel(),
                                                       stream.stream());
    check(rc, "obb_non_max_suppression_obb");
    {
      py::gil_scoped_release nogil;
      wait_words(meta, stream);
    }
    sb.clean = true;                                                   // every launch of the call was made: its last kernel zeroes the counters
    const int64_t st0 = meta.p[bs], st1 = meta.p[bs + 1];
    seg_max = (st1 >> 32) & 0x1fffffffll;
    memo.small_resolved = ((st1 >> 61) & 1) != 0;
    cand_max = st1 & 0xffffffffll;
    memo.small_boxes = ((st1 >> 62) & 1) != 0;
    if (st0 == -1) {                                                   // a segment above the small kernel's limit: nothing is valid
      memo.seg = std::max<int64_t>(seg_max, OBB_NMS_SMALL_SEG + 1);
      memo.hold_seg = kHold;
      continue;
    }
    int64_t mn = 0;
    for (int64_t b = 0; b < bs; b++) mn = std::min(mn, meta.p[b]);
    if (mn < 0) {                                                      // a team barrier of the NMS kernel timed out
      if (retry.capped)
        throw std::runtime_error("obb_non_max_suppression_obb: the NMS kernel aborted (a workgroup barrier timed out); results are invalid");
      retry.cap();                                                     // once more with a grid that is resident under any CU mask
      continue;
    }
    if (st0 > cap) {                                                   // an image produced more candidates than slots
      cap = std::min(worst, std::max<int64_t>(st0, 2 * cap));
      continue;
    }
    if (hint > 0 && hint <= OBB_NMS_SORT_LDS_HINT && cand_max > OBB_NMS_SORT_LDS_MAX) {   // the hint undersold this batch: such images were left out
      memo.cand = cand_max;
      memo.hold_cand = kHold;
      continue;
    }
    break;
  }
  memo.cap = cap;
  // hints of the next call, with hysteresis after a repeated call
  if (memo.hold_cand > 0 && cand_max > OBB_NMS_SORT_LDS_HINT * 3 / 4) { memo.hold_cand--; memo.cand = std::max<int64_t>(cand_max, OBB_NMS_SORT_LDS_HINT + 1); }
  else { memo.hold_cand = 0; memo.cand = cand_max; }
  if (memo.hold_seg > 0 && (seg_max == 0 || seg_max > OBB_NMS_SMALL_SEG * 3 / 4)) { memo.hold_seg--; memo.seg = std::max<int64_t>(seg_max, OBB_NMS_SMALL_SEG + 1); }
  else { memo.hold_seg = 0; memo.seg = seg_max; }                      // (0: the sort path of this call does not report it)
  result.reserve((size_t)bs);
  for (int64_t b = 0; b < bs; b++) result.push_back(view_of(out, b * max_det * 7, {meta.p[b], 7}, {7, 1}));
  return result;
}

// ---------------------------------------------------------------- the post-NMS tail of val.py for a batch (val.py:209-250)
constexpr int64_t kTailMaxBs = 64;       // csrc/head.hip kValTailMaxBs: images per obb_val_tail_batch_f32 call

// shapes[j] = ((h, w), ((gain, gain), (pad_x, pad_y))) as LoadImagesAndLabels yields them -> {pad_x, pad_y, gain, w, h}
void img5_of(const py::handle& shape_j, float* o) {
  const py::sequence s = py::reinterpret_borrow<py::sequence>(shape_j);
  const py::sequence hw = s[0].cast<py::sequence>();
  const py::sequence rp = s[1].cast<py::sequence>();
  const py::sequence ratio = rp[0].cast<py::sequence>();
  const py::sequence pad = rp[1].cast<py::sequence>();
  o[0] = pad[0].cast<float>(); o[1] = pad[1].cast<float>(); o[2] = ratio[0].cast<float>();
  o[3] = hw[1].cast<float>(); o[4] = hw[0].cast<float>();
}

struct HostRows { at::Tensor t; int64_t rows = 0, cols = 0; };

py::object val_tail_batch(const std::vector<at::Tensor>& preds, const at::Tensor& targets, const py::sequence& shapes, const at::Tensor& iouv,
                          bool want_boxes) {
  need_api();
  const int64_t bs = (int64_t)preds.size();
  if (bs == 0) return want_boxes ? py::object(py::make_tuple(py::list(), py::none())) : py::object(py::list());
  const at::Device dev = preds[0].device();
  require_cuda(preds[0], "pred");
  std::vector<int64_t> offs((size_t)bs + 1, 0);
  for (int64_t b = 0; b < bs; b++) offs[b + 1] = offs[b] + preds[b].size(0);
  const int64_t n = offs[bs];
  const int64_t niou = iouv.size(0);
  // the detections where they lie: views of ONE buffer of 7-float rows (what non_max_suppression_obb returns: image b at row
  // b * max_det; a packed list as well) are addressed by their first rows; anything else is concatenated
  at::Tensor base;                                                     // the tensor whose data pointer the rows count from
  std::vector<int64_t> rows((size_t)bs, 0);
  if (n) {
    bool ok = true;
    const char* p0 = nullptr;
    for (int64_t b = 0; b < bs && ok; b++) {
      const at::Tensor& p = preds[(size_t)b];
      if (p.size(0) == 0) continue;
      if (p.device() != dev || p.scalar_type() != at::kFloat || p.dim() != 2 || p.size(1) != 7 || !p.is_contiguous()) { require_cuda(p, "pred"); ok = false; break; }
      const char* q = (const char*)p.data_ptr();
      if (!p0) { p0 = q; base = p; }
      const int64_t diff = q - p0;
      if (diff < 0 || diff % 28 != 0 || diff / 28 > 0x7fffffff - p.size(0) || !p.is_alias_of(base)) { ok = false; break; }
      rows[(size_t)b] = diff / 28;
    }
    if (!ok) {
      std::vector<at::Tensor> f;
      for (const at::Tensor& p : preds) f.push_back(p.to(dev, at::kFloat));
      base = at::cat(f, 0).contiguous();
      for (int64_t b = 0; b < bs; b++) rows[(size_t)b] = offs[(size_t)b];
    }
  }
  at::Tensor tg;
  int64_t nt = 0, tcols = 0;
  if (targets.dim() == 2 && targets.size(0) > 0) {
    tg = targets.to(dev, at::kFloat).contiguous();
    nt = tg.size(0); tcols = tg.size(1);
  }
  const at::Tensor iv = iouv.to(dev, at::kFloat).contiguous();
  static thread_local std::map<int64_t, HostRows*> pin_rows;           // per thread and row width; grown geometrically; never freed (see pinned_words)
  HostRows*& hrp = pin_rows[niou + 2];
  if (!hrp) hrp = new HostRows();
  HostRows& hr = *hrp;
  if (!hr.t.defined() || hr.rows < n) {
    hr.rows = std::max<int64_t>(1024, 2 * n); hr.cols = niou + 2;
    hr.t = at::empty({hr.rows, hr.cols}, at::TensorOptions().dtype(at::kFloat).pinned_memory(true));
  }
  float* host = hr.t.data_ptr<float>();
  at::Tensor boxes[4];
  if (want_boxes) {
    const int64_t w[4] = {10, 6, 10, 6};
    for (int k = 0; k < 4; k++) boxes[k] = at::empty({n, w[k]}, at::TensorOptions().dtype(at::kFloat).device(dev));
  }
  if (n) {
    c10::DeviceGuard guard(dev);
    const auto stream = c10::hip::getCurrentHIPStream(dev.index());
    at::Tensor ws = workspace(api.obb_val_tail_batch_workspace_bytes(n, nt), dev);
    Pinned& flag = pinned_words(dev.index(), 2, 1);
    std::vector<int64_t> doff, drow;
    std::vector<float> img5;
    for (int64_t b0 = 0; b0 < bs; b0 += kTailMaxBs) {                  // (one call for any batch size val.py uses)
      const int64_t b1 = std::min(bs, b0 + kTailMaxBs), k = b1 - b0;
      const int64_t lo = offs[b0], hi = offs[b1];
      if (hi == lo) continue;
      doff.assign((size_t)k + 1, 0);
      for (int64_t j = 0; j <= k; j++) doff[j] = offs[b0 + j] - lo;
      drow.assign(rows.begin() + b0, rows.begin() + b1);
      img5.assign((size_t)(5 * k), 0.f);
      for (int64_t j = 0; j < k; j++) img5_of(shapes[(size_t)(b0 + j)], img5.data() + 5 * j);
      at::Tensor tgk = tg;
      int64_t ntk = nt;
      if (nt && (b0 || b1 < bs)) {                                     // a chunk of a very large batch: its labels, re-based
        const at::Tensor img = tg.select(1, 0);
        tgk = tg.index({(img >= (double)b0) & (img < (double)b1)}).clone();
        tgk.select(1, 0).sub_((double)b0);
        ntk = tgk.size(0);
      }
      arm(flag);
      const int rc = api.obb_val_tail_batch_rows_f32(
          base.data_ptr<float>(), drow.data(), doff.data(), k, ntk ? tgk.data_ptr<float>() : nullptr, ntk, tcols, img5.data(), iv.data_ptr<float>(), (int)niou,
          want_boxes ? boxes[0].data_ptr<float>() + lo * 10 : nullptr, want_boxes ? boxes[1].data_ptr<float>() + lo * 6 : nullptr,
          want_boxes ? boxes[2].data_ptr<float>() + lo * 10 : nullptr, want_boxes ? boxes[3].data_ptr<float>() + lo * 6 : nullptr,
          host + lo * (niou + 2), ws.data_ptr(), (size_t)ws.numel(), stream.stream(), flag.p);
      check(rc, "obb_val_tail_batch_f32");
      {
        py::gil_scoped_release nogil;
        wait_words(flag, stream);                                      // every row of this chunk is in `host`
      }
    }
  }
  // what val.py:250 appends, per image: (correct bool (n_i, niou), conf (n_i), cls (n_i)) on the host.  Two arrays for the
  // whole batch (the pinned buffer is reused by the next call), per-image views of them
  // (one pass over the pinned rows: the bool matrix and the (conf, cls) pairs; no torch CPU op -- a parallel one would wake the
  //  whole intra-op pool, see profiles/r5_host_stall.md)
  at::Tensor correct = at::empty({n, niou}, at::TensorOptions().dtype(at::kBool));
  at::Tensor cc = at::empty({n, 2}, at::TensorOptions().dtype(at::kFloat));
  {
    const float* r = host;
    bool* c = correct.data_ptr<bool>();
    float* q = cc.data_ptr<float>();
    for (int64_t i = 0; i < n; i++, r += niou + 2, c += niou, q += 2) {
      for (int64_t j = 0; j < niou; j++) c[j] = r[j] > 0.5f;
      q[0] = r[niou]; q[1] = r[niou + 1];
    }
  }
  py::list out((size_t)bs);
  for (int64_t b = 0; b < bs; b++) {
    const int64_t c = offs[b + 1] - offs[b];
    out[(size_t)b] = py::make_tuple(view_of(correct, offs[b] * niou, {c, niou}, {niou, 1}), view_of(cc, offs[b] * 2, {c}, {2}), view_of(cc, offs[b] * 2 + 1, {c}, {2}));
  }
  if (!want_boxes) return std::move(out);
  return py::make_tuple(out, py::make_tuple(py::make_tuple(boxes[0], boxes[1], boxes[2], boxes[3]), offs));
}

// ---------------------------------------------------------------- module
void init(const std::string& path) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (api.handle && api.path == path) return;
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) throw std::runtime_error(std::string("nms_rotated_ext: cannot load ") + path + ": " + dlerror());
#define X(name)                                                                                          \
  {                                                                                                      \
    void* s = dlsym(h, #name);                                                                           \
    if (!s) throw std::runtime_error(std::string("nms_rotated_ext: ") + path + " does not export " #name); \
    api.name = reinterpret_cast<decltype(api.name)>(s);                                                  \
  }
  OBB_API(X)
#undef X
  api.handle = h;
  api.path = path;
}

std::string library() { return api.path; }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled binding of libobb_hip.so (nms_rotated, nms_poly: utils/nms_rotated/src/nms_rotated_ext.cpp:57-60; "
            "non_max_suppression_obb: utils/general.py:772-862; val_tail_batch: val.py:209-250)";
  m.def("init", &init, "bind the C ABI of the given libobb_hip.so (dlopen / dlsym)");
  m.def("library", &library);
  m.def("_view_of", [](const at::Tensor& base, int64_t offset, std::vector<int64_t> sizes, std::vector<int64_t> strides) { return view_of(base, offset, sizes, strides); },
        "the binding's direct view constructor (tests/test_cabi.py compares it with torch.as_strided)");
  m.def("nms_rotated", &nms_rotated, "NMS for rotated boxes", py::arg("dets"), py::arg("scores"), py::arg("iou_threshold"));
  m.def("nms_poly", &nms_poly, "NMS for quadrilaterals", py::arg("dets"), py::arg("iou_threshold"));
  m.def("nms_rotated_opts", &nms_rotated_opts, py::arg("dets"), py::arg("scores"), py::arg("iou_threshold"), py::arg("flags") = 0, py::arg("max_keep") = 0);
  m.def("non_max_suppression_obb", &non_max_suppression_obb, py::arg("prediction"), py::arg("conf_thres") = 0.25, py::arg("iou_thres") = 0.45,
        py::arg("classes") = py::none(), py::arg("agnostic") = false, py::arg("multi_label") = false, py::arg("extra") = py::none(),
        py::arg("max_det") = 1500, py::arg("objcol") = py::none());
  m.def("abort_retries", []() { return (long long)g_abort_retries.load(std::memory_order_relaxed); },
        "calls of this process that a barrier time-out of the persistent NMS kernel sent to the 8-workgroup grid");
  m.def("hints_clear", &hints_clear, "forget the hint memo of the calling thread");
  m.def("hint_get", &hint_get, py::arg("device_index"), py::arg("A"), py::arg("nc"), py::arg("multi_label"), py::arg("conf_thres"));
  m.def("hint_set", &hint_set, py::arg("device_index"), py::arg("A"), py::arg("nc"), py::arg("multi_label"), py::arg("conf_thres"),
        py::arg("cand") = -1, py::arg("seg") = -1);
  m.def("val_tail_batch", &val_tail_batch, py::arg("preds"), py::arg("targets"), py::arg("shapes"), py::arg("iouv"), py::arg("want_boxes") = false);
}
