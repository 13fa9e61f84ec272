// Rotated / quad NMS entry points (C ABI) -- host driver for nms_core.h.
//
// Replaces, on device tensors:
//   nms_rotated_ext.nms_rotated  (utils/nms_rotated/src/nms_rotated_ext.cpp:25-39 ->
//                                 nms_rotated_cuda, nms_rotated_cuda.cu:71-134)
//   nms_rotated_ext.nms_poly     (nms_rotated_ext.cpp:42-55 -> poly_nms_cuda, poly_nms_cuda.cu:197-261)
//   _poly_nms of the devkit      (DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu:277-329)
// Everything is stream-ordered on the caller's stream; nothing is copied to the
// host; the caller reads *num_keep when it needs the count.
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "nms_core.h"
#include "obb_hip.h"

namespace obb {

// ---------------------------------------------------------------- small kernels
// Sort key: segments ascending, score descending, NaN first (torch's order),
// -0 == +0, ties keep ascending original index (LSD radix sort is stable) or the
// caller's explicit tie word.  Boxes flagged invalid sort last in their segment.
__device__ __forceinline__ uint32_t score_desc_key(float s) {
  uint32_t u = __float_as_uint(s);
  uint32_t k;
  if (s != s) k = 0xFFFFFFFFu;
  else {
    if (s == 0.f) u = 0u;
    k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  return ~k;
}

__global__ void k_make_keys(const float* __restrict__ scores, int score_stride, const int32_t* __restrict__ seg_id,
                            const uint32_t* __restrict__ tie, int tie_bits, const float* __restrict__ dets5,
                            int drop_small, int n, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k = score_desc_key(scores[(size_t)i * score_stride]);
  if (drop_small) {
    // nms_rotated_wrapper.py:32  too_small = dets[:, [2, 3]].min(1)[0] < 0.001
    float w = dets5[(size_t)i * 5 + 2], h = dets5[(size_t)i * 5 + 3];
    float mn = (h < w) ? h : w;           // torch.min propagates NaN; NaN < 0.001 is False either way
    if (mn < 0.001f) k = 0xFFFFFFFFu;
  }
  uint64_t seg = seg_id ? (uint64_t)(uint32_t)seg_id[i] : 0ull;
  uint64_t key = (seg << (32 + tie_bits)) | ((uint64_t)k << tie_bits);
  if (tie_bits) key |= (uint64_t)(tie[i] & ((1u << tie_bits) - 1u));
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

__global__ void k_seg_bounds(const uint64_t* __restrict__ keys, int n, int nseg, int shift, int* __restrict__ seg_begin,
                             int* keep_cnt, int* nrows, int* nedges) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > nseg) return;
  if (g == nseg) { seg_begin[g] = n; return; }
  // lower_bound of (g << shift)
  uint64_t target = (uint64_t)g << shift;
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  seg_begin[g] = lo;
  keep_cnt[g] = 0; nrows[g] = 0; nedges[g] = 0;
}

__global__ void k_prep_rot(const float* __restrict__ dets5, const uint32_t* __restrict__ order, int drop_small, int n,
                           float* __restrict__ feat, uint8_t* __restrict__ dead) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float* d = dets5 + (size_t)order[p] * 5;
  float x = d[0], y = d[1], w = d[2], h = d[3], a = d[4];
  RBoxFeat f = rbox_make_feat(x, y, w, h, a);
  feat[(size_t)0 * n + p] = f.x;  feat[(size_t)1 * n + p] = f.y;  feat[(size_t)2 * n + p] = f.w;   feat[(size_t)3 * n + p] = f.h;
  feat[(size_t)4 * n + p] = f.sh; feat[(size_t)5 * n + p] = f.cw; feat[(size_t)6 * n + p] = f.ch;  feat[(size_t)7 * n + p] = f.sw;
  feat[(size_t)8 * n + p] = f.r;  feat[(size_t)9 * n + p] = f.c;  feat[(size_t)10 * n + p] = f.s;  feat[(size_t)11 * n + p] = f.area;
  float mn = (h < w) ? h : w;
  dead[p] = (drop_small && mn < 0.001f) ? 1 : 0;
}

__global__ void k_prep_quad(const float* __restrict__ polys, int stride, const uint32_t* __restrict__ order, int n,
                            float* __restrict__ feat, uint8_t* __restrict__ dead) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float* d = polys + (size_t)order[p] * stride;
#pragma unroll
  for (int k = 0; k < 8; k++) feat[(size_t)k * n + p] = d[k];
  dead[p] = 0;
}

__global__ void k_finalize(const int* __restrict__ keep_cnt, const int* __restrict__ seg_begin, int nseg, long long max_keep,
                           int64_t* __restrict__ num_keep, int64_t* __restrict__ seg_begin_out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > nseg) return;
  if (seg_begin_out) seg_begin_out[g] = seg_begin[g];
  if (g == nseg) return;
  long long c = keep_cnt[g];
  if (max_keep > 0 && c > max_keep) c = max_keep;
  num_keep[g] = c;
}

// ---------------------------------------------------------------- workspace
static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

static int chunk_size() {
  static int c = 0;
  if (!c) {
    const char* e = getenv("OBB_NMS_CHUNK");
    int v = e ? atoi(e) : 2048;
    if (v < 64) v = 64;
    if (v > 4096) v = 4096;
    v = (v + 63) / 64 * 64;
    c = v;
  }
  return c;
}

struct Carve {
  uint64_t *keys_a, *keys_b;
  uint32_t *vals_a, *vals_b;
  void* sort_tmp; size_t sort_tmp_bytes;
  float* feat; uint8_t* dead;
  int *seg_begin, *keep_cnt, *nrows, *nedges;
  uint32_t *rows, *edges;
  long long ecap;
  size_t total;
};

static hipError_t sort_tmp_query(size_t n, size_t* bytes) {
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                   (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0, false);
}

static int carve(void* base, int64_t n, int64_t nseg, int nf, int C, Carve* cv) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return base ? (char*)base + o : (char*)nullptr; };
  size_t nn = (size_t)(n > 0 ? n : 1), ns = (size_t)(nseg > 0 ? nseg : 1);
  cv->keys_a = (uint64_t*)take(nn * 8); cv->keys_b = (uint64_t*)take(nn * 8);
  cv->vals_a = (uint32_t*)take(nn * 4); cv->vals_b = (uint32_t*)take(nn * 4);
  if (sort_tmp_query(nn, &cv->sort_tmp_bytes) != hipSuccess) return OBB_ERR_INTERNAL;
  cv->sort_tmp = take(cv->sort_tmp_bytes ? cv->sort_tmp_bytes : 16);
  cv->feat = (float*)take(nn * nf * 4);
  cv->dead = (uint8_t*)take(nn);
  cv->seg_begin = (int*)take((ns + 1) * 4); cv->keep_cnt = (int*)take(ns * 4);
  cv->nrows = (int*)take(ns * 4); cv->nedges = (int*)take(ns * 4);
  cv->rows = (uint32_t*)take(ns * (size_t)C * 4);
  cv->ecap = (long long)C * (C - 1) / 2; if (cv->ecap < 1) cv->ecap = 1;
  cv->edges = (uint32_t*)take(ns * (size_t)cv->ecap * 4);
  cv->total = off;
  return OBB_OK;
}

// kind: 0 rotated (5 floats + score array), 1 quad (rows of `stride` floats, score in column 8)
static int run_nms(int kind, const float* boxes, int stride, const float* scores, int score_stride, const int32_t* seg_id,
                   const uint32_t* tie, int tie_bits, int64_t n, int64_t nseg, int64_t max_seg, float thr, int flags,
                   int64_t max_keep, int64_t* keep_out, int64_t* num_keep, int64_t* seg_begin_out, void* ws, size_t ws_bytes,
                   hipStream_t st) {
  if (n < 0 || nseg < 1 || n > 0x7fffffffLL) return OBB_ERR_BAD_ARG;
  if (!num_keep || (n > 0 && (!boxes || !scores || !keep_out))) return OBB_ERR_BAD_ARG;
  const int C = chunk_size();
  const int nf = kind == 0 ? RotGeom::NF : QuadGeom::NF;
  Carve cv;
  int rc = carve(ws, n, nseg, nf, C, &cv);
  if (rc) return rc;
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  int seg_bits = 0;
  while ((1ll << seg_bits) < nseg) seg_bits++;
  if (32 + tie_bits + seg_bits > 64) return OBB_ERR_BAD_ARG;
  const int T = 256;
  const int drop_small = (flags & OBB_NMS_DROP_SMALL) ? 1 : 0;

  if (n == 0) {
    hipMemsetAsync(cv.keep_cnt, 0, nseg * 4, st);
    hipMemsetAsync(cv.seg_begin, 0, (nseg + 1) * 4, st);
    k_finalize<<<(unsigned)((nseg + 1 + T - 1) / T), T, 0, st>>>(cv.keep_cnt, cv.seg_begin, (int)nseg, max_keep, num_keep, seg_begin_out);
    return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
  }

  const unsigned gb = (unsigned)((n + T - 1) / T);
  k_make_keys<<<gb, T, 0, st>>>(scores, score_stride, seg_id, tie, tie_bits, kind == 0 ? boxes : nullptr,
                                kind == 0 ? drop_small : 0, (int)n, cv.keys_a, cv.vals_a);
  size_t tmp = cv.sort_tmp_bytes;
  if (rocprim::radix_sort_pairs(cv.sort_tmp, tmp, cv.keys_a, cv.keys_b, cv.vals_a, cv.vals_b, (size_t)n, 0,
                                (unsigned)(32 + tie_bits + seg_bits), st, false) != hipSuccess)
    return OBB_ERR_LAUNCH;
  k_seg_bounds<<<(unsigned)((nseg + 1 + T - 1) / T), T, 0, st>>>(cv.keys_b, (int)n, (int)nseg, 32 + tie_bits, cv.seg_begin,
                                                                cv.keep_cnt, cv.nrows, cv.nedges);
  if (kind == 0) k_prep_rot<<<gb, T, 0, st>>>(boxes, cv.vals_b, drop_small, (int)n, cv.feat, cv.dead);
  else k_prep_quad<<<gb, T, 0, st>>>(boxes, stride, cv.vals_b, (int)n, cv.feat, cv.dead);

  NmsArgs a;
  a.feat = cv.feat; a.order = cv.vals_b; a.dead = cv.dead; a.seg_begin = cv.seg_begin; a.keep_cnt = cv.keep_cnt;
  a.keep_out = keep_out; a.rows = cv.rows; a.nrows = cv.nrows; a.edges = cv.edges; a.nedges = cv.nedges;
  a.ecap = cv.ecap; a.n = (int)n; a.C = C;
  a.max_keep = (int)(max_keep > 0x7fffffffLL ? 0x7fffffffLL : (max_keep < 0 ? 0 : max_keep));
  a.thr = thr;
  a.cull = (thr >= 0.f) ? 1 : 0;      // rejects predict IoU <= 0 or IoU <= thr; with thr < 0 even IoU == 0 suppresses

  if (max_seg <= 0 || max_seg > n) max_seg = n;
  const int nbmax = C / 64;
  const int64_t nsteps = (max_seg + C - 1) / C;
  for (int64_t s = 0; s < nsteps; s++) {
    dim3 ga((unsigned)(nbmax * nbmax), (unsigned)nseg);
    if (kind == 0) k_chunk_pairs<RotGeom><<<ga, 64, 0, st>>>(a, (int)s);
    else k_chunk_pairs<QuadGeom><<<ga, 64, 0, st>>>(a, (int)s);
    k_chunk_resolve<<<(unsigned)nseg, 1024, (size_t)2 * C, st>>>(a, (int)s);
    int64_t rest = max_seg - (s + 1) * C;
    if (rest > 0) {
      dim3 gc((unsigned)((rest + 63) / 64), (unsigned)nseg);
      if (kind == 0) k_cross<RotGeom><<<gc, 64, 0, st>>>(a, (int)s);
      else k_cross<QuadGeom><<<gc, 64, 0, st>>>(a, (int)s);
    }
  }
  k_finalize<<<(unsigned)((nseg + 1 + T - 1) / T), T, 0, st>>>(cv.keep_cnt, cv.seg_begin, (int)nseg, max_keep, num_keep, seg_begin_out);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // namespace obb

using namespace obb;

extern "C" {

size_t obb_nms_workspace_bytes(int64_t n, int64_t nseg, int kind) {
  Carve cv;
  if (n < 0 || nseg < 1) return 0;
  if (carve(nullptr, n, nseg, kind == 0 ? RotGeom::NF : QuadGeom::NF, chunk_size(), &cv)) return 0;
  return cv.total;
}

int obb_nms_rotated_f32(const float* dets5, const float* scores, int64_t n, float iou_thr, int flags, int64_t max_keep,
                        int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  return run_nms(0, dets5, 5, scores, 1, nullptr, nullptr, 0, n, 1, n, iou_thr, flags, max_keep, keep_out, num_keep, nullptr,
                 ws, ws_bytes, (hipStream_t)stream);
}

int obb_nms_rotated_batched_f32(const float* dets5, const float* scores, const int32_t* seg_id, const uint32_t* tie,
                                int tie_bits, int64_t n, int64_t nseg, int64_t max_seg, float iou_thr, int flags,
                                int64_t max_keep, int64_t* keep_out, int64_t* num_keep, int64_t* seg_begin_out, void* ws,
                                size_t ws_bytes, void* stream) {
  return run_nms(0, dets5, 5, scores, 1, seg_id, tie, tie_bits, n, nseg, max_seg, iou_thr, flags, max_keep, keep_out,
                 num_keep, seg_begin_out, ws, ws_bytes, (hipStream_t)stream);
}

int obb_nms_poly_f32(const float* polys, int64_t row_stride, int64_t n, float iou_thr, int64_t max_keep, int64_t* keep_out,
                     int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  if (row_stride < 9) return OBB_ERR_BAD_ARG;
  return run_nms(1, polys, (int)row_stride, polys ? polys + 8 : nullptr, (int)row_stride, nullptr, nullptr, 0, n, 1, n, iou_thr,
                 0, max_keep, keep_out, num_keep, nullptr, ws, ws_bytes, (hipStream_t)stream);
}


// Devkit host-pointer API (DOTA_devkit/poly_nms_gpu/poly_nms.hpp:9-10, poly_nms_kernel.cu:277-329): the rows
// arrive pre-sorted by the caller (poly_nms.pyx:18-21) and are scanned in the given order; keep_out receives
// POSITIONS into that order.  Synchronous; errors are printed like the reference's CUDA_CHECK.
void _poly_nms(int* keep_out_host, int* num_out_host, const float* polys_host, int polys_num, int polys_dim,
               float nms_overlap_thresh, int device_id) {
  if (num_out_host) *num_out_host = 0;
  if (polys_num <= 0 || polys_dim < 8) return;
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) { fprintf(stderr, "_poly_nms: no HIP device\n"); return; }
  if (device_id >= 0 && device_id != cur) hipSetDevice(device_id);
  const size_t n = (size_t)polys_num;
  float *dp = nullptr, *ds = nullptr; int64_t *dk = nullptr, *dn = nullptr; void* ws = nullptr;
  const size_t wsb = obb_nms_workspace_bytes(polys_num, 1, 1);
  float* hs = (float*)malloc(n * 4);
  int64_t* hk = (int64_t*)malloc(n * 8);
  for (size_t i = 0; i < n; i++) hs[i] = (float)(n - i);      // strictly decreasing: keeps the given order (n < 2^24)
  hipError_t e = hipMalloc(&dp, n * polys_dim * 4);
  if (e == hipSuccess) e = hipMalloc(&ds, n * 4);
  if (e == hipSuccess) e = hipMalloc(&dk, n * 8);
  if (e == hipSuccess) e = hipMalloc(&dn, 8);
  if (e == hipSuccess) e = hipMalloc(&ws, wsb);
  if (e == hipSuccess) e = hipMemcpy(dp, polys_host, n * polys_dim * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ds, hs, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    int rc = run_nms(1, dp, polys_dim, ds, 1, nullptr, nullptr, 0, polys_num, 1, polys_num, nms_overlap_thresh, 0, 0, dk, dn,
                     nullptr, ws, wsb, (hipStream_t)0);
    if (rc) fprintf(stderr, "_poly_nms: launch failed (%d)\n", rc);
    int64_t cnt = 0;
    e = hipMemcpy(&cnt, dn, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && cnt > 0) e = hipMemcpy(hk, dk, (size_t)cnt * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
      for (int64_t i = 0; i < cnt; i++) keep_out_host[i] = (int)hk[i];
      if (num_out_host) *num_out_host = (int)cnt;
    }
  }
  if (e != hipSuccess) fprintf(stderr, "_poly_nms: %s\n", hipGetErrorString(e));
  hipFree(dp); hipFree(ds); hipFree(dk); hipFree(dn); hipFree(ws);
  free(hs); free(hk);
  if (device_id >= 0 && device_id != cur) hipSetDevice(cur);
}

const char* obb_version(void) { return "obb_hip 0.1 (gfx950)"; }

int obb_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return OBB_ERR_NO_DEVICE;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return OBB_ERR_NO_DEVICE;
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (arch_name && arch_name_len > 0) { strncpy(arch_name, p.gcnArchName, arch_name_len - 1); arch_name[arch_name_len - 1] = 0; }
  return OBB_OK;
}

}  // extern "C"
