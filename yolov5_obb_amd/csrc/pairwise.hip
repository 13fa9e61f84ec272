// Pairwise IoU entry points: element-wise pairs, dense matrices, and the devkit's
// rbox overlaps (DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu:280-353).
//
// Layout: rotated IoU: a 64 x 64 output tile per one-wave workgroup, lanes own COLUMNS (each store instruction writes 64
// consecutive floats of one output row), the row box is wave-uniform and read from LDS as a broadcast.  Quad IoU: 16 rows x a chunk of
// columns per one-wave workgroup, exact zeros written at once and the clips run from a compacted queue (k_quad_strip).  The <= 24 / 20
// clip points of each lane live in an LDS column (bank == lane).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "geom.h"
#include "obb_hip.h"

namespace obb {

__global__ __launch_bounds__(64) void k_riou_pairs(const float* __restrict__ a5, const float* __restrict__ b5, long long n,
                                                   float* __restrict__ out) {
  __shared__ float scr[RotGeom::SCR * 64];
  long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* a = a5 + i * 5; const float* b = b5 + i * 5;
  RBoxFeat A = rbox_make_feat(a[0], a[1], a[2], a[3], a[4]);
  RBoxFeat B = rbox_make_feat(b[0], b[1], b[2], b[3], b[4]);
  out[i] = rot_iou_value(A, B, scr + threadIdx.x);
}

__global__ __launch_bounds__(64) void k_riou_matrix(const float* __restrict__ a5, long long n, const float* __restrict__ b5,
                                                    long long k, float* __restrict__ out) {
  __shared__ float4 rowrec[64 * 4];
  __shared__ float scr[RotGeom::SCR * 64];
  const int lane = threadIdx.x;
  const long long i0 = (long long)blockIdx.y * 64, j = (long long)blockIdx.x * 64 + lane;
  {
    long long i = i0 + lane;
    RBoxFeat f = {};
    if (i < n) { const float* a = a5 + i * 5; f = rbox_make_feat(a[0], a[1], a[2], a[3], a[4]); }
    RotGeom::pack(f, rowrec + lane * 4);
  }
  RBoxFeat B = {};
  if (j < k) { const float* b = b5 + j * 5; B = rbox_make_feat(b[0], b[1], b[2], b[3], b[4]); }
  __syncthreads();
  const int nr = (int)((n - i0) < 64 ? (n - i0) : 64);
  for (int r = 0; r < nr; r++) {
    RBoxFeat A = RotGeom::unpack(rowrec[r * 4], rowrec[r * 4 + 1], rowrec[r * 4 + 2], rowrec[r * 4 + 3]);
    if (j < k) {
      // the reject only fires where the reference returns exactly 0 (riou_device.h)
      float v = rbox_certainly_disjoint(A, B) ? 0.f : rot_iou_value(A, B, scr + lane);
      out[(i0 + r) * k + j] = v;
    }
  }
}

// RotBox2Poly (poly_overlaps_kernel.cu:280-297): fp32 cos/sin, corner arithmetic in double
// (the "/ 2.0" literals promote), one rounding to float per coordinate.
// The reference's `float cs = cos(dbox[4])` is CUDA's device cosf: within 2 ulp of the true value, not correctly rounded, and
// not reproducible by any other libm (ocml's cosf and glibc's differ from it and from each other in the last bit for some
// angles: up to 5e-5 of IoU on a 4-pixel box).  Both this kernel and the oracle take the CORRECTLY ROUNDED float -- the
// double-precision cos / sin rounded once -- which lies inside the reference's own error band and makes the two sides agree
// bit for bit (round 3 used each side's cosf: 1e-4 max between them).
__device__ __forceinline__ void rbox_to_quad_devkit(const float* d, float* qx, float* qy) {
  float cs = (float)cos((double)d[4]), ss = (float)sin((double)d[4]);
  double w = d[2], h = d[3], x = d[0], y = d[1];
  qx[0] = (float)(x + cs * (w / 2.0) - ss * (-h / 2.0));
  qx[1] = (float)(x + cs * (w / 2.0) - ss * (h / 2.0));
  qx[2] = (float)(x + cs * (-w / 2.0) - ss * (h / 2.0));
  qx[3] = (float)(x + cs * (-w / 2.0) - ss * (-h / 2.0));
  qy[0] = (float)(y + ss * (w / 2.0) + cs * (-h / 2.0));
  qy[1] = (float)(y + ss * (w / 2.0) + cs * (h / 2.0));
  qy[2] = (float)(y + ss * (-w / 2.0) + cs * (h / 2.0));
  qy[3] = (float)(y + ss * (-w / 2.0) + cs * (-h / 2.0));
}

// Dense quad IoU: one WAVE per unit of 16 rows x `chunk` columns (devPolyIoU, utils/nms_rotated/src/poly_nms_cuda.cu:122-142;
// DEVKIT: the rows / columns are rboxes turned into quads by RotBox2Poly, DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu:280-353).
// The wave walks its 16 rows over the column tiles of 64 (a column per lane), writes the exact zeros of the pairs one of the two
// PROVED cone rules vouches for (piou_device.h: the column quad counter-clockwise of the row quad as seen from the origin, or
// clockwise of it -- tier 1 on the row's extended cone, tier 2 on its plain cone plus the pair check) and pushes the others
// into a ring queue in LDS; the clip (48 half-plane cuts, ~2.5 x 10^4 instructions per wave) only ever runs on 64 queued pairs at
// a time, whatever tile they come from.  On S-uniform rboxes over 1024 px nine pairs in ten are exact zeros.  One-wave
// workgroups of 11 KB LDS: 14 per CU, handed out by the dispatcher as they finish (the units' clip counts differ), no
// workgroup barrier anywhere; the row records live in the first 16 lanes' registers (v_readlane per row), the queued pair's
// column quad is read again from global memory when its clip runs.
#ifndef OBB_QS_ROWS
#define OBB_QS_ROWS 16
#endif
constexpr int kQsRows = OBB_QS_ROWS;      // <= 16 (four bits of a queue entry)
template <bool DEVKIT>
__global__ __launch_bounds__(64) void k_quad_strip(const float* __restrict__ a, long long sa, long long n, const float* __restrict__ b,
                                                   long long sb, long long k, float* __restrict__ out, int chunk, int nchunks) {
  __shared__ float4 rowq[kQsRows * 2];
  __shared__ uint32_t queue[128];
  __shared__ float scr[QuadGeom::SCR * 64];
  const int lane = threadIdx.x;
  const long long strip = blockIdx.x / (unsigned)nchunks, ch = blockIdx.x % (unsigned)nchunks;
  const long long i0 = strip * kQsRows, j0 = ch * chunk;
  const long long j1 = (k - j0) < chunk ? k : j0 + chunk;
  auto load_quad = [&](const float* src, long long st, long long idx) {
    QuadFeat q = {};
    if (DEVKIT) rbox_to_quad_devkit(src + idx * 5, q.x, q.y);
    else {
#pragma unroll
      for (int c = 0; c < 4; c++) { q.x[c] = src[idx * st + 2 * c]; q.y[c] = src[idx * st + 2 * c + 1]; }
    }
    return q;
  };
  uint32_t rc = kConeNone, re = kConeNone, rr = kRmNone;      // lane r < 16: row r's plain cone, extended cone, (r, M)
  if (lane < kQsRows) {
    QuadFeat q = {};
    if (i0 + lane < n) {
      q = load_quad(a, sa, i0 + lane);
      rc = quad_cone_bits(q);
      const QuadCone2 c2 = quad_cone2_bits(q);
      re = c2.ext; rr = c2.rm;
    }
    rowq[lane * 2] = make_float4(q.x[0], q.y[0], q.x[1], q.y[1]);
    rowq[lane * 2 + 1] = make_float4(q.x[2], q.y[2], q.x[3], q.y[3]);
  }
  __builtin_amdgcn_wave_barrier();
  int head = 0, count = 0;                               // wave-uniform
  auto drain = [&](int cnt) {
    __builtin_amdgcn_wave_barrier();
    const bool want = lane < cnt;
    int r = 0;
    long long j = 0;
    QuadFeat A = {}, B = {};
    if (want) {
      const uint32_t e = queue[(head + lane) & 127];
      r = (int)(e >> 28);
      j = j0 + (long long)(e & 0x0fffffffu);
      A = QuadGeom::unpack(rowq[r * 2], rowq[r * 2 + 1]); B = load_quad(b, sb, j);
    }
    if (cnt <= QuadGeom::kCoopMax) {                      // a partial queue: sixteen lanes per pair (geom.h), same bits
      const float v[16] = {A.x[0], A.y[0], A.x[1], A.y[1], A.x[2], A.y[2], A.x[3], A.y[3], B.x[0], B.y[0], B.x[1], B.y[1], B.x[2], B.y[2], B.x[3], B.y[3]};
      const float val = QuadGeom::iou_coop(want, v, scr);
      if (want) out[(i0 + r) * k + j] = val;
    } else if (want) {
      out[(i0 + r) * k + j] = QuadGeom::iou(A, B, scr + lane);
    }
    head = (head + cnt) & 127; count -= cnt;
    __builtin_amdgcn_wave_barrier();
  };
  const int nr = (int)((n - i0) < kQsRows ? (n - i0) : kQsRows);
  for (long long jt = j0; jt < j1; jt += 64) {
    const long long j = jt + lane;
    const bool cvalid = j < j1;
    QuadFeat B = {};
    uint32_t cone_b = kConeNone, rm_b = kRmNone;
    if (cvalid) { B = load_quad(b, sb, j); cone_b = quad_cone_bits(B); rm_b = quad_cone2_bits(B).rm; }
    for (int r = 0; r < nr; r++) {
      const uint32_t prc = (uint32_t)__builtin_amdgcn_readlane((int)rc, r), pre = (uint32_t)__builtin_amdgcn_readlane((int)re, r),
                     prr = (uint32_t)__builtin_amdgcn_readlane((int)rr, r);
      // either cone rule: all 16 terms of the reference's sum are exactly zero -> IoU = +0, no clip
      bool skip = quad_cone_skip(prc, cone_b) || quad_cone2_skip(pre, prr, cone_b, rm_b);
      if (!skip && quad_cone2_skip(prc, prr, cone_b, rm_b))
        skip = quad_cone2_nofuzzy(QuadGeom::unpack(rowq[r * 2], rowq[r * 2 + 1]), B);
      if (cvalid && skip) out[(i0 + r) * k + j] = 0.f;
      const bool work = cvalid && !skip;
      const unsigned long long m = __ballot(work);
      if (work) queue[(head + count + __popcll(m & ((1ull << lane) - 1ull))) & 127] = ((uint32_t)r << 28) | (uint32_t)(j - j0);
      count += __popcll(m);
      if (count >= 64) drain(64);
    }
  }
  if (count > 0) drain(count);
}

// units of 16 rows x chunk columns, chunk a multiple of 64 chosen for >= ~3 units per wave slot of the device (256 CUs x 14)
static int quad_strip_launch(bool devkit, const float* a, long long sa, long long n, const float* b, long long sb, long long k, float* out,
                             hipStream_t st) {
  const long long strips = (n + kQsRows - 1) / kQsRows;
  long long chunk = strips * k / 10752 / 64 * 64;
  chunk = chunk < 64 ? 64 : (chunk > 1024 ? 1024 : chunk);
  const long long nchunks = (k + chunk - 1) / chunk;
  if (strips * nchunks > 0x7fffffffLL || nchunks > 0x7fffffffLL) return OBB_ERR_BAD_ARG;
  if (devkit) k_quad_strip<true><<<(unsigned)(strips * nchunks), 64, 0, st>>>(a, sa, n, b, sb, k, out, (int)chunk, (int)nchunks);
  else k_quad_strip<false><<<(unsigned)(strips * nchunks), 64, 0, st>>>(a, sa, n, b, sb, k, out, (int)chunk, (int)nchunks);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

// DOTA Task-1 evaluation, the det x GT part of voc_eval (DOTA_devkit/dota_evaluation_task1.py:168-223): for every
// detection, over the ground-truth quads of ITS image, the horizontal-box gate with the +1 convention (:181-204,
// overlaps > 0), polyiou.cpp's iou_poly(GT, det) in double for the survivors (:206-213), np.max / np.argmax (:215-218: the
// first maximum in GT order; any NaN makes the maximum NaN and argmax the first NaN).
// One wave per detection (grid-stride), lanes over the image's GT list, exact stage in two half-wave passes on a
// 32-column LDS scratch.
constexpr int kEvalWaves = 4;
__global__ __launch_bounds__(64 * kEvalWaves) void k_eval_best_gt(const double* __restrict__ dets8, const int32_t* __restrict__ det_img,
                                                                 long long nd, const double* __restrict__ gts8,
                                                                 const int32_t* __restrict__ gt_off, double* __restrict__ ovmax,
                                                                 int32_t* __restrict__ jmax) {
  __shared__ double scr_all[kEvalWaves][40 * 32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* base = scr_all[wv] + (lane & 31);
  for (long long d = (long long)blockIdx.x * kEvalWaves + wv; d < nd; d += (long long)gridDim.x * kEvalWaves) {
    QuadFeatT<double> D;
    const double2* dp = reinterpret_cast<const double2*>(dets8 + d * 8);
#pragma unroll
    for (int k = 0; k < 4; k++) { double2 v = dp[k]; D.x[k] = v.x; D.y[k] = v.y; }
    D.minx = D.maxx = D.miny = D.maxy = 0.0;
    double bx1, by1, bx2, by2;
    QuadGeom64::hbb(D, &bx1, &by1, &bx2, &by2);
    const double barea = (bx2 - bx1 + 1.) * (by2 - by1 + 1.);
    const int img = det_img[d];
    const int g0 = gt_off[img], g1 = gt_off[img + 1];
    double best = -INFINITY;
    int bestj = -1, nanj = 0x7fffffff;
    for (int gb = g0; gb < g1; gb += 64) {
      const int j = gb + lane;
      bool pass = false;
      QuadFeatT<double> Gq;
      Gq.minx = Gq.maxx = Gq.miny = Gq.maxy = 0.0;
      if (j < g1) {
        const double2* gp = reinterpret_cast<const double2*>(gts8 + (long long)j * 8);
#pragma unroll
        for (int k = 0; k < 4; k++) { double2 v = gp[k]; Gq.x[k] = v.x; Gq.y[k] = v.y; }
        double gx1, gy1, gx2, gy2;
        QuadGeom64::hbb(Gq, &gx1, &gy1, &gx2, &gy2);
        const double iw = fmax(fmin(gx2, bx2) - fmax(gx1, bx1) + 1., 0.), ih = fmax(fmin(gy2, by2) - fmax(gy1, by1) + 1., 0.);
        const double inters = iw * ih;
        const double uni = barea + (gx2 - gx1 + 1.) * (gy2 - gy1 + 1.) - inters;
        pass = inters / uni > 0;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { Gq.x[k] = 0.0; Gq.y[k] = 0.0; }
      }
      if (__ballot(pass) == 0) continue;
      int nhalf = 2;
      asm volatile("" : "+s"(nhalf));   // two passes stay two passes: lanes l and l + 32 share a scratch column
      for (int half = 0; half < nhalf; half++) {
        if (pass && (lane >> 5) == half) {
          const double iou = quad_iou_t<32, false, double>(Gq, D, base, base + 10 * 32, base + 20 * 32, base + 30 * 32);
          if (iou != iou) { if (j - g0 < nanj) nanj = j - g0; }
          else if (iou > best) { best = iou; bestj = j - g0; }
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ob = __shfl_xor(best, off);
      const int oj = __shfl_xor(bestj, off), on = __shfl_xor(nanj, off);
      if (oj >= 0 && (bestj < 0 || ob > best || (ob == best && oj < bestj))) { best = ob; bestj = oj; }
      if (on < nanj) nanj = on;
    }
    if (lane == 0) {
      const bool isn = nanj != 0x7fffffff;
      ovmax[d] = isn ? (double)NAN : best;
      jmax[d] = isn ? nanj : bestj;
    }
  }
}

}  // namespace obb

using namespace obb;

#define OBB_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH)

extern "C" {

int obb_rotated_iou_pairs_f32(const float* a5, const float* b5, int64_t n, float* out, void* stream) {
  if (n < 0 || (n > 0 && (!a5 || !b5 || !out))) return OBB_ERR_BAD_ARG;
  if (n == 0) return OBB_OK;
  k_riou_pairs<<<(unsigned)((n + 63) / 64), 64, 0, (hipStream_t)stream>>>(a5, b5, n, out);
  return OBB_CHECK_LAUNCH();
}

int obb_rotated_iou_matrix_f32(const float* a5, int64_t n, const float* b5, int64_t k, float* out, void* stream) {
  if (n < 0 || k < 0 || (n > 0 && k > 0 && (!a5 || !b5 || !out))) return OBB_ERR_BAD_ARG;
  if (n == 0 || k == 0) return OBB_OK;
  dim3 g((unsigned)((k + 63) / 64), (unsigned)((n + 63) / 64));
  if (g.y > 65535) return OBB_ERR_BAD_ARG;
  k_riou_matrix<<<g, 64, 0, (hipStream_t)stream>>>(a5, n, b5, k, out);
  return OBB_CHECK_LAUNCH();
}

int obb_quad_iou_matrix_f32(const float* a, int64_t a_stride, int64_t n, const float* b, int64_t b_stride, int64_t k,
                            float* out, void* stream) {
  if (n < 0 || k < 0 || a_stride < 8 || b_stride < 8 || (n > 0 && k > 0 && (!a || !b || !out))) return OBB_ERR_BAD_ARG;
  if (n == 0 || k == 0) return OBB_OK;
  return quad_strip_launch(false, a, a_stride, n, b, b_stride, k, out, (hipStream_t)stream);
}

int obb_rbox_overlaps_f32(const float* boxes5, int64_t n, const float* query5, int64_t k, float* out, void* stream) {
  if (n < 0 || k < 0 || (n > 0 && k > 0 && (!boxes5 || !query5 || !out))) return OBB_ERR_BAD_ARG;
  if (n == 0 || k == 0) return OBB_OK;
  return quad_strip_launch(true, boxes5, 5, n, query5, 5, k, out, (hipStream_t)stream);
}


int obb_eval_best_gt_f64(const double* dets8, const int32_t* det_img, int64_t nd, const double* gts8, const int32_t* gt_off,
                         int64_t n_img, double* ovmax, int32_t* jmax, void* stream) {
  if (nd < 0 || n_img < 0 || (nd > 0 && (!dets8 || !det_img || !gt_off || !ovmax || !jmax))) return OBB_ERR_BAD_ARG;
  if (nd == 0) return OBB_OK;
  int64_t nb = (nd + kEvalWaves - 1) / kEvalWaves;
  if (nb > 256 * 16) nb = 256 * 16;
  k_eval_best_gt<<<(unsigned)nb, 64 * kEvalWaves, 0, (hipStream_t)stream>>>(dets8, det_img, nd, gts8, gt_off, ovmax, jmax);
  return OBB_CHECK_LAUNCH();
}


// Devkit host-pointer API (DOTA_devkit/poly_nms_gpu/poly_overlaps.hpp:1, poly_overlaps_kernel.cu:368-427):
// allocate, upload, run, download, free; synchronous; errors are printed, not raised (CUDA_CHECK there
// prints and continues, poly_nms_kernel.cu:20-27).
void _overlaps(float* overlaps_host, const float* boxes_host, const float* query_boxes_host, int n, int k, int device_id) {
  if (n <= 0 || k <= 0) return;
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) { fprintf(stderr, "_overlaps: no HIP device\n"); return; }
  if (device_id >= 0 && device_id != cur) hipSetDevice(device_id);
  float *db = nullptr, *dq = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc(&db, (size_t)n * 5 * 4);
  if (e == hipSuccess) e = hipMalloc(&dq, (size_t)k * 5 * 4);
  if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n * k * 4);
  if (e == hipSuccess) e = hipMemcpy(db, boxes_host, (size_t)n * 5 * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dq, query_boxes_host, (size_t)k * 5 * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    int rc = obb_rbox_overlaps_f32(db, n, dq, k, dout, nullptr);
    if (rc) fprintf(stderr, "_overlaps: launch failed (%d)\n", rc);
    e = hipMemcpy(overlaps_host, dout, (size_t)n * k * 4, hipMemcpyDeviceToHost);
  }
  if (e != hipSuccess) fprintf(stderr, "_overlaps: %s\n", hipGetErrorString(e));
  hipFree(db); hipFree(dq); hipFree(dout);
  if (device_id >= 0 && device_id != cur) hipSetDevice(cur);
}

// (the reference declares it with C++ linkage, poly_overlaps.hpp:1: the same entry point under the mangled name its Cython
// source binds)
void obb_cxx_overlaps(float* overlaps_host, const float* boxes_host, const float* query_boxes_host, int n, int k, int device_id)
    __asm__("_Z9_overlapsPfPKfS1_iii");
void obb_cxx_overlaps(float* overlaps_host, const float* boxes_host, const float* query_boxes_host, int n, int k, int device_id) {
  _overlaps(overlaps_host, boxes_host, query_boxes_host, n, k, device_id);
}

}  // extern "C"
