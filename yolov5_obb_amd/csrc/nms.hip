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
#include <hip/hip_fp16.h>
#include "launch_once.h"
#include "nms_core.h"
#include "nms_mk.h"
#include "obb_hip.h"
#include "psrs_sort.h"
#include "nms_small.h"
#include "segsort.h"

namespace obb {

// ---------------------------------------------------------------- optional per-stage timing (HIP events)
// bench.py switches this on to time the dominant kernels on the stream they are launched on.
enum { PROF_DECODE = 0, PROF_SEGSORT, PROF_PREP, PROF_STEPS, PROF_GATHER, PROF_NMS_SORT, PROF_NMS_PREP, PROF_NMS_STEPS, PROF_N };
struct ProfState {
  int on = 0;          // 0 off, 1 every stage, 2 only the NMS kernels (PROF_STEPS / PROF_NMS_STEPS: the dominant kernel of a call)
  static constexpr int kMax = 8192;
  hipEvent_t ev0[kMax], ev1[kMax];
  int id[kMax];
  int created = 0, used = 0;
};
static ProfState g_prof;
struct ProfScope {
  int slot = -1; hipStream_t st;
  ProfScope(int stage, hipStream_t s) : st(s) {
    if (!g_prof.on || g_prof.used >= ProfState::kMax) return;
    if (g_prof.on == 2 && stage != PROF_STEPS && stage != PROF_NMS_STEPS) return;
    slot = g_prof.used++;
    if (slot >= g_prof.created) { hipEventCreate(&g_prof.ev0[slot]); hipEventCreate(&g_prof.ev1[slot]); g_prof.created = slot + 1; }
    g_prof.id[slot] = stage;
    hipEventRecord(g_prof.ev0[slot], st);
  }
  ~ProfScope() { if (slot >= 0) hipEventRecord(g_prof.ev1[slot], st); }
};

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

// min / max of four ordered ints over the workgroup (<= 1024 threads); every thread returns the result
__device__ __forceinline__ void block_minmax4(int& a0, int& a1, int& b0, int& b1, int (*s_red)[4]) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a0 = min(a0, __shfl_xor(a0, d)); a1 = min(a1, __shfl_xor(a1, d));
    b0 = max(b0, __shfl_xor(b0, d)); b1 = max(b1, __shfl_xor(b1, d));
  }
  const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_red[wv][0] = a0; s_red[wv][1] = a1; s_red[wv][2] = b0; s_red[wv][3] = b1; }
  __syncthreads();
  for (int k = 0; k < nw; k++) { a0 = min(a0, s_red[k][0]); a1 = min(a1, s_red[k][1]); b0 = max(b0, s_red[k][2]); b1 = max(b1, s_red[k][3]); }
}

// Single list: the runs of the sort (psrs_sort.h) straight from the scores -- key = descending-score bits in the high word,
// original index in the low word (unique: ties keep ascending index, the documented rule) -- plus everything the old key
// kernel did on the side: the single segment's table, the zeroing of the team-barrier block and of the index / slab block,
// and the bounding-box partial of the block's boxes (one per run) for the spatial index.
struct LocalExtras {
  int *seg_begin, *seg_end, *keep_cnt;
  uint4* bar16; long long n_bar16;
  uint4* grid16; long long n_grid16;
  uint4* mk16; long long n_mk16;   // control block of the phase-kernel path (nms_mk.h)
  int* bbpart;                 // [runs][kBbInts] or NULL
  u64* tstart;                 // optional: the device's wall clock when the call's first kernel runs (k_finalize reports the call's duration)
};
__device__ __forceinline__ void local_extras(const LocalExtras& x, const float* __restrict__ dets5, int drop_small, int n, int i, bool in_range,
                                             int (*s_red)[4], uint32_t* key_out) {
  if (i == 0) { x.seg_begin[0] = 0; x.seg_end[0] = n; x.keep_cnt[0] = 0; if (x.tstart) *x.tstart = wall_clock64(); }
  for (long long k = i; k < x.n_bar16; k += (long long)gridDim.x * blockDim.x) x.bar16[k] = make_uint4(0u, 0u, 0u, 0u);
  for (long long k = i; k < x.n_grid16; k += (long long)gridDim.x * blockDim.x) x.grid16[k] = make_uint4(0u, 0u, 0u, 0u);   // GridMeta + slot counters
  for (long long k = i; k < x.n_mk16; k += (long long)gridDim.x * blockDim.x) x.mk16[k] = make_uint4(0u, 0u, 0u, 0u);
  int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = (int)0x80000000, by1 = (int)0x80000000;
  int d2 = 0;                                          // largest w^2 + h^2 of the block, as float bits (>= 0: ordered like ints)
  float rsum = 0.f; int rcnt = 0;
  if (in_range) {
    bool ok = true;
    float w = 0.f, h = 0.f;
    if (drop_small || x.bbpart != nullptr) { w = dets5[(size_t)i * 5 + 2]; h = dets5[(size_t)i * 5 + 3]; }
    if (drop_small) {
      // nms_rotated_wrapper.py:32  too_small = dets[:, [2, 3]].min(1)[0] < 0.001   (torch.min propagates NaN; NaN < 0.001 is False)
      float mn = (h < w) ? h : w;
      if (mn < 0.001f) { *key_out = 0xFFFFFFFFu; ok = false; }
    }
    if (x.bbpart != nullptr && ok) {
      // bounding box of the finite centres of the boxes that take part: the extent of the data for the spatial index
      // (grid.h), one partial per block, reduced by the blocks of the prep kernel (no atomics)
      const float cx = dets5[(size_t)i * 5], cy = dets5[(size_t)i * 5 + 1];
      if ((cx - cx == 0.f) && (cy - cy == 0.f)) { bx0 = bx1 = grid_f2o(cx); by0 = by1 = grid_f2o(cy); }
      const float q = w * w + h * h;
      if (q - q == 0.f) { d2 = __float_as_int(q); rsum = sqrtf(q); rcnt = 1; }
    }
  }
  if (x.bbpart != nullptr) {
    int d2b = d2, dummy = d2;
    block_minmax4(bx0, by0, bx1, by1, s_red);
    { int lo0 = 0x7fffffff, lo1 = 0x7fffffff; block_minmax4(lo0, lo1, d2b, dummy, s_red); }
    // (nms_mk.h: the mean diagonal of the finite boxes, for the radius limit of its tables -- a heuristic, any order of summation will do)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { rsum += __shfl_xor(rsum, d); rcnt += __shfl_xor(rcnt, d); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = __float_as_int(rsum); s_red[threadIdx.x >> 6][1] = rcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float rs = 0.f; int rc = 0;
      for (int k = 0; k < (int)((blockDim.x + 63) >> 6); k++) { rs += __int_as_float(s_red[k][0]); rc += s_red[k][1]; }
      int* o = x.bbpart + (size_t)blockIdx.x * kBbInts;
      o[0] = bx0; o[1] = by0; o[2] = bx1; o[3] = by1; o[4] = d2b; o[5] = __float_as_int(rs); o[6] = rc; o[7] = 0;
    }
  }
}
__global__ __launch_bounds__(kPsRun) void k_ps_local_scores(PsBuf b, const float* __restrict__ scores, int score_stride,
                                                            const float* __restrict__ dets5, int drop_small, LocalExtras x) {
  __shared__ unsigned long long s_k[kPsRun];
  __shared__ uint32_t s_v[kPsRun];
  __shared__ int s_red[16][4];
  const int n = b.n, tid = threadIdx.x;
  const int i = blockIdx.x * kPsRun + tid;
  uint32_t k32 = 0xFFFFFFFFu;
  if (i < n) k32 = score_desc_key(scores[(size_t)i * score_stride]);
  local_extras(x, dets5, drop_small, n, i, i < n, s_red, &k32);
  s_k[tid] = ((unsigned long long)k32 << 32) | (unsigned long long)(uint32_t)i;      // (a pad: 0xFFFFFFFF | position >= n)
  s_v[tid] = (uint32_t)i;
  ps_local_tail(b, n, s_k, s_v);
}
// ... and the plain key kernel for lists the three-launch sort does not take (n > kPsMaxN): 64-bit keys (score bits << 32 |
// index) for segsort.h's LSD radix sort over the four score bytes (stable: ties keep ascending index)
__global__ __launch_bounds__(kPsRun) void k_make_keys_wide(const float* __restrict__ scores, int score_stride, const float* __restrict__ dets5,
                                                           int drop_small, int n, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                                                           LocalExtras x) {
  __shared__ int s_red[16][4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t k32 = 0xFFFFFFFFu;
  if (i < n) k32 = score_desc_key(scores[(size_t)i * score_stride]);
  local_extras(x, dets5, drop_small, n, i, i < n, s_red, &k32);
  if (i < n) { keys[i] = ((unsigned long long)k32 << 32) | (unsigned long long)(uint32_t)i; vals[i] = (uint32_t)i; }
}

// ---------------------------------------------------------------- spatial index (grid.h): counting sort by cell
// (everything except the bounding-box partials is produced inside the NMS kernel, when a step needs the index: nms_core.h grid_build)
struct GridDev {
  GridMeta* meta;              // zeroed before the launch
  int* bbpart; int nparts;     // per-block bounding boxes written by k_make_keys32
  int* cnt;                    // [M + 4] boxes per table slot (zeroed before the launch; the in-kernel build leaves zeros)
  int* start;                  // [M + 4] exclusive prefix, start[M] = total
  int* wsum;                   // [kMaxTeams] per-workgroup totals of the in-kernel scan
  float4* sorted;              // [n] {x, y, r, position}
  uint32_t* slot_of;           // [n] position -> entry of `sorted`
  uint32_t* ulist;             // [n] positions kept out of the index
  uint32_t mask;               // M - 1
  // independent slabs (grid.h; nms_core.h slab_setup): coverage bitmap + flag live in the zeroed block behind GridMeta
  uint32_t* slab_cover; int* slab_flag;
  int* slab_cnt;               // [kMaxTeams][kMaxSlabs] boxes per (workgroup, slab)
  int* slab_tot;               // [1 + kMaxTeams / 16][kMaxSlabs] totals (in the zeroed block)
  int* slab_keep;              // [kMaxSlabs] kept boxes per slab
  float4* rec2; uint32_t* order2; uint32_t* pos_old; u64* alive2; u64* kept_bits;   // the slab-major copy of the list
  size_t alive2_words, kept_words;
  SlabPlan* slab_plan;         // written by k_slab_split
};

// the extent of the data from the key kernel's per-block partials (every block reduces them itself: a few hundred int4)
__device__ __forceinline__ GridPlan plan_from_partials(const int* __restrict__ bbpart, int nparts, int (*s_red)[4], float* max_w2h2) {
  int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = (int)0x80000000, by1 = (int)0x80000000, d2 = 0, d2b = 0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    const int4 q = reinterpret_cast<const int4*>(bbpart)[2 * i];
    bx0 = min(bx0, q.x); by0 = min(by0, q.y); bx1 = max(bx1, q.z); by1 = max(by1, q.w);
    d2 = max(d2, bbpart[(size_t)i * kBbInts + 4]);
  }
  block_minmax4(bx0, by0, bx1, by1, s_red);
  { int lo0 = 0x7fffffff, lo1 = 0x7fffffff; d2b = d2; block_minmax4(lo0, lo1, d2, d2b, s_red); }
  *max_w2h2 = __int_as_float(d2);
  const int bb[4] = {bx0, by0, bx1, by1};
  return grid_plan(bb);
}

// blockDim.x is a multiple of 64 and p starts at 0: every wave covers one word of the alive bitmap
// slab_cover != NULL: every box that takes part also marks the x bins its circle interval touches (grid.h, "independent
// slabs"), first in a bitmap of the block in LDS, then with one atomicOr per non-zero word.
__global__ __launch_bounds__(256) void k_prep_rot(const float* __restrict__ dets5, const uint32_t* __restrict__ order, int drop_small, int n,
                                                  float4* __restrict__ rec, u64* __restrict__ alive, const int* __restrict__ bbpart,
                                                  int nparts, uint32_t* __restrict__ slab_cover, int* __restrict__ slab_flag) {
  __shared__ int s_red[16][4];
  __shared__ uint32_t s_cover[kSlabWords];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false, bad = false;
  float x = 0.f, y = 0.f, r = 0.f, ms2 = 0.f;
  if (p < n) {
    const float* d = dets5 + (size_t)order[p] * 5;
    float w = d[2], h = d[3], a = d[4];
    x = d[0]; y = d[1];
    RBoxFeat f = rbox_make_feat(x, y, w, h, a);
    float4 q[4];
    RotGeom::pack(f, q);
#pragma unroll
    for (int k = 0; k < 4; k++) rec[(size_t)p * 4 + k] = q[k];
    float mn = (h < w) ? h : w;
    ok = !(drop_small && mn < 0.001f);
    r = q[0].z; ms2 = q[0].w;
  }
  const u64 m = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && (p & ~63) < n) alive[p >> 6] = m;
  if (p < 8) alive[((n + 63) >> 6) + p] = 0ull;      // guard words behind the last box (the bitmap is not memset)
  if (slab_cover == nullptr) return;
  // ---- independent slabs (grid.h): is the data wide enough to look for them at all (slab_flag[2], written by block 0;
  // every block derives the same answer from the same partials)?  If so, the x bins this block's boxes touch, first in
  // a bitmap of the block in LDS, then OR-ed into one of kSlabCopies global copies
  float max_w2h2 = 0.f;
  const GridPlan gp = plan_from_partials(bbpart, nparts, s_red, &max_w2h2);
  const bool gate = slab_gate(gp, max_w2h2);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    slab_flag[2] = gate ? 1 : 0;
    slab_flag[4] = __float_as_int(gp.x0);                    // the bins' origin and scale: the NMS kernel reads them instead of
    slab_flag[5] = __float_as_int(slab_inv_bin(gp));         // reducing the partials once more
  }
  if (!gate) return;
  for (int k = threadIdx.x; k < kSlabWords; k += blockDim.x) s_cover[k] = 0u;
  __syncthreads();
  const float inv = slab_inv_bin(gp);
  if (ok) {
    if (!slab_box_ok(gp, x, y, r, ms2)) bad = true;
    else {
      const float hw = slab_halfwidth(gp, x, r);
      const int b0 = slab_bin(x - hw, gp.x0, inv), b1 = slab_bin(x + hw, gp.x0, inv);
      for (int wd = b0 >> 5; wd <= (b1 >> 5); wd++) {
        const int lo = max(b0, wd * 32) & 31, hi = min(b1, wd * 32 + 31) & 31;
        const uint32_t mk = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
        if ((s_cover[wd] & mk) != mk) atomicOr(&s_cover[wd], mk);
      }
    }
  }
  const int anybad = __syncthreads_or(bad ? 1 : 0);
  if (anybad && threadIdx.x == 0) atomicOr(slab_flag, 1);
  uint32_t* dst = slab_cover + (size_t)(blockIdx.x % kSlabCopies) * kSlabWords;
  for (int k = threadIdx.x; k < kSlabWords; k += blockDim.x) {
    const uint32_t v = s_cover[k];
    if (v && (__hip_atomic_load(dst + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & v) != v) atomicOr(dst + k, v);
  }
}

// ---- float64 rotated boxes (RotGeom64): 64-bit keys from the double scores, records from the double boxes
__device__ __forceinline__ uint64_t score_desc_key64(double s) {
  uint64_t u = (uint64_t)__double_as_longlong(s);
  uint64_t k;
  if (s != s) k = ~0ull;                                   // NaN first (torch's order), as score_desc_key
  else {
    if (s == 0.0) u = 0ull;                                // -0 == +0
    k = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
  }
  return ~k;
}
__global__ void k_make_keys_f64(const double* __restrict__ scores, const double* __restrict__ dets5, int drop_small, int n,
                                unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, int* seg_begin, int* seg_end, int* keep_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { seg_begin[0] = 0; seg_end[0] = n; keep_cnt[0] = 0; }
  if (i >= n) return;
  uint64_t k = score_desc_key64(scores[i]);
  if (drop_small) {
    const double w = dets5[(size_t)i * 5 + 2], h = dets5[(size_t)i * 5 + 3];
    const double mn = (h < w) ? h : w;
    if (mn < 0.001) k = ~0ull;                             // nms_rotated_wrapper.py:32, compared in the tensor's dtype
  }
  keys[i] = k;
  vals[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_prep_rot64(const double* __restrict__ dets5, const uint32_t* __restrict__ order, int drop_small, int n,
                                                    float4* __restrict__ rec, u64* __restrict__ alive) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (p < n) {
    const double* d = dets5 + (size_t)order[p] * 5;
    const double x = d[0], y = d[1], w = d[2], h = d[3], a = d[4];
    float4 q[RotGeom64::RECQ];
    RotGeom64::pack(x, y, w, h, a, q);
#pragma unroll
    for (int k = 0; k < RotGeom64::RECQ; k++) rec[(size_t)p * RotGeom64::RECQ + k] = q[k];
    const double mn = (h < w) ? h : w;
    ok = !(drop_small && mn < 0.001);
  }
  const u64 m = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && (p & ~63) < n) alive[p >> 6] = m;
  if (p < 8) alive[((n + 63) >> 6) + p] = 0ull;      // guard words behind the last box (the bitmap is not memset)
}

__global__ void k_prep_quad(const float* __restrict__ polys, int stride, const uint32_t* __restrict__ order, int n, float thr, int skip,
                            float4* __restrict__ rec, u64* __restrict__ alive) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    const float* d = polys + (size_t)order[p] * stride;
    float c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = d[k];
    // skip bit 1: the bounding-box rule (thr <= 0 or bit clear: budget -inf); bit 0: the exact cone rule
    const QuadFeat qf = quad_make_feat(c);
    const QuadSkip sk = quad_skip_record(qf, (skip & 2) ? thr : 0.f);
    const uint32_t cone = (skip & 1) ? quad_cone_bits(qf) : kConeNone;
    uint32_t w0 = sk.lo, w1 = sk.hi;
    QuadCone2 c2w; c2w.ext = kConeNone; c2w.rm = kRmNone;
    if (skip & 1) c2w = quad_cone2_bits(qf);
    if ((skip & 1) && sk.f == -__builtin_huge_valf()) {
      // a quad without a budget takes no part in the bounding-box rule: its two box words carry what the SECOND proved rule
      // needs instead (piou_device.h: the extended cone and (r, M); QuadGeom::cheap_reject reads them when both budgets are -inf)
      w0 = c2w.ext; w1 = c2w.rm;
    }
    float4* r = rec + (size_t)p * QuadGeom::RECQ;
    r[0] = make_float4(__builtin_bit_cast(float, w0), __builtin_bit_cast(float, w1), sk.f, __builtin_bit_cast(float, cone));
    r[1] = make_float4(c[0], c[1], c[2], c[3]);
    r[2] = make_float4(c[4], c[5], c[6], c[7]);
    r[3] = make_float4(__builtin_bit_cast(float, c2w.ext), __builtin_bit_cast(float, c2w.rm), 0.f, 0.f);     // the second cone rule's words (QuadGeom::classify_quick)
  }
  const u64 m = __ballot(p < n);
  if ((threadIdx.x & 63) == 0 && (p & ~63) < n) alive[p >> 6] = m;
  if (p < 8) alive[((n + 63) >> 6) + p] = 0ull;      // guard words behind the last box (the bitmap is not memset)
}

// merge NMS (QuadGeom64): rows arrive as (n, 9) doubles, `order` lists the row indices in processing order, segment by
// segment.  Record = fp32 AABB rounded outward + the 8 doubles.
__device__ __forceinline__ float f32_down(double v) { float f = (float)v; return ((double)f > v) ? nextafterf(f, -INFINITY) : f; }
__device__ __forceinline__ float f32_up(double v) { float f = (float)v; return ((double)f < v) ? nextafterf(f, INFINITY) : f; }
// allow_reject = 0 (a negative or NaN threshold: even a ratio of 0 removes a box) or a coordinate that is not finite: the hot
// loop's box is the whole plane, i.e. the pair always reaches hit_exact, which applies numpy's rules to it
__global__ void k_prep_quad64(const double* __restrict__ dets9, const int32_t* __restrict__ order, int n, int allow_reject,
                              float4* __restrict__ rec, u64* __restrict__ alive) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    const double* d = dets9 + (size_t)order[p] * 9;
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = d[k];
    const double x1 = fmin(fmin(v[0], v[2]), fmin(v[4], v[6])), x2 = fmax(fmax(v[0], v[2]), fmax(v[4], v[6]));
    const double y1 = fmin(fmin(v[1], v[3]), fmin(v[5], v[7])), y2 = fmax(fmax(v[1], v[3]), fmax(v[5], v[7]));
    float4* r = rec + (size_t)p * QuadGeom64::RECQ;
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 8; k++) fin = fin && (v[k] - v[k] == 0.0);
    const float inf = __builtin_huge_valf();
    r[0] = (allow_reject && fin) ? make_float4(f32_down(x1), f32_down(y1), f32_up(x2), f32_up(y2)) : make_float4(-inf, -inf, inf, inf);
    double2* q = reinterpret_cast<double2*>(r + 1);
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = make_double2(v[2 * k], v[2 * k + 1]);
  }
  const u64 m = __ballot(p < n);
  if ((threadIdx.x & 63) == 0 && (p & ~63) < n) alive[p >> 6] = m;
  if (p < 8) alive[((n + 63) >> 6) + p] = 0ull;      // guard words behind the last box (the bitmap is not memset)
}

// records of the horizontal-box merge NMS (HbbGeom64): columns 0..3 of rows of `stride` doubles, in processing order
__global__ void k_prep_hbb64(const double* __restrict__ dets, int stride, const int32_t* __restrict__ order, int n, int allow_reject,
                             float4* __restrict__ rec, u64* __restrict__ alive) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    const double* d = dets + (size_t)order[p] * stride;
    const double x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
    const double area = (x2 - x1 + 1) * (y2 - y1 + 1);
    const bool fin = (x1 - x1 == 0.0) && (y1 - y1 == 0.0) && (x2 - x2 == 0.0) && (y2 - y2 == 0.0);
    float4* r = rec + (size_t)p * HbbGeom64::RECQ;
    const float inf = __builtin_huge_valf();
    r[0] = (allow_reject && fin && area > 0.0 && area - area == 0.0) ? make_float4(f32_down(x1), f32_down(y1), f32_up(x2 + 1), f32_up(y2 + 1))
                                                                    : make_float4(-inf, -inf, inf, inf);
    double2* q = reinterpret_cast<double2*>(r + 1);
    q[0] = make_double2(x1, y1); q[1] = make_double2(x2, y2);
  }
  const u64 m = __ballot(p < n);
  if ((threadIdx.x & 63) == 0 && (p & ~63) < n) alive[p >> 6] = m;
  if (p < 8) alive[((n + 63) >> 6) + p] = 0ull;      // guard words behind the last box (the bitmap is not memset)
}

__global__ void k_seg_from_offsets(const int32_t* __restrict__ seg_off, int nseg, int* seg_begin, int* seg_end, int* keep_cnt) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nseg) return;
  seg_begin[g] = seg_off[g]; seg_end[g] = seg_off[g + 1];
  keep_cnt[g] = 0;
}

// num_keep[g] = -1 when the persistent kernel gave up on a barrier (abort flag): the host layer raises
// feedback != NULL (a long single list of rotated boxes): what the calling thread's NEXT call of this size class chooses its path
// by (mk_choose below) -- the number of kept boxes and whether the list fell apart into independent slabs.
__global__ void k_finalize(const int* __restrict__ keep_cnt, const int* __restrict__ seg_begin, int nseg, long long max_keep,
                           const int* __restrict__ abort_flag, int64_t* __restrict__ num_keep, int64_t* __restrict__ seg_begin_out,
                           int* feedback = nullptr, const SlabPlan* slab_plan = nullptr, int feedback_slab = 0, const u64* tstart = nullptr,
                           const NmsResume* mk_ctl = nullptr, int mk_enqueued = 0) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nseg) return;
  if (seg_begin_out) seg_begin_out[g] = seg_begin[g];
  long long c = keep_cnt[g];
  if (max_keep > 0 && c > max_keep) c = max_keep;
  num_keep[g] = (abort_flag && *abort_flag) ? -1 : c;
  if (feedback != nullptr && g == 0) {
    __hip_atomic_store(feedback + 1, (int)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (only a call that looked for slabs reports on them: the phase-kernel path leaves the word alone)
    if (feedback_slab) __hip_atomic_store(feedback + 2, (slab_plan != nullptr && slab_plan->mode == 1) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // how long the call took on the device (10 ns ticks from its first kernel to this one) and which path it was: [4 + path]
    // (a phase-kernel call that merely ran out of enqueued steps -- a stale step count from other data of this size class -- and was
    //  finished by the persistent kernel says nothing about the phase kernels' speed: its time is not recorded, the thread's next
    //  call enqueues twice the steps (hint -2) and reports then.  A call that bailed out, or had 32 steps, does report.)
    const bool starved = mk_ctl != nullptr && mk_ctl->done == 0 && mk_ctl->bail == 0 && mk_enqueued < 32;
    if (tstart != nullptr && !starved) {
      const u64 dtk = wall_clock64() - *tstart;
      __hip_atomic_store(feedback + 4 + (feedback_slab ? 0 : 1), (int)(dtk > 0x3fffffffull ? 0x3fffffffull : dtk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // ... and on how many kept boxes: a time measured on other data of the size class is not compared (mk_choose)
      __hip_atomic_store(feedback + 6 + (feedback_slab ? 0 : 1), (int)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------- workspace
static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Chunk capacities: the first chunk has OBB_NMS_CHUNK boxes (default 2048), later chunks double up to
// OBB_NMS_CHUNK_MAX (default 8192 for a single list, 2048 per segment for batches: the edge list is sized for
// the worst case cap*(cap-1)/2 per segment).
static int env_int(const char* name, int dflt, int lo, int hi) {
  int v = obb_dev_switch(name, dflt);
  if (v < lo) v = lo;
  if (v > hi) v = hi;
  return (v + 63) / 64 * 64;
}
static int cap_first() { static const int c = env_int("OBB_NMS_CHUNK", 2048, 64, 8192); return c; }   // clipped to cap_max in the kernel
// (the edge list of a team is sized for the worst case cap*(cap-1)/2 of one chunk: 134 MB at 8192, 8.4 MB at 2048,
//  2 MB at 1024 -- times the number of teams)
static int cap_max(int64_t nseg) {
  static const int c1 = env_int("OBB_NMS_CHUNK_MAX", 8192, 64, 16384);
  static const int cb = env_int("OBB_NMS_CHUNK_MAX_BATCHED", 2048, 64, 16384);
  static const int cm = env_int("OBB_NMS_CHUNK_MAX_MANY", 1024, 64, 16384);
  return nseg == 1 ? c1 : (nseg <= 64 ? cb : cm);
}

constexpr int kMaxTeams = 1024;     // >= number of CUs of any gfx950 part

struct Carve {
  unsigned long long *keys_a, *keys_b;
  uint32_t *vals_a, *vals_b;
  void* sort_tmp; size_t sort_tmp_bytes;
  float4* rec; u64* alive; size_t alive_bytes;
  int *seg_begin, *seg_end, *keep_cnt;
  int* bar; size_t bar_bytes;          // team barrier counters, abort flag, per-team nrows / nedges (zeroed before every launch)
  int *abort_flag, *nrows, *nedges;
  u64* prof;                           // in-kernel phase timing (OBB_NMS_PHASE_PROF=1)
  int4* plan;                          // per-workgroup team plan (k_plan_teams)
  uint32_t *rows, *edges;
  long long ecap;
  GridDev grid;                        // spatial index (rotated boxes, single list); grid.meta == NULL: not carved
  // the phase-kernel path of a long single list (nms_mk.h); mk_cidx == NULL: not carved.  Its control block is the head of `bar`.
  int* mk_ctl; uint32_t* mk_cidx; float4* mk_ent; uint16_t* mk_start; u64* mk_kbits; uint4* mk_pend1; uint8_t* mk_hasin; u64* tstart;   // mk_ctl: 1 KB (control block, edge counter at int 64)
  size_t grid_zero_bytes;              // GridMeta + slot counters: one contiguous block, zeroed before every build
  size_t total;
};

// table slots of the spatial index (power of two, multiple of 4096)
// cells of side 2 R_L / 2^fine (grid.h): 1 measured best at 100k (K=3000: 716 -> 648 us, uniform 2361 -> 2138; 2: no further gain)
static int grid_fine() { static const int f = [] { const int v = obb_dev_switch("OBB_GRID_FINE", 1); return (v < 0 || v > 2) ? 1 : v; }(); return f; }
// Skip rules of the quad NMS (piou_device.h): bit 0 = the two exact cone rules (proved), bit 1 = the bounding-box rule (measured noise
// bound).  OBB_NMS_POLY_STRICT=1: cone rule only (every other pair is clipped); =2: no rule at all, every pair is clipped.
static int quad_skip() {
  static const int f = [] { const char* e = getenv("OBB_NMS_POLY_STRICT"); const int v = e ? atoi(e) : 0; return v == 1 ? 1 : (v >= 2 ? 0 : 3); }();
  return f;
}
static uint32_t grid_slots(int64_t n) { return (n >= 262144 || (grid_fine() > 0 && n >= 32768)) ? 65536u : 16384u; }
constexpr int64_t kGridMinN = 8192;    // below this the exhaustive cross phase is cheaper than building the index

// One persistent launch runs the whole step loop (nms_core.h).  Grid: one 512-thread workgroup per CU at most --
// all workgroups must be resident because they meet at team barriers; smaller problems get fewer workgroups
// (cheaper barriers).
static thread_local int g_max_grid = 0;   // obb_nms_set_max_grid: per calling thread (a retry of one thread never shrinks another thread's launches)
static int hw_cu_count();
static int cu_count() {
  const int c = hw_cu_count();
  return (g_max_grid > 0 && g_max_grid < c) ? g_max_grid : c;
}
static int hw_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus < 1) cus = 1;
    if (cus > kMaxTeams) cus = kMaxTeams;
  }
  return cus;
}

// scratch of the sorts: samples + cut table of the three-launch sort (psrs_sort.h), tile histograms of the radix sort (segsort.h)
static size_t sort_tmp_bytes_for(size_t n) {
  return align_up(ps_scratch_bytes()) + (((n + kSrsTile - 1) / kSrsTile) + 1) * 256 * 4;
}

static int carve(void* base, int64_t n, int64_t nseg, int recq, int C, Carve* cv) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return base ? (char*)base + o : (char*)nullptr; };
  size_t nn = (size_t)(n > 0 ? n : 1), ns = (size_t)(nseg > 0 ? nseg : 1);
  cv->keys_a = (unsigned long long*)take(nn * 8); cv->keys_b = (unsigned long long*)take(nn * 8);
  cv->vals_a = (uint32_t*)take(nn * 4); cv->vals_b = (uint32_t*)take(nn * 4);
  cv->sort_tmp_bytes = sort_tmp_bytes_for(nn);
  cv->sort_tmp = take(cv->sort_tmp_bytes);
  cv->rec = (float4*)take(nn * recq * 16);
  cv->alive_bytes = (nn / 64 + 10) * 8;      // + guard words (zeroed by the prep kernels)
  cv->alive = (u64*)take(cv->alive_bytes);
  cv->bar_bytes = ((size_t)kMaxTeams * 128 + 64 + 2 * (size_t)kMaxTeams + 2 * (size_t)kBarGroups * 64) * 4;   // (+ the group counters of the two grid-wide barriers)
  cv->bar = (int*)take(cv->bar_bytes);
  cv->abort_flag = cv->bar + (size_t)kMaxTeams * 128;
  cv->nrows = cv->abort_flag + 64;
  cv->nedges = cv->nrows + kMaxTeams;
  cv->prof = (u64*)take(56 * 8);
  cv->tstart = (u64*)take(64);
  cv->plan = (int4*)take((size_t)kMaxTeams * 16);
  cv->seg_begin = (int*)take(ns * 4); cv->seg_end = (int*)take(ns * 4);
  cv->keep_cnt = (int*)take(ns * 4);
  // scratch that only the segment a team is working on needs: one copy per team, not per segment
  size_t nteams = ns < (size_t)cu_count() ? ns : (size_t)cu_count();
  cv->rows = (uint32_t*)take((nn + 64 * kMaxSlabs) * 4);      // (+ the padding of the slab-major layout)
  cv->ecap = (long long)C * (C - 1) / 2; if (cv->ecap < 1) cv->ecap = 1;
  cv->edges = (uint32_t*)take(nteams * (size_t)cv->ecap * 4);
  cv->grid = GridDev{}; cv->grid_zero_bytes = 0;
  cv->mk_ctl = nullptr; cv->mk_cidx = nullptr; cv->mk_ent = nullptr; cv->mk_start = nullptr; cv->mk_kbits = nullptr; cv->mk_pend1 = nullptr; cv->mk_hasin = nullptr;
  if (recq == RotGeom::RECQ && nseg == 1 && n >= kGridMinN) {
    const uint32_t M = grid_slots(n);
    cv->grid.mask = M - 1;
    const size_t slab_tot_bytes = (size_t)(1 + kMaxTeams / 16) * kMaxSlabs * 4;   // slab totals + group totals (slab_setup)
    const size_t slab_zero = align_up((size_t)kSlabCopies * kSlabWords * 4 + 64) + align_up(slab_tot_bytes);
    cv->grid_zero_bytes = align_up(sizeof(GridMeta)) + slab_zero + ((size_t)M + 4) * 4;
    char* z = take(cv->grid_zero_bytes);
    cv->grid.meta = (GridMeta*)z;
    cv->grid.slab_cover = (uint32_t*)(z ? z + align_up(sizeof(GridMeta)) : nullptr);
    cv->grid.slab_flag = (int*)(z ? z + align_up(sizeof(GridMeta)) + (size_t)kSlabCopies * kSlabWords * 4 : nullptr);
    cv->grid.slab_tot = (int*)(z ? z + align_up(sizeof(GridMeta)) + align_up((size_t)kSlabCopies * kSlabWords * 4 + 64) : nullptr);
    cv->grid.cnt = (int*)(z ? z + align_up(sizeof(GridMeta)) + slab_zero : nullptr);
    cv->grid.start = (int*)take(((size_t)M + 4) * 4);
    cv->grid.wsum = (int*)take((size_t)1024 * 4);
    cv->grid.nparts = (int)((nn + 255) / 256);
    cv->grid.bbpart = (int*)take((size_t)cv->grid.nparts * kBbInts * 4);
    cv->grid.sorted = (float4*)take(nn * 16);
    cv->grid.slot_of = (uint32_t*)take(nn * 4);
    cv->grid.ulist = (uint32_t*)take(nn * 4);
    const size_t n2 = nn + 64 * (size_t)kMaxSlabs;            // every slab starts on a 64-position boundary
    cv->grid.slab_cnt = (int*)take((size_t)kMaxTeams * kMaxSlabs * 4);
    cv->grid.slab_keep = (int*)take((size_t)kMaxSlabs * 4);
    cv->grid.rec2 = (float4*)take(n2 * RotGeom::RECQ * 16);
    cv->grid.order2 = (uint32_t*)take(n2 * 4);
    cv->grid.pos_old = (uint32_t*)take(n2 * 4);
    cv->grid.alive2_words = n2 / 64 + 10;
    cv->grid.alive2 = (u64*)take(cv->grid.alive2_words * 8);
    cv->grid.kept_words = nn / 64 + 2;
    cv->grid.kept_bits = (u64*)take(cv->grid.kept_words * 8);
    cv->grid.slab_plan = (SlabPlan*)take(sizeof(SlabPlan));
    cv->mk_ctl = (int*)take(1024);
    cv->mk_cidx = (uint32_t*)take((size_t)kMkCapMax * 4);
    cv->mk_ent = (float4*)take((size_t)kMkCapMax * 16);
    cv->mk_start = (uint16_t*)take((size_t)(kMkSlots + 8) * 2);
    cv->mk_kbits = (u64*)take((size_t)(kMkCapMax / 64) * 8);
    cv->mk_hasin = (uint8_t*)take((size_t)kMkCapMax);
    cv->mk_pend1 = (uint4*)take((size_t)kMkPend1 * 16);
  }
  cv->total = off;
  return OBB_OK;
}

// window of positions opened at a time when the caller limits the number of kept boxes (nms_core.h)
static int nms_window(long long max_keep) {
  if (max_keep <= 0) return 0;
  long long w = 4 * max_keep;
  if (w < 8192) w = 8192;
  if (w > (1 << 30)) w = 1 << 30;
  return (int)((w + 63) / 64 * 64);
}

constexpr size_t kPersistLdsMax = 159 * 1024;   // of the 160 KB per CU: exactly one workgroup per CU

template <class G, bool GRID>
static int launch_persist(const NmsArgs& a, unsigned nb, hipStream_t st) {   // nb <= number of CUs (nms_grid); static teams and plans are made for that grid
  static OncePerDevice attr;
  if (const int attr_dev = attr.need(); attr_dev != OncePerDevice::kDone) {
    if (hipFuncSetAttribute((const void*)k_nms_persist<G, GRID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersistLdsMax) != hipSuccess)
      return OBB_ERR_LAUNCH;
    attr.mark(attr_dev);
  }
  // every workgroup of the launch must be resident at once (team barriers): the grid is bounded by what the occupancy
  // calculation gives for this instantiation with its largest LDS footprint -- asked once, not assumed
  static int wg_per_cu = -1;
  if (wg_per_cu < 0) {
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)k_nms_persist<G, GRID>, kNmsThreads, kPersistLdsMax) != hipSuccess) return OBB_ERR_LAUNCH;
    wg_per_cu = nblk;
  }
  if (wg_per_cu < 1) return OBB_ERR_LAUNCH;
  { const long long resident = (long long)wg_per_cu * hw_cu_count(); if ((long long)nb > resident) nb = (unsigned)resident; }
  size_t lds = sizeof(WaveLds<G>) * kNmsWaves;                // pair phases: one scratch block per wave
  if (lds < (size_t)6 * a.capmax) return OBB_ERR_INTERNAL;    // (aliased by resolve: state + blocked bytes of one chunk + its ordered output list)
  lds += (size_t)a.capmax * 4;                                // + this workgroup's copy of the chunk list
  if (lds > kPersistLdsMax) return OBB_ERR_INTERNAL;
  // (a cooperative launch -- the runtime guarantees residency instead of the kernel finding out by a barrier time-out -- was
  //  measured in round 3: +18 us per call; the abort + retry with 8 workgroups stays the answer to a grid that is not resident)
  k_nms_persist<G, GRID><<<nb, kNmsThreads, lds, st>>>(a);
  return OBB_OK;
}

// grid of the persistent launch.  n_slots: number of sorted positions that can hold a box (bounds the useful parallelism)
static int nms_grid(int64_t nseg, int64_t n_slots, int cap_first_) {
  const int cus = cu_count();
  int64_t nb = (n_slots + 511) / 512;                 // 8 waves x 64 columns per workgroup
  {                                                   // ... and one wave per tile of the first chunk's triangle
    const int64_t per_seg = (n_slots + nseg - 1) / nseg;
    const int64_t c = per_seg < cap_first_ ? per_seg : cap_first_;
    const int64_t tiles = ((c + 63) / 64) * ((c + 63) / 64 + 1) / 2 * nseg;
    if (nb < (tiles + 7) / 8) nb = (tiles + 7) / 8;
  }
  if (n_slots >= 8192) nb = cus;                      // enough work for the whole chip
  if (nb < nseg) nb = nseg;
  if (nb > cus) nb = cus;
  if (nb < 1) nb = 1;
  return (int)nb;
}

constexpr int kNmsBarZeroed = 1;    // the caller has zeroed the barrier block on this stream already
constexpr int kNmsPlanned = 2;      // the team plan for this launch has been written already (fused sort/prep kernel)
static int nms_steps(int kind, NmsArgs& a, const Carve& cv, int64_t nseg, int64_t n_slots, hipStream_t st, int pre = 0) {
  if (!(pre & kNmsBarZeroed) && hipMemsetAsync(cv.bar, 0, cv.bar_bytes, st) != hipSuccess) return OBB_ERR_LAUNCH;
  a.bar = cv.bar; a.abort_flag = cv.abort_flag; a.nseg = (int)nseg;
  a.bar_sub = cv.nedges + kMaxTeams;                 // group counters of the grid-wide barrier, behind the per-team words
  a.cap_first = cap_first();
  { static const int grow = [] { const int v = obb_dev_switch("OBB_NMS_GROW", 2); return (v < 2 || v > 8) ? 2 : v; }(); a.grow_sparse = grow; }
  { static const int lpt = obb_dev_switch("OBB_NMS_LPT", 1) & 1; a.lpt = lpt; }      // A/B switch (development builds): rows of a chunk largest first
  static const int phase_prof = [] { const char* e = getenv("OBB_NMS_PHASE_PROF"); return (e && atoi(e)) ? 1 : 0; }();
  a.prof = nullptr;
  if (phase_prof && a.resume == nullptr) {   // development aid: print the previous call's phase times (synchronises!)
    u64 h[56];
    if (hipMemcpy(h, cv.prof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[6] > 0 && h[6] < (1ull << 40)) {
      fprintf(stderr, "[nms phases, wg0, us] select %.1f pairs %.1f wait-resolve %.1f cross %.1f barrier %.1f steps %llu | resolve (any wg) %.1f rounds %llu [first round %.1f other rounds %.1f output %.1f]\n",
              h[1] * 0.01, h[2] * 0.01, h[3] * 0.01, h[5] * 0.01, h[0] * 0.01, h[6], h[9] * 0.01, h[11], h[12] * 0.01, h[13] * 0.01, h[14] * 0.01);
      fprintf(stderr, "    resolve: edges total %llu, max per chunk %llu, chunk sizes total %llu, chunks resolved from LDS %llu\n", h[25], h[27], h[28], h[26]);
      fprintf(stderr, "    pairs phase per workgroup (wave 0): longest %.1f us, sum over steps and workgroups / workgroups %.1f us; cross: longest single %.1f us\n", h[29] * 0.01, h[24] ? h[30] * 0.01 / h[24] : 0.0, h[31] * 0.01);
      fprintf(stderr, "    workgroups: %llu, busy time max %.1f us, mean %.1f us; cross phases: mean over workgroups %.1f us (sum over steps; wg0: cross + barrier = the slowest workgroup of every step)\n",
              h[24], h[22] * 0.01, h[24] ? h[23] * 0.01 / h[24] : 0.0, h[24] ? h[51] * 0.01 / h[24] : 0.0);
      fprintf(stderr, "    cross, wave 0: items %llu row-loops / scans %.1f us, stage-1 drains %llu = %.1f us | exhaustive: stage-2 drains %llu = %.1f us | indexed: per-item setup %.1f us, blocks %llu, queued pairs %llu, parts (sum over steps) %llu\n",
              h[21], h[16] * 0.01, h[19], h[17] * 0.01, h[20], h[18] * 0.01, h[18] * 0.01, h[20], h[15], h[10]);
      fprintf(stderr, "    pairs, wg0 wave 0: items %llu = %.1f us (loads %.1f), stage-1a drains %llu = %.1f us, stage-1b drains %llu = %.1f us, exact drains %llu = %.1f us | slab set-up %.1f us, merge %.1f us (wg0)\n",
              h[32], h[33] * 0.01, h[40] * 0.01, h[34], h[35] * 0.01, h[36], h[37] * 0.01, h[38], h[39] * 0.01, h[41] * 0.01, h[8] * 0.01);
      if (h[43]) fprintf(stderr, "    slab set-up (wg0): runs %.1f, count %.1f, barrier %.1f, table %.1f (copy %.1f, sums %.1f, decision %.1f), scatter %.1f + plan %.1f, barrier %.1f us\n",
                         h[42] * 0.01, h[43] * 0.01, h[44] * 0.01, (h[48] + h[49] + h[45]) * 0.01, h[48] * 0.01, h[49] * 0.01, h[45] * 0.01, h[50] * 0.01,
                         h[46] * 0.01, h[47] * 0.01);
    }
    if (hipMemsetAsync(cv.prof, 0, 56 * 8, st) != hipSuccess) return OBB_ERR_LAUNCH;
    a.prof = cv.prof;
  }
  const int nb = nms_grid(nseg, n_slots, a.cap_first);
  a.plan = nullptr;
  if (nseg > 1) {                                      // workgroups in proportion to the (non-empty) segments' sizes
    int c1 = a.cap_first < a.capmax ? a.cap_first : a.capmax;
    if (!(pre & kNmsPlanned)) k_plan_teams<<<1, 1024, 0, st>>>(a.seg_begin, a.seg_end, (int)nseg, (int)nb, c1, cv.plan);
    a.plan = cv.plan;
  }
  if (kind == 0 && a.gmeta != nullptr && a.slab_cover != nullptr && a.slab_plan != nullptr && nseg == 1 && a.max_keep <= 0 && a.cull != 0 &&
      a.bbpart != nullptr) {
    // independent slabs (grid.h): the set-up is a launch of its own on the persistent kernel's grid (nms_core.h: k_slab_split)
    k_slab_split<RotGeom><<<(unsigned)nb, kNmsThreads, 0, st>>>(a, const_cast<SlabPlan*>(a.slab_plan));
  } else {
    a.slab_plan = nullptr;
  }
  if (kind == 5) return launch_persist<HbbGeom64, false>(a, (unsigned)nb, st);
  if (kind == 4) return launch_persist<QuadGeom64All, false>(a, (unsigned)nb, st);
  if (kind == 3) return launch_persist<RotGeom64, false>(a, (unsigned)nb, st);
  if (kind == 2) return launch_persist<QuadGeom64, false>(a, (unsigned)nb, st);
  if (kind == 1) return launch_persist<QuadGeom, false>(a, (unsigned)nb, st);
  return a.gmeta != nullptr ? launch_persist<RotGeom, true>(a, (unsigned)nb, st) : launch_persist<RotGeom, false>(a, (unsigned)nb, st);
}

// ---------------------------------------------------------------- the phase-kernel path of a long single list (nms_mk.h)
// Steps are enqueued without knowing how many the data needs (stream-ordered: nothing is read back).  Every kernel of a step
// returns at once when the control block says the call is complete, and whatever the enqueued steps leave undone is finished by
// the persistent kernel launched behind them (NmsResume) -- which returns at once when nothing is left.  The device records the
// number of steps a call needed in a pinned word of the calling thread (one per size class); the thread's next call of that size
// enqueues that many.
constexpr int64_t kMkMinN = 16384;
constexpr int kMkMinKept = 256;    // below this many kept boxes (a handful of dense clusters: huge conflict lists per chunk) the phase kernels are not even tried
static int mk_enabled() { const char* e = getenv("OBB_NMS_MK"); return e ? atoi(e) : 2; }   // 0: never, 1: always, 2: by feedback (read per call: tests switch paths in one process)
// Per calling thread and size class (floor(log2 n)): four pinned words the DEVICE writes when a call completes -- [0] the steps
// the phase-kernel path needed, [1] the boxes the call kept, [2] whether the persistent kernel found independent slabs -- and
// the host reads, without synchronising, when the thread's next call of that size is set up.  Nothing but the choice of path and
// the number of enqueued steps depends on them; the result of a call does not.
struct MkFeedback { int* words; unsigned calls; };
static MkFeedback* mk_feedback(int64_t n) {
  static thread_local int* base = nullptr;
  static thread_local MkFeedback fb[64];
  if (!base) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64 * 8 * sizeof(int), hipHostMallocPortable) != hipSuccess) return nullptr;
    base = (int*)p;
    for (int i = 0; i < 64 * 8; i++) base[i] = -1;
    for (int i = 0; i < 64; i++) { fb[i].words = base + 8 * i; fb[i].calls = 0; }
  }
  int b = 0;
  while ((n >> b) > 1 && b < 63) b++;
  return &fb[b];
}
// Which path the next call takes.  The phase kernels win where a chunk keeps many rows (thousands of objects, sparse data); the
// persistent kernel wins where two or three steps with a few hundred kept rows each do it, and where the list falls apart into
// independent slabs (class offsets: one team per slab steps concurrently).  Both report what the rule needs; the first call of a
// size class, and every 64th after it, takes the persistent kernel (it is the one that can see slabs).
// Round 6: the rule above only says where the phase kernels are WORTH TRYING.  Both paths report how long the call took on the device
// ([4] persistent kernel, [5] phase kernels: 10 ns ticks between the call's first kernel and its k_finalize), and where both are
// known the faster one is taken (S-clustered K=3000 + 18 class offsets keeps 40,000 boxes, finds no slabs -- and still runs 0.58 ms
// on the persistent kernel against 0.67 ms here); the other one is measured again every 64th call, the data may have changed.
static bool mk_choose(MkFeedback* f) {
  const int mode = mk_enabled();
  if (mode == 0) return false;
  if (mode == 1 || f == nullptr) return mode == 1;
  const unsigned k = f->calls++;
  const int kept = *(volatile int*)(f->words + 1), slab = *(volatile int*)(f->words + 2);
  if (kept < 0 || (k & 63u) == 0u) return false;
  if (!(slab != 1 && kept >= kMkMinKept)) return false;
  const int t_persist = *(volatile int*)(f->words + 4), t_mk = *(volatile int*)(f->words + 5);
  // a time only counts for data like the data it was measured on: the kept count it came with ([6], [7]) within a quarter of the
  // previous call's.  (bench.py switches regimes of one size class every 24 calls: K=3000 + 18 class offsets inherited "phase
  // kernels are faster" from K=3000 and stayed on them at 0.62 ms against the persistent kernel's 0.51 until the 64th call.)
  const int k_persist = *(volatile int*)(f->words + 6), k_mk = *(volatile int*)(f->words + 7);
  auto stale = [&](int kk) { const int d = kk > kept ? kk - kept : kept - kk; return kk < 0 || d > kept / 4 + 16; };
  if (t_mk <= 0 || stale(k_mk)) return true;           // (the phase kernels have not reported on such data yet: try them)
  if (t_persist <= 0 || stale(k_persist)) return false;
  if ((k & 63u) == 32u) return t_mk >= t_persist;      // the slower one's turn
  return t_mk < t_persist;
}
// Which cross probe: the table of the kept rows in every workgroup's LDS (k_mk_cross_lds) where a step keeps a few thousand rows
// at most -- the previous call of the size class kept <= 2 passes' worth in all -- the chunk's table in global memory otherwise
// (S-uniform keeps 36,000 of 100,000: six passes over the queries would cost more than the L2 round trips they save).
// OBB_NMS_MK_XLDS = 0 / 1 pins the choice (read per call: tests run both on the same data).
static bool mk_cross_in_lds(int kept_prev) {
  const char* e = getenv("OBB_NMS_MK_XLDS");
  if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
  return kept_prev >= 0 && kept_prev <= 2 * kMkXRows;
}
static int mk_steps(MkArgs& a, int kept_prev, hipStream_t st) {
  static OncePerDevice attr;
  static_assert(sizeof(MkLdsSelect) <= kMkSerialLds && (size_t)RotGeom::SCR * 64 * 4 * kMkWaves <= kMkSerialLds, "serial-phase LDS");
  if (const int attr_dev = attr.need(); attr_dev != OncePerDevice::kDone) {
    if (hipFuncSetAttribute((const void*)k_mk_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMkSerialLds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mk_decide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMkSerialLds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mk_decide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMkSerialLds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_mk_cross_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MkLdsCross)) != hipSuccess)
      return OBB_ERR_LAUNCH;
    attr.mark(attr_dev);
  }
  static const int fixed_steps = [] { const char* e = getenv("OBB_NMS_MK_STEPS"); return e ? atoi(e) : 0; }();   // (measurement aid)
  // (-2: the previous call of this size class was finished by the persistent kernel behind its steps: twice as many this time)
  int steps = 4;
  if (a.hint_host) {
    const int h = *(volatile int*)a.hint_host;
    if (h >= 0) steps = h;
    else if (h == -2) { const int last = *(volatile int*)(a.hint_host + 3); steps = last > 0 ? 2 * last : 8; }
  }
  if (fixed_steps > 0) steps = fixed_steps;
  if (steps > 32) steps = 32;
  if (steps < 1) steps = 1;
  if (a.hint_host) { *(volatile int*)(a.hint_host + 3) = steps; }
  const unsigned cus = (unsigned)hw_cu_count();
  const unsigned gp = 4 * cus;                         // (k_mk_probe: four 256-thread workgroups per CU)
  const bool xlds = mk_cross_in_lds(kept_prev);
  k_mk_select<<<1, kMkThreads, kMkSerialLds, st>>>(a);
  for (int s = 0; s < steps; s++) {
    k_mk_probe<false><<<gp, kMkProbeThreads, 0, st>>>(a);
    k_mk_decide<false><<<cus, kMkThreads, kMkSerialLds, st>>>(a);      // ... + resolve in its last workgroup
    // (the step that completes a call has nothing behind its chunk: no cross half for the last enqueued step -- two empty launches,
    //  ~10 us; a call that needs more steps than were enqueued goes on in the persistent kernel with this cross phase, NmsResume stage 3)
    if (s == steps - 1) break;
    if (xlds) k_mk_cross_lds<<<2 * cus, kMkXThreads, sizeof(MkLdsCross), st>>>(a);
    else k_mk_probe<true><<<gp, kMkProbeThreads, 0, st>>>(a);
    k_mk_decide<true><<<cus, kMkThreads, kMkSerialLds, st>>>(a);       // ... + the next select in its last workgroup
  }
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

// One list sorted by descending score, ties by ascending index (nms_rotated_cuda.cu:81-82 leaves the tie order to an
// unstable sort; this is the documented rule here): sorted keys in cv.keys_b, order in cv.vals_b.  Up to kPsMaxN elements:
// three launches (psrs_sort.h); longer lists: key kernel + LSD radix sort over the four score bytes (segsort.h).
// (Writing the NMS records in the tail of the sort's last kernel instead of a launch of their own was built and measured in
//  round 4: the bucket kernel went from 16.6 to 34 us -- one workgroup per CU because of its 104 KB of LDS, the double-precision
//  sin / cos and the gathers of the boxes at that occupancy -- against 8 us for k_prep_rot: 9 us slower per call.  Not kept.)
static int sort_single_list(const float* scores, int score_stride, const float* dets5, int drop_small, int64_t n, const Carve& cv,
                            const LocalExtras& x, int* nparts_out, hipStream_t st) {
  if (n <= kPsMaxN) {
    PsBuf b{};
    b.run_k = cv.keys_a; b.run_v = cv.vals_a; b.out_k = cv.keys_b; b.out_v = cv.vals_b; b.n = (int)n; b.n_dev = nullptr; b.err = nullptr;
    ps_carve_scratch(cv.sort_tmp, &b);
    const int runs = (int)((n + kPsRun - 1) / kPsRun);
    k_ps_local_scores<<<(unsigned)runs, kPsRun, 0, st>>>(b, scores, score_stride, dets5, drop_small, x);
    *nparts_out = runs;
    return ps_finish(b, runs, st);
  }
  const unsigned gb = (unsigned)((n + kPsRun - 1) / kPsRun);
  k_make_keys_wide<<<gb, kPsRun, 0, st>>>(scores, score_stride, dets5, drop_small, (int)n, cv.keys_a, cv.vals_a, x);
  *nparts_out = (int)gb;
  uint32_t* hist = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(cv.sort_tmp) + align_up(ps_scratch_bytes()));
  return seg_radix_sort_large(cv.keys_a, cv.keys_b, cv.vals_a, cv.vals_b, cv.seg_begin, cv.seg_end, 1, n, n, 0xF0u, hist, st);
}

// kind: 0 rotated (5 floats + score array), 1 quad (rows of `stride` floats, score in column 8)
static int run_nms(int kind, const float* boxes, int stride, const float* scores, int score_stride, int64_t n, float thr, int flags,
                   int64_t max_keep, int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, hipStream_t st) {
  const int64_t nseg = 1;
  if (n < 0 || n > 0x7fffffffLL) return OBB_ERR_BAD_ARG;
  if (!num_keep || (n > 0 && (!boxes || !scores || !keep_out))) return OBB_ERR_BAD_ARG;
  const int C = cap_max(nseg);
  const int recq = kind == 0 ? RotGeom::RECQ : QuadGeom::RECQ;
  Carve cv;
  int rc = carve(ws, n, nseg, recq, C, &cv);
  if (rc) return rc;
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  const int T = 256;
  const int drop_small = (flags & OBB_NMS_DROP_SMALL) ? 1 : 0;
  const unsigned gseg = 1;

  if (n == 0) {
    if (hipMemsetAsync(cv.keep_cnt, 0, nseg * 4, st) != hipSuccess) return OBB_ERR_LAUNCH;
    if (hipMemsetAsync(cv.seg_begin, 0, nseg * 4, st) != hipSuccess) return OBB_ERR_LAUNCH;
    k_finalize<<<gseg, T, 0, st>>>(cv.keep_cnt, cv.seg_begin, (int)nseg, max_keep, nullptr, num_keep, nullptr);
    return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
  }

  const unsigned gb = (unsigned)((n + T - 1) / T);
  int pre = 0;
  // spatial index for the cross phases (grid.h): rotated boxes, one list, conservative rejects allowed (thr >= 0)
  static const int no_grid = obb_dev_switch("OBB_NMS_NO_GRID", 0) != 0;          // A/B switch (development builds)
  const bool use_grid = !no_grid && kind == 0 && cv.grid.meta != nullptr && thr >= 0.f && n < (1ll << 24);
  static const int no_slabs = obb_dev_switch("OBB_NMS_NO_SLABS", 0) != 0;        // A/B switch (development builds)
  // long lists without a limit on the kept boxes: the phase-kernel path (nms_mk.h) -- no index of all boxes, no slab decomposition
  MkFeedback* const fbk = (use_grid && cv.mk_cidx != nullptr && n >= kMkMinN && max_keep <= 0) ? mk_feedback(n) : nullptr;
  const bool use_mk = use_grid && cv.mk_cidx != nullptr && n >= kMkMinN && max_keep <= 0 && mk_choose(fbk);
  const bool use_slabs = use_grid && !use_mk && !no_slabs && max_keep <= 0;   // (a limit on the kept boxes keeps the call one list: the windows are per list)
  {
    ProfScope ps(PROF_NMS_SORT, st);
    LocalExtras x{};
    x.seg_begin = cv.seg_begin; x.seg_end = cv.seg_end; x.keep_cnt = cv.keep_cnt;
    x.bar16 = reinterpret_cast<uint4*>(cv.bar); x.n_bar16 = (long long)(cv.bar_bytes / 16);
    x.grid16 = reinterpret_cast<uint4*>(cv.grid.meta); x.n_grid16 = use_grid ? (long long)(cv.grid_zero_bytes / 16) : 0ll;
    x.mk16 = reinterpret_cast<uint4*>(cv.mk_ctl); x.n_mk16 = use_mk ? 64 : 0;
    x.bbpart = (use_grid && kind == 0) ? cv.grid.bbpart : nullptr;
    x.tstart = fbk ? cv.tstart : nullptr;
    rc = sort_single_list(scores, score_stride, kind == 0 ? boxes : nullptr, kind == 0 ? drop_small : 0, n, cv, x, &cv.grid.nparts, st);
    if (rc) return rc;
    pre = kNmsBarZeroed;
  }
  {
    ProfScope ps(PROF_NMS_PREP, st);
    if (kind == 0) k_prep_rot<<<gb, T, 0, st>>>(boxes, cv.vals_b, drop_small, (int)n, cv.rec, cv.alive, cv.grid.bbpart, cv.grid.nparts,
                                                use_slabs ? cv.grid.slab_cover : nullptr, cv.grid.slab_flag);
    else k_prep_quad<<<gb, T, 0, st>>>(boxes, stride, cv.vals_b, (int)n, thr, quad_skip(), cv.rec, cv.alive);
  }

  MkArgs m{};
  if (use_mk) {
    static_assert(sizeof(MkCtl) <= 256, "the control block and its counters share one zeroed KB");
    m.rec = cv.rec; m.order = cv.vals_b; m.alive = cv.alive; m.n = (int)n;
    m.ctl = reinterpret_cast<MkCtl*>(cv.mk_ctl);
    m.cidx = cv.mk_cidx; m.ent = cv.mk_ent; m.start = cv.mk_start; m.kbits = cv.mk_kbits; m.hasin = cv.mk_hasin;
    m.edges = cv.edges; m.nedges = cv.mk_ctl + 64; m.ecap = cv.ecap;
    m.rows = cv.rows; m.nrows = cv.mk_ctl + 128; m.keep_cnt = cv.keep_cnt; m.keep_out = keep_out;
    m.bbpart = cv.grid.bbpart; m.nparts = cv.grid.nparts;
    // first chunk: 2048, or 4096 when the previous call of the size class kept more than an eighth of its boxes (sparse data: few
    // conflicts inside a chunk, the steps are what costs -- measured at 100k: S-uniform 1.05 -> 0.94 ms, S-clustered K=3000 0.54 -> 0.58).
    // (the edge list holds the worst case of kMkTile = C members; a larger chunk that outgrows it hands over to the persistent kernel)
    static const int mk_cap_env = [] { const char* e = getenv("OBB_NMS_MK_CHUNK"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > kMkCapMax ? kMkCapMax : v); }();   // (measurement aid)
    const int kept_prev = fbk ? *(volatile int*)(fbk->words + 1) : -1;
    m.capmax = kMkCapMax; m.cap_first = mk_cap_env >= 64 ? mk_cap_env : ((kept_prev >= 0 && (int64_t)kept_prev * 8 > n) ? 4096 : 2048);
    static_assert(kMkTile == 8192, "cap_max(1)");
    m.thr = thr;
    // (OBB_NMS_MK_PEND: a smaller pending list, so that tests reach the overflow hand-over without two million undecided pairs)
    // The pending pairs go straight to the exact clip: nothing compacts a wave between the interval and the clip there, so a wave
    // ran the clip unless the interval had decided all 64 of its pairs -- the interval stage was 10-17 us of every call at 100k (K=300
    // 0.287 -> 0.277 ms, K=3000 0.446 -> 0.433, uniform 0.854 -> 0.837 on one box).  OBB_NMS_MK_NOFULL=0 puts it back (read per call).
    m.skip_full = [] { const char* e = getenv("OBB_NMS_MK_NOFULL"); return (e && *e == '0') ? 0 : 1; }();
    static const int mk_pend_cap = [] { const char* e = getenv("OBB_NMS_MK_PEND"); const int v = e ? atoi(e) : 0; return (v > 0 && v < kMkPend1) ? v : kMkPend1; }();
    m.pend1 = cv.mk_pend1; m.cap1 = mk_pend_cap; m.num_keep = nullptr;    // (k_finalize below writes the count)
    m.hint_host = fbk ? fbk->words : nullptr;
    static const int mk_prof = [] { const char* e = getenv("OBB_NMS_PHASE_PROF"); return (e && atoi(e)) ? 2 : 0; }();
    if (mk_prof) {   // development aid: print the previous call's serial-phase times (synchronises!)
      u64 h[56];
      if (hipMemcpy(h, cv.prof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[37] > 0 && h[37] < (1ull << 20)) {
        fprintf(stderr, "[mk phases, us, sums over %llu steps] pairs: decide+ticket %.1f (pending %llu) resolve %.1f [first round %.1f other rounds %.1f output %.1f; rounds %llu] | "
                        "cross: decide+ticket %.1f (pending %llu) select %.1f chunk-table %.1f | edges %llu chunk members %llu kept %llu\n",
                h[37], h[32] * 0.01, h[42], h[33] * 0.01, h[12] * 0.01, h[13] * 0.01, h[14] * 0.01, h[11], h[41] * 0.01, h[43], h[35] * 0.01, h[36] * 0.01, h[38], h[39], h[40]);
        fprintf(stderr, "    chunk table: loads %.1f statistics %.1f counts %.1f scan %.1f scatter %.1f\n", h[44] * 0.01, h[45] * 0.01, h[46] * 0.01, h[47] * 0.01, h[48] * 0.01);
        if (h[53]) fprintf(stderr, "    cross probe in LDS, per workgroup (%llu workgroup-steps): table %.1f (longest %.1f) wave 0's items %.1f (longest %.1f) whole %.1f (longest %.1f)\n",
                           h[53], h[49] * 0.01 / h[53], h[54] * 0.01, h[50] * 0.01 / h[53], h[55] * 0.01, h[51] * 0.01 / h[53], h[52] * 0.01);
      }
      if (hipMemsetAsync(cv.prof, 0, 56 * 8, st) != hipSuccess) return OBB_ERR_LAUNCH;
      m.prof = cv.prof;
    }
  }

  NmsArgs a{};
  if (use_grid) {
    a.gmeta = cv.grid.meta; a.bbpart = cv.grid.bbpart; a.nparts = cv.grid.nparts; a.gcnt = cv.grid.cnt; a.gstart = cv.grid.start;
    a.gsorted = cv.grid.sorted; a.gslot = cv.grid.slot_of; a.gwsum = cv.grid.wsum; a.ulist = cv.grid.ulist; a.gmask = cv.grid.mask; a.gfine = grid_fine();
  }
  if (use_slabs) {
    a.slab_cover = cv.grid.slab_cover; a.slab_flag = cv.grid.slab_flag; a.slab_cnt = cv.grid.slab_cnt; a.slab_tot = cv.grid.slab_tot; a.slab_keep = cv.grid.slab_keep;
    a.rec2 = cv.grid.rec2; a.order2 = cv.grid.order2; a.pos_old = cv.grid.pos_old; a.alive2 = cv.grid.alive2; a.kept_bits = cv.grid.kept_bits;
    a.alive2_words = (int)cv.grid.alive2_words; a.kept_words = (int)cv.grid.kept_words;
    a.slab_plan = cv.grid.slab_plan;

    // chunk capacity of a slab team: 1024 measured best at 100k / 18 slabs (512: 363 us, 1024: 330, 1536: 362, 1920: 383 --
    // a team has ~14 workgroups: the pair phase grows with the square of the chunk, smaller chunks add steps)
    // Round 6: twice that when the previous call of the size class kept more than an eighth of its boxes (thousands of objects per
    // slab: few conflicts inside a chunk, the steps are what costs) -- S-clustered K=3000 + 18 class offsets (40,000 of 100,000 kept)
    // 0.579 -> 0.506 ms, while K=300 + 18 offsets (6,900 kept) would lose 0.05 ms with it.
    static const int slab_cap = [] { const int v = obb_dev_switch("OBB_NMS_SLAB_CAP", 0); return v < 0 ? 0 : v; }();
    const int kept_before = fbk ? *(volatile int*)(fbk->words + 1) : -1;
    const int slab_auto = (kept_before >= 0 && (int64_t)kept_before * 8 > n) ? 2048 : 1024;
    a.slab_cap = ((slab_cap ? slab_cap : slab_auto) + 63) / 64 * 64;
  }
  a.rec = cv.rec; a.order = cv.vals_b; a.alive = cv.alive; a.seg_begin = cv.seg_begin; a.seg_end = cv.seg_end;
  a.keep_cnt = cv.keep_cnt; a.keep_out = keep_out;
  a.rows = cv.rows; a.nrows = cv.nrows; a.edges = cv.edges; a.nedges = cv.nedges;
  a.ecap = cv.ecap; a.n = (int)n; a.capmax = C;
  a.max_keep = (int)(max_keep > 0x7fffffffLL ? 0x7fffffffLL : (max_keep < 0 ? 0 : max_keep));
  a.window = nms_window(a.max_keep);
  a.thr = thr; a.thr64 = thr;
  a.cull = (thr >= 0.f) ? 1 : 0;      // rejects predict IoU <= 0 or IoU <= thr; with thr < 0 even IoU == 0 suppresses

  {
    ProfScope ps(PROF_NMS_STEPS, st);
    if (use_mk) {
      rc = mk_steps(m, fbk ? *(volatile int*)(fbk->words + 1) : -1, st);
      if (rc) return rc;
      a.resume = reinterpret_cast<const NmsResume*>(cv.mk_ctl);   // the persistent kernel behind them: returns at once when they completed the call
    }
    rc = nms_steps(kind, a, cv, nseg, n, st, pre);
    if (rc) return rc;
  }
  k_finalize<<<gseg, T, 0, st>>>(cv.keep_cnt, cv.seg_begin, (int)nseg, max_keep, cv.abort_flag, num_keep, nullptr, fbk ? fbk->words : nullptr,
                                 a.slab_plan, use_mk ? 0 : 1, fbk ? cv.tstart : nullptr,
                                 use_mk ? reinterpret_cast<const NmsResume*>(cv.mk_ctl) : nullptr, (use_mk && fbk) ? *(volatile int*)(fbk->words + 3) : 0);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

static int kind_recq(int kind) {
  return kind == 0 ? RotGeom::RECQ : (kind == 1 ? QuadGeom::RECQ : (kind == 2 || kind == 4 ? QuadGeom64::RECQ : (kind == 5 ? HbbGeom64::RECQ : RotGeom64::RECQ)));
}

// Tile -> full-image merge NMS: nseg independent lists, the caller fixes the processing order (numpy's argsort()[::-1] of
// the reference is not a stable sort: its tie order is the host's business).  keep_out receives ROW indices, the kept rows of
// segment g at keep_out[seg_off[g] ...], in processing order.
// kind 2: py_cpu_nms_poly_fast (rows of 9 doubles), 4: py_cpu_nms_poly (the same rows, no horizontal-box gate), 5: py_cpu_nms
// (horizontal boxes in columns 0..3 of rows of `stride` doubles).
static int run_merge_nms(int kind, const double* dets9, int stride, int64_t n, const int32_t* order, const int32_t* seg_off, int64_t nseg,
                         double thr, int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, hipStream_t st) {
  if (n < 0 || nseg < 1 || n > 0x7fffffffLL || !num_keep || !seg_off) return OBB_ERR_BAD_ARG;
  if (n > 0 && (!dets9 || !order || !keep_out)) return OBB_ERR_BAD_ARG;
  if (kind == 5 && stride < 4) return OBB_ERR_BAD_ARG;
  const int C = cap_max(nseg);
  Carve cv;
  int rc = carve(ws, n, nseg, kind_recq(kind), C, &cv);
  if (rc) return rc;
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  const int T = 256;
  const unsigned gseg = (unsigned)((nseg + T - 1) / T);
  k_seg_from_offsets<<<gseg, T, 0, st>>>(seg_off, (int)nseg, cv.seg_begin, cv.seg_end, cv.keep_cnt);
  if (n == 0) {
    k_finalize<<<gseg, T, 0, st>>>(cv.keep_cnt, cv.seg_begin, (int)nseg, 0, nullptr, num_keep, nullptr);
    return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
  }
  if (kind == 5) k_prep_hbb64<<<(unsigned)((n + T - 1) / T), T, 0, st>>>(dets9, stride, order, (int)n, (thr >= 0.0) ? 1 : 0, cv.rec, cv.alive);
  else k_prep_quad64<<<(unsigned)((n + T - 1) / T), T, 0, st>>>(dets9, order, (int)n, (thr >= 0.0) ? 1 : 0, cv.rec, cv.alive);
  NmsArgs a{};
  a.rec = cv.rec; a.order = reinterpret_cast<const uint32_t*>(order); a.alive = cv.alive;
  a.seg_begin = cv.seg_begin; a.seg_end = cv.seg_end; a.keep_cnt = cv.keep_cnt; a.keep_out = keep_out;
  a.rows = cv.rows; a.nrows = cv.nrows; a.edges = cv.edges; a.nedges = cv.nedges;
  a.ecap = cv.ecap; a.n = (int)n; a.capmax = C; a.max_keep = 0; a.window = 0;
  a.thr = (float)thr; a.thr64 = thr; a.cull = 1;
  rc = nms_steps(kind, a, cv, nseg, n, st);
  if (rc) return rc;
  k_finalize<<<gseg, T, 0, st>>>(cv.keep_cnt, cv.seg_begin, (int)nseg, 0, cv.abort_flag, num_keep, nullptr);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

// float64 rotated NMS (one list): the reference's double instantiation (nms_rotated_cuda.cu:96), RotGeom64
static int run_nms_rot64(const double* dets5, const double* scores, int64_t n, float thr, int flags, int64_t max_keep, int64_t* keep_out,
                         int64_t* num_keep, void* ws, size_t ws_bytes, hipStream_t st) {
  if (n < 0 || n > 0x7fffffffLL || !num_keep || (n > 0 && (!dets5 || !scores || !keep_out))) return OBB_ERR_BAD_ARG;
  const int C = cap_max(1);
  Carve cv;
  int rc = carve(ws, n, 1, RotGeom64::RECQ, C, &cv);
  if (rc) return rc;
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  const int T = 256;
  if (n == 0) {
    if (hipMemsetAsync(cv.keep_cnt, 0, 4, st) != hipSuccess || hipMemsetAsync(cv.seg_begin, 0, 4, st) != hipSuccess) return OBB_ERR_LAUNCH;
    k_finalize<<<1, T, 0, st>>>(cv.keep_cnt, cv.seg_begin, 1, max_keep, nullptr, num_keep, nullptr);
    return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
  }
  const unsigned gb = (unsigned)((n + T - 1) / T);
  const int drop_small = (flags & OBB_NMS_DROP_SMALL) ? 1 : 0;
  k_make_keys_f64<<<gb, T, 0, st>>>(scores, dets5, drop_small, (int)n, cv.keys_a, cv.vals_a, cv.seg_begin, cv.seg_end, cv.keep_cnt);
  // 64-bit score keys are not unique: the stable LSD radix sort of segsort.h keeps ties in ascending index (eight passes; this
  // entry serves float64 callers, the double clip behind it costs orders of magnitude more than its sort)
  {
    uint32_t* hist = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(cv.sort_tmp) + align_up(ps_scratch_bytes()));
    rc = seg_radix_sort_large(cv.keys_a, cv.keys_b, cv.vals_a, cv.vals_b, cv.seg_begin, cv.seg_end, 1, n, n, 0xFFu, hist, st);
    if (rc) return rc;
  }
  k_prep_rot64<<<gb, T, 0, st>>>(dets5, cv.vals_b, drop_small, (int)n, cv.rec, cv.alive);
  NmsArgs a{};
  a.rec = cv.rec; a.order = cv.vals_b; a.alive = cv.alive; a.seg_begin = cv.seg_begin; a.seg_end = cv.seg_end;
  a.keep_cnt = cv.keep_cnt; a.keep_out = keep_out;
  a.rows = cv.rows; a.nrows = cv.nrows; a.edges = cv.edges; a.nedges = cv.nedges;
  a.ecap = cv.ecap; a.n = (int)n; a.capmax = C;
  a.max_keep = (int)(max_keep > 0x7fffffffLL ? 0x7fffffffLL : (max_keep < 0 ? 0 : max_keep));
  a.window = nms_window(a.max_keep);
  a.thr = thr; a.thr64 = (double)thr;                      // the kernel's threshold is a float (nms_rotated_cuda.cu:14)
  a.cull = (thr >= 0.f) ? 1 : 0;
  rc = nms_steps(3, a, cv, 1, n, st);
  if (rc) return rc;
  k_finalize<<<1, T, 0, st>>>(cv.keep_cnt, cv.seg_begin, 1, max_keep, cv.abort_flag, num_keep, nullptr);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // namespace obb

#include "nmsobb_impl.h"

using namespace obb;

extern "C" {

size_t obb_nms_workspace_bytes(int64_t n, int64_t nseg, int kind) {
  Carve cv;
  if (n < 0 || nseg < 1) return 0;
  if (kind < 0 || kind > 5) return 0;
  if (carve(nullptr, n, nseg, kind_recq(kind), cap_max(nseg), &cv)) return 0;
  return cv.total;
}

int obb_nms_rotated_f32(const float* dets5, const float* scores, int64_t n, float iou_thr, int flags, int64_t max_keep,
                        int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  return run_nms(0, dets5, 5, scores, 1, n, iou_thr, flags, max_keep, keep_out, num_keep, ws, ws_bytes, (hipStream_t)stream);
}

int obb_nms_rotated_f64(const double* dets5, const double* scores, int64_t n, float iou_thr, int flags, int64_t max_keep,
                        int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  return run_nms_rot64(dets5, scores, n, iou_thr, flags, max_keep, keep_out, num_keep, ws, ws_bytes, (hipStream_t)stream);
}

int obb_nms_poly_f32(const float* polys, int64_t row_stride, int64_t n, float iou_thr, int64_t max_keep, int64_t* keep_out,
                     int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  if (row_stride < 9) return OBB_ERR_BAD_ARG;
  return run_nms(1, polys, (int)row_stride, polys ? polys + 8 : nullptr, (int)row_stride, n, iou_thr, 0, max_keep, keep_out, num_keep,
                 ws, ws_bytes, (hipStream_t)stream);
}


int obb_merge_nms_poly_f64(const double* dets9, int64_t n, const int32_t* order, const int32_t* seg_off, int64_t nseg, double thresh,
                           int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  return run_merge_nms(2, dets9, 9, n, order, seg_off, nseg, thresh, keep_out, num_keep, ws, ws_bytes, (hipStream_t)stream);
}

int obb_merge_nms_poly_all_f64(const double* dets9, int64_t n, const int32_t* order, const int32_t* seg_off, int64_t nseg, double thresh,
                               int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  return run_merge_nms(4, dets9, 9, n, order, seg_off, nseg, thresh, keep_out, num_keep, ws, ws_bytes, (hipStream_t)stream);
}

int obb_merge_nms_hbb_f64(const double* dets, int64_t row_stride, int64_t n, const int32_t* order, const int32_t* seg_off, int64_t nseg,
                          double thresh, int64_t* keep_out, int64_t* num_keep, void* ws, size_t ws_bytes, void* stream) {
  if (row_stride < 4 || row_stride > 0x7fffffff) return OBB_ERR_BAD_ARG;
  return run_merge_nms(5, dets, (int)row_stride, n, order, seg_off, nseg, thresh, keep_out, num_keep, ws, ws_bytes, (hipStream_t)stream);
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
    int rc = run_nms(1, dp, polys_dim, ds, 1, polys_num, nms_overlap_thresh, 0, 0, dk, dn, ws, wsb, (hipStream_t)0);
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

// The reference declares this function with C++ linkage (poly_nms.hpp:9-10: no extern "C"), so a build of its Cython source
// (poly_nms.pyx, `cdef extern from "poly_nms.hpp"`) binds the MANGLED name: the same entry point under that name.
void obb_cxx_poly_nms(int* keep_out_host, int* num_out_host, const float* polys_host, int polys_num, int polys_dim,
                      float nms_overlap_thresh, int device_id) __asm__("_Z9_poly_nmsPiS_PKfiifi");
void obb_cxx_poly_nms(int* keep_out_host, int* num_out_host, const float* polys_host, int polys_num, int polys_dim,
                      float nms_overlap_thresh, int device_id) {
  _poly_nms(keep_out_host, num_out_host, polys_host, polys_num, polys_dim, nms_overlap_thresh, device_id);
}

size_t obb_nms_obb_workspace_bytes(int64_t bs, int64_t cap_img, int64_t nc, int agnostic) {
  ObbCarve cv;
  if (bs < 1 || cap_img < 1 || nc < 1 || nc > 256 || obb_carve(nullptr, bs, cap_img, agnostic ? 1 : nc, &cv)) return 0;
  return cv.total;
}

int obb_non_max_suppression_obb(const void* pred, int dtype, int64_t bs, int64_t A, int64_t no, float conf_thres,
                                float iou_thres, const int32_t* classes_host, int n_classes, int agnostic, int multi_label,
                                int64_t max_det, int64_t max_nms, float max_wh, const float* extra8, int64_t n_extra,
                                int64_t cap_img, int64_t expected_cand, float* out, int out_packed, int64_t* out_count,
                                int64_t* status, void* ws, size_t ws_bytes, void* stream) {
  return obb_non_max_suppression_obb_col(pred, nullptr, dtype, bs, A, no, conf_thres, iou_thres, classes_host, n_classes, agnostic,
                                         multi_label, max_det, max_nms, max_wh, extra8, n_extra, cap_img, expected_cand, out,
                                         out_packed, out_count, status, ws, ws_bytes, stream);
}

int obb_non_max_suppression_obb_col(const void* pred, const void* objcol, int dtype, int64_t bs, int64_t A, int64_t no,
                                    float conf_thres, float iou_thres, const int32_t* classes_host, int n_classes, int agnostic,
                                    int multi_label, int64_t max_det, int64_t max_nms, float max_wh, const float* extra8,
                                    int64_t n_extra, int64_t cap_img, int64_t expected_cand, float* out, int out_packed,
                                    int64_t* out_count, int64_t* status, void* ws, size_t ws_bytes, void* stream) {
  return run_nms_obb(pred, objcol, dtype, bs, A, no, conf_thres, iou_thres, classes_host, n_classes, agnostic, multi_label, max_det,
                     max_nms, max_wh, extra8, n_extra, cap_img, expected_cand, out, out_packed ? 1 : 0, out_count, status, ws, ws_bytes,
                     (hipStream_t)stream);
}

size_t obb_nms_obb_state_bytes(int64_t bs) { return bs < 1 ? 0 : obb_state_bytes(bs); }

int obb_non_max_suppression_obb_st(const void* pred, const void* objcol, int dtype, int64_t bs, int64_t A, int64_t no,
                                   float conf_thres, float iou_thres, const int32_t* classes_host, int n_classes, int agnostic,
                                   int multi_label, int64_t max_det, int64_t max_nms, float max_wh, const float* extra8,
                                   int64_t n_extra, int64_t cap_img, int64_t expected_cand, float* out, int out_packed,
                                   int64_t* out_count, int64_t* status, void* ws, size_t ws_bytes, void* state, size_t state_bytes,
                                   void* stream) {
  if (!state) return OBB_ERR_BAD_ARG;
  return run_nms_obb(pred, objcol, dtype, bs, A, no, conf_thres, iou_thres, classes_host, n_classes, agnostic, multi_label, max_det,
                     max_nms, max_wh, extra8, n_extra, cap_img, expected_cand, out, out_packed ? 1 : 0, out_count, status, ws, ws_bytes,
                     (hipStream_t)stream, state, state_bytes);
}

int obb_profile_enable(int on) {
  g_prof.on = on == 2 ? 2 : (on != 0 ? 1 : 0);
  g_prof.used = 0;
  // the events exist before the first call that records one: creating them lazily put 2 x hipEventCreate per stage into the
  // caller's timed loop (bench.py's first loop ran 9-30 us per step slower than its later ones, depending on the host)
  if (g_prof.on) {
    const int want = 2048 < ProfState::kMax ? 2048 : ProfState::kMax;
    while (g_prof.created < want) {
      if (hipEventCreate(&g_prof.ev0[g_prof.created]) != hipSuccess) return OBB_ERR_INTERNAL;
      if (hipEventCreate(&g_prof.ev1[g_prof.created]) != hipSuccess) { hipEventDestroy(g_prof.ev0[g_prof.created]); return OBB_ERR_INTERNAL; }
      g_prof.created++;
    }
  }
  return OBB_OK;
}

int obb_profile_collect(double* ms_sum, int64_t* count, int n_stages) {
  for (int i = 0; i < n_stages; i++) { ms_sum[i] = 0.0; count[i] = 0; }
  for (int k = 0; k < g_prof.used; k++) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.ev1[k]) != hipSuccess) return OBB_ERR_INTERNAL;
    if (hipEventElapsedTime(&ms, g_prof.ev0[k], g_prof.ev1[k]) != hipSuccess) return OBB_ERR_INTERNAL;
    const int id = g_prof.id[k];
    if (id < n_stages) { ms_sum[id] += ms; count[id]++; }
  }
  g_prof.used = 0;
  return OBB_OK;
}

int obb_nms_set_max_grid(int max_workgroups) {
  g_max_grid = max_workgroups > 0 ? max_workgroups : 0;
  return OBB_OK;
}

const char* obb_version(void) { return "obb_hip 0.2 (gfx950)"; }

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

#ifdef OBB_SMALL_TRACE
// (development builds, tools/small_trace.sh) the stamps of the last k_nms_small launch; both buffers are cleared
extern "C" int obb_debug_small_trace2(unsigned long long* words) {
  static unsigned long long zero2[2048 * 8];
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(words, HIP_SYMBOL(obb::g_small_trace2), sizeof(obb::g_small_trace2)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(obb::g_small_trace2), zero2, sizeof(obb::g_small_trace2)) == hipSuccess ? 0 : -1;
}
extern "C" int obb_debug_small_trace(unsigned long long* seg_words, unsigned long long* tail_words) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(seg_words, HIP_SYMBOL(obb::g_small_trace), sizeof(obb::g_small_trace)) != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(tail_words, HIP_SYMBOL(obb::g_small_trace_tail), sizeof(obb::g_small_trace_tail)) != hipSuccess) return -3;
  static unsigned long long zero[2048 * 16];
  if (hipMemcpyToSymbol(HIP_SYMBOL(obb::g_small_trace), zero, sizeof(obb::g_small_trace)) != hipSuccess) return -4;
  if (hipMemcpyToSymbol(HIP_SYMBOL(obb::g_small_trace_tail), zero, sizeof(obb::g_small_trace_tail)) != hipSuccess) return -5;
  return 0;
}
#endif
