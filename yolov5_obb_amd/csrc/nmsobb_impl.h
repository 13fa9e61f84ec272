// Fused non_max_suppression_obb for gfx950 (included by nms.hip, namespace obb).
//
// Replaces the per-image Python loop of utils/general.py:772-862 (>= 10 tiny ATen kernels and >= 3 host
// syncs per image, then obb_nms with a device->host mask copy) by ONE stream-ordered call for the whole batch:
//
//   k_decode      workgroup = 256..1024 anchors: objectness of every row (the dense column a coupled Detect decode left, or
//                 2-4 B of each 400-800 B row), rows that pass `obj > conf` compacted into LDS; each listed row is then
//                 processed by a whole wave -- coalesced class scores, conf = obj*cls rounded in the input dtype,
//                 multi-label expansion by ballot, CSL decode = wave arg-max over the 180 angle bins (first maximum),
//                 theta = (idx-90)/180*3.141592, class filter -- and appended to the image's candidate region together
//                 with a 64-bit sort key (descending conf, then ascending anchor*nc+class: a deterministic tie rule).
//   sort per image (in LDS, k_sort_prep_lds; larger images: segsort.h / psrs_sort.h), top max_nms (30000) kept   (:845-846)
//   k_prep_cand   class offset xy += cls*max_wh (:849-851), rotated-box records, too-small filter of obb_nms
//   LC-NMS        (nms_core.h) with max_keep = max_det                           (:853-855)
//   k_gather_out  rows [x y l s theta conf cls] of the kept candidates, [bs][max_det][7] + counts
//
// Class segmentation.  The reference runs ONE greedy NMS per image over boxes shifted by cls*4096 so that boxes of
// different classes never overlap (:849-851).  Whenever that premise provably holds for what the reference's IoU
// code computes -- no candidate with 0.001 <= min(l,s) < 1 px (a box that thin, seen from >= 2.5k px away, is where the
// reference's fp32 corner rounding can fabricate an overlap, riou_device.h), at most max_nms candidates, not agnostic
// -- the image is split into one NMS segment per class: the sort key gets the class in its top byte (k_rekey), the
// segments are found by binary search (k_class_bounds), and the per-class kept lists (each in descending score) are
// merged back into the reference's global descending-score order by rank counting (k_gather_out).  The kept set and
// its order are identical to the single-list NMS; the work drops by about the number of classes and the segments
// (bs*nc of them) each get their own workgroup, so no inter-workgroup barrier is left on the path.  Images that do not
// qualify keep the single-list path inside the same launch (segment 0 of the image holds everything).
//
// The only host<->device traffic is the caller reading the bs counts (+ the overflow word) afterwards.
#pragma once
#include "dtype_device.h"
#include "segsort.h"

namespace obb {

struct ClassMask { unsigned long long w[4]; int all; };   // allowed classes (nc <= 256), all != 0: no filter

// product rounded to the tensor dtype: x[:, 5:ci] *= x[:, 4:5] happens in the input dtype (utils/general.py:820)
template <typename T> __device__ __forceinline__ float mul_in_dtype(float a, float b);
template <> __device__ __forceinline__ float mul_in_dtype<float>(float a, float b) { return a * b; }
template <> __device__ __forceinline__ float mul_in_dtype<__half>(float a, float b) { return __half2float(__float2half_rn(a * b)); }
// python float threshold compared against a tensor of the input dtype: the scalar is cast to that dtype
template <typename T> __device__ __forceinline__ float thr_in_dtype(float t);
template <> __device__ __forceinline__ float thr_in_dtype<float>(float t) { return t; }
template <> __device__ __forceinline__ float thr_in_dtype<__half>(float t) { return __half2float(__float2half_rn(t)); }

// wave arg-max with "first maximum" tie rule (torch.max over a dimension: utils/general.py:822, :830).  Value and index
// travel as ONE 64-bit key -- order-preserving image of the float in the high word (-0 counts as +0; a NaN ranks above
// everything, as in torch), ~index in the low word -- so the reduction is a plain unsigned max: four DPP steps inside the
// 16-lane rows (quad swaps, half-row and row mirrors: a few cycles each, where ds_bpermute costs an LDS round trip per
// step), then the four row results meet through v_readlane.
__device__ __forceinline__ unsigned long long argmax_key(float v, int i) {
  const uint32_t b = __float_as_uint(v + 0.0f);
  const uint32_t m = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)m << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long k) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, CTRL, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), CTRL, 0xF, 0xF, true);
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long k) {    // all 64 lanes active
  k = dpp_max_u64<0xB1>(k);      // quad_perm [1,0,3,2]
  k = dpp_max_u64<0x4E>(k);      // quad_perm [2,3,0,1]
  k = dpp_max_u64<0x141>(k);     // row_half_mirror
  return dpp_max_u64<0x140>(k);  // row_mirror: every lane holds the max of its 16-lane row
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {   // all 64 lanes active
  k = row_max_u64(k);
  unsigned long long r[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
    r[j] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), j * 16) << 32) |
           (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, j * 16);
  const unsigned long long a = r[0] > r[1] ? r[0] : r[1], b = r[2] > r[3] ? r[2] : r[3];
  return a > b ? a : b;
}
__device__ __forceinline__ void argmax_unkey(unsigned long long k, float& v, int& i) {
  const uint32_t m = (uint32_t)(k >> 32);
  v = __uint_as_float((m & 0x80000000u) ? (m ^ 0x80000000u) : ~m);
  i = (int)(0xFFFFFFFFu - (uint32_t)k);
}
__device__ __forceinline__ void wave_argmax_first(float& v, int& i) {
  argmax_unkey(wave_max_u64(argmax_key(v, i)), v, i);
}

struct DecodeArgs {
  const void* pred;          // [bs][A][no]
  const void* objcol;        // [bs][A] = pred[..., 4] stored densely by the producer (obb_detect_decode_col), or null
  long long A;
  int no, nc, bs;
  float conf_thres;
  int multi_label;
  int rows_per_thread;       // G: a workgroup of k_decode covers kDecThreads * G rows
  ClassMask cm;
  long long cap_img;         // candidate slots per image
  float4* cand;              // [bs*cap_img][2]: {x,y,l,s} {theta,conf,cls,0}
  unsigned long long* keys;  // [bs*cap_img]
  uint32_t* vals;            // [bs*cap_img] slot index inside the image region
  int* cnt;                  // [bs * kCntPad] candidates produced (may exceed cap_img: overflow), one 256-B line each
  int* tiny;                 // [bs] flags of the image (kImg*): what decides whether its classes may run as independent NMS segments
  float win_lo, win_hi;      // every candidate's circle must lie in win_lo < x < win_hi (a window narrower than max_wh), else kImgWide
  // state of the LATER launches that this kernel zeroes on its way (a call with caller-kept counters has no reset launch:
  // obb_non_max_suppression_obb_st); all NULL / 0: k_reset_state did it
  int* z_ticket; int n_ticket;
  uint4* z_bar16; long long n_bar16;
  uint4* z_alive16; long long n_alive16;
};

// Per-image flags (DecodeArgs::tiny).  The reference runs ONE list per image with xy += cls * max_wh (utils/general.py:849-851);
// one NMS segment per class gives the same kept set exactly when no pair of boxes of DIFFERENT classes can interact:
//   kImgSmall   a candidate has a short side in [0.001, 1) px: the reference's fp32 corner arithmetic is ill conditioned for such a
//               box against a partner tens of thousands of pixels away (riou_device.h: rbox_pair_well_conditioned) and can
//               report an overlap there.  Such an image keeps its class segments only if k_tiny_cross has CHECKED every
//               ill-conditioned cross-class pair with the exact clip and found no IoU > thr (round 5; rounds 1-4: single list);
//   kImgWide    a candidate's circle leaves the window [win_lo, win_hi] on x (an oversized or far-out box, NaN / inf): circles
//               of different classes could really touch -- always the single list;
//   kImgChecked k_tiny_cross ran for the image and completed its list of small boxes;   kImgCross  it found a cross-class hit.
constexpr int kImgSmall = 1, kImgWide = 2, kImgChecked = 4, kImgCross = 8;
__host__ __device__ __forceinline__ bool img_single_list(int t) {
  return (t & kImgWide) || ((t & kImgSmall) && (!(t & kImgChecked) || (t & kImgCross)));
}
__device__ __forceinline__ int cand_flags(float x, float l, float s, float win_lo, float win_hi) {
  const float mn = (s < l) ? s : l;
  const float rr = sqrtf(l * l + s * s) * 0.501f + 0.5f;
  int f = (mn >= 0.001f && mn < 1.0f) ? kImgSmall : 0;
  if (!(x - rr > win_lo && x + rr < win_hi)) f |= kImgWide;     // (NaN / inf anywhere: the comparisons fail)
  return f;
}

__device__ __forceinline__ bool class_allowed(const ClassMask& cm, int c) {
  const int q = (c >> 6) & 3;   // select chain instead of a dynamic index: the mask lives in kernel-argument SGPRs
  const unsigned long long w = q == 0 ? cm.w[0] : q == 1 ? cm.w[1] : q == 2 ? cm.w[2] : cm.w[3];
  return cm.all || ((w >> (c & 63)) & 1ull);
}

// Candidate counters are padded to one 256-byte line each: same-word device atomics retire at ~11 ns apiece
// (MI355X_MICROARCH.md "fanin"/"dequeue"), and 16 counters packed in one line would share one L2 channel.
constexpr int kCntPad = 64;   // ints

// Rows of a workgroup = kDecThreads * G (G = 1, 2 or 4 rows per thread, chosen by the host so that small batches still fill the
// chip).  History: with one wave per 64 consecutive rows that also processed ITS passing rows one after the other, the
// kernel took 52 us on a warm and 78-91 us on a freshly written configs[1] tensor -- even with the dense objectness column:
// detector output is clustered (an object fires on neighbouring cells), so a few waves carried long serial chains of cold
// row reads while most had none.  Now the workgroup compacts its passing rows into LDS and its four waves take them
// round-robin, kDecDepth rows in flight per wave.
constexpr int kDecThreads = 512;      // 8 waves: a workgroup whose share of rows is three times the median still needs one pass
constexpr int kDecWaves = kDecThreads / 64;
constexpr int kDecMaxG = 4;
constexpr int kDecDepth = 2;          // groups of four rows in flight per wave
constexpr int kDecStage = 128;         // staged candidates per wave (LDS) before a flush

// Phase 2 of k_decode works on FOUR rows per wave, one per 16-lane DPP row: a row of the head output holds ~200 elements,
// and with a whole wave per row most of every instruction was bookkeeping (measured: ~450 wave instructions per row,
// VALU-bound).  Lane l16 of a group holds angle bins l16, l16+16, ... (12 registers), classes l16 and 16+l16 (further
// class groups of 16 are loaded on demand: nc > 32), and the box.  Everything a group needs is requested up front.
constexpr int kDecCslRegs = 12;          // ceil(180 / 16)
constexpr int kDecClsRegs = 2;           // class groups of 16 that travel with the row
template <typename T>
struct DecQuad {
  float csl[kDecCslRegs];
  float cls[kDecClsRegs];
  float box[4];
  float obj;
  uint32_t row;
  bool valid;
};

template <typename T>
__device__ __forceinline__ DecQuad<T> dec_load_quad(const T* img, int no, int nc, const uint32_t* s_row, const float* s_obj,
                                                    int j, int n_rows, int l16) {
  DecQuad<T> r;
  r.valid = j < n_rows;
  r.row = r.valid ? s_row[j] : 0u;
  r.obj = r.valid ? s_obj[j] : 0.f;
  const T* base = img + (size_t)r.row * no;
  const T* csl = base + 5 + nc;
#pragma unroll
  for (int k = 0; k < kDecCslRegs; k++) {
    const int bin = k * 16 + l16;
    r.csl[k] = (r.valid && bin < 180) ? ld_as_float<T>(csl + bin) : -__builtin_inff();
  }
#pragma unroll
  for (int g = 0; g < kDecClsRegs; g++) {
    const int c = g * 16 + l16;
    r.cls[g] = (r.valid && c < nc) ? ld_as_float<T>(base + 5 + c) : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) r.box[k] = r.valid ? ld_as_float<T>(base + k) : 0.f;
  return r;
}

// k_decode: workgroup = 512 threads x G rows.  Phase 1 reads the objectness of every row -- from the dense column when the
// producer stored one (obb_detect_decode_col), else one 2-4 byte element of each 400-800 byte row -- and compacts the rows
// with obj > conf into an LDS list (wave ballot + one LDS atomic per wave).  Phase 2: the four waves take the listed rows
// round-robin in groups of four (DecQuad above), kDecDepth groups in flight per wave, and stage the candidates in LDS.  Slots in the image's candidate region are claimed with ONE atomic per workgroup (plus
// one per wave whenever its 128-entry stage fills up) and the staged records are written out coalesced.
template <typename T>
__global__ __launch_bounds__(kDecThreads) void k_decode(DecodeArgs a) {
  __shared__ float4 s_c0[kDecWaves][kDecStage], s_c1[kDecWaves][kDecStage];
  __shared__ unsigned long long s_key[kDecWaves][kDecStage];
  __shared__ int s_cnt[kDecWaves], s_base, s_n;
  __shared__ uint32_t s_row[kDecThreads * kDecMaxG];
  __shared__ float s_obj[kDecThreads * kDecMaxG];

  const T* pred = (const T*)a.pred;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  const int G = a.rows_per_thread;
  const float thr = thr_in_dtype<T>(a.conf_thres);
  const T* img = pred + (size_t)b * a.A * a.no;
  float4* cand = a.cand + (size_t)b * a.cap_img * 2;
  unsigned long long* keys = a.keys + (size_t)b * a.cap_img;
  uint32_t* vals = a.vals + (size_t)b * a.cap_img;
  float4* c0s = s_c0[wv]; float4* c1s = s_c1[wv]; unsigned long long* kys = s_key[wv];
  int staged = 0;   // wave-uniform
  int flags_seen = 0;
  if (tid == 0) s_n = 0;
  if (a.z_ticket != nullptr) {                                   // (kernel-uniform) what k_reset_state zeroes for the sort and NMS launches
    const long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * kDecThreads + tid, n = (long long)gridDim.x * gridDim.y * kDecThreads;
    if (i < a.n_ticket) a.z_ticket[i] = 0;
    for (long long k = i; k < a.n_bar16; k += n) a.z_bar16[k] = make_uint4(0u, 0u, 0u, 0u);
    for (long long k = i; k < a.n_alive16; k += n) a.z_alive16[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  auto write_out = [&](long long base, int count) {
    for (int i = lane; i < count; i += 64) {
      const long long slot = base + i;
      if (slot < a.cap_img) {
        cand[slot * 2] = c0s[i];
        cand[slot * 2 + 1] = c1s[i];
        keys[slot] = kys[i];
        vals[slot] = (uint32_t)slot;
      }
    }
  };
  auto flush_wave = [&]() {   // mid-kernel flush of a full stage: one atomic for the wave
    int base = 0;
    if (lane == 0) base = atomicAdd(&a.cnt[b * kCntPad], staged);
    base = __shfl(base, 0);
    write_out(base, staged);
    staged = 0;
  };

  // ---- phase 1: objectness of the workgroup's rows                    :785  xc = prediction[..., 4] > conf_thres
  float obj[kDecMaxG];
  // 16-row chunks are dealt to the workgroups of the image round-robin (chunk c -> workgroup c % gridDim.x): detector output
  // is clustered -- a coarse level holds hundreds of passing rows in a few thousand consecutive rows.  Contiguous 1024-row
  // blocks left a handful of workgroups with ten times the average number of rows to reduce, 64-row chunks still three
  // times (0 / 14 / 44 rows: min / median / max on the planted-object batch); the objectness reads do not care (a strided
  // read is one line per row anyway, the dense column is read in 32-byte pieces).
  auto row_of = [&](int q) -> long long { return ((long long)(q * (kDecThreads / 16) + (tid >> 4)) * gridDim.x + blockIdx.x) * 16 + (tid & 15); };
#pragma unroll
  for (int q = 0; q < kDecMaxG; q++) {
    const long long row = row_of(q);
    obj[q] = (q >= G || row >= a.A) ? 0.f
             : a.objcol ? ld_as_float<T>((const T*)a.objcol + (size_t)b * a.A + row)       // 128 bytes per wave
                        : ld_as_float<T>(img + (size_t)row * a.no + 4);                     // one line per row
  }
#pragma unroll
  for (int q = 0; q < kDecMaxG; q++) {
    const bool p = q < G && row_of(q) < a.A && obj[q] > thr;
    const unsigned long long mk = __ballot(p);
    if (mk) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_n, __popcll(mk));
      base = __shfl(base, 0);
      if (p) {
        const int i = base + __popcll(mk & lanemask_lt());
        s_row[i] = (uint32_t)row_of(q);
        s_obj[i] = obj[q];
      }
    }
  }
  __syncthreads();
  const int n_rows = s_n;

  // ---- phase 2: the listed rows in groups of four (one per 16-lane row); wave wv takes groups wv, wv + 8, ...,
  // kDecDepth groups requested before the first is reduced
  const int l16 = lane & 15, sub = lane >> 4;
  const int ncg = (a.nc + 15) >> 4;
  auto reduce_quad = [&](const DecQuad<T>& cur) {
    // CSL decode: first arg-max over the 180 bins (:822-823), inside the 16-lane row
    unsigned long long tk = argmax_key(cur.csl[0], l16);
#pragma unroll
    for (int k = 1; k < kDecCslRegs; k++) { const unsigned long long kk = argmax_key(cur.csl[k], k * 16 + l16); tk = kk > tk ? kk : tk; }
    float tv; int ti;
    argmax_unkey(row_max_u64(tk), tv, ti);
    const float theta = ((float)(ti - 90) / 180.0f) * 3.141592f;
    const float bx = cur.box[0], by = cur.box[1], bl = cur.box[2], bs_ = cur.box[3];
    const int bflags = cand_flags(bx, bl, bs_, a.win_lo, a.win_hi);
    const long long rw = (long long)cur.row;
    auto stage = [&](bool p, float conf, int c) {        // one candidate per lane with p set
      const unsigned long long pb = __ballot(p);
      const int np = __popcll(pb);
      if (np == 0) return;
      if (staged + np > kDecStage) flush_wave();
      if (p) {
        const int i = staged + __popcll(pb & lanemask_lt());
        c0s[i] = make_float4(bx, by, bl, bs_);
        c1s[i] = make_float4(theta, conf, (float)c, 0.f);
        kys[i] = ((unsigned long long)score_desc_key(conf) << 32) | (unsigned long long)(uint32_t)(rw * a.nc + c);
        flags_seen |= bflags;
      }
      staged += np;
    };
    // class confidences (:820 conf = obj * cls in the input dtype)
    float bestv = -__builtin_inff(); int besti = 0x7fffffff;
    for (int g = 0; g < ncg; g++) {
      const int c = g * 16 + l16;
      float raw;
      if (g < kDecClsRegs) raw = (g == 0) ? cur.cls[0] : cur.cls[1];
      else raw = (cur.valid && c < a.nc) ? ld_as_float<T>(img + (size_t)cur.row * a.no + 5 + c) : 0.f;
      const float v = (cur.valid && c < a.nc) ? mul_in_dtype<T>(raw, cur.obj) : -__builtin_inff();
      if (a.multi_label) stage(cur.valid && c < a.nc && v > thr && class_allowed(a.cm, c), v, c);             // :827, :835
      else if (c < a.nc && v > bestv) { bestv = v; besti = c; }            // a lane sees ascending c: first max kept
    }
    if (!a.multi_label) {
      float bv; int bi;
      argmax_unkey(row_max_u64(argmax_key(bestv, besti)), bv, bi);                                            // :830
      stage(cur.valid && l16 == 0 && bv > thr && class_allowed(a.cm, bi), bv, bi);                            // :831, :835
    }
  };
  static_assert(kDecClsRegs == 2, "reduce_quad selects cls[0] / cls[1] explicitly");
  for (int g0 = wv; g0 * 4 < n_rows; g0 += kDecWaves * kDecDepth) {
    DecQuad<T> quads[kDecDepth];
#pragma unroll
    for (int i = 0; i < kDecDepth; i++) {
      const int g = g0 + kDecWaves * i;
      if (g * 4 < n_rows) quads[i] = dec_load_quad<T>(img, a.no, a.nc, s_row, s_obj, g * 4 + sub, n_rows, l16);
    }
#pragma unroll
    for (int i = 0; i < kDecDepth; i++) {
      const int g = g0 + kDecWaves * i;
      if (g * 4 < n_rows) reduce_quad(quads[i]);
    }
  }

  {
    const int fl = (__ballot(flags_seen & kImgSmall) ? kImgSmall : 0) | (__ballot(flags_seen & kImgWide) ? kImgWide : 0);
    if (fl && lane == 0) atomicOr(&a.tiny[b], fl);
  }
  // ---- one atomic per workgroup for whatever is still staged
  if (lane == 0) s_cnt[wv] = staged;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < kDecWaves; w++) tot += s_cnt[w];
    s_base = tot ? atomicAdd(&a.cnt[b * kCntPad], tot) : 0;
  }
  __syncthreads();
  int base = s_base;
  for (int w = 0; w < wv; w++) base += s_cnt[w];
  write_out(base, staged);
}

// apriori label rows (utils/general.py:807-813), prepared by the host layer as [img, x, y, l, s, theta, conf, cls]
__global__ void k_append_extra(const float* __restrict__ extra8, int m, long long A, int nc, DecodeArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float* e = extra8 + (size_t)i * 8;
  const int b = (int)e[0];
  if (b < 0 || b >= a.bs) return;
  // a label row is a candidate like any other (:807-813 append it in front of the filters): conf = 1 > conf_thres (:827 / :831) and
  // the class filter (:835) apply to it -- until round 6 the class filter did not (found by tools/self_fuzz.py: `labels` + `classes`)
  if (!(e[6] > a.conf_thres) || !class_allowed(a.cm, (int)e[7])) return;
  const int slot = atomicAdd(&a.cnt[b * kCntPad], 1);
  { const int fl = cand_flags(e[1], e[3], e[4], a.win_lo, a.win_hi); if (fl) atomicOr(&a.tiny[b], fl); }
  if (slot >= a.cap_img) return;
  const size_t g = (size_t)b * a.cap_img + slot;
  a.cand[g * 2] = make_float4(e[1], e[2], e[3], e[4]);
  a.cand[g * 2 + 1] = make_float4(e[5], e[6], e[7], 0.f);
  a.keys[g] = ((unsigned long long)score_desc_key(e[6]) << 32) | (unsigned long long)(uint32_t)(A * nc + i);
  a.vals[g] = (uint32_t)slot;
}

// Cross-class check of the images that hold short-sided boxes (kImgSmall, see the flag table above): every pair (short-sided
// box T, box B of ANOTHER class) that rbox_pair_well_conditioned does not vouch for goes through the reference's own arithmetic
// in the reference's argument order (the higher-scored box first, nms_rotated_cuda.cu:44-60), in the coordinates the reference
// uses (xy + cls * max_wh): first the COUNT of the clip's candidate points (rbox_npoints: the 16 edge crossings and the contained
// corners in the reference's fp32 operations, ~400 flops, no scratch) -- at most two points and the reference returns IoU = 0
// (box_iou_rotated_utils.h:322-324), which is what happens for all but a few thin boxes that point along the diagonal of the class
// offsets -- and the whole clip only for the pairs with more.  No IoU > thr among them: the classes of the image cannot interact
// (well-conditioned cross-class pairs have disjoint circles inside their windows: the reference returns exactly 0), its class
// segments give the reference's kept set.  Otherwise the image keeps the reference's single list -- and so does an image with more
// than kTinyMax short-sided boxes or more than kTinyPairs (box, candidate) combinations: the check is for the stray sub-pixel box of
// a trained detector (microseconds), and bounded at a few tens of microseconds per image.  (Measured on the conv stand-in's random-
// initialised heads, which put hundreds of sub-pixel boxes into some images: with bounds of 512 / 2 x 10^6 the check cost 0.7 ms per
// batch there and saved 0.4 ms of NMS kernel; such images keep the single list.)
// Launched only when the caller's previous call of the shape met such boxes (expected_cand bit 62): grid (kTinyParts, bs),
// blocks of images without the flag return at once.  Candidates beyond the top-max_nms cut are tested as well: conservative.
constexpr int kTinyMax = 64;              // short-sided boxes of an image the check takes ...
constexpr long long kTinyPairs = 262144;  // ... and candidates x short-sided boxes: above either the image is not checked (single list)
constexpr int kTinyParts = 32;
__global__ __launch_bounds__(256) void k_tiny_cross(const float4* __restrict__ cand, const unsigned long long* __restrict__ keys,
                                                    const int* __restrict__ cnt, long long cap_img, float class_offset, float thr,
                                                    int* __restrict__ tiny) {
  __shared__ float scr[RotGeom::SCR * 256];                     // clip scratch (the rare full clip): one column per thread inside its wave's block
  __shared__ RBoxFeat s_feat[kTinyMax];                         // the short-sided boxes in the reference's coordinates
  __shared__ unsigned long long s_key[kTinyMax];
  __shared__ float s_cls[kTinyMax];
  __shared__ int s_list[kTinyMax];
  __shared__ int s_n;
  const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if ((tiny[b] & (kImgSmall | kImgWide)) != kImgSmall) return;
  long long n = cnt[b * kCntPad];
  if (n > cap_img) n = cap_img;                                  // (a call that overflowed its slots is repeated by the caller)
  const size_t base = (size_t)b * cap_img;
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (long long j = tid; j < n; j += 256) {                     // the image's short-sided candidates: every part builds the same set
    const float4 c0 = cand[(base + j) * 2];
    const float mn = c0.w < c0.z ? c0.w : c0.z;
    if (mn >= 0.001f && mn < 1.0f) { const int k = atomicAdd(&s_n, 1); if (k < kTinyMax) s_list[k] = (int)j; }
  }
  __syncthreads();
  const int nt = s_n;
  if (nt > kTinyMax || (long long)nt * n > kTinyPairs) return;   // not checked: single list
  if (part == 0 && tid == 0) atomicOr(&tiny[b], kImgChecked);
  for (int t = tid; t < nt; t += 256) {
    const int jt = s_list[t];
    const float4 t0 = cand[(base + jt) * 2], t1 = cand[(base + jt) * 2 + 1];
    const float offT = t1.z * class_offset;
    s_feat[t] = rbox_make_feat(t0.x + offT, t0.y + offT, t0.z, t0.w, t1.x);
    s_key[t] = keys[base + jt];
    s_cls[t] = t1.z;
  }
  __syncthreads();
  bool hit = false;
  float* myscr = scr + wv * (RotGeom::SCR * 64) + lane;
  for (long long j = (long long)part * 256 + tid; j < n; j += (long long)gridDim.x * 256) {
    const float4 c0 = cand[(base + j) * 2], c1 = cand[(base + j) * 2 + 1];
    if (fminf(c0.z, c0.w) < 0.001f) continue;                    // dropped by obb_nms (nms_rotated_wrapper.py:32): never a partner
    const float offB = c1.z * class_offset;
    const RBoxFeat B = rbox_make_feat(c0.x + offB, c0.y + offB, c0.z, c0.w, c1.x);
    const unsigned long long kB = keys[base + j];
    for (int t = 0; t < nt; t++) {
      if (s_cls[t] == c1.z) continue;                            // same class: decided inside the class segment, as in the single list
      const RBoxFeat T = s_feat[t];
      if (rbox_pair_well_conditioned(T, B)) continue;            // circles apart (different windows) and well conditioned: exactly 0
      const bool t_first = s_key[t] < kB;                        // smaller key = sorts first = the row box of the reference's kernel
      const int np = t_first ? rbox_npoints(T, B) : rbox_npoints(B, T);
      if (np <= 2) continue;                                     // the reference returns 0 (:322-324 / :353)
      const float v = t_first ? rbox_iou<64>(T, B, myscr, myscr + 24 * 64) : rbox_iou<64>(B, T, myscr, myscr + 24 * 64);
      if (v > thr) hit = true;
    }
  }
  if (__ballot(hit) && lane == 0) atomicOr(&tiny[b], kImgCross);
}

// per image: sort range, number of positions that take part, mode:
//   0  single list (the reference's formulation)
//   1  one NMS segment per class, class in the top key byte (k_rekey) -- needs <= max_nms candidates
//   2  more than max_nms candidates: single-list sort first (the top max_nms by confidence must be cut, :845-846), then one
//      stable pass groups the survivors by class (seg_group_by_class); segments per class as in mode 1
__global__ void k_cand_segments(const int* __restrict__ cnt, const int* __restrict__ tiny, int bs, long long cap_img, long long max_nms,
                                int class_ok, int group_ok, int* sort_begin, int* sort_end, int* img_end, int* mode, int* grp_begin,
                                int* grp_end) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= bs) return;
  long long c = cnt[g * kCntPad];
  const bool over_cap = c > cap_img;
  if (over_cap) c = cap_img;
  const int b0 = (int)(g * cap_img);
  sort_begin[g] = b0; sort_end[g] = b0 + (int)c;
  const bool over_nms = max_nms > 0 && c > max_nms;
  if (over_nms) c = max_nms;                                 // :845-846 top max_nms by confidence
  img_end[g] = b0 + (int)c;
  const int m = (class_ok && !over_cap && !img_single_list(tiny[g])) ? (over_nms ? (group_ok ? 2 : 0) : 1) : 0;   // group_ok: the host launches the pass
  mode[g] = m;
  grp_begin[g] = b0; grp_end[g] = (m == 2) ? b0 + (int)c : b0;      // range of the class-grouping pass (empty unless mode 2)
}

// class-segmented images: key (score_desc << 32 | anchor*nc + cls)  ->  (cls << 56 | score_desc << 24 | anchor)
// (ascending key order inside a class = descending score, ties by ascending anchor; extra rows count as anchors A, A+1, ..)
__global__ void k_rekey(const float4* __restrict__ cand, unsigned long long* __restrict__ keys, const int* __restrict__ sort_begin,
                        const int* __restrict__ sort_end, const int* __restrict__ mode, long long A, int nc) {
  const int g = blockIdx.y;
  if (mode[g] != 1) return;
  const int p = sort_begin[g] + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= sort_end[g]) return;
  const unsigned long long k = keys[p];
  const unsigned long long score = k >> 32, tie = k & 0xffffffffull;
  const unsigned long long cls = (unsigned long long)(int)cand[(size_t)p * 2 + 1].z;
  const unsigned long long lim = (unsigned long long)A * nc;
  const unsigned long long anchor = tie < lim ? tie / nc : (unsigned long long)A + (tie - lim);
  keys[p] = (cls << 56) | (score << 24) | (anchor & 0xffffffull);
}

__device__ __forceinline__ int key_lower_bound_range(const unsigned long long* keys, int lo, int hi, unsigned long long target) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// segment table: ncs segments per image (ncs = nc, or 1 when the call is class-agnostic)
__global__ void k_class_bounds(const unsigned long long* __restrict__ keys_sorted, const int* __restrict__ sort_begin,
                               const int* __restrict__ img_end, const int* __restrict__ mode, const uint32_t* __restrict__ digit_base,
                               int bs, int ncs, int* __restrict__ seg_begin, int* __restrict__ seg_end, int* __restrict__ keep_cnt) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= bs * ncs) return;
  const int g = s / ncs, c = s - g * ncs;
  const int b0 = sort_begin[g], e0 = img_end[g];
  int lo = b0, hi = b0;
  if (mode[g] == 1) {
    lo = key_lower_bound_range(keys_sorted, b0, e0, (unsigned long long)c << 56);
    hi = (c >= 255) ? e0 : key_lower_bound_range(keys_sorted, b0, e0, (unsigned long long)(c + 1) << 56);
  } else if (mode[g] == 2) {                                   // class runs of the grouping pass
    lo = b0 + (int)digit_base[(size_t)g * 256 + c];
    hi = (c + 1 < ncs && c + 1 < 256) ? b0 + (int)digit_base[(size_t)g * 256 + c + 1] : e0;
  } else if (c == 0) {
    hi = e0;                                                   // single list: everything in segment 0 of the image
  }
  seg_begin[s] = lo; seg_end[s] = hi; keep_cnt[s] = 0;
}

// mode-2 images: the grouped range comes back from the ping buffers
__global__ void k_copy_grouped(const unsigned long long* __restrict__ ksrc, unsigned long long* __restrict__ kdst,
                               const uint32_t* __restrict__ vsrc, uint32_t* __restrict__ vdst, const int* __restrict__ grp_begin,
                               const int* __restrict__ grp_end) {
  const int g = blockIdx.y;
  const int p = grp_begin[g] + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= grp_end[g]) return;
  kdst[p] = ksrc[p]; vdst[p] = vsrc[p];
}

__global__ void k_prep_cand(const float4* __restrict__ cand, const uint32_t* __restrict__ vals_sorted,
                            const int* __restrict__ img_begin, const int* __restrict__ img_end, long long cap_img,
                            float class_offset, float4* __restrict__ rec, u64* __restrict__ alive) {
  // img_begin[g] = g * cap_img with cap_img a multiple of 64: every wave covers exactly one word of the alive bitmap
  const int g = blockIdx.y;
  const int p = img_begin[g] + blockIdx.x * blockDim.x + threadIdx.x;
  const int se = img_end[g];
  bool ok = false;
  if (p < se) {
    const size_t ci = (size_t)g * cap_img + vals_sorted[p];
    const float4 c0 = cand[ci * 2], c1 = cand[ci * 2 + 1];
    const float off = c1.z * class_offset;                       // :849  c = x[:, 6:7] * (0 if agnostic else max_wh)
    const float x = c0.x + off, y = c0.y + off;                  // :851
    RBoxFeat f = rbox_make_feat(x, y, c0.z, c0.w, c1.x);
    float4 q[4];
    RotGeom::pack(f, q);
#pragma unroll
    for (int k = 0; k < 4; k++) rec[(size_t)p * 4 + k] = q[k];
    const float mn = (c0.w < c0.z) ? c0.w : c0.z;
    ok = !(mn < 0.001f);                                         // nms_rotated_wrapper.py:32
  }
  const u64 m = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && m) alive[p >> 6] = m;           // the bitmap was zeroed before
}

// One launch instead of four memsets: candidate counters, tiny-box flags, the caller's status words, the fused kernel's
// ticket and the team-barrier block of the NMS kernel.
__global__ void k_reset_state(int* __restrict__ cnt, int n_cnt, int* __restrict__ tiny, int bs, int64_t* __restrict__ status,
                              int* __restrict__ ticket, int n_ticket, uint4* __restrict__ bar16, long long n_bar16, uint4* __restrict__ alive16,
                              long long n_alive16) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, n = (long long)gridDim.x * blockDim.x;
  for (long long k = i; k < n_cnt; k += n) cnt[k] = 0;
  for (long long k = i; k < bs; k += n) tiny[k] = 0;
  (void)status;                                                 // (both status words are written by the output stage)
  if (i < n_ticket) ticket[i] = 0;                              // [0] the planner's ticket, [4] largest NMS segment, [8] a segment k_nms_small left out,
                                                                // [16 .. 16 + bs] the tickets of k_nms_small's output stage
  for (long long k = i; k < n_bar16; k += n) bar16[k] = make_uint4(0u, 0u, 0u, 0u);
  // (the in-LDS sort path ORs its alive bits into zeroed words: several workgroups share the words of an image)
  for (long long k = i; k < n_alive16; k += n) alive16[k] = make_uint4(0u, 0u, 0u, 0u);
}

// the slots behind the image's candidates get the largest key (single image, device-wide sort over the whole capacity)
__global__ void k_pad_keys(unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, const int* __restrict__ sort_end, long long cap) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap && i >= (long long)sort_end[0]) { keys[i] = ~0ull; vals[i] = 0u; }
}

// Small and medium images (at most kSortLdsMax candidates each, the regime of the reference's default thresholds): ONE
// workgroup per image does everything between the decode kernel and the NMS kernel -- segment bookkeeping
// (k_cand_segments), class keys (k_rekey), the sort (a bitonic network on (key, slot) pairs in LDS), the class segment table (k_class_bounds), the NMS records and the alive
// bitmap including its zero words (k_prep_cand + memset): six launches become one.  An image with more candidates than
// the network takes is left empty; the caller sees its count in status[1] and calls again with that hint.
constexpr int kSortLdsMax = OBB_NMS_SORT_LDS_MAX;
constexpr int64_t kSortLdsHint = OBB_NMS_SORT_LDS_HINT;   // use the in-LDS path when the previous call's largest image was at most this
__global__ __launch_bounds__(1024) void k_sort_prep_lds(const float4* __restrict__ cand, const unsigned long long* __restrict__ keys_in,
                                                        const uint32_t* __restrict__ vals_in, unsigned long long* __restrict__ keys_out,
                                                        uint32_t* __restrict__ vals_out, const int* __restrict__ cnt,
                                                        const int* __restrict__ tiny, int bs, long long cap_img, long long max_nms,
                                                        int class_ok, long long A, int nc, int ncs, float class_offset, int* sort_begin,
                                                        int* sort_end, int* img_end, int* mode, int* grp_begin, int* grp_end,
                                                        int* __restrict__ seg_begin, int* __restrict__ seg_end, int* __restrict__ keep_cnt,
                                                        float4* __restrict__ rec, u64* __restrict__ alive, int* __restrict__ ticket,
                                                        int plan_nb, int plan_chunk, int4* __restrict__ plan, int* __restrict__ seg_size, int P,
                                                        int helpers, int* __restrict__ help_work, int* __restrict__ seg_np, int* __restrict__ seg_ticket) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ PlanLds s_plan;
  __shared__ int s_hist[256], s_cls_off[256], s_cls_cur[256], s_cls_task[257], s_cls_max, s_cls_ntask;
  __shared__ u64 s_abits[kSortLdsMax / 2 / 64];
  unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_raw);
  uint32_t* s_vals = reinterpret_cast<uint32_t*>(s_keys + kSortLdsMax);
  // blockIdx.x = image * P + part (class-segment images: part q orders and prepares the classes c with c % P == q; every
  // part reads the image's keys and builds the whole histogram itself), the last block is the planner
  const int g = blockIdx.x / P, q = blockIdx.x - g * P, tid = threadIdx.x, T = 1024;
  if (g == bs) {
    // The planner block (launched only when there is a plan to make): the NMS launch is planned from the segment SIZES,
    // which every image's workgroup knows after its first pass over the keys -- long before its sort is done.  This block
    // waits for the bs size tables (the other workgroups never wait for anything, so it cannot hang however the blocks
    // are scheduled) and plans while they sort: the 16 us of plan_teams_block used to be the tail of the last workgroup.
#ifdef OBB_SORT_TRACE
    const unsigned long long tp0 = wall_clock64();
#endif
    if (tid == 0) {
      while (__hip_atomic_load(ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < bs) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
#ifdef OBB_SORT_TRACE
    const unsigned long long tp1 = wall_clock64();
#endif
    plan_teams_block(nullptr, nullptr, seg_size, bs * ncs, plan_nb, plan_chunk, plan, s_plan);
#ifdef OBB_SORT_TRACE
    __syncthreads();
    if (tid == 0) printf("planner: waited %llu planned %llu (x10 ns)\n", tp1 - tp0, wall_clock64() - tp1);
#endif
    return;
  }
  long long c = cnt[g * kCntPad];
  const bool over_cap = c > cap_img;
  if (over_cap) c = cap_img;
  const int b0 = (int)(g * cap_img);
  if (c > kSortLdsMax) c = 0;                                  // not this kernel's regime: reported through status[1]
  const bool over_nms = max_nms > 0 && c > max_nms;
  const int e = (int)(over_nms ? max_nms : c);                 // :845-846 top max_nms by confidence
  const int m = (class_ok && !over_cap && !img_single_list(tiny[g]) && !over_nms) ? 1 : 0;
  const int n = (int)c;
  if (q > 0 && m != 1) return;                                 // one list per image: part 0 does it all
  if (tid == 0 && q == 0) {
    sort_begin[g] = b0; sort_end[g] = b0 + n; img_end[g] = b0 + e; mode[g] = m;
    grp_begin[g] = b0; grp_end[g] = b0;
  }
  if (tid < 256) s_hist[tid] = 0;
  if (tid < kSortLdsMax / 2 / 64) s_abits[tid] = 0ull;
  __syncthreads();
#ifdef OBB_SORT_TRACE
  unsigned long long tt[12]; int ti_ = 0;
#define TSTAMP() do { __syncthreads(); tt[ti_++] = wall_clock64(); } while (0)
#else
#define TSTAMP() do {} while (0)
#endif
  TSTAMP();
  int npad = 64;
  while (npad < n) npad <<= 1;
  const unsigned long long lim = (unsigned long long)A * nc;
  for (int i = tid; i < npad; i += T) {
    unsigned long long k = ~0ull;
    uint32_t v = 0;
    if (i < n) {
      k = keys_in[b0 + i]; v = vals_in[b0 + i];
      if (m == 1) {                                            // k_rekey: (cls << 56 | score_desc << 24 | anchor)
        const unsigned long long score = k >> 32, tie = k & 0xffffffffull;
        const unsigned long long cls = (unsigned long long)(int)cand[(size_t)(b0 + i) * 2 + 1].z;
        const unsigned long long anchor = tie < lim ? tie / nc : (unsigned long long)A + (tie - lim);
        k = (cls << 56) | (score << 24) | (anchor & 0xffffffull);
        atomicAdd(&s_hist[(int)cls & 255], 1);
      }
    }
    s_keys[i] = k; s_vals[i] = v;
  }
  __syncthreads();
  if (plan != nullptr && q == 0) {                             // the size table of this image, for the planner block
    // write-through stores, drained, then a RELAXED arrival: a release fence here writes back every dirty line of this XCD's
    // L2 -- the records the other parts are storing at this very moment (measured: this handshake 4.8 -> 10.4 us)
    for (int sgm = tid; sgm < ncs; sgm += T) stg_agent(seg_size + g * ncs + sgm, (m == 1) ? s_hist[sgm] : (sgm == 0 ? e : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  TSTAMP();
  // ascending; keys are unique (the tie word), so the result is the one total order
  // elements per thread: as few as the 1024 threads allow (measured: 8 per thread with 256 busy threads moves four more
  // levels into registers but is 16 us slower at 2048 candidates -- the lane shuffles become the bottleneck)
  // Class-segment images of up to 4096 candidates (the reference's default thresholds: ~1.7k per image in 15-18 classes): the
  // order wanted is class-major, so the candidates are first dealt into their class buckets (the histogram exists already;
  // scan, one LDS atomic per element) and every bucket is then ordered on its own, without a workgroup barrier inside --
  // instead of the 66 stages + 5 rank-merge levels of the 2048-element network (19.3 us).
  bool by_class = false;
  if (m == 1 && n <= kSortLdsMax / 2) {              // (workgroup-uniform)
    if (tid < 64) {
      int c4[4], sum = 0, mx = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) { c4[u] = s_hist[tid * 4 + u]; sum += c4[u]; mx = c4[u] > mx ? c4[u] : mx; }
      int incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (tid >= d) incl += up; }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
      int run = incl - sum;
#pragma unroll
      for (int u = 0; u < 4; u++) { s_cls_off[tid * 4 + u] = run; s_cls_cur[tid * 4 + u] = run; run += c4[u]; }
      // tasks of the rank-counting pass below: one per 64 elements of a class
      int t4[4], tsum = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) { t4[u] = ((tid * 4 + u) % P == q) ? (c4[u] + 63) >> 6 : 0; tsum += t4[u]; }
      int tincl = tsum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(tincl, d); if (tid >= d) tincl += up; }
      int trun = tincl - tsum;
#pragma unroll
      for (int u = 0; u < 4; u++) { s_cls_task[tid * 4 + u] = trun; trun += t4[u]; }
      if (tid == 63) { s_cls_task[256] = tincl; s_cls_ntask = tincl; }
      if (tid == 0) s_cls_max = mx;
    }
    __syncthreads();
    // (the rank counting is quadratic in the bucket: beyond 512 elements -- two classes, one dominant class -- the 16-wave network takes over)
    by_class = s_cls_max <= 512;
  }
  // the call's largest NMS segment (feedback for the caller's choice of the NMS kernel, include/obb_hip.h: status[1])
  if (q == 0 && tid == 0) atomicMax(ticket + 4, (m == 1 && n <= kSortLdsMax / 2) ? s_cls_max : e);
  TSTAMP();
  if (!by_class) {
    if (q > 0) return;                               // (workgroup-uniform) the network orders the whole image in part 0
    for (int sgm = tid; sgm < ncs; sgm += T) {
      seg_begin[g * ncs + sgm] = b0; seg_end[g * ncs + sgm] = b0; keep_cnt[g * ncs + sgm] = 0;
      if (helpers > 0) seg_np[g * ncs + sgm] = 1;                // (segments of this path stay whole)
    }
    __syncthreads();
  }
  if (by_class) {
    // segment table of this part's classes: straight from the histogram
    for (int c = q + P * tid; c < ncs; c += P * T) {
      const int cc = s_hist[c], sb = cc ? b0 + s_cls_off[c] : b0;
      seg_begin[g * ncs + c] = sb; seg_end[g * ncs + c] = sb + cc; keep_cnt[g * ncs + c] = 0;
      if (helpers > 0) {
        // k_nms_small shares a large segment among several workgroups (nms_small.h: SmallArgs): the helpers are handed out here,
        // where the sizes are known -- ticket[12] counts the slots; a segment that does not get its full set stays whole
        int np = (cc <= kSmallMax) ? small_parts(cc) : 1, base = 0;
        if (np > 1) {
          base = atomicAdd(ticket + 12, np - 1);
          const bool fits = base + np - 1 <= helpers;
          for (int p = 1; p < np; p++) if (base + p - 1 < helpers) help_work[base + p - 1] = fits ? ((g * ncs + c) | (p << 24)) : -1;
          if (!fits) np = 1;
        }
        seg_np[g * ncs + c] = np | (base << 8);
        seg_ticket[g * ncs + c] = 0;
      }
    }
    unsigned long long* s_keys2 = s_keys + kSortLdsMax / 2;
    uint32_t* s_vals2 = s_vals + kSortLdsMax / 2;
    unsigned long long kr[4];
    uint32_t vr[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = tid + u * T; if (i < n) { kr[u] = s_keys[i]; vr[u] = s_vals[i]; } }
    __syncthreads();                                 // (the scatter's target overlaps nothing that is still being read: n <= 4096)
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = tid + u * T;
      if (i < n && (int)(kr[u] >> 56) % P == q) {
        const int pos = atomicAdd(&s_cls_cur[(int)(kr[u] >> 56)], 1);
        s_keys2[pos] = kr[u]; s_vals2[pos] = vr[u];
      }
    }
    __syncthreads();
    TSTAMP();
    // Every bucket is ordered by RANK COUNTING: a task is (class, 64 of its elements); each lane holds one of them and counts
    // the bucket's keys below its own -- the bucket is read 64 keys at a time (one LDS read per lane) and handed round with
    // v_readlane, so a comparison is two scalar broadcasts, one 64-bit compare and one add-with-carry, no LDS traffic.  Keys
    // are unique, so the rank is the element's place.  (The in-wave bitonic network on the same buckets: 11.3 us -- three
    // ds_bpermute per element and stage, 16 waves on one LDS; this: see DESIGN 4.2b.)
    const int wv = tid >> 6, lane = tid & 63;
    const int ntask = s_cls_ntask;
    for (int t = wv; t < ntask; t += T / 64) {
      int c = 0;
      {                                              // the class whose task range holds t (<= 256 classes: four per lane)
        u64 found = 0ull;
        int base_c = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int cc = u * 64 + lane;
          const bool in = t >= s_cls_task[cc] && t < s_cls_task[cc + 1];
          const u64 b = __ballot(in);
          if (b) { found = b; base_c = u * 64; }
        }
        c = base_c + __builtin_ctzll(found);
      }
      const int off = s_cls_off[c], cnt = s_hist[c], mine_i = (t - s_cls_task[c]) * 64 + lane;
      const bool have = mine_i < cnt;
      const unsigned long long mine = have ? s_keys2[off + mine_i] : 0ull;
      const uint32_t myv = have ? s_vals2[off + mine_i] : 0u;
      int rank = 0;
      for (int cb = 0; cb < cnt; cb += 64) {
        const unsigned long long ck = (cb + lane < cnt) ? s_keys2[off + cb + lane] : ~0ull;   // pads above every key: never counted
        const uint32_t clo = (uint32_t)ck, chi = (uint32_t)(ck >> 32);
        const int mm = min(64, cnt - cb);
        int j = 0;
        for (; j + 4 <= mm; j += 4) {
#pragma unroll
          for (int z = 0; z < 4; z++) {
            const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)chi, j + z) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)clo, j + z);
            rank += (kj < mine) ? 1 : 0;
          }
        }
        for (; j < mm; j++) {
          const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)chi, j) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)clo, j);
          rank += (kj < mine) ? 1 : 0;
        }
      }
      // the element's place is known: its sorted (key, slot) pair, its NMS record and its alive bit go out from here (the
      // generic tail below is the network's)
      if (have) {
        const int p = off + rank;
        keys_out[b0 + p] = mine; vals_out[b0 + p] = myv;
        const size_t ci = (size_t)b0 + myv;
        const float4 c0 = cand[ci * 2], c1 = cand[ci * 2 + 1];
        const float coff = c1.z * class_offset;                  // :849
        RBoxFeat f = rbox_make_feat(c0.x + coff, c0.y + coff, c0.z, c0.w, c1.x);
        float4 rq[4];
        RotGeom::pack(f, rq);
#pragma unroll
        for (int u = 0; u < 4; u++) rec[(size_t)(b0 + p) * 4 + u] = rq[u];
        const float mn = (c0.w < c0.z) ? c0.w : c0.z;
        if (!(mn < 0.001f)) atomicOr(&s_abits[p >> 6], 1ull << (p & 63));   // nms_rotated_wrapper.py:32
      }
    }
    __syncthreads();
    // the image's bitmap words were zeroed by k_reset_state; a word can hold positions of classes of several parts
    if (tid < kSortLdsMax / 2 / 64) { const u64 w = s_abits[tid]; if (w) atomicOr(alive + ((size_t)b0 >> 6) + tid, w); }
    if (g == bs - 1 && q == 0 && tid < 8) alive[(((size_t)bs * cap_img) >> 6) + tid] = 0ull;   // the bitmap's guard words
    TSTAMP();
#ifdef OBB_SORT_TRACE
    if (tid == 0 && g == 0) { printf("sortprep g %d part %d n %d by class, max %d:", g, q, n, s_cls_max); for (int z = 1; z < ti_; z++) printf(" %llu", tt[z] - tt[z - 1]); printf(" (x10 ns)\n"); }
#endif
    return;
  } else {
    switch (npad >> 10) {
      case 0: case 1: sort_lds_regs<1>(s_keys, s_vals, npad, tid); break;
      case 2: sort_lds_regs<2>(s_keys, s_vals, npad, tid); break;
      case 4: sort_lds_regs<4>(s_keys, s_vals, npad, tid); break;
      default: sort_lds_regs<8>(s_keys, s_vals, npad, tid); break;
    }
  }
  TSTAMP();
  for (int i = tid; i < n; i += T) { keys_out[b0 + i] = s_keys[i]; vals_out[b0 + i] = s_vals[i]; }
  // segment table (k_class_bounds): class runs of the sorted keys, or everything in segment 0
  if (m == 1) {
    for (int i = tid; i < e; i += T) {
      const int cls = (int)(s_keys[i] >> 56);
      if (i == 0 || cls != (int)(s_keys[i - 1] >> 56)) seg_begin[g * ncs + cls] = b0 + i;
      if (i == e - 1 || cls != (int)(s_keys[i + 1] >> 56)) seg_end[g * ncs + cls] = b0 + i + 1;
    }
  } else if (tid == 0) {
    seg_end[g * ncs] = b0 + e;
  }
  TSTAMP();
  // NMS records + every word of the image's part of the alive bitmap (k_prep_cand; cap_img is a multiple of 64)
  const int e64 = (e + 63) & ~63;
  for (long long w = (e64 >> 6) + tid; w < (cap_img >> 6); w += T) alive[((size_t)b0 >> 6) + w] = 0ull;   // words behind the boxes
  for (int base = 0; base < e64; base += T) {
    const int i = base + tid;
    bool ok = false;
    if (i < e) {
      const size_t ci = (size_t)b0 + s_vals[i];
      const float4 c0 = cand[ci * 2], c1 = cand[ci * 2 + 1];
      const float off = c1.z * class_offset;                     // :849
      RBoxFeat f = rbox_make_feat(c0.x + off, c0.y + off, c0.z, c0.w, c1.x);
      float4 q[4];
      RotGeom::pack(f, q);
#pragma unroll
      for (int u = 0; u < 4; u++) rec[(size_t)(b0 + i) * 4 + u] = q[u];
      const float mn = (c0.w < c0.z) ? c0.w : c0.z;
      ok = !(mn < 0.001f);                                       // nms_rotated_wrapper.py:32
    }
    const u64 bits = __ballot(ok);
    if ((tid & 63) == 0 && i < e64) alive[(size_t)(b0 + i) >> 6] = bits;
  }
  TSTAMP();
  if (g == bs - 1 && tid < 8) alive[(((size_t)bs * cap_img) >> 6) + tid] = 0ull;   // the bitmap's guard words
#ifdef OBB_SORT_TRACE
  TSTAMP();
  if (tid == 0 && g == 0) { printf("sortprep g %d n %d by_class %d max %d:", g, n, (int)by_class, s_cls_max); for (int q = 1; q < ti_; q++) printf(" %llu", tt[q] - tt[q - 1]); printf(" (x10 ns)\n"); }
#endif
#undef TSTAMP
}

// Output: the kept boxes of the image's segments merged into descending-score order (the order of the reference's single
// greedy pass), first max_det of them.  One workgroup per image.  Each segment's kept list is already in ascending key
// order; the global rank of an entry = its index in its own list + the number of entries of every other list that
// precede it -- binary searches on merge keys (score, anchor, class) staged in LDS.
constexpr int kMergeLds = 4096;
__global__ __launch_bounds__(256) void k_gather_out(const float4* __restrict__ cand, const uint32_t* __restrict__ vals_sorted,
                                                    const unsigned long long* __restrict__ keys_sorted, const int64_t* __restrict__ keep,
                                                    const int* __restrict__ seg_begin, const int* __restrict__ keep_cnt,
                                                    const int* __restrict__ mode, int ncs, long long max_det, float* __restrict__ out,
                                                    int64_t* __restrict__ out_count, const int* __restrict__ cnt, long long cap_img,
                                                    int64_t* __restrict__ status, const int* __restrict__ abort_flag, int packed,
                                                    const int* __restrict__ info, const int* __restrict__ tiny, int* __restrict__ clean_cnt,
                                                    int* __restrict__ clean_tiny) {
  __shared__ int s_pre[257], s_seg[256];
  __shared__ long long s_rows[4], s_mx[4];
  __shared__ int s_tf[4];
  __shared__ unsigned long long s_key[kMergeLds];
  __shared__ uint32_t s_val[kMergeLds];
  const int g = blockIdx.x, tid = threadIdx.x;
  // blockIdx.y = part of gridDim.y: the entries of an image with many kept rows are shared out (every part repeats the prologue;
  // part 0 writes the counters).  One part per image in the small-image regime.
  const int part = blockIdx.y, nparts = gridDim.y;
  // The kernel is a chain of dependent global reads (counts -> kept positions -> keys / slots -> candidate rows); everything
  // that does not depend on an earlier read is requested up front: seven latencies became four.
  // kept per segment (clipped to max_det: a class contributes at most max_det rows to the first max_det overall)
  for (int c = tid; c < ncs; c += 256) {
    long long k = keep_cnt[g * ncs + c];
    if (max_det > 0 && k > max_det) k = max_det;
    s_pre[c + 1] = (int)k;
    s_seg[c] = seg_begin[g * ncs + c];
  }
  if (tid == 0) s_pre[0] = 0;
  // (packed) rows of the images before this one; (workgroup 0) the largest candidate count -- same round trip as above
  long long mine = 0, mx = 0;
  int tf = 0;
  if (packed) {
    for (int b2 = tid; b2 < g; b2 += 256) {
      long long t = 0;
      for (int c = 0; c < ncs; c++) { long long k = keep_cnt[b2 * ncs + c]; if (max_det > 0 && k > max_det) k = max_det; t += k; }
      if (max_det > 0 && t > max_det) t = max_det;
      mine += t;
    }
  }
  if (g == 0)
    for (int b2 = tid; b2 < (int)gridDim.x; b2 += 256) {
      const long long c = cnt[b2 * kCntPad];
      if (c > mx) mx = c;
      const int t = tiny[b2];
      tf |= t & kImgSmall;
      if ((t & kImgSmall) && !img_single_list(t)) tf |= 16;      // ... and kept its class segments (k_tiny_cross vouched for it)
    }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { mine += __shfl_xor(mine, d); const long long o = __shfl_xor(mx, d); if (o > mx) mx = o; tf |= __shfl_xor(tf, d); }
  if ((tid & 63) == 0) { s_rows[tid >> 6] = mine; s_mx[tid >> 6] = mx; s_tf[tid >> 6] = tf; }
  __syncthreads();
  if (tid == 0) for (int c = 0; c < ncs; c++) s_pre[c + 1] += s_pre[c];
  __syncthreads();
  const int total = s_pre[ncs];
  if (tid == 0 && part == 0) {
    long long nk = total;
    if (max_det > 0 && nk > max_det) nk = max_det;
    out_count[g] = *abort_flag ? -1 : nk;                        // -1: the NMS kernel gave up on a barrier (host raises)
  }
  // status[0]: the largest candidate count if an image overflowed its slots (the caller retries), else 0; status[1]: the largest
  // candidate count (feedback for the caller's next call).  Plain stores by one workgroup -- out_count and status may be pinned
  // host memory (the host layer reads them without a device->host copy).
  if (g == 0 && tid == 0 && part == 0) {
    long long m4 = s_mx[0];
    for (int k = 1; k < 4; k++) if (s_mx[k] > m4) m4 = s_mx[k];
    // info[8] != 0: k_nms_small met a segment above its limit -- nothing of this call is valid, the caller repeats it with
    // the persistent kernel (status[0] = -1); info[4]: the largest NMS segment where the sort kernel knew it (else 0)
    status[0] = info[8] ? -1 : (m4 > cap_img ? m4 : 0);
    // bit 62: an image of this call held short-sided boxes (kImgSmall) -- the caller's next call asks for k_tiny_cross;
    // bit 61: such an image kept its class segments in THIS call (informational)
    const int tfa = s_tf[0] | s_tf[1] | s_tf[2] | s_tf[3];
    const long long small_seen = ((tfa & kImgSmall) ? (1ll << 62) : 0ll) | ((tfa & 16) ? (1ll << 61) : 0ll);
    status[1] = m4 | (((long long)(info[8] > info[4] ? info[8] : info[4]) & 0x1fffffffll) << 32) | small_seen;
  }
  // caller-kept counters (obb_non_max_suppression_obb_st) are left zeroed for the next call: this workgroup was their last
  // reader whose reading matters (the other parts of image 0 read them too, but only this one writes the status words)
  if (clean_cnt != nullptr && g == 0 && part == 0)
    for (int b2 = tid; b2 < (int)gridDim.x; b2 += 256) { clean_cnt[b2 * kCntPad] = 0; clean_tiny[b2] = 0; }
  // first output row of the image: g * max_det, or (packed) the number of rows of the images before it
  const long long row0 = packed ? s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3] : (long long)g * max_det;
  const int md = mode[g];
  const bool single = md == 0;
  // merge key of entry e = (class c, index k): the mode-1 key rotated so that it orders by (score, anchor, class);
  // mode 2 keeps the single-list key (score, anchor*nc + class), which already is the global order
  auto entry_pos = [&](int c, int k) -> uint32_t { return (uint32_t)keep[(size_t)s_seg[c] + k]; };
  auto mkey_at = [&](uint32_t p) -> unsigned long long {
    const unsigned long long k = keys_sorted[p];
    return md == 1 ? ((k << 8) | (k >> 56)) : k;
  };
  const bool in_lds = total <= kMergeLds;
  if (!single && in_lds) {
    for (int e = tid; e < total; e += 256) {
      int c = 0;
      while (s_pre[c + 1] <= e) c++;
      const uint32_t p = entry_pos(c, e - s_pre[c]);
      s_key[e] = mkey_at(p);
      s_val[e] = vals_sorted[p];
    }
    __syncthreads();
  }
  for (int e = tid + 256 * part; e < total; e += 256 * nparts) {
    int c = 0;
    while (s_pre[c + 1] <= e) c++;
    const int k = e - s_pre[c];
    const bool staged = !single && in_lds;                      // key and slot of the entry are in LDS
    const uint32_t p = staged ? 0u : entry_pos(c, k);
    long long rank = k;
    if (!single) {
      const unsigned long long mk = in_lds ? s_key[e] : mkey_at(p);
      for (int c2 = 0; c2 < ncs; c2++) {
        if (c2 == c) continue;
        int lo = 0, hi = s_pre[c2 + 1] - s_pre[c2];
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const unsigned long long mk2 = in_lds ? s_key[s_pre[c2] + mid] : mkey_at(entry_pos(c2, mid));
          if (mk2 < mk) lo = mid + 1; else hi = mid;
        }
        rank += lo;
      }
    }
    if (max_det > 0 && rank >= max_det) continue;
    const size_t ci = (size_t)g * cap_img + (staged ? s_val[e] : vals_sorted[p]);
    const float4 c0 = cand[ci * 2], c1 = cand[ci * 2 + 1];
    float* o = out + ((size_t)row0 + rank) * 7;
    o[0] = c0.x; o[1] = c0.y; o[2] = c0.z; o[3] = c0.w; o[4] = c1.x; o[5] = c1.y; o[6] = c1.z;
  }
}

// The FRONT of k_nms_small without a sort launch (round 6, fourth part).  At the reference's default thresholds an image holds a few
// thousand candidates in ~16 classes: k_sort_prep_lds spent 17 us (+ its launch) on four workgroups per image ordering what the 256
// segment workgroups of the NMS launch can order themselves -- every workgroup reads its image's keys (8 bytes per candidate, from
// L2), keeps the candidates of ITS class (the class is the key's tie word modulo nc; label rows carry it in their record), ranks
// them by counting (keys are unique: the rank is the place; <= 384 members, the keys broadcast from LDS two at a time) and builds
// records, alive bits (nms_rotated_wrapper.py:32) and publishing keys where the pair phase wants them: in LDS.  Same keys, same
// order, same records as the sort kernel's (k_sort_prep_lds: "by_class"); an image that keeps ONE list (img_single_list, more than
// max_nms candidates) is segment 0's if it fits.  What does not fit a segment raises too_big like a segment of the sort's: the caller
// repeats the call on the persistent kernel.
// (compile-time switches of the front end, measured within 1 us of each other on the bs16 step: keys per thread and trip, the first
//  trip requested together with the image's counter, the rank count with eight reads in flight)
#ifndef OBB_SELF_KPT
#define OBB_SELF_KPT 8
#endif
#ifndef OBB_SELF_SPEC
#define OBB_SELF_SPEC 1
#endif
#ifndef OBB_SELF_RU
#define OBB_SELF_RU 1
#endif
struct SmallSelfSort {
  static constexpr bool kSelf = true;
  // an image's candidate count, its top-max_nms cut and its mode exactly as k_sort_prep_lds decides them (class_ok holds: the
  // small-segment kernel is only chosen with it)
  static __device__ __forceinline__ int image_mode(long long c, int tf, long long cap_img, long long max_nms, int& n, int& e) {
    const bool over_cap = c > cap_img;
    if (over_cap) c = cap_img;
    if (c > kSortLdsMax) c = 0;                                  // not this path's regime: reported through status[1]
    const bool over_nms = max_nms > 0 && c > max_nms;
    e = (int)(over_nms ? max_nms : c);                           // :845-846
    n = (int)c;
    return (!over_cap && !img_single_list(tf) && !over_nms) ? 1 : 0;
  }
  // returns boxes of the segment (> kSmallMax: too many, nothing was built) | mode of the image << 16
  template <class G>
  static __device__ __forceinline__ int load(const SmallArgs& a, int seg, float4* s_rec, u64* s_mask, unsigned long long* s_pk,
                                                       uint32_t* s_pv, int* s_n) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = seg / a.ncs, c = seg - g * a.ncs;
    constexpr int KPT = OBB_SELF_KPT;                            // keys per thread and trip
    const size_t b0 = (size_t)g * (size_t)a.cap_img;
    // the first trip's keys are requested TOGETHER with the image's counter (slots behind the count hold stale keys of the
    // workspace: read, never used) -- one round trip less in front of every segment
    const int spec = !OBB_SELF_SPEC ? 0 : (a.cap_img < (long long)KPT * kSmallThreads ? (int)a.cap_img : KPT * kSmallThreads);
    unsigned long long k[KPT];
#pragma unroll
    for (int u = 0; u < KPT; u++) { const int i = u * kSmallThreads + tid; k[u] = i < spec ? a.keys_in[b0 + i] : 0ull; }
    int n, e;
    const int m = image_mode(a.cnt[g * kCntPad], a.tiny[g], a.cap_img, a.max_nms, n, e);
    if (n == 0 || (m == 0 && c != 0)) return m << 16;            // (workgroup-uniform)
    u64* s_alive = s_mask + (size_t)kSmallMax * kSmallWords;
    unsigned long long* st_key = reinterpret_cast<unsigned long long*>(s_mask);        // staged members: the bit matrix is not in use yet
    uint32_t* st_slot = reinterpret_cast<uint32_t*>(st_key + kSmallMax);
    if (tid == 0) *s_n = 0;
    if (tid < 8) s_alive[tid] = 0ull;
    __syncthreads();
    const uint32_t nc = (uint32_t)a.nc, lim = (uint32_t)((unsigned long long)a.A * nc);   // (A * nc + label rows < 2^32: checked by the launcher)
    for (int i0 = 0; i0 < n; i0 += KPT * kSmallThreads) {        // (workgroup-uniform trip count: ballots inside)
      if (i0 > 0 || !OBB_SELF_SPEC) {
#pragma unroll
        for (int u = 0; u < KPT; u++) { const int i = i0 + u * kSmallThreads + tid; k[u] = i < n ? a.keys_in[b0 + i] : 0ull; }
      }
#pragma unroll
      for (int u = 0; u < KPT; u++) {
        const int i = i0 + u * kSmallThreads + tid;
        if (i0 + u * kSmallThreads >= n) break;                  // (workgroup-uniform)
        bool mine = false;
        unsigned long long nk = k[u];
        if (i < n) {
          if (m == 1) {                                          // k_rekey: (cls << 56 | score_desc << 24 | anchor)
            const uint32_t tie = (uint32_t)k[u];
            uint32_t cls, anchor;
            if (tie < lim) { anchor = tie / nc; cls = tie - anchor * nc; }
            else { cls = (uint32_t)(int)a.cand[(b0 + i) * 2 + 1].z; anchor = (uint32_t)a.A + (tie - lim); }     // (an apriori label row)
            mine = cls == (uint32_t)c;
            nk = ((unsigned long long)cls << 56) | ((k[u] >> 32) << 24) | (unsigned long long)(anchor & 0xffffffu);
          } else mine = true;
        }
        const u64 mk = __ballot(mine);
        if (mk) {                                                // (wave-uniform)
          int base = 0;
          if (lane == 0) base = atomicAdd(s_n, __popcll(mk));
          base = __shfl(base, 0);
          const int at = base + __popcll(mk & lanemask_lt());
          if (mine && at < kSmallMax) { st_key[at] = nk; st_slot[at] = (uint32_t)i; }
        }
      }
    }
    __syncthreads();
    const int nm = *s_n;
    if (nm > kSmallMax) return (nm > 0xffff ? 0xffff : nm) | (m << 16);
    const int ec = (m == 1 || nm < e) ? nm : e;                  // one list: the first max_nms by key
    if (tid == 0 && nm > 0) atomicMax(a.seg_max, ec);
    if (tid < nm) {
      const unsigned long long mine = st_key[tid];
      const uint32_t slot = st_slot[tid];
      const size_t ci = b0 + slot;
      const float4 c0 = a.cand[ci * 2], c1 = a.cand[ci * 2 + 1]; // under way during the count
      int rank = 0, j = 0;
      const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(st_key);
      for (; OBB_SELF_RU && j + 16 <= nm; j += 16) {                            // eight 16-byte broadcast reads in flight (one at a time: 64 cycles each)
        ulonglong2 v[8];
#pragma unroll
        for (int z = 0; z < 8; z++) v[z] = p2[(j >> 1) + z];
#pragma unroll
        for (int z = 0; z < 8; z++) rank += (v[z].x < mine ? 1 : 0) + (v[z].y < mine ? 1 : 0);
      }
      for (; j + 2 <= nm; j += 2) { const ulonglong2 v = p2[j >> 1]; rank += (v.x < mine ? 1 : 0) + (v.y < mine ? 1 : 0); }
      if (j < nm) rank += st_key[j] < mine ? 1 : 0;
      if (rank < ec) {
        s_pk[rank] = mine; s_pv[rank] = slot;
        const float coff = c1.z * a.class_offset;                // :849
        const RBoxFeat f = rbox_make_feat(c0.x + coff, c0.y + coff, c0.z, c0.w, c1.x);
        float4 rq[4];
        G::pack(f, rq);
#pragma unroll
        for (int u = 0; u < 4; u++) s_rec[rank * G::RECQ + u] = rq[u];
        const float mn = (c0.w < c0.z) ? c0.w : c0.z;
        if (!(mn < 0.001f)) atomicOr(&s_alive[rank >> 6], 1ull << (rank & 63));        // nms_rotated_wrapper.py:32
      }
    }
    __syncthreads();                                             // the staged members are read: their place becomes the bit matrix
    for (int t = tid; t < ec * kSmallWords; t += kSmallThreads) s_mask[t] = 0ull;
    return ec | (m << 16);
  }
};

// The same output stage INSIDE k_nms_small (its TAIL): the segments of an image count themselves on a ticket and the workgroup
// that arrives last merges the image's kept lists and writes its rows -- the launch of k_gather_out, its start-up and two of its
// four dependent round trips (kept positions -> keys / slots: the segments publish merge keys and slots themselves) are gone
// from the step.  Rows of image g start at g * max_det (a packed output needs the totals of ALL earlier images: that stays with
// k_gather_out).  The image that finishes last writes the two status words.  No workgroup waits for another one.
struct SmallGather {
  struct Args {
    const float4* cand; const int* cnt; const int* tiny; const int* info;
    int* ticket;               // [bs] segments of the image that are done, then [1] images that are done (zeroed by k_reset_state or k_decode)
    float* out; int64_t* out_count; int64_t* status;
    long long cap_img, max_det;
    int bs;
    int* clean_cnt; int* clean_tiny;   // caller-kept counters to leave zeroed (obb_non_max_suppression_obb_st), or NULL
  };
  static constexpr int kLds = kSortLdsMax;                       // entries staged in LDS: an image of the in-LDS sort has no more candidates
  static constexpr size_t kLdsBytes = 2064 + (size_t)kLds * 16;
  static __device__ __forceinline__ void run(const SmallArgs& a, const Args& ga, unsigned char* s_raw, int seg) {
    __shared__ int s_flag, s_maxc;
    const int tid = threadIdx.x, lane = tid & 63, ncs = a.ncs, g = seg / ncs;
#ifdef OBB_SMALL_TRACE
    unsigned long long tt[10]; int ti_ = 0;
#define GSTAMP() do { tt[ti_++] = wall_clock64(); } while (0)
#else
#define GSTAMP() do {} while (0)
#endif
    GSTAMP();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's published entries are written through
    __syncthreads();
    // The stage is a chain of dependent round trips (~1 us each), so everything is requested as early as its address is known:
    // thread (c, k) of the first `spec` threads per segment asks for the k-th published entry of segment c TOGETHER with the
    // segment's count (most segments keep fewer than `spec` boxes: no second trip); the segment's first position comes from the
    // sort kernel and is requested while the ticket is under way.
    int spec = 2;
    while (spec * 2 * ncs <= kSmallThreads) spec <<= 1;          // (ncs <= 256)
    const int myc = tid / spec, myk = tid - myc * spec;
    const bool mine = myc < ncs;
    const bool self = a.keys_in != nullptr;                      // (kernel-uniform) self-sorting segments publish at fixed places
    // (the image's mode: from its counter, read BEFORE this workgroup arrives -- the image that is done last zeroes the counters)
    int single_self = 0;
    if (self) { int n_, e_; single_self = SmallSelfSort::image_mode(ga.cnt[g * kCntPad], ga.tiny[g], ga.cap_img, a.max_nms, n_, e_) == 0 ? 1 : 0; }
    asm volatile("" : "+v"(single_self));
    const int sb_c = !mine ? 0 : (self ? (int)((long long)g * ga.cap_img) + myc * kSmallMax : a.seg_begin[g * ncs + myc]);
    int t2 = -1;                                                 // (thread 0) images that were done before this one
    if (tid == 0) {
      const int last = __hip_atomic_fetch_add(ga.ticket + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ncs - 1 ? 1 : 0;
      // every segment of this image is done -- the image counts as done for the status words (which do not wait for its rows)
      if (last) t2 = __hip_atomic_fetch_add(ga.ticket + ga.bs, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_flag = last;
    }
    __syncthreads();
    if (!s_flag) return;                                         // (workgroup-uniform)
    GSTAMP();
    // ---- the image's output (the segment state in s_raw is dead).  What other workgroups of this launch wrote is read with
    // agent-scope loads (they wrote it through); the rest comes from earlier launches.
    int* s_pre = reinterpret_cast<int*>(s_raw);                  // [257]
    int* s_seg = s_pre + 260;                                    // [256]
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_raw + 2064);
    uint32_t* s_val = reinterpret_cast<uint32_t*>(s_key + kLds);
    uint32_t* s_ck = s_val + kLds;                               // (segment << 16 | index in its list) of a staged entry
    const long long max_det = ga.max_det;
    int cnt_c = 0;
    unsigned long long key = 0ull;
    uint32_t val = 0u;
    if (mine) {
      cnt_c = ldg_agent(a.keep_cnt + g * ncs + myc);
      size_t p = (size_t)sb_c + myk;
      if (p >= (size_t)a.n_pos) p = (size_t)a.n_pos - 1;        // (beyond the segment's count the entry is not used)
      val = ldg_agent(a.pub_val + p);
      key = ldg_agent(a.pub_key + p);
    }
    // (wave 0) the inputs of the status words come from earlier launches: under way during the trips above
    long long mx = 0;
    int tf = 0, big = 0;
    if (tid < 64) {
      for (int b2 = lane; b2 < ga.bs; b2 += 64) {
        const long long c = ga.cnt[b2 * kCntPad];
        if (c > mx) mx = c;
        const int t = ga.tiny[b2];
        tf |= t & kImgSmall;
        if ((t & kImgSmall) && !img_single_list(t)) tf |= 16;
      }
      // the image that is done last reads the too-big word now: every segment of every image has arrived
      if (tid == 0 && t2 == ga.bs - 1) big = ldg_agent(ga.info + 8);
    }
    if (max_det > 0 && cnt_c > max_det) cnt_c = (int)max_det;   // a class contributes at most max_det rows to the first max_det overall
    if (mine && myk == 0) { s_pre[myc + 1] = cnt_c; s_seg[myc] = sb_c; }
    if (tid == 0) s_pre[0] = 0;
    const int more = __syncthreads_or(cnt_c > spec ? 1 : 0);
    if (tid < 64) {                                              // inclusive scan of s_pre[1 .. ncs], four per lane (ncs <= 256), and the largest
      int v[4], sum = 0, m = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) { const int c = lane * 4 + j; v[j] = c < ncs ? s_pre[c + 1] : 0; sum += v[j]; m = v[j] > m ? v[j] : m; }
      int inc = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(m, d); m = o > m ? o : m; }
      int run = inc - sum;
#pragma unroll
      for (int j = 0; j < 4; j++) { const int c = lane * 4 + j; run += v[j]; if (c < ncs) s_pre[c + 1] = run; }
      if (lane == 0) s_maxc = m;
    }
    __syncthreads();
    const int total = s_pre[ncs], maxc = s_maxc;
    GSTAMP();
    if (tid == 0) ga.out_count[g] = (max_det > 0 && total > max_det) ? max_det : (long long)total;
    const bool single = self ? single_self != 0 : a.mode[g] == 0;
    const bool in_lds = total <= kLds;
    auto seg_of = [&](int e) -> int {                            // the segment c with s_pre[c] <= e < s_pre[c + 1]
      int lo = 0, hi = ncs - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[mid] <= e) lo = mid; else hi = mid - 1; }
      return lo;
    };
    if (in_lds) {
      if (mine && myk < cnt_c) { const int e = s_pre[myc] + myk; s_val[e] = val; s_key[e] = key; s_ck[e] = ((uint32_t)myc << 16) | (uint32_t)myk; }
      if (more)                                                  // (workgroup-uniform) a segment kept more than `spec` boxes: the rest of it
        for (int e = tid; e < total; e += kSmallThreads) {
          const int c = seg_of(e), k = e - s_pre[c];
          if (k < spec) continue;
          const size_t p = (size_t)s_seg[c] + k;
          s_val[e] = ldg_agent(a.pub_val + p);
          s_key[e] = ldg_agent(a.pub_key + p);
          s_ck[e] = ((uint32_t)c << 16) | (uint32_t)k;
        }
      __syncthreads();
    }
    GSTAMP();
    // rank of an entry = its index in its own list + the entries of every other list in front of it (binary searches on the
    // merge keys); up to 16 lanes share the lists of one entry when the image has few entries.  A lane runs the searches of eight
    // lists in lock step (same number of halving steps for all: the loads of a step are independent of each other).
    int tpe = 1;
    if (!single) while (tpe < 16 && total * tpe * 2 <= kSmallThreads) tpe <<= 1;
    int top = 1;
    while (top * 2 <= maxc) top <<= 1;                           // the largest power of two <= the longest list
    const int sub = tid & (tpe - 1), per_round = kSmallThreads / tpe, nlist = (ncs + tpe - 1) / tpe;
    for (int e0 = 0; e0 < total; e0 += per_round) {
      const int e = e0 + tid / tpe;
      const bool valid = e < total;
      int c = 0, k = 0, rank = 0;
      if (valid) {
        if (in_lds) { const uint32_t ck = s_ck[e]; c = (int)(ck >> 16); k = (int)(ck & 0xffffu); }
        else { c = seg_of(e); k = e - s_pre[c]; }
      }
      const bool writer = valid && sub == 0;
      float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
      if (writer) {                                              // the row is under way while the rank is computed
        const uint32_t slot = in_lds ? s_val[e] : ldg_agent(a.pub_val + (size_t)s_seg[c] + k);
        const size_t ci = (size_t)g * ga.cap_img + slot;
        c0 = ga.cand[ci * 2]; c1 = ga.cand[ci * 2 + 1];
      }
      GSTAMP();
      if (valid && !single) {
        if (in_lds) {
          const unsigned long long mk = s_key[e];
          for (int j0 = 0; j0 < nlist; j0 += 8) {
            int base[8], len[8], pos[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int c2 = sub + (j0 + j) * tpe;
              const bool ok = c2 < ncs && c2 != c;
              base[j] = ok ? s_pre[c2] : 0;
              len[j] = ok ? s_pre[c2 + 1] - base[j] : 0;
              pos[j] = 0;
            }
            for (int st = top; st >= 1; st >>= 1) {
              unsigned long long k2[8];                          // (all eight loads first: written as one conditional per list the
#pragma unroll                                                   //  compiler branches around each load and waits for it in turn)
              for (int j = 0; j < 8; j++) {
                const int q = pos[j] + st;
                int at = base[j] + (q <= len[j] ? q : len[j]) - 1;
                at = at < 0 ? 0 : at;
                k2[j] = s_key[at];
              }
#pragma unroll
              for (int j = 0; j < 8; j++) {
                const int q = pos[j] + st;
                const bool take = (q <= len[j]) & (k2[j] < mk);
                pos[j] = take ? q : pos[j];
              }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) rank += pos[j];
          }
        } else {
          const unsigned long long mk = ldg_agent(a.pub_key + (size_t)s_seg[c] + k);
          for (int c2 = sub; c2 < ncs; c2 += tpe) {
            if (c2 == c) continue;
            int lo = 0, hi = s_pre[c2 + 1] - s_pre[c2];
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (ldg_agent(a.pub_key + (size_t)s_seg[c2] + mid) < mk) lo = mid + 1; else hi = mid;
            }
            rank += lo;
          }
        }
      }
      GSTAMP();
      for (int d = 1; d < tpe; d <<= 1) rank += __shfl_xor(rank, d);
      rank += k;
#ifdef OBB_SMALL_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      GSTAMP();
#endif
      if (!writer || (max_det > 0 && rank >= max_det)) continue;
      float* o = ga.out + ((size_t)g * max_det + rank) * 7;
      o[0] = c0.x; o[1] = c0.y; o[2] = c0.z; o[3] = c0.w; o[4] = c1.x; o[5] = c1.y; o[6] = c1.z;
    }
    GSTAMP();
#ifdef OBB_SMALL_TRACE
    if (tid == 0 && blockIdx.x < 2048) { unsigned long long* o_ = g_small_trace_tail + blockIdx.x * 4; o_[0] = 1ull + (unsigned long long)g; o_[1] = (unsigned long long)total; o_[2] = tt[0]; o_[3] = tt[7]; }
#endif
#undef GSTAMP
    // ---- the call's status words, by the image that was done last (k_gather_out: workgroup 0; see there for their meaning)
    if (tid >= 64) return;
    const int writes = __shfl(t2, 0) == ga.bs - 1;               // (wave-uniform)
    if (!writes) return;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const long long o = __shfl_xor(mx, d); if (o > mx) mx = o; tf |= __shfl_xor(tf, d); }
    // (the counters were read above; the images still at work on their rows do not use theirs)
    if (ga.clean_cnt != nullptr)
      for (int b2 = lane; b2 < ga.bs; b2 += 64) { ga.clean_cnt[b2 * kCntPad] = 0; ga.clean_tiny[b2] = 0; }
    if (lane == 0) {
      const int seg4 = ldg_agent(ga.info + 4);                  // (self-sorting segments raise it in this launch)
      ga.status[0] = big ? -1 : (mx > ga.cap_img ? mx : 0);
      const long long small_seen = ((tf & kImgSmall) ? (1ll << 62) : 0ll) | ((tf & 16) ? (1ll << 61) : 0ll);
      ga.status[1] = mx | (((long long)(big > seg4 ? big : seg4) & 0x1fffffffll) << 32) | small_seen;
    }
  }
};
static_assert(SmallGather::kLdsBytes <= (size_t)kSmallMax * RotGeom::RECQ * 16 + (size_t)kSmallMax * kSmallWords * 8 + 64 + sizeof(SmallWave<RotGeom>) * kSmallWaves,
              "SmallGather: the merge lists live where the segment state was");

static inline size_t obb_state_bytes(int64_t bs) { return align_up((size_t)bs * kCntPad * 4 + (size_t)bs * 4); }   // cnt lines + tiny flags

struct ObbCarve {
  float4* cand; unsigned long long *keys_a, *keys_b; uint32_t *vals_a, *vals_b; int* cnt; int *sort_begin, *sort_end;
  int *img_end, *mode, *tiny, *grp_begin, *grp_end, *ticket;
  uint32_t *srs_hist, *digit_base;
  int64_t* keep;
  Carve nms;          // rec/dead/segment state reuse the NMS carve (keys/vals/sort_tmp of it unused)
  void* nms_base;
  size_t total;
};

static inline int64_t round_cap(int64_t cap_img) { return (cap_img + 63) / 64 * 64; }   // image regions start on alive-bitmap words

static int obb_carve(void* base, int64_t bs, int64_t cap_img, int64_t ncs, ObbCarve* cv) {
  cap_img = round_cap(cap_img);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return base ? (char*)base + o : (char*)nullptr; };
  const size_t n = (size_t)bs * cap_img;
  cv->cand = (float4*)take(n * 32);
  cv->keys_a = (unsigned long long*)take(n * 8); cv->keys_b = (unsigned long long*)take(n * 8);
  cv->vals_a = (uint32_t*)take(n * 4); cv->vals_b = (uint32_t*)take(n * 4);
  cv->cnt = (int*)take(bs * 4 * kCntPad); cv->sort_begin = (int*)take(bs * 4); cv->sort_end = (int*)take(bs * 4);
  cv->img_end = (int*)take(bs * 4); cv->mode = (int*)take(bs * 4); cv->tiny = (int*)take(bs * 4);
  cv->grp_begin = (int*)take(bs * 4); cv->grp_end = (int*)take(bs * 4);
  cv->ticket = (int*)take(64 + (size_t)(bs + 1) * 4);         // 16 words + k_nms_small's output stage: [bs] + [1]
  cv->digit_base = (uint32_t*)take((size_t)bs * 256 * 4);
  cv->srs_hist = (uint32_t*)take((size_t)bs * ((size_t)(cap_img + kSrsTile - 1) / kSrsTile) * 256 * 4);
  cv->keep = (int64_t*)take(n * 8);
  // NMS state sized for bs * cap_img positions
  cv->nms_base = base ? (char*)base + off : nullptr;
  int rc = carve(cv->nms_base, (int64_t)n, bs * ncs, RotGeom::RECQ, cap_max(bs * ncs), &cv->nms);
  if (rc) return rc;
  off += cv->nms.total;
  cv->total = off;
  return OBB_OK;
}

static int run_nms_obb(const void* pred, const void* objcol, int dtype, int64_t bs, int64_t A, int64_t no, float conf_thres, float iou_thres,
                       const int32_t* classes_host, int n_classes, int agnostic, int multi_label, int64_t max_det, int64_t max_nms,
                       float max_wh, const float* extra8, int64_t n_extra, int64_t cap_img, int64_t expected_cand, float* out,
                       int out_packed, int64_t* out_count, int64_t* status, void* ws, size_t ws_bytes, hipStream_t st, void* state = nullptr,
                       size_t state_bytes = 0) {
  const int nc = (int)(no - 5 - 180);                              // :784
  if (bs < 1 || A < 1 || nc < 1 || nc > 256 || max_det < 1 || cap_img < 1 || !pred || !out || !out_count || !status)
    return OBB_ERR_BAD_ARG;
  if (A * nc + n_extra > 0xffffffffLL || bs * cap_img > 0x7fffffffLL) return OBB_ERR_BAD_ARG;
  if (dtype != 0 && dtype != 1) return OBB_ERR_BAD_ARG;
  // expected_cand: low 32 bits = the previous call's largest candidate count of an image (0: unknown), high 32 bits = its
  // largest NMS segment (0: unknown) -- both as status[1] reported them
  const int64_t seg_hint = (expected_cand >> 32) & 0x1fffffff;
  const bool small_hint = ((expected_cand >> 62) & 1) != 0;        // the previous call met short-sided boxes: run k_tiny_cross
  expected_cand &= 0xffffffffll;
  cap_img = round_cap(cap_img);
  const int ncs = agnostic ? 1 : nc;                               // NMS segments per image
  // class segmentation needs the class in 8 and the anchor index in 24 key bits
  static const int no_class_seg = obb_dev_switch("OBB_NO_CLASS_SEG", 0) != 0;    // A/B switch (development builds)
  // (iou_thres < 0: IoU = 0 > thr, boxes of different classes DO suppress each other in the reference's single list)
  const int class_ok = (!no_class_seg && !agnostic && nc > 1 && A + n_extra < (1ll << 24) && iou_thres >= 0.f && max_wh > 0.f) ? 1 : 0;
  // the class-grouping pass for images with more than max_nms candidates is only worth launching when such images are expected
  // (off by default: per-class segments cannot share the max_det early stop of the single list, which usually ends the
  //  NMS of such images after the first ~2000 of 30000 candidates; OBB_NMS_GROUP_AFTER_CUT=1 enables it)
  static const int group_cut = obb_dev_switch("OBB_NMS_GROUP_AFTER_CUT", 0) != 0;
  const int group_ok = (group_cut && class_ok && max_nms > 0 && expected_cand > max_nms) ? 1 : 0;
  ObbCarve cv;
  int rc = obb_carve(ws, bs, cap_img, ncs, &cv);
  if (rc) return rc;
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  // caller-kept counters: zero when the call starts (the caller's memset, or the previous call's last kernel) -- no reset launch
  const bool kept = state != nullptr;
  if (kept) {
    if (state_bytes < obb_state_bytes(bs) || ((uintptr_t)state & 255u)) return OBB_ERR_WORKSPACE;
    cv.cnt = (int*)state;
    cv.tiny = cv.cnt + bs * kCntPad;
  }

  DecodeArgs d;
  d.pred = pred; d.objcol = objcol; d.A = A; d.no = (int)no; d.nc = nc; d.bs = (int)bs; d.conf_thres = conf_thres;
  d.multi_label = (multi_label && nc > 1) ? 1 : 0;                 // :797
  d.cm.all = (classes_host == nullptr || n_classes <= 0) ? 1 : 0;
  for (int i = 0; i < 4; i++) d.cm.w[i] = 0ull;
  for (int i = 0; i < n_classes && classes_host; i++) {
    int c = classes_host[i];
    if (c >= 0 && c < 256) d.cm.w[c >> 6] |= 1ull << (c & 63);
  }
  d.cap_img = cap_img; d.cand = cv.cand; d.keys = cv.keys_a; d.vals = cv.vals_a; d.cnt = cv.cnt; d.tiny = cv.tiny;
  d.win_lo = -0.35f * max_wh; d.win_hi = 0.6f * max_wh;            // 0.95 max_wh wide: circles of different classes cannot touch

  static const int no_lds_sort = obb_dev_switch("OBB_NO_LDS_SORT", 0) != 0;      // A/B switch (development builds)
  const bool lds_sort = !no_lds_sort && expected_cand > 0 && expected_cand <= kSortLdsHint && !group_ok;
  d.z_ticket = nullptr; d.n_ticket = 0; d.z_bar16 = nullptr; d.n_bar16 = 0; d.z_alive16 = nullptr; d.n_alive16 = 0;
  {
    Carve& nv0 = cv.nms;
    // (cap_img is a multiple of 64 -> bs * cap_img / 64 words; + the guard words, rounded up to 16 bytes)
    const long long alive16 = lds_sort ? (long long)((((size_t)bs * cap_img) >> 6) + 8 + 1) / 2 : 0ll;
    if (kept) {                                                    // the filter kernel zeroes what the later launches need
      d.z_ticket = cv.ticket; d.n_ticket = (int)(16 + bs + 1);
      d.z_bar16 = reinterpret_cast<uint4*>(nv0.bar); d.n_bar16 = (long long)(nv0.bar_bytes / 16);
      d.z_alive16 = reinterpret_cast<uint4*>(nv0.alive); d.n_alive16 = alive16;
    } else {
      k_reset_state<<<256, 256, 0, st>>>(cv.cnt, (int)(bs * kCntPad), cv.tiny, (int)bs, status, cv.ticket, (int)(16 + bs + 1),
                                         reinterpret_cast<uint4*>(nv0.bar), (long long)(nv0.bar_bytes / 16),
                                         reinterpret_cast<uint4*>(nv0.alive), alive16);
    }
  }
  // rows per workgroup: 2048 when that still gives two workgroups per CU, else 1024 / 512 (small batches, the TTA tensor)
  d.rows_per_thread = (bs * A >= 4LL * kDecThreads * 480) ? 4 : (bs * A >= 2LL * kDecThreads * 480) ? 2 : 1;
  dim3 gd((unsigned)((A + kDecThreads * d.rows_per_thread - 1) / (kDecThreads * d.rows_per_thread)), (unsigned)bs);
  {
    ProfScope ps(PROF_DECODE, st);
    if (dtype == 0) k_decode<float><<<gd, kDecThreads, 0, st>>>(d);
    else k_decode<__half><<<gd, kDecThreads, 0, st>>>(d);
  }
  if (n_extra > 0 && extra8) k_append_extra<<<(unsigned)((n_extra + 255) / 256), 256, 0, st>>>(extra8, (int)n_extra, A, nc, d);
  if (small_hint && class_ok)
    k_tiny_cross<<<dim3(kTinyParts, (unsigned)bs), 256, 0, st>>>(cv.cand, cv.keys_a, cv.cnt, cap_img, max_wh, iou_thres, cv.tiny);
  const unsigned gs = (unsigned)((bs + 255) / 256);
  Carve& nv = cv.nms;
  const int64_t max_seg = (max_nms > 0 && max_nms < cap_img) ? max_nms : cap_img;
  // grid of the NMS launch (needed by the planner inside the fused kernel)
  const int nms_capmax = cap_max(bs * ncs);
  const int plan_chunk = cap_first() < nms_capmax ? cap_first() : nms_capmax;
  // Which NMS kernel: one workgroup per segment, everything in LDS (nms_small.h), when the previous call's largest segment fits it
  // (class segments, the in-LDS sort, thr >= 0: its first decision stage uses the conservative bounds); else the persistent kernel.
  const bool small_nms = lds_sort && class_ok && seg_hint > 0 && seg_hint <= kSmallMax && iou_thres >= 0.f && bs * ncs <= 65535;
  const int plan_nb = (bs * ncs > 1 && !small_nms) ? nms_grid(bs * ncs, bs * max_seg, cap_first()) : 0;
  // k_nms_small's helper workgroups (nms_small.h: a large segment is shared by several workgroups).  A workgroup takes a whole CU
  // (160 KB of LDS), so helpers only run at once with the segments' own workgroups on the CUs the segments leave free: as many as
  // that, none for the bs16 x 16-class step whose 256 segments fill the device (measured there with 256 helpers: 0.121 -> 0.132 ms,
  // the helpers start when the first small segments are through).  Their lists and the parts' bit matrices live in the persistent
  // kernel's edge lists, which this path does not use.  OBB_NMS_SMALL_HELPERS = n pins the number (0: every segment stays whole).
  const int helpers_env = [] { const char* e = getenv("OBB_NMS_SMALL_HELPERS"); const int v = (e && *e) ? atoi(e) : -1; return v > kSmallHelpMax ? kSmallHelpMax : v; }();   // (read per call: tests switch in one process)
  static const int no_fused_out = obb_dev_switch("OBB_NO_FUSED_OUT", 0) != 0;    // A/B switch (development builds)
  // OBB_NMS_SELF_SORT (read per call): 0 = always the sort kernel, 1 = self-sorting segments where no helpers would run, 2 = default:
  // self-sorting segments wherever they are possible (no helpers then: the sort kernel is what hands them out)
  const int self_env = [] { const char* e = getenv("OBB_NMS_SELF_SORT"); return (e && *e >= '0' && *e <= '2') ? *e - '0' : 2; }();
  const bool self_possible = small_nms && !out_packed && !no_fused_out && (int64_t)ncs * kSmallMax <= cap_img && self_env != 0;
  int helpers = 0;
  if (small_nms && !(self_possible && self_env == 2 && helpers_env < 0)) {
    const int64_t free_cus = (int64_t)hw_cu_count() - bs * ncs;
    helpers = helpers_env >= 0 ? helpers_env : (int)(free_cus < 0 ? 0 : (free_cus > kSmallHelpMax ? kSmallHelpMax : free_cus));
  }
  int *help_work = nullptr, *seg_np = nullptr, *seg_ticket = nullptr;
  u64 *part_main = nullptr, *part_help = nullptr;
  if (helpers > 0) {
    const size_t nseg_b = align_up((size_t)(bs * ncs) * 4), work_b = align_up((size_t)helpers * 4);
    const size_t main_b = align_up((size_t)bs * cap_img * kSmallWords * 8), help_b = align_up((size_t)helpers * kSmallMax * kSmallWords * 8);
    const size_t have = (size_t)((bs * ncs < (int64_t)cu_count()) ? bs * ncs : (int64_t)cu_count()) * (size_t)nv.ecap * 4;   // (carve: nteams x ecap)
    if (work_b + 2 * nseg_b + main_b + help_b > have) helpers = 0;
    else {
      char* p = reinterpret_cast<char*>(nv.edges);
      help_work = (int*)p; p += work_b;
      seg_np = (int*)p; p += nseg_b;
      seg_ticket = (int*)p; p += nseg_b;
      part_main = (u64*)p; p += main_b;
      part_help = (u64*)p;
    }
  }
  // Self-sorting segments (SmallSelfSort): no sort launch when the small-segment kernel writes the rows itself and every segment is
  // one workgroup's.  OBB_NMS_SELF_SORT=0 keeps the sort kernel (read per call: tests compare the two in one process).
  const bool self_sort = self_possible && helpers == 0;
  if (self_sort) {
  } else if (lds_sort) {
    ProfScope ps(PROF_SEGSORT, st);
    static OncePerDevice attr;
    const size_t lds = (size_t)kSortLdsMax * 12;
    if (const int attr_dev = attr.need(); attr_dev != OncePerDevice::kDone) {
      if (hipFuncSetAttribute((const void*)k_sort_prep_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return OBB_ERR_LAUNCH;
      attr.mark(attr_dev);
    }
    const int parts = (class_ok && ncs >= 8) ? 4 : 1;           // class-segment images: four workgroups each take every fourth class
    k_sort_prep_lds<<<(unsigned)(bs * parts) + (plan_nb > 0 ? 1u : 0u), 1024, lds, st>>>(cv.cand, cv.keys_a, cv.vals_a, cv.keys_b, cv.vals_b, cv.cnt, cv.tiny, (int)bs, cap_img,
                                                   max_nms, class_ok, A, nc, ncs, agnostic ? 0.f : max_wh, cv.sort_begin, cv.sort_end,
                                                   cv.img_end, cv.mode, cv.grp_begin, cv.grp_end, nv.seg_begin, nv.seg_end, nv.keep_cnt,
                                                   nv.rec, nv.alive, cv.ticket, plan_nb, plan_chunk, plan_nb > 0 ? nv.plan : nullptr,
                                                   reinterpret_cast<int*>(cv.digit_base), parts,   // (digit_base: the class-bounds table of the other sort paths, free here)
                                                   helpers, help_work, seg_np, seg_ticket);
  } else {
   {
    ProfScope ps(PROF_SEGSORT, st);
    k_cand_segments<<<gs, 256, 0, st>>>(cv.cnt, cv.tiny, (int)bs, cap_img, max_nms, class_ok, group_ok, cv.sort_begin, cv.sort_end,
                                        cv.img_end, cv.mode, cv.grp_begin, cv.grp_end);
    if (class_ok) {
      dim3 gr((unsigned)((cap_img + 255) / 256), (unsigned)bs);
      k_rekey<<<gr, 256, 0, st>>>(cv.cand, cv.keys_a, cv.sort_begin, cv.sort_end, cv.mode, A, nc);
    }
    {
      // Several images, or one: every image spread over many workgroups (segsort.h's LSD radix sort, two launches per key byte) --
      // or, ONE large image (the TTA tensor) of up to kPsMaxN slots: the three-launch sort of psrs_sort.h over the image's
      // candidates (keys are unique: the tie word; the count lives on the device).  This branch is what an un-hinted first call of a
      // shape takes as well (the fused in-LDS path needs the previous call's largest candidate count).
      int tb = 0;
      while ((1ll << tb) < A * nc + n_extra + 1) tb++;                 // significant bits of the tie word
      unsigned mask = 0xF0u;                                            // single-list key: score in bytes 4..7
      for (int d = 0; d < 4; d++) if (tb > d * 8) mask |= 1u << d;
      if (class_ok) mask = 0xFFu;                                       // + class-mode key: anchor 0..2, score 3..6, class 7
      if (bs == 1 && !group_ok && cap_img <= kPsMaxN && nv.sort_tmp_bytes >= ps_scratch_bytes()) {
        PsBuf b{};
        b.run_k = cv.keys_a; b.run_v = cv.vals_a; b.out_k = cv.keys_b; b.out_v = cv.vals_b;      // (the runs are sorted in place)
        b.n = (int)cap_img; b.n_dev = cv.sort_end; b.err = nullptr;
        ps_carve_scratch(nv.sort_tmp, &b);
        const int runs = (int)((cap_img + kPsRun - 1) / kPsRun);
        k_ps_local_pairs<<<(unsigned)runs, kPsRun, 0, st>>>(b, cv.keys_a, cv.vals_a);
        rc = ps_finish(b, runs, st);
        if (rc) return rc;
      } else {
        rc = seg_radix_sort_large(cv.keys_a, cv.keys_b, cv.vals_a, cv.vals_b, cv.sort_begin, cv.sort_end, (int)bs, cap_img, bs * cap_img,
                                  mask, cv.srs_hist, st);
        if (rc) return rc;
      }
    }
    if (group_ok) {
      // images with more than max_nms candidates: group the top max_nms (now in score order) by class, one stable pass
      rc = seg_group_by_class(cv.keys_b, cv.keys_a, cv.vals_b, cv.vals_a, cv.grp_begin, cv.grp_end, (int)bs, cap_img, cv.cand, cv.srs_hist,
                              cv.digit_base, st);
      if (rc) return rc;
      dim3 gc((unsigned)((max_nms + 255) / 256), (unsigned)bs);
      k_copy_grouped<<<gc, 256, 0, st>>>(cv.keys_a, cv.keys_b, cv.vals_a, cv.vals_b, cv.grp_begin, cv.grp_end);
    }
    const int64_t nseg = bs * ncs;
    k_class_bounds<<<(unsigned)((nseg + 255) / 256), 256, 0, st>>>(cv.keys_b, cv.sort_begin, cv.img_end, cv.mode, cv.digit_base, (int)bs,
                                                                   ncs, nv.seg_begin, nv.seg_end, nv.keep_cnt);
   }
   dim3 gp((unsigned)((max_seg + 255) / 256), (unsigned)bs);
   {
    ProfScope ps(PROF_PREP, st);
    if (hipMemsetAsync(nv.alive, 0, nv.alive_bytes, st) != hipSuccess) return OBB_ERR_LAUNCH;
    k_prep_cand<<<gp, 256, 0, st>>>(cv.cand, cv.vals_b, cv.sort_begin, cv.img_end, cap_img, agnostic ? 0.f : max_wh, nv.rec, nv.alive);
   }
  }

  NmsArgs a{};
  a.rec = nv.rec; a.order = nullptr; a.alive = nv.alive; a.seg_begin = nv.seg_begin; a.seg_end = nv.seg_end;   // keep_out: sorted positions
  a.keep_cnt = nv.keep_cnt; a.keep_out = cv.keep;
  a.rows = nv.rows; a.nrows = nv.nrows; a.edges = nv.edges; a.nedges = nv.nedges;
  a.ecap = nv.ecap; a.n = (int)(bs * cap_img); a.capmax = cap_max(bs * ncs);
  a.max_keep = (int)max_det; a.window = nms_window(max_det); a.thr = iou_thres; a.cull = (iou_thres >= 0.f) ? 1 : 0;
  // the output stage runs inside k_nms_small unless the caller wants packed rows (SmallGather)
  const bool fused_out = small_nms && !out_packed && !no_fused_out;
  if (small_nms) {
    ProfScope ps(PROF_STEPS, st);
    static OncePerDevice attr;
    const size_t lds = small_lds_bytes<RotGeom>();
    if (const int attr_dev = attr.need(); attr_dev != OncePerDevice::kDone) {
      if (hipFuncSetAttribute((const void*)k_nms_small<RotGeom, SmallNoTail>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_nms_small<RotGeom, SmallGather>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_nms_small<RotGeom, SmallGather, SmallSelfSort>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return OBB_ERR_LAUNCH;
      attr.mark(attr_dev);
    }
    SmallArgs sa{};
    sa.rec = nv.rec; sa.alive = nv.alive; sa.seg_begin = nv.seg_begin; sa.seg_end = nv.seg_end; sa.keep_cnt = nv.keep_cnt;
    sa.keep_out = cv.keep; sa.too_big = cv.ticket + 8; sa.thr = iou_thres; sa.max_keep = (int)max_det;
    sa.ncs = ncs; sa.mode = cv.mode;
    sa.helpers = helpers; sa.help_cnt = cv.ticket + 12; sa.work = help_work; sa.seg_np = seg_np; sa.seg_ticket = seg_ticket;
    sa.part_main = part_main; sa.part_help = part_help;
    const unsigned gsmall = (unsigned)(bs * ncs + helpers);      // (the helpers are the FIRST blocks: dispatched before the segments' own workgroups)
    if (fused_out) {
      // (keys_a / vals_a, the sort's input, are free: they take what the segments publish)
      sa.keys_sorted = cv.keys_b; sa.vals_sorted = cv.vals_b; sa.pub_key = cv.keys_a; sa.pub_val = cv.vals_a; sa.n_pos = bs * cap_img;
      SmallGather::Args ga{};
      ga.cand = cv.cand; ga.cnt = cv.cnt; ga.tiny = cv.tiny; ga.info = cv.ticket; ga.ticket = cv.ticket + 16;
      ga.out = out; ga.out_count = out_count; ga.status = status; ga.cap_img = cap_img; ga.max_det = max_det; ga.bs = (int)bs;
      ga.clean_cnt = kept ? cv.cnt : nullptr; ga.clean_tiny = kept ? cv.tiny : nullptr;
      if (self_sort) {
        // (the segments read keys_a, the filter kernel's output, while others publish: keys_b / vals_b, the sort's output places, are free)
        sa.pub_key = cv.keys_b; sa.pub_val = cv.vals_b; sa.keys_sorted = nullptr; sa.vals_sorted = nullptr;
        sa.keys_in = cv.keys_a; sa.cand = cv.cand; sa.cnt = cv.cnt; sa.tiny = cv.tiny; sa.cap_img = cap_img; sa.max_nms = max_nms; sa.A = A; sa.nc = nc;
        sa.class_offset = max_wh; sa.seg_max = cv.ticket + 4;
        k_nms_small<RotGeom, SmallGather, SmallSelfSort><<<gsmall, kSmallThreads, lds, st>>>(sa, ga);
      } else
      k_nms_small<RotGeom, SmallGather><<<gsmall, kSmallThreads, lds, st>>>(sa, ga);
    } else {
      k_nms_small<RotGeom, SmallNoTail><<<gsmall, kSmallThreads, lds, st>>>(sa, SmallNoTail::Args{});
    }
  } else {
    ProfScope ps(PROF_STEPS, st);
    rc = nms_steps(0, a, nv, bs * ncs, bs * max_seg, st, kNmsBarZeroed | ((lds_sort && plan_nb > 0) ? kNmsPlanned : 0));
    if (rc) return rc;
  }
  if (!fused_out) {
    ProfScope ps(PROF_GATHER, st);
    // parts per image: one up to ~2k expected candidates (the kept rows fit the kernel's LDS and 256 threads), then one per 1024
    const unsigned gparts = expected_cand <= 2048 ? 1u : (unsigned)((expected_cand + 1023) / 1024 > 16 ? 16 : (expected_cand + 1023) / 1024);
    k_gather_out<<<dim3((unsigned)bs, gparts), 256, 0, st>>>(cv.cand, cv.vals_b, cv.keys_b, cv.keep, nv.seg_begin, nv.keep_cnt, cv.mode, ncs, max_det,
                                              out, out_count, cv.cnt, cap_img, status, nv.abort_flag, out_packed, cv.ticket, cv.tiny,
                                              kept ? cv.cnt : nullptr, kept ? cv.tiny : nullptr);
  }
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // namespace obb
