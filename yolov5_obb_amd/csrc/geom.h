// Geometry policies shared by the NMS and pairwise kernels.
//
// A box travels through the NMS as an AoS record of RECQ float4 (16-byte) quads,
// stored in score order.  Quad 0 is everything the hot pair loop needs
// (x, y, inflated circumradius, short side): one 16-byte load per box, read as a
// wave-uniform LDS broadcast on the row side.  The remaining quads are only
// gathered for the few pairs that survive the cheap test.
#pragma once
#include <hip/hip_runtime.h>
#include "obb_device.h"
#include "riou_device.h"
#include "riou64_device.h"
#include "piou_device.h"

namespace obb {

// Two rows of the hot pair loops per instruction: gfx950 issues v_pk_{add,mul}_f32 on a pair of fp32 values at the rate of
// the scalar forms (the loops are VALU-issue bound: ~16 instructions per row test, 64 columns wide).  ColPk = the column
// box of a lane with every component doubled, built once per tile; the row values arrive as pairs of v_readlane results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct ColPk { f32x2 x, y, z, w; };
__device__ __forceinline__ ColPk col_splat(const float4& c) {
  ColPk p;
  p.x = f32x2{c.x, c.x}; p.y = f32x2{c.y, c.y}; p.z = f32x2{c.z, c.z}; p.w = f32x2{c.w, c.w};
  return p;
}

struct RotGeom {
  static constexpr int RECQ = 4;
  static constexpr int SCR = 48;  // 24 clip points x (x,y) per lane
  // q0 = {x, y, r, ms^2}  q1 = {w, h, c, s}  q2 = {sh, cw, ch, sw}  q3 = {area, 0, 0, 0}
  static OBB_HD void pack(const RBoxFeat& f, float4* q) {
    float ms = fminf(fabsf(f.w), fabsf(f.h));
    q[0] = make_float4(f.x, f.y, f.r, ms * ms);
    q[1] = make_float4(f.w, f.h, f.c, f.s);
    q[2] = make_float4(f.sh, f.cw, f.ch, f.sw);
    q[3] = make_float4(f.area, 0.f, 0.f, 0.f);
  }
  static OBB_HD RBoxFeat unpack(const float4& q0, const float4& q1, const float4& q2, const float4& q3) {
    RBoxFeat f;
    f.x = q0.x; f.y = q0.y; f.r = q0.z;
    f.w = q1.x; f.h = q1.y; f.c = q1.z; f.s = q1.w;
    f.sh = q2.x; f.cw = q2.y; f.ch = q2.z; f.sw = q2.w;
    f.area = q3.x;
    return f;
  }
  // Hot-loop test on quad 0 only: circumscribed circles clearly apart AND the pair is well
  // conditioned (riou_device.h) -> the reference returns exactly 0.  ~11 VALU ops.
  // Conditioning: riou_device.h asks for min side >= 2e-5 * M with M = |dx| + |dy| + rs.  When the circles are apart
  // (d > rs) M <= (sqrt(2) + 1) * d, so  ms^2 >= 2.34e-9 * d^2  (= (2e-5 * 2.4143)^2 rounded up) is sufficient --
  // a slightly stricter guard than the original, never a weaker one.
  static __device__ __forceinline__ bool cheap_reject(const float4& a, const float4& b) {
    const float dx = b.x - a.x, dy = b.y - a.y;
    const float rs = a.z + b.z;
    const float d2 = dx * dx + dy * dy;
    return (d2 > rs * rs) && (fminf(a.w, b.w) >= 2.34e-9f * d2);
  }
  // the same test for two row boxes (components interleaved: ax = {row0.x, row1.x}, ...) against the column box c: the
  // same operations in the same order on each half, no contraction -- bit-identical decisions
  static constexpr bool PACKED = true;
  static __device__ __forceinline__ void cheap_reject2(f32x2 ax, f32x2 ay, f32x2 az, f32x2 aw, const ColPk& c, bool& r0, bool& r1) {
    const f32x2 dx = c.x - ax, dy = c.y - ay;
    const f32x2 rs = az + c.z;
    const f32x2 d2 = dx * dx + dy * dy;
    const f32x2 r2 = rs * rs;
    const f32x2 lim = d2 * 2.34e-9f;
    r0 = (d2.x > r2.x) && (fminf(aw.x, c.w.x) >= lim.x);
    r1 = (d2.y > r2.y) && (fminf(aw.y, c.w.y) >= lim.y);
  }
  // Two-stage decision of "IoU > thr" for the pairs that survive the hot loop.
  //   classify_quick  registers only: separating-axis reject, area-ratio bound, the slab / bounding-box bounds
  //             (rbox_quick_bounds) -- decides most pairs of a detector's output;
  //   classify_full   the IoU interval of rbox_fast_iou_bounds for the rest -- 0 (no) / 1 (yes) whenever the whole interval
  //             lies on one side of the threshold, 2 = undecided (a few % of the pairs).  The two run as separate queue
  //             stages so that a wave never executes the expensive one for a handful of lanes;
  //   hit_exact the reference's clip + Graham scan, bit for bit, on the LDS scratch column of the lane.
  static constexpr bool HAS_FAST = true;
  static constexpr bool HAS_GRID = true;       // the spatial index of grid.h applies (circle test on quad 0; quad 3.y = brute flag)
  template <class A> static __device__ __forceinline__ float thr_of(const A& a) { return a.thr; }
  // stage 1a (cheap, ~150 flops): 0 / 1 = decided, 2 = straight to the exact clip (no shortcuts allowed), 3 = needs the interval
  static __device__ __forceinline__ int classify_quick(const float4* ra, const float4* rb, float thr, bool cull) {
    if (!cull) return 2;
    RBoxFeat A = unpack(ra[0], ra[1], ra[2], ra[3]);
    RBoxFeat B = unpack(rb[0], rb[1], rb[2], rb[3]);
    if (rbox_certainly_disjoint(A, B)) return 0;
    if (rbox_iou_upper_bound(A, B) <= thr) return 0;
    IouBounds qb;
    if (rbox_quick_bounds(A, B, &qb)) {
      if (qb.lo > thr) return 1;                           // near-duplicates, the bulk of a detector's pairs
      if (qb.hi <= thr) return 0;                          // neighbours that barely touch
    }
    return 3;
  }
  // stage 1b (~1500 instructions): the IoU interval; 2 = undecided
  static __device__ __forceinline__ int classify_full(const float4* ra, const float4* rb, float thr) {
    RBoxFeat A = unpack(ra[0], ra[1], ra[2], ra[3]);
    RBoxFeat B = unpack(rb[0], rb[1], rb[2], rb[3]);
    IouBounds bd;
    if (rbox_fast_iou_bounds(A, B, &bd)) {
      if (bd.lo > thr) return 1;
      if (bd.hi <= thr) return 0;
    }
    return 2;
  }
  static __device__ __forceinline__ bool hit_exact(const float4* ra, const float4* rb, float thr, float* scr) {
    RBoxFeat A = unpack(ra[0], ra[1], ra[2], ra[3]);
    RBoxFeat B = unpack(rb[0], rb[1], rb[2], rb[3]);
    return rbox_iou<64>(A, B, scr, scr + 24 * 64) > thr;
  }
};

struct QuadGeom {
  static constexpr int RECQ = 4;               // q3 = {extended cone, (r, M), -, -}: the second cone rule (classify_quick)
  static constexpr int SCR = 40;  // 2 x 10 points x (x,y) per lane
  // q0 = {fp16 minx | miny, fp16 maxx | maxy (rounded outward), budget f, cone (piou_device.h: quad_cone_bits)}   q1 = {x0, y0, x1, y1}   q2 = {x2, y2, x3, y3}
  //      (f = -inf: the first two words are the extended cone and (r, M) of piou_device.h's second cone rule instead)
  // The reference's quad IoU sums signed triangle areas taken from the coordinate origin; for disjoint quads the terms
  // cancel only up to rounding noise that grows with the square of the coordinates, so "IoU == 0" cannot be predicted
  // from a bounding-box test alone.  Two rules (piou_device.h): a pair is skipped when the column box's cone lies
  // counter-clockwise of the row box's (all 16 terms are then EXACTLY zero: proved), or when the bounding boxes are disjoint AND the two boxes' areas are
  // large enough against their coordinate magnitudes that the noise cannot reach the threshold (piou_device.h:
  // quad_skip_record / quad_skip_pair; k_prep_quad computes the per-box budget, the threshold is known there); every
  // other pair is clipped, as the reference does.
  // (a = the row box = the first argument of the IoU, b = the column box: the cone rule is not symmetric)
  static __device__ __forceinline__ bool cheap_reject(const float4& a, const float4& b) {
    QuadSkip A, B;
    A.lo = __builtin_bit_cast(uint32_t, a.x); A.hi = __builtin_bit_cast(uint32_t, a.y); A.f = a.z;
    B.lo = __builtin_bit_cast(uint32_t, b.x); B.hi = __builtin_bit_cast(uint32_t, b.y); B.f = b.z;
    if (quad_cone_skip(__builtin_bit_cast(uint32_t, a.w), __builtin_bit_cast(uint32_t, b.w)) || quad_skip_pair(A, B)) return true;
    // two quads without a budget (f = -inf: OBB_NMS_POLY_STRICT=1, thr <= 0, outside the searched envelope) carry the second
    // proved rule's words in place of their boxes (k_prep_quad): the row quad's extended cone counter-clockwise of the column quad's cone
    const float ninf = -__builtin_huge_valf();
    return a.z == ninf && b.z == ninf && quad_cone2_skip(A.lo, A.hi, __builtin_bit_cast(uint32_t, b.w), B.hi);
  }
  static constexpr bool PACKED = false;
  static OBB_HD QuadFeat unpack(const float4& q0, const float4& q1) {
    QuadFeat f;
    f.x[0] = q0.x; f.y[0] = q0.y; f.x[1] = q0.z; f.y[1] = q0.w;
    f.x[2] = q1.x; f.y[2] = q1.y; f.x[3] = q1.z; f.y[3] = q1.w;
    f.minx = f.maxx = f.miny = f.maxy = 0.f;
    return f;
  }
  static __device__ __forceinline__ float iou(const QuadFeat& A, const QuadFeat& B, float* scr) {
    return quad_iou<64>(A, B, scr, scr + 10 * 64, scr + 20 * 64, scr + 30 * 64);
  }
  static constexpr bool HAS_FAST = false;      // every pair that is not skipped is clipped: no value bounds for this formulation
  static constexpr bool HAS_GRID = false;
  template <class A> static __device__ __forceinline__ float thr_of(const A& a) { return a.thr; }
  // The second PROVED cone rule (piou_device.h: the row quad counter-clockwise of the column quad as seen from the origin): tier 1 on
  // the row's extended cone, tier 2 on its plain cone + the pair check on the coordinates.  IoU = +0 exactly: not a hit for thr >= 0.
  static __device__ __forceinline__ int classify_quick(const float4* ra, const float4* rb, float thr, bool cull) {
    if (!cull || !(thr >= 0.f)) return 2;
    const float4 a3 = ra[3], b3 = rb[3];
    const uint32_t a_rm = __builtin_bit_cast(uint32_t, a3.y), b_rm = __builtin_bit_cast(uint32_t, b3.y);
    const uint32_t cb = __builtin_bit_cast(uint32_t, rb[0].w);
    if (quad_cone2_skip(__builtin_bit_cast(uint32_t, a3.x), a_rm, cb, b_rm)) return 0;
    if (quad_cone2_skip(__builtin_bit_cast(uint32_t, ra[0].w), a_rm, cb, b_rm) &&
        quad_cone2_nofuzzy(unpack(ra[1], ra[2]), unpack(rb[1], rb[2]))) return 0;
    return 2;
  }
  static __device__ __forceinline__ int classify_full(const float4*, const float4*, float) { return 2; }
  static __device__ __forceinline__ bool hit_exact(const float4* ra, const float4* rb, float thr, float* scr) {
    return iou(unpack(ra[1], ra[2]), unpack(rb[1], rb[2]), scr) > thr;
  }
  // The same decision for FEW pairs per wave, sixteen lanes per pair: the 16 terms of the reference's sum (a triangle of the
  // first ring against a triangle of the second, poly_nms_cuda.cu:88-105) are independent, a lane computes one of them in its
  // own scratch column, the group's first lane adds them in the reference's order (t = 4 i + j, from +0) and decides.  Four
  // pairs per pass: a wave holding a dozen undecided pairs is through in three term times instead of sixteen.  Every lane of
  // the wave calls; `want` lanes hold a pair and get its decision.  scr_wave: the wave's scratch (lane columns).
  static constexpr bool HAS_COOP = true;
#ifndef OBB_QUAD_COOP_MAX
#define OBB_QUAD_COOP_MAX 40
#endif
  static constexpr int kCoopMax = OBB_QUAD_COOP_MAX;            // above this many pairs the lane-per-pair form is as fast
  static __device__ __forceinline__ bool hit_exact_coop(bool want, const float4* ra, const float4* rb, float thr, float* scr_wave) {
    float v[16];
    if (want) {
      const float4 a1 = ra[1], a2 = ra[2], b1 = rb[1], b2 = rb[2];
      v[0] = a1.x; v[1] = a1.y; v[2] = a1.z; v[3] = a1.w; v[4] = a2.x; v[5] = a2.y; v[6] = a2.z; v[7] = a2.w;
      v[8] = b1.x; v[9] = b1.y; v[10] = b1.z; v[11] = b1.w; v[12] = b2.x; v[13] = b2.y; v[14] = b2.z; v[15] = b2.w;
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = 0.f;
    }
    return iou_coop(want, v, scr_wave) > thr;       // (lanes without a pair: NaN > thr is false)
  }
  // v: the lane's pair as x0 y0 .. x3 y3 of the first quad, then of the second; returns the pair's IoU to its lane (NaN to the others)
  static __device__ __forceinline__ float iou_coop(bool want, const float (&v)[16], float* scr_wave) {
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(want);
    const int np = __popcll(m);
    const unsigned long long below = (1ull << lane) - 1ull;
    // lane k learns which lane holds the pair of rank k (the others fill the permutation behind the np pairs)
    const int rank = want ? __popcll(m & below) : np + __popcll(~m & below);
    const int holder = __builtin_amdgcn_ds_permute(rank << 2, lane);
    const int sub = lane >> 4, t = lane & 15, ti = t >> 2, tj = t & 3;
    float mine = __builtin_nanf("");
    for (int p = 0; p * 4 < np; p++) {
      const int kk = p * 4 + sub;
      const int src = __shfl(holder, kk < np ? kk : 0);
      float w[16];
#pragma unroll
      for (int k = 0; k < 16; k++) w[k] = __shfl(v[k], src);
      // the rings as the reference orients them (reversed when the signed area is negative, poly_nms_cuda.cu:108-109)
      float px_[4] = {w[0], w[2], w[4], w[6]}, py_[4] = {w[1], w[3], w[5], w[7]};
      float qx_[4] = {w[8], w[10], w[12], w[14]}, qy_[4] = {w[9], w[11], w[13], w[15]};
      const float s1 = quad_signed_area(px_, py_), s2 = quad_signed_area(qx_, qy_);
      float ax[4], ay[4], bx[4], by[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ax[i] = (s1 < 0) ? px_[3 - i] : px_[i]; ay[i] = (s1 < 0) ? py_[3 - i] : py_[i];
        bx[i] = (s2 < 0) ? qx_[3 - i] : qx_[i]; by[i] = (s2 < 0) ? qy_[3 - i] : qy_[i];
      }
      const int i1 = (ti + 1) & 3, j1 = (tj + 1) & 3;
      const float pax = ti == 0 ? ax[0] : ti == 1 ? ax[1] : ti == 2 ? ax[2] : ax[3];
      const float pay = ti == 0 ? ay[0] : ti == 1 ? ay[1] : ti == 2 ? ay[2] : ay[3];
      const float pbx = i1 == 0 ? ax[0] : i1 == 1 ? ax[1] : i1 == 2 ? ax[2] : ax[3];
      const float pby = i1 == 0 ? ay[0] : i1 == 1 ? ay[1] : i1 == 2 ? ay[2] : ay[3];
      const float qcx = tj == 0 ? bx[0] : tj == 1 ? bx[1] : tj == 2 ? bx[2] : bx[3];
      const float qcy = tj == 0 ? by[0] : tj == 1 ? by[1] : tj == 2 ? by[2] : by[3];
      const float qdx = j1 == 0 ? bx[0] : j1 == 1 ? bx[1] : j1 == 2 ? bx[2] : bx[3];
      const float qdy = j1 == 0 ? by[0] : j1 == 1 ? by[1] : j1 == 2 ? by[2] : by[3];
      float term = 0.f;
      float* c0 = scr_wave + lane;
      if (kk < np) term = ptri_tri<64>(pax, pay, pbx, pby, qcx, qcy, qdx, qdy, c0, c0 + 10 * 64, c0 + 20 * 64, c0 + 30 * 64);
      float inter = 0.f;
#pragma unroll
      for (int k = 0; k < 16; k++) inter += __shfl(term, (lane & 48) + k);
      const float ua = pabs(quad_signed_area(ax, ay)) + pabs(quad_signed_area(bx, by)) - inter;
      const float iouv = (ua == 0.f) ? (inter + 1.f) / (ua + 1.f) : inter / ua;     // :136-137
      const float got = __shfl(iouv, (rank & 3) << 4);
      if (want && (rank >> 2) == p) mine = got;
    }
    return mine;
  }
};

// Double-precision quads for the tile -> full-image merge (DOTA_devkit/ResultMerge_multi_process.py:62-123,
// py_cpu_nms_poly_fast): a pair is only looked at when the horizontal bounding boxes overlap strictly
// (hbb_ovr > 0, :82-98), then DOTA_devkit/polyiou.cpp's iou_poly decides; the box survives iff iou <= thresh
// (:115), so a NaN IoU (two empty rings) suppresses.
//   q0 = {minx, miny, maxx, maxy} rounded OUTWARD to fp32 (hot-loop reject; never rejects an overlapping pair)
//   q1..q4 = the 8 double coordinates
struct QuadGeom64 {
  static constexpr int RECQ = 5;
  static constexpr int SCR = 40;   // 32 lanes x 40 doubles: the exact stage runs in two half-wave passes
  static constexpr bool HAS_FAST = false;
  static constexpr bool HAS_GRID = false;
  template <class A> static __device__ __forceinline__ double thr_of(const A& a) { return a.thr64; }
  static __device__ __forceinline__ bool cheap_reject(const float4& a, const float4& b) {
    // (a box of {-inf, -inf, +inf, +inf} meets every partner: non-finite coordinates, negative thresholds -- k_prep_quad64)
    const bool always = (a.x == -__builtin_huge_valf()) || (b.x == -__builtin_huge_valf());
    return !always && !(fminf(a.z, b.z) > fmaxf(a.x, b.x) && fminf(a.w, b.w) > fmaxf(a.y, b.y));
  }
  static constexpr bool PACKED = false;
  static __device__ __forceinline__ int classify_quick(const float4*, const float4*, double, bool) { return 2; }
  static __device__ __forceinline__ int classify_full(const float4*, const float4*, double) { return 2; }
  // np.maximum / np.minimum hand a NaN operand through (fmax / fmin would drop it)
  static __device__ __forceinline__ double npmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
  static __device__ __forceinline__ double npmin(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
  static __device__ __forceinline__ void hbb(const QuadFeatT<double>& f, double* x1, double* y1, double* x2, double* y2) {
    *x1 = npmin(npmin(f.x[0], f.x[1]), npmin(f.x[2], f.x[3])); *x2 = npmax(npmax(f.x[0], f.x[1]), npmax(f.x[2], f.x[3]));   // np.min / np.max: NaN wins
    *y1 = npmin(npmin(f.y[0], f.y[1]), npmin(f.y[2], f.y[3])); *y2 = npmax(npmax(f.y[0], f.y[1]), npmax(f.y[2], f.y[3]));
  }
  static __device__ __forceinline__ QuadFeatT<double> unpack(const float4* r) {
    const double2* d = reinterpret_cast<const double2*>(r + 1);
    QuadFeatT<double> f;
#pragma unroll
    for (int k = 0; k < 4; k++) { double2 v = d[k]; f.x[k] = v.x; f.y[k] = v.y; }
    f.minx = f.maxx = f.miny = f.maxy = 0.0;
    return f;
  }
  // scr = this lane's column of the wave's scratch (column index = lane); the double flavour shares the 10 KB block
  // between the two half-waves, one after the other
  static __device__ __forceinline__ bool hit_exact(const float4* ra, const float4* rb, double thr, float* scr) {
    return hit_exact_t<true>(ra, rb, thr, scr);
  }
  // GATE: the horizontal-box gate of py_cpu_nms_poly_fast; without it every pair goes through iou_poly (py_cpu_nms_poly)
  template <bool GATE>
  static __device__ __forceinline__ bool hit_exact_t(const float4* ra, const float4* rb, double thr, float* scr) {
    const int lane = threadIdx.x & 63;
    double* base = reinterpret_cast<double*>(scr - lane) + (lane & 31);
    const QuadFeatT<double> A = unpack(ra), B = unpack(rb);
    bool look = true, hit = false;
    if constexpr (GATE) {
      double ax1, ay1, ax2, ay2, bx1, by1, bx2, by2;
      hbb(A, &ax1, &ay1, &ax2, &ay2); hbb(B, &bx1, &by1, &bx2, &by2);
      // :70,87-96  areas with the +1 convention, intersection without it
      const double area_a = (ax2 - ax1 + 1) * (ay2 - ay1 + 1), area_b = (bx2 - bx1 + 1) * (by2 - by1 + 1);
      const double w = npmax(0.0, npmin(ax2, bx2) - npmax(ax1, bx1)), h = npmax(0.0, npmin(ay2, by2) - npmax(ay1, by1));
      const double hi = w * h;
      const double hv = hi / (area_a + area_b - hi);
      look = hv > 0;
      // a pair the gate keeps out of iou_poly is judged on the horizontal ratio itself (:115: np.where(hbb_ovr <= thresh)):
      // 0 <= thresh keeps it for thresh >= 0; a NaN ratio (non-finite coordinates) or a negative threshold removes it
      hit = !look && !(hv <= thr);
    }
    int nhalf = 2;
    asm volatile("" : "+s"(nhalf));   // opaque trip count: the two passes must stay two passes (lanes l and l + 32 share a column)
    for (int half = 0; half < nhalf; half++) {
      if (look && (lane >> 5) == half) {
        const double iou = quad_iou_t<32, false, double>(A, B, base, base + 10 * 32, base + 20 * 32, base + 30 * 32);
        hit = !(iou <= thr);
      }
    }
    return hit;
  }
};

// py_cpu_nms_poly (DOTA_devkit/ResultMerge_multi_process.py:24-60): the merge NMS WITHOUT the horizontal-box gate -- every
// pair of a kept row and a remaining candidate goes through polyiou.cpp's iou_poly.  Nothing may be skipped: iou_poly of two
// quads whose horizontal boxes are apart is not 0 by construction (rounding noise, NaN for degenerate quads).
struct QuadGeom64All : QuadGeom64 {
  static __device__ __forceinline__ bool cheap_reject(const float4&, const float4&) { return false; }
  static __device__ __forceinline__ bool hit_exact(const float4* ra, const float4* rb, double thr, float* scr) {
    return QuadGeom64::hit_exact_t<false>(ra, rb, thr, scr);
  }
};

// py_cpu_nms (DOTA_devkit/ResultMerge_multi_process.py:125-157; what mergebyrec hands to mergebase): horizontal boxes
// [x1 y1 x2 y2] with the "+ 1" pixel convention in areas and intersections, numpy double arithmetic.
//   q0 = {x1 rounded down, y1 rounded down, x2 + 1 rounded up, y2 + 1 rounded up} in fp32 for the hot loop -- or
//        {-inf, -inf, +inf, +inf} for a box that must meet every partner: a coordinate that is not finite, an area <= 0
//        (0 / 0 and negative unions give NaN or a negative ratio: np.where(ovr <= thresh) decides those), or a threshold < 0 /
//        NaN (then even an overlap of 0 removes a box; the prep kernel knows the threshold);
//   q1 = {x1, y1}, q2 = {x2, y2} as doubles.
// A pair the hot loop rejects has x2a + 1 < x1b (or the like) in exact arithmetic, so numpy's w = max(0, xx2 - xx1 + 1) is
// exactly 0, inter = 0 and, both areas being positive, ovr = 0 <= thresh: never removed.
struct HbbGeom64 {
  static constexpr int RECQ = 3;
  static constexpr int SCR = 40;   // (unused by the clip-free test; the wave blocks double as the resolve phase's LDS: same footprint as QuadGeom64)
  static constexpr bool HAS_FAST = false;
  static constexpr bool HAS_GRID = false;
  static constexpr bool PACKED = false;
  template <class A> static __device__ __forceinline__ double thr_of(const A& a) { return a.thr64; }
  static __device__ __forceinline__ bool cheap_reject(const float4& a, const float4& b) {
    return (a.z < b.x) || (b.z < a.x) || (a.w < b.y) || (b.w < a.y);
  }
  static __device__ __forceinline__ int classify_quick(const float4*, const float4*, double, bool) { return 2; }
  static __device__ __forceinline__ int classify_full(const float4*, const float4*, double) { return 2; }
  // np.maximum / np.minimum hand a NaN operand through (fmax / fmin would drop it)
  static __device__ __forceinline__ double npmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
  static __device__ __forceinline__ double npmin(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
  static __device__ __forceinline__ bool hit_exact(const float4* ra, const float4* rb, double thr, float*) {
    const double2 a1 = *reinterpret_cast<const double2*>(ra + 1), a2 = *reinterpret_cast<const double2*>(ra + 2);
    const double2 b1 = *reinterpret_cast<const double2*>(rb + 1), b2 = *reinterpret_cast<const double2*>(rb + 2);
    const double area_a = (a2.x - a1.x + 1) * (a2.y - a1.y + 1), area_b = (b2.x - b1.x + 1) * (b2.y - b1.y + 1);   // :133
    const double w = npmax(0.0, npmin(a2.x, b2.x) - npmax(a1.x, b1.x) + 1);                                      // :143-149
    const double h = npmax(0.0, npmin(a2.y, b2.y) - npmax(a1.y, b1.y) + 1);
    const double inter = w * h;
    const double ovr = inter / (area_a + area_b - inter);                                                        // :151
    return !(ovr <= thr);                                                                                        // :153
  }
};

// Double-precision rotated boxes: float64 tensors (nms_rotated_cuda.cu:96 dispatches double; the kernel compares the double
// IoU with the FLOAT threshold of its signature, nms_rotated_cuda.cu:14,60).
//   q0 = {x, y, r, ms^2} in fp32 for the hot loop -- the centre rounded to fp32, r = the inflated circumradius rounded UP
//        plus the rounding of the centre, ms^2 = short side squared rounded DOWN: the circle test never rejects a pair the
//        double-precision test of the same kind would keep, and its conditioning guard is the float flavour's (stricter
//        than double arithmetic needs: it only ever rejects less);
//   q1..q4 = the doubles {x, y, sh, cw, ch, sw, area, 0} (riou64_device.h).
// No cheap decision stages: every pair that passes the circle test runs the exact double clip (two half-wave passes over
// the wave's 12 KB scratch block, like QuadGeom64).
struct RotGeom64 {
  static constexpr int RECQ = 5;
  static constexpr int SCR = 48;   // 32 lanes x 48 doubles
  static constexpr bool HAS_FAST = false;
  static constexpr bool HAS_GRID = false;
  template <class A> static __device__ __forceinline__ double thr_of(const A& a) { return a.thr64; }
  static __device__ __forceinline__ bool cheap_reject(const float4& a, const float4& b) { return RotGeom::cheap_reject(a, b); }
  static constexpr bool PACKED = true;
  static __device__ __forceinline__ void cheap_reject2(f32x2 ax, f32x2 ay, f32x2 az, f32x2 aw, const ColPk& c, bool& r0, bool& r1) {
    RotGeom::cheap_reject2(ax, ay, az, aw, c, r0, r1);
  }
  static __device__ __forceinline__ int classify_quick(const float4*, const float4*, double, bool) { return 2; }
  static __device__ __forceinline__ int classify_full(const float4*, const float4*, double) { return 2; }
  static __device__ __forceinline__ void pack(double x, double y, double w, double h, double a, float4* q) {
    const RBoxFeat64 f = rbox_make_feat64(x, y, w, h, a);
    const double r = sqrt(w * w + h * h) * 0.5005 + (fabs(x) + fabs(y)) * 1.2e-7;
    float rf = (float)r;
    if ((double)rf < r) rf = f32_next_up(rf);
    const double ms = fmin(fabs(w), fabs(h)), ms2 = ms * ms;
    float mf = (float)ms2;
    if ((double)mf > ms2) mf = f32_next_down(mf);
    q[0] = make_float4((float)x, (float)y, rf, mf);
    double2* d = reinterpret_cast<double2*>(q + 1);
    d[0] = make_double2(f.x, f.y); d[1] = make_double2(f.sh, f.cw); d[2] = make_double2(f.ch, f.sw); d[3] = make_double2(f.area, 0.0);
  }
  static __device__ __forceinline__ RBoxFeat64 unpack(const float4* r) {
    const double2* d = reinterpret_cast<const double2*>(r + 1);
    RBoxFeat64 f;
    const double2 a = d[0], b = d[1], c = d[2], e = d[3];
    f.x = a.x; f.y = a.y; f.sh = b.x; f.cw = b.y; f.ch = c.x; f.sw = c.y; f.area = e.x;
    return f;
  }
  static __device__ __forceinline__ bool hit_exact(const float4* ra, const float4* rb, double thr, float* scr) {
    const int lane = threadIdx.x & 63;
    double* base = reinterpret_cast<double*>(scr - lane) + (lane & 31);
    const RBoxFeat64 A = unpack(ra), B = unpack(rb);
    bool hit = false;
    int nhalf = 2;
    asm volatile("" : "+s"(nhalf));   // opaque trip count: the two passes must stay two passes (lanes l and l + 32 share a column)
    for (int half = 0; half < nhalf; half++) {
      if ((lane >> 5) == half) hit = rbox_iou_f64<32>(A, B, base, base + 24 * 32) > thr;   // strict, nms_rotated_cuda.cu:60
    }
    return hit;
  }
};

// exact rotated IoU value with the per-lane scratch column (pairwise kernels)
__device__ __forceinline__ float rot_iou_value(const RBoxFeat& A, const RBoxFeat& B, float* scr) {
  return rbox_iou<64>(A, B, scr, scr + 24 * 64);
}

}  // namespace obb
