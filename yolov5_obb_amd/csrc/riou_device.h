// Rotated-rectangle IoU for gfx950 -- written from scratch for this project.
//
// Behavioural contract (what the result bits must equal):
//   reference  utils/nms_rotated/src/box_iou_rotated_utils.h:333-360
//   (single_box_iou_rotated<float>, *device* hull branch :195-218), evaluated
//   in IEEE fp32 without FMA contraction -- the same contract the CPU oracle
//   (oracle/riou_impl.inc) is pinned to.  Compile with -ffp-contract=off.
//
// What is different from the reference's formulation (and why it is legal):
//   * the double-precision cos/sin depend on one box only, so they are hoisted
//     into a per-box feature record computed once per NMS call (N evaluations
//     instead of N^2) -- rbox_make_feat(); the four products s2*h, c2*w, c2*h,
//     s2*w are hoisted with them (each is a single rounded fp32 product in the
//     reference, so the hoisted value is the same bits);
//   * a conservative reject (rbox_certainly_disjoint) answers "IoU == 0"
//     without running the clip when the rectangles are separated by a clear
//     margin: in that case the reference finds no edge crossing and no contained
//     corner (num <= 2, :322-324) and returns exactly 0;
//   * the <=24 candidate points live in a caller-provided scratch with a
//     compile-time lane stride (LDS column per lane on the GPU: bank = lane, no
//     conflicts; stride 1 on the host check) instead of per-thread local arrays
//     that would spill to scratch memory;
//   * the hull is built in place (p and q of the reference alias) and the
//     squared distances are recomputed instead of being swapped along.
#pragma once
#include "obb_device.h"

namespace obb {

struct RBoxFeat {
  float x, y, w, h;      // raw box (centre, size)
  float sh, cw, ch, sw;  // (float)sin(a)*0.5f*h, (float)cos(a)*0.5f*w, ...*h, ...*w
  float r;               // inflated circumradius (cull)
  float c, s;            // cos(a), sin(a) in fp32 (cull only)
  float area;            // w*h
};

OBB_HD RBoxFeat rbox_make_feat(float x, float y, float w, float h, float a) {
  RBoxFeat f;
  double th = (double)a;
  float c2 = (float)cos(th) * 0.5f;  // box_iou_rotated_utils.h:63-65
  float s2 = (float)sin(th) * 0.5f;
  f.x = x; f.y = y; f.w = w; f.h = h;
  f.sh = s2 * h; f.cw = c2 * w; f.ch = c2 * h; f.sw = s2 * w;
  f.c = c2 * 2.0f; f.s = s2 * 2.0f;
  f.r = sqrtf(w * w + h * h) * 0.5005f;  // half diagonal, +0.1 %
  f.area = w * h;                         // :351-352
  return f;
}

// Locality guard for the two shortcuts below.  The reference rounds the four
// corners of each box AFTER shifting both centres to the pair's midpoint, so a
// corner carries an absolute error of about ulp(M), M = |centre distance|/2 +
// size.  A box whose short side is not far above that error degenerates: an edge
// vector rounds to (nearly) zero, the "corner inside rectangle" test (:115-154)
// turns the box into an unbounded strip, and the reference reports IoU ~ 1 for
// boxes that are hundreds of pixels apart (reproduced in tests/).  The shortcuts
// are therefore only taken when both short sides are >= 2e-5 * M (relative edge
// error <= ~1 %), where the usual geometric reasoning holds; everything else
// goes through the full clip and inherits the reference's behaviour bit for bit.
OBB_HD bool rbox_pair_well_conditioned(const RBoxFeat& A, const RBoxFeat& B) {
  float dx = B.x - A.x, dy = B.y - A.y;
  float M = fabsf(dx) + fabsf(dy) + A.r + B.r;
  float ms = fminf(fminf(fabsf(A.w), fabsf(A.h)), fminf(fabsf(B.w), fabsf(B.h)));
  return ms >= 2e-5f * M;   // false on NaN
}

// True only when the reference returns IoU == 0 for this (well-conditioned) pair:
// separation by >= 0.1 % of the involved extents on the centre line or on one
// of the four edge normals -> no edge crossing, no contained corner (num <= 2,
// :322-324).  NaN anywhere makes every test false (not culled).
OBB_HD bool rbox_certainly_disjoint(const RBoxFeat& A, const RBoxFeat& B) {
  if (!rbox_pair_well_conditioned(A, B)) return false;
  float dx = B.x - A.x, dy = B.y - A.y;
  float rs = A.r + B.r;
  if (dx * dx + dy * dy > rs * rs) return true;
  const float m = 1.001f;
  float hwA = fabsf(A.w) * 0.5f, hhA = fabsf(A.h) * 0.5f;
  float hwB = fabsf(B.w) * 0.5f, hhB = fabsf(B.h) * 0.5f;
  // corner = centre +- w/2 * (c, -s) +- h/2 * (s, c)
  float cc = fabsf(A.c * B.c + A.s * B.s);  // |cos(dtheta)|
  float ss = fabsf(B.s * A.c - B.c * A.s);  // |sin(dtheta)|
  if (fabsf(dx * A.c - dy * A.s) > (hwA + hwB * cc + hhB * ss) * m) return true;
  if (fabsf(dx * A.s + dy * A.c) > (hhA + hwB * ss + hhB * cc) * m) return true;
  if (fabsf(dx * B.c - dy * B.s) > (hwB + hwA * cc + hhA * ss) * m) return true;
  if (fabsf(dx * B.s + dy * B.c) > (hhB + hwA * ss + hhA * cc) * m) return true;
  return false;
}

// Upper bound on the IoU the reference returns for a well-conditioned pair (area
// ratio, +0.1 %): IoU = I/(a1+a2-I) with I <= min(a1,a2).  Used to skip the clip
// when the bound is already <= the NMS threshold.  +inf when not applicable.
OBB_HD float rbox_iou_upper_bound(const RBoxFeat& A, const RBoxFeat& B) {
  if (!rbox_pair_well_conditioned(A, B)) return __builtin_inff();
  float lo = fminf(A.area, B.area), hi = fmaxf(A.area, B.area);
  if (!(lo > 0.f) || !(hi < 1e30f)) return __builtin_inff();
  return lo / hi * 1.001f;
}

// ---------------------------------------------------------------------------------------------------------------
// Register-only IoU BOUNDS (no scratch, no sort, no data-dependent indexing): a filter in front of the exact clip.
//
// 2*Area(A n B) = sum over the edges of A of (t1-t0)+ * cross(a_k, e_k)  +  the same over the edges of B, where
// [t0,t1] is the part of the edge inside the other rectangle (Liang-Barsky against an axis-aligned box in the other
// rectangle's own frame) -- Green's theorem around the intersection polygon, origin = midpoint of the two centres.
// Unlike the reference's point set + Graham scan this formulation has a FIRST-order error at shallow crossings (the
// two rectangles locate the same crossing independently), so the function returns an interval [lo, hi] that contains
// the IoU the reference computes, with the width derived from the crossing angles, and reports `false` whenever it
// cannot vouch for the interval: nearly coincident edges, tiny or badly conditioned boxes, non-finite input.  The NMS
// only acts on it when the whole interval lies on one side of the threshold; everything else runs the exact clip.
// tests/native/host_check_riou.cpp checks the containment against the oracle on tens of millions of seeded pairs.
struct IouBounds { float lo, hi; };

// 1-ulp reciprocal: the filter only needs t = q/p to a relative 1e-6 (its error terms are orders of magnitude larger);
// a correctly rounded division costs ~15 instructions on gfx950, v_rcp_f32 one.
OBB_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

OBB_HD bool rbox_fast_iou_bounds(const RBoxFeat& A, const RBoxFeat& B, IouBounds* out) {
  const float wA = A.w, hA = A.h, wB = B.w, hB = B.h;
  const float mn = fminf(fminf(wA, hA), fminf(wB, hB));
  if (!(mn >= 1.0f)) return false;                       // small / negative / NaN sizes: the exact path decides
  const float dx = B.x - A.x, dy = B.y - A.y;
  const float R = 0.5f * (fabsf(dx) + fabsf(dy)) + A.r + B.r;
  if (!(R < 1e6f) || !(mn >= 1e-3f * R)) return false;   // extreme aspect / spread: not worth a bound
  const float hx = 0.5f * dx, hy = 0.5f * dy;            // centres: A = (-hx,-hy), B = (+hx,+hy)
  const float eps = 4e-6f * R;                           // error of a signed distance computed below
  const float tol = 1e-4f * R;
  const float tolc = 5e-5f * R;                          // ~100x the reference's rounding of a corner-to-edge distance
  float sum2 = 0.f, err = 0.f;
  bool safe = true;

  // one direction: edges of P (centre (px,py), axes from (c,s), half sizes hw,hh) clipped by Q
  auto clip_edges = [&](float pcx, float pcy, float pc, float ps, float phw, float phh, float qcx, float qcy, float qc, float qs,
                        float qhw, float qhh) __attribute__((always_inline)) {
    // P's corners relative to the midpoint, counter-clockwise: +w+h, -w+h, -w-h, +w-h with e_w = (c,-s), e_h = (s,c)
    const float wx = phw * pc, wy = -phw * ps, hxv = phh * ps, hyv = phh * pc;
    float cx[4], cy[4];
    cx[0] = pcx + wx + hxv; cy[0] = pcy + wy + hyv;
    cx[1] = pcx - wx + hxv; cy[1] = pcy - wy + hyv;
    cx[2] = pcx - wx - hxv; cy[2] = pcy - wy - hyv;
    cx[3] = pcx + wx - hxv; cy[3] = pcy + wy - hyv;
    // the same corners in Q's frame
    float ux[4], uy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float rx = cx[k] - qcx, ry = cy[k] - qcy;
      ux[k] = rx * qc - ry * qs;      // . e_w(Q) = (c, -s)
      uy[k] = rx * qs + ry * qc;      // . e_h(Q) = (s, c)
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int k1 = (k + 1) & 3;
      const float len = (k & 1) ? 2.f * phh : 2.f * phw;
      const float x0 = ux[k], y0 = uy[k], ddx = ux[k1] - ux[k], ddy = uy[k1] - uy[k];
      float t0 = 0.f, t1 = 1.f;
      // boundary: q(t) = q - p*t >= 0 is inside
      auto side = [&](float p, float q) __attribute__((always_inline)) {
        const float q1 = q - p;
        if (fabsf(q) < tol && fabsf(q1) < tol) safe = false;               // the edge lies on Q's boundary line
        // a corner within rounding distance of the other rectangle's boundary is where the REFERENCE is fragile: its
        // corner-inside test and the two adjacent edge crossings (t ~ 0 or 1) can all miss by one ulp and drop a
        // vertex of the intersection polygon (seen: IoU 0.29 instead of 0.88 for thin boxes).  Never vouch there.
        if (fabsf(q) < tolc) safe = false;
        if ((q < 0.f) != (q1 < 0.f)) err += eps * len * fast_rcp(fmaxf(fabsf(p), 1e-30f));   // a crossing: located to +-eps/sin(phi)
        const float r = q * fast_rcp(p);                                    // p == 0: +-inf or NaN, handled below
        if (p < 0.f) t0 = fmaxf(t0, r);
        else if (p > 0.f) t1 = fminf(t1, r);
        else if (q < 0.f) t1 = -1.f;                                        // parallel and outside
      };
      side(-ddx, x0 + qhw);
      side(ddx, qhw - x0);
      side(-ddy, y0 + qhh);
      side(ddy, qhh - y0);
      const float ex = cx[k1] - cx[k], ey = cy[k1] - cy[k];
      const float span = fmaxf(t1 - t0, 0.f);
      sum2 += span * (cx[k] * ey - cy[k] * ex);
    }
  };
  clip_edges(-hx, -hy, A.c, A.s, 0.5f * wA, 0.5f * hA, hx, hy, B.c, B.s, 0.5f * wB, 0.5f * hB);
  clip_edges(hx, hy, B.c, B.s, 0.5f * wB, 0.5f * hB, -hx, -hy, A.c, A.s, 0.5f * wA, 0.5f * hA);
  if (!safe) return false;
  const float aA = A.area, aB = B.area;
  // a mislocated crossing opens / overlaps the boundary by its location error: area error <= 0.5 * R * that length;
  // plus the rounding of eight products of magnitude R * len
  const float e_area = 0.5f * R * err + 2e-5f * R * R;
  float inter = 0.5f * sum2;
  if (!(inter == inter) || !(e_area == e_area)) return false;
  float ilo = fmaxf(inter - e_area, 0.f), ihi = fminf(fmaxf(inter + e_area, 0.f), fminf(aA, aB));
  if (ilo > ihi) ilo = ihi;
  const float s = aA + aB;
  out->lo = ilo / (s - ilo) - 3e-5f;
  out->hi = ihi / (s - ihi) + 3e-5f;
  return out->lo == out->lo && out->hi == out->hi;
}

// Cheap BOUNDS on the IoU for the pairs a detector produces most: near-duplicates (lower bound) and loosely overlapping
// neighbours (upper bound).  With P's corners in Q's frame (u, v):
//   the part of P outside Q is covered by four slabs, one per side of Q: thickness = how far P's farthest corner
//   sticks out beyond that side, length = P's extent along the side      ->  |P n Q| >= |P| - sum(excess * extent);
//   P lies inside its own bounding box in Q's frame                      ->  |P n Q| <= overlap_u * overlap_v.
// Both directions are evaluated and the better one of each kind is kept.  ~150 flops, no division chain, no branches
// on data -- against the ~1500 instructions of rbox_fast_iou_bounds, which it spares for every pair it decides.
// Returns false when it does not vouch: same conditioning rules as rbox_fast_iou_bounds, in particular never when a
// corner lies within `tol` of the other rectangle's boundary lines (where the reference itself is fragile).  Otherwise
// lo <= (the IoU the reference computes) <= hi; tests/native/host_check_fastiou.cpp checks it against the oracle.
OBB_HD bool rbox_quick_bounds(const RBoxFeat& A, const RBoxFeat& B, IouBounds* out) {
  const float wA = fabsf(A.w), hA = fabsf(A.h), wB = fabsf(B.w), hB = fabsf(B.h);
  const float mn = fminf(fminf(wA, hA), fminf(wB, hB));
  if (!(mn >= 1.0f)) return false;
  const float dx = B.x - A.x, dy = B.y - A.y;
  const float R = 0.5f * (fabsf(dx) + fabsf(dy)) + A.r + B.r;
  if (!(R < 1e6f) || !(mn >= 1e-3f * R)) return false;
  const float tol = 1e-4f * R;
  bool safe = true;
  float in_hi = 3.4e38f;
  // area of P outside Q (upper estimate) from P's corners in Q's frame (centre of P relative to Q: (rx, ry))
  auto outside = [&](float rx, float ry, float pc, float ps, float phw, float phh, float qc, float qs, float qhw, float qhh)
      __attribute__((always_inline)) -> float {
    const float cxq = rx * qc - ry * qs, cyq = rx * qs + ry * qc;            // P's centre in Q's frame
    // P's half axes in Q's frame: e_w(P) = (pc, -ps), e_h(P) = (ps, pc)
    const float wux = phw * (pc * qc + ps * qs), wuy = phw * (pc * qs - ps * qc);
    const float hux = phh * (ps * qc - pc * qs), huy = phh * (ps * qs + pc * qc);
    float umax = -3.4e38f, umin = 3.4e38f, vmax = -3.4e38f, vmin = 3.4e38f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float sw = (k == 0 || k == 3) ? 1.f : -1.f, sh = (k < 2) ? 1.f : -1.f;
      const float u = cxq + sw * wux + sh * hux, v = cyq + sw * wuy + sh * huy;
      if (fabsf(u - qhw) < tol || fabsf(u + qhw) < tol || fabsf(v - qhh) < tol || fabsf(v + qhh) < tol) safe = false;
      umax = fmaxf(umax, u); umin = fminf(umin, u); vmax = fmaxf(vmax, v); vmin = fminf(vmin, v);
    }
    const float eu = umax - umin, ev = vmax - vmin;
    const float ou = fmaxf(fminf(umax, qhw) - fmaxf(umin, -qhw), 0.f), ov = fmaxf(fminf(vmax, qhh) - fmaxf(vmin, -qhh), 0.f);
    in_hi = fminf(in_hi, ou * ov);
    return (fmaxf(umax - qhw, 0.f) + fmaxf(-qhw - umin, 0.f)) * ev + (fmaxf(vmax - qhh, 0.f) + fmaxf(-qhh - vmin, 0.f)) * eu;
  };
  const float outA = outside(-dx, -dy, A.c, A.s, 0.5f * wA, 0.5f * hA, B.c, B.s, 0.5f * wB, 0.5f * hB);
  const float outB = outside(dx, dy, B.c, B.s, 0.5f * wB, 0.5f * hB, A.c, A.s, 0.5f * wA, 0.5f * hA);
  if (!safe) return false;
  const float aA = A.area, aB = B.area, e_area = 2e-5f * R * R;               // rounding of products of magnitude R * len
  float ilo = fmaxf(aA - outA, aB - outB) - e_area;
  float ihi = fminf(in_hi + e_area, fminf(aA, aB));
  if (!(ilo == ilo) || !(ihi == ihi)) return false;
  ilo = fminf(fmaxf(ilo, 0.f), ihi);
  const float sum = aA + aB;
  out->lo = ilo / (sum - ilo) - 3e-5f;
  out->hi = ihi / (sum - ihi) + 3e-5f;
  return out->lo == out->lo && out->hi == out->hi;
}

// The first half of the clip -- the candidate points of the intersection polygon: the 16 edge / edge crossings and the corners of
// either box inside the other (box_iou_rotated_utils.h:93-154) -- as a function of its own: STORE = true writes them to the
// scratch columns (rbox_iou below), STORE = false only COUNTS them, in the same arithmetic.  A count of <= 2 means the reference
// returns IoU = 0 (:322-324: `if (num <= 2) return 0.0`): the cross-class check of the fused NMS driver (nmsobb_impl.h:
// k_tiny_cross) screens its far-apart pairs with the count alone, ~400 flops and no scratch instead of the whole clip.
template <int STRIDE, bool STORE>
OBB_HD int rbox_points(const RBoxFeat& A, const RBoxFeat& B, float* px, float* py) {
  constexpr float kDetLe = f32_floor(1e-14);   // fabs(det) <= 1e-14

  // centre shift, evaluated in double like the reference (:338-346)
  double mx = (double)(A.x + B.x) * 0.5, my = (double)(A.y + B.y) * 0.5;
  float ax = (float)((double)A.x - mx), ay = (float)((double)A.y - my);
  float bx = (float)((double)B.x - mx), by = (float)((double)B.y - my);

  float v1x[4], v1y[4], v2x[4], v2y[4];
  v1x[0] = ax + A.sh + A.cw; v1y[0] = ay + A.ch - A.sw;
  v1x[1] = ax - A.sh + A.cw; v1y[1] = ay - A.ch - A.sw;
  v1x[2] = 2 * ax - v1x[0];  v1y[2] = 2 * ay - v1y[0];
  v1x[3] = 2 * ax - v1x[1];  v1y[3] = 2 * ay - v1y[1];
  v2x[0] = bx + B.sh + B.cw; v2y[0] = by + B.ch - B.sw;
  v2x[1] = bx - B.sh + B.cw; v2y[1] = by - B.ch - B.sw;
  v2x[2] = 2 * bx - v2x[0];  v2y[2] = 2 * by - v2y[0];
  v2x[3] = 2 * bx - v2x[1];  v2y[3] = 2 * by - v2y[1];

  float e1x[4], e1y[4], e2x[4], e2y[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    e1x[i] = v1x[(i + 1) & 3] - v1x[i]; e1y[i] = v1y[(i + 1) & 3] - v1y[i];
    e2x[i] = v2x[(i + 1) & 3] - v2x[i]; e2y[i] = v2y[(i + 1) & 3] - v2y[i];
  }

  int n = 0;
  // 16 edge/edge crossings (:93-112)
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float det = e2x[j] * e1y[i] - e1x[i] * e2y[j];
      if (fabsf(det) <= kDetLe) continue;
      float dx = v2x[j] - v1x[i], dy = v2y[j] - v1y[i];
      float t1 = (e2x[j] * dy - dx * e2y[j]) / det;
      float t2 = (e1x[i] * dy - dx * e1y[i]) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        if constexpr (STORE) {
          px[n * STRIDE] = v1x[i] + e1x[i] * t1;
          py[n * STRIDE] = v1y[i] + e1y[i] * t1;
        }
        n++;
      }
    }
  }
  // corners of A inside B (:115-135)
  {
    float abab = e2x[0] * e2x[0] + e2y[0] * e2y[0];
    float adad = e2x[3] * e2x[3] + e2y[3] * e2y[3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float apx = v1x[i] - v2x[0], apy = v1y[i] - v2y[0];
      float apab = apx * e2x[0] + apy * e2y[0];
      float apad = -(apx * e2x[3] + apy * e2y[3]);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        if constexpr (STORE) { px[n * STRIDE] = v1x[i]; py[n * STRIDE] = v1y[i]; }
        n++;
      }
    }
  }
  // corners of B inside A (:138-154)
  {
    float abab = e1x[0] * e1x[0] + e1y[0] * e1y[0];
    float adad = e1x[3] * e1x[3] + e1y[3] * e1y[3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float apx = v2x[i] - v1x[0], apy = v2y[i] - v1y[0];
      float apab = apx * e1x[0] + apy * e1y[0];
      float apad = -(apx * e1x[3] + apy * e1y[3]);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        if constexpr (STORE) { px[n * STRIDE] = v2x[i]; py[n * STRIDE] = v2y[i]; }
        n++;
      }
    }
  }
  return n;
}

// the count alone; -1: a box of (numerically) no area -- the reference returns 0 before it looks at any point (:353)
OBB_HD int rbox_npoints(const RBoxFeat& A, const RBoxFeat& B) {
  constexpr float kAreaLt = f32_ceil(1e-14);   // area < 1e-14
  if (A.area < kAreaLt || B.area < kAreaLt) return -1;
  return rbox_points<1, false>(A, B, nullptr, nullptr);
}

// Full clip.  A = higher-scored ("row") box, B = lower-scored ("column") box:
// the argument order is part of the contract (nms_rotated_cuda.cu:60).
// px/py: scratch for 24 points, element i at [i * STRIDE].
template <int STRIDE>
OBB_HD float rbox_iou(const RBoxFeat& A, const RBoxFeat& B, float* px, float* py) {
  constexpr float kAreaLt = f32_ceil(1e-14);   // area < 1e-14
  constexpr float kCpLtNeg = f32_ceil(-1e-6);  // cp < -1e-6
  constexpr float kCpLt = f32_ceil(1e-6);      // fabs(cp) < 1e-6
  constexpr float kD2Gt = f32_floor(1e-8);     // dist > 1e-8

  if (A.area < kAreaLt || B.area < kAreaLt) return 0.f;  // :353
  const int n = rbox_points<STRIDE, true>(A, B, px, py);

  float inter = 0.f;
  if (n > 2) {
    // ---- Graham scan in place (:159-291) ----
    int t = 0;
    float tx = px[0], ty = py[0];
    for (int i = 1; i < n; i++) {
      float x = px[i * STRIDE], y = py[i * STRIDE];
      if (y < ty || (y == ty && x < tx)) { t = i; tx = x; ty = y; }
    }
    for (int i = 0; i < n; i++) { px[i * STRIDE] -= tx; py[i * STRIDE] -= ty; }
    {
      float x0 = px[0], y0 = py[0];
      px[0] = px[t * STRIDE]; py[0] = py[t * STRIDE];
      px[t * STRIDE] = x0; py[t * STRIDE] = y0;
    }
    // exchange sort by polar angle around the pivot, ties by distance (:205-218)
    for (int i = 1; i < n - 1; i++) {
      float qix = px[i * STRIDE], qiy = py[i * STRIDE];
      for (int j = i + 1; j < n; j++) {
        float qjx = px[j * STRIDE], qjy = py[j * STRIDE];
        float cp = qix * qjy - qjx * qiy;
        bool sw = cp < kCpLtNeg;
        if (!sw && fabsf(cp) < kCpLt) sw = (qix * qix + qiy * qiy) > (qjx * qjx + qjy * qjy);
        if (sw) {
          px[j * STRIDE] = qix; py[j * STRIDE] = qiy;
          qix = qjx; qiy = qjy;
        }
      }
      px[i * STRIDE] = qix; py[i * STRIDE] = qiy;
    }
    // first point that is not a duplicate of the pivot (:239-249)
    int k = 1;
    for (; k < n; k++) {
      float x = px[k * STRIDE], y = py[k * STRIDE];
      if (x * x + y * y > kD2Gt) break;
    }
    if (k < n) {
      px[1 * STRIDE] = px[k * STRIDE]; py[1 * STRIDE] = py[k * STRIDE];
      int m = 2;
      for (int i = k + 1; i < n; i++) {
        float qx = px[i * STRIDE], qy = py[i * STRIDE];
        while (m > 1) {
          float bx2 = px[(m - 2) * STRIDE], by2 = py[(m - 2) * STRIDE];
          float q1x = qx - bx2, q1y = qy - by2;
          float q2x = px[(m - 1) * STRIDE] - bx2, q2y = py[(m - 1) * STRIDE] - by2;
          if (q1x * q2y >= q2x * q1y) m--; else break;  // two rounded products (:266)
        }
        px[m * STRIDE] = qx; py[m * STRIDE] = qy; m++;
      }
      // fan area (:293-305)
      if (m > 2) {
        float q0x = px[0], q0y = py[0];
        float acc = 0.f;
        float pxx = px[1 * STRIDE] - q0x, pyy = py[1 * STRIDE] - q0y;
        for (int i = 1; i < m - 1; i++) {
          float nx = px[(i + 1) * STRIDE] - q0x, ny = py[(i + 1) * STRIDE] - q0y;
          acc += fabsf(pxx * ny - nx * pyy);
          pxx = nx; pyy = ny;
        }
        inter = acc * 0.5f;  // area / 2.0 (exact scaling either way)
      }
    }
  }
  return inter / (A.area + B.area - inter);  // :358
}

}  // namespace obb
