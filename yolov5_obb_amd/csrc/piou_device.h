// Quadrilateral IoU (polygon clipping) for gfx950 -- written for this project.
//
// Behavioural contract: the float device function devPolyIoU of the reference
//   utils/nms_rotated/src/poly_nms_cuda.cu:26-142
//   (same text in DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu:36-212 and
//    poly_overlaps_kernel.cu:36-328),
// evaluated in IEEE fp32 without FMA contraction -- the contract the CPU oracle
// (oracle/piou_impl.inc) restates.  The reference's per-thread arrays sized by
// `maxn` (10 / 51 / 510 float2 depending on the copy) are replaced by a
// 2 x 10-point scratch with a compile-time lane stride (an LDS column per lane
// on the GPU); the clip ping-pongs between the two halves.  One pinning of undefined behaviour: the one
// indeterminate read of the reference (lineCross leaving pp[m] unwritten): that
// vertex is defined as (0,0) here and in the oracle.
#pragma once
#include "obb_device.h"

namespace obb {

template <typename T>
struct QuadFeatT {
  T x[4], y[4];
  T minx, maxx, miny, maxy;  // AABB (cull)
};
typedef QuadFeatT<float> QuadFeat;

OBB_HD QuadFeat quad_make_feat(const float* p8) {
  QuadFeat q;
#pragma unroll
  for (int i = 0; i < 4; i++) { q.x[i] = p8[2 * i]; q.y[i] = p8[2 * i + 1]; }
  q.minx = fminf(fminf(q.x[0], q.x[1]), fminf(q.x[2], q.x[3]));
  q.maxx = fmaxf(fmaxf(q.x[0], q.x[1]), fmaxf(q.x[2], q.x[3]));
  q.miny = fminf(fminf(q.y[0], q.y[1]), fminf(q.y[2], q.y[3]));
  q.maxy = fmaxf(fmaxf(q.y[0], q.y[1]), fmaxf(q.y[2], q.y[3]));
  return q;
}

// sig(): the reference compares against the double constant 1e-8 (float operands are widened)
OBB_HD int psgn(float d) {
  constexpr float kHi = f32_floor(1e-8);   // d >  1e-8
  constexpr float kLo = f32_ceil(-1e-8);   // d < -1e-8
  return (int)(d > kHi) - (int)(d < kLo);
}
OBB_HD int psgn(double d) { return (int)(d > 1e-8) - (int)(d < -1e-8); }
OBB_HD float pabs(float v) { return fabsf(v); }
OBB_HD double pabs(double v) { return fabs(v); }

template <typename T>
OBB_HD T ptri(T ox, T oy, T ax, T ay, T bx, T by) {
  return (ax - ox) * (by - oy) - (bx - ox) * (ay - oy);
}

// keep the part of polygon p (n vertices, scratch stride STRIDE) left of a->b; the result goes to q (the caller swaps the two
// scratch halves between clips).  The reference (poly_nms_cuda.cu:46-72, cut) writes the raw crossings / kept vertices to a second
// array, copies them back dropping every point within 1e-8 of its RAW predecessor, then drops trailing points within 1e-8 of the
// first.  Same points, same arithmetic, one pass: a point is compared with its raw predecessor (held in registers) as it is
// produced and stored only when it stays; the closing edge reads vertex 0 again instead of a sentinel copy behind the last one.
template <int STRIDE, typename T>
OBB_HD int pclip(const T* px, const T* py, int n, T ax, T ay, T bx, T by, T* qx, T* qy) {
  T cx = px[0], cy = py[0];
  T s1 = ptri(ax, ay, bx, by, cx, cy);
  int g1 = psgn(s1);
  int k = 0;
  bool first = true;
  T lx = 0, ly = 0, fx = 0, fy = 0;          // the raw predecessor; the first stored point
  for (int i = 0; i < n; i++) {
    const int nx = (i + 1 == n) ? 0 : i + 1;
    T dx = px[nx * STRIDE], dy = py[nx * STRIDE];
    T s2 = ptri(ax, ay, bx, by, dx, dy);
    int g2 = psgn(s2);
    if (g1 > 0) {
      if (first || !(psgn(cx - lx) == 0 && psgn(cy - ly) == 0)) {
        qx[k * STRIDE] = cx; qy[k * STRIDE] = cy;
        if (k == 0) { fx = cx; fy = cy; }
        k++;
      }
      lx = cx; ly = cy; first = false;
    }
    if (g1 != g2) {
      T den = s2 - s1;
      bool ok = psgn(den) != 0;   // (both-zero cannot occur here: the signs differ)
      // unwritten in the reference when !ok (indeterminate); pinned to (0,0) here and in the oracle
      const T ix = ok ? (cx * s2 - dx * s1) / den : T(0);
      const T iy = ok ? (cy * s2 - dy * s1) / den : T(0);
      if (first || !(psgn(ix - lx) == 0 && psgn(iy - ly) == 0)) {
        qx[k * STRIDE] = ix; qy[k * STRIDE] = iy;
        if (k == 0) { fx = ix; fy = iy; }
        k++;
      }
      lx = ix; ly = iy; first = false;
    }
    cx = dx; cy = dy; s1 = s2; g1 = g2;
  }
  while (k > 1 && psgn(qx[(k - 1) * STRIDE] - fx) == 0 && psgn(qy[(k - 1) * STRIDE] - fy) == 0) k--;
  return k;
}

// signed area of triangle(o,a,b) ∩ triangle(o,c,d), o = origin
template <int STRIDE, typename T>
OBB_HD T ptri_tri(T ax, T ay, T bx, T by, T cx, T cy, T dx, T dy, T* px, T* py, T* qx, T* qy) {
  const T z = 0;
  int s1 = psgn(ptri(z, z, ax, ay, bx, by));
  int s2 = psgn(ptri(z, z, cx, cy, dx, dy));
  if (s1 == 0 || s2 == 0) return z;
  if (s1 == -1) { T t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; }
  if (s2 == -1) { T t = cx; cx = dx; dx = t; t = cy; cy = dy; dy = t; }
  px[0] = z; py[0] = z;
  px[1 * STRIDE] = ax; py[1 * STRIDE] = ay;
  px[2 * STRIDE] = bx; py[2 * STRIDE] = by;
  int n = 3;
  n = pclip<STRIDE>(px, py, n, z, z, cx, cy, qx, qy);
  n = pclip<STRIDE>(qx, qy, n, cx, cy, dx, dy, px, py);
  n = pclip<STRIDE>(px, py, n, dx, dy, z, z, qx, qy);
  // shoelace (the polygon sits in q; the closing edge returns to vertex 0)
  T acc = 0;
  T x0 = qx[0], y0 = qy[0];
  for (int i = 0; i < n; i++) {
    const int nx = (i + 1 == n) ? 0 : i + 1;
    T x1 = qx[nx * STRIDE], y1 = qy[nx * STRIDE];
    acc += x0 * y1 - y0 * x1;
    x0 = x1; y0 = y1;
  }
  T r = pabs(acc * T(0.5));
  return (s1 * s2 == -1) ? -r : r;
}

template <typename T>
OBB_HD T quad_signed_area(const T* x, const T* y) {
  T acc = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) acc += x[i] * y[(i + 1) & 3] - y[i] * x[(i + 1) & 3];
  return acc * T(0.5);
}

// px,py,qx,qy: 4 scratch columns of 10 values each (stride STRIDE).
// DEGEN: the float device flavour's union == 0 rule (poly_nms_cuda.cu:136-137); the double host flavour
// (DOTA_devkit/polyiou.cpp:108-128) has none and returns inter / union as is (NaN for two empty rings).
template <int STRIDE, bool DEGEN, typename T>
OBB_HD T quad_iou_t(const QuadFeatT<T>& P, const QuadFeatT<T>& Q, T* px, T* py, T* qx, T* qy) {
  T ax[4], ay[4], bx[4], by[4];
  T a1 = quad_signed_area(P.x, P.y);
  T a2 = quad_signed_area(Q.x, Q.y);
  // reverse to counter-clockwise when the signed area is negative (:108-109)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    ax[i] = (a1 < 0) ? P.x[3 - i] : P.x[i]; ay[i] = (a1 < 0) ? P.y[3 - i] : P.y[i];
    bx[i] = (a2 < 0) ? Q.x[3 - i] : Q.x[i]; by[i] = (a2 < 0) ? Q.y[3 - i] : Q.y[i];
  }
  T inter = 0;
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      int i1 = (i + 1) & 3, j1 = (j + 1) & 3;
      // select with static indices to keep the vertex arrays in registers
      T pax = i == 0 ? ax[0] : i == 1 ? ax[1] : i == 2 ? ax[2] : ax[3];
      T pay = i == 0 ? ay[0] : i == 1 ? ay[1] : i == 2 ? ay[2] : ay[3];
      T pbx = i1 == 0 ? ax[0] : i1 == 1 ? ax[1] : i1 == 2 ? ax[2] : ax[3];
      T pby = i1 == 0 ? ay[0] : i1 == 1 ? ay[1] : i1 == 2 ? ay[2] : ay[3];
      T qcx = j == 0 ? bx[0] : j == 1 ? bx[1] : j == 2 ? bx[2] : bx[3];
      T qcy = j == 0 ? by[0] : j == 1 ? by[1] : j == 2 ? by[2] : by[3];
      T qdx = j1 == 0 ? bx[0] : j1 == 1 ? bx[1] : j1 == 2 ? bx[2] : bx[3];
      T qdy = j1 == 0 ? by[0] : j1 == 1 ? by[1] : j1 == 2 ? by[2] : by[3];
      inter += ptri_tri<STRIDE>(pax, pay, pbx, pby, qcx, qcy, qdx, qdy, px, py, qx, qy);
    }
  }
  // areas of the (possibly reversed) rings, as the reference recomputes them (:134)
  T ua = pabs(quad_signed_area(ax, ay)) + pabs(quad_signed_area(bx, by)) - inter;
  if (DEGEN && ua == T(0)) return (inter + T(1)) / (ua + T(1));   // :136-137
  return inter / ua;
}

template <int STRIDE>
OBB_HD float quad_iou(const QuadFeat& P, const QuadFeat& Q, float* px, float* py, float* qx, float* qy) {
  return quad_iou_t<STRIDE, true, float>(P, Q, px, py, qx, qy);
}

// ---- skipping pairs in the NMS hot loop ------------------------------------------------------------------------------
// quad_iou sums 16 signed triangle intersections taken from the coordinate ORIGIN.  For two quads whose bounding boxes are
// disjoint the exact sum is 0 (the product of the two winding numbers vanishes everywhere); the fp32 sum is rounding noise:
// every product in a term is O(M^2), M = largest |coordinate| of the pair, so |inter| <= c u M^2 with u = 2^-24 ("units").
//   * measured: tests/native/host_check_quadcull.cpp, ten adversarial families of bounding-box-disjoint pairs (slivers, bow
//     ties, edges along rays from the origin, all quadrants, touching boxes, a vertex at the origin, |coord| 8 .. 70000):
//     the largest noise over 1.6 x 10^10 pairs is 10.3 units, 22.9 after greedy ascents (profiles/r3_quad_noise.md; a sum of a
//     few hundred roundings of <= 1 unit each);
//   * worst case, every rounding aligned: shoelace (<= 6 products + accumulation per term) ~ 25 units per term, clip
//     vertices (<= 4 per term, each moved by a few ulps of M against edges <= 2.9 M) ~ 25 units per term, 16 terms:
//     several hundred units.
// The bound used is c = 1024 units: above the aligned-roundings accounting, > 40x the largest value found by search.  It is an
// engineering bound, not a machine-checked proof; OBB_NMS_POLY_STRICT=1 turns the skip off (every pair clipped).
// The absolute 1e-8 sign threshold of the reference drops or adds triangles of area <= 1e-8 each: kQuadSlack.
//
// A pair is a hit iff inter / (a1 + a2 - inter) > thr.  With |inter| <= E that cannot happen when E <= t (a1 + a2),
// t = thr / (1 + thr), and E <= c u (M1^2 + M2^2): with the per-box budget
//     f_i = t a_i - c u M_i^2 - slack
// a bounding-box-disjoint pair is skipped iff f_i + f_j >= 0 (the sign of a float sum is exact).  A big box near a small
// one pays for both; two small boxes far from the origin are always clipped, and so are two zero-area quads (both
// budgets negative), which keeps the reference's union == 0 rule.  thr <= 0 or a non-finite coordinate: f = -inf / NaN,
// never skipped.
//
// Hot-loop record (one 16-byte load per box, QuadGeom::q0): the bounding box as four fp16 values rounded OUTWARD
// (1 px steps at |coord| ~ 1000, +-inf beyond 65504: conservative) in two words, and f.
constexpr float kQuadNoiseUnits = 1024.f;
constexpr float kQuadNoise = kQuadNoiseUnits / 16777216.f;   // c u
constexpr float kQuadSlack = 1e-6f;
// The bound is a SEARCHED one: it is only used inside the envelope the searches covered (round 5; include/obb_hip.h states it at
// obb_nms_poly_f32) -- every coordinate |x|, |y| <= kQuadEnvCoord and a bounding box of at most kQuadEnvSize in either direction
// (the ten families of tests/native/host_check_quadcull.cpp: extents 8 .. 70,000 px, boxes 0.01 .. 600 px).  A quad outside it gets
// no budget (f = -inf): its pairs are decided by the proved cone rule or clipped.
constexpr float kQuadEnvCoord = 70000.f;
constexpr float kQuadEnvSize = 600.f;

// largest fp16 <= x (up = false) or smallest fp16 >= x (up = true), as bits; NaN -> NaN
OBB_HD uint32_t f16_bits_toward(float x, bool up) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const uint32_t sign = u >> 31, ax = u & 0x7fffffffu;
  if (ax > 0x7f800000u) return 0x7e00u;
  const bool away = up != (sign != 0u);                 // the direction grows the magnitude
  uint32_t h;
  const int e = (int)(ax >> 23) - 127;
  if (e >= 16) h = away ? 0x7c00u : 0x7bffu;            // beyond the largest finite fp16 (65504 < 2^16)
  else if (e < -14) {                                   // fp16 subnormal range: multiples of 2^-24
    const float sc = __builtin_bit_cast(float, ax) * 16777216.f;
    h = (uint32_t)sc;
    if (away && (float)h != sc) h++;
  } else {
    const uint32_t mant = ax & 0x7fffffu;
    h = ((uint32_t)(e + 15) << 10) | (mant >> 13);
    if (away && (mant & 0x1fffu)) h++;                  // the carry runs into the exponent, up to 0x7c00 = inf
  }
  return (sign << 15) | h;
}
OBB_HD float f16_bits_to_float(uint32_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
#else
  const uint32_t sign = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
  float v;
  if (e == 0) v = (float)m * (1.0f / 16777216.f);
  else if (e == 31) v = m ? __builtin_nanf("") : __builtin_huge_valf();
  else v = __builtin_bit_cast(float, ((e + 112u) << 23) | (m << 13));
  return sign ? -v : v;
#endif
}

// ---- the exact rule: the second quad's cone lies counter-clockwise of the first's ------------------------------------
// PROVED, at any coordinate magnitude (DESIGN.md section 4.1 spells the argument out).  quad_iou(P, Q) clips every triangle
// (o, a, b) of P -- o the coordinate origin, (a, b) an edge of P -- first against the line o -> c, c a vertex of Q, with
//     s(v) = fl(fl(cx * vy) - fl(vx * cy))          (ptri with the origin as the line's first point: no subtraction rounds)
// and keeps what lies to its LEFT (s > 1e-8).  If every vertex of P lies CLOCKWISE of every vertex of Q as seen from the
// origin, by an angle in [2e-4, pi - 4e-4], and no vertex is closer than 1 to the origin, then s(a) and s(b) are below -1e-8
// (exact cross product <= -2e-4 |c||v| <= -2.2e-4, rounding <= 2^-23 |c||v|: the sign is safe with or without FMA
// contraction), s(o) = 0 exactly, the two crossings are computed as (0 * s2 - v * 0) / s2 = 0 exactly, the clipped polygon is
// the single point (0, 0), every later clip keeps or drops that point, the shoelace sum over <= 1 point is 0: ALL 16 terms are
// exactly 0, inter = +0, and with a non-zero area on either side IoU = +0 -- never above a threshold >= 0.  (With P
// counter-clockwise of Q the last clip line runs d -> o and the intermediate polygon is not a point: no such statement, and
// none is used.)  The cone of a quad = the polar angles of its four vertices, as 16-bit fixed point (2 pi / 65536 per unit)
// widened by two units on either side, which also covers atan2f's error on any libm; a quad around the origin, across the
// negative x axis, with a vertex within 1.5 (L1) of the origin, with a non-finite coordinate or with zero computed area has no
// cone and is never skipped by this rule.
constexpr uint32_t kConeNone = 0x0000FFFFu;            // lo = 0xFFFF > hi = 0
OBB_HD uint32_t quad_cone_bits(const QuadFeat& q) {
  float lo = 4.f, hi = -4.f;
  bool ok = quad_signed_area(q.x, q.y) != 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float x = q.x[i], y = q.y[i];
    ok = ok && (x - x == 0.f) && (y - y == 0.f) && (fabsf(x) + fabsf(y) >= 1.5f);
    const float t = atan2f(y, x);
    lo = fminf(lo, t); hi = fmaxf(hi, t);
  }
  if (!ok || !(hi - lo < 3.1f)) return kConeNone;
  const float sc = 65536.f / 6.2831855f;
  float ulo = floorf((lo + 3.1415927f) * sc) - 2.f, uhi = ceilf((hi + 3.1415927f) * sc) + 2.f;
  ulo = ulo < 0.f ? 0.f : ulo; uhi = uhi > 65535.f ? 65535.f : uhi;
  return (uint32_t)ulo | ((uint32_t)uhi << 16);
}
// P = the FIRST argument of quad_iou (the NMS's row box), Q = the second (the column box)
OBB_HD bool quad_cone_skip(uint32_t p, uint32_t q) {
  const int plo = (int)(p & 0xffffu), phi = (int)(p >> 16), qlo = (int)(q & 0xffffu), qhi = (int)(q >> 16);
  return plo <= phi && qlo <= qhi && phi < qlo && (qhi - plo) < 32768 - 4;
}

// ---- the second exact rule: the FIRST quad's cone lies counter-clockwise of the second's -----------------------------
// PROVED under the conditions below for this file's contract, IEEE fp32 WITHOUT FMA contraction (DESIGN.md section 4.3 spells
// the argument out; tests/native/host_check_quadcone.cpp searches for counter-examples on the rule's edges).  (With contraction
// clip 3's s(Z) = fma(dx, dy, -fl(dx dy)) is a rounding residual instead of 0 and the statement does not hold; the first rule
// holds either way.)  Notation as above: P = the first
// argument, (a, b) an edge of P, (c, d) an edge of Q ordered so that the computed cross(c, d) > 1e-8, o the origin.  With every
// vertex of P counter-clockwise of every vertex of Q:
//   clip 1 (line o -> c): s(o) = 0 exactly, s(a), s(b) > 1e-8 (exact cross >= sin(margin) |c||a|, rounding <= 3 u |c||a|):
//     the polygon stays [Z, a, b], Z = (+-0, +-0);
//   clip 2 (line c -> d): s(Z) = cross(c, d) up to 4 u |d - c||c| -- safely positive when the edge's exact cross is
//     >= 2^-16 max(|c|, |d|)^2 (edge condition); a, b fall on either side.  The clipped polygon's vertices are Z, a, b, the
//     pinned (0, 0), and crossings: t a or t b with t = s(Z) / (s(Z) - s(.)) (on the rays o -> a, o -> b, no closer to the origin
//     than 0.49 x the distance r of the line c-d from it), or lambda a + (1 - lambda) b computed from the two SIGNS-DIFFER
//     values: lambda in (0, 1) when both lie outside +-1e-8, and within (-1, 2) when one of them lies inside (then
//     |s(b) - s(a)| > 1e-8 >= |s(a)| bounds the extrapolation by one edge length on either side: the EXTENDED edge);
//   clip 3 (line d -> o): s(v) = -cross(d, v) up to 3.1 u |d|(|v| + |d|); s(Z) = 0 exactly; for every other vertex v the exact
//     value is <= -|d||v| sin(margin) and |v| >= 0.49 r (M = the largest norm involved; M / r is bounded): s(v) < -1e-8.  Zero
//     points are never emitted as vertices, a crossing between a zero point and a negative one is (+-0 - +-0) / s = +-0:
//     the polygon is empty or the single point (0, 0), the shoelace sum +0.
// So all 16 terms are exactly 0.  What the rule needs per quad (quad_cone2_bits): the smallest distance r of an edge's LINE from
// the origin (rounded down; it bounds every vertex and every point of an extended edge from below), the largest norm M of
// the twelve points v_i, 2 v_i - v_(i+-1) (rounded up), the edge condition on all four edges, M <= 2^30; the cone of the four
// vertices (quad_cone_bits) and the cone of the twelve points.  Per pair, with k = max(M) / min(r) <= 4096:
//   * how far apart the cones must be: a crossing's direction moves by <= 17 u M / (0.49 r) = 2.1e-6 k rad, clip 3's sign needs
//     sin(angle) > 3.1 u (1 + M / (0.49 r)) = 3.8e-7 k: 0.026 k units of 2 pi / 65536 in all.  Asked for: 5 + k / 16 units between
//     the cones, both widened by two units (9 + k / 16 between any two points: k = 128: 17 against 3.5 needed, k = 4096: 265
//     against 112);
//   * the whole span stays below pi - 8 units;
//   * tier 1: the first quad's EXTENDED cone is used -- nothing else to check;
//   * tier 2: the first quad's plain cone is used and quad_cone2_nofuzzy vouches that none of clip 2's sixteen possible sign
//     values (a vertex of the first quad against an edge line of the second) lies within +-1e-8: every crossing on an edge
//     (a, b) is then a proper convex combination and stays inside the plain cone.
struct QuadCone2 {
  uint32_t ext;   // cone of the twelve points, 16-bit fixed point lo | hi << 16 (kConeNone: no tier 1)
  uint32_t rm;    // bf16 bits: r rounded down | M rounded up << 16   (0xffff0000: the quad takes no part in the rule)
};
constexpr uint32_t kRmNone = 0xffff0000u;
OBB_HD QuadCone2 quad_cone2_bits(const QuadFeat& q) {
  QuadCone2 out; out.ext = kConeNone; out.rm = kRmNone;
  float lo = 4.f, hi = -4.f, m2 = 0.f, r = __builtin_huge_valf();
  bool ok = quad_signed_area(q.x, q.y) != 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = (i + 1) & 3;
    const float ux = q.x[i], uy = q.y[i], wx = q.x[j], wy = q.y[j];
    ok = ok && (ux - ux == 0.f) && (uy - uy == 0.f) && (fabsf(ux) + fabsf(uy) >= 1.5f);
    const float cr = fabsf(ux * wy - wx * uy);
    const float nu = ux * ux + uy * uy, nw = wx * wx + wy * wy;
    // edge condition (computed cross >= 2^-15 max^2 covers the 3 u max^2 of its own roundings)
    ok = ok && (cr >= fmaxf(nu, nw) * (1.f / 32768.f));
    const float ex = wx - ux, ey = wy - uy;
    const float len = sqrtf(ex * ex + ey * ey);
    r = fminf(r, cr / len);
    // the three points of this edge's line that can carry a vertex: u, 2u - w (behind u), 2w - u (beyond w)
    const float px[3] = {ux, ux - ex, wx + ex}, py[3] = {uy, uy - ey, wy + ey};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float t = atan2f(py[k], px[k]);
      lo = fminf(lo, t); hi = fmaxf(hi, t);
      m2 = fmaxf(m2, px[k] * px[k] + py[k] * py[k]);
    }
  }
  const float M = sqrtf(m2) * 1.001f;
  r *= 0.99f;
  if (!ok || !(M <= 1073741824.f) || !(r > 0.f)) return out;
  const uint32_t rb = __builtin_bit_cast(uint32_t, r) >> 16;                       // truncation rounds a positive float down
  const uint32_t mu = __builtin_bit_cast(uint32_t, M);
  const uint32_t mb = (mu >> 16) + ((mu & 0xffffu) ? 1u : 0u);
  out.rm = rb | (mb << 16);
  if (!(hi - lo < 3.1f)) return out;
  const float sc = 65536.f / 6.2831855f;
  float ulo = floorf((lo + 3.1415927f) * sc) - 2.f, uhi = ceilf((hi + 3.1415927f) * sc) + 2.f;
  ulo = ulo < 0.f ? 0.f : ulo; uhi = uhi > 65535.f ? 65535.f : uhi;
  out.ext = (uint32_t)ulo | ((uint32_t)uhi << 16);
  return out;
}
// p_cone: the FIRST argument's extended cone (tier 1) or plain cone (tier 2, with quad_cone2_nofuzzy); q_cone: the second
// argument's plain cone; p_rm / q_rm: their (r, M)
OBB_HD bool quad_cone2_skip(uint32_t p_cone, uint32_t p_rm, uint32_t q_cone, uint32_t q_rm) {
  const int plo = (int)(p_cone & 0xffffu), phi = (int)(p_cone >> 16), qlo = (int)(q_cone & 0xffffu), qhi = (int)(q_cone >> 16);
  const int rmin = (int)((p_rm & 0xffffu) < (q_rm & 0xffffu) ? (p_rm & 0xffffu) : (q_rm & 0xffffu));
  const int mmax = (int)((p_rm >> 16) > (q_rm >> 16) ? (p_rm >> 16) : (q_rm >> 16));
  // bf16 bits of positive numbers order like the numbers; 128 bits per octave, the mantissa's piecewise-linear logarithm is
  // off by < 0.09 octaves: k is taken as the next power of two above the bits' difference (a true ratio <= 1.07 k)
  const int d = mmax - rmin;
  if (d > 12 * 128) return false;
  const int kexp = d <= 0 ? 0 : (d + 127) >> 7;
  const int gap = 5 + (((1 << kexp) + 15) >> 4);
  return plo <= phi && qlo <= qhi && plo - qhi >= gap && (phi - qlo) < 32768 - 8;
}
// Tier 2's pair check: t = cross(w - u, v - u) for every edge (u, w) of Q and vertex v of P, in one orientation; clip 2 computes
// the same quantity from either end of the edge with <= 4 u (|ex| + |ey|)(|v - c|_1) of rounding.  Asked for:
// |t| > 2 x 2^-20 L + 2e-8, L = (|ex| + |ey|)(|v - u|_1 + |e|_1): four times the roundings of t and of the clip's own value.
OBB_HD bool quad_cone2_nofuzzy(const QuadFeat& P, const QuadFeat& Q) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int j1 = (j + 1) & 3;
    const float ux = Q.x[j], uy = Q.y[j], ex = Q.x[j1] - ux, ey = Q.y[j1] - uy;
    const float e1 = fabsf(ex) + fabsf(ey);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float vx = P.x[i] - ux, vy = P.y[i] - uy;
      const float t = ex * vy - vx * ey;
      const float L = e1 * (fabsf(vx) + fabsf(vy) + e1);
      ok = ok && (fabsf(t) > L * (1.f / 524288.f) + 2e-8f);
    }
  }
  return ok;
}

struct QuadSkip {
  uint32_t lo;   // fp16 minx | fp16 miny << 16   (rounded down)
  uint32_t hi;   // fp16 maxx | fp16 maxy << 16   (rounded up)
  float f;       // budget (see above)
};

OBB_HD QuadSkip quad_skip_record(const QuadFeat& q, float thr) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) m = fmaxf(m, fmaxf(fabsf(q.x[i]), fabsf(q.y[i])));
  bool nan = false;                                      // fmaxf drops NaNs: look for them explicitly
#pragma unroll
  for (int i = 0; i < 4; i++) nan = nan || q.x[i] != q.x[i] || q.y[i] != q.y[i];
  const float area = fabsf(quad_signed_area(q.x, q.y));
  const float t = thr / (1.f + thr);
  QuadSkip r;
  r.lo = f16_bits_toward(q.minx, false) | (f16_bits_toward(q.miny, false) << 16);
  r.hi = f16_bits_toward(q.maxx, true) | (f16_bits_toward(q.maxy, true) << 16);
  // the 1e-3 margins cover the roundings of this expression
  const bool in_env = m <= kQuadEnvCoord && (q.maxx - q.minx) <= kQuadEnvSize && (q.maxy - q.miny) <= kQuadEnvSize;   // (false on NaN)
  r.f = (thr > 0.f && !nan && in_env) ? area * t * 0.999f - kQuadNoise * 1.001f * m * m - kQuadSlack : -__builtin_huge_valf();
  return r;
}
OBB_HD bool quad_skip_pair(const QuadSkip& a, const QuadSkip& b) {
  const float aminx = f16_bits_to_float(a.lo & 0xffffu), aminy = f16_bits_to_float(a.lo >> 16);
  const float amaxx = f16_bits_to_float(a.hi & 0xffffu), amaxy = f16_bits_to_float(a.hi >> 16);
  const float bminx = f16_bits_to_float(b.lo & 0xffffu), bminy = f16_bits_to_float(b.lo >> 16);
  const float bmaxx = f16_bits_to_float(b.hi & 0xffffu), bmaxy = f16_bits_to_float(b.hi >> 16);
  const bool apart = (aminx > bmaxx) || (bminx > amaxx) || (aminy > bmaxy) || (bminy > amaxy);
  return apart && (a.f + b.f >= 0.f);
}

}  // namespace obb
