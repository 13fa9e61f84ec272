// How far from zero does the reference's quad "intersection" land for quads whose bounding boxes are disjoint?
//
// devPolyIoU (utils/nms_rotated/src/poly_nms_cuda.cu:99-142) sums 16 signed triangle intersections taken from the coordinate
// origin; for disjoint quads the exact sum is 0, the fp32 sum is rounding noise that scales with the SQUARE of the coordinate
// magnitude M.  The NMS hot loop skips a pair only when a bound on that noise cannot reach the threshold
// (piou_device.h: quad_skip_record / quad_skip_pair, noise bound kQuadNoise * (M_i^2 + M_j^2)).  This program measures the noise over adversarial families of
// AABB-disjoint pairs with the product's own device function (bit-checked against the oracle by host_check_piou) and
//   * reports   max |inter| / (2^-24 * M^2)   per family  (the bound is kQuadNoiseUnits of these units),
//   * counts pairs the cull would skip whose IoU exceeds the threshold (must be 0),
//   * with a third argument: hill-climbs from the noisiest random pairs (ulp-sized and larger moves of single coordinates
//     that keep the boxes disjoint) to look for inputs whose roundings line up.
//   usage: host_check_quadcull <n_pairs> <seed> [climb_steps [climb_from_units]]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include "piou_device.h"

static std::mt19937 g;
static float U() { return std::uniform_real_distribution<float>(0.f, 1.f)(g); }

static void rect(float cx, float cy, float w, float h, float a, float* q) {
  float c = cosf(a), s = sinf(a);
  const float sx[4] = {1, 1, -1, -1}, sy[4] = {1, -1, -1, 1};
  for (int k = 0; k < 4; k++) { q[2 * k] = cx + sx[k] * w / 2 * c - sy[k] * h / 2 * s; q[2 * k + 1] = cy + sx[k] * w / 2 * s + sy[k] * h / 2 * c; }
}

static const float thrs[4] = {0.01f, 0.1f, 0.4f, 0.9f};
static const float kSpan[6] = {100.f, 1000.f, 1024.f, 5000.f, 70000.f, 8.f};

// one quad of family f around a random centre
static void make(int f, float span, float* q) {
  float cx = U() * span, cy = U() * span;
  switch (f) {
    case 0: rect(cx, cy, U() * 60 + 4, U() * 60 + 4, (U() - 0.5f) * 3.14159265f, q); break;
    case 1: rect(cx, cy, U() * 2 + 0.01f, U() * 490 + 10, (U() - 0.5f) * 3.14159265f, q); break;               // slivers
    case 2: { float d = U() * 100 + 1; for (int k = 0; k < 8; k++) q[k] = ((k & 1) ? cy : cx) + (U() - 0.5f) * d; } break;   // any 4 points (also bow ties)
    case 3: { float r = U() * span + 1, phi = U() * 1.5707963f;                                                  // edges along rays from the origin
              rect(r * cosf(phi), r * sinf(phi), U() * 60 + 4, U() * 20 + 0.5f, phi + (U() - 0.5f) * 1e-3f * (float)(g() % 3), q); } break;
    case 4: rect(cx - span / 2, cy - span / 2, U() * 60 + 4, U() * 60 + 4, (U() - 0.5f) * 3.14159265f, q); break;   // all four quadrants
    case 5: rect(cx, cy, U() * 600 + 4, U() * 600 + 4, (U() - 0.5f) * 3.14159265f, q); break;                     // large boxes
    case 6: rect(cx, cy, U() * 60 + 4, U() * 60 + 4, (U() - 0.5f) * 3.14159265f, q); for (int k = 0; k < 8; k++) q[k] = roundf(q[k]); break;
    case 7: rect(cx, cy, U() * 60 + 4, U() * 60 + 4, 0.f, q); break;                                              // axis aligned (touching AABBs below)
    case 8: rect(cx, cy, U() * 60 + 4, U() * 60 + 4, (U() - 0.5f) * 3.14159265f, q); q[2 * (g() % 4)] = 0.f; q[2 * (g() % 4) + 1] = 0.f; break;  // a vertex on an axis / at the origin
    default: rect(cx, cy, U() * 6 + 0.05f, U() * 6 + 0.05f, (U() - 0.5f) * 3.14159265f, q); break;                // tiny boxes
  }
}

// The 16-term sum of quad_iou_t itself (same calls in the same order: piou_device.h quad_iou_t), so that the noise is read off
// directly instead of being backed out of the IoU (inter = iou A / (1 + iou) loses everything when the noise dwarfs the areas
// and the IoU rounds to -1).  `check`: the IoU rebuilt from this sum must have the bits of quad_iou.
static float inter_of(const obb::QuadFeat& P, const obb::QuadFeat& Q, float* iou_rebuilt) {
  float s0[10], s1[10], s2[10], s3[10];
  float ax[4], ay[4], bx[4], by[4];
  const float a1 = obb::quad_signed_area(P.x, P.y), a2 = obb::quad_signed_area(Q.x, Q.y);
  for (int i = 0; i < 4; i++) {
    ax[i] = (a1 < 0) ? P.x[3 - i] : P.x[i]; ay[i] = (a1 < 0) ? P.y[3 - i] : P.y[i];
    bx[i] = (a2 < 0) ? Q.x[3 - i] : Q.x[i]; by[i] = (a2 < 0) ? Q.y[3 - i] : Q.y[i];
  }
  float inter = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      inter += obb::ptri_tri<1>(ax[i], ay[i], ax[(i + 1) & 3], ay[(i + 1) & 3], bx[j], by[j], bx[(j + 1) & 3], by[(j + 1) & 3], s0, s1, s2, s3);
  const float ua = fabsf(obb::quad_signed_area(ax, ay)) + fabsf(obb::quad_signed_area(bx, by)) - inter;
  *iou_rebuilt = (ua == 0.f) ? (inter + 1.f) / (ua + 1.f) : inter / ua;
  return inter;
}
static long g_mirror_bad = 0, g_iou_minus_one = 0;

static double noise_units(const float* p, const float* q) {   // < 0: not a bounding-box-disjoint pair
  float s0[10], s1[10], s2[10], s3[10];
  obb::QuadFeat P = obb::quad_make_feat(p), Q = obb::quad_make_feat(q);
  if (!(P.minx > Q.maxx || Q.minx > P.maxx || P.miny > Q.maxy || Q.miny > P.maxy)) return -1.0;
  (void)s0; (void)s1; (void)s2; (void)s3;
  float rebuilt;
  const float inter = inter_of(P, Q, &rebuilt);
  float m = 0.f;
  for (int k = 0; k < 8; k++) m = fmaxf(m, fmaxf(fabsf(p[k]), fabsf(q[k])));
  if (inter != inter || m == 0.f) return -1.0;
  return fabs((double)inter) / (ldexp(1.0, -24) * (double)m * m);
}

// greedy ascent on the noise: single-coordinate moves of 1..64 ulps or of a random fraction of a pixel
static double climb(float* p, float* q, long steps) {
  double best = noise_units(p, q);
  for (long s = 0; s < steps; s++) {
    float* v = (g() & 1) ? p : q;
    const int k = (int)(g() % 8);
    const float old = v[k];
    if (g() & 1) { int n = 1 + (int)(g() % 64); float t = old; for (int i = 0; i < n; i++) t = nextafterf(t, (g() & 1) ? INFINITY : -INFINITY); v[k] = t; }
    else v[k] = old + (U() - 0.5f) * (float)(1 << (g() % 4)) * 0.25f;
    const double r = noise_units(p, q);
    if (r > best) best = r; else v[k] = old;
  }
  return best;
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 2000000;
  const long climb_steps = argc > 3 ? atol(argv[3]) : 0;
  const double climb_from = argc > 4 ? atof(argv[4]) : 2.5;      // noise (units) from which a random pair starts an ascent
  double climbed = 0;
  g.seed(argc > 2 ? (unsigned)atol(argv[2]) : 0u);
  const int NF = 10;
  double worst[NF] = {0}, worst_sum[NF] = {0};
  long used[NF] = {0}, culled = 0, wrong = 0;
  long cone_skipped = 0, cone_wrong = 0, cone_overlapping_boxes = 0;
  float s0[10], s1[10], s2[10], s3[10];
  for (long i = 0; i < n; i++) {
    const int f = (int)(i % NF);
    const float span = kSpan[(i / NF) % 6];
    float p[8], q[8];
    make(f, span, p); make(f == 8 ? 0 : f, span, q);
    if (f == 7 && (i & 1)) {   // shift q so that the boxes touch up to one ulp
      obb::QuadFeat P = obb::quad_make_feat(p), Q = obb::quad_make_feat(q);
      const float dx = nextafterf(P.maxx, INFINITY) - Q.minx;
      for (int k = 0; k < 4; k++) q[2 * k] += dx;
    }
    obb::QuadFeat P = obb::quad_make_feat(p), Q = obb::quad_make_feat(q);
    const bool disjoint = P.minx > Q.maxx || Q.minx > P.maxx || P.miny > Q.maxy || Q.miny > P.maxy;
    // The exact rule (piou_device.h quad_cone_skip): whenever it fires -- in either role -- the 16-term sum must be EXACTLY +0
    // and the IoU +0 (a property a single counter-example would falsify; the bounding boxes need not even be disjoint).
    for (int role = 0; role < 2; role++) {
      const obb::QuadFeat& A = role ? Q : P; const obb::QuadFeat& B = role ? P : Q;
      if (obb::quad_cone_skip(obb::quad_cone_bits(A), obb::quad_cone_bits(B))) {
        cone_skipped++;
        if (!disjoint) cone_overlapping_boxes++;
        float rebuilt;
        const float it = inter_of(A, B, &rebuilt);
        const float v = obb::quad_iou<1>(A, B, s0, s1, s2, s3);
        uint32_t ib, vb; memcpy(&ib, &it, 4); memcpy(&vb, &v, 4);
        if (ib != 0u || vb != 0u) cone_wrong++;
      }
    }
    if (!disjoint) {   // the skip works on outward-rounded fp16 boxes: it must never fire for overlapping bounding boxes
      for (int t = 0; t < 4; t++)
        if (obb::quad_skip_pair(obb::quad_skip_record(P, thrs[t]), obb::quad_skip_record(Q, thrs[t]))) { if (wrong < 5) printf("WRONG: skipped an overlapping pair\n"); wrong++; }
      continue;
    }
    const float iou = obb::quad_iou<1>(P, Q, s0, s1, s2, s3);
    const double A = (double)fabsf(obb::quad_signed_area(P.x, P.y)) + (double)fabsf(obb::quad_signed_area(Q.x, Q.y));
    float mp = 0.f, mq = 0.f;
    for (int k = 0; k < 8; k++) { mp = fmaxf(mp, fabsf(p[k])); mq = fmaxf(mq, fabsf(q[k])); }
    float rebuilt;
    const float inter_f = inter_of(P, Q, &rebuilt);
    if (memcmp(&rebuilt, &iou, 4) != 0 && !(rebuilt != rebuilt && iou != iou)) { if (g_mirror_bad < 5) printf("MIRROR family %d iou %.9g rebuilt %.9g\n", f, iou, rebuilt); g_mirror_bad++; }
    if (iou == -1.0f) g_iou_minus_one++;      // noise far above the two areas: the back-out formula of an earlier version divided by 0 here
    if (inter_f == inter_f) {
      const double inter = fabs((double)inter_f);
      const double M = fmax(mp, mq), u = ldexp(1.0, -24);
      const double r = inter / (u * M * M), rs = inter / (u * ((double)mp * mp + (double)mq * mq));
      if (r > worst[f]) worst[f] = r;
      if (rs > worst_sum[f]) worst_sum[f] = rs;
      used[f]++;
      if (climb_steps > 0 && r > climb_from) { float p2[8], q2[8]; memcpy(p2, p, 32); memcpy(q2, q, 32); climbed = fmax(climbed, climb(p2, q2, climb_steps)); }
    }
    for (int t = 0; t < 4; t++) {
      const obb::QuadSkip sp = obb::quad_skip_record(P, thrs[t]), sq = obb::quad_skip_record(Q, thrs[t]);
      if (obb::quad_skip_pair(sp, sq)) { culled++; if (iou > thrs[t]) { if (wrong < 5) printf("WRONG family %d thr %g iou %.9g\n", f, thrs[t], iou); wrong++; } }
    }
  }
  // the outward fp16 rounding of the record: down <= x <= up, and each is the nearest such fp16
  long bad16 = 0;
  for (long i = 0; i < 4000000; i++) {
    const int k = (int)(i % 6);
    float x = (U() - 0.5f) * (k == 0 ? 2e-7f : k == 1 ? 2e-4f : k == 2 ? 2.f : k == 3 ? 4096.f : k == 4 ? 140000.f : 1e9f);
    if (i % 97 == 0) x = roundf(x);
    if (i % 1001 == 0) x = (i & 1) ? INFINITY : -INFINITY;
    const uint32_t d = obb::f16_bits_toward(x, false), u2 = obb::f16_bits_toward(x, true);
    const float fd = obb::f16_bits_to_float(d), fu = obb::f16_bits_to_float(u2);
    bool ok = fd <= x && x <= fu;
    // tightness: the next fp16 above `down` is > x (unless down == x), the next below `up` is < x
    if (ok && fd != x && fabsf(x) < 65000.f) {
      const uint32_t nd = (d & 0x8000u) ? ((d & 0x7fffu) == 0 ? 0x0000u : d - 1) : d + 1;          // next fp16 toward +inf
      ok = obb::f16_bits_to_float(nd) > x || (d == 0x8000u && x > 0.f);
    }
    if (ok && fu != x && fabsf(x) < 65000.f) {
      const uint32_t nu = (u2 & 0x8000u) ? u2 + 1 : ((u2 & 0x7fffu) == 0 ? 0x8001u : u2 - 1);      // next fp16 toward -inf
      ok = obb::f16_bits_to_float(nu) < x;
    }
    if (!ok) { if (bad16 < 5) printf("FP16 ROUNDING x %.9g down %.9g up %.9g\n", x, fd, fu); bad16++; }
  }
  printf("fp16_rounding_errors=%ld\n", bad16);
  printf("mirror_mismatches=%ld pairs_with_iou_exactly_minus_one=%ld\n", g_mirror_bad, g_iou_minus_one);
  wrong += bad16 + g_mirror_bad;
  double w = 0, ws = 0;
  for (int f = 0; f < NF; f++) {
    printf("family %d: disjoint pairs %ld  max noise %.1f units (by M_i^2+M_j^2: %.1f)\n", f, used[f], worst[f], worst_sum[f]);
    w = fmax(w, worst[f]); ws = fmax(ws, worst_sum[f]);
  }
  if (climb_steps > 0) printf("worst_after_climb_units=%.1f\n", climbed);
  printf("cone_skipped=%ld cone_wrong=%ld cone_with_overlapping_boxes=%ld\n", cone_skipped, cone_wrong, cone_overlapping_boxes);
  printf("worst_noise_units=%.1f bound_units=%.0f culled=%ld wrong=%ld\n", w, (double)obb::kQuadNoiseUnits, culled, wrong);
  return wrong ? 1 : 0;
}
