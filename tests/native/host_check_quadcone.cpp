// The exact skip rule of the quad IoU (piou_device.h: quad_cone_bits / quad_cone_skip): whenever the rule
// fires, the reference's 16-term sum must be EXACTLY +0 and the IoU +0.  Pairs are generated to sit ON the rule's edges:
// angular gaps from the smallest the fixed-point cones can resolve, spans up to pi, vertices at the minimum distance from the
// origin, coordinates from 2 to 10^7, slivers, bow ties, clockwise rings, edges along rays, all quadrants and the wrap at the
// negative x axis.  Compiled twice by tests/test_host_geometry.py: without and WITH FMA contraction (nvcc's default for the
// reference): the statement holds either way.
//   usage: host_check_quadcone <n_pairs> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include "piou_device.h"

static std::mt19937 g;
static float U() { return std::uniform_real_distribution<float>(0.f, 1.f)(g); }

// a quad inside the polar sector [t0, t1] x [r0, r1]: four points (any order: bow ties and clockwise rings included)
static void sector_quad(float t0, float t1, float r0, float r1, float* q) {
  for (int k = 0; k < 4; k++) {
    const float t = t0 + (t1 - t0) * U(), r = r0 + (r1 - r0) * U();
    q[2 * k] = r * cosf(t); q[2 * k + 1] = r * sinf(t);
  }
  if (g() % 4 == 0) {                                      // a proper rectangle-like ring from the extreme angles
    q[0] = r0 * cosf(t0); q[1] = r0 * sinf(t0); q[2] = r1 * cosf(t0); q[3] = r1 * sinf(t0);
    q[4] = r1 * cosf(t1); q[5] = r1 * sinf(t1); q[6] = r0 * cosf(t1); q[7] = r0 * sinf(t1);
  }
  if (g() % 3 == 0) for (int k = 0; k < 2; k++) { std::swap(q[2 * k], q[6 - 2 * k]); std::swap(q[2 * k + 1], q[7 - 2 * k]); }   // reversed winding
}

// a rotated rectangle (detector-like quad): centre at polar (t, dist), sides w x h, turned by phi
static void rect_quad(float t, float dist, float w, float h, float phi, float* q) {
  const float cx = dist * cosf(t), cy = dist * sinf(t), c = cosf(phi), s = sinf(phi);
  const float dx[4] = {w / 2, w / 2, -w / 2, -w / 2}, dy[4] = {-h / 2, h / 2, h / 2, -h / 2};
  for (int k = 0; k < 4; k++) { q[2 * k] = cx + c * dx[k] - s * dy[k]; q[2 * k + 1] = cy + s * dx[k] + c * dy[k]; }
}
static void rotate_quad(float* q, float a) {
  const double c = cos((double)a), s = sin((double)a);
  for (int k = 0; k < 4; k++) { const double x = q[2 * k], y = q[2 * k + 1]; q[2 * k] = (float)(c * x - s * y); q[2 * k + 1] = (float)(s * x + c * y); }
}
static const float kUnit = 6.2831855f / 65536.f;
// the second rule as k_quad_strip applies it: tier 1 (extended cone) or tier 2 (plain cone + the pair check); *tier = which
static long g_tier[3] = {0, 0, 0};
static bool rule2_fires(const obb::QuadFeat& A, const obb::QuadFeat& B) {
  const obb::QuadCone2 a2 = obb::quad_cone2_bits(A), b2 = obb::quad_cone2_bits(B);
  const uint32_t cb = obb::quad_cone_bits(B);
  if (obb::quad_cone2_skip(a2.ext, a2.rm, cb, b2.rm)) { g_tier[1]++; return true; }
  if (obb::quad_cone2_skip(obb::quad_cone_bits(A), a2.rm, cb, b2.rm) && obb::quad_cone2_nofuzzy(A, B)) { g_tier[2]++; return true; }
  return false;
}

// Families aimed at the SECOND rule (the first argument counter-clockwise of the second):
//   0: two rectangles, the gap between the first one's extended cone and the second one's cone set to -3 .. 12 units around the
//      rule's edge (5); distances chosen so that max(M) / min(r) straddles 128 in a quarter of the cases;
//   1: an edge (a, b) of the first quad ON the line through an edge (c, d) of the second, beyond d, moved off it by 0 .. 3 ulps:
//      clip 2's two values are rounding noise around +-1e-8 (the extrapolated crossing);
//   2: as 1 with the first quad's edge nearly parallel at a distance of 1e-9 .. 1e-3 of the line;
//   3: integer coordinates (exact zeros in the sign tests).
static bool family2_pair(long i, float* p, float* q, float* gap_units) {
  const int fam = (int)(i % 4);
  if (fam == 1 && (i / 4) % 2 == 0) {
    // 1b: both edges on a line parallel to an axis at a small distance x0 from the origin, moved off it by 1e-10 .. 1e-7: the
    // products in clip 2's sign values are small enough for the values to land within +-2e-8 (the extrapolated crossing)
    const float M = 2.f + 14.f * U(), x0 = M / 128.f * (1.05f + 3.f * U());
    float y[4]; y[0] = 1.5f + 0.1f * M * U(); y[1] = y[0] + 0.02f * M + 0.1f * M * U(); y[2] = y[1] + 0.1f * M + 0.2f * M * U(); y[3] = y[2] + 0.02f * M + 0.05f * M * U();
    float dl[4]; for (int k = 0; k < 4; k++) dl[k] = powf(10.f, -10.f + 3.f * U()) * (g() % 2 ? 1.f : -1.f) * (g() % 4 ? 1.f : 0.f);
    const float wq = 0.3f + 2.f * U(), wp = 0.05f + 0.5f * U();
    q[0] = x0 + wq; q[1] = y[0]; q[2] = x0 + dl[0]; q[3] = y[0]; q[4] = x0 + dl[1]; q[5] = y[1]; q[6] = x0 + wq; q[7] = y[1];
    p[0] = x0 + dl[2]; p[1] = y[2]; p[2] = x0 + dl[3]; p[3] = y[3]; p[4] = x0 - wp; p[5] = y[3] + wp * U(); p[6] = x0 - wp; p[7] = y[2] + wp * U();
    if (g() % 2) { std::swap(p[0], p[6]); std::swap(p[1], p[7]); std::swap(p[2], p[4]); std::swap(p[3], p[5]); }
    if (g() % 2) { std::swap(q[0], q[6]); std::swap(q[1], q[7]); std::swap(q[2], q[4]); std::swap(q[3], q[5]); }
    for (int rot = (int)(g() % 4); rot > 0; rot--)                         // exact quarter turns: the same case on every axis
      for (int k = 0; k < 4; k++) { float t = p[2 * k]; p[2 * k] = -p[2 * k + 1]; p[2 * k + 1] = t; t = q[2 * k]; q[2 * k] = -q[2 * k + 1]; q[2 * k + 1] = t; }
    *gap_units = 100.f;
    return true;
  }
  static const float kDist[6] = {8.f, 60.f, 500.f, 1400.f, 20000.f, 3e6f};
  const float dq = kDist[(i / 4) % 6] * (0.5f + U());
  const float tq = (U() - 0.5f) * 5.5f;
  rect_quad(tq, dq, 2.f + U() * (g() % 3 ? 60.f : 600.f), 2.f + U() * 60.f, U() * 3.14159f, q);
  if (fam == 0 || fam == 3) {
    const float dp = (g() % 4 == 0) ? dq * (U() < 0.5f ? 0.01f + 0.02f * U() : 30.f + 80.f * U()) : dq * (0.3f + 2.f * U());
    rect_quad(tq, dp, 2.f + U() * (g() % 3 ? 60.f : 600.f), 2.f + U() * 60.f, U() * 3.14159f, p);
    if (fam == 3) for (int k = 0; k < 8; k++) { p[k] = roundf(p[k]); q[k] = roundf(q[k]); }
  } else {
    // edge 1 -> 2 of q prolonged beyond vertex 2 (or, for the reversed order, beyond vertex 1)
    const bool rev = g() % 2;
    const float cx = rev ? q[4] : q[2], cy = rev ? q[5] : q[3], dx = rev ? q[2] : q[4], dy = rev ? q[3] : q[5];
    const float ex = dx - cx, ey = dy - cy, len = sqrtf(ex * ex + ey * ey);
    const float mu1 = 0.2f + 3.f * U(), mu2 = mu1 + 0.3f + 3.f * U();
    float ax = dx + mu1 * ex, ay = dy + mu1 * ey, bx = dx + mu2 * ex, by = dy + mu2 * ey;
    const float off = fam == 1 ? 0.f : powf(10.f, -9.f + 6.f * U()) * (g() % 2 ? 1.f : -1.f);
    ax += -ey / len * off; ay += ex / len * off; bx += -ey / len * off * (0.5f + U()); by += ex / len * off * (0.5f + U());
    if (fam == 1) for (int k = (int)(g() % 4); k > 0; k--) { ax = nextafterf(ax, g() % 2 ? 1e30f : -1e30f); by = nextafterf(by, g() % 2 ? 1e30f : -1e30f); }
    const float hh = (2.f + U() * 40.f) * (g() % 2 ? 1.f : -1.f);
    p[0] = ax; p[1] = ay; p[2] = bx; p[3] = by; p[4] = bx - ey / len * hh; p[5] = by + ex / len * hh; p[6] = ax - ey / len * hh; p[7] = ay + ex / len * hh;
    if (g() % 2) { std::swap(p[0], p[6]); std::swap(p[1], p[7]); std::swap(p[2], p[4]); std::swap(p[3], p[5]); }
    if (fam == 2) return true;          // stays where it is: whether the rule fires is up to the geometry
    *gap_units = 0.f;
    return true;
  }
  // turn p about the origin until the gap (first extended cone's begin - second cone's end) has the wanted size
  const obb::QuadFeat Q = obb::quad_make_feat(q);
  const uint32_t cq = obb::quad_cone_bits(Q);
  for (int it = 0; it < 3; it++) {
    const obb::QuadCone2 c2 = obb::quad_cone2_bits(obb::quad_make_feat(p));
    if (c2.ext == obb::kConeNone || cq == obb::kConeNone) return false;
    // the gap the rule asks for grows with max(M) / min(r): aim at -3 .. +12 units around it, for the extended cone (tier 1) or
    // the plain one (tier 2)
    const obb::QuadFeat Pf = obb::quad_make_feat(p);
    const obb::QuadCone2 q2 = obb::quad_cone2_bits(Q);
    const bool plain = (i / 8) % 2;
    const uint32_t pc = plain ? obb::quad_cone_bits(Pf) : c2.ext;
    if (pc == obb::kConeNone) return false;
    int need = 5;
    for (int gq = 5; gq < 300; gq++) {                      // the smallest gap quad_cone2_skip accepts for this pair's (r, M)
      const uint32_t fake_p = 40000u | (40100u << 16), fake_q = (uint32_t)(40000 - gq - 50) | ((uint32_t)(40000 - gq) << 16);
      if (obb::quad_cone2_skip(fake_p, c2.rm, fake_q, q2.rm)) { need = gq; break; }
      need = 300;
    }
    if (need >= 300) return false;
    const float want = (float)need - 8.f + 15.f * U();
    const float have = (float)(int)(pc & 0xffffu) - (float)(int)(cq >> 16);
    rotate_quad(p, (want - have) * kUnit);
    *gap_units = want - (float)need + 5.f;
  }
  return true;
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4000000;
  g.seed(argc > 2 ? (unsigned)atol(argv[2]) : 0u);
  const bool rule2 = argc > 3 ? atol(argv[3]) != 0 : true;      // the second rule's contract is "no FMA contraction": the FMA build passes 0
  long fired = 0, wrong = 0, near_edge = 0, fired2 = 0, wrong2 = 0, near_edge2 = 0;
  float s0[10], s1[10], s2[10], s3[10];
  static const float kScale[8] = {2.f, 10.f, 300.f, 1024.f, 5000.f, 70000.f, 1e6f, 1e7f};
  static const float kGap[8] = {1e-4f, 2e-4f, 3e-4f, 5e-4f, 1e-3f, 1e-2f, 0.3f, 1.5f};
  for (long i = 0; i < n; i++) {
    const float scale = kScale[i % 8];
    const float base = (U() - 0.5f) * 6.2831853f;                 // anywhere, incl. across the negative x axis
    const float wp = U() < 0.5f ? U() * 0.02f : U() * 1.2f, wq = U() < 0.5f ? U() * 0.02f : U() * 1.2f;
    const float gap = kGap[(i / 8) % 8] * (0.5f + U());
    float p[8], q[8];
    const float r0 = (g() % 5 == 0) ? 1.5f : scale * (0.05f + U());
    sector_quad(base, base + wp, r0, r0 + scale * U(), p);                                  // P: clockwise side
    sector_quad(base + wp + gap, base + wp + gap + wq, scale * (0.05f + U()), scale * (1.f + U()), q);   // Q: counter-clockwise side
    if (g() % 16 == 0) for (int k = 0; k < 8; k++) { p[k] = roundf(p[k]); q[k] = roundf(q[k]); }
    const obb::QuadFeat P = obb::quad_make_feat(p), Q = obb::quad_make_feat(q);
    for (int role = 0; rule2 && role < 2; role++) {
      // the second rule (first argument counter-clockwise of the second): same statement, exact +0
      const obb::QuadFeat& A = role ? Q : P; const obb::QuadFeat& B = role ? P : Q;
      if (!rule2_fires(A, B)) continue;
      fired2++;
      if (gap < 1.2e-3f) near_edge2++;
      const float v = obb::quad_iou<1>(A, B, s0, s1, s2, s3);
      uint32_t vb; memcpy(&vb, &v, 4);
      if (vb != 0u) { wrong2++; if (wrong2 < 5) fprintf(stderr, "counter-example (rule 2): iou bits %08x gap %g scale %g\n", vb, gap, scale); }
    }
    for (int role = 0; role < 2; role++) {
      const obb::QuadFeat& A = role ? Q : P; const obb::QuadFeat& B = role ? P : Q;
      if (!obb::quad_cone_skip(obb::quad_cone_bits(A), obb::quad_cone_bits(B))) continue;
      fired++;
      if (gap < 6e-4f) near_edge++;
      const float v = obb::quad_iou<1>(A, B, s0, s1, s2, s3);
      uint32_t vb; memcpy(&vb, &v, 4);
      if (vb != 0u) { wrong++; if (wrong < 5) fprintf(stderr, "counter-example: iou bits %08x gap %g scale %g\n", vb, gap, scale); }
    }
  }
  long fam_fired[4] = {0, 0, 0, 0}, edge2 = 0;
  for (long i = 0; rule2 && i < n; i++) {
    float p[8], q[8], gu = 100.f;
    if (!family2_pair(i, p, q, &gu)) continue;
    const obb::QuadFeat P = obb::quad_make_feat(p), Q = obb::quad_make_feat(q);
    if (!rule2_fires(P, Q)) continue;
    fired2++; fam_fired[i % 4]++;
    if (gu < 8.f) edge2++;
    const float v = obb::quad_iou<1>(P, Q, s0, s1, s2, s3);
    uint32_t vb; memcpy(&vb, &v, 4);
    if (vb != 0u) { wrong2++; if (wrong2 < 5) fprintf(stderr, "counter-example (rule 2, family %ld): iou bits %08x\n", i % 4, vb); }
  }
  printf("family2_fired=%ld,%ld,%ld,%ld at_edge2=%ld tier1=%ld tier2=%ld ", fam_fired[0], fam_fired[1], fam_fired[2], fam_fired[3], edge2, g_tier[1], g_tier[2]);
  printf("pairs=%ld fired=%ld near_edge=%ld wrong=%ld fired2=%ld near_edge2=%ld wrong2=%ld\n", n, fired, near_edge, wrong, fired2, near_edge2, wrong2);
  return (wrong || wrong2) ? 1 : 0;
}
