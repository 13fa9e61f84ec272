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

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4000000;
  g.seed(argc > 2 ? (unsigned)atol(argv[2]) : 0u);
  long fired = 0, wrong = 0, near_edge = 0;
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
  printf("pairs=%ld fired=%ld near_edge=%ld wrong=%ld\n", n, fired, near_edge, wrong);
  return wrong ? 1 : 0;
}
