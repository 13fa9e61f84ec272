// Host-side bit check of the product's quad-IoU device code against the CPU oracle.
//   usage: host_check_piou <n_pairs> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include "piou_device.h"
extern "C" float oracle_piou_f32(const float*, const float*);
extern "C" double oracle_piou_f64(const double*, const double*);

static void mk(std::mt19937& g, float spread, float off, float* q) {
  std::uniform_real_distribution<float> U(0.f, 1.f);
  float cx = U(g) * spread + off, cy = U(g) * spread + off, w = U(g) * 60 + 4, h = U(g) * 60 + 4, a = (U(g) - 0.5f) * 3.14159265f;
  float c = cosf(a), s = sinf(a);
  const float sx[4] = {1, 1, -1, -1}, sy[4] = {1, -1, -1, 1};
  for (int k = 0; k < 4; k++) { q[2 * k] = cx + sx[k] * w / 2 * c - sy[k] * h / 2 * s; q[2 * k + 1] = cy + sx[k] * w / 2 * s + sy[k] * h / 2 * c; }
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 100000;
  unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 0;
  std::mt19937 g(seed);
  long mism = 0, nonzero = 0, disjoint = 0, mism64 = 0;
  double d0[10], d1[10], d2[10], d3[10];
  double worst_noise = 0;
  float s0[10], s1[10], s2[10], s3[10];
  for (long i = 0; i < n; i++) {
    float p[9], q[9];
    int mode = i % 8;
    float off = (mode == 3) ? 5000.f : 0.f;
    mk(g, mode < 4 ? 100.f : 30.f, off, p); mk(g, mode < 4 ? 100.f : 30.f, off, q);
    if (mode == 1) memcpy(q, p, sizeof p);
    if (mode == 2) { for (int k = 0; k < 8; k++) { p[k] = roundf(p[k]); q[k] = roundf(q[k]); } }
    if (mode == 5) { float t[8]; memcpy(t, q, 32); for (int k = 0; k < 4; k++) { q[2 * k] = t[2 * (3 - k)]; q[2 * k + 1] = t[2 * (3 - k) + 1]; } }  // clockwise ring
    if (mode == 6) { for (int k = 0; k < 8; k++) q[k] = p[0 + (k & 1)]; }   // degenerate point
    if (mode == 7) { for (int k = 0; k < 8; k++) { p[k] = p[k & 1]; q[k] = q[k & 1]; } }  // both degenerate -> rule
    obb::QuadFeat P = obb::quad_make_feat(p), Q = obb::quad_make_feat(q);
    float ref = oracle_piou_f32(p, q);
    float got = obb::quad_iou<1>(P, Q, s0, s1, s2, s3);
    if (memcmp(&ref, &got, 4) != 0 && !(ref != ref && got != got)) { if (mism < 5) printf("MISMATCH mode %d ref %.9g got %.9g\n", mode, ref, got); mism++; }
    {   // double flavour (DOTA_devkit/polyiou.cpp; no degenerate rule: NaN for two empty rings)
      double pd[8], qd[8];
      obb::QuadFeatT<double> PD, QD;
      for (int k = 0; k < 8; k++) { pd[k] = (double)p[k] * 1.000000123 + 0.1; qd[k] = (double)q[k] * 1.000000123 + 0.1; }
      if (mode == 1) memcpy(qd, pd, sizeof pd);
      for (int k = 0; k < 4; k++) { PD.x[k] = pd[2 * k]; PD.y[k] = pd[2 * k + 1]; QD.x[k] = qd[2 * k]; QD.y[k] = qd[2 * k + 1]; }
      double r64 = oracle_piou_f64(pd, qd);
      double g64 = obb::quad_iou_t<1, false, double>(PD, QD, d0, d1, d2, d3);
      if (memcmp(&r64, &g64, 8) != 0 && !(r64 != r64 && g64 != g64)) { if (mism64 < 5) printf("MISMATCH64 mode %d ref %.17g got %.17g\n", mode, r64, g64); mism64++; }
    }
    if (ref > 0) nonzero++;
    if (P.minx > Q.maxx || Q.minx > P.maxx || P.miny > Q.maxy || Q.miny > P.maxy) { disjoint++; if (fabs(ref) > worst_noise && mode != 7) worst_noise = fabs(ref); }
  }
  printf("mismatches64=%ld\n", mism64);
  printf("mismatches=%ld nonzero=%ld aabb_disjoint=%ld worst_disjoint_iou=%.3g n=%ld\n", mism, nonzero, disjoint, worst_noise, n);
  return (mism || mism64) ? 1 : 0;
}
