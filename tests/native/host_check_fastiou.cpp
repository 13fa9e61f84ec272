// Host-side check (g++, no GPU) of the PRODUCT's register-only IoU interval filter (riou_device.h:
// rbox_fast_iou_bounds) against the CPU oracle on detector-like pair distributions: whenever the filter vouches
// for a pair, the oracle's IoU must lie inside [lo, hi]; prints how many pairs the filter decides per threshold.
//   usage: host_check_fastiou <n_pairs> <seed>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include "riou_device.h"
extern "C" float oracle_riou_f32(const float*, const float*);

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 1000000;
  unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 0;
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::normal_distribution<float> N(0.f, 1.f);
  const float thrs[6] = {0.0f, 0.1f, 0.2f, 0.4f, 0.45f, 0.6f};
  long viol = 0, vouched[8] = {0}, tot[8] = {0}, decided[8][6] = {{0}};
  long qviol = 0, qvouch[8] = {0}, qdec[8][6] = {{0}};
  for (long i = 0; i < n; i++) {
    const int mode = (int)(i % 8);
    float a[5], b[5];
    a[0] = U(g) * 1024; a[1] = U(g) * 1024; a[2] = U(g) * 60 + 8; a[3] = U(g) * 60 + 8; a[4] = (U(g) - 0.5f) * 3.14159265f;
    for (int k = 0; k < 5; k++) b[k] = a[k];
    float jit = 2.f, sz = 0.1f, ang = 0.05f;
    if (mode == 1) ang = powf(10.f, -1.f - 5.f * U(g));                       // nearly parallel: 1e-1 .. 1e-6 rad
    if (mode == 2) { a[2] = U(g) * 300 + 100; a[3] = U(g) * 8 + 3; b[2] = a[2]; b[3] = a[3]; }   // thin, aspect up to 100
    if (mode == 3) { float off = 4096.f * (float)(1 + (int)(U(g) * 17)); a[0] += off; a[1] += off; b[0] = a[0]; b[1] = a[1]; }
    if (mode == 4) { a[2] = U(g) * 1000 + 300; a[3] = U(g) * 600 + 100; b[2] = a[2]; b[3] = a[3]; jit = 40.f; }
    if (mode == 5) { jit = 25.f; sz = 0.4f; ang = 1.5f; }                      // loosely related boxes, all IoU values
    if (mode == 6) { jit = 0.f; sz = 0.f; ang = 0.f; b[0] += (U(g) < 0.5f ? a[2] : 0.f) * (0.5f + U(g)); }   // aligned copies / abutting
    if (mode == 7) { a[2] = U(g) * 6 + 1; a[3] = U(g) * 3 + 1; b[2] = a[2]; b[3] = a[3]; jit = 0.7f; }       // few-pixel boxes
    b[0] += N(g) * jit; b[1] += N(g) * jit;
    b[2] *= fminf(fmaxf(1.f + sz * N(g), 0.5f), 1.5f); b[3] *= fminf(fmaxf(1.f + sz * N(g), 0.5f), 1.5f);
    b[4] += ang * N(g);
    obb::RBoxFeat A = obb::rbox_make_feat(a[0], a[1], a[2], a[3], a[4]);
    obb::RBoxFeat B = obb::rbox_make_feat(b[0], b[1], b[2], b[3], b[4]);
    tot[mode]++;
    {   // the cheap bounds in front of the interval filter: must contain the oracle's value whenever they vouch
      obb::IouBounds qb;
      if (obb::rbox_quick_bounds(A, B, &qb)) {
        const float refq = oracle_riou_f32(a, b);
        qvouch[mode]++;
        if (!(qb.lo <= refq && refq <= qb.hi)) {
          if (qviol < 10) printf("QUICK-BOUNDS VIOLATION mode %d ref %.9g lo %.9g hi %.9g  a=(%g %g %g %g %g) b=(%g %g %g %g %g)\n", mode, refq, qb.lo,
                                 qb.hi, a[0], a[1], a[2], a[3], a[4], b[0], b[1], b[2], b[3], b[4]);
          qviol++;
        }
        for (int q = 0; q < 6; q++) if (qb.lo > thrs[q] || qb.hi <= thrs[q]) qdec[mode][q]++;
      }
    }
    obb::IouBounds bd;
    if (!obb::rbox_fast_iou_bounds(A, B, &bd)) continue;
    vouched[mode]++;
    const float ref = oracle_riou_f32(a, b);
    if (!(bd.lo <= ref && ref <= bd.hi)) {
      if (viol < 10) printf("VIOLATION mode %d ref %.9g lo %.9g hi %.9g  a=(%g %g %g %g %g) b=(%g %g %g %g %g)\n", mode, ref, bd.lo, bd.hi,
                            a[0], a[1], a[2], a[3], a[4], b[0], b[1], b[2], b[3], b[4]);
      viol++;
    }
    for (int q = 0; q < 6; q++) if (bd.lo > thrs[q] || bd.hi <= thrs[q]) decided[mode][q]++;
  }
  const char* names[8] = {"near-duplicate", "nearly-parallel", "thin", "class-offset", "large", "loose", "aligned/abutting", "few-pixel"};
  for (int m = 0; m < 8; m++)
    printf("mode %-17s pairs %8ld vouched %5.1f%%  decided@0/.1/.2/.4/.45/.6 = %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %% of all pairs\n", names[m],
           tot[m], 100.0 * vouched[m] / tot[m], 100.0 * decided[m][0] / tot[m], 100.0 * decided[m][1] / tot[m], 100.0 * decided[m][2] / tot[m],
           100.0 * decided[m][3] / tot[m], 100.0 * decided[m][4] / tot[m], 100.0 * decided[m][5] / tot[m]);
  for (int m = 0; m < 8; m++)
    printf("quick bounds, mode %-17s vouched %5.1f%%  decided@0/.1/.2/.4/.45/.6 = %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %% of all pairs\n",
           names[m], 100.0 * qvouch[m] / tot[m], 100.0 * qdec[m][0] / tot[m], 100.0 * qdec[m][1] / tot[m], 100.0 * qdec[m][2] / tot[m],
           100.0 * qdec[m][3] / tot[m], 100.0 * qdec[m][4] / tot[m], 100.0 * qdec[m][5] / tot[m]);
  printf("quick_bounds_violations=%ld\n", qviol);
  printf("violations=%ld n=%ld\n", viol, n);
  return (viol || qviol) ? 1 : 0;
}
