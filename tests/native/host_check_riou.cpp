// Host-side bit check of the *product's* device geometry code (compiled with g++,
// no GPU needed) against the CPU oracle.  Built and run by tests/test_host_geometry.py.
//   usage: host_check_riou <n_pairs> <seed>
// prints: mismatches=<k> cull_violations=<k> ub_violations=<k> culled=<k> nonzero=<k>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include "riou_device.h"
extern "C" float oracle_riou_f32(const float*, const float*);

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 200000;
  unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 0;
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  long mism = 0, cullv = 0, ubv = 0, culled = 0, nonzero = 0;
  long fb_safe = 0, fb_viol = 0, fb_cand = 0, fb_decided[4] = {0, 0, 0, 0};
  const float thrs[4] = {0.1f, 0.2f, 0.4f, 0.45f};
  double fb_width = 0.0;
  float px[24], py[24];
  for (long i = 0; i < n; i++) {
    float a[5], b[5];
    int mode = i % 10;
    float spread = (mode < 4) ? 100.f : 30.f;
    a[0] = U(g) * spread; a[1] = U(g) * spread; a[2] = U(g) * 60 + 4; a[3] = U(g) * 60 + 4; a[4] = (U(g) - 0.5f) * 3.14159265f;
    b[0] = U(g) * spread; b[1] = U(g) * spread; b[2] = U(g) * 60 + 4; b[3] = U(g) * 60 + 4; b[4] = (U(g) - 0.5f) * 3.14159265f;
    if (mode == 1) memcpy(b, a, sizeof a);                       // identical
    if (mode == 2) { a[4] = 0; b[4] = 0; for (int k = 0; k < 4; k++) { a[k] = roundf(a[k]); b[k] = roundf(b[k]); } }  // grid aligned
    if (mode == 3) { b[0] = a[0] + 4096.f * 3; a[0] += 4096.f * 3; }   // class-offset magnitudes, same class
    if (mode == 5) { b[2] = a[2]; b[3] = a[3]; b[4] = a[4] + 1.57079633f; b[0] = a[0]; b[1] = a[1]; }  // crossed
    if (mode == 6) { a[2] = 200; a[3] = 3; b[2] = 150; b[3] = 2; }     // thin
    if (mode == 8) { a[3] = 1e-9f; a[0] += 300.f; }                     // thin box far away: reference gives garbage IoU ~ 1
    if (mode == 9) { b[2] = 3e-6f; b[1] += 150.f; a[0] += 4096.f * 7; b[0] += 4096.f * 7; }
    if (mode == 7) { b[0] = a[0] + a[2]; b[1] = a[1]; b[4] = a[4] = 0; b[2] = a[2]; } // edge-touching
    obb::RBoxFeat A = obb::rbox_make_feat(a[0], a[1], a[2], a[3], a[4]);
    obb::RBoxFeat B = obb::rbox_make_feat(b[0], b[1], b[2], b[3], b[4]);
    float ref = oracle_riou_f32(a, b);
    float got = obb::rbox_iou<1>(A, B, px, py);
    if (memcmp(&ref, &got, 4) != 0) { if (mism < 5) printf("MISMATCH mode %d ref %.9g got %.9g\n", mode, ref, got); mism++; }
    if (obb::rbox_certainly_disjoint(A, B)) { culled++; if (ref != 0.f) cullv++; }
    if (ref > obb::rbox_iou_upper_bound(A, B)) ubv++;
    if (ref > 0) nonzero++;
    // fast interval filter: must contain the reference value whenever it vouches for the pair
    if (!obb::rbox_certainly_disjoint(A, B)) {
      fb_cand++;
      obb::IouBounds bd;
      if (obb::rbox_fast_iou_bounds(A, B, &bd)) {
        fb_safe++;
        fb_width += (double)(bd.hi - bd.lo);
        if (!(bd.lo <= ref && ref <= bd.hi)) { if (fb_viol < 8) printf("BOUND VIOLATION mode %d ref %.9g lo %.9g hi %.9g\n", mode, ref, bd.lo, bd.hi); fb_viol++; }
        for (int q = 0; q < 4; q++) if (bd.lo > thrs[q] || bd.hi <= thrs[q]) fb_decided[q]++;
      }
    }
  }
  printf("fast_bounds: candidates=%ld vouched=%ld violations=%ld mean_width=%.3g decided@0.1/0.2/0.4/0.45=%ld/%ld/%ld/%ld\n", fb_cand, fb_safe,
         fb_viol, fb_safe ? fb_width / fb_safe : 0.0, fb_decided[0], fb_decided[1], fb_decided[2], fb_decided[3]);
  if (fb_viol) return 1;
  printf("mismatches=%ld cull_violations=%ld ub_violations=%ld culled=%ld nonzero=%ld n=%ld\n", mism, cullv, ubv, culled, nonzero, n);
  return (mism || cullv || ubv) ? 1 : 0;
}
