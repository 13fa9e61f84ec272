// Host-side bit check of the product's DOUBLE-precision rotated IoU (csrc/riou64_device.h, compiled with g++) against
// the CPU oracle's double flavour (oracle/riou_impl.inc with REAL = double, pinned to the reference's own header).
// Built and run by tests/test_host_geometry.py.   usage: host_check_riou64 <n_pairs> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include "riou64_device.h"
extern "C" double oracle_riou_f64(const double*, const double*);

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 200000;
  unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 0;
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  long mism = 0, nonzero = 0;
  double px[24], py[24];
  for (long i = 0; i < n; i++) {
    double a[5], b[5];
    int mode = i % 10;
    double spread = (mode < 4) ? 100.0 : 30.0;
    for (int k = 0; k < 2; k++) { a[k] = U(g) * spread; b[k] = U(g) * spread; }
    for (int k = 2; k < 4; k++) { a[k] = U(g) * 60 + 4; b[k] = U(g) * 60 + 4; }
    a[4] = (U(g) - 0.5) * 3.14159265358979; b[4] = (U(g) - 0.5) * 3.14159265358979;
    if (mode == 1) memcpy(b, a, sizeof a);                                                   // identical
    if (mode == 2) { a[4] = 0; b[4] = 0; for (int k = 0; k < 4; k++) { a[k] = round(a[k]); b[k] = round(b[k]); } }   // grid aligned
    if (mode == 3) { b[0] = a[0] + 4096.0 * 3 + U(g); a[0] += 4096.0 * 3; }                  // class-offset magnitudes
    if (mode == 5) { b[2] = a[2]; b[3] = a[3]; b[4] = a[4] + 1.5707963267948966; b[0] = a[0]; b[1] = a[1]; }   // crossed
    if (mode == 6) { a[2] = 200; a[3] = 3; b[2] = 150; b[3] = 2; }                           // thin
    if (mode == 7) { b[0] = a[0] + a[2]; b[1] = a[1]; b[4] = a[4] = 0; b[2] = a[2]; }        // edge-touching
    if (mode == 8) { a[3] = 1e-9; a[0] += 300.0; }                                           // needle far away
    if (mode == 9) { memcpy(b, a, sizeof a); b[0] += 1e-7 * (U(g) - 0.5); b[4] += 1e-9; }    // near-identical: float32 cannot tell them apart
    obb::RBoxFeat64 A = obb::rbox_make_feat64(a[0], a[1], a[2], a[3], a[4]);
    obb::RBoxFeat64 B = obb::rbox_make_feat64(b[0], b[1], b[2], b[3], b[4]);
    const double ref = oracle_riou_f64(a, b);
    const double got = obb::rbox_iou_f64<1>(A, B, px, py);
    if (memcmp(&ref, &got, 8) != 0) { if (mism < 5) printf("MISMATCH mode %d ref %.17g got %.17g\n", mode, ref, got); mism++; }
    if (ref > 0) nonzero++;
  }
  printf("mismatches64=%ld nonzero=%ld n=%ld\n", mism, nonzero, n);
  return mism ? 1 : 0;
}
