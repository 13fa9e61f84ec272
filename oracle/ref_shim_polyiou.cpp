// TEST INFRASTRUCTURE.  extern "C" door onto the REFERENCE's DOTA_devkit/polyiou.cpp
// (double-precision polygon IoU, iou_poly at polyiou.cpp:108-128), compiled from
// /root/reference by oracle/Makefile.  Nothing is copied into this repository.
#include <cstdint>
#include <vector>
#include "polyiou.h"

extern "C" double ref_iou_poly(const double* p, const double* q) {
  return iou_poly(std::vector<double>(p, p + 8), std::vector<double>(q, q + 8));
}
extern "C" void ref_iou_poly_pairs(const double* p, const double* q, int64_t n, double* out) {
  for (int64_t i = 0; i < n; i++)
    out[i] = iou_poly(std::vector<double>(p + 8 * i, p + 8 * i + 8), std::vector<double>(q + 8 * i, q + 8 * i + 8));
}
