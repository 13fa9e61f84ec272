// TEST INFRASTRUCTURE.  Thin extern "C" door onto the REFERENCE's own header
// (utils/nms_rotated/src/box_iou_rotated_utils.h), which is compiled from where
// it lies under /root/reference (-I on the command line; nothing is copied).
// Built twice by oracle/Makefile:
//   REF_TAG=host : the header's CPU branch (std::sort hull ordering, :219-234)
//   REF_TAG=dev  : the header's CUDA/HIP branch (:195-218) emulated on the host
//                  with -D__HIP__=1 -D__host__= -D__device__= -D__forceinline__=inline
//   REF_TAG=devfma : the same branch compiled with -ffp-contract=fast -mfma, i.e. with the FMA contraction nvcc applies
//                  by default (the header itself comments on it at :262-266) -- used to QUANTIFY how many kept
//                  indices can differ between a contracted and an uncontracted build (tests/test_fma_sensitivity.py)
#include <algorithm>
#include <cstdint>
#include "box_iou_rotated_utils.h"

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define NAME(n) CAT(CAT(ref_, REF_TAG), CAT(_, n))

extern "C" float NAME(riou_f32)(const float* a, const float* b) {
  return single_box_iou_rotated<float>(a, b);
}
extern "C" double NAME(riou_f64)(const double* a, const double* b) {
  return single_box_iou_rotated<double>(a, b);
}
extern "C" void NAME(riou_pairs_f32)(const float* a, const float* b, int64_t n, float* out) {
  for (int64_t i = 0; i < n; i++) out[i] = single_box_iou_rotated<float>(a + 5 * i, b + 5 * i);
}
extern "C" void NAME(riou_pairs_f64)(const double* a, const double* b, int64_t n, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = single_box_iou_rotated<double>(a + 5 * i, b + 5 * i);
}

// Greedy scan of nms_rotated_cuda.cu:109-128 (strict >) over boxes already in descending-score order, with the reference's
// own IoU.  Pairs whose circumscribed circles are clearly apart are skipped (IoU == 0 for them in either build; the
// sizes here are far from the ill-conditioned regime), so that N = 100k finishes in seconds; the inner loop is split
// over the host's cores (same result for any thread count: rows are processed in order, columns independently).
#include <cmath>
#include <vector>
extern "C" int64_t NAME(nms_gt_sorted)(const float* d, int64_t n, float thr, int64_t* keep_pos) {
  std::vector<unsigned char> dead((size_t)n, 0);
  std::vector<float> rad((size_t)n);
  for (int64_t i = 0; i < n; i++) rad[i] = std::sqrt(d[5 * i + 2] * d[5 * i + 2] + d[5 * i + 3] * d[5 * i + 3]) * 0.51f;
  int64_t k = 0;
  for (int64_t i = 0; i < n; i++) {
    if (dead[i]) continue;
    keep_pos[k++] = i;
    const float* a = d + 5 * i;
#pragma omp parallel for schedule(static)
    for (int64_t j = i + 1; j < n; j++) {
      if (dead[j]) continue;
      const float dx = d[5 * j] - a[0], dy = d[5 * j + 1] - a[1], rs = rad[i] + rad[j];
      if (dx * dx + dy * dy > rs * rs) continue;
      if (single_box_iou_rotated<float>(a, d + 5 * j) > thr) dead[j] = 1;
    }
  }
  return k;
}
