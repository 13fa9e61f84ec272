// TEST INFRASTRUCTURE.  Thin extern "C" door onto the REFERENCE's own header
// (utils/nms_rotated/src/box_iou_rotated_utils.h), which is compiled from where
// it lies under /root/reference (-I on the command line; nothing is copied).
// Built twice by oracle/Makefile:
//   REF_TAG=host : the header's CPU branch (std::sort hull ordering, :219-234)
//   REF_TAG=dev  : the header's CUDA/HIP branch (:195-218) emulated on the host
//                  with -D__HIP__=1 -D__host__= -D__device__= -D__forceinline__=inline
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
