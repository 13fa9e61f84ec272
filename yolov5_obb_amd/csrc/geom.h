// Geometry policies shared by the NMS and pairwise kernels: how a box is laid
// out in an LDS tile (SoA, [field][64 lanes]), its conservative reject and its
// exact IoU with the per-lane clip scratch in an LDS column.
#pragma once
#include <hip/hip_runtime.h>
#include "obb_device.h"
#include "riou_device.h"
#include "piou_device.h"

namespace obb {

struct RotGeom {
  static constexpr int NF = 12;
  static constexpr int SCR = 48;  // 24 points x (x,y)
  using Feat = RBoxFeat;
  static __device__ __forceinline__ Feat load(const float* t, int i) {
    Feat f;
    f.x = t[0 * 64 + i]; f.y = t[1 * 64 + i]; f.w = t[2 * 64 + i]; f.h = t[3 * 64 + i];
    f.sh = t[4 * 64 + i]; f.cw = t[5 * 64 + i]; f.ch = t[6 * 64 + i]; f.sw = t[7 * 64 + i];
    f.r = t[8 * 64 + i]; f.c = t[9 * 64 + i]; f.s = t[10 * 64 + i]; f.area = t[11 * 64 + i];
    return f;
  }
  static __device__ __forceinline__ bool reject(const Feat& A, const Feat& B, float thr) {
    if (rbox_certainly_disjoint(A, B)) return true;
    return rbox_iou_upper_bound(A, B) <= thr;
  }
  static __device__ __forceinline__ float iou(const Feat& A, const Feat& B, float* scr) {
    return rbox_iou<64>(A, B, scr, scr + 24 * 64);
  }
};

struct QuadGeom {
  static constexpr int NF = 8;
  static constexpr int SCR = 40;  // 2 x 10 points x (x,y)
  using Feat = QuadFeat;
  static __device__ __forceinline__ Feat load(const float* t, int i) {
    Feat f;
#pragma unroll
    for (int k = 0; k < 4; k++) { f.x[k] = t[(2 * k) * 64 + i]; f.y[k] = t[(2 * k + 1) * 64 + i]; }
    f.minx = f.maxx = f.miny = f.maxy = 0.f;
    return f;
  }
  // The reference's quad IoU sums signed triangle areas taken from the coordinate
  // origin; for disjoint quads the terms cancel only up to rounding, so "IoU == 0"
  // cannot be predicted cheaply (measured up to 0.06 at |coord| ~ 5000).  No reject.
  static __device__ __forceinline__ bool reject(const Feat&, const Feat&, float) { return false; }
  static __device__ __forceinline__ float iou(const Feat& A, const Feat& B, float* scr) {
    return quad_iou<64>(A, B, scr, scr + 10 * 64, scr + 20 * 64, scr + 30 * 64);
  }
};

}  // namespace obb
