// Rotated-rectangle IoU in DOUBLE precision for gfx950 -- the reference's double instantiation.
//
// Behavioural contract: utils/nms_rotated/src/box_iou_rotated_utils.h:333-360 with T = double (what
// nms_rotated_cuda.cu:96, AT_DISPATCH_FLOATING_TYPES, runs for float64 tensors), device hull branch (:195-218),
// IEEE double without FMA contraction -- the contract of oracle/riou_impl.inc with REAL = double, which is pinned to
// the reference's own header by tests/golden (riou_ref_dev_f64) and tests/test_oracle_vs_ref.py.
//
// Same formulation as riou_device.h (the float flavour): cos / sin and the four products s2*h, c2*w, c2*h, s2*w depend on
// one box only and are hoisted into a per-box record (each is a single rounded product in the reference too); the <= 24
// candidate points live in a caller-provided scratch column with a compile-time stride; the hull is built in place.
// No shortcuts: a float64 call is a precision request, every pair that passes the (conservative, fp32) circle test of
// the hot loop runs this clip.
#pragma once
#include "obb_device.h"

namespace obb {

struct RBoxFeat64 {
  double x, y;            // centre
  double sh, cw, ch, sw;  // sin(a)*0.5*h, cos(a)*0.5*w, cos(a)*0.5*h, sin(a)*0.5*w
  double area;            // w*h
};

OBB_HD RBoxFeat64 rbox_make_feat64(double x, double y, double w, double h, double a) {
  RBoxFeat64 f;
  const double c2 = cos(a) * 0.5f;   // (T)cos(theta) * 0.5f, box_iou_rotated_utils.h:63-65 (0.5f widens exactly)
  const double s2 = sin(a) * 0.5f;
  f.x = x; f.y = y;
  f.sh = s2 * h; f.cw = c2 * w; f.ch = c2 * h; f.sw = s2 * w;
  f.area = w * h;                    // :351-352
  return f;
}

// Full clip.  A = higher-scored ("row") box, B = lower-scored ("column") box (nms_rotated_cuda.cu:60).
// px/py: scratch for 24 points, element i at [i * STRIDE].
template <int STRIDE>
OBB_HD double rbox_iou_f64(const RBoxFeat64& A, const RBoxFeat64& B, double* px, double* py) {
  if (A.area < 1e-14 || B.area < 1e-14) return 0.0;  // :353

  const double mx = (A.x + B.x) / 2.0, my = (A.y + B.y) / 2.0;   // :338-339
  const double ax = A.x - mx, ay = A.y - my, bx = B.x - mx, by = B.y - my;

  double v1x[4], v1y[4], v2x[4], v2y[4];
  v1x[0] = ax + A.sh + A.cw; v1y[0] = ay + A.ch - A.sw;
  v1x[1] = ax - A.sh + A.cw; v1y[1] = ay - A.ch - A.sw;
  v1x[2] = 2 * ax - v1x[0];  v1y[2] = 2 * ay - v1y[0];
  v1x[3] = 2 * ax - v1x[1];  v1y[3] = 2 * ay - v1y[1];
  v2x[0] = bx + B.sh + B.cw; v2y[0] = by + B.ch - B.sw;
  v2x[1] = bx - B.sh + B.cw; v2y[1] = by - B.ch - B.sw;
  v2x[2] = 2 * bx - v2x[0];  v2y[2] = 2 * by - v2y[0];
  v2x[3] = 2 * bx - v2x[1];  v2y[3] = 2 * by - v2y[1];

  double e1x[4], e1y[4], e2x[4], e2y[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    e1x[i] = v1x[(i + 1) & 3] - v1x[i]; e1y[i] = v1y[(i + 1) & 3] - v1y[i];
    e2x[i] = v2x[(i + 1) & 3] - v2x[i]; e2y[i] = v2y[(i + 1) & 3] - v2y[i];
  }

  int n = 0;
  // 16 edge/edge crossings (:93-112)
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double det = e2x[j] * e1y[i] - e1x[i] * e2y[j];
      if (fabs(det) <= 1e-14) continue;
      const double dx = v2x[j] - v1x[i], dy = v2y[j] - v1y[i];
      const double t1 = (e2x[j] * dy - dx * e2y[j]) / det;
      const double t2 = (e1x[i] * dy - dx * e1y[i]) / det;
      if (t1 >= 0.0 && t1 <= 1.0 && t2 >= 0.0 && t2 <= 1.0) {
        px[n * STRIDE] = v1x[i] + e1x[i] * t1;
        py[n * STRIDE] = v1y[i] + e1y[i] * t1;
        n++;
      }
    }
  }
  // corners of A inside B (:115-135), then corners of B inside A (:138-154)
  {
    const double abab = e2x[0] * e2x[0] + e2y[0] * e2y[0];
    const double adad = e2x[3] * e2x[3] + e2y[3] * e2y[3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double apx = v1x[i] - v2x[0], apy = v1y[i] - v2y[0];
      const double apab = apx * e2x[0] + apy * e2y[0];
      const double apad = -(apx * e2x[3] + apy * e2y[3]);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        px[n * STRIDE] = v1x[i]; py[n * STRIDE] = v1y[i]; n++;
      }
    }
  }
  {
    const double abab = e1x[0] * e1x[0] + e1y[0] * e1y[0];
    const double adad = e1x[3] * e1x[3] + e1y[3] * e1y[3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double apx = v2x[i] - v1x[0], apy = v2y[i] - v1y[0];
      const double apab = apx * e1x[0] + apy * e1y[0];
      const double apad = -(apx * e1x[3] + apy * e1y[3]);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        px[n * STRIDE] = v2x[i]; py[n * STRIDE] = v2y[i]; n++;
      }
    }
  }

  double inter = 0.0;
  if (n > 2) {
    // ---- Graham scan in place (:159-291) ----
    int t = 0;
    double tx = px[0], ty = py[0];
    for (int i = 1; i < n; i++) {
      const double x = px[i * STRIDE], y = py[i * STRIDE];
      if (y < ty || (y == ty && x < tx)) { t = i; tx = x; ty = y; }
    }
    for (int i = 0; i < n; i++) { px[i * STRIDE] -= tx; py[i * STRIDE] -= ty; }
    {
      const double x0 = px[0], y0 = py[0];
      px[0] = px[t * STRIDE]; py[0] = py[t * STRIDE];
      px[t * STRIDE] = x0; py[t * STRIDE] = y0;
    }
    // exchange sort by polar angle around the pivot, ties by distance (:205-218)
    for (int i = 1; i < n - 1; i++) {
      double qix = px[i * STRIDE], qiy = py[i * STRIDE];
      for (int j = i + 1; j < n; j++) {
        const double qjx = px[j * STRIDE], qjy = py[j * STRIDE];
        const double cp = qix * qjy - qjx * qiy;
        bool sw = cp < -1e-6;
        if (!sw && fabs(cp) < 1e-6) sw = (qix * qix + qiy * qiy) > (qjx * qjx + qjy * qjy);
        if (sw) {
          px[j * STRIDE] = qix; py[j * STRIDE] = qiy;
          qix = qjx; qiy = qjy;
        }
      }
      px[i * STRIDE] = qix; py[i * STRIDE] = qiy;
    }
    // first point that is not a duplicate of the pivot (:239-249)
    int k = 1;
    for (; k < n; k++) {
      const double x = px[k * STRIDE], y = py[k * STRIDE];
      if (x * x + y * y > 1e-8) break;
    }
    if (k < n) {
      px[1 * STRIDE] = px[k * STRIDE]; py[1 * STRIDE] = py[k * STRIDE];
      int m = 2;
      for (int i = k + 1; i < n; i++) {
        const double qx = px[i * STRIDE], qy = py[i * STRIDE];
        while (m > 1) {
          const double bx2 = px[(m - 2) * STRIDE], by2 = py[(m - 2) * STRIDE];
          const double q1x = qx - bx2, q1y = qy - by2;
          const double q2x = px[(m - 1) * STRIDE] - bx2, q2y = py[(m - 1) * STRIDE] - by2;
          if (q1x * q2y >= q2x * q1y) m--; else break;  // two rounded products (:266)
        }
        px[m * STRIDE] = qx; py[m * STRIDE] = qy; m++;
      }
      // fan area (:293-305)
      if (m > 2) {
        const double q0x = px[0], q0y = py[0];
        double acc = 0.0;
        double pxx = px[1 * STRIDE] - q0x, pyy = py[1 * STRIDE] - q0y;
        for (int i = 1; i < m - 1; i++) {
          const double nx = px[(i + 1) * STRIDE] - q0x, ny = py[(i + 1) * STRIDE] - q0y;
          acc += fabs(pxx * ny - nx * pyy);
          pxx = nx; pyy = ny;
        }
        inter = acc / 2.0;
      }
    }
  }
  return inter / (A.area + B.area - inter);  // :358
}

}  // namespace obb
