// Host build of the PRODUCT's loss math (yolov5_obb_amd/csrc/loss_math.h) as a tiny shared library so that
// tests/test_loss_math_host.py can compare it with torch autograd on the CPU (no GPU needed).
#include "loss_math.h"
extern "C" {
// in: n x (px py pw ph tx ty tw th); out: n x (ciou, d/dpx, d/dpy, d/dpw, d/dph)
void hc_ciou(const float* in, long n, float* out) {
  for (long i = 0; i < n; i++) {
    const float* b = in + i * 8;
    obb::CiouOut o = obb::ciou_fwd_bwd(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]);
    out[i * 5] = o.ciou;
    for (int k = 0; k < 4; k++) out[i * 5 + 1 + k] = o.d[k];
  }
}
// in: n x (x, t); out: n x (loss, dloss/dx)
void hc_focal(const float* in, long n, float pw, float gamma, float* out) {
  for (long i = 0; i < n; i++) {
    out[i * 2] = obb::bce_focal(in[i * 2], in[i * 2 + 1], pw, gamma);
    out[i * 2 + 1] = obb::bce_focal_grad(in[i * 2], in[i * 2 + 1], pw, gamma);
  }
}
void hc_bce(const float* in, long n, float pw, float* out) {
  for (long i = 0; i < n; i++) {
    out[i * 2] = obb::bce_logits(in[i * 2], in[i * 2 + 1], pw);
    out[i * 2 + 1] = obb::bce_logits_grad(in[i * 2], in[i * 2 + 1], pw);
  }
}
// in: n x (l0 l1 l2 l3 aw ah); out: n x (x y w h dx dy dw dh)
void hc_pred(const float* in, long n, float* out) {
  for (long i = 0; i < n; i++) {
    const float* b = in + i * 6;
    obb::PredBox p = obb::loss_pred_box(b[0], b[1], b[2], b[3], b[4], b[5]);
    float* o = out + i * 8;
    o[0] = p.x; o[1] = p.y; o[2] = p.w; o[3] = p.h; o[4] = p.dx; o[5] = p.dy; o[6] = p.dw; o[7] = p.dh;
  }
}
void hc_rem1(const float* in, long n, float* out) { for (long i = 0; i < n; i++) out[i] = obb::remainder1_f(in[i]); }
}
