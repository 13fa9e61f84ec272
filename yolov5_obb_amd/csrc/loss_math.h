// Scalar math of the OBB training loss -- plain C++ (host + device) so that tests/native can compile it with g++
// and compare it with torch autograd before any GPU time is spent.
//
// Behavioural contract (values within 1e-5 of the reference, which evaluates the same expressions in fp32):
//   BCEWithLogitsLoss(pos_weight)      utils/loss.py:98-100  (ATen: (1-t)*x - (1+(pw-1)*t)*logsigmoid(x))
//   bbox_iou(..., x1y1x2y2=False, CIoU=True)   utils/metrics.py:201-243 (alpha under no_grad :233-236)
//   pxy = sigmoid*2-0.5, pwh = (sigmoid*2)^2*anchor     utils/loss.py:148-149
// The gradients are written out by hand (the kernels do forward and backward in one pass each, there is no
// autograd tape on the device).
#pragma once
#include "obb_device.h"

namespace obb {

OBB_HD float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

OBB_HD float logsigmoid_f(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

// loss = (1-t)*x - lw*logsigmoid(x),  lw = 1 + (pw-1)*t
OBB_HD float bce_logits(float x, float t, float pw) {
  const float lw = 1.0f + (pw - 1.0f) * t;
  return (1.0f - t) * x - lw * logsigmoid_f(x);
}

// d loss / d x = (1-t) - lw*(1-sigmoid(x))
OBB_HD float bce_logits_grad(float x, float t, float pw) {
  const float lw = 1.0f + (pw - 1.0f) * t;
  return (1.0f - t) - lw * (1.0f - sigmoid_f(x));
}

// FocalLoss around BCEWithLogitsLoss (utils/loss.py:35-62, wrapped around BCEcls / BCEtheta / BCEobj when hyp['fl_gamma'] > 0,
// :107-110; alpha = 0.25, the constructor default):
//   loss = bce * af * (1 - p_t)^gamma,   p = sigmoid(x),  p_t = t p + (1-t)(1-p),  af = t alpha + (1-t)(1-alpha)
// gamma <= 0: the plain BCE (the reference does not wrap then).
constexpr float kFocalAlpha = 0.25f;
OBB_HD float bce_focal(float x, float t, float pw, float gamma) {
  const float b = bce_logits(x, t, pw);
  if (!(gamma > 0.0f)) return b;
  const float p = sigmoid_f(x);
  const float pt = t * p + (1.0f - t) * (1.0f - p);
  const float af = t * kFocalAlpha + (1.0f - t) * (1.0f - kFocalAlpha);
  return b * (af * powf(1.0f - pt, gamma));
}
// d loss / d x = af * (bce' * m + bce * m'),  m = (1 - p_t)^gamma,  m' = -gamma (1 - p_t)^(gamma-1) (2t - 1) p (1 - p)
OBB_HD float bce_focal_grad(float x, float t, float pw, float gamma) {
  const float db = bce_logits_grad(x, t, pw);
  if (!(gamma > 0.0f)) return db;
  const float b = bce_logits(x, t, pw);
  const float p = sigmoid_f(x);
  const float pt = t * p + (1.0f - t) * (1.0f - p);
  const float af = t * kFocalAlpha + (1.0f - t) * (1.0f - kFocalAlpha);
  const float u = 1.0f - pt;
  const float m = powf(u, gamma);
  const float dm = -gamma * powf(u, gamma - 1.0f) * ((2.0f * t - 1.0f) * p * (1.0f - p));
  return af * (db * m + b * dm);
}

struct CiouOut {
  float ciou;        // utils/metrics.py:237
  float d[4];        // d ciou / d (px, py, pw, ph) of box1 (the prediction), alpha held constant
};

// box1 = prediction (x, y, w, h), box2 = target (x, y, w, h); eps = 1e-7.
// minimum/maximum ties split the gradient in half and clamp(0) passes the gradient at equality, as ATen does.
OBB_HD CiouOut ciou_fwd_bwd(float px, float py, float pw, float ph, float tx, float ty, float tw, float th) {
  const float eps = 1e-7f;
  const float a1 = px - pw / 2, a2 = px + pw / 2, c1 = py - ph / 2, c2 = py + ph / 2;     // :207-208
  const float B1 = tx - tw / 2, B2 = tx + tw / 2, D1 = ty - th / 2, D2 = ty + th / 2;     // :209-210
  const float iw_raw = fminf(a2, B2) - fmaxf(a1, B1);
  const float ih_raw = fminf(c2, D2) - fmaxf(c1, D1);
  const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
  const float inter = iw * ih;                                                            // :213-214
  const float w1 = a2 - a1, h1 = c2 - c1 + eps;                                           // :217
  const float w2 = B2 - B1, h2 = D2 - D1 + eps;                                           // :218
  const float uni = w1 * h1 + w2 * h2 - inter + eps;                                      // :219
  const float iou = inter / uni;                                                          // :221
  const float cw = fmaxf(a2, B2) - fminf(a1, B1);                                         // :223
  const float ch = fmaxf(c2, D2) - fminf(c1, D1);                                         // :224
  const float c2v = cw * cw + ch * ch + eps;                                              // :226
  const float sx = B1 + B2 - a1 - a2, sy = D1 + D2 - c1 - c2;
  const float rho2 = (sx * sx + sy * sy) / 4;                                             // :227-228
  const float kpi = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
  const float dat = atanf(w2 / h2) - atanf(w1 / h1);
  const float v = kpi * (dat * dat);                                                      // :233
  const float alpha = v / (v - iou + (1 + eps));                                          // :235 (no grad)
  CiouOut o;
  o.ciou = iou - (rho2 / c2v + v * alpha);                                                // :237

  // partial derivatives with respect to the four edges of box1
  auto half_if_tie = [](float x, float y, bool pick_x_when) { return x == y ? 0.5f : (pick_x_when ? 1.0f : 0.0f); };
  const float g_iw = (iw_raw >= 0.f) ? 1.0f : 0.0f, g_ih = (ih_raw >= 0.f) ? 1.0f : 0.0f;
  // d iw / d a2 = [a2 is the min], d iw / d a1 = -[a1 is the max]
  const float diw_a2 = g_iw * half_if_tie(a2, B2, a2 < B2), diw_a1 = -g_iw * half_if_tie(a1, B1, a1 > B1);
  const float dih_c2 = g_ih * half_if_tie(c2, D2, c2 < D2), dih_c1 = -g_ih * half_if_tie(c1, D1, c1 > D1);
  const float dcw_a2 = half_if_tie(a2, B2, a2 > B2), dcw_a1 = -half_if_tie(a1, B1, a1 < B1);
  const float dch_c2 = half_if_tie(c2, D2, c2 > D2), dch_c1 = -half_if_tie(c1, D1, c1 < D1);
  const float den = w1 * w1 + h1 * h1;
  const float dv_w1 = -2.0f * kpi * dat * h1 / den, dv_h1 = 2.0f * kpi * dat * w1 / den;

  auto edge = [&](float dinter, float dw1, float dh1, float dcw, float dch, float drho2) {
    const float duni = dw1 * h1 + w1 * dh1 - dinter;
    const float diou = (dinter * uni - inter * duni) / (uni * uni);
    const float dc2v = 2.0f * cw * dcw + 2.0f * ch * dch;
    const float dpen = (drho2 * c2v - rho2 * dc2v) / (c2v * c2v);
    const float dv = dv_w1 * dw1 + dv_h1 * dh1;
    return diou - dpen - alpha * dv;
  };
  const float g_a1 = edge(diw_a1 * ih, -1.f, 0.f, dcw_a1, 0.f, -sx / 2);
  const float g_a2 = edge(diw_a2 * ih, 1.f, 0.f, dcw_a2, 0.f, -sx / 2);
  const float g_c1 = edge(iw * dih_c1, 0.f, -1.f, 0.f, dch_c1, -sy / 2);
  const float g_c2 = edge(iw * dih_c2, 0.f, 1.f, 0.f, dch_c2, -sy / 2);
  o.d[0] = g_a1 + g_a2;
  o.d[1] = g_c1 + g_c2;
  o.d[2] = (g_a2 - g_a1) * 0.5f;
  o.d[3] = (g_c2 - g_c1) * 0.5f;
  return o;
}

// Prediction decode of the loss (utils/loss.py:148-149) and its derivative with respect to the four logits.
struct PredBox {
  float x, y, w, h;       // pxy, pwh
  float dx, dy, dw, dh;   // d(pxy)/d(logit), d(pwh)/d(logit)
};
OBB_HD PredBox loss_pred_box(float l0, float l1, float l2, float l3, float aw, float ah) {
  const float s0 = sigmoid_f(l0), s1 = sigmoid_f(l1), s2 = sigmoid_f(l2), s3 = sigmoid_f(l3);
  PredBox p;
  p.x = s0 * 2 - 0.5f; p.y = s1 * 2 - 0.5f;
  const float t2 = s2 * 2, t3 = s3 * 2;
  p.w = t2 * t2 * aw; p.h = t3 * t3 * ah;
  p.dx = 2 * s0 * (1 - s0); p.dy = 2 * s1 * (1 - s1);
  p.dw = 2 * t2 * (2 * s2 * (1 - s2)) * aw; p.dh = 2 * t3 * (2 * s3 * (1 - s3)) * ah;
  return p;
}

// torch.remainder(x, 1) (utils/loss.py:247-248 `gxy % 1`): fmod, then shifted into [0, 1) for negative x
OBB_HD float remainder1_f(float x) {
  float m = fmodf(x, 1.0f);
  if (m != 0.f && m < 0.f) m += 1.0f;
  return m;
}

}  // namespace obb
