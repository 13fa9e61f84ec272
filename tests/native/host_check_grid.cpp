// CPU check of the spatial index of the NMS cross phase (yolov5_obb_amd/csrc/grid.h), compiled with g++.
//
// Builds the index exactly as the kernels do (k_prep_rot's bounding box, k_grid_count's brute rule / level / slot,
// counting sort) and runs nms_cross_grid's query arithmetic for sampled rows.  Invariant: for every pair (row i,
// column j) of non-brute boxes that RotGeom::cheap_reject (geom.h) does NOT reject, j is among the candidates the query
// visits.  (Pairs with a brute box never rely on the index: the kernel sends them through the exhaustive path.)
// usage: host_check_grid <n> <seed>      prints  "violations=<k>" and statistics; exit code 0 iff k == 0
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "grid.h"

using namespace obb;

struct Box { float x, y, r, ms2; };

static uint64_t g_rng = 88172645463325252ull;
static double urand() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (double)(g_rng >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-12, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

static Box make_box(float x, float y, float w, float h) {
  Box b; b.x = x; b.y = y;
  b.r = sqrtf(w * w + h * h) * 0.5005f;                    // rbox_make_feat (riou_device.h)
  const float ms = fminf(fabsf(w), fabsf(h)); b.ms2 = ms * ms;   // RotGeom::pack (geom.h)
  return b;
}

// RotGeom::cheap_reject (geom.h), same fp32 operations
static bool cheap_reject(const Box& a, const Box& b) {
  const float dx = b.x - a.x, dy = b.y - a.y;
  const float rs = a.r + b.r;
  const float d2 = dx * dx + dy * dy;
  return (d2 > rs * rs) && (fminf(a.ms2, b.ms2) >= 2.34e-9f * d2);
}

struct Index {
  GridPlan gp; uint32_t mask; uint32_t level_mask;
  std::vector<int> start; std::vector<int> sorted; std::vector<uint8_t> brute;
};

static int g_fine = 0;    // GridPlan::fine (argv[3]): cells of side 2 R_L / 2^fine
static Index build(const std::vector<Box>& bx, uint32_t M) {
  Index ix; ix.mask = M - 1; ix.level_mask = 0;
  int bb[4] = {0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000};
  for (const Box& b : bx)
    if ((b.x - b.x == 0.f) && (b.y - b.y == 0.f)) {
      bb[0] = std::min(bb[0], grid_f2o(b.x)); bb[1] = std::min(bb[1], grid_f2o(b.y));
      bb[2] = std::max(bb[2], grid_f2o(b.x)); bb[3] = std::max(bb[3], grid_f2o(b.y));
    }
  ix.gp = grid_plan(bb);
  ix.gp.fine = g_fine;
  const int n = (int)bx.size();
  ix.brute.assign(n, 0);
  std::vector<uint32_t> slot(n, 0xffffffffu);
  std::vector<int> cnt(M + 1, 0);
  for (int p = 0; p < n; p++) {
    const Box& b = bx[p];
    if (!ix.gp.ok || grid_is_brute(ix.gp, b.x, b.y, b.r, b.ms2)) { ix.brute[p] = 1; continue; }
    const int lv = grid_level(ix.gp, b.r);
    const float inv = grid_level_inv_cell(ix.gp, lv);
    const int cx = grid_cell(b.x, ix.gp.x0, inv, grid_last_cell(ix.gp.xr, inv));
    const int cy = grid_cell(b.y, ix.gp.y0, inv, grid_last_cell(ix.gp.yr, inv));
    slot[p] = grid_slot(lv, cx, cy, ix.mask);
    ix.level_mask |= 1u << lv;
    if (lv < 0 || lv >= kGridLevels) { printf("bad level %d\n", lv); exit(2); }
    cnt[slot[p]]++;
  }
  ix.start.assign(M + 1, 0);
  for (uint32_t i = 0; i < M; i++) ix.start[i + 1] = ix.start[i] + cnt[i];
  ix.sorted.assign(ix.start[M], -1);
  std::vector<int> fill(ix.start.begin(), ix.start.end() - 1);
  for (int p = 0; p < n; p++) if (slot[p] != 0xffffffffu) ix.sorted[fill[slot[p]]++] = p;
  return ix;
}

// the query of nms_cross_grid: marks every visited candidate, returns the number of entries scanned
static long long query(const Index& ix, const std::vector<Box>& bx, int i, std::vector<int>& mark, int stamp) {
  const Box& rq = bx[i];
  const int M = (int)ix.mask + 1;
  long long scanned = 0;
  auto scan = [&](int s, int e) { for (int k = s; k < e; k++) { mark[ix.sorted[k]] = stamp; scanned++; } };
  // (same order of decisions as nms_cross_grid: windows with more cells than half the table -> one scan of everything)
  bool whole = false;
  for (uint32_t lm = ix.level_mask; lm; lm &= lm - 1) {
    const int lv = __builtin_ctz(lm);
    const float nc = floorf(2.f * grid_query_halfwidth(ix.gp, lv, rq.x, rq.y, rq.r) * grid_level_inv_cell(ix.gp, lv)) + 2.f;
    if (!(nc * nc < 0.5f * (float)M)) whole = true;
  }
  if (whole) { scan(0, ix.start[M]); return scanned; }
  for (uint32_t lm = ix.level_mask; lm; lm &= lm - 1) {
    const int lv = __builtin_ctz(lm);
    const float inv = grid_level_inv_cell(ix.gp, lv);
    const float d = grid_query_halfwidth(ix.gp, lv, rq.x, rq.y, rq.r);
    const int lastx = grid_last_cell(ix.gp.xr, inv), lasty = grid_last_cell(ix.gp.yr, inv);
    const int cx0 = grid_cell(rq.x - d, ix.gp.x0, inv, lastx), cx1 = grid_cell(rq.x + d, ix.gp.x0, inv, lastx);
    const int cy0 = grid_cell(rq.y - d, ix.gp.y0, inv, lasty), cy1 = grid_cell(rq.y + d, ix.gp.y0, inv, lasty);
    const int len = cx1 - cx0 + 1;
    for (int cy = cy0; cy <= cy1; cy++) {
      const uint32_t i0 = grid_slot(lv, cx0, cy, ix.mask);
      const int e1 = (int)i0 + len;
      scan(ix.start[i0], ix.start[e1 <= M ? e1 : M]);
      if (e1 > M) scan(0, ix.start[e1 - M]);
    }
  }
  return scanned;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
  g_fine = argc > 3 ? atoi(argv[3]) : 0;
  g_rng ^= (uint64_t)(argc > 2 ? atoll(argv[2]) : 1) * 0x9E3779B97F4A7C15ull;
  long long violations = 0, checked = 0;
  const char* names[] = {"uniform", "clustered", "clustered+18cls", "uniform+18cls", "unit-square", "mixed-sizes", "outliers+degenerate", "huge-coords"};
  for (int dist = 0; dist < 8; dist++) {
    std::vector<Box> bx; bx.reserve(n);
    std::vector<float> ccx(300), ccy(300), cw(300), ch(300);
    for (int k = 0; k < 300; k++) { ccx[k] = (float)(urand() * 1024); ccy[k] = (float)(urand() * 1024); cw[k] = (float)(urand() * 60 + 8); ch[k] = (float)(urand() * 60 + 8); }
    for (int p = 0; p < n; p++) {
      float x, y, w, h;
      const int cls = (int)(urand() * 18);
      switch (dist) {
        case 0: x = (float)(urand() * 1024); y = (float)(urand() * 1024); w = (float)(urand() * 60 + 4); h = (float)(urand() * 60 + 4); break;
        case 1: case 2: {
          const int k = (int)(urand() * 300);
          x = ccx[k] + (float)nrand() * 2; y = ccy[k] + (float)nrand() * 2;
          w = cw[k] * (float)fmin(1.5, fmax(0.5, 1 + 0.1 * nrand())); h = ch[k] * (float)fmin(1.5, fmax(0.5, 1 + 0.1 * nrand()));
          if (dist == 2) { x += cls * 4096.f; y += cls * 4096.f; }
          break; }
        case 3: x = (float)(urand() * 1024) + cls * 4096.f; y = (float)(urand() * 1024) + cls * 4096.f; w = (float)(urand() * 60 + 4); h = (float)(urand() * 60 + 4); break;
        case 4: x = (float)urand(); y = (float)urand(); w = (float)(urand() * 0.06 + 0.004); h = (float)(urand() * 0.06 + 0.004); break;
        case 5: { const double sc = pow(2.0, urand() * 11 - 2); x = (float)(urand() * 4096); y = (float)(urand() * 4096); w = (float)(sc * (0.5 + urand())); h = (float)(sc * (0.5 + urand())); break; }
        case 6: {
          x = (float)(urand() * 1024); y = (float)(urand() * 1024); w = (float)(urand() * 60 + 4); h = (float)(urand() * 60 + 4);
          const double u = urand();
          if (u < 0.002) { x = (float)(urand() * 1e6); }                       // far outliers
          else if (u < 0.004) { w = (float)(urand() * 1e-3); }                 // needle boxes (ill conditioned: brute)
          else if (u < 0.005) { w = 5000.f; h = 3000.f; }                      // larger than the data
          else if (u < 0.006) { x = NAN; }
          else if (u < 0.007) { y = INFINITY; }
          else if (u < 0.008) { w = INFINITY; }
          else if (u < 0.009) { w = 0.f; h = 0.f; }
          break; }
        default: x = 3.0e6f + (float)(urand() * 2048); y = -7.0e6f + (float)(urand() * 2048); w = (float)(urand() * 60 + 4); h = (float)(urand() * 60 + 4); break;
      }
      bx.push_back(make_box(x, y, w, h));
    }
    const uint32_t M = n >= 65536 ? 65536u : 16384u;
    const Index ix = build(bx, M);
    long long nbrute = 0; for (int p = 0; p < n; p++) nbrute += ix.brute[p];
    std::vector<int> mark(n, -1);
    const int nq = 400;
    long long scanned = 0, must = 0, v0 = violations;
    for (int q = 0; q < nq; q++) {
      const int i = (int)(urand() * n);
      if (ix.brute[i]) continue;
      scanned += query(ix, bx, i, mark, q);
      for (int j = 0; j < n; j++) {
        if (j == i || ix.brute[j]) continue;
        if (!cheap_reject(bx[i], bx[j])) { must++; if (mark[j] != q) violations++; }
        checked++;
      }
    }
    printf("%-20s n=%d ok=%d e_base=%d levels=0x%x brute=%lld scanned/query=%.1f must-test/query=%.1f violations=%lld\n", names[dist], n, ix.gp.ok,
           ix.gp.e_base, ix.level_mask, nbrute, (double)scanned / nq, (double)must / nq, violations - v0);
  }
  printf("pairs checked=%lld\nviolations=%lld\n", checked, violations);
  return violations ? 1 : 0;
}
