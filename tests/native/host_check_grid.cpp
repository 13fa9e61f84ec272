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

// RotGeom::cheap_reject as nms_cross_blocks applies it to a column of the cell order: the column's short side replaced by +inf
static bool cheap_reject_indexed_col(const Box& row, const Box& col) {
  Box c = col; c.ms2 = INFINITY;
  return cheap_reject(row, c);
}

struct Plan { GridPlan gp; };

static GridPlan plan_of(const std::vector<Box>& bx) {
  int bb[4] = {0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000};
  for (const Box& b : bx)
    if ((b.x - b.x == 0.f) && (b.y - b.y == 0.f)) {
      bb[0] = std::min(bb[0], grid_f2o(b.x)); bb[1] = std::min(bb[1], grid_f2o(b.y));
      bb[2] = std::max(bb[2], grid_f2o(b.x)); bb[3] = std::max(bb[3], grid_f2o(b.y));
    }
  return grid_plan(bb);
}

static uint32_t slot_of(const GridPlan& gp, const Box& b, uint32_t mask, int* level = nullptr) {
  const int lv = grid_level(gp, b.r);
  if (lv < 0 || lv >= kGridLevels) { printf("bad level %d\n", lv); exit(2); }
  const float inv = grid_level_inv_cell(gp, lv);
  const int cx = grid_cell(b.x, gp.x0, inv, grid_last_cell(gp.xr, inv));
  const int cy = grid_cell(b.y, gp.y0, inv, grid_last_cell(gp.yr, inv));
  if (level) *level = lv;
  return grid_slot(lv, cx, cy, mask);
}

// the columns in cell order (k_grid_count / k_grid_scan / k_grid_scatter): blocks of <= 64 boxes of one slot
static std::vector<std::vector<int>> column_blocks(const GridPlan& gp, const std::vector<Box>& bx, uint32_t M, std::vector<uint8_t>& brute) {
  const int n = (int)bx.size();
  brute.assign(n, 0);
  std::vector<std::vector<int>> per_slot(M);
  for (int p = 0; p < n; p++) {
    if (!gp.ok || grid_is_brute(gp, bx[p].x, bx[p].y, bx[p].r, bx[p].ms2)) { brute[p] = 1; continue; }
    per_slot[slot_of(gp, bx[p], M - 1)].push_back(p);
  }
  std::vector<std::vector<int>> blocks;
  for (auto& v : per_slot)
    for (size_t k = 0; k < v.size(); k += 64) blocks.emplace_back(v.begin() + k, v.begin() + std::min(v.size(), k + 64));
  return blocks;
}

// the rows' grid of one slab (nms_cross_blocks, build part)
constexpr int kSlabRows = 1216, kSlabSlots = 2048;
struct Slab { std::vector<int> ent; std::vector<int> start; uint32_t level_mask; int n_idx, ns; };
static Slab build_slab(const GridPlan& gp, const std::vector<Box>& bx, const std::vector<int>& rows) {
  Slab S; S.level_mask = 0; S.ns = (int)rows.size();
  std::vector<int> slot(rows.size());
  std::vector<int> cnt(kSlabSlots + 2, 0);
  for (size_t k = 0; k < rows.size(); k++) {
    const Box& b = bx[rows[k]];
    if (grid_is_brute(gp, b.x, b.y, b.r, b.ms2)) slot[k] = kSlabSlots;
    else { int lv; slot[k] = (int)slot_of(gp, b, kSlabSlots - 1, &lv); S.level_mask |= 1u << lv; }
    cnt[slot[k]]++;
  }
  S.start.assign(kSlabSlots + 2, 0);
  for (int i = 0; i <= kSlabSlots; i++) S.start[i + 1] = S.start[i] + cnt[i];
  S.ent.assign(rows.size(), -1);
  std::vector<int> fill(S.start.begin(), S.start.end());
  for (size_t k = 0; k < rows.size(); k++) S.ent[fill[slot[k]]++] = rows[k];
  S.n_idx = S.start[kSlabSlots];
  return S;
}

// the rows a block enumerates (nms_cross_blocks, range part): marks them, returns how many entries were enumerated
static long long block_rows(const GridPlan& gp, const std::vector<Box>& bx, const std::vector<int>& block, const Slab& S, std::vector<int>& mark,
                            int stamp) {
  float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, rmax = 0.f;
  for (int p : block) {
    bx0 = fminf(bx0, bx[p].x); bx1 = fmaxf(bx1, bx[p].x); by0 = fminf(by0, bx[p].y); by1 = fmaxf(by1, bx[p].y); rmax = fmaxf(rmax, bx[p].r);
  }
  long long scanned = 0;
  auto enumerate = [&](int s, int t) { for (int e = s; e < t; e++) { mark[S.ent[e]] = stamp; scanned++; } };
  bool whole = false;
  struct Comb { int lv, cx0, cy, len; };
  std::vector<Comb> comb;
  const float mag = fmaxf(fabsf(bx0), fabsf(bx1)) + fmaxf(fabsf(by0), fabsf(by1));
  for (uint32_t lm = S.level_mask; lm; lm &= lm - 1) {
    const int lv = __builtin_ctz(lm);
    const float inv = grid_level_inv_cell(gp, lv);
    const float d = grid_query_halfwidth_mag(gp, lv, mag, rmax);
    const int lastx = grid_last_cell(gp.xr, inv), lasty = grid_last_cell(gp.yr, inv);
    const int cx0 = grid_cell(bx0 - d, gp.x0, inv, lastx), cy0 = grid_cell(by0 - d, gp.y0, inv, lasty);
    const int len = grid_cell(bx1 + d, gp.x0, inv, lastx) - cx0 + 1;
    const int nyr = grid_cell(by1 + d, gp.y0, inv, lasty) - cy0 + 1;
    if ((long long)len * nyr >= kSlabSlots / 2) whole = true;
    for (int dy = 0; dy < nyr && comb.size() <= 64; dy++) comb.push_back({lv, cx0, cy0 + dy, len});
  }
  if (comb.size() > 64) whole = true;
  if (whole) { enumerate(0, S.ns); return scanned; }
  for (const Comb& c : comb) {
    const uint32_t i0 = grid_slot(c.lv, c.cx0, c.cy, kSlabSlots - 1);
    const int e1 = (int)i0 + c.len;
    enumerate(S.start[i0], S.start[e1 <= kSlabSlots ? e1 : kSlabSlots]);
    if (e1 > kSlabSlots) enumerate(0, S.start[e1 - kSlabSlots]);
  }
  enumerate(S.n_idx, S.ns);                        // the brute rows
  return scanned;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
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
    const uint32_t M = n >= 32768 ? 16384u : 4096u;
    const GridPlan gp = plan_of(bx);
    std::vector<uint8_t> brute;
    const std::vector<std::vector<int>> blocks = column_blocks(gp, bx, M, brute);
    long long nbrute = 0; for (int p = 0; p < n; p++) nbrute += brute[p];
    long long scanned = 0, must = 0, v0 = violations, lanes = 0, nblk = 0;
    std::vector<int> mark(n, -1);
    int stamp = 0;
    for (int rep = 0; rep < 3 && gp.ok && !blocks.empty(); rep++) {
      // a slab of "kept rows": a random subset (dense and sparse ones), brute boxes included
      const int ns = rep == 0 ? kSlabRows : (rep == 1 ? 300 : 97);
      std::vector<int> rows;
      for (int k = 0; k < ns; k++) rows.push_back((int)(urand() * n));
      std::sort(rows.begin(), rows.end()); rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
      const Slab S = build_slab(gp, bx, rows);
      for (int q = 0; q < 300; q++) {
        const std::vector<int>& blk = blocks[(size_t)(urand() * blocks.size())];
        scanned += block_rows(gp, bx, blk, S, mark, ++stamp);
        lanes += (long long)blk.size(); nblk++;
        for (int j : blk)
          for (int i : rows) {
            if (i == j) continue;
            const bool rej = cheap_reject(bx[i], bx[j]);
            if (rej != cheap_reject_indexed_col(bx[i], bx[j])) violations++;          // the +inf substitution changes nothing
            if (!rej) { must++; if (mark[i] != stamp) violations++; }
            checked++;
          }
      }
    }
    printf("%-20s n=%d ok=%d e_base=%d brute=%lld blocks=%zu lanes/block=%.1f rows enumerated/block=%.1f must-test/block=%.1f violations=%lld\n",
           names[dist], n, gp.ok, gp.e_base, nbrute, blocks.size(), nblk ? (double)lanes / nblk : 0.0, nblk ? (double)scanned / nblk : 0.0,
           nblk ? (double)must / nblk : 0.0, violations - v0);
  }
  printf("pairs checked=%lld\nviolations=%lld\n", checked, violations);
  return violations ? 1 : 0;
}
