// Spatial index over the boxes of one NMS call (rotated boxes, single score-ordered list) -- the arithmetic shared by the
// build and the queries (nms_core.h: grid_build, nms_cross_grid) and by the CPU check (tests/native/host_check_grid.cpp).
//
// Why: the cross phase of the lazy chunked NMS (nms_core.h) tests every kept row of a chunk against every still-alive
// later position -- O(kept x alive) circle tests.  With a few hundred kept boxes (S-clustered, K = 300) that is cheap;
// with thousands (class offsets, K = 3000: the natural shape of BASELINE configs[3]) or tens of thousands (S-uniform) it
// is the whole run time.  Only pairs whose circumscribed circles touch can have IoU > 0, so a kept row only needs the
// boxes of the cells around it.  The index is built INSIDE the persistent kernel, once, the first time a step keeps
// enough rows to need it, from the boxes that are still alive behind that chunk (classify + count, scan, scatter: a
// counting sort by cell); every later cross phase queries it.
//
// Layout.  Level L (0..kGridLevels-1) holds the boxes whose inflated circumradius r is below R_L = 2^(e_base + L) (and
// not below R_(L-1)); its cells are squares of side S_L = 2 R_L, so a query for a row of radius r_i looks at the cells
// within r_i + R_L of its centre: 3 x 3 cells when the row is no larger than the level's boxes, more (in proportion to
// its own area) when it is larger.  e_base is chosen from the extent of the data (bounding box of the centres): the top
// level ends at about a quarter of the extent; larger boxes are not indexed at all ("brute" boxes, see below) -- their
// neighbourhood is most of the domain anyway.  A cell's table slot is a hash of (level, cx, cy) that is LINEAR in cx, so
// the cells of one cell row of a query are consecutive slots and their boxes one contiguous range of the cell-sorted
// array; colliding cells only add candidates that the circle test removes again.
//
// Exactness.  The index may only skip a pair that RotGeom::cheap_reject (geom.h) would reject.  That test has two
// parts: circles apart, AND the pair is well conditioned (both short sides >= 4.84e-5 * centre distance; otherwise the
// reference's own fp32 corner rounding can fabricate an overlap between boxes that are far apart, riou_device.h).
// A box is "brute" when the second part could fail for ANY partner inside the data's bounding box (short side^2 <
// 2.34e-9 * diagonal^2), when a coordinate or the radius is not finite, or when it is too large for the top level.
// Brute boxes are kept out of the index and go through the exhaustive path (every kept row x brute columns, brute kept
// rows x every column): the result is bit-identical to the exhaustive scan by construction.  When more than 1/16 of the
// boxes are brute the build stops after its counting pass and the call stays exhaustive.
#pragma once
#include <stdint.h>
#include "obb_device.h"

namespace obb {

constexpr int kGridLevels = 13;
constexpr uint32_t kGridHashY = 0x9E3779B1u;      // odd: cy -> pseudo-random table row
constexpr uint32_t kGridHashL = 0x85EBCA6Bu;
constexpr float kGridIllCond = 2.34e-9f;           // the constant of RotGeom::cheap_reject

struct GridMeta {
  int bb[4];               // ordered-int encodings of min x, min y, max x, max y over the finite boxes that take part
  int n_brute;             // boxes kept out of the index
  uint32_t level_mask;     // levels that hold at least one box
  int pad[10];
};

// float <-> int with the same ordering (finite values and infinities; NaN must be filtered by the caller)
OBB_HD int grid_f2o(float f) { const int i = __builtin_bit_cast(int, f); return i >= 0 ? i : (i ^ 0x7fffffff); }
OBB_HD float grid_o2f(int o) { return __builtin_bit_cast(float, o >= 0 ? o : (o ^ 0x7fffffff)); }

struct GridPlan {          // derived from GridMeta::bb by every thread that needs it (a handful of flops)
  float x0, y0, xr, yr;    // origin and extent of the centres
  float dmax2;             // upper bound of any squared centre distance
  int e_base;              // R_L = 2^(e_base + L)
  int ok;                  // 0: no usable extent (no finite box, zero or infinite extent)
  int fine;                // cells of side 2 R_L / 2^fine (0: the original 2 R_L)
};

OBB_HD GridPlan grid_plan(const int* bb) {
  GridPlan p;
  p.ok = 0; p.x0 = p.y0 = p.xr = p.yr = p.dmax2 = 0.f; p.e_base = 0; p.fine = 0;
  if (bb[0] > bb[2] || bb[1] > bb[3]) return p;                 // no box seen
  const float x0 = grid_o2f(bb[0]), y0 = grid_o2f(bb[1]), x1 = grid_o2f(bb[2]), y1 = grid_o2f(bb[3]);
  const float xr = x1 - x0, yr = y1 - y0;
  const float D = xr > yr ? xr : yr;
  if (!(D > 0.f) || !(D < 1e30f)) return p;
  p.x0 = x0; p.y0 = y0; p.xr = xr; p.yr = yr;
  p.dmax2 = (xr * xr + yr * yr) * 1.001f;
  int e; (void)frexpf(D * 0.25f, &e);                          // D/4 = m * 2^e, m in [0.5, 1): 2^e in (D/4, D/2]
  p.e_base = e - (kGridLevels - 1);
  p.ok = 1;
  return p;
}

// level of a box of inflated circumradius r (finite, >= 0): smallest L with r < 2^(e_base + L); >= kGridLevels: too large
OBB_HD int grid_level(const GridPlan& p, float r) {
  int e; (void)frexpf(r, &e);                                   // r = m * 2^e, m in [0.5, 1)  ->  r < 2^e   (r == 0: e = 0)
  if (!(r > 0.f)) return 0;
  const int L = e - p.e_base;
  return L < 0 ? 0 : L;
}
OBB_HD float grid_level_radius(const GridPlan& p, int L) { return ldexpf(1.0f, p.e_base + L); }
OBB_HD float grid_level_inv_cell(const GridPlan& p, int L) { return ldexpf(1.0f, -(p.e_base + L + 1 - p.fine)); }   // 2^fine / (2 R_L)
// cell coordinate along one axis (v0 = origin, vmax_cell = last cell of the level on that axis)
OBB_HD int grid_cell(float v, float v0, float inv_cell, int last) {
  const float f = floorf((v - v0) * inv_cell);
  int c = f > 0.f ? (f < (float)last ? (int)f : last) : 0;
  return c;
}
OBB_HD int grid_last_cell(float extent, float inv_cell) {
  const float f = floorf(extent * inv_cell);
  return f < 1e9f ? (int)f : 1000000000;
}
OBB_HD uint32_t grid_slot(int L, int cx, int cy, uint32_t mmask) {
  return ((uint32_t)cx + (uint32_t)cy * kGridHashY + (uint32_t)L * kGridHashL) & mmask;
}
// true: the box must stay out of the index (see "Exactness" above); q0 = {x, y, r, short side^2}
OBB_HD bool grid_is_brute(const GridPlan& p, float x, float y, float r, float ms2) {
  const bool finite = (x - x == 0.f) && (y - y == 0.f) && (r - r == 0.f);
  if (!finite) return true;
  if (!(ms2 >= kGridIllCond * p.dmax2)) return true;            // an ill-conditioned far pair is possible (false on NaN)
  return !(r < ldexpf(1.0f, p.e_base + kGridLevels - 1));       // too large for the top level
}
// half width of the query window of a row (x, y, r) at level L: r + R_L, plus the rounding of the fp32 differences
// that the circle test and the cell arithmetic form (coordinates up to |x| + |x0|)
OBB_HD float grid_query_halfwidth(const GridPlan& p, int L, float x, float y, float r) {
  const float mag = fabsf(x) + fabsf(y) + fabsf(p.x0) + fabsf(p.y0) + p.xr + p.yr;
  return (r + grid_level_radius(p, L)) * 1.0001f + mag * 4e-7f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Independent slabs (nms_core.h: slab_setup).  The callers of the reference keep classes apart by adding cls * 4096 to the
// box centres before ONE nms_rotated call (utils/general.py:849-851): the list then consists of groups of boxes that cannot
// overlap each other.  Groups are found on the x axis: every box marks the bins its (slightly widened) circle interval
// [x - r, x + r] touches in a bitmap of kSlabBins bins over the extent of the centres; a maximal run of marked bins is a
// slab.  Two boxes in different slabs have an unmarked bin between their centres, so their circles are apart by more than
// the margin RotGeom::cheap_reject needs -- the exhaustive scan would reject the pair in its hot loop, provided both boxes
// are well conditioned for ANY partner inside the data's bounding box (the first half of grid_is_brute).  One box that is
// not (or is not finite) switches the decomposition off for the call.  With the gate below a box touches at most
// kSlabBins / 32 + 2 bins.
constexpr int kSlabBins = 1024;
constexpr int kSlabWords = kSlabBins / 32;
constexpr int kSlabCopies = 4;                     // the prep kernel's blocks OR into copy (block & 3): a quarter of the same-address traffic
constexpr int kMaxSlabs = 64;
constexpr int kBbInts = 8;                         // per-block partial of the key kernel: min x, min y, max x, max y (ordered ints), max w^2+h^2 (float bits), 3 spare
// The decomposition is only looked for when the data is much wider than its largest box (xr > 64 half-diagonals): a single
// image's detections never qualify and pay nothing; two classes 4096 px apart do unless their boxes are huge.
OBB_HD bool slab_gate(const GridPlan& p, float max_w2h2) { return p.ok && (p.xr * p.xr > 1024.f * max_w2h2); }
constexpr int kSlabMaxSeg = 16384;                 // a slab larger than this: the call stays one list (index path)

OBB_HD float slab_inv_bin(const GridPlan& p) { return (p.ok && p.xr > 0.f) ? (float)kSlabBins / p.xr : 0.f; }
OBB_HD int slab_bin(float v, float x0, float inv) {            // monotone in v; NaN -> 0 (such boxes raise the flag anyway)
  const float f = floorf((v - x0) * inv);
  return f > 0.f ? (f < (float)(kSlabBins - 1) ? (int)f : kSlabBins - 1) : 0;
}
// half width of the interval a box marks: its inflated circumradius plus the rounding of the fp32 differences involved
OBB_HD float slab_halfwidth(const GridPlan& p, float x, float r) {
  return r * 1.0001f + (fabsf(x) + fabsf(p.x0) + p.xr) * 4e-7f;
}
// false: this box forbids the decomposition (not finite, or an ill-conditioned far pair is possible: kGridIllCond)
OBB_HD bool slab_box_ok(const GridPlan& p, float x, float y, float r, float ms2) {
  const bool finite = (x - x == 0.f) && (y - y == 0.f) && (r - r == 0.f);
  return finite && (ms2 >= kGridIllCond * p.dmax2);
}

}  // namespace obb
