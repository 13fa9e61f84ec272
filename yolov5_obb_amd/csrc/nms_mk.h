// Rotated NMS of ONE long score-ordered list as a chain of ordinary launches ("phase kernels") -- round 6.
//
// Same algorithm as the persistent kernel of nms_core.h (lazy chunked greedy NMS: select a chunk of the next alive positions,
// find the conflict edges inside it, resolve them like the reference's sequential scan, let the chunk's KEPT rows remove what
// they suppress among the later positions; nms_rotated_cuda.cu:60,109-128), and the same decision stages (geom.h), but
//   * every phase is a kernel of its own with its own register / LDS budget (<= 128 VGPRs, no scratch, two 512-thread
//     workgroups per CU = 4 waves per SIMD) instead of one 250-register workgroup per CU that spins on grid barriers: nothing
//     here needs co-residency, a launch boundary is the barrier;
//   * both pair phases are QUERY-centric probes of a small spatial hash of the step's chunk: "pairs" = every chunk member against
//     the earlier members around it, "cross" = every still-alive later position against the chunk's KEPT rows around it.  The
//     table holds at most one chunk (<= kMkCapMax entries, rebuilt per step by one workgroup), so there is no index of all n
//     boxes to build, nothing dead or earlier to scan past, and class-offset layouts (utils/general.py:849-851) need no slab
//     decomposition: far groups simply never share a cell;
//   * a query is served by kMkL = 8 adjacent lanes that read adjacent 16-byte entries (one 128-byte line per group and trip).
//
// What the enqueued steps leave undone -- fewer steps than the data needs (their number is an estimate: nothing is read back), a
// pending list that overflowed, a chunk with so many ill-conditioned boxes that probing is the wrong tool -- is finished by the
// persistent kernel of nms_core.h, launched behind them on the same state (NmsResume); it returns at once when the call is complete.
//
// Exactness.  A pair may only be dropped without a decision when RotGeom::cheap_reject would reject it: circumscribed circles
// apart AND both boxes well conditioned for a partner at that distance.  A box that is not well conditioned for ANY partner
// inside the data's bounding box (short side^2 < 2.34e-9 * diagonal^2, grid.h: kGridIllCond), not finite, or far larger than
// its chunk's mean is a "brute" entry: it sits in front of the table and is handed to every query; a query with that property
// reads the whole table.  Everything that is not dropped goes through classify_quick -> classify_full -> hit_exact exactly
// like in nms_core.h.
#pragma once
#include <cstddef>
#include "nms_core.h"

namespace obb {

constexpr int kMkThreads = 512;
constexpr int kMkWaves = kMkThreads / 64;
constexpr int kMkSlots = 8192;            // table slots of a chunk's spatial hash: 128 table rows of 64 cells
constexpr int kMkL = 8;                   // lanes per query
constexpr int kMkQB = 64 / kMkL;          // queries of a wave in flight
constexpr int kMkNE = 4;                  // entries a lane requests per trip of the global-table probes
constexpr int kMkCapMax = 16384;          // largest chunk (chunk-local indices are 16-bit; the edge list holds the worst case of kMkTile members)
constexpr int kMkTile = 8192;             // members a table build holds in registers at a time
constexpr int kMkPer = kMkTile / kMkThreads;
constexpr int kMkPend1 = 1 << 21;         // pending-list capacity (pairs); an overflow hands the call to the persistent kernel
constexpr int kMkTabPad = kMkSlots / 32;  // LDS counters are padded one word per 32: a thread's 16 consecutive slots spread over the banks

struct MkGrid {                           // geometry of a step's table (written by one workgroup, read by the probe kernels of later launches)
  float x0, y0, inv, rmax, mag;           // origin, 1 / cell side, radius limit of an indexed entry, |x0| + |y0| + extent
  int cxl, cyl;                           // last cell per axis (clamp)
  int total, nbrute;                      // entries; the first nbrute are handed to every query
  int pad[7];
};
struct MkCtl {                            // control block of a call (zeroed before the first launch)
  int cur, kept, done, step;
  int cn, cap, nrow, chunk_first;
  float dmax2;                            // (extent of the centres)^2: a box is ill conditioned when its short side^2 < kGridIllCond * dmax2
  int stage;                              // 0: nothing selected yet, 1: a chunk is selected (pairs / resolve pending), 3: its rows are resolved (cross pending or done)
  int bail;                               // the phase kernels stand back (pending / edge list overflow, too many brute boxes): the persistent kernel behind them carries on
  int n1;                                 // pending pairs of the phase in flight: undecided after the quick tests
  int ticket;                             // workgroups of a decide kernel that are through (the last one runs the serial phase behind it)
  // the data as a whole (nms.hip: local_extras' per-workgroup partials, reduced by the first select): the bounding box of the finite
  // centres and the radius limit of every table -- rcap = min(largest, 4 x mean) inflated circumradius.  Fixed per call: a table
  // build has no statistics pass of its own.
  float dx0, dy0, dxr, dyr, drcap, dmag; int pad;
  MkGrid g;                               // the chunk's table (both pair phases)
};

static_assert(offsetof(MkCtl, cur) == offsetof(NmsResume, cur) && offsetof(MkCtl, kept) == offsetof(NmsResume, kept) &&
              offsetof(MkCtl, done) == offsetof(NmsResume, done) && offsetof(MkCtl, cap) == offsetof(NmsResume, cap) &&
              offsetof(MkCtl, nrow) == offsetof(NmsResume, nrow) && offsetof(MkCtl, chunk_first) == offsetof(NmsResume, chunk_first) &&
              offsetof(MkCtl, stage) == offsetof(NmsResume, stage) && offsetof(MkCtl, bail) == offsetof(NmsResume, bail),
              "the persistent kernel reads the head of the control block as NmsResume");

struct MkArgs {
  const float4* rec; const uint32_t* order; u64* alive; int n;
  MkCtl* ctl;
  uint32_t* cidx;                         // [capmax] positions of the chunk members, ascending
  float4* ent; uint16_t* start;           // the chunk's table: entries {x, y, radius (rounded up to 16 bits) | chunk-local index, position}, slot starts [kMkSlots + 1]
  u64* kbits;                             // [capmax / 64] bit j: chunk member j was kept (written by resolve, read by the cross probe)
  uint8_t* hasin;                         // [capmax] != 0: an edge points at chunk member j (cleared by select, set with every edge: resolve starts from
                                          // the members nobody points at instead of finding them in a first pass over the edge list)
  uint32_t* edges; int* nedges; long long ecap;
  uint32_t* rows; int* nrows; int* keep_cnt; int64_t* keep_out;
  int64_t* num_keep;                      // written by whoever completes the call
  const int* bbpart; int nparts;
  int capmax, cap_first;
  int skip_full;                          // decide kernels: 1 = every open pending pair goes straight to the exact clip (see mk_decide_phase)
  float thr;
  uint4* pend1; int cap1;                 // {query position, entry position, entry index << 16 | query index (pairs), -} the quick tests left undecided
  int* hint_host;                         // pinned words: [0] steps the call needed, [1] boxes it kept (read by the NEXT call of this thread; may be NULL)
  u64* prof;                              // optional [56]: wall-clock ticks (10 ns) of the serial phases (OBB_NMS_PHASE_PROF=1; development aid)
};

__device__ __forceinline__ bool mk_finite3(float x, float y, float r) { return (x - x == 0.f) && (y - y == 0.f) && (r - r == 0.f); }
// cell along one axis: monotone in v (fp subtraction, multiplication by a positive factor and floor are); NaN -> 0
__device__ __forceinline__ int mk_cell(float v, float v0, float inv, int last) {
  const float f = floorf((v - v0) * inv);
  return f > 0.f ? (f < (float)last ? (int)f : last) : 0;
}
// table slot of a cell: consecutive cells of one cell row stay consecutive inside a block of 64; the table row mixes the cell
// row with the block so that the diagonal blocks of a class-offset layout spread over the table
__device__ __forceinline__ int mk_trow(int cy, int xb) { return (cy + 37 * xb) & 127; }
__device__ __forceinline__ int mk_slot(int cx, int cy) { return (cx & 63) + 64 * mk_trow(cy, cx >> 6); }
// an entry's third word: the radius rounded UP to its 16 leading bits (never smaller: the circle test only gets more cautious)
// next to the member's chunk-local index
__device__ __forceinline__ uint32_t mk_rup(float r) { return (__float_as_uint(r) + 0xFFFFu) & 0xFFFF0000u; }   // r >= 0 or +inf
__device__ __forceinline__ int mk_pad(int slot) { return slot + (slot >> 5); }

// ---- workgroup reductions (kMkThreads threads; every thread returns the result)
__device__ __forceinline__ float mk_block_sum(float v, float* s_red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < kMkWaves; k++) t += s_red[k];
  return t;
}
__device__ __forceinline__ int mk_block_sumi(int v, int* s_red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int k = 0; k < kMkWaves; k++) t += s_red[k];
  return t;
}
__device__ __forceinline__ void mk_block_minmax(float& lo0, float& lo1, float& hi0, float& hi1, float& hi2, float (*s_red)[8]) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo0 = fminf(lo0, __shfl_xor(lo0, d)); lo1 = fminf(lo1, __shfl_xor(lo1, d));
    hi0 = fmaxf(hi0, __shfl_xor(hi0, d)); hi1 = fmaxf(hi1, __shfl_xor(hi1, d)); hi2 = fmaxf(hi2, __shfl_xor(hi2, d));
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { float* o = s_red[threadIdx.x >> 6]; o[0] = lo0; o[1] = lo1; o[2] = hi0; o[3] = hi1; o[4] = hi2; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kMkWaves; k++) {
    lo0 = fminf(lo0, s_red[k][0]); lo1 = fminf(lo1, s_red[k][1]);
    hi0 = fmaxf(hi0, s_red[k][2]); hi1 = fmaxf(hi1, s_red[k][3]); hi2 = fmaxf(hi2, s_red[k][4]);
  }
}

// development aid: thread 0 of a serial phase adds the ticks since *t0 to prof[slot] and restarts the clock
__device__ __forceinline__ void mk_lap(const MkArgs& a, int slot, u64* t0) {
  if (a.prof != nullptr && threadIdx.x == 0) { const u64 t = wall_clock64(); a.prof[32 + slot] += t - *t0; *t0 = t; }
}

struct MkData { float x0, y0, xr, yr, rcap, mag, dmax2; };   // the data as a whole (mk_data_stats)

struct MkLdsBuild {
  int tab[kMkSlots + kMkTabPad + 40];     // padded counters (mk_pad); the brute block's counter sits behind them
  float red[kMkWaves][8];
  int redi[16];
};
constexpr int kMkBruteCtr = kMkSlots + kMkTabPad + 8;

// ------------------------------------------------------------------ the table of a chunk (ONE workgroup of kMkThreads)
// members k = 0 .. cnt-1 (cnt <= kMkCapMax) at positions list[k] (LDS, ascending).
// Quad 0 of every member is fetched ONCE (up to 2 x kMkPer independent loads per thread) and stays in registers through the two passes
// of the counting sort (counts, scatter): the passes cost LDS time, not a chain of memory round trips.
// A box above the data's radius limit (MkData::rcap) would widen every query's window; it becomes a "brute" entry like the boxes
// that are not finite or not well conditioned for any partner inside the data's bounding box.
__device__ void mk_build_table(const MkArgs& a, const float4* __restrict__ rec, const uint32_t* s_list, int cnt, const MkData& D, MkGrid* g_out,
                               float4* __restrict__ ent, uint16_t* __restrict__ start_g, MkLdsBuild& S) {
  const int tid = threadIdx.x;
  u64 tb = a.prof ? wall_clock64() : 0ull;
  // A chunk has at most two tiles of kMkTile members.  BOTH stay in registers from their one fetch to the scatter -- x, y, radius,
  // position and slot of 2 x kMkPer members per thread (round 6, second half: a chunk above kMkTile members went through the two
  // passes tile by tile and fetched its quads once per pass: four dependent rounds of gathers instead of one, 24 us of the 27 us
  // a 16,384-member table took one workgroup).
  const bool two = cnt > kMkTile;                        // (workgroup-uniform)
  // the geometry: the data's bounding box and radius limit (fixed per call, MkData) and a cell side chosen by the density of THIS
  // chunk -- 3/4 of the radius limit when a query meets many entries (their number decides the work), twice the limit when it
  // meets a handful (then the number of cell rows a query walks decides it)
  const float rcap = D.rcap, dmax2 = D.dmax2;
  MkGrid g;
  g.total = cnt; g.rmax = __uint_as_float(mk_rup(rcap));
  {
    const float area = fmaxf(D.xr, rcap) * fmaxf(D.yr, rcap);
    const float expect = (float)cnt * (25.f * rcap * rcap) / area;              // entries in a (5 rcap)^2 window
    float side = rcap * (expect < 16.f ? 2.0f : 0.75f);
    // (a cell side far below the extent / 2^20 buys nothing and would overflow the int cell arithmetic)
    const float ext = fmaxf(D.xr, D.yr);
    if (!(side > ext * 1e-6f)) side = ext * 1e-6f;
    if (!(side > 1e-30f)) side = 1.0f;
    g.x0 = D.x0; g.y0 = D.y0; g.inv = 1.0f / side; g.mag = D.mag;
    const float fx = floorf(D.xr * g.inv), fy = floorf(D.yr * g.inv);
    g.cxl = (fx < 1e9f) ? (int)fx : 1000000000; g.cyl = (fy < 1e9f) ? (int)fy : 1000000000;
    if (!(fx >= 0.f)) g.cxl = 0;
    if (!(fy >= 0.f)) g.cyl = 0;
  }
  for (int k = 0; k < 7; k++) g.pad[k] = 0;
  auto slot_of = [&](const float4& v) -> int {
    const bool brute = !mk_finite3(v.x, v.y, v.z) || !(v.w >= kGridIllCond * dmax2) || (v.z > rcap);
    return brute ? kMkBruteCtr : mk_pad(mk_slot(mk_cell(v.x, g.x0, g.inv, g.cxl), mk_cell(v.y, g.y0, g.inv, g.cyl)));
  };
  float ex[2][kMkPer], ey[2][kMkPer], ez[2][kMkPer];
  uint32_t pos[2][kMkPer];
  int slot[2][kMkPer];
  // one fetch: the positions (LDS), then every quad (2 x kMkPer independent loads per thread), the slot computed as the quad arrives
  auto fetch = [&](int t) {
    float4 q[kMkPer];
#pragma unroll
    for (int u = 0; u < kMkPer; u++) { const int k = t * kMkTile + tid + u * kMkThreads; pos[t][u] = k < cnt ? s_list[k] : 0u; }
#pragma unroll
    for (int u = 0; u < kMkPer; u++) {
      const int k = t * kMkTile + tid + u * kMkThreads;
      q[u] = k < cnt ? rec[(size_t)pos[t][u] * 4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kMkPer; u++) {
      const int k = t * kMkTile + tid + u * kMkThreads;
      ex[t][u] = q[u].x; ey[t][u] = q[u].y; ez[t][u] = q[u].z;
      slot[t][u] = k < cnt ? slot_of(q[u]) : -1;
    }
  };
  for (int s2 = tid; s2 < kMkSlots + kMkTabPad + 40; s2 += kMkThreads) S.tab[s2] = 0;
  fetch(0);
  if (two) fetch(1);
  __syncthreads();
  if (a.prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mk_lap(a, 12, &tb); }
  // pass 1: counts per slot
#pragma unroll
  for (int u = 0; u < kMkPer; u++) if (slot[0][u] >= 0) atomicAdd(&S.tab[slot[0][u]], 1);
  if (two) {
#pragma unroll
    for (int u = 0; u < kMkPer; u++) if (slot[1][u] >= 0) atomicAdd(&S.tab[slot[1][u]], 1);
  }
  __syncthreads();
  mk_lap(a, 14, &tb);
  const int nbrute = S.tab[kMkBruteCtr];
  g.nbrute = nbrute;
  // exclusive scan of the kMkSlots counters: 16 consecutive slots per thread, then the thread totals
  constexpr int PER = kMkSlots / kMkThreads;
  int loc[PER]; int tsum = 0;
#pragma unroll
  for (int u = 0; u < PER; u++) { loc[u] = tsum; tsum += S.tab[mk_pad(tid * PER + u)]; }
  int incl = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if ((tid & 63) >= d) incl += v; }
  if ((tid & 63) == 63) S.redi[tid >> 6] = incl;
  __syncthreads();
  int wpre = 0;
#pragma unroll
  for (int k = 0; k < kMkWaves; k++) if (k < (tid >> 6)) wpre += S.redi[k];
  const int tbase = wpre + incl - tsum;
  {
    // the slot starts go out as 16-byte stores (16 consecutive uint16 per thread)
    uint32_t w[PER / 2];
#pragma unroll
    for (int u = 0; u < PER; u += 2) {
      const int s0 = tbase + loc[u], s1 = tbase + loc[u + 1];
      S.tab[mk_pad(tid * PER + u)] = nbrute + s0;        // become the running fill pointers of the slots (entry index)
      S.tab[mk_pad(tid * PER + u + 1)] = nbrute + s1;
      w[u / 2] = (uint32_t)s0 | ((uint32_t)s1 << 16);    // (cnt <= kMkCapMax < 65536)
    }
    uint4* dst = reinterpret_cast<uint4*>(start_g + tid * PER);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
  if (tid == 0) { start_g[kMkSlots] = (uint16_t)(cnt - nbrute); S.tab[kMkBruteCtr] = 0; }   // (the brute block fills from entry 0)
  __syncthreads();
  mk_lap(a, 15, &tb);
  // pass 2: scatter
  auto scatter = [&](int t) {
#pragma unroll
    for (int u = 0; u < kMkPer; u++) {
      if (slot[t][u] >= 0) {
        const int k = t * kMkTile + tid + u * kMkThreads;
        const int e = atomicAdd(&S.tab[slot[t][u]], 1);
        ent[e] = make_float4(ex[t][u], ey[t][u], __uint_as_float(mk_rup(ez[t][u]) | (uint32_t)k), __uint_as_float(pos[t][u]));
      }
    }
  };
  scatter(0);
  if (two) scatter(1);
  if (tid == 0) *g_out = g;
  if (a.prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __syncthreads();
  mk_lap(a, 16, &tb);
}
static_assert(kMkCapMax == 2 * kMkTile, "mk_build_table holds a chunk as two tiles");
static_assert(kMkSlots / kMkThreads == 16, "the slot starts are written as two 16-byte stores per thread");

// The data as a whole from the key kernel's per-workgroup partials (nms.hip: local_extras): bounding box of the finite centres,
// (xr^2 + yr^2) * 1.001 (+inf when there is no usable extent: then every box is brute and the call goes to the persistent kernel),
// and the radius limit min(largest, 4 x mean) of the inflated circumradius 0.5005 * sqrt(w^2 + h^2) (riou_device.h).
__device__ __forceinline__ MkData mk_data_stats(const int* __restrict__ bbpart, int nparts, int* s_redi) {
  int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = (int)0x80000000, by1 = (int)0x80000000, d2 = 0, rc = 0;
  float rs = 0.f;
  for (int i = threadIdx.x; i < nparts; i += kMkThreads) {
    const int4 q = reinterpret_cast<const int4*>(bbpart)[2 * i], q2 = reinterpret_cast<const int4*>(bbpart)[2 * i + 1];
    bx0 = min(bx0, q.x); by0 = min(by0, q.y); bx1 = max(bx1, q.z); by1 = max(by1, q.w);
    d2 = max(d2, q2.x); rs += __int_as_float(q2.y); rc += q2.z;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    bx0 = min(bx0, __shfl_xor(bx0, d)); by0 = min(by0, __shfl_xor(by0, d));
    bx1 = max(bx1, __shfl_xor(bx1, d)); by1 = max(by1, __shfl_xor(by1, d));
    d2 = max(d2, __shfl_xor(d2, d)); rs += __shfl_xor(rs, d); rc += __shfl_xor(rc, d);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    int* o = s_redi + (threadIdx.x >> 6) * 8;
    o[0] = bx0; o[1] = by0; o[2] = bx1; o[3] = by1; o[4] = d2; o[5] = __float_as_int(rs); o[6] = rc;
  }
  __syncthreads();
  bx0 = s_redi[0]; by0 = s_redi[1]; bx1 = s_redi[2]; by1 = s_redi[3]; d2 = s_redi[4]; rs = __int_as_float(s_redi[5]); rc = s_redi[6];
  for (int k = 1; k < kMkWaves; k++) {
    const int* o = s_redi + 8 * k;
    bx0 = min(bx0, o[0]); by0 = min(by0, o[1]); bx1 = max(bx1, o[2]); by1 = max(by1, o[3]); d2 = max(d2, o[4]); rs += __int_as_float(o[5]); rc += o[6];
  }
  __syncthreads();
  MkData D;
  D.x0 = D.y0 = D.xr = D.yr = D.mag = 0.f; D.dmax2 = 0.f; D.rcap = 0.f;
  if (bx0 > bx1 || by0 > by1) return D;                           // no finite centre at all: every box is brute anyway
  D.x0 = grid_o2f(bx0); D.y0 = grid_o2f(by0);
  D.xr = grid_o2f(bx1) - D.x0; D.yr = grid_o2f(by1) - D.y0;
  const float dm = (D.xr * D.xr + D.yr * D.yr) * 1.001f;
  D.dmax2 = (dm == dm) ? dm : __builtin_huge_valf();
  D.mag = fabsf(D.x0) + fabsf(D.y0) + D.xr + D.yr;
  const float rmax = sqrtf(__int_as_float(d2)) * 0.5006f;          // (>= the largest inflated circumradius: the records' 0.5005 factor, rounded up)
  const float big = rc > 0 ? 4.0f * 0.5005f * (rs / (float)rc) : 0.f;
  D.rcap = rmax < big ? rmax : big;
  return D;
}

struct MkLdsSelect {
  MkLdsBuild b;
  uint32_t list[kMkCapMax];
  int s_i[16];
  int s_redi[8 * kMkWaves];
};

// the call is complete: the count for the caller, the step count for the thread's next call
__device__ __forceinline__ void mk_finish(const MkArgs& a, MkCtl* c, int kept, int steps) {
  c->done = 1;
  if (a.num_keep) *a.num_keep = (int64_t)kept;
  if (a.hint_host) {
    __hip_atomic_store(a.hint_host, steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.hint_host + 1, kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------ A: the next chunk and its table (ONE workgroup)
__device__ void mk_select_phase(const MkArgs& a, MkLdsSelect& S, u64* t0) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  MkCtl* c = a.ctl;
  const int step = c->step;
  int cur = c->cur, cap = c->cap;
  MkData D;
  D.x0 = c->dx0; D.y0 = c->dy0; D.xr = c->dxr; D.yr = c->dyr; D.rcap = c->drcap; D.mag = c->dmag; D.dmax2 = c->dmax2;
  if (step == 0) {
    cap = a.cap_first;
    D = mk_data_stats(a.bbpart, a.nparts, S.s_redi);
    if (tid == 0) {
      c->dx0 = D.x0; c->dy0 = D.y0; c->dxr = D.xr; c->dyr = D.yr; c->drcap = D.rcap; c->dmag = D.mag; c->dmax2 = D.dmax2;
      // (the step count is not known before the call completes: -2 = "started", read by the thread's next call if nothing better follows)
      if (a.hint_host) __hip_atomic_store(a.hint_host, -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  const float dmax2 = D.dmax2;
  if (cap > a.capmax) cap = a.capmax;
  // the first `cap` alive positions of [cur, n)
  const int se = a.n;
  const int w_first = cur >> 6, w_last = (se - 1) >> 6;
  int off = 0;
  __syncthreads();
  if (tid == 0) S.s_i[8] = se;
  for (int wbase = w_first; wbase <= w_last && off < cap; wbase += kMkThreads) {
    const int w = wbase + tid;
    u64 m = 0ull;
    if (w <= w_last) {
      m = ldg_agent(a.alive + w);
      const long long lo = (long long)w * 64;
      if (lo < cur) m &= ~((1ull << (cur - lo)) - 1ull);
      if (lo + 64 > se) m &= (1ull << (se - lo)) - 1ull;
    }
    const int cnt = __popcll(m);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    __syncthreads();
    if (lane == 63) S.s_i[wv] = incl;
    __syncthreads();
    int wpre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kMkWaves; k++) { const int t = S.s_i[k]; if (k < wv) wpre += t; tot += t; }
    int slot = off + wpre + incl - cnt;
    while (m && slot < cap) {
      const int b = __builtin_ctzll(m);
      m &= m - 1;
      const uint32_t pos = (uint32_t)(w * 64 + b);
      S.list[slot] = pos;
      if (slot == cap - 1) S.s_i[8] = (int)pos + 1;
      slot++;
    }
    off += tot;
  }
  __syncthreads();
  const int newcur = S.s_i[8];
  const int cn = off < cap ? off : cap;
  mk_lap(a, 3, t0);
  if (cn == 0) {                                        // nothing alive is left: the call is complete
    if (tid == 0) { c->cn = 0; c->cur = se; mk_finish(a, c, c->kept, step); }
    return;
  }
  for (int k = tid; k < cn; k += kMkThreads) a.cidx[k] = S.list[k];
  for (int k = tid; k * 16 < cn; k += kMkThreads) reinterpret_cast<uint4*>(a.hasin)[k] = make_uint4(0u, 0u, 0u, 0u);
  mk_build_table(a, a.rec, S.list, cn, D, &c->g, a.ent, a.start, S.b);
  mk_lap(a, 4, t0);
  if (tid == 0) {
    c->cn = cn; c->chunk_first = cur; c->cur = newcur; c->cap = cap; c->nrow = 0; c->stage = 1; c->n1 = 0;
    stg_agent(a.nedges, 0);
    // Brute boxes (not finite, or ill conditioned against the extent of the data) meet every partner: a few stray ones are fine,
    // a chunk full of them means quadratic work here -- the persistent kernel's exhaustive scans are the better tool for such data.
    if (c->g.nbrute * 64 > cn) c->bail = 1;
  }
}

// ------------------------------------------------------------------ queues of the probe kernels (three words per entry)
// a = the query's id: its position (cross) or entry index << 16 | query index, both chunk-local (pairs: the edge); b = the
// entry's position; c = the query's position
struct MkQueue {
  uint32_t *a, *b, *c;      // LDS, 128 entries each
  int head, count;
  __device__ __forceinline__ void push(bool pass, uint32_t va, uint32_t vb, uint32_t vc) {
    const u64 m = __ballot(pass);
    if (pass) { const int i = (head + count + __popcll(m & lanemask_lt())) & 127; a[i] = va; b[i] = vb; c[i] = vc; }
    count += __popcll(m);
  }
};
struct MkWaveLds { uint32_t qa[128], qb[128], qc[128]; };
constexpr int kMkStagePend = 384, kMkStageEdge = 448;   // (the probe workgroup stays at 32 KB of LDS: five per CU)
template <int NW> struct MkLdsProbe {
  uint16_t start[kMkSlots + 8];
  u64 kb[kMkCapMax / 64];
  MkWaveLds w[NW];
  // Appends are staged per WORKGROUP: one device atomic per workgroup and list instead of one per drain -- same-address device
  // atomics retire at ~11 ns apiece, and 12,500 drains on one counter were most of a 96 us cross phase.
  uint4 pbuf[kMkStagePend];
  uint32_t ebuf[kMkStageEdge];
  int pcnt, ecnt, pbase, ebase;
};
constexpr int kMkProbeWaves = 4;             // the lean probe kernels: 256 threads
constexpr int kMkProbeThreads = kMkProbeWaves * 64;

// a hit of either pair phase: "pairs" appends the edge (earlier member << 16 | later member), "cross" clears the query's alive bit
// (write-through store / device atomics: the last workgroup of a decide kernel reads the edges of the others)
template <bool CROSS>
__device__ __forceinline__ void mk_hit(const MkArgs& a, bool h, uint32_t qa) {
  if constexpr (CROSS) {
    if (h) atomicAnd(a.alive + (qa >> 6), ~(1ull << (qa & 63)));
  } else {
    const u64 hm = __ballot(h);
    if (hm) {
      const int lane = threadIdx.x & 63, first = (int)__builtin_ctzll(hm);
      int base = 0;
      if (lane == first) base = atomicAdd(a.nedges, __popcll(hm));
      base = __shfl(base, first);
      if (h) {
        const long long p = (long long)base + __popcll(hm & lanemask_lt());
        if (p < a.ecap) stg_agent(a.edges + p, qa); else stg_agent(&a.ctl->bail, 1);   // (a chunk above kMkTile members can outgrow the list)
        stg_agent(a.hasin + (qa & 0xffffu), (uint8_t)1);
      }
    }
  }
}
// append the flagged pairs of a wave to the pending list (one atomic per wave); an overflow raises the bail flag
__device__ __forceinline__ void mk_defer(const MkArgs& a, MkCtl* c, bool p, const uint4& v) {
  const u64 m = __ballot(p);
  if (!m) return;
  const int lane = threadIdx.x & 63, first = (int)__builtin_ctzll(m);
  int base = 0;
  if (lane == first) base = atomicAdd(&c->n1, __popcll(m));
  base = __shfl(base, first);
  if (p) {
    const long long i = (long long)base + __popcll(m & lanemask_lt());
    if (i < a.cap1) a.pend1[i] = v;
    else stg_agent(&c->bail, 1);
  }
}
// staged append (lean probe kernels): the wave's flagged values go to the workgroup's LDS buffer (one LDS atomic per wave);
// whatever does not fit goes straight to the global list.  mk_probe_phase publishes the buffer at the end.
template <typename T, typename F>
__device__ __forceinline__ void mk_stage(bool p, const T& v, T* buf, int* cnt, int cap, F&& direct) {
  const u64 m = __ballot(p);
  if (!m) return;
  const int lane = threadIdx.x & 63, first = (int)__builtin_ctzll(m);
  int base = 0;
  if (lane == first) base = atomicAdd(cnt, __popcll(m));
  base = __shfl(base, first);
  const int i = base + __popcll(m & lanemask_lt());
  const bool fits = p && i < cap;
  if (fits) buf[i] = v;
  direct(p && !fits);
}

// ------------------------------------------------------------------ B / D: the probe phases
// CROSS = false: queries = the chunk members, every EARLIER member i around a member j: IoU > thr becomes the edge (i << 16 | j);
// CROSS = true:  queries = the alive positions behind the chunk, every KEPT member around a query: IoU > thr clears the query's alive bit.
// Both read the chunk's one table.  An item = kMkQB = 8 consecutive query slots (positions, or chunk-local indices), kMkL = 8
// lanes each: lane k of a group walks the entries k, k + 8, ... of every cell-row range of the query's window, two entries per trip.
// What passes the circle test is queued per wave and goes through the register-only tests 64 pairs at a time (classify_quick);
// what those leave undecided is handed to the decide kernel behind this launch.
template <bool CROSS, int NW>
__device__ void mk_probe_phase(const MkArgs& a, MkLdsProbe<NW>& S, int wg, int nwg) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  MkCtl* c = a.ctl;
  const MkGrid g = c->g;
  const int cn = c->cn, cur = c->cur, n = a.n;
  const float dmax2 = c->dmax2, thr = a.thr;
  const float4* __restrict__ ent = a.ent;
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.start);
    uint4* dst = reinterpret_cast<uint4*>(S.start);
    for (int k = tid; k < (kMkSlots + 8) / 8; k += NT) dst[k] = src[k];
    if constexpr (CROSS) { for (int k = tid; k < (cn + 63) / 64; k += NT) S.kb[k] = a.kbits[k]; }
    if (tid == 0) { S.pcnt = 0; S.ecnt = 0; }
  }
  __syncthreads();
  MkWaveLds& L = S.w[wv];
  MkQueue Q{L.qa, L.qb, L.qc, 0, 0};
  const int nbrute = g.nbrute, total = g.total;
  auto rec_of = [&](uint32_t pos) -> const float4* { return a.rec + (size_t)pos * 4; };
  auto drain = [&](int cnt) {
    wave_sync();
    int res = 0; uint32_t qa = 0, qb = 0, qc = 0;
    if (lane < cnt) {
      const int i = (Q.head + lane) & 127;
      qa = L.qa[i]; qb = L.qb[i]; qc = L.qc[i];
      res = RotGeom::classify_quick(rec_of(qb), rec_of(qc), thr, true);
    }
    if constexpr (!CROSS) {
      mk_stage(res == 1, qa, S.ebuf, &S.ecnt, kMkStageEdge, [&](bool ov) { mk_hit<false>(a, ov, qa); });
    } else {
      mk_hit<true>(a, res == 1, qa);
    }
    Q.head = (Q.head + cnt) & 127; Q.count -= cnt;
    const uint4 v = make_uint4(qc, qb, qa, 0u);
    mk_stage(res >= 2, v, S.pbuf, &S.pcnt, kMkStagePend, [&](bool ov) { mk_defer(a, c, ov, v); });
    wave_sync();
  };

  const int w0 = CROSS ? (cur >> 6) : 0;
  const int nwords = CROSS ? (cur < n ? ((n - 1) >> 6) - w0 + 1 : 0) : ((cn + 63) >> 6);
  const int nitems = nwords * kMkQB;
  // (measured and not kept: items drawn from a counter in LDS with the next item's words requested ahead -- the extra live
  //  registers spill at the 96 the five-waves-per-SIMD budget allows: uniform 0.94 -> 1.02 ms)
  for (int it = wv * nwg + wg; it < nitems; it += nwg * NW) {          // (a few hundred items -- a 2048-member chunk -- spread over all CUs)
    const int word = it / kMkQB, part = it - word * kMkQB;
    u64 m;
    int lo;
    if constexpr (CROSS) {
      const int w = w0 + word;
      lo = w * 64;
      m = ldg_agent(a.alive + w);
      if (lo < cur) m &= ~((1ull << (cur - lo)) - 1ull);
      if (lo + 64 > n) m &= (1ull << (n - lo)) - 1ull;
    } else {
      lo = word * 64;
      m = (lo + 64 <= cn) ? ~0ull : ((1ull << (cn - lo)) - 1ull);
    }
    if (((m >> (part * kMkQB)) & ((1ull << kMkQB) - 1ull)) == 0ull) continue;       // none of this item's eight slots holds a query
    const int qlane = part * kMkQB + (lane / kMkL), k = lane & (kMkL - 1);
    const bool valid = (m >> qlane) & 1ull;
    const uint32_t qid = (uint32_t)(lo + qlane);                                    // position (CROSS) or chunk-local index
    uint32_t qpos = qid;
    if constexpr (!CROSS) qpos = valid ? a.cidx[qid] : 0u;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) q0 = a.rec[(size_t)qpos * 4];
    const float qx = q0.x, qy = q0.y, qr = q0.z;
    const bool qnever = !mk_finite3(qx, qy, qr) || !(q0.w >= kGridIllCond * dmax2);   // no pair of this query may be dropped unseen
    const float R = (qr + g.rmax) * 1.0001f + (fabsf(qx) + fabsf(qy) + g.mag) * 4e-7f;
    const int cx0 = mk_cell(qx - R, g.x0, g.inv, g.cxl), cx1 = mk_cell(qx + R, g.x0, g.inv, g.cxl);
    const int cy0 = mk_cell(qy - R, g.y0, g.inv, g.cyl), cy1 = mk_cell(qy + R, g.y0, g.inv, g.cyl);
    const bool all = qnever || (cx1 - cx0 > 62) || (cy1 - cy0 > 126);                 // the whole table in one range
    const int xb0 = cx0 >> 6, xb1 = cx1 >> 6;
    int cy = cy0, xb = xb0;
    int e = k, e1 = valid ? (all ? total : nbrute) : 0;
    bool more = valid && !all && total > nbrute;
    // one entry against this lane's query: false = the pair is dropped here (never for a brute entry or a brute query)
    auto test = [&](const float4& en, bool is_brute, uint32_t& qa) -> bool {
      const uint32_t zb = __float_as_uint(en.z), el = zb & 0xFFFFu;
      const float er = __uint_as_float(zb & 0xFFFF0000u);
      const float dx = qx - en.x, dy = qy - en.y, rs = er + qr;
      const float d2 = dx * dx + dy * dy;
      bool ok = qnever || is_brute || !(d2 > rs * rs);
      if constexpr (CROSS) { ok = ok && ((S.kb[el >> 6] >> (el & 63)) & 1ull); qa = qid; }
      else { ok = ok && (el < qid); qa = (el << 16) | qid; }
      return ok;
    };
    for (;;) {
      bool act = e < e1;
      if (!act && more) {                                     // the next range: cells [cx0, cx1] of block xb in cell row cy
        const int l0 = (cx0 > (xb << 6) ? cx0 : (xb << 6)) & 63, l1 = (cx1 < (xb << 6) + 63 ? cx1 : (xb << 6) + 63) & 63;
        const int base = 64 * mk_trow(cy, xb);
        e = nbrute + (int)S.start[base + l0] + k; e1 = nbrute + (int)S.start[base + l1 + 1];
        if (++xb > xb1) { xb = xb0; if (++cy > cy1) more = false; }
        act = e < e1;
      }
      if (__ballot(act || more) == 0ull) break;
      // kMkNE entries per lane and trip, requested together: a trip is a dependent L2 round trip whatever it carries, and a dense
      // range (S-uniform: ~110 entries per cell row of a 16,384-member chunk) is walked in a quarter of the trips two entries took
      bool pass[kMkNE]; uint32_t qa[kMkNE], ep[kMkNE];
#pragma unroll
      for (int u = 0; u < kMkNE; u++) { pass[u] = false; qa[u] = 0u; ep[u] = 0u; }
      if (act) {
        float4 en[kMkNE];
#pragma unroll
        for (int u = 0; u < kMkNE; u++) en[u] = (e + u * kMkL < e1) ? ent[e + u * kMkL] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < kMkNE; u++)
          if (e + u * kMkL < e1) { pass[u] = test(en[u], e + u * kMkL < nbrute, qa[u]); ep[u] = __float_as_uint(en[u].w); }
        e += kMkNE * kMkL;
      }
#pragma unroll
      for (int u = 0; u < kMkNE; u++) {
        if (__ballot(pass[u])) {
          Q.push(pass[u], qa[u], ep[u], qpos);
          if (Q.count >= 64) drain(64);
        }
      }
    }
  }
  if (Q.count > 0) drain(Q.count);
  // publish the workgroup's staged appends: one device atomic per list
  __syncthreads();
  const int np = S.pcnt < kMkStagePend ? S.pcnt : kMkStagePend, ne = S.ecnt < kMkStageEdge ? S.ecnt : kMkStageEdge;
  if (tid == 0) {
    S.pbase = np > 0 ? atomicAdd(&c->n1, np) : 0;
    S.ebase = ne > 0 ? atomicAdd(a.nedges, ne) : 0;
  }
  __syncthreads();
  for (int i = tid; i < np; i += NT) {
    const long long d = (long long)S.pbase + i;
    if (d < a.cap1) a.pend1[d] = S.pbuf[i]; else stg_agent(&c->bail, 1);
  }
  for (int i = tid; i < ne; i += NT) {
    const long long d = (long long)S.ebase + i;
    if (d < a.ecap) stg_agent(a.edges + d, S.ebuf[i]); else stg_agent(&c->bail, 1);
    stg_agent(a.hasin + (S.ebuf[i] & 0xffffu), (uint8_t)1);
  }
}

// ------------------------------------------------------------------ D': the cross probe on a table of the KEPT rows in LDS
// The chunk's table holds every member, the cross phase needs the kept ones (300 of 2048 in S-clustered K=300), and its entries
// sit in global memory: a query walks five or six cell-row ranges, every trip a dependent L2 round trip -- 61-67 us per step at
// 100k in the clustered regimes, nearly all of it latency.  Here every workgroup builds its OWN table of the step's kept rows
// (resolve left their positions in a.rows) in LDS: <= kMkXRows rows per pass (more rows: more passes over the queries), a
// counting sort of a few thousand quads, ~5 us, after which a trip costs two LDS reads.  Same geometry rules, same exactness
// rules as mk_build_table / mk_probe_phase: brute rows in front, a query that is not well conditioned reads everything.
constexpr int kMkXThreads = 512;
constexpr int kMkXWaves = kMkXThreads / 64;
constexpr int kMkXRows = 3072;            // kept rows of a pass
constexpr int kMkXPer = kMkXRows / kMkXThreads;
constexpr int kMkXSlots = 4096;           // 64 table rows of 64 cells
constexpr int kMkXTabPad = kMkXSlots / 32;
constexpr int kMkXBruteCtr = kMkXSlots + kMkXTabPad + 8;
__device__ __forceinline__ int mk_xtrow(int cy, int xb) { return (cy + 37 * xb) & 63; }
__device__ __forceinline__ int mk_xslot(int cx, int cy) { return (cx & 63) + 64 * mk_xtrow(cy, cx >> 6); }
struct MkLdsCross {
  float4 ent[kMkXRows];                   // {x, y, radius rounded up to 16 bits, position}
  uint16_t start[kMkXSlots + 8];
  union {
    int tab[kMkXSlots + kMkXTabPad + 40]; // the build's padded counters
    struct { MkWaveLds w[kMkXWaves]; uint4 pbuf[kMkStagePend]; } q;   // the probe's queues and its staged appends
  } u;
  int redi[16];
  int pcnt, pbase;
};
static_assert(sizeof(MkLdsCross) <= 80 * 1024, "two workgroups per CU");

__device__ void mk_cross_lds_phase(const MkArgs& a, MkLdsCross& S, int wg, int nwg) {
  constexpr int NT = kMkXThreads, NW = kMkXWaves;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  MkCtl* c = a.ctl;
  const int cur = c->cur, n = a.n, nrow = c->nrow;
  const uint32_t* __restrict__ rows = a.rows + (c->kept - nrow);
  const float dmax2 = c->dmax2, thr = a.thr, rcap = c->drcap;
  const float dx0 = c->dx0, dy0 = c->dy0, dxr = c->dxr, dyr = c->dyr, dmag = c->dmag;
  const int w0 = cur >> 6;
  const int nwords = cur < n ? ((n - 1) >> 6) - w0 + 1 : 0;
  const int nitems = nwords * kMkQB;
  auto rec_of = [&](uint32_t pos) -> const float4* { return a.rec + (size_t)pos * 4; };
  const bool xprof = a.prof != nullptr && tid == 0;
  u64 tx0 = xprof ? wall_clock64() : 0ull, tx_build = 0ull, tx_items = 0ull;
  for (int t0 = 0; t0 < nrow; t0 += kMkXRows) {
    const int cnt = (nrow - t0) < kMkXRows ? (nrow - t0) : kMkXRows;
    u64 txa = xprof ? wall_clock64() : 0ull;
    // ---- the pass's table
    float4 q[kMkXPer]; uint32_t pos[kMkXPer]; int slot[kMkXPer];
#pragma unroll
    for (int u = 0; u < kMkXPer; u++) { const int k = tid + u * NT; pos[u] = k < cnt ? rows[t0 + k] : 0u; }
#pragma unroll
    for (int u = 0; u < kMkXPer; u++) { const int k = tid + u * NT; q[u] = k < cnt ? a.rec[(size_t)pos[u] * 4] : make_float4(0.f, 0.f, 0.f, 0.f); }
    for (int s2 = tid; s2 < kMkXSlots + kMkXTabPad + 40; s2 += NT) S.u.tab[s2] = 0;
    float gx0, gy0, ginv, grmax; int cxl, cyl;
    {
      const float area = fmaxf(dxr, rcap) * fmaxf(dyr, rcap);
      const float expect = (float)cnt * (25.f * rcap * rcap) / area;                // rows in a (5 rcap)^2 window
      float side = rcap * (expect < 16.f ? 2.0f : 0.75f);
      const float ext = fmaxf(dxr, dyr);
      if (!(side > ext * 1e-6f)) side = ext * 1e-6f;
      if (!(side > 1e-30f)) side = 1.0f;
      gx0 = dx0; gy0 = dy0; ginv = 1.0f / side; grmax = __uint_as_float(mk_rup(rcap));
      const float fx = floorf(dxr * ginv), fy = floorf(dyr * ginv);
      cxl = (fx < 1e9f) ? (int)fx : 1000000000; cyl = (fy < 1e9f) ? (int)fy : 1000000000;
      if (!(fx >= 0.f)) cxl = 0;
      if (!(fy >= 0.f)) cyl = 0;
    }
    auto brute = [&](const float4& v) -> bool { return !mk_finite3(v.x, v.y, v.z) || !(v.w >= kGridIllCond * dmax2) || (v.z > rcap); };
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kMkXPer; u++) {
      const int k = tid + u * NT;
      slot[u] = -1;
      if (k < cnt) {
        const int s = brute(q[u]) ? kMkXBruteCtr : mk_xslot(mk_cell(q[u].x, gx0, ginv, cxl), mk_cell(q[u].y, gy0, ginv, cyl));
        slot[u] = s == kMkXBruteCtr ? s : s + (s >> 5);
        atomicAdd(&S.u.tab[slot[u]], 1);
      }
    }
    __syncthreads();
    const int nbrute = S.u.tab[kMkXBruteCtr];
    {
      constexpr int PER = kMkXSlots / NT;   // 8 consecutive slots per thread
      int loc[PER]; int tsum = 0;
#pragma unroll
      for (int u = 0; u < PER; u++) { const int s = tid * PER + u; loc[u] = tsum; tsum += S.u.tab[s + (s >> 5)]; }
      int incl = tsum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
      if (lane == 63) S.redi[wv] = incl;
      __syncthreads();
      int wpre = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) if (k < wv) wpre += S.redi[k];
      const int tbase = wpre + incl - tsum;
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int s = tid * PER + u;
        S.u.tab[s + (s >> 5)] = nbrute + tbase + loc[u];        // the running fill pointer of the slot
        S.start[s] = (uint16_t)(tbase + loc[u]);
      }
      if (tid == 0) { S.start[kMkXSlots] = (uint16_t)(cnt - nbrute); S.u.tab[kMkXBruteCtr] = 0; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kMkXPer; u++) {
      if (slot[u] >= 0) {
        const int e = atomicAdd(&S.u.tab[slot[u]], 1);
        S.ent[e] = make_float4(q[u].x, q[u].y, __uint_as_float(mk_rup(q[u].z)), __uint_as_float(pos[u]));
      }
    }
    __syncthreads();                         // the counters are dead: their memory serves the queues from here on
    if (tid == 0) S.pcnt = 0;
    __syncthreads();
    if (xprof) { const u64 t = wall_clock64(); tx_build += t - txa; txa = t; }
    // ---- the probe
    MkWaveLds& L = S.u.q.w[wv];
    MkQueue Q{L.qa, L.qb, L.qc, 0, 0};
    auto drain = [&](int dcnt) {
      wave_sync();
      int res = 0; uint32_t qa = 0, qb = 0;
      if (lane < dcnt) {
        const int i = (Q.head + lane) & 127;
        qa = L.qa[i]; qb = L.qb[i];
        res = RotGeom::classify_quick(rec_of(qb), rec_of(qa), thr, true);
      }
      mk_hit<true>(a, res == 1, qa);
      Q.head = (Q.head + dcnt) & 127; Q.count -= dcnt;
      const uint4 v = make_uint4(qa, qb, qa, 0u);
      mk_stage(res >= 2, v, S.u.q.pbuf, &S.pcnt, kMkStagePend, [&](bool ov) { mk_defer(a, c, ov, v); });
      wave_sync();
    };
    // An item's two loads -- the alive word of its eight positions and their quads -- do not depend on each other (the quad of a
    // dead position is simply not used), and the NEXT item's are requested before this one is walked: one exposed round trip per
    // wave instead of two per item.
    // (measured and not kept: the workgroup's waves drawing their items from a counter in LDS instead of the fixed deal -- the mean
    //  workgroup spends 16 of its 31 us in wave 0's items at S-clustered K=300, 3 in the table; 30.6 against 31.8 us, and 8 spilled registers)
    const int stride = nwg * NW;
    auto request = [&](int it, u64& m, float4& q0) {
      const int word = it / kMkQB, part = it - word * kMkQB;
      const int w = w0 + word;
      const long long p = (long long)w * 64 + part * kMkQB + (lane / kMkL);
      m = ldg_agent(a.alive + w);
      q0 = p < n ? a.rec[(size_t)p * 4] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    u64 m_nx = 0ull; float4 q_nx = make_float4(0.f, 0.f, 0.f, 0.f);
    int it = wg * NW + wv;
    if (it < nitems) request(it, m_nx, q_nx);
    for (; it < nitems; it += stride) {
      const int word = it / kMkQB, part = it - word * kMkQB;
      const int w = w0 + word, lo = w * 64;
      u64 m = m_nx; const float4 q0 = q_nx;
      if (it + stride < nitems) request(it + stride, m_nx, q_nx);
      if (lo < cur) m &= ~((1ull << (cur - lo)) - 1ull);
      if (lo + 64 > n) m &= (1ull << (n - lo)) - 1ull;
      if (((m >> (part * kMkQB)) & ((1ull << kMkQB) - 1ull)) == 0ull) continue;
      const int qlane = part * kMkQB + (lane / kMkL), k = lane & (kMkL - 1);
      const bool valid = (m >> qlane) & 1ull;
      const uint32_t qpos = (uint32_t)(lo + qlane);
      const float qx = q0.x, qy = q0.y, qr = q0.z;
      const bool qnever = !mk_finite3(qx, qy, qr) || !(q0.w >= kGridIllCond * dmax2);
      const float R = (qr + grmax) * 1.0001f + (fabsf(qx) + fabsf(qy) + dmag) * 4e-7f;
      const int cx0 = mk_cell(qx - R, gx0, ginv, cxl), cx1 = mk_cell(qx + R, gx0, ginv, cxl);
      const int cy0 = mk_cell(qy - R, gy0, ginv, cyl), cy1 = mk_cell(qy + R, gy0, ginv, cyl);
      const bool all = qnever || (cx1 - cx0 > 62) || (cy1 - cy0 > 62);
      const int xb0 = cx0 >> 6, xb1 = cx1 >> 6;
      int cy = cy0, xb = xb0;
      int e = k, e1 = valid ? (all ? cnt : nbrute) : 0;
      bool more = valid && !all && cnt > nbrute;
      for (;;) {
        bool act = e < e1;
        if (!act && more) {
          const int l0 = (cx0 > (xb << 6) ? cx0 : (xb << 6)) & 63, l1 = (cx1 < (xb << 6) + 63 ? cx1 : (xb << 6) + 63) & 63;
          const int base = 64 * mk_xtrow(cy, xb);
          e = nbrute + (int)S.start[base + l0] + k; e1 = nbrute + (int)S.start[base + l1 + 1];
          if (++xb > xb1) { xb = xb0; if (++cy > cy1) more = false; }
          act = e < e1;
        }
        if (__ballot(act || more) == 0ull) break;
        bool pass = false; uint32_t ep = 0;
        if (act) {
          const float4 en = S.ent[e];
          const float dx = qx - en.x, dy = qy - en.y, rs = en.z + qr;
          pass = qnever || (e < nbrute) || !(dx * dx + dy * dy > rs * rs);
          ep = __float_as_uint(en.w);
          e += kMkL;
        }
        if (__ballot(pass)) {
          Q.push(pass, qpos, ep, 0u);
          if (Q.count >= 64) drain(64);
        }
      }
    }
    if (Q.count > 0) drain(Q.count);
    if (xprof) { const u64 t = wall_clock64(); tx_items += t - txa; txa = t; }      // (wave 0's own items)
    __syncthreads();
    const int np = S.pcnt < kMkStagePend ? S.pcnt : kMkStagePend;
    if (tid == 0) S.pbase = np > 0 ? atomicAdd(&c->n1, np) : 0;
    __syncthreads();
    for (int i = tid; i < np; i += NT) {
      const long long d = (long long)S.pbase + i;
      if (d < a.cap1) a.pend1[d] = S.u.q.pbuf[i]; else stg_agent(&c->bail, 1);
    }
    __syncthreads();
  }
  if (xprof) {   // development aid (OBB_NMS_PHASE_PROF): per-workgroup times of this kernel, sums and maxima over workgroups and steps
    const u64 tt = wall_clock64() - tx0;
    atomicAdd(a.prof + 49, tx_build); atomicAdd(a.prof + 50, tx_items); atomicAdd(a.prof + 51, tt); atomicAdd(a.prof + 53, 1ull);
    atomicMax(a.prof + 52, tt); atomicMax(a.prof + 54, tx_build); atomicMax(a.prof + 55, tx_items);
  }
}

// ------------------------------------------------------------------ the interval and the exact clip on the pending list (dense: one lane per pair)
// The list is short against the machine (at most a trip or two per wave), so what counts is the length of a wave's dependent
// chain: the pair's positions come with the entry (no look-up), and the lanes the interval leaves undecided run the exact clip
// right there on the records they already hold instead of queueing for a full wave.
template <bool CROSS>
__device__ __forceinline__ void mk_decide_phase(const MkArgs& a, float* scr_wave) {
  MkCtl* c = a.ctl;
  const int lane = threadIdx.x & 63;
  int n1 = c->n1; if (n1 > a.cap1) n1 = a.cap1;
  // (wave w of workgroup b takes trips w * #workgroups + b, ...: a short list -- a third of the waves have a trip -- spreads over all
  //  CUs, one or two waves each, instead of filling the first third of the workgroups with eight waves that share four SIMDs)
  const int gw = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x), nw = (int)((gridDim.x * blockDim.x) >> 6);
  for (int base = gw * 64; base < n1; base += nw * 64) {
    const int i = base + lane;
    int res = 0; uint4 p = make_uint4(0u, 0u, 0u, 0u);
    const float4 *ra = a.rec, *rb = a.rec;
    if (i < n1) {
      p = a.pend1[i];
      ra = a.rec + (size_t)p.y * 4; rb = a.rec + (size_t)p.x * 4;          // (the entry = the earlier box first)
      // (cross: a query that another row has removed meanwhile -- most have a quick hit from the row of their own object -- needs
      //  no further decision)
      bool open = true;
      if constexpr (CROSS) open = (ldg_agent(a.alive + (p.z >> 6)) >> (p.z & 63)) & 1ull;
      if (open) res = a.skip_full ? 2 : RotGeom::classify_full(ra, rb, a.thr);
    }
    bool h = res == 1;
    if (__ballot(res == 2)) {
      if (res == 2) h = RotGeom::hit_exact(ra, rb, a.thr, scr_wave + lane);
    }
    mk_hit<CROSS>(a, h, p.z);
  }
}

// ------------------------------------------------------------------ C: resolve the chunk, append its kept boxes to the output, publish the kept bits (ONE workgroup)
__device__ void mk_resolve_phase(const MkArgs& a, uint8_t* smem, size_t smem_bytes, int* s_i, u64* t0) {
  const int tid = threadIdx.x;
  MkCtl* c = a.ctl;
  const int cn = c->cn, kept_before = c->kept, cur = c->cur, step = c->step, cap = c->cap;
  const int E = ldg_agent(a.nedges);
  NmsArgs r{};
  r.rec = a.rec; r.order = a.order; r.keep_cnt = a.keep_cnt; r.keep_out = a.keep_out; r.rows = a.rows; r.nrows = a.nrows;
  r.edges = a.edges; r.nedges = a.nedges; r.ecap = a.ecap; r.n = a.n; r.nseg = 1; r.capmax = a.capmax; r.max_keep = 0; r.lpt = 0; r.prof = a.prof;
  __syncthreads();
  const int total = nms_resolve(r, 0, 0, 0, cn, kept_before, a.cidx, smem, smem_bytes, s_i, a.hasin);
  // the kept bits of the chunk members for the cross probe (nms_resolve leaves the members' states at the head of its LDS block)
  const bool last = cur >= a.n;                         // nothing behind the chunk: no cross phase, the call is complete
  if (!last) {
    const uint8_t* state = smem;
    for (int j0 = (tid >> 6) * 64; j0 < cn; j0 += kMkThreads) {
      const int j = j0 + (tid & 63);
      const u64 m = __ballot(j < cn && state[j] == 1);
      if ((tid & 63) == 0) a.kbits[j0 >> 6] = m;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  mk_lap(a, 1, t0);
  if (a.prof != nullptr && tid == 0) { a.prof[38] += (u64)E; a.prof[39] += (u64)cn; a.prof[40] += (u64)total; a.prof[37] += 1; }
  if (tid == 0) {
    c->kept = kept_before + total;
    c->nrow = last ? 0 : total;
    c->step = step + 1;
    c->stage = 3; c->n1 = 0;
    // the next chunk: twice as large, four times after a sparse one (few conflicts inside the chunk)
    const long long nc = (long long)cap * ((4LL * E <= cn) ? 4 : 2);
    c->cap = (int)(nc > a.capmax ? a.capmax : nc);
    if (last) mk_finish(a, c, kept_before + total, step + 1);
  }
}

constexpr size_t kMkSerialLds = 128 * 1024;   // dynamic LDS of the kernels that run a serial phase (resolve: the edge list in LDS)

// Every kernel of a step first looks at the control block: the call may be complete (done), the phase kernels may have stood back
// (bail: the persistent kernel behind them carries on), or the step's turn may not have come (stage).
__global__ __launch_bounds__(kMkThreads) void k_mk_select(MkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mk_smem[];
  const MkCtl* c = a.ctl;
  if (c->done || c->bail || c->stage != 0) return;
  u64 t0 = a.prof ? wall_clock64() : 0ull;
  mk_select_phase(a, *reinterpret_cast<MkLdsSelect*>(mk_smem), &t0);
}
template <bool CROSS>
__global__ __launch_bounds__(kMkProbeThreads, 4) void k_mk_probe(MkArgs a) {
  __shared__ MkLdsProbe<kMkProbeWaves> S;
  const MkCtl* c = a.ctl;
  if (c->done || c->bail || c->stage != (CROSS ? 3 : 1)) return;
  if (CROSS && c->nrow == 0) return;
  mk_probe_phase<CROSS, kMkProbeWaves>(a, S, (int)blockIdx.x, (int)gridDim.x);
}
// the cross probe on a table of the kept rows in LDS (mk_cross_lds_phase): two workgroups per CU
__global__ __launch_bounds__(kMkXThreads, 4) void k_mk_cross_lds(MkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mk_smem[];
  const MkCtl* c = a.ctl;
  if (c->done || c->bail || c->stage != 3 || c->nrow == 0) return;
  mk_cross_lds_phase(a, *reinterpret_cast<MkLdsCross*>(mk_smem), (int)blockIdx.x, (int)gridDim.x);
}
// The decide kernel of a pair phase: the interval and the exact clip for the pairs its probe kernel left undecided, on every
// workgroup; the workgroup that is through LAST then runs the serial phase behind it -- resolve after "pairs", select (+ the next
// chunk's table) after "cross".  Nobody waits for anybody: no co-residency assumption.
template <bool CROSS>
__global__ __launch_bounds__(kMkThreads) void k_mk_decide(MkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mk_smem[];
  __shared__ int s_i[16];
  __shared__ int s_last;
  MkCtl* c = a.ctl;
  if (c->done || c->bail || c->stage != (CROSS ? 3 : 1)) return;
  u64 t0 = a.prof ? wall_clock64() : 0ull;
  if (c->n1 > 0) mk_decide_phase<CROSS>(a, reinterpret_cast<float*>(mk_smem) + (threadIdx.x >> 6) * (RotGeom::SCR * 64));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this workgroup's edges / kills are out (write-through stores, atomics)
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(&c->ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (int)gridDim.x - 1) ? 1 : 0;
    if (s_last) { stg_agent(&c->ticket, 0); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
  }
  __syncthreads();
  if (!s_last) return;
  if (ldg_agent(&c->bail)) return;                            // (raised by another workgroup of this launch: the persistent kernel takes over)
  mk_lap(a, CROSS ? 9 : 0, &t0);                              // (this workgroup's decide phase + the wait for its ticket)
  if (a.prof != nullptr && threadIdx.x == 0) a.prof[CROSS ? 43 : 42] += (u64)c->n1;
  if constexpr (CROSS) mk_select_phase(a, *reinterpret_cast<MkLdsSelect*>(mk_smem), &t0);
  else mk_resolve_phase(a, mk_smem, kMkSerialLds, s_i, &t0);
}

}  // namespace obb
