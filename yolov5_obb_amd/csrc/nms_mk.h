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
// Exactness.  A pair may only be dropped without a decision when RotGeom::cheap_reject would reject it: circumscribed circles
// apart AND both boxes well conditioned for a partner at that distance.  A box that is not well conditioned for ANY partner
// inside the data's bounding box (short side^2 < 2.34e-9 * diagonal^2, grid.h: kGridIllCond), not finite, or far larger than
// its chunk's mean is a "brute" entry: it sits in front of the table and is handed to every query; a query with that property
// reads the whole table.  Everything that is not dropped goes through classify_quick -> classify_full -> hit_exact exactly
// like in nms_core.h.
#pragma once
#include "nms_core.h"

namespace obb {

constexpr int kMkThreads = 512;
constexpr int kMkWaves = kMkThreads / 64;
constexpr int kMkSlots = 8192;            // table slots of a chunk's spatial hash: 128 table rows of 64 cells
constexpr int kMkL = 8;                   // lanes per query
constexpr int kMkQB = 64 / kMkL;          // queries of a wave in flight
constexpr int kMkCapMax = 8192;           // largest chunk (edge list: cap * (cap - 1) / 2 entries)
constexpr int kMkPer = kMkCapMax / kMkThreads;   // members per thread while a table is built
constexpr int kMkPend1 = 1 << 22;         // pending-list capacity (pairs); an overflow hands the call to the tail kernel
constexpr int kMkScr = 2;                 // exact-clip scratch blocks of the tail kernel's workgroup (12 KB each, taken under a lock)

struct MkGrid {                           // geometry of one table (written by one workgroup, read by the probe kernels of later launches)
  float x0, y0, inv, rmax, mag;           // origin, 1 / cell side, radius limit of an indexed entry, |x0| + |y0| + extent
  int cxl, cyl;                           // last cell per axis (clamp)
  int total, nbrute;                      // entries; the first nbrute are handed to every query
  int pad[7];
};
struct MkCtl {                            // control block of a call (zeroed before the first launch)
  int cur, kept, done, step;
  int cn, cap, nrow, chunk_first;
  float dmax2;
  int stage;                              // 0: nothing selected yet, 1: a chunk is selected (pairs / resolve pending), 3: its rows are resolved (cross pending or done)
  int bail;                               // the pending list overflowed: the phase kernels stand back, the tail kernel redoes the stage in flight
  int n1;                                 // pending pairs of the phase in flight: undecided after the quick tests
  int ticket;                             // workgroups of a decide kernel that are through (the last one runs the serial phase behind it)
  int pad[3];
  MkGrid gc, gr;                          // chunk members (pairs), kept rows (cross)
};

struct MkArgs {
  const float4* rec; const uint32_t* order; u64* alive; int n;
  MkCtl* ctl;
  uint32_t* cidx;                         // [capmax] positions of the chunk members, ascending
  float4* ent_c; uint16_t* start_c;       // chunk table: entries {x, y, r, chunk-local index}, slot starts [kMkSlots + 1]
  float4* ent_r; uint16_t* start_r;       // row table:   entries {x, y, r, position}
  uint32_t* edges; int* nedges; long long ecap;
  uint32_t* rows; int* nrows; int* keep_cnt; int64_t* keep_out;
  int64_t* num_keep;                      // written by whoever completes the call
  const int* bbpart; int nparts;
  int capmax, cap_first;
  float thr;
  uint2* pend1; int cap1;                 // (query, entry) pairs the quick tests left undecided -> the decide kernels
  int* hint_host;                         // pinned word: steps the call needed (read by the NEXT call of this thread; may be NULL)
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

struct MkLdsBuild {
  int tab[kMkSlots + 8];
  float red[kMkWaves][8];
  int redi[16];
};

// ------------------------------------------------------------------ the table of one member list (ONE workgroup of kMkThreads)
// members k = 0 .. cnt-1 (cnt <= kMkCapMax) at positions list[k] (LDS); entry id = k (LOCAL: chunk-local index) or the position.
// Quad 0 of every member is fetched ONCE, kMkPer independent loads per thread, and stays in registers through the three passes
// of the counting sort (statistics, counts, scatter): the passes cost LDS time, not a chain of memory round trips.
// Radius limit: rcap = min(largest radius, 4 x mean radius).  A box above it would widen every query's window; it becomes a
// "brute" entry like the boxes that are not finite or not well conditioned for any partner inside the data's bounding box.
template <bool LOCAL>
__device__ void mk_build_table(const float4* __restrict__ rec, const uint32_t* s_list, int cnt, float dmax2, MkGrid* g_out, float4* __restrict__ ent,
                               uint16_t* __restrict__ start_g, MkLdsBuild& S) {
  const int tid = threadIdx.x;
  float4 q[kMkPer];
  uint32_t pos[kMkPer];
#pragma unroll
  for (int u = 0; u < kMkPer; u++) {
    const int k = tid + u * kMkThreads;
    pos[u] = k < cnt ? s_list[k] : 0u;
  }
#pragma unroll
  for (int u = 0; u < kMkPer; u++) {
    const int k = tid + u * kMkThreads;
    q[u] = k < cnt ? rec[(size_t)pos[u] * 4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inf = __builtin_huge_valf();
  auto sound = [&](const float4& v) -> bool { return mk_finite3(v.x, v.y, v.z) && (v.w >= kGridIllCond * dmax2); };
  // pass 1: mean and largest radius, bounding box of the sound boxes
  float sr = 0.f, lx = inf, ly = inf, hx = -inf, hy = -inf, hr = 0.f; int ns = 0;
#pragma unroll
  for (int u = 0; u < kMkPer; u++) {
    const int k = tid + u * kMkThreads;
    if (k < cnt && sound(q[u])) {
      sr += q[u].z; ns++;
      lx = fminf(lx, q[u].x); ly = fminf(ly, q[u].y); hx = fmaxf(hx, q[u].x); hy = fmaxf(hy, q[u].y); hr = fmaxf(hr, q[u].z);
    }
  }
  const float sum_r = mk_block_sum(sr, S.red[0]);
  const int n_sound = mk_block_sumi(ns, S.redi);
  mk_block_minmax(lx, ly, hx, hy, hr, S.red);
  const float big = n_sound > 0 ? 4.0f * (sum_r / (float)n_sound) : 0.f;
  const float rcap = hr < big ? hr : big;
  auto brute = [&](const float4& v) -> bool { return !sound(v) || (v.z > rcap); };
  MkGrid g;
  g.total = cnt; g.rmax = rcap;
  if (n_sound > 0) {
    const float xr = hx - lx, yr = hy - ly;
    // cell side: 3/4 of the radius limit when a query meets many entries (their number decides the work), twice the limit when
    // it meets a handful (then the number of cell rows a query walks decides it)
    const float area = fmaxf(xr, rcap) * fmaxf(yr, rcap);
    const float expect = (float)n_sound * (25.f * rcap * rcap) / area;          // entries in a (5 rcap)^2 window
    float side = rcap * (expect < 16.f ? 2.0f : 0.75f);
    // (a cell side far below the extent / 2^20 buys nothing and would overflow the int cell arithmetic)
    const float ext = fmaxf(xr, yr);
    if (!(side > ext * 1e-6f)) side = ext * 1e-6f;
    if (!(side > 1e-30f)) side = 1.0f;
    g.x0 = lx; g.y0 = ly; g.inv = 1.0f / side;
    g.mag = fabsf(lx) + fabsf(ly) + xr + yr;
    const float fx = floorf(xr * g.inv), fy = floorf(yr * g.inv);
    g.cxl = fx < 1e9f ? (int)fx : 1000000000; g.cyl = fy < 1e9f ? (int)fy : 1000000000;
  } else {
    g.x0 = g.y0 = 0.f; g.inv = 1.f; g.mag = 0.f; g.cxl = g.cyl = 0;
  }
  for (int k = 0; k < 7; k++) g.pad[k] = 0;
  // pass 2: counts per slot (slot kMkSlots: the brute block)
  for (int s2 = tid; s2 < kMkSlots + 8; s2 += kMkThreads) S.tab[s2] = 0;
  __syncthreads();
  int slot[kMkPer];
#pragma unroll
  for (int u = 0; u < kMkPer; u++) {
    const int k = tid + u * kMkThreads;
    slot[u] = -1;
    if (k < cnt) {
      slot[u] = brute(q[u]) ? kMkSlots : mk_slot(mk_cell(q[u].x, g.x0, g.inv, g.cxl), mk_cell(q[u].y, g.y0, g.inv, g.cyl));
      atomicAdd(&S.tab[slot[u]], 1);
    }
  }
  __syncthreads();
  const int nbrute = S.tab[kMkSlots];
  g.nbrute = nbrute;
  __syncthreads();
  // exclusive scan of the kMkSlots counters: 16 per thread, then the thread totals
  constexpr int PER = kMkSlots / kMkThreads;
  int loc[PER]; int tsum = 0;
#pragma unroll
  for (int u = 0; u < PER; u++) { loc[u] = tsum; tsum += S.tab[tid * PER + u]; }
  int incl = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if ((tid & 63) >= d) incl += v; }
  if ((tid & 63) == 63) S.redi[tid >> 6] = incl;
  __syncthreads();
  int wpre = 0;
#pragma unroll
  for (int k = 0; k < kMkWaves; k++) if (k < (tid >> 6)) wpre += S.redi[k];
  const int tbase = wpre + incl - tsum;
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int st = tbase + loc[u];
    S.tab[tid * PER + u] = nbrute + st;              // becomes the running fill pointer of the slot (entry index)
    start_g[tid * PER + u] = (uint16_t)st;           // (cnt <= kMkCapMax < 65536)
  }
  if (tid == 0) { start_g[kMkSlots] = (uint16_t)(cnt - nbrute); S.tab[kMkSlots] = 0; }   // tab[kMkSlots]: fill pointer of the brute block
  __syncthreads();
  // pass 3: scatter
#pragma unroll
  for (int u = 0; u < kMkPer; u++) {
    const int k = tid + u * kMkThreads;
    if (k < cnt) {
      const int e = atomicAdd(&S.tab[slot[u]], 1);
      ent[e] = make_float4(q[u].x, q[u].y, q[u].z, __uint_as_float(LOCAL ? (uint32_t)k : pos[u]));
    }
  }
  if (tid == 0) *g_out = g;
  __syncthreads();
}

// the extent of the data from the key kernel's per-workgroup partials (nms.hip: local_extras): (xr^2 + yr^2) * 1.001, +inf when there
// is no usable extent (then every box is brute and the call is exhaustive)
__device__ __forceinline__ float mk_extent_dmax2(const int* __restrict__ bbpart, int nparts, int* s_redi) {
  int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = (int)0x80000000, by1 = (int)0x80000000;
  for (int i = threadIdx.x; i < nparts; i += kMkThreads) {
    const int4 q = reinterpret_cast<const int4*>(bbpart)[2 * i];
    bx0 = min(bx0, q.x); by0 = min(by0, q.y); bx1 = max(bx1, q.z); by1 = max(by1, q.w);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    bx0 = min(bx0, __shfl_xor(bx0, d)); by0 = min(by0, __shfl_xor(by0, d));
    bx1 = max(bx1, __shfl_xor(bx1, d)); by1 = max(by1, __shfl_xor(by1, d));
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { int* o = s_redi + (threadIdx.x >> 6) * 4; o[0] = bx0; o[1] = by0; o[2] = bx1; o[3] = by1; }
  __syncthreads();
  int bb[4] = {s_redi[0], s_redi[1], s_redi[2], s_redi[3]};
  for (int k = 1; k < kMkWaves; k++) {
    bb[0] = min(bb[0], s_redi[4 * k]); bb[1] = min(bb[1], s_redi[4 * k + 1]); bb[2] = max(bb[2], s_redi[4 * k + 2]); bb[3] = max(bb[3], s_redi[4 * k + 3]);
  }
  __syncthreads();
  if (bb[0] > bb[2] || bb[1] > bb[3]) return 0.f;                // no finite centre at all: nothing can be indexed anyway
  const float xr = grid_o2f(bb[2]) - grid_o2f(bb[0]), yr = grid_o2f(bb[3]) - grid_o2f(bb[1]);
  const float d2 = (xr * xr + yr * yr) * 1.001f;
  return (d2 == d2) ? d2 : __builtin_huge_valf();
}

struct MkLdsSelect {
  MkLdsBuild b;
  uint32_t list[kMkCapMax];
  int s_i[16];
  int s_redi[4 * kMkWaves];
};

// the call is complete: the count for the caller, the step count for the thread's next call
__device__ __forceinline__ void mk_finish(const MkArgs& a, MkCtl* c, int kept, int steps) {
  c->done = 1;
  if (a.num_keep) *a.num_keep = (int64_t)kept;
  if (a.hint_host) __hip_atomic_store(a.hint_host, steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ A: the next chunk and its table (ONE workgroup)
__device__ void mk_select_phase(const MkArgs& a, MkLdsSelect& S) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  MkCtl* c = a.ctl;
  const int step = c->step;
  int cur = c->cur, cap = c->cap;
  float dmax2 = c->dmax2;
  if (step == 0) {
    cap = a.cap_first < a.capmax ? a.cap_first : a.capmax;
    dmax2 = mk_extent_dmax2(a.bbpart, a.nparts, S.s_redi);
  }
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
  if (cn == 0) {                                        // nothing alive is left: the call is complete
    if (tid == 0) { c->cn = 0; c->cur = se; c->dmax2 = dmax2; mk_finish(a, c, c->kept, step); }
    return;
  }
  for (int k = tid; k < cn; k += kMkThreads) a.cidx[k] = S.list[k];
  mk_build_table<true>(a.rec, S.list, cn, dmax2, &c->gc, a.ent_c, a.start_c, S.b);
  if (tid == 0) {
    c->cn = cn; c->chunk_first = cur; c->cur = newcur; c->cap = cap; c->dmax2 = dmax2; c->nrow = 0; c->stage = 1; c->n1 = 0;
    stg_agent(a.nedges, 0);
  }
}

// ------------------------------------------------------------------ queues of the probe kernels (two words per entry)
struct MkQueue {
  uint32_t *a, *b;          // LDS, 128 entries each
  int head, count;
  __device__ __forceinline__ void push(bool pass, uint32_t va, uint32_t vb) {
    const u64 m = __ballot(pass);
    if (pass) { const int i = (head + count + __popcll(m & lanemask_lt())) & 127; a[i] = va; b[i] = vb; }
    count += __popcll(m);
  }
};
// LEAN (the probe kernels proper): only the quick stage runs here, undecided pairs go to the pending list of the decide kernel;
// otherwise (the tail kernel's one workgroup) all three stages run in place, as in nms_core.h.
template <bool LEAN> struct MkWaveLds;
template <> struct MkWaveLds<true> { uint32_t qa[128], qb[128]; };
template <> struct MkWaveLds<false> { uint32_t qa[128], qb[128], q1a[128], q1b[128], q2a[128], q2b[128]; };
template <bool LEAN, int NW> struct MkLdsProbe;
constexpr int kMkStagePend = 1024, kMkStageEdge = 992;   // (the lean probe workgroup stays at 32 KB of LDS: five per CU)
template <int NW> struct MkLdsProbe<true, NW> {
  uint16_t start[kMkSlots + 8];
  MkWaveLds<true> w[NW];
  // Appends are staged per WORKGROUP: one device atomic per workgroup and list instead of one per drain -- same-address device
  // atomics retire at ~11 ns apiece, and 12,500 drains on one counter were most of a 96 us cross phase.
  uint2 pbuf[kMkStagePend];
  uint32_t ebuf[kMkStageEdge];
  int pcnt, ecnt, pbase, ebase;
};
template <int NW> struct MkLdsProbe<false, NW> {
  uint16_t start[kMkSlots + 8];
  MkWaveLds<false> w[NW];
  float scr[kMkScr][RotGeom::SCR * 64];
  uint32_t pa[NW * 64], pb[NW * 64];         // pooled leftovers of the workgroup's waves
  int lock[kMkScr];
  int pool_n[4];
};
constexpr int kMkProbeWaves = 4;             // the lean probe kernels: 256 threads
constexpr int kMkProbeThreads = kMkProbeWaves * 64;

// a hit of either pair phase: "pairs" appends the edge (earlier member << 16 | later member), "cross" clears the query's alive bit
// (write-through store / device atomics: the last workgroup of a decide kernel reads the edges of the others)
template <bool CROSS>
__device__ __forceinline__ void mk_hit(const MkArgs& a, bool h, uint32_t q, uint32_t e) {
  if constexpr (CROSS) {
    if (h) atomicAnd(a.alive + (q >> 6), ~(1ull << (q & 63)));
  } else {
    const u64 hm = __ballot(h);
    if (hm) {
      const int lane = threadIdx.x & 63, first = (int)__builtin_ctzll(hm);
      int base = 0;
      if (lane == first) base = atomicAdd(a.nedges, __popcll(hm));
      base = __shfl(base, first);
      if (h) {
        const long long p = (long long)base + __popcll(hm & lanemask_lt());
        if (p < a.ecap) stg_agent(a.edges + p, (e << 16) | q);
      }
    }
  }
}
// append the flagged pairs of a wave to the pending list (one atomic per wave); an overflow raises the bail flag
__device__ __forceinline__ void mk_defer(MkCtl* c, int* counter, uint2* list, int cap, bool p, uint32_t q, uint32_t e) {
  const u64 m = __ballot(p);
  if (!m) return;
  const int lane = threadIdx.x & 63, first = (int)__builtin_ctzll(m);
  int base = 0;
  if (lane == first) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, first);
  if (p) {
    const long long i = (long long)base + __popcll(m & lanemask_lt());
    if (i < cap) list[i] = make_uint2(q, e);
    else stg_agent(&c->bail, 1);
  }
}
// staged append (lean probe kernels): the wave's flagged values go to the workgroup's LDS buffer (one LDS atomic per wave);
// whatever does not fit goes straight to the global list.  Returns nothing: mk_stage_flush publishes the buffer at the end.
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
template <bool CROSS> __device__ __forceinline__ const float4* mk_rec(const MkArgs& a, uint32_t id) {
  return a.rec + (size_t)(CROSS ? id : a.cidx[id]) * 4;
}

// ------------------------------------------------------------------ B / D: the probe phases
// CROSS = false: queries = the chunk members, table = the chunk members, a pair (earlier member i, later member j) with
//                IoU > thr becomes the edge (i << 16 | j);
// CROSS = true:  queries = the alive positions behind the chunk, table = the chunk's kept rows, IoU > thr clears the query's alive bit.
// An item = kMkQB = 8 consecutive query slots (positions, or chunk-local indices), kMkL = 8 lanes each: lane k of a group walks
// the entries k, k + 8, ... of every cell-row range of the query's window, two entries per trip.
template <bool CROSS, bool LEAN, int NW>
__device__ void mk_probe_phase(const MkArgs& a, MkLdsProbe<LEAN, NW>& S, int wg, int nwg) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  MkCtl* c = a.ctl;
  const MkGrid g = CROSS ? c->gr : c->gc;
  const int cn = c->cn, cur = c->cur, n = a.n;
  const float dmax2 = c->dmax2, thr = a.thr;
  const float4* __restrict__ ent = CROSS ? a.ent_r : a.ent_c;
  {
    const uint4* src = reinterpret_cast<const uint4*>(CROSS ? a.start_r : a.start_c);
    uint4* dst = reinterpret_cast<uint4*>(S.start);
    for (int k = tid; k < (kMkSlots + 8) / 8; k += NT) dst[k] = src[k];
    if constexpr (!LEAN) { if (tid < kMkScr) S.lock[tid] = 0; }
    else { if (tid == 0) { S.pcnt = 0; S.ecnt = 0; } }
  }
  __syncthreads();
  MkWaveLds<LEAN>& L = S.w[wv];
  MkQueue Q{L.qa, L.qb, 0, 0};
  [[maybe_unused]] MkQueue Q1{nullptr, nullptr, 0, 0}, Q2{nullptr, nullptr, 0, 0};
  if constexpr (!LEAN) { Q1.a = L.q1a; Q1.b = L.q1b; Q2.a = L.q2a; Q2.b = L.q2b; }
  const int nbrute = g.nbrute, total = g.total;

  // stage 2: the exact clip, on one of the workgroup's scratch blocks
  [[maybe_unused]] auto drain2 = [&](int cnt) {
    if constexpr (!LEAN) {
      wave_sync();
      const int sb = wv & (kMkScr - 1);
      if (lane == 0) { while (atomicCAS(&S.lock[sb], 0, 1) != 0) __builtin_amdgcn_s_sleep(2); }
      wave_sync();
      bool h = false; uint32_t q = 0, e = 0;
      if (lane < cnt) {
        const int i = (Q2.head + lane) & 127;
        q = Q2.a[i]; e = Q2.b[i];
        h = nms_stage_exact<RotGeom>(mk_rec<CROSS>(a, e), mk_rec<CROSS>(a, q), thr, S.scr[sb] + lane);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wave_sync();
      if (lane == 0) atomicExch(&S.lock[sb], 0);
      mk_hit<CROSS>(a, h, q, e);
      Q2.head = (Q2.head + cnt) & 127; Q2.count -= cnt;
      wave_sync();
    }
  };
  // stage 1b: the IoU interval
  [[maybe_unused]] auto drain1b = [&](int cnt) {
    if constexpr (!LEAN) {
      wave_sync();
      int res = 0; uint32_t q = 0, e = 0;
      if (lane < cnt) {
        const int i = (Q1.head + lane) & 127;
        q = Q1.a[i]; e = Q1.b[i];
        res = nms_stage_full<RotGeom>(mk_rec<CROSS>(a, e), mk_rec<CROSS>(a, q), thr);
      }
      mk_hit<CROSS>(a, res == 1, q, e);
      Q1.head = (Q1.head + cnt) & 127; Q1.count -= cnt;
      Q2.push(res == 2, q, e);
      wave_sync();
      if (Q2.count >= 64) drain2(64);
    }
  };
  // stage 1a: the register-only tests
  auto drain = [&](int cnt) {
    wave_sync();
    int res = 0; uint32_t q = 0, e = 0;
    if (lane < cnt) {
      const int i = (Q.head + lane) & 127;
      q = L.qa[i]; e = L.qb[i];
      res = RotGeom::classify_quick(mk_rec<CROSS>(a, e), mk_rec<CROSS>(a, q), thr, true);
    }
    if constexpr (LEAN && !CROSS) {
      mk_stage(res == 1, (e << 16) | q, S.ebuf, &S.ecnt, kMkStageEdge, [&](bool ov) { mk_hit<false>(a, ov, q, e); });
    } else {
      mk_hit<CROSS>(a, res == 1, q, e);
    }
    Q.head = (Q.head + cnt) & 127; Q.count -= cnt;
    if constexpr (LEAN) {
      mk_stage(res >= 2, make_uint2(q, e), S.pbuf, &S.pcnt, kMkStagePend, [&](bool ov) { mk_defer(c, &c->n1, a.pend1, a.cap1, ov, q, e); });
      wave_sync();
    } else {
      Q1.push(res == 3, q, e);
      Q2.push(res == 2, q, e);
      wave_sync();
      if (Q2.count >= 64) drain2(64);
      if (Q1.count >= 64) drain1b(64);
    }
  };

  const int w0 = CROSS ? (cur >> 6) : 0;
  const int nwords = CROSS ? (cur < n ? ((n - 1) >> 6) - w0 + 1 : 0) : ((cn + 63) >> 6);
  const int nitems = nwords * kMkQB;
  for (int it = wg * NW + wv; it < nitems; it += nwg * NW) {
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
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) q0 = *mk_rec<CROSS>(a, qid);
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
      const bool act2 = e + kMkL < e1;
      bool pass = false, pass2 = false; uint32_t eid = 0, eid2 = 0;
      if (act) {
        const float4 en = ent[e];
        float4 en2 = en;
        if (act2) en2 = ent[e + kMkL];
        {
          eid = __float_as_uint(en.w);
          const float dx = qx - en.x, dy = qy - en.y, rs = en.z + qr;
          const float d2 = dx * dx + dy * dy;
          pass = (qnever || e < nbrute || !(d2 > rs * rs)) && (CROSS || eid < qid);
        }
        if (act2) {
          eid2 = __float_as_uint(en2.w);
          const float dx = qx - en2.x, dy = qy - en2.y, rs = en2.z + qr;
          const float d2 = dx * dx + dy * dy;
          pass2 = (qnever || e + kMkL < nbrute || !(d2 > rs * rs)) && (CROSS || eid2 < qid);
        }
        e += 2 * kMkL;
      }
      if (__ballot(pass)) {
        Q.push(pass, qid, eid);
        if (Q.count >= 64) drain(64);
      }
      if (__ballot(pass2)) {
        Q.push(pass2, qid, eid2);
        if (Q.count >= 64) drain(64);
      }
    }
  }
  if constexpr (LEAN) {
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
      if (d < a.ecap) stg_agent(a.edges + d, S.ebuf[i]);
    }
  } else {
    // ---- leftovers: pooled per workgroup, stage by stage, and drained by as few waves as it takes
    auto pool = [&](MkQueue& q, int slot) -> int {
      __syncthreads();
      if (tid == 0) S.pool_n[slot] = 0;
      __syncthreads();
      int off = 0;
      if (lane == 0 && q.count > 0) off = atomicAdd(&S.pool_n[slot], q.count);
      off = __builtin_amdgcn_readfirstlane(off);
      for (int i = lane; i < q.count; i += 64) { S.pa[off + i] = q.a[(q.head + i) & 127]; S.pb[off + i] = q.b[(q.head + i) & 127]; }
      q.head = 0; q.count = 0;
      __syncthreads();
      return S.pool_n[slot];
    };
    auto take = [&](MkQueue& q, int c0, int cnt) {              // 64 pooled entries back into this wave's (empty) queue
      wave_sync();
      if (lane < cnt) { q.a[lane] = S.pa[c0 + lane]; q.b[lane] = S.pb[c0 + lane]; }
      q.head = 0; q.count = cnt;
      wave_sync();
    };
    {
      const int t = pool(Q, 0);
      for (int c0 = wv * 64; c0 < t; c0 += NW * 64) { const int cnt = min(64, t - c0); take(Q, c0, cnt); drain(cnt); }
    }
    {
      // (a handful of undecided pairs skip the interval stage: the exact clip decides them anyway, one drain instead of two)
      __syncthreads();
      if (tid == 0) S.pool_n[3] = 0;
      __syncthreads();
      if (lane == 0 && Q1.count + Q2.count > 0) atomicAdd(&S.pool_n[3], Q1.count + Q2.count);
      __syncthreads();
      const bool skip = S.pool_n[3] <= 64;
      const int t = pool(Q1, 1);
      if (!skip) {
        for (int c0 = wv * 64; c0 < t; c0 += NW * 64) { const int cnt = min(64, t - c0); take(Q1, c0, cnt); drain1b(cnt); }
      } else if (wv == 0 && t > 0) {
        // straight into wave 0's exact queue (t + its own leftovers <= 64 entries)
        wave_sync();
        const uint32_t va = lane < t ? S.pa[lane] : 0u, vb = lane < t ? S.pb[lane] : 0u;
        Q2.push(lane < t, va, vb);
        wave_sync();
      }
    }
    {
      const int t = pool(Q2, 2);
      for (int c0 = wv * 64; c0 < t; c0 += NW * 64) { const int cnt = min(64, t - c0); take(Q2, c0, cnt); drain2(cnt); }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ the interval and the exact clip on the pending list (dense: one lane per pair)
struct MkLdsDecideWave { float scr[RotGeom::SCR * 64]; uint32_t q2a[128], q2b[128]; };
template <bool CROSS>
__device__ __forceinline__ void mk_decide_phase(const MkArgs& a, MkLdsDecideWave& W) {
  MkCtl* c = a.ctl;
  const int lane = threadIdx.x & 63;
  int n1 = c->n1; if (n1 > a.cap1) n1 = a.cap1;
  const int gw = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nw = (int)((gridDim.x * blockDim.x) >> 6);
  MkQueue Q2{W.q2a, W.q2b, 0, 0};
  auto drain2 = [&](int cnt) {
    wave_sync();
    bool h = false; uint32_t q = 0, e = 0;
    if (lane < cnt) {
      const int i = (Q2.head + lane) & 127;
      q = W.q2a[i]; e = W.q2b[i];
      h = RotGeom::hit_exact(mk_rec<CROSS>(a, e), mk_rec<CROSS>(a, q), a.thr, W.scr + lane);
    }
    mk_hit<CROSS>(a, h, q, e);
    Q2.head = (Q2.head + cnt) & 127; Q2.count -= cnt;
    wave_sync();
  };
  for (int base = gw * 64; base < n1; base += nw * 64) {
    const int i = base + lane;
    int res = 0; uint2 p = make_uint2(0u, 0u);
    if (i < n1) {
      p = a.pend1[i];
      res = RotGeom::classify_full(mk_rec<CROSS>(a, p.y), mk_rec<CROSS>(a, p.x), a.thr);
    }
    mk_hit<CROSS>(a, res == 1, p.x, p.y);
    Q2.push(res == 2, p.x, p.y);
    if (Q2.count >= 64) drain2(64);
  }
  if (Q2.count > 0) drain2(Q2.count);
}

// ------------------------------------------------------------------ C: resolve the chunk, append its kept rows, build their table (ONE workgroup)
__device__ void mk_resolve_phase(const MkArgs& a, uint8_t* smem, size_t smem_bytes, int* s_i) {
  const int tid = threadIdx.x;
  MkCtl* c = a.ctl;
  const int cn = c->cn, kept_before = c->kept, cur = c->cur, step = c->step, cap = c->cap;
  const float dmax2 = c->dmax2;
  const int E = ldg_agent(a.nedges);
  NmsArgs r{};
  r.rec = a.rec; r.order = a.order; r.keep_cnt = a.keep_cnt; r.keep_out = a.keep_out; r.rows = a.rows; r.nrows = a.nrows;
  r.edges = a.edges; r.nedges = a.nedges; r.ecap = a.ecap; r.n = a.n; r.nseg = 1; r.capmax = a.capmax; r.max_keep = 0; r.lpt = 0;
  __syncthreads();
  const int total = nms_resolve(r, 0, 0, 0, cn, kept_before, a.cidx, smem, smem_bytes, s_i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const bool last = cur >= a.n;                         // nothing behind the chunk: no cross phase, the call is complete
  if (!last && total > 0) {
    MkLdsSelect& B = *reinterpret_cast<MkLdsSelect*>(smem);
    for (int k = tid; k < total; k += kMkThreads) B.list[k] = ldg_agent(a.rows + kept_before + k);
    __syncthreads();
    mk_build_table<false>(a.rec, B.list, total, dmax2, &c->gr, a.ent_r, a.start_r, B.b);
  }
  if (tid == 0) {
    c->kept = kept_before + total;
    c->nrow = last ? 0 : total;
    c->step = step + 1;
    c->stage = 3; c->n1 = 0;
    // the next chunk: twice as large, four times after a sparse one (few conflicts inside the chunk: the probes cost next to nothing)
    long long nc = (long long)cap * ((4LL * E <= cn) ? 4 : 2);
    c->cap = (int)(nc > a.capmax ? a.capmax : nc);
    if (last) mk_finish(a, c, kept_before + total, step + 1);
  }
}

constexpr size_t kMkSerialLds = 128 * 1024;   // dynamic LDS of the kernels that run a serial phase (resolve: the edge list in LDS)

// Every kernel of a step first looks at the control block: the call may be complete (done), the pending list may have overflowed
// (bail: the tail kernel takes over), or the step's turn may not have come (stage).
__global__ __launch_bounds__(kMkThreads) void k_mk_select(MkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mk_smem[];
  const MkCtl* c = a.ctl;
  if (c->done || c->bail || c->stage != 0) return;
  mk_select_phase(a, *reinterpret_cast<MkLdsSelect*>(mk_smem));
}
template <bool CROSS>
__global__ __launch_bounds__(kMkProbeThreads) void k_mk_probe(MkArgs a) {
  __shared__ MkLdsProbe<true, kMkProbeWaves> S;
  const MkCtl* c = a.ctl;
  if (c->done || c->bail || c->stage != (CROSS ? 3 : 1)) return;
  if (CROSS && c->nrow == 0) return;
  mk_probe_phase<CROSS, true, kMkProbeWaves>(a, S, (int)blockIdx.x, (int)gridDim.x);
}
// The decide kernel of a pair phase: the interval and the exact clip for the pairs its probe kernel left undecided, on every
// workgroup; the workgroup that is through LAST then runs the serial phase behind it -- resolve (+ the rows' table) after "pairs",
// select (+ the next chunk's table) after "cross".  Nobody waits for anybody: no co-residency assumption.
template <bool CROSS>
__global__ __launch_bounds__(kMkThreads) void k_mk_decide(MkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mk_smem[];
  __shared__ int s_i[16];
  __shared__ int s_last;
  MkCtl* c = a.ctl;
  if (c->done || c->bail || c->stage != (CROSS ? 3 : 1)) return;
  if (c->n1 > 0) mk_decide_phase<CROSS>(a, reinterpret_cast<MkLdsDecideWave*>(mk_smem)[threadIdx.x >> 6]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this workgroup's edges / kills are out (write-through stores, atomics)
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(&c->ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (int)gridDim.x - 1) ? 1 : 0;
    if (s_last) { stg_agent(&c->ticket, 0); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
  }
  __syncthreads();
  if (!s_last) return;
  if (ldg_agent(&c->bail)) return;                            // (raised by another workgroup of this launch: the tail kernel takes over)
  if constexpr (CROSS) mk_select_phase(a, *reinterpret_cast<MkLdsSelect*>(mk_smem));
  else mk_resolve_phase(a, mk_smem, kMkSerialLds, s_i);
}

// The tail: whatever the enqueued steps left undone -- too few steps for this data, or a pending list that overflowed -- is
// finished by ONE workgroup that runs the same phases in a loop, all decision stages in place (no co-residency, no barrier
// between workgroups; slow, and only reached when the caller's step estimate was too low: the next call of the thread knows
// better, MkArgs::hint_host).
__global__ __launch_bounds__(kMkThreads) void k_mk_tail(MkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mk_smem[];
  __shared__ int s_i[16];
  MkCtl* c = a.ctl;
  if (ldg_agent(&c->done)) return;
  using ProbeLds = MkLdsProbe<false, kMkWaves>;
  // where the phase kernels stopped: a selected chunk whose pairs are incomplete, or rows whose cross phase is (only after a bail)
  bool need_pairs = ldg_agent(&c->stage) == 1;
  bool need_cross = ldg_agent(&c->stage) == 3 && ldg_agent(&c->bail) != 0;
  for (int guard = 0; guard < (1 << 22); guard++) {
    __syncthreads();
    if (!need_pairs && !need_cross) {
      mk_select_phase(a, *reinterpret_cast<MkLdsSelect*>(mk_smem));
      __threadfence(); __syncthreads();
      if (ldg_agent(&c->done)) return;
      need_pairs = true;
    }
    if (need_pairs) {
      if (threadIdx.x == 0) stg_agent(a.nedges, 0);
      __threadfence(); __syncthreads();
      mk_probe_phase<false, false, kMkWaves>(a, *reinterpret_cast<ProbeLds*>(mk_smem), 0, 1);
      __threadfence(); __syncthreads();
      mk_resolve_phase(a, mk_smem, kMkSerialLds, s_i);
      __threadfence(); __syncthreads();
      if (ldg_agent(&c->done)) return;
      need_pairs = false; need_cross = true;
    }
    if (need_cross) {
      if (ldg_agent(&c->nrow) > 0) mk_probe_phase<true, false, kMkWaves>(a, *reinterpret_cast<ProbeLds*>(mk_smem), 0, 1);
      __threadfence(); __syncthreads();
      need_cross = false;
    }
  }
}

}  // namespace obb
