// Lazy chunked greedy NMS ("LC-NMS") for gfx950 -- device side.
//
// What it computes: exactly the kept set (and order) of the reference's greedy
// NMS -- sort by score, a box is dropped iff an earlier *kept* box has
// IoU(kept, box) > thr   (nms_rotated_cuda.cu:60,116-128; poly_nms_cuda.cu:187,242-254).
//
// How (MI355X-first, not the reference's N x N/64 bitmask + host scan):
//   per step, per segment (= image):
//   S   k_select_chunk  the next `cap` still-alive boxes in score order become the chunk
//                       (ordered compaction of the alive flags by one workgroup).
//   A1  k_chunk_pairs   64x64 tiles of the chunk's upper triangle, one wave each:
//                       columns in registers, rows as 16-byte LDS broadcasts, a ~13-op
//                       circle test in the hot loop; survivors are compacted through an
//                       LDS ring (wave ballot + popcount prefix) so that the expensive
//                       part (separating axes, area bound, exact clip) always runs on
//                       full waves; pairs with IoU > thr go to a per-segment edge list.
//   A2  k_chunk_resolve one workgroup per segment: lexicographically-first maximal
//                       independent set over the edge list by parallel rounds (== the
//                       sequential greedy scan), ordered compaction of the kept boxes,
//                       output write.
//   B   k_cross         only the chunk's *kept* rows are tested against the still-alive
//                       later boxes (lazy: a suppressed box never generates work again).
//   Work is O(kept x alive) instead of N^2, memory is O(N): the N x N/64 mask of the
//   reference (1.25 GB at N = 100k) is never materialised, nothing is copied to the
//   host, and `max_keep` (the caller's max_det) stops a segment early.  Segments run
//   side by side in gridDim.y; grids are fixed-size and stride over work items whose
//   count is only known on the device.
#pragma once
#include <hip/hip_runtime.h>
#include "obb_device.h"
#include "geom.h"

namespace obb {

struct NmsArgs {
  const float4* rec;         // [n][RECQ] AoS records in sorted order
  const uint32_t* order;     // sorted position -> original index
  uint8_t* dead;             // [n] 1 = suppressed / invalid
  const int* seg_begin;      // [nseg] first sorted position of the segment
  const int* seg_end;        // [nseg] one past the last position considered (top-k cap applied)
  int* cursor;               // [nseg] next position not yet placed in a chunk
  int* keep_cnt;             // [nseg]
  int64_t* keep_out;         // segment g writes at keep_out[seg_begin[g] + k]
  uint32_t* cidx;            // [nseg][capmax] positions of the current chunk
  int* ccount;               // [nseg]
  uint32_t* rows;            // [nseg][capmax] kept rows of the current chunk (positions)
  int* nrows;                // [nseg]
  uint32_t* edges;           // [nseg][ecap] (i << 16 | j), chunk-local indices, i < j
  int* nedges;               // [nseg]
  long long ecap;
  int n;
  int capmax;
  int max_keep;              // 0 = unlimited
  float thr;
  int cull;                  // 1: conservative rejects allowed (thr >= 0)
};

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// Ring queue of pending (row, col) pairs in LDS; all bookkeeping is wave-uniform.
struct PairQueue {
  uint32_t* q;  // LDS, 128 entries
  int head, count;
  __device__ __forceinline__ void push(bool pass, uint32_t item) {
    unsigned long long m = __ballot(pass);
    if (pass) q[(head + count + __popcll(m & lanemask_lt())) & 127] = item;
    count += __popcll(m);
  }
};

__device__ __forceinline__ bool seg_done(const NmsArgs& a, int g) {
  return a.max_keep > 0 && a.keep_cnt[g] >= a.max_keep;
}

// ------------------------------------------------------------------ S
// One workgroup (1024 threads) per segment: chunk = the first `cap` alive positions >= cursor.
__global__ __launch_bounds__(1024) void k_select_chunk(NmsArgs a, int cap) {
  __shared__ int s_wave[16];
  __shared__ int s_newcur;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int se = a.seg_end[g];
  const int cur = a.cursor[g];
  if (cur >= se || seg_done(a, g)) {
    if (tid == 0) a.ccount[g] = 0;
    return;
  }
  uint32_t* cidx = a.cidx + (size_t)g * a.capmax;
  if (tid == 0) s_newcur = se;
  int off = 0;
  for (int base = cur; base < se && off < cap; base += 4096) {
    const int p0 = base + tid * 4;
    bool al[4];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int pos = p0 + k;
      al[k] = (pos < se) && !a.dead[pos];
      cnt += al[k] ? 1 : 0;
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int v = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += v;
    }
    __syncthreads();                       // previous iteration's s_wave readers are done
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    int wpre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
      int t = s_wave[w];
      if (w < (tid >> 6)) wpre += t;
      tot += t;
    }
    int slot = off + wpre + incl - cnt;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (al[k]) {
        if (slot < cap) {
          cidx[slot] = (uint32_t)(p0 + k);
          if (slot == cap - 1) s_newcur = p0 + k + 1;
        }
        slot++;
      }
    }
    off += tot;
  }
  __syncthreads();
  if (tid == 0) {
    a.ccount[g] = off < cap ? off : cap;
    a.cursor[g] = s_newcur;
  }
}

// ------------------------------------------------------------------ A1
template <class G>
__global__ __launch_bounds__(64) void k_chunk_pairs(NmsArgs a) {
  __shared__ float4 rowq0[64];
  __shared__ uint32_t rowpos[64], colpos[64];
  __shared__ float scr[G::SCR * 64];
  __shared__ uint32_t qbuf[128];

  const int g = blockIdx.y, lane = threadIdx.x;
  const int cn = a.ccount[g];
  if (cn == 0) return;
  const int nb = (cn + 63) >> 6;
  const int items = nb * nb;
  const uint32_t* cidx = a.cidx + (size_t)g * a.capmax;
  uint32_t* edges = a.edges + (size_t)g * a.ecap;
  const bool cull = a.cull != 0;
  PairQueue Q{qbuf, 0, 0};

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int rb = item / nb, cb = item - rb * nb;
    if (rb > cb) continue;
    const int r = rb * 64 + lane, c = cb * 64 + lane;
    const bool rvalid = r < cn, cvalid = c < cn;
    const uint32_t rp = rvalid ? cidx[r] : 0u, cp = cvalid ? cidx[c] : 0u;
    __syncthreads();
    rowpos[lane] = rp; colpos[lane] = cp;
    rowq0[lane] = rvalid ? a.rec[(size_t)rp * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 cq = cvalid ? a.rec[(size_t)cp * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int nrow = min(64, cn - rb * 64);
    const bool diag = rb == cb;
    __syncthreads();

    auto drain = [&](int cnt) {   // wave-uniform cnt <= 64
      __syncthreads();
      bool hit = false;
      uint32_t packed = 0;
      if (lane < cnt) {
        const uint32_t it = qbuf[(Q.head + lane) & 127];
        const int rr = it >> 8, cc = it & 255;
        hit = G::hit(a.rec + (size_t)rowpos[rr] * G::RECQ, a.rec + (size_t)colpos[cc] * G::RECQ, a.thr, cull, scr + lane);
        packed = ((uint32_t)(rb * 64 + rr) << 16) | (uint32_t)(cb * 64 + cc);
      }
      const unsigned long long hm = __ballot(hit);
      if (hm) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&a.nedges[g], __popcll(hm));
        base = __shfl(base, 0);
        if (hit) {
          const long long pos = (long long)base + __popcll(hm & lanemask_lt());
          if (pos < a.ecap) edges[pos] = packed;
        }
      }
      Q.head = (Q.head + cnt) & 127;
      Q.count -= cnt;
      __syncthreads();
    };

    for (int rr = 0; rr < nrow; rr++) {
      const float4 rq = rowq0[rr];
      bool pass = cvalid && (!diag || lane > rr);
      if (pass && cull) pass = !G::cheap_reject(rq, cq);
      if (__ballot(pass)) {
        Q.push(pass, ((uint32_t)rr << 8) | (uint32_t)lane);
        if (Q.count >= 64) drain(64);
      }
    }
    if (Q.count > 0) drain(Q.count);
  }
}

// ------------------------------------------------------------------ A2
// One workgroup (1024 threads) per segment.
__global__ __launch_bounds__(1024) void k_chunk_resolve(NmsArgs a) {
  extern __shared__ uint8_t smem[];   // state[capmax] | blocked[capmax]
  __shared__ int s_remain, s_wave_tot[16], s_total;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int cn = a.ccount[g];
  if (cn == 0) {
    if (tid == 0) { a.nrows[g] = 0; a.nedges[g] = 0; }
    return;
  }
  uint8_t* state = smem;              // 0 undecided, 1 kept, 2 dead
  uint8_t* blocked = smem + a.capmax;
  for (int j = tid; j < a.capmax; j += 1024) { state[j] = (j < cn) ? 0 : 2; blocked[j] = 0; }
  long long E = a.nedges[g];
  if (E > a.ecap) E = a.ecap;         // cannot happen: ecap is the worst case capmax*(capmax-1)/2
  const uint32_t* edges = a.edges + (size_t)g * a.ecap;
  __syncthreads();

  for (int round = 0;; round++) {
    if (round > 0) {
      for (long long k = tid; k < E; k += 1024) {
        const uint32_t ed = edges[k];
        const int i = ed >> 16, j = ed & 0xffff;
        if (state[i] == 1 && state[j] == 0) state[j] = 2;
      }
      __syncthreads();
    }
    for (long long k = tid; k < E; k += 1024) {
      const uint32_t ed = edges[k];
      const int i = ed >> 16, j = ed & 0xffff;
      if (state[i] == 0 && state[j] == 0) blocked[j] = 1;
    }
    if (tid == 0) s_remain = 0;
    __syncthreads();
    bool rem = false;
    for (int j = tid; j < cn; j += 1024) {
      if (state[j] == 0) {
        if (blocked[j]) { rem = true; blocked[j] = 0; }
        else state[j] = 1;
      }
    }
    if (rem) s_remain = 1;
    __syncthreads();
    if (!s_remain) break;
    __syncthreads();
  }

  // ordered compaction of the kept boxes: thread t owns chunk positions [t*per, (t+1)*per)
  const int per = (a.capmax + 1023) / 1024;
  int mine = 0;
  for (int q = 0; q < per; q++) {
    const int j = tid * per + q;
    if (j < cn && state[j] == 1) mine++;
  }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int v = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += v;
  }
  if ((tid & 63) == 63) s_wave_tot[tid >> 6] = incl;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int w = 0; w < 16; w++) { int t = s_wave_tot[w]; s_wave_tot[w] = acc; acc += t; }
    s_total = acc;
  }
  __syncthreads();
  int rank = s_wave_tot[tid >> 6] + incl - mine;
  const int cnt0 = a.keep_cnt[g];
  const int sb = a.seg_begin[g];
  const uint32_t* cidx = a.cidx + (size_t)g * a.capmax;
  uint32_t* rows = a.rows + (size_t)g * a.capmax;
  for (int q = 0; q < per; q++) {
    const int j = tid * per + q;
    if (j < cn && state[j] == 1) {
      const uint32_t pos = cidx[j];
      rows[rank] = pos;
      const long long o = (long long)cnt0 + rank;
      if (a.max_keep <= 0 || o < a.max_keep) a.keep_out[(size_t)sb + o] = (int64_t)a.order[pos];
      rank++;
    }
  }
  __syncthreads();
  if (tid == 0) {
    a.nrows[g] = s_total;
    a.keep_cnt[g] = cnt0 + s_total;
    a.nedges[g] = 0;
  }
}

// ------------------------------------------------------------------ B
// Work item = (tile of 64 later positions, tile of 64 kept rows); one wave per item.
template <class G>
__global__ __launch_bounds__(64) void k_cross(NmsArgs a) {
  __shared__ float4 rowq0[64];
  __shared__ uint32_t rowpos[64];
  __shared__ float scr[G::SCR * 64];
  __shared__ uint32_t qbuf[128];
  __shared__ uint8_t cdead[64];

  const int g = blockIdx.y, lane = threadIdx.x;
  const int nr = a.nrows[g];
  if (nr == 0 || seg_done(a, g)) return;
  const int c0 = a.cursor[g], se = a.seg_end[g];
  if (c0 >= se) return;
  const int nct = (se - c0 + 63) >> 6, nrt = (nr + 63) >> 6;
  const long long items = (long long)nct * nrt;
  const uint32_t* rows = a.rows + (size_t)g * a.capmax;
  const bool cull = a.cull != 0;
  PairQueue Q{qbuf, 0, 0};

  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const int ct = (int)(item / nrt), rt = (int)(item - (long long)ct * nrt);
    const int cbase = c0 + ct * 64;
    const int c = cbase + lane;
    const bool cvalid = c < se;
    const bool alive0 = cvalid && !a.dead[c];
    if (__ballot(alive0) == 0ull) continue;
    const int rr0 = rt * 64;
    const int nrow = min(64, nr - rr0);
    __syncthreads();
    {
      const uint32_t rp = lane < nrow ? rows[rr0 + lane] : 0u;
      rowpos[lane] = rp;
      rowq0[lane] = lane < nrow ? a.rec[(size_t)rp * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
      cdead[lane] = alive0 ? 0 : 1;
    }
    const float4 cq = cvalid ? a.rec[(size_t)c * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    bool alive = alive0;
    __syncthreads();

    auto drain = [&](int cnt) {
      __syncthreads();
      if (lane < cnt) {
        const uint32_t it = qbuf[(Q.head + lane) & 127];
        const int rr = it >> 8, cc = it & 255;
        if (!cdead[cc]) {
          if (G::hit(a.rec + (size_t)rowpos[rr] * G::RECQ, a.rec + (size_t)(cbase + cc) * G::RECQ, a.thr, cull, scr + lane))
            cdead[cc] = 1;
        }
      }
      Q.head = (Q.head + cnt) & 127;
      Q.count -= cnt;
      __syncthreads();
    };

    for (int rr = 0; rr < nrow; rr++) {
      const float4 rq = rowq0[rr];
      bool pass = alive;
      if (pass && cull) pass = !G::cheap_reject(rq, cq);
      if (__ballot(pass)) {
        Q.push(pass, ((uint32_t)rr << 8) | (uint32_t)lane);
        if (Q.count >= 64) {
          drain(64);
          alive = alive && !cdead[lane];
        }
      }
    }
    if (Q.count > 0) drain(Q.count);
    if (alive0 && cdead[lane]) a.dead[c] = 1;
  }
}

}  // namespace obb
