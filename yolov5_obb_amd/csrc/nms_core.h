// Lazy chunked greedy NMS ("LC-NMS") for gfx950 -- device side.
//
// What it computes: exactly the kept set (and order) of the reference's greedy
// NMS -- sort by score, a box is dropped iff an earlier *kept* box has
// IoU(kept, box) > thr   (nms_rotated_cuda.cu:60,116-128; poly_nms_cuda.cu:187,242-254).
//
// How (MI355X-first, not the reference's N x N/64 bitmask + host scan):
//   boxes are processed in score order in chunks of C (<= 4096) boxes.
//   A1  k_chunk_pairs   grid of wave-sized workgroups, one 64x64 tile of the
//                       chunk's upper triangle each: conservative reject in
//                       registers, survivors compacted through an LDS queue
//                       (wave ballot + popcount prefix) so the expensive clip
//                       always runs on 64 busy lanes; pairs with IoU > thr
//                       are appended to a per-segment edge list.
//   A2  k_chunk_resolve one workgroup per segment: lexicographically-first
//                       maximal independent set over the edge list by parallel
//                       rounds (== the sequential greedy scan), ordered
//                       compaction of the kept boxes, output write.
//   B   k_cross         only the chunk's *kept* rows are tested against the
//                       still-alive later boxes (lazy: suppressed boxes never
//                       generate work; a column stops as soon as it dies).
//   Work is O(kept x N) instead of N^2, memory is O(N): the N x N/64 mask of
//   the reference (1.25 GB at N = 100k) is never materialised, nothing is
//   copied to the host, and `max_keep` (the caller's max_det) stops a segment
//   early.  Segments (= images of a batch) run side by side in gridDim.y.
#pragma once
#include <hip/hip_runtime.h>
#include "obb_device.h"
#include "geom.h"

namespace obb {

struct NmsArgs {
  const float* feat;         // SoA [NF][n]   (sorted order)
  const uint32_t* order;     // sorted position -> original index
  uint8_t* dead;             // [n] 1 = suppressed / invalid
  const int* seg_begin;      // [nseg+1] sorted positions
  int* keep_cnt;             // [nseg]
  int64_t* keep_out;         // [n] segment s writes at seg_begin[s]...
  uint32_t* rows;            // [nseg][C] kept rows of the current chunk (sorted positions)
  int* nrows;                // [nseg]
  uint32_t* edges;           // [nseg][ecap]
  int* nedges;               // [nseg]
  long long ecap;
  int n;
  int C;
  int max_keep;              // 0 = unlimited
  float thr;
  int cull;                  // 1: conservative rejects allowed (thr >= 0)
};

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

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

// ------------------------------------------------------------------ A1
template <class G>
__global__ __launch_bounds__(64) void k_chunk_pairs(NmsArgs a, int step) {
  __shared__ float rowf[G::NF * 64];
  __shared__ float colf[G::NF * 64];
  __shared__ float scr[G::SCR * 64];
  __shared__ uint32_t qbuf[128];

  const int g = blockIdx.y, lane = threadIdx.x;
  const int sb = a.seg_begin[g], se = a.seg_begin[g + 1];
  const int b = sb + step * a.C;
  if (b >= se) return;
  if (a.max_keep > 0 && a.keep_cnt[g] >= a.max_keep) return;
  const int e = min(b + a.C, se);
  const int nb = (e - b + 63) >> 6;
  const int nbmax = a.C >> 6;
  const int rb = blockIdx.x / nbmax, cb = blockIdx.x % nbmax;
  if (rb > cb || cb >= nb) return;

  const int r = b + rb * 64 + lane, c = b + cb * 64 + lane;
  const bool rvalid = r < e, cvalid = c < e;
#pragma unroll
  for (int k = 0; k < G::NF; k++) {
    rowf[k * 64 + lane] = rvalid ? a.feat[(size_t)k * a.n + r] : 0.f;
    colf[k * 64 + lane] = cvalid ? a.feat[(size_t)k * a.n + c] : 0.f;
  }
  const bool ralive = rvalid && !a.dead[r];
  const bool calive = cvalid && !a.dead[c];
  unsigned long long cmask = __ballot(calive);
  if (__ballot(ralive) == 0ull || cmask == 0ull) return;
  __syncthreads();

  const typename G::Feat R = G::load(rowf, lane);
  PairQueue Q{qbuf, 0, 0};
  uint32_t* edges = a.edges + (size_t)g * a.ecap;
  const bool diag = rb == cb;

  auto drain = [&](int cnt) {   // wave-uniform cnt <= 64
    __syncthreads();
    const bool valid = lane < cnt;
    bool hit = false;
    uint32_t packed = 0;
    if (valid) {
      uint32_t it = qbuf[(Q.head + lane) & 127];
      int rr = it >> 8, cc = it & 255;
      typename G::Feat A = G::load(rowf, rr), B = G::load(colf, cc);
      float v = G::iou(A, B, scr + lane);
      hit = v > a.thr;
      packed = ((uint32_t)(rb * 64 + rr) << 16) | (uint32_t)(cb * 64 + cc);
    }
    unsigned long long hm = __ballot(hit);
    if (hm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&a.nedges[g], __popcll(hm));
      base = __shfl(base, 0);
      if (hit) {
        long long pos = (long long)base + __popcll(hm & lanemask_lt());
        if (pos < a.ecap) edges[pos] = packed;
      }
    }
    Q.head = (Q.head + cnt) & 127;
    Q.count -= cnt;
    __syncthreads();
  };

  while (cmask) {
    const int cc = __builtin_ctzll(cmask);
    cmask &= cmask - 1;
    const typename G::Feat Cc = G::load(colf, cc);
    bool pass = ralive && (!diag || cc > lane);
    if (pass && a.cull) pass = !G::reject(R, Cc, a.thr);
    Q.push(pass, ((uint32_t)lane << 8) | (uint32_t)cc);
    if (Q.count >= 64) drain(64);
  }
  if (Q.count > 0) drain(Q.count);
}

// ------------------------------------------------------------------ A2
// One workgroup (1024 threads) per segment.
__global__ __launch_bounds__(1024) void k_chunk_resolve(NmsArgs a, int step) {
  extern __shared__ uint8_t smem[];   // state[C] | blocked[C]
  __shared__ int s_remain, s_wave_tot[16], s_total;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int sb = a.seg_begin[g], se = a.seg_begin[g + 1];
  const int b = sb + step * a.C;
  const bool done = (b >= se) || (a.max_keep > 0 && a.keep_cnt[g] >= a.max_keep);
  if (done) {
    if (tid == 0) { a.nrows[g] = 0; a.nedges[g] = 0; }
    return;
  }
  const int e = min(b + a.C, se);
  const int cn = e - b;
  uint8_t* state = smem;          // 0 undecided, 1 kept, 2 dead
  uint8_t* blocked = smem + a.C;
  for (int j = tid; j < a.C; j += 1024) {
    state[j] = (j < cn) ? (a.dead[b + j] ? 2 : 0) : 2;
    blocked[j] = 0;
  }
  long long E = a.nedges[g];
  if (E > a.ecap) E = a.ecap;     // cannot happen: ecap is the worst case C*(C-1)/2
  const uint32_t* edges = a.edges + (size_t)g * a.ecap;
  __syncthreads();

  for (int round = 0;; round++) {
    if (round > 0) {
      for (long long k = tid; k < E; k += 1024) {
        uint32_t ed = edges[k];
        int i = ed >> 16, j = ed & 0xffff;
        if (state[i] == 1 && state[j] == 0) state[j] = 2;
      }
      __syncthreads();
    }
    for (long long k = tid; k < E; k += 1024) {
      uint32_t ed = edges[k];
      int i = ed >> 16, j = ed & 0xffff;
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

  // ordered compaction of the kept boxes: thread t owns the contiguous run
  // [t*per, (t+1)*per) of chunk positions
  const int per = (a.C + 1023) / 1024;
  int mine = 0;
  for (int q = 0; q < per; q++) {
    int j = tid * per + q;
    if (j < cn && state[j] == 1) mine++;
  }
  // wave inclusive scan
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
  uint32_t* rows = a.rows + (size_t)g * a.C;
  for (int q = 0; q < per; q++) {
    int j = tid * per + q;
    if (j < cn && state[j] == 1) {
      rows[rank] = (uint32_t)(b + j);
      long long o = (long long)cnt0 + rank;
      if (a.max_keep <= 0 || o < a.max_keep) a.keep_out[(size_t)sb + o] = (int64_t)a.order[b + j];
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
// One wave per 64 later columns; rows = kept boxes of the chunk just resolved.
template <class G>
__global__ __launch_bounds__(64) void k_cross(NmsArgs a, int step) {
  __shared__ float rowf[G::NF * 64];
  __shared__ float colf[G::NF * 64];
  __shared__ float scr[G::SCR * 64];
  __shared__ uint32_t qbuf[128];
  __shared__ uint8_t cdead[64];

  const int g = blockIdx.y, lane = threadIdx.x;
  const int sb = a.seg_begin[g], se = a.seg_begin[g + 1];
  const int b = sb + step * a.C;
  const int e = b + a.C;              // first column after the chunk
  const int c = e + blockIdx.x * 64 + lane;
  if (e + (int)blockIdx.x * 64 >= se) return;
  const int nr = a.nrows[g];
  if (nr == 0) return;
  if (a.max_keep > 0 && a.keep_cnt[g] >= a.max_keep) return;
  const bool cvalid = c < se;
  bool alive = cvalid && !a.dead[c];
  if (__ballot(alive) == 0ull) return;
#pragma unroll
  for (int k = 0; k < G::NF; k++) colf[k * 64 + lane] = cvalid ? a.feat[(size_t)k * a.n + c] : 0.f;
  cdead[lane] = alive ? 0 : 1;
  __syncthreads();
  const typename G::Feat Cc = G::load(colf, lane);
  const uint32_t* rows = a.rows + (size_t)g * a.C;
  PairQueue Q{qbuf, 0, 0};

  auto drain = [&](int cnt) {
    __syncthreads();
    if (lane < cnt) {
      uint32_t it = qbuf[(Q.head + lane) & 127];
      int rr = it >> 8, cc = it & 255;
      if (!cdead[cc]) {
        typename G::Feat A = G::load(rowf, rr), B = G::load(colf, cc);
        float v = G::iou(A, B, scr + lane);
        if (v > a.thr) cdead[cc] = 1;
      }
    }
    Q.head = (Q.head + cnt) & 127;
    Q.count -= cnt;
    __syncthreads();
  };

  for (int r0 = 0; r0 < nr; r0 += 64) {
    const int nrb = min(64, nr - r0);
    if (Q.count > 0) drain(Q.count);     // the queue refers to the previous row tile
    __syncthreads();
    {
      const bool rv = lane < nrb;
      const uint32_t rp = rv ? rows[r0 + lane] : 0u;
#pragma unroll
      for (int k = 0; k < G::NF; k++) rowf[k * 64 + lane] = rv ? a.feat[(size_t)k * a.n + rp] : 0.f;
    }
    __syncthreads();
    alive = alive && !cdead[lane];
    if (__ballot(alive) == 0ull) break;
    for (int rr = 0; rr < nrb; rr++) {
      const typename G::Feat Rr = G::load(rowf, rr);
      bool pass = alive;
      if (pass && a.cull) pass = !G::reject(Rr, Cc, a.thr);
      Q.push(pass, ((uint32_t)rr << 8) | (uint32_t)lane);
      if (Q.count >= 64) {
        drain(64);
        alive = alive && !cdead[lane];
      }
    }
  }
  if (Q.count > 0) drain(Q.count);
  if (cvalid && cdead[lane] && !a.dead[c]) a.dead[c] = 1;
}

}  // namespace obb
