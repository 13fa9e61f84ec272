// Lazy chunked greedy NMS ("LC-NMS") for gfx950 -- device side, ONE persistent launch per call.
//
// What it computes: exactly the kept set (and order) of the reference's greedy NMS -- sort by score, a box is
// dropped iff an earlier *kept* box has IoU(kept, box) > thr   (nms_rotated_cuda.cu:60,116-128;
// poly_nms_cuda.cu:187,242-254).
//
// How (MI355X-first, not the reference's N x N/64 bitmask + device->host copy + host scan):
//   * the boxes live in score order as 16-byte-quad records (geom.h); "still alive" is one BIT per position
//     (64 positions = one u64 = one wave ballot);
//   * per step and segment (= image):
//       S   select   the next `cap` alive positions become the chunk (popcount prefix over the bitmap);
//       A1  pairs    64x64 tiles of the chunk's upper triangle, one wave each: columns in registers, rows as
//                    16-byte LDS broadcasts, a ~13-op circle test in the hot loop; survivors are compacted
//                    through an LDS ring (ballot + popcount prefix) so that the expensive part (separating
//                    axes, area bound, exact clip) always runs on full waves; hits go to an edge list;
//       A2  resolve  lexicographically-first maximal independent set over the edge list by parallel rounds
//                    (== the sequential greedy scan), ordered compaction of the kept boxes, output;
//       B   cross    only the chunk's KEPT rows are tested against the still-alive later positions; kills are
//                    one atomicAnd per 64 positions.  A suppressed box never generates work again.
//     Work is O(kept x alive) instead of N^2, memory is O(N): the reference's N x N/64 mask (1.25 GB at
//     N = 100k) never exists and nothing is copied to the host.
//   * the whole step loop runs inside ONE launch: one 512-thread workgroup per CU, grouped into teams (one team
//     per segment, NB/nseg workgroups each).  The control state (cursor, kept count, chunk size) is a replicated
//     state machine: every workgroup of a team computes S itself from the shared bitmap (identical result, chunk
//     list in its own LDS), so no control word travels between workgroups.  A step needs two team barriers: after
//     A1 the workgroup that arrives LAST runs A2 while the others wait (no separate launch, no host round trip),
//     after B a plain barrier makes the kills visible.  Unlike a host-driven loop no step is ever issued for a
//     segment that is already finished.  What workgroups do exchange (bitmap words, edges, kept rows) goes through
//     agent-scope (sc1) stores / atomics and either sc1 loads or ONE agent acquire followed by plain loads
//     (cdna_hip_programming.md guideline 16); nothing depends on dispatch order or XCD placement; spins are
//     bounded and raise an abort flag instead of hanging the device.
#pragma once
#include <hip/hip_runtime.h>
#include "obb_device.h"
#include "geom.h"
#include "grid.h"

namespace obb {

typedef unsigned long long u64;

struct NmsArgs {
  const float4* rec;         // [n][RECQ] AoS records in sorted order (read-only in this kernel)
  const uint32_t* order;     // sorted position -> original index (NULL: keep_out receives the sorted position itself)
  u64* alive;                // [n/64 + 2] bit (p & 63) of word (p >> 6): still a candidate (only B clears bits)
  const int* seg_begin;      // [nseg] first sorted position of the segment
  const int* seg_end;        // [nseg] one past the last position considered (top-k cap applied)
  int* keep_cnt;             // [nseg] (output)
  int64_t* keep_out;         // segment g writes at keep_out[seg_begin[g] + k]
  uint32_t* rows;            // [n] kept rows (sorted positions) of segment g at rows[seg_begin[g] + k], k = kept index
  int* nrows;                // [nteams] rows kept in the team's current chunk
  uint32_t* edges;           // [nteams][ecap] (i << 16 | j), chunk-local indices, i < j
  int* nedges;               // [nteams]   (a team works on one segment at a time: scratch is per team)
  int* bar;                  // [nteams][2][64] arrive / go counters, one 256-byte line each
  int* abort_flag;           // [1] set when a spin gave up
  u64* prof;                 // optional [32]: wall-clock ticks (10 ns) per phase (development aid)
  const int4* plan;          // optional [gridDim.x] {segment, team, index in team, team size} per workgroup (k_plan_teams):
                             // workgroups in proportion to the segment sizes; NULL or plan[0].w == 0: static teams
  long long ecap;
  int n, nseg;
  int capmax, cap_first;
  int max_keep;              // 0 = unlimited
  int window;                // 0: a segment is one window; > 0 (used with max_keep): positions are opened window by window
  float thr;
  double thr64;              // QuadGeom64 only (the merge threshold is a Python float)
  int cull;                  // 1: conservative rejects allowed (thr >= 0)
  // Spatial index of the cross phase (grid.h, nms_cross_blocks); gmeta == NULL: none, the cross phase stays exhaustive.
  const GridMeta* gmeta;     // built before the launch: extent of the data, number of blocks, number of brute boxes, on / off
  const float4* gsorted;     // the boxes in cell order {x, y, r, sorted position}, every slot padded to 64 entries
  u64* calive;               // alive bitmap in cell order (padding: 0)
  const uint32_t* ulist;     // sorted positions of the boxes kept out of the cell order (brute)
};

// ---- cost model shared by the planner (k_plan_teams) and the workgroups that follow its plan
constexpr int kPlanInfoSlot = 1023;        // the plan buffer has kMaxTeams = 1024 entries, workgroups use the first NB <= #CUs
__host__ __device__ __forceinline__ long long plan_cost(long long sz, long long chunk) {
  if (sz <= 0) return 0;
  const long long c = sz < chunk ? sz : chunk;
  const long long nb = (c + 63) / 64;
  return nb * (nb + 1) / 2 * 16 + 48 + (sz > chunk ? (sz - chunk) / 16 : 0);
}

// ------------------------------------------------------------------ agent-scope accessors
template <typename T> __device__ __forceinline__ T ldg_agent(const T* p) {
  return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ void stg_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
// wave-uniform lane -> scalar register: the row box of the hot loop is broadcast with v_readlane (no LDS round trip)
__device__ __forceinline__ float rdlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float4 rdlane4(const float4& v, int l) {
  return make_float4(rdlane(v.x, l), rdlane(v.y, l), rdlane(v.z, l), rdlane(v.w, l));
}

constexpr int kNmsThreads = 512;
constexpr int kNmsWaves = kNmsThreads / 64;
constexpr unsigned kSpinLimit = 1u << 22;

// Team barriers on one monotonic arrive counter (every barrier of either kind adds T arrivals) and one go word.
struct TeamBar {
  int* arrive; int* go; int T; int epoch; int* abort_flag;
};
// spin until *word >= target; false when the spin gave up or another workgroup raised the abort flag
__device__ __forceinline__ bool spin_until(int* word, int target, int* abort_flag) {
  for (unsigned spins = 0; ldg_agent(word) < target; spins++) {
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 1023u) == 1023u && ldg_agent(abort_flag)) return false;
    if (spins > kSpinLimit) { stg_agent(abort_flag, 1); return false; }
  }
  return true;
}
// plain barrier: returns false on abort
__device__ __forceinline__ bool team_barrier(TeamBar& b, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave drains its write-through stores / atomics
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(b.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = spin_until(b.arrive, b.T * (b.epoch + 1), b.abort_flag) ? 0 : 1;
  }
  __syncthreads();
  b.epoch++;
  return *s_flag == 0;
}
// barrier with a serial section: true in the workgroup that arrived last (it runs the serial part, then serial_end)
__device__ __forceinline__ bool serial_begin(TeamBar& b, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(b.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == b.T * (b.epoch + 1) - 1) ? 1 : 0;
    // the last arriver reads what the others published in bulk: ONE agent acquire, then plain loads (guideline 16, R1)
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}
// Both sides end with ONE agent acquire: what the serial section published in bulk (the kept rows) is read with
// plain, cacheable loads afterwards (the resolver's own CU may hold lines of the previous step as well).
__device__ __forceinline__ void serial_end(TeamBar& b) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    stg_agent(b.go, b.epoch + 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  b.epoch++;
}
__device__ __forceinline__ bool serial_wait(TeamBar& b, int* s_flag) {
  if (threadIdx.x == 0) {
    *s_flag = spin_until(b.go, b.epoch + 1, b.abort_flag) ? 0 : 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  b.epoch++;
  return *s_flag == 0;
}

// ------------------------------------------------------------------ per-wave LDS scratch of the pair phases
template <class G>
struct WaveLds {
  uint32_t rowpos[64], colpos[64];
  float scr[G::SCR * 64];
  uint32_t qbuf[128];      // stage 1: pairs that passed the hot loop
  uint32_t qbuf1b[128];    // stage 1b: pairs the cheap tests could not decide (IoU interval)
  uint32_t qbuf2[128];     // stage 2: pairs the register-only classifier could not decide (exact clip)
  uint8_t q1bcol[128];     // cross phase: column lane of a stage-1b entry
  uint8_t q2col[128];      // cross phase: column lane of a stage-2 entry
  uint8_t cdead[64];
  uint8_t pad[64];
  uint32_t ranges[128];    // indexed cross: ranges of slab rows within reach of the block
  float4 align16[0];
};

// The two expensive decision stages are real functions (one body per geometry) shared by the pair phase and both forms
// of the cross phase.
template <class G, class TH>
__device__ __attribute__((noinline)) int nms_stage_full(const float4* ra, const float4* rb, TH thr) {
  return G::classify_full(ra, rb, thr);
}
template <class G, class TH>
__device__ __attribute__((noinline)) bool nms_stage_exact(const float4* ra, const float4* rb, TH thr, float* scr) {
  return G::hit_exact(ra, rb, thr, scr);
}

// Ring queue of pending (row, col) pairs in LDS; all bookkeeping is wave-uniform.
struct PairQueue {
  uint32_t* q;  // LDS, 128 entries
  int head, count;
  __device__ __forceinline__ void push(bool pass, uint32_t item) {
    const u64 m = __ballot(pass);
    if (pass) q[(head + count + __popcll(m & lanemask_lt())) & 127] = item;
    count += __popcll(m);
  }
};

// ------------------------------------------------------------------ S: the next chunk (every workgroup, identical result)
// chunk = the first `cap` alive positions in [cur, se); their positions go to this workgroup's LDS list; returns the
// chunk size and moves `cur` behind the last member.  Members keep their alive bit: nothing reads positions below
// the cursor again.
__device__ __forceinline__ int nms_select(const NmsArgs& a, int se, int& cur, int cap, uint32_t* cidx, int* s_i) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int w_first = cur >> 6, w_last = (se - 1) >> 6;
  int off = 0;
  __syncthreads();
  if (tid == 0) s_i[8] = se;
  for (int wbase = w_first; wbase <= w_last && off < cap; wbase += kNmsThreads) {
    const int w = wbase + tid;
    u64 m = 0ull;
    if (w <= w_last) {
      m = ldg_agent(a.alive + w);
      const long long lo = (long long)w * 64;
      if (lo < cur) m &= ~((1ull << (cur - lo)) - 1ull);              // positions below the cursor / of the previous segment
      if (lo + 64 > se) m &= (1ull << (se - lo)) - 1ull;
    }
    const int cnt = __popcll(m);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (lane >= d) incl += v;
    }
    __syncthreads();                       // previous iteration's readers of s_i are done
    if (lane == 63) s_i[wv] = incl;
    __syncthreads();
    int wpre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kNmsWaves; k++) { const int t = s_i[k]; if (k < wv) wpre += t; tot += t; }
    int slot = off + wpre + incl - cnt;
    while (m && slot < cap) {
      const int b = __builtin_ctzll(m);
      m &= m - 1;
      const uint32_t pos = (uint32_t)(w * 64 + b);
      cidx[slot] = pos;
      if (slot == cap - 1) s_i[8] = (int)pos + 1;
      slot++;
    }
    off += tot;
  }
  __syncthreads();
  cur = s_i[8];
  return off < cap ? off : cap;
}

// ------------------------------------------------------------------ A1: pairs inside the chunk (all waves of the team)
template <class G>
__device__ __forceinline__ void nms_pairs(const NmsArgs& a, int tm, int cn, const uint32_t* cidx, int tw, int ntw, WaveLds<G>& L) {
  const int lane = threadIdx.x & 63;
  const int nb = (cn + 63) >> 6;
  const int items = nb * nb;
  uint32_t* edges = a.edges + (size_t)tm * a.ecap;
  const bool cull = a.cull != 0;
  PairQueue Q{L.qbuf, 0, 0}, Q1{L.qbuf1b, 0, 0}, Q2{L.qbuf2, 0, 0};
  auto emit = [&](bool hit, uint32_t packed) {           // hits -> the segment's edge list
    const u64 hm = __ballot(hit);
    if (hm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&a.nedges[tm], __popcll(hm));
      base = __shfl(base, 0);
      if (hit) {
        const long long pos = (long long)base + __popcll(hm & lanemask_lt());
        if (pos < a.ecap) stg_agent(edges + pos, packed);
      }
    }
  };
  // stage 2: exact clip; entries are chunk-local (i << 16 | j), so the queue lives across tiles
  auto drain2 = [&](int cnt) {
    wave_sync();
    bool hit = false;
    uint32_t packed = 0;
    if (lane < cnt) {
      packed = L.qbuf2[(Q2.head + lane) & 127];
      const uint32_t pi = cidx[packed >> 16], pj = cidx[packed & 0xffff];
      hit = nms_stage_exact<G>(a.rec + (size_t)pi * G::RECQ, a.rec + (size_t)pj * G::RECQ, G::thr_of(a), L.scr + lane);
    }
    const u64 hm = __ballot(hit);
    if (hm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&a.nedges[tm], __popcll(hm));
      base = __shfl(base, 0);
      if (hit) {
        const long long pos = (long long)base + __popcll(hm & lanemask_lt());
        if (pos < a.ecap) stg_agent(edges + pos, packed);
      }
    }
    Q2.head = (Q2.head + cnt) & 127;
    Q2.count -= cnt;
    wave_sync();
  };
  // stage 1b: the IoU interval, on full waves only (entries chunk-local like stage 2)
  auto drain1b = [&](int cnt) {
    wave_sync();
    int res = 0;
    uint32_t packed = 0;
    if (lane < cnt) {
      packed = L.qbuf1b[(Q1.head + lane) & 127];
      const uint32_t pi = cidx[packed >> 16], pj = cidx[packed & 0xffff];
      res = nms_stage_full<G>(a.rec + (size_t)pi * G::RECQ, a.rec + (size_t)pj * G::RECQ, G::thr_of(a));
    }
    emit(res == 1, packed);
    Q1.head = (Q1.head + cnt) & 127;
    Q1.count -= cnt;
    Q2.push(res == 2, packed);
    wave_sync();
    if (Q2.count >= 64) drain2(64);
  };

  // small chunks: split every 64-row tile into 2 or 4 row slices so that all waves of the team have work
  const int tri = nb * (nb + 1) / 2;
  const int nsub = (tri < 4 * ntw) ? 4 : ((tri < 8 * ntw) ? 2 : 1);   // fewer than 4 (8) tiles per wave: quarter (half) tiles balance better
  const int rows_sub = 64 / nsub;
  const int items_sub = items * nsub;
  for (int it2 = tw; it2 < items_sub; it2 += ntw) {
    const int item = it2 / nsub, sub = it2 - item * nsub;
    const int rb = item / nb, cb = item - rb * nb;
    if (rb > cb) continue;
    const int r = rb * 64 + lane, c = cb * 64 + lane;
    const bool rvalid = r < cn, cvalid = c < cn;
    const uint32_t rp = rvalid ? cidx[r] : 0u, cp = cvalid ? cidx[c] : 0u;
    wave_sync();
    L.rowpos[lane] = rp; L.colpos[lane] = cp;
    const float4 myrow = rvalid ? a.rec[(size_t)rp * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 cq = cvalid ? a.rec[(size_t)cp * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int nrow = min(64, cn - rb * 64);
    const int rr_lo = sub * rows_sub, rr_hi = min(nrow, rr_lo + rows_sub);
    const bool diag = rb == cb;
    wave_sync();

    auto drain = [&](int cnt) {   // stage 1a, wave-uniform cnt <= 64
      wave_sync();
      int res = 0;
      uint32_t packed = 0;
      if (lane < cnt) {
        const uint32_t it = L.qbuf[(Q.head + lane) & 127];
        const int rr = it >> 8, cc = it & 255;
        res = G::classify_quick(a.rec + (size_t)L.rowpos[rr] * G::RECQ, a.rec + (size_t)L.colpos[cc] * G::RECQ, G::thr_of(a), cull);
        packed = ((uint32_t)(rb * 64 + rr) << 16) | (uint32_t)(cb * 64 + cc);
      }
      emit(res == 1, packed);
      Q.head = (Q.head + cnt) & 127;
      Q.count -= cnt;
      Q1.push(res == 3, packed);
      Q2.push(res == 2, packed);
      wave_sync();
      if (Q1.count >= 64) drain1b(64);
      if (Q2.count >= 64) drain2(64);
    };

#pragma unroll 2
    for (int rr = rr_lo; rr < rr_hi; rr++) {
      const float4 rq = rdlane4(myrow, rr);
      bool pass = cvalid && (!diag || lane > rr);
      if (pass && cull) pass = !G::cheap_reject(rq, cq);
      if (__ballot(pass)) {
        Q.push(pass, ((uint32_t)rr << 8) | (uint32_t)lane);
        if (Q.count >= 64) drain(64);
      }
    }
    if (Q.count > 0) drain(Q.count);
  }
  if (Q1.count > 0) drain1b(Q1.count);
  if (Q2.count > 0) drain2(Q2.count);
}

// ------------------------------------------------------------------ A2: resolve the chunk (serial section, 512 threads)
// Greedy NMS inside the chunk == the lexicographically-first maximal independent set of the conflict graph (edges
// i < j, i the higher score).  Parallel rounds: a node is kept as soon as none of its lower-index neighbours is still
// undecided; kept nodes kill their higher-index neighbours.  The edge list lives in LDS as one contiguous block per
// thread and PRUNES itself: an edge is dropped the moment its target is decided or its source is dead / has delivered
// its kill, so the rounds get cheaper geometrically.  A list too long for LDS is streamed read-only from global memory
// for the first rounds, until what is left of it fits.
// LDS (aliasing the wave scratch): state[capmax] | blocked[capmax] | edges[...]
// returns the number of kept boxes of the chunk (also published in nrows[g])
__device__ __forceinline__ int nms_resolve(const NmsArgs& a, int g, int tm, int cn, int kept_before, const uint32_t* cidx, uint8_t* smem, size_t smem_bytes,
                           int* s_i) {
  const int tid = threadIdx.x;
  uint8_t* state = smem;              // 0 undecided, 1 kept, 2 dead
  uint8_t* blocked = smem + a.capmax;
  uint32_t* ledges = reinterpret_cast<uint32_t*>(smem + 2 * (size_t)a.capmax);
  const long long lcap = ((long long)smem_bytes - 2LL * a.capmax) / 4;
  for (int j = tid; j < cn; j += kNmsThreads) { state[j] = 0; blocked[j] = 0; }
  long long E = ldg_agent(a.nedges + tm);
  if (E > a.ecap) E = a.ecap;         // cannot happen: ecap is the worst case capmax*(capmax-1)/2
  const uint32_t* edges = a.edges + (size_t)tm * a.ecap;   // plain loads: acquired in serial_begin
  // The list fits in LDS when every thread's share does (one contiguous block per thread, odd length: conflict-free banks).
  const int percap = (int)(((lcap / kNmsThreads) - 1) | 1);
  int per = (int)((E + kNmsThreads - 1) / kNmsThreads);
  per |= 1;
  bool lds_mode = per <= percap;
  if (!lds_mode) per = percap;
  int mycnt = 0;
  uint32_t* mine_e = ledges + (size_t)tid * per;
  if (lds_mode) {
    // thread t takes edges t, t+512, ... (coalesced global reads, 8 in flight) into its own LDS block
    for (long long k0 = tid; k0 < E; k0 += 8 * kNmsThreads) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const long long k = k0 + (long long)u * kNmsThreads; v[u] = k < E ? edges[k] : 0u; }
#pragma unroll
      for (int u = 0; u < 8; u++) if (k0 + (long long)u * kNmsThreads < E) mine_e[mycnt++] = v[u];
    }
  }
  __syncthreads();
  u64 tp = 0;
  if (a.prof && tid == 0) { tp = wall_clock64(); atomicAdd(a.prof + 25, (u64)E); atomicMax(a.prof + 27, (u64)E); atomicAdd(a.prof + 28, (u64)cn); atomicAdd(a.prof + 26, (u64)(lds_mode ? 1 : 0)); }
  auto plap = [&](int slot) { if (a.prof && tid == 0) { const u64 t = wall_clock64(); atomicAdd(a.prof + slot, t - tp); tp = t; } };

  // A list that does not fit is first streamed READ-ONLY from global memory (16-byte plain loads, 8 in flight: the
  // list was published write-through and acquired in serial_begin, nothing writes it during the serial section):
  // the rounds work exactly as below but nothing is pruned.  Each such round counts the edges that still have two
  // undecided ends; these can only become fewer, so once every thread's count fits its LDS block the next round
  // copies the survivors into LDS and the self-pruning rounds take over.
  bool compact = false;
  // one edge (i < j) against the states read for it: true = both ends undecided, the edge stays and j waits for i
  auto decide = [&](uint32_t ed, uint8_t sj, uint8_t si) -> bool {
    if (sj != 0) return false;                       // target decided: the edge is done
    if (si == 1) { state[ed & 0xffff] = 2; return false; }   // kept source kills the target
    if (si == 2) return false;                       // dead source never matters again
    blocked[ed & 0xffff] = 1;
    return true;
  };
  for (int round = 0;; round++) {
    int remaining = 0;
    if (lds_mode) {
      // four edges per trip: their 4 + 8 LDS reads are issued together (the byte reads conflict on banks and would
      // otherwise serialise edge by edge); deciding on a slightly stale state only delays a decision by a round
      int w = 0, k = 0;
      for (; k + 4 <= mycnt; k += 4) {
        uint32_t ed[4]; uint8_t sj[4], si[4];
#pragma unroll
        for (int c = 0; c < 4; c++) ed[c] = mine_e[k + c];
#pragma unroll
        for (int c = 0; c < 4; c++) { sj[c] = state[ed[c] & 0xffff]; si[c] = state[ed[c] >> 16]; }
#pragma unroll
        for (int c = 0; c < 4; c++) if (decide(ed[c], sj[c], si[c])) mine_e[w++] = ed[c];
      }
      for (; k < mycnt; k++) {
        const uint32_t ed = mine_e[k];
        if (decide(ed, state[ed & 0xffff], state[ed >> 16])) mine_e[w++] = ed;
      }
      mycnt = w;
    } else {
      const uint4* e4 = reinterpret_cast<const uint4*>(edges);
      const long long nvec = (E + 3) >> 2;
      int w = 0;
      for (long long v0 = tid; v0 < nvec; v0 += 8 * kNmsThreads) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const long long k = v0 + (long long)u * kNmsThreads;
          v[u] = k < nvec ? e4[k] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const long long k = v0 + (long long)u * kNmsThreads;
          if (k >= nvec) break;
          const uint32_t e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          uint8_t sj[4], si[4];
#pragma unroll
          for (int c = 0; c < 4; c++) { sj[c] = state[e[c] & 0xffff]; si[c] = state[e[c] >> 16]; }   // (garbage past E: 16-bit indices, in range of the block)
#pragma unroll
          for (int c = 0; c < 4; c++) {
            if (4 * k + c >= E) break;
            if (decide(e[c], sj[c], si[c])) {
              remaining++;
              if (compact) mine_e[w++] = e[c];
            }
          }
        }
      }
      if (compact) mycnt = w;
    }
    __syncthreads();
    bool rem = false;
    for (int j = tid; j < cn; j += kNmsThreads) {
      if (state[j] == 0) {
        if (blocked[j]) { rem = true; blocked[j] = 0; }
        else state[j] = 1;
      }
    }
    const int any = __syncthreads_or(rem ? 1 : 0);     // barrier + "somebody is still undecided" in one
    if (!any) { if (a.prof && tid == 0) atomicAdd(a.prof + 11, (u64)(round + 1)); break; }
    if (!lds_mode) {
      if (compact) lds_mode = true;                                        // the survivors are in LDS now
      else compact = __syncthreads_or(remaining > per ? 1 : 0) == 0;       // block-uniform: next round copies them
    }
    if (round == 0) plap(12);
  }

  plap(13);
  // ordered compaction of the kept boxes: thread t owns chunk positions [t*per_n, (t+1)*per_n)
  const int per_n = (cn + kNmsThreads - 1) / kNmsThreads;
  int mine = 0;
  for (int q = 0; q < per_n; q++) {
    const int j = tid * per_n + q;
    if (j < cn && state[j] == 1) mine++;
  }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += v;
  }
  if ((tid & 63) == 63) s_i[tid >> 6] = incl;
  __syncthreads();
  int wpre = 0, total = 0;
#pragma unroll
  for (int k = 0; k < kNmsWaves; k++) { const int t = s_i[k]; if (k < (tid >> 6)) wpre += t; total += t; }
  int rank = wpre + incl - mine;
  const int sb = a.seg_begin[g];
  uint32_t* rows = a.rows + (size_t)sb + kept_before;         // appended to the segment's kept-row list
  for (int q = 0; q < per_n; q++) {
    const int j = tid * per_n + q;
    if (j < cn && state[j] == 1) {
      const uint32_t pos = cidx[j];
      stg_agent(rows + rank, pos);
      const long long o = (long long)kept_before + rank;
      if (a.max_keep <= 0 || o < a.max_keep) a.keep_out[(size_t)sb + o] = a.order ? (int64_t)a.order[pos] : (int64_t)pos;
      rank++;
    }
  }
  if (tid == 0) {
    stg_agent(a.nrows + tm, total);
    stg_agent(a.nedges + tm, 0);
    stg_agent(a.keep_cnt + g, kept_before + total);   // write-through: resolvers of different steps sit on different XCDs
  }
  __syncthreads();
  plap(14);
  return total;
}

// ------------------------------------------------------------------ B: kept rows x still-alive later positions
template <class G>
__device__ __forceinline__ void nms_cross(const NmsArgs& a, const uint32_t* rows, int nr, int c0, int se, int tw, int ntw, WaveLds<G>& L) {
  const int lane = threadIdx.x & 63;
  const int w0 = c0 >> 6, w1 = (se - 1) >> 6;
  const int ncw = w1 - w0 + 1, nrt = (nr + 63) >> 6;
  // one wave per column word; the row tiles of a word are split over several waves only when there are fewer
  // column words than waves
  int rgn = ntw / ncw;
  if (rgn < 1) rgn = 1;
  if (rgn > nrt) rgn = nrt;
  const int rt_per = (nrt + rgn - 1) / rgn;
  const long long items = (long long)ncw * rgn;
  // rows: plain loads -- published write-through by the resolver, acquired after the serial section
  const bool cull = a.cull != 0;
  PairQueue Q{L.qbuf, 0, 0}, Q1{L.qbuf1b, 0, 0}, Q2{L.qbuf2, 0, 0};
  const bool cprof = a.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  u64 ct0 = 0, c_loop = 0, c_d1 = 0, c_d2 = 0, c_n1 = 0, c_n2 = 0, c_items = 0;
  auto ctick = [&]() { if (cprof) ct0 = wall_clock64(); };
  auto ctock = [&](u64& acc) { if (cprof) acc += wall_clock64() - ct0; };

  for (long long item = tw; item < items; item += ntw) {
    c_items++;
    const int cw = (int)(item / rgn), rgi = (int)(item - (long long)cw * rgn);
    const int w = w0 + cw;
    const int cbase = w * 64;
    const int c = cbase + lane;
    const int rt_lo = rgi * rt_per, rt_hi = min(nrt, rt_lo + rt_per);
    if (rt_lo >= rt_hi) continue;
    // first row tile's loads are issued before the (slow, write-through) bitmap word arrives
    uint32_t rp0 = (rt_lo * 64 + lane < nr) ? rows[rt_lo * 64 + lane] : 0u;
    u64 m = ldg_agent(a.alive + w);
    if (cbase < c0) m &= ~((1ull << (c0 - cbase)) - 1ull);
    if (cbase + 64 > se) m &= (1ull << (se - cbase)) - 1ull;
    if (m == 0ull) continue;
    const bool alive0 = (m >> lane) & 1ull;
    const float4 cq = alive0 ? a.rec[(size_t)c * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rq0 = a.rec[(size_t)rp0 * G::RECQ];
    uint32_t rp1 = (rt_lo + 1 < rt_hi && (rt_lo + 1) * 64 + lane < nr) ? rows[(rt_lo + 1) * 64 + lane] : 0u;
    bool alive = alive0;
    wave_sync();
    L.cdead[lane] = alive0 ? 0 : 1;
    auto drain2 = [&](int cnt) {                   // stage 2: exact clip; entries carry the row position itself
      wave_sync();
      if (lane < cnt) {
        const int slot = (Q2.head + lane) & 127;
        const uint32_t rowp = L.qbuf2[slot];
        const int cc = L.q2col[slot];
        if (!L.cdead[cc]) {
          if (nms_stage_exact<G>(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)(cbase + cc) * G::RECQ, G::thr_of(a), L.scr + lane)) L.cdead[cc] = 1;
        }
      }
      Q2.head = (Q2.head + cnt) & 127;
      Q2.count -= cnt;
      wave_sync();
    };
    auto push2 = [&](bool undecided, uint32_t rowp, uint32_t cc) {   // (row position, column lane) into stage 2
      const u64 m2 = __ballot(undecided);
      if (undecided) {
        const int slot = (Q2.head + Q2.count + __popcll(m2 & lanemask_lt())) & 127;
        L.qbuf2[slot] = rowp;
        L.q2col[slot] = (uint8_t)cc;
      }
      Q2.count += __popcll(m2);
    };
    auto drain1b = [&](int cnt) {                  // stage 1b: the IoU interval, on full waves only
      wave_sync();
      int res = 0;
      uint32_t rowp = 0, cc = 0;
      if (lane < cnt) {
        const int slot = (Q1.head + lane) & 127;
        rowp = L.qbuf1b[slot];
        cc = L.q1bcol[slot];
        if (!L.cdead[cc]) {
          res = nms_stage_full<G>(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)(cbase + cc) * G::RECQ, G::thr_of(a));
          if (res == 1) L.cdead[cc] = 1;
        }
      }
      Q1.head = (Q1.head + cnt) & 127;
      Q1.count -= cnt;
      push2(res == 2, rowp, cc);
      wave_sync();
      if (Q2.count >= 64) drain2(64);
    };

    for (int rt = rt_lo; rt < rt_hi; rt++) {
      if (__ballot(alive) == 0ull) break;
      const int rr0 = rt * 64;
      const int nrow = min(64, nr - rr0);
      wave_sync();
      L.rowpos[lane] = rp0;
      const float4 myrow = rq0;
      // next tile's record and the positions of the tile after it travel while this tile is processed
      rp0 = rp1;
      if (rt + 1 < rt_hi) rq0 = a.rec[(size_t)rp1 * G::RECQ];
      rp1 = (rt + 2 < rt_hi && (rt + 2) * 64 + lane < nr) ? rows[(rt + 2) * 64 + lane] : 0u;
      wave_sync();

      auto drain = [&](int cnt) {                 // stage 1a: the cheap register-only tests
        wave_sync();
        int res = 0;
        uint32_t rowp = 0, cc = 0;
        if (lane < cnt) {
          const uint32_t it = L.qbuf[(Q.head + lane) & 127];
          const int rr = it >> 8;
          cc = it & 255;
          rowp = L.rowpos[rr];
          if (!L.cdead[cc]) {
            res = G::classify_quick(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)(cbase + cc) * G::RECQ, G::thr_of(a), cull);
            if (res == 1) L.cdead[cc] = 1;
          }
        }
        Q.head = (Q.head + cnt) & 127;
        Q.count -= cnt;
        {                                          // needs the interval: (row position, column lane) into stage 1b
          const u64 m1 = __ballot(res == 3);
          if (res == 3) {
            const int slot = (Q1.head + Q1.count + __popcll(m1 & lanemask_lt())) & 127;
            L.qbuf1b[slot] = rowp;
            L.q1bcol[slot] = (uint8_t)cc;
          }
          Q1.count += __popcll(m1);
        }
        push2(res == 2, rowp, cc);
        wave_sync();
        if (Q1.count >= 64) drain1b(64);
        if (Q2.count >= 64) drain2(64);
      };

      ctick();
      auto one_row = [&](int rr, bool pass) {
        if (__ballot(pass)) {
          Q.push(pass, ((uint32_t)rr << 8) | (uint32_t)lane);
          if (Q.count >= 64) {
            ctock(c_loop);
            ctick(); drain(64); ctock(c_d1); c_n1++;
            alive = alive && !L.cdead[lane];
            ctick();
          }
        }
      };
      // four rows per trip: their broadcasts and cheap tests are issued together and ONE ballot decides whether any of
      // them has a pair to queue (most rows of a kept set pass for no column of a given 64-column word); a column that a
      // drain kills in the middle of a group may still queue its later rows, which the drains skip
      int rr = 0;
      for (; rr + 4 <= nrow; rr += 4) {
        bool ps[4];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float4 rq = rdlane4(myrow, rr + k);
          ps[k] = alive && !(cull && G::cheap_reject(rq, cq));
          any = any || ps[k];
        }
        if (__ballot(any)) {
#pragma unroll
          for (int k = 0; k < 4; k++) one_row(rr + k, ps[k]);
        }
      }
      for (; rr < nrow; rr++) {
        const float4 rq = rdlane4(myrow, rr);
        one_row(rr, alive && !(cull && G::cheap_reject(rq, cq)));
      }
      ctock(c_loop);
      if (Q.count > 0) { ctick(); drain(Q.count); ctock(c_d1); c_n1++; alive = alive && !L.cdead[lane]; }
    }
    if (Q1.count > 0) { ctick(); drain1b(Q1.count); ctock(c_d1); c_n1++; alive = alive && !L.cdead[lane]; }
    if (Q2.count > 0) { ctick(); drain2(Q2.count); ctock(c_d2); c_n2++; alive = alive && !L.cdead[lane]; }
    const u64 kill = __ballot(alive0 && !alive);
    if (kill && lane == 0) {
      // RETURNING atomic whose result is consumed: the wave's vmcnt then covers the completed read-modify-write
      const u64 old = atomicAnd(a.alive + w, ~kill);
      asm volatile("; kill applied %0" ::"v"((unsigned)(old >> 32) ^ (unsigned)old));
    }
  }
  if (cprof) {
    a.prof[16] += c_loop; a.prof[17] += c_d1; a.prof[18] += c_d2; a.prof[19] += c_n1; a.prof[20] += c_n2; a.prof[21] += c_items;
  }
}

// ------------------------------------------------------------------ B': the cross phase through two spatial indices
// The exhaustive form above tests every kept row of a chunk against every still-alive later position: O(kept x alive)
// circle tests -- cheap with a few hundred kept boxes (S-clustered, K = 300), the whole run time with thousands (class
// offsets, K = 3000: the natural shape of BASELINE configs[3]) or tens of thousands (S-uniform).  Only pairs whose
// circumscribed circles touch can have IoU > 0.
//
// COLUMNS: the boxes of the call are stored once more in CELL order (grid.h: levels by circumradius, square cells of
// side 2 x the level's largest radius, slot = hash of (level, cx, cy); built before the launch: count, scan, scatter),
// every slot padded to a multiple of 64 entries, with an alive bitmap in the same order.  A wave takes one 64-entry block
// = boxes of similar size out of one cell: their bounding box is about one cell.
// ROWS: per slab of <= kSlabRows kept rows every workgroup builds its own copy of a small hashed grid over the rows'
// centres in LDS (the chunk list's 32 KB are free between resolve and the next select), same levels and hash.
// The wave then enumerates only the rows of the cells within reach of its block's bounding box -- a handful of contiguous
// LDS ranges (the slot hash is linear in cx) -- and tests each of them against its 64 columns exactly as the exhaustive
// form does: wave-uniform control flow, the row record an LDS broadcast, full lanes.
//
// Exactness: the indices decide which pairs get RotGeom::cheap_reject at all; a pair they skip is one that test would
// have rejected.  That holds for pairs of well-conditioned boxes (grid.h "brute" rule: the conditioning part of
// cheap_reject cannot fail anywhere inside the data's bounding box).  Brute rows sit in an extra slot that every block
// enumerates; brute columns are not in the cell order at all: they come as a plain list whose blocks enumerate every row.
// Both see the full cheap_reject.  tests/native/host_check_grid.cpp checks the window arithmetic on the CPU.
constexpr int kSlabRows = 1216;
constexpr int kSlabSlots = 2048;                   // power of two; slot kSlabSlots = the brute rows
constexpr int kSlabMinRows = 96;                   // fewer kept rows: the exhaustive form is cheaper than building the index
struct SlabLds {
  float4 q0[kSlabRows];                            // {x, y, r, short side^2} in slot order
  uint32_t pos[kSlabRows];                         // sorted position of the row
  uint32_t start[kSlabSlots + 2];                  // first entry of slot i; [kSlabSlots + 1] = number of rows in the slab
};

template <class G>
__device__ __forceinline__ void nms_cross_blocks(const NmsArgs& a, const GridPlan& gp, const uint32_t* rows, int nr, int c0, int se, int tw,
                                                 int ntw, WaveLds<G>& L, SlabLds& S, int* s_i) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool cull = a.cull != 0;
  constexpr uint32_t kMask = kSlabSlots - 1;
  const int nwords = a.gmeta->n_words, n_brute = a.gmeta->n_brute;
  const int nitems = nwords + ((n_brute + 63) >> 6);
  const float inf = __builtin_inff();
  const bool cprof = a.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  u64 ct0 = 0, c_enum = 0, c_drain = 0, c_build = 0, c_rng = 0, c_nd = 0, c_items = 0, c_rows = 0, c_whole = 0;
  auto ctick = [&]() { if (cprof) ct0 = wall_clock64(); };
  auto ctock = [&](u64& acc) { if (cprof) acc += wall_clock64() - ct0; };
  for (int r0 = 0; r0 < nr; r0 += kSlabRows) {
    const int ns = min(kSlabRows, nr - r0);
    // ---- build the rows' grid: every workgroup its own copy (identical content; the order inside a slot does not matter)
    ctick();
    __syncthreads();                               // the previous slab's readers / the chunk list's readers are done
    for (int i = tid; i < kSlabSlots + 2; i += kNmsThreads) S.start[i] = 0u;
    if (tid == 0) s_i[10] = 0;
    __syncthreads();
    constexpr int kPer = (kSlabRows + kNmsThreads - 1) / kNmsThreads;
    float4 rq[kPer];
    uint32_t rpos[kPer];
    int rslot[kPer];
    uint32_t lbits = 0u;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int j = tid + k * kNmsThreads;
      rslot[k] = -1;
      if (j < ns) {
        rpos[k] = rows[r0 + j];
        rq[k] = a.rec[(size_t)rpos[k] * G::RECQ];
        if (grid_is_brute(gp, rq[k].x, rq[k].y, rq[k].z, rq[k].w)) rslot[k] = kSlabSlots;
        else {
          const int lv = grid_level(gp, rq[k].z);
          const float inv = grid_level_inv_cell(gp, lv);
          const int cx = grid_cell(rq[k].x, gp.x0, inv, grid_last_cell(gp.xr, inv));
          const int cy = grid_cell(rq[k].y, gp.y0, inv, grid_last_cell(gp.yr, inv));
          rslot[k] = (int)grid_slot(lv, cx, cy, kMask);
          lbits |= 1u << lv;
        }
        atomicAdd(&S.start[rslot[k] + 1], 1u);
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lbits |= __shfl_xor(lbits, d);
    if (lane == 0 && lbits) atomicOr(&s_i[10], (int)lbits);
    __syncthreads();
    {                                              // exclusive prefix over start[1 .. kSlabSlots + 1], in place
      constexpr int kEach = (kSlabSlots + 1 + kNmsThreads - 1) / kNmsThreads;
      uint32_t v[kEach], sum = 0u;
#pragma unroll
      for (int k = 0; k < kEach; k++) {
        const int i = 1 + tid * kEach + k;
        v[k] = i <= kSlabSlots + 1 ? S.start[i] : 0u;
        sum += v[k];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
      if (lane == 63) s_i[wv] = (int)incl;
      __syncthreads();
      uint32_t run = incl - sum;
#pragma unroll
      for (int k = 0; k < kNmsWaves; k++) if (k < wv) run += (uint32_t)s_i[k];
#pragma unroll
      for (int k = 0; k < kEach; k++) {
        const int i = 1 + tid * kEach + k;
        if (i <= kSlabSlots + 1) S.start[i] = run;
        run += v[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      if (rslot[k] >= 0) {
        const uint32_t at = atomicAdd(&S.start[rslot[k] + 1], 1u);     // start[s + 1] walks from begin(s) to end(s) = begin(s + 1)
        S.q0[at] = rq[k];
        S.pos[at] = rpos[k];
      }
    }
    __syncthreads();
    const uint32_t level_mask = (uint32_t)s_i[10];
    const int n_idx = (int)S.start[kSlabSlots];    // indexed rows: entries [0, n_idx); brute rows: [n_idx, ns)
    ctock(c_build);

    // ---- columns: one wave per 64-entry block of the cell order, then per 64 entries of the brute list
    PairQueue Q{L.qbuf, 0, 0}, Q1{L.qbuf1b, 0, 0}, Q2{L.qbuf2, 0, 0};
    for (int item = tw; item < nitems; item += ntw) {
      const bool listed = item >= nwords;          // a block of the brute list
      uint32_t cpos = 0u;
      bool alive0 = false;
      float4 cq = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!listed) {
        const u64 m = ldg_agent(a.calive + item);
        if (m == 0ull) continue;
        if ((m >> lane) & 1ull) {
          const float4 ent = a.gsorted[(size_t)item * 64 + lane];
          cpos = __float_as_uint(ent.w);
          alive0 = (int)cpos >= c0 && (int)cpos < se;
          cq = make_float4(ent.x, ent.y, ent.z, inf);      // indexed = well conditioned: the short side never decides
        }
      } else {
        const int ui = (item - nwords) * 64 + lane;
        if (ui < n_brute) {
          cpos = a.ulist[ui];
          alive0 = (int)cpos >= c0 && (int)cpos < se && ((ldg_agent(a.alive + (cpos >> 6)) >> (cpos & 63)) & 1ull);
          if (alive0) cq = a.rec[(size_t)cpos * G::RECQ];
        }
      }
      if (__ballot(alive0) == 0ull) continue;
      c_items++;
      ctick();
      bool alive = alive0;
      wave_sync();
      L.cdead[lane] = alive0 ? 0 : 1;
      L.colpos[lane] = cpos;
      // ---- the rows within reach of the block: ranges of slab entries, packed (first | end << 16) in L.ranges
      int nranges = 0;
      {
        float bx0 = alive0 ? cq.x : inf, bx1 = alive0 ? cq.x : -inf, by0 = alive0 ? cq.y : inf, by1 = alive0 ? cq.y : -inf;
        float rmax = alive0 ? cq.z : 0.f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          bx0 = fminf(bx0, __shfl_xor(bx0, d)); bx1 = fmaxf(bx1, __shfl_xor(bx1, d));
          by0 = fminf(by0, __shfl_xor(by0, d)); by1 = fmaxf(by1, __shfl_xor(by1, d));
          rmax = fmaxf(rmax, __shfl_xor(rmax, d));
        }
        bool whole = listed;
        int ncomb = 0, my_lv = -1, my_cx0 = 0, my_cy = 0, my_len = 0;
        if (!whole) {
          const float mag = fmaxf(fabsf(bx0), fabsf(bx1)) + fmaxf(fabsf(by0), fabsf(by1));
          for (uint32_t lm = level_mask; lm; lm &= lm - 1) {
            const int lv = __builtin_ctz(lm);
            const float inv = grid_level_inv_cell(gp, lv);
            const float d = grid_query_halfwidth_mag(gp, lv, mag, rmax);
            const int lastx = grid_last_cell(gp.xr, inv), lasty = grid_last_cell(gp.yr, inv);
            const int cx0 = grid_cell(bx0 - d, gp.x0, inv, lastx), cy0 = grid_cell(by0 - d, gp.y0, inv, lasty);
            const int len = grid_cell(bx1 + d, gp.x0, inv, lastx) - cx0 + 1;
            const int nyr = grid_cell(by1 + d, gp.y0, inv, lasty) - cy0 + 1;
            if ((long long)len * nyr >= kSlabSlots / 2) whole = true;      // would visit entries several times over
            if (lane >= ncomb && lane < ncomb + nyr) { my_lv = lv; my_cx0 = cx0; my_cy = cy0 + (lane - ncomb); my_len = len; }
            ncomb += nyr;
          }
          if (ncomb > 64) whole = true;
        }
        wave_sync();
        if (whole) {
          c_whole++;
          if (lane == 0) L.ranges[0] = 0u | ((uint32_t)ns << 16);
          nranges = 1;
        } else {
          int s = 0, t = 0, t2 = 0;
          if (my_lv >= 0) {                        // lane = one cell row of one level's window: its slots [i0, i0 + len)
            const uint32_t i0 = grid_slot(my_lv, my_cx0, my_cy, kMask);
            const int e1 = (int)i0 + my_len;
            s = (int)S.start[i0];
            t = (int)S.start[e1 <= kSlabSlots ? e1 : kSlabSlots];
            if (e1 > kSlabSlots) t2 = (int)S.start[e1 - kSlabSlots];       // the slots wrap: second piece [0, e1 - M)
          }
          const u64 m1 = __ballot(t > s), m2 = __ballot(t2 > 0);
          const int n1 = __popcll(m1), n2 = __popcll(m2);
          if (t > s) L.ranges[__popcll(m1 & lanemask_lt())] = (uint32_t)s | ((uint32_t)t << 16);
          if (t2 > 0) L.ranges[n1 + __popcll(m2 & lanemask_lt())] = 0u | ((uint32_t)t2 << 16);
          nranges = n1 + n2;
          if (ns > n_idx) { if (lane == 0) L.ranges[nranges] = (uint32_t)n_idx | ((uint32_t)ns << 16); nranges++; }   // the brute rows
        }
        wave_sync();
      }
      auto drain2 = [&](int cnt) {                 // stage 2: exact clip; entries (row position, column lane)
        wave_sync();
        if (lane < cnt) {
          const int slot = (Q2.head + lane) & 127;
          const uint32_t rowp = L.qbuf2[slot];
          const int cc = L.q2col[slot];
          if (!L.cdead[cc]) {
            if (nms_stage_exact<G>(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)L.colpos[cc] * G::RECQ, G::thr_of(a), L.scr + lane)) L.cdead[cc] = 1;
          }
        }
        Q2.head = (Q2.head + cnt) & 127;
        Q2.count -= cnt;
        wave_sync();
      };
      auto push2 = [&](bool undecided, uint32_t rowp, uint32_t cc) {
        const u64 m2 = __ballot(undecided);
        if (undecided) {
          const int slot = (Q2.head + Q2.count + __popcll(m2 & lanemask_lt())) & 127;
          L.qbuf2[slot] = rowp;
          L.q2col[slot] = (uint8_t)cc;
        }
        Q2.count += __popcll(m2);
      };
      auto drain1b = [&](int cnt) {                // stage 1b: the IoU interval
        wave_sync();
        int res = 0;
        uint32_t rowp = 0, cc = 0;
        if (lane < cnt) {
          const int slot = (Q1.head + lane) & 127;
          rowp = L.qbuf1b[slot];
          cc = L.q1bcol[slot];
          if (!L.cdead[cc]) {
            res = nms_stage_full<G>(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)L.colpos[cc] * G::RECQ, G::thr_of(a));
            if (res == 1) L.cdead[cc] = 1;
          }
        }
        Q1.head = (Q1.head + cnt) & 127;
        Q1.count -= cnt;
        push2(res == 2, rowp, cc);
        wave_sync();
        if (Q2.count >= 64) drain2(64);
      };
      auto drain = [&](int cnt) {                  // stage 1a: the cheap register-only tests; entries (slab entry << 8 | column lane)
        wave_sync();
        int res = 0;
        uint32_t rowp = 0, cc = 0;
        if (lane < cnt) {
          const uint32_t it = L.qbuf[(Q.head + lane) & 127];
          cc = it & 255;
          rowp = S.pos[it >> 8];
          if (!L.cdead[cc]) {
            res = G::classify_quick(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)L.colpos[cc] * G::RECQ, G::thr_of(a), cull);
            if (res == 1) L.cdead[cc] = 1;
          }
        }
        Q.head = (Q.head + cnt) & 127;
        Q.count -= cnt;
        {
          const u64 m1 = __ballot(res == 3);
          if (res == 3) {
            const int slot = (Q1.head + Q1.count + __popcll(m1 & lanemask_lt())) & 127;
            L.qbuf1b[slot] = rowp;
            L.q1bcol[slot] = (uint8_t)cc;
          }
          Q1.count += __popcll(m1);
        }
        push2(res == 2, rowp, cc);
        wave_sync();
        if (Q1.count >= 64) drain1b(64);
        if (Q2.count >= 64) drain2(64);
        alive = alive && !L.cdead[lane];
      };
      ctock(c_rng);
      ctick();
      // every range is walked in tiles of 64 rows exactly like the exhaustive form walks its row tiles: lane k holds row k of
      // the tile (one conflict-free LDS read), the rows are broadcast with v_readlane, four rows per trip; the next range's
      // first tile is in flight while the current one is processed
      auto one_row = [&](int e, bool pass) {
        if (__ballot(pass)) {
          Q.push(pass, ((uint32_t)e << 8) | (uint32_t)lane);
          if (Q.count >= 64) { ctock(c_enum); ctick(); drain(64); ctock(c_drain); c_nd++; ctick(); }
        }
      };
      uint32_t pr_next = nranges > 0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.ranges[0]) : 0u;
      float4 tile_next = S.q0[min((int)(pr_next & 0xffffu) + lane, kSlabRows - 1)];
      for (int ri = 0; ri < nranges; ri++) {
        if (__ballot(alive) == 0ull) break;
        const int e_begin = (int)(pr_next & 0xffffu), e_end = (int)(pr_next >> 16);
        float4 myrow = tile_next;
        if (ri + 1 < nranges) {
          pr_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.ranges[ri + 1]);
          tile_next = S.q0[min((int)(pr_next & 0xffffu) + lane, kSlabRows - 1)];
        }
        c_rows += (u64)(e_end - e_begin);
        for (int e0 = e_begin; e0 < e_end; e0 += 64) {
          if (e0 != e_begin) myrow = S.q0[min(e0 + lane, kSlabRows - 1)];
          const int nrow = min(64, e_end - e0);
          int rr = 0;
          for (; rr + 4 <= nrow; rr += 4) {
            bool ps[4];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float4 rq1 = rdlane4(myrow, rr + k);
              ps[k] = alive && !(cull && G::cheap_reject(rq1, cq));
              any = any || ps[k];
            }
            if (__ballot(any)) {
#pragma unroll
              for (int k = 0; k < 4; k++) one_row(e0 + rr + k, ps[k]);
            }
          }
          for (; rr < nrow; rr++) {
            const float4 rq1 = rdlane4(myrow, rr);
            one_row(e0 + rr, alive && !(cull && G::cheap_reject(rq1, cq)));
          }
        }
      }
      ctock(c_enum);
      ctick();
      if (Q.count > 0) { drain(Q.count); c_nd++; }
      if (Q1.count > 0) drain1b(Q1.count);
      if (Q2.count > 0) drain2(Q2.count);
      ctock(c_drain);
      alive = alive && !L.cdead[lane];
      // RETURNING atomics whose result is consumed: the wave's vmcnt then covers the completed read-modify-write.
      // A box dies once: its bit of the score-order bitmap (select reads that one) and, for a block of the cell order,
      // the block's word.
      const bool killed = alive0 && !alive;
      u64 seen = 0ull;
      if (killed) seen = atomicAnd(a.alive + (cpos >> 6), ~(1ull << (cpos & 63)));
      const u64 kill = __ballot(killed);
      if (kill && !listed && lane == 0) seen ^= atomicAnd(a.calive + item, ~kill);
      asm volatile("; kills applied %0" ::"v"((unsigned)(seen >> 32) ^ (unsigned)seen));
    }
  }
  __syncthreads();                                 // the slab is read until here; the caller's next select reuses the memory
  if (cprof) {
    a.prof[16] += c_enum; a.prof[17] += c_drain; a.prof[18] += c_build; a.prof[19] += c_nd; a.prof[20] += c_rows; a.prof[21] += c_items;
    a.prof[15] += c_rng; a.prof[10] += c_whole;
  }
}

// ------------------------------------------------------------------ the persistent kernel
// dynamic LDS: [kNmsWaves x WaveLds<G>] (aliased by the resolve state) | chunk list [capmax] u32
// One workgroup per CU (the LDS footprint allows no second one) = 2 waves per SIMD: let the compiler use the whole
// 256-register budget of such a wave instead of spilling to scratch (measured: 20 MB of scratch writes per launch).
template <class G>
__global__ __launch_bounds__(kNmsThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_nms_persist(NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_i[16];
  __shared__ int s_flag;
  const int tid = threadIdx.x, wv = tid >> 6;
  const int NB = gridDim.x;
  int nteams = a.nseg < NB ? a.nseg : NB;
  int T = NB / nteams;
  int team = blockIdx.x / T, wg = blockIdx.x - team * T;
  int g_first = team, g_step = nteams, g_last = a.nseg - 1;
  long long skip_cost = -1;                            // >= 0: segments at least this expensive belong to a team of their own
  if (a.plan != nullptr && a.plan[0].w > 0) {          // planned (k_plan_teams): a team on one segment, or one workgroup on a run of small ones
    const int4 pl = a.plan[blockIdx.x];
    if (pl.x < 0) return;
    g_first = pl.x; g_step = 1; team = pl.y; T = pl.w;
    if (T == 1) { wg = 0; g_last = pl.x + pl.z; skip_cost = (long long)(unsigned)a.plan[kPlanInfoSlot].x; }
    else { wg = pl.z; g_last = pl.x; }
  } else if (team >= nteams) {
    return;
  }
  WaveLds<G>& L = reinterpret_cast<WaveLds<G>*>(smem)[wv];
  uint32_t* cidx = reinterpret_cast<uint32_t*>(smem + sizeof(WaveLds<G>) * kNmsWaves);
  const int tw = wg * kNmsWaves + wv, ntw = T * kNmsWaves;
  TeamBar bar{a.bar + (size_t)team * 128, a.bar + (size_t)team * 128 + 64, T, 0, a.abort_flag};

  const bool prof = a.prof != nullptr && blockIdx.x == 0 && tid == 0;
  u64 t0 = prof ? wall_clock64() : 0ull;
  const u64 t_wg0 = (a.prof && tid == 0) ? wall_clock64() : 0ull;
  auto lap = [&](int slot) {
    if (prof) { const u64 t1 = wall_clock64(); a.prof[slot] += t1 - t0; t0 = t1; }
  };

  // Spatial index for the cross phase: built before the launch (grid.h); the rows' side needs the chunk list's LDS
  GridPlan gp = {};
  bool slab_on = false;
  if constexpr (G::HAS_GRID) {
    if (a.gmeta != nullptr && a.cull != 0 && (size_t)a.capmax * 4 >= sizeof(SlabLds) && a.gmeta->on != 0) {
      gp = grid_plan(a.gmeta->bb);
      slab_on = gp.ok != 0;
    }
  }
  // kept rows x alive positions of [c0, c1): through the index of the rows when it pays (a handful of rows: exhaustive)
  auto cross = [&](const uint32_t* rows, int nr, int c0, int c1) {
    if constexpr (G::HAS_GRID) {
      if (slab_on && nr >= kSlabMinRows) {
        nms_cross_blocks<G>(a, gp, rows, nr, c0, c1, tw, ntw, L, *reinterpret_cast<SlabLds*>(cidx), s_i);
        return;
      }
    }
    nms_cross<G>(a, rows, nr, c0, c1, tw, ntw, L);
  };

  const int plan_chunk = a.cap_first < a.capmax ? a.cap_first : a.capmax;
  for (int g = g_first; g <= g_last; g += g_step) {
    const int sb = a.seg_begin[g], se = a.seg_end[g];
    // (the first segment of a plan entry is always the workgroup's own: a run starts with a small segment, and a big
    //  segment whose team has one member is an entry of its own)
    if (skip_cost >= 0 && g != g_first && plan_cost(se - sb, plan_chunk) >= skip_cost) continue;
    int cur = sb, kept = 0;
    int cap = a.cap_first < a.capmax ? a.cap_first : a.capmax;
    // Windows (only with max_keep): positions are opened a window at a time -- a new window is first tested against every
    // row kept so far, then processed like a segment of its own.  The greedy result is unchanged (every (kept row, later
    // box) pair is still tested before the box can be selected), but once max_keep rows are kept the remaining windows
    // are never touched: with max_det = 1500 and 30000 candidates most of the cross work disappears.
    int wend = (a.window > 0 && sb + a.window < se) ? sb + a.window : se;
    // ONE call site for the cross phase (the compiler inlines it there; two call sites made it a 250-register function
    // whose callers spill around the call): whoever needs one leaves a job, the head of the loop runs it.
    const uint32_t* jrows = nullptr;
    int jnr = 0, jc0 = 0, jc1 = 0;
    for (;;) {
      if (jnr > 0) {
        const u64 tcz = (a.prof && tid == 0) ? wall_clock64() : 0ull;
        cross(jrows, jnr, jc0, jc1);
        if (a.prof && tid == 0) { atomicMax(a.prof + 31, wall_clock64() - tcz); }
        lap(5);
        if (!team_barrier(bar, &s_flag)) return;           // the kills are visible before anybody selects again
        lap(0);
        jnr = 0;
      }
      if (!(cur < se && !(a.max_keep > 0 && kept >= a.max_keep))) break;
      if (cur >= wend) {                                   // open the next window: every row kept so far against it
        const int wnew = (cur + a.window < se) ? cur + a.window : se;
        wend = wnew;
        if (kept > 0) { jrows = a.rows + sb; jnr = kept; jc0 = cur; jc1 = wnew; continue; }
      }
      const int cn = nms_select(a, wend, cur, cap, cidx, s_i);
      lap(1);
      if (cn == 0) continue;                               // nothing alive in the rest of the window (cur == wend now)
      const u64 tpz = (a.prof && tid == 0) ? wall_clock64() : 0ull;
      nms_pairs<G>(a, team, cn, cidx, tw, ntw, L);
      if (a.prof && tid == 0) { const u64 d = wall_clock64() - tpz; atomicMax(a.prof + 29, d); atomicAdd(a.prof + 30, d); }
      lap(2);
      // ---- all edges are out: the last arriver resolves the chunk, the others wait for its rows
      if (serial_begin(bar, &s_flag)) {
        lap(3);
        const u64 ts = (a.prof && tid == 0) ? wall_clock64() : 0ull;
        nms_resolve(a, g, team, cn, kept, cidx, smem, sizeof(WaveLds<G>) * kNmsWaves, s_i);
        serial_end(bar);
        if (a.prof && tid == 0) { atomicAdd(a.prof + 9, wall_clock64() - ts); }
        lap(4);
      } else {
        if (!serial_wait(bar, &s_flag)) return;
        lap(3);
      }
      const int nr = ldg_agent(a.nrows + team);
      const bool more = cur < wend && !(a.max_keep > 0 && kept + nr >= a.max_keep);
      if (nr > 0 && more) { jrows = a.rows + sb + kept; jnr = nr; jc0 = cur; jc1 = wend; }   // the chunk's kept rows against what is left of the window
      kept += nr;
      if (prof) a.prof[6] += 1;
      if (cap < a.capmax) { cap *= 2; if (cap > a.capmax) cap = a.capmax; }
    }
  }
  if (a.prof && tid == 0) { const u64 el = wall_clock64() - t_wg0; atomicMax(a.prof + 22, el); atomicAdd(a.prof + 23, el); atomicAdd(a.prof + 24, 1ull); }
}

// ------------------------------------------------------------------ team planning for multi-segment launches
// One workgroup decides what the NB workgroups of the persistent launch do.  Segment cost = 64x64 tiles of the first
// chunk's triangle + a fixed part (plan_cost).  With L = total cost / NB:
//   big segments   (cost >= 2L) get a team of floor(cost / L') workgroups (L' = their share of the grid), plan[w] =
//                  {segment, team, index in team, team size};
//   small segments are packed, in segment order, into the remaining workgroups by equal cuts of their running cost:
//                  plan[w] = {first segment, team, segments in the run - 1, 1}; a run may enclose empty and big segments,
//                  which the workgroup skips (plan[kPlanInfoSlot].x = the cost from which a segment is big);
//   plan[w].x = -1: nothing to do.  plan[0].w == 0: nothing planned (no non-empty segment), the static teams are used.
// More non-empty segments than workgroups is fine (a class file of a whole test set in the merge NMS: ~900 images).
// exclusive prefix + total over the 1024 threads of the block (two values at once)
__device__ __forceinline__ void block_scan2(long long va, long long vb, long long* sa, long long* sb, long long& pa, long long& pb,
                                            long long& ta, long long& tb) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  long long ia = va, ib = vb;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long xa = __shfl_up(ia, d), xb = __shfl_up(ib, d);
    if (lane >= d) { ia += xa; ib += xb; }
  }
  __syncthreads();                                  // previous users of sa / sb are done
  if (lane == 63) { sa[wv] = ia; sb[wv] = ib; }
  __syncthreads();
  long long wa = 0, wb = 0; ta = 0; tb = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) { const long long xa = sa[k], xb = sb[k]; if (k < wv) { wa += xa; wb += xb; } ta += xa; tb += xb; }
  pa = wa + ia - va; pb = wb + ib - vb;
}

struct PlanLds { long long a[16], b[16]; int first[1024], last[1024]; };

// the planner proper: called by all 1024 threads of one workgroup (k_plan_teams, or the last workgroup of the fused
// sort/prep kernel); seg_begin / seg_end must be visible to the caller
__device__ __forceinline__ void plan_teams_block(const int* seg_begin, const int* seg_end, int nseg, int NB, int chunk, int4* plan,
                                                 PlanLds& S) {
  long long* s_a = S.a; long long* s_b = S.b;
  int* s_first = S.first; int* s_last = S.last;
  const int tid = threadIdx.x;
  const int per = (nseg + 1023) / 1024;
  const int g0 = tid * per, g1 = (g0 + per < nseg) ? g0 + per : nseg;
  auto cost_of = [&](int g) -> long long { return plan_cost((long long)(seg_end[g] - seg_begin[g]), chunk); };
  // ---- totals
  long long myc = 0, myn = 0, pa, pb, total, nne;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); myc += c; myn += c > 0 ? 1 : 0; }
  block_scan2(myc, myn, s_a, s_b, pa, pb, total, nne);
  if (nne == 0) {
    for (int w = tid; w < NB; w += 1024) plan[w] = make_int4(-1, 0, 0, 0);
    return;
  }
  const long long L = (total + NB - 1) / NB;
  const long long big_from = 2 * L;
  // ---- split of the grid: r workgroups for the small segments, NB - r for the teams of the big ones
  long long mybc = 0, mybn = 0, big_cost, n_big;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); if (c >= big_from) { mybc += c; mybn++; } }
  long long pbc, pbn;
  block_scan2(mybc, mybn, s_a, s_b, pbc, pbn, big_cost, n_big);
  const long long small_cost = total - big_cost, n_small = nne - n_big;
  long long r = 0;
  if (n_small > 0) {
    r = (small_cost + L - 1) / L;
    if (r < 1) r = 1;
    if (r > n_small) r = n_small;
    if (r > NB - n_big) r = NB - n_big;             // (n_big <= NB / 2: every big segment costs at least 2L)
  }
  // enough workgroups for everybody: no packing, every small segment gets its own workgroup (piece = its rank)
  const bool one_each = n_small > 0 && n_small + (n_big > 0 ? (big_cost + L - 1) / L : 0) <= NB;
  if (one_each) r = n_small;
  const long long NBb = NB - r;
  const long long Lb = n_big > 0 ? (big_cost + NBb - 1) / NBb : 1;
  auto team_size = [&](long long c) -> long long {
    long long t = c / Lb;
    const long long useful = (c + 63) / 64;          // more workgroups than ~tiles/8 only add barrier cost
    if (t > useful) t = useful;
    return t < 1 ? 1 : t;
  };
  // ---- teams of the big segments: workgroups r + [prefix of team sizes), team ids r + [prefix of count)
  long long myt = 0;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); if (c >= big_from) myt += team_size(c); }
  long long pt, pcnt, used_big, dummy;
  block_scan2(myt, mybn, s_a, s_b, pt, pcnt, used_big, dummy);
  if (r + used_big > NB) {                           // (cannot happen with the bounds above; the static teams are always valid)
    for (int w = tid; w < NB; w += 1024) plan[w] = make_int4(-1, 0, 0, 0);
    return;
  }
  {
    long long w0 = r + pt, team = r + pcnt;
    for (int g = g0; g < g1; g++) {
      const long long c = cost_of(g);
      if (c >= big_from) {
        const long long t = team_size(c);
        for (int i = 0; i < (int)t; i++) plan[w0 + i] = make_int4(g, (int)team, i, (int)t);
        w0 += t; team++;
      }
    }
  }
  for (int w = (int)(r + used_big) + tid; w < NB; w += 1024) plan[w] = make_int4(-1, 0, 0, 0);
  // ---- runs of small segments: piece = running small cost / Ls
  long long mysc = 0, mysn = 0, psc, psn, tsc, tsn;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); if (c > 0 && c < big_from) { mysc += c; mysn++; } }
  block_scan2(mysc, mysn, s_a, s_b, psc, psn, tsc, tsn);
  const long long Ls = r > 0 ? (small_cost + r - 1) / r : 1;
  s_first[tid] = 0x7fffffff; s_last[tid] = -1;
  __syncthreads();
  if (r > 0) {
    long long run = psc, rank = psn;
    for (int g = g0; g < g1; g++) {
      const long long c = cost_of(g);
      if (c > 0 && c < big_from) {
        const int piece = one_each ? (int)rank : (int)(run / Ls);   // < r: run < small_cost <= r * Ls
        atomicMin(&s_first[piece], g); atomicMax(&s_last[piece], g);
        run += c; rank++;
      }
    }
  }
  __syncthreads();
  for (int w = tid; w < (int)r; w += 1024) {
    const int f = s_first[w], l = s_last[w];
    plan[w] = (l >= f) ? make_int4(f, w, l - f, 1) : make_int4(-1, 0, 0, 0);       // (no segment starts in this piece: idle)
  }
  if (tid == 0) plan[kPlanInfoSlot] = make_int4((int)(big_from > 0x7fffffffLL ? 0x7fffffffLL : big_from), 0, 0, 0);
}

__global__ __launch_bounds__(1024) void k_plan_teams(const int* __restrict__ seg_begin, const int* __restrict__ seg_end, int nseg, int NB,
                                                     int chunk, int4* __restrict__ plan) {
  __shared__ PlanLds S;
  plan_teams_block(seg_begin, seg_end, nseg, NB, chunk, plan, S);
}

}  // namespace obb
