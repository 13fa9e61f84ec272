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

struct SlabPlan;

// Where a single-list call stands when the persistent kernel is launched BEHIND the phase kernels of nms_mk.h (the head of their
// control block): it returns at once when they completed the call, and otherwise carries on from their state -- the first step of a
// call they handed over, the chunk they had selected, or the cross phase of a chunk whose rows are kept already.
struct NmsResume {
  int cur, kept, done, step;
  int cn, cap, nrow, chunk_first;
  float dmax2;
  int stage;                 // 1: a chunk [chunk_first, cur) is selected but not resolved; 3: the rows of the last chunk are kept
  int bail;                  // != 0: the phase kernels stood back in the stage named (stage 3: somewhere in the cross phase).  Stage 3 means
                             // "cross phase not known to be complete" either way -- the last enqueued step has no cross half -- and kills
                             // are idempotent: the rows are crossed (again) before anything else
};

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
  int* bar_sub;              // [kBarGroups][64] group counters of the whole grid's two-level barrier (zero before the launch)
  int* abort_flag;           // [1] set when a spin gave up
  u64* prof;                 // optional [56]: wall-clock ticks (10 ns) per phase (development aid)
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
  // Spatial index over the boxes (grid.h); gmeta == NULL: none.  Everything about it happens INSIDE this kernel, and only
  // when a step keeps enough rows to need it (grid_build): a call whose chunks keep a few hundred rows pays nothing but
  // the bounding-box partials of the key kernel.
  GridMeta* gmeta;           // zero before the launch: n_brute / level_mask are counted by grid_build
  const int* bbpart;         // [nparts][kBbInts] bounding box of the centres (+ largest w^2+h^2), one partial per workgroup of the key kernel
  int nparts;
  int* gcnt;                 // [gmask + 1] zero before the launch and after every build
  int* gstart;               // [gmask + 2] first entry of every table slot in gsorted (exclusive prefix; [gmask + 1] = total)
  float4* gsorted;           // {x, y, r, sorted position as bits | dead flag in bit 31} in slot order
  uint32_t* gslot;           // [n] sorted position -> its entry of gsorted (boxes in the index)
  int* gwsum;                // [grid size] per-workgroup totals of the distributed scan
  uint32_t* ulist;           // [gmeta->n_brute] sorted positions of the boxes kept out of the index (filled by grid_build)
  uint32_t gmask;            // table size - 1 (power of two)
  int gfine;                 // GridPlan::fine
  // Independent slabs (grid.h): slab_cover == NULL: none.  One list whose boxes fall into groups that cannot overlap each
  // other (the callers' cls * 4096 offsets) is re-laid out slab by slab INSIDE this kernel (slab_setup) and run as that many
  // concurrent segments, one team each; the kept boxes meet again in score order through a bitmap over the original
  // positions (slab_merge).
  const uint32_t* slab_cover; // [kSlabCopies][kSlabWords] x bins touched by a box (written by the prep kernel; OR the copies)
  const int* slab_flag;       // [0] != 0: a box that cannot be placed (not finite / ill conditioned): no decomposition;
                              // [2] != 0: the data is wide enough to look for slabs (slab_gate) and the bins are marked
  int* slab_cnt;              // [gridDim.x][kMaxSlabs]
  int* slab_tot;              // [1 + ceil(gridDim.x / 16)][kMaxSlabs] slab totals, then the totals of every group of 16 workgroups (zero before the launch)
  int* slab_keep;             // [kMaxSlabs]
  float4* rec2; uint32_t* order2; uint32_t* pos_old; u64* alive2; u64* kept_bits;
  int alive2_words, kept_words;
  const SlabPlan* slab_plan;  // written by k_slab_split in front of this launch (NULL: no decomposition was looked for)
  int slab_cap;               // > 0: upper limit of a slab team's chunk capacity
  int grow_sparse;            // chunk growth factor after a sparse chunk (<= 2: always double)
  int lpt;                    // 1: the resolver hands the kept rows of a chunk to the indexed cross phase LARGEST FIRST (nms_resolve)
  const NmsResume* resume;    // optional (one list, no slabs): see NmsResume
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

// rows r and r + 1 of a tile's row records (one per lane) against the splatted column box: the packed circle test
template <class G>
__device__ __forceinline__ void rows_reject2(const float4& myrow, int r, const ColPk& c, bool& r0, bool& r1) {
  const f32x2 ax = {rdlane(myrow.x, r), rdlane(myrow.x, r + 1)}, ay = {rdlane(myrow.y, r), rdlane(myrow.y, r + 1)};
  const f32x2 az = {rdlane(myrow.z, r), rdlane(myrow.z, r + 1)}, aw = {rdlane(myrow.w, r), rdlane(myrow.w, r + 1)};
  G::cheap_reject2(ax, ay, az, aw, c, r0, r1);
}

constexpr int kResolveTail = 256;           // live edges from which ONE wave finishes a chunk's resolve (nms_resolve)
constexpr int kBarGroups = 64;             // groups of 16 workgroups: grids of up to 1024
constexpr int kNmsThreads = 512;
constexpr int kNmsWaves = kNmsThreads / 64;
constexpr unsigned kSpinLimit = 1u << 22;

// Team barriers on monotonic counters (every barrier of either kind adds one arrival per workgroup) and one go word on a line
// of its own, written by whoever completes the barrier: the waiting workgroups poll THAT word, the arrivals do not share a
// channel with 255 pollers.  Wide teams (the whole grid of a single-list call) arrive in two levels: same-address device
// atomics retire at ~11 ns apiece (MI355X_MICROARCH.md), 256 arrivals on one counter are 2.8 us of every barrier; with groups
// of 16 on counters 256 bytes apart and one arrival per group on the top counter it is 16 + 16.
struct TeamBar {
  int* arrive; int* go; int T; int epoch; int* abort_flag;
  int* sub;      // group counters (one 256-byte line each) or NULL: one level
  int wg;        // this workgroup's index in the team
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
// one arrival (thread 0 of the workgroup); true in the workgroup whose arrival completes the barrier
__device__ __forceinline__ bool bar_arrive(const TeamBar& b) {
  if (b.sub == nullptr) {
    const int t = __hip_atomic_fetch_add(b.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t == b.T * (b.epoch + 1) - 1;
  }
  const int g = b.wg >> 4, G = (b.T + 15) >> 4;
  const int sz = (b.T - (g << 4)) < 16 ? (b.T - (g << 4)) : 16;
  const int t = __hip_atomic_fetch_add(b.sub + g * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t != sz * (b.epoch + 1) - 1) return false;
  const int u = __hip_atomic_fetch_add(b.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u == G * (b.epoch + 1) - 1;
}
// plain barrier: returns false on abort
__device__ __forceinline__ bool team_barrier(TeamBar& b, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave drains its write-through stores / atomics
  __syncthreads();
  if (threadIdx.x == 0) {
    if (bar_arrive(b)) { stg_agent(b.go, b.epoch + 1); *s_flag = 0; }
    else *s_flag = spin_until(b.go, b.epoch + 1, b.abort_flag) ? 0 : 1;
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
    const int last = bar_arrive(b) ? 1 : 0;
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
  uint32_t qcol1b[128];    // grid cross: column position of a stage-1b entry
  uint32_t qcol2[128];     // grid cross: column position of a stage-2 entry
  uint8_t q1bcol[128];     // cross phase: column lane of a stage-1b entry
  uint8_t q2col[128];      // cross phase: column lane of a stage-2 entry
  uint8_t cdead[64];
  uint8_t pad[64];
  float4 align16[0];
};

// Phases that run once per call or once per step: forceinline by default; a build may turn any of them into a real
// function (-DOBB_COLD_x="__device__ __attribute__((noinline))") to shorten the live ranges of the persistent kernel.
#ifndef OBB_COLD_RESOLVE
#define OBB_COLD_RESOLVE __device__ __forceinline__
#endif
#ifndef OBB_COLD_GRID
#define OBB_COLD_GRID __device__ __forceinline__
#endif
#ifndef OBB_COLD_SLAB
#define OBB_COLD_SLAB __device__ __forceinline__
#endif
#ifndef OBB_COLD_SELECT
#define OBB_COLD_SELECT __device__ __forceinline__
#endif
#ifndef OBB_STAGE_ATTR
#define OBB_STAGE_ATTR __attribute__((noinline))
#endif
// The two expensive decision stages are real functions (one body per geometry): the pair phase and the forms of the
// cross phase all drain their queues through them.
template <class G, class TH>
__device__ OBB_STAGE_ATTR int nms_stage_full(const float4* ra, const float4* rb, TH thr) {
  return G::classify_full(ra, rb, thr);
}
template <class G, class TH>
__device__ OBB_STAGE_ATTR bool nms_stage_exact(const float4* ra, const float4* rb, TH thr, float* scr) {
  return G::hit_exact(ra, rb, thr, scr);
}

// FN = true: through the shared functions (the single-list kernel with the indexed cross phase, which is at the register
// limit); false: inlined into the drains (the multi-segment kernel of the fused driver: no index code, registers to spare,
// and its ~100-box segments are bound by the latency of exactly these stages).
template <class G, bool FN, class TH>
__device__ __forceinline__ int stage_full(const float4* ra, const float4* rb, TH thr) {
  if constexpr (FN) return nms_stage_full<G>(ra, rb, thr);
  else return G::classify_full(ra, rb, thr);
}
template <class G, bool FN, class TH>
__device__ __forceinline__ bool stage_exact(const float4* ra, const float4* rb, TH thr, float* scr) {
  if constexpr (FN) return nms_stage_exact<G>(ra, rb, thr, scr);
  else return G::hit_exact(ra, rb, thr, scr);
}

// Wave-level form of the exact stage: every lane of the wave calls, `want` lanes hold a pair.  A geometry whose exact stage
// splits into independent parts (QuadGeom: the 16 terms of the reference's sum) decides a FEW pairs with several lanes per
// pair (geom.h: hit_exact_coop) -- a leftover drain of a dozen quad pairs costs three term times instead of sixteen.
template <class G> struct has_coop { template <class T> static constexpr bool f(decltype(T::HAS_COOP)*) { return T::HAS_COOP; } template <class T> static constexpr bool f(...) { return false; } static constexpr bool value = f<G>(nullptr); };
template <class G, class TH>
__device__ OBB_STAGE_ATTR bool nms_stage_exact_coop(bool want, const float4* ra, const float4* rb, TH thr, float* scr_wave) {
  if constexpr (has_coop<G>::value) return G::hit_exact_coop(want, ra, rb, thr, scr_wave);
  else return false;
}
template <class G, bool FN, class TH>
__device__ __forceinline__ bool stage_exact_wave(bool want, const float4* ra, const float4* rb, TH thr, float* scr_wave) {
  if constexpr (has_coop<G>::value) {
    if (__popcll(__ballot(want)) <= G::kCoopMax) return nms_stage_exact_coop<G>(want, ra, rb, thr, scr_wave);
  }
  bool hit = false;
  if (want) hit = stage_exact<G, FN>(ra, rb, thr, scr_wave + (threadIdx.x & 63));
  return hit;
}
// how many pooled leftover pairs a wave takes per trip: 64, or -- where few pairs are decided faster by several lanes each --
// an even share of the pool (a multiple of four: four pairs per pass)
template <class G>
__device__ __forceinline__ int exact_pool_share(int total) {
  if constexpr (has_coop<G>::value) {
    if (total <= kNmsWaves * G::kCoopMax) { const int s = ((total + kNmsWaves - 1) / kNmsWaves + 3) & ~3; return s < 4 ? 4 : s; }
  }
  return 64;
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
OBB_COLD_SELECT int nms_select(const NmsArgs& a, int se, int& cur, int cap, uint32_t* cidx, int* s_i) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int w_first = cur >> 6, w_last = (se - 1) >> 6;
  int off = 0;
  __syncthreads();
  if (tid == 0) s_i[8] = se;
  // the first four trips' words are requested together: a late step scans the whole bitmap (1563 words at N = 100k) and paid a
  // coherent round trip to memory per trip (a dense early chunk ends after the first trip and wastes three loads)
  u64 pre[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { const int w = w_first + k * kNmsThreads + tid; pre[k] = (w <= w_last) ? ldg_agent(a.alive + w) : 0ull; }
  int trip = 0;
  for (int wbase = w_first; wbase <= w_last && off < cap; wbase += kNmsThreads, trip++) {
    const int w = wbase + tid;
    u64 m = 0ull;
    if (w <= w_last) {
      m = trip == 0 ? pre[0] : trip == 1 ? pre[1] : trip == 2 ? pre[2] : trip == 3 ? pre[3] : ldg_agent(a.alive + w);
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
template <class G, bool FN>
__device__ __forceinline__ void nms_pairs(const NmsArgs& a, int tm, int cn, const uint32_t* cidx, int tw, int ntw, WaveLds<G>& L, int* s_next) {
  const int lane = threadIdx.x & 63;
  const int nb = (cn + 63) >> 6;
  const int items = nb * nb;
  uint32_t* edges = a.edges + (size_t)tm * a.ecap;
  const bool cull = a.cull != 0;
  PairQueue Q{L.qbuf, 0, 0}, Q1{L.qbuf1b, 0, 0}, Q2{L.qbuf2, 0, 0};
  const bool pprof = a.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  u64 pt0 = 0, p_items = 0, p_loop = 0, p_load = 0, p_n1a = 0, p_t1a = 0, p_n1b = 0, p_t1b = 0, p_n2 = 0, p_t2 = 0;
  auto ptick = [&]() { if (pprof) pt0 = wall_clock64(); };
  auto ptock = [&](u64& acc) { if (pprof) { const u64 t = wall_clock64(); acc += t - pt0; pt0 = t; } };
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
    uint32_t packed = 0;
    uint32_t pi = 0, pj = 0;
    if (lane < cnt) {
      packed = L.qbuf2[(Q2.head + lane) & 127];
      pi = cidx[packed >> 16]; pj = cidx[packed & 0xffff];
    }
    const bool hit = stage_exact_wave<G, FN>(lane < cnt, a.rec + (size_t)pi * G::RECQ, a.rec + (size_t)pj * G::RECQ, G::thr_of(a), L.scr);
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
      res = stage_full<G, FN>(a.rec + (size_t)pi * G::RECQ, a.rec + (size_t)pj * G::RECQ, G::thr_of(a));
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
  // Only the tiles of the upper triangle are enumerated (t = 0 .. tri-1, row-major: row tile rb starts at off(rb) =
  // rb * nb - rb (rb - 1) / 2): walking all nb x nb tiles and skipping rb > cb gave a wave a FIXED column (the number of
  // waves is a multiple of nb for the usual chunk sizes), i.e. between 0 and 2x the average number of real tiles.
  const int items_sub = tri * nsub;
  (void)items;
  auto row_off = [&](int rb) { return rb * nb - ((rb * (rb - 1)) >> 1); };
  // stage 1a: the cheap register-only tests.  Entries are chunk-local (i << 16 | j) like those of the later stages, so
  // the queue lives ACROSS tiles: a sparse chunk (a few passing pairs per tile) pays the latency of a drain -- LDS look-up,
  // two record fetches, the edge counter's atomic -- once per 64 pairs instead of once per tile.
  auto drain = [&](int cnt) {   // wave-uniform cnt <= 64
    wave_sync();
    int res = 0;
    uint32_t packed = 0;
    if (lane < cnt) {
      packed = L.qbuf[(Q.head + lane) & 127];
      const uint32_t pi = cidx[packed >> 16], pj = cidx[packed & 0xffff];
      res = G::classify_quick(a.rec + (size_t)pi * G::RECQ, a.rec + (size_t)pj * G::RECQ, G::thr_of(a), cull);
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
  // (items are dealt to the workgroups of the team, w, w + T, ...; a workgroup's waves take them from a counter in LDS:
  //  a tile costs between nothing -- no pair passes the circle test -- and several drains)
  const int wgi = tw / kNmsWaves, Tw = ntw / kNmsWaves;
  __syncthreads();
  if (threadIdx.x == 0) *s_next = 0;
  __syncthreads();
  for (;;) {
    int kk = 0;
    if (lane == 0) kk = atomicAdd(s_next, 1);
    kk = __builtin_amdgcn_readfirstlane(kk);
    const long long it_ll = (long long)wgi + (long long)kk * Tw;
    if (it_ll >= items_sub) break;
    const int it2 = (int)it_ll;
    const int item = it2 / nsub, sub = it2 - item * nsub;
    int rb = (int)(((float)(2 * nb + 1) - sqrtf((float)((2 * nb + 1) * (2 * nb + 1) - 8 * item))) * 0.5f);
    rb = rb < 0 ? 0 : (rb > nb - 1 ? nb - 1 : rb);
    while (rb + 1 < nb && row_off(rb + 1) <= item) rb++;       // (float square root: fix the estimate up)
    while (row_off(rb) > item) rb--;
    const int cb = rb + (item - row_off(rb));
    const int r = rb * 64 + lane, c = cb * 64 + lane;
    const bool rvalid = r < cn, cvalid = c < cn;
    const uint32_t rp = rvalid ? cidx[r] : 0u, cp = cvalid ? cidx[c] : 0u;
    ptick();
    const float4 myrow = rvalid ? a.rec[(size_t)rp * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 cq = cvalid ? a.rec[(size_t)cp * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (pprof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ptock(p_load); p_items++; }
    const int nrow = min(64, cn - rb * 64);
    const int rr_lo = sub * rows_sub, rr_hi = min(nrow, rr_lo + rows_sub);
    const bool diag = rb == cb;
    const uint32_t pk_col = (uint32_t)(cb * 64 + lane), pk_row0 = (uint32_t)(rb * 64) << 16;

    // four rows per trip: their broadcasts and cheap tests are issued together and ONE ballot decides whether any of them
    // has a pair to queue (in a sparse chunk most rows pass for no column of the tile)
    auto one_row = [&](int rr, bool pass) {
      if (__ballot(pass)) {
        Q.push(pass, (pk_row0 + ((uint32_t)rr << 16)) | pk_col);
        if (Q.count >= 64) { ptock(p_loop); drain(64); ptock(p_t1a); p_n1a++; }
      }
    };
    int rr = rr_lo;
    [[maybe_unused]] ColPk cpk;
    if constexpr (G::PACKED) cpk = col_splat(cq);
    for (; rr + 4 <= rr_hi; rr += 4) {
      bool ps[4];
      bool any = false;
      if constexpr (G::PACKED) {                 // two rows per packed instruction (geom.h)
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          bool r0, r1;
          rows_reject2<G>(myrow, rr + k, cpk, r0, r1);
          ps[k] = cvalid && (!diag || lane > rr + k) && !(cull && r0);
          ps[k + 1] = cvalid && (!diag || lane > rr + k + 1) && !(cull && r1);
          any = any || ps[k] || ps[k + 1];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float4 rq = rdlane4(myrow, rr + k);
          ps[k] = cvalid && (!diag || lane > rr + k) && !(cull && G::cheap_reject(rq, cq));
          any = any || ps[k];
        }
      }
      if (__ballot(any)) {
#pragma unroll
        for (int k = 0; k < 4; k++) one_row(rr + k, ps[k]);
      }
    }
    for (; rr < rr_hi; rr++) {
      const float4 rq = rdlane4(myrow, rr);
      one_row(rr, cvalid && (!diag || lane > rr) && !(cull && G::cheap_reject(rq, cq)));
    }
    ptock(p_loop);
  }
  ptick();
  // The stage-1a leftovers of the workgroup's waves are pooled like the exact-clip leftovers below: full drains on a few
  // waves instead of eight partial ones that share the SIMDs.
  {
    WaveLds<G>* Lw = &L - (threadIdx.x >> 6);
    auto pool_at = [&](int idx) -> uint32_t& { return reinterpret_cast<uint32_t*>(&Lw[idx >> 7])[idx & 127]; };
    int* s_pool = s_next + 1;
    __syncthreads();                                           // every wave is out of its item loop
    if (threadIdx.x == 0) *s_pool = 0;
    __syncthreads();
    int off = 0;
    if (lane == 0 && Q.count > 0) off = atomicAdd(s_pool, Q.count);
    off = __builtin_amdgcn_readfirstlane(off);
    for (int k = lane; k < Q.count; k += 64) pool_at(off + k) = L.qbuf[(Q.head + k) & 127];
    Q.head = 0; Q.count = 0;
    __syncthreads();
    const int total = *s_pool;
    for (int c0 = (threadIdx.x >> 6) * 64; c0 < total; c0 += kNmsWaves * 64) {
      const int cnt = min(64, total - c0);
      wave_sync();
      if (lane < cnt) L.qbuf[lane] = pool_at(c0 + lane);       // back into this wave's (empty) queue: the ordinary drain takes it from there
      Q.head = 0; Q.count = cnt;
      drain(cnt);
      p_n1a++;
    }
    ptock(p_t1a);
    __syncthreads();                                           // the pool is free again (the exact-clip leftovers use it next)
  }
  // Leftovers.  A drain costs its latency whatever it holds (LDS look-ups, record fetches, a call into ~1500 instructions:
  // 10-18 us for the interval stage on a wave that shares its SIMD): a handful of undecided pairs skip the interval and
  // go straight to the exact clip, which decides them anyway -- one drain instead of two at the end of every step.
  if (Q1.count > 0 && Q1.count + Q2.count <= 64) {
    wave_sync();
    const bool mv = lane < Q1.count;
    const uint32_t e = mv ? L.qbuf1b[(Q1.head + lane) & 127] : 0u;
    Q1.head = (Q1.head + Q1.count) & 127;
    Q1.count = 0;
    Q2.push(mv, e);
    wave_sync();
  }
  if (Q1.count > 0) { drain1b(Q1.count); ptock(p_t1b); p_n1b++; }
  // The exact-clip leftovers of the workgroup's eight waves are pooled (LDS: the row / column position arrays of the wave
  // blocks, unused in this phase) and drained 64 at a time by as few waves as it takes -- waves 0..3 sit on different
  // SIMDs: eight waves each clipping a dozen pairs share the four SIMDs and take about twice as long as one full wave alone.
  {
    WaveLds<G>* Lw = &L - (threadIdx.x >> 6);                 // the workgroup's wave blocks
    auto pool_at = [&](int idx) -> uint32_t& { return reinterpret_cast<uint32_t*>(&Lw[idx >> 7])[idx & 127]; };   // rowpos[64] | colpos[64], the first 512 bytes of wave block idx / 128
    int* s_pool = s_next + 1;
    __syncthreads();                                           // every wave is out of its item loop (s_next is free: see the reset above)
    if (threadIdx.x == 0) *s_pool = 0;
    __syncthreads();
    int off = 0;
    if (lane == 0 && Q2.count > 0) off = atomicAdd(s_pool, Q2.count);
    off = __builtin_amdgcn_readfirstlane(off);
    for (int k = lane; k < Q2.count; k += 64) pool_at(off + k) = L.qbuf2[(Q2.head + k) & 127];
    Q2.head = (Q2.head + Q2.count) & 127;
    Q2.count = 0;
    __syncthreads();
    const int total = *s_pool;
    const int share = exact_pool_share<G>(total);
    for (int c0 = (threadIdx.x >> 6) * share; c0 < total; c0 += kNmsWaves * share) {
      const int cnt = min(share, total - c0);
      uint32_t packed = 0;
      uint32_t pi = 0, pj = 0;
      if (lane < cnt) {
        packed = pool_at(c0 + lane);
        pi = cidx[packed >> 16]; pj = cidx[packed & 0xffff];
      }
      const bool hit = stage_exact_wave<G, FN>(lane < cnt, a.rec + (size_t)pi * G::RECQ, a.rec + (size_t)pj * G::RECQ, G::thr_of(a), L.scr);
      emit(hit, packed);
      p_n2++;
    }
    ptock(p_t2);
  }
  if (pprof) {
    a.prof[32] += p_items; a.prof[33] += p_loop + p_load; a.prof[40] += p_load; a.prof[34] += p_n1a; a.prof[35] += p_t1a; a.prof[36] += p_n1b;
    a.prof[37] += p_t1b; a.prof[38] += p_n2; a.prof[39] += p_t2;
  }
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
// hasin (optional, [cn] bytes, nms_mk.h): != 0 where at least one edge points AT the member, set while the edges were written.  The
// members nobody points at are kept before the first round instead of after it: the first pass over the list already kills and prunes.
OBB_COLD_RESOLVE int nms_resolve(const NmsArgs& a, int g, int sb, int tm, int cn, int kept_before, const uint32_t* cidx, uint8_t* smem,
                           size_t smem_bytes, int* s_i, const uint8_t* hasin = nullptr) {
  const int tid = threadIdx.x;
  uint8_t* state = smem;              // 0 undecided, 1 kept, 2 dead
  uint8_t* blocked = smem + a.capmax;
  uint32_t* ledges = reinterpret_cast<uint32_t*>(smem + 2 * (size_t)a.capmax);
  const long long lcap = ((long long)smem_bytes - 2LL * a.capmax) / 4;
  if (hasin != nullptr) {
    // (four members per 32-bit word; the bytes behind cn stay inside the capmax-sized arrays and are never read)
    for (int j4 = tid; j4 * 4 < cn; j4 += kNmsThreads) {
      const uint32_t w = reinterpret_cast<const uint32_t*>(hasin)[j4];
      uint32_t st = 0u;
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) st |= (((w >> (8 * c4)) & 0xffu) ? 0u : 1u) << (8 * c4);
      reinterpret_cast<uint32_t*>(state)[j4] = st;
      reinterpret_cast<uint32_t*>(blocked)[j4] = 0u;
    }
  } else {
    for (int j = tid; j < cn; j += kNmsThreads) { state[j] = 0; blocked[j] = 0; }
  }
  long long E = ldg_agent(a.nedges + tm);
  if (E > a.ecap) E = a.ecap;         // cannot happen: ecap is the worst case capmax*(capmax-1)/2
  const uint32_t* edges = a.edges + (size_t)tm * a.ecap;   // plain loads: acquired in serial_begin
  // The list fits in LDS when every thread's share does (one contiguous block per thread, odd length: conflict-free banks).
  const int percap = (int)(((lcap / kNmsThreads) - 1) | 1);
  int per = (int)((E + kNmsThreads - 1) / kNmsThreads) + 5;   // (+ the slack of dealing the list out four edges at a time: a thread holds at most 4 * ceil(E / 2048) <= E / 512 + 4 edges)
  per |= 1;
  bool lds_mode = per <= percap;
  if (!lds_mode) per = percap;
  int mycnt = 0;
  uint32_t* mine_e = ledges + (size_t)tid * per;
  if (lds_mode) {
    // thread t takes the 16-byte groups t, t+512, ... (coalesced global reads, eight groups = 32 edges in flight) into its
    // own LDS block; the list was written by every workgroup of the team and sits in L2 / memory: this is a latency chain
    const uint4* e4 = reinterpret_cast<const uint4*>(edges);      // (every team's list starts on a 16-byte boundary)
    const long long nvec = E >> 2;
    for (long long v0 = tid; v0 < nvec; v0 += 8 * kNmsThreads) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const long long k = v0 + (long long)u * kNmsThreads; v[u] = k < nvec ? e4[k] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (v0 + (long long)u * kNmsThreads < nvec) {
          mine_e[mycnt] = v[u].x; mine_e[mycnt + 1] = v[u].y; mine_e[mycnt + 2] = v[u].z; mine_e[mycnt + 3] = v[u].w;
          mycnt += 4;
        }
    }
    if (tid < (int)(E & 3)) mine_e[mycnt++] = edges[(nvec << 2) + tid];
  }
  if (tid == 0) { s_i[9] = 0; s_i[10] = 0; }                  // live-edge counters of the rounds (by round parity)
  __syncthreads();
  u64 tp = 0;
  if (a.prof && tid == 0) { tp = wall_clock64(); atomicAdd(a.prof + 25, (u64)E); atomicMax(a.prof + 27, (u64)E); atomicAdd(a.prof + 28, (u64)cn); atomicAdd(a.prof + 26, (u64)(lds_mode ? 1 : 0)); }
  auto plap = [&](int slot) { if (a.prof && tid == 0) { const u64 t = wall_clock64(); atomicAdd(a.prof + slot, t - tp); tp = t; } };

  // A list that does not fit is first streamed READ-ONLY from global memory (16-byte plain loads, 8 in flight: the
  // list was published write-through and acquired in serial_begin, nothing writes it during the serial section):
  // the rounds work exactly as below but nothing is pruned.  Each such round counts the edges that still have two
  // undecided ends and copies them into the thread's LDS block as long as they fit (round 6: optimistically, in the same
  // pass -- counting first and copying in the NEXT round cost a third 45 us pass over a 100,000-edge list); these can only
  // become fewer, so once every thread's survivors fit the self-pruning rounds take over.
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
              if (remaining < per) mine_e[remaining] = e[c];
              remaining++;
            }
          }
        }
      }
    }
    // live edges of the whole list (LDS mode): two counters by round parity, the idle one is cleared for the next round
    int* live_cnt = &s_i[9 + (round & 1)];
    if (lds_mode && mycnt > 0) atomicAdd(live_cnt, mycnt);
    __syncthreads();
    if (tid == 0) s_i[9 + ((round + 1) & 1)] = 0;
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
      if (__syncthreads_or(remaining > per ? 1 : 0) == 0) { lds_mode = true; mycnt = remaining; }   // block-uniform: the survivors are in LDS now
    } else if (*live_cnt <= kResolveTail && (long long)kNmsThreads * per + kResolveTail <= lcap) {
      // The tail: a few hundred live edges, a few dozen undecided boxes, and a dependency chain that still needs several
      // rounds (a round here decides one level of the chain: 9-10 rounds for the 1873-box chunk of S-clustered K=300, of
      // which the first leaves 29 846 -> ~100 edges).  A block-wide round costs ~3 us whatever it holds (three workgroup
      // barriers, a pass over all cn states); ONE wave with the edges in registers runs a round in a few hundred cycles:
      // kills by the sources kept so far; blocks by undecided sources; targets nobody blocks are kept; edges whose target is
      // decided or whose source is dead are dropped.  Same fixed point: a box is kept once every higher-scored neighbour is
      // dead, dropped as soon as one is kept.
      uint32_t* tail_e = ledges + lcap - kResolveTail;   // (behind the per-thread blocks: checked above)
      if (tid == 0) s_i[14] = 0;
      __syncthreads();
      if (mycnt > 0) {
        const int tb = atomicAdd(&s_i[14], mycnt);
        for (int k = 0; k < mycnt; k++) tail_e[tb + k] = mine_e[k];
      }
      __syncthreads();
      if (tid < 64) {
        const int et = s_i[14];
        uint32_t ed[kResolveTail / 64];
        bool lv[kResolveTail / 64];
#pragma unroll
        for (int c = 0; c < kResolveTail / 64; c++) { const int k = tid + 64 * c; lv[c] = k < et; ed[c] = lv[c] ? tail_e[k] : 0u; }
        auto wsync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
        for (int guard = 0; guard < 70000; guard++) {
#pragma unroll
          for (int c = 0; c < kResolveTail / 64; c++)
            if (lv[c] && state[ed[c] >> 16] == 1) state[ed[c] & 0xffff] = 2;                          // kills
          wsync();
#pragma unroll
          for (int c = 0; c < kResolveTail / 64; c++)
            if (lv[c] && state[ed[c] & 0xffff] == 0 && state[ed[c] >> 16] == 0) blocked[ed[c] & 0xffff] = 1;   // blocks
          wsync();
#pragma unroll
          for (int c = 0; c < kResolveTail / 64; c++)
            if (lv[c] && state[ed[c] & 0xffff] == 0 && !blocked[ed[c] & 0xffff]) state[ed[c] & 0xffff] = 1;      // nobody blocks it: kept
          wsync();
          bool any_live = false;
#pragma unroll
          for (int c = 0; c < kResolveTail / 64; c++)
            if (lv[c]) {
              const uint8_t sj = state[ed[c] & 0xffff], si = state[ed[c] >> 16];
              blocked[ed[c] & 0xffff] = 0;
              if (sj != 0 || si == 2) lv[c] = false; else any_live = true;
            }
          wsync();
          if (__ballot(any_live) == 0ull) break;
        }
      }
      __syncthreads();
      if (a.prof && tid == 0) atomicAdd(a.prof + 11, (u64)(round + 2));
      break;
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
  uint32_t* rows = a.rows + (size_t)sb + kept_before;         // appended to the segment's kept-row list
  // Two passes: the kept positions first go to an ordered list in LDS (the edge blocks are free now),
  // then thread t emits entries t, t + 512, ... four at a time -- the gather of the original indices (or of the slab copy's
  // old positions) is four independent loads per thread instead of a chain of up to 16 dependent ones (measured: 22 us of
  // every 8192-box chunk of the uniform regime sat in this loop).
  uint32_t* olist = ledges;                                   // cn <= capmax entries fit (lcap >= capmax, checked by the launcher's LDS bound)
  __syncthreads();
  for (int q = 0; q < per_n; q++) {
    const int j = tid * per_n + q;
    if (j < cn && state[j] == 1) { olist[rank] = cidx[j]; rank++; }
  }
  __syncthreads();
  // Largest rows first (round 5).  The indexed cross phase deals (row, part) items to the workgroups in row order, and an item's
  // cost grows with the row's radius (the query windows, and the number of candidates whose circles touch): with the rows in score
  // order -- i.e. in random size order -- the last item a workgroup started was as likely a heavy one as not, and every step ended
  // with 255 workgroups waiting for it (S-uniform at 100k, in-kernel timers: the mean workgroup spent 560 us in its cross phases, the
  // slowest of every step added up to 1010 us).  The order of `rows` is free: the cross phases only need the SET (kills are
  // idempotent), the output order comes from `olist`.  Four buckets by radius relative to the chunk's largest, in bucket order.
  uint32_t* lcode = olist + a.capmax;                         // [total] bucket << 16 | slot in the bucket   (lcap >= 3 capmax + 8: checked below)
  int* lcnt = reinterpret_cast<int*>(olist + 3 * (size_t)a.capmax);   // [0..3] bucket sizes, [4] largest radius (float bits; radii are >= 0)
  const bool lpt = (a.lpt & 1) != 0 && a.gmeta != nullptr && a.nseg == 1 && a.keep_out != nullptr && total >= 512 && lcap >= 3LL * a.capmax + 8;
  if (lpt) {
    if (tid < 5) lcnt[tid] = 0;
    __syncthreads();
    int mx = 0;
    // (the radius travels together with the original index the output needs: one round trip per four entries, as without the
    //  ordering -- a first version fetched them in two passes and its resolver ran 12 us per step longer, most of what it saved)
    for (int k0 = tid; k0 < total; k0 += 4 * kNmsThreads) {
      float rv[4];
      uint32_t ov[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * kNmsThreads;
        const uint32_t pos = k < total ? olist[k] : 0u;
        rv[u] = k < total ? a.rec[(size_t)pos * 4].z : 0.f;
        ov[u] = (k < total && a.order) ? a.order[pos] : pos;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * kNmsThreads;
        if (k < total) {
          const int rb = __float_as_int(rv[u] >= 0.f ? rv[u] : 0.f);                  // (NaN: 0)
          lcode[k] = (uint32_t)rb; mx = rb > mx ? rb : mx;
          const long long o = (long long)kept_before + k;
          if (a.max_keep <= 0 || o < a.max_keep) a.keep_out[(size_t)sb + o] = (int64_t)ov[u];
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
    if ((tid & 63) == 0 && mx > 0) atomicMax(&lcnt[4], mx);
    __syncthreads();
    const float rmax = __int_as_float(lcnt[4]);
    for (int k0 = 0; k0 < total; k0 += kNmsThreads) {          // (wave-uniform trip count: ballots inside)
      const int k = k0 + tid;
      int b = -1;
      if (k < total) { const float r = __int_as_float((int)lcode[k]); b = r >= 0.75f * rmax ? 0 : (r >= 0.5f * rmax ? 1 : (r >= 0.3f * rmax ? 2 : 3)); }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u64 m = __ballot(b == q);
        if (m) {
          int base = 0;
          if ((tid & 63) == (int)__builtin_ctzll(m)) base = atomicAdd(&lcnt[q], __popcll(m));
          base = __shfl(base, (int)__builtin_ctzll(m));
          if (b == q) lcode[k] = ((uint32_t)q << 16) | (uint32_t)(base + __popcll(m & lanemask_lt()));
        }
      }
    }
    __syncthreads();
  }
  if (lpt) {                                                    // the rows in bucket order: scattered in LDS, stored coalesced
    uint32_t* olist2 = lcode + a.capmax;                        // (behind lcode; lcnt moved behind it: 3 capmax + 8 <= lcap)
    const int lb1 = lcnt[0], lb2 = lb1 + lcnt[1], lb3 = lb2 + lcnt[2];
    for (int k = tid; k < total; k += kNmsThreads) {
      const uint32_t c = lcode[k];
      const int q = (int)(c >> 16);
      olist2[(q == 0 ? 0 : (q == 1 ? lb1 : (q == 2 ? lb2 : lb3))) + (int)(c & 0xffffu)] = olist[k];
    }
    __syncthreads();
    for (int k = tid; k < total; k += kNmsThreads) stg_agent(rows + k, olist2[k]);
  }
  for (int k0 = tid; k0 < total && !lpt; k0 += 4 * kNmsThreads) {
    uint32_t pv[4], ov[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u * kNmsThreads;
      ok[u] = k < total;
      pv[u] = ok[u] ? olist[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t pos = pv[u];
      ov[u] = 0u;
      if (ok[u]) ov[u] = a.keep_out != nullptr ? (a.order ? a.order[pos] : pos) : a.pos_old[pos];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (!ok[u]) continue;
      const int k = k0 + u * kNmsThreads;
      const uint32_t pos = pv[u];
      stg_agent(rows + k, pos);
      const long long o = (long long)kept_before + k;
      if (a.keep_out != nullptr) {
        if (a.max_keep <= 0 || o < a.max_keep) a.keep_out[(size_t)sb + o] = (int64_t)ov[u];
      } else {
        // slab mode: the kept boxes of all slabs meet again in score order through a bitmap over the ORIGINAL sorted
        // positions (slab_merge); a returning atomic whose result is consumed: the wave's vmcnt covers the completed update
        const u64 old = atomicOr(a.kept_bits + (ov[u] >> 6), 1ull << (ov[u] & 63));
        asm volatile("; kept bit set %0" ::"v"((unsigned)(old >> 32) ^ (unsigned)old));
      }
    }
  }
  __syncthreads();
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
// Exhaustive form: every row against every column.  Columns are either the positions [c0, se) (clist == NULL: one wave
// per 64-position word of the bitmap, one atomicAnd per word) or the entries of a position list that fall into
// [c0, se) (the boxes the spatial index leaves out, grid.h; one atomicAnd per killed box).
template <class G, bool FN>
__device__ __forceinline__ void nms_cross(const NmsArgs& a, const uint32_t* rows, int nr, int c0, int se, const uint32_t* clist, int ncl,
                                          int tw, int ntw, WaveLds<G>& L) {
  const bool LIST = clist != nullptr;            // (wave-uniform; one body serves both forms)
  const int lane = threadIdx.x & 63;
  const int w0 = LIST ? 0 : (c0 >> 6), w1 = LIST ? ((ncl + 63) >> 6) - 1 : ((se - 1) >> 6);
  const int ncw = w1 - w0 + 1, nrt = (nr + 63) >> 6;
  if (ncw <= 0 || nr <= 0) return;
  // one wave per column word; the row tiles of a word are split over rgn waves when that shortens the longest wave:
  // rounds of items per wave x (row tiles per item x ~3 + 1 for the item's own loads), smallest over rgn = 1 .. 16
  // (a slab team of 14 workgroups with 57 column words and 5 row tiles: 1 round x 5 tiles -> 3 rounds x 1 tile)
  int rgn = 1;
  {
    long long best = -1;
    const int rmax = nrt < 16 ? nrt : 16;
    for (int r = 1; r <= rmax; r++) {
      const long long rounds = ((long long)ncw * r + ntw - 1) / ntw;
      const long long cost = rounds * ((long long)((nrt + r - 1) / r) * 3 + 1);
      if (best < 0 || cost < best) { best = cost; rgn = r; }
    }
  }
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
    const int rt_lo = rgi * rt_per, rt_hi = min(nrt, rt_lo + rt_per);
    if (rt_lo >= rt_hi) continue;
    // first row tile's loads are issued before the (slow, write-through) bitmap word arrives
    uint32_t rp0 = (rt_lo * 64 + lane < nr) ? rows[rt_lo * 64 + lane] : 0u;
    uint32_t c;                                            // this lane's column position
    bool alive0;
    if (LIST) {
      const int ci = w * 64 + lane;
      c = ci < ncl ? clist[ci] : 0u;
      alive0 = ci < ncl && (int)c >= c0 && (int)c < se;
      if (alive0) alive0 = (ldg_agent(a.alive + (c >> 6)) >> (c & 63)) & 1ull;
      if (__ballot(alive0) == 0ull) continue;
    } else {
      const int cbase = w * 64;
      c = (uint32_t)(cbase + lane);
      u64 m = ldg_agent(a.alive + w);
      if (cbase < c0) m &= ~((1ull << (c0 - cbase)) - 1ull);
      if (cbase + 64 > se) m &= (1ull << (se - cbase)) - 1ull;
      if (m == 0ull) continue;
      alive0 = (m >> lane) & 1ull;
    }
    const float4 cq = alive0 ? a.rec[(size_t)c * G::RECQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    [[maybe_unused]] ColPk cpk;
    if constexpr (G::PACKED) cpk = col_splat(cq);
    float4 rq0 = a.rec[(size_t)rp0 * G::RECQ];
    uint32_t rp1 = (rt_lo + 1 < rt_hi && (rt_lo + 1) * 64 + lane < nr) ? rows[(rt_lo + 1) * 64 + lane] : 0u;
    bool alive = alive0;
    wave_sync();
    L.cdead[lane] = alive0 ? 0 : 1;
    L.colpos[lane] = c;
    auto drain2 = [&](int cnt) {                   // stage 2: exact clip; entries carry the row position itself
      wave_sync();
      {
        uint32_t rowp = 0, colp = 0;
        int cc = 0;
        bool want = false;
        if (lane < cnt) {
          const int slot = (Q2.head + lane) & 127;
          rowp = L.qbuf2[slot];
          cc = L.q2col[slot];
          want = !L.cdead[cc];
          colp = L.colpos[cc];
        }
        if (stage_exact_wave<G, FN>(want, a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)colp * G::RECQ, G::thr_of(a), L.scr)) L.cdead[cc] = 1;
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
          res = stage_full<G, FN>(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)L.colpos[cc] * G::RECQ, G::thr_of(a));
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
            res = G::classify_quick(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)L.colpos[cc] * G::RECQ, G::thr_of(a), cull);
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
        if constexpr (G::PACKED) {               // two rows per packed instruction (geom.h)
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            bool r0, r1;
            rows_reject2<G>(myrow, rr + k, cpk, r0, r1);
            ps[k] = alive && !(cull && r0);
            ps[k + 1] = alive && !(cull && r1);
            any = any || ps[k] || ps[k + 1];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float4 rq = rdlane4(myrow, rr + k);
            ps[k] = alive && !(cull && G::cheap_reject(rq, cq));
            any = any || ps[k];
          }
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
    if (Q1.count > 0 && Q1.count + Q2.count <= 64) {     // a few leftovers: straight to the exact clip (see nms_pairs)
      wave_sync();
      const bool mv = lane < Q1.count;
      const int sl = (Q1.head + lane) & 127;
      const uint32_t rowp = mv ? L.qbuf1b[sl] : 0u;
      const uint32_t cc = mv ? (uint32_t)L.q1bcol[sl] : 0u;
      Q1.head = (Q1.head + Q1.count) & 127;
      Q1.count = 0;
      push2(mv, rowp, cc);
      wave_sync();
    }
    if (Q1.count > 0) { ctick(); drain1b(Q1.count); ctock(c_d1); c_n1++; alive = alive && !L.cdead[lane]; }
    if (Q2.count > 0) { ctick(); drain2(Q2.count); ctock(c_d2); c_n2++; alive = alive && !L.cdead[lane]; }
    // RETURNING atomics whose result is consumed: the wave's vmcnt then covers the completed read-modify-write
    if (LIST) {
      if (alive0 && !alive) {
        const u64 old = atomicAnd(a.alive + (c >> 6), ~(1ull << (c & 63)));
        asm volatile("; kill applied %0" ::"v"((unsigned)(old >> 32) ^ (unsigned)old));
      }
    } else {
      const u64 kill = __ballot(alive0 && !alive);
      if (kill && lane == 0) {
        const u64 old = atomicAnd(a.alive + w, ~kill);
        asm volatile("; kill applied %0" ::"v"((unsigned)(old >> 32) ^ (unsigned)old));
      }
    }
  }
  if (cprof) {
    a.prof[16] += c_loop; a.prof[17] += c_d1; a.prof[18] += c_d2; a.prof[19] += c_n1; a.prof[20] += c_n2; a.prof[21] += c_items;
  }
}

#ifndef OBB_SCAN_BATCH
#define OBB_SCAN_BATCH 4
#endif
constexpr int kScanBatch = OBB_SCAN_BATCH;          // blocks of candidates in flight per wave
// Indexed form (grid.h): a kept row only needs the boxes of the cells around it.  The boxes of the call sit once more in
// CELL order (built inside the kernel: count, scan, scatter); a query window is a few cell rows per level, and the boxes of
// one cell row are ONE contiguous range of that array (the slot hash is linear in cx), read coalesced by the 64 lanes
// in blocks of 64 entries.  Candidates that lie behind the chunk (position in [c0, se)) and whose circle touches the
// row's go to the same staged decision as above -- cheap bounds on full waves, the IoU interval, the exact clip -- with
// explicit (row position, column position) entries; the alive bit is checked when an entry is taken out of a queue (a
// box is usually dead by the time the second of two overlapping kept rows gets to it), kills are one atomicAnd per box.
// Brute rows (flag in quad 3) are skipped here: the caller runs the exhaustive form for them.
//
// Work distribution: an item is (row, part): part j of kw takes every kw-th block of the row's candidate ranges, so that
// a step with few kept rows still occupies every wave of the team; items go round-robin over the team's waves.  Per item
// the blocks are first listed in LDS (lanes = cell rows of a window, in parallel), then read four at a time -- four
// independent 16-byte loads per lane in flight: the phase is bound by memory latency, not by arithmetic.
//
// Measured and not kept (round 4, steps of the 100k call, us; this form = 547 at K=3000 / 1553 uniform on the same box):
//  * workgroup batches of eight rows, ONE set-up per row (the parts above repeat it: 4-7 us each), its 64-lane range lists
//    left in LDS, all eight waves drawing blocks of all eight rows from one LDS counter, batches handed out by an agent-scope
//    ticket: correct, 633 / 1612.  The set-up did shrink (52 -> 3 us per wave at K=3000), but every drawn block paid a
//    two-level search of the lists (ballot + four read-lanes + four LDS reads) and a batch cost three workgroup barriers;
//  * the same with explicit block descriptors (first entry | count) written by the producer waves into fixed 512-entry
//    parts of an LDS queue, rows per batch adapted to the phase (nr / 2T, at most eight): 572 / 1624;
//  * eight blocks in flight instead of four: 605 / 1516 on the first variant, 604 / 1912 on the second (registers).
// The counters say why none of it pays: the blocks are 72 % full (46 of 64 lanes), a wave gets through one in 0.43 us
// whichever way it is handed out -- the time is the round trip of the loads at two waves per SIMD (250 VGPRs), not the
// hand-out.  What would pay is fewer blocks per row (now ~50: a tighter index), not a cheaper way to reach them.
template <class G>
__device__ __forceinline__ void nms_cross_grid(const NmsArgs& a, const GridPlan& gp, uint32_t level_mask, const uint32_t* rows, int nr,
                                               int c0, int se, int tw, int ntw, WaveLds<G>& L, int* s_next) {
  const int lane = threadIdx.x & 63;
  const uint32_t mmask = a.gmask;
  const int M = (int)mmask + 1;
  int kw = (3 * ntw + nr - 1) / nr;                // (measured: ~3 items per wave balance best; fewer, larger items were slower)
  kw = kw < 1 ? 1 : (kw > 16 ? 16 : kw);
  const int n_items = nr * kw;
  const bool cprof = a.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  u64 ct0 = 0, c_pro = 0, c_scan = 0, c_drain = 0, c_nd = 0, c_items = 0, c_blocks = 0, c_pass = 0;
  auto ctick = [&]() { if (cprof) ct0 = wall_clock64(); };
  auto ctock = [&](u64& acc) { if (cprof) acc += wall_clock64() - ct0; };
  uint32_t* blist = L.rowpos;                      // rowpos[64] | colpos[64]: 128 block entries (first entry << 7 | entries)
  PairQueue Q{L.qbuf, 0, 0}, Q1{L.qbuf1b, 0, 0}, Q2{L.qbuf2, 0, 0};
  auto col_alive = [&](uint32_t cp) -> bool { return (ldg_agent(a.alive + (cp >> 6)) >> (cp & 63)) & 1ull; };
  // kills: RETURNING atomics whose results are consumed at the end of the phase -- the wave's vmcnt then covers the
  // completed read-modify-writes, and no drain waits for its own
  u64 seen = 0ull;
  auto kill = [&](bool hit, uint32_t cp) {
    if (hit) {
      seen ^= atomicAnd(a.alive + (cp >> 6), ~(1ull << (cp & 63)));
      // ... and the box's entry of the cell-order array gets its "dead" bit (top bit of the position word): later scans drop
      // it with the circle test instead of fetching its alive word.  Write-through; a reader that still sees the old entry
      // queues a dead box, which the first decision stage then finds dead: the flag is a filter, the bitmap is the truth
      stg_agent(reinterpret_cast<uint32_t*>(a.gsorted + a.gslot[cp]) + 3, cp | 0x80000000u);
    }
  };
  auto drain2 = [&](int cnt) {                     // stage 2: exact clip
    wave_sync();
    uint32_t cp = 0, rowp = 0;
    bool want = false;
    if (lane < cnt) {
      const int slot = (Q2.head + lane) & 127;
      rowp = L.qbuf2[slot];
      cp = L.qcol2[slot];
      want = col_alive(cp);
    }
    const bool hit = stage_exact_wave<G, true>(want, a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)cp * G::RECQ, G::thr_of(a), L.scr);
    kill(hit, cp);
    Q2.head = (Q2.head + cnt) & 127;
    Q2.count -= cnt;
    wave_sync();
  };
  auto push2 = [&](bool undecided, uint32_t rowp, uint32_t cp) {
    const u64 m2 = __ballot(undecided);
    if (undecided) {
      const int slot = (Q2.head + Q2.count + __popcll(m2 & lanemask_lt())) & 127;
      L.qbuf2[slot] = rowp;
      L.qcol2[slot] = cp;
    }
    Q2.count += __popcll(m2);
  };
  auto drain1b = [&](int cnt) {                    // stage 1b: the IoU interval
    wave_sync();
    int res = 0;
    uint32_t rowp = 0, cp = 0;
    if (lane < cnt) {
      const int slot = (Q1.head + lane) & 127;
      rowp = L.qbuf1b[slot];
      cp = L.qcol1b[slot];
      if (col_alive(cp)) res = nms_stage_full<G>(a.rec + (size_t)rowp * G::RECQ, a.rec + (size_t)cp * G::RECQ, G::thr_of(a));
    }
    kill(res == 1, cp);
    Q1.head = (Q1.head + cnt) & 127;
    Q1.count -= cnt;
    push2(res == 2, rowp, cp);
    wave_sync();
    if (Q2.count >= 64) drain2(64);
  };

  // Work distribution: workgroup w of the team owns the items w, w + T, w + 2T, ...; its eight waves take them from a
  // counter in LDS, one at a time.  An item costs between a few and a hundred-odd blocks (small box in an empty corner /
  // large box in a crowd), a wave gets 3-6 of them: with a fixed deal per wave the slowest of 2048 waves took more than
  // twice the average (uniform, 100k: 2069 -> 1841 us with the pooled deal).  Tickets of four rows drawn by the
  // workgroups from one agent-scope counter were measured as well (1847 us): the tail of a phase is one ITEM long
  // (30-80 us), whoever draws it -- not kept.
  // (Round 5, with the rows largest first -- nms_resolve -- so that the LAST items are the cheap ones: the last quarter of the items
  //  drawn by the waves from one agent-scope counter, four items per draw, the rest dealt statically.  Measured on one box: S-uniform
  //  1.83-1.86 ms against 1.575 with the static deal alone, K=3000 0.66 against 0.63 -- a draw of four items is a coarser tail than
  //  one item of the static deal, and the waves of all workgroups queue on one address.  Not kept.)
  const int wgi = tw / kNmsWaves, Tw = ntw / kNmsWaves;
  __syncthreads();
  if (threadIdx.x == 0) *s_next = 0;
  __syncthreads();
  // (holding the NEXT item too, so that its row position and record travel while the current item is scanned, was measured:
  //  no gain -- K=3000 597 against 582 us, uniform 1723 against 1654)
  for (;;) {
    int k = 0;
    if (lane == 0) k = atomicAdd(s_next, 1);
    k = __builtin_amdgcn_readfirstlane(k);
    const long long item_ll = (long long)wgi + (long long)k * Tw;
    if (item_ll >= n_items) break;
    const int item = (int)item_ll;
    const int row = item / kw, part = item - row * kw;
    c_items++;
    ctick();
    const uint32_t rp = rows[row];
    const float4 rq = a.rec[(size_t)rp * G::RECQ];
    if (grid_is_brute(gp, rq.x, rq.y, rq.z, rq.w)) continue;     // brute row: the caller runs the exhaustive form for it
    auto drain = [&](int cnt) {                    // stage 1a: the cheap register-only tests (entries: column positions)
      wave_sync();
      int res = 0;
      uint32_t cp = 0;
      if (lane < cnt) {
        cp = L.qbuf[(Q.head + lane) & 127];              // (queued without a look at the bitmap: the scan only saw the dead flag)
        // the bitmap word travels together with the two records (no branch around the test: a dependent round trip less);
        // a box that died in the meantime just loses its result
        // (a plain, cacheable load: a stale word can only show a dead box as alive -- bits go 1 -> 0 -- which costs a redundant
        //  test and an idempotent kill, never a missed one; the coherent load is a round trip to memory)
        const bool live = (a.alive[cp >> 6] >> (cp & 63)) & 1ull;
        res = G::classify_quick(a.rec + (size_t)rp * G::RECQ, a.rec + (size_t)cp * G::RECQ, G::thr_of(a), true);
        if (!live) res = 0;
      }
      kill(res == 1, cp);
      Q.head = (Q.head + cnt) & 127;
      Q.count -= cnt;
      {
        const u64 m1 = __ballot(res == 3);
        if (res == 3) {
          const int slot = (Q1.head + Q1.count + __popcll(m1 & lanemask_lt())) & 127;
          L.qbuf1b[slot] = rp;
          L.qcol1b[slot] = cp;
        }
        Q1.count += __popcll(m1);
      }
      push2(res == 2, rp, cp);
      wave_sync();
      if (Q1.count >= 64) drain1b(64);
      if (Q2.count >= 64) drain2(64);
    };
    // the window of every level first: a window with more cells than half the table has slots would visit every entry
    // several times -- then the whole array is scanned once instead (every indexed box is a candidate)
    bool whole = false;
    for (uint32_t lm = level_mask; lm; lm &= lm - 1) {
      const int lv = __builtin_ctz(lm);
      const float nc = floorf(2.f * grid_query_halfwidth(gp, lv, rq.x, rq.y, rq.z) * grid_level_inv_cell(gp, lv)) + 2.f;
      if (!(nc * nc < 0.5f * (float)M)) whole = true;
    }
    // Lanes = the cell rows of the windows of ALL levels (one table look-up round trip for the whole row, not one per
    // level); more than 64 of them: several passes.
    int ncomb = 0;
    if (whole) ncomb = 1;
    else
      for (uint32_t lm = level_mask; lm; lm &= lm - 1) {
        const int lv = __builtin_ctz(lm);
        const float inv = grid_level_inv_cell(gp, lv);
        const float d = grid_query_halfwidth(gp, lv, rq.x, rq.y, rq.z);
        const int lasty = grid_last_cell(gp.yr, inv);
        ncomb += grid_cell(rq.y + d, gp.y0, inv, lasty) - grid_cell(rq.y - d, gp.y0, inv, lasty) + 1;
      }
    int bc = 0;                                    // blocks of this row listed so far (all parts)
    for (int cb = 0; cb < ncomb; cb += 64) {
      // this lane's cell row: its slots [i0, i0 + len) (possibly wrapping) -> up to two pieces [s, e), [0, e2)
      int s = 0, e = 0, e2 = 0;
      if (whole) {
        if (lane == 0) e = a.gstart[M];
      } else {
        const int my = cb + lane;
        int before = 0, my_lv = -1, my_cx0 = 0, my_cy = 0, my_len = 0;
        for (uint32_t lm = level_mask; lm; lm &= lm - 1) {
          const int lv = __builtin_ctz(lm);
          const float inv = grid_level_inv_cell(gp, lv);
          const float d = grid_query_halfwidth(gp, lv, rq.x, rq.y, rq.z);
          const int lastx = grid_last_cell(gp.xr, inv), lasty = grid_last_cell(gp.yr, inv);
          const int cx0 = grid_cell(rq.x - d, gp.x0, inv, lastx), cy0 = grid_cell(rq.y - d, gp.y0, inv, lasty);
          const int len = grid_cell(rq.x + d, gp.x0, inv, lastx) - cx0 + 1;
          const int nyr = grid_cell(rq.y + d, gp.y0, inv, lasty) - cy0 + 1;
          if (my >= before && my < before + nyr) { my_lv = lv; my_cx0 = cx0; my_cy = cy0 + (my - before); my_len = len; }
          before += nyr;
        }
        if (my_lv >= 0) {
          const uint32_t i0 = grid_slot(my_lv, my_cx0, my_cy, mmask);
          const int e1 = (int)i0 + my_len;
          s = a.gstart[i0];
          e = a.gstart[e1 <= M ? e1 : M];
          if (e1 > M) e2 = a.gstart[e1 - M];
        }
      }
      {
        const int nb1 = e > s ? (e - s + 63) >> 6 : 0, nb2 = (e2 + 63) >> 6;
        int incl = nb1 + nb2;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        const int o0 = bc + incl - (nb1 + nb2);    // ordinal of this lane's first block
        bc += tot;
        // owned ordinals: o % kw == part; their ranks (o - part) / kw run on from the previous batch without gaps
        const int first_rank = (o0 - part + kw - 1) / kw;                        // (>= 0: o0 >= 0, part < kw)
        const int rank_lo = (bc - tot - part + kw - 1) / kw, rank_hi = (bc - part + kw - 1) / kw;   // ranks [rank_lo, rank_hi) belong to this batch
        for (int base = rank_lo; base < rank_hi; base += 128) {
          wave_sync();
          {
            int o = first_rank * kw + part;        // this lane's first owned ordinal
            for (; o < o0 + nb1 + nb2; o += kw) {
              const int rk = (o - part) / kw - base;
              if (rk < 0) continue;
              if (rk >= 128) break;
              const int b = o - o0;
              const int k0 = b < nb1 ? s + b * 64 : (b - nb1) * 64, kend = b < nb1 ? e : e2;
              blist[rk] = ((uint32_t)k0 << 7) | (uint32_t)(min(64, kend - k0));
            }
          }
          wave_sync();
          const int cnt = min(128, rank_hi - base);
          c_blocks += (u64)cnt;
          ctock(c_pro);
          ctick();
          for (int i = 0; i < cnt; i += kScanBatch) {
            uint32_t be[kScanBatch];
            float4 cq[kScanBatch];
            bool val[kScanBatch];
#pragma unroll
            for (int u = 0; u < kScanBatch; u++) {
              be[u] = (i + u < cnt) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)blist[i + u]) : 0u;
              val[u] = lane < (int)(be[u] & 127u);
              cq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (val[u]) {
#ifdef OBB_GRID_COHERENT_LOADS
                const u64* src = reinterpret_cast<const u64*>(a.gsorted + (size_t)(be[u] >> 7) + lane);
                const u64 lo = ldg_agent(src), hi = ldg_agent(src + 1);
                cq[u] = make_float4(__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                                    __uint_as_float((uint32_t)(hi >> 32)));
#else
                cq[u] = a.gsorted[(size_t)(be[u] >> 7) + lane];
#endif
              }
            }
            // circle test + the entry's dead flag (set by whoever killed the box, see `kill`): most candidates of a later
            // step are dead already and never reach a queue; what passes is queued without a look at the alive bitmap (that
            // was a second, dependent round trip per batch) -- the first decision stage checks it
#pragma unroll
            for (int u = 0; u < kScanBatch; u++) {
              const uint32_t cw = __float_as_uint(cq[u].w), cp = cw & 0x7fffffffu;
              const float dx = cq[u].x - rq.x, dy = cq[u].y - rq.y, rs = rq.z + cq[u].z;
              const bool go = val[u] && !(cw >> 31) && (int)cp >= c0 && (int)cp < se && !(dx * dx + dy * dy > rs * rs);
              if (__ballot(go)) {
                c_pass += (u64)__popcll(__ballot(go));
                Q.push(go, cp);
                if (Q.count >= 64) { ctock(c_scan); ctick(); drain(64); ctock(c_drain); c_nd++; ctick(); }
              }
            }
          }
          ctock(c_scan);
          ctick();
        }
      }
    }
    ctock(c_pro);
    ctick();
    if (Q.count > 0) { drain(Q.count); c_nd++; }
    ctock(c_drain);
  }
  if (cprof) {
    a.prof[16] += c_scan; a.prof[17] += c_drain; a.prof[18] += c_pro; a.prof[19] += c_nd; a.prof[20] += c_blocks; a.prof[21] += c_items;
    a.prof[15] += c_pass; a.prof[10] += (u64)kw;
  }
  if (Q1.count > 0 && Q1.count + Q2.count <= 64) {       // a few leftovers: straight to the exact clip (see nms_pairs)
    wave_sync();
    const bool mv = lane < Q1.count;
    const int sl = (Q1.head + lane) & 127;
    const uint32_t rowp = mv ? L.qbuf1b[sl] : 0u, cp = mv ? L.qcol1b[sl] : 0u;
    Q1.head = (Q1.head + Q1.count) & 127;
    Q1.count = 0;
    push2(mv, rowp, cp);
    wave_sync();
  }
  if (Q1.count > 0) drain1b(Q1.count);
  if (Q2.count > 0) drain2(Q2.count);
  asm volatile("; kills applied %0" ::"v"((unsigned)(seen >> 32) ^ (unsigned)seen));
}

constexpr int kGridMinRows = 512;

// Counting sort of the still-alive positions [c0, n) by table slot, by the whole team (= the whole grid: one segment):
// classify + count -> barrier -> distributed scan (every workgroup a slice of the table, then the prefix of the workgroup
// totals) -> barrier -> scatter -> barrier.  A box is classified on the fly from quad 0 (grid.h): brute boxes are appended
// to the brute list, the others counted into their table slot.  Everything another workgroup reads afterwards is written
// through (agent scope); the readers take ONE agent acquire after the last barrier and use plain loads from then on
// (guideline 16).  The slot counters are back to zero when the scatter is done.
// Returns 0: built; 1: barrier abort; 2: not worth using (no usable extent, or more than 1/16 of the boxes are brute).
template <class G>
OBB_COLD_GRID int grid_build(const NmsArgs& a, const GridPlan& gp, int c0, int wg, int T, TeamBar& bar, int* s_flag, int* s_i,
                                          uint32_t& level_mask, int& n_brute) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int M = (int)a.gmask + 1;
  const int per = (M + T - 1) / T;                 // table slots of one workgroup (<= kNmsThreads, checked by the caller)
  // 0: not a column any more (dead), 1: indexed (slot), 2: brute.  Coherent alive read: count and scatter must agree.
  auto classify = [&](int p, uint32_t& slot, float4& q0, int& lv) -> int {
    if (!((ldg_agent(a.alive + (p >> 6)) >> (p & 63)) & 1ull)) return 0;
    q0 = a.rec[(size_t)p * G::RECQ];
    if (grid_is_brute(gp, q0.x, q0.y, q0.z, q0.w)) return 2;
    lv = grid_level(gp, q0.z);
    const float inv = grid_level_inv_cell(gp, lv);
    slot = grid_slot(lv, grid_cell(q0.x, gp.x0, inv, grid_last_cell(gp.xr, inv)), grid_cell(q0.y, gp.y0, inv, grid_last_cell(gp.yr, inv)), a.gmask);
    return 1;
  };
  uint32_t lbits = 0u;
  for (int p0 = c0 + wg * kNmsThreads; p0 < a.n; p0 += T * kNmsThreads) {
    const int p = p0 + tid;
    uint32_t slot = 0u;
    float4 q0;
    int lv = 0;
    const int kind = p < a.n ? classify(p, slot, q0, lv) : 0;
    if (kind == 1) { __hip_atomic_fetch_add(a.gcnt + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); lbits |= 1u << lv; }
    const u64 bm = __ballot(kind == 2);
    if (bm) {
      int base = 0;
      if (lane == 0) base = __hip_atomic_fetch_add(&a.gmeta->n_brute, __popcll(bm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      base = __shfl(base, 0);
      if (kind == 2) stg_agent(a.ulist + base + __popcll(bm & lanemask_lt()), (uint32_t)p);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) lbits |= __shfl_xor(lbits, d);
  if (lane == 0 && lbits) __hip_atomic_fetch_or(&a.gmeta->level_mask, lbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!team_barrier(bar, s_flag)) return 1;
  level_mask = ldg_agent(&a.gmeta->level_mask);
  n_brute = ldg_agent(&a.gmeta->n_brute);
  if (level_mask == 0u || (long long)n_brute * 16 > (long long)(a.n - c0)) return 2;     // (uniform decision: same values everywhere)
  // exclusive prefix of this workgroup's slice + its total
  auto block_excl = [&](int v, int& total) -> int {
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    __syncthreads();
    if (lane == 63) s_i[wv] = incl;
    __syncthreads();
    int pre = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < kNmsWaves; k++) { const int t = s_i[k]; if (k < wv) pre += t; total += t; }
    return pre + incl - v;
  };
  const int slot_i = wg * per + tid;
  const bool mine = tid < per && slot_i < M;
  int tot_wg = 0;
  const int excl = block_excl(mine ? ldg_agent(a.gcnt + slot_i) : 0, tot_wg);
  if (tid == 0) stg_agent(a.gwsum + wg, tot_wg);
  if (!team_barrier(bar, s_flag)) return 1;
  int grand = 0, base = 0;
  {
    const int v = tid < T ? ldg_agent(a.gwsum + tid) : 0;       // T <= kNmsThreads
    const int ex = block_excl(v, grand);
    __syncthreads();
    if (tid == wg) s_i[11] = ex;
    __syncthreads();
    base = s_i[11];
  }
  if (mine) stg_agent(a.gstart + slot_i, base + excl);
  if (wg == 0 && tid < 2) stg_agent(a.gstart + M + tid, grand);
  if (!team_barrier(bar, s_flag)) return 1;
  for (int p = c0 + wg * kNmsThreads + tid; p < a.n; p += T * kNmsThreads) {
    uint32_t slot = 0u;
    float4 q0;
    int lv = 0;
    if (classify(p, slot, q0, lv) == 1) {
      const int k = __hip_atomic_fetch_sub(a.gcnt + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;   // the order inside a slot does not matter
      const int at = ldg_agent(a.gstart + slot) + k;
      u64* o = reinterpret_cast<u64*>(a.gsorted + (size_t)at);
      stg_agent(o, ((u64)__float_as_uint(q0.y) << 32) | (u64)__float_as_uint(q0.x));
      stg_agent(o + 1, ((u64)(uint32_t)p << 32) | (u64)__float_as_uint(q0.z));
      stg_agent(a.gslot + p, (uint32_t)at);                    // position -> entry (the killers flag the entry)
    }
  }
  if (!team_barrier(bar, s_flag)) return 1;
  // ONE acquire per workgroup (the caches it invalidates are the CU's and the XCD's, not the wave's), as in serial_wait: with every
  // wave of every workgroup issuing its own, 2048 invalidates queued up behind each other -- ~14 us in k_slab_split, where the
  // timers showed it (round 5).
  // What relies on it: everything another workgroup wrote before this barrier is read with PLAIN loads afterwards --
  // gstart / gsorted / gslot / the unindexed list here, the slab-major records, order, old positions and alive2 behind slab_setup's
  // barrier (which has no fence of its own: its readers use ldg_agent on slab_cnt / slab_tot), kept_bits behind slab_merge's.  The
  // writers publish with write-through stores (stg_agent) or device atomics; a NEW plain read of such a buffer behind one of these
  // barriers is only safe behind this fence + __syncthreads, a new read without them must be an ldg_agent.  A build with
  // -DOBB_NMS_FULL_FENCE (make HIPCC="hipcc -DOBB_NMS_FULL_FENCE") issues the fence from every wave again: the A/B for a suspected
  // stale read.
#ifdef OBB_NMS_FULL_FENCE
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  __syncthreads();
  return 0;
}


// ------------------------------------------------------------------ independent slabs (grid.h): set-up and merge
// What the set-up kernel (k_slab_split) leaves behind for the persistent kernel, in global memory
struct SlabPlan {
  int mode;                               // 0: the call stays one list; 1: slab mode
  int nslab;
  int cap; int pad0;                      // chunk capacity of one team
  long long ecap;                         // edge-list capacity of one team (the single list's buffer is shared out)
  int segb[kMaxSlabs], sege[kMaxSlabs];   // positions [segb, sege) of slab s in the slab-major layout
  int4 wg[1024];                          // per workgroup of the persistent launch: {slab or -1, team id, index in the team, team size}
};

// A launch of its own in front of the persistent kernel, same grid (round 4; rounds 2-3 ran it at the start of the
// persistent kernel: 65 us, of which the timers explained 50 -- code that runs once per call inside a 230 KB kernel is fetched
// cold, instruction line by instruction line; as a kernel of a few KB it is not, and a list that does not decompose pays
// an empty launch instead of nothing).  Leaves mode 0 (one list, nothing changed) or 1 (slab mode: the plan is filled in and the
// slab-major copy of the list is complete); a barrier abort leaves mode 0 and the abort flag.
//   1. runs of marked x bins -> slab of a bin (every workgroup from the same bitmap: same result);
//   2. every workgroup counts the alive boxes of its contiguous block of positions per slab          -> team barrier
//   3. totals, this workgroup's offsets, the 64-aligned start of every slab; uniform decision (>= 2 non-empty slabs, none
//      larger than kSlabMaxSeg, one workgroup per slab available, a useful chunk capacity);
//   4. STABLE scatter of records / original indices / original positions / alive bits: the order inside a slab is the
//      score order of the list                                                                      -> team barrier
template <class G>
__device__ __forceinline__ int slab_setup(const NmsArgs& a, float bin_x0, float inv, unsigned char* smem, float4* st_rec, uint4* st_meta, TeamBar& gbar,
                                          int* s_flag, SlabPlan& SL) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NB = gridDim.x, wg = blockIdx.x;
  uint32_t* starts = reinterpret_cast<uint32_t*>(smem);       // [kSlabWords] bins where a run starts
  int* wpre = reinterpret_cast<int*>(starts + kSlabWords);    // [kSlabWords] runs that start before the word
  int* cnt = wpre + kSlabWords;                               // [kMaxSlabs] this workgroup's boxes per slab
  int* tot = cnt + kMaxSlabs;                                 // [kMaxSlabs] all boxes of the slab
  int* pre = tot + kMaxSlabs;                                 // [kMaxSlabs] ... in workgroups below this one
  int* base = pre + kMaxSlabs;                                // [kMaxSlabs] first position of the slab
  int* run = base + kMaxSlabs;                                // [kMaxSlabs] ... placed by earlier tiles of this workgroup
  int* wcnt = run + kMaxSlabs;                                // [kNmsWaves][kMaxSlabs] per wave of the current tile
  int* misc = wcnt + kNmsWaves * kMaxSlabs;                   // [4]
  if (*a.slab_flag != 0 || NB > kNmsThreads) return 0;        // (written by the prep kernel: uniform)
  const bool sprof = a.prof != nullptr && blockIdx.x == 0 && tid == 0;
  u64 st0 = sprof ? wall_clock64() : 0ull;
  auto slap = [&](int slot) { if (sprof) { const u64 t = wall_clock64(); a.prof[slot] += t - st0; st0 = t; } };
  __syncthreads();
  int mypre = 0;
  auto cover = [&](int k) -> uint32_t {                       // the OR of the prep kernel's kSlabCopies copies
    uint32_t w = 0u;
#pragma unroll
    for (int c = 0; c < kSlabCopies; c++) w |= a.slab_cover[c * kSlabWords + k];
    return w;
  };
  if (tid < kSlabWords) {
    const uint32_t w = cover(tid);
    const uint32_t prev_msb = tid > 0 ? (cover(tid - 1) >> 31) : 0u;
    starts[tid] = w & ~((w << 1) | prev_msb);
  }
  __syncthreads();
  if (tid < kSlabWords) for (int k = 0; k < tid; k++) mypre += __popc(starts[k]);
  if (tid < kSlabWords) wpre[tid] = mypre;
  if (tid == kSlabWords - 1) misc[0] = mypre + __popc(starts[tid]);
  __syncthreads();
  const int nruns = misc[0];
  slap(42);
  if (nruns < 2) return 0;
  const int S = nruns < kMaxSlabs ? nruns : kMaxSlabs;        // (the runs beyond the last id share it: still independent of the others)
  auto slab_of = [&](float x) -> int {
    const int b = slab_bin(x, bin_x0, inv);
    int r = wpre[b >> 5] + __popc(starts[b >> 5] & (0xffffffffu >> (31 - (b & 31)))) - 1;
    return r < 0 ? 0 : (r < S ? r : S - 1);
  };
  // ---- 2: counts of this workgroup's block of positions (a multiple of 64 positions: whole words of the bitmap)
  const int chunk = (((a.n + NB - 1) / NB) + 63) & ~63;
  const int p0 = wg * chunk < a.n ? wg * chunk : a.n, p1 = (p0 + chunk < a.n) ? p0 + chunk : a.n;
  if (tid < kMaxSlabs) { cnt[tid] = 0; run[tid] = 0; tot[tid] = 0; pre[tid] = 0; }
  for (int k = tid; k < kNmsWaves * kMaxSlabs; k += kNmsThreads) wcnt[k] = 0;
  for (int k = wg * kNmsThreads + tid; k < a.alive2_words; k += NB * kNmsThreads) stg_agent(a.alive2 + k, 0ull);
  for (int k = wg * kNmsThreads + tid; k < a.kept_words; k += NB * kNmsThreads) stg_agent(a.kept_bits + k, 0ull);
  if (wg == 0 && tid < kMaxSlabs) stg_agent(a.slab_keep + tid, 0);
  __syncthreads();
  // A block of at most one tile (every list of up to 131072 boxes on a full grid): the box's record, slab and place in its wave are
  // kept in registers from this pass to the scatter behind the barrier -- which then only adds offsets and stores (round 5: the
  // scatter re-read and re-ranked everything: 14 us of a 50 us kernel).
  const bool one_tile = p1 - p0 <= kNmsThreads;                // (uniform over the grid)
  int sl1 = -1, rank1 = 0;
  uint32_t order1 = 0u;
  float4 q1[G::RECQ];
  for (int pb = p0; pb < p1; pb += kNmsThreads) {
    const int p = pb + tid;
    int sl = -1;
    if (p < p1 && ((a.alive[p >> 6] >> (p & 63)) & 1ull)) {
      if (one_tile) {
#pragma unroll
        for (int k = 0; k < G::RECQ; k++) q1[k] = a.rec[(size_t)p * G::RECQ + k];
        order1 = a.order[p];
        sl = slab_of(q1[0].x);
      } else {
        sl = slab_of(a.rec[(size_t)p * G::RECQ].x);
      }
    }
    u64 todo = __ballot(sl >= 0);
    while (todo) {                                             // one LDS atomic per (wave, slab present in it)
      const int l0 = __builtin_ctzll(todo);
      const int s0 = __shfl(sl, l0);
      const u64 m = __ballot(sl == s0);
      if (sl == s0) rank1 = __popcll(m & lanemask_lt());
      if (lane == l0) { atomicAdd(&cnt[s0], __popcll(m)); if (one_tile) wcnt[wv * kMaxSlabs + s0] = __popcll(m); }
      todo &= ~m;
    }
    sl1 = sl;
  }
  __syncthreads();
  // this workgroup's counts: its own row of the table, and -- one returning atomic each -- the slab totals and the totals of
  // its group of 16 workgroups: afterwards a workgroup needs the totals, the groups below its own and the rows of its own
  // group, not all 256 rows (that copy took 15 us)
  if (tid < kMaxSlabs) {
    const int c = cnt[tid];
    stg_agent(a.slab_cnt + (size_t)wg * kMaxSlabs + tid, c);
    if (c) {
      int seen = __hip_atomic_fetch_add(a.slab_tot + tid, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      seen ^= __hip_atomic_fetch_add(a.slab_tot + (size_t)(1 + (wg >> 4)) * kMaxSlabs + tid, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("; counts added %0" ::"v"(seen));
    }
  }
  slap(43);
  if (!team_barrier(gbar, s_flag)) return -1;
  // (no acquire fence: what other workgroups of this launch wrote -- the count table and its totals -- is read with agent-scope
  //  loads below; everything else this kernel reads comes from earlier launches.  Round 5: the fence, one cache invalidate per
  //  wave of every workgroup, was most of the 14 us this step was charged with)
  slap(44);
  // ---- 3: totals and offsets: thread (slab, part) adds up its share of the group totals below this workgroup's group and
  // of the rows of its own group below itself
  {
    const int s0 = tid & (kMaxSlabs - 1), part = tid >> 6;     // kNmsThreads / kMaxSlabs = 8 parts
    const int grp = wg >> 4;
    if (s0 < S) {
      // (at most two group totals and two rows per thread: all requested before the first is used -- the loop form waited
      //  for every load in turn, four round trips to lines other workgroups had just written: 14.5 us of the set-up)
      const int g2a = part, g2b = part + kNmsWaves, wa = (grp << 4) + part, wb = wa + kNmsWaves;
      const int v0 = g2a < grp ? ldg_agent(a.slab_tot + (size_t)(1 + g2a) * kMaxSlabs + s0) : 0;
      const int v1 = g2b < grp ? ldg_agent(a.slab_tot + (size_t)(1 + g2b) * kMaxSlabs + s0) : 0;
      const int v2 = wa < wg ? ldg_agent(a.slab_cnt + (size_t)wa * kMaxSlabs + s0) : 0;
      const int v3 = wb < wg ? ldg_agent(a.slab_cnt + (size_t)wb * kMaxSlabs + s0) : 0;
      const int vt = part == 0 ? ldg_agent(a.slab_tot + s0) : 0;
      int below = v0 + v1 + v2 + v3;
      for (int g2 = part + 2 * kNmsWaves; g2 < grp; g2 += kNmsWaves) below += ldg_agent(a.slab_tot + (size_t)(1 + g2) * kMaxSlabs + s0);   // (grids beyond 256 workgroups)
      if (below) atomicAdd(&pre[s0], below);
      if (part == 0) tot[s0] = vt;
    }
    slap(48);
  }
  __syncthreads();
  slap(49);
  if (tid == 0) {
    int acc = 0, mx = 0, nonempty = 0;
    for (int s0 = 0; s0 < S; s0++) { base[s0] = acc; acc += (tot[s0] + 63) & ~63; mx = mx > tot[s0] ? mx : tot[s0]; nonempty += tot[s0] > 0 ? 1 : 0; }
    // chunk capacity of a team: the single list's edge buffer shared out, cap (cap - 1) / 2 <= ecap / teams
    int c = 0;                                                 // largest multiple of 64 with c (c - 1) / 2 * teams <= ecap, at most capmax
    {
      const long long teams = nonempty > 0 ? nonempty : 1;
      auto fits = [&](long long v) { return v <= a.capmax && v * (v - 1) / 2 * teams <= a.ecap; };
      // (an estimate from the square root, then exact steps of 64 either way: the plain search from 0 took up to 128 dependent
      //  64-bit multiplications on one thread with every other thread of every workgroup waiting)
      long long est = (long long)sqrt(2.0 * (double)a.ecap / (double)teams);
      est = (est > a.capmax ? a.capmax : est) & ~63ll;
      while (est > 0 && !fits(est)) est -= 64;
      while (fits(est + 64)) est += 64;
      c = (int)est;
    }
    if (a.slab_cap > 0 && c > a.slab_cap) c = a.slab_cap;      // (OBB_NMS_SLAB_CAP: measurements)
    misc[1] = (mx <= kSlabMaxSeg && nonempty >= 2 && nonempty <= NB && c >= 512) ? 1 : 0;
    misc[2] = nonempty; misc[3] = c;
    if (wg == 0) { SL.cap = c; SL.ecap = (long long)c * (c - 1) / 2; }
  }
  __syncthreads();
  slap(45);
  if (!misc[1]) return 0;                                      // (uniform: every workgroup read the same table)
  // ---- 4: stable scatter: from the registers of the counting pass, or tile by tile
  if (one_tile) {
    // The boxes of this block go out slab by slab: in the copy the boxes a workgroup sends to one slab are neighbours (stable
    // scatter), so they are first put into that order in LDS and then stored by consecutive lanes -- whole runs of 64-byte
    // records instead of 4 x 64 scattered 16-byte pieces per store instruction (whose completion, ~10 us, the kernel's end waited for).
    if (tid < 64) {                                             // this workgroup's own prefix over the slabs (kMaxSlabs = 64)
      const int c = cnt[tid];
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
      run[tid] = incl - c;                                       // (run[] is free in this form: first local index of the slab)
      if (tid == 63) misc[0] = incl;
    }
    __syncthreads();
    if (sl1 >= 0) {
      int off = 0;
      for (int w2 = 0; w2 < wv; w2++) off += wcnt[w2 * kMaxSlabs + sl1];
      const int li = run[sl1] + off + rank1;
      const int qn = base[sl1] + pre[sl1] + off + rank1;
#pragma unroll
      for (int k = 0; k < G::RECQ; k++) st_rec[li * G::RECQ + k] = q1[k];
      st_meta[li] = make_uint4((uint32_t)qn, order1, (uint32_t)(p0 + tid), 0u);
    }
    __syncthreads();
    const int total = misc[0];
    for (int t = tid; t < total * G::RECQ; t += kNmsThreads) {
      const int j = t / G::RECQ, k = t - j * G::RECQ;
      a.rec2[(size_t)st_meta[j].x * G::RECQ + k] = st_rec[t];
    }
    for (int j = tid; j < total; j += kNmsThreads) {
      const uint4 mt = st_meta[j];
      a.order2[mt.x] = mt.y;
      a.pos_old[mt.x] = mt.z;
    }
    // the alive bits of the boxes this workgroup sends to a slab are ONE run of positions: an atomic per touched word (at most
    // eight per slab) instead of one per box -- 100,000 same-word atomics were ~10 us of completion time at the kernel's end
    if (tid < S && cnt[tid] > 0) {
      const int b0 = base[tid] + pre[tid], b1 = b0 + cnt[tid];
      for (int w = b0 >> 6; w <= (b1 - 1) >> 6; w++) {
        const int lo = b0 > (w << 6) ? b0 - (w << 6) : 0, hi = b1 < ((w + 1) << 6) ? b1 - (w << 6) : 64;     // bits [lo, hi) of word w
        const u64 mask = (hi == 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
        __hip_atomic_fetch_or(a.alive2 + w, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else
  for (int pb = p0; pb < p1; pb += kNmsThreads) {
    const int p = pb + tid;
    int sl = -1;
    float4 q[G::RECQ];
    if (p < p1 && ((a.alive[p >> 6] >> (p & 63)) & 1ull)) {
#pragma unroll
      for (int k = 0; k < G::RECQ; k++) q[k] = a.rec[(size_t)p * G::RECQ + k];
      sl = slab_of(q[0].x);
    }
    __syncthreads();                                           // the previous tile's readers of wcnt / writers of run are done
    for (int k = tid; k < kNmsWaves * kMaxSlabs; k += kNmsThreads) wcnt[k] = 0;
    __syncthreads();
    int rank_in_wave = 0;
    u64 todo = __ballot(sl >= 0);
    while (todo) {
      const int l0 = __builtin_ctzll(todo);
      const int s0 = __shfl(sl, l0);
      const u64 m = __ballot(sl == s0);
      if (sl == s0) rank_in_wave = __popcll(m & lanemask_lt());
      if (lane == l0) wcnt[wv * kMaxSlabs + s0] = __popcll(m);
      todo &= ~m;
    }
    __syncthreads();
    if (sl >= 0) {
      int off = run[sl];
      for (int w2 = 0; w2 < wv; w2++) off += wcnt[w2 * kMaxSlabs + sl];
      const int qn = base[sl] + pre[sl] + off + rank_in_wave;
      // (plain 16-byte stores: the copy is read by the NEXT launch -- as part of the persistent kernel, rounds 2-3, these were
      //  write-through 8-byte stores followed by a grid barrier.  The alive bits stay atomics on words zeroed write-through in
      //  front of this kernel's barrier: several workgroups share a word.)
      float4* dst = a.rec2 + (size_t)qn * G::RECQ;
#pragma unroll
      for (int k = 0; k < G::RECQ; k++) dst[k] = q[k];
      a.order2[qn] = a.order[p];
      a.pos_old[qn] = (uint32_t)p;
      __hip_atomic_fetch_or(a.alive2 + (qn >> 6), 1ull << (qn & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < kMaxSlabs) {
      int add = 0;
      for (int w2 = 0; w2 < kNmsWaves; w2++) add += wcnt[w2 * kMaxSlabs + tid];
      run[tid] += add;
    }
  }
  slap(50);
  // ---- the plan: one team per non-empty slab, the spare workgroups in proportion to the sizes (written by workgroup 0)
  __syncthreads();
  if (wg == 0) {
    // (all threads: the serial form -- thread 0 writing the NB entries -- took 10 us at the end of the one workgroup every other one
    //  had already left behind: a fifth of the kernel)
    if (tid < kMaxSlabs) { SL.segb[tid] = tid < S ? base[tid] : 0; SL.sege[tid] = tid < S ? base[tid] + tot[tid] : 0; }
    int* tsz = cnt;                                            // [kMaxSlabs] team size of a slab (0: empty), then its first workgroup in `run`
    int* tfirst = run;                                         // (both arrays are free after the scatter)
    __syncthreads();
    if (tid == 0) {
      long long total = 0;
      for (int s0 = 0; s0 < S; s0++) total += tot[s0];
      const int spare = NB - misc[2];
      int w = 0;
      for (int s0 = 0; s0 < S; s0++) {
        const int T = tot[s0] > 0 ? 1 + (int)((long long)spare * tot[s0] / total) : 0;
        tsz[s0] = T; tfirst[s0] = w; w += T;
      }
      misc[0] = w;                                             // workgroups with a slab
      SL.nslab = S;
    }
    __syncthreads();
    for (int w = tid; w < NB; w += kNmsThreads) {
      int4 e = make_int4(-1, 0, 0, 1);
      if (w < misc[0]) {
        int team = 0;
        for (int s0 = 0; s0 < S; s0++) {
          if (tsz[s0] <= 0) continue;
          if (w >= tfirst[s0] && w < tfirst[s0] + tsz[s0]) { e = make_int4(s0, team, w - tfirst[s0], tsz[s0]); break; }
          team++;
        }
      }
      SL.wg[w] = e;
    }
  }
  slap(46);
  // (No barrier behind the scatter: the set-up is a kernel of its own since round 4 and the persistent launch behind it on the
  //  stream only starts when every workgroup of this one has finished -- the barrier that used to stand here cost 9 us of a 56 us
  //  kernel.  The stores are write-through; the persistent kernel's first loads of the copy come after a kernel boundary.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  slap(47);
  return 1;
}

// the set-up as a kernel: grid = the persistent launch's grid (its per-workgroup count table), one two-level grid barrier of its own
template <class G>
__global__ __launch_bounds__(kNmsThreads) void k_slab_split(NmsArgs a, SlabPlan* sp) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * kSlabWords + 6 * kMaxSlabs + kNmsWaves * kMaxSlabs + 8) * 4];
  __shared__ float4 st_rec[kNmsThreads * G::RECQ];             // one tile of records in slab order (slab_setup, one-tile scatter)
  __shared__ uint4 st_meta[kNmsThreads];                       // {place in the copy, original index, original position}
  __shared__ int s_flag;
  if (blockIdx.x == 0 && threadIdx.x == 0) sp->mode = 0;
  if (a.slab_flag[2] == 0) return;                             // the data is not wide enough to look for slabs (k_prep_rot / the sort's record tail)
  // the barrier line of team 1 (unused by a single-list call) and the second group-counter block
  TeamBar gbar{a.bar + 128, a.bar + 128 + 64, (int)gridDim.x, 0, a.abort_flag, gridDim.x > 32 ? a.bar_sub + kBarGroups * 64 : nullptr, (int)blockIdx.x};
  const int st = slab_setup<G>(a, __int_as_float(a.slab_flag[4]), __int_as_float(a.slab_flag[5]), smem, st_rec, st_meta, gbar, &s_flag, *sp);
  if (st == 1 && blockIdx.x == 0 && threadIdx.x == 0) sp->mode = 1;
}

// After every team has finished its slab (and one more barrier of the whole grid): the kept boxes are the set bits of a
// bitmap over the ORIGINAL sorted positions; position order is score order, so the output is an ordered compaction of
// that bitmap.  Every workgroup scans all words for the ranks (n / 64 words: 1563 at N = 100k) and emits its share.
OBB_COLD_SLAB void slab_merge(const u64* kept_bits, int n, const uint32_t* order, int64_t* keep_out, int* keep_cnt, int* s_i) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NB = gridDim.x, wg = blockIdx.x;
  const int W = (n + 63) >> 6;
  const int per = (W + kNmsThreads - 1) / kNmsThreads;
  const int k0 = tid * per < W ? tid * per : W, k1 = (k0 + per < W) ? k0 + per : W;
  int mine = 0;
  for (int k = k0; k < k1; k++) mine += __popcll(kept_bits[k]);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
  __syncthreads();
  if (lane == 63) s_i[wv] = incl;
  __syncthreads();
  int wpre = 0, total = 0;
#pragma unroll
  for (int k = 0; k < kNmsWaves; k++) { const int t = s_i[k]; if (k < wv) wpre += t; total += t; }
  int rank = wpre + incl - mine;
  for (int k = k0; k < k1; k++) {
    u64 m = kept_bits[k];
    if ((k % NB) == wg) {
      int r = rank;
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t p = (uint32_t)(k * 64 + b);
        keep_out[r++] = order ? (int64_t)order[p] : (int64_t)p;
      }
    }
    rank += __popcll(kept_bits[k]);
  }
  if (wg == 0 && tid == 0) stg_agent(keep_cnt, total);
}

// ------------------------------------------------------------------ the persistent kernel
// dynamic LDS: [kNmsWaves x WaveLds<G>] (aliased by the resolve state) | chunk list [capmax] u32
// One workgroup per CU (the LDS footprint allows no second one) = 2 waves per SIMD: let the compiler use the whole
// 256-register budget of such a wave instead of spilling to scratch (measured: 20 MB of scratch writes per launch).
// GRID: with the indexed cross phase (single list); without it the kernel is the exhaustive one only.
template <class G, bool GRID>
__global__ __launch_bounds__(kNmsThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_nms_persist(NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_i[16];
  __shared__ int s_flag;
  __shared__ int s_bb[kNmsWaves][4];
  const int tid = threadIdx.x, wv = tid >> 6;
  const int NB = gridDim.x;
  if (a.resume != nullptr && a.resume->done != 0) return;     // (the phase kernels in front of this launch completed the call)
  // the extent of the data from the key kernel's per-workgroup partials (every workgroup reduces them itself)
  auto data_extent = [&]() -> GridPlan {
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = (int)0x80000000, by1 = (int)0x80000000;
    for (int i = tid; i < a.nparts; i += kNmsThreads) {
      const int4 q = reinterpret_cast<const int4*>(a.bbpart)[2 * i];          // (kBbInts = 8 ints per partial)
      bx0 = min(bx0, q.x); by0 = min(by0, q.y); bx1 = max(bx1, q.z); by1 = max(by1, q.w);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      bx0 = min(bx0, __shfl_xor(bx0, d)); by0 = min(by0, __shfl_xor(by0, d));
      bx1 = max(bx1, __shfl_xor(bx1, d)); by1 = max(by1, __shfl_xor(by1, d));
    }
    __syncthreads();
    if ((tid & 63) == 0) { s_bb[wv][0] = bx0; s_bb[wv][1] = by0; s_bb[wv][2] = bx1; s_bb[wv][3] = by1; }
    __syncthreads();
    int bb[4] = {s_bb[0][0], s_bb[0][1], s_bb[0][2], s_bb[0][3]};
    for (int k = 1; k < kNmsWaves; k++) {
      bb[0] = min(bb[0], s_bb[k][0]); bb[1] = min(bb[1], s_bb[k][1]); bb[2] = max(bb[2], s_bb[k][2]); bb[3] = max(bb[3], s_bb[k][3]);
    }
    return grid_plan(bb);
  };
  // ---- independent slabs (grid.h): a single list that falls apart into groups that cannot overlap runs as that many segments
  TeamBar gbar{a.bar, a.bar + 64, NB, 0, a.abort_flag, NB > 32 ? a.bar_sub : nullptr, (int)blockIdx.x};      // the whole grid (the barrier line of team 0)
  bool slab_mode = false;
  const uint32_t* order0 = a.order;
  int64_t* keep_out0 = a.keep_out;
  int* keep_cnt0 = a.keep_cnt;
  const int n0 = a.n;
  const SlabPlan* sp = nullptr;
  if constexpr (G::HAS_GRID && GRID) {
    // (k_slab_split ran in front of this launch and decided: one word of its plan)
    if (a.slab_plan != nullptr && a.nseg == 1 && a.slab_plan->mode == 1) {
      sp = a.slab_plan;
      slab_mode = true;
      a.rec = a.rec2; a.order = a.order2; a.alive = a.alive2; a.keep_out = nullptr; a.keep_cnt = a.slab_keep;
      a.nseg = sp->nslab; a.capmax = sp->cap; a.ecap = sp->ecap;
    }
  }
  int nteams = a.nseg < NB ? a.nseg : NB;
  int T = NB / nteams;
  int team = blockIdx.x / T, wg = blockIdx.x - team * T;
  int g_first = team, g_step = nteams, g_last = a.nseg - 1;
  int bar_line = team;
  long long skip_cost = -1;                            // >= 0: segments at least this expensive belong to a team of their own
  bool idle = false;
  if (slab_mode) {                                     // one team per slab (k_slab_split's plan); barrier lines 64.. (line 0 is the grid's)
    const int4 pl = sp->wg[blockIdx.x];
    g_step = 1; team = pl.y; wg = pl.z; T = pl.w; bar_line = 64 + team;
    if (pl.x < 0) { idle = true; g_first = 0; g_last = -1; } else { g_first = g_last = pl.x; }
  } else if (a.plan != nullptr && a.plan[0].w > 0) {   // planned (k_plan_teams): a team on one segment, or one workgroup on a run of small ones
    const int4 pl = a.plan[blockIdx.x];
    if (pl.x < 0) return;
    g_first = pl.x; g_step = 1; team = pl.y; T = pl.w; bar_line = team;
    if (T == 1) { wg = 0; g_last = pl.x + pl.z; skip_cost = (long long)(unsigned)a.plan[kPlanInfoSlot].x; }
    else { wg = pl.z; g_last = pl.x; }
  } else if (team >= nteams) {
    return;
  }
  (void)idle;
  WaveLds<G>& L = reinterpret_cast<WaveLds<G>*>(smem)[wv];
  uint32_t* cidx = reinterpret_cast<uint32_t*>(smem + sizeof(WaveLds<G>) * kNmsWaves);
  const int tw = wg * kNmsWaves + wv, ntw = T * kNmsWaves;
  TeamBar bar{a.bar + (size_t)bar_line * 128, a.bar + (size_t)bar_line * 128 + 64, T, 0, a.abort_flag,
              (bar_line == 0 && T == NB && NB > 32) ? a.bar_sub : nullptr, wg};       // (the one team of a single list shares the grid's counters)

  const bool prof = a.prof != nullptr && blockIdx.x == 0 && tid == 0;
  u64 t0 = prof ? wall_clock64() : 0ull;
  const u64 t_wg0 = (a.prof && tid == 0) ? wall_clock64() : 0ull;
  auto lap = [&](int slot) {
    if (prof) { const u64 t1 = wall_clock64(); a.prof[slot] += t1 - t0; t0 = t1; }
  };

  // spatial index (grid.h): the cross phases query it instead of scanning every alive position
  bool grid_on = false, grid_built = false, aborted = false;
  GridPlan gp = {};
  uint32_t glevels = 0;
  int n_brute = 0;
  if constexpr (G::HAS_GRID && GRID) {
    grid_on = a.gmeta != nullptr && a.nseg == 1 && T == NB && ((int)a.gmask + T) / T <= kNmsThreads && T <= kNmsThreads && a.cull != 0;
  }
  // kept rows x alive positions of [c0, c1)
  auto cross = [&](const uint32_t* rows, int nr, int c0, int c1) {
    if constexpr (G::HAS_GRID && GRID) {
      if (grid_on && nr >= kGridMinRows) {           // (a few hundred kept rows: the exhaustive form is the cheaper one)
        if (!grid_built) {                           // first use: sort what is still alive behind the chunk into its cells
          gp = data_extent();
          gp.fine = a.gfine;
          const int st = gp.ok ? grid_build<G>(a, gp, c0, wg, T, bar, &s_flag, s_i, glevels, n_brute) : 2;
          if (st == 1) { aborted = true; return; }
          grid_built = true;
          if (st == 2) grid_on = false;                // no usable extent / too many brute boxes: exhaustive from here on
        }
      }
      if (grid_on && nr >= kGridMinRows) {
        nms_cross_grid<G>(a, gp, glevels, rows, nr, c0, c1, tw, ntw, L, &s_i[12]);
        if (n_brute > 0) {
          // the boxes the index leaves out: brute kept rows against every column, every kept row against the brute columns
          // (the chunk list in LDS is free between resolve and the next select: it takes the brute rows, capmax at a time)
          // (every workgroup builds the SAME list in the SAME order -- ordered compaction, no atomics: the exhaustive form
          //  splits the row list over waves of different workgroups)
          for (int j0 = 0; j0 < nr; j0 += a.capmax) {
            const int j1 = min(nr, j0 + a.capmax);
            int nbr = 0;
            for (int jb = j0; jb < j1; jb += kNmsThreads) {
              const int j = jb + tid;
              uint32_t rp = 0u;
              bool flag = false;
              if (j < j1) {
                rp = rows[j];
                const float4 q0 = a.rec[(size_t)rp * G::RECQ];
                flag = grid_is_brute(gp, q0.x, q0.y, q0.z, q0.w);
              }
              const u64 fm = __ballot(flag);
              __syncthreads();                                   // previous users of s_i / readers of the list are done
              if ((tid & 63) == 0) s_i[wv] = __popcll(fm);
              __syncthreads();
              int pre = 0, tot = 0;
#pragma unroll
              for (int k = 0; k < kNmsWaves; k++) { const int t = s_i[k]; if (k < wv) pre += t; tot += t; }
              if (flag) cidx[nbr + pre + __popcll(fm & lanemask_lt())] = rp;
              nbr += tot;
            }
            __syncthreads();
            if (nbr > 0) nms_cross<G, GRID>(a, cidx, nbr, c0, c1, nullptr, 0, tw, ntw, L);
            __syncthreads();                                     // the list is read until here
          }
          nms_cross<G, GRID>(a, rows, nr, c0, c1, a.ulist, n_brute, tw, ntw, L);
        }
        return;
      }
    }
    nms_cross<G, GRID>(a, rows, nr, c0, c1, nullptr, 0, tw, ntw, L);
  };

  const int plan_chunk = a.cap_first < a.capmax ? a.cap_first : a.capmax;
  for (int g = g_first; g <= g_last; g += g_step) {
    const int sb = slab_mode ? sp->segb[g] : a.seg_begin[g], se = slab_mode ? sp->sege[g] : a.seg_end[g];
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
    if (a.resume != nullptr) {                            // carry on where the phase kernels stopped (every workgroup reads the same words)
      const NmsResume& r = *a.resume;
      kept = r.kept;
      cur = (r.stage == 1) ? r.chunk_first : r.cur;
      if (cur < sb) cur = sb;
      if (r.cap > 0) cap = r.cap < a.capmax ? r.cap : a.capmax;
      if (r.stage == 3 && r.nrow > 0 && r.nrow <= kept && cur < wend) { jrows = a.rows + sb + (kept - r.nrow); jnr = r.nrow; jc0 = cur; jc1 = wend; }
    }
    for (;;) {
      if (jnr > 0) {
        const u64 tcz = (a.prof && tid == 0) ? wall_clock64() : 0ull;
        cross(jrows, jnr, jc0, jc1);
        if (aborted) return;
        if (a.prof && tid == 0) { const u64 dcz = wall_clock64() - tcz; atomicMax(a.prof + 31, dcz); atomicAdd(a.prof + 51, dcz); }
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
      nms_pairs<G, GRID>(a, team, cn, cidx, tw, ntw, L, &s_i[12]);
      if (a.prof && tid == 0) { const u64 d = wall_clock64() - tpz; atomicMax(a.prof + 29, d); atomicAdd(a.prof + 30, d); }
      lap(2);
      // ---- all edges are out: the last arriver resolves the chunk, the others wait for its rows
      // (Round 4 measured the alternative -- a fixed resolver publishes the rows no conflict edge points at after its first
      //  pass, the other workgroups cross those while the rounds run, the few rows kept in later rounds get a second pass:
      //  K=300 230 against 221 us, 18 classes 283 / 280, K=3000 550 / 543, uniform 1600 / 1597 -- the second pass costs what
      //  the overlap saves, a chunk's resolve is only long when there is no cross phase behind it.  Not kept.)
      if (serial_begin(bar, &s_flag)) {
        lap(3);
        const u64 ts = (a.prof && tid == 0) ? wall_clock64() : 0ull;
        nms_resolve(a, g, sb, team, cn, kept, cidx, smem, sizeof(WaveLds<G>) * kNmsWaves, s_i);
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
      // (measured: jumping to the largest chunk when most of a chunk is kept -- sparse data -- is slower, 0.78 -> 0.96 ms at
      //  100k with 18 class offsets: the pair phase grows with the square of the chunk)
      // (a.grow_sparse > 2: a chunk that kept more than half of its boxes -- sparse data -- is followed by one that many times
      //  larger instead of twice; OBB_NMS_GROW, measurements)
      if (cap < a.capmax) { cap *= (a.grow_sparse > 2 && 2 * nr > cn) ? a.grow_sparse : 2; if (cap > a.capmax) cap = a.capmax; }
    }
  }
  if constexpr (G::HAS_GRID && GRID) {
    if (slab_mode) {                                   // every slab is done: the kept boxes meet again in score order
      lap(7);
      if (!team_barrier(gbar, &s_flag)) return;
      if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // (one per workgroup: see grid_build)
      __syncthreads();
      slab_merge(a.kept_bits, n0, order0, keep_out0, keep_cnt0, s_i);
      lap(8);
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

struct PlanLds { long long a[16], b[16]; int first[1024], last[1024], size[1024]; };

// the planner proper: called by all 1024 threads of one workgroup (k_plan_teams, or the planner block of the fused
// sort/prep kernel); seg_begin / seg_end -- or the segment sizes, when seg_size is given -- must be visible to the caller
__device__ __forceinline__ void plan_teams_block(const int* seg_begin, const int* seg_end, const int* seg_size, int nseg, int NB, int chunk,
                                                 int4* plan, PlanLds& S) {
  long long* s_a = S.a; long long* s_b = S.b;
  int* s_first = S.first; int* s_last = S.last;
  const int tid = threadIdx.x;
  const int per = (nseg + 1023) / 1024;
  const int g0 = tid * per, g1 = (g0 + per < nseg) ? g0 + per : nseg;
  // (the sizes are read from global memory ONCE: the five passes below each cost a dependent round trip otherwise)
  const bool staged = nseg <= 1024;
  if (staged) {
    // (seg_size comes from the other workgroups of the SAME launch -- the planner block of the fused sort kernel -- written with
    //  agent-scope stores behind a relaxed ticket: read it with agent-scope loads, so that no stale line of this CU's cache can
    //  stand in for a size and leave a segment without a workgroup)
    if (tid < nseg) S.size[tid] = seg_size ? ldg_agent(seg_size + tid) : seg_end[tid] - seg_begin[tid];
    __syncthreads();
  }
  auto cost_of = [&](int g) -> long long {
    return plan_cost((long long)(staged ? S.size[g] : (seg_size ? ldg_agent(seg_size + g) : seg_end[g] - seg_begin[g])), chunk);
  };
  // ---- totals
  long long myc = 0, myn = 0, pa, pb, total, nne;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); myc += c; myn += c > 0 ? 1 : 0; }
  block_scan2(myc, myn, s_a, s_b, pa, pb, total, nne);
  if (nne == 0) {
    for (int w = tid; w < NB; w += 1024) plan[w] = make_int4(-1, 0, 0, 0);
    return;
  }
  const long long L = (total + NB - 1) / NB;
  long long big_from = 2 * L;
  // ---- split of the grid: r workgroups for the small segments, NB - r for the teams of the big ones
  long long mybc = 0, mybn = 0, big_cost, n_big;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); if (c >= big_from) { mybc += c; mybn++; } }
  long long pbc, pbn;
  block_scan2(mybc, mybn, s_a, s_b, pbc, pbn, big_cost, n_big);
  long long small_cost = total - big_cost, n_small = nne - n_big;
  long long r = 0;
  if (n_small > 0) {
    r = (small_cost + L - 1) / L;
    if (r < 1) r = 1;
    if (r > n_small) r = n_small;
    if (r > NB - n_big) r = NB - n_big;             // (n_big <= NB / 2: every big segment costs at least 2L)
  }
  // enough workgroups for everybody: no packing, every small segment gets its own workgroup (piece = its rank)
  bool one_each = n_small > 0 && n_small + (n_big > 0 ? (big_cost + L - 1) / L : 0) <= NB;
  // Spare workgroups.  With one workgroup per segment and none big, the launch lasts as long as its heaviest segment
  // (the configs[1] step: 240 class segments, busiest workgroup 80 us, mean 49) while NB - nne workgroups idle.  The
  // NB - nne heaviest segments whose cost is above the average (a team pays ~10 us of barriers) get a second workgroup
  // each.  Costs are coarse (a function of the tile count), so "heaviest" is decided on the unique key cost * 1024 +
  // (1023 - segment): `big` is then a per-segment property, not a cost threshold -- which is fine here, because with one
  // segment per run no workgroup ever has to recognise somebody else's big segment (plan[kPlanInfoSlot] = "none").
  long long forced_team = 0;
  bool spare_mode = false;
  long long big_key = 0;
  auto ekey = [&](int g, long long c) -> long long { return c * 1024 + (1023 - g); };
  if (one_each && n_big == 0 && per == 1 && nne < NB && nseg <= 1024) {
    const long long spare = NB - nne;
    const long long myc1 = g0 < g1 ? cost_of(g0) : 0;
    unsigned long long* s_min = reinterpret_cast<unsigned long long*>(s_last);
    __syncthreads();
    s_first[tid] = (int)(myc1 > 0x7fffffffLL ? 0x7fffffffLL : myc1);
    if (tid == 0) *s_min = ~0ull;
    __syncthreads();
    if (g0 < g1 && myc1 >= L + L / 16 && myc1 >= 128) {
      const long long mine_k = ekey(g0, myc1);
      int at_or_above = 0;
      for (int h = 0; h < nseg; h++) at_or_above += ekey(h, (long long)s_first[h]) >= mine_k ? 1 : 0;
      if (at_or_above <= spare) atomicMin(s_min, (unsigned long long)mine_k);
    }
    __syncthreads();
    const unsigned long long kmin = *s_min;
    __syncthreads();
    if (kmin != ~0ull) {
      spare_mode = true;
      big_key = (long long)kmin;
      forced_team = 2;
      const bool mb = g0 < g1 && ekey(g0, myc1) >= big_key;
      mybc = mb ? myc1 : 0; mybn = mb ? 1 : 0;
      block_scan2(mybc, mybn, s_a, s_b, pbc, pbn, big_cost, n_big);
      small_cost = total - big_cost; n_small = nne - n_big;
      one_each = n_small > 0;                        // n_small + 2 n_big = nne + n_big <= NB
      r = n_small;
      big_from = 0x7fffffffLL;                       // what the workgroups are told: nobody skips by cost
    }
  }
  auto is_big = [&](int g, long long c) -> bool { return spare_mode ? (c > 0 && ekey(g, c) >= big_key) : (c >= big_from); };
  if (one_each) r = n_small;
  const long long NBb = NB - r;
  const long long Lb = n_big > 0 ? (big_cost + NBb - 1) / NBb : 1;
  auto team_size = [&](long long c) -> long long {
    long long t = forced_team ? forced_team : c / Lb;
    const long long useful = (c + 63) / 64;          // more workgroups than ~tiles/8 only add barrier cost
    if (t > useful) t = useful;
    return t < 1 ? 1 : t;
  };
  // ---- teams of the big segments: workgroups r + [prefix of team sizes), team ids r + [prefix of count)
  long long myt = 0;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); if (is_big(g, c)) myt += team_size(c); }
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
      if (is_big(g, c)) {
        const long long t = team_size(c);
        for (int i = 0; i < (int)t; i++) plan[w0 + i] = make_int4(g, (int)team, i, (int)t);
        w0 += t; team++;
      }
    }
  }
  for (int w = (int)(r + used_big) + tid; w < NB; w += 1024) plan[w] = make_int4(-1, 0, 0, 0);
  // ---- runs of small segments: piece = running small cost / Ls
  long long mysc = 0, mysn = 0, psc, psn, tsc, tsn;
  for (int g = g0; g < g1; g++) { const long long c = cost_of(g); if (c > 0 && !is_big(g, c)) { mysc += c; mysn++; } }
  block_scan2(mysc, mysn, s_a, s_b, psc, psn, tsc, tsn);
  const long long Ls = r > 0 ? (small_cost + r - 1) / r : 1;
  s_first[tid] = 0x7fffffff; s_last[tid] = -1;
  __syncthreads();
  if (r > 0) {
    long long run = psc, rank = psn;
    for (int g = g0; g < g1; g++) {
      const long long c = cost_of(g);
      if (c > 0 && !is_big(g, c)) {
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
  plan_teams_block(seg_begin, seg_end, nullptr, nseg, NB, chunk, plan, S);
}

}  // namespace obb
