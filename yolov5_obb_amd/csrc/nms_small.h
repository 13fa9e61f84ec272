// Greedy NMS of MANY SMALL segments, one workgroup per segment, nothing shared between workgroups -- gfx950 (included by
// nms.hip behind nms_core.h, namespace obb).
//
// The regime: the fused driver at the reference's default thresholds (utils/general.py:772-862 with conf 0.25): 16 images x
// 16 classes = 256 segments of ~100 boxes (a few above 200).  The persistent kernel of nms_core.h is built for lists of 10^5
// boxes -- chunks, a replicated state machine on a shared bitmap, edge lists in global memory, rounds of a maximal independent
// set, team barriers, a planner in front: for a 100-box segment its workgroup spends 7 us selecting, 12 us resolving and every
// record it touches is a global round trip (busiest workgroup 69 us, mean 47).  Here a segment lives in LDS from its first
// load to its last store:
//   1. records (64 bytes per box) and the segment's slice of the alive bitmap -> LDS -- as the sort kernel left them (FRONT =
//      SmallFromSort), or built here from the image's candidate keys with no sort launch in front of the kernel (FRONT =
//      SmallSelfSort of nmsobb_impl.h, round 6: the fused driver's default);
//   2. pairs: 64 x 16 slices of the upper triangle, drawn by the eight waves from an LDS counter; the circle test of
//      RotGeom::cheap_reject on quad 0, then the same three decision stages as everywhere else (classify_quick, classify_full,
//      hit_exact -- the decisions ARE the reference's), each with its own LDS queue so that a stage always runs on a full wave;
//      "IoU > thr" sets bit j of row i of a bit matrix in LDS;
//   3. the reference's scan itself, by one wave: the lowest alive position is kept and clears its row's bits -- one
//      iteration per KEPT box;
//   4. the kept positions and their count -- or, with an output stage behind the segments (the kernel's TAIL: SmallGather of
//      nmsobb_impl.h, round 5), the merge key and candidate slot of every kept box, written through for the workgroup that
//      finishes the image.
// No barrier between workgroups, no co-residency assumption, no planner (the TAIL counts arrivals on a ticket; nobody waits).  A segment with more than kSmallMax boxes raises a
// flag and is left alone: the host layer repeats the call on the persistent kernel (it chooses this kernel from the
// previous call's largest segment, include/obb_hip.h: expected_cand).
#pragma once

namespace obb {

#ifdef OBB_SMALL_TRACE
// (development builds, tools/small_trace.sh) 16 words per workgroup: block, seg | part << 24 | np << 28, n, kept, stamps..., drains
__device__ unsigned long long g_small_trace[2048 * 16];
__device__ unsigned long long g_small_trace_tail[2048 * 4];
__device__ unsigned long long g_small_trace2[2048 * 8];
#define STRACE(kind, ...) do { if (threadIdx.x == 0 && blockIdx.x < 2048) { const unsigned long long v_[] = {__VA_ARGS__}; unsigned long long* o_ = g_small_trace + blockIdx.x * 16; o_[0] = (kind); for (int z_ = 0; z_ < (int)(sizeof(v_) / 8) && z_ < 15; z_++) o_[1 + z_] = v_[z_]; } } while (0)
#endif
constexpr int kSmallMax = OBB_NMS_SMALL_SEG;        // boxes per segment (include/obb_hip.h)
constexpr int kSmallWords = kSmallMax / 64;         // bit-matrix words per row
constexpr int kSmallThreads = 512;
constexpr int kSmallWaves = kSmallThreads / 64;

struct SmallArgs {
  const float4* rec;         // [n][RECQ] records in sorted order
  const u64* alive;          // bit p: position p takes part (the small-box filter clears bits)
  const int* seg_begin;      // [nseg]
  const int* seg_end;        // [nseg]
  int* keep_cnt;             // [nseg] (output)
  int64_t* keep_out;         // segment g writes sorted positions at keep_out[seg_begin[g] + k]
  int* too_big;              // set to the size of a segment this kernel does not take
  float thr;
  int max_keep;              // 0 = unlimited
  // With an output stage behind the segments (the kernel's TAIL, nmsobb_impl.h: SmallGather) a segment publishes what that stage
  // merges -- the merge key and the candidate slot of every kept box, at [seg_begin[g] + k] -- instead of keep_out's positions;
  // pub_key == nullptr: keep_out.
  const unsigned long long* keys_sorted;   // [n] sort keys in sorted order
  const uint32_t* vals_sorted;             // [n] candidate slots in sorted order
  const int* mode;                         // [images] 1: the class-segment key (rotated into the merge key), else as it is
  int ncs;                                 // segments per image
  unsigned long long* pub_key;             // [n_pos]
  uint32_t* pub_val;
  long long n_pos;
  // A LARGE segment is shared by several workgroups (round 6, third part): the sort kernel, which knows every segment's size, gives a
  // segment of more than kSmallSplit1 boxes np - 1 HELPER workgroups (small_parts) from a pool of `helpers` (the first blocks of
  // the grid: they are dispatched before the segments' own workgroups) -- work[h] = segment | part << 24, seg_np[segment] =
  // np | first helper << 8.  Part p takes every np-th slice of the pair triangle into its own LDS bit matrix, writes the matrix
  // through to part_main (part 0: at the segment's positions) or part_help (helper h), and arrives on seg_ticket[segment]; the part
  // that arrives LAST ors the others' matrices into its own and goes on with the scan, the output and the TAIL.  Nobody waits.
  // helpers == 0: every segment is one workgroup's.
  int helpers;
  const int* help_cnt;                     // [1] helper slots handed out (may exceed `helpers`: the slots beyond were not)
  const int* work;                         // [helpers]  (-1: a slot whose segment got no full set of helpers and stays whole)
  const int* seg_np;                       // [nseg]
  int* seg_ticket;                         // [nseg] zeroed by the sort kernel
  u64* part_main;                          // [n_pos][kSmallWords]
  u64* part_help;                          // [helpers][kSmallMax][kSmallWords]
  // SELF-SORTING segments (round 6, fourth part; the FRONT of nmsobb_impl.h: SmallSelfSort): no sort launch in front of this kernel --
  // the workgroup of (image, class) reads the image's candidate keys as the filter kernel left them, keeps its class, orders it by
  // rank counting and builds records, alive words and publishing keys in LDS.  rec / alive / seg_begin / seg_end / keys_sorted /
  // vals_sorted / mode are not read; segment (g, c) publishes at g * cap_img + c * kSmallMax.  keys_in == nullptr: segments from the sort.
  const unsigned long long* keys_in;       // [images * cap_img] (score_desc << 32 | row * nc + class), slot order
  const float4* cand;                      // [images * cap_img][2]
  const int* cnt;                          // [images * kCntPad] candidates per image
  const int* tiny;                         // [images] flags (kImg*)
  long long cap_img, max_nms, A;
  int nc;
  float class_offset;
  int* seg_max;                            // [1] the call's largest segment (feedback, status[1])
};
struct SmallFromSort { static constexpr bool kSelf = false; };   // FRONT of k_nms_small: segments as the sort kernel left them
// parts of a segment of n boxes: the work is the pairs that survive the circle test (~n^2), a part should hold what a 128-box
// segment holds
constexpr int kSmallClipWaves = 4;                 // the waves of a workgroup sit on four SIMDs: up to this many clip drains run side by side at full speed
constexpr int kSmallSplit1 = 128;
constexpr int kSmallHelpMax = 256;
__host__ __device__ __forceinline__ int small_parts(int n) {
  return n <= kSmallSplit1 ? 1 : (n <= 192 ? 2 : (n <= 256 ? 4 : (n <= 320 ? 6 : 8)));
}
// no output stage: the caller's next launch reads keep_out / keep_cnt
struct SmallNoTail {
  struct Args {};
  static __device__ __forceinline__ void run(const SmallArgs&, const Args&, unsigned char*, int) {}
};

// The three decision stages are real functions here (one body each, called from the full drains and from the pooled
// leftovers): a workgroup runs most of this kernel's code exactly once, so its time is instruction FETCH as much as
// execution -- with the stages inlined at both call sites the kernel was 17.5k instructions, and a wave that found no work
// took 2 us to get through it.
template <class G>
__device__ __attribute__((noinline)) int small_quick(const float4* ra, const float4* rb, float thr) {
  return G::classify_quick(ra, rb, thr, true);
}

// The circle test of one 16-row x 64-column slice (round 6, third part).  Lane = column; the 16 row quads are fetched by lanes
// 0..15 in ONE LDS read and handed round with v_readlane (scalar operands of the test), every row costs its ~11 VALU instructions
// and nothing else -- no branch, no exec-mask change, no push -- and leaves ONE BIT in the lane's word; the caller turns the set
// bits into queue entries, one push per bit PLANE instead of one per row.  (The loop this replaces tested, balloted and pushed row
// by row with short-circuit conditions: the compiler made ~100 issue slots per row of it, half of them scalar spills through
// v_readlane / v_writelane -- 4.2 us per slice, 25 of the 31 us a 249-box segment spent in its slices.)  The arithmetic is
// RotGeom::cheap_reject's, operation for operation; a function of its own so that it gets a register allocation of its own.
template <class G>
__device__ __attribute__((noinline)) uint32_t small_slice_bits(const float4* s_rec, int i0, int n, uint32_t arow16, float4 cq, int j, bool jv) {
  const int lane = threadIdx.x & 63;
  const int ir = i0 + (lane & 15);
  const float4 rq = s_rec[(ir < n ? ir : 0) * G::RECQ];
  uint32_t bits = 0u;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const float4 a = rdlane4(rq, r);                            // (wave-uniform: the row box)
    const float dx = cq.x - a.x, dy = cq.y - a.y;
    const float rs = a.z + cq.z;
    const float d2 = dx * dx + dy * dy;
    const bool apart = (d2 > rs * rs) & (fminf(a.w, cq.w) >= 2.34e-9f * d2);
    const bool row_ok = ((arow16 >> r) & 1u) != 0u && i0 + r < n;   // (wave-uniform)
    const bool pass = !apart & jv & (j > i0 + r) & row_ok;
    bits |= pass ? (1u << r) : 0u;
  }
  return bits;
}

template <class G>
struct SmallWave {                                  // per wave
  float scr[G::SCR * 64];                           // exact-clip scratch, one column per lane
  uint32_t q0[128], q1[128], q2[128];               // pending (i << 16 | j) pairs of the three stages
};

// returns the segment when this workgroup completed it (the TAIL follows), -1 when it has nothing (more) to do
template <class G, class FRONT>
__device__ __forceinline__ int small_segment(const SmallArgs& a, unsigned char* s_raw) {
  __shared__ int s_next, s_nkept, s_qcnt[kSmallWaves], s_qhead[kSmallWaves], s_qcnt2[kSmallWaves], s_last;
  __shared__ uint32_t s_kept[kSmallMax];
  __shared__ unsigned long long s_pk[kSmallMax];                 // (publishing) sort key and candidate slot of every position
  __shared__ uint32_t s_pv[kSmallMax];
  float4* s_rec = reinterpret_cast<float4*>(s_raw);                                  // [kSmallMax][RECQ]
  u64* s_mask = reinterpret_cast<u64*>(s_rec + (size_t)kSmallMax * G::RECQ);         // [kSmallMax][kSmallWords]
  u64* s_alive = s_mask + (size_t)kSmallMax * kSmallWords;                           // [kSmallWords] + pad
  SmallWave<G>* s_wave = reinterpret_cast<SmallWave<G>*>(s_alive + 8);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int seg = (int)blockIdx.x - a.helpers, part = 0, np = 1, hbase = 0;
  if (seg < 0) {                                                 // a helper (workgroup-uniform): which part of which segment, if any
    const int h = (int)blockIdx.x;
    if (h >= *a.help_cnt) {
#ifdef OBB_SMALL_TRACE
      STRACE(1ull, (unsigned long long)wall_clock64());
#endif
      return -1;
    }
    const int e = a.work[h];
    if (e < 0) return -1;
    seg = e & 0xffffff; part = e >> 24;
  }
  if (a.helpers > 0) { const int v = a.seg_np[seg]; if ((v & 255) > 1) { np = v & 255; hbase = v >> 8; } }
#ifdef OBB_SMALL_TRACE
  unsigned long long tt[8]; int ti_ = 0, nd0 = 0, nd1 = 0, nd2 = 0, nit = 0;
  unsigned long long acc_d0 = 0, acc_d1 = 0, acc_d2 = 0, acc_draw = 0, acc_l0 = 0, acc_l1 = 0, acc_l2 = 0;
#define SSTAMP() do { tt[ti_++] = wall_clock64(); } while (0)
#else
#define SSTAMP() do {} while (0)
#endif
  SSTAMP();                                                      // (self-sorting segments: the front end counts as "load")
  int sb, n, md_self = 0;
  if constexpr (FRONT::kSelf) {
    // (no helpers in this mode: part == 0)  records, alive words, publishing keys and the zeroed bit matrix are in LDS when load() returns
    const int r = FRONT::template load<G>(a, seg, reinterpret_cast<float4*>(s_raw),
                                          reinterpret_cast<u64*>(reinterpret_cast<float4*>(s_raw) + (size_t)kSmallMax * G::RECQ), s_pk, s_pv, &s_next);
    n = r & 0xffff; md_self = r >> 16;
    sb = (int)((long long)(seg / a.ncs) * a.cap_img) + (seg % a.ncs) * kSmallMax;
    if (n <= 0 || n > kSmallMax) {                               // (workgroup-uniform) nobody zeroed keep_cnt in this mode
      if (tid == 0) { stg_agent(a.keep_cnt + seg, 0); if (n > kSmallMax) atomicMax(a.too_big, n); }
      return seg;
    }
  } else {
    sb = a.seg_begin[seg]; n = a.seg_end[seg] - sb;
    if (n <= 0) return part == 0 ? seg : -1;                     // (keep_cnt is zero already)
    if (n > kSmallMax) { if (tid == 0 && part == 0) atomicMax(a.too_big, n); return part == 0 ? seg : -1; }
  }
  // ---- 1. the segment -> LDS
  const bool pub = a.pub_key != nullptr;                         // (kernel-uniform)
  if constexpr (!FRONT::kSelf) {
    for (int t = tid; t < n * G::RECQ; t += kSmallThreads) s_rec[t] = a.rec[(size_t)sb * G::RECQ + t];
    for (int t = tid; t < n * kSmallWords; t += kSmallThreads) s_mask[t] = 0ull;
    if (pub)                                                     // requested with the records: no latency of its own
      for (int t = tid; t < n; t += kSmallThreads) { s_pk[t] = a.keys_sorted[(size_t)sb + t]; s_pv[t] = a.vals_sorted[(size_t)sb + t]; }
    if (tid < kSmallWords) {
      const int w0 = (sb >> 6) + tid, sh = sb & 63;
      u64 v = a.alive[w0] >> sh;
      if (sh) v |= a.alive[w0 + 1] << (64 - sh);
      const int left = n - tid * 64;                             // positions of this word that belong to the segment
      if (left <= 0) v = 0ull; else if (left < 64) v &= (1ull << left) - 1ull;
      s_alive[tid] = v;
    }
  }
  if (tid == 0) s_next = 0;
  __syncthreads();
  SSTAMP();
  // ---- 2. pairs
  const float thr = a.thr;
  SmallWave<G>& W = s_wave[wv];
  PairQueue Q0{W.q0, 0, 0}, Q1{W.q1, 0, 0}, Q2{W.q2, 0, 0};
  auto hit = [&](bool h, int i, int j) { if (h) atomicOr(&s_mask[i * kSmallWords + (j >> 6)], 1ull << (j & 63)); };
  auto drain2 = [&](int cnt) {                                   // the exact clip
#ifdef OBB_SMALL_TRACE
    nd2++;
#endif
    wave_sync();
    bool h = false;
    int i = 0, j = 0;
    if (lane < cnt) {
      const uint32_t e = W.q2[(Q2.head + lane) & 127];
      i = (int)(e >> 16); j = (int)(e & 0xffffu);
      h = nms_stage_exact<G>(s_rec + i * G::RECQ, s_rec + j * G::RECQ, thr, W.scr + lane);
    }
    hit(h, i, j);
    Q2.head = (Q2.head + cnt) & 127; Q2.count -= cnt;
    wave_sync();
  };
  auto drain1 = [&](int cnt) {                                   // the IoU interval
#ifdef OBB_SMALL_TRACE
    nd1++;
#endif
    wave_sync();
    int res = 0, i = 0, j = 0;
    uint32_t e = 0;
    if (lane < cnt) {
      e = W.q1[(Q1.head + lane) & 127];
      i = (int)(e >> 16); j = (int)(e & 0xffffu);
      res = nms_stage_full<G>(s_rec + i * G::RECQ, s_rec + j * G::RECQ, thr);
    }
    hit(res == 1, i, j);
    Q1.head = (Q1.head + cnt) & 127; Q1.count -= cnt;
    Q2.push(res == 2, e);
    wave_sync();
    if (Q2.count >= 64) drain2(64);
  };
  auto drain0 = [&](int cnt) {                                   // the register-only bounds
#ifdef OBB_SMALL_TRACE
    nd0++;
#endif
    wave_sync();
    int res = 0, i = 0, j = 0;
    uint32_t e = 0;
    if (lane < cnt) {
      e = W.q0[(Q0.head + lane) & 127];
      i = (int)(e >> 16); j = (int)(e & 0xffffu);
      res = small_quick<G>(s_rec + i * G::RECQ, s_rec + j * G::RECQ, thr);
    }
    hit(res == 1, i, j);
    Q0.head = (Q0.head + cnt) & 127; Q0.count -= cnt;
    Q1.push(res == 3, e);
    Q2.push(res == 2, e);
    wave_sync();
    if (Q2.count >= 64) drain2(64);                              // (first: drain1 may add up to 64 more)
    if (Q1.count >= 64) drain1(64);
  };
  {
    const int nb = (n + 63) >> 6;
    const int nitems = nb * (nb + 1) / 2 * 4;                    // upper-triangle tiles x four 16-row slices
    for (;;) {
      int it = 0;
#ifdef OBB_SMALL_TRACE
      const unsigned long long tdr_ = wall_clock64();
#endif
      if (lane == 0) it = atomicAdd(&s_next, 1);
      it = __builtin_amdgcn_readfirstlane(it) * np + part;       // (a shared segment: every np-th slice is this part's)
#ifdef OBB_SMALL_TRACE
      acc_draw += wall_clock64() - tdr_;
#endif
      if (it >= nitems) break;
#ifdef OBB_SMALL_TRACE
      nit++;
#endif
      const int tile = it >> 2, quarter = it & 3;
      int ti = 0, rem = tile;                                    // tile -> (ti, tj), ti <= tj: row ti holds nb - ti tiles
      while (rem >= nb - ti) { rem -= nb - ti; ti++; }
      const int tj = ti + rem;
      const int j = tj * 64 + lane;
      const bool jv = j < n && ((s_alive[tj] >> lane) & 1ull);
      const float4 cq = s_rec[(j < n ? j : 0) * G::RECQ];
      const int i0 = ti * 64 + quarter * 16;
      const uint32_t arow16 = (uint32_t)(s_alive[ti] >> (quarter * 16)) & 0xffffu;
      if (i0 >= n || arow16 == 0u) continue;                     // (wave-uniform)
      // the slice's circle tests leave a bit per (row, column) in the column's lane; a push per bit plane (small_slice_bits).
      // What the per-segment stamps say about the rest (bs16 step, segments of 200-250 boxes): 17-20 us in the pooled leftovers
      // (quick 3-7, interval 5.8, one clip drain 8 us), 3-4 us in the scan; a wave of such a workgroup is bound by its own
      // instruction stream, not by memory.  Measured and not kept: running the interval stage and the clip lazily (rings
      // drained only above 64 before a push, everything else left to the pooled passes): 66.6 -> 67.1 us.
      uint32_t bits = small_slice_bits<G>(s_rec, i0, n, arow16, cq, j, jv);
      while (__ballot(bits != 0u)) {                             // (wave-uniform)
        const bool has = bits != 0u;
        const int r = __builtin_ctz(bits | 0x10000u);
        Q0.push(has, ((uint32_t)(i0 + r) << 16) | (uint32_t)j);
        bits &= bits - 1u;
#ifdef OBB_SMALL_TRACE
        if (Q0.count >= 64) { const unsigned long long td_ = wall_clock64(); drain0(64); acc_d0 += wall_clock64() - td_; }
#else
        if (Q0.count >= 64) drain0(64);
#endif
      }
    }
    SSTAMP();
    // Leftovers: every wave holds a few pairs per stage.  A drain costs its instructions whatever it holds (150 / 1500 / ~4000
    // per stage), and eight nearly empty ones share four SIMDs -- so the leftovers of a stage are POOLED: the eight rings are
    // read as one list, 64 entries per wave, by as few waves as it takes (what a wave decides goes to its own next ring).
    for (int stage = 0; stage < 3; stage++) {
      PairQueue& Q = stage == 0 ? Q0 : (stage == 1 ? Q1 : Q2);
#ifdef OBB_SMALL_TRACE
      const unsigned long long tl_ = wall_clock64();
      struct LeftT { unsigned long long& a; unsigned long long t; __device__ ~LeftT() { a += wall_clock64() - t; } } lt_{stage == 0 ? acc_l0 : (stage == 1 ? acc_l1 : acc_l2), tl_};
#endif
      for (;;) {                                                 // (one pass unless the rings hold more than 8 x 64 pairs)
        // room for what this pass may add to the wave's next rings (64 each; a ring holds 128).  The first pass of a stage finds it
        // (the slices leave every ring below 64); a full drain in front of a LATER pass is the rare case.  (Until round 6 the drains
        // followed the pushes: one wave's interval or clip drain of 64 stood between the pooled quick pass and everybody else's next
        // stage -- 3-5 us, 16 in a 73-box segment.)
        if (stage <= 1 && Q2.count > 64) drain2(64);
        if (stage == 0 && Q1.count > 64) drain1(64);
        if (lane == 0) { s_qcnt[wv] = Q.count; s_qhead[wv] = Q.head; if (stage == 1) s_qcnt2[wv] = Q2.count; }
        __syncthreads();
        if (stage == 1) {
          // What is left for the interval stage and the exact clip together fits FOUR waves -- one per SIMD -- and every wave's two
          // rings fit one: the interval stage (1700 instructions, 5.8 us, to spare some pairs the clip) would only stand in front of
          // a clip drain (8 us) that runs anyway -- its pairs join the clip's rings and the stage is skipped.  (Round 6: the limit
          // was one wave's worth; the bs16 step's segments bring 43 pairs to this point in the median, 230 at most, none of them
          // bound for the clip yet: every segment paid both stages in a row.)
          int t12 = 0, fit = 1;
#pragma unroll
          for (int w = 0; w < kSmallWaves; w++) { t12 += s_qcnt[w] + s_qcnt2[w]; fit &= (s_qcnt[w] + s_qcnt2[w] <= 128) ? 1 : 0; }
#ifdef OBB_SMALL_TRACE
          { int t1_ = 0; for (int w = 0; w < kSmallWaves; w++) t1_ += s_qcnt[w]; if (tid == 0 && blockIdx.x < 2048 && g_small_trace2[blockIdx.x * 8 + 7] == 0ull) g_small_trace2[blockIdx.x * 8 + 7] = 1ull + (unsigned long long)t1_ + ((unsigned long long)(t12 - t1_) << 20); }
#endif
          if (t12 <= 64 * kSmallClipWaves && fit) {              // (workgroup-uniform)
            while (Q1.count > 0) {
              wave_sync();
              const int c1 = Q1.count < 64 ? Q1.count : 64;
              const bool mv = lane < c1;
              const uint32_t e1 = mv ? W.q1[(Q1.head + lane) & 127] : 0u;
              Q1.head = (Q1.head + c1) & 127; Q1.count -= c1;
              Q2.push(mv, e1);
              wave_sync();
            }
            __syncthreads();
            break;
          }
        }
        int pre[kSmallWaves + 1];
        pre[0] = 0;
#pragma unroll
        for (int w = 0; w < kSmallWaves; w++) pre[w + 1] = pre[w] + s_qcnt[w];
        const int total = pre[kSmallWaves];
        if (total == 0) { __syncthreads(); break; }              // (workgroup-uniform; the barrier: s_qcnt is rewritten by the next stage)
        const int g = wv * 64 + lane;                            // wave wv takes pooled entries [64 wv, 64 wv + 64)
        int res = 0, i = 0, j = 0;
        uint32_t e = 0;
        const bool busy = wv * 64 < total;                       // (wave-uniform)
        if (busy && g < total) {
          int w = 0;
#pragma unroll
          for (int u = 1; u < kSmallWaves; u++) w += (g >= pre[u]) ? 1 : 0;
          const uint32_t* ring = stage == 0 ? s_wave[w].q0 : (stage == 1 ? s_wave[w].q1 : s_wave[w].q2);
          e = ring[(s_qhead[w] + (g - pre[w])) & 127];
          i = (int)(e >> 16); j = (int)(e & 0xffffu);
          const float4* ra = s_rec + i * G::RECQ;
          const float4* rb = s_rec + j * G::RECQ;
          if (stage == 0) res = small_quick<G>(ra, rb, thr);
          else if (stage == 1) res = nms_stage_full<G>(ra, rb, thr);
          else res = nms_stage_exact<G>(ra, rb, thr, W.scr + lane) ? 1 : 0;
        }
        hit(res == 1, i, j);
#ifdef OBB_SMALL_TRACE
        if (busy) { if (stage == 0) nd0++; else if (stage == 1) nd1++; else nd2++; }
#endif
        __syncthreads();                                         // every reader is done with the rings of this stage ...
        {                                                        // this wave's ring lost what the pass took of it
          const int room = kSmallWaves * 64 - pre[wv];
          const int took = room <= 0 ? 0 : (room < s_qcnt[wv] ? room : s_qcnt[wv]);
          Q.head = (Q.head + took) & 127; Q.count -= took;
        }
        if (busy) {                                              // ... before anybody's next ring grows
          if (stage == 0) { Q1.push(res == 3, e); Q2.push(res == 2, e); }
          else if (stage == 1) Q2.push(res == 2, e);
          wave_sync();
        }
        __syncthreads();                                         // (s_qcnt is rewritten by the next pass)
      }
    }
  }
  SSTAMP();
  __syncthreads();
  if (np > 1) {                                                  // (workgroup-uniform) a shared segment: publish, arrive, the last part merges
    auto part_buf = [&](int p) -> u64* {
      return p == 0 ? a.part_main + (size_t)sb * kSmallWords : a.part_help + (size_t)(hbase + p - 1) * kSmallMax * kSmallWords;
    };
    u64* mine = part_buf(part);
    for (int t = tid; t < n * kSmallWords; t += kSmallThreads) stg_agent(mine + t, s_mask[t]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // written through before the arrival
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(a.seg_ticket + seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == np - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) {
#ifdef OBB_SMALL_TRACE
      STRACE(2ull, (unsigned long long)seg | ((unsigned long long)part << 24) | ((unsigned long long)np << 28), (unsigned long long)n, 0ull, tt[0], tt[1], tt[2], tt[3], (unsigned long long)wall_clock64(), 0ull, 0ull, (unsigned long long)nit, (unsigned long long)nd0, (unsigned long long)nd1, (unsigned long long)nd2);
#endif
      return -1;
    }
    for (int p = 0; p < np; p++) {
      if (p == part) continue;
      const u64* src = part_buf(p);
      for (int t = tid; t < n * kSmallWords; t += kSmallThreads) { const u64 v = ldg_agent(src + t); if (v) s_mask[t] |= v; }
    }
    __syncthreads();
  }
  SSTAMP();
  // ---- 3. the scan (nms_rotated_cuda.cu:109-128): the lowest alive position is kept and removes what it overlaps
  if (wv == 0) {
    u64 cur = lane < kSmallWords ? s_alive[lane] : 0ull;
    int k = 0;
    const int limit = a.max_keep > 0 ? a.max_keep : 0x7fffffff;
    const int nw = (n + 63) >> 6;
    for (int w = 0; w < nw && k < limit; w++) {
      u64 pending = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(cur >> 32), w) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)cur, w);
      while (pending && k < limit) {
        const int b = __builtin_ctzll(pending), i = w * 64 + b;
        if (lane == 0) s_kept[k] = (uint32_t)i;
        k++;
        if (lane < kSmallWords) cur &= ~s_mask[i * kSmallWords + lane];
        pending = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(cur >> 32), w) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)cur, w);
        pending = b < 63 ? (pending & (~0ull << (b + 1))) : 0ull;
      }
    }
    if (lane == 0) s_nkept = k;
  }
  __syncthreads();
  SSTAMP();
  // ---- 4. out
  const int nk = s_nkept;
  if (pub) {
    // write-through stores: the output stage of the image runs in whichever workgroup finishes last, on any XCD
    const int md = FRONT::kSelf ? md_self : a.mode[seg / a.ncs];
    for (int k = tid; k < nk; k += kSmallThreads) {
      const int i = (int)s_kept[k];
      const unsigned long long key = s_pk[i];
      stg_agent(a.pub_key + (size_t)sb + k, md == 1 ? ((key << 8) | (key >> 56)) : key);
      stg_agent(a.pub_val + (size_t)sb + k, s_pv[i]);
    }
    if (tid == 0) stg_agent(a.keep_cnt + seg, nk);
  } else {
    for (int k = tid; k < nk; k += kSmallThreads) a.keep_out[(size_t)sb + k] = (int64_t)(sb + (int)s_kept[k]);
    if (tid == 0) a.keep_cnt[seg] = nk;
  }
#ifdef OBB_SMALL_TRACE
  SSTAMP();
  STRACE(3ull, (unsigned long long)seg | ((unsigned long long)part << 24) | ((unsigned long long)np << 28), (unsigned long long)n, (unsigned long long)nk, tt[0], tt[1], tt[2], tt[3], tt[4], tt[5], tt[6], (unsigned long long)nit, (unsigned long long)nd0, (unsigned long long)nd1, (unsigned long long)nd2);
  if (tid == 0 && blockIdx.x < 2048) { unsigned long long* o_ = g_small_trace2 + blockIdx.x * 8; o_[0] = acc_d0; o_[1] = acc_d1; o_[2] = acc_d2; o_[3] = acc_draw; o_[4] = acc_l0; o_[5] = acc_l1; o_[6] = acc_l2; }
  // (slot 7: the pairs waiting for the interval stage / the clip when the pooled interval pass starts, written there)
#endif
#undef SSTAMP
  return seg;
}

// TAIL::run follows the segment in every workgroup (also those of empty segments): the output stage of the fused driver counts
// the segments of an image there and runs in the one that arrives last.
template <class G, class TAIL, class FRONT = SmallFromSort>
__global__ __launch_bounds__(kSmallThreads) void k_nms_small(SmallArgs a, typename TAIL::Args ta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int seg = small_segment<G, FRONT>(a, s_raw);
  if (seg < 0) return;                                           // (workgroup-uniform)
  TAIL::run(a, ta, s_raw, seg);
}

// (the whole segment state of one workgroup must fit the CU's 160 KB: records + bit matrix + alive words + eight wave blocks,
//  + the kernel's few static words)
static_assert((size_t)kSmallMax * RotGeom::RECQ * 16 + (size_t)kSmallMax * kSmallWords * 8 + 64 + sizeof(SmallWave<RotGeom>) * kSmallWaves +
              kSmallMax * 4 + kSmallMax * 12 + 256 <= 160 * 1024, "k_nms_small: LDS budget");
static_assert(kSmallMax % 64 == 0 && kSmallMax <= 65535, "k_nms_small: pair entries are (i << 16 | j)");

template <class G>
static size_t small_lds_bytes() {
  return (size_t)kSmallMax * G::RECQ * 16 + (size_t)kSmallMax * kSmallWords * 8 + 8 * 8 + sizeof(SmallWave<G>) * kSmallWaves;
}

}  // namespace obb
