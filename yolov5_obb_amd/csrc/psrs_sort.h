// Device-wide sort of (u64 key, u32 value) pairs with UNIQUE keys -- gfx950 (included by nms.hip, namespace obb).
//
// Replaces the library sort in front of the NMS (nms_rotated_cuda.cu:81-82: scores.sort(0, descending) + index_select) for
// lists of up to kPsMaxN elements.  Parallel sorting by regular sampling in three launches, no workgroup ever waits for
// another one (no co-residency assumption, nothing spins):
//   k_ps_local_*   workgroup r makes / reads the keys of run r (kPsRun consecutive elements), sorts them in registers + LDS
//                  (sort_lds_regs below), stores the sorted run and kPsSamp REGULAR samples of it (one per stride of kPsStride elements, at a
//                  phase that differs from run to run);
//   k_ps_split     workgroup r ranks its own samples among all samples (all R x 16 of them sit in LDS: <= 32 KB); a sample
//                  whose rank is a multiple of 16 is a splitter; for every splitter it owns, the workgroup finds the cut of
//                  EVERY run (the run's samples narrow it to a window of 31 elements, five probes in global memory);
//   k_ps_bucket    workgroup j gathers, from every run, the piece between splitters j-1 and j (its output offset is the sum
//                  of the lower cuts -- no scan over buckets), sorts it in LDS and writes it to its place.
// Regular sampling bounds the bucket size whatever the input looks like: a run contributes at most kPsStride elements per
// sample of it that falls into the bucket, plus kPsStride -- so a bucket never holds more than kPsRun + R * kPsStride
// elements (8704 at 256 runs), which fits the LDS of one workgroup.  A typical bucket (random, sorted or constant scores) has
// ~kPsRun elements and goes through the same register / LDS network as the runs; larger ones (an adversarial interleaving,
// tests/test_nms_gpu.py) are merged by rank counting over their sorted pieces.  Ties cannot occur: the callers make the keys
// unique (score bits in the high word, original index / tie word in the low one), which IS the reference's documented
// order here -- descending score, ties by ascending index.  Lists longer than kPsMaxN take segsort.h's LSD radix sort.
#pragma once

namespace obb {

// Sort of npad (a power of two, 64 .. 1024*E) (key, value) pairs that sit in LDS, ascending, by npad / E (<= 1024) threads holding E
// consecutive elements each in registers.
//   1. every wave sorts its run of 64*E elements with a bitonic network that never leaves the wave: compare-exchange
//      distances below E stay inside a thread, the others are lane shuffles;
//   2. the runs are merged pairwise, log2(npad / (64*E)) levels: every element finds its rank in the sibling run with a
//      binary search in LDS (strict on one side, non-strict on the other: equal keys -- the padding -- keep distinct
//      ranks) and is written to its place; in place, two workgroup barriers per level.
// (The first version ran the whole bitonic network, 10 of its 66 stages at 2048 elements through LDS with a barrier each:
// 21.7 us on the configs[1] batch, 19.3 us now.  Replacing the lane shuffles of step 1 by DPP modifiers and the gfx950
// permlane swaps -- no LDS crossbar at all -- was measured SLOWER, 22.6 us: the phase is bound by VALU issue with 16 waves
// per CU, not by ds_bpermute.)
template <int E>
__device__ __forceinline__ void sort_lds_regs(unsigned long long* s_keys, uint32_t* s_vals, int npad, int tid) {
  constexpr int R = 64 * E;                        // run length of a wave
  const bool active = tid * E < npad;              // wave-uniform: npad is a multiple of 64
  unsigned long long k[E];
  uint32_t v[E];
#pragma unroll
  for (int e = 0; e < E; e++) { k[e] = active ? s_keys[tid * E + e] : ~0ull; v[e] = active ? s_vals[tid * E + e] : 0u; }
  const int kk_top = npad < R ? npad : R;
  if (active) {
    for (int kk = 2; kk <= kk_top; kk <<= 1) {
      const bool last = kk == R;                   // the run's final phase: every run ascending
      int j = kk >> 1;
      for (; j >= E; j >>= 1) {                    // partner element i ^ j lives in lane ^ (j / E), same slot e
        const int lx = j / E;
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int i = tid * E + e;
          const unsigned long long ok = __shfl_xor(k[e], lx);
          const uint32_t ov = __shfl_xor(v[e], lx);
          const bool take_min = ((i & j) == 0) == (last || (i & kk) == 0);
          if (take_min ? (ok < k[e]) : (ok > k[e])) { k[e] = ok; v[e] = ov; }
        }
      }
#pragma unroll
      for (int jj = E >> 1; jj >= 1; jj >>= 1) {   // distances inside the thread (compile-time slots)
        if (jj <= (kk >> 1)) {
#pragma unroll
          for (int e = 0; e < E; e++) {
            if ((e & jj) == 0) {
              const int e2 = e | jj;
              const bool up = last || (((tid * E + e) & kk) == 0);
              if ((k[e] > k[e2]) == up) {
                const unsigned long long tk = k[e]; k[e] = k[e2]; k[e2] = tk;
                const uint32_t tv = v[e]; v[e] = v[e2]; v[e2] = tv;
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int e = 0; e < E; e++) { s_keys[tid * E + e] = k[e]; s_vals[tid * E + e] = v[e]; }
  }
  __syncthreads();
  for (int len = R; len < npad; len <<= 1) {       // (workgroup-uniform)
    int dst[E];
    if (active) {
      int lo[E], hi[E];
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int idx = tid * E + e;
        const int sib = ((idx / len) ^ 1) * len;   // first element of the sibling run
        lo[e] = sib; hi[e] = sib + len;
      }
      const bool right = ((tid * E) / len) & 1;    // (all E elements of a thread sit in the same run: E divides len)
      for (int step = len; step > 0; step >>= 1) { // len is a power of two: log2(len) + 1 probes close every interval
#pragma unroll
        for (int e = 0; e < E; e++) {
          if (lo[e] < hi[e]) {
            const int mid = (lo[e] + hi[e]) >> 1;
            const unsigned long long km = s_keys[mid];
            const bool below = right ? (km <= k[e]) : (km < k[e]);     // sibling entries that precede mine
            if (below) lo[e] = mid + 1; else hi[e] = mid;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int idx = tid * E + e;
        const int run = idx / len, pos = idx - run * len;
        const int sib = (run ^ 1) * len;
        dst[e] = (run >> 1) * 2 * len + pos + (lo[e] - sib);
      }
    }
    __syncthreads();                               // every search has read the old arrangement
    if (active) {
#pragma unroll
      for (int e = 0; e < E; e++) { s_keys[dst[e]] = k[e]; s_vals[dst[e]] = v[e]; }
    }
    __syncthreads();
    if (active && (len << 1) < npad) {
#pragma unroll
      for (int e = 0; e < E; e++) { k[e] = s_keys[tid * E + e]; v[e] = s_vals[tid * E + e]; }
    }
  }
}

constexpr int kPsRun = 512;                         // elements per run = threads of the run's workgroup
constexpr int kPsSamp = 16;                         // regular samples per run
constexpr int kPsStride = kPsRun / kPsSamp;         // 32
constexpr int kPsMaxRuns = 256;
constexpr int kPsMaxN = kPsRun * kPsMaxRuns;        // 131072
constexpr int kPsBucketCap = kPsRun + kPsMaxRuns * kPsStride;   // 8704: the regular-sampling bound (see above)
constexpr int kPsSortMax = 2048;                    // buckets up to this size are sorted by the register / LDS network
constexpr unsigned long long kPsPadHi = 0xFFFFFFFF00000000ull;   // pad key = kPsPadHi | position: above every real key, unique

struct PsBuf {
  unsigned long long* run_k; uint32_t* run_v;       // sorted runs: run r at [r * kPsRun, r * kPsRun + len_r)
  unsigned long long* samp;                         // [R][kPsSamp] regular samples (keys; pad keys behind a short last run)
  int* cut;                                         // [kPsMaxRuns][kPsMaxRuns]: cut[j][r] = elements of run r that are <= splitter j
  unsigned long long* out_k; uint32_t* out_v;       // the sorted list
  const int* n_dev;                                 // optional: the element count lives on the device (<= n)
  int n;
  int* err;                                         // optional: set when a bucket exceeds its bound (cannot happen)
};
__host__ __device__ __forceinline__ int ps_phase(int r) { return (r * 13 + 5) & (kPsStride - 1); }
__device__ __forceinline__ int ps_count(const PsBuf& b) {
  int n = b.n;
  if (b.n_dev) { const int d = *b.n_dev; n = d < n ? d : n; if (n < 0) n = 0; }
  return n;
}

// the run's kPsRun (key, value) pairs sit in LDS (pads behind a short last run): sort, store the run and its samples
__device__ __forceinline__ void ps_local_tail(const PsBuf& b, int n, unsigned long long* s_k, uint32_t* s_v) {
  const int tid = threadIdx.x, r = blockIdx.x, base = r * kPsRun;
  __syncthreads();
  sort_lds_regs<1>(s_k, s_v, kPsRun, tid);
  const int len = (n - base) < kPsRun ? (n - base) : kPsRun;
  if (tid < len) { b.run_k[base + tid] = s_k[tid]; b.run_v[base + tid] = s_v[tid]; }
  // sample k = element 32k + phase(r).  The phase differs from run to run: equal-sized runs of similar data put their k-th
  // samples at nearly the same global rank, so with one phase for all the sorted samples come in 16 tight clusters and the
  // buckets that straddle a gap between clusters get several thousand elements (measured: 4807); with dithered phases the
  // samples spread evenly and a bucket holds 512 +- 300.
  if (tid < kPsSamp) b.samp[r * kPsSamp + tid] = s_k[tid * kPsStride + ps_phase(r)];
}

// runs from explicit (key, value) pairs (the fused driver's candidates of ONE large image; the count lives on the device)
__global__ __launch_bounds__(kPsRun) void k_ps_local_pairs(PsBuf b, const unsigned long long* __restrict__ kin, const uint32_t* __restrict__ vin) {
  __shared__ unsigned long long s_k[kPsRun];
  __shared__ uint32_t s_v[kPsRun];
  const int n = ps_count(b), tid = threadIdx.x, base = blockIdx.x * kPsRun;
  if (base >= n) return;
  const int i = base + tid;
  s_k[tid] = i < n ? kin[i] : (kPsPadHi | (unsigned long long)(uint32_t)i);
  s_v[tid] = i < n ? vin[i] : 0u;
  ps_local_tail(b, n, s_k, s_v);
}

__global__ __launch_bounds__(kPsRun) void k_ps_split(PsBuf b) {
  __shared__ unsigned long long s_samp[kPsMaxRuns * kPsSamp];
  __shared__ int s_rank[kPsSamp], s_spj[kPsSamp], s_nsp;
  __shared__ unsigned long long s_spk[kPsSamp];
  const int n = ps_count(b), R = (n + kPsRun - 1) / kPsRun, r = blockIdx.x, tid = threadIdx.x;
  if (r >= R || R < 2) return;                       // one run: one bucket, no splitter
  const int S = R * kPsSamp;
  for (int j = tid; j < S; j += kPsRun) s_samp[j] = b.samp[j];
  if (tid < kPsSamp) s_rank[tid] = 0;
  if (tid == 0) s_nsp = 0;
  __syncthreads();
  // rank of this run's sample q among all samples: thread (q, p) counts the entries p, p + 32, ...
  const int q = tid & (kPsSamp - 1), p = tid >> 4;
  const unsigned long long mine = s_samp[r * kPsSamp + q];
  int cnt = 0;
  for (int j = p; j < S; j += kPsRun / kPsSamp) cnt += (s_samp[j] < mine) ? 1 : 0;
  cnt += __shfl_xor(cnt, 16);
  cnt += __shfl_xor(cnt, 32);
  if ((tid & 63) < kPsSamp) atomicAdd(&s_rank[q], cnt);
  __syncthreads();
  // splitter j (j = 0 .. R-2) = the sample of rank 16 (j + 1) - 1
  if (tid < kPsSamp) {
    const int g = s_rank[tid] + 1;
    if ((g & (kPsSamp - 1)) == 0 && (g >> 4) <= R - 1) {
      const int k = atomicAdd(&s_nsp, 1);
      s_spj[k] = (g >> 4) - 1;
      s_spk[k] = s_samp[r * kPsSamp + tid];
    }
  }
  __syncthreads();
  const int nsp = s_nsp;
  for (int w = tid; w < nsp * R; w += kPsRun) {
    const int k = w / R, r2 = w - k * R;
    const unsigned long long sp = s_spk[k];
    int c;                                           // samples of run r2 that are <= sp (pads are above every splitter)
    { int lo = 0, hi = kPsSamp; while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_samp[r2 * kPsSamp + mid] <= sp) lo = mid + 1; else hi = mid; } c = lo; }
    const int len = (n - r2 * kPsRun) < kPsRun ? (n - r2 * kPsRun) : kPsRun;
    // sample c - 1 (element 32 (c - 1) + phase) is <= sp, sample c (element 32c + phase) is above it: the cut lies in
    // [32c + phase - 31, 32c + phase] (c = 0: [0, phase]; c = 16: up to the run's end); found in two rounds of independent probes
    // (3, then 8) instead of five dependent ones
    const int ph = ps_phase(r2);
    int lo = c * kPsStride + ph - (kPsStride - 1), hi = c * kPsStride + ph;
    lo = lo < 0 ? 0 : lo; lo = lo < len ? lo : len; hi = hi < len ? hi : len;
    const unsigned long long* run = b.run_k + (size_t)r2 * kPsRun;
    {
      // elements lo + 8, + 16, + 24 (where they exist below hi): the cut is behind the last one that is <= sp
      unsigned long long pr[3];
#pragma unroll
      for (int t = 0; t < 3; t++) pr[t] = (lo + 8 * (t + 1) - 1 < hi) ? run[lo + 8 * (t + 1) - 1] : ~0ull;
      int adv = 0;
#pragma unroll
      for (int t = 0; t < 3; t++) if (pr[t] <= sp) adv = 8 * (t + 1);
      lo += adv;
      unsigned long long pf[8];
#pragma unroll
      for (int t = 0; t < 8; t++) pf[t] = (lo + t < hi) ? run[lo + t] : ~0ull;
      int cntle = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) cntle += (pf[t] <= sp) ? 1 : 0;
      lo += cntle;
    }
    b.cut[(size_t)s_spj[k] * kPsMaxRuns + r2] = lo;
  }
}

__global__ __launch_bounds__(kPsRun) void k_ps_bucket(PsBuf b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];   // keys [kPsBucketCap] | values [kPsBucketCap]
  __shared__ int s_off[kPsMaxRuns + 1], s_lo[kPsMaxRuns], s_w[8][2];
  unsigned long long* s_k = reinterpret_cast<unsigned long long*>(s_raw);
  uint32_t* s_v = reinterpret_cast<uint32_t*>(s_k + kPsBucketCap);
  const int n = ps_count(b), R = (n + kPsRun - 1) / kPsRun, j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (j >= R) return;
  // piece of run `tid` that belongs to bucket j: (splitter j-1, splitter j]
  int lo = 0, hi = 0;
  if (tid < R) {
    const int len = (n - tid * kPsRun) < kPsRun ? (n - tid * kPsRun) : kPsRun;
    lo = j > 0 ? b.cut[(size_t)(j - 1) * kPsMaxRuns + tid] : 0;
    hi = j < R - 1 ? b.cut[(size_t)j * kPsMaxRuns + tid] : len;
    if (hi < lo) hi = lo;                            // (cannot happen: the cuts of a run are monotone in j)
  }
  int il = hi - lo, sl = lo;                         // inclusive scan of the piece lengths, sum of the lower cuts
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(il, d); if (lane >= d) il += u; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sl += __shfl_xor(sl, d);
  if (lane == 63) s_w[wv][0] = il;
  if (lane == 0) s_w[wv][1] = sl;
  __syncthreads();
  int pre = 0, m = 0, base = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { if (k < wv) pre += s_w[k][0]; m += s_w[k][0]; base += s_w[k][1]; }
  if (tid < R) { s_off[tid] = pre + il - (hi - lo); s_lo[tid] = lo; }
  if (tid == 0) s_off[R] = m;
  __syncthreads();
  if (m > kPsBucketCap) { if (tid == 0 && b.err) *b.err = 1; m = 0; }     // (cannot happen: the regular-sampling bound)
  // gather: element q of the bucket = element q - off[r] of piece r (r by binary search over the offsets)
  for (int q = tid; q < m; q += kPsRun) {
    int a0 = 0, a1 = R;                              // largest r with off[r] <= q
    while (a1 - a0 > 1) { const int mid = (a0 + a1) >> 1; if (s_off[mid] <= q) a0 = mid; else a1 = mid; }
    const size_t src = (size_t)a0 * kPsRun + s_lo[a0] + (q - s_off[a0]);
    s_k[q] = b.run_k[src]; s_v[q] = b.run_v[src];
  }
  if (m > 1024 && m <= kPsSortMax) {
    // A bucket just above 1024 elements (one in 196 at 100k: 1048 in S-clustered K=3000) padded to 2048 ran the four-per-thread
    // network and the whole launch waited for it (31.8 us instead of 16.9).  Two blocks instead -- the first 1024 elements and the
    // rest, padded to ITS power of two -- sorted one after the other and merged by rank on the way out (keys are unique).
    const int mb = m - 1024;
    int npb = 64;
    while (npb < mb) npb <<= 1;
    for (int q = m + tid; q < 1024 + npb; q += kPsRun) { s_k[q] = ~0ull; s_v[q] = 0u; }
    __syncthreads();
    sort_lds_regs<2>(s_k, s_v, 1024, tid);
    if (npb <= 512) sort_lds_regs<1>(s_k + 1024, s_v + 1024, npb, tid);
    else sort_lds_regs<2>(s_k + 1024, s_v + 1024, npb, tid);
    for (int q = tid; q < m; q += kPsRun) {
      const unsigned long long e = s_k[q];
      const bool first = q < 1024;
      int l2 = first ? 1024 : 0, h2 = first ? m : 1024;
      const int l0 = l2;
      while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (s_k[mid] < e) l2 = mid + 1; else h2 = mid; }
      const int rank = (first ? q : q - 1024) + (l2 - l0);
      b.out_k[(size_t)base + rank] = e; b.out_v[(size_t)base + rank] = s_v[q];
    }
  } else if (m <= kPsSortMax) {
    int npad = 64;
    while (npad < m) npad <<= 1;
    for (int q = m + tid; q < npad; q += kPsRun) { s_k[q] = ~0ull; s_v[q] = 0u; }
    __syncthreads();
    if (npad <= 512) sort_lds_regs<1>(s_k, s_v, npad, tid);
    else if (npad <= 1024) sort_lds_regs<2>(s_k, s_v, npad, tid);
    else sort_lds_regs<4>(s_k, s_v, npad, tid);
    for (int q = tid; q < m; q += kPsRun) { b.out_k[(size_t)base + q] = s_k[q]; b.out_v[(size_t)base + q] = s_v[q]; }
  } else {
    // large bucket: rank of an element = its index in its own piece + the elements of every other piece below it
    __syncthreads();
    for (int q = tid; q < m; q += kPsRun) {
      int a0 = 0, a1 = R;
      while (a1 - a0 > 1) { const int mid = (a0 + a1) >> 1; if (s_off[mid] <= q) a0 = mid; else a1 = mid; }
      const unsigned long long e = s_k[q];
      int rank = q - s_off[a0];
      for (int r2 = 0; r2 < R; r2++) {
        int l2 = s_off[r2], h2 = s_off[r2 + 1];
        if (r2 == a0 || l2 == h2) continue;
        const int l0 = l2;
        while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (s_k[mid] < e) l2 = mid + 1; else h2 = mid; }
        rank += l2 - l0;
      }
      b.out_k[(size_t)base + rank] = e; b.out_v[(size_t)base + rank] = s_v[q];
    }
  }
}

constexpr size_t kPsBucketLds = (size_t)kPsBucketCap * 12;

// scratch behind the run / output arrays: samples + cut table
static inline size_t ps_scratch_bytes() { return (size_t)kPsMaxRuns * kPsSamp * 8 + (size_t)kPsMaxRuns * kPsMaxRuns * 4 + 256; }
static inline void ps_carve_scratch(void* scratch, PsBuf* b) {
  b->samp = reinterpret_cast<unsigned long long*>(scratch);
  b->cut = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + (size_t)kPsMaxRuns * kPsSamp * 8);
}
// launches 2 and 3 (the caller has launched its k_ps_local_* flavour over `runs` workgroups)
static int ps_finish(const PsBuf& b, int runs, hipStream_t st) {
  static OncePerDevice attr;
  if (const int attr_dev = attr.need(); attr_dev != OncePerDevice::kDone) {
    if (hipFuncSetAttribute((const void*)k_ps_bucket, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPsBucketLds) != hipSuccess) return OBB_ERR_LAUNCH;
    attr.mark(attr_dev);
  }
  if (runs > 1) k_ps_split<<<(unsigned)runs, kPsRun, 0, st>>>(b);
  k_ps_bucket<<<(unsigned)runs, kPsRun, kPsBucketLds, st>>>(b);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // namespace obb
