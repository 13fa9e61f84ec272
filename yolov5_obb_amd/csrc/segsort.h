// Segmented LSD radix sort of (u64 key, u32 value) pairs for LARGE segments -- gfx950 (included by nms.hip, namespace obb).
//
// Why: a sort that gives every segment to ONE workgroup (the library's segmented sort, used in round 1; the in-LDS path of
// the fused driver) is the right shape for the speed-test regime (16 images x ~2k candidates: 0.04 ms) and the wrong one for
// val.py's default conf_thres = 0.001, where every image brings ~65k candidates: 16 workgroups ground through 1M pairs in
// 2.6 ms.  Here every segment is
// spread over as many workgroups as it has 2048-element tiles (grid = tiles x segments), one 8-bit digit per pass,
// two small kernels per pass: tile histograms -> stable scatter with the offset scan built in (wave multi-split: lanes with the
// same digit find each other with 8 ballots, so the order inside a digit is the input order).  Passes over digits that
// are constant for the whole call (unused tie / class bits) are skipped by the host.
#pragma once

namespace obb {

constexpr int kSrsTile = 2048;      // elements per workgroup (256 threads x 8 rounds of one element per lane ... per wave 512)
constexpr int kSrsThreads = 256;

struct SrsArgs {
  const unsigned long long* kin; unsigned long long* kout;
  const uint32_t* vin; uint32_t* vout;
  const int* seg_begin; const int* seg_end;     // [nseg]
  uint32_t* hist;                               // [nseg][tiles][256] tile histograms (k_srs_hist)
  int tiles;                                    // tiles per segment (capacity / kSrsTile)
  int shift;
  const float4* cand; long long cap_img;        // cand != NULL: the digit is the class of the candidate the value points to
  uint32_t* digit_base;                         // optional [nseg][256]: start of every digit's run inside the segment
};

__device__ __forceinline__ uint32_t srs_digit(const SrsArgs& a, unsigned long long key, uint32_t val, int g) {
  if (a.cand) return (uint32_t)(int)a.cand[((size_t)g * a.cap_img + val) * 2 + 1].z & 255u;
  return (uint32_t)(key >> a.shift) & 255u;
}

__global__ __launch_bounds__(kSrsThreads) void k_srs_hist(SrsArgs a) {
  __shared__ uint32_t s_h[256];
  const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int b0 = a.seg_begin[g] + tile * kSrsTile, se = a.seg_end[g];
  uint32_t* out = a.hist + ((size_t)g * a.tiles + tile) * 256;
  if (b0 >= se) { out[tid] = 0u; return; }
  s_h[tid] = 0u;
  __syncthreads();
  const int e0 = min(se, b0 + kSrsTile);
  for (int p = b0 + tid; p < e0; p += kSrsThreads) atomicAdd(&s_h[srs_digit(a, a.kin[p], a.vin[p], g)], 1u);
  __syncthreads();
  out[tid] = s_h[tid];
}

// The scatter kernel derives its tile's digit offsets itself: thread d sums the counts of digit d over the tiles before its own and
// over all tiles of the segment (8 independent loads in flight, rows of 1 KB), the 256 totals are scanned in the workgroup.
// (Round 2 had a third kernel in between, one workgroup per segment walking the tiles serially: 12 us per pass for the 31 tiles
//  of the TTA tensor, plus a launch -- the TTA call went from 0.430 to 0.323 ms without it.  Counting the NEXT pass's tile
//  histograms inside the scatter, one global atomic per element into the tile it lands in, was measured too: 0.385 ms -- the
//  elements of a tile mostly share the next digit, 2048 atomics on one counter -- so the histogram stays a launch of its own.)
__global__ __launch_bounds__(kSrsThreads) void k_srs_scatter(SrsArgs a) {
  __shared__ uint32_t s_run[4][256];     // per wave: running count of each digit inside the wave's 512-element slice
  const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sb = a.seg_begin[g], se = a.seg_end[g];
  const int b0 = sb + tile * kSrsTile;
  if (b0 >= se) return;
  __shared__ uint32_t s_off[256];
  __shared__ uint32_t s_wtot[4];
  {
    const uint32_t* h = a.hist + (size_t)g * a.tiles * 256 + tid;      // digit tid's column of the segment's tile counts
    const int nt = (se - sb + kSrsTile - 1) / kSrsTile;
    uint32_t before = 0u, tot = 0u;
    for (int t0 = 0; t0 < nt; t0 += 8) {
      uint32_t c[8];
#pragma unroll
      for (int u = 0; u < 8; u++) c[u] = (t0 + u < nt) ? h[(size_t)(t0 + u) * 256] : 0u;
#pragma unroll
      for (int u = 0; u < 8; u++) { tot += c[u]; before += (t0 + u < tile) ? c[u] : 0u; }
    }
    uint32_t incl = tot;                                                // exclusive prefix of the digit totals
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t v = __shfl_up(incl, dd); if (lane >= dd) incl += v; }
    if (lane == 63) s_wtot[wv] = incl;
    for (int k = tid; k < 4 * 256; k += kSrsThreads) (&s_run[0][0])[k] = 0u;
    __syncthreads();
    uint32_t base = incl - tot;
    for (int w = 0; w < wv; w++) base += s_wtot[w];
    s_off[tid] = base + before;
    if (a.digit_base && tile == 0) a.digit_base[(size_t)g * 256 + tid] = base;
  }
  const uint32_t* off = s_off;
  __syncthreads();
  const int w0 = b0 + wv * 512;
  unsigned long long key[8]; uint32_t val[8]; uint32_t rnk[8]; bool ok[8];
  // pass A: per round the rank among the lanes of the wave with the same digit, and the wave's digit totals
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int p = w0 + r * 64 + lane;
    ok[r] = p < se && p < b0 + kSrsTile;
    key[r] = ok[r] ? a.kin[p] : ~0ull;
    val[r] = ok[r] ? a.vin[p] : 0u;
    const uint32_t dg = ok[r] ? srs_digit(a, key[r], val[r], g) : 255u;
    unsigned long long peers = __ballot(ok[r]);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
      const unsigned long long m = __ballot((dg >> bit) & 1u);
      peers &= ((dg >> bit) & 1u) ? m : ~m;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t before = (uint32_t)__popcll(peers & lt);
    uint32_t base = 0;
    if (ok[r]) {
      base = s_run[wv][dg];                                  // all peers read the same value ...
    }
    __builtin_amdgcn_wave_barrier();
    if (ok[r] && before == 0) s_run[wv][dg] = base + (uint32_t)__popcll(peers);   // ... the first peer advances it
    __builtin_amdgcn_wave_barrier();
    rnk[r] = base + before;
  }
  __syncthreads();                                           // s_run now holds the waves' digit totals
  // pass B: global position = tile offset of the digit + same digit in earlier waves + rank inside the wave
#pragma unroll
  for (int r = 0; r < 8; r++) {
    if (!ok[r]) continue;
    const uint32_t dg = srs_digit(a, key[r], val[r], g);
    uint32_t pos = off[dg] + rnk[r];
    for (int w = 0; w < wv; w++) pos += s_run[w][dg];
    a.kout[(size_t)sb + pos] = key[r];
    a.vout[(size_t)sb + pos] = val[r];
  }
}

// Sorts every segment [seg_begin[g], seg_end[g]) (capacity `cap` elements each, `total` elements in the arrays) by the key
// bytes selected in `digit_mask` (bit d set = byte d of the key takes part), result in (kb, vb).
static int seg_radix_sort_large(unsigned long long* ka, unsigned long long* kb, uint32_t* va, uint32_t* vb, const int* seg_begin,
                                const int* seg_end, int nseg, long long cap, long long total, unsigned digit_mask, uint32_t* hist,
                                hipStream_t st) {
  SrsArgs a;
  a.seg_begin = seg_begin; a.seg_end = seg_end; a.hist = hist;
  a.tiles = (int)((cap + kSrsTile - 1) / kSrsTile);
  a.cand = nullptr; a.cap_img = 0; a.digit_base = nullptr;
  bool a_to_b = true;
  dim3 gt((unsigned)a.tiles, (unsigned)nseg);
  for (int d = 0; d < 8; d++) {
    if (!((digit_mask >> d) & 1)) continue;
    a.shift = d * 8;
    a.kin = a_to_b ? ka : kb; a.kout = a_to_b ? kb : ka;
    a.vin = a_to_b ? va : vb; a.vout = a_to_b ? vb : va;
    k_srs_hist<<<gt, kSrsThreads, 0, st>>>(a);
    k_srs_scatter<<<gt, kSrsThreads, 0, st>>>(a);
    a_to_b = !a_to_b;
  }
  if (a_to_b) {                                              // even number of passes: the result sits in (ka, va)
    if (hipMemcpyAsync(kb, ka, (size_t)total * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return OBB_ERR_LAUNCH;
    if (hipMemcpyAsync(vb, va, (size_t)total * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return OBB_ERR_LAUNCH;
  }
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

// One stable pass that groups a single-list range (already in score order; values = candidate slots of the image) by the
// candidates' class: (kin, vin) -> (kout, vout); digit_base[g][c] receives the start of class c inside segment g.
static int seg_group_by_class(const unsigned long long* kin, unsigned long long* kout, const uint32_t* vin, uint32_t* vout,
                              const int* seg_begin, const int* seg_end, int nseg, long long cap, const float4* cand, uint32_t* hist,
                              uint32_t* digit_base, hipStream_t st) {
  SrsArgs a;
  a.seg_begin = seg_begin; a.seg_end = seg_end; a.hist = hist;
  a.tiles = (int)((cap + kSrsTile - 1) / kSrsTile);
  a.shift = 0; a.cand = cand; a.cap_img = cap; a.digit_base = digit_base;
  a.kin = kin; a.kout = kout; a.vin = vin; a.vout = vout;
  dim3 gt((unsigned)a.tiles, (unsigned)nseg);
  k_srs_hist<<<gt, kSrsThreads, 0, st>>>(a);
  k_srs_scatter<<<gt, kSrsThreads, 0, st>>>(a);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // namespace obb
