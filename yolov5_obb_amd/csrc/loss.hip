// ComputeLoss for the OBB head on gfx950: build_targets + loss forward + loss backward (C ABI).
//
// Replaces utils/loss.py:122-275 (ComputeLoss.__call__ / build_targets) and the autograd graph behind it.
// The reference runs, per level, ~40 small ATen kernels with boolean-mask gathers (host syncs), replicates the
// (nt,187) target rows x3 anchors x5 offsets, and its backward materialises three dense zero tensors of the head's
// size before scatter-adding.  Here:
//
//   k_bt_count/_write   anchor-ratio match + the 5 neighbour-cell candidates (two passes: per-block counts, then an
//                       ordered write that reproduces the reference's row order: offset-major, anchor-major, target order);
//                       every entry links itself into a per-cell list (atomicExch on a dense int32 head map) so that
//                       cells hit by several entries can be resolved deterministically later.  No host sync: the
//                       counts stay on the device.  CSL rows are NOT replicated -- entries keep the target index.
//   k_loss_dense_fwd    objectness BCE of every anchor against target 0: reads only the obj logit of each row
//                       (one 64-byte sector of each 800-byte row); fixed-order partial sums.
//   k_loss_entries_fwd  one wave per entry: coalesced read of the prediction row, CIoU box term, class BCE, CSL
//                       BCE (wave reductions); the wave that owns a cell applies the objectness correction
//                       BCE(x, iou) - BCE(x, 0) with the reference's last-writer-wins rule (utils/loss.py:159).
//   k_loss_finalize     fixed-order reduction, means, gains, balance -> (loss*bs, lbox, lobj, lcls, ltheta).
//   k_loss_bwd_dense    writes the WHOLE gradient tensor once: zeros plus d lobj/d obj-logit in channel 4, 16-byte
//                       stores (the reference writes the tensor >= 3 times).
//   k_loss_entries_bwd  the owner wave of each matched cell sums the hand-written gradients of all entries of the
//                       cell in ascending entry order (deterministic) and overwrites the cell's row.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string.h>
#include "obb_hip.h"
#include "dtype_device.h"
#include "loss_math.h"

namespace obb {

constexpr int kLv = OBB_LOSS_MAX_LEVELS;
constexpr int kNa = OBB_LOSS_MAX_ANCHORS;
constexpr int kCslBins = 180;
constexpr int kDenseBlocks = 512;   // partial sums per level of the dense objectness pass
constexpr int kChunks = 7;          // ceil((5 + 256 + 180) / 64): channels of one row held by a wave

// Everything the kernels need, built on the host per call and copied to the workspace by k_loss_setup (so that the
// kernels can index the per-level tables dynamically from memory instead of from kernel-argument registers).
struct LossDev {
  int nl, na, nc, no, bs, nt, tcols, cap, sort_obj_iou, dtype;
  int ny[kLv], nx[kLv];
  long long rows[kLv], cell_off[kLv];
  float anchors[kLv][kNa][2];
  float stride[kLv], balance[kLv];
  float anchor_t, cp, cn, cls_pw, theta_pw, obj_pw, g_box, g_obj, g_cls, g_theta, gr, fl_gamma;
  int csl_from_theta;       // 1: targets are (nt,7): the CSL row is regenerated from theta (column 6)
  float csl_radius;
  float csl_lut[kCslBins];  // gaussian window y_sig[j], j = 0..179 (utils/rboxs_utils.py:22-23), filled by k_loss_setup
  const void* p[kLv];
  void* grad[kLv];
  const float* targets;
  const float* gscale;
  float* out;
  // workspace
  int* counts;            // [kLv] entries per level, [kLv] = bad-target flag
  int* head;              // [sum rows] 1 + index of the entry that linked itself last, 0 = untouched cell
  int *e_cell, *e_t, *e_ao, *prev;
  float4* e_tbox;
  float *part_box, *part_cls, *part_th, *part_obj;
  double* dense_part;     // [nl][kDenseBlocks]
  float* objcol;          // [sum rows] the objectness logit of every anchor row, written densely by the forward's dense pass
                          // (round 5): the backward reads 4 contiguous bytes per row instead of one 128-byte line of an 800-byte row
};

__global__ void k_loss_setup(LossDev d, LossDev* dst) {
  if (threadIdx.x == 0) { *dst = d; __threadfence(); }      // the struct copy (incl. a zeroed table) lands first
  __syncthreads();
  // y_sig[j] = exp(-(x_j - u)^2 / (2 sig^2)), x_j = j - 90, u = 0: evaluated in double like numpy, stored as float
  for (int j = threadIdx.x; j < kCslBins; j += blockDim.x) {
    const double x = (double)j - 90.0, sg = (double)d.csl_radius;
    dst->csl_lut[j] = (float)exp(-(x * x) / (2.0 * sg * sg));
  }
}

// CSL label of angle bin k for a target row: either stored behind the 7 geometry columns (the dataloader's (nt,187) rows,
// utils/datasets.py:637-659) or regenerated from theta (column 6) exactly as gaussian_label_cpu rolls its window
// (utils/rboxs_utils.py:24-26: index = int(90 - angle), label = concat(y[index:], y[:index]), angle = theta*180/pi + 90).
struct CslRow {
  const float* stored; const float* lut; int shift;
  __device__ __forceinline__ float at(int k) const {
    if (stored) return stored[k];
    int j = k + shift;
    if (j >= kCslBins) j -= kCslBins;
    return lut[j];
  }
};
__device__ __forceinline__ CslRow csl_row(const LossDev& d, const float* tr) {
  CslRow r;
  r.lut = d.csl_lut; r.shift = 0; r.stored = nullptr;
  if (!d.csl_from_theta) { r.stored = tr + 7; return r; }
  const double angle = (double)tr[6] * 180.0 / 3.141592 + 90.0;      // utils/rboxs_utils.py:72 with pi = 3.141592
  const long long index = (long long)(90.0 - angle);                   // int() truncates toward zero
  if (index <= kCslBins && index >= -(long long)kCslBins) { long long sft = index % kCslBins; if (sft < 0) sft += kCslBins; r.shift = (int)sft; }
  return r;
}

__device__ __forceinline__ unsigned long long lanemask_lt_l() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// ------------------------------------------------------------------ build_targets (utils/loss.py:194-275)
struct BtCand {          // one (offset, anchor, target) candidate of a level
  bool ok, bad;
  int t, a, o, b;
  float gx, gy, gl, gs;
};

__device__ __forceinline__ BtCand bt_eval(const LossDev& d, int lv, long long idx, long long E) {
  BtCand c;
  c.ok = false; c.bad = false; c.t = c.a = c.o = c.b = 0; c.gx = c.gy = c.gl = c.gs = 0.f;
  if (idx >= E) return c;
  const int nt = d.nt, na = d.na, nx = d.nx[lv], ny = d.ny[lv];
  const float st = d.stride[lv];
  c.o = (int)(idx / ((long long)na * nt));
  const long long rem = idx - (long long)c.o * na * nt;
  c.a = (int)(rem / nt);
  c.t = (int)(rem - (long long)c.a * nt);
  const float* tr = d.targets + (size_t)c.t * d.tcols;
  c.gx = tr[2] / st; c.gy = tr[3] / st; c.gl = tr[4] / st; c.gs = tr[5] / st;          // :234
  const float r0 = c.gl / d.anchors[lv][c.a][0], r1 = c.gs / d.anchors[lv][c.a][1];    // :237
  bool ok = fmaxf(fmaxf(r0, 1.0f / r0), fmaxf(r1, 1.0f / r1)) < d.anchor_t;           // :238
  if (r0 != r0 || r1 != r1) ok = false;                                                // torch.max propagates NaN
  if (ok && c.o > 0) {                                                                 // :243-250, g = 0.5
    if (c.o == 1) ok = remainder1_f(c.gx) < 0.5f && c.gx > 1.0f;
    else if (c.o == 2) ok = remainder1_f(c.gy) < 0.5f && c.gy > 1.0f;
    else if (c.o == 3) { const float xi = (float)nx - c.gx; ok = remainder1_f(xi) < 0.5f && xi > 1.0f; }
    else { const float yi = (float)ny - c.gy; ok = remainder1_f(yi) < 0.5f && yi > 1.0f; }
  }
  if (ok) {
    c.b = (int)tr[0];                                                                  // :256  .long() truncates
    const int cls = (int)tr[1];
    if (c.b < 0 || c.b >= d.bs || cls < 0 || cls >= d.nc) { ok = false; c.bad = true; }   // the reference raises IndexError
  }
  c.ok = ok;
  return c;
}

// pass 1: matches per 1024-candidate block
__global__ __launch_bounds__(1024) void k_bt_count(const LossDev* __restrict__ dp, int* __restrict__ blkcnt) {
  __shared__ int s_wave[16];
  const LossDev& d = *dp;
  const int lv = blockIdx.y, tid = threadIdx.x;
  const long long E = 5LL * d.na * d.nt;
  const BtCand c = bt_eval(d, lv, (long long)blockIdx.x * 1024 + tid, E);
  const unsigned long long bal = __ballot(c.ok);
  if ((tid & 63) == 0) s_wave[tid >> 6] = __popcll(bal);
  if (c.bad) atomicOr(&d.counts[kLv], 1);
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) tot += s_wave[w];
    blkcnt[lv * gridDim.x + blockIdx.x] = tot;
  }
}

// pass 2: ordered write -- row order = candidate order (offset-major, anchor-major, target order), as the reference
__global__ __launch_bounds__(1024) void k_bt_write(const LossDev* __restrict__ dp, const int* __restrict__ blkcnt) {
  __shared__ int s_wave[16];
  __shared__ int s_part[16];
  const LossDev& d = *dp;
  const int lv = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nx = d.nx[lv], ny = d.ny[lv], na = d.na;
  const long long E = 5LL * na * d.nt;
  // matches in the blocks before this one (and, for the last block, the level's total)
  int before = 0;
  for (int k = tid; k < (int)blockIdx.x; k += 1024) before += blkcnt[lv * gridDim.x + k];
#pragma unroll
  for (int s2 = 32; s2 >= 1; s2 >>= 1) before += __shfl_xor(before, s2);
  if (lane == 0) s_part[wv] = before;
  const BtCand c = bt_eval(d, lv, (long long)blockIdx.x * 1024 + tid, E);
  const unsigned long long bal = __ballot(c.ok);
  const int rank = __popcll(bal & lanemask_lt_l());
  if (lane == 0) s_wave[wv] = __popcll(bal);
  __syncthreads();
  int base = 0, wpre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { base += s_part[w]; const int q = s_wave[w]; if (w < wv) wpre += q; tot += q; }
  if (c.ok) {
    const int pos = base + wpre + rank;
    const int eg = lv * d.cap + pos;
    const float ox = (c.o == 1) ? 0.5f : (c.o == 3) ? -0.5f : 0.0f;                  // :220-224 off * g
    const float oy = (c.o == 2) ? 0.5f : (c.o == 4) ? -0.5f : 0.0f;
    int gi = (int)(c.gx - ox), gj = (int)(c.gy - oy);                                // :261  (gxy - offsets).long()
    gi = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);                                   // :267  clamp_ (mutates gij)
    gj = gj < 0 ? 0 : (gj > ny - 1 ? ny - 1 : gj);
    const int cell = ((c.b * na + c.a) * ny + gj) * nx + gi;
    d.e_cell[eg] = cell; d.e_t[eg] = c.t; d.e_ao[eg] = c.a | (c.o << 8);
    d.e_tbox[eg] = make_float4(c.gx - (float)gi, c.gy - (float)gj, c.gl, c.gs);      // :268
    d.prev[eg] = atomicExch(&d.head[d.cell_off[lv] + cell], eg + 1);
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) d.counts[lv] = base + tot;
}

// rows of the build_targets() return value for one level (utils/loss.py:265-272)
__global__ void k_bt_export(const LossDev* __restrict__ dp, int lv, int n, int64_t* __restrict__ idx4, float* __restrict__ tbox4,
                            float* __restrict__ anch2, int64_t* __restrict__ tcls, float* __restrict__ csl) {
  const LossDev& d = *dp;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e >= n) return;
  const int eg = lv * d.cap + e;
  const int cell = d.e_cell[eg], t = d.e_t[eg], a = d.e_ao[eg] & 255;
  const int nx = d.nx[lv], ny = d.ny[lv];
  if (lane == 0) {
    const int gi = cell % nx, r1 = cell / nx, gj = r1 % ny, r2 = r1 / ny, b = r2 / d.na;
    idx4[(size_t)e * 4 + 0] = b; idx4[(size_t)e * 4 + 1] = a; idx4[(size_t)e * 4 + 2] = gj; idx4[(size_t)e * 4 + 3] = gi;
    const float4 tb = d.e_tbox[eg];
    tbox4[(size_t)e * 4 + 0] = tb.x; tbox4[(size_t)e * 4 + 1] = tb.y; tbox4[(size_t)e * 4 + 2] = tb.z; tbox4[(size_t)e * 4 + 3] = tb.w;
    anch2[(size_t)e * 2 + 0] = d.anchors[lv][a][0]; anch2[(size_t)e * 2 + 1] = d.anchors[lv][a][1];
    tcls[e] = (int64_t)(int)d.targets[(size_t)t * d.tcols + 1];
  }
  const CslRow row = csl_row(d, d.targets + (size_t)t * d.tcols);
  for (int k = lane; k < kCslBins; k += 64) csl[(size_t)e * kCslBins + k] = row.at(k);
}

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

// ------------------------------------------------------------------ dense objectness pass (utils/loss.py:178-179, tobj = 0)
template <typename T>
__global__ __launch_bounds__(256) void k_loss_dense_fwd(const LossDev* __restrict__ dp) {
  __shared__ double s_part[4];
  const LossDev& d = *dp;
  const int lv = blockIdx.y, tid = threadIdx.x;
  const long long rows = d.rows[lv];
  const T* p = (const T*)d.p[lv];
  const int no = d.no;
  const float pw = d.obj_pw, fg = d.fl_gamma;
  float* col = d.objcol + d.cell_off[lv];              // the level's slice of the dense logit column (the backward's input)
  float acc = 0.f;
  const long long step = (long long)gridDim.x * 256;
  long long r = (long long)blockIdx.x * 256 + tid;
  for (; r + 3 * step < rows; r += 4 * step) {       // 4 independent strided loads in flight
    const float x0 = ld_as_float<T>(p + (size_t)r * no + 4);
    const float x1 = ld_as_float<T>(p + (size_t)(r + step) * no + 4);
    const float x2 = ld_as_float<T>(p + (size_t)(r + 2 * step) * no + 4);
    const float x3 = ld_as_float<T>(p + (size_t)(r + 3 * step) * no + 4);
    col[r] = x0; col[r + step] = x1; col[r + 2 * step] = x2; col[r + 3 * step] = x3;
    acc += bce_focal(x0, 0.f, pw, fg); acc += bce_focal(x1, 0.f, pw, fg); acc += bce_focal(x2, 0.f, pw, fg); acc += bce_focal(x3, 0.f, pw, fg);
  }
  for (; r < rows; r += step) { const float x = ld_as_float<T>(p + (size_t)r * no + 4); col[r] = x; acc += bce_focal(x, 0.f, pw, fg); }
  const double w = wave_sum_d((double)acc);
  if ((tid & 63) == 0) s_part[tid >> 6] = w;
  __syncthreads();
  if (tid == 0) d.dense_part[lv * kDenseBlocks + blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// ------------------------------------------------------------------ per-entry terms
template <typename T>
struct RowRegs { float x[kChunks]; };

template <typename T>
__device__ __forceinline__ RowRegs<T> load_row(const T* row, int no, int lane) {
  RowRegs<T> r;
#pragma unroll
  for (int k = 0; k < kChunks; k++) {
    const int ch = lane + 64 * k;
    r.x[k] = (ch < no) ? ld_as_float<T>(row + ch) : 0.f;
  }
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void k_loss_entries_fwd(const LossDev* __restrict__ dp) {
  const LossDev& d = *dp;
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  const int cap_tot = d.nl * d.cap;
  const int no = d.no, nc = d.nc;
  for (int eg = blockIdx.x * 4 + (threadIdx.x >> 6); eg < cap_tot; eg += nw) {
    const int lv = eg / d.cap, pos = eg - lv * d.cap;
    if (pos >= d.counts[lv]) continue;
    const int cell = d.e_cell[eg], t = d.e_t[eg], a = d.e_ao[eg] & 255;
    const T* row = (const T*)d.p[lv] + (size_t)cell * no;
    const RowRegs<T> R = load_row<T>(row, no, lane);
    const float l0 = __shfl(R.x[0], 0), l1 = __shfl(R.x[0], 1), l2 = __shfl(R.x[0], 2), l3 = __shfl(R.x[0], 3);
    const float x4 = __shfl(R.x[0], 4);
    const PredBox pb = loss_pred_box(l0, l1, l2, l3, d.anchors[lv][a][0], d.anchors[lv][a][1]);    // :148-149
    const float4 tb = d.e_tbox[eg];
    const CiouOut co = ciou_fwd_bwd(pb.x, pb.y, pb.w, pb.h, tb.x, tb.y, tb.z, tb.w);               // :151
    const float* tr = d.targets + (size_t)t * d.tcols;
    const int tc = (int)tr[1];
    const CslRow csl = csl_row(d, tr);
    float scls = 0.f, sth = 0.f;
#pragma unroll
    for (int k = 0; k < kChunks; k++) {
      const int ch = lane + 64 * k;
      if (ch >= 5 && ch < 5 + nc) scls += bce_focal(R.x[k], (ch - 5 == tc) ? d.cp : d.cn, d.cls_pw, d.fl_gamma);   // :162-168
      else if (ch >= 5 + nc && ch < no) sth += bce_focal(R.x[k], csl.at(ch - 5 - nc), d.theta_pw, d.fl_gamma);     // :171-172
    }
    scls = wave_sum_f(scls); sth = wave_sum_f(sth);
    float corr = 0.f;
    if (d.head[d.cell_off[lv] + cell] == eg + 1) {            // this entry owns the cell: resolve tobj[b,a,gj,gi] (:159)
      int w = eg;
      float best = co.ciou;
      for (int j = d.prev[eg]; j > 0; j = d.prev[j - 1]) {
        const int jj = j - 1;
        if (!d.sort_obj_iou) { if (jj > w) w = jj; }           // last writer in row order wins
        else {                                                  // :156-158 rows sorted by iou: the largest is written last
          const float4 tj = d.e_tbox[jj];
          const float cj = ciou_fwd_bwd(pb.x, pb.y, pb.w, pb.h, tj.x, tj.y, tj.z, tj.w).ciou;
          if (cj > best || (cj == best && jj > w)) { best = cj; w = jj; }
        }
      }
      float iw = best;
      if (!d.sort_obj_iou && w != eg) {
        const float4 tj = d.e_tbox[w];
        iw = ciou_fwd_bwd(pb.x, pb.y, pb.w, pb.h, tj.x, tj.y, tj.z, tj.w).ciou;
      }
      const float tobj = round_to_dtype<T>((1.0f - d.gr) + d.gr * round_to_dtype<T>(fmaxf(iw, 0.f)));   // :155,159
      corr = bce_focal(x4, tobj, d.obj_pw, d.fl_gamma) - bce_focal(x4, 0.f, d.obj_pw, d.fl_gamma);
    }
    if (lane == 0) {
      d.part_box[eg] = 1.0f - co.ciou;                                                           // :152
      d.part_cls[eg] = scls; d.part_th[eg] = sth; d.part_obj[eg] = corr;
    }
  }
}

__device__ __forceinline__ double block_sum_1024(double v, double* s_tmp) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < 16; w++) t += s_tmp[w];
  return t;
}

// one workgroup per level reduces the level's partial sums in a fixed order; the workgroup that finishes last (ticket)
// combines the levels in level order.  Inter-workgroup hand-off through agent-scope (write-through) stores / loads.
__global__ __launch_bounds__(1024) void k_loss_finalize(const LossDev* __restrict__ dp, float* __restrict__ lvl_out, int* __restrict__ ticket) {
  __shared__ double s_tmp[16];
  __shared__ int s_last;
  const LossDev& d = *dp;
  const int tid = threadIdx.x, lv = blockIdx.x;
  {
    const int n = d.counts[lv];
    double sb = 0.0, sc = 0.0, st = 0.0, so = 0.0, sd = 0.0;
    for (int e = tid; e < n; e += 1024) {
      const int eg = lv * d.cap + e;
      sb += (double)d.part_box[eg]; sc += (double)d.part_cls[eg]; st += (double)d.part_th[eg]; so += (double)d.part_obj[eg];
    }
    for (int k = tid; k < kDenseBlocks; k += 1024) sd += d.dense_part[lv * kDenseBlocks + k];
    sb = block_sum_1024(sb, s_tmp); sc = block_sum_1024(sc, s_tmp); st = block_sum_1024(st, s_tmp);
    so = block_sum_1024(so, s_tmp); sd = block_sum_1024(sd, s_tmp);
    if (tid == 0) {
      float lb = 0.f, lc = 0.f, lt = 0.f;
      if (n > 0) {
        lb = (float)(sb / (double)n);                                      // :152  (1.0 - iou).mean()
        if (d.nc > 1) lc = (float)(sc / ((double)n * d.nc));               // :168
        lt = (float)(st / ((double)n * kCslBins));                         // :172
      }
      const float obji = (float)((sd + so) / (double)d.rows[lv]);          // :178
      __hip_atomic_store(lvl_out + lv * 4 + 0, lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(lvl_out + lv * 4 + 1, lc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(lvl_out + lv * 4 + 2, lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(lvl_out + lv * 4 + 3, obji, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
  }
  if (s_last && tid == 0) {
    float lbox = 0.f, lobj = 0.f, lcls = 0.f, lth = 0.f;
    for (int l = 0; l < d.nl; l++) {
      lbox += __hip_atomic_load(lvl_out + l * 4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lcls += __hip_atomic_load(lvl_out + l * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lth += __hip_atomic_load(lvl_out + l * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float obji = __hip_atomic_load(lvl_out + l * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lobj += obji * d.balance[l];                                          // :179
      d.out[5 + l] = obji;
    }
    lbox *= d.g_box; lobj *= d.g_obj; lcls *= d.g_cls; lth *= d.g_theta;    // :185-188
    float total = (lbox + lobj + lcls + lth) * (float)d.bs;                 // :192
    if (d.counts[kLv]) total = __builtin_nanf("");                          // a target row pointed outside the batch / classes
    d.out[0] = total; d.out[1] = lbox; d.out[2] = lobj; d.out[3] = lcls; d.out[4] = lth;
  }
}

// ------------------------------------------------------------------ backward
template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int V = 4; };
template <> struct Vec16<__half> { static constexpr int V = 8; };

typedef unsigned int loss_u32x4 __attribute__((ext_vector_type(4)));

// NTS: the gradient tensor is written once and read much later (the conv's backward): streaming stores
template <typename T, bool NTS>
__global__ __launch_bounds__(256) void k_loss_bwd_dense(const LossDev* __restrict__ dp) {
  constexpr int V = Vec16<T>::V;
  const LossDev& d = *dp;
  const int lv = blockIdx.y, lane = threadIdx.x & 63;
  const long long rows = d.rows[lv];
  const int no = d.no;
  const float* col = d.objcol + d.cell_off[lv];
  T* g = (T*)d.grad[lv];
  const float inv_no = 1.0f / (float)no;
  // a target row that named an image or class outside the batch (the reference raises IndexError) made the loss NaN in
  // the forward pass; the gradient is NaN as well, so that a GradScaler skips the step instead of applying a partial one
  const float gin = d.counts[kLv] ? __builtin_nanf("") : d.gscale[0];
  const float gs = gin * d.g_obj * (float)d.bs * d.balance[lv] / (float)rows;
  const long long nreg = (rows + 63) >> 6;
  for (long long rg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); rg < nreg; rg += (long long)gridDim.x * 4) {
    const long long R0 = rg << 6;
    const int nr = (int)((rows - R0) < 64 ? (rows - R0) : 64);
    // d BCE(x, 0)/dx = sigmoid(x) for every pos_weight (FocalLoss: the general form)
    // (the logit from the column the forward left: one coalesced 256-byte read per 64 rows -- reading p[row][4] cost a 128-byte line
    //  per row, 132 MB next to the 830 MB this kernel writes; the value is the same float)
    const float gv = (lane < nr) ? bce_focal_grad(col[R0 + lane], 0.f, d.obj_pw, d.fl_gamma) * gs : 0.f;
    const int nel = nr * no;
    T* gb = g + (size_t)R0 * no;                      // 64*no*sizeof(T) bytes per region: 16-byte aligned
    const int nch = (nel + V - 1) / V;
    for (int c0 = 0; c0 < nch; c0 += 64) {
      const int c = c0 + lane;
      const int i0 = c * V;
      int k0 = (int)((float)i0 * inv_no);
      if (k0 * no > i0) k0--;
      if ((k0 + 1) * no <= i0) k0++;
      const int ch0 = i0 - k0 * no;
      int j = -1, krow = 0;                           // position of a channel-4 element inside this chunk, its row
      if (ch0 <= 4 && 4 < ch0 + V) { j = 4 - ch0; krow = k0; }
      else if (ch0 + V > no + 4) { j = no + 4 - ch0; krow = k0 + 1; }
      const float val = __shfl(gv, krow & 63);        // executed by all lanes
      if (c < nch) {
        if (i0 + V <= nel) {
          if constexpr (V == 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j == 0) v.x = val; else if (j == 1) v.y = val; else if (j == 2) v.z = val; else if (j == 3) v.w = val;
            if constexpr (NTS) __builtin_nontemporal_store(__builtin_bit_cast(loss_u32x4, v), reinterpret_cast<loss_u32x4*>(gb + i0));
            else *reinterpret_cast<float4*>(gb + i0) = v;
          } else {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (j >= 0) {
              const uint32_t h = (uint32_t)__half_as_ushort(__float2half_rn(val)) << ((j & 1) * 16);
              const int q = j >> 1;
              if (q == 0) v.x = h; else if (q == 1) v.y = h; else if (q == 2) v.z = h; else v.w = h;
            }
            if constexpr (NTS) __builtin_nontemporal_store(__builtin_bit_cast(loss_u32x4, v), reinterpret_cast<loss_u32x4*>(gb + i0));
            else *reinterpret_cast<uint4*>(gb + i0) = v;
          }
        } else {
          for (int q = 0; q < V && i0 + q < nel; q++) st_from_float<T>(gb + i0 + q, q == j ? val : 0.f);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_loss_entries_bwd(const LossDev* __restrict__ dp) {
  const LossDev& d = *dp;
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  const int cap_tot = d.nl * d.cap;
  const int no = d.no, nc = d.nc;
  const float gsc = d.counts[kLv] ? __builtin_nanf("") : d.gscale[0];      // (bad target rows: NaN, see k_loss_bwd_dense)
  for (int eg = blockIdx.x * 4 + (threadIdx.x >> 6); eg < cap_tot; eg += nw) {
    const int lv = eg / d.cap, pos = eg - lv * d.cap;
    const int n = d.counts[lv];
    if (pos >= n) continue;
    const int cell = d.e_cell[eg];
    if (d.head[d.cell_off[lv] + cell] != eg + 1) continue;     // only the owner of the cell writes its row
    const int a = d.e_ao[eg] & 255;
    const T* row = (const T*)d.p[lv] + (size_t)cell * no;
    const RowRegs<T> R = load_row<T>(row, no, lane);
    const float l0 = __shfl(R.x[0], 0), l1 = __shfl(R.x[0], 1), l2 = __shfl(R.x[0], 2), l3 = __shfl(R.x[0], 3);
    const float x4 = __shfl(R.x[0], 4);
    const PredBox pb = loss_pred_box(l0, l1, l2, l3, d.anchors[lv][a][0], d.anchors[lv][a][1]);
    const float fb = (float)d.bs;
    const float s_box = gsc * d.g_box * fb / (float)n;
    const float s_cls = gsc * d.g_cls * fb / ((float)n * (float)nc);
    const float s_th = gsc * d.g_theta * fb / ((float)n * (float)kCslBins);
    const float s_obj = gsc * d.g_obj * fb * d.balance[lv] / (float)d.rows[lv];
    float acc[kChunks];
#pragma unroll
    for (int k = 0; k < kChunks; k++) acc[k] = 0.f;
    int cur = -1, w = eg;
    float best = -__builtin_inff(), iou_w = 0.f;
    for (;;) {                                                   // entries of the cell in ascending index order
      int nxt = 0x7fffffff;
      for (int j = eg + 1; j > 0; j = d.prev[j - 1]) { const int jj = j - 1; if (jj > cur && jj < nxt) nxt = jj; }
      if (nxt == 0x7fffffff) break;
      cur = nxt;
      const float4 tb = d.e_tbox[cur];
      const CiouOut co = ciou_fwd_bwd(pb.x, pb.y, pb.w, pb.h, tb.x, tb.y, tb.z, tb.w);
      if (!d.sort_obj_iou) { w = cur; iou_w = co.ciou; }
      else if (co.ciou >= best) { best = co.ciou; w = cur; iou_w = co.ciou; }
      const float* tr = d.targets + (size_t)d.e_t[cur] * d.tcols;
      const int tc = (int)tr[1];
      const CslRow csl = csl_row(d, tr);
      // d(1 - ciou)/d logit for channels 0..3
      const float gbox = (lane == 0) ? -co.d[0] * pb.dx : (lane == 1) ? -co.d[1] * pb.dy : (lane == 2) ? -co.d[2] * pb.dw : -co.d[3] * pb.dh;
#pragma unroll
      for (int k = 0; k < kChunks; k++) {
        const int ch = lane + 64 * k;
        if (k == 0 && ch < 4) acc[k] += gbox * s_box;
        else if (ch >= 5 && ch < 5 + nc) { if (nc > 1) acc[k] += bce_focal_grad(R.x[k], (ch - 5 == tc) ? d.cp : d.cn, d.cls_pw, d.fl_gamma) * s_cls; }
        else if (ch >= 5 + nc && ch < no) acc[k] += bce_focal_grad(R.x[k], csl.at(ch - 5 - nc), d.theta_pw, d.fl_gamma) * s_th;
      }
    }
    (void)w;
    const float tobj = round_to_dtype<T>((1.0f - d.gr) + d.gr * round_to_dtype<T>(fmaxf(iou_w, 0.f)));
    const float g4 = bce_focal_grad(x4, tobj, d.obj_pw, d.fl_gamma) * s_obj;
    T* grow = (T*)d.grad[lv] + (size_t)cell * no;
#pragma unroll
    for (int k = 0; k < kChunks; k++) {
      const int ch = lane + 64 * k;
      if (ch < no) st_from_float<T>(grow + ch, (ch == 4) ? g4 : acc[k]);
    }
  }
}

// ------------------------------------------------------------------ host side
static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

struct LossCarve {
  LossDev* dev;
  int* counts; int* head; int *e_cell, *e_t, *e_ao, *prev; float4* e_tbox;
  float *part_box, *part_cls, *part_th, *part_obj; double* dense_part; float* objcol;
  int* blkcnt; float* lvl_out;
  size_t head_bytes, total;
};

static int loss_check(const obb_loss_config* c, int64_t nt) {
  if (!c || c->nl < 1 || c->nl > kLv || c->na < 1 || c->na > kNa || c->nc < 1 || c->nc > 256 || c->bs < 1 || nt < 0) return OBB_ERR_BAD_ARG;
  if (c->no != 5 + c->nc + kCslBins) return OBB_ERR_BAD_ARG;
  long long tot = 0;
  for (int i = 0; i < c->nl; i++) {
    if (c->ny[i] < 1 || c->nx[i] < 1) return OBB_ERR_BAD_ARG;
    const long long r = (long long)c->bs * c->na * c->ny[i] * c->nx[i];
    if (r > 0x7fffffffLL / 2) return OBB_ERR_BAD_ARG;
    tot += r;
  }
  if (tot > 0x7fffffffLL || 5LL * c->na * nt * c->nl > 0x3fffffffLL) return OBB_ERR_BAD_ARG;
  return OBB_OK;
}

static void loss_carve(void* base, const obb_loss_config* c, int64_t nt, LossCarve* cv) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += al(bytes ? bytes : 16); return base ? (char*)base + o : (char*)nullptr; };
  long long tot = 0;
  for (int i = 0; i < c->nl; i++) tot += (long long)c->bs * c->na * c->ny[i] * c->nx[i];
  const size_t ce = (size_t)5 * c->na * (size_t)nt * c->nl;
  cv->dev = (LossDev*)take(sizeof(LossDev));
  cv->counts = (int*)take((kLv + 2) * 4);      // [kLv] entries per level | bad-target flag | finalize ticket
  cv->head_bytes = (size_t)tot * 4;
  cv->head = (int*)take(cv->head_bytes);
  cv->e_cell = (int*)take(ce * 4); cv->e_t = (int*)take(ce * 4); cv->e_ao = (int*)take(ce * 4); cv->prev = (int*)take(ce * 4);
  cv->e_tbox = (float4*)take(ce * 16);
  cv->part_box = (float*)take(ce * 4); cv->part_cls = (float*)take(ce * 4); cv->part_th = (float*)take(ce * 4); cv->part_obj = (float*)take(ce * 4);
  cv->dense_part = (double*)take((size_t)kLv * kDenseBlocks * 8);
  cv->objcol = (float*)take((size_t)tot * 4);
  cv->lvl_out = (float*)take(kLv * 4 * 4);
  cv->blkcnt = (int*)take(((size_t)5 * c->na * (size_t)nt / 1024 + 2) * c->nl * 4);
  cv->total = off;
}

static void loss_fill(LossDev& d, const obb_loss_config* c, const LossCarve& cv, const float* targets, int64_t nt, int64_t tcols) {
  memset(&d, 0, sizeof d);
  d.nl = c->nl; d.na = c->na; d.nc = c->nc; d.no = c->no; d.bs = c->bs; d.nt = (int)nt; d.tcols = (int)tcols;
  d.cap = (int)(5LL * c->na * nt); d.sort_obj_iou = c->sort_obj_iou;
  long long off = 0;
  for (int i = 0; i < c->nl; i++) {
    d.ny[i] = c->ny[i]; d.nx[i] = c->nx[i];
    d.rows[i] = (long long)c->bs * c->na * c->ny[i] * c->nx[i];
    d.cell_off[i] = off; off += d.rows[i];
    d.stride[i] = c->stride[i]; d.balance[i] = c->balance[i];
    for (int a = 0; a < c->na; a++) { d.anchors[i][a][0] = c->anchors[i][a][0]; d.anchors[i][a][1] = c->anchors[i][a][1]; }
  }
  d.anchor_t = c->anchor_t; d.cp = c->cp; d.cn = c->cn; d.cls_pw = c->cls_pw; d.theta_pw = c->theta_pw; d.obj_pw = c->obj_pw;
  d.g_box = c->gain_box; d.g_obj = c->gain_obj; d.g_cls = c->gain_cls; d.g_theta = c->gain_theta; d.gr = c->gr;
  d.fl_gamma = c->fl_gamma > 0.f ? c->fl_gamma : 0.f;
  d.csl_from_theta = (tcols < 7 + kCslBins) ? 1 : 0;
  d.csl_radius = c->csl_radius > 0.f ? c->csl_radius : 2.0f;
  d.targets = targets;
  d.counts = cv.counts; d.head = cv.head; d.e_cell = cv.e_cell; d.e_t = cv.e_t; d.e_ao = cv.e_ao; d.prev = cv.prev;
  d.e_tbox = cv.e_tbox; d.part_box = cv.part_box; d.part_cls = cv.part_cls; d.part_th = cv.part_th; d.part_obj = cv.part_obj;
  d.dense_part = cv.dense_part; d.objcol = cv.objcol;
}

static int run_match(const obb_loss_config* c, const LossCarve& cv, const LossDev& d, hipStream_t st) {
  if (hipMemsetAsync(cv.counts, 0, (kLv + 2) * 4, st) != hipSuccess) return OBB_ERR_LAUNCH;
  if (hipMemsetAsync(cv.head, 0, cv.head_bytes, st) != hipSuccess) return OBB_ERR_LAUNCH;
  k_loss_setup<<<1, 256, 0, st>>>(d, cv.dev);
  if (d.nt > 0) {
    const unsigned nblk = (unsigned)((5LL * c->na * d.nt + 1023) / 1024);
    dim3 gb(nblk, c->nl);
    k_bt_count<<<gb, 1024, 0, st>>>(cv.dev, cv.blkcnt);
    k_bt_write<<<gb, 1024, 0, st>>>(cv.dev, cv.blkcnt);
  }
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

static unsigned entry_grid(const LossDev& d) {
  long long g = ((long long)d.nl * d.cap + 3) / 4;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace obb

using namespace obb;

extern "C" {

size_t obb_loss_workspace_bytes(const obb_loss_config* cfg, int64_t nt) {
  if (loss_check(cfg, nt)) return 0;
  LossCarve cv;
  loss_carve(nullptr, cfg, nt, &cv);
  return cv.total;
}

int obb_loss_build_targets(const obb_loss_config* cfg, const float* targets, int64_t nt, int64_t tcols, int32_t* counts_out,
                           void* ws, size_t ws_bytes, void* stream) {
  int rc = loss_check(cfg, nt);
  if (rc) return rc;
  if ((nt > 0 && !targets) || (nt > 0 && tcols < 7) || !counts_out) return OBB_ERR_BAD_ARG;
  LossCarve cv;
  loss_carve(ws, cfg, nt, &cv);
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  LossDev d;
  loss_fill(d, cfg, cv, targets, nt, tcols);
  rc = run_match(cfg, cv, d, st);
  if (rc) return rc;
  if (hipMemcpyAsync(counts_out, cv.counts, (kLv + 1) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return OBB_ERR_LAUNCH;
  return OBB_OK;
}

int obb_loss_export_targets(const obb_loss_config* cfg, int64_t nt, int level, int64_t n, int64_t* indices4, float* tbox4,
                            float* anch2, int64_t* tcls, float* csl180, void* ws, size_t ws_bytes, void* stream) {
  int rc = loss_check(cfg, nt);
  if (rc) return rc;
  if (level < 0 || level >= cfg->nl || n < 0 || n > 5LL * cfg->na * nt) return OBB_ERR_BAD_ARG;
  if (n == 0) return OBB_OK;
  if (!indices4 || !tbox4 || !anch2 || !tcls || !csl180) return OBB_ERR_BAD_ARG;
  LossCarve cv;
  loss_carve(ws, cfg, nt, &cv);
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  k_bt_export<<<(unsigned)((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(cv.dev, level, (int)n, indices4, tbox4, anch2, tcls, csl180);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

int obb_loss_forward(const obb_loss_config* cfg, const void* const* p_levels_host, int dtype, const float* targets, int64_t nt,
                     int64_t tcols, float* loss_out, void* ws, size_t ws_bytes, void* stream) {
  int rc = loss_check(cfg, nt);
  if (rc) return rc;
  if (!p_levels_host || !loss_out || (nt > 0 && !targets) || (nt > 0 && tcols < 7) || (dtype != 0 && dtype != 1))
    return OBB_ERR_BAD_ARG;
  LossCarve cv;
  loss_carve(ws, cfg, nt, &cv);
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  LossDev d;
  loss_fill(d, cfg, cv, targets, nt, tcols);
  d.dtype = dtype; d.out = loss_out;
  for (int i = 0; i < cfg->nl; i++) { if (!p_levels_host[i]) return OBB_ERR_BAD_ARG; d.p[i] = p_levels_host[i]; }
  rc = run_match(cfg, cv, d, st);
  if (rc) return rc;
  dim3 gd(kDenseBlocks, cfg->nl);
  if (dtype == 0) k_loss_dense_fwd<float><<<gd, 256, 0, st>>>(cv.dev);
  else k_loss_dense_fwd<__half><<<gd, 256, 0, st>>>(cv.dev);
  if (d.cap > 0) {
    if (dtype == 0) k_loss_entries_fwd<float><<<entry_grid(d), 256, 0, st>>>(cv.dev);
    else k_loss_entries_fwd<__half><<<entry_grid(d), 256, 0, st>>>(cv.dev);
  }
  k_loss_finalize<<<cfg->nl, 1024, 0, st>>>(cv.dev, cv.lvl_out, cv.counts + kLv + 1);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

int obb_loss_backward(const obb_loss_config* cfg, const void* const* p_levels_host, int dtype, const float* targets, int64_t nt,
                      int64_t tcols, const float* grad_scale, void* const* grad_levels_host, void* ws, size_t ws_bytes,
                      void* stream) {
  int rc = loss_check(cfg, nt);
  if (rc) return rc;
  if (!p_levels_host || !grad_levels_host || !grad_scale || (nt > 0 && !targets) || (dtype != 0 && dtype != 1)) return OBB_ERR_BAD_ARG;
  LossCarve cv;
  loss_carve(ws, cfg, nt, &cv);
  if (!ws || ws_bytes < cv.total) return OBB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  LossDev d;
  loss_fill(d, cfg, cv, targets, nt, tcols);     // entries / counts / head in `ws` are the ones obb_loss_forward left there
  d.dtype = dtype; d.gscale = grad_scale;
  for (int i = 0; i < cfg->nl; i++) {
    if (!p_levels_host[i] || !grad_levels_host[i]) return OBB_ERR_BAD_ARG;
    d.p[i] = p_levels_host[i]; d.grad[i] = grad_levels_host[i];
  }
  k_loss_setup<<<1, 256, 0, st>>>(d, cv.dev);
  dim3 gd(2048, cfg->nl);
  static const int nts = obb_dev_switch("OBB_LOSS_NT", 1) != 0 ? 1 : 0;      // 0: plain stores (development builds: measurements)
  if (dtype == 0) { if (nts) k_loss_bwd_dense<float, true><<<gd, 256, 0, st>>>(cv.dev); else k_loss_bwd_dense<float, false><<<gd, 256, 0, st>>>(cv.dev); }
  else { if (nts) k_loss_bwd_dense<__half, true><<<gd, 256, 0, st>>>(cv.dev); else k_loss_bwd_dense<__half, false><<<gd, 256, 0, st>>>(cv.dev); }
  if (d.cap > 0) {
    if (dtype == 0) k_loss_entries_bwd<float><<<entry_grid(d), 256, 0, st>>>(cv.dev);
    else k_loss_entries_bwd<__half><<<entry_grid(d), 256, 0, st>>>(cv.dev);
  }
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // extern "C"
