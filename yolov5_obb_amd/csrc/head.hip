// Detect-head inference decode, CSL label encode and the small box conversions -- gfx950 kernels + C ABI.
//
//   obb_detect_decode   models/yolo.py:61-79 (Detect.forward, inference branch) for one level in ONE pass over the
//                       1x1-conv output: the reference runs view+permute+contiguous (1 read + 1 write of the level),
//                       sigmoid (1+1), two in-place slice updates and a torch.cat (1+1).  Here a workgroup moves a
//                       64-position x no-channel tile through LDS (coalesced 128-256 byte reads along the spatial
//                       axis, one contiguous 64*no-element write per output) and emits both results of the
//                       reference: the permuted raw head x[i] (bs,na,ny,nx,no) and the decoded rows, written straight
//                       into their slice of the concatenated (bs, sum A_i, no) prediction tensor.
//   obb_csl_encode      utils/rboxs_utils.py:9-26 gaussian_label_cpu for a batch of angles on the device.
//   obb_rbox2poly       utils/rboxs_utils.py:106-145 (+ poly2hbb :147-181) for (n,5) long-edge boxes.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>
#include "obb_hip.h"
#include "dtype_device.h"
#include "loss_math.h"

namespace obb {

struct DetectArgs {
  const void* in;      // (bs, na*no, ny, nx) conv output
  void* xperm;         // (bs, na, ny, nx, no) or null
  void* z;             // (bs, a_total, no) or null
  void* objcol;        // (bs, a_total) or null: z[..., 4] once more, densely (what the confidence filter of the NMS reads first)
  int bs, na, no, ny, nx;
  long long a_total, a_off;
  float stride;
  float anchor_px[OBB_LOSS_MAX_ANCHORS][2];   // anchors * stride (anchor_grid, models/yolo.py:90-91)
};

// y = x.sigmoid() in the tensor dtype, then (models/yolo.py:71-74, inplace branch)
//   xy = (y*2 - 0.5 + grid) * stride     y*2 and -0.5 in the tensor dtype; grid is a float32 tensor -> fp32 from there
//   wh = (y*2)**2 * anchor_grid          (y*2)**2 in the tensor dtype; anchor_grid is float32 -> fp32 product
// and the result is rounded to the tensor dtype by the slice assignment.
// sigmoid in the precision the comparison with the reference allows: fp32 tensors get the correctly rounded expf and
// division; fp16 tensors are rounded to 11 bits right after, so the hardware exp2 / reciprocal (1 ulp of fp32 each) can
// only move a result that sits within 2^-12 relative of an fp16 rounding boundary -- the same 1-ulp-of-fp16 freedom the
// reference's own libm has (tests/test_head_gpu.py states the tolerance)
template <typename T> __device__ __forceinline__ float detect_sigmoid(float x) { return sigmoid_f(x); }
template <> __device__ __forceinline__ float detect_sigmoid<__half>(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

template <typename T>
__device__ __forceinline__ float detect_decode_one(float raw, int ch, float gx, float gy, float stride, float aw, float ah) {
  const float y = round_to_dtype<T>(detect_sigmoid<T>(raw));
  if (ch >= 4) return y;
  const float t = round_to_dtype<T>(y * 2.0f);
  if (ch < 2) {
    const float u = round_to_dtype<T>(t - 0.5f);
    return round_to_dtype<T>((u + (ch == 0 ? gx : gy)) * stride);
  }
  const float q = round_to_dtype<T>(t * t);
  return round_to_dtype<T>(q * (ch == 2 ? aw : ah));
}

// positions per tile (template parameter kTileHW): 128 = 256-byte runs of every channel row (fp16) on the read side when
// the tile fits LDS comfortably, 64 otherwise

template <typename T> struct Pack16;                                   // 16 bytes of T
template <> struct Pack16<float> { static constexpr int V = 4; };
template <> struct Pack16<__half> { static constexpr int V = 8; };

// VEC: the input rows are 4-element aligned (HW % 4 == 0) and both outputs are 16-byte aligned at every tile start,
// so the tile is read with 8/16-byte loads and written with 16-byte stores; otherwise element-wise accesses.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// NTS: the tile is read and both outputs are written with non-temporal accesses (every byte is touched once)
template <typename T, bool VEC, int kTileHW, int NT, int SK, bool NTS, bool NTL>
__device__ __forceinline__ void detect_decode_tile(const DetectArgs& d, const int tile_x, const int ba /* b * na + a */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);            // [no] rows of 64 positions, skewed (tix)
  const int tid = threadIdx.x;
  const int HW = d.ny * d.nx;
  const int a = ba % d.na, b = ba / d.na;
  const int hw0 = tile_x * kTileHW;
  const int nhw = min(kTileHW, HW - hw0);
  const int no = d.no;
  const T* in = (const T*)d.in + ((size_t)ba * no) * HW + hw0;
  // LDS layout: channel row c starts at c*64 + 2*(c>>3) elements.  The store phase reads 8 consecutive channels of one
  // position per lane, i.e. lane stride = 8 rows = 512 + 2 elements: one 32-bit bank further per lane (fp16; two for fp32)
  // instead of the same eight banks again (8-way conflict with a plain 64 / 65 pitch).
  auto tix = [](int c, int hw) -> int { return c * kTileHW + (c >> 3) * SK + hw; };   // SK = 2: one bank per lane in the store phase; 4: two banks, and 4-position groups stay 8-byte aligned (one LDS store per load)

  // ---- load (coalesced along hw)
  if constexpr (VEC) {
    // kTileHW/4 threads x 4 positions cover the positions of one channel row; 256/(kTileHW/4) channel rows per step
    constexpr int TPR = kTileHW / 4, RPS = NT / TPR;
    const int q = tid % TPR, c16 = tid / TPR;
    const int hw = q * 4;
    if (nhw == kTileHW) {
      // full tile (all but the last of a plane): kLoadBatch channel rows in flight per thread before the first LDS store --
      // the loop over 200 rows otherwise waits for every load (one HBM round trip per row and thread)
      constexpr int kLoadBatch = 8;
      using Vec = typename std::conditional<sizeof(T) == 2, uint2, float4>::type;
      for (int c0 = c16; c0 < no; c0 += RPS * kLoadBatch) {
        Vec v[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; u++) {
          const int c = c0 + u * RPS;
          if (c < no) {
            if constexpr (NTL) {
              using NV = typename std::conditional<sizeof(T) == 2, u32x2_t, u32x4_t>::type;
              const NV t = __builtin_nontemporal_load(reinterpret_cast<const NV*>(in + (size_t)c * HW + hw));
              v[u] = __builtin_bit_cast(Vec, t);
            } else v[u] = *reinterpret_cast<const Vec*>(in + (size_t)c * HW + hw);
          }
        }
#pragma unroll
        for (int u = 0; u < kLoadBatch; u++) {
          const int c = c0 + u * RPS;
          if (c < no) {
            if constexpr (SK == 4) *reinterpret_cast<Vec*>(&tile[tix(c, hw)]) = v[u];
            else {
              const T* e = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
              for (int j = 0; j < 4; j++) tile[tix(c, hw + j)] = e[j];
            }
          }
        }
      }
    } else
    for (int c = c16; c < no; c += RPS) {
      if (hw + 3 < nhw) {
        T v[4];
        if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(v) = *reinterpret_cast<const uint2*>(in + (size_t)c * HW + hw);
        else *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(in + (size_t)c * HW + hw);
        if constexpr (SK == 4) {                              // aligned group: one 8- / 16-byte LDS store
          if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(&tile[tix(c, hw)]) = *reinterpret_cast<const uint2*>(v);
          else *reinterpret_cast<float4*>(&tile[tix(c, hw)]) = *reinterpret_cast<const float4*>(v);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) tile[tix(c, hw + j)] = v[j];
        }
      } else {
        for (int j = 0; j < 4; j++) if (hw + j < nhw) tile[tix(c, hw + j)] = in[(size_t)c * HW + hw + j];
      }
    }
  } else {
    const int hw = tid % kTileHW, c4 = tid / kTileHW;
    for (int c = c4; c < no; c += NT / kTileHW)
      if (hw < nhw) tile[tix(c, hw)] = in[(size_t)c * HW + hw];
  }
  __syncthreads();

  // ---- store: the tile's nhw*no output elements are contiguous in both outputs
  const int nel = nhw * no;
  if (d.objcol && tid < nhw)   // the objectness of the tile's positions, bit-identical to z[..., 4]
    st_from_float<T>((T*)d.objcol + (size_t)b * d.a_total + d.a_off + (size_t)a * HW + hw0 + tid,
                     round_to_dtype<T>(detect_sigmoid<T>(ld_as_float<T>(&tile[tix(4, tid)]))));
  T* xo = d.xperm ? (T*)d.xperm + ((size_t)ba * HW + hw0) * no : nullptr;
  T* zo = d.z ? (T*)d.z + ((size_t)b * d.a_total + d.a_off + (size_t)a * HW + hw0) * no : nullptr;
  const float aw = d.anchor_px[a][0], ah = d.anchor_px[a][1];
  auto decode_at = [&](T raw, int c, int hw) -> float {
    const float r = ld_as_float<T>(&raw);
    if (c >= 4) return round_to_dtype<T>(detect_sigmoid<T>(r));      // 196 of the 200 channels: no grid arithmetic
    const int pos = hw0 + hw;
    const int gyi = pos / d.nx;
    return detect_decode_one<T>(r, c, (float)(pos - gyi * d.nx), (float)gyi, d.stride, aw, ah);
  };
  if constexpr (VEC) {
    constexpr int V = Pack16<T>::V;
    const int nvec = nel / V;                           // full 16-byte groups; the tail (< V elements) goes element-wise
    for (int g = tid; g < nvec; g += NT) {
      const int e0 = g * V;
      int hw = e0 / no, c = e0 - hw * no;
      alignas(16) T rawv[V];
      alignas(16) T decv[V];
#pragma unroll
      for (int j = 0; j < V; j++) {
        const T raw = tile[tix(c, hw)];
        rawv[j] = raw;
        if (zo) { T o; st_from_float<T>(&o, decode_at(raw, c, hw)); decv[j] = o; }
        if (++c == no) { c = 0; hw++; }
      }
      if constexpr (NTS) {
        if (xo) __builtin_nontemporal_store(*reinterpret_cast<const u32x4_t*>(rawv), reinterpret_cast<u32x4_t*>(xo + e0));
        if (zo) __builtin_nontemporal_store(*reinterpret_cast<const u32x4_t*>(decv), reinterpret_cast<u32x4_t*>(zo + e0));
      } else {
        if (xo) *reinterpret_cast<uint4*>(xo + e0) = *reinterpret_cast<const uint4*>(rawv);
        if (zo) *reinterpret_cast<uint4*>(zo + e0) = *reinterpret_cast<const uint4*>(decv);
      }
    }
    for (int e = nvec * V + tid; e < nel; e += NT) {
      const int hw = e / no, c = e - hw * no;
      const T raw = tile[tix(c, hw)];
      if (xo) xo[e] = raw;
      if (zo) st_from_float<T>(zo + e, decode_at(raw, c, hw));
    }
  } else {
    int c = tid % no, hw = tid / no;
    const int dc = NT % no, dh = NT / no;
    for (int e = tid; e < nel; e += NT) {
      const T raw = tile[tix(c, hw)];
      if (xo) xo[e] = raw;
      if (zo) st_from_float<T>(zo + e, decode_at(raw, c, hw));
      c += dc; hw += dh;
      if (c >= no) { c -= no; hw++; }
    }
  }
}

template <typename T, bool VEC, int kTileHW, int NT, int SK, bool NTS = false, bool NTL = NTS>
__global__ __launch_bounds__(NT) void k_detect_decode(DetectArgs d) {
  detect_decode_tile<T, VEC, kTileHW, NT, SK, NTS, NTL>(d, (int)blockIdx.x, (int)blockIdx.y);
}

// All levels of the head in ONE launch: blockIdx.x runs over the tiles of level 0, then level 1, ...; blockIdx.y = b * na + a.
// Three back-to-back launches leave the chip draining and refilling twice (the 32 x 32 level alone is a 20 us launch for
// 5 % of the bytes); in one grid the small levels' tiles fill the gaps of the large one.
constexpr int kDetectMaxLevels = 4;                      // P3 .. P6
struct DetectLevels {
  DetectArgs lv[kDetectMaxLevels];
  int tile_end[kDetectMaxLevels];                        // running sum of the levels' tile counts
  int nl;
};
template <typename T, bool VEC, int kTileHW, int NT, int SK, bool NTS>
__global__ __launch_bounds__(NT) void k_detect_decode_levels(DetectLevels m) {
  const int bx = (int)blockIdx.x;
  int l = 0;
  while (l + 1 < m.nl && bx >= m.tile_end[l]) l++;
  detect_decode_tile<T, VEC, kTileHW, NT, SK, NTS, NTS>(m.lv[l], bx - (l ? m.tile_end[l - 1] : 0), (int)blockIdx.y);
}

// ------------------------------------------------------------------ CSL encode (utils/rboxs_utils.py:9-26)
// out[i][k] = y_sig[(k + index_i) mod n],  y_sig[j] = exp(-((j - n/2) - u)^2 / (2 sig^2)),  index_i = int(n/2 - label_i)
// (python slice semantics: |index| > n leaves the window unrolled).  Evaluated in double like numpy, stored as float.
__global__ void k_csl_encode(const float* __restrict__ labels, long long n, int num_class, double u, double sig,
                             float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * num_class) return;
  const long long i = e / num_class;
  const int k = (int)(e - i * num_class);
  const double half = (double)num_class / 2.0;
  const double v = half - (double)labels[i];
  long long index = (long long)v;                       // int() truncates toward zero
  long long shift = 0;
  if (index <= num_class && index >= -(long long)num_class) { shift = index % num_class; if (shift < 0) shift += num_class; }
  const int j = (int)((k + shift) % num_class);
  const double x = (double)j - half;
  out[e] = (float)exp(-((x - u) * (x - u)) / (2.0 * sig * sig));
}

// ------------------------------------------------------------------ rbox2poly / poly2hbb (utils/rboxs_utils.py:106-181)
__global__ void k_rbox2poly(const float* __restrict__ rb, long long n, int stride_in, float* __restrict__ poly8,
                            float* __restrict__ hbb4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = rb + i * stride_in;
  const float x = r[0], y = r[1], w = r[2], h = r[3], th = r[4];
  const float Cos = cosf(th), Sin = sinf(th);
  const float v1x = w / 2 * Cos, v1y = -w / 2 * Sin;      // vector1 :127
  const float v2x = -h / 2 * Sin, v2y = -h / 2 * Cos;     // vector2 :129
  float p[8];
  p[0] = x + v1x + v2x; p[1] = y + v1y + v2y;             // :131-134
  p[2] = x + v1x - v2x; p[3] = y + v1y - v2y;
  p[4] = x - v1x - v2x; p[5] = y - v1y - v2y;
  p[6] = x - v1x + v2x; p[7] = y - v1y + v2y;
  if (poly8) {
#pragma unroll
    for (int k = 0; k < 8; k++) poly8[i * 8 + k] = p[k];
  }
  if (hbb4) {                                              // :166-171
    const float xmax = fmaxf(fmaxf(p[0], p[2]), fmaxf(p[4], p[6])), xmin = fminf(fminf(p[0], p[2]), fminf(p[4], p[6]));
    const float ymax = fmaxf(fmaxf(p[1], p[3]), fmaxf(p[5], p[7])), ymin = fminf(fminf(p[1], p[3]), fminf(p[5], p[7]));
    hbb4[i * 4 + 0] = (xmax + xmin) / 2.0f; hbb4[i * 4 + 1] = (ymax + ymin) / 2.0f;
    hbb4[i * 4 + 2] = xmax - xmin; hbb4[i * 4 + 3] = ymax - ymin;
  }
}

}  // namespace obb

using namespace obb;

extern "C" {

int obb_detect_decode(const void* conv_out, int dtype, int64_t bs, int64_t na, int64_t no, int64_t ny, int64_t nx,
                      const float* anchors_px_host, float stride, void* x_perm_out, void* z_out, int64_t a_total,
                      int64_t a_offset, void* stream) {
  return obb_detect_decode_col(conv_out, dtype, bs, na, no, ny, nx, anchors_px_host, stride, x_perm_out, z_out, a_total, a_offset,
                               nullptr, stream);
}

int obb_detect_decode_col(const void* conv_out, int dtype, int64_t bs, int64_t na, int64_t no, int64_t ny, int64_t nx,
                          const float* anchors_px_host, float stride, void* x_perm_out, void* z_out, int64_t a_total,
                          int64_t a_offset, void* objcol_out, void* stream) {
  if (!conv_out || bs < 1 || na < 1 || na > OBB_LOSS_MAX_ANCHORS || no < 6 || no > 5 + 256 + 180 || ny < 1 || nx < 1 ||
      (dtype != 0 && dtype != 1) || !anchors_px_host)
    return OBB_ERR_BAD_ARG;
  if (!x_perm_out && !z_out && !objcol_out) return OBB_OK;
  if ((z_out || objcol_out) && (a_offset < 0 || a_offset + na * ny * nx > a_total)) return OBB_ERR_BAD_ARG;
  if (bs * na > 65535 || ny * nx > 0x7fffffffLL) return OBB_ERR_BAD_ARG;
  DetectArgs d;
  d.in = conv_out; d.xperm = x_perm_out; d.z = z_out; d.objcol = objcol_out;
  d.bs = (int)bs; d.na = (int)na; d.no = (int)no; d.ny = (int)ny; d.nx = (int)nx;
  d.a_total = a_total; d.a_off = a_offset; d.stride = stride;
  for (int a = 0; a < OBB_LOSS_MAX_ANCHORS; a++) {
    d.anchor_px[a][0] = a < na ? anchors_px_host[a * 2] : 0.f;
    d.anchor_px[a][1] = a < na ? anchors_px_host[a * 2 + 1] : 0.f;
  }
  const int HW = (int)(ny * nx);
  const size_t esz = dtype == 0 ? 4 : 2;
  // 64-position tiles, 256 threads, 2-element skew, eight channel rows in flight per thread, non-temporal loads and stores
  // (every byte is touched once; a streaming store does not push the lines other workgroups are about to read out of L2).
  // Measured against it in round 3 on (16, 3*200, 128..32, ..) and removed: a 4-element skew with one LDS store per 4-position
  // group (fp32 0.80 against 0.48 ms: the 16-byte LDS stores conflict), 128-position tiles with 512 threads (fp16 0.277
  // against 0.266, fp32 0.95), non-temporal stores only (0.268-0.280) / loads only (0.281).
  const int tile_hw = 64;
  const int skew = 2;
  const size_t lds = ((size_t)no * tile_hw + (size_t)skew * (size_t)(no / 8 + 2)) * esz;   // skewed rows (tix in the kernel)
  dim3 grid((unsigned)((HW + tile_hw - 1) / tile_hw), (unsigned)(bs * na));
  hipStream_t st = (hipStream_t)stream;
  // vector path: input rows 4-element aligned, every tile of both outputs 16-byte aligned
  auto al16 = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
  bool vec = (HW % 4 == 0) && al16(conv_out) && al16(x_perm_out) && al16(z_out);
  vec = vec && ((size_t)HW * no * esz) % 16 == 0 && ((size_t)a_total * no * esz) % 16 == 0 && ((size_t)a_offset * no * esz) % 16 == 0;
  if (lds > 150 * 1024) return OBB_ERR_BAD_ARG;
#define OBB_LAUNCH_DETECT(T, VEC, TILE, NTH, SKW, ...)                                                                             \
  do {                                                                                                                              \
    if (lds > 48 * 1024 &&                                                                                                          \
        hipFuncSetAttribute((const void*)k_detect_decode<T, VEC, TILE, NTH, SKW, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return OBB_ERR_LAUNCH;                                                                                                        \
    k_detect_decode<T, VEC, TILE, NTH, SKW, ##__VA_ARGS__><<<grid, NTH, lds, st>>>(d);                                              \
  } while (0)
#define OBB_LAUNCH_DETECT_T(T)                                                                                                      \
  do {                                                                                                                              \
    if (vec) { OBB_LAUNCH_DETECT(T, true, 64, 256, 2, true); } else { OBB_LAUNCH_DETECT(T, false, 64, 256, 2); }                    \
  } while (0)
  if (dtype == 0) OBB_LAUNCH_DETECT_T(float); else OBB_LAUNCH_DETECT_T(__half);
#undef OBB_LAUNCH_DETECT_T
#undef OBB_LAUNCH_DETECT
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

int obb_detect_decode_levels(int nl, const void* const* conv_out, int dtype, int64_t bs, int64_t na, int64_t no, const int64_t* ny,
                             const int64_t* nx, const float* anchors_px_host, const float* strides_host, void* const* x_perm_out,
                             void* z_out, int64_t a_total, void* objcol_out, void* stream) {
  if (nl < 1 || nl > kDetectMaxLevels || !conv_out || !ny || !nx || !anchors_px_host || !strides_host || bs < 1 || na < 1 ||
      na > OBB_LOSS_MAX_ANCHORS || no < 6 || no > 5 + 256 + 180 || (dtype != 0 && dtype != 1))
    return OBB_ERR_BAD_ARG;
  if (bs * na > 65535) return OBB_ERR_BAD_ARG;
  const size_t esz = dtype == 0 ? 4 : 2;
  constexpr int tile_hw = 64, skew = 2;
  auto al16 = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
  DetectLevels m;
  m.nl = nl;
  int64_t off = 0, tiles = 0;
  bool vec = al16(z_out) && ((size_t)a_total * no * esz) % 16 == 0;
  bool any_out = z_out != nullptr || objcol_out != nullptr;
  for (int l = 0; l < kDetectMaxLevels; l++) {
    DetectArgs& d = m.lv[l];
    if (l >= nl) { d = m.lv[0]; m.tile_end[l] = (int)tiles; continue; }
    if (!conv_out[l] || ny[l] < 1 || nx[l] < 1 || ny[l] * nx[l] > 0x7fffffffLL) return OBB_ERR_BAD_ARG;
    void* xp = x_perm_out ? x_perm_out[l] : nullptr;
    any_out = any_out || xp != nullptr;
    d.in = conv_out[l]; d.xperm = xp; d.z = z_out; d.objcol = objcol_out;
    d.bs = (int)bs; d.na = (int)na; d.no = (int)no; d.ny = (int)ny[l]; d.nx = (int)nx[l];
    d.a_total = a_total; d.a_off = off; d.stride = strides_host[l];
    for (int a = 0; a < OBB_LOSS_MAX_ANCHORS; a++) {
      d.anchor_px[a][0] = a < na ? anchors_px_host[((size_t)l * na + a) * 2] : 0.f;
      d.anchor_px[a][1] = a < na ? anchors_px_host[((size_t)l * na + a) * 2 + 1] : 0.f;
    }
    const int64_t HW = ny[l] * nx[l];
    vec = vec && (HW % 4 == 0) && al16(conv_out[l]) && al16(xp) && ((size_t)HW * no * esz) % 16 == 0 && ((size_t)off * no * esz) % 16 == 0;
    tiles += (HW + tile_hw - 1) / tile_hw;
    if (tiles > 0x7fffffffLL) return OBB_ERR_BAD_ARG;
    m.tile_end[l] = (int)tiles;
    off += na * HW;
  }
  if ((z_out || objcol_out) && off > a_total) return OBB_ERR_BAD_ARG;
  if (!any_out) return OBB_OK;
  const size_t lds = ((size_t)no * tile_hw + (size_t)skew * (size_t)(no / 8 + 2)) * esz;
  if (lds > 150 * 1024) return OBB_ERR_BAD_ARG;
  dim3 grid((unsigned)tiles, (unsigned)(bs * na));
  hipStream_t st = (hipStream_t)stream;
#define OBB_LAUNCH_LEVELS(T, VEC, NTS)                                                                                              \
  do {                                                                                                                              \
    if (lds > 48 * 1024 &&                                                                                                          \
        hipFuncSetAttribute((const void*)k_detect_decode_levels<T, VEC, tile_hw, 256, skew, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return OBB_ERR_LAUNCH;                                                                                                        \
    k_detect_decode_levels<T, VEC, tile_hw, 256, skew, NTS><<<grid, 256, lds, st>>>(m);                                             \
  } while (0)
  if (dtype == 0) { if (vec) OBB_LAUNCH_LEVELS(float, true, true); else OBB_LAUNCH_LEVELS(float, false, false); }
  else { if (vec) OBB_LAUNCH_LEVELS(__half, true, true); else OBB_LAUNCH_LEVELS(__half, false, false); }
#undef OBB_LAUNCH_LEVELS
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

int obb_csl_encode_f32(const float* labels, int64_t n, int num_class, double u, double sig, float* out, void* stream) {
  if (n < 0 || num_class < 1 || !(sig > 0.0)) return OBB_ERR_BAD_ARG;
  if (n == 0) return OBB_OK;
  if (!labels || !out) return OBB_ERR_BAD_ARG;
  const long long tot = (long long)n * num_class;
  k_csl_encode<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(labels, n, num_class, u, sig, out);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

int obb_rbox2poly_f32(const float* rboxes, int64_t n, int64_t row_stride, float* poly8, float* hbb4, void* stream) {
  if (n < 0 || row_stride < 5) return OBB_ERR_BAD_ARG;
  if (n == 0) return OBB_OK;
  if (!rboxes || (!poly8 && !hbb4)) return OBB_ERR_BAD_ARG;
  k_rbox2poly<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(rboxes, n, (int)row_stride, poly8, hbb4);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // extern "C"

// ====================================================================== post-NMS tail of val.py
namespace obb {

// val.py:226-236 for one image's detections (n,7) [x y l s theta conf cls]:
//   pred_poly  (n,10) = [rbox2poly, conf, cls]                       model-input space
//   pred_hbb   (n,6)  = [xywh2xyxy(poly2hbb(poly)), conf, cls]
//   pred_polyn (n,10) = scale_polys(pred_poly): (x - pad_x) / gain, (y - pad_y) / gain     native image space
//   pred_hbbn  (n,6)  = [xywh2xyxy(poly2hbb(polyn)), conf, cls]
// (utils/rboxs_utils.py:106-181, utils/general.py:590-597 xywh2xyxy, :636-650 scale_polys)
// corners of one rbox (utils/rboxs_utils.py:106-145) and the horizontal box around 8 coordinates (poly2hbb + xywh2xyxy)
__device__ __forceinline__ void vt_rbox2poly(float x, float y, float w, float h, float th, float* p) {
  const float Cos = cosf(th), Sin = sinf(th);
  const float v1x = w / 2 * Cos, v1y = -w / 2 * Sin, v2x = -h / 2 * Sin, v2y = -h / 2 * Cos;
  p[0] = x + v1x + v2x; p[1] = y + v1y + v2y; p[2] = x + v1x - v2x; p[3] = y + v1y - v2y;
  p[4] = x - v1x - v2x; p[5] = y - v1y - v2y; p[6] = x - v1x + v2x; p[7] = y - v1y + v2y;
}
__device__ __forceinline__ void vt_hbb_xyxy(const float* q, float* o) {
  const float xmax = fmaxf(fmaxf(q[0], q[2]), fmaxf(q[4], q[6])), xmin = fminf(fminf(q[0], q[2]), fminf(q[4], q[6]));
  const float ymax = fmaxf(fmaxf(q[1], q[3]), fmaxf(q[5], q[7])), ymin = fminf(fminf(q[1], q[3]), fminf(q[5], q[7]));
  const float xc = (xmax + xmin) / 2.0f, yc = (ymax + ymin) / 2.0f, bw = xmax - xmin, bh = ymax - ymin;   // poly2hbb
  o[0] = xc - bw / 2; o[1] = yc - bh / 2; o[2] = xc + bw / 2; o[3] = yc + bh / 2;                          // xywh2xyxy
}
// the four outputs of val.py:226-236 for detection row i (any output pointer may be NULL); returns pred_hbbn's box in o4n
__device__ __forceinline__ void vt_post_one(const float* __restrict__ r, long long i, float pad_x, float pad_y, float gain,
                                            float* __restrict__ poly10, float* __restrict__ hbb6, float* __restrict__ polyn10,
                                            float* __restrict__ hbbn6, float* o4n) {
  const float conf = r[5], cls = r[6];
  float p[8], o4[4];
  vt_rbox2poly(r[0], r[1], r[2], r[3], r[4], p);
  if (poly10) { for (int k = 0; k < 8; k++) poly10[i * 10 + k] = p[k]; poly10[i * 10 + 8] = conf; poly10[i * 10 + 9] = cls; }
  if (hbb6) { vt_hbb_xyxy(p, o4); for (int k = 0; k < 4; k++) hbb6[i * 6 + k] = o4[k]; hbb6[i * 6 + 4] = conf; hbb6[i * 6 + 5] = cls; }
  float pn[8];
#pragma unroll
  for (int k = 0; k < 8; k++) pn[k] = (p[k] - ((k & 1) ? pad_y : pad_x)) / gain;
  if (polyn10) { for (int k = 0; k < 8; k++) polyn10[i * 10 + k] = pn[k]; polyn10[i * 10 + 8] = conf; polyn10[i * 10 + 9] = cls; }
  vt_hbb_xyxy(pn, o4n);
  if (hbbn6) { for (int k = 0; k < 4; k++) hbbn6[i * 6 + k] = o4n[k]; hbbn6[i * 6 + 4] = conf; hbbn6[i * 6 + 5] = cls; }
}
__global__ void k_val_post(const float* __restrict__ det7, long long n, float pad_x, float pad_y, float gain,
                           float* __restrict__ poly10, float* __restrict__ hbb6, float* __restrict__ polyn10,
                           float* __restrict__ hbbn6) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o4n[4];
  vt_post_one(det7 + i * 7, i, pad_x, pad_y, gain, poly10, hbb6, polyn10, hbbn6, o4n);
}

// val.py:69-90 process_batch.  The reference keeps, per detection, its highest-IoU label among those with the same class and
// IoU >= iouv[0]; then, per label, the match with the LOWEST detection index (np.unique after the first de-duplication
// has re-ordered the matches by detection index; the re-sort by IoU is commented out at val.py:86).
__global__ void k_pb_best(const float* __restrict__ det6, int n, const float* __restrict__ lab5, int m, const float* __restrict__ iouv,
                          int* __restrict__ best_label, float* __restrict__ best_iou, int* __restrict__ winner) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  const float thr0 = iouv[0];                                            // val.py:81  iou >= iouv[0]
  const float* b2 = det6 + (size_t)d * 6;
  const float area2 = (b2[2] - b2[0]) * (b2[3] - b2[1]);
  int bl = -1; float bi = -1.f;
  for (int l = 0; l < m; l++) {
    const float* b1 = lab5 + (size_t)l * 5 + 1;
    if (lab5[(size_t)l * 5] != b2[5]) continue;
    const float area1 = (b1[2] - b1[0]) * (b1[3] - b1[1]);
    const float iw = fmaxf(fminf(b1[2], b2[2]) - fmaxf(b1[0], b2[0]), 0.f);
    const float ih = fmaxf(fminf(b1[3], b2[3]) - fmaxf(b1[1], b2[1]), 0.f);
    const float inter = iw * ih;
    const float iou = inter / (area1 + area2 - inter);                 // utils/metrics.py:265-268
    if (iou >= thr0 && iou > bi) { bi = iou; bl = l; }
  }
  best_label[d] = bl; best_iou[d] = bi;
  if (bl >= 0) atomicMin(&winner[bl], d);
}
__global__ void k_pb_correct(const int* __restrict__ best_label, const float* __restrict__ best_iou, const int* __restrict__ winner,
                             const float* __restrict__ iouv, int n, int niou, uint8_t* __restrict__ correct) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  const int bl = best_label[d];
  const bool win = bl >= 0 && winner[bl] == d;
  for (int k = 0; k < niou; k++) correct[(size_t)d * niou + k] = (win && best_iou[d] >= iouv[k]) ? 1 : 0;
}


// ---- the tail for ALL images of a batch (val.py:209-250): two launches (round 4: three) and one device -> host copy per batch instead of
// two launches + a memset + three copies per image.
constexpr int kValTailMaxBs = 64;
struct ValTailImgs {                       // by value in the kernel arguments (bs <= kValTailMaxBs)
  int det_off[kValTailMaxBs + 1];          // detections of image b: rows [det_off[b], det_off[b + 1]) of the packed list (and of every output)
  int det_row[kValTailMaxBs];              // row of det7 that holds the FIRST detection of image b (packed input: det_off[b])
  float pad_x[kValTailMaxBs], pad_y[kValTailMaxBs], gain[kValTailMaxBs], shape_w[kValTailMaxBs], shape_h[kValTailMaxBs];
  int bs;
};
// a label's box (val.py:238-241, in the reference's operation order): rbox2poly -> poly2hbb -> xywh2xyxy in the letterboxed frame,
// THEN scale_coords: subtract the pad, divide by the gain, clip to the native shape (utils/general.py:621-633) -> [x1 y1 x2 y2]
__device__ __forceinline__ void vt_label_box(const float* t, const ValTailImgs& im, int b, float* o) {
  float p[8];
  vt_rbox2poly(t[2], t[3], t[4], t[5], t[6], p);                // t = [img cls cx cy l s theta ...]
  vt_hbb_xyxy(p, o);
  o[0] -= im.pad_x[b]; o[2] -= im.pad_x[b]; o[1] -= im.pad_y[b]; o[3] -= im.pad_y[b];
#pragma unroll
  for (int k = 0; k < 4; k++) o[k] /= im.gain[b];
  o[0] = fminf(fmaxf(o[0], 0.f), im.shape_w[b]); o[2] = fminf(fmaxf(o[2], 0.f), im.shape_w[b]);
  o[1] = fminf(fmaxf(o[1], 0.f), im.shape_h[b]); o[3] = fminf(fmaxf(o[3], 0.f), im.shape_h[b]);
}
// Round 5: two launches (round 4: three -- a label kernel in front, and an atomicMin "winner" table that needed it for its
// initialisation).  The label boxes are computed where they are used (a detection meets one or two labels of its image and class),
// and the winner of a label -- the lowest-indexed detection of the image whose best label it is (process_batch's first np.unique
// pass, val.py:84-86) -- is found by the stats kernel with a scan over the earlier detections of the image.
// per detection: the four outputs of val.py:226-236 and its best label (process_batch, see k_pb_best) among the labels of ITS image.
// The label boxes of the one or two images a workgroup's 128 detections belong to are computed once into LDS (a detection
// computing the box of every label it meets -- double-precision sin / cos on divergent lanes -- made the kernel slower than the
// launch it saved); more than kVtLabLds labels in range: computed where they are met.
constexpr int kVtLabLds = 512;
__global__ void k_vt_dets(const float* __restrict__ det7, int n, ValTailImgs im, const float* __restrict__ targets, int nt, int tcols,
                          const float* __restrict__ iouv, float* __restrict__ poly10, float* __restrict__ hbb6, float* __restrict__ polyn10,
                          float* __restrict__ hbbn6, int* __restrict__ best_label, float* __restrict__ best_iou, int* __restrict__ counter) {
  __shared__ float s_lab[kVtLabLds][4];
  __shared__ float s_lc[kVtLabLds];
  __shared__ int s_li[kVtLabLds], s_lb[kVtLabLds];
  __shared__ int s_nl;
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter = 0;       // k_vt_stats' arrival counter (that kernel runs behind this one)
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const int d_first = blockIdx.x * blockDim.x, d_last = (d_first + (int)blockDim.x < n ? d_first + (int)blockDim.x : n) - 1;
  int b_lo = 0, b_hi = 0;
  while (b_lo + 1 < im.bs && d_first >= im.det_off[b_lo + 1]) b_lo++;   // (bs <= 64: short scans of kernel-argument registers)
  b_hi = b_lo;
  while (b_hi + 1 < im.bs && d_last >= im.det_off[b_hi + 1]) b_hi++;
  if (threadIdx.x == 0) s_nl = 0;
  __syncthreads();
  for (int l = threadIdx.x; l < nt; l += blockDim.x) {
    const float* t = targets + (size_t)l * tcols;
    const int lb = (int)t[0];
    if (lb >= b_lo && lb <= b_hi) {
      const int k = atomicAdd(&s_nl, 1);
      if (k < kVtLabLds) { s_li[k] = l; s_lb[k] = lb; s_lc[k] = t[1]; vt_label_box(t, im, lb, s_lab[k]); }
    }
  }
  __syncthreads();
  const int nl = s_nl;
  if (d >= n) return;
  int b = b_lo;
  while (b + 1 < im.bs && d >= im.det_off[b + 1]) b++;
  float b2[4];
  const float* row = det7 + ((size_t)im.det_row[b] + (d - im.det_off[b])) * 7;
  vt_post_one(row, d, im.pad_x[b], im.pad_y[b], im.gain[b], poly10, hbb6, polyn10, hbbn6, b2);
  const float cls = row[6];
  const float thr0 = iouv[0];                                            // val.py:81  iou >= iouv[0]
  const float area2 = (b2[2] - b2[0]) * (b2[3] - b2[1]);
  int bl = -1; float bi = -1.f;
  auto consider = [&](const float* b1, int l) {
    const float area1 = (b1[2] - b1[0]) * (b1[3] - b1[1]);
    const float iw = fmaxf(fminf(b1[2], b2[2]) - fmaxf(b1[0], b2[0]), 0.f);
    const float ih = fmaxf(fminf(b1[3], b2[3]) - fmaxf(b1[1], b2[1]), 0.f);
    const float inter = iw * ih;
    const float iou = inter / (area1 + area2 - inter);                 // utils/metrics.py:265-268
    // the best label = the first maximum in label order (the staged list is in arrival order: the tie goes to the lower index)
    if (iou >= thr0 && (iou > bi || (iou == bi && l < bl))) { bi = iou; bl = l; }
  };
  if (nl <= kVtLabLds) {
    for (int k = 0; k < nl; k++)
      if (s_lb[k] == b && s_lc[k] == cls) consider(s_lab[k], s_li[k]);
  } else {
    for (int l = 0; l < nt; l++) {
      const float* t = targets + (size_t)l * tcols;
      if ((int)t[0] != b || t[1] != cls) continue;
      float b1[4];
      vt_label_box(t, im, b, b1);
      consider(b1, l);
    }
  }
  best_label[d] = bl; best_iou[d] = bi;
}
// stats row of detection d: correct[0 .. niou) as 0 / 1, then conf, then cls (val.py:250's tuple, one copy for the batch)
// `done` (optional, pinned host memory like `stats` may be): receives n once EVERY row has landed -- each workgroup makes its
// rows visible system-wide (fence), then arrives on a device counter; the last one stores the flag.  The host polls it instead of
// copying the rows back and waiting for the stream.
__global__ void k_vt_stats(const float* __restrict__ det7, const int* __restrict__ best_label, const float* __restrict__ best_iou,
                           ValTailImgs im, const float* __restrict__ iouv, int n, int niou, float* __restrict__ stats,
                           int* __restrict__ counter, long long* __restrict__ done) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  // the winner of a label = the lowest-indexed detection of the image that chose it (detection indices grow inside the image).  A
  // matched detection looks for an earlier one with the same label; the wave scans that prefix TOGETHER, 64 entries per trip (a
  // lane on its own walked up to a few hundred dependent loads: the first version of this kernel took longer than the launch it saved)
  int bl = -1, off_b = 0, row_b = 0;
  if (d < n) {
    bl = best_label[d];
    int b = 0;
    while (b + 1 < im.bs && d >= im.det_off[b + 1]) b++;
    off_b = im.det_off[b]; row_b = im.det_row[b];
  }
  bool win = bl >= 0;
  for (unsigned long long todo = __ballot(win); todo; todo &= todo - 1) {
    const int l0 = __builtin_ctzll(todo);
    const int bl0 = __shfl(bl, l0), d0 = __shfl(d, l0), o0 = __shfl(off_b, l0);
    bool found = false;
    for (int e0 = o0; e0 < d0 && !found; e0 += 64) {
      const int e = e0 + lane;
      found = __ballot(e < d0 && best_label[e] == bl0) != 0ull;
    }
    if (found && lane == l0) win = false;
  }
  if (d < n) {
    float* o = stats + (size_t)d * (niou + 2);
    for (int k = 0; k < niou; k++) o[k] = (win && best_iou[d] >= iouv[k]) ? 1.f : 0.f;
    const float* row = det7 + ((size_t)row_b + (d - off_b)) * 7;
    o[niou] = row[5]; o[niou + 1] = row[6];
  }
  if (done != nullptr) {                                          // (kernel-uniform)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(counter, 1) == (int)gridDim.x - 1)
      __hip_atomic_store(done, (long long)n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace obb

extern "C" {

int obb_val_postprocess_f32(const float* det7, int64_t n, float pad_x, float pad_y, float gain, float* poly10, float* hbb6,
                            float* polyn10, float* hbbn6, void* stream) {
  if (n < 0 || !(gain > 0.f)) return OBB_ERR_BAD_ARG;
  if (n == 0) return OBB_OK;
  if (!det7) return OBB_ERR_BAD_ARG;
  obb::k_val_post<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(det7, n, pad_x, pad_y, gain, poly10, hbb6, polyn10, hbbn6);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

size_t obb_val_tail_batch_workspace_bytes(int64_t n_det, int64_t nt) {
  return (size_t)(n_det > 0 ? n_det : 1) * 8 + (size_t)(nt > 0 ? nt : 1) * 20 + 512;
}

static int val_tail_batch_impl(const float* det7, const int64_t* det_row_host, const int64_t* det_off_host, int64_t bs, const float* targets, int64_t nt, int64_t tcols,
                               const float* img5_host, const float* iouv, int niou, float* poly10, float* hbb6, float* polyn10,
                               float* hbbn6, float* stats, void* ws, size_t ws_bytes, void* stream, int64_t* done) {
  if (bs < 1 || bs > obb::kValTailMaxBs || nt < 0 || niou < 1 || !det_off_host || !img5_host || !iouv) return OBB_ERR_BAD_ARG;
  if (nt > 0 && (!targets || tcols < 7)) return OBB_ERR_BAD_ARG;
  const int64_t n = det_off_host[bs];
  if (n < 0 || n > 0x7fffffff || nt > 0x7fffffff || det_off_host[0] != 0) return OBB_ERR_BAD_ARG;
  obb::ValTailImgs im;
  im.bs = (int)bs;
  for (int b = 0; b <= (int)bs; b++) {
    if (b > 0 && det_off_host[b] < det_off_host[b - 1]) return OBB_ERR_BAD_ARG;
    im.det_off[b] = (int)det_off_host[b];
  }
  for (int b = 0; b < (int)bs; b++) {
    const int64_t r = det_row_host ? det_row_host[b] : det_off_host[b];
    if (r < 0 || r > 0x7fffffff - (det_off_host[b + 1] - det_off_host[b])) return OBB_ERR_BAD_ARG;
    im.det_row[b] = (int)r;
  }
  for (int b = 0; b < (int)bs; b++) {
    const float* q = img5_host + (size_t)b * 5;                 // pad_x, pad_y, gain, native width, native height
    if (!(q[2] > 0.f)) return OBB_ERR_BAD_ARG;
    im.pad_x[b] = q[0]; im.pad_y[b] = q[1]; im.gain[b] = q[2]; im.shape_w[b] = q[3]; im.shape_h[b] = q[4];
  }
  if (n == 0) { if (done) *done = 0; return OBB_OK; }            // (host-visible memory: nothing to wait for)
  if (!det7 || !stats) return OBB_ERR_BAD_ARG;
  if (!ws || ws_bytes < obb_val_tail_batch_workspace_bytes(n, nt)) return OBB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* best_label = (int*)ws;
  float* best_iou = (float*)(best_label + n);
  int* counter = (int*)(((uintptr_t)(best_iou + n) + 255) & ~(uintptr_t)255);              // (inside the 512 spare bytes)
  obb::k_vt_dets<<<(unsigned)((n + 127) / 128), 128, 0, st>>>(det7, (int)n, im, targets, (int)nt, (int)tcols, iouv, poly10, hbb6, polyn10, hbbn6,
                                                            best_label, best_iou, counter);
  obb::k_vt_stats<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(det7, best_label, best_iou, im, iouv, (int)n, niou, stats, counter,
                                                              reinterpret_cast<long long*>(done));
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

int obb_val_tail_batch_f32(const float* det7, const int64_t* det_off_host, int64_t bs, const float* targets, int64_t nt, int64_t tcols,
                           const float* img5_host, const float* iouv, int niou, float* poly10, float* hbb6, float* polyn10,
                           float* hbbn6, float* stats, void* ws, size_t ws_bytes, void* stream) {
  return val_tail_batch_impl(det7, nullptr, det_off_host, bs, targets, nt, tcols, img5_host, iouv, niou, poly10, hbb6, polyn10, hbbn6, stats, ws,
                             ws_bytes, stream, nullptr);
}

int obb_val_tail_batch_polled_f32(const float* det7, const int64_t* det_off_host, int64_t bs, const float* targets, int64_t nt,
                                  int64_t tcols, const float* img5_host, const float* iouv, int niou, float* poly10, float* hbb6,
                                  float* polyn10, float* hbbn6, float* stats, void* ws, size_t ws_bytes, void* stream, int64_t* done) {
  if (!done) return OBB_ERR_BAD_ARG;
  return val_tail_batch_impl(det7, nullptr, det_off_host, bs, targets, nt, tcols, img5_host, iouv, niou, poly10, hbb6, polyn10, hbbn6, stats, ws,
                             ws_bytes, stream, done);
}

int obb_val_tail_batch_rows_f32(const float* det7, const int64_t* det_row_host, const int64_t* det_off_host, int64_t bs, const float* targets,
                                int64_t nt, int64_t tcols, const float* img5_host, const float* iouv, int niou, float* poly10, float* hbb6,
                                float* polyn10, float* hbbn6, float* stats, void* ws, size_t ws_bytes, void* stream, int64_t* done) {
  if (!det_row_host) return OBB_ERR_BAD_ARG;
  return val_tail_batch_impl(det7, det_row_host, det_off_host, bs, targets, nt, tcols, img5_host, iouv, niou, poly10, hbb6, polyn10, hbbn6, stats,
                             ws, ws_bytes, stream, done);
}

size_t obb_process_batch_workspace_bytes(int64_t n, int64_t m) { return (size_t)(n > 0 ? n : 1) * 8 + (size_t)(m > 0 ? m : 1) * 4 + 256; }

int obb_process_batch_f32(const float* det6, int64_t n, const float* lab5, int64_t m, const float* iouv, int niou, uint8_t* correct,
                          void* ws, size_t ws_bytes, void* stream) {
  if (n < 0 || m < 0 || niou < 1 || n > 0x7fffffff || m > 0x7fffffff) return OBB_ERR_BAD_ARG;
  if (n == 0) return OBB_OK;
  if (!det6 || !iouv || !correct || (m > 0 && !lab5)) return OBB_ERR_BAD_ARG;
  if (!ws || ws_bytes < obb_process_batch_workspace_bytes(n, m)) return OBB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* best_label = (int*)ws;
  float* best_iou = (float*)(best_label + n);
  int* winner = (int*)(best_iou + n);
  if (hipMemsetAsync(winner, 0x7f, (size_t)(m > 0 ? m : 1) * 4, st) != hipSuccess) return OBB_ERR_LAUNCH;
  obb::k_pb_best<<<(unsigned)((n + 127) / 128), 128, 0, st>>>(det6, (int)n, lab5, (int)m, iouv, best_label, best_iou, winner);
  obb::k_pb_correct<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(best_label, best_iou, winner, iouv, (int)n, niou, correct);
  return hipGetLastError() == hipSuccess ? OBB_OK : OBB_ERR_LAUNCH;
}

}  // extern "C"
