// Microbenchmark (development aid): how fast can one 2-byte element of every 400-byte row be read on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ub_stride tools/ub/stride.hip && gpurun_out/ub_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int G>
__global__ __launch_bounds__(256) void k_scan(const uint16_t* __restrict__ p, long long rows, int no, int off, unsigned long long* out, unsigned long long magic) {
  const long long r0 = ((long long)blockIdx.x * 256 + threadIdx.x);
  const long long step = (long long)gridDim.x * 256;
  uint16_t v[G];
#pragma unroll
  for (int q = 0; q < G; q++) { const long long r = r0 + q * step; v[q] = r < rows ? p[r * no + off] : 0; }
  unsigned long long acc = 0;
#pragma unroll
  for (int q = 0; q < G; q++) acc += __popcll(__ballot(v[q] > 0x3c00));
  if (acc == magic) out[0] = acc;
}
// same rows, consecutive rows per lane group (row r0*G + q): a wave covers 64*G consecutive rows
template <int G>
__global__ __launch_bounds__(256) void k_scan_blk(const uint16_t* __restrict__ p, long long rows, int no, int off, unsigned long long* out, unsigned long long magic) {
  const long long w = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  uint16_t v[G];
#pragma unroll
  for (int q = 0; q < G; q++) { const long long r = (w * G + q) * 64 + lane; v[q] = r < rows ? p[r * no + off] : 0; }
  unsigned long long acc = 0;
#pragma unroll
  for (int q = 0; q < G; q++) acc += __popcll(__ballot(v[q] > 0x3c00));
  if (acc == magic) out[0] = acc;
}
__global__ void k_fill(uint16_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint16_t)(i * 2654435761u >> 20);
}
__global__ void k_trash(float4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1, 2, 3, 4);
}

int main() {
  const long long rows = 16LL * 64512; const int no = 200;
  uint16_t* p; unsigned long long* out; float4* trash; const size_t tn = (512u << 20) / 16;
  CK(hipMalloc(&p, rows * no * 2)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&trash, tn * 16));
  k_fill<<<2048, 256>>>(p, (size_t)rows * no);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, bool cold) {
    float best = 1e9, sum = 0; const int it = 10;
    for (int i = 0; i < it + 2; i++) {
      if (cold) k_trash<<<2048, 256>>>(trash, tn);
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-34s %s  avg %.1f us  min %.1f us  (%.2f TB/s of 128-byte lines)\n", name, cold ? "cold" : "warm", sum / it * 1e3, best * 1e3, rows * 128.0 / (sum / it * 1e-3) / 1e12);
  };
  for (int cold = 0; cold < 2; cold++) {
#define RUN(G) run("interleaved G=" #G, [&]() { const int wg = (int)((rows + 256LL * G - 1) / (256LL * G)); k_scan<G><<<wg, 256>>>(p, rows, no, 4, out, 12345ull); }, cold); \
               run("blocked     G=" #G, [&]() { const int wg = (int)((rows + 256LL * G - 1) / (256LL * G)); k_scan_blk<G><<<wg, 256>>>(p, rows, no, 4, out, 12345ull); }, cold);
    RUN(1) RUN(2) RUN(4) RUN(8) RUN(16)
  }
  CK(hipDeviceSynchronize());
  return 0;
}
