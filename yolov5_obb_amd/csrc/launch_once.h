// Host-side launch helper (included by nms.hip only: the device headers are also compiled for the host by tests/native/*.cpp,
// without a HIP runtime).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

namespace obb {

// hipFuncSetAttribute applies to the function ON THE CURRENT DEVICE: a process that drives several GPUs has to raise a kernel's
// dynamic-LDS limit on each of them.  One of these per kernel (a function-local static).  need() returns the index of the current
// device when the attribute still has to be set there (or kAlways when the device cannot be told: set it every time), kDone when
// it was; the caller hands that index back to mark() AFTER the attribute call succeeded.  The index lives in the caller's frame,
// not in this shared object: two host threads on different GPUs cannot mark each other's device, and two threads on
// the same GPU at worst both set the attribute, which is harmless.
struct OncePerDevice {
  static constexpr int kDone = -1, kAlways = -2;
  std::atomic<bool> seen[64] = {};
  int need() const {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return kAlways;
    return seen[dev].load(std::memory_order_acquire) ? kDone : dev;
  }
  void mark(int dev) { if (dev >= 0 && dev < 64) seen[dev].store(true, std::memory_order_release); }
};

}  // namespace obb
