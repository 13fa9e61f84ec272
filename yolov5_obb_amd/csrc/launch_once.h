// Host-side launch helper (included by nms.hip only: the device headers are also compiled for the host by tests/native/*.cpp,
// without a HIP runtime).
#pragma once
#include <hip/hip_runtime.h>

namespace obb {

// hipFuncSetAttribute applies to the function ON THE CURRENT DEVICE: a process that drives several GPUs has to raise a kernel's
// dynamic-LDS limit on each of them.  One of these per kernel (a function-local static): true until the current device has been seen
// by mark().  (Two threads may both find it unseen: the attribute is then set twice, which is harmless.)
struct OncePerDevice {
  bool seen[64] = {};
  int dev = -1;
  bool need() {
    dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev = -1; return true; }
    return !seen[dev];
  }
  void mark() { if (dev >= 0) seen[dev] = true; }
};

}  // namespace obb
