// Common device-side helpers for the oriented-box kernels (gfx950, wave64).
//
// The geometry functions in riou_device.h / piou_device.h are plain C++ so that
// tests/host_check.cpp can compile them with g++ and compare them bit-for-bit
// with the CPU oracle before any GPU time is spent; kernels (*.hip) add the
// wave-level machinery around them.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OBB_DEV __device__ __forceinline__
#define OBB_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define OBB_DEV inline
#define OBB_HD inline
#endif

#include <stdlib.h>

namespace obb {

constexpr int kWave = 64;  // gfx950 wavefront width (hard-coded on purpose)

// Measurement switches: environment variables that select an A/B code path exist only in a development build
// (`make DEV=1` = -DOBB_DEV_SWITCHES); the production library ignores them, so a stray variable cannot change what runs.
// Callers keep the value in a function-local `static const` (initialised once, thread-safe).  The two switches that stay
// in every build are documented in include/obb_hip.h: OBB_NMS_POLY_STRICT (tests) and OBB_NMS_PHASE_PROF (in-kernel timers).
inline int obb_dev_switch(const char* name, int dflt) {
#ifdef OBB_DEV_SWITCHES
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// ---- float thresholds equivalent to the reference's double-literal compares ----
// The reference compares float values with double literals (1e-14, 1e-6, 1e-8):
// the float is widened first.  For a float x and a double d
//     (double)x <  d   <=>  x <  f32_ceil(d)      (smallest float >= d)
//     (double)x <= d   <=>  x <= f32_floor(d)     (largest  float <= d)
//     (double)x >  d   <=>  x >  f32_floor(d)
// which lets the kernels stay in fp32 while deciding exactly like the reference.
constexpr float f32_next_up(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if (f == 0.0f) return __builtin_bit_cast(float, (uint32_t)1);
  u = (f > 0.0f) ? u + 1 : u - 1;
  return __builtin_bit_cast(float, u);
}
constexpr float f32_next_down(float f) { return -f32_next_up(-f); }
constexpr float f32_ceil(double d) {
  float f = (float)d;
  return ((double)f >= d) ? f : f32_next_up(f);
}
constexpr float f32_floor(double d) {
  float f = (float)d;
  return ((double)f <= d) ? f : f32_next_down(f);
}

static_assert((double)f32_ceil(1e-6) >= 1e-6 && (double)f32_next_down(f32_ceil(1e-6)) < 1e-6, "f32_ceil");
static_assert((double)f32_floor(1e-14) <= 1e-14 && (double)f32_next_up(f32_floor(1e-14)) > 1e-14, "f32_floor");

}  // namespace obb
