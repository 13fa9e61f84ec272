// Element access in the tensor's own dtype (fp32 / fp16) for the kernels that read the Detect head (HIP only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

namespace obb {

template <typename T> __device__ __forceinline__ float ld_as_float(const T* p);
template <> __device__ __forceinline__ float ld_as_float<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_as_float<__half>(const __half* p) { return __half2float(*p); }

template <typename T> __device__ __forceinline__ void st_from_float(T* p, float v);
template <> __device__ __forceinline__ void st_from_float<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_from_float<__half>(__half* p, float v) { *p = __float2half_rn(v); }

// value rounded to the tensor dtype and widened again (what an op "in the input dtype" produces)
template <typename T> __device__ __forceinline__ float round_to_dtype(float v);
template <> __device__ __forceinline__ float round_to_dtype<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to_dtype<__half>(float v) { return __half2float(__float2half_rn(v)); }

}  // namespace obb
