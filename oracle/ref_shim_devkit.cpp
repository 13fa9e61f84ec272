// TEST INFRASTRUCTURE -- the reference-side binding of boundary B3, as far as this image can build it.
//
// The reference's devkit binds `_poly_nms` / `_overlaps` from Cython (DOTA_devkit/poly_nms_gpu/poly_nms.pyx:6-7,
// poly_overlaps.pyx:4-5: `cdef extern from "poly_nms.hpp"`), i.e. from a C++ translation unit that includes the reference's own
// headers and therefore asks the linker for the C++-MANGLED names.  The shipped Cython output (poly_nms.cpp, poly_overlaps.cpp) does
// not compile against this image's numpy 2.2 (PyArray_Descr::subarray is gone) and the .pyx does not re-cythonize with Cython 3.2
// (np.int_t / np.float are no longer types) -- so this file is the same binding without the Cython layer: it includes the
// reference's headers IN PLACE (-I /root/reference/DOTA_devkit/poly_nms_gpu, nothing is copied), calls the two functions with the
// declarations those headers give them, and is linked against yolov5_obb_amd/libobb_hip.so.  If the library did not export the
// mangled names, this would not link.  tests/test_devkit_binding.py drives it like poly_nms.pyx:9-24 / poly_overlaps.pyx:7-12 do.
#include "poly_nms.hpp"
#include "poly_overlaps.hpp"

extern "C" void ref_devkit_poly_nms(int* keep_out, int* num_out, const float* polys_host, int polys_num, int polys_dim, float thresh, int device_id) {
  _poly_nms(keep_out, num_out, polys_host, polys_num, polys_dim, thresh, device_id);
}
extern "C" void ref_devkit_overlaps(float* overlaps, const float* boxes, const float* query_boxes, int n, int k, int device_id) {
  _overlaps(overlaps, boxes, query_boxes, n, k, device_id);
}
