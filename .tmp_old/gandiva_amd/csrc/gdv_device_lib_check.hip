// Build-time check only: compiles the whole device function library for gfx950 as an
// ordinary translation unit (the product hands the same header to hipRTC at Make time).
#include <hip/hip_runtime.h>

#include "gdv_device_lib.hpp"

extern "C" __global__ void gdv_device_lib_check(const double* a, double* o, unsigned* err) {
  gdv_ctx ctx{err};
  o[threadIdx.x] = divide_float64_float64(ctx, add_float64_float64(a[threadIdx.x], 1.0), 2.0) +
                   (double)hash64_float64(a[threadIdx.x], true) +
                   (double)extractYear_timestamp((gdv_int64)a[threadIdx.x]);
}
