// Test-only HOST build of the product's device function library (gandiva_amd/csrc/
// gdv_device_lib.hpp): the per-row functions are plain C++, so compiled for x86 with the
// GPU intrinsics stubbed they can be driven over dense random inputs on a CPU-only machine
// and compared with the oracle.  Wave-level helpers (DPP scans, ballots, LDS staging) are
// stubbed and NOT exercised here; they are covered by the GPU parity suite.
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __forceinline__ inline
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_amdgcn_readlane(v, l) (v)
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) (old)
#define __builtin_amdgcn_wave_barrier() ((void)0)
static inline unsigned long long __ballot(bool x) { return x ? 1ull : 0ull; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; std::memcpy(&r, &v, 8); return r; }
static inline unsigned __float_as_uint(float f) { unsigned r; std::memcpy(&r, &f, 4); return r; }
static inline float __uint_as_float(unsigned v) { float r; std::memcpy(&r, &v, 4); return r; }

#include "../../gandiva_amd/csrc/gdv_device_lib.hpp"

extern "C" {

// op: 0 add, 1 subtract, 2 multiply, 3 divide, 4 mod.  Values are 16-byte little-endian.
int host_decimal_binary(int op, const void* xv, int xp, int xs, const void* yv, int yp, int ys, int op_, int os,
                        void* out, long n) {
  const gdv_int128* x = static_cast<const gdv_int128*>(xv);
  const gdv_int128* y = static_cast<const gdv_int128*>(yv);
  gdv_int128* r = static_cast<gdv_int128*>(out);
  unsigned err = 0;
  gdv_ctx ctx{&err};
  for (long i = 0; i < n; i++) {
    switch (op) {
      case 0: r[i] = add_decimal128_decimal128(x[i], xp, xs, y[i], yp, ys, op_, os); break;
      case 1: r[i] = subtract_decimal128_decimal128(x[i], xp, xs, y[i], yp, ys, op_, os); break;
      case 2: r[i] = multiply_decimal128_decimal128(x[i], xp, xs, y[i], yp, ys, op_, os); break;
      case 3: r[i] = divide_decimal128_decimal128(ctx, x[i], xp, xs, y[i], yp, ys, op_, os); break;
      default: r[i] = mod_decimal128_decimal128(ctx, x[i], xp, xs, y[i], yp, ys, op_, os); break;
    }
  }
  return static_cast<int>(err);
}

int host_decimal_cast(const void* xv, int xp, int xs, int op, int os, void* out, long n) {
  const gdv_int128* x = static_cast<const gdv_int128*>(xv);
  gdv_int128* r = static_cast<gdv_int128*>(out);
  for (long i = 0; i < n; i++) r[i] = castDECIMAL_decimal128(x[i], xp, xs, op, os);
  return 0;
}
int host_decimal_from_int64(const long long* v, int op, int os, void* out, long n) {
  gdv_int128* r = static_cast<gdv_int128*>(out);
  for (long i = 0; i < n; i++) r[i] = castDECIMAL_int64(v[i], op, os);
  return 0;
}
int host_decimal_to_int64(const void* xv, int xp, int xs, long long* out, long n) {
  const gdv_int128* x = static_cast<const gdv_int128*>(xv);
  for (long i = 0; i < n; i++) out[i] = castBIGINT_decimal128(x[i], xp, xs, 0, 0);
  return 0;
}
int host_decimal_compare(const void* xv, int xp, int xs, const void* yv, int yp, int ys, signed char* out, long n) {
  const gdv_int128* x = static_cast<const gdv_int128*>(xv);
  const gdv_int128* y = static_cast<const gdv_int128*>(yv);
  for (long i = 0; i < n; i++) out[i] = static_cast<signed char>(gdv_dec_compare(x[i], xp, xs, y[i], yp, ys));
  return 0;
}

}  // extern "C"
