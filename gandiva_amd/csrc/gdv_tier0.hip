// Tier 0 (round 6): ONE ahead-of-time kernel that evaluates a projection or a filter predicate by INTERPRETING a small
// post-fix program over 64-row sub-tiles — what a Projector / Filter runs on while hipRTC is still compiling its
// specialised kernel (0.25-0.9 s for an unseen tree; an empty translation unit alone costs hipRTC ~0.2 s, so no amount
// of header trimming brings Make near the reference's tens of milliseconds).  SURVEY.md §3.1 option (a).
//
// Same argument block as the generated kernels (gdv_planner.h: ArgLayout — the engine binds inputs and outputs exactly
// as for the specialised kernel), same arithmetic (this file includes the device library and calls its functions, compiled
// with the same -ffp-contract=off): results are bit-identical, only slower — the program is walked per sub-tile, operands
// live on an LDS stack, nothing is shared between expressions.
//
// One wave = one sub-tile of 64 rows (lane = row) at a time.  Values are 64-bit slots: signed integers sign-extended,
// unsigned zero-extended, float32 in the low half, bool 0 / 1.  Validity is a wave-uniform 64-bit word per stack entry.
#include <hip/hip_runtime.h>

#include <algorithm>

#define GDV_U 1
#define GDV_WAVES 4
#include "gdv_device_lib.hpp"
#include "gdv_tier0.h"

namespace gdv {
namespace {

using namespace tier0;

__device__ __forceinline__ gdv_uint64 Normalise(gdv_uint64 v, int tk) {
  switch (tk) {
    case kTI8: return (gdv_uint64)(gdv_int64)(gdv_int8)v;
    case kTU8: return (gdv_uint8)v;
    case kTI16: return (gdv_uint64)(gdv_int64)(gdv_int16)v;
    case kTU16: return (gdv_uint16)v;
    case kTI32: return (gdv_uint64)(gdv_int64)(gdv_int32)v;
    case kTU32: return (gdv_uint32)v;
    default: return v;
  }
}
__device__ __forceinline__ gdv_float32 AsF32(gdv_uint64 v) { return __uint_as_float((gdv_uint32)v); }
__device__ __forceinline__ gdv_float64 AsF64(gdv_uint64 v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ gdv_uint64 FromF32(gdv_float32 f) { return (gdv_uint64)__float_as_uint(f); }
__device__ __forceinline__ gdv_uint64 FromF64(gdv_float64 d) { return (gdv_uint64)__double_as_longlong(d); }

// word `w` of a bitmap bound by the engine (gdv_bitmap: pointer, bit shift, readable words; index clamped)
__device__ __forceinline__ gdv_uint64 BitmapWord(const gdv_uint8* slot, gdv_int64 w) {
  gdv_bitmap bm;
  __builtin_memcpy(&bm, slot, sizeof(bm));
  const gdv_int64 last = bm.nwords - 1;
  const gdv_int64 i0 = w < last ? w : last;
  const gdv_int64 i1 = w + 1 < last ? w + 1 : last;
  const gdv_uint64 lo = bm.p[i0], hi = bm.p[i1];
  return (lo >> bm.shift) | ((hi << 1) << (63 - bm.shift));
}

__device__ __forceinline__ gdv_uint64 LoadValue(const void* data, gdv_int64 row, int tk) {
  switch (tk) {
    case kTI8: return (gdv_uint64)(gdv_int64)((const gdv_int8*)data)[row];
    case kTU8: return ((const gdv_uint8*)data)[row];
    case kTI16: return (gdv_uint64)(gdv_int64)((const gdv_int16*)data)[row];
    case kTU16: return ((const gdv_uint16*)data)[row];
    case kTI32: return (gdv_uint64)(gdv_int64)((const gdv_int32*)data)[row];
    case kTU32: case kTF32: return ((const gdv_uint32*)data)[row];
    default: return ((const gdv_uint64*)data)[row];
  }
}
__device__ __forceinline__ void StoreValue(void* data, gdv_int64 row, int tk, gdv_uint64 v) {
  switch (tk) {
    case kTI8: case kTU8: ((gdv_uint8*)data)[row] = (gdv_uint8)v; break;
    case kTI16: case kTU16: ((gdv_uint16*)data)[row] = (gdv_uint16)v; break;
    case kTI32: case kTU32: case kTF32: ((gdv_uint32*)data)[row] = (gdv_uint32)v; break;
    default: ((gdv_uint64*)data)[row] = v; break;
  }
}

__device__ __forceinline__ gdv_uint64 Arith(int op, int tk, gdv_uint64 a, gdv_uint64 b) {
  if (tk == kTF64) {
    const gdv_float64 x = AsF64(a), y = AsF64(b);
    return FromF64(op == kAdd ? add_float64_float64(x, y) : op == kSub ? subtract_float64_float64(x, y) : multiply_float64_float64(x, y));
  }
  if (tk == kTF32) {
    const gdv_float32 x = AsF32(a), y = AsF32(b);
    return FromF32(op == kAdd ? add_float32_float32(x, y) : op == kSub ? subtract_float32_float32(x, y) : multiply_float32_float32(x, y));
  }
  // integers wrap: 64-bit two's-complement arithmetic, then back to the type's width
  const gdv_uint64 r = op == kAdd ? a + b : op == kSub ? a - b : a * b;
  return Normalise(r, tk);
}

__device__ __forceinline__ bool Compare(int kind, int tk, gdv_uint64 a, gdv_uint64 b) {
  int lt, eq;
  if (tk == kTF64) { const gdv_float64 x = AsF64(a), y = AsF64(b); lt = x < y; eq = x == y; if (kind == kLe) return x <= y; if (kind == kGe) return x >= y; if (kind == kGt) return x > y; if (kind == kNe) return x != y; }
  else if (tk == kTF32) { const gdv_float32 x = AsF32(a), y = AsF32(b); lt = x < y; eq = x == y; if (kind == kLe) return x <= y; if (kind == kGe) return x >= y; if (kind == kGt) return x > y; if (kind == kNe) return x != y; }
  else if (tk == kTU8 || tk == kTU16 || tk == kTU32 || tk == kTU64 || tk == kTBool) { lt = a < b; eq = a == b; }
  else { lt = (gdv_int64)a < (gdv_int64)b; eq = a == b; }
  switch (kind) {
    case kEq: return eq;
    case kNe: return !eq;
    case kLt: return lt;
    case kLe: return lt || eq;
    case kGt: return !lt && !eq;
    default: return !lt;
  }
}

__device__ __forceinline__ gdv_uint64 Cast(int from, int to, gdv_uint64 v) {
  const bool from_unsigned = from == kTU8 || from == kTU16 || from == kTU32 || from == kTU64;
  if (to == kTF32) {
    if (from == kTF64) return FromF32(castFLOAT4_float64(AsF64(v)));
    if (from == kTI32) return FromF32(castFLOAT4_int32((gdv_int32)v));
    return FromF32(from_unsigned ? (gdv_float32)v : castFLOAT4_int64((gdv_int64)v));
  }
  if (to == kTF64) {
    if (from == kTF32) return FromF64(castFLOAT8_float32(AsF32(v)));
    if (from == kTI32) return FromF64(castFLOAT8_int32((gdv_int32)v));
    return FromF64(from_unsigned ? (gdv_float64)v : castFLOAT8_int64((gdv_int64)v));
  }
  if (from == kTF32) return to == kTI32 ? (gdv_uint64)(gdv_int64)castINT_float32(AsF32(v)) : (gdv_uint64)castBIGINT_float32(AsF32(v));
  if (from == kTF64) return to == kTI32 ? (gdv_uint64)(gdv_int64)castINT_float64(AsF64(v)) : (gdv_uint64)castBIGINT_float64(AsF64(v));
  return Normalise(v, to);  // integer -> integer: truncate / extend
}

}  // namespace

__global__ void __launch_bounds__(256) Tier0Kernel(const tier0::Args P) {
  __shared__ gdv_uint64 stack[4][tier0::kMaxDepth][64];
  __shared__ gdv_uint64 vstack[4][tier0::kMaxDepth];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const gdv_uint8* const block = P.block;
  gdv_int64 n;
  __builtin_memcpy(&n, block + 0, 8);
  const gdv_int64 nwords = (n + 63) >> 6;
  const int U = P.filter ? P.subtiles : 1;                      // filter: one count per wave tile of `subtiles` words
  const gdv_int64 ntiles = (nwords + U - 1) / U;
  gdv_uint64* const mask = P.filter ? *(gdv_uint64* const*)(block + 24) : nullptr;
  gdv_uint32* const counts = P.filter ? *(gdv_uint32* const*)(block + 32) : nullptr;
  const gdv_uint8* const in_base = block + 64;
  const gdv_uint8* const out_base = in_base + (P.n_in > 0 ? P.n_in : 1) * 64;
  for (gdv_int64 t = (gdv_int64)blockIdx.x * 4 + wave; t < ntiles; t += (gdv_int64)gridDim.x * 4) {
    gdv_uint32 fcount = 0;
    for (int u = 0; u < U; u++) {
      const gdv_int64 w = t * U + u;
      if (w >= nwords) break;
      const gdv_int64 row = w * 64 + lane;
      const bool live = row < n;
      const gdv_uint64 livemask = __ballot(live);
      int sp = 0;
      for (int pc = 0; pc < P.ncode; pc++) {
        const gdv_uint32 ins = P.code[pc];
        const int op = ins & 0xff, a = (ins >> 8) & 0xff, b = (ins >> 16) & 0xff, c = (ins >> 24) & 0xff;
        switch (op) {
          case kLoad: {  // a = input slot, b = type, c = bit 0 values needed, bit 1 validity needed
            const gdv_uint8* slot = in_base + a * 64;
            gdv_uint64 v = 0;
            if (c & 1) {
              if (b == kTBool) v = (BitmapWord(slot + 32, w) >> lane) & 1;
              else if (live) v = LoadValue(*(const void* const*)slot, row, b);
            }
            stack[wave][sp][lane] = v;
            if (lane == 0) vstack[wave][sp] = (c & 2) ? BitmapWord(slot + 8, w) : ~0ull;
            sp++;
            break;
          }
          case kLit:  // a = literal, b = is null
            stack[wave][sp][lane] = P.lits[a];
            if (lane == 0) vstack[wave][sp] = b ? 0ull : ~0ull;
            sp++;
            break;
          case kAdd: case kSub: case kMul: {  // b = type
            const gdv_uint64 y = stack[wave][sp - 1][lane], x = stack[wave][sp - 2][lane];
            stack[wave][sp - 2][lane] = Arith(op, b, x, y);
            if (lane == 0) vstack[wave][sp - 2] &= vstack[wave][sp - 1];
            sp--;
            break;
          }
          case kCmp: {  // a = comparison, b = operand type
            const gdv_uint64 y = stack[wave][sp - 1][lane], x = stack[wave][sp - 2][lane];
            stack[wave][sp - 2][lane] = Compare(a, b, x, y) ? 1 : 0;
            if (lane == 0) vstack[wave][sp - 2] &= vstack[wave][sp - 1];
            sp--;
            break;
          }
          case kCast:  // a = from, b = to
            stack[wave][sp - 1][lane] = Cast(a, b, stack[wave][sp - 1][lane]);
            break;
          case kNot:
            stack[wave][sp - 1][lane] ^= 1;
            break;
          case kIsNull: case kIsNotNull: {
            const bool valid = (vstack[wave][sp - 1] >> lane) & 1;
            __builtin_amdgcn_wave_barrier();
            stack[wave][sp - 1][lane] = (op == kIsNull) != valid ? 1 : 0;
            if (lane == 0) vstack[wave][sp - 1] = ~0ull;
            break;
          }
          case kAnd2: case kOr2: {  // SQL three-valued logic: a definite FALSE (AND) / TRUE (OR) decides whatever the other side is
            const gdv_uint64 vy = vstack[wave][sp - 1], vx = vstack[wave][sp - 2];
            const bool y = stack[wave][sp - 1][lane] & 1, x = stack[wave][sp - 2][lane] & 1;
            const bool yv = (vy >> lane) & 1, xv = (vx >> lane) & 1;
            const bool dom = op == kOr2;  // the dominating value
            const bool decided = (xv && x == dom) || (yv && y == dom);
            const bool valid = decided || (xv && yv);
            const bool value = decided ? dom : (op == kAnd2 ? (x && y) : (x || y));
            __builtin_amdgcn_wave_barrier();
            stack[wave][sp - 2][lane] = value ? 1 : 0;
            const gdv_uint64 vw = __ballot(valid);
            if (lane == 0) vstack[wave][sp - 2] = vw;
            sp--;
            break;
          }
          case kIf: {  // condition, then, else on the stack: a NULL condition takes the else branch
            const gdv_uint64 vc = vstack[wave][sp - 3], vt = vstack[wave][sp - 2], ve = vstack[wave][sp - 1];
            const bool take = ((vc >> lane) & 1) && (stack[wave][sp - 3][lane] & 1);
            const gdv_uint64 v = take ? stack[wave][sp - 2][lane] : stack[wave][sp - 1][lane];
            const bool valid = take ? (vt >> lane) & 1 : (ve >> lane) & 1;
            __builtin_amdgcn_wave_barrier();
            stack[wave][sp - 3][lane] = v;
            const gdv_uint64 vw = __ballot(valid);
            if (lane == 0) vstack[wave][sp - 3] = vw;
            sp -= 2;
            break;
          }
          case kOut: {  // a = output, b = type
            const gdv_uint8* slot = out_base + a * 32;
            void* data = *(void* const*)slot;
            gdv_uint64* valid = *(gdv_uint64* const*)(slot + 8);
            const gdv_uint64 v = stack[wave][sp - 1][lane];
            const gdv_uint64 vw = vstack[wave][sp - 1] & livemask;
            if (b == kTBool) {
              const gdv_uint64 bits = __ballot(live && (v & 1));
              if (lane == 0) ((gdv_uint64*)data)[w] = bits;
            } else if (live) {
              StoreValue(data, row, b, v);
            }
            if (lane == 0) valid[w] = vw;
            sp--;
            break;
          }
          case kFilterOut: {
            const gdv_uint64 fm = __ballot(live && (stack[wave][sp - 1][lane] & 1)) & vstack[wave][sp - 1];
            fcount += (gdv_uint32)__popcll(fm);
            if (lane == 0) mask[w] = fm;
            sp--;
            break;
          }
          default:
            break;
        }
        __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave execute in order: ordering for the compiler only)
      }
    }
    if (P.filter && lane == 0) counts[t] = fcount;
  }
}

hipError_t LaunchTier0(const tier0::Args& args, int64_t rows, int num_cus, hipStream_t stream) {
  const int64_t nwords = (rows + 63) / 64;
  const int U = args.filter ? args.subtiles : 1;
  const int64_t ntiles = (nwords + U - 1) / U;
  const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((ntiles + 3) / 4, static_cast<int64_t>(num_cus) * 8));
  hipLaunchKernelGGL(Tier0Kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, args);
  return hipGetLastError();
}

}  // namespace gdv
