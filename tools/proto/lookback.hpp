// Prototype of the decoupled look-back shared by the single-pass filter (K2) and the
// single-pass var-len projection (K4).  State granule: one 8-byte word per (tile, stream):
// bits 63..62 = status (0 nothing, 1 tile aggregate, 2 inclusive prefix), bits 61..0 = value.
// Written and read with relaxed agent-scope atomics (R2 "the data is the flag" granules of
// cdna_hip_programming.md Guideline 16): visible across XCDs, L1 bypassed, no fences needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define LB_A (1ull << 62)
#define LB_P (2ull << 62)
#define LB_VAL ((1ull << 62) - 1)

__device__ __forceinline__ void lb_store(uint64_t* p, uint64_t v) {
  __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t lb_load(const uint64_t* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_add(int acc, int src) {
  return acc + __builtin_amdgcn_update_dpp(0, src, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int wave_scan_incl(int v) {
  int r = v;
  r = dpp_add<0x111, 0xf, 0xf>(r, v);
  r = dpp_add<0x112, 0xf, 0xf>(r, v);
  r = dpp_add<0x113, 0xf, 0xf>(r, v);
  r = dpp_add<0x114, 0xf, 0xe>(r, r);
  r = dpp_add<0x118, 0xf, 0xc>(r, r);
  r = dpp_add<0x142, 0xa, 0xf>(r, r);
  r = dpp_add<0x143, 0xc, 0xf>(r, r);
  return r;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane(wave_scan_incl((int)v), 63);
}
// sum of 64 values < 2^62 (16-bit limbs: every limb sum stays below 2^22)
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) s += (uint64_t)wave_sum_u32((uint32_t)(v >> (16 * k)) & 0xffffu) << (16 * k);
  return s;
}

// Exclusive prefix of `agg` over tiles [0, tile) of stream `state` (stride 1 between tiles).
// Publishes the aggregate first, the inclusive prefix last.  All 64 lanes call it; the
// result is wave-uniform.  SMALL: aggregates are < 2^25 so 64 of them sum in 32 bits.
template <bool SMALL>
__device__ __forceinline__ uint64_t lookback(uint64_t* state, int64_t tile, uint64_t agg, int lane) {
  if (tile == 0) {
    if (lane == 0) lb_store(state, LB_P | agg);
    return 0;
  }
  if (lane == 0) lb_store(state + tile, LB_A | agg);
  uint64_t excl = 0;
  int64_t pos = tile - 1;
  for (;;) {
    const int64_t idx = pos - lane;
    const uint64_t s = idx >= 0 ? lb_load(state + idx) : LB_P;  // a virtual prefix 0 before tile 0
    const uint32_t st = (uint32_t)(s >> 62);
    const uint64_t missing = __ballot(st == 0);
    const uint64_t pmask = __ballot(st == 2);
    // window = lanes up to and including the nearest prefix (or all 64 when there is none)
    const int fp = pmask ? __builtin_ctzll(pmask) : 63;
    const uint64_t need = fp == 63 ? ~0ull : ((2ull << fp) - 1);
    if (missing & need) {
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    const uint64_t v = lane <= fp ? (s & LB_VAL) : 0;
    if (SMALL) {
      // the prefix value may be large: take it separately, sum the small aggregates in 32 bits
      const uint64_t pv = pmask ? (((uint64_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), fp) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((uint32_t)v, fp)) : 0;
      excl += pv + wave_sum_u32((pmask && lane == fp) ? 0u : (uint32_t)v);
    } else {
      excl += wave_sum_u64(v);
    }
    if (pmask) break;
    pos -= 64;
  }
  if (lane == 0) lb_store(state + tile, LB_P | (excl + agg));
  return excl;
}

// Wide window: every lane polls W predecessor granules per step (window = 64*W tiles), so the
// chain of inclusive prefixes advances 64*W tiles per visibility round trip.
template <int W>
__device__ __forceinline__ uint64_t lookback_wide(uint64_t* state, int64_t tile, uint64_t agg, int lane) {
  if (tile == 0) {
    if (lane == 0) lb_store(state, LB_P | agg);
    return 0;
  }
  if (lane == 0) lb_store(state + tile, LB_A | agg);
  uint64_t excl = 0;
  int64_t pos = tile - 1;
  for (;;) {
    // lane handles predecessors pos - (W*lane + k), k = 0..W-1 (nearest first)
    uint64_t s[W];
#pragma unroll
    for (int k = 0; k < W; k++) {
      const int64_t idx = pos - (W * lane + k);
      s[k] = idx >= 0 ? lb_load(state + idx) : LB_P;
    }
    int firstp = W;
    bool ok = true;
    uint64_t sum = 0;
#pragma unroll
    for (int k = 0; k < W; k++) {
      const uint32_t st = (uint32_t)(s[k] >> 62);
      if (firstp == W) {
        ok = ok && st != 0;
        sum += s[k] & LB_VAL;
        if (st == 2) firstp = k;
      }
    }
    const uint64_t pmask = __ballot(firstp < W);
    const int fp = pmask ? __builtin_ctzll(pmask) : 63;
    const uint64_t need = fp == 63 ? ~0ull : ((2ull << fp) - 1);
    if (__ballot(!ok) & need) {
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    excl += wave_sum_u64(lane <= fp ? sum : 0);
    if (pmask) break;
    pos -= 64 * W;
  }
  if (lane == 0) lb_store(state + tile, LB_P | (excl + agg));
  return excl;
}

// ---------------------------------------------------------------------------------------------
// Two-level decoupled look-back over NV streams at once.
//   T[tile * NV + e]  tile granule   (A | aggregate of the tile)
//   G[group * NV + e] group granule  (A | aggregate of the 64 tiles of the group, later
//                                     P | inclusive prefix through the end of the group)
// A tile waits only on AGGREGATES of the <= 63 earlier tiles of its own group (posted as soon
// as their lengths are known) and on the group granules before it; inclusive prefixes are
// chained at group granularity (64 groups = 4096 tiles per visibility round trip), so the
// chain keeps up with any realistic tile rate.  The last tile of a group publishes the group
// aggregate before its own group-level look-back and the group prefix after it.
template <int NV>
__device__ __forceinline__ void lookback2(uint64_t* T, uint64_t* G, int64_t tile, const uint64_t (&agg)[NV],
                                          uint64_t (&excl)[NV], int lane) {
  const int64_t g = tile >> 6;
  const int i = (int)(tile & 63);
#pragma unroll
  for (int e = 0; e < NV; e++)
    if (lane == e) lb_store(T + tile * NV + e, LB_A | agg[e]);
  uint64_t in_group[NV];
#pragma unroll
  for (int e = 0; e < NV; e++) in_group[e] = 0;
  if (i > 0) {
    for (;;) {
      uint64_t s[NV];
      bool ok = true;
#pragma unroll
      for (int e = 0; e < NV; e++) {
        s[e] = lane < i ? lb_load(T + ((g << 6) + lane) * NV + e) : LB_A;
        ok = ok && (s[e] >> 62) != 0;
      }
      if (__ballot(!ok) != 0) {
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
#pragma unroll
      for (int e = 0; e < NV; e++) in_group[e] = wave_sum_u64(s[e] & LB_VAL);
      break;
    }
  }
  if (i == 63 && g > 0) {
#pragma unroll
    for (int e = 0; e < NV; e++)
      if (lane == e) lb_store(G + g * NV + e, LB_A | (in_group[e] + agg[e]));
  }
  uint64_t gex[NV];
#pragma unroll
  for (int e = 0; e < NV; e++) gex[e] = 0;
  if (g > 0) {
    bool done[NV];
#pragma unroll
    for (int e = 0; e < NV; e++) done[e] = false;
    int64_t pos = g - 1;
    for (;;) {
      const int64_t idx = pos - lane;
      uint64_t s[NV];
      int fp[NV];
      bool hasp[NV];
      bool retry = false;
#pragma unroll
      for (int e = 0; e < NV; e++) {
        s[e] = idx >= 0 ? lb_load(G + idx * NV + e) : LB_P;
        const uint32_t st = (uint32_t)(s[e] >> 62);
        const uint64_t missing = __ballot(st == 0);
        const uint64_t pmask = __ballot(st == 2);
        hasp[e] = pmask != 0;
        fp[e] = pmask ? __builtin_ctzll(pmask) : 63;
        const uint64_t need = fp[e] == 63 ? ~0ull : ((2ull << fp[e]) - 1);
        retry = retry || (!done[e] && (missing & need) != 0);
      }
      if (retry) {
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      bool all = true;
#pragma unroll
      for (int e = 0; e < NV; e++) {
        if (!done[e]) {
          gex[e] += wave_sum_u64(lane <= fp[e] ? (s[e] & LB_VAL) : 0);
          done[e] = hasp[e];
        }
        all = all && done[e];
      }
      if (all) break;
      pos -= 64;
    }
  }
  if (i == 63) {
#pragma unroll
    for (int e = 0; e < NV; e++)
      if (lane == e) lb_store(G + g * NV + e, LB_P | (gex[e] + in_group[e] + agg[e]));
  }
#pragma unroll
  for (int e = 0; e < NV; e++) excl[e] = gex[e] + in_group[e];
}
