// Prototype (end of round 3, UNMEASURED — to be run first thing next round: tools/proto/ev_k4r.sh).
// The product's C5 main kernel is VALU-bound (profiles/r03_c5_tuning.txt): it sweeps one 64-row sub-tile
// at a time so that the span fits an LDS mirror, and those steps are three quarters full on average.
// c5_main_ring below sweeps the tile's span in FULL 1024-byte steps that run just ahead of the row
// loop (while swept < end of the sub-tile's span) into a RING mirror + ring match bitmap (4 KB of span
// per wave): the full-step efficiency of the tile-wide sweep AND no second read of the rows' bytes.
// The k4h variants are kept as the reference point inside the same binary.
//
// (k4h:) BASELINE config C5 (like '%spark%', substr(s,2,5), upper(s)) WITHOUT any
// in-kernel hand-off.  Under the optimistic-ASCII assumption the length of substr(s,2,5) is a
// function of the input offsets alone, so
//   P  pre-pass: offsets in (0.4 GB) -> byte total of every wave tile (1.5 MB)
//   S  exclusive scan of the wave-tile totals (rocprim here; ScanReduce/Spine/Apply in the product)
//   M  main kernel: every WAVE is independent — its output base is one scalar load; no scanner, no
//      look-back, no workgroup barrier, no LDS shared between waves
// Variants (template flags F): 1 = flat copy straight from the sweep's registers, 2 = one offsets
// load per sub-tile (ob from the next lane by DPP), 4 = all sweep loads issued before the first use.
// Standalone; hipcc --offload-arch=gfx950 -O3 -o k4h_proto k4h_proto.hip
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "lookback.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define B80 0x8080808080808080ull
#define B01 0x0101010101010101ull

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
__global__ void gen_lens(int32_t* lens, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    lens[i] = 4 + (int32_t)(mix(3 * i + 1) % 17);
}
__global__ void gen_bytes(const int32_t* off, uint8_t* data, int64_t n) {
  const char* letters = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ";
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = off[i], len = off[i + 1] - a;
    for (int k = 0; k < len; k++) data[a + k] = letters[mix(i * 32 + k + 77) % 52];
    const uint64_t h = mix(3 * i + 2);
    if (h % 20 == 0 && len >= 5) {
      const int p = (int)((h >> 20) % (len - 4));
      for (int k = 0; k < 5; k++) data[a + p + k] = "spark"[k];
    }
  }
}
__global__ void ref_lens(const int32_t* off, int32_t* sub_len, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t len = off[i + 1] - off[i];
    sub_len[i] = len <= 1 ? 0 : (len - 1 < 5 ? len - 1 : 5);
  }
}
__global__ void ref_check(const int32_t* off, const uint8_t* data, int64_t n, const uint64_t* like_bits,
                          const int32_t* sub_off_ref, const int32_t* sub_off, const uint8_t* sub_dat,
                          const int32_t* up_off, const uint8_t* up_dat, unsigned* bad) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = off[i], len = off[i + 1] - a;
    bool hit = false;
    for (int p = 0; p + 5 <= len && !hit; p++) {
      hit = true;
      for (int k = 0; k < 5; k++) hit = hit && data[a + p + k] == (uint8_t)"spark"[k];
    }
    if ((((like_bits[i >> 6] >> (i & 63)) & 1) != 0) != hit) atomicOr(bad, 1u);
    if (sub_off[i] != sub_off_ref[i] || sub_off[i + 1] != sub_off_ref[i + 1]) atomicOr(bad, 2u);
    else for (int k = 0; k < sub_off[i + 1] - sub_off[i]; k++)
      if (sub_dat[sub_off[i] + k] != data[a + 1 + k]) atomicOr(bad, 4u);
    if (up_off[i] != a || up_off[i + 1] != off[i + 1]) atomicOr(bad, 8u);
    else for (int k = 0; k < len; k++) {
      uint8_t c = data[a + k];
      if (c >= 'a' && c <= 'z') c -= 32;
      if (up_dat[a + k] != c) atomicOr(bad, 16u);
    }
  }
}

struct Args {
  int64_t n;
  const int32_t* off; const uint8_t* data;
  uint64_t *like_bits, *like_valid, *sub_valid, *up_valid;
  int32_t* sub_off; uint8_t* sub_dat; int32_t* up_off; uint8_t* up_dat;
  uint32_t* wt_tot;       // per wave tile: bytes of the substr output (pre-pass)
  uint32_t* wt_base;      // exclusive scan of wt_tot
  unsigned* flags;        // bit 0: a tile was not ASCII (host re-runs on the general kernel)
  int64_t cap_sub, cap_up;
};

__device__ __forceinline__ uint64_t upper8(uint64_t w) {
  const uint64_t h = w & 0x7f7f7f7f7f7f7f7full, ascii = ~w & B80;
  const uint64_t in_range = (h + 0x1f1f1f1f1f1f1f1full) & ~(h + 0x0505050505050505ull) & ascii;
  return w ^ (in_range >> 2);
}
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }

__device__ __forceinline__ uint32_t match8(uint64_t cur, uint64_t nxt, uint64_t first, uint64_t mask,
                                           uint64_t splat0, uint64_t splat1) {
  const uint64_t x = cur ^ splat0;
  uint64_t cand = (x - B01) & ~x & B80;
  const uint64_t y = ((cur >> 8) | (nxt << 56)) ^ splat1;
  cand &= (y - B01) & ~y & B80;
  uint32_t m = 0;
  while (cand) {
    const int k = __builtin_ctzll(cand) >> 3;
    cand &= cand - 1;
    const uint64_t win = k == 0 ? cur : ((cur >> (8 * k)) | (nxt << (64 - 8 * k)));
    if ((win & mask) == first) m |= 1u << k;
  }
  return m;
}
// any bit in [lo, hi) of the bitmap; ranges up to 64 positions are branch-free
__device__ __forceinline__ bool range_any(const uint64_t* bm, int lo, int hi) {
  const int nbits = hi - lo;
  const int w = lo >> 6, s = lo & 63;
  const uint64_t x = (bm[w] >> s) | ((bm[w + 1] << 1) << (63 - s));
  const uint64_t m = nbits >= 64 ? ~0ull : ((1ull << (nbits > 0 ? nbits : 0)) - 1ull);
  bool any = nbits > 0 && (x & m) != 0;
  if (nbits > 64 && !any) {
    for (int p = lo + 64; p < hi && !any; p++) any = (bm[p >> 6] >> (p & 63)) & 1;
  }
  return any;
}
__device__ __forceinline__ int32_t sub_len_ascii(int32_t len) {
  int32_t sl = len - 1 < 5 ? len - 1 : 5;
  return sl < 0 ? 0 : sl;
}

// ---- P: offsets -> per-wave-tile byte totals of the substr output
template <int U, int W>
__global__ void __launch_bounds__(W * 64) c5_prepass(const Args A) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n = A.n;
  const int64_t nwt = (n + 64 * U - 1) / (64 * U);
  for (int64_t wt = (int64_t)blockIdx.x * W + wave; wt < nwt; wt += (int64_t)gridDim.x * W) {
    const int64_t row0 = wt * (64 * U);
    int32_t o[U + 1];
    if (row0 + 64 * U < n) {
#pragma unroll
      for (int u = 0; u < U; u++) o[u] = __builtin_nontemporal_load(A.off + row0 + u * 64 + lane);
      o[U] = A.off[row0 + U * 64];
    } else {
#pragma unroll
      for (int u = 0; u <= U; u++) { const int64_t r = row0 + u * 64 + lane; o[u] = A.off[r < n ? r : n]; }
    }
    int32_t s = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      // ob = the next lane's oa (lane 63: the next sub-tile's lane 0)
      int32_t nx = __builtin_amdgcn_update_dpp(0, o[u], 0x130, 0xf, 0xf, false);  // wave_shl:1
      const int32_t first_next = __builtin_amdgcn_readfirstlane(o[u + 1]);
      if (lane == 63) nx = first_next;
      s += sub_len_ascii(nx - o[u]);
    }
    const uint32_t t = wave_sum_u32((uint32_t)s);
    if (lane == 0) A.wt_tot[wt] = t;
  }
}

// ---- M: one independent wave per tile of 64*U rows
template <int U, int W, int F>
__global__ void __launch_bounds__(W * 64) c5_main(const Args A) {
  constexpr int IN_WIN = U * 64 * 16;    // span bytes the hit bitmap covers: 16 per row on average
  constexpr int OUT_WIN = U * 64 * 8;    // staged substr bytes per wave tile
  constexpr int NIT = IN_WIN / 1024 + 1;
  __shared__ __attribute__((aligned(16))) uint8_t outwin[W][OUT_WIN + 16];
  __shared__ __attribute__((aligned(16))) uint64_t hitmap[W][IN_WIN / 64 + 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n = A.n;
  const int64_t wt = (int64_t)blockIdx.x * W + wave;
  const int64_t row0 = wt * (64 * U);
  if (row0 >= n) return;
  const int32_t* __restrict__ off = A.off;
  const uint8_t* __restrict__ data = A.data;
  const bool full = row0 + 64 * U < n;
  // wave-uniform: span of the tile and the output base (scalar loads, independent of everything)
  const int64_t rend = full ? row0 + 64 * U : n;
  const int32_t s0 = off[row0], s1 = off[rend], so0 = off[0];
  const int64_t sub_base = A.wt_base[wt];
  const int32_t base = s0 & ~15;
  const bool big = s1 - base > IN_WIN;

  int32_t oa[U], ob[U];
  if (F & 2) {
    int32_t o[U + 1];
    if (full) {
#pragma unroll
      for (int u = 0; u < U; u++) o[u] = off[row0 + u * 64 + lane];
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) { const int64_t r = row0 + u * 64 + lane; o[u] = off[r < n ? r : n]; }
    }
    o[U] = s1;
#pragma unroll
    for (int u = 0; u < U; u++) {
      int32_t nx = __builtin_amdgcn_update_dpp(0, o[u], 0x130, 0xf, 0xf, false);
      const int32_t first_next = u + 1 < U ? __builtin_amdgcn_readfirstlane(o[u + 1]) : s1;
      if (lane == 63) nx = first_next;
      oa[u] = o[u]; ob[u] = nx;
    }
  } else if (full) {
#pragma unroll
    for (int u = 0; u < U; u++) { oa[u] = off[row0 + u * 64 + lane]; ob[u] = off[row0 + u * 64 + lane + 1]; }
  } else {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t r = row0 + u * 64 + lane;
      oa[u] = off[r < n ? r : n];
      ob[u] = off[r + 1 < n ? r + 1 : n];
    }
  }

  const uint64_t needle = 0x6b72617073ull, nmask = 0xffffffffffull;
  const uint64_t splat0 = 0x73 * B01, splat1 = 0x70 * B01;
  uint64_t acc = 0;
  uint8_t* __restrict__ up_dst = A.up_dat;
  const bool up_fits = (int64_t)s1 - so0 <= A.cap_up;
  auto sweep_piece = [&](int32_t a, const uint64_t (&w)[2], uint64_t tail) {
    const uint64_t lo = w[0], hi = w[1];
    uint64_t nxt = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(lo >> 32), 0x130, 0xf, 0xf, false) << 32) |
                   (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)lo, 0x130, 0xf, 0xf, false);
    if (lane == 63) nxt = tail;
    acc |= lo | hi;
    const uint32_t m = match8(lo, hi, needle, nmask, splat0, splat1) | (match8(hi, nxt, needle, nmask, splat0, splat1) << 8);
    if (a < s1) ((uint16_t*)hitmap[wave])[(a - base) >> 4] = (uint16_t)m;
    if ((F & 1) && up_fits && a < s1) {
      // flat output straight from the sweep's registers: bytes [a, a+16) of the input are bytes
      // [a - so0, ...) of the output; only the bytes inside this wave's span [s0, s1) are its own
      const uint64_t m0 = upper8(lo), m1 = upper8(hi);
      if (a >= s0 && a + 16 <= s1) {
        uint64_t q[2] = {m0, m1};
        __builtin_memcpy(up_dst + (a - so0), q, 16);
      } else {
        const int k0 = a < s0 ? s0 - a : 0, k1 = s1 - a < 16 ? s1 - a : 16;
        for (int k = k0; k < k1; k++) up_dst[a - so0 + k] = (uint8_t)((k < 8 ? m0 : m1) >> (8 * (k & 7)));
      }
    }
  };
  if (!big) {
    if (F & 4) {
      uint64_t w[NIT][2];
      uint64_t tail[NIT];
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        const int32_t a = base + k * 1024 + 16 * lane;
        w[k][0] = 0; w[k][1] = 0; tail[k] = 0;
        if (a < s1) __builtin_memcpy(w[k], __builtin_assume_aligned(data + a, 16), 16);
      }
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        // lane 63's halo is lane 0 of the next step: no extra load
        const uint64_t t = k + 1 < NIT ? (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)w[k + 1 < NIT ? k + 1 : k][0]) |
                                         ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(w[k + 1 < NIT ? k + 1 : k][0] >> 32)) << 32)
                                       : 0ull;
        if (base + k * 1024 < s1) sweep_piece(base + k * 1024 + 16 * lane, w[k], t);
      }
    } else {
      for (int32_t c = base; c < s1; c += 1024) {
        const int32_t a = c + 16 * lane;
        uint64_t w[2] = {0, 0};
        if (a < s1) __builtin_memcpy(w, __builtin_assume_aligned(data + a, 16), 16);
        uint64_t tail = 0;
        if (lane == 63 && a + 16 < s1) tail = ld8(data + a + 16);
        sweep_piece(a, w, tail);
      }
    }
  }
  const bool tile_ascii = !big && __ballot((acc & B80) != 0) == 0;
  if (!tile_ascii && lane == 0) atomicOr(A.flags, 1u);  // pre-pass assumed ASCII: host re-runs
  if (!(F & 1) && up_fits) {
    uint8_t* __restrict__ dst = up_dst + (s0 - so0);
    const uint8_t* __restrict__ src = data + s0;
    const int32_t cnt = s1 - s0;
    if (cnt >= 16) {
      for (int32_t i = lane * 16; i < cnt; i += 1024) {
        const int32_t j = i + 16 <= cnt ? i : cnt - 16;
        uint64_t w[2];
        __builtin_memcpy(w, src + j, 16);
        w[0] = upper8(w[0]); w[1] = upper8(w[1]);
        __builtin_memcpy(dst + j, w, 16);
      }
    } else if (lane < cnt) {
      const uint8_t ch = src[lane];
      dst[lane] = (ch >= 'a' && ch <= 'z') ? ch - 32 : ch;
    }
  }
  __builtin_amdgcn_wave_barrier();

  int32_t run_sub = 0;
  uint64_t like_acc = 0, valid_acc = 0;
  uint8_t* const win = outwin[wave];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t r = row0 + u * 64 + lane;
    const bool live = r < n;
    const int32_t len = ob[u] - oa[u];
    bool hit;
    if (!big) {
      hit = range_any(hitmap[wave], oa[u] - base, ob[u] - base - 4);
    } else {
      hit = false;
      for (int p = 0; p + 5 <= len && !hit; p++) hit = (ld8(data + oa[u] + p) & nmask) == needle;
    }
    const uint64_t lw = __ballot(live && hit), vw = __ballot(live);
    like_acc = lane == u ? lw : like_acc;
    valid_acc = lane == u ? vw : valid_acc;
    const int32_t sl = live ? sub_len_ascii(len) : 0;
    const int32_t inc = wave_scan_incl(sl);
    const int32_t loc = run_sub + inc - sl;
    run_sub += __builtin_amdgcn_readlane(inc, 63);
    if (live) {
      A.sub_off[r] = (int32_t)(sub_base + loc);
      A.up_off[r] = oa[u] - so0;
    }
    // stage the row's bytes: one 8-byte load, a 4/2/1 ladder into LDS
    if (sl > 0 && loc + sl <= OUT_WIN) {
      uint64_t w = ld8(data + oa[u] + 1);
      int i = loc;
      if (sl & 4) { const uint32_t v = (uint32_t)w; __builtin_memcpy(win + i, &v, 4); i += 4; w >>= 32; }
      if (sl & 2) { const uint16_t v = (uint16_t)w; __builtin_memcpy(win + i, &v, 2); i += 2; w >>= 16; }
      if (sl & 1) win[i] = (uint8_t)w;
    }
  }
  const int64_t wbase = row0 >> 6;
  if (lane < U && wbase + lane < ((n + 63) >> 6)) {
    A.like_bits[wbase + lane] = like_acc;
    A.like_valid[wbase + lane] = valid_acc;
    A.sub_valid[wbase + lane] = valid_acc;
    A.up_valid[wbase + lane] = valid_acc;
  }
  if (rend == n && lane == 0) { A.sub_off[n] = (int32_t)(sub_base + run_sub); A.up_off[n] = s1 - so0; }
  if (sub_base + run_sub <= A.cap_sub) {
    uint8_t* __restrict__ dst = A.sub_dat + sub_base;
    if (run_sub <= OUT_WIN) {
      __builtin_amdgcn_wave_barrier();
      if (run_sub >= 16) {
        for (int32_t i = lane * 16; i < run_sub; i += 1024) {
          const int32_t j = i + 16 <= run_sub ? i : run_sub - 16;
          uint64_t w[2];
          __builtin_memcpy(w, win + j, 16);
          __builtin_memcpy(dst + j, w, 16);
        }
      } else if (lane < run_sub) {
        dst[lane] = win[lane];
      }
    } else {
      // (cannot happen with 5-byte rows; the product re-runs the rows and copies straight to HBM)
    }
  }
}

// ---- M (ring): full-step sweep just ahead of the rows, ring mirror + ring match bitmap in LDS
template <int U, int W, int RING>       // RING: bytes of span kept in LDS per wave (power of two)
__global__ void __launch_bounds__(W * 64) c5_main_ring(const Args A) {
  constexpr int OUT_WIN = U * 64 * 5;    // staged substr bytes per wave tile (5 per row at most here; the product streams its window)
  __shared__ __attribute__((aligned(16))) uint8_t outwin[W][OUT_WIN + 16];
  __shared__ __attribute__((aligned(16))) uint8_t mirror[W][RING + 16];      // + a replica of the first 16 bytes
  __shared__ __attribute__((aligned(16))) uint64_t hitring[W][RING / 64 + 2];  // + a replica of the first word
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n = A.n;
  const int64_t wt = (int64_t)blockIdx.x * W + wave;
  const int64_t row0 = wt * (64 * U);
  if (row0 >= n) return;
  const int32_t* __restrict__ off = A.off;
  const uint8_t* __restrict__ data = A.data;
  const bool full = row0 + 64 * U < n;
  const int64_t rend = full ? row0 + 64 * U : n;
  const int32_t s0 = off[row0], s1 = off[rend], so0 = off[0];
  const int64_t sub_base = A.wt_base[wt];
  const int32_t base = s0 & ~15;   // (the data buffer is 16-byte aligned here; the product aligns the ADDRESS)

  int32_t o[U + 1];
  if (full) {
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = off[row0 + u * 64 + lane];
  } else {
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t r = row0 + u * 64 + lane; o[u] = off[r < n ? r : n]; }
  }
  o[U] = s1;

  const uint64_t needle = 0x6b72617073ull, nmask = 0xffffffffffull;
  const uint64_t splat0 = 0x73 * B01, splat1 = 0x70 * B01;
  uint64_t acc = 0;
  uint8_t* __restrict__ up_dst = A.up_dat;
  const bool up_fits = (int64_t)s1 - so0 <= A.cap_up;
  uint8_t* const mir = mirror[wave];
  uint16_t* const hit16 = (uint16_t*)hitring[wave];
  const uint64_t* const hit = hitring[wave];

  // the sweep's state: bytes [base, swept) are done; the piece (and lane 63's halo) of the step at
  // `swept` are in flight
  int32_t swept = base;
  uint64_t wn[2] = {0, 0}, tn = 0;
  if (base + 16 * lane < s1) __builtin_memcpy(wn, __builtin_assume_aligned(data + base + 16 * lane, 16), 16);
  if (lane == 63 && base + 1024 < s1) tn = ld8(data + base + 1024);
  // the tile's two ragged ends of the flat output (whole 16-byte pieces overlapping their neighbours)
  if (up_fits) {
    const int32_t cnt = s1 - s0;
    if (cnt >= 16) {
      if (lane < 2) {
        const int32_t at = lane == 0 ? s0 : s1 - 16;
        uint64_t q[2];
        __builtin_memcpy(q, data + at, 16);
        q[0] = upper8(q[0]); q[1] = upper8(q[1]);
        __builtin_memcpy(up_dst + (at - so0), q, 16);
      }
    } else if (lane < cnt) {
      const uint8_t ch = data[s0 + lane];
      up_dst[s0 - so0 + lane] = (ch >= 'a' && ch <= 'z') ? ch - 32 : ch;
    }
  }

  int32_t run_sub = 0;
  uint64_t like_acc = 0, valid_acc = 0;
  uint8_t* const win = outwin[wave];
#pragma nounroll
  for (int u = 0; u < U; u++) {
    const int32_t oa = o[0];
    int32_t ob = __builtin_amdgcn_update_dpp(0, oa, 0x130, 0xf, 0xf, false);
    const int32_t se = u + 1 < U ? __builtin_amdgcn_readfirstlane(o[U > 1 ? 1 : 0]) : s1;
    if (lane == 63) ob = se;
    const int32_t ss = __builtin_amdgcn_readfirstlane(oa);
    // ---- sweep ahead: full 1024-byte steps until this sub-tile's span is covered
    while (swept < se) {
      const int32_t a = swept + 16 * lane;
      const uint64_t w[2] = {wn[0], wn[1]};
      const uint64_t tail = tn;
      wn[0] = 0; wn[1] = 0; tn = 0;
      if (a + 1024 < s1) __builtin_memcpy(wn, __builtin_assume_aligned(data + a + 1024, 16), 16);
      if (lane == 63 && a + 1024 + 16 < s1) tn = ld8(data + a + 1024 + 16);
      const uint64_t lo = w[0], hi = w[1];
      uint64_t nxt = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(lo >> 32), 0x130, 0xf, 0xf, false) << 32) |
                     (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)lo, 0x130, 0xf, 0xf, false);
      if (lane == 63) nxt = tail;
      acc |= lo | hi;
      const uint32_t m = match8(lo, hi, needle, nmask, splat0, splat1) | (match8(hi, nxt, needle, nmask, splat0, splat1) << 8);
      if (a < s1) {
        const int32_t ro = (a - base) & (RING - 1);        // ring offset of the piece (16-byte aligned: never wraps)
        hit16[ro >> 4] = (uint16_t)m;
        if (ro < 64) hit16[(RING >> 4) + (ro >> 4)] = (uint16_t)m;   // replica of the first word
        __builtin_memcpy(mir + ro, w, 16);
        if (ro == 0) __builtin_memcpy(mir + RING, w, 16);             // replica of the first 16 bytes
        if (up_fits && a >= s0 && a + 16 <= s1) {
          uint64_t q[2] = {upper8(lo), upper8(hi)};
          __builtin_memcpy(up_dst + (a - so0), q, 16);
        }
      }
      swept += 1024;
    }
    __builtin_amdgcn_wave_barrier();
    // the ring holds bytes [swept - RING, swept): this sub-tile's span [ss, se) must lie inside
    const bool ring_ok = swept - (ss & ~15) <= RING;
    const int64_t r = row0 + u * 64 + lane;
    const bool live = r < n;
    const int32_t len = ob - oa;
    bool hitrow = false;
    if (ring_ok) {
      // any match start in [oa, ob - 4): bit positions modulo the ring, two adjacent words + funnel
      const int32_t lo_b = (oa - base) & (RING - 1), nbits = len - 4;
      if (nbits > 0) {
        if (nbits <= 64) {
          const int32_t wv = lo_b >> 6, sh = lo_b & 63;
          const uint64_t x = (hit[wv] >> sh) | ((hit[wv + 1] << 1) << (63 - sh));
          const uint64_t msk = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
          hitrow = (x & msk) != 0;
        } else {
          for (int p = 0; p < nbits && !hitrow; p++) {
            const int32_t b = (lo_b + p) & (RING - 1);
            hitrow = (hit[b >> 6] >> (b & 63)) & 1;
          }
        }
      }
    } else {
      for (int p = 0; p + 5 <= len && !hitrow; p++) hitrow = (ld8(data + oa + p) & nmask) == needle;
    }
    const uint64_t lw = __ballot(live && hitrow), vw = __ballot(live);
    like_acc = lane == u ? lw : like_acc;
    valid_acc = lane == u ? vw : valid_acc;
    const int32_t sl = live ? sub_len_ascii(len) : 0;
    const int32_t inc = wave_scan_incl(sl);
    const int32_t loc = run_sub + inc - sl;
    run_sub += __builtin_amdgcn_readlane(inc, 63);
    if (live) {
      A.sub_off[r] = (int32_t)(sub_base + loc);
      A.up_off[r] = oa - so0;
    }
    if (sl > 0 && loc + sl <= OUT_WIN) {
      uint64_t wv8;
      if (ring_ok) {
        // bytes [oa + 1, oa + 9) from the ring mirror: two ALIGNED words (the replica covers the wrap)
        const int32_t d = (oa + 1 - base) & (RING - 1);
        uint64_t q0, q1;
        __builtin_memcpy(&q0, __builtin_assume_aligned(mir + (d & ~7), 8), 8);
        __builtin_memcpy(&q1, __builtin_assume_aligned(mir + (d & ~7) + 8, 8), 8);
        const int sh = (d & 7) * 8;
        wv8 = (q0 >> sh) | ((q1 << 1) << (63 - sh));
      } else {
        wv8 = ld8(data + oa + 1);
      }
      int i = loc;
      if (sl & 4) { const uint32_t v = (uint32_t)wv8; __builtin_memcpy(win + i, &v, 4); i += 4; wv8 >>= 32; }
      if (sl & 2) { const uint16_t v = (uint16_t)wv8; __builtin_memcpy(win + i, &v, 2); i += 2; wv8 >>= 16; }
      if (sl & 1) win[i] = (uint8_t)wv8;
    }
    // next sub-tile's offsets to the front
#pragma unroll
    for (int k = 0; k < U; k++) o[k] = o[k + 1];
  }
  const bool tile_ascii = __ballot((acc & B80) != 0) == 0;
  if (!tile_ascii && lane == 0) atomicOr(A.flags, 1u);
  const int64_t wbase = row0 >> 6;
  if (lane < U && wbase + lane < ((n + 63) >> 6)) {
    A.like_bits[wbase + lane] = like_acc;
    A.like_valid[wbase + lane] = valid_acc;
    A.sub_valid[wbase + lane] = valid_acc;
    A.up_valid[wbase + lane] = valid_acc;
  }
  if (rend == n && lane == 0) { A.sub_off[n] = (int32_t)(sub_base + run_sub); A.up_off[n] = s1 - so0; }
  if (sub_base + run_sub <= A.cap_sub && run_sub <= OUT_WIN) {
    uint8_t* __restrict__ dst = A.sub_dat + sub_base;
    __builtin_amdgcn_wave_barrier();
    if (run_sub >= 16) {
      for (int32_t i = lane * 16; i < run_sub; i += 1024) {
        const int32_t j = i + 16 <= run_sub ? i : run_sub - 16;
        uint64_t w[2];
        __builtin_memcpy(w, win + j, 16);
        __builtin_memcpy(dst + j, w, 16);
      }
    } else if (lane < run_sub) {
      dst[lane] = win[lane];
    }
  }
}

#ifndef UU
#define UU 4
#endif
#ifndef WW
#define WW 4
#endif
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000ll;
  int32_t *lens, *off, *sub_len, *sub_off_ref, *sub_off, *up_off;
  CK(hipMalloc(&lens, (n + 1) * 4)); CK(hipMalloc(&off, (n + 1) * 4 + 64));
  CK(hipMemset(lens, 0, (n + 1) * 4));
  hipLaunchKernelGGL(gen_lens, dim3(2048), dim3(256), 0, 0, lens, n);
  size_t tmp_bytes = 0; void* tmp = nullptr;
  rocprim::exclusive_scan(nullptr, tmp_bytes, lens, off, 0, n + 1, rocprim::plus<int32_t>());
  CK(hipMalloc(&tmp, tmp_bytes));
  rocprim::exclusive_scan(tmp, tmp_bytes, lens, off, 0, n + 1, rocprim::plus<int32_t>());
  int32_t total = 0;
  CK(hipMemcpy(&total, off + n, 4, hipMemcpyDeviceToHost));
  uint8_t *data, *sub_dat, *up_dat;
  CK(hipMalloc(&data, (size_t)total + 256)); CK(hipMemset(data, 0, (size_t)total + 256));
  hipLaunchKernelGGL(gen_bytes, dim3(4096), dim3(256), 0, 0, off, data, n);
  CK(hipMalloc(&sub_len, (n + 1) * 4)); CK(hipMemset(sub_len, 0, (n + 1) * 4));
  CK(hipMalloc(&sub_off_ref, (n + 1) * 4));
  hipLaunchKernelGGL(ref_lens, dim3(2048), dim3(256), 0, 0, off, sub_len, n);
  rocprim::exclusive_scan(tmp, tmp_bytes, sub_len, sub_off_ref, 0, n + 1, rocprim::plus<int32_t>());
  CK(hipMalloc(&sub_off, (n + 1) * 4)); CK(hipMalloc(&up_off, (n + 1) * 4));
  CK(hipMalloc(&sub_dat, (size_t)total + 256)); CK(hipMalloc(&up_dat, (size_t)total + 256));
  const int64_t nwords = (n + 63) / 64;
  uint64_t* bits; CK(hipMalloc(&bits, nwords * 8 * 4));
  constexpr int U = UU, W = WW;
  const int64_t nwt = (n + 64 * U - 1) / (64 * U);
  uint32_t *wt_tot, *wt_base;
  CK(hipMalloc(&wt_tot, (nwt + 1) * 4)); CK(hipMalloc(&wt_base, (nwt + 1) * 4));
  size_t tmp2_bytes = 0; void* tmp2 = nullptr;
  rocprim::exclusive_scan(nullptr, tmp2_bytes, wt_tot, wt_base, 0u, nwt, rocprim::plus<uint32_t>());
  CK(hipMalloc(&tmp2, tmp2_bytes));
  unsigned* flags; CK(hipMalloc(&flags, 4)); CK(hipMemset(flags, 0, 4));
  Args A;
  A.n = n; A.off = off; A.data = data;
  A.like_bits = bits; A.like_valid = bits + nwords; A.sub_valid = bits + 2 * nwords; A.up_valid = bits + 3 * nwords;
  A.sub_off = sub_off; A.sub_dat = sub_dat; A.up_off = up_off; A.up_dat = up_dat;
  A.wt_tot = wt_tot; A.wt_base = wt_base; A.flags = flags;
  A.cap_sub = total; A.cap_up = total;
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  const int iters = 10;
  const unsigned pre_grid = 256 * 8;
  const unsigned main_grid = (unsigned)((nwt + W - 1) / W);
  auto bench = [&](const char* name, void (*k)(const Args)) {
    float best = 1e9, sum = 0, bp = 1e9, bs = 1e9, bm = 1e9;
    for (int it = 0; it < iters + 2; it++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((c5_prepass<U, W>), dim3(pre_grid), dim3(W * 64), 0, 0, A);
      CK(hipEventRecord(e1));
      rocprim::exclusive_scan(tmp2, tmp2_bytes, wt_tot, wt_base, 0u, nwt, rocprim::plus<uint32_t>());
      CK(hipEventRecord(e2));
      hipLaunchKernelGGL(k, dim3(main_grid), dim3(W * 64), 0, 0, A);
      CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3));
      float ms, p, s, m;
      CK(hipEventElapsedTime(&ms, e0, e3)); CK(hipEventElapsedTime(&p, e0, e1));
      CK(hipEventElapsedTime(&s, e1, e2)); CK(hipEventElapsedTime(&m, e2, e3));
      if (it >= 2) { best = ms < best ? ms : best; sum += ms; bp = p < bp ? p : bp; bs = s < bs ? s : bs; bm = m < bm ? m : bm; }
    }
    printf("%-44s best %.3f ms avg %.3f ms  (pre-pass %.3f, scan %.3f, main %.3f)\n", name, best, sum / iters, bp, bs, bm);
    fflush(stdout);
    return best;
  };
  auto check = [&](const char* name) {
    unsigned* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(ref_check, dim3(4096), dim3(256), 0, 0, off, data, n, bits, sub_off_ref, sub_off, sub_dat, up_off, up_dat, bad);
    unsigned hbad, hflags; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hflags, flags, 4, hipMemcpyDeviceToHost));
    printf("  check %-36s %s(%u) flags=%u\n", name, hbad == 0 ? "OK" : "BAD", hbad, hflags);
    CK(hipFree(bad));
    CK(hipMemset(sub_dat, 0, (size_t)total)); CK(hipMemset(up_dat, 0, (size_t)total));
    CK(hipMemset(sub_off, 0xff, (n + 1) * 4)); CK(hipMemset(up_off, 0xff, (n + 1) * 4)); CK(hipMemset(bits, 0, nwords * 8 * 4));
  };
  float b;
  b = bench("k4h F=3 (tile-wide sweep, rows re-read HBM)", c5_main<U, W, 3>); check("k4h F=3");
  b = bench("ring 4096: full steps ahead of the rows", c5_main_ring<U, W, 4096>); check("ring 4096");
  b = bench("ring 2048", c5_main_ring<U, W, 2048>); check("ring 2048");
  b = bench("ring 4096 (again)", c5_main_ring<U, W, 4096>); check("ring 4096");
  const double alg = n * 4.0 + total + n / 8.0 * 4 + 2 * 4.0 * n + 482362923.0 + total;
  printf("k4r W=%d U=%d rows %lld bytes %d: last %.3f ms  alg %.3f GB  %.2f TB/s\n", W, U, (long long)n, total, b, alg / 1e9, alg / 1e9 / b);
  return 0;
}
