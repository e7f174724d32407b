// Test-only HOST build of the product's device function library (gandiva_amd/csrc/
// gdv_device_lib.hpp): the per-row functions are plain C++, so compiled for x86 with the
// GPU intrinsics stubbed they can be driven over dense random inputs on a CPU-only machine
// and compared with the oracle.  Wave-level helpers (DPP scans, ballots, LDS staging) are
// stubbed and NOT exercised here; they are covered by the GPU parity suite.
#include <cmath>
#include <cstdint>
#include <cstring>

#define GDV_HOST_BUILD 1
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
// round / truncate / ceil / floor (round 5): mode 0 round, 1 truncate, 2 ceil, 3 floor; k = fractional digits kept
void host_decimal_round(int mode, const void* xv, int xp, int xs, const int* k, int op, int os, void* out, long n) {
  const gdv_int128* x = (const gdv_int128*)xv;
  gdv_int128* r = (gdv_int128*)out;
  for (long i = 0; i < n; i++) {
    switch (mode) {
      case 0: r[i] = k ? round_decimal128_int32(x[i], xp, xs, k[i], op, os) : round_decimal128(x[i], xp, xs, op, os); break;
      case 1: r[i] = k ? truncate_decimal128_int32(x[i], xp, xs, k[i], op, os) : truncate_decimal128(x[i], xp, xs, op, os); break;
      case 2: r[i] = ceil_decimal128(x[i], xp, xs, op, os); break;
      default: r[i] = floor_decimal128(x[i], xp, xs, op, os); break;
    }
  }
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

// ---------------------------------------------------------------- utf8 / binary
// A column is (int32 offsets, bytes, byte count); the byte buffer must be readable up to
// max(size, 8) bytes (the engine guarantees the same).  Row views are built exactly as the
// generated kernels build them.
struct HostCol { const int* off; const unsigned char* data; long size; };
static gdv_str host_row(const HostCol& c, long i) {
  const gdv_uint8* lim = c.data + (c.size < 8 ? 8 : c.size);
  return gdv_make_str(c.data, c.off[i], c.off[i + 1], lim);
}
static gdv_str host_lit(const unsigned char* lit, int len) { return gdv_make_str(lit, 0, len, lit + len + 8); }

// predicates against a literal (readable 8 bytes past its end)
void host_str_pred_lit(int fn, const int* off, const unsigned char* data, long size, long n,
                       const unsigned char* lit, int litlen, int map, unsigned char* out) {
  HostCol c{off, data, size};
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    s.map = map;
    const gdv_str l = host_lit(lit, litlen);
    bool r = false;
    switch (fn) {
      case 0: r = gdv_like_contains(s, lit, litlen); break;
      case 1: r = gdv_like_prefix(s, lit, litlen); break;
      case 2: r = gdv_like_suffix(s, lit, litlen); break;
      case 3: r = gdv_like_equal(s, lit, litlen); break;
      case 4: r = equal_utf8_utf8(s, l); break;
      case 5: r = less_than_utf8_utf8(s, l); break;
      case 6: r = starts_with_utf8_utf8(s, l); break;
      case 7: r = ends_with_utf8_utf8(s, l); break;
      case 8: r = greater_than_or_equal_to_utf8_utf8(s, l); break;
      default: break;
    }
    out[i] = r;
  }
}
// predicates between two columns
void host_str_pred_col(int fn, const int* offa, const unsigned char* da, long sa, const int* offb,
                       const unsigned char* db, long sb, long n, unsigned char* out) {
  HostCol a{offa, da, sa}, b{offb, db, sb};
  for (long i = 0; i < n; i++) {
    const gdv_str x = host_row(a, i), y = host_row(b, i);
    out[i] = fn == 0 ? equal_utf8_utf8(x, y) : fn == 1 ? less_than_utf8_utf8(x, y)
           : fn == 2 ? starts_with_utf8_utf8(x, y) : ends_with_utf8_utf8(x, y);
  }
}
// integer-valued functions
void host_str_int(int fn, const int* off, const unsigned char* data, long size, long n,
                  const unsigned char* lit, int litlen, int map, long long* out) {
  HostCol c{off, data, size};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    s.map = map;
    switch (fn) {
      case 0: out[i] = octet_length_utf8(s); break;
      case 1: out[i] = char_length_utf8(s); break;
      case 2: out[i] = hash32_utf8(s, true); break;
      case 3: out[i] = hash64_utf8(s, true); break;
      case 4: out[i] = ascii_utf8(s); break;
      case 5: out[i] = locate_utf8_utf8(ctx, host_lit(lit, litlen), s); break;
      case 6: out[i] = gdv_str_is_ascii(s) ? 1 : 0; break;
      case 7: err = 0; out[i] = castINT_utf8(ctx, s); if (err) out[i] = (1LL << 40); break;     // marker: raised
      case 8: err = 0; out[i] = castBIGINT_utf8(ctx, s); if (err) out[i] = -1234567890123456789LL; break;
      default: out[i] = 0; break;
    }
  }
}
// view-producing functions, materialised with gdv_str_copy the way the byte pass does
long host_str_view(int fn, const int* off, const unsigned char* data, long size, long n, long long a,
                   long long b, int map, int* out_off, unsigned char* out_data) {
  HostCol c{off, data, size};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    gdv_str r = s;
    switch (fn) {
      case 0: r = s; break;
      case 1: r = substr_utf8_int64_int64(s, a, b); break;
      case 2: r = ltrim_utf8(s); break;
      case 3: r = rtrim_utf8(s); break;
      case 4: r = btrim_utf8(s); break;
      case 5: r = left_utf8_int32(s, (int)a); break;
      case 6: r = right_utf8_int32(s, (int)a); break;
      case 7: r = castVARCHAR_utf8_int64(ctx, s, a); break;
      case 8: r = substr_utf8_int64(s, a); break;
      default: break;
    }
    if (map == 1) r = upper_utf8(r);
    if (map == 2) r = lower_utf8(r);
    gdv_str_copy(out_data + at, r);
    at += r.len;
    out_off[i + 1] = (int)at;
  }
  return at;
}
// general LIKE matcher over a pattern compiled the way gdv_planner.cc compiles it
void host_str_like(const int* off, const unsigned char* data, long size, long n, const unsigned char* pbyte,
                   const unsigned char* pkind, int plen, int map, unsigned char* out) {
  HostCol c{off, data, size};
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    s.map = map;
    out[i] = gdv_like(s, pbyte, pkind, plen);
  }
}
// IN over a literal list (concatenated bytes readable 8 past the end, n+1 offsets)
void host_str_in(const int* off, const unsigned char* data, long size, long n, const unsigned char* bytes,
                 const int* loffs, int nlits, unsigned char* out) {
  HostCol c{off, data, size};
  for (long i = 0; i < n; i++) out[i] = gdv_in_strings(host_row(c, i), bytes, loffs, nlits);
}

// ---- registry tail (round 2): values that only the output copy can read, and the two-piece pads
// reverse(s) materialised by gdv_str_copy; `ascii` sets the tile-wide ASCII flag the byte sweep
// would have set (only ever pass 1 for all-ASCII buffers).  Returns the error bits.
unsigned host_str_reverse(const int* off, const unsigned char* data, long size, long n, int map, int ascii,
                          int* out_off, unsigned char* out_data) {
  HostCol c{off, data, size};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    if (ascii) s.flags |= GDV_STR_ASCII;
    if (map == 1) s = upper_utf8(s);
    if (map == 2) s = lower_utf8(s);
    const gdv_str r = reverse_utf8(ctx, s);
    gdv_str_copy(out_data + at, r);
    at += r.len;
    out_off[i + 1] = (int)at;
  }
  return err;
}
// initcap(s) (round 5), materialised by gdv_str_copy; map: 0 the column, 1 upper(column), 2 lower(column)
void host_str_initcap(const int* off, const unsigned char* data, long size, long n, int map, int* out_off, unsigned char* out_data) {
  HostCol c{off, data, size};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    if (map == 1) s = upper_utf8(s);
    if (map == 2) s = lower_utf8(s);
    const gdv_str r = initcap_utf8(s);
    gdv_str_copy(out_data + at, r);
    at += r.len;
    out_off[i + 1] = (int)at;
  }
}
// message digests (round 5): algo 0 SHA-256, 1 SHA-1, 2 MD5 over string rows (valid[i] = 0: NULL -> the empty message) ...
void host_str_digest(int algo, const int* off, const unsigned char* data, long size, long n, const unsigned char* valid, int map,
                     unsigned char* out /* n x 64 bytes */) {
  HostCol c{off, data, size};
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    if (map == 1) s = upper_utf8(s);
    const gdv_str r = algo == 0 ? hashSHA256_utf8(s, valid[i] != 0) : algo == 1 ? hashSHA1_utf8(s, valid[i] != 0) : hashMD5_utf8(s, valid[i] != 0);
    gdv_str_copy(out + 64 * i, r);
  }
}
// ... and over numbers (hashed as the 8 bytes of the double)
void host_f64_digest(int algo, const double* v, long n, const unsigned char* valid, unsigned char* out) {
  for (long i = 0; i < n; i++) {
    const gdv_str r = algo == 0 ? hashSHA256_float64(v[i], valid[i] != 0) : algo == 1 ? hashSHA1_float64(v[i], valid[i] != 0) : hashMD5_float64(v[i], valid[i] != 0);
    gdv_str_copy(out + 64 * i, r);
  }
}
// castVARCHAR(float64 / float32, n) (round 5): 32 bytes per row, out_len[i] = the length
int host_real_text(int is32, const void* v, long n, long cut, unsigned char* out, int* out_len) {
  unsigned err = 0;
  gdv_ctx ctx{&err};
  for (long i = 0; i < n; i++) {
    const gdv_str r = is32 ? castVARCHAR_float32_int64(ctx, ((const float*)v)[i], cut) : castVARCHAR_float64_int64(ctx, ((const double*)v)[i], cut);
    out_len[i] = r.len;
    if (r.len > 0) gdv_str_copy(out + 32 * i, r);
  }
  return (int)err;
}
// to_date(text, pattern) (round 5): ops = the planner's compiled pattern; out_valid[i] = parsed; returns the error bits
int host_parse_date(const int* off, const unsigned char* data, long size, long n, const unsigned char* ops, int nops, int suppress,
                    long long* out, unsigned char* out_valid) {
  HostCol c{off, data, size};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  for (long i = 0; i < n; i++) {
    bool ov = false;
    out[i] = gdv_parse_date(ctx, host_row(c, i), ops, nops, suppress, true, &ov);
    out_valid[i] = ov ? 1 : 0;
  }
  return (int)err;
}
void host_to_timestamp(int kind, const void* v, long n, long long* ts, int* tm) {
  for (long i = 0; i < n; i++) {
    if (kind == 0) { ts[i] = to_timestamp_int32(((const int*)v)[i]); tm[i] = to_time_int32(((const int*)v)[i]); }
    else if (kind == 1) { ts[i] = to_timestamp_int64(((const long long*)v)[i]); tm[i] = to_time_int64(((const long long*)v)[i]); }
    else if (kind == 2) { ts[i] = to_timestamp_float32(((const float*)v)[i]); tm[i] = to_time_float32(((const float*)v)[i]); }
    else { ts[i] = to_timestamp_float64(((const double*)v)[i]); tm[i] = to_time_float64(((const double*)v)[i]); }
  }
}
// replace / lpad / rpad with per-row arguments (round 5): three string columns (text, from | fill, to) and a length column
long host_replace_row(const int* off0, const unsigned char* d0, long s0, const int* off1, const unsigned char* d1, long s1, const int* off2,
                      const unsigned char* d2, long s2, long n, int text_map, int* out_off, unsigned char* out_data, int* err_out) {
  HostCol c0{off0, d0, s0}, c1{off1, d1, s1}, c2{off2, d2, s2};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c0, i);
    if (text_map == 1) s = upper_utf8(s);
    const gdv_str r = gdv_replace_row(ctx, s, host_row(c1, i), host_row(c2, i));
    if (r.len > 0) gdv_str_copy(out_data + at, r);
    at += r.len;
    out_off[i + 1] = (int)at;
  }
  *err_out = (int)err;
  return at;
}
long host_pad_row(int right, const int* off0, const unsigned char* d0, long s0, const int* want, const int* off1, const unsigned char* d1, long s1,
                  long n, int* out_off, unsigned char* out_data, int* err_out) {
  HostCol c0{off0, d0, s0}, c1{off1, d1, s1};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    const gdv_str s = host_row(c0, i);
    const gdv_str text = gdv_pad_text(s, want[i]), pad = gdv_pad_fill_row(ctx, s, want[i], host_row(c1, i));
    const gdv_str& first = right ? text : pad;
    const gdv_str& second = right ? pad : text;
    if (first.len > 0) gdv_str_copy(out_data + at, first);
    at += first.len;
    if (second.len > 0) gdv_str_copy(out_data + at, second);
    at += second.len;
    out_off[i + 1] = (int)at;
  }
  *err_out = (int)err;
  return at;
}
// regexp_like (round 5): `table` = what gdv_compile_regex lays out; map 1 = the text read through upper()
void host_regex_search(const int* off, const unsigned char* data, long size, long n, const unsigned char* table, int map, unsigned char* out) {
  HostCol c{off, data, size};
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    if (map == 1) s = upper_utf8(s);
    out[i] = gdv_regex_search(s, table) ? 1 : 0;
  }
}
// lpad (right = 0) / rpad (right = 1): `tab` is the fill repeated to `want` characters, readable
// 8 bytes past its end (what the planner lays out in the constant block)
long host_str_pad(int right, const int* off, const unsigned char* data, long size, long n, int want,
                  const unsigned char* tab, int tab_len, int tab_ascii, int* out_off, unsigned char* out_data) {
  HostCol c{off, data, size};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    const gdv_str s = host_row(c, i);
    const gdv_str text = gdv_pad_text(s, want), pad = gdv_pad_fill(s, want, tab, tab_len, tab_ascii != 0);
    const gdv_str& first = right ? text : pad;
    const gdv_str& second = right ? pad : text;
    if (first.len > 0) gdv_str_copy(out_data + at, first);
    at += first.len;
    if (second.len > 0) gdv_str_copy(out_data + at, second);
    at += second.len;
    out_off[i + 1] = (int)at;
  }
  return at;
}
// replace with a table laid out as the planner lays it out: int32 from_len, int32 to_len, 8 unused
// bytes, from, (16-byte aligned) to
// inbuf != 0: the data buffer is readable 8 bytes past `size` — rows carry GDV_STR_INBUF and take
// the word-at-a-time search (gdv_find_raw), as every tile but a batch's last few does on the device
unsigned host_str_replace(const int* off, const unsigned char* data, long size, long n, int map,
                          const unsigned char* table, int* out_off, unsigned char* out_data, int inbuf) {
  HostCol c{off, data, size};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    if (inbuf) s.flags |= GDV_STR_INBUF;
    if (map == 1) s = upper_utf8(s);
    if (map == 2) s = lower_utf8(s);
    const gdv_str r = gdv_replace(ctx, s, table);
    if (r.len > 0) gdv_str_copy(out_data + at, r);
    at += r.len;
    out_off[i + 1] = (int)at;
  }
  return err;
}
// replace answered by the byte sweep (round 3): the match bitmap of the whole column is built the way
// the generated kernels build it — 16-byte pieces, two gdv_match8 per piece, the next piece's first
// word as halo — then every row counts its bits (gdv_replace_hits) and is copied along them
// (gdv_copy_replaced_hits).  The data buffer must be readable 16 bytes past `size`.
unsigned host_str_replace_hits(const int* off, const unsigned char* data, long size, long n, int map,
                               const unsigned char* table, int* out_off, unsigned char* out_data) {
  HostCol c{off, data, size};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  const gdv_int32 fl = ((const gdv_int32*)table)[0];
  const gdv_uint64 mask = gdv_low_bytes_mask(fl);
  const gdv_uint64 nd = gdv_load8_raw(table + 16) & mask;
  const gdv_uint32 s0 = (gdv_uint32)(nd & 0xffull) * 0x01010101u, s1 = (gdv_uint32)((nd >> 8) & 0xffull) * 0x01010101u;
  const long pieces = (size + 15) / 16;
  gdv_uint64* bm = new gdv_uint64[pieces / 4 + 4]();
  for (long q = 0; q < pieces; q++) {
    const long a = q * 16;
    gdv_uint64 w[2] = {0, 0}, nxw = 0;
    std::memcpy(w, data + a, 16);                     // (readable: 16 bytes of padding behind the column)
    if (a + 16 < size) std::memcpy(&nxw, data + a + 16, 8);
    const gdv_uint64 lo = gdv_map8(w[0], map), hi = gdv_map8(w[1], map), nx = gdv_map8(nxw, map);
    const gdv_uint32 m = gdv_match8(lo, hi, nd, mask, s0, s1) | (gdv_match8(hi, nx, nd, mask, s0, s1) << 8);
    ((gdv_uint16*)bm)[q] = (gdv_uint16)m;
  }
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    s.flags |= GDV_STR_INBUF;
    if (map == 1) s = upper_utf8(s);
    if (map == 2) s = lower_utf8(s);
    const gdv_str r = gdv_replace_hits(ctx, s, table, bm, off[i]);
    if (r.len > 0) {
      if (r.map & GDV_MAP_HITS) gdv_copy_replaced_hits(out_data + at, r, gdv_rd_hbm{r.p}, bm, off[i]);
      else gdv_str_copy(out_data + at, r);
    }
    at += r.len;
    out_off[i + 1] = (int)at;
  }
  delete[] bm;
  return err;
}
// locate(needle, s, start) with the needle read through a view, rows through `map`
unsigned host_str_locate(const int* off, const unsigned char* data, long size, long n, const unsigned char* lit,
                         int litlen, int start, int map, int* out) {
  HostCol c{off, data, size};
  unsigned err = 0;
  gdv_ctx ctx{&err};
  for (long i = 0; i < n; i++) {
    gdv_str s = host_row(c, i);
    s.map = map;
    out[i] = locate_utf8_utf8_int32(ctx, host_lit(lit, litlen), s, start);
  }
  return err;
}
unsigned host_cast_varchar_int64(const long long* v, long n, long long len, int* out_off, unsigned char* out_data) {
  unsigned err = 0;
  gdv_ctx ctx{&err};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    const gdv_str r = castVARCHAR_int64_int64(ctx, v[i], len);
    gdv_str_copy(out_data + at, r);
    at += r.len;
    out_off[i + 1] = (int)at;
  }
  return err;
}
// castVARCHAR(decimal128(p, s), len): values are 16-byte little-endian
unsigned host_cast_varchar_decimal(const void* xv, int xp, int xs, long n, long long len, int* out_off, unsigned char* out_data) {
  const gdv_int128* x = static_cast<const gdv_int128*>(xv);
  unsigned err = 0;
  gdv_ctx ctx{&err};
  long at = 0;
  out_off[0] = 0;
  for (long i = 0; i < n; i++) {
    const gdv_str r = castVARCHAR_decimal128_int64(ctx, x[i], xp, xs, len, 0, 0);
    gdv_str_copy(out_data + at, r);
    at += r.len;
    out_off[i + 1] = (int)at;
  }
  return err;
}
// ---------------------------------------------------------------- numeric / date / hash scalars (round 3)
// op: 0 round_float64 (-> double), 1 truncate_float64 (-> double), 2 castBIGINT_float64 (-> int64),
//     3 castINT_float64 (-> int32 widened to int64)
void host_f64_op(int op, const double* x, long n, double* outd, long long* outi) {
  for (long i = 0; i < n; i++) {
    switch (op) {
      case 0: outd[i] = round_float64(x[i]); break;
      case 1: outd[i] = truncate_float64(x[i]); break;
      case 2: outi[i] = castBIGINT_float64(x[i]); break;
      default: outi[i] = castINT_float64(x[i]); break;
    }
  }
}
// op: 0 Year 1 Month 2 Day 3 Hour 4 Minute 5 Second 6 Doy 7 Dow 8 Quarter 9 Epoch 10 Decade 11 Century 12 Millennium
void host_extract_timestamp(int op, const long long* t, long n, long long* out) {
  for (long i = 0; i < n; i++) {
    switch (op) {
      case 0: out[i] = extractYear_timestamp(t[i]); break;
      case 1: out[i] = extractMonth_timestamp(t[i]); break;
      case 2: out[i] = extractDay_timestamp(t[i]); break;
      case 3: out[i] = extractHour_timestamp(t[i]); break;
      case 4: out[i] = extractMinute_timestamp(t[i]); break;
      case 5: out[i] = extractSecond_timestamp(t[i]); break;
      case 6: out[i] = extractDoy_timestamp(t[i]); break;
      case 7: out[i] = extractDow_timestamp(t[i]); break;
      case 8: out[i] = extractQuarter_timestamp(t[i]); break;
      case 9: out[i] = extractEpoch_timestamp(t[i]); break;
      case 10: out[i] = extractDecade_timestamp(t[i]); break;
      case 11: out[i] = extractCentury_timestamp(t[i]); break;
      default: out[i] = extractMillennium_timestamp(t[i]); break;
    }
  }
}
// op: 0..10 date_trunc_{Second, Minute, Hour, Day, Week, Month, Quarter, Year, Decade, Century, Millennium},
//     11 extractWeek, 12 last_day — over timestamps (ms)
void host_date_trunc_timestamp(int op, const long long* t, long n, long long* out) {
  for (long i = 0; i < n; i++) {
    switch (op) {
      case 0: out[i] = date_trunc_Second_timestamp(t[i]); break;
      case 1: out[i] = date_trunc_Minute_timestamp(t[i]); break;
      case 2: out[i] = date_trunc_Hour_timestamp(t[i]); break;
      case 3: out[i] = date_trunc_Day_timestamp(t[i]); break;
      case 4: out[i] = date_trunc_Week_timestamp(t[i]); break;
      case 5: out[i] = date_trunc_Month_timestamp(t[i]); break;
      case 6: out[i] = date_trunc_Quarter_timestamp(t[i]); break;
      case 7: out[i] = date_trunc_Year_timestamp(t[i]); break;
      case 8: out[i] = date_trunc_Decade_timestamp(t[i]); break;
      case 9: out[i] = date_trunc_Century_timestamp(t[i]); break;
      case 10: out[i] = date_trunc_Millennium_timestamp(t[i]); break;
      case 11: out[i] = extractWeek_timestamp(t[i]); break;
      default: out[i] = last_day_timestamp(t[i]); break;
    }
  }
}
// hash32 / hash64 of int64, float64 and utf8 values (valid rows; seed 0)
void host_hash_fixed(int is_f64, const void* v, long n, int* h32, long long* h64) {
  for (long i = 0; i < n; i++) {
    if (is_f64) {
      h32[i] = hash32_float64(((const double*)v)[i], true);
      h64[i] = hash64_float64(((const double*)v)[i], true);
    } else {
      h32[i] = hash32_int64(((const long long*)v)[i], true);
      h64[i] = hash64_int64(((const long long*)v)[i], true);
    }
  }
}
void host_hash_utf8(const int* off, const unsigned char* data, long size, long n, int* h32, long long* h64) {
  HostCol c{off, data, size};
  for (long i = 0; i < n; i++) {
    const gdv_str s = host_row(c, i);
    h32[i] = hash32_utf8(s, true);
    h64[i] = hash64_utf8(s, true);
  }
}
// castBIGINT / castINT of text (blanks trimmed, decimal or 0x.. hexadecimal); bad[i] != 0: the row raised
void host_parse_int(int wide, const int* off, const unsigned char* data, long size, long n, long long* out, unsigned char* bad) {
  HostCol c{off, data, size};
  for (long i = 0; i < n; i++) {
    unsigned err = 0;
    gdv_ctx ctx{&err};
    const gdv_str s = host_row(c, i);
    out[i] = wide ? castBIGINT_utf8(ctx, s) : (long long)castINT_utf8(ctx, s);
    bad[i] = err != 0;
  }
}
void host_months_between(const long long* s, const long long* e, long n, int unit, int* out) {
  for (long i = 0; i < n; i++)
    out[i] = unit == 0 ? timestampdiffMonth_timestamp_timestamp(s[i], e[i])
           : unit == 1 ? timestampdiffQuarter_timestamp_timestamp(s[i], e[i])
                       : timestampdiffYear_timestamp_timestamp(s[i], e[i]);
}

}  // extern "C"
