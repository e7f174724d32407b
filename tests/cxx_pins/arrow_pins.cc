// Test-only second pin for the decimal128 and calendar semantics (round-1 verdict item 7).
// The reference lineage builds its decimal functions on arrow::BasicDecimal128 / 256 and its
// date functions on the vendored Hinnant date library; both ship in this image with pyarrow
// (libarrow.so + headers).  This file restates the operators ONLY in terms of those
// primitives — scale up (IncreaseScaleBy), 256-bit multiply / divide, scale down with
// ReduceScaleBy(round = true), FitsInPrecision — and is compared with the oracle in
// tests/test_arrow_pins.py.  It shares no code with the oracle or the product.
#include <arrow/util/basic_decimal.h>
#include <arrow/util/decimal.h>
#include <arrow/vendored/datetime/date.h>
#include <arrow/type.h>
#include <arrow/util/formatting.h>
#include <arrow/util/value_parsing.h>

#include <cstdint>
#include <cstring>

using arrow::BasicDecimal128;
using arrow::BasicDecimal256;
namespace date = arrow_vendored::date;

static BasicDecimal256 Widen(const uint8_t* le16) {
  uint64_t lo, hi;
  std::memcpy(&lo, le16, 8);
  std::memcpy(&hi, le16 + 8, 8);
  const uint64_t ext = (hi >> 63) ? ~0ull : 0ull;
  return BasicDecimal256(std::array<uint64_t, 4>{lo, hi, ext, ext});  // little-endian words
}
static void Narrow(const BasicDecimal256& v, uint8_t* le16) {
  const auto& w = v.little_endian_array();
  std::memcpy(le16, &w[0], 8);
  std::memcpy(le16 + 8, &w[1], 8);
}

extern "C" {

// op: 0 add, 1 subtract, 2 multiply, 3 divide.  x, y, out: n 16-byte little-endian values.
// A result that does not fit 38 digits is 0 (the reference's overflow convention).
// Returns 1 if a divisor was zero (rows with a zero divisor are written as 0).
int pin_decimal_binary(int op, const uint8_t* x, int xs, const uint8_t* y, int ys, int os, uint8_t* out, long n) {
  int div_zero = 0;
  for (long i = 0; i < n; i++) {
    BasicDecimal256 a = Widen(x + 16 * i), b = Widen(y + 16 * i), r;
    if (op <= 1) {
      const int s = xs > ys ? xs : ys;
      a = a.IncreaseScaleBy(s - xs);
      b = b.IncreaseScaleBy(s - ys);
      r = op == 0 ? a + b : a + (-b);
      if (s > os) r = r.ReduceScaleBy(s - os, true);
      else if (s < os) r = r.IncreaseScaleBy(os - s);
    } else if (op == 2) {
      r = a * b;
      const int s = xs + ys;
      if (s > os) r = r.ReduceScaleBy(s - os, true);
      else if (s < os) r = r.IncreaseScaleBy(os - s);
    } else {
      if (b == BasicDecimal256(0)) {
        div_zero = 1;
        r = BasicDecimal256(0);
      } else {
        // x / y at scale os: (x * 10^(os + ys - xs)) / y, rounded half away from zero
        const int up = os + ys - xs;
        BasicDecimal256 num = up >= 0 ? a.IncreaseScaleBy(up) : a.ReduceScaleBy(-up, false);
        BasicDecimal256 q, rem;
        num.Divide(b, &q, &rem);
        BasicDecimal256 twice = BasicDecimal256::Abs(rem) + BasicDecimal256::Abs(rem);
        if (twice >= BasicDecimal256::Abs(b)) {
          const bool neg = (num < BasicDecimal256(0)) != (b < BasicDecimal256(0));
          q = neg ? q + BasicDecimal256(-1) : q + BasicDecimal256(1);
        }
        r = q;
      }
    }
    if (!r.FitsInPrecision(38)) r = BasicDecimal256(0);
    Narrow(r, out + 16 * i);
  }
  return div_zero;
}

// days since 1970-01-01 -> civil fields through date::year_month_day / weekday
void pin_civil(const int64_t* days, long n, int32_t* year, int32_t* month, int32_t* day, int32_t* doy,
               int32_t* dow_sunday1) {
  for (long i = 0; i < n; i++) {
    const date::sys_days sd{date::days{days[i]}};
    const date::year_month_day ymd{sd};
    year[i] = static_cast<int>(ymd.year());
    month[i] = static_cast<int>(static_cast<unsigned>(ymd.month()));
    day[i] = static_cast<int>(static_cast<unsigned>(ymd.day()));
    const date::sys_days jan1{ymd.year() / date::January / 1};
    doy[i] = static_cast<int>((sd - jan1).count()) + 1;
    dow_sunday1[i] = static_cast<int>(date::weekday{sd}.c_encoding()) + 1;  // Sunday = 1
  }
}

// millis since epoch + `months` calendar months; a day past the end of the target month is
// clamped to its last day (the SQL rule); the time of day is kept
void pin_add_months(const int64_t* ms, long n, int32_t months, int64_t* out) {
  for (long i = 0; i < n; i++) {
    const int64_t day_ms = 86400000;
    int64_t d = ms[i] / day_ms, tod = ms[i] % day_ms;
    if (tod < 0) { tod += day_ms; d -= 1; }
    const date::year_month_day ymd{date::sys_days{date::days{d}}};
    date::year_month_day t = ymd + date::months{months};
    if (!t.ok()) t = t.year() / t.month() / date::last;
    out[i] = static_cast<int64_t>(date::sys_days{t}.time_since_epoch().count()) * day_ms + tod;
  }
}

// ---- round 2: integer <-> text through the Arrow primitives the reference lineage is believed to
// call (gdv_function_stubs.cc: arrow::internal::StringFormatter for castVARCHAR(integer, n),
// arrow::internal::ParseValue on the blank-trimmed text for castINT / castBIGINT).
// out: 24 bytes per value, lens[i] = number of characters
void pin_format_int64(const long long* v, long n, char* out, int* lens) {
  arrow::internal::StringFormatter<arrow::Int64Type> fmt;
  for (long i = 0; i < n; i++) {
    char* dst = out + 24 * i;
    int* len = lens + i;
    (void)fmt(static_cast<int64_t>(v[i]), [&](std::string_view sv) {
      std::memcpy(dst, sv.data(), sv.size());
      *len = static_cast<int>(sv.size());
      return arrow::Status::OK();
    });
  }
}
// returns 1 and *out on success, 0 when the text is not an integer of that width
int pin_parse_int(const char* text, int len, int bits, long long* out) {
  while (len > 0 && *text == ' ') { text++; len--; }
  while (len > 0 && text[len - 1] == ' ') len--;
  if (bits == 32) {
    int32_t v = 0;
    if (!arrow::internal::ParseValue<arrow::Int32Type>(text, static_cast<size_t>(len), &v)) return 0;
    *out = v;
    return 1;
  }
  int64_t v = 0;
  if (!arrow::internal::ParseValue<arrow::Int64Type>(text, static_cast<size_t>(len), &v)) return 0;
  *out = v;
  return 1;
}

}  // extern "C"
