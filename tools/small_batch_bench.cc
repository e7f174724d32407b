// Small-batch latency of the C ABI itself (no Python in the loop): the reference is fed 4K-64K-row
// batches by its query engine, where per-call overhead, not HBM, sets the rate.
// HBM-resident C2-shaped batches (4 float64 columns with validity, 10 expressions):
//   sync      gdv_projector_evaluate, one batch per call, the call waits
//   async     gdv_projector_evaluate + GDV_EVAL_ASYNC on one stream, one wait per 64 batches
//   many      gdv_projector_evaluate_many, 64 batches per call (ONE launch), the call waits
// and the filter (a > k1 AND b < k2 over int64): sync vs gdv_filter_evaluate_async.
// Prints microseconds per batch.   g++ -O2 -std=c++17 tools/small_batch_bench.cc -Iinclude
//   -Lgandiva_amd -lgandiva_amd -Wl,-rpath,$PWD/gandiva_amd -o tools/small_batch_bench
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gandiva_amd.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d %s [%s]\n", __FILE__, __LINE__, #c, gdv_last_error()); exit(1); } } while (0)

static double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const gdv_type_t f64 = {GDV_TYPE_DOUBLE, 0, 0}, i64 = {GDV_TYPE_INT64, 0, 0}, boolean = {GDV_TYPE_BOOL, 0, 0};
  gdv_schema_t* schema = gdv_schema_new();
  const char* names[4] = {"a", "b", "c", "d"};
  gdv_node_t* f[4];
  for (int k = 0; k < 4; k++) {
    CHECK(gdv_schema_add_field(schema, names[k], f64, 1) == GDV_OK);
    f[k] = gdv_node_field(names[k], f64);
  }
  auto fn = [&](const char* name, gdv_node_t* x, gdv_node_t* y) {
    gdv_node_t* args[2] = {x, y};
    return gdv_node_function(name, args, 2, f64);
  };
  gdv_node_t *a = f[0], *b = f[1], *c = f[2], *d = f[3];
  gdv_node_t* roots[10] = {fn("add", a, b), fn("subtract", a, b), fn("multiply", a, b), fn("add", c, d), fn("multiply", c, d),
                           fn("multiply", fn("add", a, b), c), fn("multiply", fn("subtract", a, b), d),
                           fn("add", fn("multiply", a, b), fn("multiply", c, d)),
                           fn("multiply", fn("add", a, b), fn("subtract", c, d)),
                           fn("multiply", fn("multiply", fn("multiply", a, b), c), d)};
  gdv_expression_t* exprs[10];
  for (int e = 0; e < 10; e++) {
    char nm[8];
    snprintf(nm, sizeof(nm), "e%d", e);
    exprs[e] = gdv_expression_new(roots[e], nm, f64);
  }
  gdv_projector_t* proj = nullptr;
  CHECK(gdv_projector_make(schema, exprs, 10, GDV_SEL_NONE, nullptr, &proj) == GDV_OK);

  gdv_schema_t* fschema = gdv_schema_new();
  CHECK(gdv_schema_add_field(fschema, "a", i64, 1) == GDV_OK);
  CHECK(gdv_schema_add_field(fschema, "b", i64, 1) == GDV_OK);
  const int64_t k1 = 499, k2 = 250;
  gdv_node_t* ga[2] = {gdv_node_field("a", i64), gdv_node_literal(i64, &k1, 0)};
  gdv_node_t* gb[2] = {gdv_node_field("b", i64), gdv_node_literal(i64, &k2, 0)};
  gdv_node_t* conj[2] = {gdv_node_function("greater_than", ga, 2, boolean), gdv_node_function("less_than", gb, 2, boolean)};
  gdv_filter_t* flt = nullptr;
  CHECK(gdv_filter_make(fschema, gdv_condition_new(gdv_node_and(conj, 2)), nullptr, &flt) == GDV_OK);

  const int NB = 64;
  printf("%8s %10s %10s %10s   %12s %12s %12s   %10s %10s   (microseconds per batch, %d batches per round; host+reg = host buffers in registered memory)\n", "rows", "proj sync",
         "proj async", "proj many", "filter sync", "filter async", "filter many", "proj host", "host+reg", NB);
  for (int64_t rows : {1024, 4096, 16384, 65536, 262144}) {
    const int64_t vbytes = ((rows + 63) / 64) * 8;
    // NB independent batches: inputs + outputs in HBM
    std::vector<std::vector<gdv_column_t>> cols(NB, std::vector<gdv_column_t>(4));
    std::vector<std::vector<gdv_out_column_t>> outs(NB, std::vector<gdv_out_column_t>(10));
    std::vector<gdv_batch_t> batches(NB);
    std::vector<double> host(rows);
    for (int64_t i = 0; i < rows; i++) host[i] = (double)(i % 1000) * 0.25 - 100.0;
    std::vector<uint8_t> bits(vbytes, 0xEF);
    for (int bch = 0; bch < NB; bch++) {
      for (int k = 0; k < 4; k++) {
        memset(&cols[bch][k], 0, sizeof(gdv_column_t));
        void *dv, *dd;
        CHECK(gdv_device_alloc(vbytes, &dv) == GDV_OK);
        CHECK(gdv_device_alloc(rows * 8, &dd) == GDV_OK);
        CHECK(gdv_memcpy_h2d(dv, bits.data(), vbytes) == GDV_OK);
        CHECK(gdv_memcpy_h2d(dd, host.data(), rows * 8) == GDV_OK);
        cols[bch][k].validity = dv; cols[bch][k].validity_size = vbytes;
        cols[bch][k].data = dd; cols[bch][k].data_size = rows * 8;
      }
      for (int e = 0; e < 10; e++) {
        memset(&outs[bch][e], 0, sizeof(gdv_out_column_t));
        void *dv, *dd;
        CHECK(gdv_device_alloc(vbytes, &dv) == GDV_OK);
        CHECK(gdv_device_alloc(rows * 8, &dd) == GDV_OK);
        outs[bch][e].validity = dv; outs[bch][e].validity_size = vbytes;
        outs[bch][e].data = dd; outs[bch][e].data_size = rows * 8;
      }
      batches[bch] = gdv_batch_t{rows, cols[bch].data(), 4, outs[bch].data(), 10};
    }
    auto time_rounds = [&](auto&& round) {
      round();  // warm-up
      CHECK(gdv_device_synchronize() == GDV_OK);
      double best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        const double t0 = Now();
        round();
        CHECK(gdv_device_synchronize() == GDV_OK);
        best = std::min(best, Now() - t0);
      }
      return best / NB * 1e6;
    };
    const double p_sync = time_rounds([&] {
      for (int bch = 0; bch < NB; bch++)
        CHECK(gdv_projector_evaluate(proj, rows, cols[bch].data(), 4, nullptr, outs[bch].data(), 10, GDV_MEM_DEVICE, nullptr, 0) == GDV_OK);
    });
    const double p_async = time_rounds([&] {
      for (int bch = 0; bch < NB; bch++)
        CHECK(gdv_projector_evaluate(proj, rows, cols[bch].data(), 4, nullptr, outs[bch].data(), 10, GDV_MEM_DEVICE, nullptr, GDV_EVAL_ASYNC) == GDV_OK);
    });
    const double p_many = time_rounds([&] { CHECK(gdv_projector_evaluate_many(proj, batches.data(), NB, nullptr, 0) == GDV_OK); });
    // filter over the first two columns reinterpreted as int64 (the bit patterns do not matter for timing)
    void* idx;
    CHECK(gdv_device_alloc(rows * 4 + 64, &idx) == GDV_OK);
    void* cnt;
    CHECK(gdv_device_alloc(64, &cnt) == GDV_OK);
    const double f_sync = time_rounds([&] {
      int64_t count = 0;
      for (int bch = 0; bch < NB; bch++)
        CHECK(gdv_filter_evaluate(flt, rows, cols[bch].data(), 2, GDV_SEL_UINT32, idx, rows, &count, GDV_MEM_DEVICE, nullptr) == GDV_OK);
    });
    const double f_async = time_rounds([&] {
      for (int bch = 0; bch < NB; bch++)
        CHECK(gdv_filter_evaluate_async(flt, rows, cols[bch].data(), 2, GDV_SEL_UINT32, idx, rows, cnt, nullptr) == GDV_OK);
    });
    // host buffers in and out (what pyarrow / JNI callers pass): staged through HBM by the library
    std::vector<std::vector<double>> hout(10, std::vector<double>(rows));
    std::vector<std::vector<uint8_t>> hval(10, std::vector<uint8_t>((rows + 7) / 8 + 8));
    gdv_column_t hc[4];
    gdv_out_column_t ho[10];
    for (int k = 0; k < 4; k++) {
      memset(&hc[k], 0, sizeof(hc[k]));
      hc[k].validity = bits.data(); hc[k].validity_size = (rows + 7) / 8;
      hc[k].data = host.data(); hc[k].data_size = rows * 8;
    }
    for (int e = 0; e < 10; e++) {
      memset(&ho[e], 0, sizeof(ho[e]));
      ho[e].validity = hval[e].data(); ho[e].validity_size = (rows + 7) / 8;
      ho[e].data = hout[e].data(); ho[e].data_size = rows * 8;
    }
    const double p_host = time_rounds([&] {
      for (int bch = 0; bch < NB; bch++)
        CHECK(gdv_projector_evaluate(proj, rows, hc, 4, nullptr, ho, 10, GDV_MEM_HOST, nullptr, 0) == GDV_OK);
    });
    // the same call with every buffer in page-locked memory of the library (gdv_host_alloc): the kernel reads and
    // writes the host buffers in place, nothing is staged
    double p_reg = 0;
    {
      const int64_t vb = ((rows + 63) / 64) * 8 + 64, db = rows * 8 + 64;
      void* block = nullptr;
      CHECK(gdv_host_alloc(4 * (vb + db) + 10 * (vb + db), &block) == GDV_OK);
      char* at = (char*)block;
      gdv_column_t rc[4];
      gdv_out_column_t ro[10];
      for (int k = 0; k < 4; k++) {
        memset(&rc[k], 0, sizeof(rc[k]));
        memcpy(at, bits.data(), (rows + 7) / 8); rc[k].validity = at; rc[k].validity_size = vb - 64; at += vb;
        memcpy(at, host.data(), rows * 8); rc[k].data = at; rc[k].data_size = rows * 8; at += db;
      }
      for (int e = 0; e < 10; e++) {
        memset(&ro[e], 0, sizeof(ro[e]));
        ro[e].validity = at; ro[e].validity_size = vb - 64; at += vb;
        ro[e].data = at; ro[e].data_size = rows * 8; at += db;
      }
      const long long staged0 = gdv_host_staged_bytes();
      p_reg = time_rounds([&] {
        for (int bch = 0; bch < NB; bch++)
          CHECK(gdv_projector_evaluate(proj, rows, rc, 4, nullptr, ro, 10, GDV_MEM_HOST, nullptr, 0) == GDV_OK);
      });
      if (gdv_host_staged_bytes() != staged0) { fprintf(stderr, "registered buffers were staged\n"); return 1; }
      if (memcmp(ro[3].data, hout[3].data(), rows * 8) != 0) { fprintf(stderr, "registered path: output differs\n"); return 1; }
      CHECK(gdv_host_free(block) == GDV_OK);
    }
    std::vector<void*> idxs(NB);
    std::vector<gdv_filter_batch_t> fb(NB);
    for (int bch = 0; bch < NB; bch++) {
      CHECK(gdv_device_alloc(rows * 4 + 64, &idxs[bch]) == GDV_OK);
      fb[bch] = gdv_filter_batch_t{rows, cols[bch].data(), 2, idxs[bch], rows};
    }
    std::vector<int64_t> counts(NB);
    const double f_many = time_rounds([&] {
      CHECK(gdv_filter_evaluate_many(flt, fb.data(), NB, GDV_SEL_UINT32, counts.data(), nullptr, nullptr, 0) == GDV_OK);
    });
    for (int bch = 0; bch < NB; bch++) gdv_device_free(idxs[bch]);
    printf("%8lld %10.1f %10.1f %10.1f   %12.1f %12.1f %12.1f   %10.1f %10.1f\n", (long long)rows, p_sync, p_async, p_many, f_sync, f_async, f_many, p_host, p_reg);
    fflush(stdout);
    for (int bch = 0; bch < NB; bch++) {
      for (int k = 0; k < 4; k++) { gdv_device_free((void*)cols[bch][k].validity); gdv_device_free((void*)cols[bch][k].data); }
      for (int e = 0; e < 10; e++) { gdv_device_free(outs[bch][e].validity); gdv_device_free(outs[bch][e].data); }
    }
    gdv_device_free(idx); gdv_device_free(cnt);
  }
  return 0;
}
