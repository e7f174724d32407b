// In-process multi-device evaluation ("one host thread + stream per device"), through
// the C ABI only.  ONE logical batch is row-sharded with gdv_shard_bounds over N device contexts,
// every shard evaluated by its own host thread on its own context (gdv_set_device), HBM-resident;
// the concatenation must equal the UNSHARDED evaluation of the same handles bit for bit:
//   C2 shape  float64 projection with nulls (validity words merged per 64 rows)
//   C3 shape  int64 filter -> uint32 selection vector (local indices + shard base, ascending)
//   C5 shape  utf8 like / substr / upper (var-len outputs: per-shard offsets rebased)
// With fewer physical GPUs than N the contexts are virtual (gdv_set_virtual_devices): N contexts
// share the GPUs round-robin — the same code path, which is how this runs on a one-GPU box; on a
// multi-GPU node the same binary uses real devices.
// Round 4: "equal to its own unsharded run" says nothing about the VALUES.  With a third argument the
// inputs and the unsharded results are written to that directory as raw Arrow buffers; the Python
// wrapper (tests/test_multi_device.py) rebuilds the batch and compares the results with the oracle —
// sharded == unsharded (checked here) and unsharded == oracle (checked there).
//   multi_device_test [N=2] [rows=300000] [dump_dir]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gandiva_amd.h"

#define CHECK(cond)                                                                           \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      fprintf(stderr, "FAILED %s:%d: %s  [%s]\n", __FILE__, __LINE__, #cond, gdv_last_error()); \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

namespace {

uint64_t Mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

const gdv_type_t kF64 = {GDV_TYPE_DOUBLE, 0, 0}, kI64 = {GDV_TYPE_INT64, 0, 0}, kBool = {GDV_TYPE_BOOL, 0, 0},
                 kStr = {GDV_TYPE_STRING, 0, 0};

struct HostColumn {
  std::vector<uint8_t> validity;  // empty: no nulls
  std::vector<uint8_t> data;
  std::vector<int32_t> offsets;   // var-len only
};

// device copy of rows [lo, hi) of a host column (lo a multiple of 1024: bitmaps slice at whole bytes)
struct DeviceColumn {
  void *validity = nullptr, *data = nullptr, *offsets = nullptr;
  gdv_column_t view{};
  void Upload(const HostColumn& h, int width, int64_t lo, int64_t hi) {
    const int64_t n = hi - lo;
    memset(&view, 0, sizeof(view));
    if (!h.validity.empty()) {
      const int64_t bytes = (n + 7) / 8;
      CHECK(gdv_device_alloc(bytes + 64, &validity) == GDV_OK);
      CHECK(gdv_memcpy_h2d(validity, h.validity.data() + lo / 8, bytes) == GDV_OK);
      view.validity = validity;
      view.validity_size = bytes;
    }
    if (!h.offsets.empty()) {
      // the shard keeps the whole byte buffer's numbering: offsets are copied as they are, the data
      // pointer stays at the start of the (shard-local copy of the) byte range they address
      const int32_t b0 = h.offsets[lo], b1 = h.offsets[hi];
      std::vector<int32_t> local(h.offsets.begin() + lo, h.offsets.begin() + hi + 1);
      for (auto& o : local) o -= b0;
      CHECK(gdv_device_alloc((n + 1) * 4 + 64, &offsets) == GDV_OK);
      CHECK(gdv_memcpy_h2d(offsets, local.data(), (n + 1) * 4) == GDV_OK);
      CHECK(gdv_device_alloc((b1 - b0) + 64, &data) == GDV_OK);
      CHECK(gdv_memcpy_h2d(data, h.data.data() + b0, b1 - b0) == GDV_OK);
      view.offsets = offsets;
      view.offsets_size = (n + 1) * 4;
      view.data = data;
      view.data_size = (b1 - b0) + 64;  // (padding is readable: the kernels' 8-byte loads may touch it)
    } else {
      CHECK(gdv_device_alloc(n * width + 64, &data) == GDV_OK);
      CHECK(gdv_memcpy_h2d(data, h.data.data() + lo * width, n * width) == GDV_OK);
      view.data = data;
      view.data_size = n * width;
    }
  }
  void Free() {
    gdv_device_free(validity); gdv_device_free(data); gdv_device_free(offsets);
  }
};

struct Output {  // one output column, host side
  std::vector<uint8_t> validity, data;
  std::vector<int32_t> offsets;
};

bool BitAt(const std::vector<uint8_t>& bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }

void DumpBytes(const std::string& dir, const std::string& name, const void* p, size_t n) {
  FILE* f = fopen((dir + "/" + name).c_str(), "wb");
  CHECK(f != nullptr);
  if (n > 0) CHECK(fwrite(p, 1, n, f) == n);
  CHECK(fclose(f) == 0);
}

}  // namespace

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2;
  const int64_t rows = argc > 2 ? atoll(argv[2]) : 300000;
  if (gdv_physical_device_count() < 1) {
    fprintf(stderr, "no HIP device\n");
    return 2;
  }
  if (gdv_physical_device_count() < N) CHECK(gdv_set_virtual_devices(N) == GDV_OK);
  CHECK(gdv_device_count() >= N);

  // ---- data: 2 float64 columns with ~10 % nulls, 2 int64 columns, 1 utf8 column
  HostColumn a, b, k1, k2, s;
  a.data.resize(rows * 8); b.data.resize(rows * 8); k1.data.resize(rows * 8); k2.data.resize(rows * 8);
  a.validity.assign((rows + 7) / 8 + 8, 0); b.validity.assign((rows + 7) / 8 + 8, 0);
  s.validity.assign((rows + 7) / 8 + 8, 0);
  s.offsets.resize(rows + 1);
  const char* letters = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ";
  std::string bytes;
  for (int64_t i = 0; i < rows; i++) {
    const double va = (double)(int64_t)(Mix(4 * i + 1) % 2000001) / 1000.0 - 1000.0;
    const double vb = (double)(int64_t)(Mix(4 * i + 2) % 2000001) / 777.0 - 1300.0;
    memcpy(&a.data[i * 8], &va, 8); memcpy(&b.data[i * 8], &vb, 8);
    const int64_t x = (int64_t)(Mix(4 * i + 3) % 1000), y = (int64_t)(Mix(4 * i + 4) % 1000);
    memcpy(&k1.data[i * 8], &x, 8); memcpy(&k2.data[i * 8], &y, 8);
    if (Mix(9 * i + 5) % 10 != 0) a.validity[i >> 3] |= 1u << (i & 7);
    if (Mix(9 * i + 6) % 10 != 0) b.validity[i >> 3] |= 1u << (i & 7);
    s.offsets[i] = (int32_t)bytes.size();
    const bool null_s = Mix(9 * i + 7) % 10 == 0;
    if (!null_s) {
      s.validity[i >> 3] |= 1u << (i & 7);
      const int len = 4 + (int)(Mix(3 * i + 11) % 17);
      for (int q = 0; q < len; q++) bytes.push_back(letters[Mix(i * 32 + q + 77) % 52]);
      if (Mix(3 * i + 12) % 20 == 0 && len >= 5) memcpy(&bytes[bytes.size() - len + Mix(i) % (len - 4)], "spark", 5);
    }
  }
  s.offsets[rows] = (int32_t)bytes.size();
  s.data.assign(bytes.begin(), bytes.end());
  s.data.resize(s.data.size() + 64);

  // ---- plans
  gdv_schema_t* schema = gdv_schema_new();
  CHECK(gdv_schema_add_field(schema, "a", kF64, 1) == GDV_OK);
  CHECK(gdv_schema_add_field(schema, "b", kF64, 1) == GDV_OK);
  CHECK(gdv_schema_add_field(schema, "k1", kI64, 1) == GDV_OK);
  CHECK(gdv_schema_add_field(schema, "k2", kI64, 1) == GDV_OK);
  CHECK(gdv_schema_add_field(schema, "s", kStr, 1) == GDV_OK);
  gdv_node_t *fa = gdv_node_field("a", kF64), *fb = gdv_node_field("b", kF64), *fk1 = gdv_node_field("k1", kI64),
             *fk2 = gdv_node_field("k2", kI64), *fs = gdv_node_field("s", kStr);
  auto fn2 = [](const char* name, gdv_node_t* x, gdv_node_t* y, gdv_type_t t) {
    gdv_node_t* args[2] = {x, y};
    gdv_node_t* n = gdv_node_function(name, args, 2, t);
    CHECK(n != nullptr);
    return n;
  };
  // C2 shape: a + b, a * b, (a - b) * a
  gdv_expression_t* c2[3] = {
      gdv_expression_new(fn2("add", fa, fb, kF64), "e0", kF64),
      gdv_expression_new(fn2("multiply", fa, fb, kF64), "e1", kF64),
      gdv_expression_new(fn2("multiply", fn2("subtract", fa, fb, kF64), fa, kF64), "e2", kF64)};
  gdv_projector_t* p2 = nullptr;
  CHECK(gdv_projector_make(schema, c2, 3, GDV_SEL_NONE, nullptr, &p2) == GDV_OK);
  // C3 shape: k1 > 499 AND k2 < 250
  const int64_t c499 = 499, c250 = 250;
  gdv_node_t* conj[2] = {fn2("greater_than", fk1, gdv_node_literal(kI64, &c499, 0), kBool),
                         fn2("less_than", fk2, gdv_node_literal(kI64, &c250, 0), kBool)};
  gdv_filter_t* f3 = nullptr;
  CHECK(gdv_filter_make(schema, gdv_condition_new(gdv_node_and(conj, 2)), nullptr, &f3) == GDV_OK);
  // C5 shape: like '%spark%', substr(s, 2, 5), upper(s)
  const int64_t two = 2, five = 5;
  gdv_node_t* sub_args[3] = {fs, gdv_node_literal(kI64, &two, 0), gdv_node_literal(kI64, &five, 0)};
  gdv_node_t* up_args[1] = {fs};
  gdv_expression_t* c5[3] = {
      gdv_expression_new(fn2("like", fs, gdv_node_literal_bytes(kStr, "%spark%", 7, 0), kBool), "m", kBool),
      gdv_expression_new(gdv_node_function("substr", sub_args, 3, kStr), "sub", kStr),
      gdv_expression_new(gdv_node_function("upper", up_args, 1, kStr), "up", kStr)};
  gdv_projector_t* p5 = nullptr;
  CHECK(gdv_projector_make(schema, c5, 3, GDV_SEL_NONE, nullptr, &p5) == GDV_OK);

  const HostColumn* hcols[5] = {&a, &b, &k1, &k2, &s};
  const int widths[5] = {8, 8, 8, 8, 0};

  // Evaluates rows [lo, hi) on the calling thread's device; results into host vectors.
  auto evaluate = [&](int64_t lo, int64_t hi, Output (&o2)[3], std::vector<uint32_t>* sel, Output (&o5)[3]) {
    const int64_t n = hi - lo;
    DeviceColumn dc[5];
    gdv_column_t cols[5];
    for (int k = 0; k < 5; k++) {
      dc[k].Upload(*hcols[k], widths[k], lo, hi);
      cols[k] = dc[k].view;
    }
    const int64_t vbytes = ((n + 63) / 64) * 8;
    // C2
    {
      void *dv[3], *dd[3];
      gdv_out_column_t outs[3];
      memset(outs, 0, sizeof(outs));
      for (int e = 0; e < 3; e++) {
        CHECK(gdv_device_alloc(vbytes, &dv[e]) == GDV_OK);
        CHECK(gdv_device_alloc(n * 8, &dd[e]) == GDV_OK);
        outs[e].validity = dv[e]; outs[e].validity_size = vbytes;
        outs[e].data = dd[e]; outs[e].data_size = n * 8;
      }
      CHECK(gdv_projector_evaluate(p2, n, cols, 5, nullptr, outs, 3, GDV_MEM_DEVICE, nullptr, 0) == GDV_OK);
      for (int e = 0; e < 3; e++) {
        o2[e].validity.resize(vbytes); o2[e].data.resize(n * 8);
        CHECK(gdv_memcpy_d2h(o2[e].validity.data(), dv[e], vbytes) == GDV_OK);
        CHECK(gdv_memcpy_d2h(o2[e].data.data(), dd[e], n * 8) == GDV_OK);
        gdv_device_free(dv[e]); gdv_device_free(dd[e]);
      }
    }
    // C3
    {
      void* di = nullptr;
      CHECK(gdv_device_alloc(n * 4 + 64, &di) == GDV_OK);
      int64_t count = -1;
      CHECK(gdv_filter_evaluate(f3, n, cols, 5, GDV_SEL_UINT32, di, n, &count, GDV_MEM_DEVICE, nullptr) == GDV_OK);
      sel->resize(count);
      if (count > 0) CHECK(gdv_memcpy_d2h(sel->data(), di, count * 4) == GDV_OK);
      gdv_device_free(di);
    }
    // C5
    {
      void *dv[3], *dd[3] = {nullptr, nullptr, nullptr}, *dofs[3] = {nullptr, nullptr, nullptr};
      gdv_out_column_t outs[3];
      memset(outs, 0, sizeof(outs));
      const int64_t cap = (int64_t)hcols[4]->offsets[hi] - hcols[4]->offsets[lo] + 64;
      for (int e = 0; e < 3; e++) {
        CHECK(gdv_device_alloc(vbytes, &dv[e]) == GDV_OK);
        outs[e].validity = dv[e]; outs[e].validity_size = vbytes;
        if (e == 0) {
          CHECK(gdv_device_alloc(vbytes, &dd[e]) == GDV_OK);
          outs[e].data = dd[e]; outs[e].data_size = vbytes;
        } else {
          CHECK(gdv_device_alloc(cap, &dd[e]) == GDV_OK);
          CHECK(gdv_device_alloc((n + 1) * 4, &dofs[e]) == GDV_OK);
          outs[e].data = dd[e]; outs[e].data_size = cap;
          outs[e].offsets = dofs[e]; outs[e].offsets_size = (n + 1) * 4;
        }
      }
      CHECK(gdv_projector_evaluate(p5, n, cols, 5, nullptr, outs, 3, GDV_MEM_DEVICE, nullptr, 0) == GDV_OK);
      for (int e = 0; e < 3; e++) {
        o5[e].validity.resize(vbytes);
        CHECK(gdv_memcpy_d2h(o5[e].validity.data(), dv[e], vbytes) == GDV_OK);
        if (e == 0) {
          o5[e].data.resize(vbytes);
          CHECK(gdv_memcpy_d2h(o5[e].data.data(), dd[e], vbytes) == GDV_OK);
        } else {
          o5[e].offsets.resize(n + 1);
          CHECK(gdv_memcpy_d2h(o5[e].offsets.data(), dofs[e], (n + 1) * 4) == GDV_OK);
          o5[e].data.resize(outs[e].data_size);
          if (outs[e].data_size > 0) CHECK(gdv_memcpy_d2h(o5[e].data.data(), dd[e], outs[e].data_size) == GDV_OK);
        }
        gdv_device_free(dv[e]); gdv_device_free(dd[e]); gdv_device_free(dofs[e]);
      }
    }
    for (int k = 0; k < 5; k++) dc[k].Free();
  };

  // ---- unsharded, on device 0
  CHECK(gdv_set_device(0) == GDV_OK);
  Output w2[3], w5[3];
  std::vector<uint32_t> wsel;
  evaluate(0, rows, w2, &wsel, w5);

  // ---- sharded: one host thread per device context
  std::vector<Output> g2(3 * N), g5(3 * N);
  std::vector<std::vector<uint32_t>> gsel(N);
  std::vector<int64_t> los(N), his(N);
  std::vector<std::thread> threads;
  for (int r = 0; r < N; r++) {
    CHECK(gdv_shard_bounds(rows, N, r, &los[r], &his[r]) == GDV_OK);
    threads.emplace_back([&, r] {
      CHECK(gdv_set_device(r) == GDV_OK);
      CHECK(gdv_get_device() == r);
      if (his[r] == los[r]) return;
      Output o2[3], o5[3];
      evaluate(los[r], his[r], o2, &gsel[r], o5);
      for (int e = 0; e < 3; e++) { g2[3 * r + e] = std::move(o2[e]); g5[3 * r + e] = std::move(o5[e]); }
    });
  }
  for (auto& t : threads) t.join();

  // ---- concatenation == unsharded
  int64_t checked = 0;
  std::vector<uint32_t> cat_sel;
  for (int r = 0; r < N; r++) {
    CHECK(r == 0 ? los[r] == 0 : los[r] == his[r - 1]);
    const int64_t lo = los[r], n = his[r] - los[r];
    for (uint32_t i : gsel[r]) cat_sel.push_back((uint32_t)(i + lo));
    for (int64_t i = 0; i < n; i++) {
      for (int e = 0; e < 3; e++) {
        const bool v = BitAt(g2[3 * r + e].validity, i);
        CHECK(v == BitAt(w2[e].validity, lo + i));
        if (v) CHECK(memcmp(&g2[3 * r + e].data[i * 8], &w2[e].data[(lo + i) * 8], 8) == 0);
      }
      const bool mv = BitAt(g5[3 * r].validity, i);
      CHECK(mv == BitAt(w5[0].validity, lo + i));
      if (mv) CHECK(BitAt(g5[3 * r].data, i) == BitAt(w5[0].data, lo + i));
      for (int e = 1; e < 3; e++) {
        const Output &g = g5[3 * r + e], &w = w5[e];
        const bool v = BitAt(g.validity, i);
        CHECK(v == BitAt(w.validity, lo + i));
        const int32_t gl = g.offsets[i + 1] - g.offsets[i], wl = w.offsets[lo + i + 1] - w.offsets[lo + i];
        CHECK(gl == wl);
        // per-shard offsets rebased: shard r's bytes start where the unsharded output has row lo
        CHECK(g.offsets[i] + w.offsets[lo] == w.offsets[lo + i]);
        if (v && gl > 0) CHECK(memcmp(&g.data[g.offsets[i]], &w.data[w.offsets[lo + i]], gl) == 0);
      }
      checked++;
    }
  }
  CHECK(his[N - 1] == rows && checked == rows);
  CHECK(cat_sel == wsel);
  for (size_t i = 1; i < cat_sel.size(); i++) CHECK(cat_sel[i - 1] < cat_sel[i]);
  if (argc > 3) {
    const std::string dir = argv[3];
    const HostColumn* in[5] = {&a, &b, &k1, &k2, &s};
    const char* names[5] = {"a", "b", "k1", "k2", "s"};
    for (int k = 0; k < 5; k++) {
      DumpBytes(dir, std::string(names[k]) + ".data", in[k]->data.data(), in[k]->data.size());
      DumpBytes(dir, std::string(names[k]) + ".validity", in[k]->validity.data(), in[k]->validity.size());
      DumpBytes(dir, std::string(names[k]) + ".offsets", in[k]->offsets.data(), in[k]->offsets.size() * 4);
    }
    for (int e = 0; e < 3; e++) {
      DumpBytes(dir, "c2_" + std::to_string(e) + ".data", w2[e].data.data(), w2[e].data.size());
      DumpBytes(dir, "c2_" + std::to_string(e) + ".validity", w2[e].validity.data(), w2[e].validity.size());
      DumpBytes(dir, "c5_" + std::to_string(e) + ".data", w5[e].data.data(), w5[e].data.size());
      DumpBytes(dir, "c5_" + std::to_string(e) + ".validity", w5[e].validity.data(), w5[e].validity.size());
      DumpBytes(dir, "c5_" + std::to_string(e) + ".offsets", w5[e].offsets.data(), w5[e].offsets.size() * 4);
    }
    DumpBytes(dir, "c3.indices", wsel.data(), wsel.size() * 4);
  }
  printf("multi-device ok: %d contexts (%d physical), %lld rows: C2 / C3 (%zu selected) / C5 shards == unsharded\n", N,
         gdv_physical_device_count(), (long long)rows, wsel.size());
  return 0;
}
