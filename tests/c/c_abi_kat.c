/* The reference lineage's first known-answer test (pyarrow/tests/test_gandiva.py:24-63,
 * test_tree_exp_builder: if (a > b) a else b over int32) driven through the C ABI from plain
 * C99 — what a C / cgo / JNI caller of include/gandiva_amd.h writes.
 *
 *   c_abi_kat --host-only   build the trees, render them, compile the kernel for gfx950
 *                           (no device needed)
 *   c_abi_kat               also evaluate on the GPU over host buffers and check the answer
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gandiva_amd.h"

#define CHECK(cond)                                                                         \
  do {                                                                                      \
    if (!(cond)) {                                                                          \
      fprintf(stderr, "FAILED %s:%d: %s  [%s]\n", __FILE__, __LINE__, #cond, gdv_last_error()); \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

int main(int argc, char** argv) {
  const int host_only = argc > 1 && strcmp(argv[1], "--host-only") == 0;
  const gdv_type_t i32 = {GDV_TYPE_INT32, 0, 0};
  const gdv_type_t boolean = {GDV_TYPE_BOOL, 0, 0};

  gdv_schema_t* schema = gdv_schema_new();
  CHECK(schema != NULL);
  CHECK(gdv_schema_add_field(schema, "a", i32, 1) == GDV_OK);
  CHECK(gdv_schema_add_field(schema, "b", i32, 1) == GDV_OK);

  gdv_node_t* a = gdv_node_field("a", i32);
  gdv_node_t* b = gdv_node_field("b", i32);
  gdv_node_t* args[2];
  args[0] = a;
  args[1] = b;
  gdv_node_t* cond = gdv_node_function("greater_than", args, 2, boolean);
  CHECK(cond != NULL);
  gdv_node_t* if_node = gdv_node_if(cond, a, b, i32);
  CHECK(if_node != NULL);
  char* text = gdv_node_to_string(if_node);
  CHECK(text != NULL);
  printf("%s\n", text);
  CHECK(strcmp(text, "if (bool greater_than((int32) a, (int32) b)) { (int32) a } else { (int32) b }") == 0);
  gdv_free_string(text);

  gdv_expression_t* expr = gdv_expression_new(if_node, "res", i32);
  CHECK(expr != NULL);
  gdv_expression_t* exprs[1];
  exprs[0] = expr;
  CHECK(gdv_precompile_projector(schema, exprs, 1, GDV_SEL_NONE) == GDV_OK);

  /* an ill-typed tree is refused with the reference's status code */
  {
    const gdv_type_t f64 = {GDV_TYPE_DOUBLE, 0, 0};
    gdv_node_t* bad = gdv_node_function("greater_than", args, 2, f64);
    gdv_expression_t* bad_expr = gdv_expression_new(bad, "x", f64);
    gdv_expression_t* bad_list[1];
    bad_list[0] = bad_expr;
    CHECK(gdv_precompile_projector(schema, bad_list, 1, GDV_SEL_NONE) == GDV_EXPRESSION_VALIDATION_ERROR);
    gdv_expression_free(bad_expr);
    gdv_node_free(bad);
  }

  /* round 6: the tier-0 program of the same tree (what its first evaluations run on while hipRTC compiles), and the
   * refusals of the new entry points, still without a device */
  {
    char* prog = gdv_tier0_program(schema, exprs, 1, 0);
    CHECK(prog != NULL);
    CHECK(strcmp(prog, "load in0 int32\nload in1 int32\ncompare gt int32\nload in0 int32\nload in1 int32\nif\nout0 int32\n") == 0);
    gdv_free_string(prog);
    CHECK(gdv_projector_evaluate_sharded(NULL, 10, 2, 1, NULL, 0, 0) != GDV_OK);
    CHECK(gdv_filter_evaluate_host_sharded(NULL, 10, NULL, 0, GDV_SEL_UINT32, NULL, 0, NULL, NULL, 0) != GDV_OK);
    CHECK(gdv_device_pool_alloc(NULL, 16, NULL) != GDV_OK);
    CHECK(gdv_device_pool_bytes(NULL, NULL) == 0);
    {
      int64_t lo = -1, hi = -1;
      CHECK(gdv_shard_bounds(10000, 3, 1, &lo, &hi) == GDV_OK && lo == 4096 && hi == 7168);
    }
    gdv_shutdown(); /* (idempotent; a later Make waits for its compilation) */
    gdv_shutdown();
  }

  if (!host_only) {
    gdv_projector_t* proj = NULL;
    CHECK(gdv_projector_make(schema, exprs, 1, GDV_SEL_NONE, NULL, &proj) == GDV_OK);
    char* ir = gdv_projector_dump_ir(proj);
    CHECK(ir != NULL && strstr(ir, "@expr_") != NULL);
    gdv_free_string(ir);

    const int32_t va[4] = {10, 12, -20, 5}, vb[4] = {5, 15, 15, 17};
    gdv_column_t cols[2];
    memset(cols, 0, sizeof(cols));
    cols[0].data = va; cols[0].data_size = sizeof(va);
    cols[1].data = vb; cols[1].data_size = sizeof(vb);
    int64_t vbytes = 0, dbytes = 0;
    CHECK(gdv_projector_output_sizes(proj, 0, 4, GDV_MEM_HOST, &vbytes, &dbytes) == GDV_OK);
    CHECK(vbytes == 1 && dbytes == 16);
    uint8_t validity[8] = {0};
    int32_t result[4] = {0, 0, 0, 0};
    gdv_out_column_t out;
    memset(&out, 0, sizeof(out));
    out.validity = validity; out.validity_size = sizeof(validity);
    out.data = result; out.data_size = sizeof(result);
    CHECK(gdv_projector_evaluate(proj, 4, cols, 2, NULL, &out, 1, GDV_MEM_HOST, NULL, 0) == GDV_OK);
    printf("%d %d %d %d  validity %#x\n", result[0], result[1], result[2], result[3], validity[0]);
    CHECK(result[0] == 10 && result[1] == 15 && result[2] == 15 && result[3] == 17);
    CHECK((validity[0] & 0xf) == 0xf);
    /* round 6: the same batch through the one-call multi-device entry point (one device here) and into a pool-owned buffer */
    {
      const int32_t devs[1] = {0};
      int32_t result2[4] = {0, 0, 0, 0};
      uint8_t validity2[8] = {0};
      gdv_out_column_t out2;
      memset(&out2, 0, sizeof(out2));
      out2.validity = validity2; out2.validity_size = sizeof(validity2);
      out2.data = result2; out2.data_size = sizeof(result2);
      CHECK(gdv_projector_evaluate_host_sharded(proj, 4, cols, 2, &out2, 1, devs, 1) == GDV_OK);
      CHECK(memcmp(result, result2, sizeof(result)) == 0 && (validity2[0] & 0xf) == 0xf);
      gdv_device_pool_t* pool = NULL;
      void* bufs[2];
      int tried = 0, kept = -1;
      CHECK(gdv_device_pool_create(&pool) == GDV_OK);
      CHECK(gdv_device_pool_reserve_set(pool, 2, 1 << 20, 4, bufs, NULL, &tried, &kept) == GDV_OK);
      CHECK(bufs[0] != NULL && bufs[1] != NULL && bufs[0] != bufs[1] && tried == 1);
      CHECK(gdv_device_pool_free(pool, bufs[0]) == GDV_OK && gdv_device_pool_free(pool, bufs[1]) == GDV_OK);
      gdv_device_pool_destroy(pool);
    }
    gdv_projector_free(proj);
  }

  gdv_expression_free(expr);
  gdv_node_free(if_node);
  gdv_node_free(cond);
  gdv_node_free(b);
  gdv_node_free(a);
  gdv_schema_free(schema);
  printf("ok\n");
  return 0;
}
