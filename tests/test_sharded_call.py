"""Round 6: ONE call evaluates one logical batch on all the GPUs of a node (gdv_*_evaluate_sharded /
gdv_*_evaluate_host_sharded): one host thread per shard inside the library, own device context and stream each, no
exchange step (the reference's Projector::Evaluate / Filter::Evaluate are one call —
pyarrow/includes/libgandiva.pxd:218-226, 246-248).  On the one-GPU test box the N devices are virtual contexts of the
same GPU: the code path (threads, contexts, per-device code objects, peer gather) is the multi-GPU one.

sharded == unsharded == oracle, for C2 / C3 / C4 / C5, device-resident shards and host-resident batches."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import _capi, shard, workloads as W
from oracle import oracle
from helpers import assert_bit_exact


def test_the_sharded_entry_points_are_exported_and_refuse_bad_arguments():
    lib = _capi.lib()
    for name in ("gdv_projector_evaluate_sharded", "gdv_filter_evaluate_sharded", "gdv_filter_gather_sharded",
                 "gdv_projector_evaluate_host_sharded", "gdv_filter_evaluate_host_sharded"):
        assert hasattr(lib, name)
    arr = (_capi.gdv_shard_t * 1)()
    assert lib.gdv_projector_evaluate_sharded(None, 10, 0, 0, arr, 1, 0) != 0
    assert "null projector" in _capi.last_error()
    assert lib.gdv_filter_evaluate_sharded(None, 10, 0, 2, arr, 1, 0, None) != 0
    assert lib.gdv_filter_gather_sharded(arr, 0, 2, 0, None, 0) != 0


def _device_shards(batch, n):
    """rows [lo_s, hi_s) of the host batch uploaded as shard s's own DeviceBatch (what rank s would hold)"""
    out = []
    for s in range(n):
        part, _ = shard.shard_record_batch(batch, n, s)
        # (a slice of a pyarrow batch keeps the parent's buffers + an offset: materialise the shard's own buffers)
        part = pa.RecordBatch.from_arrays([pa.concat_arrays([c]) for c in part.columns], schema=part.schema)
        out.append(gandiva.DeviceBatch.from_arrow(part))
    return out


@pytest.fixture
def four_contexts():
    gandiva.set_virtual_devices(4)
    yield 4


@pytest.mark.gpu
@pytest.mark.parametrize("workload,rows", [("c2", 300_007), ("c4", 123_456), ("c5", 200_003), ("c2", 3_000)])
def test_projector_sharded_in_one_call_equals_unsharded_and_the_oracle(four_contexts, workload, rows):
    n = four_contexts
    batch = {"c2": W.c2_batch, "c4": lambda r: W.c4_batch(r, 0.05), "c5": lambda r: W.c5_batch(r, 0.05)}[workload](rows)
    exprs = {"c2": W.c2_expressions, "c4": W.c4_expressions, "c5": W.c5_expressions}[workload]()
    proj = gandiva.make_projector(batch.schema, exprs, None)
    want = oracle.project(exprs, batch)
    # device-resident shards, one call
    shards = _device_shards(batch, n)
    outs = shard.evaluate_projector_sharded(proj, shards)
    for e in range(len(exprs)):
        got = shard.concat_arrays([outs[s][e].to_arrow() for s in range(n) if shards[s].num_rows > 0], contiguous=True)
        assert_bit_exact(got, want[e], f"{workload} expression {e}, {n} device shards")
    # second call into the same output buffers (steady state), shards on devices in another order
    outs = shard.evaluate_projector_sharded(proj, shards, devices=[3, 2, 1, 0], outputs=outs)
    got = shard.concat_arrays([outs[s][0].to_arrow() for s in range(n) if shards[s].num_rows > 0], contiguous=True)
    assert_bit_exact(got, want[0], f"{workload}, second call")
    # ONE host-resident batch, sliced and staged by the library
    got_host = shard.evaluate_projector_host_sharded(proj, batch, list(range(n)))
    for e in range(len(exprs)):
        assert_bit_exact(got_host[e], want[e], f"{workload} expression {e}, host batch over {n} devices")


@pytest.mark.gpu
@pytest.mark.parametrize("rows,nulls", [(1_000_003, 0.0), (250_000, 0.1), (5_000, 0.1)])
@pytest.mark.parametrize("dtype", ["int32", "int64"])
def test_filter_sharded_in_one_call_gives_the_global_ascending_vector(four_contexts, rows, nulls, dtype):
    import torch
    n = four_contexts
    batch = W.c3_batch(rows, nulls)
    cond = W.c3_condition()
    flt = gandiva.make_filter(batch.schema, cond)
    want = oracle.filter_indices(cond, batch, dtype).to_numpy().astype(np.int64)
    shards = _device_shards(batch, n)
    sels, total, gathered = shard.evaluate_filter_sharded(flt, shards, dtype, global_indices=True, gather_on=1)
    assert total == len(want)
    assert np.array_equal(gathered.cpu().numpy().astype(np.int64), want)
    parts = [s.to_array().to_numpy().astype(np.int64) for s in sels]
    assert np.array_equal(np.concatenate(parts), want)
    # local positions + the shard's base: the round-3 convention still available
    sels, total, _ = shard.evaluate_filter_sharded(flt, shards, dtype, global_indices=False)
    bases = [shard.shard_bounds(rows, n, s)[0] for s in range(n)]
    assert np.array_equal(shard.concat_selection([s.to_array().to_numpy() for s in sels], bases), want)
    # one host batch over the four devices
    got = shard.evaluate_filter_host_sharded(flt, batch, list(range(n)), dtype)
    assert np.array_equal(got.to_array().to_numpy().astype(np.int64), want)


@pytest.mark.gpu
def test_a_failing_shard_fails_the_call_and_names_its_device(four_contexts):
    """divide raises on the shard that holds a zero divisor; the other shards run to the end, the call returns that
    shard's ExecutionError and says which device it ran on."""
    n = four_contexts
    rows = 40_000
    a = np.arange(rows, dtype=np.int64) + 1
    b = np.ones(rows, dtype=np.int64)
    b[rows - 5] = 0                                         # last shard
    batch = pa.RecordBatch.from_arrays([pa.array(a), pa.array(b)], names=["a", "b"])
    bld = gandiva.TreeExprBuilder()
    fa, fb = bld.make_field(batch.schema.field(0)), bld.make_field(batch.schema.field(1))
    e = [bld.make_expression(bld.make_function("divide", [fa, fb], pa.int64()), pa.field("q", pa.int64()))]
    proj = gandiva.make_projector(batch.schema, e, None)
    with pytest.raises(gandiva.GandivaError) as err:
        shard.evaluate_projector_sharded(proj, _device_shards(batch, n))
    assert "divide by zero" in str(err.value) and "shard 3 (device 3)" in str(err.value)
    with pytest.raises(gandiva.GandivaError):
        shard.evaluate_projector_host_sharded(proj, batch, list(range(n)))
