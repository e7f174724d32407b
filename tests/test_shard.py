"""Row-range sharding: pure host logic + the one host-side exchange,
exercised with world_size 2 over gloo on CPU.  Local evaluation in these CPU tests is done by
the oracle (there is no GPU here); what is under test is the shard plan: disjoint cover,
1024-row alignment, index rebasing, count exchange, and that rank-ordered concatenation
equals the unsharded result."""
import os
import socket

import numpy as np
import pyarrow as pa
import pytest

from gandiva_amd import shard, workloads as W
from helpers import assert_bit_exact


@pytest.mark.parametrize("n", [0, 1, 1023, 1024, 1025, 5000, 1 << 20, (1 << 20) + 77])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_bounds_cover_and_align(n, world):
    prev_hi = 0
    for r in range(world):
        lo, hi = shard.shard_bounds(n, world, r)
        assert lo == prev_hi and lo <= hi
        assert lo % shard.ALIGN == 0 or lo == n
        prev_hi = hi
    assert prev_hi == n
    sizes = [shard.shard_bounds(n, world, r)[1] - shard.shard_bounds(n, world, r)[0] for r in range(world)]
    assert max(sizes) - min(sizes) <= 2 * shard.ALIGN


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, tmpdir):
    import torch.distributed as dist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = W.c3_batch(n, 0.1)
        local, base = shard.shard_record_batch(batch, world, rank)
        cond = W.c3_condition()
        idx = oracle.filter_indices(cond, local, "int32").to_numpy() if local.num_rows else np.zeros(0, np.uint32)
        counts = shard.exchange_counts(len(idx))
        offs = shard.global_offsets(counts)
        glob = shard.rebase_indices(idx, base)
        np.save(os.path.join(tmpdir, f"idx{rank}.npy"), glob)
        np.save(os.path.join(tmpdir, f"off{rank}.npy"), offs)
        # projection shard: outputs stay sharded, validity bitmaps never straddle shards
        b2 = W.c2_batch(n)
        l2, _ = shard.shard_record_batch(b2, world, rank)
        if l2.num_rows:
            out = oracle.project(W.c2_expressions()[:2], l2)
            with pa.ipc.new_file(os.path.join(tmpdir, f"proj{rank}.arrow"),
                                 pa.schema([("e0", pa.float64()), ("e1", pa.float64())])) as w:
                w.write_batch(pa.RecordBatch.from_arrays(out, names=["e0", "e1"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [5000, 70001])
def test_world_size_2_gloo_sharded_filter_and_projection(n, tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"idx{r}.npy") for r in range(world)]
    offs = np.load(tmp_path / "off0.npy")
    assert np.array_equal(offs, np.load(tmp_path / "off1.npy"))          # every rank agrees
    assert [len(p) for p in parts] == list(np.diff(offs))
    got = np.concatenate(parts)
    want = oracle.filter_indices(W.c3_condition(), W.c3_batch(n, 0.1), "int64").to_numpy()
    assert np.array_equal(got, want.astype(np.int64))
    assert np.all(np.diff(got) > 0)                                        # globally ascending
    # projection: rank-ordered concatenation == unsharded evaluation
    full = oracle.project(W.c2_expressions()[:2], W.c2_batch(n))
    chunks = [pa.ipc.open_file(tmp_path / f"proj{r}.arrow").read_all() for r in range(world)]
    for e in range(2):
        cat = pa.concat_arrays([c.column(e).combine_chunks() for c in chunks])
        assert_bit_exact(cat, full[e], f"e{e}")
