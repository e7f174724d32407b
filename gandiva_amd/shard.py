"""Row-range sharding of one logical batch across the GPUs of a node.

The path has no exchange step: every rank evaluates the same Projector / Filter on its own
contiguous row range, resident in its own HBM.  Shard boundaries fall on multiples of
``ALIGN`` = 1024 rows (one workgroup tile = one 128-byte line of every validity bitmap), so
no bitmap word or cache line straddles two shards and a shard of a bigger Arrow array is a
plain zero-copy slice (`offset` is a multiple of 64: no funnel shift in the kernel).

The only cross-rank datum is the per-shard selected-row COUNT of a filter (one integer per
rank, exchanged on the host with ``all_gather``); selection indices are local + ``row_base``
and concatenate in rank order into a globally ascending vector.  There is no data-path
collective and no RCCL dependency here: ``torch.distributed`` (gloo on CPU, nccl=RCCL on GPU)
is used for that one tiny all-gather and for barriers.
"""
import numpy as np
import pyarrow as pa

ALIGN = 1024


def shard_bounds(num_rows, world_size, rank, align=ALIGN):
    """[lo, hi) of `rank`'s row range: near-equal shards, boundaries on `align` multiples,
    the last shard takes the ragged tail.  Every row belongs to exactly one shard."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    tiles = (num_rows + align - 1) // align
    per, extra = divmod(tiles, world_size)
    lo_tile = rank * per + min(rank, extra)
    hi_tile = lo_tile + per + (1 if rank < extra else 0)
    lo = min(lo_tile * align, num_rows)
    hi = min(hi_tile * align, num_rows)
    return lo, hi


def shard_record_batch(batch, world_size, rank):
    """Zero-copy slice of a host pyarrow.RecordBatch for `rank` (+ its row base)."""
    lo, hi = shard_bounds(batch.num_rows, world_size, rank)
    return batch.slice(lo, hi - lo), lo


def exchange_counts(local_count, group=None):
    """All ranks' selected-row counts, in rank order (host-side all_gather of one int64)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return [int(local_count)]
    world = dist.get_world_size(group)
    device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    out = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def global_offsets(counts):
    """Exclusive prefix sum: where each rank's indices start in the concatenated vector."""
    offs = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(np.asarray(counts, dtype=np.int64), out=offs[1:])
    return offs


def rebase_indices(local_indices, row_base, dtype=np.uint64):
    """Local selection indices -> global row positions (host numpy)."""
    return np.asarray(local_indices).astype(np.int64) + np.int64(row_base)


def concat_selection(per_rank_indices, per_rank_bases):
    """Shard-ordered concatenation of rebased indices: globally ascending by construction."""
    parts = [rebase_indices(i, b) for i, b in zip(per_rank_indices, per_rank_bases)]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)


def concat_arrays(per_rank_arrays, contiguous=False):
    """Projector outputs stay sharded: the logical concatenation is a chunked array.  With
    ``contiguous=True`` the shards are joined into ONE array; for utf8/binary that rebases every
    shard's int32 offsets by the bytes of the shards before it."""
    parts = list(per_rank_arrays)
    if not contiguous:
        return pa.chunked_array(parts)
    return pa.concat_arrays([p.combine_chunks() if isinstance(p, pa.ChunkedArray) else p for p in parts])


def shard_device_batch(dbatch, world_size, rank):
    """Zero-copy slice of an HBM-resident DeviceBatch for `rank` (+ its row base).  Boundaries are
    multiples of 1024 rows, so validity / bool bitmaps slice at whole bytes (array offset 0 for
    the kernel) and fixed-width buffers at whole rows; a var-len column keeps the WHOLE byte
    buffer and slices only its offsets (they stay absolute)."""
    from . import gandiva as gdv
    lo, hi = shard_bounds(dbatch.num_rows, world_size, rank)
    cols = []
    for c in dbatch.columns:
        if c.offset != 0:
            raise ValueError("shard_device_batch expects columns without an Arrow array offset")
        validity = None if c.validity is None else c.validity[lo // 8:]
        if c.offsets is not None:
            cols.append(gdv.DeviceColumn(c.type, hi - lo, validity, c.data, c.offsets[lo * 4:], 0))
        elif pa.types.is_boolean(c.type):
            cols.append(gdv.DeviceColumn(c.type, hi - lo, validity, c.data[lo // 8:], None, 0))
        else:
            w = c.type.bit_width // 8
            cols.append(gdv.DeviceColumn(c.type, hi - lo, validity, c.data[lo * w:], None, 0))
    return gdv.DeviceBatch(dbatch.schema, cols, hi - lo), lo


def concat_varlen_device(per_rank_outputs):
    """Joins the var-len output shards of consecutive ranks ON THE DEVICE: offsets of shard r are
    rebased by the byte total of shards 0..r-1, bytes are laid end to end.  Returns
    (offsets int32 tensor of rows+1 entries, bytes uint8 tensor)."""
    import torch
    offs, data, base = [], [], 0
    for i, o in enumerate(per_rank_outputs):
        n = o.length
        local = o.offsets.view(torch.int32)[:n + 1].to(torch.int64)
        used = int(local[n].item())
        offs.append((local[:n] if i + 1 < len(per_rank_outputs) else local) + base)
        data.append(o.data[:used])
        base += used
    return torch.cat(offs).to(torch.int32), torch.cat(data)
