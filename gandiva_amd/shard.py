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


# ------------------------------------------------------------------------------------------------
# Round 6: ONE call for all the GPUs of a node (include/gandiva_amd.h: gdv_*_evaluate_sharded).  The library starts one
# host thread per shard, each on its own device context and stream; nothing is exchanged between shards.

def _physical(device):
    from . import gandiva as gdv
    return device % max(gdv.physical_device_count(), 1)


def evaluate_projector_sharded(projector, shards, devices=None, outputs=None):
    """shards[s]: DeviceBatch holding rows [lo_s, hi_s) (shard_bounds) of ONE logical batch, resident on devices[s]
    (default: device s).  Returns per shard the list of output DeviceColumns (allocated on that shard's GPU unless
    ``outputs`` from an earlier call are passed back)."""
    import ctypes as C
    import torch
    from . import _capi, gandiva as gdv
    lib = _capi.lib()
    n = len(shards)
    devices = list(range(n)) if devices is None else list(devices)
    num_rows = sum(b.num_rows for b in shards)
    for s, b in enumerate(shards):
        lo, hi = shard_bounds(num_rows, n, s)
        if b.num_rows != hi - lo:
            raise ValueError(f"shard {s} holds {b.num_rows} rows, shard_bounds gives {hi - lo}")
    n_out = len(projector._out_types)
    varlen = [pa.types.is_string(t) or pa.types.is_binary(t) for t in projector._out_types]
    arr = (_capi.gdv_shard_t * n)()
    keep = []
    fresh = outputs is None
    if fresh:
        outputs = []
    for s, b in enumerate(shards):
        dev = f"cuda:{_physical(devices[s])}"
        cols = (_capi.gdv_column_t * max(len(b.columns), 1))(*[c._c() for c in b.columns])
        if fresh:
            outs_s = []
            guess = 64 + sum(c.data.numel() for c in b.columns if c.offsets is not None)
            for i, t in enumerate(projector._out_types):
                vb, db = C.c_int64(), C.c_int64()
                gdv._check(lib.gdv_projector_output_sizes(projector._h, i, b.num_rows, 1, vb, db))
                outs_s.append(gdv.DeviceColumn(
                    t, b.num_rows, torch.empty(gdv._pad64(max(vb.value, 1)), dtype=torch.uint8, device=dev),
                    torch.empty(gdv._pad64((max(db.value, guess)) if varlen[i] else max(db.value, 1)), dtype=torch.uint8, device=dev),
                    torch.empty(gdv._pad64((b.num_rows + 1) * 4), dtype=torch.uint8, device=dev) if varlen[i] else None))
            outputs.append(outs_s)
        outs = (_capi.gdv_out_column_t * n_out)()
        for i, o in enumerate(outputs[s]):
            outs[i].validity, outs[i].validity_size = o.validity.data_ptr(), o.validity.numel()
            outs[i].data, outs[i].data_size = o.data.data_ptr(), o.data.numel()
            if o.offsets is not None:
                outs[i].offsets, outs[i].offsets_size = o.offsets.data_ptr(), o.offsets.numel()
        arr[s].device, arr[s].cols, arr[s].outs = devices[s], cols, outs
        keep.append((cols, outs))
    ncols = len(shards[0].columns)
    gdv._check(lib.gdv_projector_evaluate_sharded(projector._h, num_rows, ncols, n_out, arr, n, 0))
    for s in range(n):
        for i in range(n_out):
            if varlen[i]:
                outputs[s][i].data_used = keep[s][1][i].data_size
            outputs[s][i].length = shards[s].num_rows
    return outputs


def evaluate_filter_sharded(flt, shards, dtype="int32", devices=None, global_indices=True, gather_on=None):
    """One Filter::Evaluate over device-resident shards.  Returns (per-shard SelectionVectors, total, gathered):
    with global_indices the positions are lo_s + local and the shards' vectors concatenate into the globally ascending
    one; gather_on=d also lays them end to end on device d (a torch tensor; hipMemcpyPeerAsync)."""
    import ctypes as C
    import torch
    from . import _capi, gandiva as gdv
    lib = _capi.lib()
    n = len(shards)
    devices = list(range(n)) if devices is None else list(devices)
    mode = gdv.Filter._mode_of(dtype)
    tdt = {1: torch.int16, 2: torch.int32, 3: torch.int64}[mode]
    num_rows = sum(b.num_rows for b in shards)
    arr = (_capi.gdv_shard_t * n)()
    keep, outs = [], []
    for s, b in enumerate(shards):
        lo, hi = shard_bounds(num_rows, n, s)
        if b.num_rows != hi - lo:
            raise ValueError(f"shard {s} holds {b.num_rows} rows, shard_bounds gives {hi - lo}")
        cols = (_capi.gdv_column_t * max(len(b.columns), 1))(*[c._c() for c in b.columns])
        out = torch.empty(max(b.num_rows, 1), dtype=tdt, device=f"cuda:{_physical(devices[s])}")
        arr[s].device, arr[s].cols = devices[s], cols
        arr[s].out_indices, arr[s].max_slots = out.data_ptr(), out.numel()
        keep.append(cols)
        outs.append(out)
    total = C.c_int64(0)
    gdv._check(lib.gdv_filter_evaluate_sharded(flt._h, num_rows, len(shards[0].columns), mode, arr, n,
                                               2 if global_indices else 0, C.byref(total)))
    sels = [gdv.SelectionVector(mode, outs[s], int(arr[s].num_selected), device=True) for s in range(n)]
    gathered = None
    if gather_on is not None:
        gathered = torch.empty(max(total.value, 1), dtype=tdt, device=f"cuda:{_physical(gather_on)}")
        gdv._check(lib.gdv_filter_gather_sharded(arr, n, mode, gather_on, C.c_void_p(gathered.data_ptr()), gathered.numel()))
        gathered = gathered[:total.value]
    return sels, total.value, gathered


def evaluate_projector_host_sharded(projector, batch, devices):
    """ONE host-resident pyarrow.RecordBatch evaluated by len(devices) GPUs: the library slices it on 1024-row bounds,
    every shard is staged through its own device, the results land in one set of host arrays."""
    import ctypes as C
    from . import _capi, gandiva as gdv
    lib = _capi.lib()
    gdv._check_batch(batch, projector._schema)
    cols = (_capi.gdv_column_t * max(batch.num_columns, 1))(*[gdv._column_of_array(a) for a in batch.columns])
    n_out = len(projector._out_types)
    varlen = [pa.types.is_string(t) or pa.types.is_binary(t) for t in projector._out_types]
    guess = 64 + sum(a.buffers()[2].size for a in batch.columns
                     if (pa.types.is_string(a.type) or pa.types.is_binary(a.type)) and a.buffers()[2] is not None)
    outs = (_capi.gdv_out_column_t * n_out)()
    holders = []
    for i, t in enumerate(projector._out_types):
        vb, db = C.c_int64(), C.c_int64()
        gdv._check(lib.gdv_projector_output_sizes(projector._h, i, batch.num_rows, 0, vb, db))
        v = pa.allocate_buffer(gdv._pad64(max(vb.value, 1)))
        d = pa.allocate_buffer(gdv._pad64(max(db.value, guess) if varlen[i] else max(db.value, 1)))
        o = pa.allocate_buffer(gdv._pad64((batch.num_rows + 1) * 4)) if varlen[i] else None
        holders.append((v, d, o))
        outs[i].validity, outs[i].validity_size = v.address, v.size
        outs[i].data, outs[i].data_size = d.address, d.size
        if o is not None:
            outs[i].offsets, outs[i].offsets_size = o.address, o.size
    devs = (C.c_int32 * len(devices))(*devices)
    gdv._check(lib.gdv_projector_evaluate_host_sharded(projector._h, batch.num_rows, cols, batch.num_columns, outs, n_out,
                                                       devs, len(devices)))
    result = []
    for i, t in enumerate(projector._out_types):
        v, d, o = holders[i]
        result.append(pa.Array.from_buffers(t, batch.num_rows, [v, o, d.slice(0, outs[i].data_size)] if varlen[i] else [v, d]))
    return result


def evaluate_filter_host_sharded(flt, batch, devices, dtype="int32"):
    """ONE host-resident batch filtered by len(devices) GPUs -> the global, ascending SelectionVector (host)."""
    import ctypes as C
    from . import _capi, gandiva as gdv
    lib = _capi.lib()
    gdv._check_batch(batch, flt._schema)
    mode = gdv.Filter._mode_of(dtype)
    cols = (_capi.gdv_column_t * max(batch.num_columns, 1))(*[gdv._column_of_array(a) for a in batch.columns])
    np_t = {1: np.uint16, 2: np.uint32, 3: np.uint64}[mode]
    out = np.zeros(max(batch.num_rows, 1), dtype=np_t)
    count = C.c_int64(0)
    devs = (C.c_int32 * len(devices))(*devices)
    gdv._check(lib.gdv_filter_evaluate_host_sharded(flt._h, batch.num_rows, cols, batch.num_columns, mode,
                                                    C.c_void_p(out.ctypes.data), out.size, C.byref(count), devs, len(devices)))
    return gdv.SelectionVector(mode, out, count.value)
