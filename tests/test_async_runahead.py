"""An asynchronous caller may enqueue far ahead of the GPU: the library's scratch blocks must not grow with the run-ahead
(round 6: Runtime::Alloc waits for the oldest block of a size once eight of that size are waiting for their streams)."""
import pytest


@pytest.mark.gpu
def test_scratch_of_asynchronous_evaluations_is_bounded_by_the_run_ahead_cap():
    import torch
    import gandiva_amd as gandiva
    from gandiva_amd import workloads as W
    rows = 50_000_000
    batch = W.c5_device_batch_philox(rows)
    proj = gandiva.make_projector(W.c5_schema(), W.c5_expressions(), None)
    outs, result = proj.evaluate_device_async(batch)
    torch.cuda.synchronize()
    assert int(result[0].item()) == 0
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(100):          # ~0.5 ms of GPU work each, enqueued faster than that
        outs, result = proj.evaluate_device_async(batch, outputs=outs)
    free1, _ = torch.cuda.mem_get_info()
    torch.cuda.synchronize()
    assert int(result[0].item()) == 0
    # per call: head + counts + bases + chunks + state, ~1.6 MB at this size: eight calls' worth is ~13 MB, a run-ahead of
    # 50-100 calls would be 80-160 MB.  (torch's own allocations are cached and do not move mem_get_info inside the loop.)
    assert free0 - free1 < 48 << 20, f"scratch grew by {(free0 - free1) >> 20} MiB over 100 asynchronous calls"
