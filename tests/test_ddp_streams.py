"""The data-parallel step's three concurrent activities — backward (launch stream), per-bucket AdamW (optimiser stream), gradient
all-reduce (communication stream) — must sit on three different HARDWARE queues: HIP multiplexes streams over four of them, and
two streams on one queue run strictly one after the other (a modelled 0.4 ms all-reduce on the wrong stream costs the step
1.0-1.3 ms, tools/probes/r2_ddp_model*.sh).  ``ddp.pick_streams`` measures the sharing instead of guessing it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pick_streams_finds_three_hardware_queues():
    from vit_ae_plus_plus_amd import ddp
    dev = torch.device('cuda', 0)
    main = torch.cuda.current_stream(dev)
    opt, comm, rep = ddp.pick_streams(dev)
    assert rep['ok'], rep
    assert opt.cuda_stream != comm.cuda_stream != main.cuda_stream
    # a long kernel on any of the three does not hold back the other two
    for busy, others in ((main, [opt, comm]), (opt, [comm]), (comm, [opt])):
        assert ddp.held_back_by(busy, others) == [False] * len(others)
    # and the measurement does see sharing where there is some: among 12 streams on 4 queues at least one shares with `comm`
    many = [torch.cuda.Stream(device=dev) for _ in range(12)]
    x = torch.zeros(8, device=dev)
    for s in many:
        with torch.cuda.stream(s):
            x.add_(1.0)
    assert any(ddp.held_back_by(comm, many))
    assert ddp.held_back_by(comm, [comm]) == [True]


def test_collectives_run_on_the_chosen_stream(tmp_path):
    """World of one over RCCL: the reducer's all-reduce is issued as a synchronous call under ``comm_stream`` and completes
    there (its event is recorded on that stream); values are untouched (sum over one rank)."""
    import os
    import torch.distributed as dist
    from vit_ae_plus_plus_amd import ddp
    dev = torch.device('cuda', 0)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', device_id=dev)
    try:
        flat = torch.randn(1 << 20, device=dev)
        ref = flat.clone()
        red = ddp.GradBucketReducer(flat, [(0, 1 << 19), (1 << 19, 1 << 20)], force=True, comm_dtype=torch.bfloat16)
        assert red.active and red.comm_stream is not None
        _, red.comm_stream, _ = ddp.pick_streams(dev)
        for b in range(2):
            red.launch(b)
        red.wait(copy_back=True)
        torch.cuda.synchronize()
        assert torch.equal(flat, ref.bfloat16().float())
        assert not red.pending
    finally:
        dist.destroy_process_group()
