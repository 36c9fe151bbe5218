"""CPU, world_size 2 over gloo: the bucketed gradient exchange of vit_ae_plus_plus_amd.ddp —
mean-of-ranks contract, bucket ranges covering the arena exactly once, fused metric all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import mae_ref as R
        from vit_ae_plus_plus_amd import ddp
        from vit_ae_plus_plus_amd.utils import misc
        torch.manual_seed(0)
        cfg = R.RefConfig(volume_size=(16, 16, 16), patch_size=4, in_chans=1, embed_dim=24, depth=2, num_heads=2,
                          decoder_embed_dim=24, decoder_depth=1, decoder_num_heads=2, contrastive=True)
        sd = R.init_state_dict(cfg, seed=0)

        # a stand-in with the engine's layout rules (matrices | tokens | vectors), built on CPU
        class FakeEngine:
            pass
        eng = FakeEngine()
        eng.cfg = cfg
        names = [k for k in sd if R.is_trainable(k)]
        mats = [k for k in names if not (sd[k].ndim <= 1 or k.endswith('.bias')) and k not in ('cls_token', 'mask_token')]
        toks = [k for k in names if k in ('cls_token', 'mask_token')]
        vecs = [k for k in names if sd[k].ndim <= 1 or k.endswith('.bias')]
        eng.layout, off = {}, 0
        for k in mats + toks + vecs:
            if toks and k == toks[0]:
                eng.tok_off = off
            if k == vecs[0]:
                eng.vec_off = off
            eng.layout[k] = (off, tuple(sd[k].shape))
            off += (sd[k].numel() + 3) // 4 * 4
        eng.n_total = off
        import types
        from vit_ae_plus_plus_amd.engine import HipMAEEngine
        eng.enc_chunk_bounds = types.MethodType(HipMAEEngine.enc_chunk_bounds, eng)
        for chunks in (1, 2):
            eng.enc_chunks = chunks
            ranges = ddp.engine_bucket_ranges(eng)
            assert len(ranges) == chunks + 2
            covered = sorted(ranges)
            assert covered[0][0] == 0 and covered[-1][1] == off
            assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered
        # completion order: decoder first, then encoder chunks from the top block down to offset 0, vectors last
        assert ranges[0][0] == eng.layout['decoder_embed.weight'][0] and ranges[1][1] == ranges[0][0]
        assert ranges[1][0] == eng.layout['blocks.1.attn.qkv.weight'][0] and ranges[2] == (0, ranges[1][0])
        # per-rank gradients from the oracle on different data, pre-scaled by 1/world like set_loss_weights
        params = R.make_leaf_params(sd)
        v1, v2 = R.synthetic_views((2, 1, 16, 16, 16), seed=10 + rank)
        n1, n2 = R.masking_noise(2, cfg.num_patches, seed=20 + rank)
        loss, pred, mask, p1, p2, z1, z2 = R.contr_forward(params, v1, v2, n1, n2, cfg, 0.75, 0.01)
        total = loss[0] + R.contrastive_loss(p1, p2, z1, z2, 0.001)
        (total / world).backward()
        flat = torch.zeros(off)
        for k, (o, shp) in eng.layout.items():
            flat[o:o + params[k].numel()] = params[k].grad.reshape(-1)
        local = flat.clone()
        red = ddp.GradBucketReducer(flat, ranges, max_bucket_elems=1000)
        assert red.world_size == world
        for b in range(len(ranges)):
            red.launch(b)
        red.wait()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered)          # each already carries 1/world -> mean of the raw per-rank grads
        assert torch.allclose(flat, want, rtol=1e-6, atol=1e-9)
        # bf16 on the wire: every rank ends with the same fp32 arena, equal to the bf16-rounded sum
        flat16 = local.clone()
        red16 = ddp.GradBucketReducer(flat16, ranges, comm_dtype=torch.bfloat16)
        for b in range(len(ranges)):
            red16.launch(b)
        red16.wait()
        assert float((flat16 - want).norm() / want.norm()) < 1e-2     # bf16 round-off of the operands and the sum
        both = [torch.zeros_like(flat16) for _ in range(world)]
        dist.all_gather(both, flat16)
        assert torch.equal(both[0], both[1])
        means = misc.all_reduce_means([float(rank), 2.0, float(total)])
        assert means[0] == pytest.approx((world - 1) / 2) and means[1] == 2.0
        assert misc.all_reduce_mean(float(rank)) == pytest.approx((world - 1) / 2)
        # one metric collective per logging window (utils/train_one_epoch.py): rows of per-iteration scalars
        rows = misc.all_reduce_mean_rows([[float(rank), 1.0], [2.0 * rank, 3.0], [5.0, float(rank + 1)]])
        assert rows[0] == pytest.approx([(world - 1) / 2, 1.0]) and rows[1] == pytest.approx([world - 1.0, 3.0])
        assert rows[2] == pytest.approx([5.0, (world + 1) / 2])
        # generic (non-fused) route: whole-arena exchange leaving the mean of the unscaled per-rank gradients
        arena = torch.full((off,), float(rank + 1))
        red2 = ddp.GradBucketReducer(arena, ranges)
        ddp.allreduce_mean_now(red2)
        assert torch.allclose(arena, torch.full((off,), (world + 1) / 2))
        # bench.py's rank plumbing as torch.distributed.run would drive it: the group must have --gpus ranks
        import bench
        bargs = bench.parse(['--gpus', str(world)])
        assert bench.rank_layout(bargs) == (world, rank, rank)
        assert dist.get_world_size() == bargs.gpus
        sv = misc.SmoothedValue()
        sv.update(float(rank + 1), n=rank + 1)
        sv.synchronize_between_processes()
        assert sv.count == 3 and sv.total == 1.0 * 1 + 2.0 * 2
        q.put((rank, 'ok'))
    except Exception as e:   # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == 'ok' for r in res), res


def _worker4(rank, world, port, q):
    """World of four, the data-parallel DEFAULTS: four exchange buckets with block cuts 0,1,4,8,12 (the last bucket only block 0 +
    the patch embedding), bf16 wire, a two-part decoder (VITAE_DEC_CHUNKS=2: decoder_pred + top blocks + predictor exchanged
    while the bottom blocks are still in their backward)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import types
        from oracle import mae_ref as R
        from vit_ae_plus_plus_amd import ddp
        from vit_ae_plus_plus_amd.engine import HipMAEEngine
        cfg = R.RefConfig(volume_size=(16, 16, 16), patch_size=4, in_chans=1, embed_dim=24, depth=12, num_heads=2,
                          decoder_embed_dim=24, decoder_depth=4, decoder_num_heads=2, contrastive=True)
        sd = R.init_state_dict(cfg, seed=0)

        class FakeEngine:
            pass
        eng = FakeEngine()
        eng.cfg = cfg
        names = [k for k in sd if R.is_trainable(k)]
        mats = [k for k in names if not (sd[k].ndim <= 1 or k.endswith('.bias')) and k not in ('cls_token', 'mask_token')]
        toks = [k for k in names if k in ('cls_token', 'mask_token')]
        vecs = [k for k in names if sd[k].ndim <= 1 or k.endswith('.bias')]
        eng.layout, off = {}, 0
        for k in mats + toks + vecs:
            if toks and k == toks[0]:
                eng.tok_off = off
            if k == vecs[0]:
                eng.vec_off = off
            eng.layout[k] = (off, tuple(sd[k].shape))
            off += (sd[k].numel() + 3) // 4 * 4
        eng.n_total = off
        for name in ('enc_chunk_bounds', 'set_backward_chunks', 'set_decoder_chunks'):
            setattr(eng, name, types.MethodType(getattr(HipMAEEngine, name), eng))
        eng.enc_chunks, eng.enc_cuts, eng.dec_chunks = 3, None, 1
        os.environ['VITAE_DEC_CHUNKS'] = '2'
        from vit_ae_plus_plus_amd.model.vit_autoenc import MaskedAutoencoderViT
        MaskedAutoencoderViT._set_exchange_buckets(eng, None)           # the data-parallel default cuts
        assert eng.enc_chunks == 4 and eng.enc_cuts == [0, 1, 4, 8, 12] and eng.dec_chunks == 2
        eng.dec_cut = cfg.decoder_depth // 2
        ranges = ddp.engine_bucket_ranges(eng)
        assert len(ranges) == 2 + 4 + 1                                  # two decoder buckets, four encoder buckets, tokens + vectors
        covered = sorted(ranges)
        assert covered[0][0] == 0 and covered[-1][1] == off and all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered
        bounds = eng.enc_chunk_bounds()
        assert len(bounds) == 4 and bounds[-1] == (0, 0), bounds          # the LAST encoder bucket: block 0 (+ the patch embedding)
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(off, generator=g) / world
        flat = local.clone()
        red = ddp.GradBucketReducer(flat, ranges, comm_dtype=torch.bfloat16)
        assert red.world_size == world
        for b in range(len(ranges)):
            red.launch(b)
            red.wait_bucket(b, copy_back=True)
        red.wait()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered)
        assert float((flat - want).norm() / want.norm()) < 1e-2
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert all(torch.equal(both[0], t) for t in both[1:])            # every replica holds the same reduced arena
        q.put((rank, 'ok'))
    except Exception:   # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_default_buckets_bf16_wire_world4_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == 'ok' for r in res), res


def test_single_process_reducer_is_a_noop():
    from vit_ae_plus_plus_amd import ddp
    flat = torch.arange(10.0)
    red = ddp.GradBucketReducer(flat, [(0, 4), (4, 10)])
    red.launch(0); red.launch(1); red.wait()
    assert torch.equal(flat, torch.arange(10.0)) and red.world_size == 1
