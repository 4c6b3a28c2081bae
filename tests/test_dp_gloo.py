"""world_size-2 gloo test (CPU) of the data-parallel exchange used by the SFT step: bucketed mean reduce-scatter with
ZeRO-1 shard ownership + all-gather of updated parameters == single-process average."""
import math
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vlaser_amd import dp
    n = 5000 + 37                      # ragged: not a multiple of world * 128
    buckets = [(0, 1024), (1024, 3000), (3000, n)]
    shards = dp.plan_shards(buckets, world, rank)
    g = torch.Generator().manual_seed(100 + rank)
    grads = torch.randn(n, generator=g).to(torch.bfloat16)
    params = torch.zeros(n, dtype=torch.bfloat16)
    flat_g = grads.clone()
    for b, s in zip(buckets, shards):
        dp.reduce_scatter_mean(flat_g, b, s)
    # "optimizer": every rank updates only its shard (p = -mean grad), then all-gather
    for (lo, hi), (s_lo, s_hi, per) in zip(buckets, shards):
        params[s_lo:s_hi] = -flat_g[s_lo:s_hi]
    for b, s in zip(buckets, shards):
        dp.all_gather_params(params, b, s)
    ret[rank] = (params.float(), grads.float(), shards)
    dist.destroy_process_group()


def test_bucketed_reduce_scatter_allgather_matches_mean():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29533, ret), nprocs=world, join=True)
    p0, g0, sh0 = ret[0]
    p1, g1, sh1 = ret[1]
    assert torch.equal(p0, p1), 'parameters differ across ranks after the all-gather'
    ref = -((g0 + g1) / 2).to(torch.bfloat16).float()
    assert torch.allclose(p0, ref, rtol=0, atol=1e-2)
    # ownership: shards of the two ranks tile every bucket without overlap
    for (a_lo, a_hi, per), (b_lo, b_hi, _) in zip(sh0, sh1):
        assert a_hi <= b_lo or b_hi <= a_lo
        assert per % 128 == 0


def test_plan_shards_covers_bucket():
    from vlaser_amd import dp
    buckets = [(0, 1000), (1000, 1000 + 128 * 9 + 5)]
    for world in (1, 2, 4, 8):
        cover = [set() for _ in buckets]
        for r in range(world):
            for i, (lo, hi, per) in enumerate(dp.plan_shards(buckets, world, r)):
                assert lo <= hi
                cover[i] |= set(range(lo, hi))
        for (lo, hi), c in zip(buckets, cover):
            assert c == set(range(lo, hi))


def test_cosine_lr_matches_hf_schedule():
    """HF get_cosine_schedule_with_warmup (the SFT launcher's scheduler) restated; compared with torch's LambdaLR driven by
    the published lambda."""
    from vlaser_amd.sft import cosine_lr
    total, base, ratio = 200, 2e-5, 0.03
    warm = math.ceil(total * ratio)

    def lam(cur):
        if cur < warm:
            return float(cur) / float(max(1, warm))
        prog = float(cur - warm) / float(max(1, total - warm))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 2.0 * 0.5 * prog)))
    opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=base)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    for step in range(total):
        assert abs(cosine_lr(step, total, base, ratio) - sch.get_last_lr()[0]) < 1e-12
        opt.step(); sch.step()
    assert cosine_lr(0, total, base) == 0.0 and abs(cosine_lr(warm, total, base) - base) < 1e-12
