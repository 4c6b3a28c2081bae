"""RCCL before the driver runs it: the `nccl` (= RCCL on ROCm) branch of vlaser_amd/dp.py -- in-place
`reduce_scatter_tensor(AVG, bf16)` into a slice of its own input, in-place `all_gather_into_tensor`, the comm-stream events, the
`gnorm2` all-reduce -- executed on the one-GPU box at world size 1 with VLASER_FORCE_DP=1.  At world 1 the exchange is the
identity, so two SFT steps and two VLA training steps must be BIT-identical to the same steps without a process group.
Reference semantics: zero_stage1_config.json (ZeRO-1, bf16 reduction), train.py:183-193 (DDP)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _sft(pg, clip):
    from vlaser_amd import config as C, synth
    from vlaser_amd.sft import SFTModel
    cfg = C.truncated(C.vlaser_2b(), 1, 2)
    m = SFTModel(cfg, max_seq_len=320, lr=1e-3, max_grad_norm=clip, process_group=pg, bucket_layers=1)
    m.load_state_dict(synth.vlm_state_dict(cfg, device='cuda'))
    g = torch.Generator().manual_seed(77)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.cat([torch.randint(0, 151643, (20,), generator=g), torch.full((256,), cfg.img_context_token_id),
                     torch.randint(0, 151643, (30,), generator=g)])[None]
    labels = torch.full_like(ids, -100)
    labels[0, -12:] = ids[0, -12:]
    outs = [m.step(pv, ids, labels) for _ in range(2)]
    m.wait_optimizer()
    torch.cuda.synchronize()
    return m, [float(o.loss) for o in outs], [float(o.grad_norm) for o in outs]


def _vla(pg):
    from vlaser_amd import config as C, synth
    from vlaser_amd.vla_train import VLATrainer
    vla = C.VLAConfig(base=C.truncated(C.vlaser_2b(), 2, 2))
    m = VLATrainer(vla, lr=1e-3, max_grad_norm=1.0, bucket_layers=1, process_group=pg)
    m.load_state_dict(synth.vla_state_dict(vla, with_head=True, device='cuda'))
    d = np.load(os.path.join(GOLDEN, 'g7_vla.npz'))
    f = np.load(os.path.join(GOLDEN, 'g10_flow_matching.npz'))
    smp = dict(input_ids=torch.from_numpy(d['a_input_ids']), pixel_values=torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(int(d['a_seed']))),
               proprios=torch.from_numpy(d['a_proprio']), actions=torch.from_numpy(f['a_actions']), t=torch.from_numpy(f['a_t']), x0=torch.from_numpy(f['a_x0']))
    outs = [m.step([smp]) for _ in range(2)]
    torch.cuda.synchronize()
    return m, [float(o.loss) for o in outs], [float(o.grad_norm) for o in outs]


def _worker(rank, port, out_dir, exchange):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), VLASER_FORCE_DP='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    # the same steps with the exchange on RCCL's C API (own communicator + own stream), with and without the CU masks
    mode, comm_cus = {'pg': ('pg', 0), 'capi': ('capi', 32), 'capi_nomask': ('capi', 0)}[exchange]
    os.environ.update(VLASER_DP_EXCHANGE=mode, VLASER_DP_COMM_CUS=str(comm_cus))
    if comm_cus:
        from vlaser_amd import ops
        ops.set_cu_budget(256 - comm_cus)          # the masked step sizes its GEMM grids (and split-K) for the CUs the mask leaves: the no-DP reference must do the same
    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    assert dist.get_backend() == 'nccl'
    res = {}
    # the single-rank reference takes the gradient norm from the gradient buffer here, like the data-parallel path (shard by shard, same order): the
    # norm assembled from the weight-gradient GEMM epilogues (r04, one rank only) sums in another order and is checked in test_sft_gpu.py
    os.environ['VLASER_SFT_NO_FUSED_NORM'] = '1'
    for clip in (0.0, 1.0):
        a, la, na = _sft(None, clip)
        b, lb, nb = _sft(dist.group.WORLD, clip)
        assert not a.dp_active and b.dp_active and b.comm_stream is not None and len(b.buckets) >= 3
        info = b.exchange_info()
        assert info['mode'] == mode and (b.capi is not None) == (mode == 'capi') and (b.main_stream is not None) == bool(comm_cus), info
        if mode == 'capi':
            assert info['comm_cus'] == comm_cus and info['compute_cus'] == 256 - comm_cus and b.comm_stream is b.capi.stream, info
            if comm_cus:
                assert b.opt_stream.cuda_stream not in (0, b.main_stream.cuda_stream) and b.wgrad_stream.cuda_stream != b.main_stream.cuda_stream
        res[f'sft{clip}'] = dict(same_p=torch.equal(a.fp.p, b.fp.p), same_m=torch.equal(a.m, b.m), same_v=torch.equal(a.v, b.v),
                                 same_master=torch.equal(a.master, b.master), la=la, lb=lb, na=na, nb=nb)
        if clip > 0:
            # the shard AdamW on the comm stream in front of each bucket's all-gather (r03) against the serial order (AdamW of every bucket
            # on the compute stream, then the all-gathers) -- an ordering or race bug around gnorm2 / adamw_clipped would show here, bit for bit
            os.environ['VLASER_SFT_DP_SERIAL_ADAMW'] = '1'
            c, lc, nc = _sft(dist.group.WORLD, clip)
            del os.environ['VLASER_SFT_DP_SERIAL_ADAMW']
            res['sft_serial_adamw'] = dict(same_p=torch.equal(c.fp.p, b.fp.p), same_m=torch.equal(c.m, b.m), same_v=torch.equal(c.v, b.v),
                                           same_master=torch.equal(c.master, b.master), la=lc, lb=lb, na=nc, nb=nb)
            del c
        del a, b
        torch.cuda.empty_cache()
    if exchange == 'pg':                           # (the VLA trainer's exchange stays on the process group: the switch is the SFT step's)
        a, la, na = _vla(None)
        b, lb, nb = _vla(dist.group.WORLD)
        assert not a.dp_active and b.dp_active
        res['vla'] = dict(same_p=torch.equal(a.fp.p, b.fp.p), same_m=torch.equal(a.m, b.m), same_v=torch.equal(a.v, b.v), same_master=torch.equal(a.master, b.master),
                          la=la, lb=lb, na=na, nb=nb)
    # the collectives themselves, on a ragged bucket (the padded path of dp.reduce_scatter_mean / all_gather_params) and on an even one (in place)
    from vlaser_amd import dp
    capi = None
    if mode == 'capi':
        from vlaser_amd import rccl_capi
        capi = rccl_capi.CapiExchange(dist.group.WORLD, 'cuda:0', comm_cus=comm_cus)
        assert capi.version >= 20000
    ok = True
    for n in (1000, 1024):
        g = torch.randn(n, device='cuda').to(torch.bfloat16)
        g0 = g.clone()
        shard = dp.plan_shards([(0, n)], 1, 0)[0]
        if capi is not None:
            capi.stream.wait_stream(torch.cuda.current_stream())
        dp.reduce_scatter_mean(g, (0, n), shard, dist.group.WORLD, capi=capi)
        dp.all_gather_params(g, (0, n), shard, dist.group.WORLD, capi=capi)
        torch.cuda.synchronize()
        ok = ok and torch.equal(g, g0)
    if capi is not None:
        t = torch.tensor([3.5], device='cuda')
        capi.stream.wait_stream(torch.cuda.current_stream())
        capi.all_reduce_sum(t)
        torch.cuda.synchronize()
        ok = ok and t.item() == 3.5
        capi.destroy()
    res['ragged_identity'] = ok
    torch.save(res, os.path.join(out_dir, 'res.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('exchange', ['pg', 'capi', 'capi_nomask'])
def test_rccl_world1_steps_are_bit_identical_to_no_dp(tmp_path, exchange):
    import torch.multiprocessing as mp
    port = 29900 + (os.getpid() % 50) + {'pg': 0, 'capi': 50, 'capi_nomask': 100}[exchange]
    mp.spawn(_worker, args=(port, str(tmp_path), exchange), nprocs=1, join=True)
    res = torch.load(tmp_path / 'res.pt')
    assert res['ragged_identity']
    for k in ('sft0.0', 'sft1.0', 'sft_serial_adamw') + (('vla',) if exchange == 'pg' else ()):
        r = res[k]
        assert r['la'] == r['lb'], (k, r)                                     # losses bit-equal
        assert r['na'] == r['nb'], (k, r)                                     # gradient norms: same shard order, all-reduce over one rank
        assert r['same_p'] and r['same_m'] and r['same_v'] and r['same_master'], (k, r)
