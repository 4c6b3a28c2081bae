"""Data-parallel VLA flow-matching training step (the reference trains with DDP: train.py:470-482): two processes on one GPU over gloo,
one sample each.  The bucketed exchange averages the bf16 gradients in fp32 -- exactly what a single process does when it accumulates the
two samples -- so after two optimizer steps both ranks must hold bit-identical parameters, equal to the single-process `step([a, b])`.
Clipping is off here: with it the squared norm is summed per shard and all-reduced, i.e. in another order than on one rank, and the clip
factor may differ in its last bit (the clipped exchange itself is covered by tests/test_sft_dp_gpu.py on the same dp code)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _case(case):
    d = np.load(os.path.join(GOLDEN, 'g7_vla.npz'))
    f = np.load(os.path.join(GOLDEN, 'g10_flow_matching.npz'))
    pv = torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(int(d[f'{case}_seed'])))
    return dict(input_ids=torch.from_numpy(d[f'{case}_input_ids']), pixel_values=pv, proprios=torch.from_numpy(d[f'{case}_proprio']),
                actions=torch.from_numpy(f[f'{case}_actions']), t=torch.from_numpy(f[f'{case}_t']), x0=torch.from_numpy(f[f'{case}_x0']))


def _trainer(pg=None, train_vlm=False):
    from vlaser_amd import config as C, synth
    from vlaser_amd.vla_train import VLATrainer
    torch.set_grad_enabled(False)
    vla = C.VLAConfig(base=C.truncated(C.vlaser_2b(), 2, 2))
    sd = synth.vla_state_dict(vla, with_head=True, device='cuda')
    m = VLATrainer(vla, lr=1e-3, max_grad_norm=0.0, bucket_layers=1, process_group=pg, train_vlm=train_vlm, vlm_lr=2e-4)
    m.load_state_dict(sd)
    return m


def _worker_vlm(rank, world, port, out_dir):
    """train_vlm: True at world 2 -- both parameter groups exchanged (ZeRO-1 shards of each), one sample per rank, two steps."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m = _trainer(dist.group.WORLD, train_vlm=True)
    assert m.world == 2 and m.dp_active and len(m.vg.buckets) >= 3
    smp = _case('ab'[rank])
    losses = [float(m.step([smp]).loss) for _ in range(2)]
    torch.cuda.synchronize()
    torch.save({'losses': losses, 'p': {k: v.cpu() for k, v in m.state_dict().items()}}, os.path.join(out_dir, f'rank{rank}.pt'))
    # ---- resume at world 2 with BOTH parameter groups: the VLM group's shard keys are written and read back; the third step of
    # the resumed trainer == the uninterrupted third step bit for bit (weights, masters, both optimisers' moments)
    from vlaser_amd import config as C, synth
    from vlaser_amd.vla_train import ModelAveraging
    frozen = synth.vla_state_dict(C.VLAConfig(base=C.truncated(C.vlaser_2b(), 2, 2)), with_head=True, device='cuda')
    ck = os.path.join(out_dir, 'vlm_step2.pt')
    m.save_checkpoint(ck, frozen)
    dist.barrier()
    if rank == 0:
        data = torch.load(ck, weights_only=True)
        for k in ('cnt_update', 'cnt_batch', 'model', 'action_optimizer', 'vlm_optimizer', 'action_lr_scheduler', 'vlm_lr_scheduler', 'wandb_id', 'n_averaged'):
            assert k in data, k                                  # the keys the reference's save_training writes (train.py:655-670)
    # EMA of both groups from here on (the VLM group used to leave un-averaged under an 'ema' label)
    ma = ModelAveraging(m, use_ema=True, ema_start=3, ema_decay=0.5)
    m.step([smp]); ma.maybe_initialize(3); ma.maybe_update(3)
    torch.cuda.synchronize()
    want = {k: v.cpu() for k, v in m.state_dict().items()}
    want_state = [t.cpu().clone() for t in (m.m, m.v, m.master, m.vg.m, m.vg.v, m.vg.master)]
    first = {k: v.float().cpu() for k, v in want.items()}
    m.step([smp]); ma.maybe_update(4)
    torch.cuda.synchronize()
    second = {k: v.float().cpu() for k, v in m.state_dict().items()}
    avg = ma.state_dict()
    assert avg['model_type'] == 'ema' and avg['n_averaged'] == 2
    after = m.state_dict()
    assert all(torch.equal(after[k].cpu().float(), second[k]) for k in second), 'ModelAveraging.state_dict must restore the live weights'
    moved = 0
    for k in ('vision_model.encoder.layers.0.attn.qkv.weight', 'mlp1.1.weight', 'language_model.model.layers.0.mlp.down_proj.weight', 'action_expert.model.layers.0.mlp.down_proj.weight'):
        mid = 0.5 * (first[k] + second[k])                      # decay 0.5: avg = first + (second - first) / 2, from the fp32 masters
        got = avg['state_dict'][k].float().cpu()
        assert (got - mid).abs().max().item() <= 2.0 ** -7 * max(1e-3, mid.abs().max().item()), k
        moved += int(not torch.equal(first[k], second[k]))
    assert moved == 4
    r = _trainer(dist.group.WORLD, train_vlm=True)
    r.load_checkpoint(ck)
    assert r.step_count == 2
    r.step([smp])
    torch.cuda.synchronize()
    got = r.state_dict()
    assert all(torch.equal(want[k], got[k].cpu()) for k in want), 'resumed train_vlm run diverged'
    for w, t in zip(want_state, (r.m, r.v, r.master, r.vg.m, r.vg.v, r.vg.master)):
        assert torch.equal(w, t.cpu())
    dist.barrier()
    dist.destroy_process_group()


def test_world2_train_vlm_equals_single_process_accumulation(tmp_path):
    import torch.multiprocessing as mp
    port = 29860 + (os.getpid() % 50) * 2
    mp.spawn(_worker_vlm, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    assert r0['p'].keys() == r1['p'].keys() and all(torch.equal(r0['p'][k], r1['p'][k]) for k in r0['p']), 'ranks differ'
    m = _trainer(train_vlm=True)
    a, b = _case('a'), _case('b')
    outs = [m.step([a, b]) for _ in range(2)]
    torch.cuda.synchronize()
    ref = {k: v.cpu() for k, v in m.state_dict().items()}
    assert any(k.startswith('vision_model.') for k in ref) and any(k.startswith('language_model.model.layers.') for k in ref)
    for k in ref:
        assert torch.equal(ref[k], r0['p'][k]), (k, float((ref[k].float() - r0['p'][k].float()).abs().max()))
    for s in range(2):
        assert abs(0.5 * (r0['losses'][s] + r1['losses'][s]) - float(outs[s].loss)) < 1e-5


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m = _trainer(dist.group.WORLD)
    assert m.world == 2 and m.dp_active and len(m.buckets) >= 2
    smp = _case('ab'[rank])
    losses = [float(m.step([smp]).loss) for _ in range(2)]
    torch.cuda.synchronize()
    torch.save({'losses': losses, 'p': {k: v.cpu() for k, v in m.state_dict().items()}}, os.path.join(out_dir, f'rank{rank}.pt'))      # (the flat buffer's padding depends on the world size)
    # ---- resume at world 2: rank 0 writes the reference-layout step file, EVERY rank its own optimizer shard; a fresh trainer
    # that loads them continues bit for bit (third step == the uninterrupted third step), and a shard of another rank / world is refused
    from vlaser_amd import config as C, synth
    frozen = synth.vla_state_dict(C.VLAConfig(base=C.truncated(C.vlaser_2b(), 2, 2)), with_head=True, device='cuda')
    ck = os.path.join(out_dir, 'step2.pt')
    m.save_checkpoint(ck, frozen)
    dist.barrier()
    assert os.path.exists(ck) and all(os.path.exists(f'{ck}.optimizer_rank{r:05d}_of_00002.pt') for r in range(2))
    m.step([smp])
    torch.cuda.synchronize()
    want = {k: v.cpu() for k, v in m.state_dict().items()}
    want_m, want_v, want_master = m.m.cpu(), m.v.cpu(), m.master.cpu()
    r = _trainer(dist.group.WORLD)
    r.load_checkpoint(ck)
    assert r.step_count == 2
    r.step([smp])
    torch.cuda.synchronize()
    got = r.state_dict()
    assert all(torch.equal(want[k], got[k].cpu()) for k in want), 'resumed run diverged'
    assert torch.equal(want_m, r.m.cpu()) and torch.equal(want_v, r.v.cpu()) and torch.equal(want_master, r.master.cpu())
    os.replace(f'{ck}.optimizer_rank{rank:05d}_of_00002.pt', f'{ck}.keep{rank}')
    dist.barrier()
    try:
        _trainer(dist.group.WORLD).load_checkpoint(ck)
        raise AssertionError('a missing optimizer shard must raise')
    except FileNotFoundError:
        pass
    dist.barrier()
    dist.destroy_process_group()


def test_world2_equals_single_process_accumulation(tmp_path):
    import torch.multiprocessing as mp
    port = 29800 + (os.getpid() % 50) * 2
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    assert r0['p'].keys() == r1['p'].keys() and all(torch.equal(r0['p'][k], r1['p'][k]) for k in r0['p']), 'ranks differ'      # ZeRO-1 shards all-gathered
    m = _trainer()
    a, b = _case('a'), _case('b')
    outs = [m.step([a, b]) for _ in range(2)]
    torch.cuda.synchronize()
    ref = {k: v.cpu() for k, v in m.state_dict().items()}
    for k in ref:                                                 # mean of two ranks' gradients == accumulation of the two samples
        assert torch.equal(ref[k], r0['p'][k]), (k, float((ref[k].float() - r0['p'][k].float()).abs().max()))
    # each rank reported its own sample's loss; the single process reports their mean
    for s in range(2):
        assert abs(0.5 * (r0['losses'][s] + r1['losses'][s]) - float(outs[s].loss)) < 1e-5
