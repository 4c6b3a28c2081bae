"""The data-parallel SFT step with a REAL world size of 2 on the one-GPU box: two processes share the GPU and exchange CUDA
tensors over gloo (the same dp.py / SFTModel code path that runs over RCCL on a multi-GPU node; only the backend differs).

  * both ranks end every step with bit-identical parameters (ZeRO-1 shard ownership + all-gather),
  * the result equals the single-process emulation "average the two samples' gradients, then one AdamW step" BIT FOR BIT, with clipping off and
    with the default clip at 1.0 (the emulation sums the squared norm in the ranks' order: per-rank shard partials, then their sum),
  * the loss each rank reports is its own sample's loss.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16
KEYS = ['language_model.model.layers.0.self_attn.q_proj.weight', 'language_model.model.layers.1.mlp.down_proj.weight',
        'language_model.lm_head.weight', 'language_model.model.norm.weight', 'mlp1.1.weight', 'language_model.model.embed_tokens.weight']


def _sample(cfg, rank):
    g = torch.Generator().manual_seed(500 + rank)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.cat([torch.randint(0, 151643, (20,), generator=g), torch.full((256,), cfg.img_context_token_id),
                     torch.randint(0, 151643, (30,), generator=g)])[None]
    labels = torch.full_like(ids, -100)
    labels[0, -12:] = ids[0, -12:]
    return pv, ids, labels


def _model(clip, pg=None):
    from vlaser_amd import config as C, synth
    from vlaser_amd.sft import SFTModel
    cfg = C.truncated(C.vlaser_2b(), 1, 2)
    sd = synth.vlm_state_dict(cfg, device='cuda')      # generated on the device: the CPU generator was most of this test's wall time (full 151 674-row vocabulary)
    m = SFTModel(cfg, max_seq_len=320, lr=1e-3, max_grad_norm=clip, process_group=pg, bucket_layers=1)
    m.load_state_dict(sd)
    return cfg, m


def _worker(rank, world, port, clip, out_dir, mute_rank=-1):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_grad_enabled(False)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg, m = _model(clip, dist.group.WORLD)
    assert m.world == 2 and m.dp_active and len(m.buckets) >= 3
    pv, ids, labels = _sample(cfg, rank)
    if rank == mute_rank:
        labels = torch.full_like(labels, -100)          # no supervised position on this rank
    losses = []
    for _ in range(2):
        out = m.step(pv, ids, labels)
        losses.append(float(out.loss))
    sd = m.state_dict()
    torch.save({'losses': losses, 'params': {k: sd[k].cpu() for k in KEYS}}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('clip', [0.0, 1.0])
def test_world2_step_equals_gradient_averaging(tmp_path, clip):
    import torch.multiprocessing as mp
    port = 29600 + int(clip) + (os.getpid() % 50) * 2
    mp.spawn(_worker, args=(2, port, clip, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    for k in KEYS:
        assert torch.equal(r0['params'][k], r1['params'][k]), k                 # every rank holds the same updated parameters
    # single-process emulation: mean of the two samples' gradients, then the same optimizer step
    torch.set_grad_enabled(False)
    from vlaser_amd import dp, ops, sft as sft_mod
    cfg, m = _model(clip)
    samples = [_sample(cfg, r) for r in range(2)]
    # the emulation sums the squared norm exactly as two ranks do: rank r adds its shard of every bucket, bucket by bucket, into its own fp32
    # accumulator; the all-reduce then adds the two totals.  With that the clipped step is BIT-identical too, not merely close.
    # The two ranks hold the gradients in the WORLD-2 layout (bucket boundaries padded to 2 x 128 elements), so the emulation copies its averaged
    # gradients into a buffer of that layout and lets "rank r" run the same vlaser_sumsq over the same element ranges.
    fp2, buckets2 = sft_mod.plan_flat_layout(cfg, 2, bucket_layers=1)
    off2 = {name: off for name, _, off in fp2.specs}
    g2 = torch.zeros(buckets2[-1][1], dtype=BF, device='cuda')
    acc = [torch.zeros(1, device='cuda'), torch.zeros(1, device='cuda')]
    real_sumsq = ops.sumsq
    state = {'done': False}

    def sumsq_as_two_ranks(x, out, ws):
        if state['done']:
            return                                  # the first call of the step already delivered the whole norm
        g2.zero_()
        for name, shape, off in m.fp.specs:
            n = m.fp.gview[name].numel()
            g2[off2[name]:off2[name] + n].copy_(m.fp.g[off:off + n])
        for r in range(2):
            acc[r].zero_()
            for (s_lo, s_hi, _) in dp.plan_shards(buckets2, 2, r):
                if s_hi > s_lo:
                    real_sumsq(g2[s_lo:s_hi], acc[r], ws)
        out.copy_(acc[0] + acc[1])                  # the all-reduce over two ranks
        state['done'] = True

    for step in range(2):
        grads, losses = [], []
        for pv, ids, labels in samples:
            losses.append(float(m.forward_backward(pv, ids, labels)))
            grads.append(m.fp.g.clone())
        m.fp.g.copy_(((grads[0].float() + grads[1].float()) / 2).to(BF))
        state['done'] = False
        sft_mod.ops.sumsq = sumsq_as_two_ranks
        try:
            m.optimizer_step()
        finally:
            sft_mod.ops.sumsq = real_sumsq
        assert abs(losses[0] - r0['losses'][step]) <= 1e-6 and abs(losses[1] - r1['losses'][step]) <= 1e-6, (step, losses, r0['losses'], r1['losses'])
    ref = m.state_dict()
    for k in KEYS:
        assert torch.equal(r0['params'][k].float(), ref[k].float().cpu()), k


def test_world2_rank_without_labels_issues_the_same_collectives(tmp_path):
    """a rank whose sample has no supervised label must still issue one reduce-scatter per bucket (otherwise its
    all_reduce(gnorm) meets the peers' reduce_scatter and the job hangs).  Rank 1 contributes zero gradients: the result equals
    'half of rank 0's gradient, then one AdamW step'."""
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 50) * 2
    mp.spawn(_worker, args=(2, port, 0.0, str(tmp_path), 1), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    assert r1['losses'] == [0.0, 0.0] and r0['losses'][0] > 0
    for k in KEYS:
        assert torch.equal(r0['params'][k], r1['params'][k]), k
    torch.set_grad_enabled(False)
    cfg, m = _model(0.0)
    pv, ids, labels = _sample(cfg, 0)
    for step in range(2):
        m.forward_backward(pv, ids, labels)
        m.fp.g.copy_((m.fp.g.float() / 2).to(BF))
        m.optimizer_step()
    ref = m.state_dict()
    for k in KEYS:
        assert torch.equal(r0['params'][k].float(), ref[k].float().cpu()), k
