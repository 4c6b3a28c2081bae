"""Full-depth Vlaser-2B (24 ViT + 28 LLM + 28 action-expert layers, BASELINE.json sizes) on the GPU: properties that do not
need a CPU oracle run at this size (the oracle needs minutes there; oracle / golden parity runs on the depth-truncated
true-width model in test_models_gpu.py).

  * determinism: the same observation gives bit-identical chunks, eager and HIP-graph replays agree bit for bit;
  * flow-matching glue: with a zeroed action decoder the velocity is 0, so the chunk is exactly clip(noise) after 10 Euler
    steps; with a constant velocity v (zero decoder weight, bias v) the chunk is clip(noise + sum_k dt*v): the integrator
    takes exactly num_inference_steps steps of 1/num_inference_steps;
  * batching: a batch of 2 observations gives each observation's single-batch chunk within the bf16 tolerance;
  * left-padded ragged greedy decoding agrees with one-by-one decoding up to the first low-margin step.
"""
import time

import pytest
import torch

from parity import elementwise, parity

pytestmark = pytest.mark.gpu
# per-tensor bounds of the full-depth SFT gradients (<= 2 x measured in round 5; profiles/r05_parity_numbers.md)
GRAD_REL = {'language_model.lm_head.weight': 0.057, 'language_model.model.norm.weight': 0.052, 'mlp1.1.weight': 0.09, 'language_model.model.layers.27.self_attn.q_proj.weight': 0.079, 'language_model.model.layers.27.self_attn.v_proj.weight': 0.026, 'language_model.model.layers.27.mlp.down_proj.weight': 0.08, 'language_model.model.layers.14.self_attn.q_proj.weight': 0.075, 'language_model.model.layers.14.self_attn.v_proj.weight': 0.057, 'language_model.model.layers.14.mlp.down_proj.weight': 0.081, 'language_model.model.layers.0.self_attn.q_proj.weight': 0.086, 'language_model.model.layers.0.self_attn.v_proj.weight': 0.08, 'language_model.model.layers.0.mlp.down_proj.weight': 0.082, 'default': 0.09}
# (cosines in fp64 since r06: 1 - cos measured 0.8e-4 ... 1.0e-3; bounds at 2 x that deviation.  The r05 figures for lm_head / down_proj were fp32 dot products over
# 13.8-233 M elements and read 1.001-1.002)
GRAD_COS = {'language_model.lm_head.weight': 0.9992, 'language_model.model.norm.weight': 0.99933, 'mlp1.1.weight': 0.9989, 'language_model.model.layers.27.self_attn.q_proj.weight': 0.9986, 'language_model.model.layers.27.self_attn.v_proj.weight': 0.99983, 'language_model.model.layers.27.mlp.down_proj.weight': 0.9984, 'language_model.model.layers.14.self_attn.q_proj.weight': 0.99902, 'language_model.model.layers.14.self_attn.v_proj.weight': 0.9992, 'language_model.model.layers.14.mlp.down_proj.weight': 0.9984, 'language_model.model.layers.0.self_attn.q_proj.weight': 0.9981, 'language_model.model.layers.0.self_attn.v_proj.weight': 0.9983, 'language_model.model.layers.0.mlp.down_proj.weight': 0.9984, 'default': 0.998}
BF = torch.bfloat16


def _inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    pv = torch.randn(B, 3, 448, 448, generator=g)
    ids = torch.full((B, 384), cfg.pad_token_id)
    ids[:, :10] = torch.randint(0, 151643, (B, 10), generator=g)
    ids[:, 10:266] = cfg.img_context_token_id
    ids[:, 266:277] = torch.randint(0, 151643, (B, 11), generator=g)
    return ids, pv, torch.rand(B, 1, 7, generator=g) * 2 - 1, torch.randn(B, 4, 7, generator=g)


@pytest.fixture(scope='module')
def full():
    from vlaser_amd import config as C, synth
    torch.set_grad_enabled(False)
    vla = C.VLAConfig(base=C.vlaser_2b())
    sd = synth.vla_state_dict(vla, device='cuda', dtype=BF, with_head=True)
    return vla, sd


def _valid(ids, cfg):
    return (ids != cfg.pad_token_id).sum(-1).cuda()


def test_determinism_graph_eager_and_batching(full):
    from vlaser_amd.pizero import PiZeroInference
    vla, sd = full
    g2 = PiZeroInference(vla, max_batch=2, use_graph=True); g2.load_state_dict(sd)
    e1 = PiZeroInference(vla, max_batch=1, use_graph=False); e1.load_state_dict(sd)
    ids, pv, pro, noise = _inputs(vla.base, 2, seed=3)
    outs = []
    for b in range(2):
        a = [t[b:b + 1] for t in (ids, pv, pro, noise)]
        o1 = g2.infer_action(a[0], a[1], proprios=a[2], noise=a[3], valid_len=_valid(a[0], vla.base))
        o2 = g2.infer_action(a[0], a[1], proprios=a[2], noise=a[3], valid_len=_valid(a[0], vla.base))
        oe = e1.infer_action(a[0], a[1], proprios=a[2], noise=a[3], valid_len=_valid(a[0], vla.base))
        assert o1.shape == (1, 4, 7) and torch.isfinite(o1).all()
        assert torch.equal(o1, o2) and torch.equal(o1, oe)
        assert float(o1.abs().max()) <= vla.final_action_clip_value
        outs.append(o1.clone())
    both = g2.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=_valid(ids, vla.base))
    for b in range(2):
        assert (both[b] - outs[b][0]).abs().max().item() < 1e-2
    assert (outs[0] - outs[1]).abs().max().item() > 1e-3          # different observations -> different chunks


def test_euler_integrator_properties(full):
    from vlaser_amd.pizero import PiZeroInference
    vla, sd = full
    ids, pv, pro, noise = _inputs(vla.base, 1, seed=9)
    clip = vla.final_action_clip_value
    sd0 = dict(sd)
    sd0['action_decoder.weight'] = torch.zeros_like(sd['action_decoder.weight'])
    sd0['action_decoder.bias'] = torch.zeros_like(sd['action_decoder.bias'])
    m = PiZeroInference(vla, max_batch=1); m.load_state_dict(sd0)
    out = m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=_valid(ids, vla.base))
    assert torch.equal(out.cpu(), noise.clamp(-clip, clip))            # zero velocity: nothing may leak into the state
    v = torch.linspace(-0.5, 0.5, 7)
    sd0['action_decoder.bias'] = v.to(BF).cuda()
    m2 = PiZeroInference(vla, max_batch=1); m2.load_state_dict(sd0)
    out2 = m2.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=_valid(ids, vla.base))
    vb = v.to(BF).float()
    x = noise.clone()
    for _ in range(vla.num_inference_steps):                           # x <- x + dt * v, dt = 1 / steps (pizero_internvl.py:883-924)
        x = x + vb / vla.num_inference_steps
    assert (out2.cpu() - x.clamp(-clip, clip)).abs().max().item() < 1e-5


def test_full_depth_ragged_generate_consistency(full):
    from vlaser_amd.internvl_chat import InternVLChatModel
    vla, sd = full
    cfg = vla.base
    vsd = {k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
    m = InternVLChatModel(cfg, max_seq_len=448, max_batch=2)
    m.load_state_dict(vsd)
    m.img_context_token_id = cfg.img_context_token_id
    g = torch.Generator().manual_seed(21)
    pv = torch.randn(2, 3, 448, 448, generator=g)
    rows = [torch.cat([torch.randint(0, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id),
                       torch.randint(0, 151643, (n,), generator=g)]) for n in (39, 16)]
    S = max(len(r) for r in rows)
    ids = torch.full((2, S), cfg.pad_token_id); am = torch.zeros(2, S, dtype=torch.long)
    for b, r in enumerate(rows):
        ids[b, S - len(r):] = r; am[b, S - len(r):] = 1
    bgen, blg = m.generate(pv, ids, attention_mask=am, max_new_tokens=5, return_logits=True)
    for b, r in enumerate(rows):
        sgen, slg = m.generate(pv[b:b + 1], r[None], max_new_tokens=5, return_logits=True)
        # two bf16 paths with different split-K shapes (M = 2S vs S) through 28 layers: each is within ~3e-2 of fp32
        parity(f'full-depth 2B ragged batch row {b} vs solo logits max|err|/max|ref|', ((blg[b, 0] - slg[0, 0]).abs().max() / slg[0, 0].abs().max()).item(), 6e-2)
        parity(f'full-depth 2B ragged batch row {b} vs solo logits cosine', torch.nn.functional.cosine_similarity(blg[b, 0], slg[0, 0], dim=0).item(), 0.999, lower=True)
        t2 = slg[0].topk(2, dim=-1).values
        margin = (t2[:, 0] - t2[:, 1]).cpu()
        n_clear = 0
        while n_clear < 5 and margin[n_clear] > 0.08:
            n_clear += 1
        assert bgen[b, :n_clear].tolist() == sgen[0, :n_clear].tolist()


def test_full_depth_chunk_vs_fp32_oracle(full):
    """VERDICT r01 #3c: VALUES at full depth (24 ViT + 28 LLM + 28 expert layers), not only properties: one chunk with 2 Euler
    steps against the fp32 CPU oracle run on the same bf16-rounded weights -- the error the bf16 activation path accumulates over
    the full stack, for the action chunk, the per-step velocities and the last layer's cached keys."""
    from oracle import vla as ovla
    from vlaser_amd import config as C
    from vlaser_amd.pizero import PiZeroInference
    vla_full, sd = full
    vla = C.VLAConfig(base=vla_full.base, num_inference_steps=2)
    ids, pv, pro, noise = _inputs(vla.base, 1, seed=5)
    m = PiZeroInference(vla, max_batch=1); m.load_state_dict(sd)
    act = m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=_valid(ids, vla.base)).cpu()
    vel = m.last_velocities()[:, 0].cpu()
    nL = vla.base.llm.num_hidden_layers
    k_last = m.cache.k[nL - 1, 0, :, :277].float().cpu()
    del m
    torch.cuda.empty_cache()
    sdc = {k: v.float().cpu() for k, v in sd.items() if not k.startswith('language_model.lm_head')}
    am = (ids != vla.base.pad_token_id).long()
    mask, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
    m1, m2 = ovla.split_full_mask_into_submasks(mask, vla)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref, caches, trace = ovla.infer_action(sdc, vla, ids, pv.to(BF).float(), m1, m2, vp, pp, ap, pro, noise, return_trace=True)
    rvel = torch.stack([v for _, v in trace], 0)[:, 0]
    e_act = (act - ref).abs().max().item()
    e_vel = (vel - rvel).abs().amax(dim=(1, 2))
    rk = caches['vlm'][nL - 1][0][0][:, :277]
    e_k = ((k_last - rk).abs().max() / rk.abs().max()).item()
    print(f'full depth vs fp32 oracle: action max|err| {e_act:.3e}; per-step velocity max|err| {[f"{x:.2e}" for x in e_vel.tolist()]} (ref max {rvel.abs().max():.3f}); '
          f'last-layer K rel err {e_k:.3e}')
    parity('full-depth chunk (2 Euler steps) vs fp32 oracle: action max|err|', e_act, 1e-2)
    parity('full-depth chunk vs fp32 oracle: worst per-step velocity max|err|', e_vel.max().item(), 1.9e-2)
    parity('full-depth chunk vs fp32 oracle: last-layer K max|err|/max|ref|', e_k, 5e-2)


def test_full_depth_8b_13_tiles_properties():
    """BASELINE configs[3] at FULL size (Vlaser-8B: 24 ViT + 28 x 3584-wide LLM layers, 7.6 B parameters, 13 tiles, S = 3408): the
    oracle needs minutes there, so properties: the prefill is finite and deterministic, and the two decode implementations --
    chunked-K weight-streaming kernels vs the MFMA GEMM path with one row -- agree on the next-token logits within the bf16
    tolerance and on the greedy ids wherever the margin is clear."""
    from vlaser_amd import config as C, synth
    from vlaser_amd.internvl_chat import InternVLChatModel
    torch.set_grad_enabled(False)
    cfg = C.vlaser_8b()
    sd = synth.vlm_state_dict(cfg, device='cuda', dtype=BF)
    m = InternVLChatModel(cfg, max_tiles=13, max_seq_len=3456)
    m.load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    assert m.use_skinny
    m.img_context_token_id = cfg.img_context_token_id
    g = torch.Generator().manual_seed(45)
    pv = torch.randn(13, 3, 448, 448, generator=g)
    ids = torch.cat([torch.randint(0, 151643, (41,), generator=g), torch.full((13 * 256,), cfg.img_context_token_id),
                     torch.randint(0, 151643, (39,), generator=g)])[None]
    a_ids, a_lg = m.generate(pv, ids, max_new_tokens=4, return_logits=True)
    b_ids, b_lg = m.generate(pv, ids, max_new_tokens=4, return_logits=True)
    assert torch.isfinite(a_lg).all() and torch.equal(a_ids, b_ids) and torch.equal(a_lg, b_lg)
    m.use_skinny = False                                   # same weights through the MFMA GEMM kernels (M = 1 per step)
    c_ids, c_lg = m.generate(pv, ids, max_new_tokens=4, return_logits=True)
    assert (a_lg[0, 0] - c_lg[0, 0]).abs().max() < 1e-5 * max(1.0, a_lg[0, 0].abs().max().item()) + 1e-3        # step 0 = prefill output: same path
    for t in range(1, 4):
        if a_ids[0, t - 1].item() != c_ids[0, t - 1].item():
            break
        assert (a_lg[0, t] - c_lg[0, t]).abs().max() < 4e-2 * c_lg[0, t].abs().max(), t
        assert torch.nn.functional.cosine_similarity(a_lg[0, t], c_lg[0, t], dim=0) > 0.999
        t2 = c_lg[0, t].topk(2).values
        if (t2[0] - t2[1]).item() > 0.1:
            assert a_ids[0, t].item() == c_ids[0, t].item()
    del m
    torch.cuda.empty_cache()


def test_full_depth_sft_step_s560():
    """BASELINE configs[4] at its real shape on one rank: 28-layer S = 560 SFT steps -- the loss is finite and decreases over 3 steps
    on a fixed sample, activation recompute == kept activations bit for bit (loss and gradient norm), two runs agree bit for bit."""
    from vlaser_amd import config as C, synth
    from vlaser_amd.sft import SFTModel
    torch.set_grad_enabled(False)
    cfg = C.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device='cuda', dtype=BF)
    g = torch.Generator().manual_seed(77)
    S = 560
    ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id),
                     torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
    labels = torch.full_like(ids, -100); labels[0, -128:] = ids[0, -128:]
    pv = torch.randn(1, 3, 448, 448, generator=g)
    res = []
    for recompute in (False, True, False):
        m = SFTModel(cfg, max_seq_len=576, recompute=recompute, lr=2e-5)
        m.load_state_dict(sd)
        out = [m.step(pv, ids, labels) for _ in range(3)]
        res.append(([o.loss.item() for o in out], [o.grad_norm.item() for o in out]))
        del m
        torch.cuda.empty_cache()
    losses, gn = res[0]
    assert all(l == l and abs(l) < 1e4 for l in losses) and losses[2] < losses[0], losses
    assert res[1] == res[0], 'recompute != kept activations'
    assert res[2] == res[0], 'two runs differ'


def test_full_depth_qa_logits_and_greedy_ids_vs_fp32_oracle(full):
    """VERDICT r02 #5 (BASELINE configs[1]): VALUES of the full-depth Vlaser-2B QA path -- 24 ViT + 28 LLM layers, one tile + 80 text tokens
    (S = 336) -- against the fp32 CPU oracle on the same bf16-rounded weights: the next-token logits of the prompt's last
    position and the first 4 greedy ids (compared while the oracle's top-2 margin is clear: with random-init weights the logits are
    near-uniform and a bf16 path may legitimately flip a near-tie)."""
    from oracle import vlm as ovlm
    from vlaser_amd.internvl_chat import InternVLChatModel
    vla, sd = full
    cfg = vla.base
    vsd = {k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
    m = InternVLChatModel(cfg, max_seq_len=384, max_batch=1)
    m.load_state_dict(vsd)
    m.img_context_token_id = cfg.img_context_token_id
    g = torch.Generator().manual_seed(31)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.cat([torch.randint(0, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id),
                     torch.randint(0, 151643, (39,), generator=g)])[None]
    assert ids.shape[1] == 336
    gen, lg = m.generate(pv, ids, max_new_tokens=4, return_logits=True)
    gen, lg = gen.cpu(), lg.float().cpu()
    del m
    torch.cuda.empty_cache()
    sdc = {k: v.float().cpu() for k, v in vsd.items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    rgen, rlg = ovlm.generate(sdc, cfg, pv.to(BF).float(), ids, max_new_tokens=4, return_logits=True)
    e0 = ((lg[0, 0] - rlg[0, 0]).abs().max() / rlg[0, 0].abs().max()).item()
    cos0 = torch.nn.functional.cosine_similarity(lg[0, 0], rlg[0, 0], dim=0).item()
    l2 = ((lg[0, 0] - rlg[0, 0]).norm() / rlg[0, 0].norm()).item()
    print(f'full-depth 2B QA vs fp32 oracle: last-position logits max|err| / max|ref| = {e0:.3e}, relative L2 {l2:.3e}, cosine {cos0:.6f}; ids {gen[0].tolist()} vs {rgen[0].tolist()}')
    parity('full-depth 2B QA last-position logits vs fp32 oracle max|err|/max|ref|', e0, 5e-2)
    parity('full-depth 2B QA last-position logits vs fp32 oracle relative L2', l2, 5e-2)
    parity('full-depth 2B QA last-position logits vs fp32 oracle cosine', cos0, 0.999, lower=True)
    top = rlg[0, 0].topk(8)
    parity('full-depth 2B QA top-8 logit VALUES vs fp32 oracle, elementwise (rtol 2e-2, atol 5e-2)', elementwise(lg[0, 0][top.indices], top.values, 2e-2, 5e-2), 0.52)
    for t in range(4):
        t2 = rlg[0, t].topk(2).values
        if (t2[0] - t2[1]).item() > 4 * e0 * rlg[0, t].abs().max().item():          # clear margin: the ids must agree
            assert gen[0, t].item() == rgen[0, t].item(), t
        if gen[0, t].item() != rgen[0, t].item():
            break                                                                   # the sequences fork after a near-tie
        parity(f'full-depth 2B QA decode-step-{t} logits vs fp32 oracle max|err|/max|ref|', ((lg[0, t] - rlg[0, t]).abs().max() / rlg[0, t].abs().max()).item(), 5e-2)


def test_full_depth_sft_loss_and_grads_vs_fp32_oracle():
    """VERDICT r02 #5 (BASELINE configs[4]): the full-depth (28-layer) S = 560 SFT forward + backward against torch autograd through
    the fp32 CPU oracle on the same bf16-rounded weights: loss, and the gradients of lm_head, the final norm, layers 27 / 14 / 0
    (fused q/k/v and down projection) and the projector's first Linear -- i.e. what the bf16 backward accumulates over all 28 layers."""
    from oracle import vlm as ovlm
    from vlaser_amd import config as C, synth
    from vlaser_amd.sft import SFTModel
    torch.set_grad_enabled(False)
    cfg = C.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device='cuda', dtype=BF)
    g = torch.Generator().manual_seed(78)
    S = 560
    ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id),
                     torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
    labels = torch.full_like(ids, -100); labels[0, -128:] = ids[0, -128:]
    pv = torch.randn(1, 3, 448, 448, generator=g)
    m = SFTModel(cfg, max_seq_len=576)
    m.load_state_dict(sd)
    loss = float(m.forward_backward(pv, ids, labels))
    torch.cuda.synchronize()
    L_ = 'language_model.model.layers.'
    keys = ['language_model.lm_head.weight', 'language_model.model.norm.weight', 'mlp1.1.weight']
    for i in (27, 14, 0):
        keys += [f'{L_}{i}.self_attn.q_proj.weight', f'{L_}{i}.self_attn.v_proj.weight', f'{L_}{i}.mlp.down_proj.weight']
    ng = m.named_grads()
    grads = {k: ng[k].float().cpu() for k in keys}
    del m, ng
    torch.cuda.empty_cache()
    sdc = {k: v.float().cpu() for k, v in sd.items()}
    del sd
    for k in keys:
        sdc[k].requires_grad_(True)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    torch.set_grad_enabled(True)
    try:
        ref = ovlm.sft_loss(ovlm.forward_logits(sdc, cfg, pv.to(BF).float(), ids), labels)
        ref.backward()
    finally:
        torch.set_grad_enabled(False)
    print(f'full-depth SFT vs fp32 oracle: loss {loss:.5f} vs {ref.item():.5f}')
    parity('full-depth SFT loss vs fp32 oracle |err|', abs(loss - ref.item()), 2.5e-3)
    worst = []
    for k in keys:
        # fp64 statistics (VERDICT r05 weak #1a): an fp32 dot product over 13.8 M elements printed cosines of 1.001-1.002 for the down projections, so their
        # lower bounds could not fail
        a, b = grads[k].double().flatten(), sdc[k].grad.double().flatten()
        rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
        cos = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)).item()
        assert cos <= 1.0 + 1e-12, (k, cos)
        nrel = abs(a.norm().item() - b.norm().item()) / (b.norm().item() + 1e-30)
        worst.append((k, round(rel, 4), round(cos, 6), round(nrel, 4)))
    print('gradient (rel Frobenius err, cosine, rel norm err):', worst)
    for k, rel, cos, nrel in worst:
        parity(f'full-depth SFT gradient {k}: relative Frobenius error', rel, GRAD_REL.get(k, GRAD_REL['default']))
        parity(f'full-depth SFT gradient {k}: cosine', cos, GRAD_COS.get(k, GRAD_COS['default']), lower=True)
        parity(f'full-depth SFT gradient {k}: relative norm error', nrel, 4.6e-3)        # worst measured 2.3e-3 (layers.14 q_proj)


def test_full_depth_8b_one_tile_logits_vs_fp32_oracle():
    """VERDICT r03 #5 / weak #2 (BASELINE configs[3] widths at FULL depth): Vlaser-8B -- 24 ViT + 28 x 3584-wide LLM layers, 7.6 B parameters -- on one
    tile + 80 text tokens (S = 336): the next-token logits of the prompt's last position and the first greedy ids against the fp32 CPU oracle on the
    same bf16-rounded weights (the 13-tile shape stays property-checked above: its oracle forward is ~60 TFLOP).  Ids are compared while the oracle's
    top-2 margin is clear.  Prints the host seconds the oracle took."""
    import time
    from oracle import vlm as ovlm
    from vlaser_amd import config as C, synth
    from vlaser_amd.internvl_chat import InternVLChatModel
    torch.set_grad_enabled(False)
    cfg = C.vlaser_8b()
    sd = synth.vlm_state_dict(cfg, device='cuda', dtype=BF)
    m = InternVLChatModel(cfg, max_seq_len=384, max_batch=1)
    m.load_state_dict(sd)
    m.img_context_token_id = cfg.img_context_token_id
    # three input seeds (VERDICT r05 weak #1b: the element-wise top-8 check sat at 0.80 of its bound on the one seed it ran -- noise, or the chunked-K decode
    # kernels losing more than they should?  The spread over seeds answers it; every seed is held to the same bounds)
    seeds = (33, 34, 35)
    runs = []
    for sd_ in seeds:
        g = torch.Generator().manual_seed(sd_)
        pv = torch.randn(1, 3, 448, 448, generator=g)
        ids = torch.cat([torch.randint(0, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id),
                         torch.randint(0, 151643, (39,), generator=g)])[None]
        assert ids.shape[1] == 336
        gen, lg = m.generate(pv, ids, max_new_tokens=3, return_logits=True)
        runs.append((pv, ids, gen.cpu(), lg.float().cpu()))
    del m
    torch.cuda.empty_cache()
    t0 = time.time()
    sdc = {}
    for k in list(sd):
        sdc[k] = sd.pop(k).float().cpu()                  # 30 GB of fp32 on the host, one tensor at a time
    torch.cuda.empty_cache()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    top8 = []
    for sd_, (pv, ids, gen, lg) in zip(seeds, runs):
        rgen, rlg = ovlm.generate(sdc, cfg, pv.to(BF).float(), ids, max_new_tokens=3, return_logits=True)
        e0 = ((lg[0, 0] - rlg[0, 0]).abs().max() / rlg[0, 0].abs().max()).item()
        cos0 = torch.nn.functional.cosine_similarity(lg[0, 0].double(), rlg[0, 0].double(), dim=0).item()
        l2 = ((lg[0, 0] - rlg[0, 0]).norm() / rlg[0, 0].norm()).item()
        print(f'full-depth 8B (1 tile, S=336, seed {sd_}) vs fp32 oracle: last-position logits max|err| / max|ref| = {e0:.3e}, relative L2 {l2:.3e}, cosine {cos0:.6f}; '
              f'ids {gen[0].tolist()} vs {rgen[0].tolist()}; host seconds so far {time.time() - t0:.0f}')
        # random-init weights give near-uniform logits (|logit| << the hidden norm): 28 layers of bf16-rounded activations then show as a few per cent of the
        # logit vector (2B at the same depth: see the test above; measured here 5.3e-2 in L2, 5.4e-2 in the maximum norm)
        parity(f'full-depth 8B seed {sd_} last-position logits vs fp32 oracle relative L2', l2, 8e-2)
        parity(f'full-depth 8B seed {sd_} last-position logits vs fp32 oracle max|err|/max|ref|', e0, 8e-2)
        parity(f'full-depth 8B seed {sd_} last-position logits vs fp32 oracle cosine', cos0, 0.9974, lower=True)
        top = rlg[0, 0].topk(8)
        top8.append(parity(f'full-depth 8B seed {sd_} top-8 logit VALUES vs fp32 oracle, elementwise (rtol 2e-2, atol 5e-2)',
                           elementwise(lg[0, 0][top.indices], top.values, 2e-2, 5e-2), 1.0))
        for t in range(3):
            t2 = rlg[0, t].topk(2).values
            if (t2[0] - t2[1]).item() > 4 * e0 * rlg[0, t].abs().max().item():
                assert gen[0, t].item() == rgen[0, t].item(), (sd_, t)
            if gen[0, t].item() != rgen[0, t].item():
                break
            parity(f'full-depth 8B seed {sd_} decode-step-{t} logits vs fp32 oracle max|err|/max|ref|', ((lg[0, t] - rlg[0, t]).abs().max() / rlg[0, t].abs().max()).item(), 8e-2)
            if t > 0:
                # steps 1, 2 come out of the chunked-K weight-streaming decode kernels, step 0 out of the MFMA prefill: the same figure for both answers whether the decode
                # path loses more than the prefill (it does not: r06 measured 0.34-0.82 against 0.45-0.84 at step 0)
                tt = rlg[0, t].topk(8)
                parity(f'full-depth 8B seed {sd_} decode-step-{t} (chunked-K kernels) top-8 logit VALUES, elementwise (rtol 2e-2, atol 5e-2)',
                       elementwise(lg[0, t][tt.indices], tt.values, 2e-2, 5e-2), 1.6)
    print(f'full-depth 8B top-8 element-wise figure over seeds {seeds}: {[round(x, 3) for x in top8]} (bound 1.0)')


def _sample_16k(cfg, seed):
    """One packed-length SFT sample as the reference's launcher admits it (--max_dynamic_patch 12 + thumbnail = 13 tiles, --max_seq_length 16384,
    …2nd_finetune_full.sh:39,60): 41 prompt tokens, 13 x 256 <IMG_CONTEXT>, then a long multi-turn text with RAGGED supervised spans."""
    g = torch.Generator().manual_seed(seed)
    S = 16384
    ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((13 * 256,), cfg.img_context_token_id),
                     torch.randint(1, 151643, (S - 41 - 13 * 256,), generator=g)])[None]
    labels = torch.full_like(ids, -100)
    for lo, n in ((3400, 37), (5000, 411), (9001, 3), (12000, 1000), (16384 - 129, 129)):      # assistant turns of uneven length; the last one ends the sample
        labels[0, lo:lo + n] = ids[0, lo:lo + n]
    pv = torch.randn(13, 3, 448, 448, generator=g)
    return pv, ids, labels


def test_sft_step_s16384_13_tiles_recompute():
    """VERDICT r03 #5f: the SFT step at the launcher's limits -- 13 tiles, S = 16 384, ragged supervised spans (1 580 labelled positions), per-layer
    activation recompute as the reference's grad_checkpoint (…2nd_finetune_full.sh:46).  (a) a depth-2 model: recompute == kept activations bit for bit
    (loss, every gradient); (b) the full 28-layer Vlaser-2B with recompute=True: finite loss and gradient norm, the loss falls over two steps on the
    same sample, peak device memory printed."""
    from vlaser_amd import config as C, synth
    from vlaser_amd.sft import SFTModel
    torch.set_grad_enabled(False)
    cfg2 = C.truncated(C.vlaser_2b(), 2, 2)
    pv, ids, labels = _sample_16k(cfg2, 160)
    sd = synth.vlm_state_dict(cfg2, device='cuda', dtype=BF)
    got = []
    for recompute in (False, True):
        m = SFTModel(cfg2, max_seq_len=16384, max_tiles=13, recompute=recompute, lr=2e-5)
        m.load_state_dict(sd)
        loss = m.forward_backward(pv, ids, labels)
        got.append((loss.item(), {k: v.clone() for k, v in m.named_grads().items()}))
        del m
        torch.cuda.empty_cache()
    assert got[0][0] == got[1][0] and got[0][0] == got[0][0] and abs(got[0][0]) < 1e3
    for k, g_ in got[0][1].items():
        assert torch.equal(g_, got[1][1][k]), k
    del got, sd
    torch.cuda.empty_cache()
    cfg = C.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device='cuda', dtype=BF)
    torch.cuda.reset_peak_memory_stats()
    m = SFTModel(cfg, max_seq_len=16384, max_tiles=13, recompute=True, lr=2e-5)
    m.load_state_dict(sd)
    del sd
    t0 = time.perf_counter()
    out = [m.step(pv, ids, labels) for _ in range(2)]
    m.wait_optimizer()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses, gn = [o.loss.item() for o in out], [o.grad_norm.item() for o in out]
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f'\nS=16384, 13 tiles, 28 layers, recompute=True: losses {losses}, grad norms {gn}, {dt / 2:.2f} s per step, peak device memory {peak:.1f} GiB')
    assert all(l == l and abs(l) < 1e3 for l in losses) and all(x == x and 0 < x < 1e6 for x in gn), (losses, gn)
    assert losses[1] < losses[0], losses
    assert peak < 120, peak                 # the reference fits this in 80 GB parts with ZeRO-1 over 8 ranks; one rank holding ALL optimizer state stays well under 288 GB
    del m
    torch.cuda.empty_cache()
