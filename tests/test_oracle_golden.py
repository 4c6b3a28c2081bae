"""Pin the CPU oracle against golden vectors produced by the reference itself (tools/gen_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import qwen2, vit, vla as ovla, vlm as ovlm


def _check(d, prefix, t, rtol=2e-4, atol=2e-5):
    f = t.detach().float().flatten()
    assert list(t.shape) == list(d[prefix + '_shape'])
    got = f[torch.from_numpy(d[prefix + '_idx'])].numpy()
    np.testing.assert_allclose(got, d[prefix + '_val'], rtol=rtol, atol=atol)
    st = d[prefix + '_stats']
    dd = t.detach().double()
    np.testing.assert_allclose([dd.mean().item(), dd.abs().mean().item(), dd.norm().item()], st, rtol=1e-4, atol=1e-6)


@pytest.fixture(scope='module')
def g56(golden_dir):
    return np.load(os.path.join(golden_dir, 'g5g6_vlm.npz'))


def _inputs(seed, d):
    g = torch.Generator().manual_seed(seed)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    return pv, torch.from_numpy(d['input_ids'])


def test_pixel_shuffle_bit_exact(golden_dir):
    d = np.load(os.path.join(golden_dir, 'g3g4_shuffle_masks.npz'))
    x = torch.arange(2 * 32 * 32 * 8, dtype=torch.float32).reshape(2, 32, 32, 8)
    y = vit.pixel_shuffle(x, 0.5, 'v2')
    assert np.array_equal(y.numpy().astype(np.int32), d['ps_out'])
    # closed form from SURVEY §8 a5: out[n,i,j,a*2C+b*C+k] = x[n,2i+a,2j+b,k]
    C = 8
    for (i, j, a, b, k) in [(0, 0, 0, 0, 0), (3, 5, 1, 0, 2), (15, 15, 1, 1, 7), (7, 0, 0, 1, 4)]:
        assert y[1, i, j, a * 2 * C + b * C + k] == x[1, 2 * i + a, 2 * j + b, k]


def test_vla_masks_bit_exact(golden_dir, golden_model):
    _, vla, _ = golden_model
    d = np.load(os.path.join(golden_dir, 'g3g4_shuffle_masks.npz'))
    for n_valid in (277, 384, 1, 300):
        am = torch.zeros(2, 384, dtype=torch.long)
        am[0, :n_valid] = 1
        am[1, :max(1, n_valid - 17)] = 1
        m, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
        assert set(m.unique().tolist()) <= {0.0, torch.finfo(torch.float32).min}
        assert np.array_equal((m == 0).numpy().astype(np.uint8), d[f'mask_{n_valid}_zero'])
        m1, m2 = ovla.split_full_mask_into_submasks(m, vla)
        assert list(m1.shape) + list(m2.shape) == list(d[f'mask_{n_valid}_sub_shapes'])
        assert np.array_equal((m1 == 0).numpy().astype(np.uint8), d[f'mask_{n_valid}_sub1_zero'])
        assert np.array_equal((m2 == 0).numpy().astype(np.uint8), d[f'mask_{n_valid}_sub2_zero'])
        assert np.array_equal(vp.numpy(), d[f'pos_{n_valid}_vlm'])
        assert np.array_equal(pp.numpy(), d[f'pos_{n_valid}_pro'])
        assert np.array_equal(ap.numpy(), d[f'pos_{n_valid}_act'])


def test_vit_and_projector(g56, golden_model):
    cfg, _, sd = golden_model
    pv, _ = _inputs(0, g56)
    emb = vit.embeddings(sd, cfg.vision, pv)
    _check(g56, 'vit_emb', emb)
    h, layers = vit.vision_forward(sd, cfg.vision, pv, return_layers=True)
    for i, l in enumerate(layers):
        _check(g56, f'vit_l{i}', l)
    _check(g56, 'vit_feat', vit.extract_feature(sd, cfg, pv))


def test_vlm_logits_loss_greedy(g56, golden_model):
    cfg, _, sd = golden_model
    pv, ids = _inputs(0, g56)
    logits = ovlm.forward_logits(sd, cfg, pv, ids)
    _check(g56, 'logits', logits[:, -4:], rtol=5e-4, atol=5e-5)
    top = logits[0, -1].topk(8)
    assert np.array_equal(top.indices.numpy(), g56['last_top_ids'])
    np.testing.assert_allclose(top.values.numpy(), g56['last_top_vals'], rtol=1e-4, atol=1e-5)
    labels = torch.full_like(ids, -100)
    labels[0, -16:] = ids[0, -16:]
    np.testing.assert_allclose(ovlm.sft_loss(logits, labels).item(), float(g56['sft_loss']), rtol=1e-5)
    gen, lg = ovlm.generate(sd, cfg, pv, ids, max_new_tokens=8, eos_token_id=None, return_logits=True)
    assert np.array_equal(gen.numpy(), g56['greedy_ids'])
    np.testing.assert_allclose(lg[0].topk(4, dim=-1).values.numpy(), g56['greedy_top_vals'], rtol=1e-4, atol=2e-5)


def test_ragged_batch_generate(golden_dir, golden_model):
    """batch_chat's left-padded generate (modeling_internvl_chat.py:318-341): the oracle's mask/position handling against
    the reference's own HF generate on a 2-prompt batch of different lengths."""
    cfg, _, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g6b_ragged.npz'))
    pv = torch.cat([torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(int(s))) for s in d['seeds']])
    ids, am = torch.from_numpy(d['input_ids']), torch.from_numpy(d['attention_mask'])
    assert am[0].all() and not am[1].all() and am[1, -1] == 1          # row 1 is left padded
    gen, lg = ovlm.generate(sd, cfg, pv, ids, attention_mask=am, max_new_tokens=6, eos_token_id=None, return_logits=True)
    assert np.array_equal(gen.numpy(), d['greedy_ids'])
    np.testing.assert_allclose(lg.topk(4, dim=-1).values.numpy(), d['greedy_top_vals'], rtol=1e-4, atol=2e-5)


def test_sft_gradients_vs_reference_autograd(golden_dir, golden_model, g56):
    """The oracle's loss differentiated by torch autograd against the gradients of the REFERENCE model itself (G8): pins the
    checker that the HIP backward is compared with."""
    cfg, _, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g8_sft_grads.npz'))
    pv, ids = _inputs(0, g56)
    labels = torch.full_like(ids, -100)
    labels[0, -16:] = ids[0, -16:]
    names = [str(n) for n in d['names']]
    with torch.enable_grad():
        sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))}
        loss = ovlm.sft_loss(ovlm.forward_logits(sdg, cfg, pv, ids), labels)
        loss.backward()
    np.testing.assert_allclose(loss.item(), float(d['loss']), rtol=1e-5)
    assert len(names) == 33
    for n in names:
        g = sdg[n].grad.double().flatten()
        np.testing.assert_allclose(g.norm().item(), float(d[f'norm::{n}']), rtol=2e-4, err_msg=n)
        np.testing.assert_allclose(g[torch.from_numpy(d[f'idx::{n}'])].numpy(), d[f'val::{n}'], rtol=2e-3, atol=1e-6 * float(d[f'norm::{n}']) + 1e-9, err_msg=n)


def test_infer_action_integration_methods(golden_dir, golden_model):
    """G7c: the reference's chunks under integration_method = euler / heun / rk4 (its model_step closure returns one velocity per step: heun == euler bit for bit)."""
    import dataclasses
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    c = np.load(os.path.join(golden_dir, 'g7c_integrators.npz'))
    np.testing.assert_array_equal(c['a_heun_action'], c['a_euler_action'])
    np.testing.assert_array_equal(c['a_euler_action'], d['a_action'])
    case = 'a'
    g = torch.Generator().manual_seed(int(d[f'{case}_seed']))
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.from_numpy(d[f'{case}_input_ids'])
    am = (ids != vla.base.pad_token_id).long()
    m, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
    m1, m2 = ovla.split_full_mask_into_submasks(m, vla)
    acts = {}
    for method in ('heun', 'rk4'):
        v2 = dataclasses.replace(vla, integration_method=method)
        acts[method] = ovla.infer_action(sd, v2, ids, pv, m1, m2, vp, pp, ap, torch.from_numpy(d[f'{case}_proprio']), torch.from_numpy(d[f'{case}_noise']))
        np.testing.assert_allclose(acts[method].numpy(), c[f'{case}_{method}_action'], rtol=0, atol=2e-5)
    assert (acts['rk4'] - acts['heun']).abs().max().item() < 1e-6
    with pytest.raises(ValueError, match='Unknown integration method'):
        ovla.integration_step(torch.zeros(1), 0.1, torch.zeros(1), 'midpoint')


def g7d_case(d, case, vla):
    """Inputs of one G7d case rebuilt from the fixture: (ids, pixel_values, image_text_proprio_mask, action_mask, position ids x 3, proprio, noise)."""
    T, na = vla.max_image_text_tokens, vla.num_action_tokens
    Lt = T + 1 + na
    pv = torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(int(d[f'{case}_pixel_seed'])))
    ids = torch.from_numpy(d[f'{case}_input_ids'])
    vis = torch.from_numpy(np.unpackbits(d[f'{case}_mask_bits'])[:Lt * Lt].reshape(Lt, Lt).astype(bool))
    bias = torch.from_numpy(d[f'{case}_mask_bias']).float() if d[f'{case}_mask_bias'].size else torch.zeros(Lt, Lt)
    mask = torch.where(vis, bias, torch.full((), torch.finfo(torch.float32).min))[None, None]
    m1, m2 = ovla.split_full_mask_into_submasks(mask, vla)
    _, vp, pp, ap = ovla.build_causal_mask_and_position_ids((ids != vla.base.pad_token_id).long(), torch.float32, vla)
    return ids, pv, m1, m2, vp, pp, ap, torch.from_numpy(d[f'{case}_proprio']), torch.from_numpy(d[f'{case}_noise'])


def test_infer_action_general_masks(golden_dir, golden_model):
    """G7d: the reference's own chunks under masks its builder never produces (left-padded prompt, causal text, finite biases + a hole in an action row): the additive
    tensors go straight into eager attention (joint_model.py:636-656), and the oracle adds them the same way."""
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7d_general_masks.npz'))
    for case in ('a', 'b', 'c'):
        ids, pv, m1, m2, vp, pp, ap, pro, noise = g7d_case(d, case, vla)
        act = ovla.infer_action(sd, vla, ids, pv, m1, m2, vp, pp, ap, pro, noise)
        np.testing.assert_allclose(act.numpy(), d[f'{case}_action'], rtol=0, atol=2e-5)
    assert int((torch.from_numpy(d['a_input_ids'])[0, :50] == vla.base.pad_token_id).sum()) == 50          # case a really is left-padded


def test_infer_action(golden_dir, golden_model):
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    for case in ('a', 'b'):
        seed = int(d[f'{case}_seed'])
        g = torch.Generator().manual_seed(seed)
        pv = torch.randn(1, 3, 448, 448, generator=g)
        ids = torch.from_numpy(d[f'{case}_input_ids'])
        am = (ids != vla.base.pad_token_id).long()
        assert int(am.sum()) == int(d[f'{case}_n_valid'])
        m, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
        m1, m2 = ovla.split_full_mask_into_submasks(m, vla)
        act = ovla.infer_action(sd, vla, ids, pv, m1, m2, vp, pp, ap, torch.from_numpy(d[f'{case}_proprio']),
                                torch.from_numpy(d[f'{case}_noise']))
        np.testing.assert_allclose(act.numpy(), d[f'{case}_action'], rtol=0, atol=2e-5)


def test_infer_action_trace_kv_and_naive(golden_dir, golden_model):
    """G7b: per-Euler-step velocities, K / V cache slices (first / last layer) and the cache-free path, all from the reference's
    own infer_action / JointModel.forward (tools/gen_golden.py:g7b_trace)."""
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    t = np.load(os.path.join(golden_dir, 'g7b_vla_trace.npz'))
    for case in ('a', 'b'):
        g = torch.Generator().manual_seed(int(d[f'{case}_seed']))
        pv = torch.randn(1, 3, 448, 448, generator=g)
        ids = torch.from_numpy(d[f'{case}_input_ids'])
        am = (ids != vla.base.pad_token_id).long()
        m, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
        m1, m2 = ovla.split_full_mask_into_submasks(m, vla)
        pro, noise = torch.from_numpy(d[f'{case}_proprio']), torch.from_numpy(d[f'{case}_noise'])
        np.testing.assert_array_equal(t[f'{case}_action'], d[f'{case}_action'])          # same reference run as G7
        act, caches, trace = ovla.infer_action(sd, vla, ids, pv, m1, m2, vp, pp, ap, pro, noise, return_trace=True)
        vel = torch.stack([v for _, v in trace], 0)[:, 0]
        np.testing.assert_allclose(vel.numpy(), t[f'{case}_vel'], rtol=0, atol=5e-5)
        pos = torch.from_numpy(t[f'{case}_kv_pos'])
        for li in (0, int(t[f'{case}_n_layers']) - 1):
            k, v = caches['vlm'][li]
            np.testing.assert_allclose(k[0][:, pos].numpy(), t[f'{case}_k_vlm_L{li}'], rtol=0, atol=2e-5)
            np.testing.assert_allclose(v[0][:, pos].numpy(), t[f'{case}_v_vlm_L{li}'], rtol=0, atol=2e-5)
            k, v = caches['proprio'][li]
            np.testing.assert_allclose(k[0, :, 0].numpy(), t[f'{case}_k_pro_L{li}'], rtol=0, atol=2e-5)
            np.testing.assert_allclose(v[0, :, 0].numpy(), t[f'{case}_v_pro_L{li}'], rtol=0, atol=2e-5)
        naive, ntrace = ovla.infer_action_naive(sd, vla, ids, pv, m, vp, pp, ap, pro, noise, return_trace=True)
        np.testing.assert_allclose(naive.numpy(), t[f'{case}_action_naive'], rtol=0, atol=2e-5)
        np.testing.assert_allclose(torch.stack(ntrace, 0)[:, 0].numpy(), t[f'{case}_vel_naive'], rtol=0, atol=5e-5)
        # cached == naive in fp32 (the reference's own remark, eval.py:131-137)
        assert (act - naive[:, -vla.horizon_steps:]).abs().max().item() < 1e-5


def test_flow_matching_loss_and_grads_vs_reference_autograd(golden_dir, golden_model):
    """G10: loss and action-expert gradients of the reference's own flow-matching training forward (PiZero.forward) + autograd."""
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    f = np.load(os.path.join(golden_dir, 'g10_flow_matching.npz'))
    names = [str(n) for n in f['a_names']]
    case = 'a'
    g = torch.Generator().manual_seed(int(d[f'{case}_seed']))
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.from_numpy(d[f'{case}_input_ids'])
    am = (ids != vla.base.pad_token_id).long()
    m, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
    torch.set_grad_enabled(True)
    try:
        sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
        loss = ovla.flow_matching_loss(sdg, vla, ids, pv, m, vp, pp, ap, torch.from_numpy(d[f'{case}_proprio']), torch.from_numpy(f[f'{case}_actions']),
                                       torch.from_numpy(f[f'{case}_t']), torch.from_numpy(f[f'{case}_x0']))
        loss.backward()
    finally:
        torch.set_grad_enabled(False)
    np.testing.assert_allclose(loss.item(), float(f[f'{case}_loss']), rtol=2e-5)
    assert len(names) == 35
    for n in names:
        gr = sdg[n].grad.double().flatten()
        np.testing.assert_allclose(gr.norm().item(), float(f[f'{case}_norm::{n}']), rtol=5e-4, err_msg=n)
        np.testing.assert_allclose(gr[torch.from_numpy(f[f'{case}_idx::{n}'])].numpy(), f[f'{case}_val::{n}'], rtol=5e-3,
                                   atol=1e-6 * float(f[f'{case}_norm::{n}']) + 1e-10, err_msg=n)


def test_packed_sequence_loss_vs_reference(golden_dir, golden_model):
    """G11: the reference's own forward with `loss_weight` on a packed row (block-diagonal causal 4-D mask, restarting position ids)
    vs the oracle running the sub-sequences independently + the weighted loss."""
    cfg, _, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g11_packed.npz'))
    g = torch.Generator().manual_seed(int(d['seed']))
    ids = torch.from_numpy(d['input_ids'])
    # regenerate the pixel values exactly as the generator did: the id / text draws come first
    for n_img, n_text, n_lab in ((256, 22, 7), (0, 31, 9), (256, 15, 5)):
        torch.randint(1, 151643, (9,), generator=g); torch.randint(1, 151643, (n_text,), generator=g)
    pv = torch.randn(3, 3, 448, 448, generator=g)
    logits = ovlm.packed_logits(sd, cfg, pv, ids, torch.from_numpy(d['cu_seqlens']), torch.from_numpy(d['image_flags']))
    loss = ovlm.packed_loss(logits, torch.from_numpy(d['labels']), torch.from_numpy(d['loss_weight']))
    np.testing.assert_allclose(loss.item(), float(d['loss']), rtol=2e-5)
    np.testing.assert_allclose(logits[0, -1].topk(8).values.numpy(), d['last_logits'], rtol=0, atol=2e-4)


def test_flow_matching_vlm_group_grads_vs_reference(golden_dir, golden_model):
    """G10b: `train_vlm: True` -- the reference's `PiZero.forward` + autograd with `trainable_vlm_parameters` (pizero_internvl.py:405-411: vision
    tower, projector, the VLM mixture's decoder layers) unfrozen next to the action-expert group: same loss as G10, 90 gradient tensors; the
    VLM's last-layer post-attention parameters, its q projection there and its final norm receive none."""
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    f = np.load(os.path.join(golden_dir, 'g10b_flow_matching_vlm.npz'))
    f10 = np.load(os.path.join(golden_dir, 'g10_flow_matching.npz'))
    for case in ('a', 'b'):
        names, nograd = [str(n) for n in f[f'{case}_names']], [str(n) for n in f[f'{case}_nograd']]
        assert float(f[f'{case}_loss']) == float(f10[f'{case}_loss']) and set(str(n) for n in f10[f'{case}_names']) <= set(names) and len(names) == 90
        g = torch.Generator().manual_seed(int(d[f'{case}_seed']))
        pv = torch.randn(1, 3, 448, 448, generator=g)
        ids = torch.from_numpy(d[f'{case}_input_ids'])
        am = (ids != vla.base.pad_token_id).long()
        m, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
        torch.set_grad_enabled(True)
        try:
            sdg = {k: (v.clone().requires_grad_(True) if (k in names or k in nograd) and k in sd else v) for k, v in sd.items()}
            loss = ovla.flow_matching_loss(sdg, vla, ids, pv, m, vp, pp, ap, torch.from_numpy(d[f'{case}_proprio']), torch.from_numpy(f[f'{case}_actions']),
                                           torch.from_numpy(f[f'{case}_t']), torch.from_numpy(f[f'{case}_x0']))
            loss.backward()
        finally:
            torch.set_grad_enabled(False)
        np.testing.assert_allclose(loss.item(), float(f[f'{case}_loss']), rtol=2e-5)
        for n in names:
            gr = sdg[n].grad.double().flatten()
            np.testing.assert_allclose(gr.norm().item(), float(f[f'{case}_norm::{n}']), rtol=1e-3, err_msg=n)
            np.testing.assert_allclose(gr[torch.from_numpy(f[f'{case}_idx::{n}'])].numpy(), f[f'{case}_val::{n}'], rtol=1e-2,
                                       atol=2e-6 * float(f[f'{case}_norm::{n}']) + 1e-10, err_msg=n)
        for n in nograd:
            if n in sdg and sdg[n].requires_grad:
                assert sdg[n].grad is None or float(sdg[n].grad.abs().max()) == 0.0, n
