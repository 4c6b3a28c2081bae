"""Model-level parity of the HIP path (through the C ABI) against (a) golden vectors produced by the reference
itself and (b) the fp32 CPU oracle on the same seeded inputs.  GPU box only.

Stated tolerances (bf16 storage / fp32 accumulate vs an fp32 reference):
  hidden states, visual tokens, logits : max|err| <= 3e-2 * max|ref|   (measured 0.5-1.3e-2 at 2+2 layers)
  SFT loss                             : |err| <= 5e-3
  action chunk                         : max|err| <= 1e-2 (measured 3.6-6.4e-3; the reference's own bf16-vs-fp32 remark is ~1e-3 per
                                         cached step, eval.py:131, and 10 Euler steps integrate it)
  greedy token ids, visual-token indices, top-k ids : bit-exact (golden margins >= 0.09 logit units >> bf16 noise)
"""
import os

import numpy as np
import pytest
import torch

from parity import elementwise, parity, relmax

pytestmark = pytest.mark.gpu

# Asserted bounds = at most 2 x the worst value measured on MI355X in round 5 (profiles/r05_parity_numbers.md lists every measured figure next to its bound)
TOL = {'vit_emb': 1.1e-2, 'vit_layer': 2.1e-2, 'vit_feat': 1.6e-2, 'logits': 2.7e-2, 'logits_8b': 2.9e-2, 'loss': 2e-3, 'graph_vs_eager': 2e-3, 'action': 1e-2,
       'action_rel': 2.5e-2, 'velocity': 1.5e-2, 'kv': 2.2e-2, 'naive_vs_cached': 3.4e-3}
EW = 0.7          # element-wise checks (worst |err| / (atol + rtol |ref|)): measured 0.23-0.45


@pytest.fixture(scope='module')
def vlm(golden_model):
    from vlaser_amd.internvl_chat import InternVLChatModel
    cfg, _, sd = golden_model
    m = InternVLChatModel(cfg, max_seq_len=512)
    m.load_state_dict(sd, strict=False)      # VLA superset: action_expert.* etc. are unexpected keys for the chat model
    m.img_context_token_id = cfg.img_context_token_id
    return m


@pytest.fixture(scope='module')
def g56(golden_dir):
    return np.load(os.path.join(golden_dir, 'g5g6_vlm.npz'))


def _pv(seed):
    return torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(seed))


def _golden_close(d, prefix, t, tol, name=None):
    f = t.detach().float().cpu().flatten()
    got = f[torch.from_numpy(d[prefix + '_idx'])].numpy()
    ref = d[prefix + '_val']
    return parity(name or f'golden {prefix} max|err|/max|ref|', np.abs(got - ref).max() / np.abs(ref).max(), tol)


def test_vit_layers_vs_oracle_and_golden(vlm, g56, golden_model):
    from oracle import vit as ovit
    cfg, _, sd = golden_model
    pv = _pv(0)
    feat, layers = vlm.vit.forward(vlm._to_bf16(pv), return_layers=True)
    _, olayers = ovit.vision_forward(sd, cfg.vision, pv, return_layers=True)
    parity('vit embeddings vs oracle max|err|/max|ref|', relmax(layers[0].view(1, 1025, 1024), ovit.embeddings(sd, cfg.vision, pv)), TOL['vit_emb'])
    for i, (a, b) in enumerate(zip(layers[1:], olayers)):
        parity(f'vit layer {i} vs oracle max|err|/max|ref|', relmax(a.view(1, 1025, 1024), b), TOL['vit_layer'])
        parity(f'vit layer {i} vs oracle elementwise (rtol 2e-2, atol 2e-2 max|ref|)', elementwise(a.view(1, 1025, 1024), b, 2e-2, 2e-2 * b.abs().max().item()), EW)
        _golden_close(g56, f'vit_l{i}', a.view(1, 1025, 1024), TOL['vit_layer'])
    f = vlm.extract_feature(pv)
    assert f.shape == (1, 256, cfg.llm.hidden_size)
    of = ovit.extract_feature(sd, cfg, pv)
    parity('visual tokens vs oracle max|err|/max|ref|', relmax(f, of), TOL['vit_feat'])
    parity('visual tokens vs oracle elementwise (rtol 2e-2, atol 2e-2 max|ref|)', elementwise(f, of, 2e-2, 2e-2 * of.abs().max().item()), EW)
    _golden_close(g56, 'vit_feat', f, TOL['vit_feat'])


def test_pixel_shuffle_surface_bit_exact(vlm, golden_dir):
    d = np.load(os.path.join(golden_dir, 'g3g4_shuffle_masks.npz'))
    x = torch.arange(2 * 32 * 32 * 8, dtype=torch.float32).reshape(2, 32, 32, 8) % 251     # exact in bf16
    y = vlm.pixel_shuffle(x.cuda(), 0.5)
    ref = torch.from_numpy(d['ps_out']).float() % 251
    assert torch.equal(y.float().cpu(), ref)


def test_logits_loss_topk(vlm, g56, golden_model):
    from oracle import vlm as ovlm
    cfg, _, sd = golden_model
    pv, ids = _pv(0), torch.from_numpy(g56['input_ids'])
    out = vlm.forward(pv, ids, image_flags=torch.ones(1, 1, dtype=torch.long))
    ol = ovlm.forward_logits(sd, cfg, pv, ids)
    parity('logits (all positions) vs oracle max|err|/max|ref|', relmax(out.logits, ol), TOL['logits'])
    _golden_close(g56, 'logits', out.logits[:, -4:], TOL['logits'])
    assert out.logits[0, -1].topk(8).indices.tolist() == g56['last_top_ids'].tolist()
    # element-wise on the values that decide the ids: the top-8 logits of the last position, each within atol + rtol |ref| of the fp32 oracle's
    top = ol[0, -1].topk(8)
    parity('last-position top-8 logit VALUES vs oracle, elementwise (rtol 1e-2, atol 2e-2)', elementwise(out.logits[0, -1].cpu()[top.indices], top.values, 1e-2, 2e-2), EW)
    # visual-token indices: rank workspace holds the scatter map
    sel = (ids.flatten() == cfg.img_context_token_id)
    rank = vlm.rank_ws[:ids.numel()].cpu()
    assert torch.equal(rank[sel], torch.arange(256, dtype=torch.int32)) and bool((rank[~sel] == -1).all())
    assert sel.nonzero().flatten().tolist() == list(range(41, 297))
    labels = torch.full_like(ids, -100)
    labels[0, -16:] = ids[0, -16:]
    out2 = vlm.forward(pv, ids, image_flags=torch.ones(1, 1, dtype=torch.long), labels=labels)
    parity('SFT loss vs golden |err|', abs(out2.loss.item() - float(g56['sft_loss'])), TOL['loss'])


def test_graph_decode_equals_eager_decode(vlm, g56, golden_model):
    """Uniform batches replay one captured decode step (device-resident slot / visible-key state): the ids must equal the eager step-by-step
    decode's, the logits agree to bf16 noise (the key-chunk schedule is sized for the final length instead of the current one), a second
    call replays the cached graph, and a batch of two equal prompts gives two equal rows."""
    from vlaser_amd.internvl_chat import InternVLChatModel
    cfg, _, sd = golden_model
    pv, ids = _pv(0), torch.from_numpy(g56['input_ids'])
    eager = InternVLChatModel(cfg, max_seq_len=512, max_batch=2, decode_graph=False)
    eager.load_state_dict(sd, strict=False)
    eager.img_context_token_id = cfg.img_context_token_id
    ge, le = eager.generate(pv, ids, max_new_tokens=12, return_logits=True)
    for rep in range(2):                                        # second pass: graph cached from the first
        gg, lg = vlm.generate(pv, ids, max_new_tokens=12, return_logits=True)
        assert gg.cpu().tolist() == ge.cpu().tolist()
        parity('graph decode vs eager decode logits max|err|/max|ref|', (lg - le).abs().max().item() / le.abs().max().item(), TOL['graph_vs_eager'])
    assert any(not isinstance(g, str) for g in vlm._dec_graphs.values())      # a graph was really captured and replayed
    two = vlm.generate(torch.cat([pv, pv]), torch.cat([ids, ids]), max_new_tokens=12)
    assert two[0].tolist() == two[1].tolist() == ge[0].tolist()


def test_greedy_ids_bit_exact(vlm, g56):
    pv, ids = _pv(0), torch.from_numpy(g56['input_ids'])
    assert float(g56['greedy_margin'].min()) > 0.05           # golden margins are far above bf16 logit noise
    gen, lg = vlm.generate(pv, ids, max_new_tokens=8, return_logits=True)
    assert gen.cpu().tolist() == g56['greedy_ids'].tolist()
    top = lg[0].topk(4, dim=-1).values.cpu().numpy()
    parity('greedy top-4 logit values vs golden max|err|/max|ref|', np.abs(top - g56['greedy_top_vals']).max() / np.abs(g56['greedy_top_vals']).max(), TOL['logits'])
    parity('greedy top-4 logit values vs golden elementwise (rtol 1e-2, atol 2e-2)', elementwise(top, g56['greedy_top_vals'], 1e-2, 2e-2), EW)
    # eos handling: stop at the first generated token when it is declared eos
    first = int(g56['greedy_ids'][0, 0])
    gen2 = vlm.generate(pv, ids, max_new_tokens=8, eos_token_id=first)
    assert gen2.cpu().tolist() == [[first]]


def test_ragged_batch_generate_bit_exact(vlm, golden_dir):
    """Left-padded 2-prompt batch (batch_chat's generate call): ids bit-exact against the reference's HF generate, and the
    same ids when the caller pads on the right instead."""
    d = np.load(os.path.join(golden_dir, 'g6b_ragged.npz'))
    assert float(d['greedy_margin'].min()) > 0.04
    pv = torch.cat([_pv(int(s)) for s in d['seeds']])
    ids, am = torch.from_numpy(d['input_ids']), torch.from_numpy(d['attention_mask'])
    gen, lg = vlm.generate(pv, ids, attention_mask=am, max_new_tokens=6, return_logits=True)
    assert gen.cpu().tolist() == d['greedy_ids'].tolist()
    top = lg.topk(4, dim=-1).values.cpu().numpy()
    parity('ragged batch top-4 logit values vs golden max|err|/max|ref|', np.abs(top - d['greedy_top_vals']).max() / np.abs(d['greedy_top_vals']).max(), TOL['logits'])
    parity('ragged batch top-4 logit values vs golden elementwise (rtol 1e-2, atol 2e-2)', elementwise(top, d['greedy_top_vals'], 1e-2, 2e-2), 0.9)
    n1 = int(am[1].sum())
    ids_r, am_r = ids.clone(), am.clone()
    ids_r[1, :n1], ids_r[1, n1:] = ids[1, -n1:], 151643
    am_r[1, :n1], am_r[1, n1:] = 1, 0
    assert vlm.generate(pv, ids_r, attention_mask=am_r, max_new_tokens=6).cpu().tolist() == d['greedy_ids'].tolist()
    # each prompt alone gives the same continuation
    solo = vlm.generate(pv[1:], ids[1:, -n1:], max_new_tokens=6)
    assert solo.cpu().tolist() == d['greedy_ids'][1:].tolist()


class _StubTokenizer:
    """Test stand-in for the HF Qwen2 tokenizer (absent on the GPU box).  Special tokens map to their real ids; the
    golden chat prompt maps to the ids the real tokenizer produced for it (tests/golden/g1_prompts.json); any other text
    is split on whitespace and hashed, which is enough to drive chat()/batch_chat() plumbing deterministically."""
    SPECIAL = {'<IMG_CONTEXT>': 151667, '<img>': 151665, '</img>': 151666, '<|im_end|>': 151645, '<|endoftext|>': 151643,
               '<|im_start|>': 151644}

    def __init__(self, golden):
        import hashlib
        self._sha = lambda q: hashlib.sha256(q.encode('utf-8')).hexdigest()
        c = golden['chat_1tile']
        non = list(c['ids_nonimg'])
        self.known = {c['prompt_sha']: non[:c['img_first']] + [151667] * c['img_count'] + non[c['img_first']:]}
        self.padding_side = 'right'

    def convert_tokens_to_ids(self, t):
        return self.SPECIAL[t]

    def _encode(self, q):
        import re
        import zlib
        if self._sha(q) in self.known:
            return self.known[self._sha(q)]
        out = []
        for piece in re.split('(' + '|'.join(re.escape(k) for k in self.SPECIAL) + ')', q):
            if piece in self.SPECIAL:
                out.append(self.SPECIAL[piece])
            else:
                out += [zlib.crc32(w.encode('utf-8')) % 151643 for w in piece.split()]
        return out

    def __call__(self, queries, return_tensors='pt', padding=False):
        rows = [self._encode(q) for q in ([queries] if isinstance(queries, str) else queries)]
        S = max(len(r) for r in rows)
        ids = torch.full((len(rows), S), 151643, dtype=torch.long)
        am = torch.zeros(len(rows), S, dtype=torch.long)
        for b, r in enumerate(rows):
            sl = slice(S - len(r), S) if self.padding_side == 'left' else slice(0, len(r))
            ids[b, sl] = torch.tensor(r)
            am[b, sl] = 1
        return {'input_ids': ids, 'attention_mask': am}

    def batch_decode(self, ids, skip_special_tokens=True):
        return [' '.join(f't{int(i)}' for i in row if not (skip_special_tokens and int(i) >= 151643)) for row in ids]


def test_chat_and_batch_chat_surfaces(vlm, golden_dir):
    """chat() (modeling_internvl_chat.py:343-398) and batch_chat() (:293-341): prompt assembly -> ids -> greedy decode
    -> response split; the batched (left-padded) path returns what the one-by-one path returns."""
    import json
    g = json.load(open(os.path.join(golden_dir, 'g1_prompts.json')))
    tok = _StubTokenizer(g)
    vlm.system_message = g['system_message']
    q1 = g['chat_1tile']['question']
    q2 = 'Where is the red cup?'
    pv = torch.cat([_pv(3), _pv(4)])
    gc = dict(max_new_tokens=5, do_sample=False)
    r1, hist = vlm.chat(tok, pv[:1], q1, dict(gc), return_history=True)
    assert hist == [('<image>\n' + q1, r1)] and r1.startswith('t') and len(r1.split()) <= 5
    # the ids chat() fed to generate() are the real tokenizer's ids for this prompt
    ids1 = tok(vlm_query(vlm, q1))['input_ids']
    assert ids1.shape[1] == g['chat_1tile']['n_tokens']
    direct = vlm.generate(pv[:1], ids1, max_new_tokens=5, eos_token_id=151645)
    assert tok.batch_decode(direct)[0] == r1
    gc2 = dict(gc)
    both = vlm.batch_chat(tok, pv, [q1, q2], gc2, num_patches_list=[1, 1])
    # batch_chat == decode(generate(left-padded batch)); against the one-by-one path the ids agree up to the first step
    # whose top-2 logit margin is inside bf16 noise (random-init logits are nearly flat; the golden-margin cases are
    # test_ragged_batch_generate_bit_exact), and the logits agree within the stated tolerance
    tok.padding_side = 'left'
    mi = tok([vlm_query(vlm, q1), vlm_query(vlm, q2)], padding=True)
    bgen, blg = vlm.generate(pv, mi['input_ids'], attention_mask=mi['attention_mask'], max_new_tokens=5, eos_token_id=151645,
                             return_logits=True)
    assert both == [r.strip() for r in tok.batch_decode(bgen)]
    for b, q in enumerate((q1, q2)):
        n = int(mi['attention_mask'][b].sum())
        sgen, slg = vlm.generate(pv[b:b + 1], mi['input_ids'][b:b + 1, -n:], max_new_tokens=5, eos_token_id=151645, return_logits=True)
        assert (blg[b, 0] - slg[0, 0]).abs().max() < 3e-2 * slg[0, 0].abs().max()
        t2 = slg[0].topk(2, dim=-1).values
        margin = (t2[:, 0] - t2[:, 1]).cpu()
        n_clear = 0
        while n_clear < min(sgen.shape[1], bgen.shape[1]) and n_clear < margin.shape[0] and margin[n_clear] > 0.08:
            n_clear += 1
        assert bgen[b, :n_clear].tolist() == sgen[0, :n_clear].tolist()
    assert gc2['eos_token_id'] == 151645 and tok.padding_side == 'left'      # same side effects as the reference
    with pytest.raises(NotImplementedError):
        vlm.batch_chat(tok, pv, [q1, q2], dict(gc), num_patches_list=[1, 1], return_history=True)
    # text-only chat (pixel_values=None, :347-349)
    r3 = vlm.chat(tok, None, 'Hello there', dict(gc))
    assert isinstance(r3, str)


def vlm_query(vlm, question):
    from vlaser_amd import prep
    return prep.build_chat_query(vlm.template, vlm.system_message, question, [1], vlm.num_image_token, None, True)[0]


def test_mismatched_image_tokens_raises(vlm, g56):
    pv, ids = _pv(0), torch.from_numpy(g56['input_ids']).clone()
    ids[0, 100] = 5        # 255 <IMG_CONTEXT> tokens for 256 visual tokens
    with pytest.raises(RuntimeError):
        vlm.generate(pv, ids, max_new_tokens=2)


@pytest.fixture(scope='module')
def pz(golden_model):
    from vlaser_amd.pizero import PiZeroInference
    _, vla, sd = golden_model
    m = PiZeroInference(vla, max_batch=2)
    m.load_state_dict(sd)
    return m


def _vla_inputs(d, case, pz):
    seed = int(d[f'{case}_seed'])
    pv = _pv(seed)
    ids = torch.from_numpy(d[f'{case}_input_ids'])
    am = (ids != 151643).long()
    mask, vp, pp, ap = pz.build_causal_mask_and_position_ids(am, torch.float32)
    m1, m2 = pz.split_full_mask_into_submasks(mask)
    return ids, pv, m1, m2, vp, pp, ap, torch.from_numpy(d[f'{case}_proprio']), torch.from_numpy(d[f'{case}_noise'])


def test_infer_action_vs_golden(pz, golden_dir):
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    for case in ('a', 'b'):
        ids, pv, m1, m2, vp, pp, ap, pro, noise = _vla_inputs(d, case, pz)
        act = pz.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)
        assert act.shape == (1, 4, 7)
        parity(f'action chunk {case} vs golden G7 max|err|', (act.cpu() - torch.from_numpy(d[f'{case}_action'])).abs().max().item(), TOL['action'])
        act2 = pz.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)       # graph replay is deterministic
        assert torch.equal(act, act2)


def test_infer_action_integration_methods_vs_golden(golden_model, golden_dir):
    """VERDICT r05 missing #6: `integration_method` heun / rk4 (pizero_internvl.py:164,910-922,1309-1331) against the reference's own chunks (golden G7c) at the Euler
    tolerance, and against this path's own Euler chunk: heun bit-identical (the reference's is too), rk4 within a few fp32 ulp (the reference: 2.4e-7)."""
    import dataclasses
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    c = np.load(os.path.join(golden_dir, 'g7c_integrators.npz'))
    got = {}
    for method in ('euler', 'heun', 'rk4'):
        m = PiZeroInference(dataclasses.replace(vla, integration_method=method), max_batch=1)
        m.load_state_dict(sd)
        for case in ('a', 'b'):
            ids, pv, m1, m2, vp, pp, ap, pro, noise = _vla_inputs(d, case, m)
            act = m.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise).cpu()
            got[method, case] = act
            parity(f'action chunk {case}, integration_method={method} vs golden G7c max|err|', (act - torch.from_numpy(c[f'{case}_{method}_action'])).abs().max().item(), TOL['action'])
        del m
    for case in ('a', 'b'):
        assert torch.equal(got['heun', case], got['euler', case])
        assert (got['rk4', case] - got['euler', case]).abs().max().item() < 2e-6
    with pytest.raises(ValueError, match='Unknown integration method'):
        dataclasses.replace(vla, integration_method='midpoint')


def test_infer_action_internals_vs_reference_trace(pz, golden_dir):
    """VERDICT r01 #3a/b: the action chunk within 1e-2 (and relative to the model-dependent part of the signal, action - clip(noise)),
    the decoder velocity of EVERY Euler step, and the cached K / V of the first and last layer (VLM positions + proprio token)
    against what the reference's own infer_action produced internally (golden G7b)."""
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    t = np.load(os.path.join(golden_dir, 'g7b_vla_trace.npz'))
    T = pz.max_image_text_tokens
    for case in ('a', 'b'):
        ids, pv, m1, m2, vp, pp, ap, pro, noise = _vla_inputs(d, case, pz)
        act = pz.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise).cpu()
        ref = torch.from_numpy(d[f'{case}_action'])
        err = (act - ref).abs().max().item()
        sig = (ref - noise.clamp(-1, 1))
        live = ref.abs() < 1.0                                             # entries the final clip did not saturate
        rel = ((act - ref)[live].norm() / sig[live].norm()).item()
        parity(f'action chunk {case} vs G7b max|err|', err, TOL['action'])
        parity(f'action chunk {case} vs G7b error relative to the model-dependent signal', rel, TOL['action_rel'])
        vel = pz.last_velocities()[:, 0].cpu()
        rv = torch.from_numpy(t[f'{case}_vel'])
        verr = (vel - rv).abs().amax(dim=(1, 2))
        parity(f'velocity of every Euler step {case} vs G7b worst max|err| / max(1, max|ref|)', verr.max().item() / max(1.0, rv.abs().max().item()), TOL['velocity'])
        pos = torch.from_numpy(t[f'{case}_kv_pos']).long()
        for li in (0, int(t[f'{case}_n_layers']) - 1):
            k = pz.cache.k[li, 0].float().cpu()                            # [n_kv, S_max, 128]
            vt = pz.cache.vt[li, 0].float().cpu()                          # [n_kv, 128, S_max]
            for name, got, want in [('k_vlm', k[:, pos], t[f'{case}_k_vlm_L{li}']), ('v_vlm', vt[:, :, pos].transpose(1, 2), t[f'{case}_v_vlm_L{li}']),
                                    ('k_pro', k[:, T], t[f'{case}_k_pro_L{li}']), ('v_pro', vt[:, :, T], t[f'{case}_v_pro_L{li}'])]:
                want = torch.from_numpy(want)
                e = (got - want).abs().max().item()
                parity(f'cached {name} layer {li} case {case} vs reference cache max|err|/max|ref|', e / want.abs().max().item(), TOL['kv'])


def test_infer_action_naive_equals_cached(golden_model, golden_dir):
    """VERDICT r01 #3d: the cache-free surface (`infer_action_naive`: every step re-runs the joint pass; here on the MFMA GEMM +
    prefill-attention kernels for the expert rows) against the cached path (weight-streaming kernels) and against the reference's
    cache-free result (G7b)."""
    from vlaser_amd.pizero import PiZeroInference
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    t = np.load(os.path.join(golden_dir, 'g7b_vla_trace.npz'))
    m = PiZeroInference(vla, max_batch=1, naive_support=True); m.load_state_dict(sd)
    for case in ('a', 'b'):
        ids, pv, m1, m2, vp, pp, ap, pro, noise = _vla_inputs(d, case, m)
        cached = m.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise).cpu()
        am = (ids != 151643).long()
        mask, _, _, _ = m.build_causal_mask_and_position_ids(am, torch.float32)
        naive = m.infer_action_naive(ids, pv, mask, vp, pp, ap, pro, noise=noise).cpu()
        dn = (naive - cached).abs().max().item()
        dr = (naive - torch.from_numpy(t[f'{case}_action_naive'])).abs().max().item()
        parity(f'infer_action_naive vs cached {case} max|err|', dn, TOL['naive_vs_cached'])
        parity(f'infer_action_naive vs reference naive {case} max|err|', dr, TOL['action'])


def test_infer_text_equals_chat_model_logits(golden_model):
    """VERDICT r01 #3d: `infer_text` (VLM mixture alone through the joint path + lm_head, pizero_internvl.py:1005-1046) must
    reproduce InternVLChatModel's logits."""
    from oracle import vlm as ovlm
    from vlaser_amd.internvl_chat import InternVLChatModel
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    g = torch.Generator().manual_seed(3)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.cat([torch.randint(0, 151643, (12,), generator=g), torch.full((256,), cfg.img_context_token_id),
                     torch.randint(0, 151643, (20,), generator=g)])[None]
    pz_ = PiZeroInference(vla, max_batch=1); pz_.load_state_dict(sd)
    lt = pz_.infer_text(ids, pv, torch.ones_like(ids))['logits'].cpu()
    chat = InternVLChatModel(cfg, max_seq_len=320); chat.load_state_dict({k: v for k, v in sd.items() if k.startswith(('vision_model.', 'mlp1.', 'language_model.'))})
    chat.img_context_token_id = cfg.img_context_token_id
    lc = chat.forward(pv, ids, image_flags=torch.ones(1, 1, dtype=torch.long)).logits.cpu()
    assert lt.shape == lc.shape
    parity('infer_text vs chat model logits max|err|/max|ref| (same kernels, same weights)', (lt - lc).abs().max().item() / lc.abs().max().item(), 2e-3)
    ref = ovlm.forward_logits(sd, cfg, pv, ids)
    parity('infer_text last-4 logits vs oracle max|err|/max|ref|', (lt[0, -4:] - ref[0, -4:]).abs().max().item() / ref[0, -4:].abs().max().item(), TOL['logits'])
    assert lt[0, -1].argmax().item() == ref[0, -1].argmax().item()


def test_infer_action_batch2_equals_singles(pz, golden_dir):
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    a, b = _vla_inputs(d, 'a', pz), _vla_inputs(d, 'b', pz)
    cat = [torch.cat([x, y], 0) for x, y in zip(a, b)]
    both = pz.infer_action(*cat[:8], noise=cat[8])
    ra = torch.from_numpy(d['a_action']); rb = torch.from_numpy(d['b_action'])
    parity('batch-2 chunk vs golden max|err|', max((both[0].cpu() - ra[0]).abs().max().item(), (both[1].cpu() - rb[0]).abs().max().item()), TOL['action'])


def test_graph_equals_eager(golden_model, golden_dir):
    from vlaser_amd.pizero import PiZeroInference
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    e = PiZeroInference(vla, max_batch=1, use_graph=False); e.load_state_dict(sd)
    g = PiZeroInference(vla, max_batch=1, use_graph=True); g.load_state_dict(sd)
    args = _vla_inputs(d, 'a', e)
    assert torch.equal(e.infer_action(*args[:8], noise=args[8]), g.infer_action(*args[:8], noise=args[8]))


def test_proprio_row_riding_with_step0_is_bit_identical(golden_model, golden_dir):
    """Batch 1 runs the proprio token in front of the action rows of Euler step 0 (M = 5) instead of in its own pass through the
    expert: actions and the cached proprio K / V^T (slot 384 of every layer) must equal the separate pass bit for bit."""
    from vlaser_amd.pizero import PiZeroInference
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    a = PiZeroInference(vla, max_batch=1, use_graph=False, ride_proprio=True); a.load_state_dict(sd)
    b = PiZeroInference(vla, max_batch=1, use_graph=False, ride_proprio=False); b.load_state_dict(sd)
    for case in ('a', 'b'):
        args = _vla_inputs(d, case, a)
        ra, rb = a.infer_action(*args[:8], noise=args[8]), b.infer_action(*args[:8], noise=args[8])
        assert torch.equal(ra, rb)
        T = a.max_image_text_tokens
        assert torch.equal(a.cache.k[:, :, :, T:T + 5], b.cache.k[:, :, :, T:T + 5])
        assert torch.equal(a.cache.vt[:, :, :, :, T:T + 5], b.cache.vt[:, :, :, :, T:T + 5])
        assert a.cache.k[:, :, :, T].abs().sum() > 0


def test_vlaser_8b_widths_multi_tile_generate():
    """BASELINE configs[3] geometry at true 8B widths (hidden 3584, 28 q / 4 kv heads, MLP 18944), depth-truncated, 2 tiles:
    prefill + greedy decode vs the fp32 CPU oracle.  The 3584-wide decoder decodes on the weight-streaming kernels through their
    chunked-K variants (hidden 3584 = 2 x 7 steps per wave; MLP 18944 zero-padded to 20480)."""
    from oracle import vlm as ovlm
    from vlaser_amd import config as C, synth, ops
    from vlaser_amd.internvl_chat import InternVLChatModel
    cfg = C.truncated(C.vlaser_8b(), 1, 2)
    assert ops.skinny_supported(cfg.llm) and ops.skinny_supported(C.vlaser_2b().llm)
    sd = synth.vlm_state_dict(cfg)
    m = InternVLChatModel(cfg, max_tiles=2, max_seq_len=640)
    m.load_state_dict(sd)
    m.img_context_token_id = cfg.img_context_token_id
    g = torch.Generator().manual_seed(5)
    pv = torch.randn(2, 3, 448, 448, generator=g)
    ids = torch.cat([torch.randint(0, 151643, (20,), generator=g), torch.full((512,), cfg.img_context_token_id),
                     torch.randint(0, 151643, (17,), generator=g)])[None]
    gen, lg = m.generate(pv, ids, max_new_tokens=4, return_logits=True)
    ogen, olg = ovlm.generate(sd, cfg, pv, ids, max_new_tokens=4, eos_token_id=None, return_logits=True)
    parity('8B widths 2 tiles prefill logits vs oracle max|err|/max|ref|', relmax(lg[0, 0], olg[0, 0]), TOL['logits_8b'])
    t2 = olg[0].topk(2, dim=-1).values
    margin = t2[:, 0] - t2[:, 1]
    n_clear = 0
    while n_clear < 4 and margin[n_clear] > 0.08:
        n_clear += 1
    assert gen[0, :n_clear].cpu().tolist() == ogen[0, :n_clear].tolist()
    if n_clear == 4:       # teacher-forced agreement all the way: the decode-step logits match too
        parity('8B widths 2 tiles decode-step-3 logits vs oracle max|err|/max|ref|', relmax(lg[0, 3], olg[0, 3]), TOL['logits_8b'])
    del m
    torch.cuda.empty_cache()


def test_config4_shape_8b_widths_13_tiles_vs_oracle():
    """BASELINE configs[3] at its real SHAPE: Vlaser-8B widths x 13 tiles (12 + thumbnail) -> S = 13*256 + 48 + 32 = 3408 prompt tokens,
    depth-truncated (1 ViT layer, 2 LLM layers): prefill logits + greedy ids vs the fp32 oracle, decode on the chunked-K
    weight-streaming kernels.  (r01 tested the widths at 2 tiles and 13 tiles at 2B widths, never together.)"""
    from oracle import vlm as ovlm
    from vlaser_amd import config as C, synth
    from vlaser_amd.internvl_chat import InternVLChatModel
    cfg = C.truncated(C.vlaser_8b(), 1, 2)
    sd = synth.vlm_state_dict(cfg)
    m = InternVLChatModel(cfg, max_tiles=13, max_seq_len=3456)
    m.load_state_dict(sd)
    assert m.use_skinny
    m.img_context_token_id = cfg.img_context_token_id
    g = torch.Generator().manual_seed(44)
    pv = torch.randn(13, 3, 448, 448, generator=g)
    ids = torch.cat([torch.randint(0, 151643, (41,), generator=g), torch.full((13 * 256,), cfg.img_context_token_id),
                     torch.randint(0, 151643, (39,), generator=g)])[None]
    assert ids.shape[1] == 3408
    gen, lg = m.generate(pv, ids, max_new_tokens=3, return_logits=True)
    ogen, olg = ovlm.generate(sd, cfg, pv, ids, max_new_tokens=3, eos_token_id=None, return_logits=True)
    for t in range(3):
        parity(f'8B widths 13 tiles step-{t} logits vs oracle max|err|/max|ref|', relmax(lg[0, t], olg[0, t]), TOL['logits_8b'])
        t2 = olg[0, t].topk(2).values
        if (t2[0] - t2[1]).item() < 0.08:
            break
        assert gen[0, t].item() == ogen[0, t].item()
    del m
    torch.cuda.empty_cache()


def test_euler_kernel_options_are_bit_identical(golden_model, golden_dir):
    """r03 Euler-phase kernels: 16-row lane-local units ('gu16', 'qkv16') compute the same arithmetic in the same order as the r02 kernels -- the whole
    chunk (10 Euler steps, proprio riding) is bit-identical, eager and graph."""
    from vlaser_amd.pizero import PiZeroInference
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    ids = torch.from_numpy(d['a_input_ids'])
    pv = torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(int(d['a_seed'])))
    pro, noise = torch.from_numpy(d['a_proprio']), torch.from_numpy(d['a_noise'])
    outs = {}
    for opts in ('none', 'gu16', 'qkv16', 'gu16,qkv16'):
        for graph in (False, True):
            m = PiZeroInference(vla, max_batch=1, use_graph=graph, euler_opts=opts); m.load_state_dict(sd)
            for _ in range(3):
                a = m.infer_action(ids, pv, proprios=pro, noise=noise)
            outs[(opts, graph)] = (a.clone(), m.last_velocities().clone())
    ref = outs[('none', False)]
    for k, v in outs.items():
        assert torch.equal(v[0], ref[0]) and torch.equal(v[1], ref[1]), k


@pytest.mark.parametrize('ride', [True, False])
def test_euler_glue1_one_launch_between_layer_passes(golden_model, golden_dir, ride):
    """'glue1' (vlaser_vla_step: tail of Euler step s-1 + action encoder of step s in ONE launch, linear_1 / time embedding folded into linear_2 on
    the host): the chunk and every step's velocity stay within bf16 noise of the 4-launch path (the fold drops one bf16 rounding of linear_1's
    output), eager == graph bit for bit, with the proprio token riding in step 0 and without, and against the reference's own chunk (G7)."""
    from vlaser_amd.pizero import PiZeroInference
    _, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7_vla.npz'))
    ids = torch.from_numpy(d['a_input_ids'])
    pv = torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(int(d['a_seed'])))
    pro, noise = torch.from_numpy(d['a_proprio']), torch.from_numpy(d['a_noise'])
    outs = {}
    for opts in ('qkv16', 'qkv16,glue1'):
        for graph in (False, True):
            m = PiZeroInference(vla, max_batch=1, use_graph=graph, euler_opts=opts, ride_proprio=ride); m.load_state_dict(sd)
            for _ in range(3):
                a = m.infer_action(ids, pv, proprios=pro, noise=noise)
            outs[(opts, graph)] = (a.float().cpu().clone(), m.last_velocities().float().cpu().clone())
    assert torch.equal(outs[('qkv16,glue1', False)][0], outs[('qkv16,glue1', True)][0])
    assert torch.equal(outs[('qkv16,glue1', False)][1], outs[('qkv16,glue1', True)][1])
    a0, v0 = outs[('qkv16', True)]
    a1, v1 = outs[('qkv16,glue1', True)]
    assert (a1 - a0).abs().max().item() <= 4e-3 * max(1.0, a0.abs().max().item()), (a1 - a0).abs().max().item()
    assert (v1 - v0).abs().max().item() <= 2e-2 * max(1.0, v0.abs().max().item()), (v1 - v0).abs().max().item()
    ref = torch.from_numpy(d['a_action']).float() if 'a_action' in d.files else None
    if ref is not None:
        assert (a1.view_as(ref) - ref).abs().max().item() <= (a0.view_as(ref) - ref).abs().max().item() + 4e-3
