"""GPU: general additive attention masks (ABI 8, VL_ATTN_DENSE; PiZero(general_masks=True)) -- VERDICT r05 missing #2.  The reference hands `eager_attention_forward` an arbitrary
[B,1,Sq,Skv] additive mask (joint_model.py:636-656); the default path serves the one pattern its own builder produces (pizero_internvl.py:517-603) through descriptors and refuses
the rest.  Here: the dense-mask variants of the two attention kernels against a torch fp32 reference (holes, left padding, causal text, finite biases), bit-identity with the
descriptor kernels on the builder's own masks, and the whole infer_action against the CPU oracle -- which adds the masks exactly as the reference does -- on masks the default
path refuses."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
FMIN = torch.finfo(torch.float32).min


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from vlaser_amd import ops as o
    return o


def rnd(*s, std=1.0, seed=0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * std).to(BF).cuda()


def _ref(q, k, v, scale, mask):
    """q [B,Hq,Sq,D], k / v [B,Hkv,Skv,D], mask fp32 [B,Sq,Skv] additive -> [B,Sq,Hq*D] (eager attention, fp32)."""
    rep = q.shape[1] // k.shape[1]
    k = k.repeat_interleave(rep, 1); v = v.repeat_interleave(rep, 1)
    s = (q.float() @ k.float().transpose(-1, -2)) * scale + mask[:, None]
    return (s.softmax(-1) @ v.float()).transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)


def _masks(B, Sq, Skv, seed):
    """Additive masks with the things descriptors cannot say: random holes, left padding, a causal band, finite biases; every row keeps at least one key."""
    g = torch.Generator().manual_seed(seed)
    m = torch.zeros(B, Sq, Skv)
    m[torch.rand(B, Sq, Skv, generator=g) < 0.3] = FMIN                        # holes
    m[:, :, :7] = FMIN                                                          # left padding
    i, j = torch.arange(Sq)[:, None], torch.arange(Skv)[None]
    m[0][(j > i + 40).expand(Sq, Skv)] = FMIN                                   # a causal band in batch element 0
    bias = torch.randn(B, Sq, Skv, generator=g) * 2.0                           # finite biases on the visible keys
    m = torch.where(m < -1e30, m, bias)
    m[:, :, 9] = 0.25                                                           # at least one visible key per row
    return m


@pytest.mark.parametrize('B,S', [(1, 384), (2, 384), (1, 200), (2, 77)])
def test_attn_prefill_dense_mask_vs_fp32(ops, B, S):
    from vlaser_amd import _lib as L
    nq, nkv, smax = 12, 2, 448
    q = rnd(B * S, nq * 128, seed=1)
    k = rnd(B, nkv, smax, 128, seed=2); v = rnd(B, nkv, smax, 128, seed=3)
    vt = v.transpose(-1, -2).contiguous()
    sc = 128 ** -0.5
    mask = _masks(B, S, S, seed=S)
    slot = torch.full((B, S + 5, 448), -3.0e38, dtype=torch.float32, device='cuda')       # padded rows, as the model's slot
    slot[:, :S, :S] = mask.cuda()
    out = torch.zeros(B, S, nq * 128, dtype=BF, device='cuda')
    args = (q, k, vt, out, B, S, S, nq, nkv, 128, (S * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax), (S * nq * 128, nq * 128), smax, sc)
    ops.attn_prefill(*args, L.ATTN_DENSE, dense_mask=slot[:, :S])
    ref = _ref(q.view(B, S, nq, 128).permute(0, 2, 1, 3), k[:, :, :S], v[:, :, :S], sc, mask.cuda())
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err
    out2 = torch.zeros_like(out)
    ops.attn_prefill(*args[:3], out2, *args[4:], L.ATTN_DENSE, dense_mask=slot[:, :S])
    assert torch.equal(out, out2)


def test_attn_prefill_dense_equals_prefix_on_the_builders_mask(ops):
    """The reference builder's own pattern as a dense mask == the (valid_len, blk_start) descriptors, bit for bit, on every row somebody reads."""
    from vlaser_amd import _lib as L
    B, nq, nkv, smax, S = 2, 12, 2, 448, 384
    valid = torch.tensor([277, 384], dtype=torch.int32, device='cuda')
    q = rnd(B * S, nq * 128, seed=1)
    k = rnd(B, nkv, smax, 128, seed=2); v = rnd(B, nkv, smax, 128, seed=3)
    vt = v.transpose(-1, -2).contiguous()
    sc = 128 ** -0.5
    strides = ((S * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax), (S * nq * 128, nq * 128))
    a = torch.zeros(B, S, nq * 128, dtype=BF, device='cuda'); b = torch.zeros_like(a)
    ops.attn_prefill(q, k, vt, a, B, S, S, nq, nkv, 128, *strides, smax, sc, L.ATTN_PREFIX, valid_len=valid, blk_start=384)
    slot = torch.full((B, S, 448), FMIN, dtype=torch.float32, device='cuda')
    for i in range(B):
        slot[i, :, :int(valid[i])] = 0.0
    ops.attn_prefill(q, k, vt, b, B, S, S, nq, nkv, 128, *strides, smax, sc, L.ATTN_DENSE, dense_mask=slot)
    for i in range(B):
        n = int(valid[i])
        assert torch.equal(a[i, :n], b[i, :n])


@pytest.mark.parametrize('nq_tok,B,kv_len', [(4, 1, 389), (1, 1, 385), (4, 2, 389), (4, 4, 389), (1, 3, 200)])
def test_chain_attn_dense_mask_vs_fp32(ops, nq_tok, B, kv_len):
    """chain_attn (DENSE) + chain_oproj: sum of the split-K slabs == o_proj(eager attention under the additive mask)."""
    from vlaser_amd import _lib as L
    nq, nkv, smax, H = 12, 2, 448, 768
    G, M, ks_o = nq // nkv, B * nq_tok, 3
    nsp = ops.chain_attn_splits(kv_len)
    q = rnd(M, nq * 128, seed=1)
    k = rnd(B, nkv, smax, 128, seed=2); v = rnd(B, nkv, smax, 128, seed=3)
    vt = v.transpose(-1, -2).contiguous()
    sc = 128 ** -0.5
    wo = rnd(H, nq * 128, std=0.03, seed=9)
    wp = ops.pack_skinny(wo, ks_o, 1)
    mask = _masks(B, nq_tok, kv_len, seed=kv_len + nq_tok)
    slot = torch.full((B, nq_tok + 2, 448), -3.0e38, dtype=torch.float32, device='cuda')
    slot[:, 1:1 + nq_tok, :kv_len] = mask.cuda()
    strides = ((nq_tok * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax))
    cp = ops.chain_attn_buffers(B, nkv, 'cuda')
    a = ops.attn_skinny_args(q, k, vt, (cp[0], cp[0], cp[1]), B, nq_tok, kv_len, nq, nkv, 128, *strides, smax, sc, L.ATTN_DENSE, nsp, dense_mask=slot[:, 1:1 + nq_tok])
    outs = []
    for rep in range(2):
        out = torch.full((ks_o, M, H), 5.0, dtype=torch.float32, device='cuda')
        ops.launch_chain_attn(a)
        o_args, _ = ops.skinny_args(None, wp, M, out_f32=out, attn_m=cp[0], attn_o=cp[1], attn_splits=nsp, attn_group=G, attn_nq=nq_tok)
        ops.launch_chain_oproj(o_args)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    att = _ref(q.view(B, nq_tok, nq, 128).permute(0, 2, 1, 3), k[:, :, :kv_len], v[:, :, :kv_len], sc, mask.cuda()).reshape(M, nq * 128)
    ref = att.to(BF).float() @ wo.float().t()
    err = (outs[0].sum(0) - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err


def _obs(cfg, seed, left_pad=0, B=1):
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((B, 384), cfg.pad_token_id)
    o = left_pad
    ids[:, o:o + 10] = torch.randint(0, 151643, (B, 10), generator=g)
    ids[:, o + 10:o + 266] = cfg.img_context_token_id
    ids[:, o + 266:o + 277] = torch.randint(0, 151643, (B, 11), generator=g)
    return ids, torch.randn(B, 3, 448, 448, generator=g), torch.rand(B, 1, 7, generator=g) * 2 - 1, torch.randn(B, 4, 7, generator=g)


def _general_mask(am, vla, kind, seed):
    """A full [B,1,L,L] additive mask from a 0/1 attention_mask with ARBITRARY positions of the ones (the reference builder only counts them), plus extras."""
    B, T = am.shape
    na = vla.num_action_tokens
    Lt = T + 1 + na
    m = torch.full((B, Lt, Lt), FMIN)
    for b in range(B):
        vis = am[b].bool()
        m[b, :T, :T][vis[:, None] & vis[None, :]] = 0.0                     # image / text tokens see each other (wherever they sit)
        m[b, T:, :T][:, vis] = 0.0                                           # proprio / action rows see them
    m[:, T, T] = 0.0
    m[:, T + 1:, T:] = 0.0
    g = torch.Generator().manual_seed(seed)
    if kind == 'causal_text':                                                # the text after the image tokens attends causally
        for b in range(B):
            idx = am[b].nonzero().flatten()[-11:]
            for a_, i in enumerate(idx):
                m[b, i, idx[a_ + 1:]] = FMIN
    if kind == 'bias':                                                       # finite biases on visible entries (ALiBi-like), a few action -> prefix holes
        bias = torch.randn(B, Lt, Lt, generator=g)
        m = torch.where(m < -1e30, m, bias)
        m[:, T + 2, 20:60] = FMIN
    return m[:, None]


@pytest.mark.parametrize('kind,left_pad,B', [('plain', 0, 1), ('plain', 50, 1), ('causal_text', 23, 1), ('bias', 0, 1), ('plain', 107, 2)])
def test_infer_action_general_masks_vs_oracle(golden_model, kind, left_pad, B):
    """Whole infer_action under masks the default path refuses (left padding: the valid tokens are not a prefix; causal text; finite biases) against the CPU oracle, whose
    attention adds the dense masks exactly as the reference's eager attention does; the position ids travel with the tokens, as in the reference call."""
    from oracle import vla as ovla
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    ids, pv, pro, noise = _obs(cfg, 40 + left_pad, left_pad, B)
    am = (ids != cfg.pad_token_id).long()
    full = _general_mask(am, vla, kind, seed=left_pad)
    m1, m2 = ovla.split_full_mask_into_submasks(full, vla)
    _, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
    ref = ovla.infer_action(sd, vla, ids, pv, m1, m2, vp, pp, ap, pro, noise)
    m = PiZeroInference(vla, max_batch=B, general_masks=True)
    m.load_state_dict(sd)
    act = m.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)
    err = (act.cpu() - ref).abs().max().item()
    print(f'[parity] infer_action general masks {kind} left_pad={left_pad} B={B}: max|err| vs oracle = {err:.3e} (bound 1e-2)')
    assert err < 1e-2
    again = m.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)
    assert torch.equal(act, again)
    m.check_errors()
    if left_pad or kind != 'plain':
        # the default path refuses the same call (NaN + ValueError), pointing at the switch
        d = PiZeroInference(vla, max_batch=B)
        d.load_state_dict(sd)
        bad = d.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)
        with pytest.raises(ValueError, match='general_masks=True'):
            bad.cpu()


def test_infer_action_general_masks_equal_default_on_the_builders_masks(golden_model):
    """On the reference builder's own masks the general path returns the default path's chunk (same kernels' arithmetic, the proprio row in its own pass)."""
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    ids, pv, pro, noise = _obs(cfg, 77)
    g = PiZeroInference(vla, max_batch=1, general_masks=True); g.load_state_dict(sd)
    d = PiZeroInference(vla, max_batch=1, ride_proprio=False); d.load_state_dict(sd)
    mask, vp, pp, ap = g.build_causal_mask_and_position_ids((ids != cfg.pad_token_id).long(), torch.float32)
    m1, m2 = g.split_full_mask_into_submasks(mask)
    a = g.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)
    b = d.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)
    assert (a - b).abs().max().item() < 2e-3
    for dt in (torch.bfloat16, torch.float16):                               # the reference builds the mask in the model dtype
        a2 = g.infer_action(ids, pv, m1.to(dt), m2.to(dt), vp, pp, ap, pro, noise=noise)
        assert torch.equal(a, a2)
    with pytest.raises(ValueError, match='general_masks=True'):
        g.infer_action(ids, pv, proprios=pro, noise=noise)
    # the one pattern still refused: an image / text row that sees the proprio key
    bad1 = m1.clone(); bad1[0, 0, 3, 384] = 0.0
    out = g.infer_action(ids, pv, bad1, m2, vp, pp, ap, pro, noise=noise)
    with pytest.raises(ValueError, match='sees the proprio key'):
        out.cpu()


def test_infer_action_general_masks_vs_reference_golden(golden_model, golden_dir):
    """G7d: the REFERENCE's own chunks under a left-padded prompt, causal text and finite biases + a hole in an action row (tools/gen_golden.py g7d_general_masks)."""
    import os
    import numpy as np
    from test_oracle_golden import g7d_case
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    d = np.load(os.path.join(golden_dir, 'g7d_general_masks.npz'))
    m = PiZeroInference(vla, max_batch=1, general_masks=True)
    m.load_state_dict(sd)
    for case in ('a', 'b', 'c'):
        ids, pv, m1, m2, vp, pp, ap, pro, noise = g7d_case(d, case, vla)
        act = m.infer_action(ids, pv, m1, m2, vp, pp, ap, pro, noise=noise)
        err = (act.cpu() - torch.from_numpy(d[f'{case}_action'])).abs().max().item()
        print(f'[parity] infer_action general masks, golden G7d case {case}: max|err| vs the reference = {err:.3e} (bound 1e-2)')
        assert err < 1e-2
    m.check_errors()
