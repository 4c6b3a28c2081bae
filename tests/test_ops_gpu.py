"""Op-level parity of every C-ABI kernel against a plain PyTorch fp32 reference of the same op (GPU box only)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from vlaser_amd import ops as o
    return o


def rnd(*shape, std=1.0, seed=0, dtype=BF):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * std).to(dtype).cuda()


def close(got, ref, rtol=1.6e-2, atol=None, name=''):
    got, ref = got.float(), ref.float()
    if atol is None:
        atol = 1e-2 * ref.abs().max().item() + 1e-6
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert torch.isfinite(got).all(), f'{name}: non-finite output'
    bad = (err > tol)
    assert not bad.any(), f'{name}: {int(bad.sum())}/{bad.numel()} mismatches, max err {err.max().item():.4g} (ref max {ref.abs().max().item():.4g})'


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (1025, 1024, 1024), (385, 1536, 1536), (77, 4096, 1024), (300, 1536, 8960)])
def test_gemm_plain_bias_gelu(ops, M, N, K):
    from vlaser_amd import _lib as L
    x, w, b = rnd(M, K), rnd(N, K, std=0.05), rnd(N, std=0.5)
    ref = x.float() @ w.float().t()
    close(ops.linear(x, w), ref, name='none')
    close(ops.linear(x, w, b), ref + b.float(), name='bias')
    close(ops.linear(x, w, b, epi=L.EPI_BIAS_GELU), F.gelu(ref + b.float()), name='gelu')
    if N % 4 == 0:      # aux_out (ABI 4): the same launch also keeps the rounded pre-activation == what EPI_BIAS writes, and leaves `out` unchanged
        z, g2 = torch.zeros(M, N, dtype=BF, device='cuda'), torch.zeros(M, N, dtype=BF, device='cuda')
        ops.gemm(L.EPI_BIAS_GELU, x, w, out=g2, bias=b, aux_out=z, ld_aux=N)
        assert torch.equal(z, ops.linear(x, w, b)) and torch.equal(g2, ops.linear(x, w, b, epi=L.EPI_BIAS_GELU))
    res, ls = rnd(M, N, seed=3), rnd(N, std=0.1, seed=4)
    close(ops.linear(x, w, b, epi=L.EPI_BIAS_LS_RES, res=res, ls=ls), res.float() + ls.float() * (ref + b.float()), name='ls_res')
    close(ops.linear(x, w, epi=L.EPI_RES, res=res), res.float() + ref, name='res')
    out = res.clone()       # in-place residual
    ops.linear(x, w, epi=L.EPI_RES, res=out, out=out)
    close(out, res.float() + ref, name='res_inplace')


def test_gemm_f32_edge_n(ops):
    from vlaser_amd import _lib as L
    M, N, K = 70, 1002, 1536        # N not a multiple of 4/16/128 (vocab-like edge)
    x, w = rnd(M, K), rnd(N, K, std=0.05)
    out = ops.linear(x, w, epi=L.EPI_F32)
    close(out, x.float() @ w.float().t(), rtol=2e-3, atol=2e-3, name='f32')
    outb = ops.linear(x, w)
    close(outb, x.float() @ w.float().t(), name='bf16 edge')


def test_gemm_transpose_detect(ops):
    # A = I-like check with asymmetric W (guide: symmetric inputs hide a swapped C layout)
    M = N = K = 128
    x = torch.eye(M, dtype=BF).cuda()
    w = (torch.arange(N * K).reshape(N, K) % 251).to(BF).cuda()
    out = ops.linear(x, w)
    assert torch.equal(out.float(), w.float().t())


def test_gemm_swiglu(ops):
    from vlaser_amd import _lib as L
    M, I, K = 200, 8960, 1536
    x, g, u = rnd(M, K), rnd(I, K, std=0.03, seed=1), rnd(I, K, std=0.03, seed=2)
    packed = ops.pack_gate_up(g, u)
    out = ops.linear(x, packed, epi=L.EPI_SWIGLU)
    gr = (x.float() @ g.float().t()).to(BF).float()
    ur = (x.float() @ u.float().t()).to(BF).float()
    close(out, F.silu(gr).to(BF).float() * ur, name='swiglu')


def _rope_ref(x, pos, theta=1e6):
    # x [M, heads, 128] fp32
    inv = 1.0 / (theta ** (torch.arange(0, 128, 2).float() / 128))
    f = pos.float().cpu()[:, None] * inv[None]
    cos, sin = torch.cat([f, f], -1).cos()[:, None].to(x.device), torch.cat([f, f], -1).sin()[:, None].to(x.device)
    rot = torch.cat([-x[..., 64:], x[..., :64]], -1)
    return x * cos + rot * sin


def test_gemm_qkv_rope(ops):
    from vlaser_amd import _lib as L
    B, S, H, nq, nkv, smax = 2, 100, 1536, 12, 2, 192
    M = B * S
    x = rnd(M, H)
    qw, kw, vw = rnd(nq * 128, H, std=0.03, seed=1), rnd(nkv * 128, H, std=0.03, seed=2), rnd(nkv * 128, H, std=0.03, seed=3)
    qb, kb, vb = rnd(nq * 128, std=0.3, seed=4), rnd(nkv * 128, std=0.3, seed=5), rnd(nkv * 128, std=0.3, seed=6)
    W, Bv = ops.pack_qkv(qw, kw, vw, qb, kb, vb)
    cos, sin = ops.rope_table(512)
    pos = (torch.arange(S).repeat(B) + 3).int().cuda()
    q_out = torch.zeros(M, nq * 128, dtype=BF, device='cuda')
    kc = torch.zeros(B, nkv, smax, 128, dtype=BF, device='cuda')
    vtc = torch.zeros(B, nkv, 128, smax, dtype=BF, device='cuda')
    ops.gemm(L.EPI_QKV_ROPE, x, W, bias=Bv, q_out=q_out, k_cache=kc, vt_cache=vtc, rope_cos=cos, rope_sin=sin, pos_ids=pos,
             n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=S, slot_base=5)
    xf = x.float()
    q = (xf @ qw.float().t() + qb.float()).to(BF).float().view(M, nq, 128)
    k = (xf @ kw.float().t() + kb.float()).to(BF).float().view(M, nkv, 128)
    v = (xf @ vw.float().t() + vb.float()).to(BF).float().view(M, nkv, 128)
    close(q_out.view(M, nq, 128), _rope_ref(q, pos.cpu().cuda()), name='q')
    kr = _rope_ref(k, pos).view(B, S, nkv, 128).permute(0, 2, 1, 3)
    close(kc[:, :, 5:5 + S], kr, name='k cache')
    close(vtc[:, :, :, 5:5 + S], v.view(B, S, nkv, 128).permute(0, 2, 3, 1), name='vT cache')
    assert kc[:, :, :5].abs().max() == 0 and kc[:, :, 5 + S:].abs().max() == 0
    assert vtc[:, :, :, :5].abs().max() == 0 and vtc[:, :, :, 5 + S:].abs().max() == 0


def test_gemm_vit_qkv(ops):
    from vlaser_amd import _lib as L
    T, S, C, Hn, Sp = 2, 130, 1024, 16, 192
    x, w, b = rnd(T * S, C), rnd(3 * C, C, std=0.03), rnd(3 * C, std=0.3)
    q = torch.zeros(T, Hn, Sp, 64, dtype=BF, device='cuda'); k = torch.zeros_like(q)
    vt = torch.zeros(T, Hn, 64, Sp, dtype=BF, device='cuda')
    ops.gemm(L.EPI_VIT_QKV, x, w, bias=b, vq=q, vk=k, vvt=vt, vit_heads=Hn, vit_seq=S, vit_seq_pad=Sp, q_scale=0.125)
    ref = (x.float() @ w.float().t() + b.float()).to(BF).float().view(T, S, 3, Hn, 64).permute(2, 0, 3, 1, 4)
    close(q[:, :, :S], ref[0] * 0.125, name='q'); close(k[:, :, :S], ref[1], name='k')
    close(vt[:, :, :, :S], ref[2].transpose(-1, -2), name='vT')


def _attn_ref(q, k, v, scale, vis):
    # q [B,Hq,Sq,D], k/v [B,Hkv,Skv,D], vis bool [B,Sq,Skv]
    rep = q.shape[1] // k.shape[1]
    k = k.repeat_interleave(rep, 1); v = v.repeat_interleave(rep, 1)
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    s = s.masked_fill(~vis[:, None], float('-inf'))
    return (s.softmax(-1) @ v.float()).transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)


@pytest.mark.parametrize('T,S', [(2, 1025), (1, 1025), (1, 1024), (1, 1040), (1, 1056), (1, 1153), (1, 449), (1, 200), (13, 1025)])
def test_attn_vit_full(ops, T, S):
    """InternViT attention (FULL mode, head_dim 64).  S = 1025 = 8 key tiles of 128 + ONE key and 16 workgroups of 64 query rows + ONE row: since r04 the odd key
    is a 32-key chunk straight from global memory and the odd row's workgroup splits the keys between its waves (csrc/attn.hip, `TAIL`); 1040 / 1056 / 1153 /
    200 put 16 / 32 / 1 / 8 keys and rows behind the last full tile (with and without the tail paths), 13 tiles at once, the lse output on the tail row.
    One tile (T = 1, <= 512 workgroups): the tail rows are HOSTED by the last full workgroup of their head (a second softmax state per wave on every fourth key
    tile, `HOST`): 1025 / 1040 / 1153 with the tail-key chunk, 449 with a masked last tile instead."""
    from vlaser_amd import _lib as L
    Hn, Sp = 16, (S + 63) // 64 * 64
    q = rnd(T, Hn, Sp, 64, seed=1); k = rnd(T, Hn, Sp, 64, seed=2); v = rnd(T, Hn, Sp, 64, seed=3)
    k[:, :, S - 1] *= 3.0                                   # the last key matters: a dropped tail would show
    v[:, :, S:] = 0
    vt = v.transpose(-1, -2).contiguous()
    out = torch.zeros(T, S, Hn * 64, dtype=BF, device='cuda')
    lse = torch.zeros(T, Hn, S, dtype=torch.float32, device='cuda')
    ops.attn_prefill(q, k, vt, out, T, S, S, Hn, Hn, 64, (Hn * Sp * 64, Sp * 64, 64), (Hn * Sp * 64, Sp * 64),
                     (Hn * 64 * Sp, 64 * Sp), (S * Hn * 64, Hn * 64), Sp, 0.125, L.ATTN_FULL, lse_out=lse)
    vis = torch.ones(T, S, S, dtype=torch.bool, device='cuda')
    close(out, _attn_ref(q[:, :, :S], k[:, :, :S], v[:, :, :S], 0.125, vis), name='vit attn')
    sref = (q[:, :, :S].float() @ k[:, :, :S].float().transpose(-1, -2)) * 0.125
    close(lse, torch.logsumexp(sref, -1) * 1.4426950408889634, rtol=2e-3, atol=2e-3, name='base-2 log-sum-exp')


@pytest.mark.parametrize('S,off', [(336, 0), (70, 0), (130, 40)])
def test_attn_causal_gqa(ops, S, off):
    from vlaser_amd import _lib as L
    B, nq, nkv, smax = 2, 12, 2, 448
    kv_len = S + off
    q = rnd(B * S, nq * 128, seed=1)
    k = rnd(B, nkv, smax, 128, seed=2); v = rnd(B, nkv, smax, 128, seed=3)
    vt = v.transpose(-1, -2).contiguous()
    out = torch.zeros(B, S, nq * 128, dtype=BF, device='cuda')
    sc = 128 ** -0.5
    ops.attn_prefill(q, k, vt, out, B, S, kv_len, nq, nkv, 128, (S * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128),
                     (nkv * 128 * smax, 128 * smax), (S * nq * 128, nq * 128), smax, sc, L.ATTN_CAUSAL, causal_off=off)
    i = torch.arange(S, device='cuda')[:, None]; j = torch.arange(kv_len, device='cuda')[None]
    vis = (j <= i + off)[None].expand(B, -1, -1)
    qq = q.view(B, S, nq, 128).permute(0, 2, 1, 3)
    close(out, _attn_ref(qq, k[:, :, :kv_len], v[:, :, :kv_len], sc, vis), name='causal')


@pytest.mark.parametrize('S,causal,kv_valid', [(336, True, None), (70, True, None), (389, True, None), (200, False, 137), (64, False, 64)])
def test_attn_bwd_fused_vs_autograd(ops, S, causal, kv_valid):
    """csrc/attn_bwd.hip (forward keeps the base-2 log-sum-exp; dQ and per-Q-head dK / dV without score matrices) against fp32 autograd of
    the same masked softmax attention on the same bf16 inputs.  Tolerance: bf16 outputs of sums over <= 389 terms -> 2e-2 of the tensor's max."""
    from vlaser_amd import _lib as L
    nq, nkv, hd, smax = 12, 2, 128, 448
    G = nq // nkv
    kvv = S if kv_valid is None else kv_valid
    q = rnd(S, nq * hd, seed=1)
    k = rnd(1, nkv, smax, hd, seed=2); v = rnd(1, nkv, smax, hd, seed=3)
    d_o = rnd(S, nq * hd, seed=4)
    vt = v.transpose(-1, -2).contiguous()
    out = torch.zeros(S, nq * hd, dtype=BF, device='cuda')
    lse = torch.zeros(nq * S, dtype=torch.float32, device='cuda')
    sc = hd ** -0.5
    if causal:
        ops.attn_prefill(q, k, vt, out, 1, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * smax * hd, smax * hd), (nkv * hd * smax, hd * smax),
                         (S * nq * hd, nq * hd), smax, sc, L.ATTN_CAUSAL, lse_out=lse)
    else:
        vl = torch.tensor([kvv], dtype=torch.int32, device='cuda')
        ops.attn_prefill(q, k, vt, out, 1, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * smax * hd, smax * hd), (nkv * hd * smax, hd * smax),
                         (S * nq * hd, nq * hd), smax, sc, L.ATTN_PREFIX, valid_len=vl, blk_start=smax, lse_out=lse)
    i = torch.arange(S, device='cuda')[:, None]; j = torch.arange(S, device='cuda')[None]
    vis = ((j <= i) if causal else (j < kvv).expand(S, S))
    with torch.enable_grad():
        qf = q.float().view(S, nq, hd).permute(1, 0, 2).clone().requires_grad_(True)
        kf = k[0, :, :S].float().clone().requires_grad_(True)
        vf = v[0, :, :S].float().clone().requires_grad_(True)
        s_ = (qf @ kf.repeat_interleave(G, 0).transpose(-1, -2)) * sc
        s_ = s_.masked_fill(~vis[None], float('-inf'))
        o_ref = (s_.softmax(-1) @ vf.repeat_interleave(G, 0)).permute(1, 0, 2).reshape(S, nq * hd)
        lse_ref = torch.logsumexp(s_, -1) * 1.4426950408889634
        o_ref.backward(d_o.float())
    close(out, o_ref.detach(), name='forward out')
    assert (lse.view(nq, S) - lse_ref.detach()).abs().max().item() < 2e-2
    dq = torch.full((S, nq * hd), 7.0, dtype=BF, device='cuda'); dk = torch.full_like(dq, 7.0); dv = torch.full_like(dq, 7.0)
    delta = torch.zeros(nq * S, dtype=torch.float32, device='cuda')
    ops.attn_bwd(q, k, vt, out, d_o, lse, delta, dq, dk, dv, S, nq, nkv, smax, sc, causal=causal, kv_valid=kvv)
    torch.cuda.synchronize()
    d_ref = (d_o.float() * out.float()).view(S, nq, hd).sum(-1).t()
    assert (delta.view(nq, S) - d_ref).abs().max().item() < 1e-3 * max(1.0, d_ref.abs().max().item())
    close(dq, qf.grad.permute(1, 0, 2).reshape(S, nq * hd), name='dq')
    dk_sum = dk.float().view(S, nkv, G, hd).sum(2).permute(1, 0, 2)
    dv_sum = dv.float().view(S, nkv, G, hd).sum(2).permute(1, 0, 2)
    close(dk_sum, kf.grad, name='dk')
    close(dv_sum, vf.grad, name='dv')
    # run-to-run bit-identical (no atomics)
    dq2 = torch.zeros_like(dq); dk2 = torch.zeros_like(dq); dv2 = torch.zeros_like(dq)
    ops.attn_bwd(q, k, vt, out, d_o, lse, delta, dq2, dk2, dv2, S, nq, nkv, smax, sc, causal=causal, kv_valid=kvv)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


def test_attn_prefix_block(ops):
    from vlaser_amd import _lib as L
    B, nq, nkv, smax, S = 2, 12, 2, 448, 385
    valid = torch.tensor([277, 384], dtype=torch.int32, device='cuda')
    q = rnd(B * S, nq * 128, seed=1)
    k = rnd(B, nkv, smax, 128, seed=2); v = rnd(B, nkv, smax, 128, seed=3)
    vt = v.transpose(-1, -2).contiguous()
    out = torch.zeros(B, S, nq * 128, dtype=BF, device='cuda')
    sc = 128 ** -0.5
    ops.attn_prefill(q, k, vt, out, B, S, S, nq, nkv, 128, (S * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128),
                     (nkv * 128 * smax, 128 * smax), (S * nq * 128, nq * 128), smax, sc, L.ATTN_PREFIX, valid_len=valid, blk_start=384)
    j = torch.arange(S, device='cuda')[None, None]; i = torch.arange(S, device='cuda')[None, :, None]
    vis = (j < valid[:, None, None]) | ((i >= 384) & (j >= 384))
    qq = q.view(B, S, nq, 128).permute(0, 2, 1, 3)
    ref = _attn_ref(qq, k[:, :, :S], v[:, :, :S], sc, vis)
    for b in range(B):       # rows beyond the valid prefix are "don't care" (reference: uniform softmax over masked row)
        n = int(valid[b])
        close(out[b, :n], ref[b, :n], name=f'prefix rows b{b}')
        close(out[b, 384:], ref[b, 384:], name=f'proprio row b{b}')


@pytest.mark.parametrize('nq_tok,kv_len,mode,nsp', [(4, 389, 'prefix', 4), (1, 385, 'prefix', 4), (1, 337, 'full', 3), (3, 1500, 'full', 8),
                                                    (5, 40, 'full', 1), (2, 389, 'prefix', 7)])
def test_attn_skinny_partials_and_merge(ops, nq_tok, kv_len, mode, nsp):
    """Split partials of the GQA-aware skinny attention + the merge fused into the o_proj GEMV prologue."""
    from vlaser_amd import _lib as L
    B, nq, nkv, smax = 2, 12, 2, 1536
    G = nq // nkv
    q = rnd(B * nq_tok, nq * 128, seed=1)
    k = rnd(B, nkv, smax, 128, seed=2); v = rnd(B, nkv, smax, 128, seed=3)
    vt = v.transpose(-1, -2).contiguous()
    sc = 128 ** -0.5
    valid = torch.tensor([277, 300], dtype=torch.int32, device='cuda')
    kw = dict(valid_len=valid, blk_start=384) if mode == 'prefix' else {}
    parts = ops.attn_partial_buffers(B, nkv, 'cuda', max_splits=nsp)
    ops.attn_skinny(q, k, vt, parts, B, nq_tok, kv_len, nq, nkv, 128, (nq_tok * nq * 128, 128, nq * 128),
                    (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax), smax, sc,
                    L.ATTN_PREFIX if mode == 'prefix' else L.ATTN_FULL, nsp, **kw)
    j = torch.arange(kv_len, device='cuda')[None, None]
    if mode == 'prefix':
        vis = ((j < valid[:, None, None]) | (j >= 384)).expand(B, nq_tok, kv_len)
    else:
        vis = torch.ones(B, nq_tok, kv_len, dtype=torch.bool, device='cuda')
    qq = q.view(B, nq_tok, nq, 128).permute(0, 2, 1, 3)
    ref = _attn_ref(qq, k[:, :, :kv_len], v[:, :, :kv_len], sc, vis)          # [B, tok, nq*128]
    # reference merge of the partials (base-2 running max), rows r = hg*nq_tok + tok
    pm, pl, po = parts
    M = pm.max(dim=2, keepdim=True).values
    f = torch.exp2(pm - M)
    o = (po * f[..., None]).sum(2) / (pl * f).sum(2)[..., None]                # [B, nkv, 32, 128]
    o = o[:, :, :G * nq_tok].view(B, nkv, G, nq_tok, 128).permute(0, 3, 1, 2, 4).reshape(B, nq_tok, nq * 128)
    close(o, ref, name='partials merged')
    # merge fused into the o_proj skinny GEMV
    Mrows, H = B * nq_tok, 768
    wo = rnd(H, nq * 128, std=0.03, seed=9)
    ks = 3
    part = torch.zeros(ks, Mrows, H, dtype=torch.float32, device='cuda')
    ops.skinny(L.PRO_ATTN, L.SK_PARTIAL, None, ops.pack_skinny(wo, ks), Mrows, out_f32=part, attn_m=pm, attn_l=pl, attn_o=po, attn_splits=nsp,
               attn_group=G, attn_nq=nq_tok)
    close(part.sum(0), ref.reshape(Mrows, -1).to(BF).float() @ wo.float().t(), rtol=2e-2, name='o_proj over merged attention')


@pytest.mark.parametrize('nq_tok,B,kv_len,first,valid,H,blk', [
    (4, 1, 389, 0, [277], 768, 384), (5, 1, 389, 385, [277], 768, 384), (4, 2, 389, 0, [277, 31], 768, 384), (4, 1, 389, 0, [384], 768, 384),
    (4, 1, 389, 0, [1], 768, 384), (1, 1, 389, 0, [200], 1536, 384), (4, 2, 389, 0, [384, 100], 768, 384),
    # the greedy decode step's form (one token per sequence, key schedule sized for kvmax, the visible count on the device, empty trailing block): 10 / 9 / 16 / 8 / 1 splits
    (1, 1, 640, 0, [571], 1536, 640), (1, 8, 640, 0, [561, 562, 600, 640, 1, 33, 577, 592], 1536, 640), (1, 1, 576, 0, [570], 1536, 576), (1, 2, 1024, 0, [1000, 3], 1536, 1024),
    (1, 1, 512, 0, [512], 1536, 512), (1, 1, 64, 0, [40], 1536, 64), (4, 1, 600, 0, [500], 768, 596), (1, 16, 640, 0, [561 + 5 * i for i in range(16)], 1536, 640), (4, 4, 389, 0, [277, 31, 384, 100], 768, 384)])
def test_chain_attn_oproj_vs_fp32(ops, nq_tok, B, kv_len, first, valid, H, blk):
    """r05: one wave per (kv head, key split) attention leaving (m, l) + normalised bf16 rows, merged by the o_proj launch's prologue: the sum of the split-K slabs ==
    o_proj(attention) of an fp32 reference under the VLA block mask (valid prefix + trailing block, the riding proprio row's own key limit, batches with ragged
    prefixes, a prefix of one key), and the pair agrees with vlaser_attn_skinny + vlaser_skinny(ATTN, PARTIAL) to bf16 noise; deterministic."""
    from vlaser_amd import _lib as L
    nq, nkv, smax = 12, 2, max(448, (kv_len + 63) // 64 * 64)
    G, M = nq // nkv, B * nq_tok
    ks_o = 3 if H == 768 else 2
    nsp = ops.chain_attn_splits(kv_len)
    assert nsp == ((kv_len + 31) // 32 + 1) // 2 <= 16
    assert ops.chain_oproj_supported(M, H, nq * 128, ks_o, nsp, G)
    q = rnd(M, nq * 128, seed=1)
    k = rnd(B, nkv, smax, 128, seed=2); v = rnd(B, nkv, smax, 128, seed=3)
    k[:, :, 5] *= 4.0
    vt = v.transpose(-1, -2).contiguous()
    sc = 128 ** -0.5
    vl = torch.tensor(valid, dtype=torch.int32, device='cuda')
    wo = rnd(H, nq * 128, std=0.03, seed=9)
    wp = ops.pack_skinny(wo, ks_o, 1)
    strides = ((nq_tok * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax))
    cp = ops.chain_attn_buffers(B, nkv, 'cuda')
    a = ops.attn_skinny_args(q, k, vt, (cp[0], cp[0], cp[1]), B, nq_tok, kv_len, nq, nkv, 128, *strides, smax, sc, L.ATTN_PREFIX, nsp, valid_len=vl, blk_start=blk,
                             first_tok_kv_len=first)
    outs = []
    for rep in range(2):
        out = torch.full((ks_o, M, H), 5.0, dtype=torch.float32, device='cuda')
        ops.launch_chain_attn(a)
        o_args, _ = ops.skinny_args(None, wp, M, out_f32=out, attn_m=cp[0], attn_o=cp[1], attn_splits=nsp, attn_group=G, attn_nq=nq_tok)
        ops.launch_chain_oproj(o_args)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    j = torch.arange(kv_len, device='cuda')[None, None]
    vis = ((j < vl.view(B, 1, 1)) | (j >= blk)).expand(B, nq_tok, kv_len).clone()
    if first:
        vis[:, 0] = (j[0] < vl.view(B, 1)) | ((j[0] >= blk) & (j[0] < first))
    qq = q.view(B, nq_tok, nq, 128).permute(0, 2, 1, 3)
    att = _attn_ref(qq, k[:, :, :kv_len], v[:, :, :kv_len], sc, vis).reshape(M, nq * 128)
    ref = att.to(BF).float() @ wo.float().t()
    close(outs[0].sum(0), ref, rtol=2e-2, name='sum of the split-K slabs')
    # the r01-r04 pair on the same inputs
    parts = ops.attn_partial_buffers(B, nkv, 'cuda')
    n_old = ops.attn_splits(kv_len)
    ops.attn_skinny(q, k, vt, parts, B, nq_tok, kv_len, nq, nkv, 128, *strides, smax, sc, L.ATTN_PREFIX, n_old, valid_len=vl, blk_start=blk, first_tok_kv_len=first)
    old = torch.zeros(ks_o, M, H, dtype=torch.float32, device='cuda')
    ops.skinny(L.PRO_ATTN, L.SK_PARTIAL, None, wp, M, out_f32=old, attn_m=parts[0], attn_l=parts[1], attn_o=parts[2], attn_splits=n_old, attn_group=G, attn_nq=nq_tok)
    assert (outs[0].sum(0) - old.sum(0)).abs().max().item() <= 2e-2 * max(1.0, old.sum(0).abs().max().item())


def _rms_ref(h, w, eps=1e-6):
    hf = h.float()
    return (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps)).to(BF).float() * w.float()


@pytest.mark.parametrize('M,N,K,ks', [(4, 768, 1536, 6), (4, 768, 8960, 7), (1, 1536, 8960, 5), (16, 1536, 1536, 1), (5, 3584, 3584, 2), (2, 1536, 18944, 37)])
def test_skinny_partial(ops, M, N, K, ks):
    from vlaser_amd import _lib as L
    x, w = rnd(M, K), rnd(N, K, std=0.03)
    part = torch.zeros(ks, M, N, dtype=torch.float32, device='cuda')
    ops.skinny(L.PRO_PLAIN, L.SK_PARTIAL, x, ops.pack_skinny(w, ks), M, out_f32=part)
    close(part.sum(0), x.float() @ w.float().t(), rtol=2e-3, atol=2e-3, name='partial sum')
    # each slab is the partial over its own K slice
    kb = K // ks
    close(part[ks - 1], x[:, -kb:].float() @ w[:, -kb:].float().t(), rtol=2e-3, atol=2e-3, name='last slab')


@pytest.mark.parametrize('M,N,K,ks', [(4, 768, 8960, 5), (2, 1536, 8960, 7), (5, 784, 8960, 5)])
def test_skinny_partial_16_row_units(ops, M, N, K, ks):
    """tiles_per_unit = 1: 16-row units (twice the workgroups for the narrow down_proj), incl. an N that is no multiple of 32."""
    from vlaser_amd import _lib as L
    x, w = rnd(M, K), rnd(N, K, std=0.03)
    part = torch.full((ks, M, N), 7.0, dtype=torch.float32, device='cuda')
    pw = ops.pack_skinny(w, ks, 1)
    assert pw.tpu == 1 and pw.N == (N + 15) // 16 * 16
    ops.skinny(L.PRO_PLAIN, L.SK_PARTIAL, x, pw, M, out_f32=part)
    close(part.sum(0), x.float() @ w.float().t(), rtol=2e-3, atol=2e-3, name='partial sum (16-row units)')


def test_skinny_bias_silu_f32(ops):
    from vlaser_amd import _lib as L
    M, N, K = 4, 768, 1536
    x, w, b = rnd(M, K), rnd(N, K, std=0.03), rnd(N, std=0.3)
    ref = x.float() @ w.float().t() + b.float()
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    ops.skinny(L.PRO_PLAIN, L.SK_BIAS, x, ops.pack_skinny(w), M, out=out, ldo=N, bias=b)
    close(out, ref, name='bias')
    ops.skinny(L.PRO_PLAIN, L.SK_BIAS_SILU, x, ops.pack_skinny(w), M, out=out, ldo=N, bias=b)
    close(out, F.silu(ref.to(BF).float()), name='bias_silu')
    N2 = 1002
    w2 = rnd(N2, K, std=0.03, seed=9)
    lg = torch.zeros(M, N2, dtype=torch.float32, device='cuda')
    ops.skinny(L.PRO_PLAIN, L.SK_F32, x, ops.pack_skinny(w2), M, out_f32=lg)
    close(lg, x.float() @ w2.float().t(), rtol=2e-3, atol=2e-3, name='f32 edge N')


@pytest.mark.parametrize('M,H,I,npart', [(4, 768, 8960, 7), (1, 1536, 8960, 0), (4, 768, 8960, 3), (16, 768, 8960, 8), (3, 2048, 2048, 2)])
def test_skinny_norm_swiglu(ops, M, H, I, npart):
    from vlaser_amd import _lib as L
    h, nw = rnd(M, H), (1 + 0.1 * rnd(H, seed=5).float()).to(BF)
    parts = (torch.randn(max(npart, 1), M, H, generator=torch.Generator().manual_seed(7)) * 0.3).cuda()
    g, u = rnd(I, H, std=0.03, seed=1), rnd(I, H, std=0.03, seed=2)
    W = ops.pack_gate_up(g, u)
    out = torch.zeros(M, I, dtype=BF, device='cuda'); h_out = torch.zeros(M, H, dtype=BF, device='cuda')
    if H == 768:      # 96-row units (tiles_per_unit = 6; slower, and the only skinny variant that spilled) are refused since r05
        with pytest.raises(L.VlaserHipError, match='tiles_per_unit'):
            ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, ops.pack_skinny(W, 1, 6), M, partials=parts, n_partials=npart, norm_w=nw, out=out, ldo=I)
    ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, ops.pack_skinny(W), M, partials=parts, n_partials=npart, norm_w=nw, h_out=h_out, out=out, ldo=I)
    hs = (h.float() + (parts[:npart].sum(0) if npart else 0)).to(BF)
    assert torch.equal(h_out, hs) or (h_out.float() - hs.float()).abs().max() <= 2 ** -7 * hs.float().abs().max()
    xn = _rms_ref(h_out, nw).to(BF).float()
    gr = (xn @ g.float().t()).to(BF).float(); ur = (xn @ u.float().t()).to(BF).float()
    close(out, F.silu(gr).to(BF).float() * ur, name='norm swiglu')
    if H in (768, 1536):      # 16-row lane-local units (tiles_per_unit = 1, r03): same K order per output -> bit-identical
        out16 = torch.zeros(M, I, dtype=BF, device='cuda'); h16 = torch.zeros(M, H, dtype=BF, device='cuda')
        pw = ops.pack_skinny(ops.pack_gate_up8(g, u), 1, 1)
        assert pw.tpu == 1 and pw.N == 2 * I
        ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, pw, M, partials=parts, n_partials=npart, norm_w=nw, h_out=h16, out=out16, ldo=I)
        assert torch.equal(out16, out) and torch.equal(h16, h_out)


def test_skinny_norm_qkv_rope(ops):
    from vlaser_amd import _lib as L
    B, tok, H, nq, nkv, smax = 2, 4, 768, 12, 2, 448
    M = B * tok
    h, nw = rnd(M, H), (1 + 0.1 * rnd(H, seed=5).float()).to(BF)
    qw, kw, vw = rnd(nq * 128, H, std=0.03, seed=1), rnd(nkv * 128, H, std=0.03, seed=2), rnd(nkv * 128, H, std=0.03, seed=3)
    qb, kb, vb = rnd(nq * 128, std=0.3, seed=4), rnd(nkv * 128, std=0.3, seed=5), rnd(nkv * 128, std=0.3, seed=6)
    W, Bv = ops.pack_qkv(qw, kw, vw, qb, kb, vb)
    cos, sin = ops.rope_table(64)
    pos = (torch.arange(tok).repeat(B) + 2).int().cuda()
    q_out = torch.zeros(M, nq * 128, dtype=BF, device='cuda')
    kc = torch.zeros(B, nkv, smax, 128, dtype=BF, device='cuda'); vtc = torch.zeros(B, nkv, 128, smax, dtype=BF, device='cuda')
    ops.skinny(L.PRO_NORM, L.SK_QKV_ROPE, h, ops.pack_skinny(W), M, n_partials=0, norm_w=nw, bias=Bv, q_out=q_out, k_cache=kc, vt_cache=vtc,
               rope_cos=cos, rope_sin=sin, pos_ids=pos, n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=tok, slot_base=385)
    xn = _rms_ref(h, nw).to(BF).float()
    q = (xn @ qw.float().t() + qb.float()).to(BF).float().view(M, nq, 128)
    k = (xn @ kw.float().t() + kb.float()).to(BF).float().view(M, nkv, 128)
    v = (xn @ vw.float().t() + vb.float()).to(BF).float().view(M, nkv, 128)
    close(q_out.view(M, nq, 128), _rope_ref(q, pos), name='q')
    close(kc[:, :, 385:389], _rope_ref(k, pos).view(B, tok, nkv, 128).permute(0, 2, 1, 3), name='k')
    close(vtc[:, :, :, 385:389], v.view(B, tok, nkv, 128).permute(0, 2, 3, 1), name='vT')
    # 16-row lane-local units (tiles_per_unit = 1, r03): 128 instead of 64 units, bit-identical outputs
    W16, B16 = ops.pack_qkv16(qw, kw, vw, qb, kb, vb)
    q16 = torch.zeros_like(q_out); kc16 = torch.zeros_like(kc); vtc16 = torch.zeros_like(vtc)
    ops.skinny(L.PRO_NORM, L.SK_QKV_ROPE, h, ops.pack_skinny(W16, 1, 1), M, n_partials=0, norm_w=nw, bias=B16, q_out=q16, k_cache=kc16, vt_cache=vtc16,
               rope_cos=cos, rope_sin=sin, pos_ids=pos, n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=tok, slot_base=385)
    assert torch.equal(q16, q_out) and torch.equal(kc16, kc) and torch.equal(vtc16, vtc)


@pytest.mark.parametrize('M,H,I,npart', [(4, 768, 8960, 3), (5, 768, 8960, 3), (1, 1536, 8960, 2), (8, 1536, 8960, 2), (16, 768, 8960, 3), (11, 768, 8960, 2), (3, 768, 5120, 3)])
def test_chain_gu_bit_identical_to_skinny(ops, M, H, I, npart):
    """r05 csrc/chain.hip: gate/up with every unit of the workgroup requested up front == skinny_kernel<NORM, SWIGLU> bit for bit (activations and the rounded residual
    stream), for 1-3 chunks per thread and 2 / 3 units per workgroup; ten runs beside a second stream stay identical (race screen of the reordered loads)."""
    from vlaser_amd import _lib as L
    assert ops.chain_gu_supported(M, 2 * I, H, npart)
    h, nw = rnd(M, H), (1 + 0.1 * rnd(H, seed=5).float()).to(BF)
    parts = (torch.randn(npart, M, H, generator=torch.Generator().manual_seed(7)) * 0.3).cuda()
    pw = ops.pack_skinny(ops.pack_gate_up(rnd(I, H, std=0.03, seed=1), rnd(I, H, std=0.03, seed=2)))
    o0, h0 = torch.zeros(M, I, dtype=BF, device='cuda'), torch.zeros(M, H, dtype=BF, device='cuda')
    ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, pw, M, partials=parts, n_partials=npart, norm_w=nw, h_out=h0, out=o0, ldo=I)
    a, _ = ops.skinny_args(h, pw, M, partials=parts, n_partials=npart, norm_w=nw, h_out=None, out=None, ldo=I)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device='cuda')
    for rep in range(10):
        o1, h1 = torch.full((M, I), 7.0, dtype=BF, device='cuda'), torch.full((M, H), 7.0, dtype=BF, device='cuda')
        a.out, a.h_out = o1.data_ptr(), h1.data_ptr()
        with torch.cuda.stream(side):
            junk @ junk
        ops.launch_chain_gu(a)
        assert torch.equal(o1, o0) and torch.equal(h1, h0), rep
    torch.cuda.synchronize()
    assert not ops.chain_gu_supported(M, 2 * I, 1024, npart) and not ops.chain_gu_supported(M, 2 * I, H, 5)
    # 16-row lane-local units (pack_gate_up8: 224 workgroups x 5 units for the expert): the same bits again
    if I == 8960 and ops.chain_gu_supported(M, 2 * I, H, npart, 1):
        pw8 = ops.pack_skinny(ops.pack_gate_up8(rnd(I, H, std=0.03, seed=1), rnd(I, H, std=0.03, seed=2)), 1, 1)
        a8, _ = ops.skinny_args(h, pw8, M, partials=parts, n_partials=npart, norm_w=nw, h_out=None, out=None, ldo=I)
        o8, h8 = torch.full((M, I), 7.0, dtype=BF, device='cuda'), torch.full((M, H), 7.0, dtype=BF, device='cuda')
        a8.out, a8.h_out = o8.data_ptr(), h8.data_ptr()
        ops.launch_chain_gu(a8)
        assert torch.equal(o8, o0) and torch.equal(h8, h0)


@pytest.mark.parametrize('B,tok,H', [(1, 4, 768), (1, 5, 768), (2, 4, 768), (4, 4, 768), (1, 1, 1536), (8, 1, 1536)])
def test_chain_qkv_vs_skinny_and_fp32(ops, B, tok, H):
    """r05: one wave per 16-row q/k/v unit over the whole K (hidden 768; two waves, one K half each, at hidden 1536): q, K cache and V^T cache vs the fp32 reference and vs the 8-wave skinny
    kernel (different fp32 summation order: bf16-level agreement), untouched cache slots stay untouched."""
    from vlaser_amd import _lib as L
    nq, nkv, smax = 12, 2, 448
    M = B * tok
    assert ops.chain_qkv_supported(M, (nq + 2 * nkv) * 128, H)
    h, nw = rnd(M, H), (1 + 0.1 * rnd(H, seed=5).float()).to(BF)
    qw, kw, vw = rnd(nq * 128, H, std=0.03, seed=1), rnd(nkv * 128, H, std=0.03, seed=2), rnd(nkv * 128, H, std=0.03, seed=3)
    qb, kb, vb = rnd(nq * 128, std=0.3, seed=4), rnd(nkv * 128, std=0.3, seed=5), rnd(nkv * 128, std=0.3, seed=6)
    W16, B16 = ops.pack_qkv16(qw, kw, vw, qb, kb, vb)
    pw = ops.pack_skinny(W16, 1, 1)
    cos, sin = ops.rope_table(64)
    pos = (torch.arange(tok).repeat(B) + 2).int().cuda()
    outs = []
    for chain in (False, True):
        q_out = torch.zeros(M, nq * 128, dtype=BF, device='cuda')
        kc = torch.full((B, nkv, smax, 128), 3.0, dtype=BF, device='cuda'); vtc = torch.full((B, nkv, 128, smax), 3.0, dtype=BF, device='cuda')
        a, keep = ops.skinny_args(h, pw, M, n_partials=0, norm_w=nw, bias=B16, q_out=q_out, k_cache=kc, vt_cache=vtc, rope_cos=cos, rope_sin=sin, pos_ids=pos,
                                  n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=tok, slot_base=385)
        (ops.launch_chain_qkv(a) if chain else ops.launch_skinny(L.PRO_NORM, L.SK_QKV_ROPE, a))
        outs.append((q_out, kc, vtc))
    xn = _rms_ref(h, nw).to(BF).float()
    q = (xn @ qw.float().t() + qb.float()).to(BF).float().view(M, nq, 128)
    k = (xn @ kw.float().t() + kb.float()).to(BF).float().view(M, nkv, 128)
    v = (xn @ vw.float().t() + vb.float()).to(BF).float().view(M, nkv, 128)
    q_out, kc, vtc = outs[1]
    close(q_out.view(M, nq, 128), _rope_ref(q, pos), name='q')
    close(kc[:, :, 385:385 + tok], _rope_ref(k, pos).view(B, tok, nkv, 128).permute(0, 2, 1, 3), name='k')
    close(vtc[:, :, :, 385:385 + tok], v.view(B, tok, nkv, 128).permute(0, 2, 3, 1), name='vT')
    assert float((kc[:, :, :385].float() - 3).abs().max()) == 0 and float((vtc[:, :, :, 385 + tok:].float() - 3).abs().max()) == 0
    for a_, b_ in zip(outs[0], outs[1]):           # vs the 8-wave kernel: the same values up to the fp32 summation order
        assert (a_.float() - b_.float()).abs().max().item() <= 2 ** -6 * max(1.0, b_.float().abs().max().item())
    if H == 1536:          # hidden 1536 runs two waves per unit (one K half each) by default; the one-wave kernel stays reachable and agrees to the summation order
        prev = L.lib().vlaser_chain_qkv_set_waves(1)
        try:
            assert prev == 0
            q1 = torch.zeros(M, nq * 128, dtype=BF, device='cuda')
            kc1 = torch.full((B, nkv, smax, 128), 3.0, dtype=BF, device='cuda'); vtc1 = torch.full((B, nkv, 128, smax), 3.0, dtype=BF, device='cuda')
            a, keep = ops.skinny_args(h, pw, M, n_partials=0, norm_w=nw, bias=B16, q_out=q1, k_cache=kc1, vt_cache=vtc1, rope_cos=cos, rope_sin=sin, pos_ids=pos,
                                      n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=tok, slot_base=385)
            ops.launch_chain_qkv(a)
        finally:
            L.lib().vlaser_chain_qkv_set_waves(prev)
        close(q1.view(M, nq, 128), _rope_ref(q, pos), name='q, one wave per unit')
        for a_, b_ in zip((q1, kc1, vtc1), outs[1]):
            assert (a_.float() - b_.float()).abs().max().item() <= 2 ** -6 * max(1.0, b_.float().abs().max().item())
    # slot_base < 0: the cache slot is the row's position id (graph-replayable decode)
    kc2 = torch.zeros(B, nkv, smax, 128, dtype=BF, device='cuda'); vtc2 = torch.zeros(B, nkv, 128, smax, dtype=BF, device='cuda')
    pos2 = (torch.arange(tok).repeat(B) + 100).int().cuda()
    a, keep = ops.skinny_args(h, pw, M, n_partials=0, norm_w=nw, bias=B16, q_out=q_out, k_cache=kc2, vt_cache=vtc2, rope_cos=ops.rope_table(256)[0], rope_sin=ops.rope_table(256)[1],
                              pos_ids=pos2, n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=tok, slot_base=-1)
    ops.launch_chain_qkv(a)
    close(kc2[:, :, 100:100 + tok], _rope_ref(k, pos2).view(B, tok, nkv, 128).permute(0, 2, 1, 3), name='k slot = position')
    assert float(kc2[:, :, :100].float().abs().max()) == 0


@pytest.mark.parametrize('M,N', [(4, 768), (5, 768), (1, 1536), (8, 1536), (16, 768), (13, 768), (3, 772)])
def test_chain_down_vs_fp32(ops, M, N):
    """r05: the down projection without cross-workgroup split-K (4 output columns per workgroup over the whole K = 8960, v_mfma_f32_4x4x4_16b_bf16) + residual ->
    the bf16 residual stream: vs fp32, deterministic over ten runs beside a second stream, rows / columns outside [M, N] untouched."""
    K = 8960
    assert ops.chain_down_supported(M, N, K)
    x, w, res = rnd(M, K, std=0.7), rnd(N, K, std=0.03, seed=1), rnd(M, N, std=1.5, seed=2)
    w4 = ops.pack_down4(w)
    ref = res.float() + x.float() @ w.float().t()
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device='cuda')
    first = None
    for rep in range(10):
        out = torch.full((M + 1, N), 9.0, dtype=BF, device='cuda')
        with torch.cuda.stream(side):
            junk @ junk
        ops.chain_down(x, w4, res, out, M, N, K)
        first = out.clone() if first is None else first
        assert torch.equal(out, first), rep
    torch.cuda.synchronize()
    close(first[:M], ref, rtol=8e-3, atol=8e-3 * ref.abs().max().item(), name='down4')
    assert float((first[M].float() - 9).abs().max()) == 0
    # a strided activation buffer (the engine's [16, I] workspace) and the unsupported widths
    xs = torch.zeros(16, K + 64, dtype=BF, device='cuda'); xs[:M, :K] = x
    out2 = torch.zeros(M, N, dtype=BF, device='cuda')
    ops.chain_down(xs, w4, res, out2, M, N, K)
    assert torch.equal(out2, first[:M])
    assert not ops.chain_down_supported(M, N, 18944) and not ops.chain_down_supported(17, N, K)


@pytest.mark.parametrize('M,N', [(4, 768), (5, 768), (1, 1536), (4, 1536), (8, 768), (3, 776)])
def test_chain_down2_and_qkv_slabs(ops, M, N):
    """r05: the down projection as two K halves on six-column workgroups (vlaser_chain_down2: fp32 slabs, no residual) -- the slabs sum to x @ W^T, deterministic --
    and its consumer: vlaser_chain_qkv with n_partials = 2 == the same launch on the pre-reduced stream h = bf16(res + slab 0 + slab 1) bit for bit, h stored by unit 0."""
    from vlaser_amd import _lib as L
    K = 8960
    assert bool(L.lib().vlaser_chain_down2_supported(M, N, K))
    x, w = rnd(M, K, std=0.7), rnd(N, K, std=0.03, seed=1)
    w42 = ops.pack_down4(w, k_splits=2)
    slabs = torch.full((2, M + 1, N), 9.0, device='cuda')[:, :M].contiguous()
    side = torch.cuda.Stream(); junk = torch.randn(4096, 4096, device='cuda')
    first = None
    for rep in range(6):
        out = torch.full((2 * M * N + 8,), 9.0, device='cuda')
        with torch.cuda.stream(side):
            junk @ junk
        ops.chain_down2(x, w42, out, M, N, K)
        first = out.clone() if first is None else first
        assert torch.equal(out, first), rep
    torch.cuda.synchronize()
    assert float((first[2 * M * N:] - 9).abs().max()) == 0
    sl = first[:2 * M * N].view(2, M, N)
    ref = x.float() @ w.float().t()
    close(sl.sum(0), ref, rtol=4e-3, atol=4e-3 * ref.abs().max().item(), name='sum of the two K halves')
    close(sl[0], x[:, :K // 2].float() @ w[:, :K // 2].float().t(), rtol=4e-3, atol=4e-3 * ref.abs().max().item(), name='first half')
    if N not in (768, 1536) or not bool(L.lib().vlaser_chain_qkv2_supported(M, 2048, N)):
        return
    # the consumer: q/k/v from (res, slabs) == q/k/v from h = bf16(res + s0 + s1)
    H, nq, nkv, smax, tok = N, 12, 2, 448, M
    res, nw = rnd(M, H, std=1.5, seed=2), (1 + 0.1 * rnd(H, seed=5).float()).to(BF)
    h = (sl[0] + sl[1] + res.float()).to(BF)
    W16, B16 = ops.pack_qkv16(rnd(nq * 128, H, std=0.03, seed=1), rnd(nkv * 128, H, std=0.03, seed=2), rnd(nkv * 128, H, std=0.03, seed=3), rnd(nq * 128, std=0.3, seed=4),
                              rnd(nkv * 128, std=0.3, seed=5), rnd(nkv * 128, std=0.3, seed=6))
    pw = ops.pack_skinny(W16, 1, 1)
    cos, sin = ops.rope_table(64)
    pos = (torch.arange(tok) + 2).int().cuda()
    outs = []
    from vlaser_amd import _lib as L
    prev = L.lib().vlaser_chain_qkv_set_waves(1)       # the slab form runs one wave per unit: the bit-level comparison needs the plain form in the same summation order (hidden 1536 defaults to two)
    try:
        for x_in, parts, npart in ((h, None, 0), (res, sl.contiguous(), 2)):
            q_out = torch.zeros(M, nq * 128, dtype=BF, device='cuda')
            kc = torch.zeros(1, nkv, smax, 128, dtype=BF, device='cuda'); vtc = torch.zeros(1, nkv, 128, smax, dtype=BF, device='cuda')
            hA = torch.full((M, H), 7.0, dtype=BF, device='cuda')
            a, keep = ops.skinny_args(x_in, pw, M, partials=parts, n_partials=npart, norm_w=nw, h_out=hA, bias=B16, q_out=q_out, k_cache=kc, vt_cache=vtc, rope_cos=cos, rope_sin=sin,
                                      pos_ids=pos, n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=tok, slot_base=385)
            ops.launch_chain_qkv(a)
            outs.append((q_out, kc, vtc, hA))
    finally:
        L.lib().vlaser_chain_qkv_set_waves(prev)
    for a_, b_ in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a_, b_)
    assert torch.equal(outs[1][3], h) and float((outs[0][3].float() - 7).abs().max()) == 0


def test_norms(ops):
    x = rnd(1025, 1024, std=2.0); w = (1 + 0.1 * rnd(1024, seed=1).float()).to(BF); b = rnd(1024, std=0.1, seed=2)
    close(ops.layernorm(x, w, b, 1e-6), F.layer_norm(x.float(), (1024,), w.float(), b.float(), 1e-6), name='layernorm')
    x2 = rnd(385, 1536, std=3.0); w2 = (1 + 0.1 * rnd(1536, seed=1).float()).to(BF)
    close(ops.rmsnorm(x2, w2, 1e-6), _rms_ref(x2, w2), rtol=8e-3, name='rmsnorm')


def test_pixel_shuffle_bit_exact_and_ln(ops):
    T, G, Cc = 2, 32, 1024
    x = rnd(T, G * G + 1, Cc)
    out = torch.zeros(T * 256, 4 * Cc, dtype=BF, device='cuda')
    ops.pixel_shuffle(x, out, T, G, Cc)
    xr = x[:, 1:].reshape(T, G, G, Cc)
    # closed form: out[n,i,j,a*2C+b*C+k] = x[n,2i+a,2j+b,k]
    ref = xr.view(T, 16, 2, 16, 2, Cc).permute(0, 1, 3, 2, 4, 5).reshape(T * 256, 4 * Cc)
    assert torch.equal(out, ref)
    w = (1 + 0.1 * rnd(4 * Cc, seed=1).float()).to(BF); b = rnd(4 * Cc, std=0.1, seed=2)
    o2 = torch.zeros_like(out)
    ops.pixel_shuffle_ln(x, w, b, o2, T, G, Cc, 1e-5)
    close(o2, F.layer_norm(ref.float(), (4 * Cc,), w.float(), b.float(), 1e-5), name='ps+ln')


def test_patch_embed(ops):
    from vlaser_amd import _lib as L
    T = 2
    pix = rnd(T, 3, 448, 448)
    w = rnd(1024, 3, 14, 14, std=0.05, seed=1); b = rnd(1024, std=0.1, seed=2)
    cls = rnd(1, 1, 1024, seed=3); pos = rnd(1, 1025, 1024, seed=4)
    A = torch.zeros(T * 1024, 640, dtype=BF, device='cuda')
    ops.im2col(pix, A, T, 448, 640)
    patch = ops.linear(A, ops.pack_patch_embed(w), b)
    h = torch.zeros(T, 1025, 1024, dtype=BF, device='cuda')
    ops.vit_assemble(patch, cls, pos, h, T, 1024, 1024)
    ref = F.conv2d(pix.float(), w.float(), b.float(), stride=14).to(BF).float().flatten(2).transpose(1, 2)
    ref = torch.cat([cls.float().expand(T, 1, -1), ref], 1) + pos.float()
    close(h, ref, name='patch embed')


def test_embed_merge_and_argmax(ops):
    V, H = 3000, 1536
    embed = rnd(V, H); vit = rnd(512, H, seed=5)
    ids = torch.randint(0, 1000, (2, 300))
    ids[0, 10:266] = 2999; ids[1, 40:296] = 2999; ids[1, 296:] = 1500      # 1500 = pad
    ids = ids.cuda()
    out = torch.zeros(2, 300, H, dtype=BF, device='cuda'); rank = torch.zeros(600, dtype=torch.int32, device='cuda')
    cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.embed_merge(ids, embed, vit, out, 2999, 1500, False, rank, cnt)
    ref = embed[ids].clone(); ref[ids == 2999] = vit
    assert torch.equal(out, ref) and int(cnt) == 512
    sel = (ids.flatten() == 2999)
    assert torch.equal(rank[sel].cpu(), torch.arange(512, dtype=torch.int32))       # visual-token indices bit-exact
    assert (rank[~sel] == -1).all()
    ops.embed_merge(ids, embed, vit, out, 2999, 1500, True, rank, cnt)
    ref[ids == 1500] = 0
    assert torch.equal(out, ref)
    logits = torch.randn(3, 151674, device='cuda'); logits[1, 777] = 50; logits[2, 5] = 60; logits[2, 100000] = 60
    oid = torch.zeros(3, dtype=torch.int64, device='cuda'); nh = torch.zeros(3, H, dtype=BF, device='cuda')
    big = rnd(151674, 64)
    nh = torch.zeros(3, 64, dtype=BF, device='cuda')
    ops.argmax(logits, oid, big, nh)
    assert torch.equal(oid, logits.argmax(-1)) and oid[2] == 5
    assert torch.equal(nh, big[oid])


@pytest.mark.parametrize('N', [7, 1001, 8193, 16386, 151674, 151675])
def test_argmax_widths_and_ties(ops, N):
    """vlaser_argmax == torch.argmax (lowest index wins ties) at odd widths (rows of an odd N are only 4-byte aligned: the 8-byte loads fall back), with ties
    planted across the unrolled / remainder / tail loops of the kernel"""
    g = torch.Generator(device='cuda').manual_seed(N)
    logits = torch.randn(4, N, device='cuda', generator=g)
    logits[1, N - 1] = 40.0
    logits[2, N - 1] = 40.0; logits[2, N // 2] = 40.0; logits[2, min(N - 1, 2 * 8192 + 1)] = 40.0
    logits[3, :] = 1.0
    oid = torch.full((4,), -1, dtype=torch.int64, device='cuda')
    ops.argmax(logits, oid, None, None)
    assert torch.equal(oid, logits.argmax(-1)) and oid[3] == 0 and oid[1] == N - 1
    # r05: the same rows spread over 64 workgroups each (last arriver folds the pairs and re-arms the counter): repeated launches on one workspace, + the embedding gather
    ws = ops.argmax_workspace(16, 'cuda')                 # capacity 16 rows, launches of 4 / 1 / 3 rows on it (the decode's buffers: one workspace, prefill M = 1, batch M = B)
    emb = rnd(N, 64, seed=3)
    o1 = torch.full((1,), -1, dtype=torch.int64, device='cuda')
    ops.argmax(logits[2:3], o1, None, None, ws=ws)
    assert o1[0] == logits[2].argmax()
    for rep in range(3):
        oid2 = torch.full((4,), -1, dtype=torch.int64, device='cuda'); nh = torch.zeros(4, 64, dtype=BF, device='cuda')
        ops.argmax(logits, oid2, emb, nh, ws=ws)
        assert torch.equal(oid2, oid) and torch.equal(nh, emb[oid])
        o3 = torch.full((3,), -1, dtype=torch.int64, device='cuda')
        ops.argmax(logits[1:], o3, None, None, ws=ws)
        assert torch.equal(o3, oid[1:])
    sub = logits[:, 1:].contiguous()                      # the other row alignment (N - 1 columns)
    oid3 = torch.full((4,), -1, dtype=torch.int64, device='cuda')
    ops.argmax(sub, oid3, None, None, ws=ws)
    assert torch.equal(oid3, sub.argmax(-1))
    if N >= 4096:                                         # (narrower rows take the one-workgroup kernel and never look at the workspace)
        from vlaser_amd import _lib as L
        with pytest.raises(L.VlaserHipError):
            ops.argmax(logits, oid2, None, None, ws=ws[:16])


def test_vla_glue(ops):
    W, adim, M = 768, 7, 8
    act = torch.randn(M, adim).cuda(); w1 = rnd(W, adim, std=0.2); b1 = rnd(W, std=0.1, seed=1)
    xcat = torch.zeros(M, 2 * W, dtype=BF, device='cuda')
    t = 0.3
    ops.vla_prep(act, w1, b1, xcat, M, W, adim, t, 10000.0)
    half = W // 2
    e = math.log(10000.0) / (half - 1)
    ang = t * torch.exp(torch.arange(half).float() * -e)
    temb = torch.cat([ang.sin(), ang.cos()]).cuda()
    close(xcat[:, :W], temb[None].expand(M, -1), rtol=1e-2, atol=1e-2, name='time emb')
    close(xcat[:, W:], act.to(BF).float() @ w1.float().t() + b1.float(), name='linear_1')
    # euler
    h = rnd(M, W); parts = (torch.randn(3, M, W) * 0.2).cuda(); nw = (1 + 0.1 * rnd(W, seed=2).float()).to(BF)
    wd = rnd(adim, W, std=0.02, seed=3); bd = rnd(adim, std=0.05, seed=4)
    a0 = act.clone(); vel = torch.zeros(M, adim, device='cuda')
    ops.vla_euler(h, parts, 3, M, nw, 1e-6, wd, bd, act, W, adim, 0.1, 1.0, False, vel)
    hs = (h.float() + parts.sum(0)).to(BF)
    y = _rms_ref(hs, nw).to(BF).float()
    vref = (y @ wd.float().t() + bd.float())
    close(vel, vref, rtol=1e-2, atol=5e-3, name='vel')
    close(act, a0 + 0.1 * vel, rtol=1e-5, atol=1e-6, name='euler')
    # r06: the reference's other integration methods (pizero_internvl.py:1309-1331) re-combine ONE velocity per step: the update is torch's arithmetic BIT FOR BIT
    # (every product and sum rounded separately), from the same decoder output
    from oracle.vla import integration_step
    assert torch.equal(act.cpu(), integration_step(a0.cpu(), 0.1, vel.cpu(), 'euler'))
    for method in ('heun', 'rk4'):
        a1 = a0.clone(); v1 = torch.zeros(M, adim, device='cuda')
        ops.vla_euler(h, parts, 3, M, nw, 1e-6, wd, bd, a1, W, adim, 0.1, 1.0, False, v1, method=method)
        assert torch.equal(v1, vel), method
        assert torch.equal(a1.cpu(), integration_step(a0.cpu(), 0.1, vel.cpu(), method)), method
    assert torch.equal(integration_step(a0.cpu(), 0.1, vel.cpu(), 'heun'), a0.cpu() + 0.1 * vel.cpu())        # heun == euler exactly (a power-of-two rescaling)


@pytest.mark.parametrize('bm', [32, 64, 128, 1100, 1200, 1300, 1440, 1500, 0])
def test_gemm_tile_heights(ops, bm):
    from vlaser_amd import _lib as L
    M, N, K = 385, 1536, 1536
    x, w, b = rnd(M, K), rnd(N, K, std=0.05), rnd(N, std=0.5)
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    ops.gemm(L.EPI_BIAS, x, w, out=out, bias=b, force_bm=bm)
    close(out, x.float() @ w.float().t() + b.float(), name=f'bm{bm}')
    g, u = rnd(2048, K, std=0.03, seed=1), rnd(2048, K, std=0.03, seed=2)
    o2 = torch.zeros(M, 2048, dtype=BF, device='cuda')
    ops.gemm(L.EPI_SWIGLU, x, ops.pack_gate_up(g, u), out=o2, force_bm=bm)
    gr = (x.float() @ g.float().t()).to(BF).float(); ur = (x.float() @ u.float().t()).to(BF).float()
    close(o2, F.silu(gr).to(BF).float() * ur, name=f'swiglu bm{bm}')


@pytest.mark.parametrize('bm', [32, 64, 1100, 1200, 1300, 1440, 1500])
def test_gemm_qkv_rope_small_tiles(ops, bm):
    from vlaser_amd import _lib as L
    B, S, H, nq, nkv, smax = 1, 100, 1536, 12, 2, 128
    M = B * S
    x = rnd(M, H)
    qw, kw, vw = rnd(nq * 128, H, std=0.03, seed=1), rnd(nkv * 128, H, std=0.03, seed=2), rnd(nkv * 128, H, std=0.03, seed=3)
    qb, kb, vb = rnd(nq * 128, std=0.3, seed=4), rnd(nkv * 128, std=0.3, seed=5), rnd(nkv * 128, std=0.3, seed=6)
    W, Bv = ops.pack_qkv(qw, kw, vw, qb, kb, vb)
    cos, sin = ops.rope_table(512)
    pos = (torch.arange(S).repeat(B) + 1).int().cuda()
    q_out = torch.zeros(M, nq * 128, dtype=BF, device='cuda')
    kc = torch.zeros(B, nkv, smax, 128, dtype=BF, device='cuda'); vtc = torch.zeros(B, nkv, 128, smax, dtype=BF, device='cuda')
    ops.gemm(L.EPI_QKV_ROPE, x, W, bias=Bv, q_out=q_out, k_cache=kc, vt_cache=vtc, rope_cos=cos, rope_sin=sin, pos_ids=pos,
             n_q_heads=nq, n_kv_heads=nkv, s_max=smax, tok_per_batch=S, slot_base=0, force_bm=bm)
    xf = x.float()
    q = (xf @ qw.float().t() + qb.float()).to(BF).float().view(M, nq, 128)
    k = (xf @ kw.float().t() + kb.float()).to(BF).float().view(M, nkv, 128)
    v = (xf @ vw.float().t() + vb.float()).to(BF).float().view(M, nkv, 128)
    close(q_out.view(M, nq, 128), _rope_ref(q, pos), name='q')
    close(kc[:, :, :S], _rope_ref(k, pos).view(B, S, nkv, 128).permute(0, 2, 1, 3), name='k')
    close(vtc[:, :, :, :S], v.view(B, S, nkv, 128).permute(0, 2, 3, 1), name='vT')


@pytest.mark.parametrize('M,N,K,S', [(385, 1536, 8960, 7), (1025, 1024, 4096, 4), (100, 1024, 1024, 1), (130, 3584, 3584, 2), (64, 1536, 1536, 3),
                                     (50, 2048, 1024, 2), (33, 3584, 3584, 7)])
def test_gemm_splitk_reduce_norm(ops, M, N, K, S):
    # widths 1024 / 1536 / 3584 take the exact-shape seam kernels (all loads up front), 2048 and 3584 x 7 slabs the generic one
    from vlaser_amd import _lib as L
    x, w = rnd(M, K), rnd(N, K, std=0.03)
    part = torch.zeros(S, M, N, dtype=torch.float32, device='cuda')
    ops.gemm(L.EPI_PARTIAL, x, w, out_f32=part, k_splits=S)
    ref = x.float() @ w.float().t()
    close(part.sum(0), ref, rtol=2e-3, atol=2e-3, name='partials')
    h = rnd(M, N, seed=3); nw = (1 + 0.1 * rnd(N, seed=4).float()).to(BF); nb = rnd(N, std=0.1, seed=5)
    bias = rnd(N, std=0.3, seed=6); ls = rnd(N, std=0.1, seed=7)
    # RMS seam (Qwen2): h += sum; x = rms(h) * w
    ho = torch.zeros_like(h); xo = torch.zeros_like(h)
    ops.reduce_norm(h, part, S, M, N, ho, xo, norm=1, norm_w=nw)
    href = (h.float() + ref).to(BF)
    close(ho, href.float(), rtol=8e-3, name='h rms')
    close(xo, _rms_ref(ho, nw), rtol=8e-3, name='x rms')
    # LayerNorm seam (ViT): h += ls * (sum + bias); x = LN(h); in place
    h2 = h.clone(); xo2 = torch.zeros_like(h)
    ops.reduce_norm(h2, part, S, M, N, h2, xo2, bias=bias, ls=ls, norm=2, norm_w=nw, norm_b=nb, eps=1e-6)
    href2 = h.float() + ls.float() * (ref + bias.float())
    close(h2, href2, rtol=8e-3, name='h ln')
    close(xo2, F.layer_norm(h2.float(), (N,), nw.float(), nb.float(), 1e-6), name='x ln')
    # no norm
    h3 = torch.zeros_like(h)
    ops.reduce_norm(h, part, S, M, N, h3)
    close(h3, href.float(), rtol=8e-3, name='h none')


@pytest.mark.parametrize('bm', [1100, 1105, 1200, 1300, 1440, 1500, 1506, 1532, 1564, 1900])
def test_gemm_glds_ragged_and_splitk(ops, bm):
    """LDS-DMA pipelines on shapes that do not fill their tiles: ragged M and N (fp32 logits epilogue), K shorter than the stage
    ring (look-ahead tiles are clamped re-fetches), and split-K partial slabs."""
    from vlaser_amd import _lib as L
    for (M, N, K) in [(130, 1000, 64), (257, 300, 128), (70, 520, 1536)]:
        x, w = rnd(M, K), rnd(N, K, std=0.05, seed=M)
        out = torch.zeros(M, N, dtype=torch.float32, device='cuda')
        ops.gemm(L.EPI_F32, x, w, out=out, force_bm=bm)
        close(out, x.float() @ w.float().t(), rtol=2e-3, atol=2e-3, name=f'f32 {M}x{N}x{K} cfg{bm}')
    M, N, K, S = 385, 1024, 2048, 4
    x, w = rnd(M, K), rnd(N, K, std=0.05, seed=9)
    part = torch.zeros(S, M, N, dtype=torch.float32, device='cuda')
    ops.gemm(L.EPI_PARTIAL, x, w, out_f32=part, k_splits=S, force_bm=bm)
    close(part.sum(0), x.float() @ w.float().t(), rtol=2e-3, atol=2e-3, name=f'split-K cfg{bm}')


@pytest.mark.parametrize('asym,two_stage', [(1900, 1901), (1300, 1302), (1903, 1901), (1100, 1101), (1200, 1201), (1440, 1441), (1500, 1501), (1904, 1901), (1304, 1302),
                                            (1110, 1101), (1210, 1201), (1442, 1441), (1502, 1501), (1902, 1901), (1564, 1566), (1310, 1302)])      # r06: 1440 / 1500 / 1564 / 1900 pipeline their fragment reads across the barrier (1442 / 1502 / 1566 / 1902 = the plain loops; 1110 / 1210 = lab)
def test_gemm_asymmetric_ring_is_bit_identical_to_the_two_stage_ring(ops, asym, two_stage):
    """r05: the 192x256 / 256x256 tiles carry a third stage for the W operand alone (weights two K-steps ahead, activations one), and every ring issues its refill one piece
    at a time between the K-step's MFMAs instead of as one burst (x01 / 1304 / 1904 = the burst forms).  Same arithmetic in the same order as the
    r03-r04 rings: bit-identical outputs -- NT and NN forms, K of 1 .. 24 tiles (shorter than, equal to and longer than both rings), ragged M / N, split-K slabs."""
    from vlaser_amd import _lib as L
    for (M, N, K) in [(560, 1792, 1536), (200, 520, 64), (257, 300, 128), (385, 1024, 192), (70, 777, 256), (3408 // 8, 2048, 3584)]:
        x, w = rnd(M, K), rnd(N, K, std=0.05, seed=M)
        o1, o2 = torch.zeros(M, N, dtype=BF, device='cuda'), torch.zeros(M, N, dtype=BF, device='cuda')
        ops.gemm(L.EPI_NONE, x, w, out=o1, force_bm=asym)
        ops.gemm(L.EPI_NONE, x, w, out=o2, force_bm=two_stage)
        assert torch.equal(o1, o2), (M, N, K)
        close(o1, x.float() @ w.float().t(), rtol=1e-2, atol=1e-2, name=f'nt {M}x{N}x{K} cfg{asym}')
        if N % 8 == 0 and asym not in (1564, 1310):
            wk = w.t().contiguous()
            ops.gemm_nn(L.EPI_NONE, x, wk, out=o1, force_bm=asym)
            ops.gemm_nn(L.EPI_NONE, x, wk, out=o2, force_bm=two_stage)
            assert torch.equal(o1, o2), ('nn', M, N, K)
    M, N, K, S = 385, 1024, 2048, 4
    x, w = rnd(M, K), rnd(N, K, std=0.05, seed=9)
    p1, p2 = torch.zeros(S, M, N, dtype=torch.float32, device='cuda'), torch.zeros(S, M, N, dtype=torch.float32, device='cuda')
    ops.gemm(L.EPI_PARTIAL, x, w, out_f32=p1, k_splits=S, force_bm=asym)
    ops.gemm(L.EPI_PARTIAL, x, w, out_f32=p2, k_splits=S, force_bm=two_stage)
    assert torch.equal(p1, p2)


def test_gemm_lds_attribute_is_set_per_kernel_in_any_order():
    """The dynamic-LDS limit is raised once per (kernel instantiation, device).  In a fresh process, run the LDS-DMA configurations
    from the largest stage ring to the smallest: every one must launch (a table shared by all instantiations let the first, largest
    request mask the later ones -- an order the model code did not hit until the 144-row configuration existed)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import torch\n"
        "from vlaser_amd import ops, _lib as L\n"
        "x = torch.randn(300, 512, device='cuda').bfloat16(); w = (0.05 * torch.randn(640, 512, device='cuda')).bfloat16()\n"
        "ref = x.float() @ w.float().t()\n"
        "for bm in (1200, 1440, 1100, 1300, 1500, 128, 64, 32):\n"
        "    out = torch.zeros(300, 640, dtype=torch.float32, device='cuda')\n"
        "    ops.gemm(L.EPI_F32, x, w, out=out, force_bm=bm)\n"
        "    torch.cuda.synchronize()\n"
        "    assert (out - ref).abs().max().item() < 2e-2, bm\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize('bm', [0, 1100, 1200, 1300, 1440, 1500, 1900])
def test_gemm_nn_reads_the_weight_as_stored(ops, bm):
    """NN form (dgrad dX = dY @ W with W [N_out, K_in] as the forward stores it): every LDS-DMA configuration against fp32 matmul on
    ragged M / N, a short contraction (K shorter than the stage ring), a strided B (a slice of a wider matrix), and split-K slabs."""
    from vlaser_amd import _lib as L
    for (M, N, K) in [(130, 1000, 64), (257, 304, 128), (70, 520, 1536), (560, 1536, 2048)]:
        x, w = rnd(M, K), rnd(K, N, std=0.05, seed=M)
        out = torch.zeros(M, N, dtype=BF, device='cuda')
        ops.gemm_nn(L.EPI_NONE, x, w, out=out, force_bm=bm)
        close(out, x.float() @ w.float(), rtol=1e-2, atol=1e-2, name=f'nn {M}x{N}x{K} cfg{bm}')
    # B = columns [256, 256 + 520) of a wider row-major matrix
    x, wide = rnd(96, 320), rnd(320, 1024, std=0.05, seed=5)
    b = wide[:, 256:776]
    out = torch.zeros(96, 520, dtype=BF, device='cuda')
    a = L.GemmArgs(); a.A, a.W, a.out = x.data_ptr(), b.data_ptr(), out.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldo, a.force_bm = 96, 520, 320, 320, 1024, 520, bm
    import ctypes
    L.check(L.lib().vlaser_gemm_nn(L.EPI_NONE, ctypes.byref(a), torch.cuda.current_stream().cuda_stream), 'vlaser_gemm_nn')
    close(out, x.float() @ b.float(), rtol=1e-2, atol=1e-2, name=f'nn strided cfg{bm}')
    M, N, K, S = 385, 1024, 2048, 4
    x, w = rnd(M, K), rnd(K, N, std=0.05, seed=9)
    part = torch.zeros(S, M, N, dtype=torch.float32, device='cuda')
    ops.gemm_nn(L.EPI_PARTIAL, x, w, out_f32=part, k_splits=S, force_bm=bm)
    close(part.sum(0), x.float() @ w.float(), rtol=2e-3, atol=2e-3, name=f'nn split-K cfg{bm}')


def test_skinny_chunked_k_vlaser_8b_widths(ops):
    """Chunked-K weight streaming: K / (k_splits * 256) = 14 or 16 steps per wave run as two chunks of 7 / 8 with
    the accumulators carried across chunks -- Vlaser-8B's hidden 3584 (qkv / gate+up / lm_head, NORM prologue) and its MLP width
    18944, zero-padded to 20480 = 5 splits x 16 steps (down projection)."""
    from vlaser_amd import _lib as L
    M, H, I = 5, 3584, 18944
    assert ops.skinny_geometry(I, H) == (20480, 5) and ops.skinny_geometry(H, H)[0] == H
    h = rnd(M, H); nw = (1 + 0.1 * rnd(H, seed=2).float()).to(BF)
    parts = (torch.randn(2, M, H, generator=torch.Generator().manual_seed(1)) * 0.1).cuda()
    hs = (h.float() + parts.sum(0)).to(BF)
    xn = _rms_ref(hs, nw).to(BF).float()
    # gate/up with the fused residual + RMSNorm prologue, K = 3584 (14 steps = 2 x 7)
    g, u = rnd(512, H, std=0.03, seed=3), rnd(512, H, std=0.03, seed=4)
    act = torch.zeros(M, 512, dtype=BF, device='cuda'); hout = torch.zeros(M, H, dtype=BF, device='cuda')
    ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, ops.pack_skinny(ops.pack_gate_up(g, u)), M, partials=parts, n_partials=2, norm_w=nw, h_out=hout, out=act, ldo=512)
    gr = (xn @ g.float().t()).to(BF).float(); ur = (xn @ u.float().t()).to(BF).float()
    close(act, F.silu(gr).to(BF).float() * ur, name='chunked NORM+SWIGLU')
    assert torch.equal(hout, hs)
    # fp32 logits head, ragged N
    wv = rnd(1000, H, std=0.03, seed=5)
    lg = torch.zeros(M, 1000, dtype=torch.float32, device='cuda')
    ops.skinny(L.PRO_NORM, L.SK_F32, h, ops.pack_skinny(wv), M, partials=parts, n_partials=2, norm_w=nw, out_f32=lg)
    close(lg, xn @ wv.float().t(), rtol=2e-3, atol=2e-2, name='chunked NORM+F32')
    # down projection: K = 18944 zero-padded to 20480, 5 split-K slabs of 16 steps (2 x 8)
    kp, ks = ops.skinny_geometry(I, H)
    x = torch.zeros(M, kp, dtype=BF, device='cuda'); x[:, :I] = rnd(M, I, seed=6)
    wd = rnd(256, I, std=0.03, seed=7)
    part = torch.zeros(ks, M, 256, dtype=torch.float32, device='cuda')
    ops.skinny(L.PRO_PLAIN, L.SK_PARTIAL, x, ops.pack_skinny(wd, ks, k_pad=kp), M, out_f32=part)
    close(part.sum(0), x[:, :I].float() @ wd.float().t(), rtol=2e-3, atol=2e-2, name='chunked padded down')
    # plain bias epilogue, 16 rows, K = 3584
    x16, wb, bb = rnd(16, H, seed=8), rnd(320, H, std=0.03, seed=9), rnd(320, std=0.3, seed=10)
    o = torch.zeros(16, 320, dtype=BF, device='cuda')
    ops.skinny(L.PRO_PLAIN, L.SK_BIAS, x16, ops.pack_skinny(wb), 16, out=o, ldo=320, bias=bb)
    close(o, x16.float() @ wb.float().t() + bb.float(), name='chunked PLAIN+BIAS M=16')


def test_normalize_u8_bit_exact_vs_host_preprocessing(ops):
    """VERDICT r02 #8 / SURVEY 8f-2: the device-side uint8 -> normalised bf16 kernel against the host restatements of the reference's
    two normalisations -- InternVLAProcessor (processing.py:303-311; `prep.vla_normalize_images`, pixel probe pinned by golden G1) and
    build_transform's ToTensor + Normalize (dataset.py:293-300; the arithmetic of `prep.normalize_tiles`) -- bit for bit after the one
    bf16 rounding, planar and interleaved inputs, every byte value in every channel."""
    from vlaser_amd import prep
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (2, 3, 448, 448), generator=g, dtype=torch.uint8)
    img[0, :, 0, :256] = torch.arange(256, dtype=torch.uint8)           # all 256 values in each channel
    out = torch.empty(2, 3, 448, 448, dtype=BF, device='cuda')
    ops.normalize_u8(img.cuda(), out, prep.VLA_MEAN, prep.VLA_STD, layout='chw', mode='vla')
    ref = prep.vla_normalize_images(img[:, None]).to(BF)
    assert torch.equal(out.cpu(), ref)
    # torchvision path: HWC uint8 tiles (PIL), ToTensor = /255, Normalize = (x - mean) / std
    hwc = img.permute(0, 2, 3, 1).contiguous()
    ops.normalize_u8(hwc.cuda(), out, prep.IMAGENET_MEAN, prep.IMAGENET_STD, layout='hwc', mode='totensor')
    mean, std = torch.tensor(prep.IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(prep.IMAGENET_STD).view(1, 3, 1, 1)
    ref2 = ((img.float() / 255.0 - mean) / std).to(BF)
    assert torch.equal(out.cpu(), ref2)
    assert not torch.equal(ref, ref2) or True                           # (the two orders of operations differ in fp32; after bf16 rounding they mostly coincide)


def test_infer_action_accepts_uint8_observation(golden_model):
    """The raw uint8 observation through `infer_action` == the host-normalised fp32 tensor through `infer_action` (same bf16 pixels)."""
    from vlaser_amd import prep
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (1, 3, 448, 448), generator=g, dtype=torch.uint8)
    ids = torch.full((1, 384), cfg.pad_token_id)
    ids[0, :10] = torch.randint(0, 151643, (10,), generator=g)
    ids[0, 10:266] = cfg.img_context_token_id
    ids[0, 266:277] = torch.randint(0, 151643, (11,), generator=g)
    pro, noise = torch.rand(1, 1, 7, generator=g) * 2 - 1, torch.randn(1, 4, 7, generator=g)
    m = PiZeroInference(vla, max_batch=1); m.load_state_dict(sd)
    a = m.infer_action(ids, img, proprios=pro, noise=noise)
    b = m.infer_action(ids, prep.vla_normalize_images(img[:, None]), proprios=pro, noise=noise)
    assert torch.equal(a, b)


def test_vla_stage_one_launch(ops):
    """r04: every per-call input of infer_action staged by ONE launch (vlaser_vla_stage) == the r03 sequence of copies / casts, bit for bit: ids, valid_len
    (given as int32 / int64, or counted from the pad ids), proprio, noise, pixels as bf16 / fp32 / uint8 (normalised like vlaser_normalize_u8), call counter."""
    from vlaser_amd import prep
    g = torch.Generator().manual_seed(5)
    B, T, pad = 3, 384, 151643
    ids = torch.randint(0, 151643, (B, T), generator=g)
    lens = [277, 384, 1]
    for b, n in enumerate(lens):
        ids[b, n:] = pad
    pro, nz = torch.rand(B, 7, generator=g), torch.randn(B * 4, 7, generator=g)
    u8 = torch.randint(0, 256, (B, 3, 448, 448), generator=g, dtype=torch.uint8)
    u8[0, :, 0, :256] = torch.arange(256, dtype=torch.uint8)
    f32 = torch.randn(B, 3, 448, 448, generator=g)
    ctr = torch.zeros(4, dtype=torch.int32, device='cuda')
    k = 0
    for pix, want in ((u8, prep.vla_normalize_images(u8[:, None]).to(BF)), (f32, f32.to(BF)), (f32.to(BF), f32.to(BF))):
        for valid in (None, torch.tensor(lens, dtype=torch.int32), torch.tensor([5, 6, 7], dtype=torch.int64)):
            o_ids = torch.zeros(B + 1, T, dtype=torch.int64, device='cuda')
            o_valid = torch.full((B + 1,), -1, dtype=torch.int32, device='cuda')
            o_pro, o_nz = torch.zeros(B + 1, 7, device='cuda'), torch.zeros(16, 7, device='cuda')
            o_pix = torch.zeros(B + 1, 3, 448, 448, dtype=BF, device='cuda')
            k += 1
            ops.vla_stage(ids.cuda(), o_ids, None if valid is None else valid.cuda(), o_valid, pro.cuda(), o_pro, nz.cuda(), o_nz, pix.cuda(), o_pix, pad,
                          prep.VLA_MEAN, prep.VLA_STD, call_ctr=ctr, call_no=k)
            assert torch.equal(o_ids[:B].cpu(), ids) and int(o_ids[B].abs().sum()) == 0
            assert o_valid.cpu().tolist() == (lens if valid is None else valid.tolist()) + [-1]
            assert torch.equal(o_pro[:B].cpu(), pro) and torch.equal(o_nz[:B * 4].cpu(), nz) and float(o_nz[B * 4:].abs().sum()) == 0
            assert torch.equal(o_pix[:B].cpu(), want) and float(o_pix[B].float().abs().sum()) == 0
    assert ctr.tolist() == [9, 0, 0, 0]
    with pytest.raises(ValueError):            # slot capacities are checked by the wrapper (the C side never sees them)
        ops.vla_stage(ids.cuda(), o_ids, None, o_valid, pro.cuda(), torch.zeros(2, 7, device='cuda'), nz.cuda(), o_nz, f32.cuda(), o_pix, pad, prep.VLA_MEAN, prep.VLA_STD)


def test_vla_stage_reference_tensors(ops):
    """r05 (ABI 6): the reference's dense masks and int64 position ids (eval.py:110-128) consumed by the staging launch on the device: valid_len = zero count of
    the proprio row, positions converted into the int32 slots, masks of the supported pattern leave the error word 0 (fp32 / bf16 / fp16, dense or sliced out of
    the full mask), and each kind of unsupported mask sets its bit -- in THIS call's word only."""
    from vlaser_amd import prep
    g = torch.Generator().manual_seed(6)
    B, T, na, pad = 2, 384, 4, 151643
    lens = [277, 31]
    ids = torch.randint(0, 151643, (B, T), generator=g)
    am = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate(lens):
        ids[b, n:] = pad
        am[b, :n] = 1
    pro, nz = torch.rand(B, 7, generator=g).cuda(), torch.randn(B * na, 7, generator=g).cuda()
    pix = torch.randn(B, 3, 448, 448, generator=g).to(BF).cuda()
    ctr = torch.zeros(4, dtype=torch.int32, device='cuda')
    slots = lambda: (torch.zeros(B, T, dtype=torch.int64, device='cuda'), torch.full((B,), -1, dtype=torch.int32, device='cuda'), torch.zeros(B, 7, device='cuda'),
                     torch.zeros(16, 7, device='cuda'), torch.zeros(B, 3, 448, 448, dtype=BF, device='cuda'))
    k = 0

    def stage(m1, m2, valid=None, positions=None, pos_out=None):
        nonlocal k
        k += 1
        o_ids, o_valid, o_pro, o_nz, o_pix = slots()
        ops.vla_stage(ids.cuda(), o_ids, valid, o_valid, pro, o_pro, nz, o_nz, pix, o_pix, pad, prep.VLA_MEAN, prep.VLA_STD, call_ctr=ctr, call_no=k,
                      masks=(m1, m2), n_act=na, positions=positions, pos_out=pos_out)
        c = ctr.tolist()
        assert c[0] == k and c[1 + ((k + 1) & 1)] == 0
        return o_valid.tolist(), c[1 + (k & 1)]

    for dt in (torch.float32, torch.bfloat16, torch.float16):
        full, vp, pp, ap = prep.build_causal_mask_and_position_ids(am, dt, T, 1, na)
        m1, m2 = prep.split_full_mask_into_submasks(full, T, 1, na)
        full_d = full.cuda()
        m1v, m2v = prep.split_full_mask_into_submasks(full_d, T, 1, na)           # device-side slices: strided views, no copy
        assert not m1v.is_contiguous()
        for a, b_ in ((m1.contiguous().cuda(), m2.contiguous().cuda()), (m1v, m2v)):
            assert stage(a, b_) == (lens, 0)
            assert stage(a, b_, valid=torch.tensor(lens, dtype=torch.int32, device='cuda')) == (lens, 0)
            assert stage(a, None) == (lens, 0) and stage(None, b_) == (lens, 0)      # one mask alone (valid_len from the ids when the proprio row is absent)
        # positions: int64 -> int32 slots (+ the batch-1 [proprio | action] row is refused for B = 2)
        pv_, pp_, pa_ = (torch.zeros(B, T, dtype=torch.int32, device='cuda'), torch.zeros(B, dtype=torch.int32, device='cuda'), torch.zeros(B * na, dtype=torch.int32, device='cuda'))
        assert stage(m1v, m2v, positions=(vp.cuda() + 3, pp.cuda() + 5, ap.cuda() + 7), pos_out=(pv_, pp_, pa_, None)) == (lens, 0)
        assert torch.equal(pv_.cpu().long(), vp + 3) and torch.equal(pp_.cpu().long().view(B, 1), pp + 5) and torch.equal(pa_.cpu().long().view(B, na), ap + 7)
        lo = torch.finfo(dt).min
        # (1) a hole in the valid prefix of the proprio row; (2) a prefix row that sees a padded key; (2) the proprio row blind to itself; (4) an action row that
        # does not see an action token; (4) an action row that sees a padded key; a "don't care" row (padded position) may hold anything
        cases = [(1, lambda a, b_: a[1, 0, T].__setitem__(7, lo)), (2, lambda a, b_: a[0, 0, 5].__setitem__(300, 0.0)), (2, lambda a, b_: a[0, 0, T].__setitem__(T, lo)),
                 (4, lambda a, b_: b_[1, 0, 2].__setitem__(T + 2, lo)), (4, lambda a, b_: b_[0, 0, 0].__setitem__(350, 0.0)), (0, lambda a, b_: a[1, 0, 200, :].fill_(0.0))]
        for bit, poke in cases:
            a, b_ = m1.contiguous().clone(), m2.contiguous().clone()
            poke(a, b_)
            v, word = stage(a.cuda(), b_.cuda(), valid=torch.tensor(lens, dtype=torch.int64, device='cuda'))
            assert word == bit, (dt, bit, word)
        a = m1.contiguous().clone(); a[1, 0, T, 7] = lo                       # without a given valid_len the same hole shows as a non-contiguous prefix
        v, word = stage(a.cuda(), m2.contiguous().cuda())
        assert v == [lens[0], lens[1] - 1] and word & 1


def test_infer_action_reference_signature(golden_model):
    """The reference's 8-tensor call (pizero_internvl.py:798-808, built at eval.py:110-128) served without a host round trip: == the valid_len extension bit for
    bit; a non-block `action_mask` and a non-prefix `image_text_proprio_mask` each give a NaN result and raise at the next poll (r04 ignored action_mask)."""
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    g = torch.Generator().manual_seed(13)
    ids = torch.full((1, 384), cfg.pad_token_id)
    ids[0, :10] = torch.randint(0, 151643, (10,), generator=g)
    ids[0, 10:266] = cfg.img_context_token_id
    ids[0, 266:277] = torch.randint(0, 151643, (11,), generator=g)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    pro, noise = torch.rand(1, 1, 7, generator=g) * 2 - 1, torch.randn(1, 4, 7, generator=g)
    m = PiZeroInference(vla, max_batch=1); m.load_state_dict(sd)
    assert m.output_ring == 0
    want = m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=torch.tensor([277]))
    for dt in (torch.float32, torch.bfloat16):
        mask, vp, pp, ap = m.build_causal_mask_and_position_ids((ids != cfg.pad_token_id).long(), dt)
        m1, m2 = m.split_full_mask_into_submasks(mask)
        dev = lambda t: t.to('cuda')                                      # eval.py:130: every tensor moved to the device
        got = m.infer_action(dev(ids), dev(pv), dev(m1), dev(m2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
        assert got.data_ptr() != want.data_ptr() and torch.equal(got.cpu(), want.cpu())     # a fresh tensor per call, as the reference returns
        m.check_errors()
        bad2 = m2.clone(); bad2[0, 0, 1, 384 + 2] = torch.finfo(dt).min       # action token 1 blind to action token 1
        out = m.infer_action(dev(ids), dev(pv), dev(m1), dev(bad2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
        assert torch.isnan(out).all()
        with pytest.raises(ValueError, match='action_mask'):
            m.check_errors()
        bad1 = m1.clone(); bad1[0, 0, 384, 100] = torch.finfo(dt).min        # a hole in the prefix the proprio row sees
        out = m.infer_action(dev(ids), dev(pv), dev(bad1), dev(m2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
        with pytest.raises(ValueError, match='image_text_proprio_mask'):     # ... also raised lazily, by the NEXT call
            torch.cuda.synchronize()
            m.infer_action(dev(ids), dev(pv), dev(m1), dev(m2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
        assert torch.isnan(out).all()
        m.check_errors()                                                    # the call that raised was not run; nothing outstanding
        again = m.infer_action(dev(ids), dev(pv), dev(m1), dev(m2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
        assert torch.equal(again.cpu(), want.cpu())
    # custom positions reach the kernels (different result), the defaults come back when none are passed
    other = m.infer_action(dev(ids), dev(pv), dev(m1), dev(m2), dev(vp) + 2, dev(pp), dev(ap) + 1, dev(pro), noise=dev(noise))
    assert not torch.equal(other.cpu(), want.cpu())
    assert torch.equal(m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=torch.tensor([277])).cpu(), want.cpu())
    with pytest.raises(ValueError):
        m.infer_action(ids, pv, proprios=torch.zeros(1, 1, 6), noise=noise)
    with pytest.raises(ValueError):
        m.infer_action(ids, pv, proprios=pro, noise=torch.zeros(1, 5, 7))


def test_infer_action_output_ring(golden_model):
    """output_ring = n (opt-in): infer_action returns a view of a result ring written by the chunk's last kernel (no copy launch): values == the default
    fresh-tensor path (output_ring = 0, what the reference returns), a result stays intact for n - 1 further calls, and device-resident inputs (one staging
    launch) == host inputs."""
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    g = torch.Generator().manual_seed(12)
    ids = torch.full((1, 384), cfg.pad_token_id)
    ids[0, :10] = torch.randint(0, 151643, (10,), generator=g)
    ids[0, 10:266] = cfg.img_context_token_id
    ids[0, 266:277] = torch.randint(0, 151643, (11,), generator=g)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    pro = torch.rand(1, 1, 7, generator=g) * 2 - 1
    noises = [torch.randn(1, 4, 7, generator=g) for _ in range(5)]
    m0 = PiZeroInference(vla, max_batch=1, output_ring=0); m0.load_state_dict(sd)
    want = [m0.infer_action(ids, pv, proprios=pro, noise=n).cpu() for n in noises]
    m = PiZeroInference(vla, max_batch=1, output_ring=4); m.load_state_dict(sd)
    got = [m.infer_action(ids.cuda(), pv.cuda(), proprios=pro.cuda(), noise=n.cuda(), valid_len=torch.tensor([277], device='cuda')) for n in noises]
    torch.cuda.synchronize()
    for i in range(1, 5):                                   # the last four results are all still there
        assert torch.equal(got[i].cpu(), want[i]), i
    assert got[0].data_ptr() == got[4].data_ptr()           # ... and the fifth call reused the first one's slot
    assert not torch.equal(want[0], want[4])


def _vla_inputs(cfg, seed, B=1):
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((B, 384), cfg.pad_token_id)
    ids[:, :10] = torch.randint(0, 151643, (B, 10), generator=g)
    ids[:, 10:266] = cfg.img_context_token_id
    ids[:, 266:277] = torch.randint(0, 151643, (B, 11), generator=g)
    pv = torch.randn(B, 3, 448, 448, generator=g)
    return ids, pv, torch.rand(B, 1, 7, generator=g) * 2 - 1, torch.randn(B, 4, 7, generator=g)


def test_infer_action_failed_first_call_keeps_position_state(golden_model):
    """a call that raises between `_positions_for_stage` and the staging launch (here: a wrong proprio width on the FIRST call of a batch
    size) must not leave the model believing its position-id slots are written -- the retry has to stage them, i.e. equal a fresh model's result."""
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    ids, pv, pro, noise = _vla_inputs(cfg, 21)
    vl = torch.tensor([277])
    m0 = PiZeroInference(vla, max_batch=1); m0.load_state_dict(sd)
    want = m0.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=vl).cpu()
    m = PiZeroInference(vla, max_batch=1); m.load_state_dict(sd)
    with pytest.raises(ValueError):
        m.infer_action(ids, pv, proprios=torch.zeros(1, 1, 6), noise=noise, valid_len=vl)
    assert m._pos_state is None                       # nothing was staged
    assert torch.equal(m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=vl).cpu(), want)
    assert m._pos_state == ('default', 1)
    # the same with custom positions in the failing call: the retry without any must bring the defaults back
    m = PiZeroInference(vla, max_batch=1); m.load_state_dict(sd)
    _, vp, pp, ap = m.build_causal_mask_and_position_ids((ids != cfg.pad_token_id).long(), torch.float32)
    assert torch.equal(m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=vl).cpu(), want)
    with pytest.raises(ValueError):
        m.infer_action(ids, pv, vlm_position_ids=vp + 2, proprio_position_ids=pp, action_position_ids=ap, proprios=torch.zeros(1, 1, 6), noise=noise, valid_len=vl)
    assert m._pos_state == ('default', 1)             # the slots still hold the defaults: the custom ids never reached the device
    assert torch.equal(m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=vl).cpu(), want)


def test_infer_action_groups_with_output_ring(golden_model):
    """B > max_batch runs as groups; with output_ring > 0 each group's result is a view of a ring slot, and with more groups than slots a
    later group overwrote an earlier one's slot before the final `cat` -- every row must equal its own single-observation call."""
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    B = 6                                             # 6 groups of 1 on a 4-slot ring
    ids, pv, pro, noise = _vla_inputs(cfg, 22, B)
    vl = torch.full((B,), 277)
    m0 = PiZeroInference(vla, max_batch=1); m0.load_state_dict(sd)
    want = torch.cat([m0.infer_action(ids[i:i + 1], pv[i:i + 1], proprios=pro[i:i + 1], noise=noise[i:i + 1], valid_len=vl[i:i + 1]).cpu() for i in range(B)])
    assert len({want[i].numpy().tobytes() for i in range(B)}) == B
    m = PiZeroInference(vla, max_batch=1, output_ring=4); m.load_state_dict(sd)
    got = m.infer_action(ids, pv, proprios=pro, noise=noise, valid_len=vl)
    assert torch.equal(got.cpu(), want)


def test_infer_action_one_euler_step(golden_model):
    """num_inference_steps == 1 at batch 1 -- the step that hosts the proprio row would also be the last one and never write the result ring;
    the proprio row takes its own pass instead, and the result equals the explicit ride_proprio=False model's bit for bit (and is not stale / zero)."""
    import dataclasses
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    vla1 = dataclasses.replace(vla, num_inference_steps=1)
    ids, pv, pro, noise = _vla_inputs(cfg, 23)
    vl = torch.tensor([277])
    a = PiZeroInference(vla1, max_batch=1); a.load_state_dict(sd)
    b = PiZeroInference(vla1, max_batch=1, ride_proprio=False); b.load_state_dict(sd)
    ra = [a.infer_action(ids, pv, proprios=pro, noise=s * noise, valid_len=vl).cpu() for s in (1.0, -0.5)]
    rb = [b.infer_action(ids, pv, proprios=pro, noise=s * noise, valid_len=vl).cpu() for s in (1.0, -0.5)]
    assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1])
    assert not torch.equal(ra[0], ra[1]) and ra[0].abs().max() > 0


def test_infer_action_mask_error_surfaces_at_host_transfer(golden_model):
    """VERDICT r05 weak #9 / next #6d: the chunk of a call with an unsupported dense mask is NaN and the `ValueError` is raised by the first host transfer of
    that chunk (`.cpu()` of the chunk or of anything derived from it, `.tolist()`), by `last_velocities()`, and by the next public method -- not one call late."""
    from vlaser_amd.pizero import PiZeroInference
    cfg, vla, sd = golden_model
    ids, pv, pro, noise = _vla_inputs(cfg, 24)
    m = PiZeroInference(vla, max_batch=1); m.load_state_dict(sd)
    mask, vp, pp, ap = m.build_causal_mask_and_position_ids((ids != cfg.pad_token_id).long(), torch.float32)
    m1, m2 = m.split_full_mask_into_submasks(mask)
    dev = lambda t: t.to('cuda')
    good = m.infer_action(dev(ids), dev(pv), dev(m1), dev(m2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
    host = good[0].float().cpu().numpy()              # eval.py:139, on a supported mask: a plain array
    assert host.shape == (4, 7) and type(good.cpu()) is torch.Tensor
    bad2 = m2.clone(); bad2[0, 0, 1, 384 + 2] = torch.finfo(torch.float32).min
    call = lambda: m.infer_action(dev(ids), dev(pv), dev(m1), dev(bad2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
    out = call()
    with pytest.raises(ValueError, match='check_errors'):
        out[0].float().cpu()
    m.check_errors()                                  # reported once
    out = call()
    with pytest.raises(ValueError, match='action_mask'):
        out.tolist()
    out = call()
    with pytest.raises(ValueError, match='action_mask'):
        m.last_velocities()
    out = call()
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match='action_mask'):
        m.infer_text(ids[:, :277], pv)
    assert torch.isnan(out).all()
    again = m.infer_action(dev(ids), dev(pv), dev(m1), dev(m2), dev(vp), dev(pp), dev(ap), dev(pro), noise=dev(noise))
    assert torch.equal(again.cpu(), good.cpu())


def test_avg_update_ema_swa(ops):
    """EMA / SWA kernel vs torch.optim.swa_utils semantics (model_averaging.py:8-72): first update copies, then lerp / running mean."""
    from vlaser_amd.vla_train import ModelAveraging
    from types import SimpleNamespace
    n = 100003
    ps = [torch.randn(n, generator=torch.Generator().manual_seed(s)).cuda() for s in range(4)]
    for kind in ('ema', 'swa'):
        tr = SimpleNamespace(master=ps[0].clone())
        ma = ModelAveraging(tr, use_ema=kind == 'ema', use_swa=kind == 'swa', ema_start=2, swa_start=2, ema_decay=0.9)
        ref, nav = None, 0
        for step in range(1, 6):
            tr.master.copy_(ps[step % 4])
            ma.maybe_initialize(step); ma.maybe_update(step)
            if step >= 2:
                p = ps[step % 4].double()
                ref = p.clone() if nav == 0 else (ref + (p - ref) * (0.1 if kind == 'ema' else 1.0 / (nav + 1)))
                nav += 1
                assert (ma.avg.double() - ref).abs().max().item() < 1e-6
            else:
                assert ma.avg is None and ma.state_dict() == {}
        assert ma.n_averaged == 4


@pytest.mark.parametrize('M,W,ad,npart,row_off', [(4, 768, 7, 7, 0), (4, 768, 7, 7, 1), (1, 256, 7, 0, 0), (5, 512, 14, 3, 0), (16, 1024, 7, 8, 0), (9, 768, 16, 5, 2)])
def test_vla_step_one_launch_between_layer_passes(ops, M, W, ad, npart, row_off):
    """vlaser_vla_step against the same arithmetic in torch fp32 with the kernel's rounding points: tail of an Euler step (split-K reduce + residual ->
    RMSNorm -> action decoder -> a += dt v) on rows row_off.. of the layer output, then the action encoder with linear_1 / the time embedding folded
    into linear_2 (ops.fold_action_encoder), swish, linear_3.  Every row-count / width / action-dim branch of the kernel templates; and the folded
    encoder against the un-folded reference order (modules.py:25-56) within bf16 noise."""
    torch.manual_seed(M * W + ad)
    dev = 'cuda'
    n_steps, mp, dt, eps = 10, 10000.0, 0.1, 1e-6
    w1, b1, w2, b2, w3, b3 = rnd(W, ad, std=0.3), rnd(W, std=0.1), rnd(W, 2 * W), rnd(W, std=0.1), rnd(W, W), rnd(W, std=0.1)
    wd, bd = rnd(ad, W, std=0.05), rnd(ad, std=0.1)
    nw = (1.0 + 0.1 * torch.randn(W, device=dev)).to(BF)
    rows_in = M + row_off
    h = rnd(rows_in, W, std=1.0)
    parts = (torch.randn(max(npart, 1), rows_in, W, device=dev) * 0.2).contiguous()
    a_in = torch.randn(16, ad, device=dev)
    a_out = torch.full((16, ad), 9.0, device=dev); vel = torch.full((16, ad), 9.0, device=dev)
    h_out = torch.zeros(16, W, dtype=BF, device=dev)
    w21, cs = ops.fold_action_encoder(w1, b1, w2, b2, W, ad, n_steps, mp)
    s_idx = 3
    ops.vla_step(a_in, a_out, w21, cs[s_idx], w3, b3, h_out, M, W, ad, finish=(h, parts if npart else None, npart, rows_in, row_off, nw, eps, wd, bd), vel_out=vel, dt=dt)
    torch.cuda.synchronize()
    r = lambda x: x.to(BF).float()
    hs = r(h.float()[row_off:] + (parts[:npart, row_off:].sum(0) if npart else 0.0))
    y = r(r(hs * torch.rsqrt((hs * hs).mean(-1, keepdim=True) + eps)) * nw.float())
    v_ref = r(y @ wd.float().t() + bd.float())
    a_ref = a_in[:M] + dt * v_ref
    assert (vel[:M] - v_ref).abs().max().item() <= 2e-2 * max(1.0, v_ref.abs().max().item())
    assert (a_out[:M] - a_ref).abs().max().item() <= 2e-3 * max(1.0, a_ref.abs().max().item())
    assert (a_out[M:] == 9.0).all() and (vel[M:] == 9.0).all()
    # encoder on the kernel's own a_out (so the two halves are checked separately)
    a_b = r(a_out[:M])
    e2 = r(torch.nn.functional.silu(r(a_b @ w21.t() + cs[s_idx][None])))
    h_ref = r(e2 @ w3.float().t() + b3.float())
    close(h_out[:M], h_ref, name='linear_3 of the folded encoder')
    assert (h_out[M:] == 0).all()
    # un-folded reference order with its bf16 rounding of linear_1's output and of the time embedding
    half = W // 2
    freq = torch.exp(-math.log(mp) / (half - 1) * torch.arange(half, device=dev, dtype=torch.float32))
    ang = (s_idx / n_steps) * freq
    temb = r(torch.cat([ang.sin(), ang.cos()]))
    e1 = r(a_b @ w1.float().t() + b1.float())
    pre = r(torch.cat([temb[None].expand(M, -1), e1], -1) @ w2.float().t() + b2.float())
    h_unf = r(r(torch.nn.functional.silu(pre)) @ w3.float().t() + b3.float())
    close(h_out[:M], h_unf, rtol=3e-2, name='folded vs un-folded encoder')
    # encoder only (first Euler step): a_out may alias a_in
    h2 = torch.zeros(16, W, dtype=BF, device=dev)
    ops.vla_step(a_out, a_out, w21, cs[s_idx], w3, b3, h2, M, W, ad)
    assert torch.equal(h2[:M], h_out[:M])



@pytest.mark.parametrize('case', ['nt_bias_gelu_144x128', 'nt_res_64x64', 'nn_none_128x256', 'nt_swiglu_aux_192x256'])
def test_gemm_row_store_race_screen(ops, case):
    """r04: the bf16 epilogues reuse the stage ring as scratch for whole-row stores (after a barrier that follows every wave's own vmcnt(0)) / trade fragments between lanes:
    25 runs on the same operands, an unrelated GEMM stream beside every other one, must be bit-identical -- a store that met a late LDS-DMA piece or a stale scratch row
    would differ between runs."""
    from vlaser_amd import _lib as L
    g = torch.Generator().manual_seed(len(case))
    rn = lambda *sh, sc=0.05: (torch.randn(*sh, generator=g) * sc).to(BF).cuda()
    if case == 'nt_bias_gelu_144x128':
        x, w, b = rn(1025, 1024, sc=1.0), rn(4096, 1024), rn(4096)
        run = lambda out: ops.gemm(L.EPI_BIAS_GELU, x, w, out=out, bias=b)
        shape = (1025, 4096)
    elif case == 'nt_res_64x64':
        x, w, r = rn(384, 1536, sc=1.0), rn(1536, 1536), rn(384, 1536, sc=1.0)
        run = lambda out: ops.gemm(L.EPI_RES, x, w, out=out, res=r, force_bm=1564)
        shape = (384, 1536)
    elif case == 'nn_none_128x256':
        x, w = rn(560, 1536, sc=1.0), rn(1536, 8960)
        run = lambda out: ops.gemm_nn(L.EPI_NONE, x, w, out=out)
        shape = (560, 8960)
    else:
        x, w = rn(560, 1536, sc=1.0), ops.pack_gate_up(rn(8960, 1536), rn(8960, 1536))
        aux = torch.zeros(560, 17920, dtype=BF, device='cuda')
        run = lambda out: ops.gemm(L.EPI_SWIGLU, x, w, out=out, aux_out=aux, ld_aux=aux.stride(0))
        shape = (560, 8960)
    side = torch.cuda.Stream()
    a2, w2, o2 = rn(1024, 1024, sc=1.0), rn(4096, 1024), torch.empty(1024, 4096, dtype=BF, device='cuda')
    first = None
    for it in range(25):
        out = torch.full(shape, 3.0, dtype=BF, device='cuda')
        if it % 2:
            with torch.cuda.stream(side):
                for _ in range(1 + it % 4):
                    ops.gemm(L.EPI_NONE, a2, w2, out=o2)
        run(out)
        torch.cuda.synchronize()
        keep = (out.clone(), aux.clone()) if case.startswith('nt_swiglu') else (out.clone(),)
        if first is None:
            first = keep
            assert torch.isfinite(out.float()).all() and not (out == 3.0).all()
        else:
            assert all(torch.equal(a, b) for a, b in zip(keep, first)), it
