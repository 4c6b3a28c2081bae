"""INTEGRATION.md section 2, executed: every operator-level seam stub (vlaser_amd/seams.py) runs behind the reference's own
signature against plain torch fp32 math of the statement it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rnd(*s, seed=0, std=1.0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * std).to(BF).cuda()


def _close(got, ref, tol=2e-2, name=''):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all(), name
    err = (got - ref).abs()
    assert err.max().item() < tol * max(1.0, ref.abs().max().item()), (name, err.max().item(), ref.abs().max().item(), (err > tol).nonzero()[:6].tolist())


def test_seam1_norm2fn_modules():
    from vlaser_amd.seams import HipLayerNorm, HipRMSNorm
    x = _rnd(3, 1025, 1024)
    ln = HipLayerNorm(1024, eps=1e-6).to(BF).cuda()
    ln.weight.data = 1 + 0.1 * _rnd(1024, seed=1); ln.bias.data = 0.1 * _rnd(1024, seed=2)
    _close(ln(x), F.layer_norm(x.float(), (1024,), ln.weight.float(), ln.bias.float(), 1e-6))
    rn = HipRMSNorm(1024).to(BF).cuda()
    rn.weight.data = 1 + 0.1 * _rnd(1024, seed=3)
    xf = x.float()
    _close(rn(x), (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF).float() * rn.weight.float())


def test_seam2_flash_attention_forward_signature():
    """FlashAttention.forward(qkv[B,S,3,H,D]) -> (out[B,S,H,D], None)  (modeling_intern_vit.py:51-96; ViT shape 1025 x 16 x 64)."""
    from vlaser_amd.seams import flash_attention_forward
    B, S, H, D = 2, 1025, 16, 64
    qkv = _rnd(B, S, 3, H, D, std=0.7)
    out, w = flash_attention_forward(qkv, causal=False)
    assert w is None and out.shape == (B, S, H, D)
    q, k, v = [qkv[:, :, i].permute(0, 2, 1, 3).float() for i in range(3)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).permute(0, 2, 1, 3)
    _close(out, ref)
    with pytest.raises(AssertionError):
        flash_attention_forward(qkv.float())                      # the reference asserts fp16 / bf16 CUDA input (:60-62)


def _eager(q, k, v, mask, scaling):
    G = q.shape[1] // k.shape[1]
    k, v = k.float().repeat_interleave(G, 1), v.float().repeat_interleave(G, 1)
    s = q.float() @ k.transpose(-1, -2) * scaling
    if mask is not None:
        s = s + mask
    return (torch.softmax(s, -1) @ v).transpose(1, 2).contiguous()


def test_seam3_hf_attention_interface_vla_masks():
    """The call of joint_model.py:636-656 with the VLA block masks of pizero_internvl.py:517-603: the joint prefill (385 rows) and
    one Euler step (4 action rows over 389 keys), each against HF's eager_attention_forward math."""
    from vlaser_amd import prep
    from vlaser_amd.seams import vlaser_attention_forward
    B, Hq, Hkv, D = 2, 12, 2, 128
    am = torch.zeros(B, 384, dtype=torch.long); am[0, :277] = 1; am[1, :300] = 1
    mask, _, _, _ = prep.build_causal_mask_and_position_ids(am, torch.float32, 384, 1, 4)
    m1, m2 = prep.split_full_mask_into_submasks(mask, 384, 1, 4)
    q, k, v = _rnd(B, Hq, 385, D, seed=1), _rnd(B, Hkv, 385, D, seed=2), _rnd(B, Hkv, 385, D, seed=3)
    out, w = vlaser_attention_forward(None, q, k, v, m1.cuda(), dropout=0.0, scaling=D ** -0.5)
    assert w is None and out.shape == (B, 385, Hq, D)
    ref = _eager(q, k, v, m1.cuda(), D ** -0.5)
    for b, n in enumerate((277, 300)):                             # padded text rows are "don't care" (nobody attends to them)
        _close(out[b, :n], ref[b, :n], name=f'prefill rows b{b}'); _close(out[b, 384:], ref[b, 384:], name=f'proprio row b{b}')
    qa, ka, va = _rnd(B, Hq, 4, D, seed=4), _rnd(B, Hkv, 389, D, seed=5), _rnd(B, Hkv, 389, D, seed=6)
    out2, _ = vlaser_attention_forward(None, qa, ka, va, m2.cuda(), scaling=D ** -0.5)
    assert out2.shape == (B, 4, Hq, D)
    _close(out2, _eager(qa, ka, va, m2.cuda(), D ** -0.5), name='action rows')
    # plain causal prefill (InternVLChatModel's LLM) and mask-free attention through the same interface
    S = 200
    qc, kc, vc = _rnd(1, Hq, S, D, seed=7), _rnd(1, Hkv, S, D, seed=8), _rnd(1, Hkv, S, D, seed=9)
    cm = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)[None, None].cuda()
    _close(vlaser_attention_forward(None, qc, kc, vc, cm)[0], _eager(qc, kc, vc, cm, D ** -0.5), name='causal')
    _close(vlaser_attention_forward(None, qc, kc, vc, None)[0], _eager(qc, kc, vc, None, D ** -0.5), name='full')
    with pytest.raises(NotImplementedError):
        vlaser_attention_forward(None, qc, kc, vc, torch.randn(1, 1, S, S).cuda())


def test_seam4_linear():
    from vlaser_amd.seams import HipLinear
    lin = HipLinear(1024, 4096).to(BF).cuda()
    lin.weight.data = _rnd(4096, 1024, seed=1, std=0.03); lin.bias.data = _rnd(4096, seed=2, std=0.2)
    x = _rnd(2, 300, 1024)
    _close(lin(x), x.float() @ lin.weight.float().t() + lin.bias.float())
    nb = HipLinear(1024, 512, bias=False).to(BF).cuda()
    nb.weight.data = _rnd(512, 1024, seed=3, std=0.03)
    _close(nb(x), x.float() @ nb.weight.float().t())
