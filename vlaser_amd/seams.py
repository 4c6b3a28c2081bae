"""Operator-level seams of the reference, bound to libvlaser_hip.so -- the stubs a maintainer would paste
into the reference's own modules, kept here in executable form so that tests/test_seams_gpu.py can run each one behind the
reference's signature against plain torch math.  Layout conversion is all these adapters do; the arithmetic is the C ABI's.

  seam 1  NORM2FN registry                      modeling_intern_vit.py:127-130      HipLayerNorm / HipRMSNorm
  seam 2  ViT attention core FlashAttention     modeling_intern_vit.py:51-96        flash_attention_forward
  seam 3  HF attention interface                joint_model.py:636-656              vlaser_attention_forward
  seam 4  nn.Linear call sites                  modeling_intern_vit.py:196,208,...  HipLinear
"""
import torch
from torch import nn

from . import _lib as L
from . import ops

BF = torch.bfloat16


class HipLayerNorm(nn.LayerNorm):
    """NORM2FN['layer_norm'] = HipLayerNorm."""

    def forward(self, x):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        return ops.layernorm(x2, self.weight, self.bias, self.eps).view(x.shape)


class HipRMSNorm(nn.Module):
    """Replaces Qwen2RMSNorm / InternRMSNorm (modeling_intern_vit.py:99-110)."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        return ops.rmsnorm(x2, self.weight, self.variance_epsilon).view(x.shape)


class HipLinear(nn.Linear):
    def forward(self, x):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        out = ops.linear(x2, self.weight, self.bias)
        return out.view(*x.shape[:-1], self.out_features)


def flash_attention_forward(qkv, key_padding_mask=None, causal=False, cu_seqlens=None, max_s=None, need_weights=False, softmax_scale=None):
    """FlashAttention.forward(qkv[B,S,3,H,D]) -> (out[B,S,H,D], None); same asserts as the reference (:60-62)."""
    assert not need_weights
    assert qkv.dtype in (torch.float16, torch.bfloat16) and qkv.is_cuda
    if key_padding_mask is not None or cu_seqlens is not None:
        raise NotImplementedError('the ViT path passes neither a padding mask nor cu_seqlens (modeling_intern_vit.py:232-236)')
    qkv = qkv.to(BF)
    B, S, _, H, D = qkv.shape
    Sp = (S + 63) // 64 * 64
    scale = D ** -0.5 if softmax_scale is None else softmax_scale
    q = (qkv[:, :, 0].permute(0, 2, 1, 3).float() * scale).to(BF).contiguous()                     # [B,H,S,D], pre-scaled like VL_EPI_VIT_QKV
    k = torch.zeros(B, H, Sp, D, dtype=BF, device=qkv.device); k[:, :, :S] = qkv[:, :, 1].permute(0, 2, 1, 3)
    vt = torch.zeros(B, H, D, Sp, dtype=BF, device=qkv.device); vt[..., :S] = qkv[:, :, 2].permute(0, 2, 3, 1)   # V transposed, zero padded
    out = torch.empty(B, S, H * D, dtype=BF, device=qkv.device)
    ops.attn_prefill(q, k, vt, out, B, S, S, H, H, D, (H * S * D, S * D, D), (H * Sp * D, Sp * D), (H * D * Sp, D * Sp), (S * H * D, H * D), Sp, 1.0,
                     L.ATTN_CAUSAL if causal else L.ATTN_FULL)
    return out.view(B, S, H, D), None


def _describe_mask(attention_mask, Sq, Skv):
    """additive [B,1,Sq,Skv] mask -> (mode, valid_len[B] or None, blk_start): None -> FULL, lower-triangular -> CAUSAL, the VLA block
    mask (pizero_internvl.py:517-587: every row sees a valid prefix; rows of the trailing block also see that block) -> PREFIX."""
    if attention_mask is None:
        return L.ATTN_FULL, None, 0
    m = attention_mask[:, 0].float().cpu()
    if not bool(((m == 0) | (m <= -1e9)).all()):
        raise NotImplementedError('attention_mask must be additive with entries 0 / dtype-min (no biases)')
    vis = m == 0                                                                                   # [B,Sq,Skv]
    causal = torch.ones(Sq, Skv, dtype=torch.bool).tril(Skv - Sq)
    if bool((vis == causal[None]).all()):
        return L.ATTN_CAUSAL, None, 0
    last = vis[:, -1]                                                                              # a row of the trailing block
    valid = last.int().cumprod(-1).sum(-1)                                                         # leading visible keys
    tail = last.flip(-1).int().cumprod(-1).sum(-1)                                                 # trailing visible keys
    blk = Skv - int(tail[0])
    idx = torch.arange(Skv)[None]
    rows = torch.arange(Skv - Sq, Skv)[None, :, None]                                              # global row index of each query row
    rebuilt = (idx[:, None] < valid[:, None, None]) | ((rows >= blk) & (idx[:, None] >= blk))
    # padded prefix rows (valid_len <= row < blk_start) are "don't care": the reference masks everything for them (uniform softmax
    # over an all-min row) and nobody attends to them; the kernels let them see the valid prefix
    care = (rows < valid[:, None, None]) | (rows >= blk)
    if bool((tail == tail[0]).all()) and bool(((vis == rebuilt) | ~care).all()):
        return L.ATTN_PREFIX, valid.to(torch.int32), blk
    raise NotImplementedError('attention_mask is neither causal nor a prefix + trailing-block mask')


def vlaser_attention_forward(module, query, key, value, attention_mask, dropout=0.0, scaling=None, sliding_window=None, **kw):
    """HF attention interface (`ALL_ATTENTION_FUNCTIONS["vlaser_hip"] = vlaser_attention_forward`): query [B,Hq,Sq,128],
    key / value [B,Hkv,Skv,128], additive mask [B,1,Sq,Skv] or None -> (attn_output [B,Sq,Hq,128] contiguous, None)."""
    assert dropout == 0.0 and sliding_window is None
    B, Hq, Sq, D = query.shape
    Hkv, Skv = key.shape[1], key.shape[2]
    scaling = D ** -0.5 if scaling is None else scaling
    dev = query.device
    mode, valid, blk = _describe_mask(attention_mask, Sq, Skv)
    Sp = (Skv + 63) // 64 * 64
    q = query.to(BF).contiguous()
    k = torch.zeros(B, Hkv, Sp, D, dtype=BF, device=dev); k[:, :, :Skv] = key
    vt = torch.zeros(B, Hkv, D, Sp, dtype=BF, device=dev); vt[..., :Skv] = value.transpose(2, 3)
    valid_d = None if valid is None else valid.to(dev)
    if Sq * (Hq // Hkv) > 32 or mode == L.ATTN_CAUSAL:
        out = torch.empty(B, Sq, Hq * D, dtype=BF, device=dev)
        ops.attn_prefill(q, k, vt, out, B, Sq, Skv, Hq, Hkv, D, (Hq * Sq * D, Sq * D, D), (Hkv * Sp * D, Sp * D), (Hkv * D * Sp, D * Sp),
                         (Sq * Hq * D, Hq * D), Sp, scaling, mode, causal_off=Skv - Sq, valid_len=valid_d, blk_start=blk, q_row_off=Skv - Sq)
        return out.view(B, Sq, Hq, D), None
    # <= 16 query tokens: the key-split kernel (one split here; inside the model the o_proj prologue merges the splits)
    parts = ops.attn_partial_buffers(B, Hkv, dev, max_splits=1)      # compact [B, n_kv, n_splits, ...] layout
    ops.attn_skinny(q, k, vt, parts, B, Sq, Skv, Hq, Hkv, D, (Hq * Sq * D, Sq * D, D), (Hkv * Sp * D, Sp * D), (Hkv * D * Sp, D * Sp), Sp, scaling, mode, 1,
                    valid_len=valid_d, blk_start=blk)
    G = Hq // Hkv
    o = parts[2][:, :, 0, :G * Sq] / parts[1][:, :, 0, :G * Sq, None]                              # rows r = hg*Sq + tok of each kv head
    out = o.view(B, Hkv, G, Sq, D).permute(0, 3, 1, 2, 4).reshape(B, Sq, Hq, D).to(BF).contiguous()
    return out, None
