"""Qwen2.5 decoder arithmetic (HF transformers `modeling_qwen2.py`, un-vendored third-party dependency of the
reference, pinned 4.37.2 / 4.48.0 / 4.54.0; call sites modeling_internvl_chat.py:194-203,431-438 and
joint_model.py:166,209,219,449-452,573-578,631-669,694).  Restated from the published algorithm:

  RMSNorm   : y = w * (x_f32 * rsqrt(mean(x_f32^2) + eps)).to(dtype)
  RoPE      : inv_freq_i = theta^(-2i/d); cos/sin of pos*inv_freq in fp32, cast to dtype; rotate_half convention
  attention : repeat_kv; softmax(q k^T * d^-1/2 + mask) in fp32, cast to dtype; @ v
  MLP       : down(silu(gate(x)) * up(x))
"""

import torch
import torch.nn.functional as F


def rms_norm(x, w, eps):
    dt = x.dtype
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rope_cos_sin(position_ids, head_dim, theta, dtype):
    """position_ids [B,S] int64 -> cos, sin [B,S,head_dim] in `dtype` (computed in fp32 like Qwen2RotaryEmbedding)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x, cos, sin):
    """x [B,heads,S,D]; cos/sin [B,S,D]."""
    return x * cos.unsqueeze(1) + rotate_half(x) * sin.unsqueeze(1)


def repeat_kv(x, n_rep):
    if n_rep == 1:
        return x
    b, h, s, d = x.shape
    return x[:, :, None].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def eager_attention(q, k, v, mask, scaling):
    """q [B,Hq,Sq,D], k/v [B,Hkv,Skv,D], additive mask [B,1,Sq,Skv] or None -> [B,Sq,Hq*D]."""
    n_rep = q.shape[1] // k.shape[1]
    k = repeat_kv(k, n_rep)
    v = repeat_kv(v, n_rep)
    w = torch.matmul(q, k.transpose(2, 3)) * scaling
    if mask is not None:
        w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v)
    return o.transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)


def causal_mask(sq, skv, dtype, attention_mask=None):
    """Additive causal mask [B or 1,1,Sq,Skv]; queries are the last `sq` of `skv` positions.
    attention_mask [B,Skv] (1 = keep) adds key padding."""
    i = torch.arange(sq)[:, None] + (skv - sq)
    j = torch.arange(skv)[None, :]
    keep = (j <= i)[None, None]
    if attention_mask is not None:
        keep = keep & attention_mask[:, None, None, :].bool()
    m = torch.zeros(keep.shape, dtype=dtype)
    return m.masked_fill(~keep, torch.finfo(dtype).min)


def qkv_proj(sd, p, x, llm):
    """x [B,S,H] -> q [B,Hq,S,D], k, v [B,Hkv,S,D] (no RoPE)."""
    B, S, _ = x.shape
    D = llm.head_dim
    q = F.linear(x, sd[p + 'self_attn.q_proj.weight'], sd[p + 'self_attn.q_proj.bias']).view(B, S, -1, D).transpose(1, 2)
    k = F.linear(x, sd[p + 'self_attn.k_proj.weight'], sd[p + 'self_attn.k_proj.bias']).view(B, S, -1, D).transpose(1, 2)
    v = F.linear(x, sd[p + 'self_attn.v_proj.weight'], sd[p + 'self_attn.v_proj.bias']).view(B, S, -1, D).transpose(1, 2)
    return q, k, v


def mlp(sd, p, x):
    g = F.linear(x, sd[p + 'mlp.gate_proj.weight'])
    u = F.linear(x, sd[p + 'mlp.up_proj.weight'])
    return F.linear(F.silu(g) * u, sd[p + 'mlp.down_proj.weight'])


def decoder_layer(sd, p, llm, h, cos, sin, mask, past=None):
    """One Qwen2DecoderLayer. past = (k,v) with RoPE already applied to k; returns (h, (k,v))."""
    x = rms_norm(h, sd[p + 'input_layernorm.weight'], llm.rms_norm_eps)
    q, k, v = qkv_proj(sd, p, x, llm)
    q = apply_rope(q, cos, sin)
    k = apply_rope(k, cos, sin)
    if past is not None:
        k = torch.cat([past[0], k], dim=2)
        v = torch.cat([past[1], v], dim=2)
    a = eager_attention(q, k, v, mask, llm.head_dim ** -0.5)
    h = h + F.linear(a, sd[p + 'self_attn.o_proj.weight'])
    x = rms_norm(h, sd[p + 'post_attention_layernorm.weight'], llm.rms_norm_eps)
    h = h + mlp(sd, p, x)
    return h, (k, v)


def model_forward(sd, prefix, llm, inputs_embeds, position_ids, mask, past=None, return_layers=False):
    """Qwen2Model.forward -> (final-normed hidden [B,S,H], list of (k,v))."""
    h = inputs_embeds
    cos, sin = rope_cos_sin(position_ids, llm.head_dim, llm.rope_theta, h.dtype)
    new_past, per_layer = [], []
    for i in range(llm.num_hidden_layers):
        h, kv = decoder_layer(sd, f'{prefix}model.layers.{i}.', llm, h, cos, sin, mask,
                              None if past is None else past[i])
        new_past.append(kv)
        if return_layers:
            per_layer.append(h)
    h = rms_norm(h, sd[prefix + 'model.norm.weight'], llm.rms_norm_eps)
    if return_layers:
        return h, new_past, per_layer
    return h, new_past


def lm_head(sd, prefix, h):
    return F.linear(h, sd[prefix + 'lm_head.weight'])


def greedy_generate(sd, prefix, llm, inputs_embeds, attention_mask, max_new_tokens, eos_token_id=None,
                    return_logits=False):
    """HF GenerationMixin greedy loop with inputs_embeds + KV cache (modeling_internvl_chat.py:431-438).
    Position ids are cumsum(mask)-1 (0..S-1 when unpadded).  Returns new token ids [B, n] (and last-position
    logits of every step when asked)."""
    B, S, _ = inputs_embeds.shape
    dt = inputs_embeds.dtype
    if attention_mask is None:
        attention_mask = torch.ones(B, S, dtype=torch.long)
    pos = (attention_mask.long().cumsum(-1) - 1).clamp(min=0)
    mask = causal_mask(S, S, dt, attention_mask)
    h, past = model_forward(sd, prefix, llm, inputs_embeds, pos, mask)
    logits = lm_head(sd, prefix, h[:, -1:])
    out, all_logits = [], []
    embed = sd[prefix + 'model.embed_tokens.weight']
    finished = torch.zeros(B, dtype=torch.bool)
    am = attention_mask
    for step in range(max_new_tokens):
        all_logits.append(logits[:, -1].float())
        nxt = logits[:, -1].float().argmax(-1)
        if eos_token_id is not None:
            nxt = torch.where(finished, torch.full_like(nxt, eos_token_id), nxt)
            finished = finished | (nxt == eos_token_id)
        out.append(nxt)
        if step == max_new_tokens - 1 or (eos_token_id is not None and bool(finished.all())):
            break
        am = torch.cat([am, torch.ones(B, 1, dtype=am.dtype)], dim=1)
        pos1 = am.long().sum(-1, keepdim=True) - 1
        m1 = causal_mask(1, am.shape[1], dt, am)
        x = F.embedding(nxt[:, None], embed)
        h, past = model_forward(sd, prefix, llm, x, pos1, m1, past)
        logits = lm_head(sd, prefix, h)
    ids = torch.stack(out, dim=1)
    if return_logits:
        return ids, torch.stack(all_logits, dim=1)
    return ids
