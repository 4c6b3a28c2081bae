"""Deterministic synthetic weights with the reference's checkpoint key names.

There are no Vlaser checkpoints offline, so the bench and the parity tests use random-init weights of the
right architecture.  Values come from an integer counter hash evaluated with torch int64 ops, which gives
bit-identical tensors on CPU and on the GPU (no dependence on either RNG), so the CPU oracle and the HIP
path always see exactly the same weights.

Key names mirror the HF checkpoint of InternVLChatModel (modeling_intern_vit.py:141-152,196-208,256-257,
275-279; modeling_internvl_chat.py:89-94; HF Qwen2) and the VLA additions (pizero_internvl.py:253-262,
294-320).
"""
import zlib

import torch

from .config import VlaserConfig, VLAConfig

_M1 = -7046029254386353131      # 0x9E3779B97F4A7C15 as signed int64
_M2 = -4658895280553007687      # 0xBF58476D1CE4E5B9
_M3 = -7723592293110705685      # 0x94D049BB133111EB


def _hash_uniform(n, seed, device):
    """splitmix64-style hash of arange(n) -> float32 uniform in [-1, 1). int64 ops wrap identically on CPU/GPU."""
    x = torch.arange(n, dtype=torch.int64, device=device) * _M1 + seed
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * _M2
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * _M3
    x = x ^ ((x >> 31) & 0x1FFFFFFFF)
    u = ((x >> 40) & 0xFFFFFF).to(torch.float32)          # 24 random bits
    return u * (2.0 / 16777216.0) - 1.0


def synth_tensor(name, shape, std, device='cpu', dtype=torch.float32, seed=0, mean=0.0):
    n = 1
    for s in shape:
        n *= s
    key = (zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF
    t = _hash_uniform(n, key * 1000003 + 12345, device) * (std * 1.7320508075688772) + mean
    return t.reshape(shape).to(dtype)


def _linear(sd, name, out_f, in_f, bias, std, **kw):
    sd[name + '.weight'] = synth_tensor(name + '.weight', (out_f, in_f), std, **kw)
    if bias:
        sd[name + '.bias'] = synth_tensor(name + '.bias', (out_f,), std, **kw)


def _norm(sd, name, dim, bias, **kw):
    sd[name + '.weight'] = synth_tensor(name + '.weight', (dim,), 0.05, mean=1.0, **kw)
    if bias:
        sd[name + '.bias'] = synth_tensor(name + '.bias', (dim,), 0.02, **kw)


def qwen2_state_dict(prefix, llm, with_embed=True, with_head=True, std=0.02, **kw):
    sd = {}
    H, hd = llm.hidden_size, llm.head_dim
    if with_embed:
        sd[prefix + 'model.embed_tokens.weight'] = synth_tensor(prefix + 'embed', (llm.vocab_size, H), std, **kw)
    for i in range(llm.num_hidden_layers):
        p = f'{prefix}model.layers.{i}.'
        _linear(sd, p + 'self_attn.q_proj', llm.num_attention_heads * hd, H, True, std, **kw)
        _linear(sd, p + 'self_attn.k_proj', llm.num_key_value_heads * hd, H, True, std, **kw)
        _linear(sd, p + 'self_attn.v_proj', llm.num_key_value_heads * hd, H, True, std, **kw)
        _linear(sd, p + 'self_attn.o_proj', H, llm.num_attention_heads * hd, False, std, **kw)
        _linear(sd, p + 'mlp.gate_proj', llm.intermediate_size, H, False, std, **kw)
        _linear(sd, p + 'mlp.up_proj', llm.intermediate_size, H, False, std, **kw)
        _linear(sd, p + 'mlp.down_proj', H, llm.intermediate_size, False, std, **kw)
        _norm(sd, p + 'input_layernorm', H, False, **kw)
        _norm(sd, p + 'post_attention_layernorm', H, False, **kw)
    _norm(sd, prefix + 'model.norm', H, False, **kw)
    if with_head:
        sd[prefix + 'lm_head.weight'] = synth_tensor(prefix + 'lm_head', (llm.vocab_size, H), std, **kw)
    return sd


def vlm_state_dict(cfg: VlaserConfig, device='cpu', dtype=torch.float32, seed=0, std=0.02):
    """State dict of InternVLChatModel (vision_model.*, mlp1.*, language_model.*)."""
    kw = dict(device=device, dtype=dtype, seed=seed)
    v = cfg.vision
    sd = {}
    e = 'vision_model.embeddings.'
    sd[e + 'class_embedding'] = synth_tensor(e + 'cls', (1, 1, v.hidden_size), std, **kw)
    sd[e + 'patch_embedding.weight'] = synth_tensor(e + 'pe.w', (v.hidden_size, 3, v.patch_size, v.patch_size), std, **kw)
    sd[e + 'patch_embedding.bias'] = synth_tensor(e + 'pe.b', (v.hidden_size,), std, **kw)
    sd[e + 'position_embedding'] = synth_tensor(e + 'pos', (1, v.num_positions, v.hidden_size), std, **kw)
    for i in range(v.num_hidden_layers):
        p = f'vision_model.encoder.layers.{i}.'
        _linear(sd, p + 'attn.qkv', 3 * v.hidden_size, v.hidden_size, True, std, **kw)
        _linear(sd, p + 'attn.proj', v.hidden_size, v.hidden_size, True, std, **kw)
        _linear(sd, p + 'mlp.fc1', v.intermediate_size, v.hidden_size, True, std, **kw)
        _linear(sd, p + 'mlp.fc2', v.hidden_size, v.intermediate_size, True, std, **kw)
        _norm(sd, p + 'norm1', v.hidden_size, True, **kw)
        _norm(sd, p + 'norm2', v.hidden_size, True, **kw)
        sd[p + 'ls1'] = synth_tensor(p + 'ls1', (v.hidden_size,), 0.02, mean=v.initializer_factor, **kw)
        sd[p + 'ls2'] = synth_tensor(p + 'ls2', (v.hidden_size,), 0.02, mean=v.initializer_factor, **kw)
    c4 = v.hidden_size * int(1 / cfg.downsample_ratio) ** 2
    H = cfg.llm.hidden_size
    _norm(sd, 'mlp1.0', c4, True, **kw)
    _linear(sd, 'mlp1.1', H, c4, True, std, **kw)
    _linear(sd, 'mlp1.3', H, H, True, std, **kw)
    sd.update(qwen2_state_dict('language_model.', cfg.llm, std=std, **kw))
    return sd


def vla_state_dict(cfg: VLAConfig, device='cpu', dtype=torch.float32, seed=0, std=0.02, with_head=False):
    """VLA checkpoint `data["model"]` in canonical (de-aliased) form:
    the VLM under its InternVLChatModel names plus
      action_expert.model.layers.* / action_expert.model.norm   (pizero_internvl.py:134,255-262)
      action_encoder.linear_{1,2,3}, proprio_encoder, action_decoder   (pizero_internvl.py:303-320)
    """
    kw = dict(device=device, dtype=dtype, seed=seed)
    sd = vlm_state_dict(cfg.base, std=std, **kw)
    if not with_head:
        sd.pop('language_model.lm_head.weight')
    sd.update(qwen2_state_dict('action_expert.', cfg.expert, with_embed=False, with_head=False, std=std, **kw))
    W = cfg.action_hidden_size
    _linear(sd, 'action_encoder.linear_1', W, cfg.action_dim, True, 0.2, **kw)
    _linear(sd, 'action_encoder.linear_2', W, 2 * W, True, std, **kw)
    _linear(sd, 'action_encoder.linear_3', W, W, True, std, **kw)
    _linear(sd, 'proprio_encoder', W, cfg.proprio_dim, True, 0.2, **kw)
    _linear(sd, 'action_decoder', cfg.action_dim, W, True, 0.01, **kw)
    return sd
