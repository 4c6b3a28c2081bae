"""pi0-style VLA inference on InternVL3 restated (pizero_internvl.py:517-603,706-936; joint_model.py:140-232,
410-696,740-814; modules.py:9-53; kv_cache.py).

State-dict keys: the VLM under its InternVLChatModel names, `action_expert.model.layers.*` / `.norm`
(the `proprio` and `action` mixtures share these weights, pizero_internvl.py:255-262,508-510),
`action_encoder.linear_{1,2,3}`, `proprio_encoder`, `action_decoder`.
"""
import math

import torch
import torch.nn.functional as F

from . import qwen2, vit

LM = 'language_model.'
AE = 'action_expert.'


def build_causal_mask_and_position_ids(attention_mask, dtype, vla):
    """pizero_internvl.py:517-587 (non-debug branch). attention_mask [B,384] of 0/1."""
    bsz, T = attention_mask.shape
    na, npp = vla.num_action_tokens, vla.num_proprio_tokens
    proprio_start = vla.max_image_text_tokens
    proprio_end = proprio_start + npp
    L = T + na + 1
    m = torch.full((bsz, L, L), torch.finfo(dtype).min, dtype=dtype)
    cnts = attention_mask.sum(dim=1)
    for i, c in enumerate(cnts.tolist()):
        m[i, :c, :c] = 0
        m[i, proprio_start:, :c] = 0
    m[:, proprio_start:proprio_end, proprio_start:proprio_end] = 0
    m[:, proprio_end:, proprio_start:] = 0
    m = m.unsqueeze(1)
    vlm_pos = torch.arange(1, vla.max_image_text_tokens + 1).repeat(bsz, 1)
    pro_pos = torch.arange(1, npp + 1).repeat(bsz, 1)
    act_pos = torch.arange(npp + 1, npp + na + 1).repeat(bsz, 1)
    return m, vlm_pos, pro_pos, act_pos


def split_full_mask_into_submasks(mask, vla):
    """pizero_internvl.py:589-603 (4-D branch)."""
    n = vla.max_image_text_tokens + vla.num_proprio_tokens
    return mask[..., :n, :n], mask[..., -vla.num_action_tokens:, :]


def sinusoidal_pos_emb(t, dim, max_period):
    """modules.py:9-22."""
    half = dim // 2
    e = math.log(max_period) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=t.dtype) * -e)
    e = t[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def action_encoder(sd, action, time_emb):
    """modules.py:25-53 with time_cond=True: W3 silu(W2 [time || W1 a])."""
    e = F.linear(action, sd['action_encoder.linear_1.weight'], sd['action_encoder.linear_1.bias'])
    t = time_emb.unsqueeze(1).expand(-1, action.size(1), -1)
    e = torch.cat([t, e], dim=-1)
    e = F.silu(F.linear(e, sd['action_encoder.linear_2.weight'], sd['action_encoder.linear_2.bias']))
    return F.linear(e, sd['action_encoder.linear_3.weight'], sd['action_encoder.linear_3.bias'])


def embed_image_text(sd, vla, input_ids, pixel_values):
    """_forward_siglip_and_text_embedding (pizero_internvl.py:718-796): zeros at pad positions, text embeddings at
    text positions, projected ViT features at <IMG_CONTEXT> positions."""
    cfg = vla.base
    dt = pixel_values.dtype
    emb = F.embedding(input_ids, sd[LM + 'model.embed_tokens.weight'])
    feats = vit.extract_feature(sd, cfg, pixel_values).to(dt)          # [B*n,256,H]
    B, S = input_ids.shape
    out = torch.zeros(B, S, emb.shape[-1], dtype=dt)
    text = (input_ids != cfg.img_context_token_id) & (input_ids != cfg.pad_token_id)
    img = input_ids == cfg.img_context_token_id
    out[text] = emb[text].to(dt)
    out[img] = feats.flatten(0, 1)
    return out


def _mixture_attn_layer(sd, vla, layer, hs, cos_sin, mask, caches, skip):
    """One layer of forward_mixture_layers_internvl over the active mixtures in `hs` (ordered dict name->[B,S,H]).
    caches: name -> list of (k,v) per layer (K post-RoPE).  Non-active cached mixtures contribute K/V
    ("append_non_active", joint_model.py:461-464); active mixtures in `caches` get their K/V appended once."""
    llms = {'vlm': (LM, vla.base.llm), 'proprio': (AE, vla.expert), 'action': (AE, vla.expert)}
    q_all, k_all, v_all = {}, {}, {}
    for name in caches:
        if name not in hs:
            k_all[name], v_all[name] = caches[name][layer]
    normed = {}
    for name, h in hs.items():
        pre, llm = llms[name]
        p = f'{pre}model.layers.{layer}.'
        x = qwen2.rms_norm(h, sd[p + 'input_layernorm.weight'], llm.rms_norm_eps)
        q, k, v = qwen2.qkv_proj(sd, p, x, llm)
        cos, sin = cos_sin[name]
        k = qwen2.apply_rope(k, cos, sin)
        q = qwen2.apply_rope(q, cos, sin)
        if name in caches:
            assert len(caches[name]) == layer
            caches[name].append((k, v))
        q_all[name], k_all[name], v_all[name] = q, k, v
    q = torch.cat(tuple(q_all.values()), dim=-2)
    k = torch.cat(tuple(k_all.values()), dim=-2)
    v = torch.cat(tuple(v_all.values()), dim=-2)
    a = qwen2.eager_attention(q, k, v, mask, vla.base.llm.head_dim ** -0.5)
    outs = torch.split(a, [h.shape[1] for h in hs.values()], dim=1)
    new = {}
    for (name, h), ao in zip(hs.items(), outs):
        if name in skip:
            new[name] = None
            continue
        pre, llm = llms[name]
        p = f'{pre}model.layers.{layer}.'
        h = h + F.linear(ao, sd[p + 'self_attn.o_proj.weight'])
        x = qwen2.rms_norm(h, sd[p + 'post_attention_layernorm.weight'], llm.rms_norm_eps)
        new[name] = h + qwen2.mlp(sd, p, x)
    return new


def joint_forward(sd, vla, hs, pos, mask, caches, final_skip=('vlm', 'proprio'), return_layers=False):
    """JointModel.forward (joint_model.py:740-814), INTERNVL backbone."""
    L = vla.base.llm.num_hidden_layers
    cos_sin = {}
    for name, h in hs.items():
        cos_sin[name] = qwen2.rope_cos_sin(pos[name], vla.base.llm.head_dim, vla.base.llm.rope_theta, h.dtype)
    per_layer = []
    for layer in range(L):
        hs = _mixture_attn_layer(sd, vla, layer, hs, cos_sin, mask, caches,
                                 final_skip if layer == L - 1 else ())
        if return_layers:
            per_layer.append(hs)
    out = {}
    for name, h in hs.items():
        if name not in final_skip:
            pre = LM if name == 'vlm' else AE
            eps = vla.base.llm.rms_norm_eps
            out[name] = qwen2.rms_norm(h, sd[pre + 'model.norm.weight'], eps)
    return (out, per_layer) if return_layers else out


def integration_step(action, delta_t, vel, method='euler'):
    """pizero_internvl.py:910-922 + `integration_step` :1309-1331.  The reference hands integration_step a `model_step(x, tt)` that ignores both arguments and returns the
    decoder output of THIS step's joint pass (:914-917): k1 = k2 = k3 = k4 = vel, so heun and rk4 re-combine one velocity with their own rounding."""
    if method == 'euler':
        return action + delta_t * vel
    if method == 'heun':
        return action + 0.5 * delta_t * (vel + vel)
    if method == 'rk4':
        return action + (delta_t / 6.0) * (vel + 2 * vel + 2 * vel + vel)
    raise ValueError(f'Unknown integration method: {method}')


def infer_action(sd, vla, input_ids, pixel_values, image_text_proprio_mask, action_mask, vlm_position_ids,
                 proprio_position_ids, action_position_ids, proprios, noise, return_trace=False):
    """PiZero.infer_action (pizero_internvl.py:798-936) with the noise as an explicit input."""
    dt = pixel_values.dtype
    bsz = pixel_values.shape[0]
    caches = {'vlm': [], 'proprio': []}
    embeds = embed_image_text(sd, vla, input_ids, pixel_values)
    pro = F.linear(proprios, sd['proprio_encoder.weight'], sd['proprio_encoder.bias'])
    joint_forward(sd, vla, {'vlm': embeds, 'proprio': pro},
                  {'vlm': vlm_position_ids, 'proprio': proprio_position_ids}, image_text_proprio_mask, caches)
    action = noise.clone().to(dt)
    n = vla.num_inference_steps
    dt_step = 1.0 / n
    t = torch.zeros(bsz, dtype=dt)
    trace = []
    for _ in range(n):
        temb = sinusoidal_pos_emb(t, vla.action_hidden_size, vla.time_max_period)
        ae = action_encoder(sd, action, temb)
        out = joint_forward(sd, vla, {'action': ae}, {'action': action_position_ids}, action_mask, caches,
                            final_skip=())['action']
        vel = F.linear(out, sd['action_decoder.weight'], sd['action_decoder.bias'])
        action = integration_step(action, dt_step, vel, getattr(vla, 'integration_method', 'euler'))
        t = t + dt_step
        if return_trace:
            trace.append((action.clone(), vel.clone()))
    if vla.final_action_clip_value is not None:
        action = torch.clamp(action, -vla.final_action_clip_value, vla.final_action_clip_value)
    action = action[:, -vla.horizon_steps:]
    return (action, caches, trace) if return_trace else action


def infer_action_naive(sd, vla, input_ids, pixel_values, causal_mask, vlm_position_ids, proprio_position_ids, action_position_ids,
                       proprios, noise, return_trace=False):
    """PiZero.infer_action_naive (pizero_internvl.py:938-1003): no KV cache -- every Euler step runs ONE joint pass over all three
    mixtures (cache_mode="no_append") under the full [B,1,389,389] block mask.  (The reference's method omits
    `position_embeddings_all` and raises KeyError on the InternVL path; tools/gen_golden.py drives the same JointModel.forward
    calls with that argument supplied -- golden G7b.)"""
    dt = pixel_values.dtype
    bsz = pixel_values.shape[0]
    embeds = embed_image_text(sd, vla, input_ids, pixel_values)
    pro = F.linear(proprios, sd['proprio_encoder.weight'], sd['proprio_encoder.bias'])
    action = noise.clone().to(dt)
    n = vla.num_inference_steps
    t = torch.zeros(bsz, dtype=dt)
    trace = []
    for _ in range(n):
        temb = sinusoidal_pos_emb(t, vla.action_hidden_size, vla.time_max_period)
        ae = action_encoder(sd, action, temb)
        out = joint_forward(sd, vla, {'vlm': embeds.clone(), 'proprio': pro.clone(), 'action': ae},
                            {'vlm': vlm_position_ids, 'proprio': proprio_position_ids, 'action': action_position_ids}, causal_mask, {})['action']
        vel = F.linear(out, sd['action_decoder.weight'], sd['action_decoder.bias'])
        action = action + vel / n
        t = t + 1.0 / n
        trace.append(vel.clone())
    if vla.final_action_clip_value is not None:
        action = torch.clamp(action, -vla.final_action_clip_value, vla.final_action_clip_value)
    return (action, trace) if return_trace else action



def flow_matching_loss(sd, vla, input_ids, pixel_values, causal_mask, vlm_position_ids, proprio_position_ids, action_position_ids, proprios,
                       actions, t, x0):
    """PiZero.forward (pizero_internvl.py:1064-1197), the flow-matching training loss, with x0 as an explicit input:
    psi_t = (1 - (1 - sig_min) t) x0 + t x1 (:1050-1062); ONE joint pass over {vlm, proprio, action}, no cache, last-layer
    post-attention skipped for proprio only; loss = mean((action_decoder(h_action) - (x1 - (1 - sig_min) x0))^2)."""
    sig = vla.flow_sig_min
    tt = t[:, None, None]
    psi = (1 - (1 - sig) * tt) * x0 + tt * actions
    embeds = embed_image_text(sd, vla, input_ids, pixel_values)
    pro = F.linear(proprios, sd['proprio_encoder.weight'], sd['proprio_encoder.bias'])
    temb = sinusoidal_pos_emb(t, vla.action_hidden_size, vla.time_max_period)
    ae = action_encoder(sd, psi, temb)
    out = joint_forward(sd, vla, {'vlm': embeds, 'proprio': pro, 'action': ae},
                        {'vlm': vlm_position_ids, 'proprio': proprio_position_ids, 'action': action_position_ids}, causal_mask, {}, final_skip=('proprio',))['action']
    v = F.linear(out, sd['action_decoder.weight'], sd['action_decoder.bias'])
    return torch.mean((v - (actions - (1 - sig) * x0)) ** 2)
