"""InternViT-300M + pixel_shuffle + mlp1 (reference modeling_intern_vit.py:133-431,
modeling_internvl_chat.py:257-291), restated functionally over a state dict with the checkpoint key names."""
import torch
import torch.nn.functional as F


def embeddings(sd, v, pixel_values):
    """InternVisionEmbeddings.forward (modeling_intern_vit.py:162-174).  At image_size == config size the bicubic
    position-embedding resample (:154-160) is the identity, which is the only case on the hot path."""
    p = 'vision_model.embeddings.'
    dt = sd[p + 'patch_embedding.weight'].dtype
    x = F.conv2d(pixel_values.to(dt), sd[p + 'patch_embedding.weight'], sd[p + 'patch_embedding.bias'],
                 stride=v.patch_size)
    B, C, Hh, Ww = x.shape
    assert Hh * Ww == v.num_patches, 'oracle covers the native 448-px grid only'
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + 'class_embedding'].expand(B, 1, -1).to(dt)
    x = torch.cat([cls, x], dim=1)
    return x + sd[p + 'position_embedding'].to(dt)


def attention(sd, p, v, x):
    """InternAttention._naive_attn (modeling_intern_vit.py:210-227): softmax in the activation dtype."""
    B, N, C = x.shape
    Hn = v.num_attention_heads
    qkv = F.linear(x, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'])
    qkv = qkv.reshape(B, N, 3, Hn, C // Hn).permute(2, 0, 3, 1, 4)
    q, k, vv = qkv[0], qkv[1], qkv[2]
    a = (q * (C // Hn) ** -0.5) @ k.transpose(-2, -1)
    a = a.softmax(dim=-1)
    o = (a @ vv).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'])


def encoder_layer(sd, p, v, h):
    """InternVisionEncoderLayer.forward (modeling_intern_vit.py:283-295), DropPath = identity."""
    C = v.hidden_size
    x = F.layer_norm(h, (C,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], v.layer_norm_eps).to(h.dtype)
    h = h + attention(sd, p, v, x) * sd[p + 'ls1']
    x = F.layer_norm(h, (C,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], v.layer_norm_eps).to(h.dtype)
    m = F.linear(F.gelu(F.linear(x, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'])),
                 sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])
    return h + m * sd[p + 'ls2']


def vision_forward(sd, v, pixel_values, return_layers=False):
    h = embeddings(sd, v, pixel_values)
    layers = []
    for i in range(v.num_hidden_layers):
        h = encoder_layer(sd, f'vision_model.encoder.layers.{i}.', v, h)
        if return_layers:
            layers.append(h)
    return (h, layers) if return_layers else h


def pixel_shuffle(x, scale=0.5, ps_version='v2'):
    """modeling_internvl_chat.py:257-271: out[n,i,j,a*2C+b*C+k] = x[n,2i+a,2j+b,k] for v2 (a,b in {0,1})."""
    n, w, h, c = x.shape
    x = x.view(n, w, int(h * scale), int(c / scale))
    x = x.permute(0, 2, 1, 3).contiguous()
    x = x.view(n, int(h * scale), int(w * scale), int(c / (scale * scale)))
    if ps_version != 'v1':
        x = x.permute(0, 2, 1, 3).contiguous()
    return x


def mlp1(sd, x):
    """mlp1 = LayerNorm(4096) -> Linear -> GELU(erf) -> Linear (modeling_internvl_chat.py:89-94)."""
    c4 = x.shape[-1]
    x = F.layer_norm(x, (c4,), sd['mlp1.0.weight'], sd['mlp1.0.bias'], 1e-5)
    x = F.gelu(F.linear(x, sd['mlp1.1.weight'], sd['mlp1.1.bias']))
    return F.linear(x, sd['mlp1.3.weight'], sd['mlp1.3.bias'])


def extract_feature(sd, cfg, pixel_values):
    """InternVLChatModel.extract_feature (modeling_internvl_chat.py:273-291), select_layer == -1."""
    h = vision_forward(sd, cfg.vision, pixel_values)[:, 1:, :]
    g = int(h.shape[1] ** 0.5)
    h = h.reshape(h.shape[0], g, g, -1)
    h = pixel_shuffle(h, cfg.downsample_ratio, cfg.ps_version)
    h = h.reshape(h.shape[0], -1, h.shape[-1])
    return mlp1(sd, h)
