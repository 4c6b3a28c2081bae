"""Model dimensions for the Vlaser hot path.

Values follow the reference's vendored HF config
(Vlaser_VLA/RoboTwin/policy/internvla_2B_parallel_decoding/internvl/pretrained/InternVL3-1B/config.json:
 vision_config is identical for every Vlaser size) and SURVEY.md §8 for the 2B / 8B LLM sizes
(Vlaser_VLA/Simpler/config/eval/bridge_internvl_448.yaml:46,118; pizero_internvl.py:119-131).
"""
from dataclasses import dataclass, field, replace


@dataclass(frozen=True)
class VisionConfig:
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    image_size: int = 448
    patch_size: int = 14
    layer_norm_eps: float = 1e-6
    initializer_factor: float = 0.1          # layer-scale init (modeling_intern_vit.py:278)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def num_patches(self):
        return (self.image_size // self.patch_size) ** 2

    @property
    def num_positions(self):
        return self.num_patches + 1


@dataclass(frozen=True)
class LLMConfig:
    hidden_size: int = 1536
    num_hidden_layers: int = 28
    num_attention_heads: int = 12
    num_key_value_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 8960
    vocab_size: int = 151674
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6


@dataclass(frozen=True)
class VlaserConfig:
    vision: VisionConfig = field(default_factory=VisionConfig)
    llm: LLMConfig = field(default_factory=LLMConfig)
    downsample_ratio: float = 0.5
    ps_version: str = 'v2'
    select_layer: int = -1
    template: str = 'internvl2_5'
    img_context_token_id: int = 151667
    img_start_token_id: int = 151665
    img_end_token_id: int = 151666
    eos_token_id: int = 151645               # <|im_end|>
    pad_token_id: int = 151643               # <|endoftext|>

    @property
    def num_image_token(self):
        # modeling_internvl_chat.py:57
        return int((self.vision.image_size // self.vision.patch_size) ** 2 * (self.downsample_ratio ** 2))


@dataclass(frozen=True)
class VLAConfig:
    """pi0-style head on top of a VlaserConfig (bridge_internvl_448.yaml:32-41,80; pizero_internvl.py:116-131,207-227)."""
    base: VlaserConfig = field(default_factory=VlaserConfig)
    action_hidden_size: int = 768
    action_intermediate_size: int = 8960
    horizon_steps: int = 4
    cond_steps: int = 1
    action_dim: int = 7
    proprio_dim: int = 7
    num_inference_steps: int = 10
    final_action_clip_value: float = 1.0
    max_image_text_tokens: int = 384
    time_max_period: float = 10000.0
    flow_sig_min: float = 0.001
    extra_action_tokens: int = 256           # '<a i>' tokens appended to the vocab (pizero_internvl.py:45-48,85)
    integration_method: str = 'euler'        # pizero_internvl.py:164 (`cfg.get("integration_method", "euler")`); the eval YAML does not set it

    def __post_init__(self):
        # the reference's `integration_step` (pizero_internvl.py:910-922,1309-1331): euler | heun | rk4.  Its `model_step` closure returns THIS step's decoder output
        # whatever (x, t) it is handed, so heun and rk4 re-combine one velocity per step (no extra passes through the expert): served by the same kernels with the
        # method's arithmetic (r06, golden G7c); anything else raises exactly where the reference does
        if self.integration_method not in ('euler', 'heun', 'rk4'):
            raise ValueError(f'Unknown integration method: {self.integration_method}')

    @property
    def expert(self) -> LLMConfig:
        return replace(self.base.llm, hidden_size=self.action_hidden_size,
                       intermediate_size=self.action_intermediate_size)

    @property
    def num_proprio_tokens(self):
        return 1                              # hard-coded in pizero_internvl.py:209

    @property
    def num_action_tokens(self):
        return self.horizon_steps + self.cond_steps - 1


def vlaser_2b(**over) -> VlaserConfig:
    return replace(VlaserConfig(), **over)


def vlaser_8b(**over) -> VlaserConfig:
    llm = LLMConfig(hidden_size=3584, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                    head_dim=128, intermediate_size=18944)
    return replace(VlaserConfig(llm=llm), **over)


def truncated(cfg: VlaserConfig, vit_layers: int, llm_layers: int, vocab_size: int = None) -> VlaserConfig:
    """Depth-truncated (true-width) variant used by parity tests so the CPU oracle finishes in seconds."""
    v = replace(cfg.vision, num_hidden_layers=vit_layers)
    l = replace(cfg.llm, num_hidden_layers=llm_layers,
                vocab_size=cfg.llm.vocab_size if vocab_size is None else vocab_size)
    return replace(cfg, vision=v, llm=l)


def from_hf_config(d: dict) -> VlaserConfig:
    """VlaserConfig from the dict of an InternVL3 / Vlaser HF `config.json` (the format of the reference's vendored
    InternVL3-1B/config.json: top-level template / ps_version / select_layer / downsample_ratio + `llm_config` + `vision_config`).
    Raises ValueError for architectures the kernels do not cover instead of loading something that would run wrong."""
    lc, vc = d.get('llm_config', {}), d.get('vision_config', {})
    arch = (lc.get('architectures') or ['Qwen2ForCausalLM'])[0]
    if arch != 'Qwen2ForCausalLM':
        raise ValueError(f'LLM architecture {arch} is outside the hot path (Qwen2ForCausalLM only; SURVEY.md 8)')
    nh = lc['num_attention_heads']
    head_dim = lc.get('head_dim') or lc['hidden_size'] // nh
    if head_dim != 128:
        raise ValueError(f'head_dim {head_dim}: the attention / RoPE kernels are built for head_dim 128 (Vlaser-2B / 8B)')
    if lc.get('tie_word_embeddings', d.get('tie_word_embeddings', False)):
        raise ValueError('tie_word_embeddings=True is not supported (Vlaser checkpoints carry a separate lm_head)')
    if vc.get('norm_type', 'layer_norm') != 'layer_norm' or vc.get('qk_normalization', False) or not vc.get('qkv_bias', True):
        raise ValueError('vision tower variant outside InternViT-300M-448px (layer_norm, no qk-norm, qkv bias)')
    if vc.get('hidden_act', 'gelu') != 'gelu':
        raise ValueError('vision MLP activation must be gelu')
    rope = lc.get('rope_theta') or (lc.get('rope_parameters') or {}).get('rope_theta') or 1e6      # transformers >= 5 moved it
    llm = LLMConfig(hidden_size=lc['hidden_size'], num_hidden_layers=lc['num_hidden_layers'], num_attention_heads=nh,
                    num_key_value_heads=lc.get('num_key_value_heads', nh), head_dim=head_dim, intermediate_size=lc['intermediate_size'],
                    vocab_size=lc['vocab_size'], rms_norm_eps=lc.get('rms_norm_eps', 1e-6), rope_theta=float(rope))
    img = d.get('force_image_size') or vc.get('image_size', 448)
    vision = VisionConfig(hidden_size=vc.get('hidden_size', 1024), num_hidden_layers=vc.get('num_hidden_layers', 24),
                          num_attention_heads=vc.get('num_attention_heads', 16), intermediate_size=vc.get('intermediate_size', 4096),
                          image_size=img, patch_size=vc.get('patch_size', 14), layer_norm_eps=vc.get('layer_norm_eps', 1e-6),
                          initializer_factor=vc.get('initializer_factor', 0.1))
    return VlaserConfig(vision=vision, llm=llm, downsample_ratio=d.get('downsample_ratio', 0.5), ps_version=d.get('ps_version', 'v2'),
                        select_layer=d.get('select_layer', -1), template=d.get('template', 'internvl2_5'))


def to_hf_config(cfg: VlaserConfig) -> dict:
    """The inverse of from_hf_config (what save_pretrained writes)."""
    v, l = cfg.vision, cfg.llm
    return {'architectures': ['InternVLChatModel'], 'model_type': 'internvl_chat', 'downsample_ratio': cfg.downsample_ratio,
            'ps_version': cfg.ps_version, 'select_layer': cfg.select_layer, 'template': cfg.template, 'force_image_size': v.image_size,
            'tie_word_embeddings': False, 'torch_dtype': 'bfloat16',
            'llm_config': {'architectures': ['Qwen2ForCausalLM'], 'hidden_size': l.hidden_size, 'num_hidden_layers': l.num_hidden_layers,
                           'num_attention_heads': l.num_attention_heads, 'num_key_value_heads': l.num_key_value_heads, 'head_dim': l.head_dim,
                           'intermediate_size': l.intermediate_size, 'vocab_size': l.vocab_size, 'rms_norm_eps': l.rms_norm_eps,
                           'rope_theta': l.rope_theta, 'tie_word_embeddings': False},
            'vision_config': {'hidden_size': v.hidden_size, 'num_hidden_layers': v.num_hidden_layers, 'num_attention_heads': v.num_attention_heads,
                              'intermediate_size': v.intermediate_size, 'image_size': v.image_size, 'patch_size': v.patch_size,
                              'layer_norm_eps': v.layer_norm_eps, 'initializer_factor': v.initializer_factor, 'norm_type': 'layer_norm',
                              'qk_normalization': False, 'qkv_bias': True, 'hidden_act': 'gelu'}}


def load_hf_checkpoint(path: str):
    """(config dict, state dict) of an HF-format checkpoint directory: config.json + model.safetensors, or
    model.safetensors.index.json + shards, or pytorch_model.bin (InternVLChatModel.from_pretrained's formats)."""
    import json
    import os
    with open(os.path.join(path, 'config.json')) as f:
        cfg = json.load(f)
    idx = os.path.join(path, 'model.safetensors.index.json')
    one = os.path.join(path, 'model.safetensors')
    sd = {}
    if os.path.exists(idx):
        from safetensors.torch import load_file
        with open(idx) as f:
            shards = sorted(set(json.load(f)['weight_map'].values()))
        for sh in shards:
            sd.update(load_file(os.path.join(path, sh)))
    elif os.path.exists(one):
        from safetensors.torch import load_file
        sd = load_file(one)
    elif os.path.exists(os.path.join(path, 'pytorch_model.bin')):
        import torch
        sd = torch.load(os.path.join(path, 'pytorch_model.bin'), map_location='cpu')
    else:
        raise FileNotFoundError(f'no model.safetensors(.index.json) / pytorch_model.bin under {path}')
    return cfg, sd


def save_hf_checkpoint(path: str, cfg: VlaserConfig, sd: dict, max_shard_bytes: int = 4 << 30):
    """config.json + sharded safetensors with an index (HF layout; keys unchanged)."""
    import json
    import os
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, 'config.json'), 'w') as f:
        json.dump(to_hf_config(cfg), f, indent=1)
    shards, cur, size = [], {}, 0
    for k, t in sd.items():
        b = t.numel() * t.element_size()
        if cur and size + b > max_shard_bytes:
            shards.append(cur); cur, size = {}, 0
        cur[k] = t.detach().cpu().contiguous(); size += b
    shards.append(cur)
    wm = {}
    for i, sh in enumerate(shards):
        name = f'model-{i + 1:05d}-of-{len(shards):05d}.safetensors'
        save_file(sh, os.path.join(path, name), metadata={'format': 'pt'})
        wm.update({k: name for k in sh})
    with open(os.path.join(path, 'model.safetensors.index.json'), 'w') as f:
        json.dump({'metadata': {'total_size': sum(t.numel() * t.element_size() for t in sd.values())}, 'weight_map': wm}, f, indent=1)
