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
