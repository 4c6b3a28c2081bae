"""InternVLChatModel.generate / forward restated (modeling_internvl_chat.py:143-255,400-440)."""
import torch
import torch.nn.functional as F

from . import qwen2, vit

LM = 'language_model.'


def merge_embeddings(sd, cfg, input_ids, vit_embeds):
    """embed_tokens + visual-token scatter (modeling_internvl_chat.py:418-427): rows where
    input_ids == img_context_token_id are overwritten, in row-major order, by vit_embeds.reshape(-1, C)."""
    e = F.embedding(input_ids, sd[LM + 'model.embed_tokens.weight'])
    B, N, C = e.shape
    e = e.reshape(B * N, C).clone()
    sel = input_ids.reshape(B * N) == cfg.img_context_token_id
    assert int(sel.sum()) != 0
    e[sel] = vit_embeds.reshape(-1, C).to(e.dtype)
    return e.reshape(B, N, C), sel.nonzero().flatten()


def generate(sd, cfg, pixel_values, input_ids, attention_mask=None, max_new_tokens=16, eos_token_id=None,
             return_logits=False):
    vit_embeds = vit.extract_feature(sd, cfg, pixel_values)
    embeds, _ = merge_embeddings(sd, cfg, input_ids, vit_embeds)
    return qwen2.greedy_generate(sd, LM, cfg.llm, embeds, attention_mask, max_new_tokens, eos_token_id,
                                 return_logits=return_logits)


def forward_logits(sd, cfg, pixel_values, input_ids, attention_mask=None, position_ids=None, image_flags=None):
    """InternVLChatModel.forward up to logits (:162-204)."""
    vit_embeds = vit.extract_feature(sd, cfg, pixel_values)
    if image_flags is not None:
        vit_embeds = vit_embeds[image_flags.reshape(-1) == 1]
    embeds, _ = merge_embeddings(sd, cfg, input_ids, vit_embeds)
    B, S, _ = embeds.shape
    if attention_mask is None:
        attention_mask = torch.ones(B, S, dtype=torch.long)
    if position_ids is None:
        position_ids = (attention_mask.long().cumsum(-1) - 1).clamp(min=0)
    mask = qwen2.causal_mask(S, S, embeds.dtype, attention_mask)
    h, _ = qwen2.model_forward(sd, LM, cfg.llm, embeds, position_ids, mask)
    return qwen2.lm_head(sd, LM, h)


def sft_loss(logits, labels):
    """Shifted CrossEntropyLoss, mean over labels != -100 (modeling_internvl_chat.py:231-243)."""
    sl = logits[..., :-1, :].contiguous().float()
    tl = labels[..., 1:].contiguous()
    return F.cross_entropy(sl.view(-1, sl.shape[-1]), tl.view(-1), ignore_index=-100)
