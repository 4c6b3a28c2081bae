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


def packed_logits(sd, cfg, pixel_values, input_ids, cu_seqlens, image_flags=None):
    """Logits of a PACKED row (dataset_packed.py:517-624): block-diagonal causal attention + restarting position ids make the
    sub-sequences independent -- each [cu[i], cu[i+1]) slice is an ordinary causal forward over its own tiles."""
    nt = cfg.num_image_token
    flags = None if image_flags is None else image_flags.reshape(-1)
    out, t0 = [], 0
    for lo, hi in zip(cu_seqlens[:-1].tolist(), cu_seqlens[1:].tolist()):
        ids = input_ids[:, lo:hi]
        need = int((ids == cfg.img_context_token_id).sum()) // nt
        t1, got = t0, 0
        while t1 < pixel_values.shape[0] and got < need:
            got += 1 if (flags is None or flags[t1] == 1) else 0
            t1 += 1
        if need == 0:
            if flags is not None and t1 < pixel_values.shape[0] and flags[t1] == 0:
                t1 += 1                                            # dummy tile of a text-only sub-sequence: no visual token uses it
            e = F.embedding(ids, sd[LM + 'model.embed_tokens.weight'])
            S = e.shape[1]
            h, _ = qwen2.model_forward(sd, LM, cfg.llm, e, torch.arange(S)[None], qwen2.causal_mask(S, S, e.dtype, torch.ones(1, S, dtype=torch.long)))
            out.append(qwen2.lm_head(sd, LM, h))
        else:
            fl = None if flags is None else flags[t0:t1]
            out.append(forward_logits(sd, cfg, pixel_values[t0:t1], ids, image_flags=fl))
        t0 = t1
    return torch.cat(out, 1)


def packed_loss(logits, labels, loss_weight):
    """sum(w_t ce_t) / sum(w_t) over the flat shifted row (modeling_internvl_chat.py:207-230)."""
    sl = logits[..., :-1, :].contiguous().float()
    ce = F.cross_entropy(sl.view(-1, sl.shape[-1]), labels[..., 1:].reshape(-1), ignore_index=-100, reduction='none')
    w = loss_weight[..., 1:].reshape(-1).float()
    return (ce * w).sum() / w.sum()
