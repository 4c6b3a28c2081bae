"""Drop-in mirror of the reference's `InternVLChatModel` surface (Vlaser_VLM/internvl_chat/internvl/model/
internvl_chat/modeling_internvl_chat.py:39-450) running on hand-written gfx950 kernels.

Same method names, argument meaning, attributes and error behaviour:
  __init__(config)                      :48     (config: vlaser_amd.config.VlaserConfig or an HF-style dict)
  load_state_dict(sd)                           HF checkpoint key names unchanged
  extract_feature(pixel_values)         :273    [T,3,448,448] -> [T,256,H]
  generate(pixel_values, input_ids, attention_mask, visual_features=None, **generate_kwargs)   :400
  chat(tokenizer, pixel_values, question, generation_config, history=None, return_history=False, ...)   :343
  batch_chat(...)                       :293
  forward(pixel_values, input_ids, attention_mask, position_ids, image_flags, labels, ...)      :143
Attributes read by callers: num_image_token, template, system_message, img_context_token_id, config.

There is no CPU path: constructing the model without the HIP library / a GPU raises.
"""
from types import SimpleNamespace

import os

import torch

from . import _lib as L
from . import ops, prep
from .config import VlaserConfig
from .engine import BF, KVCache, PrefillBuffers, QwenStack, SkinnyBuffers, VitEngine, prefill_begin, prefill_layer, skinny_layer


class InternVLChatModel:
    def __init__(self, config: VlaserConfig, device='cuda', max_tiles=1, max_seq_len=1024, max_batch=1, decode_graph=True):
        L.lib()   # fail loudly when the HIP library is missing
        self.decode_graph = decode_graph          # uniform batches: greedy / sampled decode steps replayed from one HIP graph
        if not torch.cuda.is_available():
            raise L.VlaserHipError('vlaser_amd needs an MI355X (gfx950) GPU: there is no CPU fallback')
        self.config = config
        self.device = torch.device(device)
        self.num_image_token = config.num_image_token
        self.template = config.template
        self.ps_version = config.ps_version
        self.select_layer = config.select_layer
        self.downsample_ratio = config.downsample_ratio
        self.conv_template = prep.get_conv_template(self.template)
        self.system_message = self.conv_template.system_message
        self.img_context_token_id = None
        self.num_samples = 0
        self._max_tiles, self._max_seq, self._max_batch = max_tiles, (max_seq_len + 63) // 64 * 64, max_batch
        self.vit = None
        self.llm = None
        if config.select_layer != -1:
            raise NotImplementedError('only select_layer == -1 (last hidden state) is implemented')

    # ------------------------------------------------------------------ weights
    @classmethod
    def from_pretrained(cls, path, device='cuda', **kw):
        """HF-format checkpoint directory (config.json + safetensors shards / pytorch_model.bin), key names unchanged
        (the reference: InternVLChatModel.from_pretrained, eval_example.py:112-122)."""
        from .config import from_hf_config, load_hf_checkpoint
        hf_cfg, sd = load_hf_checkpoint(path)
        model = cls(from_hf_config(hf_cfg), device=device, **kw)
        if hf_cfg.get('system_message'):
            model.system_message = hf_cfg['system_message']
        model.load_state_dict(sd)
        return model

    def expected_keys(self):
        """The HF key names this model consumes (module definitions: modeling_intern_vit.py:141-152,196-208,256-257,275-279,
        modeling_internvl_chat.py:89-94, HF Qwen2ForCausalLM), i.e. `InternVLChatModel(config).state_dict().keys()` of the reference."""
        v, llm = self.config.vision, self.config.llm
        keys = ['vision_model.embeddings.class_embedding', 'vision_model.embeddings.patch_embedding.weight',
                'vision_model.embeddings.patch_embedding.bias', 'vision_model.embeddings.position_embedding']
        for i in range(v.num_hidden_layers):
            p = f'vision_model.encoder.layers.{i}.'
            keys += [p + n for n in ('attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight', 'attn.proj.bias', 'mlp.fc1.weight', 'mlp.fc1.bias',
                                     'mlp.fc2.weight', 'mlp.fc2.bias', 'norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias', 'ls1', 'ls2')]
        keys += [f'mlp1.{i}.{n}' for i in (0, 1, 3) for n in ('weight', 'bias')]
        keys += ['language_model.model.embed_tokens.weight', 'language_model.model.norm.weight', 'language_model.lm_head.weight']
        for i in range(llm.num_hidden_layers):
            p = f'language_model.model.layers.{i}.'
            keys += [p + n for n in ('self_attn.q_proj.weight', 'self_attn.q_proj.bias', 'self_attn.k_proj.weight', 'self_attn.k_proj.bias',
                                     'self_attn.v_proj.weight', 'self_attn.v_proj.bias', 'self_attn.o_proj.weight', 'mlp.gate_proj.weight',
                                     'mlp.up_proj.weight', 'mlp.down_proj.weight', 'input_layernorm.weight', 'post_attention_layernorm.weight')]
        return keys

    def load_state_dict(self, sd, strict=True):
        """nn.Module.load_state_dict semantics: `strict=True` raises a RuntimeError naming every missing and unexpected key;
        `strict=False` loads what is there (tensors the kernels need must still be present) and returns both lists.  Rotary
        `inv_freq` buffers of older HF checkpoints are derived, not loaded, and are never reported."""
        need = self.expected_keys()
        have = set(sd.keys())
        missing = [k for k in need if k not in have]
        needset = set(need)
        unexpected = sorted(k for k in have if k not in needset and not k.endswith('rotary_emb.inv_freq'))
        if strict and (missing or unexpected):
            msg = [f'Error(s) in loading state_dict for {type(self).__name__}:']
            if missing:
                msg.append('\tMissing key(s) in state_dict: ' + ', '.join(f'"{k}"' for k in missing[:20]) + (' ...' if len(missing) > 20 else '') + '.')
            if unexpected:
                msg.append('\tUnexpected key(s) in state_dict: ' + ', '.join(f'"{k}"' for k in unexpected[:20]) + (' ...' if len(unexpected) > 20 else '') + '.')
            raise RuntimeError('\n'.join(msg))
        if missing:
            raise RuntimeError(f'state_dict lacks tensors the kernels need: {missing[:8]}{" ..." if len(missing) > 8 else ""}')
        self.vit = VitEngine(sd, self.config, self.device, max_tiles=self._max_tiles)
        self.use_skinny = ops.skinny_supported(self.config.llm)      # Vlaser-8B (hidden 3584) decodes through the GEMM path
        # 16-row lane-local units for the q/k/v and gate/up GEMVs of the <= 16-row path (r03 kernels; at H = 1536 both help the decode: 1.169 -> 1.12 ms per token,
        # same-box A/B r04; the action expert at K = 768 keeps gate/up on 32-row units): VLASER_DECODE_OPTS=none restores the r03 decode.
        # 'chain' (r05, csrc/chain.hip): q/k/v, gate/up and the down projection on the latency-built kernels (batch <= 8): 1.089 -> 1.038 ms per token, same-box A/B
        self.llm = QwenStack(sd, 'language_model.', self.config.llm, self.device, skinny=self.use_skinny,
                             opts=tuple(o for o in os.environ.get('VLASER_DECODE_OPTS', 'qkv16,gu16,chain' if self.config.llm.hidden_size in (768, 1536) else '').split(',')
                                        if o and o != 'none'))      # (the 16-row units are built for hidden sizes 768 / 1536: 3 / 6 K-steps per wave)
        self._alloc_llm()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def _alloc_llm(self):
        llm, dev = self.config.llm, self.device
        self.cache = KVCache(llm.num_hidden_layers, self._max_batch, llm.num_key_value_heads, self._max_seq, dev, llm.head_dim)
        self.pbuf = PrefillBuffers(self.llm, self._max_batch * self._max_seq, dev)
        self.sbuf = SkinnyBuffers(self.llm, 16, dev)
        self.rope = ops.rope_table(self._max_seq + 8, llm.head_dim, llm.rope_theta, dev)
        self.h = torch.zeros(self._max_batch * self._max_seq, llm.hidden_size, dtype=BF, device=dev)
        self.rank_ws = torch.zeros(self._max_batch * self._max_seq, dtype=torch.int32, device=dev)
        self.img_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.logits = torch.zeros(16, llm.vocab_size, dtype=torch.float32, device=dev)
        self.next_ids = torch.zeros(16, dtype=torch.int64, device=dev)
        self.argmax_ws = ops.argmax_workspace(16, dev)      # r05: the vocabulary row spread over 64 workgroups (one CU's request rate was the whole 18 us)
        self.next_h = torch.zeros(16, llm.hidden_size, dtype=BF, device=dev)
        # device-resident decode state: pos1 = position (= cache slot) of the incoming token of each sequence, vis = keys visible to it
        self.dyn = torch.zeros(32, dtype=torch.int32, device=dev)
        self.pos1, self.vis = self.dyn[:16], self.dyn[16:]
        self._dec_graphs = {}                     # (batch, key-count bound) -> captured decode step (workspaces above were just re-allocated)

    def _ensure(self, batch, seq):
        seq = (seq + 63) // 64 * 64
        if batch > self._max_batch or seq > self._max_seq:
            self._max_batch, self._max_seq = max(batch, self._max_batch), max(seq, self._max_seq)
            self._alloc_llm()

    # ------------------------------------------------------------------ vision
    def _to_bf16(self, pixel_values):
        pv = pixel_values.to(self.device)
        if pv.dtype == torch.float32:
            out = torch.empty(pv.shape, dtype=BF, device=self.device)
            ops.cast_f32_bf16(pv.contiguous(), out)
            return out
        return pv.to(BF).contiguous()

    def pixel_shuffle(self, x, scale_factor=0.5):
        """[n,w,h,c] bf16 (no CLS) -> [n, w*s, h*s, c/s^2]: same permutation as modeling_internvl_chat.py:257-271."""
        n, w, h, c = x.shape
        assert w == h and scale_factor == 0.5
        xin = torch.zeros(n, w * h + 1, c, dtype=BF, device=self.device)
        xin[:, 1:] = x.reshape(n, w * h, c).to(BF)
        out = torch.empty(n * (w // 2) * (h // 2), 4 * c, dtype=BF, device=self.device)
        ops.pixel_shuffle(xin, out, n, w, c, 1 if self.ps_version == 'v1' else 0)
        return out.view(n, w // 2, h // 2, 4 * c)

    def extract_feature(self, pixel_values):
        feats = self.vit.forward(self._to_bf16(pixel_values))
        return feats.view(pixel_values.shape[0], self.num_image_token, -1).clone()

    # ------------------------------------------------------------------ LLM plumbing
    def _embed(self, input_ids, vit_embeds, zero_pad=False):
        B, S = input_ids.shape
        ids = input_ids.to(self.device).contiguous()
        h = self.h[:B * S]
        vit2d = None if vit_embeds is None else vit_embeds.reshape(-1, vit_embeds.shape[-1])
        ops.embed_merge(ids, self.llm.embed, vit2d, h, self.img_context_token_id if self.img_context_token_id is not None else -1,
                        self.config.pad_token_id, zero_pad, self.rank_ws, self.img_count)
        return h

    def _prefill(self, h, B, S, pos_ids, final_norm=False):
        """28x Qwen2DecoderLayer; with final_norm the last fused seam also applies model.norm into self.pbuf.x."""
        layers = self.llm.layers
        prefill_begin(self.llm, self.pbuf, h, B * S)
        for i, lw in enumerate(layers):
            nxt = layers[i + 1].ln_in if i + 1 < len(layers) else (self.llm.norm if final_norm else None)
            prefill_layer(self.llm, lw, self.pbuf, h, self.cache, i, self.rope, pos_ids, B, S, L.ATTN_CAUSAL, next_norm_w=nxt)
        return h

    def _head_last(self, h_last, partials, n_partials, M, greedy=True):
        """final RMSNorm + lm_head on M rows -> fp32 logits (+ argmax and next-token embedding gather)."""
        llm = self.config.llm
        if self.use_skinny:
            ops.skinny(L.PRO_NORM, L.SK_F32, h_last, self.llm.sk_head, M, partials=partials, n_partials=n_partials, norm_w=self.llm.norm,
                       eps=llm.rms_norm_eps, out_f32=self.logits)
        else:
            assert partials is None
            ops.rmsnorm(h_last, self.llm.norm, llm.rms_norm_eps, out=self.pbuf.x[:M])
            ops.gemm(L.EPI_F32, self.pbuf.x[:M], self.llm.head, out=self.logits[:M])
        if greedy:
            ops.argmax(self.logits[:M], self.next_ids, self.llm.embed, self.next_h, ws=self.argmax_ws)

    def _decode_step(self, B, L_cur, lens=None, step=0):
        """One greedy step for B sequences: consumes self.next_h.  Uniform batches hold L_cur cached tokens each.  Ragged
        batches (lens = int32 [B] prompt lengths on the device) keep prompt b in slots [0, lens[b]) and every generated
        token in the common slots [L_cur, L_cur + step]: the PREFIX descriptor (valid_len = lens, blk_start = L_cur)
        replaces HF's left-padding + additive mask (modeling_internvl_chat.py:326-341) with the same visibility."""
        if lens is None:
            self.pos1[:B].fill_(L_cur)
            slot, kv_len, mode, kw = L_cur, L_cur + 1, L.ATTN_FULL, {}
        else:
            torch.add(lens, step, out=self.pos1[:B])
            slot, kv_len, mode, kw = L_cur + step, L_cur + step + 1, L.ATTN_PREFIX, dict(valid_len=lens, blk_start=L_cur)
        if not self.use_skinny:
            # MFMA GEMM path with one row per sequence: K/V appended at `slot`, PREFIX/CAUSAL visibility by descriptor
            h, layers = self.next_h, self.llm.layers
            prefill_begin(self.llm, self.pbuf, h, B)
            for i, lw in enumerate(layers):
                nxt = layers[i + 1].ln_in if i + 1 < len(layers) else None
                prefill_layer(self.llm, lw, self.pbuf, h, self.cache, i, self.rope, self.pos1, B, 1,
                              L.ATTN_PREFIX if lens is not None else L.ATTN_CAUSAL, causal_off=slot, kv_len=kv_len, next_norm_w=nxt,
                              slot_base=slot, **kw)
            self._head_last(h, None, 0, B)
            return
        h, parts, npart = self.next_h, None, 0
        for i, lw in enumerate(self.llm.layers):
            h, parts, npart = skinny_layer(self.llm, lw, self.sbuf, h, parts, npart, self.cache, i, self.rope, self.pos1, B, 1, slot,
                                           kv_len, mode, **kw)
        self._head_last(h, parts, npart, B)

    def _decode_dyn(self, B, kvmax):
        """One greedy step for a UNIFORM batch on the weight-streaming path with every per-step scalar on the device: the cache slot of
        the incoming token is its position id (`slot_base = -1`), the visible key count is `vis[b]` (PREFIX descriptor with an empty
        trailing block), the key-chunk schedule is sized for `kvmax` keys (keys beyond `vis` are masked; the cache is finite there).
        Nothing in the launch sequence depends on the step, so it is captured once in a HIP graph and replayed; the state advances at
        the end of the step.  (eager decode was host-bound, 141 launches through ctypes per token.)"""
        h, parts, npart = self.next_h, None, 0
        for i, lw in enumerate(self.llm.layers):
            h, parts, npart = skinny_layer(self.llm, lw, self.sbuf, h, parts, npart, self.cache, i, self.rope, self.pos1, B, 1, -1,
                                           kvmax, L.ATTN_PREFIX, valid_len=self.vis, blk_start=kvmax)
        self._head_last(h, parts, npart, B)
        self.dyn.add_(1)

    def _decode_step_graph(self, B, L_cur, step, max_new_tokens):
        """Step `step` of a uniform batch holding L_cur + step cached tokens: first use of a (B, key bound) runs eagerly (kernel
        attributes, plan cache), the second is captured, later ones are graph replays."""
        kvmax = min(self.cache.s_max, (L_cur + max_new_tokens + 63) // 64 * 64)
        if step == 0:
            self.dyn[:16].fill_(L_cur)
            self.dyn[16:].fill_(L_cur + 1)
        key = (B, kvmax)
        g = self._dec_graphs.get(key)
        if g is None:
            self._decode_dyn(B, kvmax)                          # eager (also the warm-up of the capture below)
            self._dec_graphs[key] = 'warm'
        elif g == 'warm':
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_dyn(B, kvmax)
            self._dec_graphs[key] = g
            g.replay()
        else:
            g.replay()

    @staticmethod
    def _compact_padded(input_ids, attention_mask, pad):
        """Padded batch (either side) -> right-padded ids trimmed to the longest prompt + int32 lengths.  Valid tokens keep
        their order, so the row-major <IMG_CONTEXT> scatter order (:418-427) is unchanged."""
        am = attention_mask.bool().cpu()
        lens = am.sum(1)
        if int(lens.min()) == 0:
            raise ValueError('a sequence of the batch has no valid token')
        S = int(lens.max())
        ids = torch.full((input_ids.shape[0], S), pad, dtype=input_ids.dtype)
        for b in range(input_ids.shape[0]):
            ids[b, :int(lens[b])] = input_ids[b].cpu()[am[b]]
        return ids, lens.to(torch.int32)

    # ------------------------------------------------------------------ public surfaces
    @torch.no_grad()
    def generate(self, pixel_values=None, input_ids=None, attention_mask=None, visual_features=None, generation_config=None,
                 output_hidden_states=None, max_new_tokens=None, min_new_tokens=0, do_sample=False, eos_token_id=None,
                 pad_token_id=None, return_logits=False, **generate_kwargs):
        assert self.img_context_token_id is not None
        # HF generation features this decoder does not implement must not be dropped silently
        allowed = {'temperature', 'top_k', 'top_p', 'generator', 'use_cache', 'return_dict', 'return_dict_in_generate', 'output_scores',
                   'output_attentions', 'num_return_sequences', 'num_beams', 'repetition_penalty', 'length_penalty', 'no_repeat_ngram_size',
                   'early_stopping', 'do_sample', 'max_length'}
        neutral = {'num_beams': 1, 'repetition_penalty': 1.0, 'length_penalty': 1.0, 'no_repeat_ngram_size': 0, 'num_return_sequences': 1,
                   'early_stopping': False, 'output_scores': False, 'output_attentions': False, 'return_dict_in_generate': False, 'use_cache': True}
        for k, v in generate_kwargs.items():
            if k not in allowed:
                raise TypeError(f'generate() got an unsupported generation argument {k!r}')
            if k in neutral and v is not None and v != neutral[k]:
                raise NotImplementedError(f'generate(): {k}={v!r} is not implemented (greedy / temperature / top-k / top-p sampling only)')
        if generate_kwargs.get('max_length') is not None and max_new_tokens is None:
            max_new_tokens = max(1, int(generate_kwargs['max_length']) - int(input_ids.shape[1]))
        # do_sample: HF's logits warpers in their order (temperature -> top_k -> top_p) + multinomial on the fp32 logits the
        # lm_head kernel leaves on the device; not on the hot path (the reference's eval configs decode greedily)
        sample = dict(temperature=float(generate_kwargs.pop('temperature', 1.0) or 1.0), top_k=int(generate_kwargs.pop('top_k', 0) or 0),
                      top_p=float(generate_kwargs.pop('top_p', 1.0) or 1.0), generator=generate_kwargs.pop('generator', None)) if do_sample else None
        if generation_config is not None:
            max_new_tokens = max_new_tokens or getattr(generation_config, 'max_new_tokens', None)
            eos_token_id = eos_token_id if eos_token_id is not None else getattr(generation_config, 'eos_token_id', None)
        max_new_tokens = max_new_tokens or 20          # HF default max_length heritage
        B = input_ids.shape[0]
        if B > 16:
            # the weight-streaming decode takes <= 16 rows per launch: larger batches run as consecutive groups of 16 sequences
            if return_logits or visual_features is not None:
                raise NotImplementedError('return_logits / visual_features are per-group features: call generate() with <= 16 sequences')
            tiles = ((input_ids == self.img_context_token_id).sum(1) // self.num_image_token).tolist() if pixel_values is not None else [0] * B
            pad = pad_token_id if pad_token_id is not None else (eos_token_id[0] if isinstance(eos_token_id, (list, tuple)) else eos_token_id) or 0
            outs, off = [], 0
            for lo in range(0, B, 16):
                hi = min(B, lo + 16)
                nt = sum(tiles[lo:hi])
                outs.append(self.generate(None if pixel_values is None else pixel_values[off:off + nt], input_ids[lo:hi],
                                          None if attention_mask is None else attention_mask[lo:hi], max_new_tokens=max_new_tokens,
                                          min_new_tokens=min_new_tokens, eos_token_id=eos_token_id, pad_token_id=pad_token_id,
                                          do_sample=do_sample, **(dict(sample) if sample else {})))
                off += nt
            n = max(o.shape[1] for o in outs)
            return torch.cat([torch.nn.functional.pad(o, (0, n - o.shape[1]), value=pad) for o in outs], 0)
        lens = None
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            input_ids, lens_cpu = self._compact_padded(input_ids, attention_mask, self.config.pad_token_id)
            lens = lens_cpu.to(self.device)
        S = input_ids.shape[1]
        self._ensure(B, S + max_new_tokens)
        if pixel_values is not None:
            vit_embeds = visual_features if visual_features is not None else self.vit.forward(self._to_bf16(pixel_values))
            n_sel = int((input_ids == self.img_context_token_id).sum())
            assert n_sel != 0
            if n_sel != vit_embeds.numel() // vit_embeds.shape[-1]:
                raise RuntimeError(f'shape mismatch: {n_sel} <IMG_CONTEXT> tokens vs {vit_embeds.numel() // vit_embeds.shape[-1]} visual tokens')
        else:
            vit_embeds = None
        h = self._embed(input_ids, vit_embeds)
        pos = torch.arange(S, dtype=torch.int32, device=self.device).repeat(B)
        self._prefill(h, B, S, pos)
        if lens is None:
            last = h.view(B, S, -1)[:, -1].contiguous()
        else:                                       # causal prefill: right padding never reaches a valid row
            last = h.view(B, S, -1)[torch.arange(B, device=self.device), (lens - 1).long()].contiguous()
        self._head_last(last, None, 0, B)
        out, logits_out = [], []
        eos = eos_token_id if isinstance(eos_token_id, (list, tuple)) or eos_token_id is None else [eos_token_id]
        finished = torch.zeros(B, dtype=torch.bool)
        pad = pad_token_id if pad_token_id is not None else (eos[0] if eos else 0)
        for step in range(max_new_tokens):
            if return_logits:
                logits_out.append(self.logits[:B].clone())
            if sample is not None:
                self._sample_next(B, **sample)
            nxt = self.next_ids[:B].cpu()
            nxt = torch.where(finished, torch.full_like(nxt, pad), nxt)
            out.append(nxt)
            if eos is not None and step + 1 >= min_new_tokens:
                finished = finished | torch.isin(nxt, torch.tensor(eos))
            if step == max_new_tokens - 1 or bool(finished.all()):
                break
            if lens is None and self.use_skinny and self.decode_graph:
                self._decode_step_graph(B, S, step, max_new_tokens)
            elif lens is None:
                self._decode_step(B, S + step)
            else:
                self._decode_step(B, S, lens, step)
        ids = torch.stack(out, dim=1).to(self.device)
        if return_logits:
            return ids, torch.stack(logits_out, dim=1)
        return ids

    def _sample_next(self, B, temperature=1.0, top_k=0, top_p=1.0, generator=None):
        """Replace the greedy pick in next_ids / next_h by a sample from the warped distribution (torch ops on device logits)."""
        lg = self.logits[:B] / temperature
        if top_k > 0:
            kth = lg.topk(min(top_k, lg.shape[-1]), dim=-1).values[:, -1:]
            lg = lg.masked_fill(lg < kth, float('-inf'))
        if top_p < 1.0:
            srt, idx = lg.sort(dim=-1, descending=False)
            cum = srt.softmax(-1).cumsum(-1)
            drop = cum <= (1.0 - top_p)
            drop[:, -1] = False                                   # always keep the most likely token
            lg = lg.masked_fill(drop.scatter(1, idx, drop), float('-inf'))
        pick = torch.multinomial(lg.softmax(-1), 1, generator=generator).squeeze(1)
        self.next_ids[:B].copy_(pick)
        self.next_h[:B].copy_(self.llm.embed[pick])

    def chat(self, tokenizer, pixel_values, question, generation_config, history=None, return_history=False,
             num_patches_list=None, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>', IMG_CONTEXT_TOKEN='<IMG_CONTEXT>',
             verbose=False):
        if num_patches_list is None:
            num_patches_list = [pixel_values.shape[0]] if pixel_values is not None else []
        assert pixel_values is None or len(pixel_values) == sum(num_patches_list)
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        query, question, template = prep.build_chat_query(self.template, self.system_message, question, num_patches_list,
                                                          self.num_image_token, history, pixel_values is not None)
        eos_token_id = tokenizer.convert_tokens_to_ids(template.sep.strip())
        history = [] if history is None else history
        if verbose and pixel_values is not None:
            print(f'dynamic ViT batch size: {pixel_values.shape[0]}')
        model_inputs = tokenizer(query, return_tensors='pt')
        generation_config['eos_token_id'] = eos_token_id      # the reference mutates the caller's dict too (:381)
        out = self.generate(pixel_values=pixel_values, input_ids=model_inputs['input_ids'],
                            attention_mask=model_inputs['attention_mask'], **generation_config)
        response = tokenizer.batch_decode(out, skip_special_tokens=True)[0]
        response = response.split(template.sep.strip())[0].strip()
        history.append((question, response))
        if return_history:
            return response, history
        if verbose:
            print(query.replace(IMG_CONTEXT_TOKEN, '').replace(f'{IMG_START_TOKEN}{IMG_END_TOKEN}', '<image>'), response)
        return response

    def batch_chat(self, tokenizer, pixel_values, questions, generation_config, num_patches_list=None, history=None,
                   return_history=False, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>', IMG_CONTEXT_TOKEN='<IMG_CONTEXT>',
                   verbose=False, image_counts=None):
        if history is not None or return_history:
            print('Now multi-turn chat is not supported in batch_chat.')
            raise NotImplementedError
        if image_counts is not None:
            num_patches_list = image_counts
            print('Warning: `image_counts` is deprecated. Please use `num_patches_list` instead.')
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        if verbose and pixel_values is not None:
            print(f'dynamic ViT batch size: {pixel_values.shape[0]}')
        queries, template = [], None
        for idx, num_patches in enumerate(num_patches_list):
            question = questions[idx]
            if pixel_values is not None and '<image>' not in question:
                question = '<image>\n' + question
            query, _, template = prep.build_chat_query(self.template, self.system_message, question, [num_patches],
                                                       self.num_image_token, None, pixel_values is not None)
            queries.append(query)
        tokenizer.padding_side = 'left'                    # as the reference does (:318)
        model_inputs = tokenizer(queries, return_tensors='pt', padding=True)
        eos_token_id = tokenizer.convert_tokens_to_ids(template.sep.strip())
        generation_config['eos_token_id'] = eos_token_id
        responses = []
        for b0 in range(0, len(queries), 16):              # the weight-streaming decode handles <= 16 rows per launch
            n_tiles0, n_tiles1 = sum(num_patches_list[:b0]), sum(num_patches_list[:b0 + 16])
            pv = None if pixel_values is None else pixel_values[n_tiles0:n_tiles1]
            out = self.generate(pixel_values=pv, input_ids=model_inputs['input_ids'][b0:b0 + 16],
                                attention_mask=model_inputs['attention_mask'][b0:b0 + 16], **generation_config)
            responses += tokenizer.batch_decode(out, skip_special_tokens=True)
        return [r.split(template.sep.strip())[0].strip() for r in responses]

    @torch.no_grad()
    def forward(self, pixel_values, input_ids=None, attention_mask=None, position_ids=None, image_flags=None,
                past_key_values=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, statistics=None, loss_weight=None, loss_reduction_all_gather=False):
        """Inference forward (logits + CE loss).  The trainable SFT step lives in vlaser_amd.sft."""
        if past_key_values is not None:
            raise NotImplementedError('external caches are not supported by forward() (generate() owns the KV cache)')
        if loss_weight is not None:
            return self._forward_packed(pixel_values, input_ids, attention_mask, labels, image_flags, loss_weight, loss_reduction_all_gather)
        B, S = input_ids.shape
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            am = attention_mask.bool()
            if not bool((am[:, :-1] | ~am[:, 1:]).all()):      # a 0 followed by a 1: not right padded
                raise NotImplementedError('forward() takes unpadded or right-padded batches (the SFT collator pads on the '
                                          'right, pad_data_collator.py:57-72); causal attention keeps pads out of valid rows')
        self._ensure(B, S)
        if pixel_values is None or pixel_values.shape[0] == 0:                  # text-only (sub-)sequence
            feats2d = torch.zeros(0, self.config.llm.hidden_size, dtype=BF, device=self.device)
        else:
            feats = self.vit.forward(self._to_bf16(pixel_values)).view(pixel_values.shape[0], self.num_image_token, -1)
            if image_flags is not None:
                feats = feats[image_flags.reshape(-1).to(self.device) == 1]
            feats2d = feats.reshape(-1, feats.shape[-1])
        n_sel = int((input_ids == self.img_context_token_id).sum())
        ignore_flag = False
        if n_sel != feats2d.shape[0]:
            # reference falls back to the first n_token features and zeroes the loss (:184-190, 242-243)
            print(f'warning: shape mismatch, input_embeds[selected].shape={n_sel}, vit_embeds.shape={tuple(feats2d.shape)}')
            feats2d = feats2d[:n_sel].contiguous()
            ignore_flag = True
        h = self._embed(input_ids, feats2d)
        if position_ids is None:
            pos = torch.arange(S, dtype=torch.int32, device=self.device).repeat(B)
        else:
            pos = position_ids.to(self.device).to(torch.int32).reshape(-1).contiguous()
        self._prefill(h, B, S, pos, final_norm=True)
        logits = ops.linear(self.pbuf.x[:B * S], self.llm.head, epi=L.EPI_F32).view(B, S, -1)
        loss = None
        if labels is not None:
            from .sft import ce_loss
            loss = ce_loss(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1).to(self.device))
            if ignore_flag:
                loss = loss * 0.0
        return SimpleNamespace(loss=loss, logits=logits, past_key_values=None, hidden_states=None, attentions=None)

    def _forward_packed(self, pixel_values, input_ids, cu_seqlens, labels, image_flags, loss_weight, loss_reduction_all_gather):
        """Packed-sequence forward (`--use_packed_ds`: dataset_packed.py:517-624, qwen2_packed_training_patch.py:14-101): every row of
        `input_ids` is a concatenation of sub-sequences, `attention_mask` carries their cu_seqlens [B, n+1] (:623), attention is
        block-diagonal causal (flash_attn_varlen_func, :71-96) -- i.e. the sub-sequences are independent, so each one runs through the
        ordinary causal path on its own -- and the loss is sum(w_t * ce_t) / sum(w_t) over the flat shifted row with the per-token
        `loss_weight` (modeling_internvl_chat.py:207-230)."""
        B, S = input_ids.shape
        cu = cu_seqlens.detach().to('cpu', torch.int64).reshape(B, -1)
        ids_h = input_ids.detach().to('cpu')
        nt = self.num_image_token
        flags = None if image_flags is None else image_flags.detach().to('cpu').reshape(-1)
        logits = torch.zeros(B, S, self.config.llm.vocab_size, dtype=torch.float32, device=self.device)
        t0 = 0
        for b in range(B):
            for lo, hi in zip(cu[b, :-1].tolist(), cu[b, 1:].tolist()):
                if hi <= lo:
                    continue
                need = int((ids_h[b, lo:hi] == self.img_context_token_id).sum()) // nt
                t1, got = t0, 0
                while pixel_values is not None and t1 < pixel_values.shape[0] and got < need:
                    got += 1 if (flags is None or flags[t1] == 1) else 0
                    t1 += 1
                sub = self.forward(None if pixel_values is None else pixel_values[t0:t1], ids_h[b:b + 1, lo:hi],
                                   image_flags=None if flags is None else flags[t0:t1].reshape(-1, 1))
                logits[b, lo:hi] = sub.logits[0]
                t0 = t1
        loss = None
        if labels is not None:
            w = torch.as_tensor(loss_weight, dtype=torch.float32).reshape(B, S)[:, 1:].reshape(-1).to(self.device)
            tgt = labels.to(self.device)[:, 1:].reshape(-1).contiguous()
            rows = torch.empty(tgt.numel(), dtype=torch.float32, device=self.device)
            ops.ce_rows(logits[:, :-1].reshape(-1, logits.shape[-1]), tgt, rows, None, -100)
            wsum = w.sum()
            if loss_reduction_all_gather and torch.distributed.is_available() and torch.distributed.is_initialized():
                torch.distributed.all_reduce(wsum, op=torch.distributed.ReduceOp.AVG)
            loss = (rows * w).sum() / wsum
        return SimpleNamespace(loss=loss, logits=logits, past_key_values=None, hidden_states=None, attentions=None)

    __call__ = forward

    # accessors the reference exposes (:442-450)
    @property
    def lm_head(self):
        return self.llm.head

    def get_input_embeddings(self):
        return self.llm.embed

    def get_output_embeddings(self):
        return self.llm.head

    def eval(self):
        return self
