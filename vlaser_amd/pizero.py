"""Drop-in mirror of the Vlaser-VLA inference surface `PiZero` / `PiZeroInference`
(Vlaser_VLA/Simpler/src/model/vla/pizero_internvl.py:154-336,517-603,798-936,1286-1307) on gfx950 kernels.

  infer_action(input_ids[B,384] i64, pixel_values[B*n,3,448,448], image_text_proprio_mask[B,1,385,385],
               action_mask[B,1,4,389], vlm_position_ids[B,384], proprio_position_ids[B,1],
               action_position_ids[B,4], proprios[B,1,7], noise=None, generator=None) -> [B,4,7]
               (pixel_values: normalised fp32 / bf16 as the reference passes them, or the raw uint8 observation -- normalised on the device)
  build_causal_mask_and_position_ids(attention_mask, dtype), split_full_mask_into_submasks(mask)

Extensions over the reference: an explicit `noise` / `generator` argument (the reference draws torch.randn inside
the method, :879-881) and `valid_len` descriptors derived from the dense masks (the kernels never read the
[B,1,389,389] tensors).  The whole call (ViT -> joint prefill -> 10 Euler steps) is allocation-free and is
replayed from ONE HIP graph after the first call of a given batch size.

Checkpoint: canonical (de-aliased) VLA state dict -- the InternVLChatModel keys plus action_expert.model.*,
action_encoder.*, proprio_encoder.*, action_decoder.* (see `canonicalize_vla_state_dict`).
"""
import os

import torch

from . import _lib as L
from . import ops, prep
from .config import VLAConfig
from .engine import BF, KVCache, PrefillBuffers, QwenStack, SkinnyBuffers, VitEngine, prefill_begin, prefill_layer, skinny_layer


def canonicalize_vla_state_dict(sd):
    """Map the released `.pt` `data["model"]` key aliases onto canonical names (eval.py:196-212 strips
    `_orig_mod.`; the same modules are registered under several names, pizero_internvl.py:253-288,508-510)."""
    out = {}
    rules = [('vision_tower.vision_model.', 'vision_model.'), ('internvl_model.vision_model.', 'vision_model.'),
             ('multi_modal_projector.', 'mlp1.'), ('internvl_model.mlp1.', 'mlp1.'),
             ('joint_model.mixtures.vlm.layers.', 'language_model.model.layers.'),
             ('joint_model.mixtures.vlm.norm.', 'language_model.model.norm.'),
             ('joint_model.mixtures.action.layers.', 'action_expert.model.layers.'),
             ('joint_model.mixtures.proprio.layers.', 'action_expert.model.layers.'),
             ('joint_model.mixtures.action.norm.', 'action_expert.model.norm.'),
             ('joint_model.mixtures.proprio.norm.', 'action_expert.model.norm.'),
             ('internvl_model.action_expert.model.norm.', 'action_expert.model.norm.'),
             ('internvl_model.language_model.', 'language_model.'),
             ('embed_tokens.', 'language_model.model.embed_tokens.'),
             ('lm_head.', 'language_model.lm_head.')]
    for k, v in sd.items():
        if k.startswith('_orig_mod.'):
            k = k[len('_orig_mod.'):]
        for a, b in rules:
            if k.startswith(a):
                k = b + k[len(a):]
                break
        if k in out and out[k].shape != v.shape:
            raise ValueError(f'alias collision with different shapes for {k}')
        out.setdefault(k, v)
    return out


def reference_vla_state_dict(sd, action_vocab_rows=None):
    """Inverse of `canonicalize_vla_state_dict`: canonical tensors -> the key names of the reference's `PiZero.state_dict()` INCLUDING
    every alias under which the reference registers the same module (vision_tower.vision_model == internvl_model.vision_model,
    proprio mixture == action mixture, embed_tokens, ... -- golden G9 lists all 1713 names), so that the reference's
    `load_checkpoint` (strict=False + "no missing keys", eval.py:196-212) accepts the file.  The action expert's vocabulary head
    (`internvl_model.action_expert.lm_head.weight`, never used) is written as zeros when the canonical dict has none."""
    rules = [('vision_model.', ['vision_tower.vision_model.', 'internvl_model.vision_model.']),
             ('mlp1.', ['multi_modal_projector.']),
             ('language_model.model.layers.', ['joint_model.mixtures.vlm.layers.']),
             ('language_model.model.norm.', ['joint_model.mixtures.vlm.norm.', 'internvl_model.language_model.model.norm.']),
             ('language_model.model.embed_tokens.', ['embed_tokens.', 'internvl_model.language_model.model.embed_tokens.']),
             ('language_model.lm_head.', ['internvl_model.language_model.lm_head.']),
             ('action_expert.model.layers.', ['joint_model.mixtures.action.layers.', 'joint_model.mixtures.proprio.layers.']),
             ('action_expert.model.norm.', ['joint_model.mixtures.action.norm.', 'joint_model.mixtures.proprio.norm.', 'internvl_model.action_expert.model.norm.']),
             ('action_expert.lm_head.', ['internvl_model.action_expert.lm_head.']),
             ('action_encoder.', ['action_encoder.']), ('proprio_encoder.', ['proprio_encoder.']), ('action_decoder.', ['action_decoder.'])]
    out = {}
    for k, v in sd.items():
        for a, bs in rules:
            if k.startswith(a):
                for b in bs:
                    out[b + k[len(a):]] = v
                break
        else:
            raise KeyError(f'no reference name for canonical key {k}')
    if 'internvl_model.action_expert.lm_head.weight' not in out:
        emb = sd['language_model.model.embed_tokens.weight']
        rows = emb.shape[0] if action_vocab_rows is None else action_vocab_rows
        out['internvl_model.action_expert.lm_head.weight'] = torch.zeros(rows, sd['action_expert.model.norm.weight'].shape[0], dtype=emb.dtype)
    return out


def save_vla_checkpoint(path, sd, cnt_update=0, cnt_batch=0, extra=None):
    """Write a VLA checkpoint in the reference's `step{N}.pt` layout (train.py:639-672): `{"cnt_update", "cnt_batch", "model": state_dict,
    ...}` with the reference's key names; `PiZero.load_checkpoint` / the reference's `EvalAgent.load_checkpoint` read `data["model"]`."""
    data = {'cnt_update': cnt_update, 'cnt_batch': cnt_batch, 'model': {k: v.detach().cpu() for k, v in reference_vla_state_dict(sd).items()}}
    data.update(extra or {})
    torch.save(data, path)


def stage_pixels(pixel_values, out, dev):
    """Host/device plumbing of the image input: bf16 -> copy; fp32 -> `vlaser_cast_f32_bf16`; **uint8 [N,3,H,W]** (the raw observation the
    reference's InternVLAProcessor normalises on the host, processing.py:303-311) -> `vlaser_normalize_u8` on the device: 1 byte per
    sample over PCIe instead of 4, no fp32 intermediate."""
    pv = pixel_values.to(dev)
    if pv.dtype == torch.uint8:
        ops.normalize_u8(pv.contiguous(), out, prep.VLA_MEAN, prep.VLA_STD, layout='chw', mode='vla')
    elif pv.dtype == torch.float32:
        ops.cast_f32_bf16(pv.contiguous(), out)
    else:
        out.copy_(pv)
    return out


class _ActionChunk(torch.Tensor):
    """The tensor `infer_action` returns when it was handed the reference's dense masks: an ordinary device tensor that remembers the model whose deferred mask
    check covers it.  The check itself never synchronises (pizero.py::_poll_errors); but the first thing a caller does with a chunk is bring it to the host
    (`actions[0].float().cpu().numpy()`, eval.py:139) -- that transfer waits for the chunk anyway, and the snapshot of the error word was enqueued behind the
    chunk on the same stream, so polling right AFTER a `.cpu()` / `.to('cpu')` / `.tolist()` / `.item()` costs nothing and turns the NaN chunk of an unsupported
    mask into its `ValueError` at the natural place instead of one call late."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        owner = None
        for a in args:
            if isinstance(a, _ActionChunk):
                owner = getattr(a, '_vl_owner', None)
                if owner is not None:
                    break
        out = super().__torch_function__(func, types, args, kwargs or {})
        if owner is None:
            return out
        to_host = getattr(func, '__name__', '') in ('tolist', 'item') or (isinstance(out, torch.Tensor) and out.device.type == 'cpu')
        if to_host:
            owner._poll_errors(block=False)
            return out.as_subclass(torch.Tensor) if isinstance(out, torch.Tensor) else out
        if isinstance(out, _ActionChunk):
            out._vl_owner = owner
        return out


class PiZero:
    # Euler-phase kernel options of the action expert (VLASER_EULER overrides: comma list, "none" = the r02 kernels):
    #   qkv16: 16-row lane-local units for the q/k/v weight-streaming GEMV (128 instead of 64 workgroups): -0.88 us per layer-step in-chain
    #   gu16: the same for gate/up (1120 units): +0.6 us in-chain with the r03 kernel (its five units streamed one after the other, see csrc/chain.hip) -- on again
    #         with 'chain' (r05), whose gate/up requests all five up front
    #   glue1: ONE launch between two passes through the layers (vlaser_vla_step: tail of Euler step s-1 + action encoder of step s) instead of four:
    #          10.9 vs 16.3 us in isolation, -0.085 ms per chunk in-chain (tools/micro/vla_step_lab.py, ab_chunk.py) -- ON.  Not bit-identical to the
    #          4-launch path (linear_1 / time embedding folded into linear_2 in fp32): same tolerance against oracle and goldens
    # measurements + in-kernel timelines: profiles/r03c_euler_fusion.md
    #   chain (r05, needs qkv16): q/k/v, gate/up and the down projection on the latency-built kernels of csrc/chain.hip -- every request of a launch issued up front,
    #          one wave per q/k/v unit, the down projection publishing the bf16 residual stream once (no split-K slabs for the next layer to re-reduce): ON.  Same
    #          tolerance against oracle / goldens as the skinny kernels; gate/up bit-identical, q/k/v and down sum in a different fp32 order
    EULER_DEFAULT = 'qkv16,gu16,glue1,chain'      # (gu16 ON with chain since r05: 224 workgroups x 5 units instead of 187 x 3, 11.79 -> 11.68 ms per chunk, same-box A/B)

    ERR_BITS = {1: 'image_text_proprio_mask: the keys the proprio row sees are not a contiguous valid prefix',
                2: 'image_text_proprio_mask: a valid-prefix row (or the proprio row\'s own key) does not have the prefix pattern of build_causal_mask_and_position_ids',
                4: 'action_mask: an action row does not see exactly {valid prefix, proprio, every action token}',
                8: 'image_text_proprio_mask (general_masks=True): an image / text row sees the proprio key -- the cached-prefix schedule computes the prefix rows before '
                   'the proprio token\'s K / V exist'}

    def __init__(self, cfg: VLAConfig, device='cuda', max_batch=1, use_graph=True, ride_proprio=True, naive_support=False, euler_opts=None, output_ring=0,
                 general_masks=False):
        L.lib()
        if not torch.cuda.is_available():
            raise L.VlaserHipError('vlaser_amd needs an MI355X (gfx950) GPU: there is no CPU fallback')
        self.cfg = cfg
        self.device = torch.device(device)
        self.max_image_text_tokens = cfg.max_image_text_tokens
        self.num_proprio_tokens = cfg.num_proprio_tokens
        self.num_action_tokens = cfg.num_action_tokens
        self.total_num_tokens = self.max_image_text_tokens + self.num_proprio_tokens + self.num_action_tokens
        self.num_inference_steps = cfg.num_inference_steps
        self.horizon_steps = cfg.num_action_tokens
        self.action_dim, self.proprio_dim = cfg.action_dim, cfg.proprio_dim
        self.final_action_clip_value = cfg.final_action_clip_value
        self.image_token_index = cfg.base.img_context_token_id
        self.pad_token_id = cfg.base.pad_token_id
        self.num_images = cfg.cond_steps
        self.max_batch = max_batch
        self.use_graph = use_graph
        self.naive_support = naive_support      # keep the expert's un-packed weights for infer_action_naive (tests)
        self.ride_proprio = ride_proprio        # batch 1: proprio row processed with the action rows of Euler step 0 (see _run)
        # general_masks=True (ABI 8, opt-in): the two dense additive masks of the call are SERVED as given -- any visibility pattern (left padding, holes, causal text)
        # and any finite bias -- by the VL_ATTN_DENSE variants of the attention kernels, as the reference's eager attention would (joint_model.py:636-656), instead of
        # being checked against the prefix + trailing-block pattern.  Slower (every key tile is walked with its mask values);
        # the default path and its timings are untouched.  Still refused: an image / text row that sees the proprio key (ERR_BITS[8]).
        self.general_masks = bool(general_masks)
        eo = os.environ.get('VLASER_EULER', self.EULER_DEFAULT) if euler_opts is None else euler_opts
        self.euler_opts = tuple(x for x in eo.split(',') if x and x != 'none')
        self._graphs = {}
        self._pos_state = None
        self._pos_next = None
        # output_ring = 0 (default): infer_action returns a FRESH tensor, as the reference does (pizero_internvl.py:934-936).  output_ring = n > 0 (opt-in, serving
        # loops / bench.py): it returns a VIEW of slot (call number mod n) of a small result ring written by the chunk's last kernel -- no copy launch, but the
        # view is overwritten n calls later
        self.output_ring = int(output_ring)
        self._calls = 0
        self._err_pending = []                  # (event, pinned int32[4] snapshot of call_ctr) of calls that passed dense masks and were not polled yet
        self._err_pins = None
        self._n_masked = 0
        if max_batch * cfg.num_action_tokens > 16:
            raise ValueError('the weight-streaming action path handles batch * horizon <= 16 rows')

    # ------------------------------------------------------------------ weights / workspace
    def load_checkpoint(self, path):
        """The reference's released `.pt` checkpoints: `torch.load(path)["model"]` (eval.py:196-212); aliases and `_orig_mod.`
        prefixes are canonicalised by load_state_dict."""
        data = torch.load(path, map_location='cpu', weights_only=True)
        return self.load_state_dict(data['model'] if isinstance(data, dict) and 'model' in data else data)

    def load_state_dict(self, sd, strict=True):
        sd = canonicalize_vla_state_dict(sd)
        cfg, dev = self.cfg, self.device
        base = cfg.base
        self.vit = VitEngine(sd, base, dev, max_tiles=self.max_batch * self.num_images)
        self.vlm = QwenStack(sd, 'language_model.', base.llm, dev, with_embed=True, with_head=True, gemm=True, skinny=False)   # lm_head (if present): infer_text only
        self._sd_expert = {k: v for k, v in sd.items() if k.startswith('action_expert.')} if self.naive_support else None
        self.expert_gemm = None
        self.expert = QwenStack(sd, 'action_expert.', cfg.expert, dev, with_embed=False, with_head=False, gemm=False, skinny=True, opts=self.euler_opts)
        g = lambda k: sd[k].to(device=dev, dtype=BF).contiguous()
        self.ae_w1, self.ae_b1 = g('action_encoder.linear_1.weight'), g('action_encoder.linear_1.bias')
        self.ae_w2, self.ae_b2 = ops.pack_skinny(g('action_encoder.linear_2.weight')), g('action_encoder.linear_2.bias')
        self.ae_w3, self.ae_b3 = ops.pack_skinny(g('action_encoder.linear_3.weight')), g('action_encoder.linear_3.bias')
        # 'glue1' (vlaser_vla_step): linear_1 and the time embedding folded into linear_2 (fp32 constants per Euler step), linear_3 as stored
        self.ae_w3_raw = g('action_encoder.linear_3.weight')
        self.ae_w21, self.ae_cs = ops.fold_action_encoder(self.ae_w1, self.ae_b1, g('action_encoder.linear_2.weight'), self.ae_b2, cfg.action_hidden_size,
                                                          cfg.action_dim, self.num_inference_steps, cfg.time_max_period)
        self.pe_w, self.pe_b = g('proprio_encoder.weight'), g('proprio_encoder.bias')
        self.ad_w, self.ad_b = g('action_decoder.weight'), g('action_decoder.bias')
        self._alloc()
        return self

    def _alloc(self):
        cfg, dev, B = self.cfg, self.device, self.max_batch
        self._pos_state = None                  # fresh (zeroed) position-id buffers
        self._pos_next = None
        llm = cfg.base.llm
        T = self.max_image_text_tokens
        self.s_max = (self.total_num_tokens + 63) // 64 * 64
        self.cache = KVCache(llm.num_hidden_layers, B, llm.num_key_value_heads, self.s_max, dev, llm.head_dim)
        self.pbuf = PrefillBuffers(self.vlm, B * T, dev)
        self.sb_pro = SkinnyBuffers(self.expert, 16, dev)
        self.sb_act = SkinnyBuffers(self.expert, 16, dev)
        self.rope = ops.rope_table(T + 16, llm.head_dim, llm.rope_theta, dev)
        W = cfg.action_hidden_size
        z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device=dev)
        self.h_vlm = z(B * T, llm.hidden_size)
        self.h_pro = z(16, W)
        self.xcat = z(16, 2 * W)
        self.e2 = z(16, W)
        self.h_act = z(16, W)
        self.h5 = z(16, W)                      # batch 1: [proprio row | action rows] of Euler step 0
        self.action5 = z(16, cfg.action_dim, dt=torch.float32)
        self.vel5 = z(16, cfg.action_dim, dt=torch.float32)
        self.vel_trace = z(cfg.num_inference_steps, 16, cfg.action_dim, dt=torch.float32)   # decoder output of every Euler step
        self.action = z(16, cfg.action_dim, dt=torch.float32)
        self.action_b = z(16, cfg.action_dim, dt=torch.float32)        # 'glue1': the actions ping-pong between two buffers (all workgroups read, one writes)
        self.rank_ws = z(B * T, dt=torch.int32)
        self.valid_len = z(B, dt=torch.int32)
        # static inputs of the captured graph
        self.in_ids = z(B, T, dt=torch.int64)
        self.in_pix = z(B * self.num_images, 3, cfg.base.vision.image_size, cfg.base.vision.image_size)
        self.in_proprio = z(B, cfg.proprio_dim, dt=torch.float32)
        self.in_noise = z(B * self.num_action_tokens, cfg.action_dim, dt=torch.float32)
        self.pos_vlm = z(B * T, dt=torch.int32)
        self.pos_pro = z(B, dt=torch.int32)
        self.pos_act = z(B * self.num_action_tokens, dt=torch.int32)
        self.pos5 = z(16, dt=torch.int32)
        self.call_ctr = z(4, dt=torch.int32)    # {call number, error word of even calls, error word of odd calls, pad}: vlaser_vla_stage / vlaser_vla_euler
        self.out_ring = z(max(self.output_ring, 4), 16 * cfg.action_dim, dt=torch.float32)      # >= 4 slots: `output_ring` may be switched on later without re-capturing the graph
        # general masks: fp32 [B, T + 1 + na, ld] -- rows 0..T image_text_proprio_mask, rows T+1.. action_mask (written by the staging launch, read by the dense-mask attention)
        self.mask_slot = (torch.full((B, self.total_num_tokens, (self.total_num_tokens + 63) // 64 * 64), -3.0e38, dtype=torch.float32, device=dev)
                          if self.general_masks else None)
        self._calls = 0
        self._err_pending = []
        self._pos_defaults = {}
        # where the per-call noise is staged: straight into the buffer the first launch of the Euler phase reads ('glue1': the ping-pong buffer the
        # integration starts from -- r03 copied in_noise there inside the graph)
        n = cfg.num_inference_steps
        self._glue1 = 'glue1' in self.euler_opts and n >= 2 and W % 256 == 0 and W <= (1024 if cfg.action_dim <= 8 else 768)
        self.noise_dst = (self.action, self.action_b)[(n - 1) % 2] if self._glue1 else self.in_noise

    def _positions_for_stage(self, B, vlm_position_ids, proprio_position_ids, action_position_ids):
        """The three position-id tensors the staging launch has to write this call, as int64 DEVICE tensors, or None when the slots already hold what is
        asked for.  The reference passes them on every call (eval.py:110-128; values 1..T, 1, 2..1+na: pizero_internvl.py:576-585): they are converted
        int64 -> int32 slots by `vlaser_vla_stage` itself -- r04 issued five small host-staged copies per call whenever they were given.  With all three
        None the defaults are written once per batch size."""
        T, na, dev = self.max_image_text_tokens, self.num_action_tokens, self.device
        custom = not (vlm_position_ids is None and proprio_position_ids is None and action_position_ids is None)
        self._pos_next = None
        if not custom and self._pos_state == ('default', B):
            return None
        d = self._pos_defaults.get(B)
        if d is None:
            d = self._pos_defaults[B] = (torch.arange(1, T + 1, device=dev).repeat(B, 1), torch.ones(B, 1, dtype=torch.long, device=dev),
                                         torch.arange(2, 2 + na, device=dev).repeat(B, 1))
        on = lambda t, dflt: dflt if t is None else t.to(device=dev, dtype=torch.int64, non_blocking=True).contiguous()
        # the state the slots WILL hold once the staging launch has been accepted: the caller commits it (`_pos_commit`) behind that launch -- a call that raises
        # between here and there (bad proprio / mask shape ...) must not leave a retry believing the slots are written
        self._pos_next = ('custom', B) if custom else ('default', B)
        return on(vlm_position_ids, d[0]), on(proprio_position_ids, d[1]), on(action_position_ids, d[2])

    def _pos_commit(self):
        if self._pos_next is not None:
            self._pos_state, self._pos_next = self._pos_next, None

    # ------------------------------------------------------------------ reference helpers (API parity)
    def build_causal_mask_and_position_ids(self, attention_mask, dtype):
        return prep.build_causal_mask_and_position_ids(attention_mask, dtype, self.max_image_text_tokens, self.num_proprio_tokens,
                                                       self.num_action_tokens)

    def split_full_mask_into_submasks(self, causal_mask):
        return prep.split_full_mask_into_submasks(causal_mask, self.max_image_text_tokens, self.num_proprio_tokens,
                                                  self.num_action_tokens)

    def build_mixture_caches(self):
        return self.cache

    # ------------------------------------------------------------------ the hot path (all kernel launches)
    def _ride(self, B):
        """Batch 1: the proprio row rides with the action rows of Euler step 0 (see _run_prefill).  Needs a second Euler step: the step that hosts the
        proprio row integrates M + 1 rows in `action5` and does not write the result ring, so with num_inference_steps == 1 the caller's slot (and the
        NaN poisoning of an unsupported mask) would never be written -- the proprio row takes its own pass then."""
        return self.ride_proprio and B == 1 and self.num_inference_steps >= 2      # (general masks too: rows T .. T + na of the mask slot ARE the riding launch's rows)

    def _run(self, B):
        """ViT + projector + scatter -> joint prefill -> Euler loop: three phases, separately callable so that bench.py can time each
        one from its own HIP graph (per-phase ms on the JSON line)."""
        self._run_vit(B)
        self._run_prefill(B)
        self._run_euler(B)

    def _run_vit(self, B):
        # a1-a7, a12: ViT tiles -> projector -> scatter into the (zero-padded) text embeddings
        T = self.max_image_text_tokens
        feats = self.vit.forward(self.in_pix[:B * self.num_images])
        ops.embed_merge(self.in_ids[:B], self.vlm.embed, feats, self.h_vlm[:B * T], self.image_token_index, self.pad_token_id, True, self.rank_ws)

    def _run_prefill(self, B):
        cfg = self.cfg
        llm = cfg.base.llm
        T, na = self.max_image_text_tokens, self.num_action_tokens
        nL = llm.num_hidden_layers
        h_vlm = self.h_vlm[:B * T]
        # a13: joint prefill over {vlm, proprio}; K/V of both mixtures cached (post-RoPE), last layer skips o_proj+MLP.
        # Batch 1: the proprio token does NOT get its own pass through the expert (28 x 5 weight-streaming launches, ~0.85 ms of pure
        # latency): it rides in front of the 4 action rows of Euler step 0 (M = 5).  Its rows of the block mask (prefix + itself,
        # pizero_internvl.py:517-587) only differ from the action rows' in the block keys they see (first_tok_kv_len), no row looks at
        # a later row's output, and the per-row arithmetic of the <= 16-row kernels does not depend on M -- so its K / V^T (slot T)
        # and every action are bit-identical to the separate pass, and the expert's weights are streamed once for both.
        # (An HIP-graph side branch was tried first: graph replay runs the branches back to back, 0.9 % kernel overlap in rocprof.)
        ride = self._ride(B)
        if ride:
            ops.small_linear(self.in_proprio, self.pe_w, self.pe_b, self.h5, B, cfg.action_hidden_size, cfg.proprio_dim)      # row 0
        else:
            ops.small_linear(self.in_proprio, self.pe_w, self.pe_b, self.h_pro, B, cfg.action_hidden_size, cfg.proprio_dim)
        h_pro, parts, npart = self.h_pro, None, 0
        # general masks: rows 0..T-1 of the slot for the image / text rows (keys 0..T-1), row T for the proprio token (keys 0..T)
        dm_vlm = self.mask_slot[:B, :T] if self.general_masks else None
        dm_pro = self.mask_slot[:B, T:T + 1] if self.general_masks else None
        prefill_begin(self.vlm, self.pbuf, h_vlm, B * T)
        for i in range(nL):
            last = i == nL - 1
            prefill_layer(self.vlm, self.vlm.layers[i], self.pbuf, h_vlm, self.cache, i, self.rope, self.pos_vlm, B, T,
                          L.ATTN_PREFIX, valid_len=self.valid_len, blk_start=T, skip_post_attn=last,
                          next_norm_w=None if last else self.vlm.layers[i + 1].ln_in, dense_mask=dm_vlm)
            if not ride:
                h_pro, parts, npart = skinny_layer(self.expert, self.expert.layers[i], self.sb_pro, h_pro, parts, npart, self.cache, i,
                                                   self.rope, self.pos_pro, B, 1, T, T + 1, L.ATTN_PREFIX, valid_len=self.valid_len,
                                                   blk_start=T, skip_post_attn=last, dense_mask=dm_pro)

    def _run_euler(self, B, skip=()):
        """a14: flow-matching Euler integration over the cached prefix.  `skip`: names of per-layer launches left out (bench.py's in-chain
        timing of one kernel = chain with it minus chain without it; the values are garbage then)."""
        cfg = self.cfg
        llm, ex = cfg.base.llm, cfg.expert
        T, na = self.max_image_text_tokens, self.num_action_tokens
        nL = llm.num_hidden_layers
        ride = self._ride(B)
        M = B * na
        n = self.num_inference_steps
        dt = 1.0 / n
        W = cfg.action_hidden_size
        clip = self.final_action_clip_value
        if self._glue1:
            return self._run_euler_glue1(B, skip)
        if ride:
            self.action5[1:1 + M].copy_(self.in_noise[:M])
        else:
            self.action[:M].copy_(self.in_noise[:M])
        for s in range(n):
            t = s * dt
            first = ride and s == 0
            ops.vla_prep(self.action5[1:1 + M] if first else self.action, self.ae_w1, self.ae_b1, self.xcat, M, W, cfg.action_dim, t, cfg.time_max_period)
            ops.skinny(L.PRO_PLAIN, L.SK_BIAS_SILU, self.xcat, self.ae_w2, M, out=self.e2, ldo=W, bias=self.ae_b2)
            ops.skinny(L.PRO_PLAIN, L.SK_BIAS, self.e2, self.ae_w3, M, out=self.h5[1:1 + M] if first else self.h_act, ldo=W, bias=self.ae_b3)
            if first:
                h, parts, npart = self.h5, None, 0
                for i in range(nL):
                    h, parts, npart = skinny_layer(self.expert, self.expert.layers[i], self.sb_pro, h, parts, npart, self.cache, i, self.rope,
                                                   self.pos5, B, na + 1, T, T + 1 + na, L.ATTN_PREFIX, valid_len=self.valid_len, blk_start=T,
                                                   first_tok_kv_len=0 if self.general_masks else T + 1, skip=skip,
                                                   dense_mask=self.mask_slot[:B, T:] if self.general_masks else None)      # (dense: the proprio row's own mask row hides the action keys)
                ops.vla_euler(h, parts, npart, M + 1, self.expert.norm, ex.rms_norm_eps, self.ad_w, self.ad_b, self.action5, W, cfg.action_dim, dt,
                              clip if clip is not None else 0.0, clip is not None and s == n - 1, vel_out=self.vel5, method=cfg.integration_method)
                self.action[:M].copy_(self.action5[1:1 + M])          # row 0 of action5 (the proprio row's "velocity") is scratch
                self.vel_trace[s, :M].copy_(self.vel5[1:1 + M])
                continue
            h, parts, npart = self.h_act, None, 0
            for i in range(nL):
                h, parts, npart = skinny_layer(self.expert, self.expert.layers[i], self.sb_act, h, parts, npart, self.cache, i, self.rope,
                                               self.pos_act, B, na, T + 1, T + 1 + na, L.ATTN_PREFIX, valid_len=self.valid_len,
                                               blk_start=T, skip=skip, dense_mask=self.mask_slot[:B, T + 1:] if self.general_masks else None)
            ring = (self.out_ring, self.call_ctr) if s == n - 1 else (None, None)
            ops.vla_euler(h, parts, npart, M, self.expert.norm, ex.rms_norm_eps, self.ad_w, self.ad_b, self.action, W, cfg.action_dim, dt,
                          clip if clip is not None else 0.0, clip is not None and s == n - 1, vel_out=self.vel_trace[s], ring=ring[0], ring_ctr=ring[1], method=cfg.integration_method)

    def _run_euler_glue1(self, B, skip=()):
        """The same integration with ONE launch between two passes through the expert's layers (`vlaser_vla_step`: tail of step s-1 + action
        encoder of step s) instead of four (vla_euler, vla_prep, linear_2 + swish, linear_3): 30 launches of ~5 us less per chunk.  linear_1 and the
        time embedding are folded into linear_2 at load time, so the encoder output differs from the 4-launch path in the last bf16 bit of a few
        elements (no longer bit-identical to `VLASER_EULER=qkv16`; same tolerance against the oracle and the reference goldens)."""
        cfg = self.cfg
        llm, ex = cfg.base.llm, cfg.expert
        T, na = self.max_image_text_tokens, self.num_action_tokens
        nL = llm.num_hidden_layers
        ride = self._ride(B)
        M, n, W, ad = B * na, self.num_inference_steps, cfg.action_hidden_size, cfg.action_dim
        dt = 1.0 / n
        clip = self.final_action_clip_value
        acts = [self.action, self.action_b]
        p = (n - 1) % 2                           # n - 1 ping-pongs later the actions sit in self.action, where the last step finishes in place
        assert acts[p] is self.noise_dst          # the staging launch wrote this call's noise here
        fin = None
        for s in range(n):
            first = ride and s == 0
            h_enc = self.h5[1:1 + M] if first else self.h_act
            if fin is None:
                ops.vla_step(acts[p], acts[p], self.ae_w21, self.ae_cs[s], self.ae_w3_raw, self.ae_b3, h_enc, M, W, ad)
            else:
                ops.vla_step(acts[p], acts[1 - p], self.ae_w21, self.ae_cs[s], self.ae_w3_raw, self.ae_b3, h_enc, M, W, ad, finish=fin,
                             vel_out=self.vel_trace[s - 1], dt=dt, method=cfg.integration_method)
                p = 1 - p
            if first:
                h, parts, npart = self.h5, None, 0
                for i in range(nL):
                    h, parts, npart = skinny_layer(self.expert, self.expert.layers[i], self.sb_pro, h, parts, npart, self.cache, i, self.rope,
                                                   self.pos5, B, na + 1, T, T + 1 + na, L.ATTN_PREFIX, valid_len=self.valid_len, blk_start=T,
                                                   first_tok_kv_len=0 if self.general_masks else T + 1, skip=skip,
                                                   dense_mask=self.mask_slot[:B, T:] if self.general_masks else None)      # (dense: the proprio row's own mask row hides the action keys)
                fin = (h, parts, npart, M + 1, 1, self.expert.norm, ex.rms_norm_eps, self.ad_w, self.ad_b)       # row 0 = the proprio row: skipped
            else:
                h, parts, npart = self.h_act, None, 0
                for i in range(nL):
                    h, parts, npart = skinny_layer(self.expert, self.expert.layers[i], self.sb_act, h, parts, npart, self.cache, i, self.rope,
                                                   self.pos_act, B, na, T + 1, T + 1 + na, L.ATTN_PREFIX, valid_len=self.valid_len,
                                                   blk_start=T, skip=skip, dense_mask=self.mask_slot[:B, T + 1:] if self.general_masks else None)
                fin = (h, parts, npart, M, 0, self.expert.norm, ex.rms_norm_eps, self.ad_w, self.ad_b)
        assert acts[p] is self.action
        ring = (self.out_ring, self.call_ctr)     # always: the caller's copy (a 1-slot ring when output_ring == 0) is where an unsupported mask turns into NaN
        ops.vla_euler(fin[0], fin[1], fin[2], M, self.expert.norm, ex.rms_norm_eps, self.ad_w, self.ad_b, self.action, W, ad, dt,
                      clip if clip is not None else 0.0, clip is not None, vel_out=self.vel_trace[n - 1], ring=ring[0], ring_ctr=ring[1], method=cfg.integration_method)

    @torch.no_grad()
    def infer_action(self, input_ids, pixel_values, image_text_proprio_mask=None, action_mask=None, vlm_position_ids=None,
                     proprio_position_ids=None, action_position_ids=None, proprios=None, noise=None, generator=None,
                     valid_len=None):
        """pizero_internvl.py:798-936.  The dense masks are checked ON THE DEVICE against the only visibility the kernels express (valid prefix + trailing
        block); the call never waits for that check.  An unsupported mask turns THIS call's chunk into NaN and raises `ValueError` at the next poll: bringing
        the returned chunk to the host (`.cpu()`, `.tolist()` ...), the next public method of this object, `last_velocities()` or `check_errors()`.  A NaN
        chunk therefore always means "call check_errors()"."""
        cfg, dev = self.cfg, self.device
        B = pixel_values.shape[0] // self.num_images
        T, na = self.max_image_text_tokens, self.num_action_tokens
        if B > self.max_batch:
            # larger batches run as consecutive groups of max_batch observations (<= 16 action rows per weight-streaming launch)
            outs, mb, ni = [], self.max_batch, self.num_images
            sl = lambda t, lo, hi, k=1: None if t is None else t[lo * k:hi * k]
            for lo in range(0, B, mb):
                hi = min(B, lo + mb)
                o = self.infer_action(input_ids[lo:hi], pixel_values[lo * ni:hi * ni], sl(image_text_proprio_mask, lo, hi),
                                      sl(action_mask, lo, hi), sl(vlm_position_ids, lo, hi), sl(proprio_position_ids, lo, hi),
                                      sl(action_position_ids, lo, hi), sl(proprios, lo, hi), sl(noise, lo, hi), generator,
                                      sl(valid_len, lo, hi))
                # with output_ring > 0 a group's result is a VIEW of a ring slot that a later group of this same call may overwrite (more groups than
                # slots): take the copy now, in stream order
                outs.append(o.clone() if self.output_ring > 0 else o)
            return torch.cat(outs, 0)
        if input_ids.shape != (B, T):
            raise ValueError(f'input_ids must be [B,{T}] (right-padded with pad_token_id), got {tuple(input_ids.shape)}')
        # ---- stage inputs into the static slots of the captured graph: ONE launch (vlaser_vla_stage) once the tensors are on the device.  The reference's
        # dense masks are NOT copied to the host: valid_len is counted from the mask's proprio row on the device and both masks are checked there against the
        # pattern the kernels' (valid_len, blk_start) descriptors express (an unsupported mask -> NaN result + ValueError at the next poll, never a silent
        # mis-service: `action_mask` was accepted and ignored until r04)
        self._poll_errors(block=False)
        if noise is None:
            noise = torch.randn((B, na, cfg.action_dim), generator=generator)      # reference: torch.randn inside (:879-881)
        masks = None if (image_text_proprio_mask is None and action_mask is None) else (image_text_proprio_mask, action_mask)
        if self.general_masks and (image_text_proprio_mask is None or action_mask is None):
            raise ValueError('PiZero(general_masks=True): infer_action needs image_text_proprio_mask AND action_mask (they ARE the visibility in this mode)')
        positions = self._positions_for_stage(B, vlm_position_ids, proprio_position_ids, action_position_ids)
        self._stage_inputs(B, input_ids, pixel_values, proprios, noise, valid_len, masks, positions)
        # ---- run (HIP graph replay after the first call per batch size)
        if self.use_graph:
            gs = self._graphs.get(B)
            if gs is None:
                self._run(B)                      # warm-up: sets kernel attributes, touches every buffer
                torch.cuda.synchronize()
                # VLASER_GRAPH_SPLIT=k: the chunk as k HIP graphs replayed back to back (1 = one graph of ~1 780 kernel nodes; 3 = ViT | joint prefill | Euler phase;
                # 3 + j: the Euler phase in j + 1 pieces of whole Euler steps) -- see profiles/r04*_graph_split.md
                nsplit = int(os.environ.get('VLASER_GRAPH_SPLIT', '1'))
                if nsplit <= 1:
                    parts = [lambda: self._run(B)]
                else:
                    parts = [lambda: self._run_vit(B), lambda: self._run_prefill(B), lambda: self._run_euler(B)]
                gs = []
                for fn in parts:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        fn()
                    gs.append(g)
                self._graphs[B] = gs
                # the warm-up run integrated the staged noise away (it is staged straight into the Euler phase's start buffer, and the graph holds no
                # copy of it any more): stage this call's inputs once more in front of the first replay
                self._stage_inputs(B, input_ids, pixel_values, proprios, noise, valid_len, masks, positions)
            for g in gs:
                g.replay()
        else:
            self._run(B)
        if masks is not None:
            # snapshot of {call number, error words} behind the chunk, into pinned host memory: polled at the NEXT call / by check_errors(), never waited for here
            if self._err_pins is None:
                self._err_pins = [torch.empty(4, dtype=torch.int32).pin_memory() for _ in range(12)]      # > 8 outstanding + the ones being read
            pin = self._err_pins[self._n_masked % len(self._err_pins)]      # (a counter of MASKED calls: at most 8 + the ones being read are ever outstanding)
            self._n_masked += 1
            pin.copy_(self.call_ctr, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._err_pending.append((ev, pin))
            if len(self._err_pending) > 8:        # a host running more than 8 calls ahead of the device waits for the oldest one
                self._poll_errors(block=False, at_most_pending=8)
        slot = self._calls % self.out_ring.shape[0]
        act = self.out_ring[slot, :B * na * cfg.action_dim].view(B, na, cfg.action_dim)[:, -cfg.horizon_steps:]
        act = act if self.output_ring > 0 else act.clone()
        if masks is not None:
            act = act.as_subclass(_ActionChunk)   # bringing it to the host polls the deferred mask check (see _ActionChunk)
            act._vl_owner = self
        return act

    def _poll_errors(self, block, at_most_pending=0):
        """Raise for a finished call whose dense masks the kernels cannot honour (error word written by `vlaser_vla_stage`).  block=False looks only at calls
        the device has completed (no synchronisation), except that it waits until at most `at_most_pending` snapshots are outstanding."""
        keep = []
        pend, self._err_pending = self._err_pending, []
        n_wait = len(pend) - at_most_pending if at_most_pending else 0
        bad = None
        for i, (ev, pin) in enumerate(pend):
            if block or i < n_wait:
                ev.synchronize()
            elif not ev.query():
                keep.append((ev, pin))
                continue
            k = int(pin[0])
            word = int(pin[1 + (k & 1)])
            if word and bad is None:
                bad = (k, word)
        self._err_pending = keep
        if bad is not None:
            why = '; '.join(t for b_, t in self.ERR_BITS.items() if bad[1] & b_)
            if self.general_masks:
                raise ValueError(f'infer_action call #{bad[0]}: {why}; the chunk that call returned is all NaN (a NaN chunk always means: call check_errors())')
            raise ValueError(f'infer_action call #{bad[0]}: the dense masks are not the prefix + trailing-block pattern of build_causal_mask_and_position_ids '
                             f'(pizero_internvl.py:517-603) -- the only visibility the kernels\' (valid_len, blk_start) descriptors express; the chunk that call '
                             f'returned is all NaN (a NaN chunk always means: call check_errors()).  PiZero(general_masks=True) serves arbitrary additive masks.  {why}')

    def check_errors(self):
        """Wait for every outstanding call and raise if one of them passed a mask the kernels cannot honour (the lazy check of infer_action, made now)."""
        self._poll_errors(block=True)

    def _stage_inputs(self, B, input_ids, pixel_values, proprios, noise, valid_len, masks=None, positions=None):
        """Host / device plumbing of one call's inputs (the reference moves them with .to(device) in its agent loop, eval.py:117-130): anything not yet
        on the device is copied there, then `vlaser_vla_stage` writes every slot in one launch -- ids, valid_len (given, or the zero count of the dense
        mask's proprio row, or counted from the pad ids), proprio, noise (straight into the buffer the Euler phase starts from), position ids, the pixels
        (bf16 / fp32 -> bf16, or the raw uint8 observation normalised on the device) -- checks the dense masks and sets the call number that selects the result
        slot."""
        dev, cfg = self.device, self.cfg
        na = self.num_action_tokens
        on = lambda t, dt=None: t.to(device=dev, dtype=dt, non_blocking=True).contiguous()
        ids = on(input_ids, torch.int64)
        pv = pixel_values if pixel_values.dtype in (torch.uint8, torch.float32, BF) else pixel_values.float()
        pv = on(pv)
        if tuple(pv.shape[-3:]) != tuple(self.in_pix.shape[-3:]) or pv.numel() != B * self.num_images * self.in_pix[0].numel():
            raise ValueError(f'pixel_values must be [B*{self.num_images},{",".join(map(str, self.in_pix.shape[1:]))}], got {tuple(pixel_values.shape)}')
        if proprios is None:
            raise ValueError('infer_action: proprios [B,1,proprio_dim] is required')
        if proprios.numel() != B * self.num_proprio_tokens * cfg.proprio_dim or noise.numel() != B * na * cfg.action_dim:
            # (the staging launch copies element counts it is given: a wrong proprio_dim / action_dim / horizon must not become an out-of-bounds device write)
            raise ValueError(f'proprios must be [B,{self.num_proprio_tokens},{cfg.proprio_dim}] and noise [B,{na},{cfg.action_dim}], got {tuple(proprios.shape)} / {tuple(noise.shape)}')
        pro = on(proprios.reshape(B, -1), torch.float32)
        nz = on(noise.reshape(B * na, -1), torch.float32)
        if valid_len is not None:
            valid_len = on(valid_len.reshape(-1), valid_len.dtype if valid_len.dtype in (torch.int32, torch.int64) else torch.int64)
            if valid_len.numel() != B:
                raise ValueError(f'valid_len must have {B} entries, got {valid_len.numel()}')
        if masks is not None:
            # the masks stay where they are (any batch / row stride: the reference hands out slices of the full [B,1,L,L] mask, :589-603); only a
            # non-unit innermost stride or a host tensor costs a copy
            def fix(m):
                if m is None:
                    return None
                m = m.to(dev, non_blocking=True)
                return m if m.stride(-1) == 1 else m.contiguous()
            masks = tuple(fix(m) for m in masks)
        k = self._calls + 1
        ops.vla_stage(ids, self.in_ids[:B], valid_len, self.valid_len, pro, self.in_proprio, nz, self.noise_dst, pv, self.in_pix, self.pad_token_id,
                      prep.VLA_MEAN, prep.VLA_STD, call_ctr=self.call_ctr, call_no=k, masks=masks, n_act=na, positions=positions,
                      pos_out=(self.pos_vlm, self.pos_pro, self.pos_act, self.pos5 if B == 1 else None),
                      mask_slot=self.mask_slot[:B] if (self.general_masks and masks is not None) else None)
        self._calls = k                           # only after the launch was accepted: the host's ring index cannot run ahead of the device's call number
        if positions is not None:
            self._pos_commit()                    # likewise: the slots hold the new position ids only now

    def last_velocities(self, B=1):
        """Decoder output (velocity) of every Euler step of the last infer_action call: fp32 [n_steps, B, horizon, action_dim]
        (the `action_vel` of pizero_internvl.py:911; golden G7b pins it per step).  Waits for that call and raises its deferred mask error, if any."""
        self._poll_errors(block=True)
        na = self.num_action_tokens
        return self.vel_trace[:, :B * na].view(self.num_inference_steps, B, na, -1).clone()

    @torch.no_grad()
    def infer_action_naive(self, input_ids, pixel_values, causal_mask=None, vlm_position_ids=None, proprio_position_ids=None,
                           action_position_ids=None, proprios=None, noise=None, generator=None, valid_len=None):
        """Mirror of `PiZero.infer_action_naive` (pizero_internvl.py:938-1003): no reuse of cached keys -- EVERY Euler step
        re-runs the joint pass over {vlm, proprio, action} (`cache_mode="no_append"`).  It is a self-consistency surface, not a hot
        path, and it deliberately takes the OTHER kernels: the 5 expert rows go through the MFMA GEMM kernels + the prefill
        attention kernel (block mask as two PREFIX launches) instead of the weight-streaming <= 16-row kernels + the key-split
        attention -- so `infer_action == infer_action_naive` cross-checks the two implementations (the reference remarks ~1e-3 in
        bf16, none in fp32: eval.py:131-137).  Batch 1."""
        self._poll_errors(block=False)
        cfg, dev = self.cfg, self.device
        base, llm, ex = cfg.base, cfg.base.llm, cfg.expert
        T, na, nL = self.max_image_text_tokens, self.num_action_tokens, llm.num_hidden_layers
        B = pixel_values.shape[0] // self.num_images
        if B != 1:
            raise NotImplementedError('infer_action_naive: batch 1')
        if self._sd_expert is None:
            raise RuntimeError('construct PiZero(..., naive_support=True) to keep the weights infer_action_naive needs')
        if self.expert_gemm is None:
            self.expert_gemm = QwenStack(self._sd_expert, 'action_expert.', ex, dev, with_embed=False, with_head=False, gemm=True, skinny=False)
            self.ebuf = PrefillBuffers(self.expert_gemm, 16, dev)
        # inputs exactly as infer_action stages them
        self.in_ids[:1].copy_(input_ids)
        stage_pixels(pixel_values, self.in_pix[:self.num_images], dev)
        self.in_proprio[:1].copy_(proprios.reshape(1, -1).to(torch.float32))
        if valid_len is None:
            valid_len = prep.mask_to_descriptor(causal_mask[:, :, :T + 1, :T + 1].to('cpu'), T) if causal_mask is not None else (input_ids != self.pad_token_id).sum(-1)
        self.valid_len[:1].copy_(valid_len.to(torch.int32))
        pos = self._positions_for_stage(1, vlm_position_ids, proprio_position_ids, action_position_ids)
        if pos is not None:                       # (test surface: plain torch copies; the hot path converts them inside vlaser_vla_stage)
            self.pos_vlm[:T].copy_(pos[0].reshape(-1)); self.pos_pro[:1].copy_(pos[1].reshape(-1)); self.pos_act[:na].copy_(pos[2].reshape(-1))
            self.pos5[:1].copy_(self.pos_pro[:1]); self.pos5[1:1 + na].copy_(self.pos_act[:na])
            self._pos_commit()
        if noise is None:
            noise = torch.randn((1, na, cfg.action_dim), generator=generator)
        self.action[:na].copy_(noise.reshape(na, -1).to(torch.float32))
        feats = self.vit.forward(self.in_pix[:self.num_images])
        W, n = cfg.action_hidden_size, self.num_inference_steps
        dt = 1.0 / n
        clip = self.final_action_clip_value
        eg, eb = self.expert_gemm, self.ebuf
        for s in range(n):
            h_vlm = self.h_vlm[:T]
            ops.embed_merge(self.in_ids[:1], self.vlm.embed, feats, h_vlm, self.image_token_index, self.pad_token_id, True, self.rank_ws)
            ops.small_linear(self.in_proprio, self.pe_w, self.pe_b, self.h_pro, 1, W, cfg.proprio_dim)
            ops.vla_prep(self.action, self.ae_w1, self.ae_b1, self.xcat, na, W, cfg.action_dim, s * dt, cfg.time_max_period)
            ops.skinny(L.PRO_PLAIN, L.SK_BIAS_SILU, self.xcat, self.ae_w2, na, out=self.e2, ldo=W, bias=self.ae_b2)
            ops.skinny(L.PRO_PLAIN, L.SK_BIAS, self.e2, self.ae_w3, na, out=self.h_act, ldo=W, bias=self.ae_b3)
            h_pro, h_act = self.h_pro[:1], self.h_act[:na]
            prefill_begin(self.vlm, self.pbuf, h_vlm, T)
            for i in range(nL):
                last = i == nL - 1
                prefill_layer(self.vlm, self.vlm.layers[i], self.pbuf, h_vlm, self.cache, i, self.rope, self.pos_vlm, 1, T, L.ATTN_PREFIX,
                              valid_len=self.valid_len, blk_start=T, skip_post_attn=last, next_norm_w=None if last else self.vlm.layers[i + 1].ln_in)
                lw = eg.layers[i]
                # proprio row: prefix + itself (slot T); action rows: prefix + proprio + all action rows (slots T+1..T+na)
                ops.rmsnorm(h_pro, lw.ln_in, ex.rms_norm_eps, out=eb.x[:1])
                prefill_layer(eg, lw, eb, h_pro, self.cache, i, self.rope, self.pos_pro, 1, 1, L.ATTN_PREFIX, valid_len=self.valid_len, blk_start=T,
                              kv_len=T + 1, skip_post_attn=last, slot_base=T)
                ops.rmsnorm(h_act, lw.ln_in, ex.rms_norm_eps, out=eb.x[:na])
                prefill_layer(eg, lw, eb, h_act, self.cache, i, self.rope, self.pos_act, 1, na, L.ATTN_PREFIX, valid_len=self.valid_len, blk_start=T,
                              kv_len=T + 1 + na, slot_base=T + 1)
            ops.vla_euler(h_act, None, 0, na, eg.norm, ex.rms_norm_eps, self.ad_w, self.ad_b, self.action, W, cfg.action_dim, dt,
                          clip if clip is not None else 0.0, clip is not None and s == n - 1, vel_out=self.vel_trace[s], method=cfg.integration_method)
        return self.action[:na].view(1, na, cfg.action_dim).clone()

    @torch.no_grad()
    def infer_text(self, input_ids, pixel_values, attention_mask=None, kv_cache=None):
        """Mirror of `PiZero.infer_text` (pizero_internvl.py:1005-1046): the VLM mixture ALONE through the joint model (causal mask
        of :645-702, 0-based positions `cumsum(mask) - 1`, `final_layer_post_attn_skip_names=[]`) + lm_head -> {'logits': fp32
        [B, S, V]}; it must reproduce InternVLChatModel's logits (same weights, VLA-style embedding assembly with zeroed pad
        rows).  Prefill only (kv_cache must be None), no padding inside the sequence (the reference assumes the same)."""
        self._poll_errors(block=False)
        if kv_cache is not None:
            raise NotImplementedError('infer_text: prefill only (kv_cache=None)')
        if self.vlm.head is None:
            raise ValueError('infer_text needs language_model.lm_head.weight in the checkpoint')
        cfg, dev, llm = self.cfg, self.device, self.cfg.base.llm
        B, S = input_ids.shape
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            raise NotImplementedError('infer_text: padded prompts are not supported (the reference assumes no padding, :655)')
        if B != 1 or S > self.max_image_text_tokens:
            raise NotImplementedError(f'infer_text: batch 1, at most {self.max_image_text_tokens} tokens')
        pvb = stage_pixels(pixel_values, torch.empty(pixel_values.shape, dtype=BF, device=dev), dev)
        feats = self.vit.forward(pvb)
        ids = input_ids.to(dev).contiguous()
        h = self.h_vlm[:S]
        ops.embed_merge(ids, self.vlm.embed, feats, h, self.image_token_index, self.pad_token_id, True, self.rank_ws)
        pos = torch.arange(S, dtype=torch.int32, device=dev)
        layers = self.vlm.layers
        prefill_begin(self.vlm, self.pbuf, h, S)
        for i, lw in enumerate(layers):
            nxt = layers[i + 1].ln_in if i + 1 < len(layers) else self.vlm.norm
            prefill_layer(self.vlm, lw, self.pbuf, h, self.cache, i, self.rope, pos, 1, S, L.ATTN_CAUSAL, next_norm_w=nxt)
        logits = torch.empty(S, llm.vocab_size, dtype=torch.float32, device=dev)
        ops.gemm(L.EPI_F32, self.pbuf.x[:S], self.vlm.head, out=logits)
        return {'logits': logits.view(1, S, -1)}

    def forward(self, *args, **kw):
        return self.infer_action(*args, **kw)

    __call__ = forward

    def eval(self):
        return self


class PiZeroInference(PiZero):
    """pizero_internvl.py:1286-1307."""
    pass
