"""Flow-matching training step of the Vlaser-VLA action expert on MI355X.

Mirrors the step semantics of the reference's VLA trainer without its control plane (hydra / TF-RLDS pipeline / wandb):

  * loss      = `PiZero.forward` (pizero_internvl.py:1064-1197): x0 ~ N(0, I), psi_t = (1 - (1 - sig_min) t) x0 + t x1 (:1050-1062),
                ONE joint pass over {vlm, proprio, action} with the block mask of :517-587 and no KV cache, last-layer post-attention
                skipped for the proprio row, loss = mean((action_decoder(h_action) - (x1 - (1 - sig_min) x0))^2);
  * t         = `sample_fm_time` (train.py:335-343): Beta(1.5, 1) flipped and scaled by 0.999 ("beta") or stratified uniform;
  * trained   = the reference's default parameter group `action_expert_parameters` (pizero_internvl.py:358-374; `train_vlm: False`,
                train.py:246-255): action / proprio encoders, action decoder, the expert's 28 decoder layers + final norm (proprio and
                action mixtures share them, :508-510).  The VLM is frozen, so its rows are a constant prefix: they run ONCE through the
                inference prefill kernels and only their K / V^T enter the expert rows' attention -- the joint pass restricted to what
                the loss and the trained gradients depend on;
  * step      = gradient accumulation (`no_sync`, train.py:470-482), `clip_grad_norm_` (:504-507), AdamW per group with
                `CosineAnnealingWarmupRestarts` (optim.py:31-160), DDP-mean gradients (here: the ZeRO-1 bucketed reduce-scatter /
                all-gather of vlaser_amd/dp.py).  The reference uses bitsandbytes' 8-bit AdamW; this build keeps fp32 moments.

The 5 expert rows (proprio + 4 action tokens) go through the MFMA GEMM kernels with every intermediate saved; their backward reuses
the SFT kernels (dgrad on resident W^T, TN wgrad, RMSNorm / SwiGLU / RoPE backward) plus `vlaser_attn_rows_bwd` for the joint
attention.  All arithmetic is HIP; torch holds buffers, streams and RCCL.
"""
import math
import os
from types import SimpleNamespace

import torch

from . import _lib as L
from . import dp, ops, prep
from .config import VLAConfig
from .engine import BF, KVCache, PrefillBuffers, QwenStack, VitEngine, prefill_begin, prefill_layer
from .pizero import canonicalize_vla_state_dict, stage_pixels
from .sft import FlatParams

F32 = torch.float32


def sample_fm_time(bsz, flow_sampling='beta', alpha=1.5, beta=1.0, t_max=0.999, generator=None):
    """train.py:335-343 (+ :317-323 for the Beta parameters): 'beta' -> t = t_max * (1 - z), z ~ Beta(alpha, beta);
    'uniform' -> stratified (rand + arange/bsz) mod (1 - 1e-5)."""
    if flow_sampling == 'uniform':
        return (torch.rand(1, generator=generator) + torch.arange(bsz) / bsz) % (1 - 1e-5)
    # Beta(a, b) = Ga / (Ga + Gb) from two Gamma draws (torch's Beta sampler takes no generator)
    ga = torch._standard_gamma(torch.full((bsz,), float(alpha)), generator=generator)
    gb = torch._standard_gamma(torch.full((bsz,), float(beta)), generator=generator)
    return t_max * (1 - ga / (ga + gb))


def cosine_warmup_restarts_lr(step, first_cycle_steps, max_lr, min_lr=0.0, warmup_steps=0, cycle_mult=1.0, gamma=1.0):
    """`CosineAnnealingWarmupRestarts.get_lr` (src/utils/optim.py:31-160) for optimizer step `step` (0-based)."""
    if cycle_mult == 1.0:
        cycle, in_cycle, cur = step // first_cycle_steps, step % first_cycle_steps, first_cycle_steps
    else:
        n = int(math.log(step / first_cycle_steps * (cycle_mult - 1) + 1, cycle_mult)) if step >= first_cycle_steps else 0
        cycle = n
        in_cycle = step - int(first_cycle_steps * (cycle_mult ** n - 1) / (cycle_mult - 1))
        cur = int(first_cycle_steps * cycle_mult ** n)
    mx = max_lr * gamma ** cycle
    if in_cycle < warmup_steps:
        return (mx - min_lr) * in_cycle / warmup_steps + min_lr
    return min_lr + (mx - min_lr) * (1 + math.cos(math.pi * (in_cycle - warmup_steps) / (cur - warmup_steps))) / 2


class ModelAveraging:
    """EMA / SWA of the trained parameters -- mirror of `ModelAveraging` (src/agent/model_averaging.py:8-72; driven from `TrainAgent.run`,
    train.py:524-528: `maybe_initialize(cnt_update)` then `maybe_update(cnt_update)` after every optimizer step).  The reference wraps the
    model in torch's `AveragedModel` (first update copies, later ones lerp with 1 - ema_decay or average with 1 / (n + 1)); here the
    average lives in fp32 next to the rank's ZeRO-1 master shard and is updated by ONE streaming kernel (`vlaser_avg_update`).  Like the
    reference's, it does not support resuming."""

    def __init__(self, trainer, use_ema=False, use_swa=False, ema_start=0, ema_decay=0.99, ema_freq=1, swa_start=0, swa_freq=1):
        assert not (use_ema and use_swa), 'Cannot use both EMA and SWA at once'
        self.trainer, self.use_ema, self.use_swa = trainer, use_ema, use_swa
        self.ema_start, self.ema_decay, self.ema_freq = ema_start, ema_decay, ema_freq
        self.swa_start, self.swa_freq = swa_start, swa_freq
        self.avg = None
        self.avg_vlm = None           # train_vlm=True: the reference's AveragedModel wraps the WHOLE PiZero, so the VLM group is averaged too
        self.n_averaged = 0

    def _groups(self):
        """(flat parameter owner, its average) per trained group: the action expert and, with `train_vlm`, the VLM group (vision tower, projector, VLM
        decoder layers) -- `AveragedModel(model)` (model_averaging.py:33-44) averages every parameter of the model it wraps."""
        tr = self.trainer
        out = [(tr, self.avg)]
        if getattr(tr, 'vg', None) is not None:
            out.append((tr.vg, self.avg_vlm))
        return out

    def maybe_initialize(self, cnt_update):
        if (self.use_swa and cnt_update == self.swa_start) or (self.use_ema and cnt_update == self.ema_start):
            self.avg = torch.zeros_like(self.trainer.master)
            vg = getattr(self.trainer, 'vg', None)
            self.avg_vlm = torch.zeros_like(vg.master) if vg is not None else None
            self.n_averaged = 0

    def maybe_update(self, cnt_update):
        if self.avg is None:
            return
        if (self.use_ema and cnt_update % self.ema_freq == 0) or (self.use_swa and cnt_update % self.swa_freq == 0):
            c = (1.0 - self.ema_decay) if self.use_ema else 1.0 / (self.n_averaged + 1)
            for owner, avg in self._groups():
                ops.avg_update(avg, owner.master, c, self.n_averaged == 0)
            self.n_averaged += 1

    def state_dict(self):
        """{'state_dict': averaged weights under the canonical VLA key names (bf16), 'n_averaged', 'model_type'} -- {} before the start step.
        Both trained groups carry their average (the VLM group used to be saved un-averaged under an 'ema' / 'swa' label)."""
        if self.avg is None:
            return {}
        tr = self.trainer
        groups = self._groups()
        keep = [owner.fp.p.clone() for owner, _ in groups]
        try:
            for owner, avg in groups:
                for (lo, hi, _), o in zip(owner.shards, owner.shard_off):
                    if hi > lo:
                        owner.fp.p[lo:hi].copy_(avg[o:o + hi - lo].to(BF))
                if tr.dp_active:
                    for b in range(len(owner.buckets)):
                        dp.all_gather_params(owner.fp.p, owner.buckets[b], owner.shards[b], tr.pg)
            sd = tr.state_dict()
        finally:
            for (owner, _), k in zip(groups, keep):
                owner.fp.p.copy_(k)
        return {'state_dict': sd, 'n_averaged': self.n_averaged, 'model_type': 'ema' if self.use_ema else 'swa'}


class VLATrainer:
    """`step(batch...)` = one optimizer update of the action expert on flow-matching samples (per-device batch B >= 1, sample by sample)."""

    def __init__(self, cfg: VLAConfig, device='cuda', lr=5e-5, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 process_group=None, bucket_layers=8, train_vlm=False, vlm_lr=5e-5, vlm_weight_decay=0.0):
        L.lib()
        if not torch.cuda.is_available():
            raise L.VlaserHipError('vlaser_amd needs an MI355X (gfx950) GPU: there is no CPU fallback')
        self.cfg, self.device = cfg, torch.device(device)
        self.lr, self.wd, self.betas, self.eps, self.max_grad_norm = lr, weight_decay, betas, eps, max_grad_norm
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        self.dp_active = self.world > 1 or (process_group is not None and os.environ.get('VLASER_FORCE_DP') == '1')
        self.bucket_layers = bucket_layers
        # `train_vlm: True` (train.py:270-295): the reference's second parameter group `trainable_vlm_parameters` (pizero_internvl.py:405-411) with its
        # own learning rate / weight decay (`vlm_lr`, `vlm_weight_decay`); vlaser_amd/vla_vlm_group.py
        self.train_vlm, self.vlm_lr, self.vlm_wd = train_vlm, vlm_lr, vlm_weight_decay
        self.vg = None
        self.step_count = 0
        self.T, self.na = cfg.max_image_text_tokens, cfg.num_action_tokens
        self.R = 1 + self.na                       # expert rows: proprio + action tokens

    # ------------------------------------------------------------------ parameters
    def load_state_dict(self, sd):
        sd = canonicalize_vla_state_dict(sd)
        cfg, dev, ex = self.cfg, self.device, self.cfg.expert
        base, llm = cfg.base, cfg.base.llm
        self.vit = VitEngine(sd, base, dev, max_tiles=1)
        self.vlm = QwenStack(sd, 'language_model.', llm, dev, with_embed=True, with_head=False, gemm=True, skinny=False)      # frozen
        H, I = ex.hidden_size, ex.intermediate_size
        nq, nkv, hd = ex.num_attention_heads, ex.num_key_value_heads, ex.head_dim
        NQ, A, P = (nq + 2 * nkv) * hd, cfg.action_dim, cfg.proprio_dim
        self.A8 = 8                                 # tiny operands are padded to the kernels' 8-column granularity
        fp = FlatParams(dev)
        fp.add('dec.w', (A, H)); fp.add('dec.b', (A,)); fp.add('norm', (H,))
        names = [('wqkv', (NQ, H)), ('bqkv', (NQ,)), ('wo', (H, nq * hd)), ('wgu', (2 * I, H)), ('wdown', (H, I)), ('ln_in', (H,)), ('ln_post', (H,))]
        Lyr = ex.num_hidden_layers
        bounds = []
        for j, i in enumerate(reversed(range(Lyr))):
            if j % self.bucket_layers == 0 and j > 0:
                fp.align(128 * self.world); bounds.append(fp.n)
            for nm, shp in names:
                fp.add(f'l{i}.{nm}', shp)
        fp.align(128 * self.world); bounds.append(fp.n)
        for nm, shp in [('ae1.w', (H, A)), ('ae1.b', (H,)), ('ae2.w', (H, 2 * H)), ('ae2.b', (H,)), ('ae3.w', (H, H)), ('ae3.b', (H,)), ('pe.w', (H, P)),
                        ('pe.b', (H,))]:
            fp.add(nm, shp)
        fp.finalize(pad_to=128 * self.world * 8)
        self.fp = fp
        g = lambda k: sd[k].to(device=dev, dtype=BF)
        v = fp.view
        v['dec.w'].copy_(g('action_decoder.weight')); v['dec.b'].copy_(g('action_decoder.bias')); v['norm'].copy_(g('action_expert.model.norm.weight'))
        for i in range(Lyr):
            p = f'action_expert.model.layers.{i}.'
            wqkv, bqkv = ops.pack_qkv(g(p + 'self_attn.q_proj.weight'), g(p + 'self_attn.k_proj.weight'), g(p + 'self_attn.v_proj.weight'),
                                      g(p + 'self_attn.q_proj.bias'), g(p + 'self_attn.k_proj.bias'), g(p + 'self_attn.v_proj.bias'), hd)
            v[f'l{i}.wqkv'].copy_(wqkv); v[f'l{i}.bqkv'].copy_(bqkv)
            v[f'l{i}.wo'].copy_(g(p + 'self_attn.o_proj.weight'))
            v[f'l{i}.wgu'].copy_(ops.pack_gate_up(g(p + 'mlp.gate_proj.weight'), g(p + 'mlp.up_proj.weight')))
            v[f'l{i}.wdown'].copy_(g(p + 'mlp.down_proj.weight'))
            v[f'l{i}.ln_in'].copy_(g(p + 'input_layernorm.weight')); v[f'l{i}.ln_post'].copy_(g(p + 'post_attention_layernorm.weight'))
        for nm, k in [('ae1', 'action_encoder.linear_1'), ('ae2', 'action_encoder.linear_2'), ('ae3', 'action_encoder.linear_3'), ('pe', 'proprio_encoder')]:
            v[nm + '.w'].copy_(g(k + '.weight')); v[nm + '.b'].copy_(g(k + '.bias'))
        self.buckets, lo = [], 0
        for hi in bounds + [fp.n]:
            if hi > lo:
                self.buckets.append((lo, hi)); lo = hi
        assert all((hi - lo) % (128 * self.world) == 0 for lo, hi in self.buckets)
        self.bucket_of_layer = {i: min(j // self.bucket_layers, len(self.buckets) - 2) for j, i in enumerate(reversed(range(Lyr)))}
        self.shards = dp.plan_shards(self.buckets, self.world, self.rank)
        n_shard = sum(hi - lo for lo, hi, _ in self.shards)
        self.master = torch.zeros(n_shard, dtype=F32, device=dev)
        self.m = torch.zeros(n_shard, dtype=F32, device=dev)
        self.v = torch.zeros(n_shard, dtype=F32, device=dev)
        self.shard_off, o = [], 0
        for lo, hi, _ in self.shards:
            self.master[o:o + hi - lo].copy_(fp.p[lo:hi].float())
            self.shard_off.append(o); o += hi - lo
        self._alloc()
        self._refresh_transposes()
        if self.train_vlm:
            from .vla_vlm_group import VLMGroup
            self.vg = VLMGroup(self, sd, self.bucket_layers)
            self.gacc_v = None
        return self

    def _alloc(self):
        cfg, dev, ex, llm = self.cfg, self.device, self.cfg.expert, self.cfg.base.llm
        H, I = ex.hidden_size, ex.intermediate_size
        nq, nkv, hd = ex.num_attention_heads, ex.num_key_value_heads, ex.head_dim
        NQ, Lyr, R, T = (nq + 2 * nkv) * hd, ex.num_hidden_layers, self.R, self.T
        z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device=dev)
        self.s_max = (T + R + 63) // 64 * 64
        self.cache = KVCache(llm.num_hidden_layers, 1, llm.num_key_value_heads, self.s_max, dev, llm.head_dim)
        self.pbuf = PrefillBuffers(self.vlm, T, dev)
        self.rope = ops.rope_table(T + 16, llm.head_dim, llm.rope_theta, dev)
        self.h_vlm = z(T, llm.hidden_size)
        self.rank_ws = z(T, dt=torch.int32)
        self.valid_len = z(1, dt=torch.int32)
        self.pos_vlm, self.pos5 = z(T, dt=torch.int32), z(16, dt=torch.int32)
        # the dgrad GEMMs read the weights as stored (vlaser_gemm_nn); only the 7-row action decoder keeps a zero-padded transposed copy
        # saved activations of the R expert rows (tiny: R x width per layer)
        self.h_in = z(Lyr + 1, 16, H)
        self.x1, self.x2, self.h2 = z(Lyr, 16, H), z(Lyr, 16, H), z(Lyr, 16, H)
        self.q, self.ao = z(Lyr, 16, nq * hd), z(Lyr, 16, nq * hd)
        self.gu, self.act = z(Lyr, 16, 2 * I), z(Lyr, 16, I)
        self.part = torch.zeros(16 * 16 * max(H, 2 * I), dtype=F32, device=dev)
        self.xcat, self.z2, self.e2 = z(16, 2 * H), z(16, H), z(16, H)
        self.hn, self.vout = z(16, H), torch.zeros(16, self.A8, dtype=F32, device=dev)
        self.psi = torch.zeros(16, cfg.action_dim, dtype=F32, device=dev)
        self.proprio = torch.zeros(1, cfg.proprio_dim, dtype=F32, device=dev)
        # backward buffers
        self.dh, self.dh2, self.dx = z(16, H), z(16, H), z(16, H)
        self.dact, self.dgu = z(16, I), z(16, 2 * I)
        self.dao, self.dq = z(16, nq * hd), z(16, nq * hd)
        self.dgrad_part = torch.zeros(40 * 16 * 1536, dtype=F32, device=self.device)      # split-K slabs of the long-contraction dgrads (_dgrad)
        self.arb_ws = torch.zeros(L.lib().vlaser_attn_rows_bwd_ws_floats(nq), dtype=F32, device=self.device)      # block-key P / dS of vlaser_attn_rows_bwd
        self.dk, self.dv = z(16, nkv * hd), z(16, nkv * hd)
        self.dqkv = z(16, NQ)
        self.dv64, self.decT = z(16, 64), z(H, 64)
        self.col = torch.zeros(max(2 * I, NQ, 2 * H), dtype=F32, device=dev)
        self.rowstat = torch.zeros(2 * 16 + 16 * max(2 * I, NQ, 2 * H), dtype=F32, device=dev)
        self.normw_ws = torch.zeros(8 * H, dtype=F32, device=dev)
        self.gnorm2 = torch.zeros(1, dtype=F32, device=dev)
        self.sumsq_ws = torch.zeros(1024, dtype=F32, device=dev)
        self.gacc = None
        self.comm_stream = torch.cuda.Stream(device=dev) if self.dp_active else None

    def _refresh_transposes(self):
        v = self.fp.view
        # action decoder [A, H] -> [H, 64] zero padded (contraction axis of the dgrad GEMM must be a multiple of 64)
        self.decT.zero_()
        ops.transpose(v['dec.w'], self.decT, v['dec.w'].shape[0], v['dec.w'].shape[1], v['dec.w'].shape[1], 64)

    # ------------------------------------------------------------------ small helpers
    def _dgrad(self, dY, W, out, M):
        """out[M,K] = dY[M,N] @ W[N,K], W as the forward stores it (NN GEMM).  A long contraction over M <= 16 rows is a weight STREAM: the gate/up dgrad
        ([5, 17920] @ [17920, 768]) has six output tiles, i.e. six workgroups pulling 27.5 MB (~80 us, r06ab trace); as many split-K slices as divide the
        contraction put 240 workgroups on it, the fp32 slabs (a few hundred KB) are summed in slab order by one launch -- the rule of the SFT step's head dgrad."""
        Nin, Kout = W.shape
        if Nin > 2048 and Kout <= 4096 and Kout % 8 == 0:
            sp = max(d for d in range(1, 41) if Nin % (64 * d) == 0)
            if sp > 8 and sp * M * Kout <= self.dgrad_part.numel():
                part = self.dgrad_part[:sp * M * Kout]
                ops.gemm_nn(L.EPI_PARTIAL, dY[:M], W, out_f32=part, k_splits=sp)
                ops.reduce_norm(None, part, sp, M, Kout, out[:M])
                return
        ops.gemm_nn(L.EPI_NONE, dY[:M], W, out=out[:M])

    def _wgrad(self, dY, X, out, M, bias_out=None):
        """out[N,K] = dY[:M]^T @ X[:M] (contraction over the M <= 16 rows)."""
        ops.gemm_tn(dY[:M], X[:M], out)
        if bias_out is not None:
            ops.colsum_bf16(dY[:M], bias_out, M, dY.shape[1])

    def _linear(self, x, w, b, out, M, gelu=False):
        ops.gemm(L.EPI_BIAS, x[:M], w, out=out[:M], bias=b)

    # ------------------------------------------------------------------ forward + backward of ONE sample
    def forward_backward(self, input_ids, pixel_values, proprios, actions, t, x0, vlm_position_ids=None, proprio_position_ids=None,
                         action_position_ids=None, causal_mask=None, on_bucket_ready=None):
        """Loss (device scalar) + gradients into self.fp.g.  Tensors of one sample: input_ids [1,T] (right-padded), pixel_values
        [1,3,448,448], proprios [1,1,P], actions / x0 [1,na,A], t [1]."""
        cfg, dev, ex, llm = self.cfg, self.device, self.cfg.expert, self.cfg.base.llm
        T, na, R = self.T, self.na, self.R
        H, I, A = ex.hidden_size, ex.intermediate_size, cfg.action_dim
        nq, nkv, hd = ex.num_attention_heads, ex.num_key_value_heads, ex.head_dim
        Lyr = ex.num_hidden_layers
        v, gv = self.fp.view, self.fp.gview
        ids_h = input_ids.detach().to('cpu', torch.int64).reshape(1, T)
        n_valid = int((ids_h != cfg.base.pad_token_id).sum())
        # visibility is expressed as (valid_len, blk_start) descriptors: the prompt must be strictly right-padded, and a dense mask, when
        # given, must be exactly the block mask of that pad count (`PiZero.forward` would honour any mask; this build refuses the others)
        if n_valid < T and bool((ids_h[0, :n_valid] == cfg.base.pad_token_id).any()):
            raise ValueError('input_ids must be right-padded: pad tokens inside the valid prefix are not supported')
        if causal_mask is not None:
            prep.check_block_mask(causal_mask, [n_valid], T, 1, na)
        tval = float(t.reshape(-1)[0])
        sig = cfg.flow_sig_min
        # ---- frozen prefix: ViT -> projector -> embeddings -> VLM layers (inference kernels), K / V^T of every layer cached
        pvb = stage_pixels(pixel_values, torch.empty(pixel_values.shape, dtype=BF, device=dev), dev)
        ids = ids_h.pin_memory().to(dev, non_blocking=True)
        self.valid_len.fill_(n_valid)
        bpos = lambda p, default: (default if p is None else p).to(torch.int32).reshape(-1)
        self.pos_vlm.copy_(bpos(vlm_position_ids, torch.arange(1, T + 1)))
        self.pos5[:1].copy_(bpos(proprio_position_ids, torch.ones(1, dtype=torch.long)))
        self.pos5[1:R].copy_(bpos(action_position_ids, torch.arange(2, 2 + na)))
        if self.vg is not None:
            self.vg.forward(pvb, ids)                   # train_vlm: the same rows with every intermediate kept (vla_vlm_group.py)
        else:
            feats = self.vit.forward(pvb)
            ops.embed_merge(ids, self.vlm.embed, feats, self.h_vlm, cfg.base.img_context_token_id, cfg.base.pad_token_id, True, self.rank_ws)
            nLv = llm.num_hidden_layers
            prefill_begin(self.vlm, self.pbuf, self.h_vlm, T)
            for i in range(nLv):
                last = i == nLv - 1
                prefill_layer(self.vlm, self.vlm.layers[i], self.pbuf, self.h_vlm, self.cache, i, self.rope, self.pos_vlm, 1, T, L.ATTN_PREFIX,
                              valid_len=self.valid_len, blk_start=T, skip_post_attn=last, next_norm_w=None if last else self.vlm.layers[i + 1].ln_in)
        # ---- inputs of the expert rows: proprio encoder (row 0), action encoder on psi_t (rows 1..na)
        x1a = actions.reshape(na, A).to(dev, F32)
        x0a = x0.reshape(na, A).to(dev, F32)
        self.psi[:na].copy_((1 - (1 - sig) * tval) * x0a + tval * x1a)          # psi_t (:1050-1062): 28 numbers of input preparation
        self.proprio.copy_(proprios.reshape(1, -1).to(dev, F32))
        h0 = self.h_in[0]
        ops.small_linear(self.proprio, v['pe.w'], v['pe.b'], h0, 1, H, cfg.proprio_dim)
        ops.vla_prep(self.psi, v['ae1.w'], v['ae1.b'], self.xcat, na, H, A, tval, cfg.time_max_period)       # [time embedding | linear_1(psi)]
        self._linear(self.xcat, v['ae2.w'], v['ae2.b'], self.z2, na)
        ops.silu(self.z2[:na], self.e2[:na])
        self._linear(self.e2, v['ae3.w'], v['ae3.b'], h0[1:], na)
        # ---- expert layers on the R rows, everything saved
        scale = hd ** -0.5
        ks, vs = self.cache.strides()
        for i in range(Lyr):
            h_in, x1, x2, h2, q, ao, gu, act = self.h_in[i, :R], self.x1[i, :R], self.x2[i, :R], self.h2[i, :R], self.q[i, :R], self.ao[i, :R], self.gu[i, :R], self.act[i, :R]
            ops.rmsnorm(h_in, v[f'l{i}.ln_in'], ex.rms_norm_eps, out=x1)
            ops.gemm(L.EPI_QKV_ROPE, x1, v[f'l{i}.wqkv'], bias=v[f'l{i}.bqkv'], q_out=q, k_cache=self.cache.k[i], vt_cache=self.cache.vt[i], rope_cos=self.rope[0],
                     rope_sin=self.rope[1], pos_ids=self.pos5, n_q_heads=nq, n_kv_heads=nkv, s_max=self.s_max, tok_per_batch=R, slot_base=T)
            for (r0, nr, kvl) in ((0, 1, T + 1), (1, na, T + R)):            # proprio row: prefix + itself; action rows: prefix + whole block
                ops.attn_prefill(q[r0:], self.cache.k[i], self.cache.vt[i], ao[r0:], 1, nr, kvl, nq, nkv, hd, (nr * nq * hd, hd, nq * hd), ks, vs,
                                 (nr * nq * hd, nq * hd), self.s_max, scale, L.ATTN_PREFIX, valid_len=self.valid_len, blk_start=T, q_row_off=T + r0)
            sp = ops.gemm_splits(R, H, nq * hd)
            ops.gemm(L.EPI_PARTIAL, ao, v[f'l{i}.wo'], out_f32=self.part, k_splits=sp)
            ops.reduce_norm(h_in, self.part, sp, R, H, h2, x2, norm=1, norm_w=v[f'l{i}.ln_post'], eps=ex.rms_norm_eps)
            ops.gemm(L.EPI_NONE, x2, v[f'l{i}.wgu'], out=gu)
            ops.swiglu(gu, act, R, I)
            sp = ops.gemm_splits(R, H, I)
            ops.gemm(L.EPI_PARTIAL, act, v[f'l{i}.wdown'], out_f32=self.part, k_splits=sp)
            ops.reduce_norm(h2, self.part, sp, R, H, self.h_in[i + 1, :R])
        # ---- head + loss on the action rows
        h_fin = self.h_in[Lyr, 1:R]
        hn = self.hn[:na]
        ops.rmsnorm(h_fin, v['norm'], ex.rms_norm_eps, out=hn)
        vout = self.vout[:na]
        ops.gemm(L.EPI_F32, hn, v['dec.w'], out=vout)                       # [na, A] (+ bias below), fp32
        vel = vout[:, :A] + v['dec.b'].float()
        diff = vel.to(BF).float() - (x1a - (1 - sig) * x0a)                 # the decoder is a bf16 Linear in the reference's bf16 training
        loss = (diff * diff).mean()
        dvel = diff * (2.0 / (na * A))                                      # d loss / d v
        # ================================================================ backward
        self.dv64.zero_()
        self.dv64[:na, :A] = dvel.to(BF)
        dv = self.dv64
        # action decoder: dW = dv^T hn (A rows padded to 8), db = column sum, d hn = dv @ W
        wg8 = torch.zeros(8, H, dtype=BF, device=dev)
        ops.gemm_tn(dv[:na, :8], hn, wg8)
        gv['dec.w'].copy_(wg8[:A])
        gv['dec.b'].copy_(dvel.sum(0).to(BF))
        dhn = self.dx[:na]
        ops.gemm(L.EPI_NONE, dv[:na], self.decT, out=dhn)
        dh = self.dh
        dh.zero_()
        if self.vg is not None:
            self.vg.begin_backward()
        ops.rmsnorm_bwd(dhn, h_fin, v['norm'], None, dh[1:R], na, H, ex.rms_norm_eps, dw_out=gv['norm'], dw_ws=self.normw_ws)
        for i in reversed(range(Lyr)):
            h_in, x1, x2, h2, q, ao, gu, act = self.h_in[i, :R], self.x1[i, :R], self.x2[i, :R], self.h2[i, :R], self.q[i, :R], self.ao[i, :R], self.gu[i, :R], self.act[i, :R]
            dact, dgu, dx, dh2, dao = self.dact, self.dgu, self.dx, self.dh2, self.dao
            self._dgrad(dh, v[f'l{i}.wdown'], dact, R)
            self._wgrad(dh, act, gv[f'l{i}.wdown'], R)
            ops.swiglu_bwd(gu, dact[:R], dgu[:R], R, I)
            self._dgrad(dgu, v[f'l{i}.wgu'], dx, R)
            self._wgrad(dgu, x2, gv[f'l{i}.wgu'], R)
            ops.rmsnorm_bwd(dx[:R], h2, v[f'l{i}.ln_post'], dh[:R], dh2[:R], R, H, ex.rms_norm_eps, dw_out=gv[f'l{i}.ln_post'], dw_ws=self.normw_ws)
            self._dgrad(dh2, v[f'l{i}.wo'], dao, R)
            self._wgrad(dh2, ao, gv[f'l{i}.wo'], R)
            vg = self.vg
            ops.attn_rows_bwd(q, self.cache.k[i, 0], self.cache.vt[i, 0], dao[:R], ao, self.dq[:R], self.dk[:R], self.dv[:R], R, nq, nkv, self.s_max,
                              n_valid, T, True, scale, p_out=None if vg is None else vg.p_rows, ds_out=None if vg is None else vg.ds_rows, ws=self.arb_ws)
            if vg is not None:                          # train_vlm: the prefix keys' dK / dV of this layer, then the VLM rows' own layer backward
                vg.prefix_kv_grads((q, dao[:R]), R, nq, nkv)
                vg.backward_layer(i, n_valid)
            ops.rope_bwd_pack(self.dq[:R], self.dk[:R], self.dv[:R], self.rope[0], self.rope[1], self.pos5, self.dqkv[:R], R, nq, nkv, kv_per_q_head=False)
            self._dgrad(self.dqkv, v[f'l{i}.wqkv'], dx, R)
            self._wgrad(self.dqkv, x1, gv[f'l{i}.wqkv'], R, bias_out=gv[f'l{i}.bqkv'])
            ops.rmsnorm_bwd(dx[:R], h_in, v[f'l{i}.ln_in'], dh2[:R], dh[:R], R, H, ex.rms_norm_eps, dw_out=gv[f'l{i}.ln_in'], dw_ws=self.normw_ws)
            if on_bucket_ready and (i == 0 or self.bucket_of_layer[i - 1] != self.bucket_of_layer[i]):
                on_bucket_ready(self.bucket_of_layer[i])
        # ---- encoders: proprio (row 0), action (rows 1..na)
        pw8 = torch.zeros(H, 8, dtype=BF, device=dev)
        p8 = torch.zeros(8, 8, dtype=BF, device=dev); p8[0, :cfg.proprio_dim] = self.proprio[0].to(BF)
        d8 = torch.zeros(8, H, dtype=BF, device=dev); d8[0] = dh[0]
        ops.gemm_tn(d8, p8, pw8)                                            # d pe.w = dh[0]^T proprio
        gv['pe.w'].copy_(pw8[:, :cfg.proprio_dim]); gv['pe.b'].copy_(dh[0])
        da = dh[1:R]
        self._wgrad(da, self.e2, gv['ae3.w'], na, bias_out=gv['ae3.b'])
        de2 = self.dx
        self._dgrad(da, v['ae3.w'], de2, na)
        dz2 = self.dh2
        ops.silu_bwd(self.z2[:na], de2[:na], dz2[:na])
        self._wgrad(dz2, self.xcat, gv['ae2.w'], na, bias_out=gv['ae2.b'])
        dxc = torch.zeros(16, 2 * H, dtype=BF, device=dev)
        self._dgrad(dz2, v['ae2.w'], dxc, na)
        dl1 = dxc[:na, H:].contiguous()                                     # gradient of linear_1's output (the time half has no parameters)
        a8 = torch.zeros(8, 8, dtype=BF, device=dev); a8[:na, :A] = self.psi[:na].to(BF)
        l8 = torch.zeros(8, H, dtype=BF, device=dev); l8[:na] = dl1
        aw8 = torch.zeros(H, 8, dtype=BF, device=dev)
        ops.gemm_tn(l8, a8, aw8)
        gv['ae1.w'].copy_(aw8[:, :A])
        ops.colsum_bf16(dl1, gv['ae1.b'], na, H)
        if on_bucket_ready:
            on_bucket_ready(len(self.buckets) - 1)
        if self.vg is not None:
            self.vg.backward_tail(ids_h)                # mlp1, pixel_shuffle, the vision tower and its embeddings
        return loss

    # ------------------------------------------------------------------ optimizer / data parallel (same machinery as the SFT step)
    def _exchange_bucket(self, b):
        if not self.dp_active:
            return
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            dp.reduce_scatter_mean(self.fp.g, self.buckets[b], self.shards[b], self.pg)

    def _accumulate_bucket(self, b, w, first, last):
        lo, hi = self.buckets[b]
        ops.grad_accumulate(self.fp.g[lo:hi], self.gacc[lo:hi], w, first, last)
        if last:
            self._exchange_bucket(b)

    def _vlm_grads_ready(self, w, first, last, accumulate):
        """VLM group after one sample's backward: accumulate (fp32) when the step has several samples, and hand every bucket to the exchange after
        the last one (no overlap with the backward here: the vision tower finishes last and owns most of the group's buckets anyway)."""
        vg = self.vg
        for b, (lo, hi) in enumerate(vg.buckets):
            if accumulate:
                ops.grad_accumulate(vg.fp.g[lo:hi], self.gacc_v[lo:hi], w, first, last)
            if last and self.dp_active:
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)
                    dp.reduce_scatter_mean(vg.fp.g, vg.buckets[b], vg.shards[b], self.pg)

    def optimizer_step(self, lr=None, vlm_lr=None):
        """clip_grad_norm_ over BOTH groups (train.py:504-507), then one AdamW per group with its own learning rate / weight decay (:509-520)."""
        lr = self.lr if lr is None else lr
        self.step_count += 1
        if self.dp_active:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        groups = [(self.fp, self.shards, self.shard_off, self.master, self.m, self.v, lr, self.wd, self.buckets)]
        if self.vg is not None:
            vg = self.vg
            groups.append((vg.fp, vg.shards, vg.shard_off, vg.master, vg.m, vg.v, self.vlm_lr if vlm_lr is None else vlm_lr, self.vlm_wd, vg.buckets))
        self.gnorm2.zero_()
        for fp, shards, _, _, _, _, _, _, _ in groups:
            for (s_lo, s_hi, _) in shards:
                if s_hi > s_lo:
                    ops.sumsq(fp.g[s_lo:s_hi], self.gnorm2, self.sumsq_ws)
        if self.dp_active:
            torch.distributed.all_reduce(self.gnorm2, group=self.pg)
        for fp, shards, offs, master, m, v_, glr, gwd, _ in groups:
            for (s_lo, s_hi, _), o in zip(shards, offs):
                if s_hi > s_lo:
                    n = s_hi - s_lo
                    ops.adamw_clipped(fp.p[s_lo:s_hi], master[o:o + n], m[o:o + n], v_[o:o + n], fp.g[s_lo:s_hi], glr, self.betas[0],
                                      self.betas[1], self.eps, gwd, 1.0, self.gnorm2, self.max_grad_norm, self.step_count)
        if self.dp_active:
            for fp, shards, _, _, _, _, _, _, buckets in groups:
                for b in range(len(buckets)):
                    dp.all_gather_params(fp.p, buckets[b], shards[b], self.pg)
        self._refresh_transposes()
        return self.gnorm2.sqrt()

    def step(self, samples, lr=None, grad_accumulation_steps=1, vlm_lr=None):
        """One optimizer update over `samples` = list of dicts (input_ids, pixel_values, proprios, actions, t, x0 [, position ids]):
        the per-device batch of every accumulation micro-batch, flattened; each sample weighs 1 / len(samples) (the loss is a batch
        mean, `normalized_loss = loss / grad_accumulation_steps`, train.py:479-498)."""
        n = len(samples)
        if n > 1 and self.gacc is None:
            self.gacc = torch.zeros(self.fp.n, dtype=F32, device=self.device)
        if n > 1 and self.vg is not None and self.gacc_v is None:
            self.gacc_v = torch.zeros(self.vg.fp.n, dtype=F32, device=self.device)
        loss = torch.zeros((), device=self.device)
        for j, smp in enumerate(samples):
            first, last = j == 0, j == n - 1
            hook = self._exchange_bucket if n == 1 else (lambda b, first=first, last=last: self._accumulate_bucket(b, 1.0 / n, first, last))
            loss = loss + self.forward_backward(on_bucket_ready=hook, **smp) / n
            if self.vg is not None:
                self._vlm_grads_ready(1.0 / n, first, last, n > 1)
        gnorm = self.optimizer_step(lr, vlm_lr)
        return SimpleNamespace(loss=loss, grad_norm=gnorm)

    # ------------------------------------------------------------------ export (canonical VLA key names, un-packed layouts)
    def state_dict(self, grads=False):
        ex = self.cfg.expert
        v = self.fp.gview if grads else self.fp.view
        nq, nkv, hd = ex.num_attention_heads, ex.num_key_value_heads, ex.head_dim
        inv = torch.empty(hd, dtype=torch.long); inv[ops.head_perm(hd)] = torch.arange(hd)
        out = {'action_decoder.weight': v['dec.w'].clone(), 'action_decoder.bias': v['dec.b'].clone(), 'action_expert.model.norm.weight': v['norm'].clone()}
        for nm, k in [('ae1', 'action_encoder.linear_1'), ('ae2', 'action_encoder.linear_2'), ('ae3', 'action_encoder.linear_3'), ('pe', 'proprio_encoder')]:
            out[k + '.weight'], out[k + '.bias'] = v[nm + '.w'].clone(), v[nm + '.b'].clone()
        for i in range(ex.num_hidden_layers):
            p = f'action_expert.model.layers.{i}.'
            w, b = v[f'l{i}.wqkv'], v[f'l{i}.bqkv']
            idx = (torch.arange(nq + 2 * nkv)[:, None] * hd + inv[None, :]).reshape(-1).to(w.device)
            wn, bn = w[idx], b[idx]
            out[p + 'self_attn.q_proj.weight'], out[p + 'self_attn.q_proj.bias'] = wn[:nq * hd].clone(), bn[:nq * hd].clone()
            out[p + 'self_attn.k_proj.weight'], out[p + 'self_attn.k_proj.bias'] = wn[nq * hd:(nq + nkv) * hd].clone(), bn[nq * hd:(nq + nkv) * hd].clone()
            out[p + 'self_attn.v_proj.weight'], out[p + 'self_attn.v_proj.bias'] = wn[(nq + nkv) * hd:].clone(), bn[(nq + nkv) * hd:].clone()
            out[p + 'self_attn.o_proj.weight'] = v[f'l{i}.wo'].clone()
            gu = v[f'l{i}.wgu'].view(-1, 2, 16, ex.hidden_size)
            out[p + 'mlp.gate_proj.weight'] = gu[:, 0].reshape(-1, ex.hidden_size).clone()
            out[p + 'mlp.up_proj.weight'] = gu[:, 1].reshape(-1, ex.hidden_size).clone()
            out[p + 'mlp.down_proj.weight'] = v[f'l{i}.wdown'].clone()
            out[p + 'input_layernorm.weight'] = v[f'l{i}.ln_in'].clone()
            out[p + 'post_attention_layernorm.weight'] = v[f'l{i}.ln_post'].clone()
        if self.vg is not None:
            out.update(self.vg.state_dict(grads))
        return out

    def named_grads(self):
        return self.state_dict(grads=True)

    def _opt_shard_path(self, path):
        return f'{path}.optimizer_rank{self.rank:05d}_of_{self.world:05d}.pt'

    def save_checkpoint(self, path, frozen_sd, cnt_batch=0):
        """`step{N}.pt` of the reference trainer (train.py:639-672): RANK 0 writes the full model under the reference's key names (trained
        expert group from the flat buffer + the frozen VLM tensors `frozen_sd` as loaded); EVERY rank writes its ZeRO-1 shard of the fp32
        masters + AdamW moments next to it (`<path>.optimizer_rank{r}_of_{w}.pt`, as SFTModel.save_checkpoint does) for a bit-identical
        resume.  Call it on all ranks."""
        from .pizero import save_vla_checkpoint
        if self.dp_active:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.rank == 0:
            sd = {k: v for k, v in canonicalize_vla_state_dict(frozen_sd).items() if not k.startswith(('action_expert.model.', 'action_encoder.', 'proprio_encoder.', 'action_decoder.'))}
            sd.update(self.state_dict())
            # every key the reference's `save_training` writes (train.py:655-670), so that its `load_checkpoint` (reads data["wandb_id"]) works and its
            # `load_optimizer` fails on a clear None instead of a KeyError: the optimizer state lives in the per-rank shard files below (fp32 masters +
            # moments of the ZeRO-1 shard -- not torch.optim state dicts)
            extra = {'action_optimizer': None, 'vlm_optimizer': None, 'action_lr_scheduler': None, 'vlm_lr_scheduler': None, 'wandb_id': None, 'n_averaged': 1}
            save_vla_checkpoint(path, sd, cnt_update=self.step_count, cnt_batch=cnt_batch, extra=extra)
        st = {'rank': self.rank, 'world': self.world, 'shards': self.shards, 'step_count': self.step_count, 'master': self.master.cpu(),
              'exp_avg': self.m.cpu(), 'exp_avg_sq': self.v.cpu()}
        if self.vg is not None:                      # the VLM group's shard of the second optimiser (train_vlm)
            st.update({'vlm_shards': self.vg.shards, 'vlm_master': self.vg.master.cpu(), 'vlm_exp_avg': self.vg.m.cpu(), 'vlm_exp_avg_sq': self.vg.v.cpu()})
        torch.save(st, self._opt_shard_path(path))

    def load_checkpoint(self, path, resume_optimizer=True):
        """Weights from the reference-layout `.pt`; with `resume_optimizer` also this rank's optimizer shard -- a missing shard file or one
        written for another world size / bucket layout raises (resuming with zero moments would silently change the run).  Pass
        `resume_optimizer=False` to start a fresh optimizer from released weights."""
        data = torch.load(path, map_location='cpu', weights_only=True)
        self.load_state_dict(data['model'])
        if not resume_optimizer:
            return self
        sp = self._opt_shard_path(path)
        if not os.path.exists(sp):
            raise FileNotFoundError(f'{sp}: no optimizer shard for rank {self.rank} of {self.world} (resume_optimizer=False loads the weights only)')
        st = torch.load(sp, map_location='cpu', weights_only=True)        # tensors, ints and tuples only
        if st['world'] != self.world or st['rank'] != self.rank or [tuple(x) for x in st['shards']] != [tuple(x) for x in self.shards]:
            raise ValueError(f"optimizer shard was written for rank {st['rank']} of {st['world']} / another bucket layout")
        self.step_count = st['step_count']
        self.master.copy_(st['master']); self.m.copy_(st['exp_avg']); self.v.copy_(st['exp_avg_sq'])
        for (lo, hi, _), o in zip(self.shards, self.shard_off):
            if hi > lo:
                self.fp.p[lo:hi].copy_(self.master[o:o + hi - lo].to(BF))
        if self.vg is not None:
            vg = self.vg
            if 'vlm_master' not in st or [tuple(x) for x in st['vlm_shards']] != [tuple(x) for x in vg.shards]:
                raise ValueError('the optimizer shard holds no (matching) state for the VLM parameter group (train_vlm)')
            vg.master.copy_(st['vlm_master']); vg.m.copy_(st['vlm_exp_avg']); vg.v.copy_(st['vlm_exp_avg_sq'])
            for (lo, hi, _), o in zip(vg.shards, vg.shard_off):
                if hi > lo:
                    vg.fp.p[lo:hi].copy_(vg.master[o:o + hi - lo].to(BF))
        # (the other ranks' slices of fp.p are the checkpoint's bf16 weights = bf16(their masters): rank 0 saved them after the all-gather)
        self._refresh_transposes()
        return self
