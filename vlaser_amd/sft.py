"""SFT data-parallel step of InternVLChatModel on MI355X.

Reproduces the step semantics of the reference's SFT launcher
(`internvl_chat_finetune.py:798-1068` + `shell/internvl3.0/2nd_finetune/internvl3_2b_dynamic_res_2nd_finetune_full.sh:25-69`
+ `zero_stage1_config.json`) without its control plane (HF Trainer / DeepSpeed):

  * forward  = `InternVLChatModel.forward` with labels (`modeling_internvl_chat.py:143-255`): frozen ViT, trainable mlp1 + LLM,
    visual-token scatter, shifted CrossEntropy (mean over labels != -100);
  * backward with per-layer activation recompute (the reference checkpoints every LLM layer, `internvl_chat_finetune.py:975-979`);
  * data parallelism: micro-batch sharded over ranks, gradients averaged with bucketed RCCL reduce-scatter issued as soon as a
    bucket's layers have finished their backward (overlaps the remaining backward), ZeRO-1: every rank owns 1/N of each bucket's
    fp32 master weights + AdamW moments, updates its shard, all-gathers the bf16 parameters;
  * AdamW (beta .9/.999, eps 1e-8, weight decay, bias correction) on fp32 masters, bf16 params/grads, grad-norm clipping.

All arithmetic runs in hand-written gfx950 kernels (vlaser_amd/csrc); torch is device memory, streams and RCCL.
Parameters live in ONE flat bf16 buffer in kernel layout (q/k/v fused+permuted, gate/up interleaved) ordered
[lm_head, final norm, layer L-1 ... layer 0, embed_tokens, mlp1] so gradient buckets complete front to back.
"""
import math
import contextlib
import os
from types import SimpleNamespace

import torch

from . import _lib as L
from . import dp, ops
from .config import VlaserConfig
from .engine import BF, KVCache, VitEngine

F32 = torch.float32


def ce_loss(logits, labels, ignore_index=-100):
    """Mean cross entropy over labels != ignore_index of fp32 logits [R, V] (already shifted)."""
    assert logits.dtype == torch.float32 and logits.is_cuda
    logits = logits if logits.is_contiguous() else logits.contiguous()
    labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
    R, V = logits.shape
    rows = torch.empty(R, dtype=torch.float32, device=logits.device)
    ops.ce_rows(logits, labels, rows, None, ignore_index)
    n = (labels != ignore_index).sum().clamp(min=1)
    return rows.sum() / n


def cosine_lr(step, total_steps, base_lr, warmup_ratio=0.03):
    """Learning rate used for optimizer step number `step` (0-based): HF `get_cosine_schedule_with_warmup` as the SFT
    launcher configures it (`--lr_scheduler_type cosine --warmup_ratio 0.03 --learning_rate 2e-5`, …2b…full.sh:55-58):
    linear warm-up over ceil(total*ratio) steps, then half a cosine down to 0."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


class FlatParams:
    """Flat bf16 parameter / gradient buffers with named views (+ fp32 master / moments for the local ZeRO-1 shard)."""

    def __init__(self, device):
        self.device = device
        self.specs = []          # (name, shape, offset)
        self.n = 0

    def add(self, name, shape):
        numel = 1
        for s in shape:
            numel *= s
        self.specs.append((name, tuple(shape), self.n))
        self.n += (numel + 127) // 128 * 128          # 256-byte aligned views
        return len(self.specs) - 1

    def align(self, multiple):
        """Pad so that the next tensor starts at a multiple of `multiple` elements (bucket boundaries: every ZeRO-1 bucket
        then divides evenly over the ranks and the collectives run in place, without staging copies)."""
        self.n = (self.n + multiple - 1) // multiple * multiple

    def finalize(self, pad_to=1):
        self.n = (self.n + pad_to - 1) // pad_to * pad_to
        self.p = torch.zeros(self.n, dtype=BF, device=self.device)
        self.g = torch.zeros(self.n, dtype=BF, device=self.device)
        self.view = {}
        self.gview = {}
        for name, shape, off in self.specs:
            numel = 1
            for s in shape:
                numel *= s
            self.view[name] = self.p[off:off + numel].view(shape)
            self.gview[name] = self.g[off:off + numel].view(shape)

    def offset_of(self, name):
        for n, _, off in self.specs:
            if n == name:
                return off
        raise KeyError(name)


def plan_flat_layout(cfg, world, bucket_layers=4, device='cpu'):
    """The flat parameter / gradient layout of the SFT step and its gradient buckets, without allocating anything: (FlatParams before `finalize`,
    [(lo, hi)]).  Order = the order the backward completes the gradients: [head (+ zero pad rows up to a multiple of 64) + final norm], the layers
    last to first in groups of `bucket_layers`, [embedding + projector]; every bucket boundary is a multiple of 128 * world elements, so each ZeRO-1
    bucket divides evenly over the ranks and the RCCL collectives run in place (dp.py).  Also used by the CPU test of the world-8 shard plan."""
    llm = cfg.llm
    H, I, V = llm.hidden_size, llm.intermediate_size, llm.vocab_size
    nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
    NQ = (nq + 2 * nkv) * hd
    Vp = (V + 63) // 64 * 64
    fp = FlatParams(device)
    fp.add('head', (V, H))
    if Vp > V:
        fp.add('head_pad', (Vp - V, H))       # zero rows (zero gradients, so AdamW keeps them zero): the head's dgrad contracts over Vp = V rounded up to 64
    fp.add('norm', (H,))
    fp.align(128 * world)                     # end of bucket 0 (head + final norm)
    for i in reversed(range(llm.num_hidden_layers)):
        for nm, shp in [('wqkv', (NQ, H)), ('bqkv', (NQ,)), ('wo', (H, nq * hd)), ('wgu', (2 * I, H)), ('wdown', (H, I)), ('ln_in', (H,)),
                        ('ln_post', (H,))]:
            fp.add(f'l{i}.{nm}', shp)
    fp.align(128 * world)                     # layer buckets are whole layers: already multiples for world <= 8, keep it explicit
    fp.add('embed', (V, H))
    C4 = cfg.vision.hidden_size * 4
    for nm, shp in [('m0w', (C4,)), ('m0b', (C4,)), ('m1w', (H, C4)), ('m1b', (H,)), ('m3w', (H, H)), ('m3b', (H,))]:
        fp.add('mlp1.' + nm, shp)
    n_total = (fp.n + 128 * world * 8 - 1) // (128 * world * 8) * (128 * world * 8)       # what finalize(pad_to=128 * world * 8) makes of it
    bounds = [fp.offset_of(f'l{llm.num_hidden_layers - 1}.wqkv')]
    layers_rev = list(reversed(range(llm.num_hidden_layers)))
    for j in range(bucket_layers, llm.num_hidden_layers, bucket_layers):
        bounds.append(fp.offset_of(f'l{layers_rev[j]}.wqkv'))
    bounds.append(fp.offset_of('embed'))
    bounds.append(n_total)
    buckets, lo = [], 0
    for hi in bounds:
        if hi > lo:
            buckets.append((lo, hi))
            lo = hi
    return fp, buckets


class SFTModel:
    """Trainable Vlaser-2B SFT step (per rank).  `step(pixel_values, input_ids, labels)` runs forward, backward, the
    gradient exchange and the optimizer update and returns the (rank-local) loss."""

    def __init__(self, cfg: VlaserConfig, device='cuda', max_seq_len=576, max_tiles=1, lr=2e-5, weight_decay=0.05, betas=(0.9, 0.999),
                 eps=1e-8, max_grad_norm=1.0, process_group=None, bucket_layers=4, seed_state_dict=None, recompute=False, attn_bwd_block=1024, attn_bwd='fused'):
        L.lib()
        if not torch.cuda.is_available():
            raise L.VlaserHipError('vlaser_amd needs an MI355X (gfx950) GPU: there is no CPU fallback')
        self.cfg, self.device = cfg, torch.device(device)
        self.llm = cfg.llm
        self.S_max = (max_seq_len + 63) // 64 * 64
        self.lr, self.wd, self.betas, self.eps, self.max_grad_norm = lr, weight_decay, betas, eps, max_grad_norm
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        # VLASER_FORCE_DP=1 runs the collectives even at world size 1 (single-GPU check of the RCCL call sequence)
        self.dp_active = self.world > 1 or (process_group is not None and os.environ.get('VLASER_FORCE_DP') == '1')
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        self.bucket_layers = bucket_layers
        # The reference checkpoints every LLM layer (grad_checkpoint, …full.sh:46) to fit 80 GB parts; one layer's saved
        # activations are ~39 MB at S=560, 1.1 GB for 28 layers -- noise next to 288 GB, so they are kept by default and the
        # backward re-runs nothing.  recompute=True restores the per-layer recompute (same values either way).
        self.recompute = recompute
        # the attention backward walks the query rows in blocks of this many (r03): its score matrices are [heads, block, keys], not [heads, S, S] --
        # the reference's launcher trains at --max_seq_length 16384 (…2nd_finetune_full.sh:38,60), where four S x S matrices per head are 38 GB
        self.attn_bwd_block = attn_bwd_block
        self.wgrad_lds = os.environ.get('VLASER_SFT_WGRAD', 'lds') == 'lds'      # 'tn': the register-staged TN kernel everywhere (A/B)
        self._wgrad_pad_S = None
        self.attn_bwd_mode = os.environ.get('VLASER_SFT_ATTN_BWD', attn_bwd)
        if self.attn_bwd_mode not in ('fused', 'materialised'):
            raise ValueError(f'VLASER_SFT_ATTN_BWD={self.attn_bwd_mode!r}: fused | materialised')
        self.ag_events = {}                       # bucket -> event of its last parameter all-gather (data parallel only)
        self.overlap_allgather = os.environ.get('VLASER_SFT_NO_AG_OVERLAP') != '1'
        self.overlap_optimizer = os.environ.get('VLASER_SFT_NO_OPT_OVERLAP') != '1'
        self.step_count = 0
        self.img_context_token_id = cfg.img_context_token_id
        self.max_tiles = max_tiles
        if seed_state_dict is not None:
            self.load_state_dict(seed_state_dict)

    # ------------------------------------------------------------------ parameters
    def load_state_dict(self, sd):
        cfg, llm, dev = self.cfg, self.llm, self.device
        self.wait_optimizer()
        self._embed_touched = None
        self.vit = VitEngine(sd, cfg, dev, max_tiles=self.max_tiles)     # frozen (freeze_backbone True)
        self.frozen_sd = {k: v.detach().to('cpu') for k, v in sd.items() if k.startswith('vision_model.')}      # for save_pretrained
        H, I, V = llm.hidden_size, llm.intermediate_size, llm.vocab_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        NQ = (nq + 2 * nkv) * hd
        self.Vp = (V + 63) // 64 * 64
        fp, self.buckets = plan_flat_layout(cfg, self.world, self.bucket_layers, dev)
        fp.finalize(pad_to=128 * self.world * 8)
        assert fp.n == self.buckets[-1][1], 'plan_flat_layout and FlatParams.finalize disagree on the padded length'
        self.fp = fp
        o_head = fp.offset_of('head')
        self.head_full = fp.p[o_head:o_head + self.Vp * H].view(self.Vp, H)      # [Vp, H]: lm_head rows + the zero pad rows
        g = lambda k: sd[k].to(device=dev, dtype=BF)
        fp.view['head'].copy_(g('language_model.lm_head.weight'))
        fp.view['norm'].copy_(g('language_model.model.norm.weight'))
        fp.view['embed'].copy_(g('language_model.model.embed_tokens.weight'))
        for i in range(llm.num_hidden_layers):
            p = f'language_model.model.layers.{i}.'
            wqkv, bqkv = ops.pack_qkv(g(p + 'self_attn.q_proj.weight'), g(p + 'self_attn.k_proj.weight'), g(p + 'self_attn.v_proj.weight'),
                                      g(p + 'self_attn.q_proj.bias'), g(p + 'self_attn.k_proj.bias'), g(p + 'self_attn.v_proj.bias'), hd)
            fp.view[f'l{i}.wqkv'].copy_(wqkv); fp.view[f'l{i}.bqkv'].copy_(bqkv)
            fp.view[f'l{i}.wo'].copy_(g(p + 'self_attn.o_proj.weight'))
            fp.view[f'l{i}.wgu'].copy_(ops.pack_gate_up(g(p + 'mlp.gate_proj.weight'), g(p + 'mlp.up_proj.weight')))
            fp.view[f'l{i}.wdown'].copy_(g(p + 'mlp.down_proj.weight'))
            fp.view[f'l{i}.ln_in'].copy_(g(p + 'input_layernorm.weight'))
            fp.view[f'l{i}.ln_post'].copy_(g(p + 'post_attention_layernorm.weight'))
        for nm, k in [('m0w', 'mlp1.0.weight'), ('m0b', 'mlp1.0.bias'), ('m1w', 'mlp1.1.weight'), ('m1b', 'mlp1.1.bias'),
                      ('m3w', 'mlp1.3.weight'), ('m3b', 'mlp1.3.bias')]:
            fp.view['mlp1.' + nm].copy_(g(k))
        # ZeRO-1: every rank owns the slice [lo + r*len/N, lo + (r+1)*len/N) of each bucket (bucket lengths are multiples of 128;
        # uneven division is handled by padding the reduce-scatter input)
        assert all((hi - lo) % (128 * self.world) == 0 for lo, hi in self.buckets), 'ZeRO-1 buckets must divide evenly over the ranks'
        self.shards = dp.plan_shards(self.buckets, self.world, self.rank)
        n_shard = sum(hi - lo for lo, hi, _ in self.shards)
        self.master = torch.zeros(n_shard, dtype=F32, device=dev)
        self.m = torch.zeros(n_shard, dtype=F32, device=dev)
        self.v = torch.zeros(n_shard, dtype=F32, device=dev)
        o = 0
        self.shard_off = []
        for lo, hi, _ in self.shards:
            self.master[o:o + hi - lo].copy_(fp.p[lo:hi].float())
            self.shard_off.append(o)
            o += hi - lo
        self._alloc_workspace()
        return self

    def _alloc_workspace(self):
        cfg, llm, dev = self.cfg, self.llm, self.device
        H, I, V = llm.hidden_size, llm.intermediate_size, llm.vocab_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        NQ = (nq + 2 * nkv) * hd
        S, Lyr = self.S_max, llm.num_hidden_layers
        z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device=dev)
        C4 = cfg.vision.hidden_size * 4
        # the dgrad GEMMs read the forward weights as stored (vlaser_gemm_nn): no transposed copies (r01/r02 kept 3.5 GB of W^T and
        # rebuilt it after every optimizer step: 170 launches, 1.8 ms)
        # activations
        self.h_in = z(Lyr + 1, S, H)               # layer inputs (checkpoints) + final hidden
        Lk = 1 if self.recompute else Lyr          # saved-activation slots: one (reused) or one per layer
        self.cache = KVCache(Lk, 1, nkv, S, dev, hd)    # K / V^T per slot
        self.rope = ops.rope_table(S + 8, hd, llm.rope_theta, dev)
        self.x1, self.x2, self.h2 = z(Lk, S, H), z(Lk, S, H), z(Lk, S, H)
        self.q, self.ao = z(Lk, S, nq * hd), z(Lk, S, nq * hd)
        self.gu, self.act = z(Lk, S, 2 * I), z(Lk, S, I)
        self.part = torch.zeros(8 * S * max(H, I), dtype=F32, device=dev)
        self.xn = z(S, H)
        # backward buffers
        self.dh, self.dh2, self.dx = z(S, H), z(S, H), z(S, H)
        self.dact, self.dgu = z(S, I), z(S, 2 * I)
        self.dao, self.dq, self.dk, self.dv = z(S, nq * hd), z(S, nq * hd), z(S, nq * hd), z(S, nq * hd)      # dk / dv: one partial per Q head
        self.dqkv = z(S, NQ)
        G = nq // nkv
        # attention backward: fused (csrc/attn_bwd.hip: the forward keeps the log-sum-exp, no score matrices) or, VLASER_SFT_ATTN_BWD=materialised,
        # the r02 path through [heads, rows, keys] score matrices, one block of attn_bwd_block query rows at a time
        self.lse = torch.zeros(Lk, nq * S, dtype=F32, device=dev)
        self.delta_ws = torch.zeros(nq * S, dtype=F32, device=dev)
        if self.attn_bwd_mode == 'materialised':
            QB = min(S, (self.attn_bwd_block + 63) // 64 * 64)
            self.sc = torch.zeros(nq, QB, S, dtype=F32, device=dev)
            self.dP = torch.zeros(nq, QB, S, dtype=F32, device=dev)
            self.P, self.dS = z(nq, QB, S), z(nq, QB, S)
        self.dkv_acc = None                        # fp32 [2, S, nq*hd]: dK / dV partial sums over the query blocks (allocated by the first multi-block backward)
        self.col = torch.zeros(max(2 * I, NQ, C4, H), dtype=F32, device=dev)
        self.sumsq_ws = torch.zeros(1024, dtype=F32, device=dev)
        self.normw_ws = torch.zeros((S + 3) // 4 * H, dtype=F32, device=dev)      # norm-weight gradient partials of rmsnorm_bwd
        # r04: the two norm-weight gradients of every layer of a bucket leave their partials in slots of their own and are finished by ONE launch when the
        # bucket completes (2 x bucket_layers launches less per bucket; `VLASER_SFT_NO_NORMW_BATCH=1`: A/B)
        self.normw_slot = (S + 3) // 4 * H
        self.normw_multi = None if os.environ.get('VLASER_SFT_NO_NORMW_BATCH') == '1' else torch.zeros(2 * self.bucket_layers * self.normw_slot, dtype=F32, device=dev)
        self._normw_off = {}                    # first layer of a bucket -> device int64 offsets of its ln_post / ln_in gradients in fp.g, in slot order
        self.gnorm2 = torch.zeros(1, dtype=F32, device=dev)
        self.rank_ws = torch.zeros(S, dtype=torch.int32, device=dev)
        self.pos_all = torch.arange(S, dtype=torch.int32, device=dev)
        self.gacc = None                                            # fp32 gradient accumulator (allocated by the first multi-sample step)
        self._alloc_projector_ws()
        # The exchange (data parallel only).  VLASER_DP_EXCHANGE=pg (default): torch's ProcessGroupNCCL from a comm stream.  =capi: RCCL's C API on a communicator and a
        # stream of this package's own (rccl_capi.py), that stream CU-masked to the last VLASER_DP_COMM_CUS CUs (default 32; 0 = no masks) and EVERY compute stream of the
        # step (main, weight gradients, optimizer) masked to the rest -- RCCL's channel workgroups then never share a CU with a GEMM workgroup (x1.13 instead of
        # x1.23-1.34 on the forward + backward in the one-GPU stand-in, profiles/r05j_rccl_shadow_masks.md).  The GEMM tile heuristics count on the CUs the mask leaves.
        self.exchange_mode = os.environ.get('VLASER_DP_EXCHANGE', 'pg') if self.dp_active else 'none'
        if self.exchange_mode not in ('pg', 'capi', 'none'):
            raise ValueError(f'VLASER_DP_EXCHANGE={self.exchange_mode!r}: pg | capi')
        self.capi, self.main_stream, mask = None, None, None
        if self.exchange_mode == 'capi':
            from . import rccl_capi
            self.capi = rccl_capi.CapiExchange(self.pg, dev, comm_cus=int(os.environ.get('VLASER_DP_COMM_CUS', '32')))
            self.comm_stream = self.capi.stream
            mask = self.capi.compute_mask()
            if mask is not None:
                # The masked main stream becomes the calling thread's CURRENT stream, once, here.  hipExtStreamCreateWithCUMask only makes BLOCKING streams (implicitly
                # ordered against the legacy default stream), and a step that hops default -> masked -> default on every call measured 19.1 ms of forward + backward
                # against 13.8 ms inside one stream context (profiles/r06j_capi_masks.md): a trainer in this mode is a dedicated process, so the switch is global
                self.main_stream = rccl_capi.masked_stream(*mask)
                self.main_stream.wait_stream(torch.cuda.current_stream())
                torch.cuda.set_stream(self.main_stream)
                ops.set_cu_budget(mask[1])
        else:
            self.comm_stream = torch.cuda.Stream(device=dev) if self.dp_active else None
        mk = (lambda **kw: rccl_capi.masked_stream(*mask)) if mask is not None else (lambda **kw: torch.cuda.Stream(device=dev, **kw))
        # single rank: AdamW (HBM-bound, a third of a step) runs on its own stream bucket by bucket in the order the next forward
        # consumes the parameters, so the next step's frozen-ViT / early-layer GEMMs (MFMA-bound) overlap it
        self.opt_stream = mk(priority=int(os.environ.get('VLASER_SFT_OPT_PRIORITY', '0')))
        # r04: the layer weight gradients run on a stream of their own.  dW = dY^T X depends on dY only, nothing in the backward chain depends on it, and most
        # of the chain's launches are single-round grids of 108-252 workgroups on 256 CUs (tools/micro/sft_timeline.py): the weight-gradient GEMMs fill the gaps
        self._slab_norm = os.environ.get('VLASER_SFT_NO_SLAB_NORM') != '1'      # A/B: reduce the gate/up dgrad's slabs in their own launch again
        self.wgrad_stream = None if os.environ.get('VLASER_SFT_NO_WGRAD_STREAM') == '1' else mk()

    def exchange_info(self):
        """What the step's exchange runs on (bench.py puts it on the line as `sft.exchange.mode` ...)."""
        info = {'mode': self.exchange_mode}
        if self.capi is not None:
            m = self.capi.compute_mask()
            info.update({'comm_cus': self.capi.comm_cus, 'compute_cus': m[1] if m else self.capi.total_cus, 'rccl_version_capi': self.capi.version,
                         'cu_masks': m is not None})
        return info

    @contextlib.contextmanager
    def _on_main(self):
        """capi mode with CU masks: the step's launches go to the masked main stream, ordered behind the caller's stream at entry and in front of it at exit."""
        ms = self.main_stream
        cur = torch.cuda.current_stream()
        if ms is None or cur.cuda_stream == ms.cuda_stream:
            yield
            return
        ms.wait_stream(cur)
        with torch.cuda.stream(ms):
            yield
        cur.wait_stream(ms)

    def _alloc_projector_ws(self):
        """Projector (mlp1) workspaces, sized for `max_tiles` tiles x 256 visual tokens."""
        cfg, dev = self.cfg, self.device
        H, C4 = self.llm.hidden_size, cfg.vision.hidden_size * 4
        z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device=dev)
        nt = self.max_tiles * cfg.num_image_token
        self.ps_raw, self.ps_ln = z(nt, C4), z(nt, C4)
        self.z1, self.g1, self.feat = z(nt, H), z(nt, H), z(nt, H)
        self.dvit, self.dg1, self.dz1, self.dln = z(nt, H), z(nt, H), z(nt, H), z(nt, C4)
        I = self.llm.intermediate_size
        NQ = (self.llm.num_attention_heads + 2 * self.llm.num_key_value_heads) * self.llm.head_dim
        self.rowstat = torch.zeros(2 * max(self.S_max, nt) + 16 * max(2 * I, NQ, C4, H), dtype=F32, device=dev)   # colsum scratch

    def _grow_tiles(self, T):
        """A sample with more tiles than the workspaces were built for (dynamic-resolution SFT: up to 12 + thumbnail, plus dummy
        `image_flags == 0` tiles): re-allocate the projector workspaces, as VitEngine._alloc does for the tower's own."""
        if T > self.max_tiles:
            self.max_tiles = T
            self._alloc_projector_ws()

    def _h2d(self, t):
        """Small host index tensor -> device through a pinned staging copy (`non_blocking`: the host does not wait for the stream)."""
        return t.contiguous().pin_memory().to(self.device, non_blocking=True)

    def wait_optimizer(self):
        """Make the current stream wait for parameter updates / all-gathers still in flight on the side streams."""
        if getattr(self, 'opt_stream', None) is not None:
            torch.cuda.current_stream().wait_stream(self.opt_stream)
        if self.dp_active and getattr(self, 'comm_stream', None) is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def _wait_params(self, b):
        """Block the compute stream until bucket b's parameters of the current step have been all-gathered."""
        ev = self.ag_events.get(b)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # ------------------------------------------------------------------ small helpers
    def _wgrad(self, dY, X, out, S, bias_out=None, padded=False, ssq=None):
        """out[N,K] = dY[S,N]^T @ X[S,K] (bf16) by the TN GEMM: both operands are read as they lie (contraction along their
        rows, transposing LDS reads) -- no transposed activation copies.  `padded`: dY / X are views of step buffers with ceil64(S)
        rows whose dY pad rows are zero (`_zero_wgrad_pad`): the product then runs on the LDS-DMA pipeline (r03: 97 instead of 135 us
        per layer at S = 560, tools/micro/tn_lab.py)."""
        Sp = (S + 63) // 64 * 64
        if padded and self.wgrad_lds and dY.shape[1] % 8 == 0 and X.shape[1] % 8 == 0:
            for t in (dY, X):       # the pad rows must lie inside the buffer the view was cut from
                if t.storage_offset() + (Sp - 1) * t.stride(0) + t.shape[1] > t.untyped_storage().nbytes() // t.element_size():
                    raise ValueError(f'_wgrad(padded=True): the operand has no {Sp - S} pad rows behind its {S} rows')
            ops.gemm_tn_lds(dY, X, out, Sp, sumsq_part=ssq)
        else:
            ops.gemm_tn(dY[:S], X[:S], out, sumsq_part=ssq)
        if bias_out is not None:
            ops.colsum_bf16(dY, bias_out, S, dY.shape[1])

    # ------------------------------------------------------------------ gradient norm from the producers (single rank, one sample per step)
    def _plan_fused_norm(self):
        """Slot layout of the partial sums of squares, bucket by bucket (r04): the weight-gradient GEMMs write one slot per (workgroup, wave) of their
        launch (`sumsq_part`; which slots depends on the weight's shape only, the rest stay zero), the embedding table one slot per position (the rows
        this step touched), everything small (norm weights, biases, the projector) one slot per 8192-element chunk, summed by one launch per bucket."""
        fp, dev = self.fp, self.device
        Lyr = self.llm.num_hidden_layers
        by_gemm = {'head'} | {f'l{i}.{w}' for i in range(Lyr) for w in ('wqkv', 'wo', 'wgu', 'wdown')}
        self.norm_slot, self.norm_plan, n = {}, [], 0
        for (blo, bhi) in self.buckets:
            n = (n + 3) // 4 * 4                     # a bucket's slots start 16-byte aligned (vlaser_sum_partials reads them in 16-byte pieces)
            lo, chunks = n, []
            for name, shape, off in fp.specs:
                if not (blo <= off < bhi) or name == 'head_pad':         # head_pad: zero gradients by construction
                    continue
                numel = math.prod(shape)
                if name in by_gemm:
                    cap = ops.tn_sumsq_slots(shape[0], shape[1])
                    self.norm_slot[name] = (n, cap)
                    n += cap
                elif name == 'embed':
                    self.norm_slot[name] = (n, self.S_max)
                    n += self.S_max
                else:
                    chunks += [(off + c0, min(8192, numel - c0)) for c0 in range(0, numel, 8192)]
            tab = torch.tensor(chunks, dtype=torch.int64, device=dev).reshape(-1, 2) if chunks else None
            self.norm_plan.append((lo, n + len(chunks), n, tab))
            n += len(chunks)
        self.norm_parts = torch.zeros(n, dtype=F32, device=dev)

    def _ssq(self, name):
        """The slot slice the producer of gradient `name` fills, or None when this step takes the norm from the gradient buffer."""
        if not getattr(self, '_fused_norm', False):
            return None
        lo, cap = self.norm_slot[name]
        return self.norm_parts[lo:lo + cap]

    def _head_pads(self, R, Rp):
        """The lm_head's two row-padded operands ([ceil64(R), H] inputs, [ceil64(R), Vp] dlogits), allocated once per padded row count; their rows
        R..Rp must be zero (they enter the weight gradient's contraction): cleared when R changes -- the kernels only ever write rows < R."""
        if getattr(self, '_head_pad_Rp', None) != Rp:
            self._x_pad = torch.zeros(Rp, self.llm.hidden_size, dtype=BF, device=self.device)
            self._dlog_pad = torch.zeros(Rp, self.Vp, dtype=BF, device=self.device)
            self._head_pad_Rp, self._head_pad_R = Rp, R
        elif self._head_pad_R != R:
            lo = min(R, self._head_pad_R)
            self._x_pad[lo:].zero_()
            self._dlog_pad[lo:].zero_()
            self._head_pad_R = R
        return self._x_pad, self._dlog_pad

    def _wgrad_side(self, *args, want_done=False, **kw):
        """`_wgrad` on the weight-gradient stream, behind everything the compute stream has queued so far (its dY operand); `want_done`: returns the event
        that marks its end."""
        if self.wgrad_stream is None:
            self._wgrad(*args, **kw)
            return None
        ev = torch.cuda.Event()
        ev.record()
        ws = self.wgrad_stream
        ws.wait_event(ev)
        # `_wgrad` launches only through the C ABI: its launches are redirected by handle -- torch's stream context manager costs ~10 us of host time per
        # use, which made the backward host-bound (4 weight gradients x 28 layers: tools/micro/sft_phases.py)
        prev = ops.pin_stream(ws.cuda_stream)
        try:
            self._wgrad(*args, **kw)
        finally:
            ops.pin_stream(prev)
        if not want_done:
            return None
        done = torch.cuda.Event()
        done.record(ws)
        return done

    def _join_wgrad(self):
        """The compute stream waits for every weight gradient queued so far (before their dY buffers are overwritten / the bucket is handed on)."""
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def _zero_wgrad_pad(self, S):
        """Rows S..ceil64(S) of the four dY buffers the layer weight gradients contract over: zero, so that the padded TN GEMM may read whole
        64-row tiles.  Nothing writes those rows while S stays the same (every kernel is bounded by S), so this runs when S changes."""
        Sp = (S + 63) // 64 * 64
        if Sp != S and self._wgrad_pad_S != S:
            for b in (self.dh, self.dh2, self.dgu, self.dqkv):
                b[S:Sp].zero_()
            # the X-side operands are multiplied by those zero rows, so their pad rows only have to be FINITE -- they hold whatever an earlier, longer
            # sample left there; cleared as well when S changes (<= 63 rows per buffer) so that a stale Inf / NaN cannot turn 0 * x into NaN
            for b in (self.act, self.x2, self.ao, self.x1):
                b[:, S:Sp].zero_()
        self._wgrad_pad_S = S

    def _dgrad(self, dY, W, out, S, keep_slabs=False):
        """out[S,K] = dY[S,N] @ W[N,K], W as the forward stores it (NN GEMM); long contractions over few output tiles run split-K.  `keep_slabs`: a
        split-K product is NOT reduced -- returns (slabs, count) for a consumer that sums them itself (`vlaser_rmsnorm_bwd`'s dy_partials), else None."""
        Nin, Kout = W.shape
        # measured at S = 560, 1536 outputs (tools/micro/nn_lab.py): contraction <= 2048 -> one pass (11.5-14.6 us) beats split-K slabs +
        # their reduction (9-11 + 5 us); longer contractions (8960 / 17920) keep split-K
        sp = 1 if Nin <= 2048 else ops.gemm_splits(S, Kout, Nin, nn=True)
        if sp > 1:
            part = self.part[:sp * S * Kout]
            ops.gemm_nn(L.EPI_PARTIAL, dY, W, out_f32=part, k_splits=sp)
            if keep_slabs:
                return part, sp
            ops.reduce_norm(None, part, sp, S, Kout, out)
        else:
            ops.gemm_nn(L.EPI_NONE, dY, W, out=out)
        return None

    def _norm_wgrad(self, dy, x, out, S, Cc, mode=2, eps=1e-6):
        ops.colsum_mul(dy, x, self.col, S, Cc, mode, eps, self.rowstat)
        out.copy_(self.col[:Cc])

    # ------------------------------------------------------------------ forward of one layer with everything saved for its backward
    def _layer_forward(self, i, h_in, S, pos, x1_ready=False):
        llm = self.llm
        v = self.fp.view
        H, I = llm.hidden_size, llm.intermediate_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        x1, x2, h2, q, ao, gu, act = self._saved(i, S)
        j = 0 if self.recompute else i
        if not x1_ready:            # (the forward pass gets x1 of layers 1 .. L-1 from the previous layer's down_proj seam)
            ops.rmsnorm(h_in, v[f'l{i}.ln_in'], llm.rms_norm_eps, out=x1)
        ops.gemm(L.EPI_QKV_ROPE, x1, v[f'l{i}.wqkv'], bias=v[f'l{i}.bqkv'], q_out=q, k_cache=self.cache.k[j], vt_cache=self.cache.vt[j],
                 rope_cos=self.rope[0], rope_sin=self.rope[1], pos_ids=pos, n_q_heads=nq, n_kv_heads=nkv, s_max=self.cache.s_max,
                 tok_per_batch=S, slot_base=0)
        ks, vs = self.cache.strides()
        ops.attn_prefill(q, self.cache.k[j], self.cache.vt[j], ao, 1, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), ks, vs,
                         (S * nq * hd, nq * hd), self.cache.s_max, hd ** -0.5, L.ATTN_CAUSAL, lse_out=self.lse[j])
        sp = ops.gemm_splits(S, H, nq * hd)
        ops.gemm(L.EPI_PARTIAL, ao, v[f'l{i}.wo'], out_f32=self.part, k_splits=sp)
        ops.reduce_norm(h_in, self.part, sp, S, H, h2, x2, norm=1, norm_w=v[f'l{i}.ln_post'], eps=llm.rms_norm_eps)
        ops.gemm(L.EPI_SWIGLU, x2, v[f'l{i}.wgu'], out=act, aux_out=gu, ld_aux=gu.stride(0))      # act + the pre-activations swiglu's backward needs
        return x1, x2, h2, q, ao, gu, act

    def _saved(self, i, S):
        j = 0 if self.recompute else i
        return (self.x1[j, :S], self.x2[j, :S], self.h2[j, :S], self.q[j, :S], self.ao[j, :S], self.gu[j, :S], self.act[j, :S])

    def _layer_out(self, i, S, h2, act, h_out, norm_w=None, x_out=None):
        llm = self.llm
        sp = ops.gemm_splits(S, llm.hidden_size, llm.intermediate_size)
        ops.gemm(L.EPI_PARTIAL, act, self.fp.view[f'l{i}.wdown'], out_f32=self.part, k_splits=sp)
        if norm_w is None:
            ops.reduce_norm(h2, self.part, sp, S, llm.hidden_size, h_out)
        else:
            ops.reduce_norm(h2, self.part, sp, S, llm.hidden_size, h_out, x_out, norm=1, norm_w=norm_w, eps=llm.rms_norm_eps)

    # ------------------------------------------------------------------ forward (loss) + backward (grads into self.fp.g)
    def forward_backward(self, pixel_values, input_ids, labels, image_flags=None, on_bucket_ready=None):
        """Loss + gradients of ONE sample.  The ~850 launches of the call all go to the stream that is current at entry: its handle is looked up once and
        pinned for the C-ABI launches (`torch.cuda.current_stream()` per launch was 30 % of the call's host time: tools/micro/sft_host_profile.py); the
        bucket callbacks, which switch streams themselves, run with the pin lifted."""
        with self._on_main():
            main = torch.cuda.current_stream().cuda_stream
            prev = ops.pin_stream(main)
            cb = on_bucket_ready
            if cb is not None:
                def on_bucket_ready(b, _cb=cb):
                    ops.pin_stream(prev)
                    try:
                        _cb(b)
                    finally:
                        ops.pin_stream(main)
            try:
                return self._forward_backward(pixel_values, input_ids, labels, image_flags, on_bucket_ready)
            finally:
                ops.pin_stream(prev)

    def _forward_backward(self, pixel_values, input_ids, labels, image_flags=None, on_bucket_ready=None):
        cfg, llm, dev = self.cfg, self.llm, self.device
        v, gv = self.fp.view, self.fp.gview
        B, S = input_ids.shape
        if B != 1:
            raise ValueError('forward_backward takes ONE sample; per-device batches / gradient accumulation go through train_step()')
        if S > self.S_max:
            raise ValueError(f'sequence of {S} tokens exceeds max_seq_len={self.S_max}')
        # index bookkeeping on the HOST copy of the (tiny) id / label tensors -- the collator hands CPU tensors over; CUDA inputs cost
        # one device->host copy here.  Nothing below reads a device value back, so the whole step is queued without a host sync.
        ids_h = input_ids.detach().to('cpu', torch.int64).reshape(1, S)
        lab_h = labels.detach().to('cpu', torch.int64).reshape(-1)
        if int(ids_h.min()) < 0 or int(ids_h.max()) >= llm.vocab_size:
            raise ValueError(f'input_ids outside [0, {llm.vocab_size})')
        tgt_h = torch.full((S,), -100, dtype=torch.int64)
        tgt_h[:S - 1] = lab_h[1:]                                      # shift: position t predicts token t+1
        rows_h = (tgt_h != -100).nonzero().flatten()
        R = int(rows_h.numel())
        n_img = int((ids_h == self.img_context_token_id).sum())
        H, I, V = llm.hidden_size, llm.intermediate_size, llm.vocab_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        Lyr = llm.num_hidden_layers
        ids = self._h2d(ids_h)
        pos = self.pos_all[:S]
        # ---- vision tower (frozen) + trainable projector (mlp1), with the intermediates mlp1's backward needs
        T = pixel_values.shape[0]
        self._grow_tiles(T)
        pv = pixel_values.to(dev)
        if pv.dtype != BF:
            pvb = torch.empty(pv.shape, dtype=BF, device=dev)
            ops.cast_f32_bf16(pv.float().contiguous(), pvb)
            pv = pvb
        vit_w = self.vit
        vit_w.m0w, vit_w.m0b = v['mlp1.m0w'], v['mlp1.m0b']       # the projector weights are the trainable views
        vit_w.m1w, vit_w.m1b, vit_w.m3w, vit_w.m3b = v['mlp1.m1w'], v['mlp1.m1b'], v['mlp1.m3w'], v['mlp1.m3b']
        # the frozen encoder needs no parameter of this step: it starts right away, under the previous step's AdamW (r04: it used to sit behind the
        # wait below, 1.1 ms of an idle compute stream per step -- tools/micro/sft_timeline.py)
        vit_w.forward(pv, project=False)                           # leaves the last hidden state in vit_w.h
        self._wait_params(len(self.buckets) - 1)                   # embed + projector bucket
        # every gradient tensor is fully overwritten by its wgrad / column-sum kernel each step, except the embedding rows
        # (scatter-add over the text tokens): only that slice is cleared (466 MB instead of the whole 3.6 GB buffer) -- after the
        # wait above: the previous step's AdamW may still be reading this bucket's gradients on the optimizer stream
        # r04: on one rank with one sample per step the only non-zero rows are the ones the PREVIOUS call scattered into: those rows are cleared (a 560-row
        # fill) instead of the table; anything else that wrote the bucket (gradient accumulation's finalize, a reduce-scatter) resets to the full clear
        # (only inside train_step's one-sample path -- `_fused_norm` -- where nothing but this function writes the gradient buffer between two calls; a direct
        # forward_backward() call, whose caller may do anything to fp.g in between, clears the table)
        touched = getattr(self, '_embed_touched', None)
        own = getattr(self, '_fused_norm', False) and os.environ.get('VLASER_SFT_EMBED_FULL_CLEAR') != '1'
        if touched is not None and own:
            gv['embed'].index_fill_(0, touched, 0)
        else:
            gv['embed'].zero_()
        self._embed_touched = ids.reshape(-1).clamp(0, gv['embed'].shape[0] - 1) if own else None
        nt = T * cfg.num_image_token
        C1 = cfg.vision.hidden_size
        G_ = cfg.vision.image_size // cfg.vision.patch_size
        ps_raw, ps_ln, z1, g1, feat = self.ps_raw[:nt], self.ps_ln[:nt], self.z1[:nt], self.g1[:nt], self.feat[:nt]
        ops.pixel_shuffle(vit_w.h, ps_raw, T, G_, C1, 1 if cfg.ps_version == 'v1' else 0)
        ops.pixel_shuffle_ln(vit_w.h, v['mlp1.m0w'], v['mlp1.m0b'], ps_ln, T, G_, C1, 1e-5, 1 if cfg.ps_version == 'v1' else 0)
        ops.gemm(L.EPI_BIAS_GELU, ps_ln, v['mlp1.m1w'], out=g1, bias=v['mlp1.m1b'], aux_out=z1, ld_aux=z1.stride(0))      # g1 + the pre-activation z1 (GELU backward)
        ops.gemm(L.EPI_BIAS, g1, v['mlp1.m3w'], out=feat, bias=v['mlp1.m3b'])
        if image_flags is not None:
            keep = self._h2d((image_flags.detach().to('cpu').reshape(-1) == 1).nonzero().flatten())          # indices of the real tiles
            feat_used = feat.view(T, cfg.num_image_token, H).index_select(0, keep).reshape(-1, H)
        else:
            feat_used = feat
        if n_img != feat_used.shape[0]:
            raise RuntimeError(f'shape mismatch: {n_img} <IMG_CONTEXT> tokens vs {feat_used.shape[0]} visual tokens')
        # ---- embeddings + visual-token scatter
        h0 = self.h_in[0, :S]
        ops.embed_merge(ids, v['embed'], feat_used, h0, self.img_context_token_id, cfg.pad_token_id, False, self.rank_ws)
        # ---- forward through the layers (saved activations per layer, or only the layer inputs when recompute=True)
        # r04: the down_proj seam (split-K reduce + residual) also applies the following layer's input_layernorm into its x1 slot, as the inference prefill
        # does: one launch less per layer (`VLASER_SFT_NO_SEAM_NORM=1`: A/B)
        seam = os.environ.get('VLASER_SFT_NO_SEAM_NORM') != '1'
        xn = self.xn[:S]
        bucket_of = lambda li: 1 + (Lyr - 1 - li) // self.bucket_layers          # layer buckets hold the layers in reverse order (bucket 1 = last layers)
        x1_ready = False
        for i in range(Lyr):
            self._wait_params(bucket_of(i))
            _, _, h2, _, _, _, act = self._layer_forward(i, self.h_in[i, :S], S, pos, x1_ready=x1_ready)
            # (only inside a parameter bucket: across a boundary the seam would have to wait for the NEXT bucket's AdamW / all-gather one GEMM early)
            x1_ready = seam and i + 1 < Lyr and bucket_of(i + 1) == bucket_of(i)
            if x1_ready:
                self._layer_out(i, S, h2, act, self.h_in[i + 1, :S], norm_w=v[f'l{i + 1}.ln_in'], x_out=self._saved(i + 1, S)[0])
            else:
                self._layer_out(i, S, h2, act, self.h_in[i + 1, :S])
        h_fin = self.h_in[Lyr, :S]
        # ---- loss head on the labelled rows only (rows with label -100 contribute neither loss nor gradient)
        rows = self._h2d(rows_h) if R else None
        self._wait_params(0)                                         # lm_head + final norm
        ops.rmsnorm(h_fin, v['norm'], llm.rms_norm_eps, out=xn)
        if R == 0:
            # no supervised position on this rank: zero loss, zero gradients -- but the SAME collective sequence as every other
            # rank (one reduce-scatter per bucket, in backward order)
            self.fp.g.zero_()
            self._embed_touched = None
            if getattr(self, '_fused_norm', False):
                self.norm_parts.zero_()            # nothing wrote this step's slots: the producers' partial sums of the PREVIOUS step must not become this step's norm
            if on_bucket_ready:
                for b in range(len(self.buckets)):
                    on_bucket_ready(b)
            return torch.zeros((), device=dev)
        # the R supervised rows, in buffers of ceil64(R) rows (zero pad): the head's weight gradient then contracts over whole 64-row tiles and runs on
        # the LDS-DMA pipeline (vlaser_gemm_tn_lds; r03: 237 -> ~110 us for the [151 674 x 1536] gradient)
        Rp = (R + 63) // 64 * 64
        x_pad, dlog_pad = self._head_pads(R, Rp)
        torch.index_select(xn, 0, rows, out=x_pad[:R])
        x_rows = x_pad[:R]
        t_rows = self._h2d(tgt_h.index_select(0, rows_h))
        logits = ops.linear(x_rows, v['head'], epi=L.EPI_F32)         # [R, V] fp32
        loss_rows = torch.empty(R, dtype=F32, device=dev)
        lse = torch.empty(R, dtype=F32, device=dev)
        ops.ce_rows(logits, t_rows, loss_rows, lse)
        loss = loss_rows.sum() / R
        # ================================================================ backward
        # (every bucket's `_wait_params` has been issued by now: the previous step's AdamW / all-gathers no longer touch fp.p or fp.g)
        dlog = dlog_pad[:R]
        ops.ce_dlogits(logits, lse, t_rows, dlog, 1.0 / R)
        dx_rows = torch.empty(R, H, dtype=BF, device=dev)
        # dX = dlogits @ W_head (contraction over Vp: dlogits and the pad rows are zero there): R rows against a 466 MB weight is a stream, and the stream
        # wants many slices in flight -- 30 split-K slices 114 us (4.1 TB/s), the generic chooser's 6 slices 210 us (tools/micro/head_lab.py)
        sp = max(d for d in range(1, int(os.environ.get('VLASER_SFT_HEAD_SPLITS', '32')) + 1) if self.Vp % (64 * d) == 0)
        if sp > 8 and sp * R * H <= self.part.numel():
            part = self.part[:sp * R * H]
            ops.gemm_nn(L.EPI_PARTIAL, dlog, self.head_full, out_f32=part, k_splits=sp)
            ops.reduce_norm(None, part, sp, R, H, dx_rows)          # fp32 sum of the slabs in slab order, one rounding to bf16
        else:
            self._dgrad(dlog, self.head_full, dx_rows, R)
        ops.gemm_tn(dlog_pad[:, :V], x_pad, gv['head'], sumsq_part=self._ssq('head'))             # dW_head = dlogits^T @ x over ceil64(R) rows (zero pad rows; dlogits rows are padded to Vp columns)
        dxn = self.dx[:S]
        dxn.zero_()
        dxn.index_copy_(0, rows, dx_rows)
        self._zero_wgrad_pad(S)
        dh = self.dh[:S]
        ops.rmsnorm_bwd(dxn, h_fin, v['norm'], None, dh, S, H, llm.rms_norm_eps, dw_out=gv['norm'], dw_ws=self.normw_ws)
        if on_bucket_ready:
            on_bucket_ready(0)
        G = nq // nkv
        sm = self.cache.s_max
        scale = hd ** -0.5
        bucket_of_layer = {}
        layers_rev = list(reversed(range(Lyr)))
        for j, li in enumerate(layers_rev):
            bucket_of_layer[li] = 1 + j // self.bucket_layers
        nslot = 0                       # norm-weight gradients of the current bucket whose partials wait in self.normw_multi
        for i in reversed(range(Lyr)):
            h_in = self.h_in[i, :S]
            self._join_wgrad()          # the previous layer's weight gradients still read dgu / dh2 / dqkv (and, with recompute, the one activation slot)
            x1, x2, h2, q, ao, gu, act = self._layer_forward(i, h_in, S, pos) if self.recompute else self._saved(i, S)
            kslot = 0 if self.recompute else i
            dact, dgu, dx, dh2, dao = self.dact[:S], self.dgu[:S], self.dx[:S], self.dh2[:S], self.dao[:S]
            # MLP: h3 = h2 + act Wd^T ; act = silu(g) u ; [g|u] = x2 Wgu^T ; x2 = rms(h2) w_post
            if H <= 2048 and os.environ.get('VLASER_SFT_NO_FUSED_SWIGLU_BWD') != '1':
                # d(act) = dh @ W_down with swiglu's backward as the GEMM epilogue: d(gate), d(up) straight from the accumulators
                # (the stand-alone pass read d(act) and the pre-activations back and wrote 2 I columns per row)
                ops.gemm_nn(L.EPI_SWIGLU_BWD, dh, v[f'l{i}.wdown'], out=dgu, res=gu)
            else:
                self._dgrad(dh, v[f'l{i}.wdown'], dact, S)
                ops.swiglu_bwd(gu, dact, dgu, S, I)
            ev_wdown = self._wgrad_side(dh, act, gv[f'l{i}.wdown'], S, padded=True, ssq=self._ssq(f'l{i}.wdown'), want_done=True)
            slabs = self._dgrad(dgu, v[f'l{i}.wgu'], dx, S, keep_slabs=self._slab_norm)
            self._wgrad_side(dgu, x2, gv[f'l{i}.wgu'], S, padded=True, ssq=self._ssq(f'l{i}.wgu'))
            nw = self.normw_multi
            dw_kw = dict(dw_out=gv[f'l{i}.ln_post'], dw_ws=self.normw_ws) if nw is None else dict(dw_ws=nw[nslot * self.normw_slot:(nslot + 1) * self.normw_slot])
            if slabs is not None:       # r04: the split-K slabs of the gate/up dgrad go straight into the norm's backward (one launch less per layer, same bits)
                ops.rmsnorm_bwd(None, h2, v[f'l{i}.ln_post'], dh, dh2, S, H, llm.rms_norm_eps, dy_partials=slabs[0], n_partials=slabs[1], **dw_kw)
            else:
                ops.rmsnorm_bwd(dx, h2, v[f'l{i}.ln_post'], dh, dh2, S, H, llm.rms_norm_eps, **dw_kw)
            nslot += nw is not None
            # attention block: h2 = h_in + ao Wo^T
            self._dgrad(dh2, v[f'l{i}.wo'], dao, S)
            self._wgrad_side(dh2, ao, gv[f'l{i}.wo'], S, padded=True, ssq=self._ssq(f'l{i}.wo'))
            Kc, VTc = self.cache.k[kslot, 0], self.cache.vt[kslot, 0]         # [nkv, s_max, hd], [nkv, hd, s_max]
            if self.attn_bwd_mode == 'fused':
                ops.attn_bwd(q, Kc, VTc, ao, dao, self.lse[kslot], self.delta_ws, self.dq[:S], self.dk[:S], self.dv[:S], S, nq, nkv, sm, scale, head_dim=hd)
            else:
                self._attn_backward(q, Kc, VTc, dao, ao, S, nq, G, hd, sm, scale)
            dqkv = self.dqkv[:S]
            ops.rope_bwd_pack(self.dq[:S], self.dk[:S], self.dv[:S], self.rope[0], self.rope[1], pos, dqkv, S, nq, nkv, kv_per_q_head=True)
            self._dgrad(dqkv, v[f'l{i}.wqkv'], dx, S)
            self._wgrad_side(dqkv, x1, gv[f'l{i}.wqkv'], S, bias_out=gv[f'l{i}.bqkv'], padded=True, ssq=self._ssq(f'l{i}.wqkv'))
            if ev_wdown is not None:
                torch.cuda.current_stream().wait_event(ev_wdown)          # the down_proj weight gradient read dh, which the next kernel overwrites
            dw_kw = dict(dw_out=gv[f'l{i}.ln_in'], dw_ws=self.normw_ws) if nw is None else dict(dw_ws=nw[nslot * self.normw_slot:(nslot + 1) * self.normw_slot])
            ops.rmsnorm_bwd(dx, h_in, v[f'l{i}.ln_in'], dh2, dh, S, H, llm.rms_norm_eps, **dw_kw)
            nslot += nw is not None
            last_of_bucket = i == 0 or bucket_of_layer[i - 1] != bucket_of_layer[i]
            if nw is not None and last_of_bucket:
                # the bucket's layers are i .. i + nslot/2 - 1, visited last to first: slots = (ln_post, ln_in) of layer i + nslot/2 - 1, ..., of layer i
                key = (i, nslot)
                if key not in self._normw_off:
                    offs = []
                    for li in reversed(range(i, i + nslot // 2)):
                        offs += [self.fp.offset_of(f'l{li}.ln_post'), self.fp.offset_of(f'l{li}.ln_in')]
                    self._normw_off[key] = torch.tensor(offs, dtype=torch.int64, device=dev)
                ops.colsum_partials_multi(nw, self.normw_slot, nslot, (S + 3) // 4, H, self.fp.g, self._normw_off[key])
                nslot = 0
            if on_bucket_ready and last_of_bucket:
                self._join_wgrad()
                on_bucket_ready(bucket_of_layer[i])
        self._join_wgrad()
        # ---- embeddings (text rows) and projector (image rows)
        ops.embed_scatter_add(ids, self.rank_ws, dh, gv['embed'], S, H, sumsq_part=self._ssq('embed'))
        img_rows = self._h2d((ids_h.reshape(-1) == self.img_context_token_id).nonzero().flatten())
        dfeat_used = dh.index_select(0, img_rows).contiguous()
        dvit = self.dvit[:nt]
        if image_flags is not None:
            dvit.zero_()
            dvit.view(T, cfg.num_image_token, H).index_copy_(0, keep, dfeat_used.view(-1, cfg.num_image_token, H))
        else:
            dvit.copy_(dfeat_used)
        dg1, dz1, dln = self.dg1[:nt], self.dz1[:nt], self.dln[:nt]
        ops.gemm_nn(L.EPI_NONE, dvit, v['mlp1.m3w'], out=dg1)
        self._wgrad(dvit, g1, gv['mlp1.m3w'], nt, bias_out=gv['mlp1.m3b'])
        ops.gelu_bwd(z1, dg1, dz1)
        self._wgrad(dz1, ps_ln, gv['mlp1.m1w'], nt, bias_out=gv['mlp1.m1b'])
        ops.gemm_nn(L.EPI_NONE, dz1, v['mlp1.m1w'], out=dln)
        C4 = C1 * 4
        ops.colsum_mul(dln, ps_raw, self.col, nt, C4, 3, 1e-5, self.rowstat)
        gv['mlp1.m0w'].copy_(self.col[:C4])
        ops.colsum_mul(dln, None, self.col, nt, C4, 0, ws=self.rowstat)
        gv['mlp1.m0b'].copy_(self.col[:C4])
        if on_bucket_ready:
            on_bucket_ready(len(self.buckets) - 1)
        return loss

    def _attn_backward(self, q, Kc, VTc, dao, ao, S, nq, G, hd, sm, scale):
        """Causal attention backward through materialised score matrices, one BLOCK of query rows at a time: for rows [q0, q1) only the keys
        [0, q1) are visible, so the block's matrices are [heads, q1 - q0, q1] -- S <= attn_bwd_block is one block (the r02 path, unchanged);
        longer sequences accumulate dK / dV over the blocks in fp32 (vlaser_grad_accumulate) and round once.  Leaves dQ in self.dq and one
        dK / dV partial per Q head in self.dk / self.dv (summed over the kv group by rope_bwd_pack)."""
        QB = self.sc.shape[1]
        nblk = -(-S // QB)
        if nblk > 1 and self.dkv_acc is None:
            self.dkv_acc = torch.zeros(2, self.S_max, nq * hd, dtype=F32, device=self.device)
        q1_prev = 0
        for bi in range(nblk):
            q0, q1 = bi * QB, min(S, (bi + 1) * QB)
            nb, kp = q1 - q0, (q1 + 63) // 64 * 64
            n = nq * nb * kp
            sc, dP = self.sc.view(-1)[:n].view(nq, nb, kp), self.dP.view(-1)[:n].view(nq, nb, kp)
            P, dS = self.P.view(-1)[:n].view(nq, nb, kp), self.dS.view(-1)[:n].view(nq, nb, kp)
            ops.gemm_raw(L.EPI_F32, q[q0:], Kc, sc, nb, q1, hd, nq * hd, hd, kp, batch=nq, a_bs=hd, w_bs=sm * hd, o_bs=nb * kp, w_group=G)     # Q K^T
            # dP = dO V^T and (below) dQ = dS K in the NN form: V^T [hd, keys] and K [keys, hd] are read as the cache holds them (r01/r02
            # transposed both per layer); dS is zero beyond the causal range, so cache rows past S only need to be finite
            ops.gemm_raw_nn(L.EPI_F32, dao[q0:], VTc, dP, nb, kp, hd, nq * hd, sm, kp, batch=nq, a_bs=hd, w_bs=hd * sm, o_bs=nb * kp, w_group=G)   # dO V^T
            ops.attn_bwd_pds_masked(sc, dP, dao[q0:], ao[q0:], P, dS, nq, nb, kp, hd, scale, True, kp, q0)      # P = softmax(S), dS = P o (dP - D) * scale
            ops.gemm_raw_nn(L.EPI_NONE, dS, Kc, self.dq[q0:], nb, hd, kp, kp, hd, nq * hd, batch=nq, a_bs=nb * kp, w_bs=sm * hd, o_bs=hd, w_group=G)  # dQ = dS K
            # dK[kvh] = sum_g dS[kvh*G+g]^T Q_g, dV[kvh] = sum_g P[kvh*G+g]^T dO_g: contraction along the rows (q) of both operands,
            # summed over the q heads of the kv group, straight from dS / P [head, q, k] and q / dO [q, head*hd]
            # one TN GEMM per Q head (108 workgroups instead of 18 serial ones); the sum over the group happens in rope_bwd_pack
            ops.gemm_tn_grouped(dS, q[q0:], self.dk, q1, hd, nb, kp, nq * hd, nq * hd, 1, 0, 0, nq, nb * kp, hd, hd)
            ops.gemm_tn_grouped(P, dao[q0:], self.dv, q1, hd, nb, kp, nq * hd, nq * hd, 1, 0, 0, nq, nb * kp, hd, hd)
            if nblk > 1:
                last = bi == nblk - 1
                for g_, acc in ((self.dk, self.dkv_acc[0]), (self.dv, self.dkv_acc[1])):
                    if q1_prev > 0:
                        ops.grad_accumulate(g_[:q1_prev], acc[:q1_prev], 1.0, False, last)       # keys earlier blocks already reached
                    ops.grad_accumulate(g_[q1_prev:q1], acc[q1_prev:q1], 1.0, True, last)        # keys first reached by this block
                q1_prev = q1

    # ------------------------------------------------------------------ optimizer / data parallel
    def _norm_bucket(self, b):
        """Single rank: bucket b's share of the squared gradient norm as soon as its gradients are complete, on the optimizer stream
        (a one-workgroup-per-CU streaming kernel under the backward's GEMMs) instead of a 0.6 ms pass between backward and AdamW.
        Buckets complete in index order, so the partial sums are added in the same order as the loop in optimizer_step."""
        self._norm_buckets_seen = getattr(self, '_norm_buckets_seen', 0) + 1
        if getattr(self, '_fused_norm', False):
            # r04: the producers left the partial sums (weight-gradient GEMM epilogues, the touched embedding rows); the small tensors are summed here in
            # one launch, then one workgroup adds the bucket's slots in a fixed order -- on the compute stream: two ~5 us launches per bucket instead of
            # 3.6 GB re-read under the backward's GEMMs.  Every AdamW of the previous step was waited for by the forward, so nothing reads gnorm2 now.
            lo, hi, c_lo, tab = self.norm_plan[b]
            if tab is not None:
                ops.sumsq_chunks(self.fp.g, tab, self.norm_parts[c_lo:hi])
            ops.sum_partials(self.norm_parts[lo:hi], self.gnorm2, accumulate=b != 0)
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.opt_stream):
            self.opt_stream.wait_event(ev)
            if b == 0:
                self.gnorm2.zero_()          # on this stream: behind the previous step's AdamW, which reads the old norm here
            s_lo, s_hi, _ = self.shards[b]
            if s_hi > s_lo:
                ops.sumsq(self.fp.g[s_lo:s_hi], self.gnorm2, self.sumsq_ws)

    def _exchange_bucket(self, b):
        """mean reduce-scatter of bucket b on the comm stream (RCCL), issued as soon as its gradients are complete."""
        if not self.dp_active:
            return
        lo, hi = self.buckets[b]
        s_lo, s_hi, per = self.shards[b]
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            dp.reduce_scatter_mean(self.fp.g, self.buckets[b], self.shards[b], self.pg, capi=self.capi)

    def optimizer_step(self, lr=None):
        with self._on_main():
            return self._optimizer_step(lr)

    def _optimizer_step(self, lr=None):
        lr = self.lr if lr is None else lr
        self.step_count += 1
        if self.dp_active:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        # global gradient norm over the (reduced) shards -> clip factor
        if getattr(self, '_norm_early', False):
            torch.cuda.current_stream().wait_stream(self.opt_stream)     # the per-bucket sums queued from the backward (`_norm_bucket`)
            self._norm_early = False
        else:
            self.gnorm2.zero_()
            for (s_lo, s_hi, _) in self.shards:
                if s_hi > s_lo:
                    ops.sumsq(self.fp.g[s_lo:s_hi], self.gnorm2, self.sumsq_ws)
        if self.capi is not None:                                  # the 4-byte sum on the package's own communicator / stream too
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            self.capi.all_reduce_sum(self.gnorm2)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        elif self.dp_active:
            torch.distributed.all_reduce(self.gnorm2, group=self.pg)
        gnorm = self.gnorm2.sqrt()                                  # device tensor: reading it is the caller's (only) sync

        def adamw_bucket(b):
            (s_lo, s_hi, per), o = self.shards[b], self.shard_off[b]
            if s_hi > s_lo:
                n = s_hi - s_lo
                # clip factor max_norm / (||g|| + 1e-6) resolved inside the kernel from the device-resident squared norm
                ops.adamw_clipped(self.fp.p[s_lo:s_hi], self.master[o:o + n], self.m[o:o + n], self.v[o:o + n], self.fp.g[s_lo:s_hi], lr, self.betas[0],
                                  self.betas[1], self.eps, self.wd, 1.0, self.gnorm2, self.max_grad_norm, self.step_count)

        if not self.dp_active and self.overlap_optimizer:
            # pipelined with the NEXT step: bucket by bucket on `opt_stream`, embed + projector first, lm_head last; the next forward
            # waits per bucket (`_wait_params`); by the time its backward starts every bucket has been waited for.  The
            # update is therefore still in flight when this returns: `wait_optimizer()` before touching parameter buffers directly.
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.opt_stream):
                self.opt_stream.wait_event(ev)
                for b in reversed(range(len(self.buckets))):
                    adamw_bucket(b)
                    self.ag_events[b] = torch.cuda.Event()
                    self.ag_events[b].record()
            return gnorm
        if self.dp_active and self.overlap_allgather and os.environ.get('VLASER_SFT_DP_SERIAL_ADAMW') != '1':
            # ZeRO-1 (r03): this rank's shard of every bucket is updated ON THE COMM STREAM right in front of that bucket's parameter all-gather, bucket by
            # bucket in the order the NEXT forward consumes them (embed + projector first, lm_head last); the forward waits per bucket (`_wait_params`), so
            # AdamW (1/N of the parameters) and the exchange both run under the frozen ViT and the earlier layers instead of between two steps
            # (world 1 with the exchange forced on: 31.0 -> 28.9 ms per step, profiles/r03dp_force_dp_world1.md)
            ev = torch.cuda.Event()
            ev.record()
            if self.main_stream is not None:
                # capi mode with CU masks: the comm stream owns only RCCL's share of the CUs -- the shard AdamW (HBM-bound) runs on the compute-masked optimizer stream,
                # bucket by bucket, and each bucket's all-gather follows it on the comm stream behind an event
                self.opt_stream.wait_event(ev)
                for b in reversed(range(len(self.buckets))):
                    with torch.cuda.stream(self.opt_stream):
                        adamw_bucket(b)
                        eb = torch.cuda.Event()
                        eb.record()
                    with torch.cuda.stream(self.comm_stream):
                        self.comm_stream.wait_event(eb)
                        dp.all_gather_params(self.fp.p, self.buckets[b], self.shards[b], self.pg, capi=self.capi)
                        self.ag_events[b] = torch.cuda.Event()
                        self.ag_events[b].record()
                return gnorm
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                for b in reversed(range(len(self.buckets))):
                    adamw_bucket(b)
                    dp.all_gather_params(self.fp.p, self.buckets[b], self.shards[b], self.pg, capi=self.capi)
                    self.ag_events[b] = torch.cuda.Event()
                    self.ag_events[b].record()
            return gnorm
        for b in range(len(self.buckets)):
            adamw_bucket(b)
        if self.dp_active:
            # all-gather the updated bf16 parameters bucket by bucket on the comm stream, in the order the NEXT forward consumes them
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                for b in reversed(range(len(self.buckets))):
                    dp.all_gather_params(self.fp.p, self.buckets[b], self.shards[b], self.pg, capi=self.capi)
                    self.ag_events[b] = torch.cuda.Event()
                    self.ag_events[b].record()
            if not self.overlap_allgather:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        return gnorm

    def step(self, pixel_values, input_ids, labels, image_flags=None, lr=None, total_steps=None):
        """One optimizer step on one per-device batch ([B, S] ids / labels as the collator pads them, B >= 1).  With `total_steps` the
        learning rate follows the launcher's cosine schedule."""
        return self.train_step([(pixel_values, input_ids, labels, image_flags)], lr=lr, total_steps=total_steps)

    def _split_batch(self, pixel_values, input_ids, labels, image_flags, attention_mask=None):
        """A collated per-device batch (pad_data_collator.py:57-116: ids right-padded with 0, labels with -100, tiles of all samples
        concatenated, `image_flags` 0 for dummy tiles) -> its samples, each trimmed to its own length, with the number of supervised
        positions R_b (CrossEntropyLoss averages over ALL supervised tokens of the batch: sample b weighs R_b / sum R).
        A sample's length is the collator's `attention_mask` when given; otherwise the last position that holds a non-pad id OR a
        supervised label (id 0 is a real Qwen token, "!": a sample ending in it must keep its trailing labels)."""
        ids_h = input_ids.detach().to('cpu', torch.int64)
        lab_h = labels.detach().to('cpu', torch.int64)
        am_h = None if attention_mask is None else attention_mask.detach().to('cpu').reshape(ids_h.shape) != 0
        B = ids_h.shape[0]
        nt = self.cfg.num_image_token
        flags = None if image_flags is None else image_flags.detach().to('cpu').reshape(-1)
        T = pixel_values.shape[0]
        out, t0 = [], 0
        for b in range(B):
            nz = (am_h[b] if am_h is not None else ((ids_h[b] != 0) | (lab_h[b] != -100))).nonzero().flatten()
            Lb = int(nz[-1]) + 1 if nz.numel() else 1
            need = int((ids_h[b, :Lb] == self.img_context_token_id).sum()) // nt          # real tiles of this sample
            t1, got = t0, 0
            while t1 < T and got < need:
                got += 1 if (flags is None or flags[t1] == 1) else 0
                t1 += 1
            if need == 0:                           # text-only sample: the dataset attaches ONE dummy tile with image_flags 0 (pure_text_get_item)
                if flags is None or t1 >= T or flags[t1] != 0:
                    raise ValueError('a text-only sample must come with its dummy tile (image_flags == 0), as the reference dataset emits it')
                t1 += 1
            if got < need:
                raise ValueError(f'sample {b} has {need * nt} <IMG_CONTEXT> tokens but only {got} real tiles are left in pixel_values')
            if b == B - 1 and t1 != T:
                raise ValueError(f'{T - t1} tiles of pixel_values belong to no sample')
            R = int((lab_h[b, 1:Lb] != -100).sum())
            out.append((pixel_values[t0:t1], ids_h[b:b + 1, :Lb], lab_h[b:b + 1, :Lb], None if flags is None else flags[t0:t1].reshape(-1, 1), R))
            t0 = t1
        return out

    def _accumulate_bucket(self, b, w, first, last):
        """Sample-wise gradient accumulation, bucket by bucket as the backward completes them: acc (fp32) += w * g; on the LAST sample
        the bucket is rounded back to bf16 and handed to the exchange (one reduce-scatter per bucket per step, like DDP's no_sync /
        DeepSpeed's gradient-accumulation boundary: train.py:470-482, zero_stage1_config.json)."""
        lo, hi = self.buckets[b]
        ops.grad_accumulate(self.fp.g[lo:hi], self.gacc[lo:hi], w, first, last)
        self._embed_touched = None              # (the finalised bucket holds every sample's rows: the next call clears the whole table)
        if last:
            self._exchange_bucket(b)

    def train_step(self, micro_batches, lr=None, total_steps=None):
        """One optimizer step over `len(micro_batches)` gradient-accumulation micro-batches, each a per-device batch
        `(pixel_values, input_ids [B,S], labels [B,S], image_flags[, attention_mask])` -- the reference launcher's PER_DEVICE_BATCH_SIZE x GRADIENT_ACC
        (…2nd_finetune_full.sh:5-6,49-50).  HF Trainer semantics: each micro-batch's loss is the mean over its supervised tokens,
        divided by the number of accumulation steps; gradients add up.  The kernels run one sample at a time, so sample b of a
        micro-batch enters with weight (R_b / R_batch) / n_micro; partial sums live in an fp32 accumulator."""
        if lr is None and total_steps is not None:
            lr = cosine_lr(self.step_count, total_steps, self.lr)
        GA = len(micro_batches)
        work = []
        for mb in micro_batches:
            pv, ids, lab = mb[0], mb[1], mb[2]
            fl = mb[3] if len(mb) > 3 else None
            if ids.shape[0] == 1 and GA == 1:
                work.append((pv, ids, lab, fl, 1.0))
                continue
            smp = self._split_batch(pv, ids, lab, fl, mb[4] if len(mb) > 4 else None)
            Rt = sum(x[4] for x in smp)
            for (pvb, idb, lbb, flb, Rb) in smp:
                work.append((pvb, idb, lbb, flb, (Rb / Rt if Rt else 0.0) / GA))
        return self._run_weighted(work, lr)

    def _run_weighted(self, work, lr):
        with self._on_main():
            return self._run_weighted_main(work, lr)

    def _run_weighted_main(self, work, lr):
        """Samples (pixel_values, ids, labels, image_flags, weight) -> accumulated weighted gradients -> exchange -> optimizer step."""
        self._norm_early = False                   # only trusted when THIS step's backward queued every bucket's partial norm (below)
        if len(work) == 1 and work[0][4] == 1.0:
            early = not self.dp_active and self.overlap_optimizer and os.environ.get('VLASER_SFT_NO_EARLY_NORM') != '1'
            self._norm_buckets_seen = 0
            # the norm from the producers (r04) needs every gradient element to be written exactly once by a kernel that can sum it: one sample per step
            # ... and a slot layout that depends on the weight's shape only: true for the LDS-DMA weight-gradient kernel the step takes by default; the A/B fallback
            # (VLASER_SFT_WGRAD=tn) picks its kernel -- and with it the slots it reaches -- by the sequence length, so it keeps the buffer pass
            self._fused_norm = early and self.wgrad_lds and os.environ.get('VLASER_SFT_NO_FUSED_NORM') != '1'
            if self._fused_norm and getattr(self, 'norm_parts', None) is None:
                self._plan_fused_norm()
            try:
                loss = self.forward_backward(*work[0][:4], on_bucket_ready=self._norm_bucket if early else self._exchange_bucket)
            finally:
                self._fused_norm = False
            # a forward_backward that raised never gets here, and one that returned must have signalled every bucket
            self._norm_early = early and self._norm_buckets_seen == len(self.buckets)
        else:
            if self.gacc is None:
                self.gacc = torch.zeros(self.fp.n, dtype=F32, device=self.device)
            loss = torch.zeros((), device=self.device)
            for j, (pvb, idb, lbb, flb, w) in enumerate(work):
                first, last = j == 0, j == len(work) - 1
                lj = self.forward_backward(pvb, idb, lbb, flb, on_bucket_ready=lambda b, w=w, first=first, last=last: self._accumulate_bucket(b, w, first, last))
                loss = loss + lj * w
        gnorm = self.optimizer_step(lr)
        return SimpleNamespace(loss=loss, grad_norm=gnorm)

    def train_step_packed(self, pixel_values, input_ids, labels, loss_weight, cu_seqlens, image_flags=None, lr=None, total_steps=None,
                          loss_reduction_all_gather=False):
        """One optimizer step on a PACKED batch (`--use_packed_ds`: dataset_packed.py:517-624; varlen attention of
        qwen2_packed_training_patch.py:14-101; weighted loss of modeling_internvl_chat.py:207-230).  Rows of `input_ids` concatenate
        sub-sequences delimited by `cu_seqlens` [B, n+1]; block-diagonal causal attention makes them independent, so each runs as its
        own sample.  The reference's loss sum(w_t ce_t) / sum(w_t) has w constant over the supervised tokens of a sub-sequence
        (`len2weight`, dataset_packed.py:540) and 0 on ignored labels (:622): sub-sequence j therefore enters with weight
        w_j R_j / sum_k w_k R_k (R = supervised positions)."""
        if lr is None and total_steps is not None:
            lr = cosine_lr(self.step_count, total_steps, self.lr)
        B, S = input_ids.shape
        cu = cu_seqlens.detach().to('cpu', torch.int64).reshape(B, -1)
        ids_h = input_ids.detach().to('cpu', torch.int64)
        lab_h = labels.detach().to('cpu', torch.int64)
        w_h = torch.as_tensor(loss_weight, dtype=torch.float32).reshape(B, S)
        flags = None if image_flags is None else image_flags.detach().to('cpu').reshape(-1)
        nt = self.cfg.num_image_token
        work, t0 = [], 0
        for b in range(B):
            for lo, hi in zip(cu[b, :-1].tolist(), cu[b, 1:].tolist()):
                if hi <= lo:
                    continue
                if lo > 0 and lab_h[b, lo] != -100 and w_h[b, lo] != 0:
                    raise NotImplementedError('a supervised FIRST token of a packed sub-sequence would be predicted from the previous sub-sequence')
                sup = lab_h[b, lo + 1:hi] != -100
                ws = w_h[b, lo + 1:hi][sup]
                if ws.numel() and not bool((ws == ws[0]).all()):
                    raise NotImplementedError('loss_weight must be constant over the supervised tokens of a sub-sequence (dataset_packed.py:540)')
                need = int((ids_h[b, lo:hi] == self.img_context_token_id).sum()) // nt
                t1, got = t0, 0
                while t1 < pixel_values.shape[0] and got < need:
                    got += 1 if (flags is None or flags[t1] == 1) else 0
                    t1 += 1
                if need == 0 and flags is not None and t1 < pixel_values.shape[0] and flags[t1] == 0:
                    t1 += 1                                            # the dummy tile of a text-only sub-sequence
                if t1 == t0:
                    raise ValueError('every packed sub-sequence needs at least its (dummy) tile, as the reference dataset emits it')
                wj = float(ws[0]) * int(sup.sum()) if ws.numel() else 0.0
                work.append([pixel_values[t0:t1], ids_h[b:b + 1, lo:hi], lab_h[b:b + 1, lo:hi], None if flags is None else flags[t0:t1].reshape(-1, 1), wj])
                t0 = t1
        wsum = sum(x[4] for x in work)
        if loss_reduction_all_gather and self.dp_active:
            tw = torch.tensor([wsum], dtype=torch.float64, device=self.device)
            torch.distributed.all_reduce(tw, op=torch.distributed.ReduceOp.AVG, group=self.pg)
            wsum = float(tw.item())
        for x in work:
            x[4] = x[4] / wsum if wsum else 0.0
        return self._run_weighted(work, lr)

    # ------------------------------------------------------------------ resumable training state
    def save_checkpoint(self, path):
        """Everything a run needs to resume bit-identically (HF Trainer checkpoints, internvl_chat_finetune.py:847-860,1051-1068; VLA
        `step{N}.pt`, train.py:639-672): rank 0 writes the HF-layout weights (`save_pretrained`), EVERY rank writes its ZeRO-1 shard
        of the fp32 masters + AdamW moments and the step counter (DeepSpeed shards optimizer state per rank the same way)."""
        os.makedirs(path, exist_ok=True)
        if self.rank == 0:
            self.save_pretrained(path)
        self.wait_optimizer()
        torch.save({'step_count': self.step_count, 'world': self.world, 'rank': self.rank, 'shards': self.shards, 'lr': self.lr, 'betas': self.betas,
                    'eps': self.eps, 'weight_decay': self.wd, 'master': self.master.cpu(), 'exp_avg': self.m.cpu(), 'exp_avg_sq': self.v.cpu()},
                   os.path.join(path, f'optimizer_rank{self.rank:05d}_of_{self.world:05d}.pt'))

    def load_checkpoint(self, path):
        """Weights through `load_hf_checkpoint` (key names unchanged), then this rank's optimizer shard; the bf16 parameters are
        re-derived from the fp32 masters so the resumed run continues bit for bit."""
        from .config import load_hf_checkpoint
        self.wait_optimizer()
        _, sd = load_hf_checkpoint(path)
        self.load_state_dict(sd)
        st = torch.load(os.path.join(path, f'optimizer_rank{self.rank:05d}_of_{self.world:05d}.pt'), map_location='cpu', weights_only=True)
        if st['world'] != self.world or [tuple(x) for x in st['shards']] != [tuple(x) for x in self.shards]:
            raise ValueError(f"optimizer shard was written for world size {st['world']} / another bucket layout")
        self.step_count = st['step_count']
        self.master.copy_(st['master']); self.m.copy_(st['exp_avg']); self.v.copy_(st['exp_avg_sq'])
        for (lo, hi, _), o in zip(self.shards, self.shard_off):
            if hi > lo:
                self.fp.p[lo:hi].copy_(self.master[o:o + hi - lo].to(BF))
        return self

    def save_pretrained(self, path, max_shard_bytes=4 << 30):
        """HF-layout checkpoint (config.json + sharded safetensors + index) of the current weights: trainable tensors from the flat
        buffer (HF key names, un-packed), the frozen vision tower as loaded -- `InternVLChatModel.from_pretrained(path)` reads it."""
        from .config import save_hf_checkpoint
        sd = dict(self.frozen_sd)
        sd.update(self.state_dict())
        save_hf_checkpoint(path, self.cfg, sd, max_shard_bytes)

    # ------------------------------------------------------------------ export (HF key names, un-packed layouts)
    def state_dict(self):
        self.wait_optimizer()                                              # pending parameter updates / all-gathers
        llm = self.llm
        v = self.fp.view
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        inv = torch.empty(hd, dtype=torch.long)
        inv[ops.head_perm(hd)] = torch.arange(hd)
        out = {'language_model.lm_head.weight': v['head'].clone(), 'language_model.model.norm.weight': v['norm'].clone(),
               'language_model.model.embed_tokens.weight': v['embed'].clone()}
        for i in range(llm.num_hidden_layers):
            p = f'language_model.model.layers.{i}.'
            w, b = v[f'l{i}.wqkv'], v[f'l{i}.bqkv']
            nh = nq + 2 * nkv
            idx = (torch.arange(nh)[:, None] * hd + inv[None, :]).reshape(-1).to(w.device)
            wn, bn = w[idx], b[idx]              # natural row order: packed[head*128 + inv[d]] = natural[head*128 + d]
            out[p + 'self_attn.q_proj.weight'], out[p + 'self_attn.q_proj.bias'] = wn[:nq * hd].clone(), bn[:nq * hd].clone()
            out[p + 'self_attn.k_proj.weight'], out[p + 'self_attn.k_proj.bias'] = wn[nq * hd:(nq + nkv) * hd].clone(), bn[nq * hd:(nq + nkv) * hd].clone()
            out[p + 'self_attn.v_proj.weight'], out[p + 'self_attn.v_proj.bias'] = wn[(nq + nkv) * hd:].clone(), bn[(nq + nkv) * hd:].clone()
            out[p + 'self_attn.o_proj.weight'] = v[f'l{i}.wo'].clone()
            gu = v[f'l{i}.wgu'].view(-1, 2, 16, llm.hidden_size)
            out[p + 'mlp.gate_proj.weight'] = gu[:, 0].reshape(-1, llm.hidden_size).clone()
            out[p + 'mlp.up_proj.weight'] = gu[:, 1].reshape(-1, llm.hidden_size).clone()
            out[p + 'mlp.down_proj.weight'] = v[f'l{i}.wdown'].clone()
            out[p + 'input_layernorm.weight'] = v[f'l{i}.ln_in'].clone()
            out[p + 'post_attention_layernorm.weight'] = v[f'l{i}.ln_post'].clone()
        for nm, k in [('m0w', 'mlp1.0.weight'), ('m0b', 'mlp1.0.bias'), ('m1w', 'mlp1.1.weight'), ('m1b', 'mlp1.1.bias'),
                      ('m3w', 'mlp1.3.weight'), ('m3b', 'mlp1.3.bias')]:
            out[k] = v['mlp1.' + nm].clone()
        return out

    def named_grads(self):
        """Gradients under HF key names (un-packed), for parity tests."""
        saved = self.fp.view
        self.fp.view = self.fp.gview
        try:
            return self.state_dict()
        finally:
            self.fp.view = saved
