"""The SECOND parameter group of the Vlaser-VLA flow-matching training step: `train_vlm: True`.

Reference: `trainable_vlm_parameters` (Vlaser_VLA/Simpler/src/model/vla/pizero_internvl.py:405-411) = `vision_tower` (InternViT-300M, every parameter)
+ `multi_modal_projector` (mlp1) + `joint_model.mixtures["vlm"]` (the Qwen2.5 decoder layers and final norm; `embed_tokens` is NOT in the group), with
its own AdamW + cosine-restart schedule (`vlm_lr`, src/agent/train.py:270-295) stepped beside the action optimiser (:509-520) after ONE
`clip_grad_norm_` over both groups (:504-507).

Where the gradient flows (`PiZero.forward`, :1064-1197; block mask :517-587): the loss reads the action rows only; the image/text rows influence it
through the keys / values they hand to the proprio / action rows in every layer.  So per layer, top down:
  expert rows' attention backward -> dK, dV of the prefix keys (`vlaser_attn_rows_bwd_ex` + grouped TN GEMM)
  -> the VLM rows' q/k/v projection, their own (bidirectional, valid-prefix) attention, o_proj and MLP of the layers below
  -> embeddings of the image rows -> mlp1 -> pixel_shuffle -> the 24 InternViT blocks -> patch / class / position embeddings.
The last VLM layer contributes through K / V only (its post-attention half feeds nothing: `final_layer_post_attn_skip_names`), its q projection,
o_proj, MLP, post-attention norm and the final norm receive no gradient -- exactly the reference's `grad is None` set (golden G10b).

Forward = the inference kernels with every intermediate kept (the image/text rows run ONCE per sample); backward = the SFT step's kernels
(NN dgrad on the weights as stored, TN wgrad, RMSNorm / SwiGLU / RoPE backward) with a bidirectional mask, plus LayerNorm / GELU / layer-scale /
full-attention backward for the vision tower.  All arithmetic is HIP through the C ABI; torch holds buffers and does index plumbing.
"""
import os

import torch

from . import _lib as L
from . import dp, ops
from .engine import BF
from .sft import FlatParams

F32 = torch.float32


class VLMGroup:
    def __init__(self, trainer, sd, bucket_layers=8):
        """`trainer`: the VLATrainer (frozen-prefix engines already built from `sd`); the group re-homes every trained VLM tensor in ONE flat
        bf16 buffer (kernel layouts) and points the engines at views of it, so AdamW updates are what the next forward reads."""
        tr = self.tr = trainer
        cfg, dev = tr.cfg, tr.device
        base, llm, vis = cfg.base, cfg.base.llm, cfg.base.vision
        self.llm, self.vis = llm, vis
        H, I = llm.hidden_size, llm.intermediate_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        NQ = (nq + 2 * nkv) * hd
        C, Cm, Hn = vis.hidden_size, vis.intermediate_size, vis.num_attention_heads
        C4 = 4 * C
        Lyr, Lv = llm.num_hidden_layers, vis.num_hidden_layers
        world = tr.world
        fp = FlatParams(dev)
        bounds = []
        fp.add('norm', (H,))                      # final norm of the VLM mixture: in the group, never reached by the loss (zero gradient)
        for j, i in enumerate(reversed(range(Lyr))):
            if j % bucket_layers == 0 and j > 0:
                fp.align(128 * world); bounds.append(fp.n)
            for nm, shp in [('wqkv', (NQ, H)), ('bqkv', (NQ,)), ('wo', (H, nq * hd)), ('wgu', (2 * I, H)), ('wdown', (H, I)), ('ln_in', (H,)), ('ln_post', (H,))]:
                fp.add(f'l{i}.{nm}', shp)
        fp.align(128 * world); bounds.append(fp.n)
        for nm, shp in [('m0w', (C4,)), ('m0b', (C4,)), ('m1w', (H, C4)), ('m1b', (H,)), ('m3w', (H, H)), ('m3b', (H,))]:
            fp.add('mlp1.' + nm, shp)
        for j, i in enumerate(reversed(range(Lv))):
            if j % bucket_layers == 0 and j > 0:
                fp.align(128 * world); bounds.append(fp.n)
            for nm, shp in [('wqkv', (3 * C, C)), ('bqkv', (3 * C,)), ('wproj', (C, C)), ('bproj', (C,)), ('wfc1', (Cm, C)), ('bfc1', (Cm,)), ('wfc2', (C, Cm)),
                            ('bfc2', (C,)), ('n1w', (C,)), ('n1b', (C,)), ('n2w', (C,)), ('n2b', (C,)), ('ls1', (C,)), ('ls2', (C,))]:
                fp.add(f'v{i}.{nm}', shp)
        fp.align(128 * world); bounds.append(fp.n)
        KP = tr.vit.KPAD
        for nm, shp in [('pe.w', (C, KP)), ('pe.b', (C,)), ('cls', (C,)), ('pos', (vis.num_positions, C))]:
            fp.add('emb.' + nm, shp)
        fp.finalize(pad_to=128 * world * 8)
        self.fp = fp
        v = fp.view
        # ---- fill from the engines' tensors (already in kernel layout) and re-point the engines at the views
        v['norm'].copy_(tr.vlm.norm); tr.vlm.norm = v['norm']
        for i, lw in enumerate(tr.vlm.layers):
            for nm in ('wqkv', 'bqkv', 'wo', 'wgu', 'wdown', 'ln_in', 'ln_post'):
                v[f'l{i}.{nm}'].copy_(getattr(lw, nm)); setattr(lw, nm, v[f'l{i}.{nm}'])
        vit = tr.vit
        for nm in ('m0w', 'm0b', 'm1w', 'm1b', 'm3w', 'm3b'):
            v['mlp1.' + nm].copy_(getattr(vit, nm)); setattr(vit, nm, v['mlp1.' + nm])
        for i, lw in enumerate(vit.layers):
            for nm in list(lw.keys()):
                v[f'v{i}.{nm}'].copy_(lw[nm]); lw[nm] = v[f'v{i}.{nm}']
        v['emb.pe.w'].copy_(vit.w_pe); vit.w_pe = v['emb.pe.w']
        v['emb.pe.b'].copy_(vit.b_pe); vit.b_pe = v['emb.pe.b']
        v['emb.cls'].copy_(vit.cls); vit.cls = v['emb.cls']
        v['emb.pos'].copy_(vit.pos); vit.pos = v['emb.pos']
        self.buckets, lo = [], 0
        for hi in bounds + [fp.n]:
            if hi > lo:
                self.buckets.append((lo, hi)); lo = hi
        assert all((hi - lo) % (128 * world) == 0 for lo, hi in self.buckets)
        self.shards = dp.plan_shards(self.buckets, world, tr.rank)
        n_shard = sum(hi - lo for lo, hi, _ in self.shards)
        self.master = torch.zeros(n_shard, dtype=F32, device=dev)
        self.m = torch.zeros(n_shard, dtype=F32, device=dev)
        self.v = torch.zeros(n_shard, dtype=F32, device=dev)
        self.shard_off, o = [], 0
        for lo, hi, _ in self.shards:
            self.master[o:o + hi - lo].copy_(fp.p[lo:hi].float())
            self.shard_off.append(o); o += hi - lo
        self._alloc()

    def _alloc(self):
        tr, llm, vis = self.tr, self.llm, self.vis
        dev, T = tr.device, tr.T
        H, I = llm.hidden_size, llm.intermediate_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        NQ = (nq + 2 * nkv) * hd
        Lyr, Lv = llm.num_hidden_layers, vis.num_hidden_layers
        z = lambda *s, dt=BF: torch.zeros(*s, dtype=dt, device=dev)
        # ---- saved activations of the T image/text rows, per LLM layer
        self.h_in = z(Lyr + 1, T, H)
        self.x1, self.x2, self.h2 = z(Lyr, T, H), z(Lyr, T, H), z(Lyr, T, H)
        self.q, self.ao = z(Lyr, T, nq * hd), z(Lyr, T, nq * hd)
        self.gu, self.act = z(Lyr, T, 2 * I), z(Lyr, T, I)
        self.part = torch.zeros(ops.split_slab_elems(max(T, vis.num_positions), max(H, vis.hidden_size)), dtype=F32, device=dev)
        # ---- backward buffers (LLM rows)
        self.Tp = (T + 63) // 64 * 64
        self.dh, self.dh2, self.dx = z(T, H), z(T, H), z(T, H)
        self.dgu = z(T, 2 * I)
        self.dao, self.dq, self.dk, self.dv = z(T, nq * hd), z(T, nq * hd), z(T, nq * hd), z(T, nq * hd)
        self.zkv = z(T, nkv * hd)
        self.dqkv = z(T, NQ)
        self.dkx, self.dvx = z(T, nkv * hd), z(T, nkv * hd)                    # prefix-key gradients from the expert rows of ONE layer
        self.p_rows, self.ds_rows = z(nq, 16, tr.s_max), z(nq, 16, tr.s_max)    # P / dS of the expert rows over every key
        # fused attention backward of the VLM rows (r06): the forward's base-2 log-sum-exp per layer + the kernel's delta scratch
        self.lse = torch.zeros(llm.num_hidden_layers, nq * T, dtype=F32, device=dev)
        self.delta_ws = torch.zeros(nq * T, dtype=F32, device=dev)
        self.fused_attn_bwd = os.environ.get('VLASER_VLA_VLM_ATTN_BWD', 'fused') != 'materialised'
        # ---- vision tower: saved activations per block + projector intermediates
        S, C, Cm, Hn, hdv = vis.num_positions, vis.hidden_size, vis.intermediate_size, vis.num_attention_heads, vis.head_dim
        sp = self.sp = tr.vit.s_pad
        self.vh = z(Lv + 1, S, C)
        self.vx1, self.vhm, self.vx2, self.vao = z(Lv, S, C), z(Lv, S, C), z(Lv, S, C), z(Lv, S, C)
        self.vq, self.vk, self.vvt = z(Lv, Hn, sp, hdv), z(Lv, Hn, sp, hdv), z(Lv, Hn, hdv, sp)
        self.vz, self.vf = z(Lv, S, Cm), z(Lv, S, Cm)
        nt = tr.cfg.base.num_image_token
        C4 = 4 * C
        self.ps_raw, self.ps_ln = z(nt, C4), z(nt, C4)
        self.z1, self.g1, self.feat = z(nt, H), z(nt, H), z(nt, H)
        # ---- backward buffers (vision / projector); the score matrices are shared by both towers' attention backward
        n_sc = max(nq * T * self.Tp, Hn * S * sp)
        self.sc = torch.zeros(n_sc, dtype=F32, device=dev)
        self.dP = torch.zeros(n_sc, dtype=F32, device=dev)
        self.P, self.dS = z(n_sc), z(n_sc)
        self.dvh, self.dvhm, self.dvx = z(S, C), z(S, C), z(S, C)
        self.dy, self.yrec = z(S, C), z(S, C)
        self.dvf, self.dvz = z(S, Cm), z(S, Cm)
        self.dvao, self.dvqkv = z(S, C), z(S, 3 * C)
        self.dfeat, self.dg1, self.dz1, self.dln, self.dps = z(nt, H), z(nt, H), z(nt, H), z(nt, C4), z(nt, C4)
        wmax = max(2 * I, NQ, C4, 3 * C, Cm, H)
        self.col = torch.zeros(wmax, dtype=F32, device=dev)
        self.rowstat = torch.zeros(2 * max(T, S, nt) + 16 * wmax, dtype=F32, device=dev)
        self.normw_ws = torch.zeros((T + 3) // 4 * H, dtype=F32, device=dev)

    # ------------------------------------------------------------------ helpers
    def _wgrad(self, dY, X, out, bias_out=None):
        ops.gemm_tn(dY, X, out)
        if bias_out is not None:
            ops.colsum_bf16(dY, bias_out, dY.shape[0], dY.shape[1])

    def _dgrad(self, dY, W, out):
        """out[S,K] = dY[S,N] @ W[N,K], W as stored; long contractions over few output tiles run split-K (as SFTModel._dgrad)."""
        S = dY.shape[0]
        Nin, Kout = W.shape
        sp = 1 if Nin <= 2048 else ops.gemm_splits(S, Kout, Nin, self.part.numel(), nn=True)
        if sp > 1:
            part = self.part[:sp * S * Kout]
            ops.gemm_nn(L.EPI_PARTIAL, dY, W, out_f32=part, k_splits=sp)
            ops.reduce_norm(None, part, sp, S, Kout, out)
        else:
            ops.gemm_nn(L.EPI_NONE, dY, W, out=out)

    def _colsum(self, a, b, out, S, Cc, mode, eps=1e-6):
        ops.colsum_mul(a, b, self.col, S, Cc, mode, eps, self.rowstat)
        out.copy_(self.col[:Cc])

    # ------------------------------------------------------------------ forward: image/text rows, everything kept
    def forward(self, pvb, ids):
        """ViT -> mlp1 -> embeddings -> the VLM rows through every decoder layer (last layer: q/k/v + attention only); K / V^T of every layer
        land in the trainer's cache, where the expert rows' joint attention reads them."""
        tr, llm, vis, v = self.tr, self.llm, self.vis, self.fp.view
        cfg, T = tr.cfg, tr.T
        vit = tr.vit
        S, C, Hn, hdv, sp = vis.num_positions, vis.hidden_size, vis.num_attention_heads, vis.head_dim, self.sp
        Lv, Lyr = vis.num_hidden_layers, llm.num_hidden_layers
        # ---- vision tower (modeling_intern_vit.py:162-174, 283-295)
        ops.im2col(pvb, vit.col, 1, vis.image_size, vit.KPAD)
        ops.linear(vit.col[:vis.num_patches], v['emb.pe.w'], v['emb.pe.b'], out=vit.patch[:vis.num_patches])
        ops.vit_assemble(vit.patch, v['emb.cls'], v['emb.pos'], self.vh[0], 1, vis.num_patches, C)
        sp_proj = ops.gemm_splits(S, C, C, self.part.numel())
        sp_fc2 = ops.gemm_splits(S, C, vis.intermediate_size, self.part.numel())
        ops.layernorm(self.vh[0], v['v0.n1w'], v['v0.n1b'], vis.layer_norm_eps, out=self.vx1[0])
        for l in range(Lv):
            g = lambda nm: v[f'v{l}.{nm}']
            ops.gemm(L.EPI_VIT_QKV, self.vx1[l], g('wqkv'), bias=g('bqkv'), vq=self.vq[l], vk=self.vk[l], vvt=self.vvt[l], vit_heads=Hn, vit_seq=S,
                     vit_seq_pad=sp, q_scale=hdv ** -0.5)
            ops.attn_prefill(self.vq[l], self.vk[l], self.vvt[l], self.vao[l], 1, S, S, Hn, Hn, hdv, (Hn * sp * hdv, sp * hdv, hdv), (Hn * sp * hdv, sp * hdv),
                             (Hn * hdv * sp, hdv * sp), (S * C, C), sp, 1.0, L.ATTN_FULL)
            ops.gemm(L.EPI_PARTIAL, self.vao[l], g('wproj'), out_f32=self.part, k_splits=sp_proj)
            ops.reduce_norm(self.vh[l], self.part, sp_proj, S, C, self.vhm[l], self.vx2[l], bias=g('bproj'), ls=g('ls1'), norm=2, norm_w=g('n2w'), norm_b=g('n2b'),
                            eps=vis.layer_norm_eps)
            ops.gemm(L.EPI_BIAS_GELU, self.vx2[l], g('wfc1'), out=self.vf[l], bias=g('bfc1'), aux_out=self.vz[l], ld_aux=self.vz[l].stride(0))
            ops.gemm(L.EPI_PARTIAL, self.vf[l], g('wfc2'), out_f32=self.part, k_splits=sp_fc2)
            nxt = l + 1 < Lv
            ops.reduce_norm(self.vhm[l], self.part, sp_fc2, S, C, self.vh[l + 1], self.vx1[l + 1] if nxt else None, bias=g('bfc2'), ls=g('ls2'),
                            norm=2 if nxt else 0, norm_w=v[f'v{l + 1}.n1w'] if nxt else None, norm_b=v[f'v{l + 1}.n1b'] if nxt else None, eps=vis.layer_norm_eps)
        # ---- projector (modeling_internvl_chat.py:89-94, 257-291), intermediates kept
        G_ = vis.image_size // vis.patch_size
        v1 = 1 if cfg.base.ps_version == 'v1' else 0
        ops.pixel_shuffle(self.vh[Lv], self.ps_raw, 1, G_, C, v1)
        ops.pixel_shuffle_ln(self.vh[Lv], v['mlp1.m0w'], v['mlp1.m0b'], self.ps_ln, 1, G_, C, 1e-5, v1)
        ops.gemm(L.EPI_BIAS_GELU, self.ps_ln, v['mlp1.m1w'], out=self.g1, bias=v['mlp1.m1b'], aux_out=self.z1, ld_aux=self.z1.stride(0))
        ops.gemm(L.EPI_BIAS, self.g1, v['mlp1.m3w'], out=self.feat, bias=v['mlp1.m3b'])
        # ---- embeddings (pad rows zero: pizero_internvl.py:757-791) and the decoder layers
        h0 = self.h_in[0]
        ops.embed_merge(ids, tr.vlm.embed, self.feat, h0, cfg.base.img_context_token_id, cfg.base.pad_token_id, True, tr.rank_ws)
        H, I = llm.hidden_size, llm.intermediate_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        ks, vs = tr.cache.strides()
        for i in range(Lyr):
            g = lambda nm: v[f'l{i}.{nm}']
            ops.rmsnorm(self.h_in[i], g('ln_in'), llm.rms_norm_eps, out=self.x1[i])
            ops.gemm(L.EPI_QKV_ROPE, self.x1[i], g('wqkv'), bias=g('bqkv'), q_out=self.q[i], k_cache=tr.cache.k[i], vt_cache=tr.cache.vt[i], rope_cos=tr.rope[0],
                     rope_sin=tr.rope[1], pos_ids=tr.pos_vlm, n_q_heads=nq, n_kv_heads=nkv, s_max=tr.s_max, tok_per_batch=T, slot_base=0)
            ops.attn_prefill(self.q[i], tr.cache.k[i], tr.cache.vt[i], self.ao[i], 1, T, T, nq, nkv, hd, (T * nq * hd, hd, nq * hd), ks, vs, (T * nq * hd, nq * hd),
                             tr.s_max, hd ** -0.5, L.ATTN_PREFIX, valid_len=tr.valid_len, blk_start=T, lse_out=self.lse[i])
            if i == Lyr - 1:
                break                                      # the last layer's post-attention half feeds nothing (final_layer_post_attn_skip_names)
            sp_o = ops.gemm_splits(T, H, nq * hd, self.part.numel())
            ops.gemm(L.EPI_PARTIAL, self.ao[i], g('wo'), out_f32=self.part, k_splits=sp_o)
            ops.reduce_norm(self.h_in[i], self.part, sp_o, T, H, self.h2[i], self.x2[i], norm=1, norm_w=g('ln_post'), eps=llm.rms_norm_eps)
            ops.gemm(L.EPI_SWIGLU, self.x2[i], g('wgu'), out=self.act[i], aux_out=self.gu[i], ld_aux=self.gu[i].stride(0))
            sp_d = ops.gemm_splits(T, H, I, self.part.numel())
            ops.gemm(L.EPI_PARTIAL, self.act[i], g('wdown'), out_f32=self.part, k_splits=sp_d)
            ops.reduce_norm(self.h2[i], self.part, sp_d, T, H, self.h_in[i + 1])

    # ------------------------------------------------------------------ backward
    def begin_backward(self):
        self.fp.g.zero_()                                  # tensors the loss never reaches keep a zero gradient (the reference leaves them None)
        self.dh.zero_()

    def prefix_kv_grads(self, q_rows, R, n_q_e, n_kv):
        """dK / dV of the prefix keys of ONE layer from the expert rows' P / dS (written by vlaser_attn_rows_bwd_ex into self.p_rows / self.ds_rows):
        dK[kvh] = sum_{g, r} dS[kvh G + g][r]^T q[r, kvh G + g], dV[kvh] = sum P^T dO -- contraction over the R rows, summed over the group."""
        tr = self.tr
        T, hd, sm = tr.T, self.llm.head_dim, tr.s_max
        G = n_q_e // n_kv
        q_e, dO_e = q_rows
        ops.gemm_tn_grouped(self.ds_rows, q_e, self.dkx, T, hd, R, sm, n_q_e * hd, n_kv * hd, G, 16 * sm, hd, n_kv, G * 16 * sm, G * hd, hd)
        ops.gemm_tn_grouped(self.p_rows, dO_e, self.dvx, T, hd, R, sm, n_q_e * hd, n_kv * hd, G, 16 * sm, hd, n_kv, G * 16 * sm, G * hd, hd)

    def backward_layer(self, i, n_valid):
        """Decoder layer i of the VLM rows: self.dh holds d loss / d h_in[i+1] (zero on entry for the last layer), self.dkx / self.dvx the prefix-key
        gradients of this layer's joint attention; leaves d loss / d h_in[i] in self.dh."""
        tr, llm = self.tr, self.llm
        v, gv = self.fp.view, self.fp.gview
        T, Tp = tr.T, self.Tp
        H, I = llm.hidden_size, llm.intermediate_size
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        G, sm, scale = nq // nkv, tr.s_max, hd ** -0.5
        Lyr = llm.num_hidden_layers
        g = lambda nm: v[f'l{i}.{nm}']
        gg = lambda nm: gv[f'l{i}.{nm}']
        dh, dh2, dx, dao, dqkv = self.dh, self.dh2, self.dx, self.dao, self.dqkv
        if i < Lyr - 1:
            ops.gemm_nn(L.EPI_SWIGLU_BWD, dh, g('wdown'), out=self.dgu, res=self.gu[i])
            self._wgrad(dh, self.act[i], gg('wdown'))
            self._dgrad(self.dgu, g('wgu'), dx)
            self._wgrad(self.dgu, self.x2[i], gg('wgu'))
            ops.rmsnorm_bwd(dx, self.h2[i], g('ln_post'), dh, dh2, T, H, llm.rms_norm_eps, dw_out=gg('ln_post'), dw_ws=self.normw_ws)
            self._dgrad(dh2, g('wo'), dao)
            self._wgrad(dh2, self.ao[i], gg('wo'))
            # attention backward; key k visible iff k < n_valid (bidirectional inside the valid prefix)
            Kc, VTc = tr.cache.k[i, 0], tr.cache.vt[i, 0]
            if self.fused_attn_bwd:
                # r06: the SFT step's fused kernel (csrc/attn_bwd.hip; causal = 0, kv_valid = the prefix length) instead of six launches through materialised
                # [heads, T, T] fp32 score matrices: ~200 -> ~40 us per layer (profiles/r06ab_vla_train_kernel_stats.md)
                ops.attn_bwd(self.q[i], Kc, VTc, self.ao[i], dao, self.lse[i], self.delta_ws, self.dq, self.dk, self.dv, T, nq, nkv, sm, scale, causal=False,
                             kv_valid=n_valid, head_dim=hd)
                ops.rope_bwd_pack_ex(self.dq, self.dk, self.dv, tr.rope[0], tr.rope[1], tr.pos_vlm, dqkv, T, nq, nkv, kv_per_q_head=True, dk_extra=self.dkx,
                                     dv_extra=self.dvx)
                dres = dh2
                self._dgrad(dqkv, g('wqkv'), dx)
                self._wgrad(dqkv, self.x1[i], gg('wqkv'), bias_out=gg('bqkv'))
                ops.rmsnorm_bwd(dx, self.h_in[i], g('ln_in'), dres, dh, T, H, llm.rms_norm_eps, dw_out=gg('ln_in'), dw_ws=self.normw_ws)
                return
            n = nq * T * Tp
            sc, dP = self.sc[:n].view(nq, T, Tp), self.dP[:n].view(nq, T, Tp)
            P, dS = self.P[:n].view(nq, T, Tp), self.dS[:n].view(nq, T, Tp)
            ops.gemm_raw(L.EPI_F32, self.q[i], Kc, sc, T, T, hd, nq * hd, hd, Tp, batch=nq, a_bs=hd, w_bs=sm * hd, o_bs=T * Tp, w_group=G)
            ops.gemm_raw_nn(L.EPI_F32, dao, VTc, dP, T, Tp, hd, nq * hd, sm, Tp, batch=nq, a_bs=hd, w_bs=hd * sm, o_bs=T * Tp, w_group=G)
            ops.attn_bwd_pds_masked(sc, dP, dao, self.ao[i], P, dS, nq, T, Tp, hd, scale, False, n_valid)
            ops.gemm_raw_nn(L.EPI_NONE, dS, Kc, self.dq, T, hd, Tp, Tp, hd, nq * hd, batch=nq, a_bs=T * Tp, w_bs=sm * hd, o_bs=hd, w_group=G)
            ops.gemm_tn_grouped(dS, self.q[i], self.dk, T, hd, T, Tp, nq * hd, nq * hd, 1, 0, 0, nq, T * Tp, hd, hd)
            ops.gemm_tn_grouped(P, dao, self.dv, T, hd, T, Tp, nq * hd, nq * hd, 1, 0, 0, nq, T * Tp, hd, hd)
            ops.rope_bwd_pack_ex(self.dq, self.dk, self.dv, tr.rope[0], tr.rope[1], tr.pos_vlm, dqkv, T, nq, nkv, kv_per_q_head=True, dk_extra=self.dkx,
                                 dv_extra=self.dvx)
            dres = dh2
        else:
            self.dq.zero_()
            ops.rope_bwd_pack_ex(self.dq, self.zkv, self.zkv, tr.rope[0], tr.rope[1], tr.pos_vlm, dqkv, T, nq, nkv, kv_per_q_head=False, dk_extra=self.dkx,
                                 dv_extra=self.dvx)
            dres = None
        self._dgrad(dqkv, g('wqkv'), dx)
        self._wgrad(dqkv, self.x1[i], gg('wqkv'), bias_out=gg('bqkv'))
        ops.rmsnorm_bwd(dx, self.h_in[i], g('ln_in'), dres, dh, T, H, llm.rms_norm_eps, dw_out=gg('ln_in'), dw_ws=self.normw_ws)

    def backward_tail(self, ids_h):
        """self.dh = d loss / d (input embeddings of the T rows): image rows -> mlp1 -> pixel_shuffle -> the vision tower -> its embeddings."""
        tr, llm, vis = self.tr, self.llm, self.vis
        cfg = tr.cfg
        v, gv = self.fp.view, self.fp.gview
        H = llm.hidden_size
        S, C, Cm, Hn, hdv, sp = vis.num_positions, vis.hidden_size, vis.intermediate_size, vis.num_attention_heads, vis.head_dim, self.sp
        Lv = vis.num_hidden_layers
        nt, C4 = cfg.base.num_image_token, 4 * C
        img_rows = (ids_h.reshape(-1) == cfg.base.img_context_token_id).nonzero().flatten().pin_memory().to(tr.device, non_blocking=True)
        self.dfeat.copy_(self.dh.index_select(0, img_rows))                      # index plumbing (the visual-token scatter of a7 run backwards)
        # ---- projector: feat = m3(gelu(m1(LN(ps_raw))))
        ops.gemm_nn(L.EPI_NONE, self.dfeat, v['mlp1.m3w'], out=self.dg1)
        self._wgrad(self.dfeat, self.g1, gv['mlp1.m3w'], bias_out=gv['mlp1.m3b'])
        ops.gelu_bwd(self.z1, self.dg1, self.dz1)
        self._wgrad(self.dz1, self.ps_ln, gv['mlp1.m1w'], bias_out=gv['mlp1.m1b'])
        ops.gemm_nn(L.EPI_NONE, self.dz1, v['mlp1.m1w'], out=self.dln)
        self._colsum(self.dln, self.ps_raw, gv['mlp1.m0w'], nt, C4, 3, 1e-5)
        self._colsum(self.dln, None, gv['mlp1.m0b'], nt, C4, 0)
        ops.layernorm_bwd(self.dln, self.ps_raw, v['mlp1.m0w'], None, self.dps, nt, C4, 1e-5)
        G_ = vis.image_size // vis.patch_size
        ops.pixel_unshuffle(self.dps, self.dvh, 1, G_, C, 1 if cfg.base.ps_version == 'v1' else 0)
        # ---- vision tower blocks, top down (modeling_intern_vit.py:283-295): h' = h + ls1 (proj(attn(LN1 h)) + b); h'' = h' + ls2 (fc2(gelu(fc1(LN2 h'))) + b)
        n = Hn * S * sp
        sc, dP = self.sc[:n].view(Hn, S, sp), self.dP[:n].view(Hn, S, sp)
        P, dS = self.P[:n].view(Hn, S, sp), self.dS[:n].view(Hn, S, sp)
        dvh, dvhm, dvx, dy, yrec = self.dvh, self.dvhm, self.dvx, self.dy, self.yrec
        eps = vis.layer_norm_eps
        for l in reversed(range(Lv)):
            g = lambda nm: v[f'v{l}.{nm}']
            gg = lambda nm: gv[f'v{l}.{nm}']
            # MLP branch
            ops.gemm(L.EPI_BIAS, self.vf[l], g('wfc2'), out=yrec, bias=g('bfc2'))                  # y2 again (the forward fuses it into the seam kernel)
            self._colsum(dvh, yrec, gg('ls2'), S, C, 1)
            ops.scale_cols(dvh, g('ls2'), dy, S, C)
            self._wgrad(dy, self.vf[l], gg('wfc2'), bias_out=gg('bfc2'))
            self._dgrad(dy, g('wfc2'), self.dvf)
            ops.gelu_bwd(self.vz[l], self.dvf, self.dvz)
            self._wgrad(self.dvz, self.vx2[l], gg('wfc1'), bias_out=gg('bfc1'))
            self._dgrad(self.dvz, g('wfc1'), dvx)
            self._colsum(dvx, self.vhm[l], gg('n2w'), S, C, 3, eps)
            self._colsum(dvx, None, gg('n2b'), S, C, 0)
            ops.layernorm_bwd(dvx, self.vhm[l], g('n2w'), dvh, dvhm, S, C, eps)
            # attention branch
            ops.gemm(L.EPI_BIAS, self.vao[l], g('wproj'), out=yrec, bias=g('bproj'))
            self._colsum(dvhm, yrec, gg('ls1'), S, C, 1)
            ops.scale_cols(dvhm, g('ls1'), dy, S, C)
            self._wgrad(dy, self.vao[l], gg('wproj'), bias_out=gg('bproj'))
            self._dgrad(dy, g('wproj'), self.dvao)
            q, k, vt = self.vq[l], self.vk[l], self.vvt[l]                                           # q carries the 1/sqrt(64) (VL_EPI_VIT_QKV)
            ops.gemm_raw(L.EPI_F32, q, k, sc, S, S, hdv, hdv, hdv, sp, batch=Hn, a_bs=sp * hdv, w_bs=sp * hdv, o_bs=S * sp)
            ops.gemm_raw_nn(L.EPI_F32, self.dvao, vt, dP, S, sp, hdv, C, sp, sp, batch=Hn, a_bs=hdv, w_bs=hdv * sp, o_bs=S * sp)
            ops.attn_bwd_pds_masked(sc, dP, self.dvao, self.vao[l], P, dS, Hn, S, sp, hdv, 1.0, False, S)
            dqkv = self.dvqkv
            ops.gemm_raw_nn(L.EPI_NONE, dS, k, dqkv, S, hdv, sp, sp, hdv, 3 * C, batch=Hn, a_bs=S * sp, w_bs=sp * hdv, o_bs=hdv)                    # dq (scaled q)
            ops.scale_cols(dqkv, None, dqkv, S, C, alpha=hdv ** -0.5)                                                                              # -> d(raw q)
            ops.gemm_tn_grouped(dS, q, dqkv[:, C:], S, hdv, S, sp, hdv, 3 * C, 1, 0, 0, Hn, S * sp, sp * hdv, hdv)                                  # dk = dS^T q
            ops.gemm_tn_grouped(P, self.dvao, dqkv[:, 2 * C:], S, hdv, S, sp, C, 3 * C, 1, 0, 0, Hn, S * sp, hdv, hdv)                              # dv = P^T dO
            self._wgrad(dqkv, self.vx1[l], gg('wqkv'), bias_out=gg('bqkv'))
            self._dgrad(dqkv, g('wqkv'), dvx)
            self._colsum(dvx, self.vh[l], gg('n1w'), S, C, 3, eps)
            self._colsum(dvx, None, gg('n1b'), S, C, 0)
            ops.layernorm_bwd(dvx, self.vh[l], g('n1w'), dvhm, dvh, S, C, eps)
        # ---- embeddings: h0 = [cls ; patch] + pos  (modeling_intern_vit.py:162-174)
        gv['emb.pos'].copy_(dvh)
        gv['emb.cls'].copy_(dvh[0])
        dpatch = dvh[1:]
        ops.gemm_tn(dpatch, tr.vit.col[:vis.num_patches], gv['emb.pe.w'])
        ops.colsum_bf16(dpatch, gv['emb.pe.b'], vis.num_patches, C)

    # ------------------------------------------------------------------ export (canonical key names, un-packed layouts)
    def state_dict(self, grads=False):
        llm, vis = self.llm, self.vis
        v = self.fp.gview if grads else self.fp.view
        nq, nkv, hd = llm.num_attention_heads, llm.num_key_value_heads, llm.head_dim
        inv = torch.empty(hd, dtype=torch.long); inv[ops.head_perm(hd)] = torch.arange(hd)
        out = {'language_model.model.norm.weight': v['norm'].clone()}
        for i in range(llm.num_hidden_layers):
            p = f'language_model.model.layers.{i}.'
            w, b = v[f'l{i}.wqkv'], v[f'l{i}.bqkv']
            idx = (torch.arange(nq + 2 * nkv)[:, None] * hd + inv[None, :]).reshape(-1).to(w.device)
            wn, bn = w[idx], b[idx]
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
        for nm, k in [('m0w', 'mlp1.0.weight'), ('m0b', 'mlp1.0.bias'), ('m1w', 'mlp1.1.weight'), ('m1b', 'mlp1.1.bias'), ('m3w', 'mlp1.3.weight'), ('m3b', 'mlp1.3.bias')]:
            out[k] = v['mlp1.' + nm].clone()
        names = [('wqkv', 'attn.qkv.weight'), ('bqkv', 'attn.qkv.bias'), ('wproj', 'attn.proj.weight'), ('bproj', 'attn.proj.bias'), ('wfc1', 'mlp.fc1.weight'),
                 ('bfc1', 'mlp.fc1.bias'), ('wfc2', 'mlp.fc2.weight'), ('bfc2', 'mlp.fc2.bias'), ('n1w', 'norm1.weight'), ('n1b', 'norm1.bias'), ('n2w', 'norm2.weight'),
                 ('n2b', 'norm2.bias'), ('ls1', 'ls1'), ('ls2', 'ls2')]
        for i in range(vis.num_hidden_layers):
            for nm, k in names:
                out[f'vision_model.encoder.layers.{i}.{k}'] = v[f'v{i}.{nm}'].clone()
        e = 'vision_model.embeddings.'
        kk = 3 * vis.patch_size ** 2
        out[e + 'patch_embedding.weight'] = v['emb.pe.w'][:, :kk].reshape(vis.hidden_size, 3, vis.patch_size, vis.patch_size).clone()
        out[e + 'patch_embedding.bias'] = v['emb.pe.b'].clone()
        out[e + 'class_embedding'] = v['emb.cls'].reshape(1, 1, -1).clone()
        out[e + 'position_embedding'] = v['emb.pos'].reshape(1, vis.num_positions, -1).clone()
        return out
